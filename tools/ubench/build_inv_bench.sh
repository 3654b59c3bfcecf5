#!/bin/bash
# builds the fr_inv microbenchmark variants next to this script (gfx950); run them on the GPU box one after the other
set -e
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
F="-O3 -std=c++17 --offload-arch=gfx950"
if git -C ../.. show 446aec2:circom-witnesscalc_amd/csrc/fr_gfx950.hpp > /tmp/fr_r02.hpp 2>/dev/null; then
  cp ../../circom-witnesscalc_amd/csrc/*.inc /tmp/
  $H $F -DCWC_FR_HEADER='"/tmp/fr_r02.hpp"' -o inv_bench_r02 inv_bench.hip
fi
$H $F -DCWC_SGCD_CXX_UPDATE -o inv_bench_cxx inv_bench.hip
$H $F -o inv_bench_blk inv_bench.hip
$H $F -DCWC_SGCD_UPDATE_INC='"../../tools/ubench/sgcd_update_sh32.inc"' -o inv_bench_sh32 inv_bench.hip
