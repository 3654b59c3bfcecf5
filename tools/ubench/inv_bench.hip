// Microbenchmark: cycles per fr_inv for one wave (gfx950).  Built in several variants by tools/ubench/build_inv_bench.sh:
// the round-2 header (reference point), the round-3 loop with the C++ update, with the generated update block, and with
// the block's 64-bit-shift variant.  Cases: every lane the same operand / distinct operands in all 64 lanes / four active
// lanes (the divider wave's situation: a division request of the authV2-class ladder has 2-4 active lanes, the rest 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#ifndef CWC_FR_HEADER
#define CWC_FR_HEADER "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
#endif
#include CWC_FR_HEADER
using namespace cwc;
__global__ void k(uint64_t* out, int iters, int mode) {
    Fr a = fr_r2();
    a.v[0] ^= mode ? threadIdx.x * 2654435761u : 12345u;
    a.v[3] ^= mode ? threadIdx.x * 40503u : 777u;
    if (mode == 2 && threadIdx.x >= 4) a = fr_zero();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a = fr_inv(a);
        if (mode != 2 || threadIdx.x < 4) { a.v[0] ^= 1u; a.v[7] &= 0x0fffffffu; }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a.v[1] == 0x12345) out[1] = a.v[2];
    // parity: x * inv(x) == 1 (Montgomery) on this lane's last value
    Fr x = fr_r2();
    x.v[0] ^= threadIdx.x * 2654435761u;
    x.v[2] ^= (uint32_t)iters * 97u;
    const Fr p = fr_mul(x, fr_inv(x)), one = fr_one();
    uint32_t bad = 0;
    for (int i = 0; i < 8; ++i) bad |= p.v[i] ^ one.v[i];
    if (bad) atomicAdd((unsigned long long*)&out[2], 1ull);
}
int main() {
    uint64_t* d; hipMalloc(&d, 64); uint64_t h[3];
    const char* names[3] = {"identical operands", "distinct operands in 64 lanes", "four active lanes"};
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, 64);
        k<<<1, 64>>>(d, 50, mode); hipDeviceSynchronize();
        hipMemset(d, 0, 64);
        k<<<1, 64>>>(d, 200, mode); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-32s %7.0f cycles per fr_inv (one wave)  parity mismatches %llu\n", names[mode], h[0] / 200.0, (unsigned long long)h[2]);
    }
}
