export TMPDIR=/tmp
O=gpurun_out
PROBE_T=258 python tools/gpu_classprof.py > $O/r03_classprof_macro.log 2>&1
PROBE_B=256 PROBE_T=257 python tools/gpu_classprof.py >> $O/r03_classprof_macro.log 2>&1
grep -v "amdgpu.ids" $O/r03_classprof_macro.log
