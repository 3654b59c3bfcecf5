export TMPDIR=/tmp
O=gpurun_out
rm -rf /tmp/cc14; CWC_DEBUG_COMPILE_TIMES=1 CWC_DEBUG_SINGLE=1 CWC_PROGRAM_CACHE=/tmp/cc14 CWC_DEBUG_CACHE=1 SHOTS=10 python tools/gpu_single_shot.py > $O/r03_single_shot_14.log 2>&1
echo "---- second process" >> $O/r03_single_shot_14.log
CWC_DEBUG_SINGLE=1 CWC_PROGRAM_CACHE=/tmp/cc14 CWC_DEBUG_CACHE=1 SHOTS=3 python tools/gpu_single_shot.py >> $O/r03_single_shot_14.log 2>&1
grep -v "amdgpu.ids" $O/r03_single_shot_14.log | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "macro or single_shot" 2>&1 | tail -3
