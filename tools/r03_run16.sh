export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r03_gputest_16.log 2>&1; tail -2 $O/r03_gputest_16.log
rm -rf /tmp/cc16; CWC_PROGRAM_CACHE=/tmp/cc16 CWC_DEBUG_CACHE=1 python tools/gpu_single_shot.py > $O/r03_single_shot_16.log 2>&1
echo "---- second process, program cache warm" >> $O/r03_single_shot_16.log
CWC_PROGRAM_CACHE=/tmp/cc16 CWC_DEBUG_CACHE=1 SHOTS=6 python tools/gpu_single_shot.py >> $O/r03_single_shot_16.log 2>&1
echo "---- third process, no cache" >> $O/r03_single_shot_16.log
CWC_PROGRAM_CACHE=0 SHOTS=4 python tools/gpu_single_shot.py >> $O/r03_single_shot_16.log 2>&1
grep -v amdgpu.ids $O/r03_single_shot_16.log
