export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -q -m gpu -x > $O/r03_gputest_11.log 2>&1; tail -2 $O/r03_gputest_11.log
rm -rf /tmp/cc11; CWC_PROGRAM_CACHE=/tmp/cc11 CWC_DEBUG_CACHE=1 python tools/gpu_single_shot.py > $O/r03_single_shot_11.log 2>&1
echo "---- second process, program cache warm" >> $O/r03_single_shot_11.log
CWC_PROGRAM_CACHE=/tmp/cc11 CWC_DEBUG_CACHE=1 SHOTS=6 python tools/gpu_single_shot.py >> $O/r03_single_shot_11.log 2>&1
cat $O/r03_single_shot_11.log
python bench.py --cpu-sample 0 --extras 0 2>/dev/null | python tools/show_bench.py
