export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r03_gputest_12.log 2>&1; tail -3 $O/r03_gputest_12.log
CWC_MACRO=1 SOAK_SEEDS=1500 SOAK_BASE=20261005 timeout 900 python tools/gpu_soak.py > $O/r03_soak_macro.log 2>&1; tail -3 $O/r03_soak_macro.log
bash tools/gpu_policies.sh "X=0 --" "CWC_NO_MACRO=1 --" "X=0 -- --batch-per-gpu 256" "CWC_NO_MACRO=1 -- --batch-per-gpu 256" "X=0 -- --batch-per-gpu 512" "CWC_NO_MACRO=1 -- --batch-per-gpu 512" "X=0 --" "CWC_NO_MACRO=1 --" > $O/r03_macro_ab.log 2>&1; cat $O/r03_macro_ab.log
PROBE_T=258 python tools/gpu_classprof.py > $O/r03_classprof_macro.log 2>&1
PROBE_B=256 PROBE_T=257 python tools/gpu_classprof.py >> $O/r03_classprof_macro.log 2>&1
cat $O/r03_classprof_macro.log
CWC_PROGRAM_CACHE=0 SHOTS=10 python tools/gpu_single_shot.py
