export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
CWC_FUSE=1001 SOAK_SEEDS=400 SOAK_BASE=777 timeout 900 python tools/gpu_soak.py > $O/r03_soak_fused.log 2>&1; tail -2 $O/r03_soak_fused.log
CWC_FUSE=1 SOAK_SEEDS=200 SOAK_BASE=778 timeout 600 python tools/gpu_soak.py > $O/r03_soak_fused1.log 2>&1; tail -1 $O/r03_soak_fused1.log
SOAK_SEEDS=200 SOAK_BASE=779 timeout 600 python tools/gpu_soak.py > $O/r03_soak_auto.log 2>&1; tail -1 $O/r03_soak_auto.log
timeout 900 python -m pytest tests -q -m gpu -x > $O/r03_gputest_2.log 2>&1; tail -3 $O/r03_gputest_2.log
CWC_FUSE=11 PROBE_B=256 PROBE_T=4353 timeout 300 python tools/gpu_classprof.py > $O/r03_classprof_fused.log 2>&1; cat $O/r03_classprof_fused.log
CWC_NO_FUSE=1 PROBE_B=256 PROBE_T=4353 timeout 300 python tools/gpu_classprof.py > $O/r03_classprof_nofuse.log 2>&1; cat $O/r03_classprof_nofuse.log
timeout 600 python bench.py --cpu-sample 0 --extras 0 > $O/r03_bench_2.json 2> $O/r03_bench_2.err; python tools/show_bench.py $O/r03_bench_2.json 2>/dev/null | head -3
