export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/r03_gputest_6.log 2>&1; tail -3 $O/r03_gputest_6.log
bash tools/gpu_policies.sh "X=0 --" "X=0 --" "X=0 -- --config 3" "X=0 -- --config 4" "X=0 -- --batch-per-gpu 256" > $O/r03_ab6.log 2>&1; cat $O/r03_ab6.log
CWC_FUSE=1001 SOAK_SEEDS=300 SOAK_BASE=991 timeout 900 python tools/gpu_soak.py > $O/r03_soak_fused.log 2>&1; tail -1 $O/r03_soak_fused.log
timeout 900 python bench.py --cpu-sample 0 > $O/r03_bench_6.json 2> $O/r03_bench_6.err; tail -3 $O/r03_bench_6.err; python -c "
import json; d=json.load(open('$O/r03_bench_6.json')); print(d['value']); print(d.get('e2e_json_to_wtns')); print(d.get('json_front_end'))"
