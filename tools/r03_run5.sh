export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$PWD/circom-witnesscalc_amd/libcwc_nomulf.so
bash tools/gpu_policies.sh "X=0 --" "CWC_LIB_PATH=$L --" "CWC_MODEL_CYCLES=3:73500 --" "CWC_MODEL_CYCLES=3:73500 CWC_LIB_PATH=$L --" "X=0 --" "CWC_LIB_PATH=$L --" \
   "CWC_MODEL_CYCLES=3:65000 --" "CWC_MODEL_CYCLES=3:45000 --" \
   "X=0 -- --batch-per-gpu 256" "CWC_LIB_PATH=$L -- --batch-per-gpu 256" "CWC_MODEL_CYCLES=3:73500 -- --batch-per-gpu 256" > $O/r03_regress_ab.log 2>&1; cat $O/r03_regress_ab.log
timeout 600 python -m pytest tests -q -m gpu -x -k "streaming or quick_first" > $O/r03_gputest_5.log 2>&1; tail -5 $O/r03_gputest_5.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 > $O/r03_bench_dist1.json 2> $O/r03_bench_dist1.err; grep -A8 Traceback $O/r03_bench_dist1.err | head -20; python -c "
import json; d=json.load(open('$O/r03_bench_dist1.json')); print({k:d.get(k) for k in ('value','rccl_ranks','per_rank_ms_per_step','n1_ms_per_step_same_run','efficiency_vs_n1')}); print(d.get('e2e_json_to_wtns')); print(d.get('json_front_end'))"
timeout 900 python bench.py --cpu-sample 0 > $O/r03_bench_5.json 2> $O/r03_bench_5.err; tail -3 $O/r03_bench_5.err; python -c "
import json; d=json.load(open('$O/r03_bench_5.json')); print(d['value']); print(d.get('e2e_json_to_wtns')); print(d.get('json_front_end'))"
