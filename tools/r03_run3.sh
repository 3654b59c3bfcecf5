export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
CWC_FUSE=1001 SOAK_SEEDS=600 SOAK_BASE=777 timeout 900 python tools/gpu_soak.py > $O/r03_soak_fused.log 2>&1; tail -1 $O/r03_soak_fused.log
CWC_FUSE=1 SOAK_SEEDS=200 SOAK_BASE=778 timeout 600 python tools/gpu_soak.py > $O/r03_soak_fused1.log 2>&1; tail -1 $O/r03_soak_fused1.log
SOAK_SEEDS=300 SOAK_BASE=779 timeout 600 python tools/gpu_soak.py > $O/r03_soak_auto.log 2>&1; tail -1 $O/r03_soak_auto.log
timeout 900 python -m pytest tests -q -m gpu -x > $O/r03_gputest_3.log 2>&1; tail -3 $O/r03_gputest_3.log
timeout 900 python bench.py --cpu-sample 256 > $O/r03_bench_3.json 2> $O/r03_bench_3.err; tail -3 $O/r03_bench_3.err; python tools/show_bench.py $O/r03_bench_3.json 2>/dev/null | head -12
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 > $O/r03_bench_dist1.json 2> $O/r03_bench_dist1.err; tail -3 $O/r03_bench_dist1.err; python -c "
import json; d=json.load(open('$O/r03_bench_dist1.json')); print({k:d.get(k) for k in ('value','rccl_ranks','per_rank_ms_per_step','n1_ms_per_step_same_run','efficiency_vs_n1')}); print(d.get('config4'))"
