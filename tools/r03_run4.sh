export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/r03_gputest_4.log 2>&1; tail -3 $O/r03_gputest_4.log
bash tools/gpu_policies.sh "X=0 --" "CWC_PACK=3 --" "CWC_WITNESS_SLOTS=1 --" "CWC_WITNESS_SLOTS=1 CWC_PACK=3 --" \
   "X=0 -- --config 3" "CWC_PACK=3 -- --config 3" "CWC_WITNESS_SLOTS=1 -- --config 3" "CWC_WITNESS_SLOTS=1 CWC_PACK=3 -- --config 3" \
   "X=0 -- --config 4" "CWC_PACK=3 -- --config 4" "CWC_WITNESS_SLOTS=1 -- --config 4" "CWC_WITNESS_SLOTS=1 CWC_PACK=3 -- --config 4" \
   "X=0 -- --batch-per-gpu 256" "CWC_NO_FUSE=1 -- --batch-per-gpu 256" "CWC_NO_FUSE=1 --" > $O/r03_pack_ab.log 2>&1; cat $O/r03_pack_ab.log
CWC_PROGRAM_CACHE=/tmp/cwc_cache CWC_DEBUG_CACHE=1 timeout 300 python tools/gpu_single_shot.py > $O/r03_single_shot_2.log 2>&1; tail -8 $O/r03_single_shot_2.log
CWC_PROGRAM_CACHE=/tmp/cwc_cache CWC_DEBUG_CACHE=1 timeout 300 python tools/gpu_single_shot.py > $O/r03_single_shot_3.log 2>&1; tail -6 $O/r03_single_shot_3.log
timeout 900 python bench.py --cpu-sample 256 > $O/r03_bench_4.json 2> $O/r03_bench_4.err; tail -3 $O/r03_bench_4.err; python tools/show_bench.py $O/r03_bench_4.json 2>/dev/null | head -12
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 > $O/r03_bench_dist1.json 2> $O/r03_bench_dist1.err; tail -3 $O/r03_bench_dist1.err; python -c "
import json; d=json.load(open('$O/r03_bench_dist1.json')); print({k:d.get(k) for k in ('value','rccl_ranks','per_rank_ms_per_step','n1_ms_per_step_same_run','efficiency_vs_n1')}); print(d.get('config4'))"
