export TMPDIR=/tmp
O=gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x > $O/r03_gputest_17.log 2>&1; tail -3 $O/r03_gputest_17.log
bash tools/gpu_policies.sh "X=0 --" "X=0 -- --batch-per-gpu 256" "X=0 -- --batch-per-gpu 512" "X=0 -- --config 4" "X=0 -- --config 3" "X=0 --" > $O/r03_compact_ab.log 2>&1; cat $O/r03_compact_ab.log
CWC_PROGRAM_CACHE=0 SHOTS=6 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids
SOAK_SEEDS=3000 SOAK_BASE=20261006 timeout 1200 python tools/gpu_soak.py 2>&1 | tail -2
python bench.py --config 5 --cpu-sample 0 2>/dev/null | python tools/show_bench.py /dev/stdin | head -3
