export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
cd tools/ubench; for v in r02 cxx blk sh64; do echo "== inv_bench_$v"; timeout 120 ./inv_bench_$v; done > ../../$O/r03_inv_bench.txt 2>&1; cd ../..
cat $O/r03_inv_bench.txt
timeout 900 python -m pytest tests -q -m gpu -x > $O/r03_gputest_1.log 2>&1; tail -3 $O/r03_gputest_1.log
timeout 600 python bench.py --cpu-sample 0 --extras 0 > $O/r03_bench_1.json 2> $O/r03_bench_1.err; python tools/show_bench.py $O/r03_bench_1.json 2>/dev/null | head -30
timeout 300 python tools/gpu_single_shot.py > $O/r03_single_shot_1.log 2>&1; tail -5 $O/r03_single_shot_1.log
PROBE_B=256 PROBE_T=0 timeout 300 python tools/gpu_onebatch.py > $O/r03_b256_1.log 2>&1; tail -3 $O/r03_b256_1.log
