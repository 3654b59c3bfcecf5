export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python tools/gpu_e2e.py > $O/r03_e2e_ab.log 2>&1; cat $O/r03_e2e_ab.log
timeout 300 python -m pytest tests -q -m gpu -x -k "optimiser_on_and_off" > $O/r03_gputest_7.log 2>&1; tail -3 $O/r03_gputest_7.log
timeout 600 python tools/gpu_streams.py > $O/r03_streams.log 2>&1; cat $O/r03_streams.log
