export TMPDIR=/tmp
O=gpurun_out
timeout 900 python tools/gpu_e2e.py > $O/r03_e2e_ab2.log 2>&1; cat $O/r03_e2e_ab2.log
timeout 600 python -m pytest tests -q -m gpu -x -k "optimiser_on_and_off or streaming" > $O/r03_gputest_8.log 2>&1; tail -3 $O/r03_gputest_8.log
