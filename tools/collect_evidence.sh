#!/bin/bash
# Evidence of one code state on one MI355X box (run through gpurun from the repository root):
#   gpurun --timeout 2400 -- 'bash tools/collect_evidence.sh r01'
# then, back in the container:  python tools/profile_summary.py r01  and copy the logs named in profiles/README.md.
# Each rocprofv3 pass is its own command with the program directly behind `--`; counter passes carry no trace domains.
R=${1:-r01}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
BENCH="python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --extra-batch 0 --host-path 0"
python -m pytest tests -q -m gpu > $O/gputest_$R.log 2>&1; tail -1 $O/gputest_$R.log
python bench.py > $O/bench_$R.json 2> $O/bench_$R.err
rocprofv3 --kernel-trace --stats -d $O/prof_$R -o runc -- $BENCH > $O/prof_$R.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$R -o runc -- $BENCH > $O/pmc_fetch_$R.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$R -o runc -- $BENCH > $O/pmc_write_$R.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    -d $O/pmc_sq_$R -o runc -- $BENCH > $O/pmc_sq_$R.log 2>&1
PROBE_T=2,258,4 python tools/gpu_classprof.py > $O/classprof_$R.log 2>&1
PROBE_GRAPH=bigint PROBE_B=32 PROBE_T=1 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
python tools/gpu_sweep.py > $O/sweep_$R.log 2>&1
python tools/gpu_autopick.py > $O/autopick_$R.log 2>&1
python bench.py --graph sha256 --batch-per-gpu 4096 --cpu-sample 128 --extra-batch 0 > $O/bench_sha256_$R.json 2> $O/bench_sha256_$R.err
python tools/gpu_host_path.py > $O/hostpath_$R.log 2>&1
ls $O
