R=r05
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
want=$(make -s -C circom-witnesscalc_amd/csrc print-ksrc-hash); have=$(python -c "import cwc_import; print(cwc_import.load().kernel_source_hash())" 2>/dev/null)
if [ "$want" != "$have" ]; then echo "STALE LIBRARY"; exit 1; fi
python -m pytest tests -q -m gpu > $O/gputest_$R.log 2>&1; tail -1 $O/gputest_$R.log
python bench.py > $O/bench_$R.json 2> $O/bench_$R.err
python bench.py --config 5 --cpu-sample 32 > $O/bench_config5_$R.json 2> $O/bench_config5_$R.err
python bench.py --config 5 --config5-graph bigint --cpu-sample 32 > $O/bench_config5_bigint_$R.json 2> $O/bench_config5_bigint_$R.err
RSA_MULS=310 PROBE_T=0,1,2 RSA_CHECK=32 python tools/gpu_rsa.py > $O/config5_rsa_$R.log 2>&1
BIGINT_ROUNDS=4000 PROBE_T=0 python tools/gpu_bigint.py > $O/config5_$R.log 2>&1
PROBE_T=258,2,4 python tools/gpu_classprof.py > $O/classprof_$R.log 2>&1
PROBE_B=256 PROBE_T=257 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
PROBE_GRAPH=bigint PROBE_B=32 PROBE_T=1,2 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
PROBE_GRAPH=rsa RSA_MULS=4 PROBE_B=32 PROBE_T=1,2 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
SOAK_KINDS=limb SOAK_SEEDS=3000 SOAK_BASE=20261104 python tools/gpu_soak.py > $O/soak_scan_$R.log 2>&1
SOAK_KINDS=limb SOAK_WIDE_SHARE=1.0 SOAK_SEEDS=3000 SOAK_BASE=20261105 python tools/gpu_soak.py > $O/soak_wide_$R.log 2>&1
SOAK_SEEDS=20000 SOAK_BASE=20261003 python tools/gpu_soak.py > $O/soak_$R.log 2>&1
python tools/gpu_robustness.py > $O/robustness_$R.log 2>&1
python tools/gpu_autopick.py > $O/autopick_$R.log 2>&1
tail -1 $O/soak_$R.log $O/soak_wide_$R.log $O/soak_scan_$R.log
