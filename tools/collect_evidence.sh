#!/bin/bash
# Evidence of one code state on one MI355X box (run through gpurun from the repository root):
#   gpurun --timeout 3600 -- 'bash tools/collect_evidence.sh r06'
# then, back in the container:  python tools/profile_summary.py r05  and  bash tools/copy_evidence.sh r05  (the logs named in profiles/README.md).
# Each rocprofv3 pass is its own command with the program directly behind `--`; counter passes carry no trace domains.
R=${1:-r06}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
# the library's device code must come from the kernel sources in the tree (a stale kernels.o once produced an hour of evidence)
want=$(make -s -C circom-witnesscalc_amd/csrc print-ksrc-hash); have=$(python -c "import cwc_import; print(cwc_import.load().kernel_source_hash())" 2>/dev/null)
if [ "$want" != "$have" ]; then echo "libcircom_witnesscalc_amd.so was built from other kernel sources ($have, tree: $want): run make first" | tee $O/STALE_LIBRARY_$R.txt; exit 1; fi
echo "$have" > $O/ksrc_$R.txt   # (tools/profile_summary.py stamps the PMC summary with it: bench.py --pmc-selfcheck compares)
BENCH="python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --extras 0"
python -m pytest tests -q -m gpu > $O/gputest_$R.log 2>&1; tail -1 $O/gputest_$R.log
python bench.py > $O/bench_$R.json 2> $O/bench_$R.err
# rocprofv3 passes: each its own command with the program directly behind `--`; counter passes carry no trace domains.  The rocpd databases are
# summarised HERE (tools/profile_summary.py -> $O/summary_$R/, what gets committed under profiles/) and deleted: four configurations of them do not
# fit the 64 MiB a lease copies back.
S=$O/summary_$R
mkdir -p $S
passes() {  # $1 = tag suffix ("" or _configN), $2.. = the bench command
  local sfx=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/prof_${R}$sfx -o runc -- "$@" > $O/prof_${R}$sfx.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_${R}$sfx -o runc -- "$@" > $O/pmc_fetch_${R}$sfx.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_${R}$sfx -o runc -- "$@" > $O/pmc_write_${R}$sfx.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      -d $O/pmc_sq_${R}$sfx -o runc -- "$@" > $O/pmc_sq_${R}$sfx.log 2>&1
}
passes "" $BENCH
python tools/profile_summary.py $R --graph authv2 --batch 1024 --out $S > $O/profile_summary_$R.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_icache_$R -o runc -- $BENCH > $O/pmc_icache_$R.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_INSTS_SMEM -d $O/pmc_dcache_$R -o runc -- $BENCH > $O/pmc_dcache_$R.log 2>&1
python tools/pmc_cache_summary.py $R > $O/cache_counters_$R.log 2>&1
rm -rf $O/prof_$R $O/pmc_fetch_$R $O/pmc_write_$R $O/pmc_sq_$R $O/pmc_icache_$R $O/pmc_dcache_$R
# the other BASELINE configurations as the timed metric (config 5 = the RSA-class graph, 32 sets)
for C in 3 4 5; do
  passes _config$C python3 bench.py --config $C --steps 5 --warmup 1 --cpu-sample 0 --extras 0
  case $C in 3) GK="--graph sha256 --batch 4096";; 4) GK="--graph authv2 --batch 8192";; 5) GK="--graph rsa --batch 32";; esac
  python tools/profile_summary.py $R --tag config$C $GK --cmd "python3 bench.py --config $C --steps 5 --warmup 1 --cpu-sample 0 --extras 0" --out $S > $O/profile_summary_${R}_config$C.log 2>&1
  rm -rf $O/prof_${R}_config$C $O/pmc_fetch_${R}_config$C $O/pmc_write_${R}_config$C $O/pmc_sq_${R}_config$C
done
PROBE_T=258,2,4 python tools/gpu_classprof.py > $O/classprof_$R.log 2>&1
PROBE_B=256 PROBE_T=257 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
PROBE_GRAPH=bigint PROBE_B=32 PROBE_T=1,2 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
PROBE_GRAPH=rsa RSA_MULS=4 PROBE_B=32 PROBE_T=1,2 python tools/gpu_classprof.py >> $O/classprof_$R.log 2>&1
python tools/gpu_sweep.py > $O/sweep_$R.log 2>&1
python tools/gpu_autopick.py > $O/autopick_$R.log 2>&1
python tools/gpu_robustness.py > $O/robustness_$R.log 2>&1
# (the micro-benchmarks are built on the box, into /tmp: built binaries no longer travel with the repository)
H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950"
(cd tools/ubench && $H -o /tmp/coop_mul coop_mul.hip && /tmp/coop_mul) > $O/coop_mul_$R.log 2>&1
(cd tools/ubench && $H -I../../circom-witnesscalc_amd/csrc -o /tmp/scan_par_test scan_par_test.hip && timeout 120 /tmp/scan_par_test) > $O/scan_par_$R.log 2>&1
(cd tools/ubench && $H -o /tmp/inv_bench_blk inv_bench.hip && $H -DCWC_SGCD_CXX_UPDATE -o /tmp/inv_bench_cxx inv_bench.hip && for v in cxx blk; do echo "== inv_bench_$v"; timeout 120 /tmp/inv_bench_$v; done) > $O/inv_bench_$R.log 2>&1
python tools/gpu_e2e.py > $O/e2e_$R.log 2>&1
CWC_FUSE=1001 SOAK_SEEDS=2000 SOAK_BASE=20261004 python tools/gpu_soak.py > $O/soak_fused_$R.log 2>&1
SOAK_KINDS=limb SOAK_SEEDS=3000 SOAK_BASE=20261104 python tools/gpu_soak.py > $O/soak_scan_$R.log 2>&1
SOAK_KINDS=limb SOAK_WIDE_SHARE=1.0 SOAK_SEEDS=3000 SOAK_BASE=20261105 python tools/gpu_soak.py > $O/soak_wide_$R.log 2>&1
CWC_FUSE=11 PROBE_B=256 PROBE_T=4353 python tools/gpu_classprof.py > $O/classprof_fused_$R.log 2>&1
python bench.py --config 3 --cpu-sample 128 > $O/bench_config3_$R.json 2> $O/bench_config3_$R.err
python bench.py --config 4 --cpu-sample 0 > $O/bench_config4_$R.json 2> $O/bench_config4_$R.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 \
    > $O/bench_dist1_$R.json 2> $O/bench_dist1_$R.err
python tools/gpu_host_path.py > $O/hostpath_$R.log 2>&1
python tools/gpu_calibrate.py > $O/calibration_$R.log 2>&1
rm -rf /tmp/cwc_cache_$R; CWC_PROGRAM_CACHE=/tmp/cwc_cache_$R CWC_DEBUG_CACHE=1 python tools/gpu_single_shot.py > $O/single_shot_$R.log 2>&1
echo "---- second process, program cache warm" >> $O/single_shot_$R.log
CWC_PROGRAM_CACHE=/tmp/cwc_cache_$R CWC_DEBUG_CACHE=1 SHOTS=6 python tools/gpu_single_shot.py >> $O/single_shot_$R.log 2>&1
echo "---- third process, no cache, where the first call's time goes" >> $O/single_shot_$R.log
CWC_PROGRAM_CACHE=0 CWC_DEBUG_SINGLE=1 SHOTS=2 python tools/gpu_single_shot.py >> $O/single_shot_$R.log 2>&1
BIGINT_ROUNDS=4000 PROBE_T=0 python tools/gpu_bigint.py > $O/config5_$R.log 2>&1
RSA_MULS=310 PROBE_T=0,1,2 RSA_CHECK=32 python tools/gpu_rsa.py > $O/config5_rsa_$R.log 2>&1
python bench.py --config 5 --cpu-sample 32 > $O/bench_config5_$R.json 2> $O/bench_config5_$R.err
python bench.py --config 5 --config5-graph bigint --cpu-sample 32 > $O/bench_config5_bigint_$R.json 2> $O/bench_config5_bigint_$R.err
python tools/gpu_streams.py > $O/streams_$R.log 2>&1
SOAK_SEEDS=20000 SOAK_BASE=20261003 python tools/gpu_soak.py > $O/soak_$R.log 2>&1
bash tools/gpu_policies.sh "X=0 --" "CWC_NO_COOP_MUL=1 --" "CWC_COOP_FILL=32 CWC_COOP_SLACK=4000000000 --" "CWC_COOP_FILL=16 CWC_COOP_SLACK=2 --" \
    "CWC_SCHED_MUL_COST=47 CWC_SCHED_LIN_COST=12 --" "CWC_SCHED_MUL_COST=26 CWC_SCHED_LIN_COST=14 --" "CWC_SCHED_MUL_COST=30 CWC_SCHED_LIN_COST=24 --" \
    "CWC_PACK=2 --" "CWC_PACK=2 -- --config 4" "X=0 -- --config 4" "CWC_WITNESS_SLOTS=1 -- --config 4" "X=0 -- --batch-per-gpu 256" "CWC_NO_COOP_MUL=1 -- --batch-per-gpu 256" \
    "CWC_MODEL_CYCLES=3:73500 --" "CWC_NO_FUSE=1 -- --batch-per-gpu 256" > $O/policies_$R.log 2>&1
du -sh $O; for f in $O/*_$R*.log $O/*_$R*.err; do echo "== $f: $(tail -1 $f | cut -c1-200)"; done
