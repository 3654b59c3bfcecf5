#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the logs of tools/collect_evidence.sh under their committed names
R=${1:-r06}
O=gpurun_out
P=profiles
cp $O/bench_$R.json $P/${R}_bench_line.json
cp $O/bench_config3_$R.json $P/${R}_bench_line_config3.json
cp $O/bench_config4_$R.json $P/${R}_bench_line_config4.json
cp $O/bench_config5_$R.json $P/${R}_bench_line_config5.json
cp $O/bench_dist1_$R.json $P/${R}_bench_line_rccl_1rank.json
cp $O/classprof_$R.log $P/${R}_class_profile.txt
cp $O/classprof_fused_$R.log $P/${R}_class_profile_fused.txt
cp $O/sweep_$R.log $P/${R}_sweep_batch_tile.txt
cp $O/autopick_$R.log $P/${R}_autopick.txt
cp $O/robustness_$R.log $P/${R}_robustness.txt
cp $O/coop_mul_$R.log $P/${R}_ubench_coop_mul.txt
cp $O/inv_bench_$R.log $P/${R}_inv_bench.txt
cp $O/scan_par_$R.log $P/${R}_scan_par_test.txt
cp $O/e2e_$R.log $P/${R}_e2e_ab.txt
cp $O/hostpath_$R.log $P/${R}_host_path.txt
cp $O/single_shot_$R.log $P/${R}_single_shot.txt
cp $O/config5_$R.log $P/${R}_config5_10m_nodes.txt
cp $O/config5_rsa_$R.log $P/${R}_config5_rsa_10m_nodes.txt
cp $O/bench_config5_bigint_$R.json $P/${R}_bench_line_config5_bigint.json
cp $O/soak_wide_$R.log $P/${R}_soak_wide.txt
cp $O/streams_$R.log $P/${R}_streams_ab.txt
cp $O/soak_$R.log $P/${R}_soak.txt
cp $O/soak_fused_$R.log $P/${R}_soak_fused.txt
cp $O/soak_scan_$R.log $P/${R}_soak_scan.txt
cp $O/policies_$R.log $P/${R}_policies.txt
cp $O/calibration_$R.log $P/${R}_calibration.txt
cp $O/cache_counters_$R.log $P/${R}_cache_counters.txt
[ -f $O/soak_100k_$R.log ] && (head -3 $O/soak_100k_$R.log; echo ...; tail -4 $O/soak_100k_$R.log) > $P/${R}_soak_100k.txt  # (SOAK_SEEDS=100000 python tools/gpu_soak.py, run on its own)
tail -3 $O/gputest_$R.log > $P/${R}_gputest_tail.txt
ls -la $P | grep ${R}_
