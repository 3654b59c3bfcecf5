export TMPDIR=/tmp
O=gpurun_out
bash tools/gpu_policies.sh "X=0 --" "CWC_FULL_RECORDS=1 --" "X=0 -- --batch-per-gpu 256" "CWC_FULL_RECORDS=1 -- --batch-per-gpu 256" "X=0 --" "CWC_FULL_RECORDS=1 --" > $O/r03_compact_ab.log 2>&1; cat $O/r03_compact_ab.log
