export TMPDIR=/tmp
S4=/root/repo/circom-witnesscalc_amd/libcwc_step4.so
for lib in new step4 new step4; do
  if [ $lib = step4 ]; then export CWC_LIB_PATH=$S4; else unset CWC_LIB_PATH; fi
  echo "== $lib"
  BIGINT_ROUNDS=1000 PROBE_T=1,2 python tools/gpu_bigint.py 2>&1 | grep "bigint-class"
done
unset CWC_LIB_PATH
bash tools/gpu_policies.sh "X=0 -- --config 3" "CWC_LIB_PATH=$S4 -- --config 3" "X=0 -- --config 3" "CWC_LIB_PATH=$S4 -- --config 3" "X=0 -- --config 3 --batch-per-gpu 1024" "CWC_LIB_PATH=$S4 -- --config 3 --batch-per-gpu 1024"
