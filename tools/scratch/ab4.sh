export TMPDIR=/tmp
BASE=/root/repo/circom-witnesscalc_amd/libcwc_base.so
for lib in new base new; do
  if [ $lib = base ]; then export CWC_LIB_PATH=$BASE; else unset CWC_LIB_PATH; fi
  echo "== $lib"
  BIGINT_ROUNDS=1000 PROBE_T=1 python tools/gpu_bigint.py 2>&1 | grep "bigint-class"
  CWC_FORCE_MODE3=1 BIGINT_ROUNDS=1000 PROBE_T=1 python tools/gpu_bigint.py 2>&1 | grep "bigint-class" | sed 's/^/forced MODE3: /'
  RSA_MULS=34 PROBE_T=1 RSA_CHECK=4 python tools/gpu_rsa.py 2>&1 | grep "rsa-class" | cut -c1-60,200-330
done
unset CWC_LIB_PATH
bash tools/gpu_policies.sh "X=0 --" "CWC_LIB_PATH=$BASE --" "X=0 -- --config 3" "CWC_LIB_PATH=$BASE -- --config 3" "X=0 --" "CWC_LIB_PATH=$BASE --" "X=0 -- --config 3" "CWC_LIB_PATH=$BASE -- --config 3"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
SOAK_KINDS=limb SOAK_SEEDS=2000 SOAK_BASE=779 python tools/gpu_soak.py 2>&1 | tail -1
