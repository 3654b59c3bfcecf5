export TMPDIR=/tmp
BASE=/root/repo/circom-witnesscalc_amd/libcwc_base.so
for lib in new base new base; do
  if [ $lib = base ]; then export CWC_LIB_PATH=$BASE; else unset CWC_LIB_PATH; fi
  echo "== $lib"; python tools/scratch/limbdiv.py 2>&1 | grep "limb graph"
done
unset CWC_LIB_PATH
B="CWC_LIB_PATH=$BASE"
bash tools/gpu_policies.sh "X=0 --" "$B --" "X=0 --" "$B --" "X=0 -- --batch-per-gpu 256" "$B -- --batch-per-gpu 256" "X=0 -- --batch-per-gpu 512" "$B -- --batch-per-gpu 512" "X=0 -- --batch-per-gpu 2048" "$B -- --batch-per-gpu 2048" "X=0 --" "$B --" "X=0 -- --config 4" "$B -- --config 4"
python tools/gpu_single_shot.py 2>&1 | tail -3
CWC_LIB_PATH=$BASE python tools/gpu_single_shot.py 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
SOAK_SEEDS=3000 SOAK_BASE=780 python tools/gpu_soak.py 2>&1 | tail -1
