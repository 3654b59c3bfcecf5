# Same-box A/B of kernel builds that differ in code-generation flags only: circom-witnesscalc_amd/libcwc_var_<name>.so against the in-tree library.
export TMPDIR=/tmp
ARGS=()
for rep in 1 2; do
  ARGS+=("X=0 --")
  for f in circom-witnesscalc_amd/libcwc_var_*.so; do ARGS+=("CWC_LIB_PATH=/root/repo/$f --"); done
done
for c in "--config 4" "--config 5" "--config 5 --config5-graph bigint" "--batch-per-gpu 256" "--config 3"; do
  ARGS+=("X=0 -- $c")
  for f in circom-witnesscalc_amd/libcwc_var_*.so; do ARGS+=("CWC_LIB_PATH=/root/repo/$f -- $c"); done
done
bash tools/gpu_policies.sh "${ARGS[@]}"
