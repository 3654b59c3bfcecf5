export TMPDIR=/tmp
D=/root/repo/circom-witnesscalc_amd
S4="CWC_LIB_PATH=$D/libcwc_step4.so"; B1="CWC_LIB_PATH=$D/libcwc_b1.so"
bash tools/gpu_policies.sh "X=0 --" "$B1 --" "$S4 --" "X=0 --" "$B1 --" "$S4 --" "X=0 -- --batch-per-gpu 256" "$B1 -- --batch-per-gpu 256" "$S4 -- --batch-per-gpu 256" "X=0 -- --config 4" "$B1 -- --config 4" "$S4 -- --config 4" "X=0 -- --config 3" "$B1 -- --config 3" "$S4 -- --config 3" "X=0 -- --batch-per-gpu 512" "$B1 -- --batch-per-gpu 512" "$S4 -- --batch-per-gpu 512" "X=0 --" "$B1 --" "$S4 --"
