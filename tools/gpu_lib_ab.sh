# Same-box A/B of two builds of the library through bench.py: the in-tree one against circom-witnesscalc_amd/libcwc_base.so
# (a copy of the build to compare with, e.g. `git stash; make -C circom-witnesscalc_amd/csrc; cp ...so libcwc_base.so; git stash pop; make`):
#   gpurun --timeout 1800 -- 'bash tools/gpu_lib_ab.sh'
# Box-to-box differences are ~1 %: only same-box pairs are trusted.
export TMPDIR=/tmp
O=gpurun_out
B=CWC_LIB_PATH=/root/repo/circom-witnesscalc_amd/libcwc_base.so
bash tools/gpu_policies.sh "X=0 --" "$B --" "X=0 -- --batch-per-gpu 256" "$B -- --batch-per-gpu 256" "X=0 -- --batch-per-gpu 512" "$B -- --batch-per-gpu 512" "X=0 -- --config 3" "$B -- --config 3" "X=0 -- --config 4" "$B -- --config 4" "X=0 --" "$B --" > $O/lib_ab.log 2>&1; cat $O/lib_ab.log
python bench.py --config 5 --cpu-sample 0 2>/dev/null | python tools/show_bench.py /dev/stdin | head -1
CWC_LIB_PATH=/root/repo/circom-witnesscalc_amd/libcwc_base.so python bench.py --config 5 --cpu-sample 0 2>/dev/null | python tools/show_bench.py /dev/stdin | head -1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fuzz or streams or authv2 or scan or soak or gadgets or edge" 2>&1 | tail -2
