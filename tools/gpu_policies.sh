mkdir -p gpurun_out/r2
B="python bench.py --steps 5 --warmup 1 --cpu-sample 0 --extra-batch 0 --host-path 0"
for v in "auto::" "nocoop:CWC_NO_COOP_MUL=1:" "naive:CWC_COOP_FILL=32:CWC_COOP_SLACK=100000000" "f12s0:CWC_COOP_FILL=12:CWC_COOP_SLACK=0" "f16s94:CWC_COOP_FILL=16:CWC_COOP_SLACK=94" "f20s94:CWC_COOP_FILL=20:CWC_COOP_SLACK=94" "f16s0:CWC_COOP_FILL=16:CWC_COOP_SLACK=0" "f24s0:CWC_COOP_FILL=24:CWC_COOP_SLACK=0"; do
  name=${v%%:*}; rest=${v#*:}; e1=${rest%%:*}; e2=${rest#*:}
  out=$(env $e1 $e2 $B 2>/dev/null)
  echo "$name $(echo $out | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f wit/s interp %.2f ms bundles %d" % (d["value"], d["roofline"]["avg_launch_ms"], d["config"]["bundles"]))')"
done | tee gpurun_out/r2/policies1.txt
