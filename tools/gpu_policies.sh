# A/B of scheduler / kernel knobs: each argument = one forced setting ("ENV=.. ENV=.. -- bench args") through bench.py
mkdir -p gpurun_out/r2
for v in "$@"; do
  envs=${v%%--*}; args=${v#*--}; [ "$args" = "$v" ] && args=""
  out=$(env $envs python bench.py --steps 5 --warmup 1 --cpu-sample 0 --extras 0 $args 2>/dev/null)
  echo "$v : $(echo $out | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f wit/s interp %.2f ms pack %.2f ms bundles %d T=%d div=%d" % (d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["pack_kernel_avg_ms"], d["config"]["bundles"], d["config"]["tile_width"], d["config"]["interpreter_waves_per_divider_wave"]))')"
done
