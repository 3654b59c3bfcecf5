"""Pretty-print the sub-records of a bench.py JSON line."""
import json
import sys
d = json.load(open(sys.argv[1]))
print("value %.0f %s  ms/step %.2f  n_gpus %d  roofline.frac %.3f" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["roofline"]["frac"]))
print("roofline:", {k: v for k, v in d["roofline"].items() if k not in ("note", "compute")})
print("compute:", d["roofline"].get("compute"))
for k in ("cpu_baseline", "config3", "config4_per_gpu", "config4", "json_front_end", "single_shot", "e2e_json_to_wtns", "pcie_inclusive", "rccl_ranks", "efficiency_vs_n1", "extras_seconds"):
    print(k + ":", d.get(k))
