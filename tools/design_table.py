"""Prints DESIGN.md section 5's table of current numbers from the round's committed evidence (profiles/rNN_*):
    python tools/design_table.py r06
Every row names the file its number comes from."""
import json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def load(name):
    try:
        return json.load(open(os.path.join(P, "%s_%s" % (R, name))))
    except (OSError, ValueError):
        return None
rows = []
def row(what, value, src):
    rows.append("| %s | %s | `%s_%s` |" % (what, value, R, src))
d = load("bench_line.json")
if d:
    rf = d["roofline"]
    row("**headline: BASELINE config 2** (authV2-class graph, 1024 sets, 1 GPU)", "**%.1f k witnesses/s**, %.2f ms per step (interpreter %.2f, pack %.2f), %.3g field-ops/s" % (d["value"] / 1e3, d["ms_per_step"], rf["avg_launch_ms"], rf["pack_kernel_avg_ms"], d["field_ops_per_sec"]), "bench_line.json")
    row("... what binds: lone-wave instruction issue (`roofline.bound = valu_issue`)", "%.3g modmul-equivalents/s of %.3g measured in the run = **%.3f**" % (rf["achieved"], rf["peak"], rf["frac"]), "bench_line.json")
    h = rf["hbm"]
    row("... SURVEY 8(d) algorithmic-byte model (`roofline.hbm`)", "%.0f GB/s of 8000 = %.3f (per step %.3f); counter traffic %.2f GB per launch = %.3f of peak, kernel hash matches: %s" % (h["achieved"], h["frac"], h["frac_step"], (rf["traffic"] or 0) / 1e9, h["hbm_measured_frac"] or 0, rf.get("traffic_kernel_hash_matches")), "bench_line.json, pmc_summary.json")
    c = rf["chain"]
    row("... floor of the execution model (`roofline.chain`)", "longest dependent chain %.2f M cycles = %.2f ms; achieved / floor **%.2f**" % (c["floor_cycles"] / 1e6, c["floor_ms"], c["achieved_over_floor"]), "bench_line.json")
    if "hbm_frac" in rf:  # the flat scalars the driver's record keeps (round 6): the same numbers under the keys a reader of BENCH_rNN.json finds
        row("... the same as flat keys of `roofline` (what the driver's record keeps)", "`hbm_frac` %.3f, `hbm_frac_step` %.3f, `hbm_measured_frac` %s, `traffic_over_algorithmic` %s, `chain_achieved_over_floor` %.3f, `valu_useful_issue_frac` %s, `sq_wait_any_frac` %s" % (
            rf["hbm_frac"], rf["hbm_frac_step"], "%.3f" % rf["hbm_measured_frac"] if rf.get("hbm_measured_frac") else "-", "%.3f" % rf["traffic_over_algorithmic"] if rf.get("traffic_over_algorithmic") else "-",
            rf["chain_achieved_over_floor"], "%.3f" % rf["valu_useful_issue_frac"] if rf.get("valu_useful_issue_frac") else "-", "%.3f" % rf["sq_wait_any_frac"] if rf.get("sq_wait_any_frac") else "-"), "bench_line.json")
        for k in ("config3", "config4_per_gpu", "config5", "config5_rsa"):
            if rf.get(k + "_value") is not None:
                row("... `roofline.%s_*`" % k, "value %.1f /s, %.2f ms per step, compute_frac %s, hbm_frac %s, counter traffic / algorithmic %s" % (
                    rf[k + "_value"], rf[k + "_ms_per_step"], "%.3f" % rf[k + "_compute_frac"] if rf.get(k + "_compute_frac") else "-", "%.3f" % rf[k + "_hbm_frac"] if rf.get(k + "_hbm_frac") else "-",
                    "%.3f" % rf[k + "_traffic_over_algorithmic"] if rf.get(k + "_traffic_over_algorithmic") else "-"), "bench_line.json")
    cb = d.get("cpu_baseline") or {}
    if cb:
        row("... CPU baseline (C port of `evaluate()`, same sets, byte-equal)", "%.0f witnesses/s on one pinned core, %.0f on all %d" % (cb["value"], (cb.get("all_cores") or {}).get("value", 0), (cb.get("all_cores") or {}).get("cores", 0)), "bench_line.json")
    for k, name in (("config3", "BASELINE config 3 (sha256_512, 4096 sets)"), ("config4_per_gpu", "BASELINE config 4, per-GPU share (authV2-class, 8192 sets)")):
        r = d.get(k) or {}
        if "value" in r:
            extra = ", every digest = hashlib: %s" % r.get("matches_hashlib") if k == "config3" else ""
            row(name, "%.1f k witnesses/s, %.2f ms per step (interpreter %.2f, pack %.2f); compute %.3f of the modmul ceiling%s" % (r["value"] / 1e3, r["ms_per_step"], r["interp_kernel_ms"], r["pack_kernel_ms"], (r.get("compute") or {}).get("frac") or 0, extra), "bench_line.json")
    for k, name in (("config5_rsa", "**BASELINE config 5, the named class** (zk-email RSA / long_div: 121-bit registers x 17, 10.0 M nodes, 32 sets)"), ("config5", "BASELINE config 5, first generator (64-bit limbs x 32, division by one limb, 10.5 M nodes, 32 sets)")):
        r = d.get(k) or {}
        if "value" in r:
            cpu = r.get("cpu_baseline") or {}
            a = r.get("all_256_sets_on_one_gpu") or {}
            row(name, "**%.0f witnesses/s**, %.1f ms per step, %.2f G nodes/s, %d bundles, program %.0f MB, generate %.1f s + compile %.1f s; one CPU core %.2f /s (x%.0f), all cores %.1f /s (x%.1f); all sets = oracle: %s" % (
                r["value"], r["ms_per_step"], r["nodes_per_sec"] / 1e9, r["bundles"], r["program_bytes"] / 1e6, r["generate_seconds"], r["compile_and_export_seconds"],
                cpu.get("value", 0), cpu.get("gpu_over_one_core", 0), (cpu.get("all_cores") or {}).get("value", 0), cpu.get("gpu_over_all_cores", 0), r.get("matches_oracle")), "bench_line.json")
            if "value" in a:
                row("... all 256 sets of config 5 on ONE GPU", "%.0f witnesses/s, %.1f ms per step = %.2f x the 32-set step (the 8-GPU split of config 5 buys latency, not throughput)" % (a["value"], a["ms_per_step"], a["step_time_over_32_set_step"]), "bench_line.json")
    s = d.get("single_shot") or {}
    if "first_call_ms" in s:
        row("`gw_calc_witness` (one input set, the reference's symbol)", "%.1f ms warm, %.0f ms first call on a graph; `.wtns` = oracle: %s" % (s["warm_call_ms_median_of_last_5"], s["first_call_ms"], s["matches_oracle_wtns"]), "bench_line.json")
    j = d.get("json_front_end") or {}
    e = d.get("e2e_json_to_wtns") or {}
    if j and e:
        row("host sides (SURVEY 8(f) f3)", "JSON front-end %.0f k sets/s; NDJSON -> `.wtns` files %.1f k witnesses/s (%.0f GB/s off the device)" % (j["value"] / 1e3, e["value"] / 1e3, e["link_rate_GBs"]), "bench_line.json")
    row("wall time of the default `python bench.py`", "sub-records %.0f s" % d.get("extras_seconds", 0), "bench_line.json")
for name, what in (("bench_line_config3.json", "config 3 as the timed metric"), ("bench_line_config4.json", "config 4 as the timed metric"), ("bench_line_config5.json", "config 5 (RSA class) as the timed metric"), ("bench_line_config5_bigint.json", "config 5 (first generator) as the timed metric"), ("bench_line_rccl_1rank.json", "the bench under torch.distributed.run with one rank (RCCL broadcast through the C-ABI)")):
    d2 = load(name)
    if d2:
        row(what, "%.1f %s, %.2f ms per step%s" % (d2["value"], d2["unit"], d2["ms_per_step"], (", rccl_ranks %s" % d2.get("rccl_ranks")) if "rccl" in name else ""), name)
for tag, what in (("", "headline"), ("_config3", "config 3"), ("_config4", "config 4"), ("_config5", "config 5 (RSA class)")):
    pm = load("pmc_summary%s.json" % tag)
    if pm:
        k = pm["kernels"]["interp"]
        row("rocprofv3 passes, %s: `%s`" % (what, k["name"].replace("void cwc::", "")), "avg %.3f ms over %d launches; FETCH_SIZE x2 + WRITE_SIZE = %.2f GB per launch; SQ_WAIT_ANY / SQ_WAVE_CYCLES %.3f; VALU wave-instructions %.3g; kernel sources %s" % (
            k["avg_duration_ms"], k["launches"], k["hbm_bytes_per_launch_corrected"] / 1e9, (k.get("sq_wait_any_per_launch") or 0) / max(1.0, k.get("sq_wave_cycles_per_launch") or 1.0), k.get("sq_insts_valu_per_launch") or 0,
            (pm.get("kernel_source_hash") or "?")[:8]), "pmc_summary%s.json, bench_kernel_stats%s.csv" % (tag, tag))
print("| what | number | evidence (profiles/) |\n|---|---|---|")
print("\n".join(rows))
