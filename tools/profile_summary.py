"""Turns the rocprofv3 outputs of a gpurun call (rocpd sqlite databases under gpurun_out/) into the small, tracked
evidence files under profiles/: per-kernel statistics, this repository's kernel dispatches, the PMC counter rows, and
rNN_pmc_summary.json (HBM bytes per launch with the gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md: FETCH_SIZE
counts a 128-byte request of a 16-byte-per-lane read at 64 bytes -> x2).

    python tools/profile_summary.py r01 --graph authv2 --batch 1024
"""
import argparse, csv, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def db(path):
    f = os.path.join(ROOT, "gpurun_out", path, "runc_results.db")
    if not os.path.exists(f):
        sys.exit("missing " + f)
    return sqlite3.connect(f)


def ours(name):
    return "cwc::" in name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round")
    ap.add_argument("--graph", default="authv2")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--tag", default="", help="passes of another bench command line (tools/collect_evidence.sh: config3, config4, config5): reads gpurun_out/*_<round>_<tag>, "
                                              "writes profiles/<round>_*_<tag>.*")
    ap.add_argument("--cmd", default="python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --extras 0")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"), help="where the summaries go (tools/collect_evidence.sh summarises on the GPU box, into gpurun_out/summary_<round>: "
                                                                          "the rocpd databases of four configurations do not fit what a lease copies back)")
    a = ap.parse_args()
    r = a.round
    sfx = ("_" + a.tag) if a.tag else ""
    out = a.out
    os.makedirs(out, exist_ok=True)
    # ---- kernel trace + stats ----
    con = db("prof_%s%s" % (r, sfx))
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(os.path.join(out, "%s_bench_kernel_stats%s.csv" % (r, sfx)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for n, c, t, avg, pct in rows:
            w.writerow([n, c, "%.3f" % t, "%.3f" % avg, "%.4f" % pct])
    disp = list(con.execute("select name, dispatch_id, start, end, duration, grid_x, grid_y, workgroup_x, workgroup_y, "
                            "lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start"))
    with open(os.path.join(out, "%s_bench_kernel_trace_cwc%s.csv" % (r, sfx)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatch_Id", "Start_ns", "End_ns", "Duration_ns", "Grid_X", "Grid_Y", "Workgroup_X",
                    "Workgroup_Y", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"])
        for d in disp:
            if ours(d[0]):
                w.writerow(d)
    interp = [d for d in disp if "interp_kernel" in d[0]]
    pack = [d for d in disp if "pack_kernel" in d[0]]
    tile = int(interp[0][0].split("interp_kernel<")[1].split(",")[0])
    divider = int(interp[0][0].split("interp_kernel<")[1].split(">")[0].split(",")[2].strip())
    ksrc = None  # the kernel sources the profiled library was built from (written on the GPU box by tools/collect_evidence.sh)
    try:
        ksrc = open(os.path.join(ROOT, "gpurun_out", "ksrc_%s.txt" % r)).read().strip() or None
    except OSError:
        pass
    summary = {"config": {"graph": a.graph, "batch_per_gpu": a.batch, "tile_width": tile, "interpreter_waves_per_divider_wave": divider},
               "kernel_source_hash": ksrc,
               "source": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* passes of "
                         "`%s`, one pass per command" % a.cmd,
               "kernels": {"interp": {"name": interp[0][0].split("(")[0], "launches": len(interp),
                                      "avg_duration_ms": sum(d[4] for d in interp) / len(interp) / 1e6,
                                      "vgpr": interp[0][10], "sgpr": interp[0][12], "lds_bytes": interp[0][9],
                                      "grid": interp[0][5], "workgroup": interp[0][7]},
                           "pack": {"launches": len(pack), "avg_duration_ms": sum(d[4] for d in pack) / max(len(pack), 1) / 1e6}}}
    # ---- PMC passes ----
    def counters(path, fname):
        con = db(path)
        rows = list(con.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection order by dispatch_id, counter_name"))
        with open(os.path.join(out, fname), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Dispatch_Id", "Counter_Name", "Counter_Value", "Duration_ns"])
            for x in rows:
                if ours(x[0]):
                    w.writerow([x[0].split("(")[0]] + list(x[1:]))
        return [x for x in rows if ours(x[0])]
    fetch = counters("pmc_fetch_%s%s" % (r, sfx), "%s_bench_pmc_fetch_size%s.csv" % (r, sfx))
    write = counters("pmc_write_%s%s" % (r, sfx), "%s_bench_pmc_write_size%s.csv" % (r, sfx))
    sq = counters("pmc_sq_%s%s" % (r, sfx), "%s_bench_pmc_sq%s.csv" % (r, sfx))

    def per_launch(rows, kern, ctr):
        v = [x[3] for x in rows if kern in x[0] and x[2] == ctr]
        return sum(v) / len(v) if v else None
    for k, kern in (("interp", "interp_kernel"), ("pack", "pack_kernel")):
        f_kib, w_kib = per_launch(fetch, kern, "FETCH_SIZE"), per_launch(write, kern, "WRITE_SIZE")
        e = summary["kernels"][k]
        e["fetch_size_kib_per_launch_raw"] = f_kib
        e["write_size_kib_per_launch"] = w_kib
        e["hbm_bytes_per_launch_corrected"] = (2.0 * f_kib + w_kib) * 1024.0  # FETCH_SIZE x2 on gfx950 (16 B/lane reads)
        for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            e[c.lower() + "_per_launch"] = per_launch(sq, kern, c)
    json.dump(summary, open(os.path.join(out, "%s_pmc_summary%s.json" % (r, sfx)), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
