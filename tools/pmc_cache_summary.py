"""Instruction-cache and scalar-data-cache counters of the bench's kernels, per launch, from the rocprofv3 --pmc passes
tools/collect_evidence.sh makes (gpurun_out/pmc_icache_<round>, pmc_dcache_<round>):  python tools/pmc_cache_summary.py r03"""
import collections, os, sqlite3, sys

r = sys.argv[1] if len(sys.argv) > 1 else "r03"
for d in ("pmc_icache_" + r, "pmc_dcache_" + r):
    path = os.path.join("gpurun_out", d, "runc_results.db")
    if not os.path.exists(path):
        print(d, ": no database")
        continue
    con = sqlite3.connect(path)
    view = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')") if t[0].startswith("counters_collection")][0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v in con.execute("select kernel_name, counter_name, value from %s" % view):
        if "interp_kernel" in k or "pack_kernel" in k:
            acc[k.split("(")[0]][c].append(v)
    for k, v in sorted(acc.items()):
        print(d, k)
        for c, x in sorted(v.items()):
            print("   %-32s %14.0f per launch (%d launches)" % (c, sum(x) / len(x), len(x)))
