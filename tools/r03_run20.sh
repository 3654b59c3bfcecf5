export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/pmc_icache $O/pmc_dcache
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_icache -o runc -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --extras 0 > $O/pmc_icache.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_INSTS_SMEM -d $O/pmc_dcache -o runc -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --extras 0 > $O/pmc_dcache.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_icache", "pmc_dcache"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "interp" in k or "pack" in k:
                print(d, k, {c: "%.4g" % x for c, x in v.items()})
PY
