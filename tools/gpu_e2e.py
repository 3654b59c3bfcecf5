"""End-to-end streaming pipeline (NDJSON -> .wtns files on tmpfs) under a few settings: CWC_WRITE_THREADS, CWC_E2E_SUBBATCH."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import cwc_import
pkg = cwc_import.load()
wl = bench.Workload("authv2")
g = pkg.Graph(wl.data)
for env in ({}, {}, {"CWC_WRITE_THREADS": "8"}, {"CWC_WRITE_THREADS": "12"}, {"CWC_WRITE_THREADS": "24"}, {"CWC_WRITE_THREADS": "32"}, {"CWC_WRITE_THREADS": "64"},
            {"CWC_E2E_SLICE_MB": "24"}, {"CWC_E2E_SLICE_MB": "192"}, {"CWC_E2E_SUBBATCH": "512"}, {"CWC_E2E_SUBBATCH": "2048"}, {}):
    os.environ.update(env)
    r = bench.e2e_json_to_wtns_point(wl, g, n=8192)
    for k in env:
        del os.environ[k]
    print(env, "-> %.0f witnesses/s, %.1f GB/s, wait for drain %.3f s, parse %.3f s, ok %s" % (
        r["value"], r["link_rate_GBs"], r["wait_for_drain_seconds_last_round"], r["parse_seconds_last_round"], r["matches_oracle_files"]), flush=True)
