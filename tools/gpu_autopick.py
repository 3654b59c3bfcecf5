"""Checks the automatic program choice (cost model in pipeline.cc) against neighbours: per batch size, the chosen
tile width / divider and its throughput, authV2-class and sha256 graphs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
from tools.synth import synth_inputs as _synth_inputs


def synth_inputs(kind, n_inputs, batch, seed):
    return _synth_inputs("bits" if kind == "sha256" else "field", n_inputs, batch, seed)

for kind, builder in (("authv2", C.build_authv2_class()), ("sha256", C.build_sha256(512))):
    g = pkg.Graph(builder.to_bin())
    for B in [int(x) for x in os.environ.get("BATCHES", "128,512,1024,2048,4096,8192,16384").split(",")]:
        rows = synth_inputs(kind, g.n_inputs, B, 5)
        d_in = torch.from_numpy(rows).cuda()
        d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        g.set_tile_width(0)
        t0 = time.perf_counter()
        g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
        first = time.perf_counter() - t0
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        tm = g.last_timing()
        print("%-7s B=%-6d auto -> T=%-2d%s  %7.1f ms  %9.0f wit/s  (first call incl. compiles %.1f s)" % (
            kind, B, tm["tile_width"], {0: "  ", 1: "+D", 3: "+3", 4: "+G"}[tm["divider"]], best * 1e3, B / best, first), flush=True)
        del d_in, d_out, d_st
