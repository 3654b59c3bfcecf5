"""PCIe-inclusive rate of the host-buffer entry point (gwb_calc_witness_batch_host): inputs and witness rows in host
memory, authV2-class graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cwc_import
pkg = cwc_import.load()
from tools.graphgen import circuits as C
from bench import synth_inputs
g = pkg.Graph(C.build_authv2_class().to_bin())
for B in (1024, 4096):
    rows = synth_inputs("authv2", g.n_inputs, B, 9)
    g.calc_witness_batch(rows[:64])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); wit, st = g.calc_witness_batch(rows); best = min(best, time.perf_counter() - t0)
    print("host path B=%d: %.1f ms -> %.0f wit/s (%.2f GB of witness rows back over PCIe)" % (B, best * 1e3, B / best, wit.nbytes / 1e9), flush=True)
