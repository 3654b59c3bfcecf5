import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from tools.graphgen import circuits as C
from bench import synth_inputs
g = pkg.Graph(C.build_authv2_class().to_bin())
B = 8192
rows = synth_inputs("authv2", g.n_inputs, B, 5)
d_in = torch.from_numpy(rows).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
for tw in (4, 8):
    g.set_tile_width(tw)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("  B=%d T=%d: %.1f ms" % (B, tw, best * 1e3))
