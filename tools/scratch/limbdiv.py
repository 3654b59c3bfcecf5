import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
C = pkg.graphgen.circuits
b = C.build_limb_graph_with_divisions(k=16, rounds=60)
data = b.to_bin(); g = pkg.Graph(data); og = cbind.Graph(data)
B = 32
rng = np.random.default_rng(5)
rows = np.frombuffer(rng.bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy(); rows[:, :, 8:] = 0; rows[:, 0, :] = 0; rows[:, 0, 0] = 1
d_in = torch.from_numpy(rows).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
want, wst = og.evaluate_batch(rows[:4])
for key in (1 | 0x100, 2 | 0x100, 1):
    g.set_tile_width(key)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize(); dt = time.perf_counter() - t
    tm = g.last_timing(); ps = g.program_stats(0)
    ok = np.array_equal(d_out[:4].cpu().numpy()[wst == 0], want[wst == 0])
    print("limb graph with field divisions, n_op %d, key %#x: %.2f ms interp %.2f, bundles %d, scan steps %d, divider %d, parity %s" % (g.n_op, key, dt * 1e3, tm["interp_ms"], tm["n_bundles"], ps["n_scan_steps"], tm["divider"], ok), flush=True)
