"""Run a few batches of the authV2-class graph (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
g = pkg.Graph(C.build_authv2_class().to_bin())
B = int(os.environ.get("PROBE_B", "1024")); T = int(os.environ.get("PROBE_T", "2"))
rng = np.random.default_rng(1)
rows = np.frombuffer(rng.bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy()
rows[:, :, 31] &= 0x1f; rows[:, 0, :] = 0; rows[:, 0, 0] = 1
d_in = torch.from_numpy(rows).cuda()
d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
g.set_tile_width(T)
for _ in range(2):
    g.calc_witness_batch_device(d_in, d_out, d_st)
torch.cuda.synchronize()
print(g.last_timing())
