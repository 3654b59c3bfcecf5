import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from tools.graphgen import circuits as C
from bench import synth_inputs
g = pkg.Graph(C.build_authv2_class().to_bin())
B = 1024
rows = synth_inputs("authv2", g.n_inputs, B, 5)
d_in = torch.from_numpy(rows).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
g.set_tile_width(2)
for _ in range(2):
    g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
st = d_st.cpu().numpy().view(np.uint32)[:8].astype(np.uint64)
sec = [int(st[2*q]) | (int(st[2*q+1]) << 32) for q in range(4)]
n = sec[3]
print("interp %.1f ms; per bundle (cycles): fetch/decode/forward %.0f | class arithmetic %.0f | store+rotate %.0f | bundles %d" % (g.last_timing()["interp_ms"], sec[0]/n, sec[1]/n, sec[2]/n, n))
