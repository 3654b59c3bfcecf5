"""Programs of several streams (two or four wavefronts per tile over the graph's independent parts) against the
single-stream programs: interpreter + pack time per batch size and program key, parity against the C oracle per line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
import cwc_import
C = cwc_import.load().graphgen.circuits
from tools.synth import synth_inputs
D, S2, S4 = 0x100, 0x800, 0x1000
data = C.build_authv2_class().to_bin()
g = pkg.Graph(data)
og = cbind.Graph(data)
plan = os.environ.get("PLAN", "256:0,1|D,1|D|S2,1|D|S4,2|D|S4;512:0,2|D,2|D|S2,2|D|S4,1|D|S4,4|D|S4;1024:0,2|D,2|D|S2,2|D|S4,4|D|S2,4|D|S4;2048:0,4|D|S4,4|D|S2,8|D|S4")
for part in plan.split(";"):
    B, keys = part.split(":")
    B = int(B)
    rows = synth_inputs("field", g.n_inputs, B, 11)
    want, wst = og.evaluate_batch(rows[:3])
    d_in = torch.from_numpy(rows).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    for k in keys.split(","):
        key = eval(k)
        g.set_tile_width(key)
        d_out.zero_()
        try:
            g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
        except Exception as e:
            print("B=%d key %s: %s" % (B, k, e), flush=True); continue
        ok = np.array_equal(d_out[:3].cpu().numpy(), want) and not d_st.cpu().numpy().any()
        last = d_out[B - 1].cpu().numpy()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        tm = g.last_timing()
        print("B=%-5d key %-10s T=%d div=%d: interp %.2f ms pack %.2f ms wall %.2f ms -> %.0f wit/s  parity=%s" % (
            B, k, tm["tile_width"], tm["divider"], tm["interp_ms"], tm["pack_ms"], best * 1e3, B / best, ok), flush=True)
    del d_in, d_out, d_st
