"""First-contact GPU probe: parity of every op class vs the C oracle + a tile-width timing sweep."""
import os, sys, time, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cwc_import
pkg = cwc_import.load()
from oracle import cbind, model
import cwc_import
C = cwc_import.load().graphgen.circuits

def parity(name, builder, rows, tws=(1, 4, 64)):
    data = builder.to_bin()
    g = pkg.Graph(data); og = cbind.Graph(data)
    inp = cbind.ints_to_array(rows)
    want, wst = og.evaluate_batch(inp)
    for tw in tws:
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        okm = wst == 0
        same_st = np.array_equal(st != 0, wst != 0)
        same = np.array_equal(got[okm], want[okm])
        print("parity %-10s T=%-2d sets=%d status_match=%s witness_match=%s timing=%s" % (name, tw, len(rows), same_st, same, g.last_timing()), flush=True)
        if not same:
            bad = np.argwhere((got[okm] != want[okm]).any(axis=2))
            print("   first mismatches (set, witness idx):", bad[:5].tolist())

rnd = random.Random(11)
def rrow(n, small=0.3):
    return [1] + [rnd.randrange(model.M) if rnd.random() > small else rnd.randrange(1 << 16) for _ in range(n - 1)]

t0 = time.time()
parity("circuit1", C.build_circuit1(), [[1, 105, 303]] + [rrow(3) for _ in range(9)])
parity("gadgets", C.build_gadgets(), [rrow(7) for _ in range(70)])
for seed in range(4):
    parity("dag%d" % seed, C.build_random_dag(seed, n_ops=400), [rrow(7) for _ in range(33)], tws=(1, 64))
parity("poseidon2", C.build_poseidon(2), [rrow(3, 0) for _ in range(130)], tws=(1, 2, 8, 16, 32, 64))
print("parity phase s", time.time() - t0, flush=True)

# timing sweep on authV2-class
b = C.build_authv2_class(); data = b.to_bin()
g = pkg.Graph(data); og = cbind.Graph(data)
print("authv2 info", {k: getattr(g, k) for k, _ in pkg.GraphInfo._fields_}, flush=True)
import torch
B = int(os.environ.get("PROBE_B", "1024"))
rows = np.frombuffer(np.random.default_rng(1).bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy()
rows[:, :, 31] &= 0x1f   # < 2^253 < r
rows[:, 0, :] = 0; rows[:, 0, 0] = 1
d_in = torch.from_numpy(rows).cuda()
d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
want, wst = og.evaluate_batch(rows[:4])
for tw in [int(x) for x in os.environ.get("PROBE_T", "1,2,4,8,64").split(",")]:
    g.set_tile_width(tw)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        g.calc_witness_batch_device(d_in, d_out, d_st)
        torch.cuda.synchronize(); dt = time.time() - t
    tm = g.last_timing()
    got = d_out[:4].cpu().numpy()
    print("authv2 B=%d T=%d wall %.1f ms interp %.1f ms pack %.2f ms -> %.0f wit/s, %.3g field-ops/s, roofline frac %.4f, parity(4 sets)=%s status_nonzero=%d" % (
        B, tw, dt * 1e3, tm["interp_ms"], tm["pack_ms"], B / dt, g.n_op * B / dt,
        g.algorithmic_bytes_per_set * B / (tm["interp_ms"] * 1e-3) / 8e12, np.array_equal(got, want), int((d_st != 0).sum())), flush=True)
