"""BASELINE config 5 class on one GPU: ~N-million-node bigint/long_div-class synthetic graph, small batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
import cwc_import
C = cwc_import.load().graphgen.circuits
rounds = int(os.environ.get("BIGINT_ROUNDS", "400")); k = int(os.environ.get("BIGINT_K", "32")); B = int(os.environ.get("PROBE_B", "32"))
t = time.time(); b = C.build_bigint_class(k=k, rounds=rounds); data = b.to_bin(); print("generated %d bytes in %.1fs" % (len(data), time.time() - t), flush=True)
t = time.time(); g = pkg.Graph(data); print("loaded: n_nodes=%d n_op=%d W=%d depth=%d in %.1fs" % (g.n_nodes, g.n_op, g.n_witness, g.depth, time.time() - t), flush=True)
rng = np.random.default_rng(3)
rows = np.frombuffer(rng.bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy(); rows[:, :, 31] &= 0x1f; rows[:, 0, :] = 0; rows[:, 0, 0] = 1
d_in = torch.from_numpy(rows).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
og = cbind.Graph(data); t = time.time(); want, wst = og.evaluate_batch(rows[:2]); cpu = (time.time() - t) / 2
for tw in [int(x) for x in os.environ.get("PROBE_T", "0,1").split(",")]:   # 0 = the library's choice for this batch
    g.set_tile_width(tw)
    t = time.perf_counter(); blob_len = len(g.export_blob(tw or g.pick_tile_width(B))); t_compile = time.perf_counter() - t   # (compiles that key on the host)
    print("key %#x: compiled + exported in %.1f s, program %.0f MB" % (tw or g.pick_tile_width(B), t_compile, blob_len / 1e6), flush=True)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize(); dt = time.perf_counter() - t
    tm = g.last_timing()
    tw = tm["tile_width"]
    print("bigint-class B=%d T=%d bundles=%d slots=%d launches=%d: %.1f ms -> %.1f wit/s, %.3g field-ops/s; cpu oracle %.1f ms/witness; parity=%s" % (
        B, tw, tm["n_bundles"], tm["n_slots"], tm["n_launches"], dt * 1e3, B / dt, g.n_op * B / dt, cpu * 1e3,
        np.array_equal(d_out[:2].cpu().numpy(), want) and not wst.any()), flush=True)
