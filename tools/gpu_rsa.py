"""BASELINE config 5's named class on one GPU: the zk-email RSA / long_div-class graph (graphgen.circuits.build_rsa_long_div_class),
every set against the C oracle.  RSA_N / RSA_K / RSA_MULS: register width, registers, modular multiplications; PROBE_B sets;
PROBE_T program keys (0 = the library's choice); RSA_CHECK how many sets the oracle checks (default all)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from oracle import cbind, model
C = pkg.graphgen.circuits
n, k, muls = int(os.environ.get("RSA_N", "121")), int(os.environ.get("RSA_K", "17")), int(os.environ.get("RSA_MULS", "4"))
B = int(os.environ.get("PROBE_B", "32")); n_check = int(os.environ.get("RSA_CHECK", str(B)))
t = time.time(); data = C.build_rsa_long_div_class(n=n, k=k, muls=muls).to_bin(); print("generated %d bytes in %.1fs" % (len(data), time.time() - t), flush=True)
t = time.time(); g = pkg.Graph(data); print("loaded: n_nodes=%d n_op=%d W=%d depth=%d in %.1fs" % (g.n_nodes, g.n_op, g.n_witness, g.depth, time.time() - t), flush=True)
rnd = random.Random(11)
# sets: uniform field elements (the masks make registers of them), all-ones / zero / one registers (borrows and carries that ripple, equal registers)
rows = [[1] + [rnd.randrange(model.M) if s % 4 else rnd.choice([0, 1, (1 << n) - 1, (1 << n) - 2, rnd.randrange(1 << n)]) for _ in range(g.n_inputs - 1)] for s in range(B)]
inp = cbind.ints_to_array(rows)
d_in = torch.from_numpy(inp).cuda(); d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
og = cbind.Graph(data); t = time.time(); want, wst = og.evaluate_batch(inp[:n_check]); cpu = (time.time() - t) / max(1, n_check)
for tw in [int(x) for x in os.environ.get("PROBE_T", "0,1,2").split(",")]:
    g.set_tile_width(tw)
    key = tw or g.pick_tile_width(B)
    t = time.perf_counter(); blob_len = len(g.export_blob(key)); t_compile = time.perf_counter() - t
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize(); dt = time.perf_counter() - t
    tm = g.last_timing(); ps = g.program_stats(0)
    got = d_out[:n_check].cpu().numpy()
    ok = np.array_equal(got, want) and not wst.any() and not d_st.cpu().numpy().any()
    print("rsa-class n=%d k=%d muls=%d B=%d key %#x (T=%d): compile+export %.1f s, program %.1f MB, bundles=%d %s scan steps %d: %.2f ms (interp %.2f) -> %.1f wit/s, %.3g nodes/s; cpu oracle %.1f ms/witness; parity(%d sets)=%s" % (
        n, k, muls, B, key, tm["tile_width"], t_compile, blob_len / 1e6, tm["n_bundles"], ps["class_bundles"], ps["n_scan_steps"], dt * 1e3, tm["interp_ms"], B / dt, g.n_op * B / dt, cpu * 1e3, n_check, ok), flush=True)
    if not ok:
        bad = [(s, int(np.argmax((got[s] != want[s]).any(axis=1)))) for s in range(n_check) if not np.array_equal(got[s], want[s])]
        print("  MISMATCH sets (set, first witness index):", bad[:10], "status", d_st.cpu().numpy()[:n_check].tolist()[:10], flush=True)
