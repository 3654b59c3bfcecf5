"""Throughput sweep over batch size and tile width (device-resident buffers), authV2-class and sha256 graphs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
import cwc_import
C = cwc_import.load().graphgen.circuits
sys.path.insert(0, ROOT)
from tools.synth import synth_inputs as _synth_inputs


def synth_inputs(kind, n_inputs, batch, seed):
    return _synth_inputs("bits" if kind == "sha256" else "field", n_inputs, batch, seed)

def run(kind, builder, cases):
    data = builder.to_bin()
    g = pkg.Graph(data); og = cbind.Graph(data)
    print("%s: n_op=%d W=%d depth=%d alg bytes/set=%d" % (kind, g.n_op, g.n_witness, g.depth, g.algorithmic_bytes_per_set), flush=True)
    for B, tiles in cases:
        rows = synth_inputs(kind, g.n_inputs, B, 123)
        d_in = torch.from_numpy(rows).cuda()
        d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        want, _ = og.evaluate_batch(rows[[0, B // 2, B - 1]])
        for tw in tiles:
            g.set_tile_width(tw)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                g.calc_witness_batch_device(d_in, d_out, d_st)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            tm = g.last_timing()
            ok = np.array_equal(d_out[[0, B // 2, B - 1]].cpu().numpy(), want) and int((d_st != 0).sum()) == 0
            print("  %-7s B=%-6d T=%-2d%s launches=%-2d wall %8.1f ms -> %9.0f wit/s  %.3g field-ops/s  alg-roofline frac %.3f  parity=%s" % (
                kind, B, tm["tile_width"], {0: "  ", 1: "+D", 3: "+3", 4: "+G"}.get(tm["divider"], "+?"), tm["n_launches"], best * 1e3, B / best, g.n_op * B / best,
                g.algorithmic_bytes_per_set * B / best / 8e12, ok), flush=True)
        del d_in, d_out, d_st
        torch.cuda.empty_cache()

which = os.environ.get("SWEEP", "authv2,sha256").split(",")
if "authv2" in which:
    D, G = 0x100, 0x200  # asynchronous divider programs: one divider wave per interpreter wave / per four
    cases = [(256, (1, 1 | D)), (512, (1, 1 | D, 2 | D)), (1024, (1, 2, 1 | D, 2 | D, 4 | D)), (2048, (2, 4, 2 | D, 4 | D)),
             (4096, (2, 4, 4 | D, 4 | G, 2 | G)), (8192, (4, 8, 4 | G, 8 | G, 2 | G)), (16384, (8, 16, 8 | G, 4 | G)), (32768, (16, 16 | G, 8 | G))]
    if os.environ.get("SWEEP_CASES"):  # e.g. SWEEP_CASES="8192:4,8,264,520,1032;16384:8,16,272" (program keys: T | 0x100 divider | 0x200 group | 0x400 triple)
        cases = [(int(c.split(":")[0]), tuple(int(k) for k in c.split(":")[1].split(","))) for c in os.environ["SWEEP_CASES"].split(";")]
    elif os.environ.get("SWEEP_BIG"):
        cases = [(8192, (4, 8)), (16384, (4, 8, 16)), (32768, (8, 16, 32))]
    run("authv2", C.build_authv2_class(), cases)
if "sha256" in which:
    run("sha256", C.build_sha256(512), [(1024, (1, 2, 4)), (4096, (2, 4, 8, 16)), (16384, (8, 16, 32))])
