"""Is the automatic program choice (tile width / divider mode by the runtime's cost model, narrow-bundle policy and
scheduling weights by the compiler's) tied to the one benchmark graph?  Structurally different members of the authV2
class -- tree depths permuted, all chains equal, short ladder, scaled down -- plus a Poseidon-only and the sha256 graph,
each at two batch sizes: the automatic choice against every forced alternative, device-resident, three repetitions.
Prints the ratio auto / best per case (profiles/r02_robustness.txt)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cwc_import  # noqa: E402
pkg = cwc_import.load()
from oracle import cbind  # noqa: E402
import cwc_import  # noqa: E402
C = cwc_import.load().graphgen.circuits
from tools.synth import synth_inputs  # noqa: E402

D, G, T3, S2, S4 = 0x100, 0x200, 0x400, 0x800, 0x1000
BATCHES = tuple(int(x) for x in os.environ.get("ROBUST_BATCHES", "1024,4096").split(","))  # (256: the small-batch regime, stream programs among the candidates)
VARIANTS = [
    ("authV2-class as benchmarked (40/40/64 levels, 254-bit ladder)", lambda: C.build_authv2_class(), "field"),
    ("tree depths 64/20/40", lambda: C.build_authv2_class(levels=(40, 20, 64)), "field"),
    ("all three chains 40 levels", lambda: C.build_authv2_class(levels=(40, 40, 40)), "field"),
    ("one long chain 8/8/64", lambda: C.build_authv2_class(levels=(8, 8, 64)), "field"),
    ("short ladder (96 bits)", lambda: C.build_authv2_class(ladder_bits=96), "field"),
    ("scale 0.6", lambda: C.build_authv2_class(scale=0.6), "field"),
    ("Poseidon(5) only", lambda: C.build_poseidon(5), "field"),
    ("sha256_512", lambda: C.build_sha256(512), "bits"),
]


def measure(g, d_in, d_out, d_st, key, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        gg = pkg.Graph(g) if env else None   # (compiler knobs are read at compile time: a fresh handle)
        h = gg or measure.handle
        h.set_tile_width(key)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            h.calc_witness_batch_device(d_in, d_out, d_st)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        tm = h.last_timing()
        return best, tm
    finally:
        for k in (env or {}):
            del os.environ[k]


for name, build, kind in VARIANTS:
    data = build().to_bin()
    measure.handle = pkg.Graph(data)
    g = measure.handle
    og = cbind.Graph(data)
    for B in BATCHES:
        rows = synth_inputs(kind, g.n_inputs, B, 0xC1C00002)
        d_in = torch.from_numpy(rows).cuda()
        d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        t_auto, tm = measure(data, d_in, d_out, d_st, 0)
        want, _ = og.evaluate_batch(rows[[0, B - 1]])
        ok = np.array_equal(d_out[[0, B - 1]].cpu().numpy(), want)
        alts = {}
        keys = [1 | D, 2 | D, 1, 2, 1 | D | S2, 1 | D | S4, 2 | D | S4, 1 | S4] if B <= 512 else [1 | D, 2 | D, 4 | D, 2, 4, 4 | G, 8] if B == 1024 else [2 | D, 4 | D, 4 | G, 2, 4, 8, 8 | G]
        for key in keys:
            try:
                alts["T=%d%s%s" % (key & 0xff, "+D" if key & D else "+G" if key & G else "", "+S2" if key & S2 else "+S4" if key & S4 else "")] = measure(data, d_in, d_out, d_st, key)[0]
            except pkg.WitnessCalcError:
                pass
        auto_key = tm["tile_width"] | {0: 0, 1: D, 3: T3, 4: G}[tm["divider"]] | {1: 0, 2: S2, 4: S4}[tm["streams"]]
        for label, env in (("no narrow bundles", {"CWC_NO_COOP_MUL": "1"}), ("narrow whenever it fits", {"CWC_COOP_FILL": "64", "CWC_COOP_SLACK": "4000000000"}),
                           ("round-1 weights", {"CWC_SCHED_MUL_COST": "47", "CWC_SCHED_LIN_COST": "12"})):
            alts[label] = measure(data, d_in, d_out, d_st, auto_key, env)[0]
        best_name = min(alts, key=alts.get)
        best = min(alts[best_name], t_auto)
        print("%-62s B=%-5d auto T=%d div=%d streams=%d %7.2f ms %8.0f wit/s parity=%s | best alternative %-24s %7.2f ms | auto/best %.3f" % (
            name, B, tm["tile_width"], tm["divider"], tm["streams"], t_auto * 1e3, B / t_auto, ok, best_name, alts[best_name] * 1e3, t_auto / best), flush=True)
        print("      " + "  ".join("%s %.2f" % (k, v * 1e3) for k, v in alts.items()), flush=True)
        del d_in, d_out, d_st
        torch.cuda.empty_cache()
