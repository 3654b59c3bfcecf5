"""A/B of the interpreter's workgroup shape (CWC_WAVES_PER_WORKGROUP = 1 or 4) over batch sizes and tile widths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
from tools.synth import synth_inputs as _synth_inputs


def synth_inputs(kind, n_inputs, batch, seed):
    return _synth_inputs("bits" if kind == "sha256" else "field", n_inputs, batch, seed)
for kind, builder, cases in (("authv2", C.build_authv2_class, [(512, 0x101), (512, 0x102), (1024, 0x101), (1024, 0x102), (1024, 0x104), (2048, 0x102), (2048, 0x104), (1024, 2), (1536, 2), (2048, 2), (3072, 4), (4096, 4), (6144, 8), (8192, 8)]),
                             ("sha256", lambda: C.build_sha256(512), [(512, 1), (768, 1), (1024, 1), (2048, 2), (4096, 4)])):
    g = pkg.Graph(builder().to_bin())
    for B, T in cases:
        rows = synth_inputs(kind, g.n_inputs, B, 5)
        d_in = torch.from_numpy(rows).cuda()
        d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
        g.set_tile_width(T)
        res = {}
        for pack in ("1", "4", "1", "4"):
            os.environ["CWC_WAVES_PER_WORKGROUP"] = pack
            g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
                best = min(best, g.last_timing()["interp_ms"])
            res.setdefault(pack, []).append(best)
        print("%s B=%-6d key=%#05x tiles=%-5d interp ms: single-wave workgroups %s | four-wave workgroups %s -> %.3f" % (
            kind, B, T, (B + (T & 0xff) - 1) // (T & 0xff), ["%.2f" % x for x in res["1"]], ["%.2f" % x for x in res["4"]], min(res["4"]) / min(res["1"])), flush=True)
        del d_in, d_out, d_st
