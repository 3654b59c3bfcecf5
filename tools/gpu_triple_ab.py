"""Triple-divider programs (three interpreters + one divider per workgroup) against the other divider modes at the
batch sizes where 513..768 tiles come up."""
import os, sys
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
g = pkg.Graph(C.build_authv2_class().to_bin())
D, G, T3 = 0x100, 0x200, 0x400
for B, keys in ((768, (1 | T3, 2 | D, 1 | D, 1 | G)), (1024, (2 | D, 2 | T3)), (1536, (2 | T3, 4 | D, 2 | G, 2 | D)), (3072, (4 | T3, 4 | G, 4 | D, 8 | D)),
                (6144, (8 | T3, 4, 8 | G)), (2048, (4 | D, 4 | T3))):
    rows = synth_inputs("authv2", g.n_inputs, B, 5)
    d_in = torch.from_numpy(rows).cuda()
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    out = []
    for key in keys + (0,):
        g.set_tile_width(key)
        g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
            tm = g.last_timing(); best = min(best, tm["interp_ms"] + tm["pack_ms"])
        out.append("%s T=%d+%d: %.2f ms" % ("auto ->" if key == 0 else "", tm["tile_width"], tm["divider"], best))
    print("B=%d: %s" % (B, " | ".join(out)), flush=True)
    del d_in, d_out, d_st
