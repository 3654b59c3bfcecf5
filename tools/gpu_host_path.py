"""PCIe-inclusive rate of the host-buffer entry point (gwb_calc_witness_batch_host): inputs and witness rows in host
memory, authV2-class graph.  Three destinations: a fresh pageable array per call, a reused pageable array, a pinned
array from gwb_host_alloc."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
from tools.synth import synth_inputs as _synth_inputs


def synth_inputs(kind, n_inputs, batch, seed):
    return _synth_inputs("bits" if kind == "sha256" else "field", n_inputs, batch, seed)
g = pkg.Graph(C.build_authv2_class().to_bin())
print("host cores:", os.cpu_count(), "copy threads:", os.environ.get("CWC_COPY_THREADS", "default"), flush=True)
for B in (1024, 4096):
    rows = synth_inputs("authv2", g.n_inputs, B, 9)
    g.calc_witness_batch(rows[:64])
    reused = np.empty((B, g.n_witness, 32), dtype=np.uint8); reused[:] = 0
    pinned = pkg.pinned_rows((B, g.n_witness, 32))
    ref = None
    for name, out in (("fresh pageable", None), ("reused pageable", reused), ("pinned", pinned)):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); wit, st = g.calc_witness_batch(rows, out=out); best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = wit.copy()
        tm = g.last_timing()
        print("host path B=%d %-16s: %.1f ms -> %.0f wit/s (%.2f GB of witness rows over PCIe; kernels %.1f ms; same bytes %s)" % (
            B, name, best * 1e3, B / best, wit.nbytes / 1e9, tm["interp_ms"] + tm["pack_ms"], np.array_equal(wit, ref)), flush=True)
    del pinned, reused
