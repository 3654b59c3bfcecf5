"""Re-measures the cost model's cycle table on the machine at hand (the table in csrc/compile.cc was taken on one MI355X box).

The stamped interpreter build (gwb_profile_classes) runs the bench graphs -- the authV2-class graph at the headline's
program and as a lone-wave stream program, with and without divider waves, the bigint-class graph for the integer classes --
and the cycles per bundle of every class it sees, net of the build's own time stamps, are set beside the table the
library loaded.  `--write` leaves them as "class:cycles,..." in model_cycles.txt of the program cache's directory (or the
path given), which the library reads when it is loaded (CycleTable in csrc/compile.cc; CWC_MODEL_CYCLES_FILE names another
file); without it nothing changes.

    python tools/gpu_calibrate.py [--write [PATH]]

Classes priced with operand-form / stage adjustments (BIT, IDIVMOD, CMPS, MULF, SCAN) are reported but not written: their
table entries are the all-conversions / all-stages price, not what a mixed program measures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import subprocess
# the stamped interpreter instances live in the diagnostic library (make diag): built on demand, loaded in place of the product's
_PKG = os.path.join(ROOT, "circom-witnesscalc_amd")
subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc"), "diag"])
os.environ["CWC_LIB_PATH"] = os.path.join(_PKG, "libcircom_witnesscalc_amd_diag.so")
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits

STAMPS = {"MULF": 2}   # time stamps of ~40 cycles inside a bundle's measured span (every other class path: 5)
STAMP_CYCLES = 40
ADJUSTED = ("BIT", "IDIVMOD", "CMPS", "MULF", "SCAN")
M = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def measure(graph_bytes, batch, key, seed):
    g = pkg.Graph(graph_bytes)
    rng = np.random.default_rng(seed)
    rows = np.zeros((batch, g.n_inputs, 32), dtype=np.uint8)
    rows[:, :, :31] = rng.integers(0, 256, size=(batch, g.n_inputs, 31), dtype=np.uint8)  # < 2^248 < r
    rows[:, 0, :] = 0
    rows[:, 0, 0] = 1
    d_in = torch.from_numpy(rows).cuda()
    d_out = torch.empty((batch, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(batch, dtype=torch.int32, device="cuda")
    g.set_tile_width(key)
    g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    prof = g.profile_classes(d_in, d_out, d_st)
    out = {}
    for name, v in prof.items():
        if name.startswith("_") or not v[3]:
            continue
        out[name] = (v[0] / v[3] - STAMPS.get(name, 5) * STAMP_CYCLES, v[3])
    return out


def main():
    table = pkg.model_cycles()
    authv2 = C.build_authv2_class().to_bin()
    runs = [("authV2-class, 1024 sets, T = 2 + divider waves (the headline's program)", authv2, 1024, 2 | 0x100),
            ("authV2-class, 256 sets, T = 1 + divider waves, four streams", authv2, 256, 1 | 0x100 | 0x1000),
            ("authV2-class, 1024 sets, T = 2, inline divisions", authv2, 1024, 2),
            ("bigint-class (k = 32, 40 rounds), 32 sets, T = 2", C.build_bigint_class(k=32, rounds=40).to_bin(), 32, 2)]
    best = {}
    for title, data, batch, key in runs:
        got = measure(data, batch, key, 7)
        print(title)
        for name, (cyc, n) in sorted(got.items()):
            print("   %-8s %9.0f cycles/bundle net of stamps (%d bundles sampled)   table %8.0f   measured/table %.2f" % (name, cyc, n, table[name], cyc / table[name]))
            if name not in best or n > best[name][1]:
                best[name] = (cyc, n)
    text = ",".join("%d:%.0f" % (pkg.CLASS_NAMES.index(n), c) for n, (c, _k) in sorted(best.items()) if n not in ADJUSTED)
    print("calibration (classes with the most bundles sampled; " + ", ".join(ADJUSTED) + " left at the table's values):")
    print("   CWC_MODEL_CYCLES=" + text)
    if "--write" in sys.argv:
        k = sys.argv.index("--write")
        path = sys.argv[k + 1] if k + 1 < len(sys.argv) else None
        if path is None:
            d = os.environ.get("CWC_PROGRAM_CACHE") or (os.environ.get("XDG_CACHE_HOME") and os.path.join(os.environ["XDG_CACHE_HOME"], "circom-witnesscalc-amd")) or \
                os.path.join(os.path.expanduser("~"), ".cache", "circom-witnesscalc-amd")
            if d in ("0", "off"):
                sys.exit("no cache directory (CWC_PROGRAM_CACHE=%s): give a path" % d)
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "model_cycles.txt")
        with open(path, "w") as f:
            f.write(text + "\n")
        print("   written to", path)


if __name__ == "__main__":
    main()
