"""Diagnostic: per-bundle-class cycle shares of the interpreter (stamped build) on the authV2-class graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import subprocess
# the stamped interpreter instances live in the diagnostic library (make diag): built on demand, loaded in place of the product's
_PKG = os.path.join(ROOT, "circom-witnesscalc_amd")
if os.environ.get("CLASSPROF_LIB"):  # (another diagnostic build, e.g. one whose first section row samples another class: -DCWC_PSEC_CLASS)
    os.environ["CWC_LIB_PATH"] = os.environ["CLASSPROF_LIB"]
else:
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc"), "diag"])
    os.environ["CWC_LIB_PATH"] = os.path.join(_PKG, "libcircom_witnesscalc_amd_diag.so")
import cwc_import
pkg = cwc_import.load()
import cwc_import
C = cwc_import.load().graphgen.circuits
kind = os.environ.get("PROBE_GRAPH", "authv2")
b = (C.build_authv2_class() if kind == "authv2" else C.build_sha256(512) if kind == "sha256" else
     C.build_rsa_long_div_class(n=int(os.environ.get("RSA_N", "121")), k=int(os.environ.get("RSA_K", "17")), muls=int(os.environ.get("RSA_MULS", "4"))) if kind == "rsa" else
     C.build_bigint_class(k=32, rounds=int(os.environ.get("BIGINT_ROUNDS", "400"))))
g = pkg.Graph(b.to_bin())
B = int(os.environ.get("PROBE_B", "1024"))
rng = np.random.default_rng(1)
rows = np.frombuffer(rng.bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy()
if kind != "sha256":
    rows[:, :, 31] &= 0x1f
else:
    rows[:, :, 1:] = 0; rows[:, :, 0] &= 1
rows[:, 0, :] = 0; rows[:, 0, 0] = 1
d_in = torch.from_numpy(rows).cuda()
d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
for tw in [int(x) for x in os.environ.get("PROBE_T", "1,4,64").split(",")]:
    g.set_tile_width(tw)
    g.calc_witness_batch_device(d_in, d_out, d_st); torch.cuda.synchronize()
    t = g.last_timing()
    prof = g.profile_classes(d_in, d_out, d_st)
    tiles = (B + (tw & 0xff) - 1) // (tw & 0xff)
    nw = max(1, tiles // 64 + (1 if tiles % 64 else 0))
    sections = prof.pop("_sections")
    issue_part = prof.pop("_issue_part", {})
    scan_kinds = prof.pop("_scan_kinds")
    wv = prof.pop("_waves")
    tot = sum(v[0] for v in prof.values())
    print("T=%d%s B=%d product interp %.1f ms; stamped build: sampled waves=%d total cycles/wave %.3g" % (tw & 0xff, " + divider wave" if tw & 0x100 else "", B, t["interp_ms"], nw, tot / nw))
    print("   interpreter waves %d: loop cycles min %.3g mean %.3g max %.3g" % (wv["n"], wv["min_cycles"], wv["mean_cycles"], wv["max_cycles"]))
    for k, (cyc, _a, _b, n) in prof.items():
        if n:
            print("   %-8s bundles/wave %7d  cycles/bundle %8.0f  share %.1f%%" % (k, n // nw, cyc / n, 100.0 * cyc / tot))
    for k, (cyc, _a, _b, n) in scan_kinds.items():
        if n:
            print("   scan bundles, %-11s bundles/wave %7d  cycles/bundle %8.0f" % (k, n // nw, cyc / n))
    for k, v in sections.items():
        if v[5]:
            print("   %s sections (cycles/bundle, each includes one ~40-cycle stamp): top+vmcnt wait %.0f | LDS reads + previous stores %.0f | staging issue %.0f | arithmetic %.0f | ring write %.0f" % (
                (k,) + tuple(x / v[5] for x in v[:5])))
            if issue_part.get(k):
                print("   %s section 2 split (one stamp in it, not waited for): reads + record refill + previous stores issued after %.0f cycles | wait for the LDS reads %.0f" % (
                    k, issue_part[k] / v[5], (v[1] - issue_part[k]) / v[5]))
