"""Debug aid: one limb-chain graph on the GPU, signal by signal against the oracle."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
C = pkg.graphgen.circuits
steps = int(os.environ.get("STEPS", "10")); chains = int(os.environ.get("CHAINS", "1"))
b = C.build_limb_chains(64, 64, steps, chains, True, False)
data = b.to_bin()
g = pkg.Graph(data)
rnd = random.Random(5)
rows = [[1] + [rnd.randrange(1 << 64) for _ in range(g.n_inputs - 1)] for _ in range(4)]
rows.append([1] + [(1 << 64) - 1] * (g.n_inputs - 1))
arr = cbind.ints_to_array(rows)
want, wst = cbind.Graph(data).evaluate_batch(arr)
for tw in (1, 2):
    g.set_tile_width(tw)
    got, st = g.calc_witness_batch(arr)
    print("T=%d status %s bundles %s" % (tw, st.tolist(), g.program_stats(tw)["class_bundles"]))
    for k in range(len(rows)):
        gi, wi = cbind.array_to_ints(got[k]), cbind.array_to_ints(want[k])
        base = g.n_inputs
        bad = [j - base for j in range(base, len(wi)) if gi[j] != wi[j]]
        print(" set %d: %d signals, mismatching (signal index: carry chain = limb,carry pairs first, then q,rem pairs): %s" % (k, len(wi) - base, bad[:40]))
        for j in bad[:6]:
            print("   sig %d got %x want %x" % (j, gi[base + j], wi[base + j]))
