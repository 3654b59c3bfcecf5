"""Soak / fuzz parity run on the GPU: many random DAGs (every op, panic edges included) and chain-heavy graphs, random
batch sizes and program keys (tile widths, divider modes), byte-compared with the C oracle.  Exit code 1 on mismatch.

    SOAK_SEEDS=300 python tools/gpu_soak.py
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cwc_import
pkg = cwc_import.load()
from oracle import cbind
import cwc_import
C = cwc_import.load().graphgen.circuits

M = 21888242871839275222246405745257275088548364400416034343698204186575808495617
EDGE = [0, 1, 2, 3, 63, 64, 65, 127, 128, 253, 254, 255, 256, M - 1, M - 2, M // 2, M // 2 + 1, 1 << 253, (1 << 64) - 1, 1 << 64,
        (1 << 128) - 1, 1 << 200, M & ((1 << 253) - 1), M ^ (M & ((1 << 253) - 1))]
KEYS = [1, 2, 4, 8, 16, 32, 64, 1 | 0x100, 2 | 0x100, 8 | 0x100, 32 | 0x100, 1 | 0x200, 4 | 0x200, 16 | 0x200, 1 | 0x400, 2 | 0x400, 8 | 0x400,
        1 | 0x800, 2 | 0x1000, 1 | 0x100 | 0x1000, 4 | 0x100 | 0x800, 8 | 0x1000, 16 | 0x100 | 0x1000]  # (0x800 / 0x1000: two / four streams per tile)
KINDS = os.environ.get("SOAK_KINDS", "dag,dag,dag_panic,chains,forest,forest_panic,limb").split(",")
LIMB_KEYS = [1, 2, 1 | 0x100, 2 | 0x100, 1 | 0x1000, 2 | 0x100 | 0x1000, 4, 8]  # (scan bundles exist at tile widths 1 and 2)


def run(n_seeds, base, verbose=True):
    """returns the number of mismatching (graph, program) runs"""
    rnd = random.Random(base)
    bad = 0
    t0 = time.time()
    for s in range(n_seeds):
        kind = rnd.choice(KINDS)
        if kind == "limb":  # serial limb recurrences: what the compiler runs as scan bundles (tile widths 1 and 2), every shift / base width
            if rnd.random() < float(os.environ.get("SOAK_WIDE_SHARE", "0.3")):  # round 5: registers wider than a word -- borrow chains, comparisons, multi-register long division
                os.environ.pop("CWC_CONV_ANY_WIDTH", None)
                os.environ.pop("CWC_CONV_ALWAYS", None)
                if rnd.random() < 0.6:
                    b = C.build_bit_recurrence_variants(rnd.randrange(1 << 30))
                else:
                    b = C.build_rsa_long_div_class(n=rnd.choice([55, 64, 100, 121, 121, rnd.randrange(12, 127)]), k=rnd.choice([1, 2, 3, 4, 6]), muls=rnd.randrange(1, 3),
                                                   range_checks=rnd.random() < 0.3)
            elif rnd.random() < 0.15:  # (limb products in the shapes the convolution rewrite has to tell apart)
                os.environ.pop("CWC_CONV_ANY_WIDTH", None)
                os.environ.pop("CWC_CONV_ALWAYS", None)
                if rnd.random() < 0.5:
                    os.environ["CWC_CONV_ALWAYS"] = "1"
                b = C.build_limb_product_variants(rnd.randrange(1 << 30))
            elif rnd.random() < 0.25:  # (schoolbook limb products: convolution bundles where 2k - 1 columns fit the tile width's node slots)
                b = C.build_bigint_class(k=rnd.choice([2, 3, 5, 8, 11, 16, 17, 24, 32, rnd.randrange(2, 33)]), rounds=rnd.randrange(1, 4),
                                         n_bits=rnd.choice([64, 64, 64, 16, 32, 63, 65, 100, 126]))
                # (limbs beyond 64 bits keep their unfused products unless this is set: half of the graphs run the bundle's field-arithmetic rounds)
                os.environ.pop("CWC_CONV_ANY_WIDTH", None)
                os.environ.pop("CWC_CONV_ALWAYS", None)
                if rnd.random() < 0.5:
                    os.environ["CWC_CONV_ANY_WIDTH"] = "1"
                if rnd.random() < 0.7:  # (else the unfused program competes: the cost model's pick)
                    os.environ["CWC_CONV_ALWAYS"] = "1"
            else:
                b = C.build_limb_chains(rnd.choice([1, 31, 32, 33, 63, 64, 65, 100, 121, 127, 128, 129, 200, 253, rnd.randrange(1, 254)]),
                                        rnd.choice([1, 17, 32, 63, 64, 65, 121, 128, 253, rnd.randrange(1, 254)]), rnd.randrange(1, 80), rnd.randrange(1, 4),
                                        rnd.random() < 0.5, rnd.random() < 0.4)
            n_in = b.n_inputs
        elif kind.startswith("forest"):  # independent parts behind shared inputs: what the stream programs split
            b, n_in = C.build_random_dag(rnd.randrange(1 << 30), n_ops=rnd.randrange(40, 250), panic_free=(kind == "forest"), parts=rnd.randrange(2, 6)), 7
        elif kind == "chains":
            b, n_in = C.build_chain_heavy(rnd.randrange(1 << 30), n_chains=rnd.randrange(4, 20)), 6
        else:
            b, n_in = C.build_random_dag(rnd.randrange(1 << 30), n_ops=rnd.randrange(50, 600), panic_free=(kind == "dag")), 7
        data = b.to_bin()
        B = rnd.choice([1, 2, 3, 17, 64, 65, 200])
        small = 0.3 if kind != "limb" else rnd.choice([0.0, 0.5, 1.0])  # (limb graphs: all field-sized, mixed, all limb-sized operands)
        rows = [[1] + [rnd.randrange(M) if rnd.random() >= small else rnd.choice(EDGE + [rnd.randrange(1 << 16), rnd.randrange(1 << 64), rnd.randrange(1 << 64)]) for _ in range(n_in - 1)] for _ in range(B)]
        inp = cbind.ints_to_array(rows)
        og = cbind.Graph(data)
        want, wst = og.evaluate_batch(inp)
        g = pkg.Graph(data)
        for key in rnd.sample(KEYS if kind != "limb" else LIMB_KEYS, 3) + [0]:
            g.set_tile_width(key)
            got, st = g.calc_witness_batch(inp)
            ok = wst == 0
            if not (np.array_equal(st != 0, wst != 0) and np.array_equal(got[ok], want[ok])):
                bad += 1
                print("MISMATCH seed-index %d kind %s key %#x batch %d" % (s, kind, key, B), flush=True)
        if verbose and s % 50 == 49:
            print("%d graphs, %d mismatches, %.0f s" % (s + 1, bad, time.time() - t0), flush=True)
    return bad


if __name__ == "__main__":
    n = int(os.environ.get("SOAK_SEEDS", "200"))
    bad = run(n, int(os.environ.get("SOAK_BASE", "12345")))
    print("soak done: %d graphs x 4 programs, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)
