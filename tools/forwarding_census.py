"""Where do the operands of a compiled program's bundles come from -- and for how many bundles could the wait for the LDS reads be
avoided by forwarding the dependent operand in registers and reading the other one early (round-5 review, item 1)?

Host-only: compiles the program (authV2-class graph, the headline key T = 2 + divider wave by default) and walks the exported blob.
Per class: bundles that read a result of the bundle right in front of them ("dep"); operand sources (ring cell 1..4 bundles old, the
tile's constants, other memory values through the STAGE cells); and the census the question turns on -- a bundle's LDS wait can only
go away if EVERY operand of EVERY active lane is either (a) the same lane's own previous result (register forwarding: producer and
consumer in the same node slot, at most one consumer per producer) or (b) in LDS before the previous bundle's arithmetic starts
(a ring cell two to four bundles old).  Operands that arrive through a STAGE cell (constants, older values) land with the counted
vmcnt wait at the top of the bundle's own iteration: they cannot be read a bundle early without a third STAGE buffer.

    python tools/forwarding_census.py [authv2|sha|rsa|bigint] [key]   > profiles/r06_forwarding_census.txt
"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cwc_import
import program_emulator as pe


def main():
    pkg = cwc_import.load()
    C = pkg.graphgen.circuits
    kind = sys.argv[1] if len(sys.argv) > 1 else "authv2"
    key = int(sys.argv[2]) if len(sys.argv) > 2 else 258
    b = (C.build_authv2_class() if kind == "authv2" else C.build_sha256(512) if kind == "sha" else
         C.build_rsa_long_div_class(n=121, k=17, muls=4) if kind == "rsa" else C.build_bigint_class(k=32, rounds=40))
    g = pkg.Graph(b.to_bin())
    blob = pe.Blob(g.export_blob(key))
    T, G = blob.T, blob.G
    print("graph %s, program key %d: tile width %d, %d bundles, %d stream(s)" % (kind, key, T, blob.n_bundles, blob.n_streams))
    per = collections.defaultdict(collections.Counter)
    for bi in range(blob.n_bundles):
        h = blob.hdr[bi]
        cls = pe.CLASS_NAMES[h & 15]
        rep = 4 if cls in ("MULQ", "MULF") else 1
        c = per[cls]
        c["bundles"] += 1
        dep = stage = False
        producers = collections.Counter()
        two_producers = unaligned = 0
        for js in range(0, G, rep):
            r = blob.recs[(bi * G + js) * 4:(bi * G + js) * 4 + 4]
            if not (r[2] & 8):
                continue
            c["nodes"] += 1
            mine = set()
            for q, lds in enumerate((r[3] & 0xffff, r[3] >> 16)):
                if cls == "BIT" and q == 1 and (r[2] & 7) == 5:
                    continue
                if lds < pe.LDS_STAGE_OFF:
                    d = (bi - lds // 2048) % 4 or 4
                    c["operands from the ring, %d bundle(s) old" % d] += 1
                    if d == 1:
                        dep = True
                        pos = (lds % 1024) // (16 * T)
                        mine.add(pos)
                        unaligned += pos != js
                elif r[q] == pe.OFF_NOWHERE:
                    c["operands unused"] += 1
                elif r[q] < blob.n_const * 32 * T:
                    c["operands: constants (STAGE)"] += 1
                    stage = True
                else:
                    c["operands: older values from memory (STAGE)"] += 1
                    stage = True
            two_producers += len(mine) == 2
            for x in mine:
                producers[x] += 1
        c["bundles that read the bundle right in front of them"] += dep
        c["bundles with a STAGE operand (cannot be read a bundle early)"] += stage
        if not stage:
            alignable = two_producers == 0 and all(v == 1 for v in producers.values())
            c["bundles without STAGE operands"] += 1
            c["... whose dependent operands could all sit in the consumer's own lane"] += alignable
            c["... and do so in the program as compiled"] += alignable and unaligned == 0
    tot = collections.Counter()
    for cls in sorted(per, key=lambda k: -per[k]["bundles"]):
        c = per[cls]
        print("%-8s %6d bundles, %.1f nodes per bundle" % (cls, c["bundles"], c["nodes"] / max(1, c["bundles"])))
        for k in sorted(c):
            if k not in ("bundles", "nodes"):
                print("    %-78s %8d" % (k, c[k]))
        for k in c:
            tot[k] += c[k]
    n = tot["bundles"]
    ok = tot["... whose dependent operands could all sit in the consumer's own lane"]
    print("all classes: %d bundles; %d (%.1f %%) read the bundle in front of them; %d (%.1f %%) have a STAGE operand; the wait for the LDS reads could be "
          "forwarded away in %d (%.1f %%)" % (n, tot["bundles that read the bundle right in front of them"], 100.0 * tot["bundles that read the bundle right in front of them"] / n,
                                             tot["bundles with a STAGE operand (cannot be read a bundle early)"], 100.0 * tot["bundles with a STAGE operand (cannot be read a bundle early)"] / n,
                                             ok, 100.0 * ok / n))


if __name__ == "__main__":
    main()
