import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cwc_import
pkg = cwc_import.load()
import tests.program_emulator as pe
C = pkg.graphgen.circuits
kind = sys.argv[1]
g = pkg.Graph((C.build_rsa_long_div_class(n=121, k=17, muls=4) if kind == "rsa" else C.build_bigint_class(k=32, rounds=40)).to_bin())
for T in (1, 2):
    blob = pe.Blob(g.export_blob(T))
    st = blob.stats
    print(kind, "T", T, "bundles", blob.n_bundles, {k: v for k, v in st.items() if k.startswith("n_") and v and not isinstance(v, (list, tuple))})
