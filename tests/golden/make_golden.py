"""Regenerates the golden fixtures in this directory (run from the repo root: python tests/golden/make_golden.py).

Sources of the vectors:
  * reference unit tests restated as data: src/graph.rs:779-883 (Shl, Div, Idiv, Mod, u_gte), src/lib.rs:259-271
    (inputs JSON), src/storage.rs:316-342 (length-delimited node framing) -- values copied as data, no code;
  * the reference's own test input data files test_circuits/*_inputs.json (copied verbatim as data);
  * hand-derived circuit1 graph (94 bytes) and its 204-byte .wtns (sha256 bbb1fc...);
  * generated graphs (tools/graphgen) with expected witnesses computed by oracle/model.py (pure-Python big ints),
    stored as sha256 of the .wtns so that the GPU box can check them without /root/reference.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model  # noqa: E402
import cwc_import  # noqa: E402
C = cwc_import.load().graphgen.circuits

HERE = os.path.dirname(os.path.abspath(__file__))
M = model.M

kat = {
    "reference_unit_vectors": [  # [op, a, b, expected]  (decimal strings)
        ["Shl", "4", "2", "16"],                                                                   # graph.rs:779-785
        ["Div", "2", "3", "7296080957279758407415468581752425029516121466805344781232734728858602831873"],   # :789-791
        ["Div", "6", "2", "3"],                                                                    # :793-795
        ["Div", "7", "2", "10944121435919637611123202872628637544274182200208017171849102093287904247812"],  # :797-799
        ["Idiv", "2", "3", "0"], ["Idiv", "6", "2", "3"], ["Idiv", "7", "2", "3"],                 # :802-815
        ["Mod", "7", "2", "1"], ["Mod", "7", "9", "7"],                                            # :817-826
        ["Geq", "10", "3", "1"], ["Geq", "3", "3", "1"], ["Geq", "2", "3", "0"],                   # :849-858
        ["Geq", str(M - 1), "3", "0"], ["Geq", str(M - 1), str(M - 2), "1"],                       # :860-870
        ["Geq", str(M - 2), str(M - 1), "0"], ["Geq", str(M - 2), str(M - 2), "1"],                # :872-882
    ],
    "inputs_json": {  # lib.rs:259-271
        "text": '{"key1": ["123", "456", 100500], "key2": "789", "key3": 123123}',
        "want": {"key1": ["123", "456", "100500"], "key2": ["789"], "key3": ["123123"]},
    },
    "node_framing": {  # storage.rs:316-342 / SURVEY 8(a) a13 sample encodings (hex of length-delimited records)
        "Input(0)": "020a00", "Input(1)": "040a020801", "Input(2)": "040a020802", "Const(2)": "0712050a030a0102",
        "Mul(2,3)": "06220410021803", "Add(4,0)": "06220408021004",
    },
}
# edge-case vectors of SURVEY 7.6, expected values from the big-int model; "panic" = reference panics
edge = []
E = [0, 1, 2, 3, 31, 32, 33, 64, 129, 192, 253, 254, 255, M - 1, M - 2, M // 2, M // 2 + 1,
     1 << 253, (1 << 64) - 1, 1 << 64, (1 << 128) - 1]
for op in model.DUO:
    if op == "Pow":
        continue
    for a in E:
        for b in E:
            try:
                v = str(model.eval_duo(op, a, b))
            except model.ReferencePanic:
                v = "panic"
            edge.append([op, str(a), str(b), v])
kat["edge_vectors"] = edge
kat["neg_vectors"] = [[str(a), str(model.eval_uno("Neg", a))] for a in E]
with open(os.path.join(HERE, "kat_ops.json"), "w") as f:
    json.dump(kat, f)

# circuit1 fixture
b = C.build_circuit1()
data = b.to_bin()
assert len(data) == 94
open(os.path.join(HERE, "circuit1.bin"), "wb").write(data)
w = model.calc_witness(open(os.path.join(HERE, "circuit1_inputs.json")).read(), data)
assert w == [1, 31817, 105, 303]
wt = model.wtns_from_witness(w)
assert hashlib.sha256(wt).hexdigest() == "bbb1fcd1ba5ef0d68a6bbd526b66d34c1a67d99a06ed0e6a3da5ba288961c72b"
open(os.path.join(HERE, "circuit1.wtns"), "wb").write(wt)

# generated graphs: expected .wtns digests from the Python model
exp = {}
def add(name, builder, inputs_json):
    data = builder.to_bin()
    w = model.calc_witness(inputs_json, data)
    exp[name] = {"bin_sha256": hashlib.sha256(data).hexdigest(), "inputs": inputs_json,
                 "wtns_sha256": hashlib.sha256(model.wtns_from_witness(w)).hexdigest(), "n_witness": len(w)}
add("poseidon1", C.build_poseidon(1), open(os.path.join(HERE, "circuit5_poseidon_inputs.json")).read())
add("gadgets", C.build_gadgets(), json.dumps({"x": "123456789", "y": 0, "arr": ["5", "7", 11, "13"]}))
add("sha256_512", C.build_sha256(512), open(os.path.join(HERE, "circuit8_sha256_512_inputs.json")).read())
add("authv2_class", C.build_authv2_class(), open(os.path.join(HERE, "circuit9_authV2_inputs.json")).read())
for seed in (1, 2, 3):
    add("dag%d" % seed, C.build_random_dag(seed, n_ops=300), json.dumps({"in": [str((seed * 7919 + i) ** 5 % M) for i in range(6)]}))
with open(os.path.join(HERE, "expected_wtns.json"), "w") as f:
    json.dump(exp, f, indent=1)
print("golden fixtures written:", sorted(os.listdir(HERE)))
