"""Pins the oracles (oracle/model.py, oracle/cwc_oracle.c) against the reference's own unit vectors and
fixtures; the Rust reference itself cannot be built or run here.  CPU only."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import cbind, model
import cwc_import
C = cwc_import.load().graphgen.circuits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat_ops.json")))


def test_reference_unit_vectors_both_oracles():
    # reference src/graph.rs:779-883
    for op, a, b, want in KAT["reference_unit_vectors"]:
        a, b, want = int(a), int(b), int(want)
        assert model.eval_duo(op, a, b) == want, (op, a, b)
        assert cbind.eval_op("Duo", model.DUO_CODE[op], a, b) == want, (op, a, b)


def test_edge_vectors_c_oracle_matches_model():
    # SURVEY 7.6 edge list; expected values were produced by the big-int model (make_golden.py)
    for op, a, b, want in KAT["edge_vectors"]:
        a, b = int(a), int(b)
        try:
            got = str(cbind.eval_op("Duo", model.DUO_CODE[op], a, b))
        except ArithmeticError:
            got = "panic"
        assert got == want, (op, a, b)
    for a, want in KAT["neg_vectors"]:
        assert cbind.eval_op("Uno", 0, int(a)) == int(want)


def test_survey_7_6_named_cases():
    M = model.M
    e = model.eval_duo
    assert e("Div", 5, 0) == 0 and e("Idiv", 5, 0) == 0 and e("Mod", 5, 0) == 0
    assert model.eval_uno("Neg", 0) == 0 and e("Sub", 0, 1) == M - 1 and e("Add", M - 1, 1) == 0
    assert e("Mul", M - 1, M - 1) == 1
    assert e("Lt", M - 1, 3) == 1 and e("Gt", M - 1, 3) == 0
    assert e("Geq", M // 2, M // 2 + 1) == 1  # halfM non-negative, halfM+1 negative
    assert e("Shl", 7, 0) == 7 and e("Shl", 7, 254) == 0 and e("Shl", 7, M - 1) == 0 and e("Shl", 1, 253) == 1 << 253
    assert e("Shl", 1, 255) == 0
    with pytest.raises(model.ReferencePanic):
        e("Shl", M - 1, 1)
    assert e("Shr", 7, 0) == 7 and e("Shr", 7, 254) == 0 and e("Shr", 1 << 253, 253) == 1
    A = M & ((1 << 253) - 1)
    with pytest.raises(model.ReferencePanic):
        e("Bor", A, M ^ A)  # result == r
    assert e("Bor", (1 << 253), (1 << 253) - 1) == ((1 << 254) - 1) - M  # one subtraction
    assert model.eval_tres("TernCond", 0, 11, 22) == 22 and model.eval_tres("TernCond", 5, 11, 22) == 11
    with pytest.raises(model.ReferencePanic):
        e("Pow", 2, 3)


def test_inputs_json_reference_vector():
    v = KAT["inputs_json"]  # reference src/lib.rs:259-271
    got = model.deserialize_inputs(v["text"])
    assert {k: [str(x) for x in xs] for k, xs in got.items()} == v["want"]
    for bad in ['[1]', '{"a": -1}', '{"a": 1.5}', '{"a": [[1]]}', '{"a": true}', '{"a": "0x10"}', '{"a": null}']:
        with pytest.raises(model.InputsError):
            model.deserialize_inputs(bad)


def test_node_framing_vectors():
    # reference src/storage.rs:316-342 style records; bytes from SURVEY 8(a) a13
    from tools.graphgen.pywriter import encode_node, _varint
    nodes = {"Input(0)": ("Input", 0), "Input(1)": ("Input", 1), "Input(2)": ("Input", 2), "Const(2)": ("Const", 2),
             "Mul(2,3)": ("Duo", "Mul", 2, 3), "Add(4,0)": ("Duo", "Add", 4, 0)}
    for name, hx in KAT["node_framing"].items():
        body = encode_node(nodes[name])
        assert (_varint(len(body)) + body).hex() == hx, name
        # and the oracle's reader decodes it back
        assert model._decode_node(bytes.fromhex(hx)[1:])[0] == nodes[name][0]


def test_storage_roundtrip_vector():
    """reference src/storage.rs:420-464 (test_deserialize_inputs): the graph [Input(0), Constant(1), UnoOp(Id, 4),
    Op(Mul, 5, 6), TresOp(TernCond, 7, 8, 9)] with witness [4, 1] and inputs sig1 -> (1, 3), sig2 -> (5, 1) survives
    write -> read, and the trailing u64 points at a metadata record that decodes to the same witness list / input map.
    (Operand indices are not checked at this level -- the reference's reader does not check them either.)"""
    import struct
    from tools.graphgen.pywriter import serialize_graph
    nodes = [("Input", 0), ("Const", 1), ("Uno", "Id", 4), ("Duo", "Mul", 5, 6), ("Tres", "TernCond", 7, 8, 9)]
    wit = [4, 1]
    ins = {"sig1": (1, 3), "sig2": (5, 1)}
    data = serialize_graph(nodes, wit, ins)
    got_nodes, got_wit, got_ins = model.deserialize_witnesscalc_graph(data)
    assert [tuple(n) for n in got_nodes] == nodes and list(got_wit) == wit and dict(got_ins) == ins
    (md_off,) = struct.unpack_from("<Q", data, len(data) - 8)
    # the metadata record alone: varint length + GraphMetadata{1: packed witness, 2: map entries}
    md = data[md_off:len(data) - 8]
    assert md[0] == len(md) - 1
    assert md[1:1 + 4] == bytes([0x0a, 0x02, 0x04, 0x01])                      # witnessSignals = [4, 1], packed
    assert b"sig1" in md and b"sig2" in md
    # header: magic + u64 node count (storage.rs:145, :228)
    assert data[:14] == b"wtns.graph.001" and struct.unpack_from("<Q", data, 14)[0] == 5


def test_circuit1_fixture_end_to_end():
    data = open(os.path.join(GOLD, "circuit1.bin"), "rb").read()
    assert len(data) == 94 and C.build_circuit1().to_bin() == data
    inputs = open(os.path.join(GOLD, "circuit1_inputs.json")).read()
    w = model.calc_witness(inputs, data)
    assert w == [1, 31817, 105, 303]
    wt = model.wtns_from_witness(w)
    assert wt == open(os.path.join(GOLD, "circuit1.wtns"), "rb").read()
    assert hashlib.sha256(wt).hexdigest() == "bbb1fcd1ba5ef0d68a6bbd526b66d34c1a67d99a06ed0e6a3da5ba288961c72b"
    g = cbind.Graph(data)
    out, st = g.evaluate_batch(cbind.ints_to_array([[1, 105, 303]]))
    assert st[0] == 0 and cbind.wtns_from_witness(out[0]) == wt


def test_expected_wtns_digests_c_oracle():
    exp = json.load(open(os.path.join(GOLD, "expected_wtns.json")))
    builders = {"poseidon1": lambda: C.build_poseidon(1), "gadgets": C.build_gadgets, "sha256_512": lambda: C.build_sha256(512),
                "authv2_class": C.build_authv2_class, "dag1": lambda: C.build_random_dag(1, n_ops=300),
                "dag2": lambda: C.build_random_dag(2, n_ops=300), "dag3": lambda: C.build_random_dag(3, n_ops=300)}
    for name, e in exp.items():
        data = builders[name]().to_bin()
        assert hashlib.sha256(data).hexdigest() == e["bin_sha256"], name  # generators are deterministic
        nodes, wit, ins = model.deserialize_witnesscalc_graph(data)
        buf = model.get_inputs_buffer(model.get_inputs_size(nodes))
        model.populate_inputs(model.deserialize_inputs(e["inputs"]), ins, buf)
        g = cbind.Graph(data)
        out, st = g.evaluate_batch(cbind.ints_to_array([buf]))
        assert st[0] == 0
        assert hashlib.sha256(cbind.wtns_from_witness(out[0])).hexdigest() == e["wtns_sha256"], name


def test_sha256_graph_matches_hashlib():
    """External anchor: the generated SHA-256(512) graph evaluated by the oracle equals hashlib."""
    data = C.build_sha256(512).to_bin()
    g = cbind.Graph(data)
    rnd = random.Random(4)
    msgs = [bytes(64), bytes(range(64)), bytes([255] * 64)] + [bytes(rnd.getrandbits(8) for _ in range(64)) for _ in range(3)]
    ref_in = json.load(open(os.path.join(GOLD, "circuit8_sha256_512_inputs.json")))["in"]
    msgs.append(bytes(sum(ref_in[8 * i + k] << (7 - k) for k in range(8)) for i in range(64)))
    rows = [[1] + [(m[i // 8] >> (7 - i % 8)) & 1 for i in range(512)] for m in msgs]
    out, st = g.evaluate_batch(cbind.ints_to_array(rows))
    assert not st.any()
    for m, o in zip(msgs, out):
        bits = cbind.array_to_ints(o[1:257])
        dig = hashlib.sha256(m).digest()
        assert bits == [(dig[i // 8] >> (7 - i % 8)) & 1 for i in range(256)]
    # the pure-Python model agrees on one message
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    assert model.evaluate(nodes, rows[1], wit) == cbind.array_to_ints(out[1])


def test_random_dags_c_oracle_vs_model():
    rnd = random.Random(9)
    for seed in range(12):
        b = C.build_random_dag(seed, n_ops=250, panic_free=(seed % 3 != 0))
        data = b.to_bin()
        nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
        g = cbind.Graph(data)
        rows = [[1] + [rnd.randrange(model.M) if rnd.random() < 0.6 else rnd.randrange(1 << 12) for _ in range(6)] for _ in range(4)]
        out, st = g.evaluate_batch(cbind.ints_to_array(rows))
        for r, o, s in zip(rows, out, st):
            try:
                want = model.evaluate(nodes, r, wit)
            except model.ReferencePanic:
                assert s in (1, 2)
                continue
            assert s == 0 and cbind.array_to_ints(o) == want


def test_bigint_class_graph_oracles_agree():
    rnd = random.Random(31)
    b = C.build_bigint_class(k=4, rounds=3)
    data = b.to_bin()
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    g = cbind.Graph(data)
    rows = [[1] + [rnd.randrange(model.M) if rnd.random() < 0.7 else rnd.randrange(1 << 40) for _ in range(9)] for _ in range(6)]
    out, st = g.evaluate_batch(cbind.ints_to_array(rows))
    assert not st.any()
    for r, o in zip(rows, out):
        assert model.evaluate(nodes, r, wit) == cbind.array_to_ints(o)


REF_SHAPED = {  # reference test circuits rebuilt by tools/graphgen, evaluated on the reference's own input files
    "circuit2": (C.build_circuit2, lambda a, b: (1 if a == 0 else 0) * b + 2),
    "circuit3": (C.build_circuit3, lambda a, b: a * b + 3),
    "circuit4": (C.build_circuit4, lambda a, b: (a & 1) * ((a >> 1) & 1) + b),
    "circuit6_num2bits": (C.build_circuit6, lambda a, b: (a >> 16) & ((1 << 216) - 1)),
}


@pytest.mark.parametrize("name", sorted(REF_SHAPED))
def test_reference_test_circuits_shaped(name):
    """test_circuits/circuit{2,3,4,6}.circom-shaped graphs with the reference's own `*_inputs.json`: big-int model ==
    C oracle == the circuit's arithmetic done by hand (plus a few more inputs, zero and field-edge values included)."""
    build, expect = REF_SHAPED[name]
    data = build().to_bin()
    js = open(os.path.join(GOLD, name + "_inputs.json")).read()
    d = json.loads(js)
    a, b = int(d["a"][0]), int(d["b"][0]) if "b" in d else 0
    w = model.calc_witness(js, data)
    assert w[0] == 1 and w[1] == expect(a, b) % model.M and w[2] == a
    og = cbind.Graph(data)
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    rows = [[1, a, b][:og.n_inputs]] + [[1, x, y][:og.n_inputs] for x in (0, 1, 2, 3, model.M - 1, 1 << 200) for y in (0, 7, model.M - 1)]
    got, st = og.evaluate_batch(cbind.ints_to_array(rows))
    for row, o in zip(rows, got):
        vals = cbind.array_to_ints(o)
        assert vals == model.evaluate(nodes, row, wit)
        assert vals[1] == expect(row[1], row[2] if len(row) > 2 else 0) % model.M
    assert not st.any()


def test_oracles_against_plain_integer_semantics_and_published_poseidon():
    """Anchors OUTSIDE both restatements (tests/anchors.py): every comparison, bit operation, shift, integer division and
    field operation of a graph written by the independent `.bin` writer against plain Python integers -- ordered
    comparisons as signed integers on [-(r-1)/2, (r-1)/2], Bor / Bxor / Band / Shr / Shl on the integers with the
    reference's reduction and panic rules -- on 3 000 operand pairs (edges of the sign boundary, of r, of the 254-bit
    rule); and circomlib's Poseidon with constants derived here from the Poseidon paper's Grain LFSR against the
    hashes circomlibjs publishes.  Both oracles (big-int model, C port) must agree with these."""
    import anchors
    from oracle import cbind
    data = anchors.ops_graph()
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    og = cbind.Graph(data)
    pairs = anchors.operand_pairs(1, 3000)
    out, st = og.evaluate_batch(cbind.ints_to_array([[1, a, b] for a, b in pairs]))
    n_panic = 0
    for (a, b), row, s in zip(pairs, out, st):
        want = [anchors.plain(op, a, b) for op in anchors.OPS]
        panics = any(p for _, p in want)
        n_panic += panics
        assert (s != 0) == panics, (a, b)
        for k, (op, (v, p)) in enumerate(zip(anchors.OPS, want)):
            if p:
                with pytest.raises(model.ReferencePanic):
                    model.eval_duo(op, a, b)
                continue
            assert model.eval_duo(op, a, b) == v, (op, a, b)
            if not panics:
                assert cbind.array_to_ints(row)[1 + k] == v, (op, a, b)
    assert n_panic >= 5
    import cwc_import
    C = cwc_import.load().graphgen.circuits
    for ins, want in anchors.POSEIDON_PUBLISHED.items():
        data = C.build_poseidon_circomlib(len(ins)).to_bin()
        nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
        assert model.evaluate(nodes, [1] + list(ins), wit)[1] == want
        out, st = cbind.Graph(data).evaluate_batch(cbind.ints_to_array([[1] + list(ins)]))
        assert st[0] == 0 and cbind.array_to_ints(out[0])[1] == want


def test_oracles_neg_terncond_and_inputs_above_r_against_plain_integers():
    """The three semantics that had no anchor outside the two restatements (tests/anchors.py, round 6): Neg (graph.rs:188-194: 0 -> 0,
    else r - a), TernCond (graph.rs:221-225: a == 0 ? c : b) and the reduction of inputs at or above r (graph.rs:376: r, r + 5, 2r,
    2^256 - 1 ... stand for x mod r; a selector that arrives as r selects like 0) -- a graph written by the independent `.bin` writer,
    raw input triples against plain Python integers, on both oracles."""
    import anchors
    from oracle import cbind
    data = anchors.uno_tres_graph()
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    og = cbind.Graph(data)
    rows = anchors.uno_tres_inputs(5, 1500)
    assert any(a >= anchors.R for a, _, _ in rows) and any(a % anchors.R == 0 and a for a, _, _ in rows)
    out, st = og.evaluate_batch(cbind.ints_to_array([[1, a, b, c] for a, b, c in rows]))
    assert not st.any()
    for (a, b, c), row in zip(rows, out):
        want = anchors.uno_tres_plain(a, b, c)
        assert cbind.array_to_ints(row) == want, (a, b, c)
    for a, b, c in rows[::7]:
        assert model.evaluate(nodes, [1, a, b, c], wit) == anchors.uno_tres_plain(a, b, c), (a, b, c)
    # known answers spelled out (not through the helper): Neg 0 = 0, Neg 1 = r - 1; r as a selector takes the else arm; 2^256 - 1 mod r
    assert anchors.uno_tres_plain(anchors.R, 5, 9)[1:8] == [0, 5, 9, 0, anchors.R - 5, 9, 9]
    assert anchors.uno_tres_plain(1, 0, 9)[4:8] == [anchors.R - 1, 0, 0, 1]
    assert anchors.input_plain((1 << 256) - 1) == 6350874878119819312338956282401532410528162663560392320966563075034087161850
