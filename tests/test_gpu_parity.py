"""Parity tests proper: the HIP path (through the C-ABI) against the oracle on the same seeded inputs, against
the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties.
Bar: bit-exact (integer arithmetic).  Needs a real MI355X: run with `-m gpu`."""
import hashlib
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from oracle import cbind, model
import cwc_import
C = cwc_import.load().graphgen.circuits
import cwc_import
Builder = cwc_import.load().graphgen.builder.Builder

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = model.M
EDGE = [0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 253, 254, 255, 256, M - 1, M - 2, M // 2,
        M // 2 + 1, M // 2 + 2, 1 << 253, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 200, M & ((1 << 253) - 1),
        M ^ (M & ((1 << 253) - 1))]


def _rand_row(rnd, n, small=0.3):
    return [1] + [rnd.randrange(M) if rnd.random() > small else rnd.choice([rnd.randrange(1 << 16), rnd.choice(EDGE)]) for _ in range(n - 1)]


DIVIDER = 0x100  # GWB_TILE_ASYNC_DIVIDER
GROUP = 0x200    # GWB_TILE_GROUP_DIVIDER
TRIPLE = 0x400   # GWB_TILE_TRIPLE_DIVIDER


def _check(pkg, data, rows, tiles=(1, 4, 64, 1 | DIVIDER, 4 | DIVIDER, 4 | GROUP, 2 | TRIPLE)):
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    inp = cbind.ints_to_array(rows)
    want, wst = og.evaluate_batch(inp)
    for tw in tiles:
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        # status: bit0 Shl overflow <-> oracle code 1, bit1 bit-op == r <-> oracle code 2 (reference panics)
        assert np.array_equal(st != 0, wst != 0), "tile %d" % tw
        ok = wst == 0
        assert np.array_equal(got[ok], want[ok]), "tile %d" % tw
    return g


def test_every_op_on_edge_operands(pkg):
    """One graph with every evaluable op applied to the two inputs; all edge pairs as a batch (op-level KAT)."""
    b = Builder()
    (x,) = b.input("x"); (y,) = b.input("y"); (z,) = b.input("z")
    for name in model.DUO:
        if name != "Pow":
            b.signal(b.op(name, x, y))
    b.signal(b.neg(x)); b.signal(b.tern(x, y, z)); b.signal(b.tern(y, x, z))
    data = b.to_bin()
    rows = [[1, a, c, (a * 7 + c) % M] for a in EDGE for c in EDGE]
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    inp = cbind.ints_to_array(rows)
    # a panicking op poisons the whole set in the oracle (evaluate stops), so compare op by op instead
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    for tw in (1, 16, 64, 2 | DIVIDER):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        for r, (row, o, s) in enumerate(zip(rows, got, st)):
            vals = cbind.array_to_ints(o)
            want_bits = 0
            for wi, node_idx in enumerate(wit):
                n = nodes[node_idx]
                try:
                    if n[0] == "Duo":
                        w = model.eval_duo(n[1], row[1], row[2])
                    elif n[0] == "Uno":
                        w = model.eval_uno("Neg", row[1])
                    elif n[0] == "Tres":
                        w = model.eval_tres("TernCond", *[row[nodes[k][1]] for k in n[2:5]])  # operands are Input nodes
                    else:
                        w = 1
                except model.ReferencePanic:
                    want_bits |= 1 if n[1] == "Shl" else 2
                    continue
                if w is not None:
                    assert vals[wi] == w, (tw, n, row[1], row[2])
            assert int(s) == want_bits, (tw, row[1], row[2], int(s), want_bits)
    # reference unit vectors (graph.rs:779-883) through the same graph
    kat = json.load(open(os.path.join(GOLD, "kat_ops.json")))["reference_unit_vectors"]
    g.set_tile_width(0)
    rows = [[1, int(a), int(c), 0] for _, a, c, _ in kat]
    got, st = g.calc_witness_batch(cbind.ints_to_array(rows))
    names = [nodes[i][1] if nodes[i][0] == "Duo" else None for i in wit]
    for (op, a, c, want), o in zip(kat, got):
        assert cbind.array_to_ints(o)[names.index(op)] == int(want), (op, a, c)


def test_terncond_and_inputs_ge_r(pkg):
    b = Builder()
    (x,) = b.input("x"); (y,) = b.input("y"); (z,) = b.input("z")
    b.signal(b.tern(x, y, z)); b.signal(b.add(x, b.const(0)))
    data = b.to_bin()
    rows = [[1, 0, 11, 22], [1, 5, 11, 22], [1, M - 1, 3, 4]]
    _check(pkg, data, rows)
    # inputs >= r are reduced mod r (Fr::new; "[ext] unpinned" in the reference) -- checked against the oracle's reading
    g = pkg.Graph(data)
    got, st = g.calc_witness_batch(cbind.ints_to_array([[1, M, 1, 2], [1, M + 5, 1, 2], [1, (1 << 256) - 1, 1, 2]]))
    assert [cbind.array_to_ints(o)[2] for o in got] == [0, 5, ((1 << 256) - 1) % M]


def test_terncond_third_operand_fresh_from_the_previous_bundles(pkg):
    """TernCond's third operand is loaded in place, from memory, in the iteration that runs the node (kernels.hip C_TERN) --
    also when the bundle right in front of it produced the value, whose store is issued at the top of that same iteration.
    Chains in which every selection's third operand is one, two or three operations old, at every tile width and with
    divider waves, against the oracle (round 3 found that a direct-to-LDS staging load issued in the iteration of the store
    it depends on reads stale data; this is the one place where a load follows its store that closely)."""
    rnd = random.Random(79)
    b = Builder()
    (c,) = b.input("c")
    xs = b.input("x", 6)
    one, k = b.const(1), b.const(12345)
    for lag in (1, 2, 3):
        for x in xs:
            hist = [x]
            for step in range(40):
                y = b.add(b.mul(hist[-1], hist[-1]), one)       # fresh value
                hist.append(y)
                cond = b.op("Lt", b.op("Band", y, b.const(255)), b.const(128 if step % 2 else 64))
                z = b.tern(cond if step % 3 else c, k, hist[-min(lag, len(hist))])  # third operand: `lag` operations old
                hist.append(b.signal(b.add(z, x)))
    rows = [[1] + [rnd.choice([0, 1, rnd.randrange(M)])] + [rnd.randrange(M) for _ in range(6)] for _ in range(70)]
    _check(pkg, b.to_bin(), rows, tiles=(1, 2, 4, 8, 16, 32, 64, 2 | DIVIDER, 8 | DIVIDER, 1 | 0x1000, 4 | 0x800))


def test_golden_fixtures_through_gw_calc_witness(pkg):
    """The reference's drop-in symbol, byte-compared with committed `.wtns` digests (from the big-int model)."""
    data = open(os.path.join(GOLD, "circuit1.bin"), "rb").read()
    wt = pkg.calc_witness_wtns(open(os.path.join(GOLD, "circuit1_inputs.json")).read(), data)
    assert wt == open(os.path.join(GOLD, "circuit1.wtns"), "rb").read()
    assert hashlib.sha256(wt).hexdigest() == "bbb1fcd1ba5ef0d68a6bbd526b66d34c1a67d99a06ed0e6a3da5ba288961c72b"
    exp = json.load(open(os.path.join(GOLD, "expected_wtns.json")))
    builders = {"poseidon1": lambda: C.build_poseidon(1), "gadgets": C.build_gadgets, "sha256_512": lambda: C.build_sha256(512),
                "authv2_class": C.build_authv2_class, "dag1": lambda: C.build_random_dag(1, n_ops=300),
                "dag2": lambda: C.build_random_dag(2, n_ops=300), "dag3": lambda: C.build_random_dag(3, n_ops=300)}
    for name, e in exp.items():
        gdata = builders[name]().to_bin()
        assert hashlib.sha256(gdata).hexdigest() == e["bin_sha256"]
        wt = pkg.calc_witness_wtns(e["inputs"], gdata)
        assert len(wt) == 76 + 32 * e["n_witness"]
        assert hashlib.sha256(wt).hexdigest() == e["wtns_sha256"], name
    assert pkg.calc_witness(open(os.path.join(GOLD, "circuit1_inputs.json")).read(), data) == [1, 31817, 105, 303]


def test_reference_test_circuits_shaped_through_the_c_abi(pkg):
    """circuit{2,3,4,6}-shaped graphs + the reference's input files through gw_calc_witness (GPU) == big-int model."""
    for name, build in (("circuit2", C.build_circuit2), ("circuit3", C.build_circuit3), ("circuit4", C.build_circuit4),
                        ("circuit6_num2bits", C.build_circuit6)):
        data = build().to_bin()
        js = open(os.path.join(GOLD, name + "_inputs.json")).read()
        assert pkg.calc_witness_wtns(js, data) == model.wtns_from_witness(model.calc_witness(js, data)), name


def test_cli_twin(pkg, tmp_path):
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "calc-witness")
    out = tmp_path / "w.wtns"
    r = subprocess.run([exe, os.path.join(GOLD, "circuit1.bin"), os.path.join(GOLD, "circuit1_inputs.json"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Witness generated in:" in r.stdout and "witness saved to" in r.stdout
    assert out.read_bytes() == open(os.path.join(GOLD, "circuit1.wtns"), "rb").read()


@pytest.mark.parametrize("seed", range(10))
def test_random_dag_fuzz(pkg, seed):
    rnd = random.Random(100 + seed)
    b = C.build_random_dag(seed, n_ops=400, panic_free=(seed % 3 != 0))
    _check(pkg, b.to_bin(), [_rand_row(rnd, 7) for _ in range(40)], tiles=(1, 8, 64, 8 | DIVIDER))


def test_compiler_rewrites_on_chain_heavy_graphs(pkg):
    """Exact rewrites of the host compiler (tree-height reduction, riders, request/collect divisions) on the GPU."""
    rnd = random.Random(77)
    for seed in range(4):
        b = C.build_chain_heavy(seed)
        rows = [[1] + [rnd.choice([0, 1, M - 1, rnd.randrange(M)]) for _ in range(5)] for _ in range(37)]
        _check(pkg, b.to_bin(), rows, tiles=(1, 2, 16, 64, 2 | DIVIDER, 8 | DIVIDER, 1 | GROUP, 8 | GROUP))


def test_short_soak_random_graphs_batches_programs(pkg):
    """120 random graphs (every op, panic edges, chain-heavy) x random batch sizes x random program keys."""
    from tools import gpu_soak
    assert gpu_soak.run(120, 4242, verbose=False) == 0


def test_gadgets_and_ragged_batches(pkg):
    rnd = random.Random(5)
    data = C.build_gadgets().to_bin()
    for n in (1, 2, 63, 64, 65, 130):  # tails of every tile width
        _check(pkg, data, [_rand_row(rnd, 7) for _ in range(n)], tiles=(1, 2, 4, 8, 16, 32, 64, 2 | DIVIDER, 32 | DIVIDER, 2 | GROUP, 32 | GROUP))
    g = pkg.Graph(data)
    w, s = g.calc_witness_batch(np.zeros((0, g.n_inputs, 32), dtype=np.uint8))  # empty batch
    assert w.shape == (0, g.n_witness, 32) and s.shape == (0,)


def test_poseidon_model_and_oracle(pkg):
    rnd = random.Random(6)
    for n in (1, 2, 5):
        b = C.build_poseidon(n)
        rows = [[1] + [rnd.randrange(M) for _ in range(n)] for _ in range(33)]
        g = _check(pkg, b.to_bin(), rows, tiles=(1, 16, 64))
        got, _ = g.calc_witness_batch(cbind.ints_to_array(rows[:4]))
        for r, o in zip(rows, got):
            assert cbind.array_to_ints(o[1:2])[0] == C.poseidon_model(r[1:])


def test_sha256_512_against_hashlib_and_oracle(pkg):
    """BASELINE config 3 graph (Shr/Band bit path): every set's 256 output bits equal hashlib (external anchor)."""
    data = C.build_sha256(512).to_bin()
    g = pkg.Graph(data)
    rnd = random.Random(8)
    B = 256
    msgs = [bytes(rnd.getrandbits(8) for _ in range(64)) for _ in range(B)]
    ref_in = json.load(open(os.path.join(GOLD, "circuit8_sha256_512_inputs.json")))["in"]
    msgs[0] = bytes(sum(ref_in[8 * i + k] << (7 - k) for k in range(8)) for i in range(64))
    bits = np.unpackbits(np.frombuffer(b"".join(msgs), dtype=np.uint8).reshape(B, 64), axis=1)
    inp = np.zeros((B, 513, 32), dtype=np.uint8)
    inp[:, 0, 0] = 1
    inp[:, 1:, 0] = bits
    for tw in (1, 8, 64):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        assert not st.any()
        outbits = got[:, 1:257, 0]
        assert not got[:, 1:257, 1:].any()
        want = np.unpackbits(np.frombuffer(b"".join(hashlib.sha256(m).digest() for m in msgs), dtype=np.uint8).reshape(B, 32), axis=1)
        assert np.array_equal(outbits, want), "tile %d" % tw
    og = cbind.Graph(data)
    want_full, _ = og.evaluate_batch(inp[:16])
    assert np.array_equal(got[:16], want_full)


def test_authv2_class_full_size_batch_1024(pkg):
    """BASELINE config 2 at full size (B = 1024, device-resident buffers): a sample of sets is compared with the
    oracle byte for byte; all sets are checked through properties that do not need the oracle:
    witness[0] == 1, inputs echoed at their witness positions, determinism across tile widths (checksum of
    checksums), duplicate input sets give duplicate witnesses."""
    import torch
    b = C.build_authv2_class()
    data = b.to_bin()
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    B = 1024
    rng = np.random.default_rng(2)
    inp = np.frombuffer(rng.bytes(B * g.n_inputs * 32), dtype=np.uint8).reshape(B, g.n_inputs, 32).copy()
    inp[:, :, 31] &= 0x1F
    inp[:, 0, :] = 0
    inp[:, 0, 0] = 1
    ref = C.authv2_reference_inputs()
    row0 = [1]
    for k, n in C.AUTHV2_INPUTS:
        row0 += ref[k]
    inp[0] = cbind.ints_to_array([row0])[0]          # set 0 = the reference's own input file
    inp[B - 1] = inp[17]                              # duplicate set
    d_in = torch.from_numpy(inp).cuda()
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    digests = []
    for tw in (0, 1, 64, 2, 2 | DIVIDER, 4 | GROUP):  # 0 = the library's choice (asynchronous divider at this batch size)
        g.set_tile_width(tw)
        d_out.zero_()
        g.calc_witness_batch_device(d_in, d_out, d_st)
        torch.cuda.synchronize()
        assert int((d_st != 0).sum()) == 0
        out = d_out.cpu().numpy()
        digests.append(hashlib.sha256(out.tobytes()).hexdigest())
    assert len(set(digests)) == 1
    assert (out[:, 0, 0] == 1).all() and not out[:, 0, 1:].any()
    assert np.array_equal(out[B - 1], out[17])
    sample = [0, 1, 2, 3, 511, 1022, 1023]
    want, wst = og.evaluate_batch(inp[sample])
    assert not wst.any() and np.array_equal(out[sample], want)
    exp = json.load(open(os.path.join(GOLD, "expected_wtns.json")))["authv2_class"]
    assert hashlib.sha256(pkg.wtns_from_witness(out[0])).hexdigest() == exp["wtns_sha256"]
    t = g.last_timing()
    assert t["interp_ms"] > 0


def test_workspace_chunking_gives_identical_bytes(pkg, monkeypatch):
    """A batch whose value workspace exceeds the per-descriptor window is spread over several workspace chunks,
    covered by one launch (the kernel picks the chunk per tile) or, beyond the chunk table / CWC_STREAMS, by several
    launches (pipeline.cc); the witnesses must not depend on either."""
    rnd = random.Random(12)
    data = C.build_poseidon(2).to_bin()
    rows = cbind.ints_to_array([_rand_row(rnd, 3, 0) for _ in range(301)])
    g = pkg.Graph(data)
    g.set_tile_width(4)
    a, sa = g.calc_witness_batch(rows)
    assert g.last_timing()["n_launches"] == 1
    launches = {}
    monkeypatch.setenv("CWC_WORKSPACE_GB", "0.0002")  # ~200 KB: a handful of tiles per chunk
    for streams in (None, "1", "3"):
        if streams is None:
            monkeypatch.delenv("CWC_STREAMS", raising=False)
        else:
            monkeypatch.setenv("CWC_STREAMS", streams)
        g2 = pkg.Graph(data)
        g2.set_tile_width(4)
        b2, sb = g2.calc_witness_batch(rows)
        launches[streams] = g2.last_timing()["n_launches"]
        assert np.array_equal(a, b2) and np.array_equal(sa, sb), streams
    assert launches["1"] > launches["3"] >= launches[None] >= 1 and launches["1"] >= 4


def test_replica_from_broadcast_blob(pkg):
    """gwb_graph_export -> gwb_graph_import (what non-zero ranks do after the RCCL broadcast) evaluates identically."""
    rnd = random.Random(13)
    data = C.build_gadgets().to_bin()
    g = pkg.Graph(data)
    rows = cbind.ints_to_array([_rand_row(rnd, 7) for _ in range(20)])
    g.set_tile_width(8)
    a, sa = g.calc_witness_batch(rows)
    rep = pkg.Graph.from_blob(g.export_blob(8))
    b2, sb = rep.calc_witness_batch(rows)
    assert np.array_equal(a, b2) and np.array_equal(sa, sb)
    assert np.array_equal(rep.inputs_from_json('{"x": "9", "arr": [1,2,3,4]}'), g.inputs_from_json('{"x": "9", "arr": [1,2,3,4]}'))


def test_bigint_class_graph(pkg):
    """BASELINE config 5 class (Idiv/Mod/Lt/TernCond-heavy, synthetic) at reduced size."""
    rnd = random.Random(21)
    b = C.build_bigint_class(k=8, rounds=6)
    rows = [_rand_row(rnd, 18, 0.2) for _ in range(40)]
    _check(pkg, b.to_bin(), rows, tiles=(1, 4, 64))


def test_limb_sized_divisions_take_the_short_path(pkg):
    """Idiv / Mod with dividends below 2^128 and divisors below 2^64 in every lane (the operand sizes of 64-bit-limb
    big-integer circuits): the wave-uniform short division, against the C oracle; mixed with a full-width lane."""
    b = Builder()
    (x,) = b.input("x"); (y,) = b.input("y"); (z,) = b.input("z")
    a = b.op("Band", x, b.const((1 << 128) - 1))
    d = b.add(b.op("Band", y, b.const((1 << 64) - 1)), b.const(1))
    dz = b.op("Band", y, b.const((1 << 64) - 1))                      # may be zero: b == 0 -> 0 (graph.rs:112-121)
    for num, den in ((a, d), (a, dz), (d, a), (b.mul(d, d), d)):
        b.signal(b.op("Idiv", num, den)); b.signal(b.op("Mod", num, den))
    data = b.to_bin()
    rnd = random.Random(21)
    small = [0, 1, 2, 3, (1 << 32) - 1, 1 << 32, (1 << 63), (1 << 64) - 1, 1 << 64, (1 << 96) + 5, (1 << 127), (1 << 128) - 1]
    rows = [[1, u, v, 0] for u in small for v in small] + [[1, rnd.randrange(M), rnd.randrange(M), 0] for _ in range(300)]
    _check(pkg, data, rows, tiles=(1, 4, 64))
    # a second graph where one division has full-width operands: the wave falls back to the general routine
    b2 = Builder()
    (x,) = b2.input("x"); (y,) = b2.input("y"); (z,) = b2.input("z")
    a = b2.op("Band", x, b2.const((1 << 128) - 1))
    d = b2.add(b2.op("Band", y, b2.const((1 << 64) - 1)), b2.const(1))
    b2.signal(b2.op("Idiv", a, d)); b2.signal(b2.op("Mod", x, d)); b2.signal(b2.op("Idiv", x, y)); b2.signal(b2.op("Mod", a, y))
    _check(pkg, b2.to_bin(), rows, tiles=(1, 4, 64))


def test_power_of_two_division_rewrite_on_gpu(pkg):
    b = Builder()
    (x,) = b.input("x"); (y,) = b.input("y")
    for k in (0, 1, 31, 32, 33, 64, 128, 200, 253):
        b.signal(b.op("Idiv", x, b.const(1 << k)))
        b.signal(b.op("Mod", x, b.const(1 << k)))
    b.signal(b.op("Idiv", x, y)); b.signal(b.op("Mod", x, y)); b.signal(b.op("Idiv", y, x)); b.signal(b.op("Mod", y, x))
    rows = [[1, a, c] for a in EDGE for c in (0, 1, 3, 1 << 64, M - 1, (1 << 100) + 7)]
    _check(pkg, b.to_bin(), rows, tiles=(1, 16, 64))


def test_host_rows_sliced_staged_and_pinned(pkg, monkeypatch):
    """gwb_calc_witness_batch_host brings the witness rows back in slices through pinned staging with several copy
    threads, or straight into a pinned caller buffer (gwb_host_alloc); every variant must deliver the rows of the
    oracle, whatever the slice size, thread count, or a buffer size that is not a multiple of either."""
    rnd = random.Random(21)
    data = C.build_poseidon(2).to_bin()
    og = cbind.Graph(data)
    rows = cbind.ints_to_array([_rand_row(rnd, 3, 0) for _ in range(3000)])  # 42 MB of witness rows
    want, _ = og.evaluate_batch(rows)
    g = pkg.Graph(data)
    for slice_mb, threads in ((None, None), ("1", "5"), ("3", "1"), ("1", "16")):
        for name, v in (("CWC_COPY_SLICE_MB", slice_mb), ("CWC_COPY_THREADS", threads)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, v)
        for n in (3000, 2999, 37, 1):
            got, st = g.calc_witness_batch(rows[:n])
            assert not st.any() and np.array_equal(got, want[:n]), (slice_mb, threads, n)
    pinned = pkg.pinned_rows((3000, g.n_witness, 32))
    pinned[:] = 0xAA
    got, st = g.calc_witness_batch(rows, out=pinned)
    assert got is pinned and not st.any() and np.array_equal(pinned, want)
    # a view into the middle of a pinned allocation is pinned too
    got, st = g.calc_witness_batch(rows[5:105], out=pinned[100:200])
    assert np.array_equal(pinned[100:200], want[5:105]) and np.array_equal(pinned[:100], want[:100]) and np.array_equal(pinned[200:], want[200:])


@pytest.mark.parametrize("waves", ["1", "4"])
def test_workgroup_shapes_give_identical_witnesses(pkg, monkeypatch, waves):
    """The interpreter launches single-wave workgroups or four-wave ones (four interpreters, or two interpreter +
    divider pairs: CWC_WAVES_PER_WORKGROUP, chosen by tile count otherwise); ragged batches leave interpreter waves
    (and whole pairs) of the last workgroup without a tile.  Same witnesses either way."""
    monkeypatch.setenv("CWC_WAVES_PER_WORKGROUP", waves)
    rnd = random.Random(31)
    data = C.build_gadgets().to_bin()
    og = cbind.Graph(data)
    g = pkg.Graph(data)
    for n in (1, 2, 3, 5, 9, 37):
        rows = cbind.ints_to_array([_rand_row(rnd, 7) for _ in range(n)])
        want, wst = og.evaluate_batch(rows)
        for tw in (1, 2, 4, 1 | DIVIDER, 2 | DIVIDER, 4 | GROUP, 1 | TRIPLE, 4 | TRIPLE):
            g.set_tile_width(tw)
            got, st = g.calc_witness_batch(rows)
            assert np.array_equal(st != 0, wst != 0), (n, tw)
            assert np.array_equal(got[wst == 0], want[wst == 0]), (n, tw)


def test_timing_history_covers_asynchronous_calls(pkg):
    """gwb_timing_history: the handle keeps the HIP events of its most recent launches, so a run of asynchronous calls
    can be timed after the fact (bench.py does); gwb_last_timing keeps describing the last call only."""
    import torch
    data = C.build_poseidon(2).to_bin()
    g = pkg.Graph(data)
    rnd = random.Random(41)
    rows = cbind.ints_to_array([_rand_row(rnd, 3, 0) for _ in range(96)])
    d_in = torch.from_numpy(rows).cuda()
    d_out = torch.empty((96, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(96, dtype=torch.int32, device="cuda")
    g.set_tile_width(4)
    for _ in range(5):
        g.calc_witness_batch_device(d_in, d_out, d_st)  # no synchronization in between
    interp, pack = g.timing_history(3)
    assert len(interp) == 3 and len(pack) == 3 and (interp > 0).all() and (pack > 0).all()
    interp_all, _ = g.timing_history(1000)
    assert 5 <= len(interp_all) <= 256
    tm = g.last_timing()
    assert tm["n_launches"] == 1 and abs(tm["interp_ms"] - float(interp[-1])) < 1e-6
    want, _ = cbind.Graph(data).evaluate_batch(rows)
    assert np.array_equal(d_out.cpu().numpy(), want) and not d_st.any().item()


# ---- round 2: the BASELINE configurations at their named sizes, an outside anchor for the division-heavy block, and the
# ---- callers either side of the path end to end on the GPU -------------------------------------------------
def _synth(kind, n_inputs, batch, seed, first_set=0):
    from tools.synth import synth_inputs
    return synth_inputs(kind, n_inputs, batch, seed, first_set)


def test_config4_share_authv2_class_8192_sets(pkg):
    """BASELINE config 4's per-GPU share (65536 sets over 8 GPUs = 8192 per GPU) at its named size, with the program the
    library chooses for that batch and with T = 4 + group divider: sampled sets against the oracle byte for byte, all
    sets through properties (witness[0] == 1, no error status, the same per-set checksums under both programs, a
    duplicated input set gives a duplicated witness and nothing else does, the first 1024 sets equal to a separate
    1024-set call)."""
    import torch
    from circom_witnesscalc_amd.dist import set_checksums
    b = C.build_authv2_class()
    data = b.to_bin()
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    B = 8192
    inp = _synth("field", g.n_inputs, B, 0xC1C00004)
    inp[B - 1] = inp[17]
    d_in = torch.from_numpy(inp).cuda()
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    sums = []
    for tw in (0, 4 | GROUP):
        g.set_tile_width(tw)
        d_out.zero_()
        g.calc_witness_batch_device(d_in, d_out, d_st)
        torch.cuda.synchronize()
        assert int((d_st != 0).sum()) == 0
        sums.append(set_checksums(d_out).cpu().numpy())
        assert bool((d_out[:, 0, 0] == 1).all()) and not bool(d_out[:, 0, 1:].any())
    assert np.array_equal(sums[0], sums[1])
    assert g.last_timing()["tile_width"] == 4 and g.last_timing()["divider"] == 4
    assert sums[0][B - 1] == sums[0][17] and len(set(sums[0].tolist())) == B - 1   # the duplicate set, and only it, repeats
    sample = [0, 1, 4095, 4096, 6143, 6144, 8190, 8191]
    want, wst = og.evaluate_batch(inp[sample])
    assert not wst.any() and np.array_equal(d_out[torch.tensor(sample, device="cuda")].cpu().numpy(), want)
    g.set_tile_width(0)
    d_small = torch.empty((1024, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    g.calc_witness_batch_device(d_in[:1024], d_small, d_st[:1024])
    torch.cuda.synchronize()
    assert np.array_equal(set_checksums(d_small).cpu().numpy(), sums[0][:1024])


def test_config3_sha256_512_4096_sets_every_digest_against_hashlib(pkg):
    """BASELINE config 3 at its named size: 4096 uniform 512-bit messages, EVERY set's 256 output bits against hashlib."""
    import torch
    data = C.build_sha256(512).to_bin()
    g = pkg.Graph(data)
    B = 4096
    inp = _synth("bits", g.n_inputs, B, 0xC1C00003)
    ref_in = json.load(open(os.path.join(GOLD, "circuit8_sha256_512_inputs.json")))["in"]
    inp[0, 1:, 0] = ref_in                                  # set 0 = the reference's own input file
    d_in = torch.from_numpy(inp).cuda()
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    assert int((d_st != 0).sum()) == 0 and not bool(d_out[:, 1:257, 1:].any())
    msgs = np.packbits(inp[:, 1:513, 0], axis=1)
    want = np.unpackbits(np.frombuffer(b"".join(hashlib.sha256(m.tobytes()).digest() for m in msgs), dtype=np.uint8).reshape(B, 32), axis=1)
    assert np.array_equal(d_out[:, 1:257, 0].cpu().numpy(), want)
    wfull, _ = cbind.Graph(data).evaluate_batch(inp[:8])
    assert np.array_equal(d_out[:8].cpu().numpy(), wfull)


def test_config5_class_bigint_graph_one_million_nodes(pkg):
    """BASELINE config 5's class at over a million nodes (the 10.5 M-node size runs as a tool, tools/gpu_bigint.py): 32
    input sets (the per-GPU share of 256 over 8 GPUs) against the oracle, whole witnesses."""
    b = C.build_bigint_class(k=32, rounds=400)
    data = b.to_bin()
    g = pkg.Graph(data)
    assert g.n_nodes >= 1000000
    rnd = random.Random(55)
    rows = cbind.ints_to_array([[1] + [rnd.randrange(1 << 64) for _ in range(g.n_inputs - 1)] for _ in range(32)])
    want, wst = cbind.Graph(data).evaluate_batch(rows)
    got, st = g.calc_witness_batch(rows)
    assert np.array_equal(st != 0, wst != 0) and np.array_equal(got[wst == 0], want[wst == 0])


def test_gpu_against_plain_integers_and_published_poseidon(pkg):
    """The kernels against references OUTSIDE both restatements (tests/anchors.py), through a graph written by the
    independent `.bin` writer: ordered comparisons as signed integers, Bor / Bxor / Band / Shr / Shl / Idiv / Mod on plain
    Python integers with the reference's reduction and panic rules, field operations mod r -- 3 000 operand pairs through
    four program keys; circomlib's Poseidon (constants from the paper's Grain LFSR) equals the hashes circomlibjs publishes,
    for 64 sets whose first is the published input."""
    import anchors
    data = anchors.ops_graph()
    pairs = anchors.operand_pairs(2, 3000)
    g = pkg.Graph(data)
    inp = cbind.ints_to_array([[1, a, b] for a, b in pairs])
    for tw in (1, 4, 64, 2 | DIVIDER):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        for (a, b), row, s in zip(pairs, got, st):
            vals = cbind.array_to_ints(row)
            bits = 0
            for k, op in enumerate(anchors.OPS):
                v, panics = anchors.plain(op, a, b)
                if panics:
                    bits |= 1 if op == "Shl" else 2
                else:
                    assert vals[1 + k] == v, (tw, op, a, b)
            assert int(s) == bits, (tw, a, b)
    rnd = random.Random(4)
    for ins, want in anchors.POSEIDON_PUBLISHED.items():
        data = C.build_poseidon_circomlib(len(ins)).to_bin()
        g = pkg.Graph(data)
        rows = [[1] + list(ins)] + [[1] + [rnd.randrange(M) for _ in ins] for _ in range(63)]
        for tw in (1, 2, 8):
            g.set_tile_width(tw)
            got, st = g.calc_witness_batch(cbind.ints_to_array(rows))
            assert not st.any() and cbind.array_to_ints(got[0])[1] == want
            assert all(cbind.array_to_ints(got[k])[1] == C.poseidon_model(rows[k][1:], circomlib=True) for k in range(1, 64))


def test_gpu_neg_and_terncond_against_plain_integers(pkg):
    """Neg (graph.rs:188-194: 0 -> 0, else r - a) and TernCond (graph.rs:221-225: a == 0 ? c : b) on the kernels against plain Python
    integers (tests/anchors.py: outside both restatements), through a graph written by the independent `.bin` writer, inputs below r,
    four program keys."""
    import anchors
    data = anchors.uno_tres_graph()
    rows = [(a % anchors.R, b % anchors.R, c % anchors.R) for a, b, c in anchors.uno_tres_inputs(6, 1200)]
    g = pkg.Graph(data)
    inp = cbind.ints_to_array([[1, a, b, c] for a, b, c in rows])
    for tw in (1, 4, 64, 2 | DIVIDER):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        assert not st.any()
        for (a, b, c), row in zip(rows, got):
            w = cbind.array_to_ints(row)
            assert w[4] == (0 if a == 0 else anchors.R - a) and w[5] == (0 if b == 0 else anchors.R - b), (tw, a, b)
            assert w[6] == (c if a == 0 else b) and w[7] == (a if b == 0 else c), (tw, a, b, c)
            assert w == anchors.uno_tres_plain(a, b, c), (tw, a, b, c)


def test_gpu_inputs_at_or_above_r_against_plain_integers(pkg):
    """graph.rs:376 `Fr::new(inputs[i])` on the kernels: raw inputs r, r + 5, 2r, 2^256 - 1, multiples of r ... leave as x mod r, feed
    Neg as x mod r and select like their residue (r as a selector = 0) -- against plain Python integers, four program keys; the
    single-call symbol with the same values as decimal strings in the inputs JSON gives the same `.wtns` rows."""
    import anchors
    data = anchors.uno_tres_graph()
    rows = anchors.uno_tres_inputs(7, 1200)
    assert sum(1 for r in rows if max(r) >= anchors.R) > 400
    g = pkg.Graph(data)
    inp = cbind.ints_to_array([[1, a, b, c] for a, b, c in rows])
    for tw in (1, 4, 64, 2 | DIVIDER):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        assert not st.any()
        for (a, b, c), row in zip(rows, got):
            assert cbind.array_to_ints(row) == anchors.uno_tres_plain(a, b, c), (tw, a, b, c)
    for a, b, c in [(anchors.R, 5, 9), (anchors.R + 5, anchors.R, (1 << 256) - 1), (2 * anchors.R, 2 * anchors.R + 1, 0)]:
        wtns = pkg.calc_witness_wtns(json.dumps({"a": [str(a)], "b": str(b), "c": c if c < (1 << 64) else str(c)}), data)
        body = wtns[76:]
        assert len(body) == 320 and [int.from_bytes(body[32 * k:32 * k + 32], "little") for k in range(10)] == anchors.uno_tres_plain(a, b, c)


def test_scan_bundles_on_the_gpu(pkg):
    """Round 4, class C_SCAN: carry chains and remainder chains of limb arithmetic as loops inside one bundle (pairs of node
    slots, the accumulator moving up the wave by DPP).  Every shift / base width (word-aligned and not, up to 253), chains
    longer than a bundle, forks, operands outside the limb range (the general 256-bit rounds), limb-sized ones (the straight
    rounds), d == 0; 70 input sets through tile widths 1 and 2 with and without divider waves, against the oracle."""
    from test_host_formats import SCAN_CASES, scan_rows
    rnd = random.Random(21)
    steps = 0
    for case in SCAN_CASES + [(64, 64, 64, 1, True, False), (121, 121, 20, 2, False, False), (31, 1, 12, 2, True, True)]:
        data = C.build_limb_chains(*case).to_bin()
        g = pkg.Graph(data)
        rows = scan_rows(rnd, g.n_inputs, 70)
        g = _check(pkg, data, rows, tiles=(1, 2, 1 | DIVIDER, 2 | DIVIDER, 4))
        g.set_tile_width(1)
        steps += g.program_stats(1)["class_bundles"].get("SCAN", 0)
    assert steps > 20
    # the bigint-class graph with inputs of every size (its own masks make limbs of them)
    data = C.build_bigint_class(k=8, rounds=12).to_bin()
    g = pkg.Graph(data)
    _check(pkg, data, scan_rows(rnd, g.n_inputs, 40), tiles=(1, 2, 4))


def test_parallel_scan_forms(pkg, tmp_path):
    """Round 4: chains of 64-bit limbs run all segments of a scan bundle at once (csrc/scan_gfx950.hpp: carry-lookahead over the
    wave; long division as a segmented prefix of affine maps modulo the divisor).  (1) One wave, the helpers alone: 8 000 random
    bundles -- segment starts anywhere, x up to 192 bits, all-ones runs (carries that ripple through the whole wave), divisors
    1, 2, 2^63, 2^64 - 1 -- against the serial recurrences on the host (tools/ubench/scan_par_test.hip, compiled here).
    (2) Through the C-ABI against the oracle: limb chains longer than a bundle with the edge values in every position, tile
    widths 1 and 2; accumulators / divisors that rule the parallel form out (acc >= d, d == 0, a divisor that changes inside a
    chain, x beyond 192 bits) take the serial rounds in the same bundles."""
    exe = tmp_path / "scan_par_test"
    src = os.path.join(ROOT, "tools", "ubench", "scan_par_test.hip")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "circom-witnesscalc_amd", "csrc"), src, "-o", str(exe)],
                   check=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout
    assert "scan_par_test: ok" in out and "8000" not in out and out.count(" 0 carry mismatches, 0 division mismatches") == 2, out
    rnd = random.Random(64)
    B = 1 << 64
    edge = [B - 1, B - 1, B - 1, 0, 1, B - 2, 1 << 63, (1 << 63) + 1]
    for steps, chains in ((70, 1), (33, 2), (5, 3)):
        data = C.build_limb_chains(64, 64, steps, chains, False, False).to_bin()
        g = pkg.Graph(data)
        n = g.n_inputs - 1
        rows = [[1] + [B - 1] * n, [1] + [B - 1] * (n - chains) + [1] * chains, [1] + [0] * n]
        for _ in range(20):
            rows.append([1] + [rnd.choice(edge) if rnd.random() < 0.7 else rnd.randrange(B) for _ in range(n)])
        for _ in range(8):   # wide x / accumulators: 128 .. 192 bits and beyond (the serial 256-bit rounds)
            rows.append([1] + [rnd.randrange(1 << rnd.choice([64, 128, 191, 192, 200, 253])) for _ in range(n)])
        _check(pkg, data, rows, tiles=(1, 2))
    # the bigint-class graph on limbs that are all ones / all zero / edge values (columns of 2^133, carries through every position)
    data = C.build_bigint_class(k=16, rounds=6).to_bin()
    g = pkg.Graph(data)
    n = g.n_inputs - 1
    rows = [[1] + [B - 1] * n, [1] + [0] * n, [1] + [B - 1] * (n - 1) + [0], [1] + [1] * n] + [[1] + [rnd.choice(edge) for _ in range(n)] for _ in range(12)]
    _check(pkg, data, rows, tiles=(1, 2))


def test_convolution_bundles_on_the_gpu(pkg):
    """Round 4: the column sums of a k x k schoolbook limb product as ONE bundle (rewrite.cc detect_convolutions, class SCAN with
    HDR_SCAN_CONV: lane c accumulates x_i y_(c-i) while the y's move up the wave).  Bigint-class graphs with k = 2 .. 32 limbs of
    32 / 64 bits (the 64 x 64 multiply-accumulate rounds), of 100 bits (factors beyond a limb: the field-arithmetic rounds) and
    of 130 bits (products that wrap around r), tile widths 1 and 2 where 2k - 1 columns fit; all-ones limbs (column sums of
    2^133), zeros, uniform field elements as inputs; against the oracle."""
    rnd = random.Random(33)
    from test_host_formats import scan_rows
    B = 1 << 64
    for k, rounds, nb, tiles in ((32, 2, 64, (1,)), (16, 3, 64, (1, 2)), (8, 3, 100, (1, 2)), (5, 2, 130, (1, 2)), (3, 2, 32, (1, 2, 4)), (2, 3, 64, (1, 2))):
        data = C.build_bigint_class(k=k, rounds=rounds, n_bits=nb).to_bin()
        # (the compiler makes convolution bundles of limbs known to fit 64 bits only; CWC_CONV_ANY_WIDTH lifts that for the wide cases)
        os.environ.pop("CWC_CONV_ANY_WIDTH", None)
        if nb > 64:
            os.environ["CWC_CONV_ANY_WIDTH"] = "1"
        os.environ["CWC_CONV_ALWAYS"] = "1"   # (the unfused program competes and wins for the smallest products)
        try:
            g = pkg.Graph(data)
            n = g.n_inputs - 1
            rows = [[1] + [M - 1] * n, [1] + [0] * n, [1] + [B - 1] * n, [1] + [(1 << nb) - 1] * n] + scan_rows(rnd, g.n_inputs, 36)
            g = _check(pkg, data, rows, tiles=tiles)
            st = g.program_stats(tiles[0])
            assert st["n_conv_products"] == k * k * rounds, (k, st["n_conv_products"])
        finally:
            os.environ.pop("CWC_CONV_ANY_WIDTH", None)
            os.environ.pop("CWC_CONV_ALWAYS", None)
    # a limb graph with field divisions: the library's own pick (0), without and with divider waves
    data = C.build_limb_graph_with_divisions(k=8, rounds=3).to_bin()
    g = pkg.Graph(data)
    g = _check(pkg, data, scan_rows(rnd, g.n_inputs, 40), tiles=(0, 1, 2, 1 | DIVIDER, 2 | DIVIDER))
    assert g.program_stats(1)["n_scan_steps"] > 0 and g.program_stats(1 | DIVIDER)["n_scan_steps"] > 0 and g.program_stats(1 | DIVIDER)["class_bundles"].get("DIVREQ", 0) > 0
    # the shapes the recognition has to tell apart (rectangular blocks, squares, shared factors, extra addends, holes, ...)
    for seed in range(100, 130):
        data = C.build_limb_product_variants(seed).to_bin()
        g = pkg.Graph(data)
        os.environ.pop("CWC_CONV_ALWAYS", None)
        if seed % 2:
            os.environ["CWC_CONV_ALWAYS"] = "1"
        try:
            _check(pkg, data, scan_rows(rnd, g.n_inputs, 12), tiles=(1, 2))
        finally:
            os.environ.pop("CWC_CONV_ALWAYS", None)


@pytest.mark.timeout(900)
def test_config5_named_size_ten_million_nodes_all_sets(pkg):
    """BASELINE config 5 at its NAMED size: the 10.5 M-node bigint / long_div-class graph (32 limbs x 4000 rounds), the
    library's own program for 32 sets (the per-GPU share of 256 over 8 GPUs), EVERY set against the oracle as a whole
    witness.  Graphs above two million nodes take compile branches of their own (one schedule, no all-Montgomery
    competitor): this is what covers them on hardware.  Inputs mix uniform field elements (what bench.py feeds) with
    limb-sized and edge values."""
    b = C.build_bigint_class(k=32, rounds=4000)
    data = b.to_bin()
    g = pkg.Graph(data)
    assert g.n_nodes >= 10000000
    rnd = random.Random(5)
    rows = []
    for s in range(32):
        if s % 3 == 0:
            rows.append([1] + [rnd.randrange(M) for _ in range(g.n_inputs - 1)])
        elif s % 3 == 1:
            rows.append([1] + [rnd.randrange(1 << 64) for _ in range(g.n_inputs - 1)])
        else:
            rows.append([1] + [rnd.choice(EDGE + [(1 << 64) - 1, (1 << 64) - 2, 0]) for _ in range(g.n_inputs - 1)])
    inp = cbind.ints_to_array(rows)
    og = cbind.Graph(data)
    _, want, wst = cbind.time_batch_threads(og, inp, min(32, os.cpu_count() or 1))
    key = g.pick_tile_width(32)
    g.set_tile_width(key)
    got, st = g.calc_witness_batch(inp)
    assert np.array_equal(st != 0, wst != 0)
    assert np.array_equal(got[wst == 0], want[wst == 0])
    tm = g.last_timing()
    assert tm["n_bundles"] > 40000 and g.depth > 1000000
    assert g.program_stats(key)["n_conv_products"] == 32 * 32 * 4000 and g.program_stats(key)["n_scan_steps"] > 4000 * 120


def _rsa_rows(rnd, n_inputs, n_bits, count):
    """input sets of the RSA / long_div-class graph: uniform field elements (the graph's masks make registers of them), and
    register-sized edge values -- all-ones and zero registers (borrows and carries that ripple, equal registers in the comparisons)"""
    edge = [0, 1, (1 << n_bits) - 1, (1 << n_bits) - 2, 1 << (n_bits - 1)]
    return [[1] + [rnd.randrange(M) if s % 4 else rnd.choice(edge + [rnd.randrange(1 << n_bits)]) for _ in range(n_inputs - 1)] for s in range(count)]


def test_rsa_long_div_class_one_million_nodes_every_set(pkg):
    """BASELINE config 5's named class (synthetic by necessity): the zk-email RSA / long_div-class graph -- 121-bit
    registers x 17, schoolbook products with carries, long_div by the 17-register modulus digit by digit (short_div estimate,
    long_scalar_mult, long_gt, long_sub), chained modular multiplications -- at 1.1 M nodes, every set against the oracle at tile
    widths 1 and 2, and its q / r registers of the first multiplications against Python's divmod.  The program must run the
    multi-register recurrences as scan bundles (carry chains of 121-bit registers, borrow chains, comparisons)."""
    n, k, muls = 121, 17, 34
    data = C.build_rsa_long_div_class(n=n, k=k, muls=muls).to_bin()
    g = pkg.Graph(data)
    assert g.n_nodes > 1000000
    rnd = random.Random(21)
    rows = _rsa_rows(rnd, g.n_inputs, n, 32)
    inp = cbind.ints_to_array(rows)
    og = cbind.Graph(data)
    _, want, wst = cbind.time_batch_threads(og, inp, min(32, os.cpu_count() or 1))
    assert not wst.any()
    for tw in (1, 2):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        assert not st.any() and np.array_equal(got, want), "tile width %d" % tw
        ps = g.program_stats(tw)
        assert ps["n_scan_steps"] > muls * 3000 and ps["class_bundles"].get("SCAN", 0) * 3 > ps["class_bundles"].get("TERN", 0), ps
    # outside anchor: q, r of the first two multiplications = divmod(a * b, p) on Python integers
    mask = (1 << n) - 1
    for s in (0, 1, 5):
        w = [int.from_bytes(bytes(got[s][i]), "little") for i in range(1 + 2 * k + 2 * (2 * k) * (n + 1))]
        xs, ps_ = [v & mask for v in rows[s][1:1 + k]], [v & mask for v in rows[s][1 + k:1 + 2 * k]]
        ps_[-1] = (rows[s][2 * k] & ((1 << (n - 10)) - 1)) + (1 << (n - 10))
        X, P = sum(v << (n * i) for i, v in enumerate(xs)), sum(v << (n * i) for i, v in enumerate(ps_))
        assert w[1:1 + k] == xs and w[1 + k:1 + 2 * k] == ps_
        pos, acc = 1 + 2 * k, X
        for _m in range(2):
            Q, Rm = divmod(acc * acc, P)
            regs = []
            for _i in range(2 * k):
                regs.append(w[pos])
                assert w[pos + 1:pos + 1 + n] == [(w[pos] >> j) & 1 for j in range(n)]
                pos += n + 1
            assert sum(v << (n * i) for i, v in enumerate(regs[k:])) == Rm and sum(v << (n * i) for i, v in enumerate(regs[:k])) == Q % (1 << (n * k))
            acc = Rm


@pytest.mark.timeout(900)
def test_config5_rsa_named_size_ten_million_nodes_all_sets(pkg):
    """BASELINE config 5 at its named size on the named class: ten million nodes of the RSA / long_div-class graph (121-bit registers x
    17, 310 chained modular multiplications = 18 RSA-65537 exponentiations), the 32 sets of one GPU's shard, every one against
    the oracle, with the program the cost model picks for that batch."""
    n, k, muls = 121, 17, 310
    data = C.build_rsa_long_div_class(n=n, k=k, muls=muls).to_bin()
    g = pkg.Graph(data)
    assert g.n_nodes >= 10000000
    rows = _rsa_rows(random.Random(22), g.n_inputs, n, 32)
    inp = cbind.ints_to_array(rows)
    og = cbind.Graph(data)
    _, want, wst = cbind.time_batch_threads(og, inp, min(32, os.cpu_count() or 1))
    key = g.pick_tile_width(32)
    g.set_tile_width(key)
    got, st = g.calc_witness_batch(inp)
    assert not wst.any() and not st.any() and np.array_equal(got, want)
    assert g.program_stats(key)["n_scan_steps"] > muls * 3000


def test_bit_recurrence_variants_on_the_gpu(pkg, monkeypatch):
    """Borrow chains and most-significant-difference comparisons in every shape the recognition by value has to cope with
    (graphgen.circuits.build_bit_recurrence_variants: four comparison styles, three associations per arm, constant registers,
    arms that are witness elements, dropped and kept last borrows, every pair of result bits, booleans coming in), on registers
    of 2 .. 252 bits, masked and straight from the inputs -- the kernels' register-sized paths and their general ones -- at
    tile widths 1 and 2 (scan bundles), 4 (none) and with the recognition switched off."""
    rnd = random.Random(77)
    n_scan = 0
    for seed in range(60):
        data = C.build_bit_recurrence_variants(seed).to_bin()
        og = cbind.Graph(data)
        rows = [[1] + [rnd.choice([0, 1, rnd.getrandbits(rnd.choice([8, 64, 121, 128, 200])), rnd.randrange(M), M - 1 - rnd.getrandbits(20)]) for _ in range(og.n_inputs - 1)]
                for _s in range(rnd.choice([1, 3, 67]))]
        g = _check(pkg, data, rows, tiles=(1, 2, 4, 1 | DIVIDER))
        n_scan += g.program_stats(1)["n_scan_steps"]
        if seed % 10 == 0:
            monkeypatch.setenv("CWC_NO_BIT_SCANS", "1")
            _check(pkg, data, rows, tiles=(1, 2))
            monkeypatch.delenv("CWC_NO_BIT_SCANS")
    assert n_scan > 300


# BabyJubjub in twisted Edwards form a x^2 + y^2 = 1 + d x^2 y^2 (a = 168700, d = 168696), independent of the generator's
# Montgomery-form gadgets: the unified addition law and double-and-add on Python integers.
_BJ_A, _BJ_D = 168700, 168696
_BJ_BASE8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
             16950150798460657717958625567821834550301663161624707787222815936182638968203)
_BJ_ORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041


def _ed_add(p, q):
    (x1, y1), (x2, y2) = p, q
    t = _BJ_D * x1 * x2 * y1 * y2 % M
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, M) % M, (y1 * y2 - _BJ_A * x1 * x2) * pow(1 - t, -1, M) % M)


def _ed_mul(k, p):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = _ed_add(acc, p)
        p = _ed_add(p, p)
        k >>= 1
    return acc


def _ed_to_mont(p):   # (x, y) -> (u, v) = ((1 + y) / (1 - y), u / x)
    x, y = p
    u = (1 + y) * pow(1 - y, -1, M) % M
    return u, u * pow(x, -1, M) % M


def test_division_heavy_block_against_the_babyjubjub_group_law(pkg):
    """The outside anchor for the Div / inversion-heavy block (what hashlib is for the SHA-256 block): the generator's
    circomlib-shaped curve gadgets -- MontgomeryDouble / MontgomeryAdd ladders (two dependent field divisions per bit) and
    BabyAdd (two divisions) -- evaluated on the GPU must agree with the group law of the curve computed independently in
    Edwards coordinates on Python integers: scalar_mul_any(bits, [r]B) = [(2k+1) r mod l]B and [r1]B + [r2]B = [r1+r2]B,
    for 64 random sets.  (reference semantics: Operation::Div, src/graph.rs:109, vectors :787-800)"""
    assert (_BJ_A * _BJ_BASE8[0] ** 2 + _BJ_BASE8[1] ** 2 - 1 - _BJ_D * _BJ_BASE8[0] ** 2 * _BJ_BASE8[1] ** 2) % M == 0
    assert _ed_mul(_BJ_ORDER, _BJ_BASE8) == (0, 1)
    nbits = 24
    b = Builder()
    bits = b.input("bits", nbits)
    pu, pv = b.input("p", 2)
    q1 = b.input("q1", 2)
    q2 = b.input("q2", 2)
    su, sv = C.scalar_mul_any(b, bits, (pu, pv))
    b.signal(su)
    b.signal(sv)
    ax, ay = C.baby_add(b, tuple(q1), tuple(q2))
    data = b.to_bin()
    g = pkg.Graph(data)
    nodes, wit, _ = model.deserialize_witnesscalc_graph(data)
    rnd = random.Random(2024)
    rows, want = [], []
    for _ in range(64):
        r, r1, r2 = (rnd.randrange(1, _BJ_ORDER) for _ in range(3))
        k = rnd.getrandbits(nbits)
        p_ed, q1_ed, q2_ed = _ed_mul(r, _BJ_BASE8), _ed_mul(r1, _BJ_BASE8), _ed_mul(r2, _BJ_BASE8)
        rows.append([1] + [(k >> i) & 1 for i in range(nbits)] + list(_ed_to_mont(p_ed)) + list(q1_ed) + list(q2_ed))
        want.append(_ed_to_mont(_ed_mul((2 * k + 1) * r % _BJ_ORDER, _BJ_BASE8)) + _ed_mul((r1 + r2) % _BJ_ORDER, _BJ_BASE8))
    inp = cbind.ints_to_array(rows)
    # positions of the four result nodes in the witness: the last two signals of the ladder, the last two of BabyAdd
    w_all, wst = cbind.Graph(data).evaluate_batch(inp)
    assert not wst.any()
    for tw in (0, 1, 2 | DIVIDER, 4 | GROUP, 64):
        g.set_tile_width(tw)
        got, st = g.calc_witness_batch(inp)
        assert not st.any() and np.array_equal(got, w_all), tw
    n_w = got.shape[1]
    # the ladder's (su, sv) were signalled just before BabyAdd's six signals (beta, gamma, delta, tau, xo, yo)
    idx = [n_w - 8, n_w - 7, n_w - 2, n_w - 1]
    for s in range(64):
        vals = cbind.array_to_ints(got[s][idx])
        assert tuple(vals) == tuple(want[s]), s


def test_json_to_wtns_end_to_end_on_the_gpu(pkg, tmp_path):
    """SURVEY 8(f) f3 end to end: NDJSON of 4096 authV2-class input objects -> gwb_inputs_from_json_batch (host threads) ->
    device -> rows back -> gwb_wtns_save_batch; every set's rows against a per-set checksum of a direct run, 32 of the
    files byte-compared with the oracle's `.wtns` (wtns_from_witness of the big-int model's framing)."""
    import torch
    from circom_witnesscalc_amd.dist import set_checksums
    bld = C.build_authv2_class()
    nodes, wit, inputs = bld.finalize()
    data = bld.to_bin()
    g = pkg.Graph(data)
    B = 4096
    src = _synth("field", g.n_inputs, B, 0xC1C00008)
    lines = []
    for r in src:
        vals = [int.from_bytes(r[k].tobytes(), "little") for k in range(g.n_inputs)]
        lines.append(json.dumps({name: [str(v) for v in vals[off:off + n]] for name, (off, n) in inputs.items()}))
    rows = g.inputs_from_json_batch("\n".join(lines))
    assert np.array_equal(rows, src)
    d_in = torch.from_numpy(rows).cuda()
    d_out = torch.empty((B, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    g.calc_witness_batch_device(d_in, d_out, d_st)
    torch.cuda.synchronize()
    assert int((d_st != 0).sum()) == 0
    first = d_out[:32].cpu().numpy()
    pkg.wtns_save_batch(first, str(tmp_path / "w_%04lu.wtns"))
    og = cbind.Graph(data)
    want, _ = og.evaluate_batch(src[:32])
    for i in range(32):
        vals = cbind.array_to_ints(want[i])
        assert (tmp_path / ("w_%04d.wtns" % i)).read_bytes() == model.wtns_from_witness(vals), i
    # every set: the same rows as the host-buffer entry point produces from the same JSON (another route through the library)
    got2, st2 = g.calc_witness_batch(rows[1000:1256])
    assert not st2.any() and np.array_equal(set_checksums(torch.from_numpy(got2)).numpy(), set_checksums(d_out[1000:1256]).cpu().numpy())


@pytest.mark.gpu
def test_load_time_optimiser_on_and_off_on_the_gpu(pkg, monkeypatch):
    """SURVEY 8(f) f2 on the device: the fold-heavy graphs of the emulator test (constant operations over edge values incl.
    the ones the reference panics on, identities, duplicated subexpressions, dead and dead-but-fallible code) evaluated by
    the HIP path with the load-time optimiser (default) and without it (CWC_NO_LOAD_OPTIMIZE) -- both against the oracle
    on the graph as written, status words included."""
    from tests.test_host_formats import fold_heavy_builder
    rnd = random.Random(12)
    for variant in range(4):
        data = fold_heavy_builder(variant, rnd).to_bin()
        rows = cbind.ints_to_array([[1, 0, 0], [1, 5, 5], [1, M - 1, 3], [1, 1 << 200, 2], [1, 3, 200]] + [_rand_row(rnd, 3) for _ in range(45)])
        want, wst = cbind.Graph(data).evaluate_batch(rows)
        ok = wst == 0
        for off in (False, True, "random"):  # default passes / none / the reference's probabilistic passes in front of them
            monkeypatch.delenv("CWC_NO_LOAD_OPTIMIZE", raising=False)
            monkeypatch.delenv("CWC_RANDOM_EVAL", raising=False)
            if off is True:
                monkeypatch.setenv("CWC_NO_LOAD_OPTIMIZE", "1")
            elif off == "random":
                monkeypatch.setenv("CWC_RANDOM_EVAL", "1")
            g = pkg.Graph(data)
            for key in (0, 1, 4, 64, 2 | DIVIDER):
                g.set_tile_width(key)
                got, st = g.calc_witness_batch(rows)
                assert np.array_equal(st != 0, wst != 0), (variant, off, key)
                assert np.array_equal(got[ok], want[ok]), (variant, off, key)
        if variant == 0:
            assert wst.all()  # (the constant operations that panic in the reference report on every row, folded or not)
    monkeypatch.delenv("CWC_NO_LOAD_OPTIMIZE", raising=False)
    monkeypatch.delenv("CWC_RANDOM_EVAL", raising=False)


@pytest.mark.gpu
def test_streaming_json_to_wtns_pipeline(pkg, tmp_path, monkeypatch):
    """The streaming end-to-end entry point (gwb_calc_witness_json_to_wtns): sub-batches parsed on host threads and evaluated
    while the previous one's witness rows leave HBM in slices and writer threads frame them as `.wtns` files.  Ragged sizes
    (a last sub-batch of 3 sets, a last slice of one set), a JSON array instead of NDJSON, sets that panic in the
    reference (status word, NO file -- a stale file of that name is removed; without a status buffer the call fails and
    names the first such set), an unparsable set (the call fails and names it); every file byte-equal to the oracle's
    `.wtns`."""
    bld = C.build_gadgets()
    nodes, wit, inputs = bld.finalize()
    data = bld.to_bin()
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    rnd = random.Random(9)
    B = 131
    rows = [_rand_row(rnd, g.n_inputs) for _ in range(B)]
    for k in (17, 64, 130):  # x | y == r: the reference panics in Bor (graph.rs:701)
        rows[k][1], rows[k][2] = M - 1, 1

    def obj(r):
        return {name: [str(v) for v in r[off:off + n]] for name, (off, n) in inputs.items()}
    want, wst = og.evaluate_batch(cbind.ints_to_array(rows))
    for sub, text in (("64", "\n".join(json.dumps(obj(r)) for r in rows)), ("7", json.dumps([obj(r) for r in rows])), ("1000", "\n\n".join(json.dumps(obj(r)) for r in rows))):
        monkeypatch.setenv("CWC_E2E_SUBBATCH", sub)
        d = tmp_path / ("out" + sub)
        d.mkdir()
        assert wst.any() and not wst.all()
        first_bad = int(np.nonzero(wst)[0][0])
        (d / ("w_%05d.wtns" % (40 + first_bad))).write_bytes(b"stale file of an earlier run")
        st, stats = g.json_to_wtns(text, str(d / "w_%05lu.wtns"), first_index=40)
        assert len(st) == B and stats["n_sets"] == B and np.array_equal(st != 0, wst != 0)
        assert stats["failed_sets"] == int((wst != 0).sum())
        assert sorted(os.listdir(d)) == ["w_%05d.wtns" % (40 + i) for i in range(B) if not wst[i]]
        with pytest.raises(pkg.WitnessCalcError, match="input set %d: .* of %d input sets failed" % (first_bad, B)):
            g.json_to_wtns(text, str(d / "n_%05lu.wtns"), first_index=0, with_status=False)
        assert sorted(f for f in os.listdir(d) if f.startswith("n_")) == ["n_%05d.wtns" % i for i in range(B) if not wst[i]]
        for i in range(B):
            if not wst[i]:
                assert (d / ("w_%05d.wtns" % (40 + i))).read_bytes() == model.wtns_from_witness(cbind.array_to_ints(want[i])), (sub, i)
    bad = [json.dumps(obj(r)) for r in rows[:20]]
    bad[13] = '{"x": -5}'
    with pytest.raises(pkg.WitnessCalcError, match="input set 13"):
        g.json_to_wtns("\n".join(bad), str(tmp_path / "bad_%lu.wtns"))
    with pytest.raises(pkg.WitnessCalcError, match="cannot open"):
        g.json_to_wtns(json.dumps(obj(rows[0])), str(tmp_path / "no_such_dir" / "w_%lu.wtns"))


def test_bin_writer_round_trip_on_the_gpu(pkg):
    """SURVEY 8(f) f1, against the ORACLE: generator -> product writer (C-ABI producer gwb_builder_* ->
    serialize_witnesscalc_graph, storage.rs:137-183) -> bytes equal to the independent pure-Python writer -> loaded by the
    product's reader -> witnesses on the GPU equal to the C oracle's (which parses the Python writer's bytes with its own
    reader); the loaded graph re-serializes to the same bytes."""
    from tools.graphgen.pywriter import serialize_graph
    rnd = random.Random(77)
    for builder in (C.build_gadgets(), C.build_random_dag(5, n_ops=500), C.build_poseidon(3), C.build_chain_heavy(3)):
        data = builder.to_bin()
        independent = serialize_graph(*builder.finalize())
        assert data == independent
        g = pkg.Graph(data)
        assert g.serialize() == independent
        rows = cbind.ints_to_array([_rand_row(rnd, g.n_inputs) for _ in range(40)])
        want, want_st = cbind.Graph(independent).evaluate_batch(rows)
        got, st = g.calc_witness_batch(rows)
        ok = want_st == 0
        assert np.array_equal(st != 0, want_st != 0) and np.array_equal(got[ok], want[ok])


def test_prover_handoff_montgomery_rows_and_event(pkg):
    """gwb_calc_witness_batch_handoff leaves the rows in HBM in Montgomery form (x * 2^256 mod r) and
    records the caller's event behind its last kernel; a consumer stream that waits on the event (no host
    synchronization) reads complete rows, and they are the canonical witnesses times 2^256."""
    import torch
    data = C.build_poseidon(2).to_bin()
    g = pkg.Graph(data)
    rnd = random.Random(4)
    rows = cbind.ints_to_array([_rand_row(rnd, 3, 0) for _ in range(200)])
    want, _ = cbind.Graph(data).evaluate_batch(rows)
    d_in = torch.from_numpy(rows).cuda()
    d_out = torch.zeros((200, g.n_witness, 32), dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(200, dtype=torch.int32, device="cuda")
    producer, consumer = torch.cuda.Stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    torch.cuda.synchronize()
    g.calc_witness_batch_device(d_in, d_out, d_st, stream=producer, montgomery=True, done_event=ev)
    with torch.cuda.stream(consumer):
        consumer.wait_event(ev)
        copy = d_out.clone()
    consumer.synchronize()
    got = copy.cpu().numpy()
    rinv = pow(1 << 256, -1, M)
    for s in (0, 57, 199):
        assert [v * rinv % M for v in cbind.array_to_ints(got[s])] == cbind.array_to_ints(want[s])
    # canonical form through the same entry point is the plain device call
    g.calc_witness_batch_device(d_in, d_out, d_st, stream=producer, done_event=ev)
    ev.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_calls_on_different_streams_execute_in_enqueue_order(pkg):
    """One handle, one value workspace: calls enqueued on different streams must not overlap (the advisor's finding: the
    second call's tiles / constant refill used to race the first).  Alternating batch sizes forces workspace refills."""
    import torch
    data = C.build_gadgets().to_bin()
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    rnd = random.Random(9)
    streams = [torch.cuda.Stream() for _ in range(3)]
    jobs = []
    for i in range(12):
        n = (700, 64, 333)[i % 3]
        rows = cbind.ints_to_array([_rand_row(rnd, 7) for _ in range(n)])
        d_in = torch.from_numpy(rows).cuda()
        jobs.append((rows, d_in, torch.zeros((n, g.n_witness, 32), dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()
    for i, (rows, d_in, d_out, d_st) in enumerate(jobs):
        g.set_tile_width((1, 4, 2 | DIVIDER)[i % 3])
        g.calc_witness_batch_device(d_in, d_out, d_st, stream=streams[i % 3])
    torch.cuda.synchronize()
    for rows, d_in, d_out, d_st in jobs:
        want, wst = og.evaluate_batch(rows)
        ok = wst == 0
        assert np.array_equal(d_st.cpu().numpy() != 0, wst != 0) and np.array_equal(d_out.cpu().numpy()[ok], want[ok])


def test_single_shot_cache_distinguishes_graphs_of_equal_length(pkg):
    """gw_calc_witness keeps compiled graphs keyed by the graph bytes themselves: two `.bin` images of the same length
    that differ in one constant give their own witnesses, in either order."""
    def build(c):
        b = Builder()
        (x,) = b.input("x")
        b.signal(b.add(b.mul(x, x), b.const(c)))
        return b.to_bin()
    d1, d2 = build(1000), build(2000)
    assert len(d1) == len(d2) and d1 != d2
    for _ in range(2):
        assert pkg.calc_witness('{"x": "7"}', d1)[1] == 1049
        assert pkg.calc_witness('{"x": "7"}', d2)[1] == 2049


@pytest.mark.gpu
def test_single_shot_quick_first_call_background_search_and_disk_cache(pkg, tmp_path):
    """gw_calc_witness on a graph it has never seen runs ONE quickly compiled program at once and lets the search over
    candidate programs / schedule variants finish in the background; the refined program replaces the quick one and is
    written to the on-disk cache (CWC_PROGRAM_CACHE), where the next PROCESS finds it.  Every call gives the oracle's
    bytes: the quick program, the refined one, the one imported from the cache; a damaged or truncated cache file is
    ignored (checksum / structural validation of the blob) and rewritten."""
    import subprocess
    import time
    data = C.build_authv2_class(scale=0.12).to_bin()
    gpath, ipath = tmp_path / "g.bin", tmp_path / "in.json"
    gpath.write_bytes(data)
    ins = C.authv2_reference_inputs()
    ipath.write_text(json.dumps({k: [str(x) for x in v] for k, v in ins.items()}))
    nodes, wit, in_map = model.deserialize_witnesscalc_graph(data)
    row = [1] + [0] * (model.get_inputs_size(nodes) - 1)
    for k, v in ins.items():
        off, n = in_map[k]
        row[off:off + n] = v[:n]
    want = model.wtns_from_witness(model.evaluate(nodes, row, wit))
    cache = tmp_path / "cache"
    script = (
        "import sys, time, hashlib\n"
        "sys.path.insert(0, %r)\n"
        "import cwc_import; pkg = cwc_import.load()\n"
        "data = open(%r, 'rb').read(); js = open(%r).read()\n"
        "out = []\n"
        "for i in range(int(sys.argv[1])):\n"
        "    t0 = time.perf_counter(); w = pkg.calc_witness_wtns(js, data); out.append((time.perf_counter() - t0, hashlib.sha256(w).hexdigest()))\n"
        "    time.sleep(float(sys.argv[2]))\n"
        "print('TIMES', ' '.join('%%.4f' %% t for t, _ in out)); print('DIGESTS', ' '.join(sorted(set(d for _, d in out))))\n"
    ) % (ROOT, str(gpath), str(ipath))
    env = dict(os.environ, CWC_PROGRAM_CACHE=str(cache), CWC_DEBUG_CACHE="1")

    def run(calls, pause):
        r = subprocess.run([sys.executable, "-c", script, str(calls), str(pause)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS")][0].split()[1:]
        assert digests == [hashlib.sha256(want).hexdigest()], (digests, r.stderr[-500:])
        return r.stderr

    # process 1: quick program first, the background search finishes within the run, the refined program goes to the cache
    err1 = run(12, 0.5)
    files = list(cache.glob("*.cwcprog"))
    assert "program cache: wrote" in err1 and len(files) == 1, err1[-800:]
    # process 2: served from the cache
    err2 = run(2, 0.0)
    assert "program cache: hit" in err2 and "wrote" not in err2, err2[-800:]
    # a damaged file, then a truncated one: ignored, compiled again, rewritten
    blob = bytearray(files[0].read_bytes())
    blob[len(blob) // 2] ^= 0x40
    files[0].write_bytes(bytes(blob))
    err3 = run(12, 0.5)
    assert "program cache: ignored" in err3 and "program cache: wrote" in err3, err3[-800:]
    files[0].write_bytes(files[0].read_bytes()[:1000])
    err4 = run(12, 0.5)
    assert "program cache: ignored" in err4 and "program cache: wrote" in err4, err4[-800:]
    assert "program cache: hit" in run(1, 0.0)
    # an intact entry of ANOTHER graph (or build) under this graph's file name: the file says what it is for and is refused
    good = files[0].read_bytes()
    assert good[:8] == b"CWCPROG2" and good[8:8 + 64].decode() == hashlib.sha256(data).hexdigest()
    forged = bytearray(good)
    forged[8] = ord("0") if forged[8] != ord("0") else ord("1")   # (header names a different graph image)
    files[0].write_bytes(bytes(forged))
    err5 = run(12, 0.5)
    assert "program cache: ignored" in err5 and "not this graph's" in err5 and "program cache: wrote" in err5, err5[-800:]


STREAMS2, STREAMS4 = 0x800, 0x1000  # GWB_TILE_STREAMS2 / GWB_TILE_STREAMS4


@pytest.mark.gpu
def test_streams_two_and_four_wavefronts_per_tile(pkg):
    """Programs of several streams (the graph's independent parts on wavefronts of their own, posts / waits through the
    tile's sync slot, a divider wave per stream): every key gives the witnesses of the C oracle on the authV2-class graph
    at a ragged batch size, on a graph that is one piece (compiles to one stream) and on fuzzed DAGs with panicking sets;
    the automatic choice for a small batch is a stream program and agrees too."""
    data = C.build_authv2_class(scale=0.15).to_bin()
    g = pkg.Graph(data)
    og = cbind.Graph(data)
    rows = _synth("field", g.n_inputs, 77, 5)
    want, wst = og.evaluate_batch(rows)
    assert not wst.any()
    for key in (1 | STREAMS4, 1 | DIVIDER | STREAMS4, 2 | DIVIDER | STREAMS2, 2 | STREAMS2, 4 | DIVIDER | STREAMS4, 8 | STREAMS4, 16 | DIVIDER | STREAMS2):
        g.set_tile_width(key)
        got, st = g.calc_witness_batch(rows)
        tm = g.last_timing()
        assert tm["streams"] == (4 if key & STREAMS4 else 2) and tm["tile_width"] == key & 0xff, key
        assert not st.any() and np.array_equal(got, want), key
    g.set_tile_width(0)
    got, st = g.calc_witness_batch(rows)
    assert g.last_timing()["streams"] > 1 and not st.any() and np.array_equal(got, want)
    one, st1 = g.calc_witness_batch(rows[:1])  # (the single-call shape: one tile)
    assert not st1.any() and np.array_equal(one, want[:1])
    for seed in range(4):
        data = C.build_random_dag(400 + seed, n_ops=200, panic_free=False, parts=2 + seed).to_bin()
        g2, o2 = pkg.Graph(data), cbind.Graph(data)
        rows2 = cbind.ints_to_array([[1] + [random.Random(seed * 100 + i).randrange(model.M) for _ in range(g2.n_inputs - 1)] for i in range(40)])
        want2, wst2 = o2.evaluate_batch(rows2)
        for key in (1 | STREAMS4, 2 | DIVIDER | STREAMS2, 4 | STREAMS4):
            g2.set_tile_width(key)
            got2, st2 = g2.calc_witness_batch(rows2)
            assert np.array_equal(st2 != 0, wst2 != 0), (seed, key)
            ok = wst2 == 0
            assert np.array_equal(got2[ok], want2[ok]), (seed, key)
