"""Parity tests proper: the HIP path (through the C-ABI) against the oracle on the same seeded inputs, against
the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties.
Bar: bit-exact (integer arithmetic).  Needs a real MI355X: run with `-m gpu`."""
import hashlib
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import cbind, model
from tools.graphgen import circuits as C
from tools.graphgen.builder import Builder

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
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
    launches (runtime.cc); the witnesses must not depend on either."""
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
