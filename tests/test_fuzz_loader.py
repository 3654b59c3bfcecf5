"""Robustness of the host-side loaders: corrupted `.bin` images and hostile inputs JSON must produce an error
(or a valid parse), never a crash -- the reference panics across the FFI boundary in these cases
(src/lib.rs:130,196).  CPU only."""
import random

import pytest

import cwc_import

C = cwc_import.load().graphgen.circuits


def _mutations(data, rnd, n):
    data = bytearray(data)
    for _ in range(n):
        d = bytearray(data)
        kind = rnd.randrange(6)
        if kind == 0:   # flip bytes
            for _ in range(rnd.randrange(1, 8)):
                d[rnd.randrange(len(d))] = rnd.randrange(256)
        elif kind == 1:  # truncate
            d = d[:rnd.randrange(len(d))]
        elif kind == 2:  # huge varint / length
            p = rnd.randrange(14, len(d))
            d[p:p + 1] = b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\x7f"
        elif kind == 3:  # node count lies
            d[14:22] = rnd.choice([2 ** 63, 2 ** 32, len(d), 0]).to_bytes(8, "little")
        elif kind == 4:  # splice random junk
            p = rnd.randrange(len(d))
            d[p:p] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))
        else:            # operand index garbage: set many bytes high
            for _ in range(20):
                d[rnd.randrange(22, len(d))] |= 0x80
        yield bytes(d)


def test_corrupted_graphs_never_crash(pkg):
    rnd = random.Random(77)
    ok = bad = 0
    for builder in (C.build_circuit1(), C.build_gadgets(), C.build_random_dag(3, n_ops=120)):
        base = builder.to_bin()
        for m in _mutations(base, rnd, 400):
            try:
                g = pkg.Graph(m)
                # whatever parsed must also compile for a few tile widths (validation happens there)
                for t in (1, 8, 64):
                    g.export_blob(t)
                ok += 1
            except pkg.WitnessCalcError:
                bad += 1
    assert bad > 100 and ok + bad == 1200


def test_hostile_json_never_crashes(pkg):
    g = pkg.Graph(C.build_gadgets().to_bin())
    rnd = random.Random(5)
    base = '{"x": "123", "y": 7, "arr": ["1", 2, "3", 4]}'
    cases = ['{"x": ' + "[" * 300 + "]" * 300 + "}", '{"x": "' + "9" * 5000 + '"}', '{"x": 1' + "0" * 400 + "}",
             '{"\\ud800": 1}', '{"x": "\\u00e9"}', "{" * 1000, '{"x": 1}{"y": 2}', "", "   ", "null", '{"x": 1e999}',
             '{"x": -0}', '{"arr": []}', '{"x": "1", "x": [1]}']
    for _ in range(300):
        b = bytearray(base.encode())
        for _ in range(rnd.randrange(1, 5)):
            b[rnd.randrange(len(b))] = rnd.choice(b'{}[]",:0123456789 \\eE-+.\x00\xff')
        cases.append(b.decode("latin-1"))
    for txt in cases:
        try:
            g.inputs_from_json(txt.encode("latin-1", "replace").replace(b"\x00", b" "))
        except pkg.WitnessCalcError:
            pass


def test_corrupted_program_blobs_are_rejected(pkg):
    """gwb_graph_import is what the ranks of a node feed with the broadcast program: a truncated, padded, bit-flipped or
    internally inconsistent blob must be refused before anything is uploaded or launched (checksum + structural
    validation of every offset / index / LDS address, validate_program in compile.cc).  On this CPU-only machine a
    VALID blob gets as far as the device check and fails there; everything else must fail earlier with its own message."""
    import struct
    rnd = random.Random(5)
    g = pkg.Graph(C.build_gadgets().to_bin())
    for key in (1, 4, 64, 2 | 0x100):
        blob = g.export_blob(key)
        with pytest.raises(pkg.WitnessCalcError, match="no HIP device"):
            pkg.Graph.from_blob(blob)
        # the advisor's crash: a 16-byte blob whose trailer says "program length 0"
        for tiny in (b"\0" * 16, b"\0" * 24, b"\0" * 32, blob[:7], blob[-24:]):
            with pytest.raises(pkg.WitnessCalcError, match="bad blob|bad program blob"):
                pkg.Graph.from_blob(tiny)
        # transport damage: caught by the checksum
        for _ in range(150):
            d = bytearray(blob)
            kind = rnd.randrange(3)
            if kind == 0:
                d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
            elif kind == 1:
                d = d[:rnd.randrange(len(d))]
            else:
                d += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 64)))
            with pytest.raises(pkg.WitnessCalcError, match="bad blob|bad program blob"):
                pkg.Graph.from_blob(bytes(d))
        # a hostile sender: consistent checksum, damaged contents -> the structural validation has to catch it
        from tests.program_emulator import blob_checksum as refnv
        body, (exact, padded, _) = bytearray(blob[:-24]), struct.unpack("<3Q", blob[-24:])
        hits = 0
        for _ in range(120):
            d = bytearray(body)
            pos = rnd.randrange(0, exact - 4) & ~3
            d[pos:pos + 4] = struct.pack("<I", rnd.choice([0xFFFFFFFF, 0x7FFFFFF0, 1 << 20, rnd.getrandbits(32)]))
            evil = bytes(d) + struct.pack("<3Q", exact, padded, refnv(d))
            try:
                pkg.Graph.from_blob(evil)
            except pkg.WitnessCalcError as e:
                hits += "no HIP device" not in str(e)   # (a change in a statistics field or a constant is harmless and passes)
        assert hits >= 30, hits  # (the rest landed in constants, statistics or on other valid offsets: memory-safe)


def test_oversized_input_map_is_refused(pkg):
    """An input-map entry far beyond the Input nodes (offset 0xFFFFFFFF, the advisor's case) used to wrap the 32-bit
    input count, after which the kernel would have read beyond the caller's rows; such a graph is refused at load."""
    import cwc_import
    from tools.graphgen.pywriter import serialize_graph
    Builder = cwc_import.load().graphgen.builder.Builder
    b = Builder()
    a = b.input("a", 1)[0]
    b.signal(b.mul(a, a))
    nodes, wit, inputs = b.finalize()
    assert pkg.Graph(serialize_graph(nodes, wit, inputs)).n_inputs == 2
    for off, n in ((0xFFFFFFFF, 2), (1 << 27, 1), (0xFFFFFFF0, 0x20)):
        with pytest.raises(pkg.WitnessCalcError, match="too large"):
            pkg.Graph(serialize_graph(nodes, wit, {"a": (off, n)}))
    # a sane entry beyond the Input run only grows the buffer (reference: panic at lib.rs:158-161; here: reported size)
    assert pkg.Graph(serialize_graph(nodes, wit, {"a": (1, 1), "pad": (5, 3)})).n_inputs == 8


def test_stream_programs_are_validated_for_every_stream(pkg):
    """validate_program used to compare stream_cref_first with the running row count before moving on to the stream a
    bundle belongs to, so the check only ever fired for stream 0; and the two bundles behind a wait have their staging
    loads issued in front of it, so they must be idle.  A blob that is wrong there (checksum recomputed: a hostile or
    buggy sender) is refused before anything reaches the device."""
    import struct
    from tests import program_emulator as E

    refnv = E.blob_checksum
    g = pkg.Graph(C.build_random_dag(11, n_ops=150, parts=4).to_bin())
    blob = g.export_blob(1 | 0x1000)
    bl = E.Blob(blob)
    assert bl.n_streams > 1
    body, (exact, padded, _) = bytearray(blob[:-24]), struct.unpack("<3Q", blob[-24:])
    with pytest.raises(pkg.WitnessCalcError, match="no HIP device"):
        pkg.Graph.from_blob(blob)
    tested = 0
    for s in range(1, bl.n_streams):
        if not bl.stream_count[s]:
            continue
        # (a) the stream's first third-operand row
        d = bytearray(body)
        pos = 4 * (24 + s)
        struct.pack_into("<I", d, pos, bl.stream_cref_first[s] + 1)
        with pytest.raises(pkg.WitnessCalcError, match="third-operand rows of stream %d" % s):
            pkg.Graph.from_blob(bytes(d) + struct.pack("<3Q", exact, padded, refnv(d)))
        # (b) work in the bundle right behind the wait: node count 1 in an idle bundle's header
        d = bytearray(body)
        b1 = bl.stream_first[s] + 1
        assert (bl.hdr[b1] >> 4) & 0x7F == 0
        struct.pack_into("<I", d, E.HDR_SIZE + 4 * b1, bl.hdr[b1] | (1 << 4))
        with pytest.raises(pkg.WitnessCalcError, match="behind a wait|bundle %d" % b1):
            pkg.Graph.from_blob(bytes(d) + struct.pack("<3Q", exact, padded, refnv(d)))
        tested += 1
    assert tested >= 1
