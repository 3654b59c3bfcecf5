"""Test-only emulator of a compiled program blob (gwb_graph_export) in Python big ints.

Validates the HOST side of the product -- level scheduling into bundles, liveness slot reuse, operand
encoding, witness references -- on machines without a GPU.  Arithmetic comes from oracle/model.py, so
this checks the compiler, not the HIP kernels (those are checked by the `-m gpu` parity tests)."""
import struct

from oracle import model

REF_CONST = 0x80000000
CREF_TILE = 0x80000000
CTRL_A_TILE, CTRL_B_TILE, CTRL_ACTIVE = 1, 2, 1 << 24
SRC_MEM, SRC_PREV, SRC_LDS = 0, 1, 2
HDR_A_PREV, HDR_A_LDS, HDR_B_PREV, HDR_B_LDS = 1 << 11, 1 << 12, 1 << 13, 1 << 14
RING_BUNDLES, RING_SLOT_BYTES = 8, 2048
SUB_TERN, SUB_INPUT = 33, 34
R_MONT = (1 << 256) % model.M
R_INV = pow(R_MONT, -1, model.M)
HDR_FMT = "<10I25Q"
HDR_SIZE = struct.calcsize(HDR_FMT)
CLASS_NAMES = ["INPUT", "MUL", "LIN", "DIV", "CMPZ", "CMPS", "BIT", "IDIVMOD", "TERN"]


class Blob:
    def __init__(self, data):
        h = struct.unpack_from(HDR_FMT, data, 0)
        (self.magic, self.version, self.T, self.G, self.n_bundles, self.n_slots, self.n_const, self.n_inputs,
         self.n_witness, _res) = h[:10]
        st = h[10:]
        self.stats = dict(n_nodes=st[0], n_op=st[1], n_input_nodes=st[2], n_const=st[3], n_witness=st[4], depth=st[5],
                          class_nodes=st[6:15], class_bundles=st[15:24], algorithmic_bytes_per_set=st[24])
        assert self.magic == 0x47505743 and self.G == 64 // self.T
        pos = HDR_SIZE

        def take(n):
            nonlocal pos
            v = struct.unpack_from("<%dI" % n, data, pos)
            pos += 4 * n
            return v
        self.hdr = take(self.n_bundles)
        self.recs = take(self.n_bundles * self.G * 4)
        self.crefs = take(self.n_bundles * self.G)
        k = take(self.n_const * 8)
        self.consts = [sum(k[8 * i + j] << (32 * j) for j in range(8)) * R_INV % model.M for i in range(self.n_const)]
        assert self.n_const >= 1 and self.consts[-1] == 0  # trailing dummy entry (prefetch target)
        self.witness_refs = take(self.n_witness)


def run(blob: Blob, inputs_row):
    """Evaluate one input set (list of ints) through the format-v3 program; returns (witness ints, status bits).
    Raises on any read of a slot / ring entry that was never written or already overwritten (scheduling bug) and
    checks the wave-uniform header bits against the per-record control bits."""
    T, G = blob.T, blob.G
    slot_bytes = 32 * T
    trash = blob.n_slots * slot_bytes
    slots = {}
    status = 0
    prev = []   # register results of the previous bundle, by node slot
    ring = {}   # (ring slot, node slot) -> (bundle that wrote it, value)

    def mem(off, tile_rel):
        if not tile_rel:
            assert off % slot_bytes == 0 and off // slot_bytes < blob.n_const
            return blob.consts[off // slot_bytes]
        assert off % slot_bytes == 0 and off // slot_bytes < blob.n_slots
        return slots[off // slot_bytes]

    for b in range(blob.n_bundles):
        h = blob.hdr[b]
        cls, cnt = h & 0xF, (h >> 4) & 0x7F
        assert 1 <= cnt <= G
        name = CLASS_NAMES[cls]
        results = []
        seen = {HDR_A_PREV: False, HDR_A_LDS: False, HDR_B_PREV: False, HDR_B_LDS: False}
        for j in range(cnt):
            ctrl, dst, a, bb = blob.recs[(b * G + j) * 4:(b * G + j) * 4 + 4]
            assert ctrl & CTRL_ACTIVE
            sub = (ctrl >> 16) & 0xFF
            if name == "INPUT":
                assert sub == SUB_INPUT
                v = inputs_row[a] % model.M
            else:
                ops = []
                for q, (off, tbit, sh, hp, hl) in enumerate(((a, CTRL_A_TILE, 2, HDR_A_PREV, HDR_A_LDS), (bb, CTRL_B_TILE, 4, HDR_B_PREV, HDR_B_LDS))):
                    src = (ctrl >> sh) & 3
                    if src == SRC_PREV:
                        seen[hp] = True
                        ops.append(prev[j])  # IndexError = the previous bundle had no node in this slot
                    elif src == SRC_LDS:
                        seen[hl] = True
                        rs, rem = divmod(off, RING_SLOT_BYTES)
                        assert rs < RING_BUNDLES and rem % (16 * T) == 0 and rem // (16 * T) < G
                        wb, val = ring[(rs, rem // (16 * T))]
                        assert 1 <= b - wb <= RING_BUNDLES - 1, "ring entry too old or from the future"
                        ops.append(val)
                    else:
                        assert src == SRC_MEM
                        ops.append(mem(off, bool(ctrl & tbit)))
                if name == "TERN":
                    assert sub == SUB_TERN
                    cr = blob.crefs[b * G + j]
                    v = model.eval_tres("TernCond", ops[0], ops[1], mem(cr & 0x7FFFFFFF, bool(cr & CREF_TILE)))
                else:
                    op = model.DUO[sub]
                    expect = {"Mul": "MUL", "Div": "DIV", "Add": "LIN", "Sub": "LIN", "Idiv": "IDIVMOD", "Mod": "IDIVMOD",
                              "Eq": "CMPZ", "Neq": "CMPZ", "Land": "CMPZ", "Lor": "CMPZ", "Lt": "CMPS", "Gt": "CMPS",
                              "Leq": "CMPS", "Geq": "CMPS", "Shl": "BIT", "Shr": "BIT", "Bor": "BIT", "Band": "BIT",
                              "Bxor": "BIT"}[op]
                    assert expect == name, (op, name)
                    try:
                        v = model.eval_duo(op, ops[0], ops[1])
                    except model.ReferencePanic:
                        status |= 1 if op == "Shl" else 2
                        v = 0
            results.append((dst, v))
        if name != "INPUT":  # header bits must describe the records
            for bit, s_ in seen.items():
                assert bool(h & bit) == s_, (b, bit)
        for j in range(cnt, G):  # padding: record 0 without ACTIVE, stored to the trash slot
            r0 = blob.recs[b * G * 4:b * G * 4 + 4]
            rj = blob.recs[(b * G + j) * 4:(b * G + j) * 4 + 4]
            assert rj[0] == r0[0] & ~CTRL_ACTIVE and rj[1] == trash and rj[2:] == r0[2:]
        dsts = [d for d, _ in results if d != trash]
        assert len(set(dsts)) == len(dsts), "two nodes of one bundle share a destination slot"
        for d, v in results:  # all loads of a bundle happen before its stores
            assert d % slot_bytes == 0 and d // slot_bytes <= blob.n_slots
            if d != trash:
                slots[d // slot_bytes] = v
        prev = [v for _, v in results]
        for j, (_, v) in enumerate(results):
            ring[(b % RING_BUNDLES, j)] = (b, v)
        for j in range(cnt, G):
            ring.pop((b % RING_BUNDLES, j), None)  # inactive lanes overwrite the entry with garbage

    def wit(r):
        return blob.consts[r & 0x7FFFFFFF] if r & REF_CONST else slots[r]
    return [wit(r) for r in blob.witness_refs], status
