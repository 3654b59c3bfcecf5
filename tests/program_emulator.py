"""Test-only emulator of a compiled program blob (gwb_graph_export) in Python big ints.

Validates the HOST side of the product -- level scheduling into bundles, liveness slot reuse, operand
encoding, witness references -- on machines without a GPU.  Arithmetic comes from oracle/model.py, so
this checks the compiler, not the HIP kernels (those are checked by the `-m gpu` parity tests)."""
import struct

from oracle import model

REF_CONST = 0x80000000
CREF_TILE = 0x80000000
CTRL_A_TILE, CTRL_B_TILE, CTRL_A_FWD, CTRL_B_FWD, CTRL_ACTIVE = 1, 2, 4, 8, 1 << 24
FWD_NONE, FWD_PERMUTE, FWD_SAME_SOME, FWD_SAME_ALL = 0, 1, 2, 3
LIN_MIXED, LIN_ALL_ADD, LIN_ALL_SUB = 0, 1, 2
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
    """Evaluate one input set (list of ints) through the format-v2 program; returns (witness ints, status bits).
    Raises on any read of a slot that was never written (use-before-def = scheduling bug) and checks the
    wave-uniform header fields against the per-record control bits."""
    T, G = blob.T, blob.G
    slot_bytes = 32 * T
    trash = blob.n_slots * slot_bytes
    slots = {}
    status = 0
    prev = []  # register results of the previous bundle, by node slot

    def mem(off, tile_rel):
        if not tile_rel:
            assert off % slot_bytes == 0 and off // slot_bytes < blob.n_const
            return blob.consts[off // slot_bytes]
        assert off % slot_bytes == 0 and off // slot_bytes < blob.n_slots
        return slots[off // slot_bytes]

    for b in range(blob.n_bundles):
        h = blob.hdr[b]
        cls, cnt = h & 0xF, (h >> 4) & 0x7F
        amode, bmode, lin = (h >> 11) & 3, (h >> 13) & 3, (h >> 15) & 3
        assert 1 <= cnt <= G
        name = CLASS_NAMES[cls]
        results = []
        fwd = [[], []]  # per operand: list of (lane, src) for forwarded lanes
        subs = []
        for j in range(cnt):
            ctrl, dst, a, bb = blob.recs[(b * G + j) * 4:(b * G + j) * 4 + 4]
            assert ctrl & CTRL_ACTIVE
            sub = (ctrl >> 16) & 0xFF
            subs.append(sub)
            if name == "INPUT":
                assert sub == SUB_INPUT
                v = inputs_row[a] % model.M
            else:
                ops = []
                for q, (off, tbit, fbit, sh) in enumerate(((a, CTRL_A_TILE, CTRL_A_FWD, 4), (bb, CTRL_B_TILE, CTRL_B_FWD, 10))):
                    if ctrl & fbit:
                        src = (ctrl >> sh) & 63
                        fwd[q].append((j, src))
                        assert blob.consts[off // slot_bytes] == 0  # harmless prefetch target
                        ops.append(prev[src])  # IndexError = forwarded from a node slot the previous bundle left empty
                    else:
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
        # header modes must describe the records
        if name != "INPUT":
            for q, mode in enumerate((amode, bmode)):
                f = fwd[q]
                if not f:
                    want = FWD_NONE
                elif any(j != src for j, src in f):
                    want = FWD_PERMUTE
                else:
                    want = FWD_SAME_ALL if len(f) == cnt else FWD_SAME_SOME
                assert mode == want, (b, q, mode, want)
        if name == "LIN":
            want = LIN_ALL_ADD if all(x == 2 for x in subs) else LIN_ALL_SUB if all(x == 3 for x in subs) else LIN_MIXED
            assert lin == want
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

    def wit(r):
        return blob.consts[r & 0x7FFFFFFF] if r & REF_CONST else slots[r]
    return [wit(r) for r in blob.witness_refs], status
