"""Test-only emulator of a compiled program blob (gwb_graph_export) in Python big ints.

Validates the HOST side of the product -- level scheduling into bundles, liveness slot reuse, operand
encoding, witness references -- on machines without a GPU.  Arithmetic comes from oracle/model.py, so
this checks the compiler, not the HIP kernels (those are checked by the `-m gpu` parity tests)."""
import struct

from oracle import model

REF_CONST = 0x80000000
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
        self.witness_refs = take(self.n_witness)


def run(blob: Blob, inputs_row):
    """Evaluate one input set (list of ints); returns (witness ints, status bits). Raises on any
    read of a slot that was never written (use-before-def = scheduling bug)."""
    slots = {}
    status = 0

    def ld(ref):
        if ref & REF_CONST:
            return blob.consts[ref & 0x7FFFFFFF]
        return slots[ref]
    G = blob.G
    for b in range(blob.n_bundles):
        cls, cnt = blob.hdr[b] & 0xFF, blob.hdr[b] >> 8
        assert 1 <= cnt <= G
        results = []
        for j in range(cnt):
            sub, dst, a, bb = blob.recs[(b * G + j) * 4:(b * G + j) * 4 + 4]
            name = CLASS_NAMES[cls]
            if name == "INPUT":
                assert sub == 34
                v = inputs_row[a] % model.M
            elif name == "TERN":
                assert sub == 33
                v = model.eval_tres("TernCond", ld(a), ld(bb), ld(blob.crefs[b * G + j]))
            elif sub == 32:
                assert name == "LIN" and a == bb
                v = model.eval_uno("Neg", ld(a))
            else:
                op = model.DUO[sub]
                expect = {"Mul": "MUL", "Div": "DIV", "Add": "LIN", "Sub": "LIN", "Idiv": "IDIVMOD", "Mod": "IDIVMOD",
                          "Eq": "CMPZ", "Neq": "CMPZ", "Land": "CMPZ", "Lor": "CMPZ", "Lt": "CMPS", "Gt": "CMPS",
                          "Leq": "CMPS", "Geq": "CMPS", "Shl": "BIT", "Shr": "BIT", "Bor": "BIT", "Band": "BIT",
                          "Bxor": "BIT"}[op]
                assert expect == name, (op, name)
                try:
                    v = model.eval_duo(op, ld(a), ld(bb))
                except model.ReferencePanic:
                    status |= 1 if op == "Shl" else 2
                    v = 0
            results.append((dst, v))
        for j in range(cnt, G):  # padding records must replicate record 0
            assert blob.recs[(b * G + j) * 4:(b * G + j) * 4 + 4] == blob.recs[b * G * 4:b * G * 4 + 4]
        dsts = [d for d, _ in results]
        assert len(set(dsts)) == len(dsts), "two nodes of one bundle share a destination slot"
        for d, v in results:  # all loads of a bundle happen before its stores
            assert d < blob.n_slots
            slots[d] = v
    return [ld(r) for r in blob.witness_refs], status
