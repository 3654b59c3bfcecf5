"""Graph builder + `.bin` writer for synthetic circom-witnesscalc graphs (test / bench infrastructure).

Emits the `wtns.graph.001` container exactly as the reference writer does
(reference src/storage.rs:137-183; schema protos/messages.proto) and honours the invariants of a
reference-produced graph: constants first, then Input(0) (= signal 0 = 1), then one
Input(k) per main-input scalar in declaration order, then op nodes in topological order; constants
are canonical, minimal-length little-endian; witness[0] is Input(0).

This file does not import anything from oracle/ (the oracle has its own, independent reader).
"""
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

DUO = ["Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq",
       "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor"]
DUO_CODE = {n: i for i, n in enumerate(DUO)}
UNO_CODE = {"Neg": 0, "Id": 1}
TRES_CODE = {"TernCond": 0}


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_varint(fno, v):
    # proto3: default (zero) scalars are elided, as prost does
    return b"" if v == 0 else _varint(fno << 3) + _varint(v)


def _field_bytes(fno, b):
    return _varint((fno << 3) | 2) + _varint(len(b)) + b


def encode_node(node):
    """One proto::Node message body (without the length prefix)."""
    k = node[0]
    if k == "Input":
        return _field_bytes(1, _field_varint(1, node[1]))
    if k == "Const":
        v = node[1]
        le = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "little")  # num-bigint to_bytes_le
        return _field_bytes(2, _field_bytes(1, _field_bytes(1, le)))
    if k == "Uno":
        return _field_bytes(3, _field_varint(1, UNO_CODE[node[1]]) + _field_varint(2, node[2]))
    if k == "Duo":
        return _field_bytes(4, _field_varint(1, DUO_CODE[node[1]]) + _field_varint(2, node[2])
                            + _field_varint(3, node[3]))
    if k == "Tres":
        return _field_bytes(5, _field_varint(1, TRES_CODE[node[1]]) + _field_varint(2, node[2])
                            + _field_varint(3, node[3]) + _field_varint(4, node[4]))
    raise ValueError(k)


def serialize_graph(nodes, witness_signals, input_signals):
    """serialize_witnesscalc_graph (reference src/storage.rs:137-183).
    nodes: list of tuples, witness_signals: list[int], input_signals: {name: (offset, len)}."""
    out = bytearray(b"wtns.graph.001")
    out += struct.pack("<Q", len(nodes))
    for n in nodes:
        body = encode_node(n)
        out += _varint(len(body)) + body
    md = bytearray()
    if witness_signals:
        packed = b"".join(_varint(w) for w in witness_signals)
        md += _field_bytes(1, packed)
    for name, (off, ln) in input_signals.items():
        entry = _field_bytes(1, name.encode()) + _field_bytes(2, _field_varint(1, off) + _field_varint(2, ln))
        md += _field_bytes(2, entry)
    md_off = len(out)
    out += _varint(len(md)) + md
    out += struct.pack("<Q", md_off)
    return bytes(out)


class Sig(int):
    """Symbolic node handle (an int id into Builder._sym)."""


class Builder:
    """Builds a graph symbolically, then lays it out in reference order on finalize()."""

    def __init__(self, dedup_consts=True):
        self._consts = {}     # value -> sym id
        self._sym = []        # sym id -> ("Const", v) | ("Input", k) | op tuples with sym operands
        self._inputs = {}     # name -> (offset, len)
        self._n_in = 1
        self._dedup = dedup_consts
        self._witness = []
        self.one_in = self._push(("Input", 0))  # signal 0
        self._witness.append(self.one_in)

    def _push(self, t):
        self._sym.append(t)
        return len(self._sym) - 1

    # -- leaves -------------------------------------------------------------------------------
    def const(self, v):
        v %= R
        if self._dedup and v in self._consts:
            return self._consts[v]
        s = self._push(("Const", v))
        self._consts[v] = s
        return s

    def input(self, name, n=1):
        """Declare a main input signal array; returns list of n handles."""
        off = self._n_in
        self._inputs[name] = (off, n)
        hs = [self._push(("Input", off + i)) for i in range(n)]
        self._n_in += n
        return hs

    # -- ops ----------------------------------------------------------------------------------
    def op(self, name, a, b):
        return self._push(("Duo", name, a, b))

    def neg(self, a):
        return self._push(("Uno", "Neg", a))

    def tern(self, c, a, b):
        return self._push(("Tres", "TernCond", c, a, b))

    def mul(self, a, b): return self.op("Mul", a, b)
    def add(self, a, b): return self.op("Add", a, b)
    def sub(self, a, b): return self.op("Sub", a, b)
    def div(self, a, b): return self.op("Div", a, b)

    def signal(self, h):
        """Mark a node as a witness signal (appended in call order)."""
        self._witness.append(h)
        return h

    @property
    def n_inputs(self):
        return self._n_in

    # -- layout -------------------------------------------------------------------------------
    def finalize(self):
        """-> (nodes, witness_signals, input_signals) in reference layout."""
        order = [i for i, t in enumerate(self._sym) if t[0] == "Const"]
        order += [i for i, t in enumerate(self._sym) if t[0] == "Input"]
        order += [i for i, t in enumerate(self._sym) if t[0] not in ("Const", "Input")]
        remap = {s: i for i, s in enumerate(order)}
        nodes = []
        for s in order:
            t = self._sym[s]
            if t[0] in ("Const", "Input"):
                nodes.append(t)
            elif t[0] == "Uno":
                nodes.append((t[0], t[1], remap[t[2]]))
            elif t[0] == "Duo":
                nodes.append((t[0], t[1], remap[t[2]], remap[t[3]]))
            else:
                nodes.append((t[0], t[1], remap[t[2]], remap[t[3]], remap[t[4]]))
        for i, n in enumerate(nodes):  # all references backward (graph.rs:343-356)
            for o in n[2:] if n[0] in ("Uno", "Duo", "Tres") else ():
                assert o < i
        return nodes, [remap[w] for w in self._witness], dict(self._inputs)

    def to_bin(self):
        return serialize_graph(*self.finalize())


def graph_stats(nodes, witness):
    """Node count, op histogram, dependency depth, witness length (printed by bench runs)."""
    hist = {}
    depth = [0] * len(nodes)
    for i, n in enumerate(nodes):
        k = n[0]
        if k in ("Const", "Input"):
            hist[k] = hist.get(k, 0) + 1
            continue
        hist[n[1]] = hist.get(n[1], 0) + 1
        depth[i] = 1 + max(depth[o] for o in n[2:])
    n_op = sum(v for k, v in hist.items() if k not in ("Const", "Input"))
    return {"N": len(nodes), "N_op": n_op, "W": len(witness), "depth": max(depth) if depth else 0,
            "hist": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
