"""Graph builder of the product: builds a circom-witnesscalc op graph symbolically and writes the
`wtns.graph.001` container through the C-ABI producer (`gwb_builder_*` in include/graph_witness_batch.h -> the product's
serialize_witnesscalc_graph, reference src/storage.rs:137-183).

Honours the invariants of a reference-produced graph: constants first, then Input(0) (= signal 0 = 1),
then one Input(k) per main-input scalar in declaration order, then op nodes in topological order; constants are
canonical; witness[0] is Input(0).

The independent pure-Python writer the tests compare these bytes with lives in tools/graphgen/pywriter.py (test
infrastructure, not imported here).
"""
import ctypes

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

DUO = ["Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq",
       "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor"]
DUO_CODE = {n: i for i, n in enumerate(DUO)}
UNO_CODE = {"Neg": 0, "Id": 1}
TRES_CODE = {"TernCond": 0}
BUILDER_BAD = 0xFFFFFFFF


def write_bin(nodes, witness_signals, input_signals):
    """nodes (tuples as Builder.finalize() returns them), witness list, {name: (offset, len)} -> `.bin` bytes through the
    C-ABI producer (gwb_builder_new / _input / _constant / _uno / _duo / _tres / _witness / _input_signal / _finish)."""
    from .. import GwStatus, WitnessCalcError, _libc, lib
    L = lib()
    b = L.gwb_builder_new()
    if not b:
        raise WitnessCalcError("gwb_builder_new failed")
    try:
        for n in nodes:
            k = n[0]
            if k == "Input":
                idx = L.gwb_builder_input(b, n[1])
            elif k == "Const":
                v = n[1]
                le = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "little")
                idx = L.gwb_builder_constant(b, le, len(le))
            elif k == "Uno":
                idx = L.gwb_builder_uno(b, UNO_CODE[n[1]], n[2])
            elif k == "Duo":
                idx = L.gwb_builder_duo(b, DUO_CODE[n[1]], n[2], n[3])
            elif k == "Tres":
                idx = L.gwb_builder_tres(b, TRES_CODE[n[1]], n[2], n[3], n[4])
            else:
                raise ValueError(k)
            if idx == BUILDER_BAD:
                break
        for w in witness_signals:
            L.gwb_builder_witness(b, w)
        for name, (off, ln) in input_signals.items():
            L.gwb_builder_input_signal(b, name.encode(), off, ln)
        out, n_out, st = ctypes.c_void_p(), ctypes.c_size_t(), GwStatus()
        rc = L.gwb_builder_finish(b, ctypes.byref(out), ctypes.byref(n_out), ctypes.byref(st))
        msg = ctypes.string_at(st.error_msg).decode("utf-8", "replace") if st.error_msg else ""
        L.gwb_free_status(ctypes.byref(st))
        if rc != 0:
            raise WitnessCalcError(msg or "gwb_builder_finish failed")
        data = ctypes.string_at(out.value, n_out.value)
        _libc.free(out)
        return data
    finally:
        L.gwb_builder_free(b)


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
        """the `.bin` container, written by the product's writer through the C-ABI producer"""
        return write_bin(*self.finalize())


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
