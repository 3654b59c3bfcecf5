"""Independent pure-Python writer of the `wtns.graph.001` container (test infrastructure).

Emits the container exactly as the reference writer does (reference src/storage.rs:137-183; schema
protos/messages.proto).  The product writes `.bin` files through the C-ABI producer (gwb_builder_*, graph.cc); the tests
compare the two byte for byte.  Nothing in the product imports this file.
"""
import struct

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
