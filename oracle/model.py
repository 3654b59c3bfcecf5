"""ORACLE (test infrastructure, NOT product code) -- pure-Python big-int restatement of the
calc-witness path of iden3/circom-witnesscalc (reference @ 2024-10-22, v0.2.0).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (circom-witnesscalc_amd/) never does.

Parity status: the Rust reference cannot be built in this environment (no cargo/rustc, crates
un-vendored, no network), so this model is pinned against the reference's own unit vectors
(src/graph.rs:779-883, src/lib.rs:257-280, src/storage.rs:316-465) and the hand-derived circuit1
fixture of SURVEY.md section 8(c); see tests/test_oracle_golden.py.  Everything else is
"parity unpinned" -- it rests on reading the reference source plus the mathematically determined
behaviour of its dependencies (ark-ff 0.4.2 / ark-bn254 0.4.0 Fr arithmetic, ruint 1.12.3 U256
div/rem/cmp, wtns-file 0.1.5, prost 0.13.3), none of which is vendored under /root/reference.

All values here are *canonical* integers in [0, r); the reference's Montgomery representation is
not observable (every exit goes through into_bigint, src/graph.rs:387).
"""
import json
import struct

# src/field.rs:3-4  (BN254 scalar field modulus r)
M = 21888242871839275222246405745257275088548364400416034343698204186575808495617
# src/graph.rs:720
HALF_M = 10944121435919637611123202872628637544274182200208017171849102093287904247808
assert HALF_M == M // 2
MASK256 = (1 << 256) - 1

# protos/messages.proto:5-35  wire codes
DUO = ["Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq",
       "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor"]
UNO = ["Neg", "Id"]
TRES = ["TernCond"]
DUO_CODE = {n: i for i, n in enumerate(DUO)}
UNO_CODE = {n: i for i, n in enumerate(UNO)}


class ReferencePanic(Exception):
    """The reference would panic / hit unimplemented! on this input (outside the parity domain)."""


# ---------------------------------------------------------------------------------------------
# node evaluators  (src/graph.rs:102-144, 188-197, 221-225, 621-769)
# ---------------------------------------------------------------------------------------------
def _neg(x):  # src/graph.rs:724-725  "a_neg = halfM < a"
    return x > HALF_M


def u_lt(a, b):  # src/graph.rs:759-769
    an, bn = _neg(a), _neg(b)
    if an == bn:
        return int(a < b)
    return 1 if an else 0


def u_gt(a, b):  # src/graph.rs:747-757
    an, bn = _neg(a), _neg(b)
    if an == bn:
        return int(a > b)
    return 0 if an else 1


def u_lte(a, b):  # src/graph.rs:735-745
    an, bn = _neg(a), _neg(b)
    if an == bn:
        return int(a <= b)
    return 1 if an else 0


def u_gte(a, b):  # src/graph.rs:723-733
    an, bn = _neg(a), _neg(b)
    if an == bn:
        return int(a >= b)
    return 0 if an else 1


def shl(a, b):  # src/graph.rs:621-635
    if b == 0:
        return a
    if b >= 254:  # Fr::MODULUS_BIT_SIZE
        return 0
    x = (a << b) & MASK256  # BigInt::muln drops bits shifted past 256 [ext: ark-ff]
    if x >= M:
        raise ReferencePanic("shl: from_bigint(None).unwrap()")  # :634
    return x


def shr(a, b):  # src/graph.rs:637-672
    if b == 0:
        return a
    if b >= 254:
        return 0
    return a >> b


def _bitop(d):  # src/graph.rs:682-686 / 697-701 / 712-716
    if d > M:
        d -= M
    if d >= M:
        raise ReferencePanic("bit op result == r: from_bigint(None).unwrap()")
    return d


def eval_duo(op, a, b):
    """Operation::eval_fr, src/graph.rs:102-144, on canonical representatives."""
    if op == "Mul":
        return a * b % M
    if op == "Div":
        return 0 if b == 0 else a * pow(b, M - 2, M) % M
    if op == "Add":
        return (a + b) % M
    if op == "Sub":
        return (a - b) % M
    if op == "Idiv":
        return 0 if b == 0 else a // b
    if op == "Mod":
        return 0 if b == 0 else a % b
    if op == "Eq":
        return int(a == b)
    if op == "Neq":
        return int(a != b)
    if op == "Lt":
        return u_lt(a, b)
    if op == "Gt":
        return u_gt(a, b)
    if op == "Leq":
        return u_lte(a, b)
    if op == "Geq":
        return u_gte(a, b)
    if op == "Land":
        return int(a != 0 and b != 0)
    if op == "Lor":
        return int(a != 0 or b != 0)
    if op == "Shl":
        return shl(a, b)
    if op == "Shr":
        return shr(a, b)
    if op == "Bor":
        return _bitop(a | b)
    if op == "Band":
        return _bitop(a & b)
    if op == "Bxor":
        return _bitop(a ^ b)
    raise ReferencePanic("operator %s not implemented for Montgomery" % op)  # :141-142 (Pow)


def eval_uno(op, a):  # src/graph.rs:188-197
    if op == "Neg":
        return 0 if a == 0 else M - a
    raise ReferencePanic("uno operator %s not implemented for Montgomery" % op)


def eval_tres(op, a, b, c):  # src/graph.rs:221-225
    assert op == "TernCond"
    return c if a == 0 else b


# Nodes are tuples: ("Input", idx) | ("Const", value) | ("Uno", op, a) | ("Duo", op, a, b)
#                   | ("Tres", op, a, b, c)
def evaluate(nodes, inputs, outputs):
    """graph::evaluate, src/graph.rs:367-391."""
    values = []
    for n in nodes:
        k = n[0]
        if k == "Const":
            v = n[1] % M
        elif k == "Input":
            v = inputs[n[1]] % M  # Fr::new(U256): clean reduction assumed for x >= r [ext, unpinned]
        elif k == "Duo":
            v = eval_duo(n[1], values[n[2]], values[n[3]])
        elif k == "Uno":
            v = eval_uno(n[1], values[n[2]])
        elif k == "Tres":
            v = eval_tres(n[1], values[n[2]], values[n[3]], values[n[4]])
        else:
            raise ValueError(k)
        values.append(v)
    return [values[i] for i in outputs]


# ---------------------------------------------------------------------------------------------
# .bin container  (src/storage.rs:9-48, 185-249; protos/messages.proto)
# ---------------------------------------------------------------------------------------------
MAGIC = b"wtns.graph.001"  # src/storage.rs:16


def _varint(buf, pos):
    shift = 0
    out = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(buf):
    """Yield (field_no, wire_type, value) of one protobuf message; value is int or bytes."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            if len(v) != ln:
                raise ValueError("truncated field")
            pos += ln
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


def _msg_ints(buf, want):
    d = {k: 0 for k in want}
    for fno, wt, v in _fields(buf):
        if fno in d and wt == 0:
            d[fno] = v
    return d


def _decode_node(buf):
    """proto::Node -> graph::Node, src/storage.rs:20-48."""
    node = None
    for fno, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fno == 1:  # InputNode
            node = ("Input", _msg_ints(v, [1])[1])
        elif fno == 2:  # ConstantNode{1: BigUInt{1: bytes}}
            val = b""
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1 and w3 == 2:
                            val = v3
            node = ("Const", int.from_bytes(val, "little") % M)  # from_le_bytes_mod_order :28
        elif fno == 3:
            d = _msg_ints(v, [1, 2])
            node = ("Uno", UNO[d[1]], d[2])
        elif fno == 4:
            d = _msg_ints(v, [1, 2, 3])
            node = ("Duo", DUO[d[1]], d[2], d[3])
        elif fno == 5:
            d = _msg_ints(v, [1, 2, 3, 4])
            node = ("Tres", TRES[d[1]], d[2], d[3], d[4])
    if node is None:
        raise ValueError("Node with empty oneof")  # value.node.unwrap() :22
    return node


def deserialize_witnesscalc_graph(data):
    """src/storage.rs:214-249 -> (nodes, witness_signals, input_signals{name:(offset,len)})."""
    data = bytes(data)
    if data[:len(MAGIC)] != MAGIC:
        raise ValueError("Invalid magic")
    pos = len(MAGIC)
    (n_nodes,) = struct.unpack_from("<Q", data, pos)  # :228 (u64, the ":11" comment is stale)
    pos += 8
    nodes = []
    for _ in range(n_nodes):
        ln, pos = _varint(data, pos)
        if pos + ln > len(data):
            raise ValueError("Unexpected EOF")
        nodes.append(_decode_node(data[pos:pos + ln]))
        pos += ln
    ln, pos = _varint(data, pos)
    md = data[pos:pos + ln]
    if len(md) != ln:
        raise ValueError("Unexpected EOF")
    witness = []
    inputs = {}
    for fno, wt, v in _fields(md):
        if fno == 1:
            if wt == 2:  # packed
                p = 0
                while p < len(v):
                    x, p = _varint(v, p)
                    witness.append(x)
            else:
                witness.append(v)
        elif fno == 2 and wt == 2:  # map entry {1: key, 2: SignalDescription{1: offset, 2: len}}
            key, off, ln2 = "", 0, 0
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    key = v2.decode("utf-8")
                elif f2 == 2 and w2 == 2:
                    d = _msg_ints(v2, [1, 2])
                    off, ln2 = d[1], d[2]
            inputs[key] = (off, ln2)
    return nodes, witness, inputs


# ---------------------------------------------------------------------------------------------
# host API  (src/lib.rs:114-247)
# ---------------------------------------------------------------------------------------------
class InputsError(Exception):
    pass


def _parse_u256_dec(s):
    """U256::from_str_radix(s, 10), src/lib.rs:208,223.  [ext: ruint] digits only (ruint also
    skips '_'), value must fit 256 bits; empty string parses as 0 in ruint 1.12."""
    v = 0
    for ch in s:
        if ch == "_":
            continue
        if not ("0" <= ch <= "9"):
            raise InputsError("InputFieldNumberParseError(InvalidDigit(%r))" % ch)
        v = v * 10 + (ord(ch) - 48)
        if v > MASK256:
            raise InputsError("InputFieldNumberParseError(BaseOverflow)")
    return v


def _scalar(v, key, in_array):
    if isinstance(v, str):
        return _parse_u256_dec(v)
    if isinstance(v, bool) or v is None or isinstance(v, (list, dict)):
        if in_array:
            raise InputsError("inputs must be a string: %s" % key)  # :232
        raise InputsError("value for key %s must be an a number as a string, as a number of an "
                          "array of strings of numbers" % key)  # :240-242
    if isinstance(v, int) and 0 <= v < (1 << 64):
        return v
    raise InputsError("signal value is not a positive integer")  # :213,:227


def deserialize_inputs(data):
    """src/lib.rs:195-247."""
    if isinstance(data, (bytes, bytearray)):
        data = data.decode("utf-8")
    v = json.loads(data)  # invalid JSON: reference panics (:196 unwrap); here ValueError
    if not isinstance(v, dict):
        raise InputsError("inputs must be an object")  # :201
    out = {}
    for k, val in v.items():  # duplicate keys: last wins (serde_json map / python dict alike)
        if isinstance(val, list):
            out[k] = [_scalar(x, k, True) for x in val]
        else:
            out[k] = [_scalar(val, k, False)]
    return out


def get_inputs_size(nodes):  # src/lib.rs:138-152
    start = False
    mx = 0
    for n in nodes:
        if n[0] == "Input":
            mx = max(mx, n[1])
            start = True
        elif start:
            break
    return mx + 1


def get_inputs_buffer(size):  # src/lib.rs:177-181
    buf = [0] * size
    buf[0] = 1
    return buf


def populate_inputs(input_list, inputs_info, buf):  # src/lib.rs:154-168
    for key, value in input_list.items():
        if key not in inputs_info:
            raise ReferencePanic("unknown input key %s (HashMap index panic)" % key)
        off, ln = inputs_info[key]
        if ln != len(value):
            raise ReferencePanic("Invalid input length for %s" % key)
        for i, v in enumerate(value):
            if off + i >= len(buf):
                raise ReferencePanic("input index out of range")
            buf[off + i] = v


def calc_witness(inputs_json, graph_data):  # src/lib.rs:125-136
    inputs = deserialize_inputs(inputs_json)
    nodes, signals, mapping = deserialize_witnesscalc_graph(graph_data)
    buf = get_inputs_buffer(get_inputs_size(nodes))
    populate_inputs(inputs, mapping, buf)
    return evaluate(nodes, buf, signals)


def wtns_from_witness(witness):
    """src/lib.rs:114-123 + wtns-file 0.1.5 [ext]: iden3 binfile 'wtns' v2, two sections."""
    n = len(witness)
    out = bytearray()
    out += b"wtns"
    out += struct.pack("<II", 2, 2)  # version (forced :118), nSections
    out += struct.pack("<IQ", 1, 40)  # section 1: header, 4 + 32 + 4 bytes
    out += struct.pack("<I", 32)  # n8
    out += M.to_bytes(32, "little")  # prime :117
    out += struct.pack("<I", n)
    out += struct.pack("<IQ", 2, 32 * n)  # section 2: witness body
    for w in witness:
        out += int(w).to_bytes(32, "little")  # as_le_slice :171
    return bytes(out)
