"""References that owe nothing to this repository's restatements of the reference (oracle/model.py, oracle/cwc_oracle.c)
or to its product: plain Python integers and published numbers.  Each states the reference semantics it stands for from
the source's own words (file:line), not through the oracle's code.  Graphs are written by the independent pure-Python
`.bin` writer (tools/graphgen/pywriter.py), not by the product's producer."""
import random

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
HALF = R >> 1  # graph.rs:723-769 compares with M / 2: values above it are negative

# circomlibjs test/poseidon.js ("poseidonperm_x5_254_3", "poseidonperm_x5_254_5"): published hashes of circomlib's Poseidon
POSEIDON_PUBLISHED = {(1, 2): 7853200120776062878684798364095072458815029376092732009249414926327459813530,
                      (1, 2, 3, 4): 18821383157269793795438455681495246036402687001665670618754263018637548127333}

OPS = ["Lt", "Gt", "Leq", "Geq", "Bor", "Bxor", "Band", "Shr", "Shl", "Idiv", "Mod", "Eq", "Neq", "Land", "Lor", "Mul", "Add", "Sub"]


def signed(a):
    """the integer a field element stands for in an ordered comparison: [0, (r-1)/2] as is, above it a - r"""
    return a if a <= HALF else a - R


def plain(op, a, b):
    """(value, panics) of one operation on canonical field elements a, b in plain integer arithmetic"""
    if op == "Lt":
        return int(signed(a) < signed(b)), False
    if op == "Gt":
        return int(signed(a) > signed(b)), False
    if op == "Leq":
        return int(signed(a) <= signed(b)), False
    if op == "Geq":
        return int(signed(a) >= signed(b)), False
    if op in ("Bor", "Bxor", "Band"):  # graph.rs:674-717: the bit operation on the integers, one subtraction of r if it is not below r; == r has no field element
        v = a | b if op == "Bor" else a ^ b if op == "Bxor" else a & b
        if v == R:
            return None, True
        return (v - R if v > R else v), False
    if op == "Shr":  # graph.rs:637-672: zero from 254 bits on
        return (a >> b if b < 254 else 0), False
    if op == "Shl":  # graph.rs:621-635: bits beyond 256 fall off, the rest must be a field element
        if b >= 254:
            return 0, False
        v = (a << b) & ((1 << 256) - 1)
        return (None, True) if v >= R else (v, False)
    if op == "Idiv":
        return (a // b if b else 0), False
    if op == "Mod":
        return (a % b if b else 0), False
    if op == "Eq":
        return int(a == b), False
    if op == "Neq":
        return int(a != b), False
    if op == "Land":
        return int(a != 0 and b != 0), False
    if op == "Lor":
        return int(a != 0 or b != 0), False
    if op == "Mul":
        return a * b % R, False
    if op == "Add":
        return (a + b) % R, False
    if op == "Sub":
        return (a - b) % R, False
    raise ValueError(op)


def ops_graph():
    """.bin bytes of a graph with two inputs and one witness element per operation of OPS (written by pywriter)"""
    from tools.graphgen.pywriter import serialize_graph
    nodes = [("Input", 0), ("Input", 1), ("Input", 2)] + [("Duo", op, 1, 2) for op in OPS]
    return serialize_graph(nodes, [0] + list(range(3, 3 + len(OPS))), {"a": (1, 1), "b": (2, 1)})


def operand_pairs(seed, n):
    rnd = random.Random(seed)
    edge = [0, 1, 2, HALF - 1, HALF, HALF + 1, HALF + 2, R - 2, R - 1, 253, 254, 255, (1 << 253) - 1, 1 << 253, R & ((1 << 253) - 1), R - 1 - (R & ((1 << 253) - 1)), R ^ 1]
    pairs = [(a, b) for a in edge for b in edge]
    while len(pairs) < n:
        k = rnd.random()
        a = rnd.randrange(R)
        b = rnd.randrange(R) if k < 0.5 else rnd.randrange(300) if k < 0.7 else (R - a) % R if k < 0.75 else a ^ rnd.randrange(1 << 20) if k < 0.8 else rnd.randrange(1 << rnd.randrange(1, 254))
        pairs.append((a, b % R))
    return pairs
