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


# ---- Neg, TernCond and inputs at or above r (round 6): the three semantics that rested on the two restatements agreeing with each other ----
def neg_plain(a):
    """UnoOperation::Neg on a canonical field element (graph.rs:188-194): 0 stays 0, anything else is r - a"""
    return 0 if a == 0 else R - a


def tern_plain(a, b, c):
    """TresOperation::TernCond (graph.rs:221-225): a == 0 ? c : b -- both arms are values, nothing is evaluated lazily"""
    return c if a == 0 else b


def input_plain(x):
    """graph.rs:376 `Fr::new(inputs[i])`: any 256-bit input stands for x mod r (ark-ff's Montgomery `new` multiplies by R^2 and
    reduces; lib.rs:195-247 checks no range), and leaves as the canonical residue (graph.rs:387 `into_bigint`)"""
    return x % R


def uno_tres_graph():
    """.bin bytes (pywriter) of a graph over three inputs a, b, c whose witness is [1, a, b, c, Neg a, Neg b, TernCond(a, b, c),
    TernCond(b, c, a), TernCond(Neg a, a, b), Neg TernCond(c, a, b)]: the Input nodes themselves are witness elements, so inputs at or
    above r show their reduction, and a selector that is r (or 2r) as an input must select like 0."""
    from tools.graphgen.pywriter import serialize_graph
    nodes = [("Input", 0), ("Input", 1), ("Input", 2), ("Input", 3),
             ("Uno", "Neg", 1), ("Uno", "Neg", 2), ("Tres", "TernCond", 1, 2, 3), ("Tres", "TernCond", 2, 3, 1),
             ("Tres", "TernCond", 4, 1, 2), ("Tres", "TernCond", 3, 1, 2), ("Uno", "Neg", 9)]
    return serialize_graph(nodes, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10], {"a": (1, 1), "b": (2, 1), "c": (3, 1)})


def uno_tres_plain(a, b, c):
    """the witness of uno_tres_graph for RAW inputs a, b, c < 2^256, in plain integers"""
    ra, rb, rc = input_plain(a), input_plain(b), input_plain(c)
    return [1, ra, rb, rc, neg_plain(ra), neg_plain(rb), tern_plain(ra, rb, rc), tern_plain(rb, rc, ra),
            tern_plain(neg_plain(ra), ra, rb), neg_plain(tern_plain(rc, ra, rb))]


def uno_tres_inputs(seed, n):
    """raw input triples: every combination of the values the semantics turn on (0, 1, r - 1, r, r + 1, r + 5, 2r, 2r + 1, 2^256 - 1,
    multiples of r below 2^256) and random ones below r / below 2^256"""
    rnd = random.Random(seed)
    top = (1 << 256) - 1
    edge = [0, 1, 2, R - 1, R, R + 1, R + 5, 2 * R, 2 * R + 1, 5 * R, 5 * R + 3, top, top - 1, (top // R) * R, HALF, HALF + 1, HALF + R]
    rows = [(a, b, c) for a in edge for b in edge[:9] for c in (0, 7, R, top)]
    while len(rows) < n:
        pick = lambda: rnd.choice(edge) if rnd.random() < 0.3 else rnd.randrange(R) if rnd.random() < 0.5 else rnd.randrange(1 << 256)
        rows.append((pick(), pick(), pick()))
    return rows
