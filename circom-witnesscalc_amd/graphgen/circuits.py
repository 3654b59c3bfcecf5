"""Circuit-shaped synthetic graph generators (test / bench infrastructure).

Real authV2 / sha256 `.bin` graphs cannot be produced offline (no circom toolchain, circomlib
submodules empty), so these generators emit graphs with the *node patterns* the
reference front-end emits for the corresponding circomlib templates:
witness hints `<--` become Div/Shr/Band/TernCond nodes, `<==` become Mul/Add/Sub chains, asserts are
dropped, constant-only subexpressions are folded.  Poseidon constants are seeded pseudo-random
("Poseidon-shaped"), SHA-256 uses the real FIPS 180-4 constants (so its output is checkable against
hashlib -- an oracle independent of all of this repository's code).
"""
import hashlib
import random

from .builder import Builder, R

POSEIDON_RP = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]  # t = 2..17
POSEIDON_RF = 8


def _field_stream(tag):
    """Deterministic stream of field elements (seeded constants)."""
    ctr = 0
    while True:
        h = hashlib.sha256(("%s/%d" % (tag, ctr)).encode()).digest()
        ctr += 1
        yield int.from_bytes(h, "little") % R


def poseidon_grain_constants(t, r_f=POSEIDON_RF, r_p=None, n=254):
    """The Poseidon round constants and MDS matrix of the reference parameter script (Grassi et al., Poseidon, USENIX
    Security 2021, generate_params_poseidon.sage): a Grain LFSR in self-shrinking mode seeded with (field = 1, sbox = 0 i.e.
    x^5, n, t, R_F, R_P), 160 bits discarded; round constants by rejection below r, then a Cauchy matrix 1 / (x_i + y_j)
    from 2t more elements.  These are circomlib's (non-optimised) Poseidon constants: the published hashes come out,
    poseidon([1, 2]) = 7853200120776062878684798364095072458815029376092732009249414926327459813530 (tests)."""
    r_p = POSEIDON_RP[t - 2] if r_p is None else r_p
    bits = []
    for v, w in ((1, 2), (0, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
        bits.extend(int(c) for c in bin(v)[2:].zfill(w))
    bits.extend([1] * 30)

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb
    for _ in range(160):
        step()

    def next_bit():  # self-shrinking: a 1 lets the following bit through, a 0 drops it
        while step() == 0:
            step()
        return step()

    def element_bits():
        v = 0
        for _ in range(n):
            v = (v << 1) | next_bit()
        return v
    consts = []
    while len(consts) < (r_f + r_p) * t:
        v = element_bits()
        if v < R:
            consts.append(v)
    while True:
        xy = [element_bits() % R for _ in range(2 * t)]
        if len(set(xy)) != 2 * t or any((x + y) % R == 0 for x in xy[:t] for y in xy[t:]):
            continue
        mds = [[pow((xy[i] + xy[t + j]) % R, -1, R) for j in range(t)] for i in range(t)]
        return [consts[r * t:(r + 1) * t] for r in range(r_f + r_p)], mds


class PoseidonParams:
    _cache = {}

    def __init__(self, t, circomlib=False):
        self.t = t
        self.rp = POSEIDON_RP[t - 2]
        n = POSEIDON_RF + self.rp
        if circomlib:  # the real constants (circomlib's Poseidon): anchors on published hashes
            self.C, self.Mx = poseidon_grain_constants(t)
            return
        s = _field_stream("poseidon-shaped/t=%d" % t)
        self.C = [[next(s) for _ in range(t)] for _ in range(n)]
        self.Mx = [[next(s) for _ in range(t)] for _ in range(t)]

    @classmethod
    def get(cls, t, circomlib=False):
        if (t, circomlib) not in cls._cache:
            cls._cache[(t, circomlib)] = cls(t, circomlib)
        return cls._cache[(t, circomlib)]


def poseidon(b: Builder, inputs, signals=True, circomlib=False):
    """Poseidon-shaped permutation hash of len(inputs) field elements (t = n+1), first state word out.
    Node pattern per round: Ark = Add(x, const); Sigma = 3 Mul (x^2, x^4, x^5); Mix = Mul(const, x)
    products summed by a serial Add chain (circom `lc += M[j][i]*in[j]`).  circomlib=True: the reference constants
    (poseidon_grain_constants) instead of the seeded ones -- circomlib's Poseidon itself."""
    t = len(inputs) + 1
    pp = PoseidonParams.get(t, circomlib)
    st = [b.const(0)] + list(inputs)
    half = POSEIDON_RF // 2
    for r in range(POSEIDON_RF + pp.rp):
        # ARK
        st = [b.add(st[i], b.const(pp.C[r][i])) for i in range(t)]
        # S-box
        full = r < half or r >= half + pp.rp
        for i in range(t if full else 1):
            x2 = b.mul(st[i], st[i])
            x4 = b.mul(x2, x2)
            x5 = b.mul(x4, st[i])
            if signals:
                b.signal(x2); b.signal(x4); b.signal(x5)
            st[i] = x5
        # MDS mix
        nxt = []
        for i in range(t):
            lc = b.mul(b.const(pp.Mx[i][0]), st[0])
            for j in range(1, t):
                lc = b.add(lc, b.mul(b.const(pp.Mx[i][j]), st[j]))
            if signals:
                b.signal(lc)
            nxt.append(lc)
        st = nxt
    return st[0]


def poseidon_model(inputs, circomlib=False):
    """Pure-integer model of `poseidon` above (for generator self-checks)."""
    t = len(inputs) + 1
    pp = PoseidonParams.get(t, circomlib)
    st = [0] + [x % R for x in inputs]
    half = POSEIDON_RF // 2
    for r in range(POSEIDON_RF + pp.rp):
        st = [(st[i] + pp.C[r][i]) % R for i in range(t)]
        full = r < half or r >= half + pp.rp
        for i in range(t if full else 1):
            st[i] = pow(st[i], 5, R)
        st = [sum(pp.Mx[i][j] * st[j] for j in range(t)) % R for i in range(t)]
    return st[0]


# -- bit gadgets ------------------------------------------------------------------------------------
def num2bits(b: Builder, x, n, signals=True):
    """circomlib Num2Bits: out[i] <-- (in >> i) & 1  ->  Band(Shr(x, i), 1); constraints dropped."""
    one = b.const(1)
    out = []
    for i in range(n):
        bit = b.op("Band", b.op("Shr", x, b.const(i)), one)
        if signals:
            b.signal(bit)
        out.append(bit)
    return out


def bits2num(b: Builder, bits):
    """circomlib Bits2Num: lc1 += in[i] * 2^i (serial chain); out <== lc1."""
    lc = b.mul(bits[0], b.const(1))
    for i in range(1, len(bits)):
        lc = b.add(lc, b.mul(bits[i], b.const(1 << i)))
    return b.signal(lc)


def is_zero(b: Builder, x):
    """circomlib IsZero: inv <-- in!=0 ? 1/in : 0; out <== -in*inv + 1."""
    zero, one = b.const(0), b.const(1)
    cond = b.op("Neq", x, zero)
    inv = b.signal(b.tern(cond, b.div(one, x), zero))
    return b.signal(b.add(b.mul(b.neg(x), inv), one))


def is_equal(b, x, y):
    return is_zero(b, b.sub(y, x))


def less_than(b: Builder, n, x, y):
    """circomlib LessThan(n): Num2Bits(n+1)(in0 + 2^n - in1); out <== 1 - bit[n]."""
    s = b.sub(b.add(x, b.const(1 << n)), y)
    bits = num2bits(b, s, n + 1)
    return b.signal(b.sub(b.const(1), bits[n]))


def switcher(b, sel, l, r):
    """circomlib Switcher: aux <== (R-L)*sel; outL <== aux + L; outR <== -aux + R."""
    aux = b.signal(b.mul(b.sub(r, l), sel))
    return b.signal(b.add(aux, l)), b.signal(b.add(b.neg(aux), r))


def mux1(b, sel, c0, c1):
    """circomlib Mux1: out <== (c1 - c0)*s + c0."""
    return b.signal(b.add(b.mul(b.sub(c1, c0), sel), c0))


def comp_constant(b: Builder, bits, ct):
    """circomlib CompConstant-shaped: 127 two-bit parts, each a few Mul/Add/Sub against constants,
    summed by a serial chain, then Num2Bits(135) of the sum; out = bit 127."""
    acc = None
    e = 1
    for i in range(127):
        clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
        slsb, smsb = bits[2 * i], bits[2 * i + 1]
        bb, aa = b.const((1 << 128) - e), b.const(e)
        if cmsb == 0 and clsb == 0:
            part = b.sub(b.add(b.mul(b.mul(smsb, slsb), b.neg(bb)), b.mul(smsb, bb)), b.neg(b.mul(slsb, bb)))
        elif cmsb == 0 and clsb == 1:
            part = b.sub(b.add(b.sub(b.mul(b.mul(aa, smsb), slsb), b.mul(aa, slsb)), b.mul(bb, smsb)), b.sub(b.mul(aa, smsb), aa))
        elif cmsb == 1 and clsb == 0:
            part = b.add(b.sub(b.mul(b.mul(bb, smsb), slsb), b.mul(aa, smsb)), aa)
        else:
            part = b.add(b.mul(b.mul(b.neg(aa), smsb), slsb), aa)
        b.signal(part)
        acc = part if acc is None else b.add(acc, part)
        e *= 2
    b.signal(acc)
    nb = num2bits(b, acc, 135)
    return nb[127]


# -- elliptic-curve-shaped gadgets (BabyJubjub: Montgomery A=168698, B=1; Edwards a=168700, d=168696)
def montgomery_add(b: Builder, p, q):
    """circomlib MontgomeryAdd: lamda <-- (y2-y1)/(x2-x1); x3 = B*l^2 - A - x1 - x2; y3 = l*(x1-x3) - y1."""
    (x1, y1), (x2, y2) = p, q
    lam = b.signal(b.div(b.sub(y2, y1), b.sub(x2, x1)))
    x3 = b.signal(b.sub(b.sub(b.sub(b.mul(b.const(1), b.mul(lam, lam)), b.const(168698)), x1), x2))
    y3 = b.signal(b.sub(b.mul(lam, b.sub(x1, x3)), y1))
    return x3, y3


def montgomery_double(b: Builder, p):
    """circomlib MontgomeryDouble: x1_2 = x1^2; lamda <-- (3*x1_2 + 2*A*x1 + 1)/(2*B*y1)."""
    x1, y1 = p
    x1_2 = b.signal(b.mul(x1, x1))
    num = b.add(b.add(b.mul(b.const(3), x1_2), b.mul(b.const(2 * 168698), x1)), b.const(1))
    lam = b.signal(b.div(num, b.mul(b.const(2), y1)))
    x3 = b.signal(b.sub(b.sub(b.mul(b.const(1), b.mul(lam, lam)), b.const(168698)), b.mul(b.const(2), x1)))
    y3 = b.signal(b.sub(b.mul(lam, b.sub(x1, x3)), y1))
    return x3, y3


def baby_add(b: Builder, p, q):
    """circomlib BabyAdd (twisted Edwards, two Divs)."""
    (x1, y1), (x2, y2) = p, q
    a, d = 168700, 168696
    beta = b.signal(b.mul(x1, y2))
    gamma = b.signal(b.mul(y1, x2))
    delta = b.signal(b.mul(b.add(b.mul(b.const(R - a), x1), y1), b.add(x2, y2)))
    tau = b.signal(b.mul(beta, gamma))
    one = b.const(1)
    xo = b.signal(b.div(b.add(beta, gamma), b.add(one, b.mul(b.const(d), tau))))
    yo = b.signal(b.div(b.sub(b.add(delta, b.mul(b.const(a), beta)), gamma), b.sub(one, b.mul(b.const(d), tau))))
    return xo, yo


def scalar_mul_any(b: Builder, bits, p):
    """EscalarMulAny-shaped: per bit one MontgomeryDouble + one MontgomeryAdd + two Mux1 (select by bit).
    Serial dependency chain of 2 Divs per bit."""
    acc = p
    dbl = p
    for bit in bits:
        dbl = montgomery_double(b, dbl)
        s = montgomery_add(b, acc, dbl)
        acc = (mux1(b, bit, acc[0], s[0]), mux1(b, bit, acc[1], s[1]))
    return acc


def scalar_mul_fix(b: Builder, bits, seed="fixbase"):
    """EscalarMulFix-shaped: 3-bit windows; each window = Mux3 over 8 constant points (bit products
    times constants, summed) followed by one MontgomeryAdd into the accumulator."""
    s = _field_stream(seed)
    acc = None
    for w in range(0, len(bits) - len(bits) % 3, 3):
        s0, s1, s2 = bits[w:w + 3]
        s10 = b.signal(b.mul(s1, s0))
        s20 = b.mul(s2, s0)
        s21 = b.mul(s2, s1)
        s210 = b.mul(s21, s0)
        pt = []
        for _coord in range(2):
            c = [next(s) for _ in range(8)]
            a210 = b.mul(b.const((c[7] - c[6] - c[5] + c[4] - c[3] + c[2] + c[1] - c[0]) % R), s210)
            a21 = b.mul(b.const((c[6] - c[4] - c[2] + c[0]) % R), s21)
            a20 = b.mul(b.const((c[5] - c[4] - c[1] + c[0]) % R), s20)
            a2 = b.mul(b.const((c[4] - c[0]) % R), s2)
            a10 = b.mul(b.const((c[3] - c[2] - c[1] + c[0]) % R), s10)
            a1 = b.mul(b.const((c[2] - c[0]) % R), s1)
            a0 = b.mul(b.const((c[1] - c[0]) % R), s0)
            o = b.add(b.add(b.add(b.add(b.add(b.add(b.add(a210, a21), a20), a2), a10), a1), a0), b.const(c[0]))
            pt.append(b.signal(o))
        pt = tuple(pt)
        acc = pt if acc is None else montgomery_add(b, acc, pt)
    return acc


# -- SHA-256 (real FIPS 180-4 function, circomlib gadget shapes) ---------------------------------------
_K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
_H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


class _Bits:
    """Bit-vector arithmetic where every bit is either a Python int (constant, folded as the
    reference's constant propagation would) or a node handle."""

    def __init__(self, b: Builder):
        self.b = b

    def _is_c(self, x):
        return not isinstance(x, _H)

    def lift(self, x):
        return self.b.const(x) if self._is_c(x) else x.h

    def mul(self, x, y):
        if self._is_c(x) and self._is_c(y):
            return x * y % R
        return _H(self.b.mul(self.lift(x), self.lift(y)))

    def add(self, x, y):
        if self._is_c(x) and self._is_c(y):
            return (x + y) % R
        return _H(self.b.add(self.lift(x), self.lift(y)))

    def sub(self, x, y):
        if self._is_c(x) and self._is_c(y):
            return (x - y) % R
        return _H(self.b.sub(self.lift(x), self.lift(y)))

    def sig(self, x):
        if not self._is_c(x):
            self.b.signal(x.h)
        return x

    def xor3(self, a, b_, c):
        """circomlib Xor3: mid = b*c; out = a*(1 - 2b - 2c + 4mid) + b + c - 2mid."""
        mid = self.sig(self.mul(b_, c))
        t = self.add(self.sub(self.sub(1, self.mul(2, b_)), self.mul(2, c)), self.mul(4, mid))
        return self.sig(self.sub(self.add(self.add(self.mul(a, t), b_), c), self.mul(2, mid)))

    def ch(self, a, b_, c):
        """circomlib Ch_t: out = a*(b-c) + c."""
        return self.sig(self.add(self.mul(a, self.sub(b_, c)), c))

    def maj(self, a, b_, c):
        """circomlib Maj_t: mid = b*c; out = a*(b + c - 2mid) + mid."""
        mid = self.sig(self.mul(b_, c))
        return self.sig(self.add(self.mul(a, self.sub(self.add(b_, c), self.mul(2, mid))), mid))

    def binsum(self, words, nbits=32):
        """circomlib BinSum: lin = sum_k sum_j in[j][k]*2^k (serial chain); out[k] <-- (lin >> k) & 1.
        words: list of 32-entry LSB-first bit lists. Returns the low `nbits` bits of the sum."""
        lin = 0
        for k in range(nbits):
            for w in words:
                lin = self.add(lin, self.mul(w[k], 1 << k))
        if self._is_c(lin):
            return [(lin >> k) & 1 for k in range(nbits)]
        out = []
        one = self.b.const(1)
        for k in range(nbits):
            out.append(self.sig(_H(self.b.op("Band", self.b.op("Shr", lin.h, self.b.const(k)), one))))
        return out


class _H:
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h


def _rotr(w, n):
    return [w[(i + n) % 32] for i in range(32)]


def _shr(w, n):
    return [w[i + n] if i + n < 32 else 0 for i in range(32)]


def _const_word(v):
    return [(v >> i) & 1 for i in range(32)]


def sha256_bits(b: Builder, msg_bits):
    """SHA-256 of a bit string (MSB-first per byte, as circomlib Sha256), len(msg_bits) % 8 == 0.
    msg_bits: node handles. Returns 256 output bit handles/ints (MSB-first, like circomlib `out`)."""
    bv = _Bits(b)
    n = len(msg_bits)
    padded = [_H(x) for x in msg_bits] + [1]
    while (len(padded) + 64) % 512:
        padded.append(0)
    padded += [(n >> (63 - i)) & 1 for i in range(64)]
    H = [_const_word(h) for h in _H0]  # LSB-first internally
    for blk in range(len(padded) // 512):
        w = []
        for t in range(16):
            be = padded[blk * 512 + 32 * t: blk * 512 + 32 * t + 32]  # MSB first
            w.append(list(reversed(be)))
        for t in range(16, 64):
            x = w[t - 2]
            s1 = [bv.xor3(a, c, d) for a, c, d in zip(_rotr(x, 17), _rotr(x, 19), _shr(x, 10))]
            x = w[t - 15]
            s0 = [bv.xor3(a, c, d) for a, c, d in zip(_rotr(x, 7), _rotr(x, 18), _shr(x, 3))]
            w.append(bv.binsum([s1, w[t - 7], s0, w[t - 16]]))
        a, bb, c, d, e, f, g, h = H
        for t in range(64):
            S1 = [bv.xor3(x, y, z) for x, y, z in zip(_rotr(e, 6), _rotr(e, 11), _rotr(e, 25))]
            chv = [bv.ch(x, y, z) for x, y, z in zip(e, f, g)]
            S0 = [bv.xor3(x, y, z) for x, y, z in zip(_rotr(a, 2), _rotr(a, 13), _rotr(a, 22))]
            mj = [bv.maj(x, y, z) for x, y, z in zip(a, bb, c)]
            kt = _const_word(_K[t])
            # circomlib T1 = BinSum(5)(h, S1, ch, k, w); T2 = BinSum(2)(S0, maj); e' = d + T1; a' = T1 + T2
            T1 = bv.binsum([h, S1, chv, kt, w[t]])
            T2 = bv.binsum([S0, mj])
            h, g, f = g, f, e
            e = bv.binsum([d, T1])
            d, c, bb = c, bb, a
            a = bv.binsum([T1, T2])
        H = [bv.binsum([x, y]) for x, y in zip(H, [a, bb, c, d, e, f, g, h])]
    out = []
    for wd in H:
        out += list(reversed(wd))
    return [o.h if isinstance(o, _H) else b.const(o) for o in out]


# -- whole-circuit generators --------------------------------------------------------------------------
def build_circuit1():
    """test_circuits/circuit1.circom: c <== a*b + 2; witness [1, c, a, b] (SURVEY.md 8(c) fixture)."""
    b = Builder()
    two = b.const(2)
    (a,) = b.input("a")
    (bb,) = b.input("b")
    c = b.add(b.mul(a, bb), two)
    b.signal(c); b.signal(a); b.signal(bb)
    return b


def build_circuit2():
    """test_circuits/circuit2.circom-shaped: c <== IsZero(a) * b + 2   (f2(30) = 2 is folded at compile time)."""
    b = Builder()
    (a,) = b.input("a")
    (bb,) = b.input("b")
    e = is_zero(b, a)
    c = b.add(b.mul(e, bb), b.const(2))
    b._witness = [b._witness[0], c, a, bb] + [w for w in b._witness[1:]]
    return b


def build_circuit3():
    """test_circuits/circuit3.circom-shaped: d = [a, b]; c <== d[0] * d[1] + 3 through an anonymous component."""
    b = Builder()
    (a,) = b.input("a")
    (bb,) = b.input("b")
    c = b.add(b.mul(a, bb), b.const(3))
    b.signal(c); b.signal(a); b.signal(bb)
    return b


def build_circuit4():
    """test_circuits/circuit4.circom-shaped: n = Num2Bits(2)(a); c <== n.out[0] * n.out[1] + b."""
    b = Builder()
    (a,) = b.input("a")
    (bb,) = b.input("b")
    bits = num2bits(b, a, 2)
    c = b.add(b.mul(bits[0], bits[1]), bb)
    b._witness = [b._witness[0], c, a, bb] + [w for w in b._witness[1:]]
    return b


def build_circuit6():
    """test_circuits/circuit6_num2bits.circom-shaped: idBits = Num2Bits(256)(a); c <== Bits2Num(216)(idBits[16..232)).
    (Num2Bits(256) shifts by up to 255: amounts >= 254 give 0 in the reference, src/graph.rs:642-646.)"""
    b = Builder()
    (a,) = b.input("a")
    bits = num2bits(b, a, 256)
    out = bits2num(b, bits[16:256 - 16 - 8])
    b._witness = [b._witness[0], out, a] + [w for w in b._witness[1:] if w != out]
    return b


def build_poseidon_circomlib(n_inputs=2):
    """circomlib's Poseidon(n_inputs) with its real constants: witness = [1, hash, inputs...]"""
    b = Builder()
    ins = b.input("inputs", n_inputs)
    b.signal(poseidon(b, ins, signals=False, circomlib=True))
    for h in ins:
        b.signal(h)
    return b


def build_poseidon(n_inputs=1, name="a"):
    """circuit5_poseidon-shaped: Poseidon(n) over one input array `a` (BASELINE config 1)."""
    b = Builder()
    ins = b.input(name, n_inputs)
    out = poseidon(b, ins)
    # circom witness order: 1, outputs, inputs, intermediates; we keep generation order after out/in
    b._witness = [b._witness[0], out] + ins + [w for w in b._witness[1:] if w != out]
    return b


def build_sha256(n_bits=512):
    """circuit8_sha256_512-shaped: in[n_bits] -> out[256] (BASELINE config 3)."""
    b = Builder()
    ins = b.input("in", n_bits)
    hs = sha256_bits(b, ins)
    b._witness = [b._witness[0]] + hs + ins + b._witness[1:]
    return b


def build_gadgets():
    """Small graph touching every gadget (unit-scale differential tests)."""
    b = Builder()
    (x,) = b.input("x")
    (y,) = b.input("y")
    arr = b.input("arr", 4)
    bits = num2bits(b, x, 16)
    b.signal(bits2num(b, bits))
    b.signal(is_zero(b, y))
    b.signal(is_zero(b, b.sub(x, x)))
    b.signal(less_than(b, 32, arr[0], arr[1]))
    l, r = switcher(b, bits[0], arr[2], arr[3])
    b.signal(poseidon(b, [l, r]))
    p = (arr[0], arr[1])
    q = montgomery_double(b, p)
    s = montgomery_add(b, p, q)
    e = baby_add(b, s, q)
    b.signal(e[0]); b.signal(e[1])
    for name in ("Idiv", "Mod", "Lt", "Gt", "Leq", "Geq", "Eq", "Neq", "Land", "Lor", "Bor", "Bxor", "Band"):
        b.signal(b.op(name, x, y))
    b.signal(b.op("Shl", b.op("Band", x, b.const(0xFFFF)), b.const(7)))
    return b


def smt_verifier(b: Builder, n_levels, enabled, root, siblings, old_key, old_value, is_old0, key, value, fnc):
    """circomlib SMTVerifier-shaped: Num2Bits(254) of key, per-level IsZero on siblings + level-select
    logic (Mul/Sub chains), bottom-up chain of `n_levels` Poseidon(2) hashes over Switcher outputs, two
    leaf hashes Poseidon(3), final root comparison (IsEqual)."""
    one = b.const(1)
    h1old = poseidon(b, [old_key, old_value, one])
    h1new = poseidon(b, [key, value, one])
    n2b_new = num2bits(b, key, 254)
    # SMTLevIns: levIns[i] via IsZero(siblings[i]) and a serial done[] chain
    isz = [is_zero(b, s) for s in siblings]
    lev_ins = [None] * n_levels
    done = [None] * (n_levels - 1)
    lev_ins[n_levels - 1] = b.signal(b.sub(one, isz[n_levels - 2]))
    done[n_levels - 2] = lev_ins[n_levels - 1]
    for i in range(n_levels - 2, 0, -1):
        lev_ins[i] = b.signal(b.mul(b.sub(one, done[i]), b.sub(one, isz[i - 1])))
        done[i - 1] = b.signal(b.add(lev_ins[i], done[i]))
    lev_ins[0] = b.signal(b.sub(one, done[0]))
    # SMTVerifierSM chain (top-down state machine, 5 state signals per level)
    st_top, st_i0, st_iold, st_inew, st_na = enabled, b.const(0), b.const(0), b.const(0), b.sub(one, enabled)
    sms = []
    for i in range(n_levels):
        prev_top_lev_ins = b.signal(b.mul(st_top, lev_ins[i]))
        n_top = b.signal(b.sub(st_top, prev_top_lev_ins))
        aux1 = b.signal(b.mul(prev_top_lev_ins, is_old0))
        n_iold = b.signal(b.mul(b.sub(prev_top_lev_ins, aux1), b.sub(one, fnc)))
        n_inew = b.signal(b.sub(b.sub(prev_top_lev_ins, aux1), n_iold))
        n_i0 = aux1
        n_na = b.signal(b.add(b.add(b.add(st_na, st_inew), st_iold), st_i0))
        sms.append((n_top, n_i0, n_iold, n_inew, n_na))
        st_top, st_i0, st_iold, st_inew, st_na = n_top, n_i0, n_iold, n_inew, n_na
    # SMTVerifierLevel chain, bottom-up
    child = b.const(0)
    for i in range(n_levels - 1, -1, -1):
        n_top, n_i0, n_iold, n_inew, _ = sms[i]
        l, r = switcher(b, n2b_new[i], child, siblings[i])
        ph = poseidon(b, [l, r])
        aux0 = b.signal(b.mul(ph, n_top))
        aux1 = b.signal(b.mul(h1old, n_iold))
        child = b.signal(b.add(b.add(aux0, aux1), b.mul(h1new, n_inew)))
    ok = is_equal(b, child, root)
    return b.signal(b.mul(ok, enabled))


AUTHV2_INPUTS = [  # key schema of test_circuits/circuit9_authV2_inputs.json (21 keys, 169 scalars)
    ("genesisID", 1), ("profileNonce", 1), ("authClaim", 8), ("authClaimIncMtp", 40),
    ("authClaimNonRevMtp", 40), ("authClaimNonRevMtpAuxHi", 1), ("authClaimNonRevMtpAuxHv", 1),
    ("authClaimNonRevMtpNoAux", 1), ("challenge", 1), ("challengeSignatureR8x", 1),
    ("challengeSignatureR8y", 1), ("challengeSignatureS", 1), ("claimsTreeRoot", 1), ("revTreeRoot", 1),
    ("rootsTreeRoot", 1), ("state", 1), ("gistRoot", 1), ("gistMtp", 64), ("gistMtpAuxHi", 1),
    ("gistMtpAuxHv", 1), ("gistMtpNoAux", 1)]


def build_authv2_class(scale=1.0, levels=None, ladder_bits=None):
    """authV2-class composite (BASELINE configs 2 and 4): the reference's authV2 input schema;
    three SMT verifiers (40/40/64 levels, one Poseidon(2) per level + leaf hashes), claim hashing,
    state check, and an EdDSA-Poseidon-shaped block (Num2Bits(254), CompConstant, Poseidon(5),
    three BabyDbl, EscalarMulAny over 254 bits = 508 chained Divs, EscalarMulFix over 253 bits).
    `scale` < 1 shrinks level counts / scalar widths proportionally (for fast tests); `levels` = (claims, rev, gist) tree
    depths (at most 40, 40, 64: the input schema) and `ladder_bits` = width of the scalar in EscalarMulAny give structurally
    different members of the class (tools/gpu_robustness.py)."""
    b = Builder()
    I = {name: b.input(name, n) for name, n in AUTHV2_INPUTS}
    g = lambda k: I[k][0]
    one, zero = b.const(1), b.const(0)
    lv = lambda n: max(3, int(round(n * scale)))
    nb = lambda n: max(9, int(round(n * scale)))
    l_claims, l_rev, l_gist = levels if levels else (lv(40), lv(40), lv(64))
    assert l_claims <= 40 and l_rev <= 40 and l_gist <= 64
    # claim hashes: hi = Poseidon(4)(slots 0..3), hv = Poseidon(4)(slots 4..7), hash = Poseidon(2)
    claim = I["authClaim"]
    hi = poseidon(b, claim[0:4])
    hv = poseidon(b, claim[4:8])
    b.signal(poseidon(b, [hi, hv]))
    # claim header checks: Num2Bits(256-ish) of slot 0, flags
    c0bits = num2bits(b, claim[0], nb(254))
    b.signal(b.mul(c0bits[3 % len(c0bits)], c0bits[5 % len(c0bits)]))
    # auth claim inclusion in claims tree
    smt_verifier(b, l_claims, one, g("claimsTreeRoot"), I["authClaimIncMtp"][:l_claims], zero, zero, zero, hi, hv, zero)
    # non-revocation (non-inclusion) in rev tree
    rev_nonce = bits2num(b, c0bits[:min(64, len(c0bits))])
    smt_verifier(b, l_rev, one, g("revTreeRoot"), I["authClaimNonRevMtp"][:l_rev], g("authClaimNonRevMtpAuxHi"),
                 g("authClaimNonRevMtpAuxHv"), g("authClaimNonRevMtpNoAux"), rev_nonce, zero, one)
    # state = Poseidon(3)(claimsTreeRoot, revTreeRoot, rootsTreeRoot); compare with `state`
    st = poseidon(b, [g("claimsTreeRoot"), g("revTreeRoot"), g("rootsTreeRoot")])
    b.signal(is_equal(b, st, g("state")))
    # GIST inclusion / non-inclusion (64 levels), key = Poseidon(1)(genesisID)
    gkey = poseidon(b, [g("genesisID")])
    smt_verifier(b, l_gist, one, g("gistRoot"), I["gistMtp"][:l_gist], g("gistMtpAuxHi"), g("gistMtpAuxHv"),
                 g("gistMtpNoAux"), gkey, g("state"), b.signal(is_zero(b, g("profileNonce"))))
    # profile id: Poseidon(2)(genesisID, nonce) selected by IsZero(nonce)
    prof = poseidon(b, [g("genesisID"), g("profileNonce")])
    b.signal(mux1(b, is_zero(b, g("profileNonce")), prof, g("genesisID")))
    # EdDSA-Poseidon verify of `challenge` under pubkey (claim[2], claim[3])
    ax, ay = claim[2], claim[3]
    sbits = num2bits(b, g("challengeSignatureS"), nb(253))
    if len(sbits) >= 254:
        comp_constant(b, sbits + [zero], 2736030358979909402780800718157159386076813972158567259200215660948447373040)
    hh = poseidon(b, [g("challengeSignatureR8x"), g("challengeSignatureR8y"), ax, ay, g("challenge")])
    hbits = num2bits(b, hh, nb(254))
    if len(hbits) >= 254:
        comp_constant(b, hbits, (R - 1))  # Num2Bits_strict alias check
    d1 = baby_add(b, (ax, ay), (ax, ay))
    d2 = baby_add(b, d1, d1)
    d3 = baby_add(b, d2, d2)
    is_z = is_zero(b, d3[0])
    # Edwards -> Montgomery: u = (1+y)/(1-y), v = u/x
    u = b.signal(b.div(b.add(one, d3[1]), b.sub(one, d3[1])))
    v = b.signal(b.div(u, d3[0]))
    right = scalar_mul_any(b, hbits[:ladder_bits] if ladder_bits else hbits, (u, v))
    rx = b.signal(b.div(right[0], right[1]))
    ry = b.signal(b.div(b.sub(right[0], one), b.add(right[0], one)))
    right2 = baby_add(b, (g("challengeSignatureR8x"), g("challengeSignatureR8y")), (rx, ry))
    left = scalar_mul_fix(b, sbits)
    lx = b.signal(b.div(left[0], left[1]))
    ly = b.signal(b.div(b.sub(left[0], one), b.add(left[0], one)))
    b.signal(is_equal(b, lx, right2[0]))
    b.signal(b.mul(is_equal(b, ly, right2[1]), b.sub(one, is_z)))
    return b


def authv2_reference_inputs():
    """Inputs of test_circuits/circuit9_authV2_inputs.json as {key: [ints]} (data fixture; values only)."""
    z40, z64 = [0] * 40, [0] * 64
    return {
        "genesisID": [26109404700696283154998654512117952420503675471097392618762221546565140481],
        "profileNonce": [0],
        "authClaim": [80551937543569765027552589160822318028, 0,
                      17640206035128972995519606214765283372613874593503528180869261482403155458945,
                      20634138280259599560273310290025659992320584624461316485434108770067472477956,
                      15930428023331155902, 0, 0, 0],
        "authClaimIncMtp": z40, "authClaimNonRevMtp": z40,
        "authClaimNonRevMtpAuxHi": [0], "authClaimNonRevMtpAuxHv": [0], "authClaimNonRevMtpNoAux": [1],
        "challenge": [10],
        "challengeSignatureR8x": [2436614617352067078274240654647841101298221663194055411539273018411814965042],
        "challengeSignatureR8y": [18597752099468941062473075570139025288787892531282848931228194191266230422780],
        "challengeSignatureS": [1642466479083925938589665711747519202726798003514101885795868643287098549939],
        "claimsTreeRoot": [9860409408344985873118363460916733946840214387455464863344022463808838582364],
        "revTreeRoot": [0], "rootsTreeRoot": [0],
        "state": [1648710229725601204870171311149827592640182384459240511403224642152766848235],
        "gistRoot": [11098939821764568131087645431296528907277253709936443029379587475821759259406],
        "gistMtp": z64,
        "gistMtpAuxHi": [27918766665310231445021466320959318414450284884582375163563581940319453185],
        "gistMtpAuxHv": [20177832565449474772630743317224985532862797657496372535616634430055981993180],
        "gistMtpNoAux": [0],
    }


def build_random_dag(seed, n_ops=400, n_inputs=6, ops=None, panic_free=True, parts=1):
    """Random DAG fuzzer over every evaluable op (all 20 DuoOps except Pow, Neg, TernCond).
    Operands are drawn mostly from recent nodes; small-value nodes (masks, booleans, small constants)
    are mixed in so that shifts / compares / bit ops see interesting ranges.  With `panic_free` the
    operand choice avoids the reference's two panic edges by construction (Shl only of values masked
    to < 2^100 by < 128 bits; Bor/Bxor only of values masked to 253 bits).  `parts` > 1: that many independent DAGs of
    n_ops operations each over the same inputs and a few shared operations next to them (what programs of several
    streams split over wavefronts)."""
    rnd = random.Random(seed)
    b = Builder()
    ins = b.input("in", n_inputs)
    hubs = [b.add(ins[1], ins[2]), b.mul(ins[2], ins[3])] if parts > 1 else []
    edge = [0, 1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 191, 192, 193, 253, 254, 255, 256, R - 1, R - 2,
            R // 2, R // 2 + 1, R // 2 + 2, 1 << 253, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 200]
    allops = ops or ["Mul", "Div", "Add", "Sub", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq", "Land",
                     "Lor", "Shl", "Shr", "Bor", "Band", "Bxor", "Neg", "TernCond"]

    def pick(lst=None):
        lst = lst or pool
        if rnd.random() < 0.15:
            return b.const(rnd.choice(edge) if rnd.random() < 0.7 else rnd.randrange(R))
        if rnd.random() < 0.7:
            return lst[-1 - min(len(lst) - 1, int(rnd.expovariate(0.15)))]
        return rnd.choice(lst)

    for step in range(n_ops * parts):
        if step % n_ops == 0:  # (a new part starts from the inputs and the shared operations)
            if step:
                b.signal(pool[-1])
            pool = list(ins) + hubs  # general values
            small = []               # values known < 2^100
            m253 = []                # values known < 2^253
        op = rnd.choice(allops)
        if op == "Neg":
            h = b.neg(pick())
        elif op == "TernCond":
            c = pick(small) if small and rnd.random() < 0.5 else pick()
            h = b.tern(c, pick(), pick())
        elif op == "Shl":
            if panic_free:
                if not small:
                    small.append(b.op("Band", pick(), b.const((1 << 100) - 1)))
                    pool.append(small[-1])
                h = b.op("Shl", rnd.choice(small), b.const(rnd.choice([0, 1, 7, 31, 32, 33, 63, 64, 65, 100, 127])))
            else:
                h = b.op("Shl", pick(), pick())
        elif op == "Shr":
            sh = b.const(rnd.choice([0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 253, 254, 300])) \
                if rnd.random() < 0.8 else pick()
            h = b.op("Shr", pick(), sh)
        elif op in ("Bor", "Bxor") and panic_free:
            while len(m253) < 2:
                m253.append(b.op("Band", pick(), b.const((1 << 253) - 1)))
                pool.append(m253[-1])
            h = b.op(op, rnd.choice(m253), rnd.choice(m253))
            m253.append(h)
        elif op == "Band":
            mv = rnd.choice([1, 0xFF, (1 << 64) - 1, (1 << 100) - 1, (1 << 253) - 1])
            use_mask = rnd.random() < 0.6
            h = b.op("Band", pick(), b.const(mv) if use_mask else pick())
            if use_mask and mv < (1 << 101):
                small.append(h)
            if use_mask:
                m253.append(h)
        elif op in ("Idiv", "Mod"):
            den = pick(small) if small and rnd.random() < 0.5 else pick()
            h = b.op(op, pick(), den)
        else:
            h = b.op(op, pick(), pick())
        pool.append(h)
        if op in ("Eq", "Neq", "Lt", "Gt", "Leq", "Geq", "Land", "Lor"):
            small.append(h)
        if rnd.random() < 0.5:
            b.signal(h)
    b.signal(pool[-1])
    return b


def build_bigint_class(k=8, n_bits=64, rounds=4, seed="bigint"):
    """bigint / long_div-class synthetic graph (BASELINE config 5; the reference front-end cannot compile such
    circuits at this commit, README.md:21, so this is synthetic by necessity): `rounds` iterations of
      * schoolbook k x k limb multiplication with witness-hint carries  (Mul, Add, Idiv by 2^n, Mod 2^n),
      * long division of the 2k-limb product by a single limb          (Mul, Add, Idiv, Mod per limb),
      * limb-wise comparison and conditional subtraction               (Lt, Sub, TernCond),
    chained so that every round depends on the previous one.  Values stay below 2^(2n+log k), far below r.
    Node count ~ rounds * (6 k^2 + 12 k)."""
    b = Builder()
    a_in = b.input("a", k)
    b_in = b.input("b", k)
    (d_in,) = b.input("d")
    base = b.const(1 << n_bits)
    mask = b.const((1 << n_bits) - 1)
    one = b.const(1)
    zero = b.const(0)
    # normalise inputs to n_bits limbs (they may be arbitrary field elements in synthetic batches)
    x = [b.signal(b.op("Band", v, mask)) for v in a_in]
    y = [b.signal(b.op("Band", v, mask)) for v in b_in]
    d = b.signal(b.add(b.op("Band", d_in, mask), one))  # divisor limb in [1, 2^n]
    for _ in range(rounds):
        # product columns
        cols = [None] * (2 * k)
        for i in range(k):
            for j in range(k):
                pr = b.mul(x[i], y[j])
                cols[i + j] = pr if cols[i + j] is None else b.add(cols[i + j], pr)
        cols[2 * k - 1] = zero
        # carry propagation: limb = col % 2^n ; carry = col \ 2^n
        carry = zero
        prod = []
        for c in range(2 * k):
            t = b.add(cols[c], carry)
            prod.append(b.signal(b.op("Mod", t, base)))
            carry = b.signal(b.op("Idiv", t, base))
        # long division of prod by the single limb d (most significant limb first)
        rem = zero
        quo = [None] * (2 * k)
        for c in range(2 * k - 1, -1, -1):
            t = b.add(b.mul(rem, base), prod[c])
            quo[c] = b.signal(b.op("Idiv", t, d))
            rem = b.signal(b.op("Mod", t, d))
        # compare-and-select: next x = quo low limbs, next y = limb-wise min(y, quo high) + rem folded in
        nx, ny = [], []
        for i in range(k):
            lt = b.signal(b.op("Lt", quo[i + k], y[i]))
            sel = b.signal(b.tern(lt, quo[i + k], y[i]))
            nx.append(b.signal(b.op("Band", b.add(quo[i], rem), mask)))
            ny.append(b.signal(b.op("Band", b.add(sel, one), mask)))
        x, y = nx, ny
        d = b.signal(b.add(b.op("Band", b.add(d, rem), mask), one))
    return b


def build_chain_heavy(seed, n_chains=12, n_inputs=5):
    """Long Add / Mul / mixed chains with constants (`lc += c * x`), repeated operands, witness elements in the middle of
    chains and unused tails: the shapes the compiler's exact rewrites (tree-height reduction, shared subexpressions,
    dead-node elimination, linear riders, request/collect divisions) act on."""
    rnd = random.Random(seed)
    b = Builder()
    ins = b.input("in", n_inputs)
    pool = list(ins)
    for _chain in range(n_chains):
        op = rnd.choice(["add", "mul", "mixed"])
        acc = rnd.choice(pool)
        for _step in range(rnd.randrange(3, 14)):
            x = rnd.choice(pool) if rnd.random() < 0.7 else b.const(rnd.choice([0, 1, 2, R - 1, rnd.randrange(R)]))
            if rnd.random() < 0.3:
                x = b.mul(b.const(rnd.randrange(1, R)), x)
            kind = op if op != "mixed" else rnd.choice(["add", "mul", "sub", "div"])
            acc = {"add": b.add, "mul": b.mul, "sub": b.sub, "div": b.div}[kind](acc, x)
            if rnd.random() < 0.25:
                b.signal(acc)
            if rnd.random() < 0.3:
                pool.append(acc)
        if rnd.random() < 0.7:
            b.signal(acc)
    return b


def build_limb_product_variants(seed):
    """Schoolbook limb products in the shapes the convolution rewrite has to tell apart (rewrite.cc detect_convolutions): complete
    k x k blocks, rectangular k x m blocks, squares (x * x: the cross products are shared nodes), blocks that share a factor
    vector, columns with an extra addend, products that are witness elements, column trees of different shapes, one limb
    missing from a block -- each followed by a carry chain so that the graph is a limb graph.  Whatever the compiler makes of
    them, the witness must be exact."""
    rnd = random.Random(seed)
    b = Builder()
    n_bits = rnd.choice([64, 64, 64, 32, 16, 63])
    mask, base, zero = b.const((1 << n_bits) - 1), b.const(1 << n_bits), b.const(0)
    n_vec = rnd.randrange(2, 5)
    klen = [rnd.choice([2, 3, 4, 5, 8]) for _ in range(n_vec)]
    vecs = [[b.op("Band", v, mask) for v in b.input("v%d" % i, klen[i])] for i in range(n_vec)]
    (extra,) = b.input("e")
    extra = b.op("Band", extra, mask)
    for _blk in range(rnd.randrange(1, 5)):
        shape = rnd.choice(["square_block", "square_block", "rect", "self", "shared", "extra", "signal", "hole", "chain"])
        xi = rnd.randrange(n_vec)
        yi = xi if shape == "self" else rnd.randrange(n_vec)
        x, y = vecs[xi], vecs[yi]
        if shape in ("square_block", "extra", "signal", "hole", "chain", "shared") and len(y) != len(x):
            y = y[:len(x)] if len(y) > len(x) else y + [b.op("Band", b.add(v, extra), mask) for v in x[len(y):]]
        cols = [None] * (len(x) + len(y) - 1)
        order = [(i, j) for i in range(len(x)) for j in range(len(y))]
        if rnd.random() < 0.5:
            rnd.shuffle(order)
        hole = rnd.choice(order) if shape == "hole" else None
        for i, j in order:
            if (i, j) == hole:
                continue
            pr = b.mul(x[i], y[j]) if rnd.random() < 0.5 else b.mul(y[j], x[i])
            if shape == "signal" and rnd.random() < 0.15:
                b.signal(pr)
            if cols[i + j] is None:
                cols[i + j] = pr
            elif shape == "chain" or rnd.random() < 0.7:
                cols[i + j] = b.add(cols[i + j], pr)
            else:
                cols[i + j] = b.add(pr, cols[i + j])
        if shape == "extra":
            c = rnd.randrange(len(cols))
            cols[c] = b.add(cols[c], extra)
        carry = zero
        outs = []
        for c in range(len(cols)):
            t = b.add(cols[c], carry) if cols[c] is not None else carry
            outs.append(b.signal(b.op("Mod", t, base)))
            carry = b.signal(b.op("Idiv", t, base))
        if rnd.random() < 0.5:   # the product's limbs are the next block's factors
            vecs[rnd.randrange(n_vec)] = outs[:rnd.choice([2, 3, 4, len(outs)])]
    return b


def build_limb_graph_with_divisions(k=6, rounds=2):
    """A limb graph (schoolbook product, carry chain, long division by one limb) that also holds FIELD divisions: programs for
    divider waves keep the limb arithmetic unfused (the interpreter instances with scan / convolution paths run without divider
    waves), programs without them run the divisions in line beside scan and convolution bundles -- both must be exact, and the
    cost model picks between them."""
    b = Builder()
    xs, ys = b.input("x", k), b.input("y", k)
    (dv,) = b.input("d")
    m, base, zero, one = b.const((1 << 64) - 1), b.const(1 << 64), b.const(0), b.const(1)
    x = [b.op("Band", v, m) for v in xs]
    y = [b.op("Band", v, m) for v in ys]
    d = b.add(b.op("Band", dv, m), one)
    for _ in range(rounds):
        cols = [None] * (2 * k - 1)
        for i in range(k):
            for j in range(k):
                pr = b.mul(x[i], y[j])
                cols[i + j] = pr if cols[i + j] is None else b.add(cols[i + j], pr)
        carry, prod = zero, []
        for c in range(2 * k - 1):
            t = b.add(cols[c], carry)
            prod.append(b.signal(b.op("Mod", t, base)))
            carry = b.signal(b.op("Idiv", t, base))
        rem, quo = zero, []
        for c in range(2 * k - 2, -1, -1):
            t = b.add(b.mul(rem, base), prod[c])
            quo.append(b.signal(b.op("Idiv", t, d)))
            rem = b.signal(b.op("Mod", t, d))
        inv = [b.signal(b.div(b.add(quo[i], one), b.add(prod[i], one))) for i in range(3)]
        x = [b.op("Band", b.add(quo[i], inv[i % 3]), m) for i in range(k)]
        y = [b.op("Band", b.add(prod[i], rem), m) for i in range(k)]
    return b


def build_limb_chains(n_bits=64, k_bits=64, steps=10, chains=2, mask_inputs=False, fork=False):
    """Serial limb recurrences on operands that come straight from the inputs (any field element, unless mask_inputs): per
    chain a carry chain `t = x + carry; limb = t % 2^n; carry = t \\ 2^n` and a remainder chain `t = rem * 2^k + x;
    q = t \\ d; rem = t % d` -- the idioms of limb-wise big-integer circuits that the compiler runs as scan bundles, here
    with every shift / base width and with operands outside the limb range (the general paths of the kernels).  fork:
    some accumulators feed two steps (a chain that splits), and a step's x is another step's output."""
    b = Builder()
    xs = b.input("x", steps * chains)
    acc0 = b.input("acc", chains)
    ds = b.input("d", chains)
    base_n, base_k = b.const(1 << n_bits), b.const(1 << k_bits)
    if mask_inputs:
        m = b.const((1 << min(n_bits, k_bits, 64)) - 1)
        xs = [b.op("Band", v, m) for v in xs]
        acc0 = [b.op("Band", v, m) for v in acc0]
        ds = [b.op("Band", v, m) for v in ds]
    for ch in range(chains):
        carry = acc0[ch]
        limbs = []
        for c in range(steps):
            x = xs[ch * steps + c]
            if fork and c == steps // 2 and limbs:
                x = limbs[0]                      # (an earlier step's output as this step's x)
            t = b.add(x, carry)
            limbs.append(b.signal(b.op("Mod", t, base_n)))
            nxt = b.signal(b.op("Idiv", t, base_n))
            if fork and c == steps // 3:
                t2 = b.add(xs[ch * steps], nxt)   # a second step on the same accumulator
                b.signal(b.op("Mod", t2, base_n))
                b.signal(b.op("Idiv", t2, base_n))
            carry = nxt
        rem = acc0[ch]
        for c in range(steps):
            t = b.add(b.mul(rem, base_k), xs[ch * steps + (steps - 1 - c)])
            b.signal(b.op("Idiv", t, ds[ch]))
            rem = b.signal(b.op("Mod", t, ds[ch]))
    return b


# ---- zk-email RSA / long_div-class (BASELINE config 5's named class) -------------------------------------------------------
# The reference front-end cannot compile such circuits at this commit (README.md:21: "we plan to add support ... long_div"), so
# the graph can only be synthetic.  What is restated here are the PUBLIC witness-hint algorithms of circom-bigint as zk-email's
# RSA verifier uses them (bigint_func.circom: long_scalar_mult, long_sub, long_gt, short_div_norm, short_div, long_div, and the
# schoolbook product with `% 2^n` / `\ 2^n` carries; fp.circom FpMul: q, r <-- long_div(a * b, p) with Num2Bits range checks),
# laid out as a symbolic executor would emit them node by node: every circom `var` expression is one Op node in source
# order with no algebraic simplification (`0 + x` stays an Add node, as in the reference, whose `propagate` only folds
# operations on two constants, graph.rs:394-428), and every signal-dependent branch is predicated -- both arms are ordinary
# earlier nodes and a TernCond selects (what the reference does for the branches it supports):
#     if (c) { v = A } else { v = B }                           ->  v = TernCond(c, A, B)
#     for (i = k-1 .. 0) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; } return 0
#                                                               ->  res = 0; for i = 0 .. k-1: res = TernCond(a[i] > b[i], 1, TernCond(a[i] < b[i], 0, res))
# Loop-invariant calls (the normalisation of the divisor in short_div: `scale`, `norm_b`) appear once, as they do behind the
# reference's value numbering (graph.rs:540-578), which every reference-built `.bin` has been through.
def _lsm(b, n_base, kk, a, bb, zero):
    """long_scalar_mult(n, kk, a, bb) -> kk + 1 registers: temp = out[i] + a * bb[i]; out[i] = temp % 2^n; out[i+1] = out[i+1] + temp \\ 2^n"""
    out = [zero] * (kk + 1)
    for i in range(kk):
        temp = b.add(out[i], b.mul(a, bb[i]))
        out[i] = b.op("Mod", temp, n_base)
        out[i + 1] = b.add(out[i + 1], b.op("Idiv", temp, n_base))
    return out


def _long_gt(b, kk, x, y, zero, one):
    """long_gt(n, kk, x, y): 1 iff x > y as kk-register integers (most significant differing register decides)"""
    res = zero
    for i in range(kk):
        res = b.tern(b.op("Gt", x[i], y[i]), one, b.tern(b.op("Lt", x[i], y[i]), zero, res))
    return res


def _long_sub(b, n_base, kk, x, y, zero, one):
    """long_sub(n, kk, x, y) -> kk registers of x - y (x >= y), borrow chain as the function's if / else per register"""
    diff, borrow = [], None
    for i in range(kk):
        if i == 0:
            c = b.op("Geq", x[i], y[i])
            d_then = b.sub(x[i], y[i])
            d_else = b.add(b.sub(x[i], y[i]), n_base)
        else:
            c = b.op("Geq", x[i], b.add(y[i], borrow))
            d_then = b.sub(b.sub(x[i], y[i]), borrow)
            d_else = b.sub(b.sub(b.add(n_base, x[i]), y[i]), borrow)
        diff.append(b.tern(c, d_then, d_else))
        borrow = b.tern(c, zero, one)
    return diff


def _short_div_norm(b, n_base, n_max, kk, a, bb, zero, one, two):
    """short_div_norm(n, kk, a[kk+1], bb[kk]): the quotient digit from the two leading registers, corrected at most twice"""
    qhat = b.op("Idiv", b.add(b.mul(a[kk], n_base), a[kk - 1]), bb[kk - 1])
    qhat = b.tern(b.op("Gt", qhat, n_max), n_max, qhat)
    mult = _lsm(b, n_base, kk, qhat, bb, zero)
    g1 = _long_gt(b, kk + 1, mult, a, zero, one)
    mult2 = _long_sub(b, n_base, kk + 1, mult, list(bb) + [zero], zero, one)
    g2 = _long_gt(b, kk + 1, mult2, a, zero, one)
    return b.tern(b.op("Eq", g1, one), b.tern(b.op("Eq", g2, one), b.sub(qhat, two), b.sub(qhat, one)), qhat)


def build_rsa_long_div_class(n=121, k=17, muls=2, range_checks=True, seed="rsa"):
    """zk-email RSA / long_div-class graph: a chain of `muls` modular multiplications out = a * b mod p on k registers of n
    bits (RSA-2048: n = 121, k = 17), in the pattern of x^65537 (sixteen squarings, then a multiplication by x, repeated).
    Per multiplication, as fp.circom's FpMul computes its witness:
      * the 2k-register product: schoolbook columns (Mul, Add), then the carry chain col % 2^n, col \\ 2^n   (Mod, Idiv)
      * (q, r) = long_div(n, k, k, product, p): k + 1 quotient digits, each a short_div -- normalise (long_scalar_mult by
        scale = 2^n \\ (1 + p[k-1])), estimate from the two leading registers (Idiv of a 2n-bit by an n-bit value), multiply
        back (long_scalar_mult), compare (long_gt) and correct (long_sub, long_gt) -- then remainder -= digit * p << (n i)
        (long_scalar_mult, long_sub over all 2k registers)
      * q[i], r[i] as witness signals, each with its Num2Bits(n) range check's bit signals ((v >> j) & 1) when range_checks.
    Inputs: x[k], p[k] -- any field elements; they are masked to n bits (what the circuit's range checks would constrain),
    and p's leading register is brought into [2^(n-10), 2^(n-9)) like the 112-bit leading register of a 2048-bit modulus.
    Values of a consistent input (x, p as above) stay far below r; nothing in the graph can fail (Idiv / Mod by 0 give 0).
    Node count ~ muls * (1 330 (k + 1) k ... ) -- measured: n=121, k=17: ~23.9 k operations per multiplication + 8.2 k of range checks."""
    b = Builder()
    x_in = b.input("base", k)
    p_in = b.input("modulus", k)
    n_base, n_max, mask = b.const(1 << n), b.const((1 << n) - 1), b.const((1 << n) - 1)
    zero, one, two = b.const(0), b.const(1), b.const(2)
    x = [b.signal(b.op("Band", v, mask)) for v in x_in]
    p = [b.signal(b.op("Band", v, mask)) for v in p_in[:-1]]
    top_bits = max(n - 10, 1)
    p.append(b.signal(b.add(b.op("Band", p_in[-1], b.const((1 << top_bits) - 1)), b.const(1 << top_bits))))
    # short_div's loop invariants (once per graph behind value numbering): scale and the normalised divisor
    scale = b.op("Idiv", n_base, b.add(one, p[k - 1]))
    norm_b = _lsm(b, n_base, k, scale, p, zero)  # k + 1 registers

    def short_div(a):
        norm_a = _lsm(b, n_base, k + 1, scale, a, zero)  # k + 2 registers
        wide = _short_div_norm(b, n_base, n_max, k + 1, norm_a, norm_b, zero, one, two)
        narrow = _short_div_norm(b, n_base, n_max, k, norm_a, norm_b[:k], zero, one, two)
        return b.tern(b.op("Neq", norm_b[k], zero), wide, narrow)

    def long_div(a):  # a: 2k registers -> (k + 1 quotient digits, k remainder registers)
        m = k
        rem = list(a)
        quo = [None] * (m + 1)
        for i in range(m, -1, -1):
            dividend = ([rem[j + m] for j in range(k)] + [zero]) if i == m else [rem[j + i] for j in range(k + 1)]
            quo[i] = short_div(dividend)
            mult_shift = _lsm(b, n_base, k, quo[i], p, zero)
            subtrahend = [zero] * (m + k)
            for j in range(k + 1):
                if i + j < m + k:
                    subtrahend[i + j] = mult_shift[j]
            rem = _long_sub(b, n_base, m + k, rem, subtrahend, zero, one)
        return quo, rem[:k]

    def fp_mul(u, v):
        cols = [None] * (2 * k - 1)
        for i in range(k):
            for j in range(k):
                pr = b.mul(u[i], v[j])
                cols[i + j] = pr if cols[i + j] is None else b.add(cols[i + j], pr)
        carry, prod = zero, []
        for c in range(2 * k - 1):
            t = b.add(cols[c], carry)
            prod.append(b.op("Mod", t, n_base))
            carry = b.op("Idiv", t, n_base)
        prod.append(carry)
        quo, rem = long_div(prod)
        for v_ in quo[:k] + rem:          # q[i] <-- ..., r[i] <-- ... and their range checks
            b.signal(v_)
            if range_checks:
                for j in range(n):
                    b.signal(b.op("Band", b.op("Shr", v_, b.const(j)), one))
        return rem

    acc = x
    for s in range(muls):
        acc = fp_mul(acc, x if s % 17 == 16 else acc)
    return b


def build_bit_recurrence_variants(seed):
    """The one-bit recurrences of multi-register integers in the shapes the compiler's recognition by value has to cope with
    (rewrite.cc detect_bit_scans): register-wise subtractions whose borrow test is written with any of the four ordered
    comparisons, whose arms are associated differently (x - y - b, x - (y + b), (x - b) - y; 2^n first, last, or inside), with
    constant registers, with inner nodes that something else reads, with and without the last borrow; comparisons decided by the
    most significant differing register with either strict comparison outside, swapped operands, every pair of result bits and
    any boolean coming in -- on registers of any width (also beyond 128 bits and straight from the inputs: the kernels' general
    paths), in chains longer than a bundle.  Whatever the compiler makes of them, the witness must be exact."""
    rnd = random.Random(seed)
    b = Builder()
    n = rnd.choice([2, 7, 31, 32, 33, 55, 63, 64, 65, 100, 121, 126, 127, 128, 150, 200, 252])
    k = rnd.choice([1, 2, 3, 5, 8, 17, 18, 33, 40])
    base, zero, one = b.const(1 << n), b.const(0), b.const(1)
    mask = b.const((1 << n) - 1)
    raw = rnd.random() < 0.25   # registers straight from the inputs: any field element
    xs_in, ys_in = b.input("x", k), b.input("y", k)
    (f_in,) = b.input("f")
    reg = (lambda v: v) if raw else (lambda v: b.op("Band", v, mask))
    xs, ys = [reg(v) for v in xs_in], [reg(v) for v in ys_in]
    flag = b.op("Lt", f_in, b.const(R // 3))   # some boolean
    for _chain in range(rnd.randrange(1, 4)):
        kind = rnd.choice(["sub", "sub", "cmp", "cmp", "sub_then_cmp"])
        x = list(xs)
        y = [rnd.choice([ys[i], ys[i], zero, xs[i], b.const(rnd.getrandbits(min(n, 250)))]) for i in range(k)]
        if rnd.random() < 0.3:
            x = [rnd.choice([x[i], b.const(rnd.getrandbits(min(n, 250)))]) for i in range(k)]
        if kind in ("sub", "sub_then_cmp"):
            cond_style = rnd.choice(["geq", "leq", "lt", "gt"])
            diff, borrow = [], None
            for i in range(k):
                s = y[i] if borrow is None else (b.add(y[i], borrow) if rnd.random() < 0.5 else b.add(borrow, y[i]))
                then_style, else_style = rnd.randrange(3), rnd.randrange(3)
                bw = borrow if borrow is not None else zero
                if borrow is None and rnd.random() < 0.7:   # the first register as circom-bigint writes it: no borrow term at all
                    d_then = b.sub(x[i], y[i])
                    d_else = b.add(b.sub(x[i], y[i]), base) if rnd.random() < 0.5 else b.sub(b.add(base, x[i]), y[i])
                else:
                    d_then = [lambda: b.sub(b.sub(x[i], y[i]), bw), lambda: b.sub(x[i], b.add(y[i], bw)), lambda: b.sub(b.sub(x[i], bw), y[i])][then_style]()
                    d_else = [lambda: b.sub(b.sub(b.add(base, x[i]), y[i]), bw), lambda: b.add(b.sub(b.sub(x[i], y[i]), bw), base),
                              lambda: b.sub(b.add(x[i], base), b.add(y[i], bw))][else_style]()
                if rnd.random() < 0.08:
                    b.signal(d_else)          # an arm that is a witness element stays
                if cond_style == "geq":
                    c = b.op("Geq", x[i], s)
                elif cond_style == "leq":
                    c = b.op("Leq", s, x[i])
                elif cond_style == "lt":
                    c = b.op("Lt", x[i], s)
                else:
                    c = b.op("Gt", s, x[i])
                no_borrow_when_true = cond_style in ("geq", "leq")
                diff.append(b.signal(b.tern(c, d_then, d_else) if no_borrow_when_true else b.tern(c, d_else, d_then)))
                if i + 1 < k or rnd.random() < 0.5:
                    borrow = b.tern(c, zero, one) if no_borrow_when_true else b.tern(c, one, zero)
                    if rnd.random() < 0.1:
                        b.signal(borrow)
                if rnd.random() < 0.05:
                    b.signal(c)
            x = diff
        if kind in ("cmp", "sub_then_cmp"):
            kg, kl = rnd.randrange(2), rnd.randrange(2)
            res = rnd.choice([zero, zero, one, flag])
            gt_outside = rnd.random() < 0.5
            for i in range(k):
                gt = b.op("Gt", x[i], y[i]) if rnd.random() < 0.7 else b.op("Lt", y[i], x[i])
                lt = b.op("Lt", x[i], y[i]) if rnd.random() < 0.7 else b.op("Gt", y[i], x[i])
                c_kg, c_kl = (one if kg else zero), (one if kl else zero)
                res = b.tern(gt, c_kg, b.tern(lt, c_kl, res)) if gt_outside else b.tern(lt, c_kl, b.tern(gt, c_kg, res))
                if rnd.random() < 0.1:
                    b.signal(res)
            b.signal(b.tern(b.op("Eq", res, one), xs[0], ys[0]))
            b.signal(res)
    # integer divisions by registers (the quotient-digit estimates of multi-register long division: a product of two registers, or the
    # field element an input is, by a register -- divisors of one, two and more 64-bit words, zero among them)
    for i in range(min(k, rnd.randrange(0, 5))):
        num = rnd.choice([b.mul(xs[i], ys[i]), b.add(b.mul(xs[i], base), ys[i]), xs_in[i], xs[i]])
        den = rnd.choice([ys[i], ys[i], xs[(i + 1) % k], b.op("Band", ys_in[i], b.const((1 << rnd.choice([64, 65, 100, 121, 128])) - 1)), zero])
        b.signal(b.op(rnd.choice(["Idiv", "Mod"]), num, den))
    # selections on an ordered comparison that nothing else reads (clamps, minima, maxima: one selection bundle with the comparison inside),
    # and some whose comparison is read again or is a witness element (left alone)
    for i in range(min(k, rnd.randrange(1, 6))):
        cmp_op = rnd.choice(["Lt", "Gt", "Leq", "Geq"])
        lhs, rhs = rnd.choice([xs[i], ys[i]]), rnd.choice([ys[i], mask, b.const(rnd.getrandbits(min(n, 250))), xs[(i + 1) % k]])
        c = b.op(cmp_op, lhs, rhs)
        arms = [rnd.choice([lhs, rhs, mask, zero, one, xs[0]]) for _ in range(2)]
        sel = b.signal(b.tern(c, arms[0], arms[1]))
        if rnd.random() < 0.2:
            b.signal(c)
        if rnd.random() < 0.3:
            b.signal(b.add(sel, c) if rnd.random() < 0.5 else b.op("Band", sel, mask))
    return b
