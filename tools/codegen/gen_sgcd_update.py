#!/usr/bin/env python3
"""Generates circom-witnesscalc_amd/csrc/sgcd_update_gfx950.inc: the matrix update of one safegcd batch (fr_gfx950.hpp,
sgcd_update_all: f, g <- M (f, g) / 2^30 and d, e <- M (d, e) / 2^30 mod r on signed 30-bit limbs) as ONE asm block.

Why a block: written as one inline-asm statement per v_mad_i64_i32 the compiler puts a wait state behind every one of
them (it cannot see into the statement): 50-75 of the ~230 issue slots of an update were s_nop, and the divider wave of
the interpreter is bound by exactly this instruction stream (the Div nodes of reference src/graph.rs:109 are served by
a lone wavefront, 18 batches per inversion).  Inside one block the four accumulator chains (d, e, f, g) are interleaved
limb by limb, so that no multiply-add waits for the one in front of it.

The instruction list is executed on Python integers against the big-integer definition of the update before it is
written (run with --check for more rounds).
"""
import os
import random
import sys

P30 = [0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064]
P_INT = sum(v << (30 * i) for i, v in enumerate(P30))
PINV30 = 0x10000001  # r^-1 mod 2^30
M30 = 0x3fffffff
M32 = 0xffffffff


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


class Emitter:
    """instructions as (text, function on the register file); registers are names -> u32"""

    def __init__(self):
        self.ins = []

    def mad_i64(self, pair, a, b, addend):
        lo, hi = pair

        def fn(R):
            ad = 0 if addend is None else s64(R[addend[0]] | (R[addend[1]] << 32))
            x = s32(R[a]) * s32(R[b]) + ad
            R[lo], R[hi] = x & M32, (x >> 32) & M32
        add_txt = "0" if addend is None else vp(addend)
        self.ins.append(("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (vp(pair), a, b, add_txt), fn))

    def mad_u64(self, pair, a, b, addend):
        lo, hi = pair

        def fn(R):
            x = R[a] * R[b] + (R[addend[0]] | (R[addend[1]] << 32))
            R[lo], R[hi] = x & M32, (x >> 32) & M32
        self.ins.append(("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vp(pair), a, b, vp(addend)), fn))

    # one v_ashrrev_i64 per shift; --shift32: funnel shift + 32-bit shift instead (A/B on MI355X, four active lanes:
    # 52.2 k against 53.4 k cycles per inversion, profiles/r03_inv_bench.txt)
    shift64 = True

    def sar30(self, pair):
        lo, hi = pair
        if self.shift64:
            def f0(R):
                x = s64(R[lo] | (R[hi] << 32)) >> 30
                R[lo], R[hi] = x & M32, (x >> 32) & M32
            self.ins.append(("v_ashrrev_i64 %s, 30, %s" % (vp(pair), vp(pair)), f0))
            return

        def f1(R):
            R[lo] = ((R[lo] | (R[hi] << 32)) >> 30) & M32
        self.ins.append(("v_alignbit_b32 %s, %s, %s, 30" % (lo, hi, lo), f1))

        def f2(R):
            R[hi] = (s32(R[hi]) >> 30) & M32
        self.ins.append(("v_ashrrev_i32_e32 %s, 30, %s" % (hi, hi), f2))

    def op2(self, name, d, a, b, py):
        def fn(R):
            R[d] = py(R[a] if isinstance(a, str) else a, R[b]) & M32
        at = a if isinstance(a, str) else "0x%x" % a
        self.ins.append(("%s %s, %s, %s" % (name, d, at, b), fn))


def vp(p):
    lo, hi = p
    assert lo[0] == "v" and int(hi[1:]) == int(lo[1:]) + 1 and int(lo[1:]) % 2 == 0, p
    return "v[%s:%s]" % (lo[1:], hi[1:])


def make(vbase=184):
    """d, e, f, g: %[d0]..%[g8] (in/out); matrix %[mu] %[mv] %[mq] %[mr]; modulus limbs %[p0]..%[p8] and %[pinv] in SGPRs.
    Temporaries: physical VGPRs from vbase (clobbered)."""
    E = Emitter()
    nv = [vbase]

    def pair():
        nv[0] += nv[0] & 1
        r = ("v%d" % nv[0], "v%d" % (nv[0] + 1))
        nv[0] += 2
        return r
    cd, ce, cf, cg = pair(), pair(), pair(), pair()
    md, me, tq = pair(), pair(), pair()   # (md / me sit in the low register of an aligned pair: addend of a 64-bit multiply-add)
    sd, se = tq
    d = ["%%[d%d]" % i for i in range(9)]
    e = ["%%[e%d]" % i for i in range(9)]
    f = ["%%[f%d]" % i for i in range(9)]
    g = ["%%[g%d]" % i for i in range(9)]
    p = ["%%[p%d]" % i for i in range(9)]
    mu, mv, mq, mr = "%[mu]", "%[mv]", "%[mq]", "%[mr]"
    # md = (u & sign(d)) + (v & sign(e)), me likewise with q, r
    E.op2("v_ashrrev_i32_e32", sd, 31, d[8], lambda a, b: s32(b) >> 31)
    E.op2("v_ashrrev_i32_e32", se, 31, e[8], lambda a, b: s32(b) >> 31)
    E.mad_i64(cd, mu, d[0], None)
    E.mad_i64(ce, mq, d[0], None)
    E.mad_i64(cf, mu, f[0], None)
    E.mad_i64(cg, mq, f[0], None)
    E.op2("v_and_b32_e32", md[0], mu, sd, lambda a, b: a & b)
    E.op2("v_and_b32_e32", md[1], mv, se, lambda a, b: a & b)
    E.op2("v_and_b32_e32", me[0], mq, sd, lambda a, b: a & b)
    E.op2("v_and_b32_e32", me[1], mr, se, lambda a, b: a & b)
    E.mad_i64(cd, mv, e[0], cd)
    E.mad_i64(ce, mr, e[0], ce)
    E.mad_i64(cf, mv, g[0], cf)
    E.mad_i64(cg, mr, g[0], cg)
    E.op2("v_add_u32_e32", md[0], md[0], md[1], lambda a, b: a + b)
    E.op2("v_add_u32_e32", me[0], me[0], me[1], lambda a, b: a + b)
    # md -= (pinv * low(cd) + md) mod 2^30: the multiple of r that clears the low 30 bits of cd
    E.mad_u64(tq, p_inv(), cd[0], md)
    E.sar30(cf)   # (the low 30 bits of cf, cg are zero by construction)
    E.op2("v_and_b32_e32", tq[0], M30, tq[0], lambda a, b: a & b)
    E.op2("v_sub_u32_e32", md[0], md[0], tq[0], lambda a, b: a - b)
    E.mad_u64(tq, p_inv(), ce[0], me)
    E.sar30(cg)
    E.op2("v_and_b32_e32", tq[0], M30, tq[0], lambda a, b: a & b)
    E.op2("v_sub_u32_e32", me[0], me[0], tq[0], lambda a, b: a - b)
    E.mad_i64(cd, p[0], md[0], cd)
    E.mad_i64(ce, p[0], me[0], ce)
    E.sar30(cd)
    E.sar30(ce)
    for i in range(1, 9):
        E.mad_i64(cd, mu, d[i], cd)
        E.mad_i64(ce, mq, d[i], ce)
        E.mad_i64(cf, mu, f[i], cf)
        E.mad_i64(cg, mq, f[i], cg)
        E.mad_i64(cd, mv, e[i], cd)
        E.mad_i64(ce, mr, e[i], ce)
        E.mad_i64(cf, mv, g[i], cf)
        E.mad_i64(cg, mr, g[i], cg)
        E.mad_i64(cd, p[i], md[0], cd)
        E.mad_i64(ce, p[i], me[0], ce)
        E.op2("v_and_b32_e32", f[i - 1], M30, cf[0], lambda a, b: a & b)
        E.op2("v_and_b32_e32", g[i - 1], M30, cg[0], lambda a, b: a & b)
        E.sar30(cf)
        E.sar30(cg)
        E.op2("v_and_b32_e32", d[i - 1], M30, cd[0], lambda a, b: a & b)
        E.op2("v_and_b32_e32", e[i - 1], M30, ce[0], lambda a, b: a & b)
        E.sar30(cd)
        E.sar30(ce)
    for dst, src in ((d[8], cd), (e[8], ce), (f[8], cf), (g[8], cg)):
        E.op2("v_mov_b32_e32", dst, 0, src[0], lambda a, b: b)
    return E, nv[0]


def p_inv():
    return "%[pinv]"


def fix_mov(text):
    # v_mov has one source: the emitter's two-operand helper carries a dummy first source
    if text.startswith("v_mov_b32_e32"):
        d, _, b = [x.strip() for x in text[len("v_mov_b32_e32"):].split(",")]
        return "v_mov_b32_e32 %s, %s" % (d, b)
    return text


def limbs30(x):
    """signed value -> 9 limbs, the low eight in [0, 2^30), the top one signed"""
    out = []
    for _ in range(8):
        out.append(x & M30)
        x >>= 30
    out.append(x & M32)
    return out


def value30(l):
    return sum(v << (30 * i) for i, v in enumerate(l[:8])) + (s32(l[8]) << 240)


def divsteps30(eta, f0, g0):
    """the variable-time batch of fr_gfx950.hpp (sgcd_divsteps_30_var) on Python integers -> (eta, u, v, q, r)"""
    u, v, q, r = 1, 0, 0, 1
    f, g, i = f0 & M32, g0 & M32, 30
    while True:
        zeros = 0
        while zeros < i and not (g >> zeros) & 1:
            zeros += 1
        g >>= zeros
        u <<= zeros
        v <<= zeros
        eta -= zeros
        i -= zeros
        if i == 0:
            break
        if eta < 0:
            eta, f, g, u, q, v, r = -eta, g, (-f) & M32, q, -u, r, -v
        limit = min(eta + 1, i, 6)
        w = (g * f * (f * f - 2)) & ((1 << limit) - 1)
        g = (g + f * w) & M32
        q += u * w
        r += v * w
    return eta, u, v, q, r


def check(rounds, seed=3):
    E, _ = make()
    rnd = random.Random(seed)
    for it in range(rounds):
        # a state of the inversion: f odd, |f|, |g| < 2^256, d, e in (-2r, r); the matrix of the next 30 divsteps
        bits = rnd.choice([256, 256, 200, 90, 31, 8])
        fv = rnd.randrange(-(1 << bits) + 1, 1 << bits) | 1
        gv = rnd.randrange(-(1 << bits) + 1, 1 << bits)
        if it % 7 == 0:
            gv <<= rnd.randrange(0, 40)
            gv = max(-(1 << 256) + 1, min((1 << 256) - 1, gv))
        if it % 11 == 0:
            fv, gv = P_INT, rnd.randrange(P_INT)
        eta = rnd.randrange(-20, 21)
        _, u, v, q, r = divsteps30(eta, fv & M30, gv & M30)
        dv = rnd.randrange(-2 * P_INT + 1, P_INT)
        ev = rnd.randrange(-2 * P_INT + 1, P_INT)
        if it % 9 == 0:
            dv, ev = rnd.choice([(0, 1), (-2 * P_INT + 1, P_INT - 1), (P_INT - 1, -2 * P_INT + 1), (0, 0)])
        R = {}
        for name, val in (("d", dv), ("e", ev), ("f", fv), ("g", gv)):
            for i, l in enumerate(limbs30(val)):
                R["%%[%s%d]" % (name, i)] = l
        for i in range(9):
            R["%%[p%d]" % i] = P30[i]
        R["%[pinv]"] = PINV30
        R["%[mu]"], R["%[mv]"], R["%[mq]"], R["%[mr]"] = u & M32, v & M32, q & M32, r & M32
        for _, fn in E.ins:
            fn(R)
        got = {n: value30([R["%%[%s%d]" % (n, i)] for i in range(9)]) for n in "defg"}
        assert (u * fv + v * gv) % (1 << 30) == 0 and (q * fv + r * gv) % (1 << 30) == 0
        assert got["f"] == (u * fv + v * gv) >> 30 and got["g"] == (q * fv + r * gv) >> 30, (it, "fg")
        for n in "defg":
            assert all(0 <= R["%%[%s%d]" % (n, i)] <= M30 for i in range(8)), (it, n, "limb range")
        # d' = (u d + v e + md r) / 2^30 with the multiple of r that makes the division exact: same residue class, range (-2r, r)
        for n, a, b in (("d", u, v), ("e", q, r)):
            want_mod = (a * dv + b * ev) * pow(1 << 30, -1, P_INT) % P_INT
            assert got[n] % P_INT == want_mod, (it, n, "residue")
            assert -2 * P_INT < got[n] < P_INT, (it, n, "range", got[n])
    return len(E.ins)


def emit(path):
    E, vend = make()
    lines = [fix_mov(t) for t, _ in E.ins]
    names = ["%s%d" % (n, i) for n in "defg" for i in range(9)]
    outs = ", ".join('[%s] "+v"(%s.v[%s])' % (nm, nm[0], nm[1]) for nm in names)
    ins = ", ".join(['[mu] "v"(t.u)', '[mv] "v"(t.v)', '[mq] "v"(t.q)', '[mr] "v"(t.r)'] + ['[p%d] "s"(p30_%d)' % (i, i) for i in range(9)] + ['[pinv] "s"(pinv30)'])
    clob = ", ".join('"v%d"' % r for r in range(184, vend)) + ', "vcc"'
    text = ["// GENERATED by tools/codegen/gen_sgcd_update.py -- do not edit.  Device-only body of sgcd_update_all(d, e, f, g, t): one asm",
            "// block, %d issue slots (90 v_mad_i64_i32, no compiler-inserted wait states: the four accumulator chains are interleaved); clobbers v184-v%d, vcc." % (len(lines), vend - 1),
            "{"]
    for i in range(9):
        text.append("    const int p30_%d = 0x%x;" % (i, P30[i]))
    text.append("    const unsigned int pinv30 = 0x%xu;" % PINV30)
    text.append('    asm volatile("' + "\\n\\t".join(lines) + '"')
    text.append("                 : " + outs)
    text.append("                 : " + ins)
    text.append("                 : " + clob + ");")
    text.append("}")
    open(path, "w").write("\n".join(text) + "\n")
    return len(lines)


if __name__ == "__main__":
    Emitter.shift64 = "--shift32" not in sys.argv
    n = check(2000 if "--check" in sys.argv else 300)
    print("sgcd update: emulation ok, %d issue slots" % n)
    if "--check" not in sys.argv:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.normpath(os.path.join(here, "..", "..", "circom-witnesscalc_amd", "csrc", "sgcd_update_gfx950.inc"))
        if "--out" in sys.argv:
            path = sys.argv[sys.argv.index("--out") + 1]
        emit(path)
        print("wrote", path)
