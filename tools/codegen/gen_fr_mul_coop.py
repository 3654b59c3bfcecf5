#!/usr/bin/env python3
"""Generates the lane-cooperative Montgomery multipliers of the interpreter's narrow multiplication bundles
(circom-witnesscalc_amd/csrc/fr_mul_coop{2,4,8}_gfx950.inc) and checks them on a lane-accurate emulator.

Why: a lone wavefront issues one instruction per ~5 cycles whatever its lanes do, and the headline configuration is
bound by the chain of dependent multiplications of the graph (reference Operation::Mul, src/graph.rs:105): the only
way to shorten a dependent multiplication is fewer instructions per product.  A bundle with few nodes leaves most of
the 64 lanes idle, so K = 2, 4 or 8 adjacent lanes share ONE product: lane k of a group holds L = 8/K limbs of b and
of the modulus, every lane holds all of a, and the 8 rounds of a word-serial Montgomery multiplication
(t += a_i * b; m = t_0 * (-r^-1); t += m * r; t >>= 32) run on L + 1 column accumulators per lane instead of 8:

    round i, lane k:   C_j += a_i * b_(kL+j)                 j < L      (v_mad_u64_u32 + carry into a third word)
                       m = C_0.lo * (-r^-1 mod 2^32) of lane 0, broadcast to the group   (DPP)
                       C_j += m * r_(kL+j)
                       shift: C_0's upper words fold into C_1, the low word of the NEXT lane's C_0 arrives as the new
                       top column (DPP; lane 0's is zero by construction, so rotating instead of shifting is exact)

After 8 rounds the lanes hold an unnormalised result below 2r: words + a small overflow into the next lane.  Carries
between lanes are resolved with one 64-bit scalar addition for the whole wave (generate / propagate masks from the
carry-out and an all-ones compare: carries = ((G | P) + G) ^ P), then r is subtracted the same way and the top lane's
final borrow selects per group.  Issue slots by count: K=8 ~105, K=4 ~150, K=2 ~235, against 322 for one lane.

Hazards (gfx940 family): a VGPR written by a VALU instruction may be read by a DPP instruction two wait states later
at the earliest; an SGPR (carry mask) written by a VALU instruction may be read by a VALU instruction two wait states
later at the earliest.  The emitter tracks the last writer of every register and pads with s_nop.

The emulator below executes the very instruction list that is printed, on 64 lanes of random operands, against
Python big integers (run: python3 tools/codegen/gen_fr_mul_coop.py --check).
"""
import os
import random
import sys

P_LIMBS = [0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72]
P_INT = sum(v << (32 * i) for i, v in enumerate(P_LIMBS))
INV32 = 0xefffffff  # -r^-1 mod 2^32
M32 = 0xffffffff
M64 = (1 << 64) - 1
WAVE = 64


# ---- a tiny virtual ISA: every instruction knows how to print itself and how to execute on the lane state ----------
class St:
    """Emulator state: v[name] = list of 64 u32, s[name] = 64-bit lane mask / scalar."""

    def __init__(self):
        self.v = {}
        self.s = {}

    def rv(self, name, lane):
        if isinstance(name, int):
            return name & M32
        if name in self.s:  # scalar operand broadcast to the lanes
            return self.s[name] & M32
        return self.v[name][lane]


def dpp_src_lane(ctrl, lane):
    """source lane of a DPP control, or None when the lane has no source (bound_ctrl:0 then reads zero)"""
    kind = ctrl[0]
    row, pos = lane // 16, lane % 16
    if kind == "quad_perm":
        return (lane & ~3) + ctrl[1][lane & 3]
    if kind == "row_shl":  # lane i reads lane i + n of its row
        return row * 16 + pos + ctrl[1] if pos + ctrl[1] < 16 else None
    if kind == "row_shr":
        return row * 16 + pos - ctrl[1] if pos - ctrl[1] >= 0 else None
    if kind == "row_newbcast":
        return row * 16 + ctrl[1]
    raise ValueError(kind)


def dpp_text(ctrl):
    if ctrl[0] == "quad_perm":
        return "quad_perm:[%d,%d,%d,%d]" % tuple(ctrl[1])
    return "%s:%d" % (ctrl[0], ctrl[1])


class I:
    """one instruction: text template + register reads/writes (for the hazard padding) + emulation"""

    def __init__(self, text, fn, vw=(), vr=(), sw=(), sr=(), dpp_reads=(), valu=True):
        self.text, self.fn = text, fn
        self.vw, self.vr, self.sw, self.sr, self.dpp_reads, self.valu = tuple(vw), tuple(vr), tuple(sw), tuple(sr), tuple(dpp_reads), valu


class Prog:
    def __init__(self):
        self.ins = []

    # -- VALU ---------------------------------------------------------------------------------------------------------
    def mad(self, pair, cout, a, b, addend):
        """pair(lo, hi) = a * b + addend (a pair, or 0); carry-out mask into cout"""
        lo, hi = pair
        add_txt = "0" if addend == 0 else "%s" % vpair(addend)

        def fn(st):
            m = 0
            nl, nh = [0] * WAVE, [0] * WAVE
            for l in range(WAVE):
                ad = 0 if addend == 0 else st.rv(addend[0], l) | (st.rv(addend[1], l) << 32)
                x = st.rv(a, l) * st.rv(b, l) + ad
                if x >> 64:
                    m |= 1 << l
                nl[l], nh[l] = x & M32, (x >> 32) & M32
            st.v[lo], st.v[hi] = nl, nh
            st.s[cout] = m
        rd = [a, b] + ([] if addend == 0 else list(addend))
        self.ins.append(I("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vpair(pair), sreg(cout), op(a), op(b), add_txt), fn,
                          vw=pair, vr=[r for r in rd if isv(r)], sw=[cout], sr=[r for r in rd if iss(r)]))

    def mul_lo(self, d, a, b):
        def fn(st):
            st.v[d] = [(st.rv(a, l) * st.rv(b, l)) & M32 for l in range(WAVE)]
        self.ins.append(I("v_mul_lo_u32 %s, %s, %s" % (op(d), op(a), op(b)), fn, vw=[d], vr=[r for r in (a, b) if isv(r)],
                          sr=[r for r in (a, b) if iss(r)]))

    def addc(self, d, cout, a, b, cin, sub=False):
        """d = a + b + cin (or a - b - cin), carry / borrow out into cout; cin None = plain add_co"""
        name = ("v_sub" if sub else "v_add") + ("_co_u32_e64" if cin is None else "")
        if cin is not None:
            name = "v_subb_co_u32_e64" if sub else "v_addc_co_u32_e64"

        def fn(st):
            m = 0
            out = [0] * WAVE
            for l in range(WAVE):
                ci = 0 if cin is None else (st.s[cin] >> l) & 1
                x = st.rv(a, l) - st.rv(b, l) - ci if sub else st.rv(a, l) + st.rv(b, l) + ci
                if x < 0 or x >> 32:
                    m |= 1 << l
                out[l] = x & M32
            st.v[d] = out
            st.s[cout] = m
        txt = "%s %s, %s, %s, %s" % (name, op(d), sreg(cout), op(a), op(b)) + ("" if cin is None else ", %s" % sreg(cin))
        self.ins.append(I(txt, fn, vw=[d], vr=[r for r in (a, b) if isv(r)], sw=[cout], sr=[] if cin is None else [cin]))

    def add_dpp(self, d, a_dpp, b, ctrl, bank_mask=0xf):
        """d = dpp(a_dpp) + b, carry-out into vcc (VOP2 encoding, the DPP control applies to the first source)"""
        def fn(st):
            m = 0
            out = list(st.v.get(d, [0] * WAVE))
            old = st.v[a_dpp]
            for l in range(WAVE):
                if not (bank_mask >> ((l % 16) // 4)) & 1:
                    continue
                sl = dpp_src_lane(ctrl, l)
                x = (old[sl] if sl is not None else 0) + st.rv(b, l)
                if x >> 32:
                    m |= 1 << l
                out[l] = x & M32
            st.v[d] = out
            st.s["vcc"] = m
        self.ins.append(I("v_add_co_u32_dpp %s, vcc, %s, %s %s row_mask:0xf bank_mask:0x%x bound_ctrl:0" % (op(d), op(a_dpp), op(b), dpp_text(ctrl), bank_mask),
                          fn, vw=[d], vr=[b], sw=["vcc"], dpp_reads=[a_dpp]))

    def mov_dpp(self, d, s, ctrl, bank_mask=0xf):
        def fn(st):
            out = list(st.v.get(d, [0] * WAVE))
            old = st.v[s]
            for l in range(WAVE):
                if not (bank_mask >> ((l % 16) // 4)) & 1:
                    continue
                sl = dpp_src_lane(ctrl, l)
                out[l] = old[sl] if sl is not None else 0
            st.v[d] = out
        self.ins.append(I("v_mov_b32_dpp %s, %s %s row_mask:0xf bank_mask:0x%x bound_ctrl:0" % (op(d), op(s), dpp_text(ctrl), bank_mask),
                          fn, vw=[d], vr=[d] if bank_mask != 0xf else [], dpp_reads=[s]))

    def mov(self, d, s):
        def fn(st):
            st.v[d] = [st.rv(s, l) for l in range(WAVE)]
        self.ins.append(I("v_mov_b32_e32 %s, %s" % (op(d), op(s)), fn, vw=[d], vr=[s] if isv(s) else [], sr=[s] if iss(s) else []))

    def and_(self, d, a, b, opn="and"):
        def fn(st):
            f = {"and": lambda x, y: x & y, "or": lambda x, y: x | y}[opn]
            st.v[d] = [f(st.rv(a, l), st.rv(b, l)) for l in range(WAVE)]
        self.ins.append(I("v_%s_b32_e32 %s, %s, %s" % (opn, op(d), op(a), op(b)), fn, vw=[d], vr=[r for r in (a, b) if isv(r)]))

    def xor_(self, d, a, b):
        def fn(st):
            st.v[d] = [st.rv(a, l) ^ st.rv(b, l) for l in range(WAVE)]
        self.ins.append(I("v_xor_b32_e32 %s, %s, %s" % (op(d), op(a), op(b)), fn, vw=[d], vr=[r for r in (a, b) if isv(r)]))

    def cmp_eq(self, cout, a, b):
        def fn(st):
            st.s[cout] = sum(1 << l for l in range(WAVE) if st.rv(a, l) == st.rv(b, l))
        self.ins.append(I("v_cmp_eq_u32_e64 %s, %s, %s" % (sreg(cout), op(a), op(b)), fn, vr=[r for r in (a, b) if isv(r)], sw=[cout]))

    def cndmask(self, d, a, b, mask):
        """d = mask ? b : a"""
        def fn(st):
            st.v[d] = [st.rv(b, l) if (st.s[mask] >> l) & 1 else st.rv(a, l) for l in range(WAVE)]
        self.ins.append(I("v_cndmask_b32_e64 %s, %s, %s, %s" % (op(d), op(a), op(b), sreg(mask)), fn, vw=[d],
                          vr=[r for r in (a, b) if isv(r)], sr=[mask]))

    # -- SALU (64-bit masks) ---------------------------------------------------------------------------------------------
    def s_op(self, name, d, a, b):
        f = {"or": lambda x, y: x | y, "and": lambda x, y: x & y, "xor": lambda x, y: x ^ y, "andn2": lambda x, y: x & ~y & M64}[name]

        def fn(st):
            st.s[d] = f(st.s[a], st.s[b]) & M64
        self.ins.append(I("s_%s_b64 %s, %s, %s" % (name, sreg(d), sreg(a), sreg(b)), fn, sw=[d], sr=[a, b], valu=False))

    def s_add64(self, d, a, b):
        """d = a + b on 64-bit masks: two 32-bit scalar adds (d, a, b are fixed SGPR pairs whose halves can be named)"""
        def fn(st):
            st.s[d] = (st.s[a] + st.s[b]) & M64
        self.ins.append(I("s_add_u32 %s, %s, %s" % (shalf(d, 0), shalf(a, 0), shalf(b, 0)), lambda st: None, sw=[d], sr=[a, b], valu=False))
        self.ins.append(I("s_addc_u32 %s, %s, %s" % (shalf(d, 1), shalf(a, 1), shalf(b, 1)), fn, sw=[d], sr=[a, b], valu=False))

    def s_expand(self, d, a, K):
        """d = every group's bit (K-1) of a, spread over the K lanes of the group: ((a >> (K-1)) & 0x..0101) * (2^K - 1)"""
        one = sum(1 << i for i in range(0, 32, K))

        def fn(st):
            x = (st.s[a] >> (K - 1)) & sum(1 << i for i in range(0, 64, K))
            st.s[d] = (x * ((1 << K) - 1)) & M64
        self.ins.append(I("s_lshr_b64 %s, %s, %d" % (sreg(d), sreg(a), K - 1), lambda st: None, sw=[d], sr=[a], valu=False))
        self.ins.append(I("s_and_b32 %s, %s, 0x%x" % (shalf(d, 0), shalf(d, 0), one), lambda st: None, sw=[d], sr=[d], valu=False))
        self.ins.append(I("s_and_b32 %s, %s, 0x%x" % (shalf(d, 1), shalf(d, 1), one), lambda st: None, sw=[d], sr=[d], valu=False))
        self.ins.append(I("s_mul_i32 %s, %s, 0x%x" % (shalf(d, 0), shalf(d, 0), (1 << K) - 1), lambda st: None, sw=[d], sr=[d], valu=False))
        self.ins.append(I("s_mul_i32 %s, %s, 0x%x" % (shalf(d, 1), shalf(d, 1), (1 << K) - 1), fn, sw=[d], sr=[d], valu=False))

    # -- scheduling + hazard padding ----------------------------------------------------------------------------------------
    def schedule(self):
        """List scheduling of the instruction DAG (register RAW / WAR / WAW dependencies, SCC included) under the
        gfx940-family wait-state rules: VALU write of a VGPR -> DPP read: 2 wait states; VALU write of an SGPR -> VALU
        read: 2 wait states (an instruction in between is one wait state).  Longest-path-first; a slot with nothing
        ready becomes s_nop.  Returns [(slot, instruction)]; replaces self.ins by the scheduled order."""
        ins = self.ins
        n = len(ins)
        preds = [[] for _ in range(n)]
        succs = [[] for _ in range(n)]
        last_w, readers = {}, {}

        def regs_r(i):
            I_ = ins[i]
            r = set(I_.vr) | set(I_.dpp_reads) | set(I_.sr)
            if I_.text.startswith("s_addc_u32"):
                r.add("scc")
            return r

        def regs_w(i):
            I_ = ins[i]
            w = set(I_.vw) | set(I_.sw)
            if not I_.valu:
                w.add("scc")
            return w
        for i in range(n):
            I_ = ins[i]
            for r in regs_r(i):
                if r in last_w:
                    j = last_w[r]
                    d = 1
                    if ins[j].valu and r in I_.dpp_reads:
                        d = 3
                    if ins[j].valu and I_.valu and r in I_.sr and r in ins[j].sw:
                        d = 3
                    preds[i].append((j, d))
                readers.setdefault(r, []).append(i)
            for w in regs_w(i):
                if w in last_w:
                    preds[i].append((last_w[w], 1))
                for j in readers.get(w, []):
                    if j != i:
                        preds[i].append((j, 1))
                readers[w] = []
                last_w[w] = i
        for i in range(n):
            for j, d in preds[i]:
                succs[j].append((i, d))
        prio = [0] * n
        for i in range(n - 1, -1, -1):
            prio[i] = 1 + max([prio[k] + d - 1 for k, d in succs[i]] or [0])
        slot_of = {}
        done = [False] * n
        order = []
        t = 0
        remaining = n
        while remaining:
            best = -1
            for i in range(n):
                if done[i]:
                    continue
                ok = True
                for j, d in preds[i]:
                    if not done[j] or slot_of[j] + d > t:
                        ok = False
                        break
                if ok and (best < 0 or prio[i] > prio[best]):
                    best = i
            if best >= 0:
                done[best] = True
                slot_of[best] = t
                order.append((t, ins[best]))
                remaining -= 1
            t += 1
        self.ins = [I_ for _, I_ in order]
        return order

    def lines(self):
        order = self.schedule()
        out = []
        n_nop = 0
        t = 0
        for slot, I_ in order:
            gap = slot - t
            while gap > 0:
                g = min(gap, 8)
                out.append("s_nop %d" % (g - 1))
                n_nop += g
                gap -= g
            out.append(I_.text)
            t = slot + 1
        return out, n_nop

    def run(self, st):
        for ins in self.ins:
            ins.fn(st)


# register naming: VGPRs "vNNN" are physical (clobbered), "%[name]" are asm operands; SGPR pairs likewise
def isv(r):
    return isinstance(r, str) and (r.startswith("v") or r.startswith("%[v") or r.startswith("%[a") or r.startswith("%[b") or r.startswith("%[n") or r.startswith("%[r"))


def iss(r):
    return isinstance(r, str) and not isv(r)


def op(r):
    return "0x%x" % r if isinstance(r, int) and r > 64 else str(r)


def vpair(p):
    lo, hi = p
    assert lo[0] == "v" and hi[0] == "v" and int(hi[1:]) == int(lo[1:]) + 1, p
    return "v[%s:%s]" % (lo[1:], hi[1:])


def sreg(s):
    if s == "vcc":
        return "vcc"
    if s.startswith("s") and s[1:].isdigit():  # fixed pair named by its first register
        return "s[%d:%d]" % (int(s[1:]), int(s[1:]) + 1)
    return s


def shalf(s, h):
    assert s.startswith("s") and s[1:].isdigit(), s
    return "s%d" % (int(s[1:]) + h)


# ---- the multiplier ---------------------------------------------------------------------------------------------------
def make(K, vbase=168, sbase=88, riders=False, lin_only=False):
    """Instruction list for lane groups of K.  Inputs: %[a0..a7] (all of a, every lane), %[b0..b(L-1)] / %[n0..n(L-1)]
    (this lane's limbs of b / of the modulus), %[inv] (scalar -r^-1), %[top] (mask of every group's top lane).
    Outputs: %[r0..r(L-1)].  Temporaries are physical VGPRs from vbase and SGPR pairs from sbase (clobbered).
    riders: groups whose %[sub] is 0 / 1 compute (a + b) / (a - b) mod r instead (linear nodes riding in a multiplication
    bundle, graph.rs:110-111): %[aq0..] = this lane's limbs of a, %[lane0] = mask of every group's lowest lane.  Their
    limb sums (a + b, or a + r + ~b + 1 = a - b + r + 2^256, both below 2r once the 2^256 is dropped at the top lane)
    replace the multiplier's words in front of the normalisation both share."""
    L = 8 // K
    p = Prog()
    nv = [vbase]

    def newpair():
        nv[0] += nv[0] & 1   # VGPR tuples are 64-bit aligned on this family
        r = ("v%d" % nv[0], "v%d" % (nv[0] + 1))
        nv[0] += 2
        return r

    def newv():
        r = "v%d" % nv[0]
        nv[0] += 1
        return r
    a = ["%%[a%d]" % i for i in range(8)]
    b = ["%%[b%d]" % j for j in range(L)]
    n = ["%%[n%d]" % j for j in range(L)]
    rr = ["%%[r%d]" % j for j in range(L)]
    carr = ["%%[sc%d]" % i for i in range(4)]   # carry masks of the multiply-accumulates, round robin (compiler-allocated pairs)
    Pm, B1 = "%[sp]", "%[sb]"
    U, X, SEL, G = ["s%d" % (sbase + 2 * i) for i in range(4)]  # fixed pairs: their halves are named (s_add_u32 / s_mul_i32)
    if K == 8:
        bcast0 = [(("row_newbcast", 0), 0x3), (("row_newbcast", 8), 0xc)]
        down = ("row_shl", 1)       # lane k reads lane k + 1 (a group's top lane reads the next group's lane 0: zero by construction)
        up = ("row_shr", 1)         # lane k reads lane k - 1 (lane 0 of a group reads the previous group's top overflow: zero)
    elif K == 4:
        bcast0 = [(("quad_perm", [0, 0, 0, 0]), 0xf)]
        down = ("quad_perm", [1, 2, 3, 0])
        up = ("quad_perm", [3, 0, 1, 2])
    elif K == 2:
        bcast0 = [(("quad_perm", [0, 0, 2, 2]), 0xf)]
        down = ("quad_perm", [1, 0, 3, 2])
        up = ("quad_perm", [1, 0, 3, 2])
    else:
        raise ValueError(K)
    if lin_only:   # (a +- b) mod r alone, in the lane layout of the products (fused stages of a narrow bundle): riders only
        assert riders
    inc, zero = newpair()   # (incoming word, 0): the 64-bit addend of a fresh top column
    if not lin_only:
        p.mov(zero, 0)
    mreg, mb = newv(), newv()
    # column accumulators: pair (w0, w1) + third word w2 that only collects carries
    cols = [{"pair": newpair(), "w2": newv(), "has2": False} for _ in range(L)]
    spare_pair = newpair() if L == 1 else None
    ci = [0]

    def nextc():
        c = carr[ci[0] % len(carr)]
        ci[0] += 1
        return c

    def acc_carry(c, cy):
        if c["has2"]:
            p.addc(c["w2"], cy, c["w2"], 0, cy)
        else:
            p.addc(c["w2"], cy, 0, 0, cy)
            c["has2"] = True
    for i in range(0 if lin_only else 8):
        # (1) C_j += a_i * b_j
        for j in range(L):
            c = cols[j]
            if i == 0:
                p.mad(c["pair"], nextc(), a[i], b[j], 0)
                c["has2"] = False
            elif j == L - 1 and L > 1:
                p.mad(c["pair"], nextc(), a[i], b[j], (inc, zero))  # fresh top column, 32-bit addend: no carry
                c["has2"] = False
            else:
                cy = nextc()
                p.mad(c["pair"], cy, a[i], b[j], c["pair"])
                acc_carry(c, cy)
        # (2) m = C_0.lo * (-r^-1) of the group's lane 0, broadcast
        p.mul_lo(mreg, cols[0]["pair"][0], "%[inv]")
        for ctrl, bank in bcast0:
            p.mov_dpp(mb, mreg, ctrl, bank)
        # (3) C_j += m * r_j
        for j in range(L):
            c = cols[j]
            cy = nextc()
            p.mad(c["pair"], cy, mb, n[j], c["pair"])
            acc_carry(c, cy)
        # (4) shift by one word: the next lane's C_0.lo arrives, C_0's upper words fold into C_1
        c0 = cols[0]
        if L == 1:
            p.add_dpp(spare_pair[0], c0["pair"][0], c0["pair"][1], down)   # w1 + incoming
            p.addc(spare_pair[1], "vcc", c0["w2"], 0, "vcc")
            cols[0], spare_pair = {"pair": spare_pair, "w2": c0["w2"], "has2": False}, c0["pair"]
        else:
            p.mov_dpp(inc, c0["pair"][0], down)
            c1 = cols[1]
            k = nextc()
            p.addc(c1["pair"][0], k, c1["pair"][0], c0["pair"][1], None)
            p.addc(c1["pair"][1], k, c1["pair"][1], c0["w2"], k)
            acc_carry(c1, k)
            old0 = cols.pop(0)   # its registers hold the next round's top column (written by that round's first mad)
            old0["has2"] = False
            cols.append(old0)
    # ---- normalisation: words W[0..L-1] + a small overflow into the next lane ----------------------------------------
    k = nextc()
    if lin_only:
        W, ov = None, None
    elif L == 1:
        W = [cols[0]["pair"][0]]
        ov = cols[0]["pair"][1]
    elif L == 2:
        c0 = cols[0]
        W = [c0["pair"][0], c0["pair"][1]]
        p.addc(W[1], k, W[1], inc, None)
        p.addc(c0["w2"], k, c0["w2"], 0, k)
        ov = c0["w2"]
    else:  # L == 4: columns C0, C1, C2 (three words each) and the incoming word at position 3
        c0, c1, c2 = cols[0], cols[1], cols[2]
        t0, t1, t2, t3 = c0["pair"][0], c0["pair"][1], c0["w2"], c1["w2"]
        p.addc(t1, k, t1, c1["pair"][0], None)
        p.addc(t2, k, t2, c1["pair"][1], k)
        p.addc(t3, k, t3, 0, k)
        u0, u1, u2 = c2["pair"][0], c2["pair"][1], c2["w2"]
        k2 = nextc()
        p.addc(u1, k2, u1, inc, None)
        p.addc(u2, k2, u2, 0, k2)
        k3 = nextc()
        p.addc(t2, k3, t2, u0, None)
        p.addc(t3, k3, t3, u1, k3)
        p.addc(u2, k3, u2, 0, k3)
        W = [t0, t1, t2, t3]
        ov = u2
    if riders:
        aq = ["%%[aq%d]" % j for j in range(L)]
        sub = "%[sub]"
        Madd, Msub = "%[sma]", "%[sms]"
        if not lin_only:
            p.cmp_eq(Madd, sub, 0)
        p.cmp_eq(Msub, sub, 1)
        mneg = newv()
        p.cndmask(mneg, 0, -1, Msub)          # all ones in subtracting lanes
        bx, nm, u = [newv() for _ in range(L)], [newv() for _ in range(L)], [newv() for _ in range(L)]
        for j in range(L):
            p.xor_(bx[j], mneg, b[j])           # b or ~b
            p.and_(nm[j], mneg, n[j])           # 0 or r
        p.s_op("and", X, Msub, "%[lane0]")     # + 1 at the lowest limb of a subtraction
        c1, c2 = nextc(), nextc()
        uov = newv()
        # two carry chains per limb: a + (0 | r), then + (b | ~b)
        p.addc(u[0], c1, aq[0], nm[0], None)
        p.addc(u[0], c2, u[0], bx[0], X)
        for j in range(1, L):
            p.addc(u[j], c1, aq[j], nm[j], c1)
            p.addc(u[j], c2, u[j], bx[j], c2)
        p.addc(uov, c1, 0, 0, c1)
        p.addc(uov, c2, uov, 0, c2)
        if lin_only:
            W, ov = u, uov
        else:
            p.s_op("or", Madd, Madd, Msub)          # rider lanes
            for j in range(L):
                p.cndmask(W[j], W[j], u[j], Madd)
            p.cndmask(ov, ov, uov, Madd)
        p.cndmask(ov, ov, 0, "%[top]")          # the 2^256 of a subtraction leaves at the group's top lane
    # + overflow of the previous lane (a group's top overflow is zero: the result is below 2r < 2^255)
    ovin = newv()
    p.mov_dpp(ovin, ov, up)
    p.addc(W[0], G, W[0], ovin, None)
    for j in range(1, L):
        p.addc(W[j], G, W[j], 0, G)
    if L == 1:
        allones = W[0]
    else:
        allones = newv()
        p.and_(allones, W[0], W[1])
        for j in range(2, L):
            p.and_(allones, allones, W[j])
    p.cmp_eq(Pm, allones, -1)
    # carry into lane k+1 = g_k | (p_k & carry into lane k): one scalar addition, carries = ((G | P) + G) ^ P.  A group's
    # top lane stays out of the sum (its own carry-in then is the sum bit itself): nothing may run into the next group --
    # a product's top word is below 2^31, but a riding subtraction drops its 2^256 exactly there
    p.s_op("andn2", X, G, "%[top]")
    p.s_op("andn2", Pm, Pm, "%[top]")
    p.s_op("or", U, X, Pm)
    p.s_add64(U, U, X)
    p.s_op("xor", X, U, Pm)
    p.addc(W[0], G, W[0], 0, X)
    for j in range(1, L):
        p.addc(W[j], G, W[j], 0, G)
    # ---- D = W - r, borrows resolved the same way; the top lane's final borrow says W < r ------------------------------
    D = [newv() for _ in range(L)]
    p.addc(D[0], G, W[0], n[0], None, sub=True)
    for j in range(1, L):
        p.addc(D[j], G, W[j], n[j], G, sub=True)
    if L == 1:
        allz = D[0]
    else:
        allz = newv()
        p.and_(allz, D[0], D[1], "or")
        for j in range(2, L):
            p.and_(allz, allz, D[j], "or")
    p.cmp_eq(Pm, allz, 0)
    p.s_op("or", B1, G, G)             # first-pass borrow of every lane (kept for the top lanes)
    # a borrow out of a group's top lane must not run into the next group: the top lanes stay out of the scalar sum,
    # their own borrow-in then is the sum bit itself
    p.s_op("andn2", X, G, "%[top]")
    p.s_op("andn2", Pm, Pm, "%[top]")
    p.s_op("or", U, X, Pm)
    p.s_add64(U, U, X)
    p.s_op("xor", X, U, Pm)
    p.addc(D[0], G, D[0], 0, X, sub=True)
    for j in range(1, L):
        p.addc(D[j], G, D[j], 0, G, sub=True)
    p.s_op("or", SEL, G, B1)            # the two passes cannot both borrow
    p.s_expand(SEL, SEL, K)
    for j in range(L):
        p.cndmask(rr[j], D[j], W[j], SEL)   # borrow: W < r already
    return p, nv[0], sbase + 8


def top_mask(K):
    return sum(1 << i for i in range(K - 1, 64, K))


# ---- emulation against big integers ------------------------------------------------------------------------------------
def check(K, rounds=200, seed=1, riders=False, lin_only=False):
    L = 8 // K
    p, vend, send = make(K, riders=riders, lin_only=lin_only)
    p.schedule()
    rnd = random.Random(seed + K)
    Rinv = pow(1 << 256, -1, P_INT)
    edge = [0, 1, P_INT - 1, P_INT - 2, (1 << 253), (1 << 254) - 1 if (1 << 254) - 1 < P_INT else P_INT - 3, (1 << 128) - 1, (1 << 224) - 1,
            0xffffffff, (P_INT - 1) // 2, sum(0xffffffff << (32 * i) for i in range(7)) % P_INT]
    for it in range(rounds):
        groups = WAVE // K
        A, B = [], []
        for g in range(groups):
            if it < 20:
                A.append(rnd.choice(edge))
                B.append(rnd.choice(edge))
            elif it % 7 == 0:  # products that end up close to a multiple of r / carry-heavy words
                A.append(rnd.randrange(P_INT))
                B.append(pow(A[-1], -1, P_INT) * rnd.choice([1, 2, P_INT - 1]) % P_INT if A[-1] else 0)
            else:
                A.append(rnd.randrange(P_INT))
                B.append(rnd.randrange(P_INT))
        st = St()
        for i in range(8):
            st.v["%%[a%d]" % i] = [(A[l // K] >> (32 * i)) & M32 for l in range(WAVE)]
        for j in range(L):
            st.v["%%[b%d]" % j] = [(B[l // K] >> (32 * ((l % K) * L + j))) & M32 for l in range(WAVE)]
            st.v["%%[n%d]" % j] = [P_LIMBS[(l % K) * L + j] for l in range(WAVE)]
        st.s["%[inv]"] = INV32
        st.s["%[top]"] = top_mask(K)
        subs = [2] * groups
        if riders:
            subs = [rnd.choice([0, 1] if lin_only else [0, 1, 2]) for _ in range(groups)]
            if it % 5 == 0:   # (a - a, a + (r - a), 0 - b: the edges of the correction)
                B = [A[g_] if subs[g_] == 1 else (P_INT - A[g_]) % P_INT if subs[g_] == 0 else B[g_] for g_ in range(groups)]
                for j in range(L):
                    st.v["%%[b%d]" % j] = [(B[l // K] >> (32 * ((l % K) * L + j))) & M32 for l in range(WAVE)]
            for j in range(L):
                st.v["%%[aq%d]" % j] = [(A[l // K] >> (32 * ((l % K) * L + j))) & M32 for l in range(WAVE)]
            st.v["%[sub]"] = [subs[l // K] for l in range(WAVE)]
            st.s["%[lane0]"] = sum(1 << i for i in range(0, 64, K))
        p.run(st)
        for g in range(groups):
            got = 0
            for k in range(K):
                for j in range(L):
                    got |= st.v["%%[r%d]" % j][g * K + k] << (32 * (k * L + j))
            want = A[g] * B[g] * Rinv % P_INT if subs[g] == 2 else (A[g] + B[g]) % P_INT if subs[g] == 0 else (A[g] - B[g]) % P_INT
            assert got == want, "K=%d round %d group %d: a=%x b=%x got %x want %x" % (K, it, g, A[g], B[g], got, want)
    lines, n_nop = p.lines()
    return len(lines) - sum(l.startswith("s_nop") for l in lines) + n_nop, n_nop, vend, send


def emit(K, path, vbase=168, sbase=88, riders=False, lin_only=False):
    L = 8 // K
    p, vend, send = make(K, vbase, sbase, riders, lin_only)
    lines, n_nop = p.lines()
    scal = ["sc0", "sc1", "sc2", "sc3", "sp", "sb"] + (["sms"] if lin_only else ["sma", "sms"] if riders else [])
    outs = ", ".join(['[r%d] "=&v"(r%d)' % (j, j) for j in range(L)] + ['[%s] "=&s"(%s)' % (x, x) for x in scal])
    ins = ", ".join(([] if lin_only else ['[a%d] "v"(a.v[%d])' % (i, i) for i in range(8)]) + ['[b%d] "v"(b%d)' % (j, j) for j in range(L)]
                    + ['[n%d] "v"(n%d)' % (j, j) for j in range(L)] + ([] if lin_only else ['[inv] "s"(inv32)']) + ['[top] "s"(top)']
                    + (['[aq%d] "v"(aq%d)' % (j, j) for j in range(L)] + ['[sub] "v"(sub)', '[lane0] "s"(lane0)'] if riders else []))
    clob = ", ".join(['"v%d"' % r for r in range(vbase, vend)] + ['"s%d"' % r for r in range(sbase, send)] + ['"vcc"', '"scc"'])
    text = ["// GENERATED by tools/codegen/gen_fr_mul_coop.py -- do not edit.  Lane-cooperative %s, groups of %d lanes" % ("(a + b) / (a - b) mod r in the lane layout of the products" if lin_only else "Montgomery product", K),
            "// (%d limbs of b and of the modulus per lane, all of a in every lane)%s: %d issue slots, %d of them wait states; clobbers v%d-v%d, s%d-s%d."
            % (L, ", groups with sub = 0 / 1 add / subtract instead" if riders else "", len(lines) - sum(l.startswith("s_nop") for l in lines) + n_nop, n_nop, vbase, vend - 1, sbase, send - 1),
            "{",
            "    unsigned int " + ", ".join("r%d" % j for j in range(L)) + ";",
            "    unsigned long long " + ", ".join(scal) + ";",
            "    const unsigned int inv32 = 0xefffffffu;",
            "    const unsigned long long top = 0x%xull, lane0 = 0x%xull;" % (top_mask(K), sum(1 << i for i in range(0, 64, K))),
            '    asm volatile("' + "\\n\\t".join(lines) + '"',
            "                 : " + outs,
            "                 : " + ins,
            "                 : " + clob + ");"]
    for j in range(L):
        text.append("    out[%d] = r%d;" % (j, j))
    text.append("}")
    open(path, "w").write("\n".join(text) + "\n")
    return len(lines), n_nop


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "..", "circom-witnesscalc_amd", "csrc")
    # K = 4 is what the interpreter uses (class C_MULQ), with and without linear riders; K = 8 and K = 2 are kept for the
    # microbenchmark (tools/ubench/coop_mul.hip: 624 / 704 / 1220 cycles per dependent product against 1436 for one lane)
    for K, riders, lin_only in ((4, False, False), (4, True, False), (4, True, True), (8, False, False), (2, False, False)):
        slots, nops, vend, send = check(K, rounds=60 if "--check" not in sys.argv else 400, riders=riders, lin_only=lin_only)
        print("K=%d%s: emulation ok, %d issue slots (%d wait states), VGPR temps up to v%d, SGPR up to s%d" % (K, " add/sub only" if lin_only else " + riders" if riders else "", slots, nops, vend - 1, send - 1))
        if "--check" not in sys.argv:
            path = os.path.normpath(os.path.join(dst, "fr_addsub_coop%d_gfx950.inc" % K if lin_only else "fr_mul_coop%d%s_gfx950.inc" % (K, "r" if riders else "")))
            emit(K, path, riders=riders, lin_only=lin_only)
            print("wrote", path)
