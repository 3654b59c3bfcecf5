// BN254 scalar field (Fr) arithmetic for gfx950 lanes: 8 x u32 limbs, Montgomery form with R = 2^256.
//
// Semantics follow the reference's use of ark_bn254::Fr / ruint::U256 in src/graph.rs:102-144,
// 188-197, 221-225, 621-769 (modulus src/field.rs:3-4).  The internal representation is free
// (every exit of the reference goes through into_bigint, src/graph.rs:387): 32-bit limbs are used
// here because the gfx950 integer multiplier is v_mad_u64_u32 (32x32+64 -> 64).
//
// The same header compiles for the host (g++) where the graph compiler converts constants.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FRD __host__ __device__ __forceinline__
#else
#define FRD inline
#endif

namespace cwc {

struct Fr {
    uint32_t v[8];
};

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
#define CWC_P0 0xf0000001u
#define CWC_P1 0x43e1f593u
#define CWC_P2 0x79b97091u
#define CWC_P3 0x2833e848u
#define CWC_P4 0x8181585du
#define CWC_P5 0xb85045b6u
#define CWC_P6 0xe131a029u
#define CWC_P7 0x30644e72u
#define CWC_INV32 0xefffffffu  // -r^-1 mod 2^32

FRD Fr fr_p() { return Fr{{CWC_P0, CWC_P1, CWC_P2, CWC_P3, CWC_P4, CWC_P5, CWC_P6, CWC_P7}}; }
// R mod r  (Montgomery form of 1)
FRD Fr fr_one() { return Fr{{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}}; }
// R^2 mod r
FRD Fr fr_r2() { return Fr{{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}}; }
// (r-1)/2  (reference src/graph.rs:720, halfM)
FRD Fr fr_half() { return Fr{{0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u}}; }
FRD Fr fr_zero() { return Fr{{0, 0, 0, 0, 0, 0, 0, 0}}; }

// a && b / a || b without the short circuit: no branch around b.  (On the GPU a divergent branch inside a region of uniform branches makes
// StructurizeCFG rewrite the uniform ones into flag registers and chains of s_cbranch_vcc*, and the interpreter's class paths lose their
// own back edges: DESIGN 5, layout sensitivity.  Inside the interpreter loop and everything it calls, per-lane conditions are selections.)
FRD bool both(bool a, bool b) { return ((uint32_t)a & (uint32_t)b) != 0u; }
FRD bool either(bool a, bool b) { return ((uint32_t)a | (uint32_t)b) != 0u; }

FRD bool u256_is_zero(const Fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
FRD bool u256_eq(const Fr& a, const Fr& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// r = a + b, returns carry out
// (clang's __builtin_addc / __builtin_subc lower to v_add_co / v_addc_co chains; the uint64 idiom below makes hipcc
// emit 64-bit adds, sign extensions and moves -- ~110 instructions for one modular addition instead of ~35)
// one limb of an add / subtract chain (clang: carry builtins, which hipcc turns into v_addc / v_subb; portable otherwise)
FRD uint32_t adc32(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(__clang__)
    unsigned c = carry;
    const uint32_t r = __builtin_addc(a, b, c, &c);
    carry = c;
    return r;
#else
    const uint64_t t = (uint64_t)a + b + carry;
    carry = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
FRD uint32_t sbb32(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(__clang__)
    unsigned c = borrow;
    const uint32_t r = __builtin_subc(a, b, c, &c);
    borrow = c;
    return r;
#else
    const uint64_t t = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(t >> 63);
    return (uint32_t)t;
#endif
}
FRD uint32_t u256_add(Fr& r, const Fr& a, const Fr& b) {
#if defined(__clang__)
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    return c;
#else
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
#endif
}
// r = a - b, returns borrow out (1 if a < b)
FRD uint32_t u256_sub(Fr& r, const Fr& a, const Fr& b) {
#if defined(__clang__)
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
    return br;
#else
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;  // arithmetic: 0 or -1
    }
    return (uint32_t)(c & 1);
#endif
}
FRD bool u256_lt(const Fr& a, const Fr& b) {
    Fr t;
    return u256_sub(t, a, b) != 0;
}
FRD Fr u256_select(bool c, const Fr& a, const Fr& b) {  // c ? a : b
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// (a + b) mod r for a, b < r
FRD Fr fr_add(const Fr& a, const Fr& b) {
    Fr s, t;
    u256_add(s, a, b);  // < 2^255, no carry out
    uint32_t br = u256_sub(t, s, fr_p());
    return u256_select(br != 0, s, t);
}
// (a - b) mod r
FRD Fr fr_sub(const Fr& a, const Fr& b) {
    Fr d, t;
    uint32_t br = u256_sub(d, a, b);
    u256_add(t, d, fr_p());
    return u256_select(br != 0, t, d);
}
#if defined(__HIPCC__)
// Interpreter versions of fr_add / fr_sub: the two carry chains (sum and correction) interleaved limb by limb in
// inline assembly, so that a lone wavefront never waits on a carry; pv = the modulus held in VGPRs by the caller.
__device__ __forceinline__ Fr fr_add_wave(const Fr& a, const Fr& b, const Fr& pv) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_add_gfx950.inc"
#else
    (void)pv;
    return fr_add(a, b);
#endif
}
__device__ __forceinline__ Fr fr_sub_wave(const Fr& a, const Fr& b, const Fr& pv) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_sub_gfx950.inc"
#else
    (void)pv;
    return fr_sub(a, b);
#endif
}
#endif
// reference src/graph.rs:188-194: 0 -> 0, else r - a   (identical in Montgomery form)
FRD Fr fr_neg(const Fr& a) {
    Fr t;
    u256_sub(t, fr_p(), a);
    return u256_select(u256_is_zero(a), a, t);
}

// Montgomery product a*b/2^256 mod r.  Requires b < r; a may be any value < 2^256
// (result < r after one conditional subtraction since (a*b + m*r)/2^256 < b + r < 2r).
FRD Fr fr_mul(const Fr& a, const Fr& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CWC_PORTABLE_FR_MUL)
#include "fr_mul_gfx950.inc"
#else
    const uint32_t p[8] = {CWC_P0, CWC_P1, CWC_P2, CWC_P3, CWC_P4, CWC_P5, CWC_P6, CWC_P7};
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.v[i] * b.v[j] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        t[8] = (uint32_t)c;  // t < 2r + 2^32 r: fits 9 words
        uint32_t m = t[0] * CWC_INV32;
        c = (uint64_t)m * p[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * p[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;  // quotient < 2r < 2^255: no ninth word after the shift
    }
    Fr r, s;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    uint32_t br = u256_sub(s, r, fr_p());
    return u256_select(br != 0, r, s);
#endif
}
#if defined(__HIPCC__)
// Per-lane add or subtract in one pass (bundles that mix both): m = all ones in subtracting lanes, subm / addm = the
// lane masks of the subtracting / adding lanes (every lane in exactly one of them).
__device__ __forceinline__ Fr fr_addsub_wave(const Fr& a, const Fr& b, const Fr& pv, uint32_t m, unsigned long long subm,
                                             unsigned long long addm) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_addsub_gfx950.inc"
#else
    (void)pv; (void)subm; (void)addm;
    return m ? fr_sub(a, b) : fr_add(a, b);
#endif
}
// Montgomery product as ONE asm block (accumulators in fixed VGPRs v160-v167): no compiler-inserted hazard padding, the
// column shift is one v_pk_mov_b32, the conditional subtraction rides in the tails of the last columns.
__device__ __forceinline__ Fr fr_mul_wave(const Fr& a, const Fr& b, const Fr& pv) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_mul_block_gfx950.inc"
#else
    (void)pv;
    return fr_mul(a, b);
#endif
}
// Lane-cooperative Montgomery product (narrow multiplication bundles, class C_MULQ): four adjacent lanes share one
// product.  Lane 4v + q passes all of a, limbs 2q and 2q+1 of b (b0, b1) and of the modulus (n0, n1) and receives limbs
// 2q, 2q+1 of a*b/2^256 mod r.  148 issue slots against 322 for fr_mul_wave; every lane of the wave must hold operands
// below r (idle groups: zeros).  Generated and emulated by tools/codegen/gen_fr_mul_coop.py.
__device__ __forceinline__ void fr_mul_coop4(const Fr& a, uint32_t b0, uint32_t b1, uint32_t n0, uint32_t n1, uint32_t* out) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_mul_coop4_gfx950.inc"
#else
    (void)a; (void)b0; (void)b1; (void)n0; (void)n1;
    out[0] = out[1] = 0;
#endif
}
// The same with linear nodes riding along (graph.rs:110-111): groups whose sub is SUB_ADD (0) / SUB_SUB (1) return
// (a + b) / (a - b) mod r; aq0, aq1 = the lane's limbs 2q, 2q+1 of a.  167 issue slots.
__device__ __forceinline__ void fr_mul_coop4r(const Fr& a, uint32_t aq0, uint32_t aq1, uint32_t b0, uint32_t b1, uint32_t n0, uint32_t n1, uint32_t sub,
                                              uint32_t* out) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_mul_coop4r_gfx950.inc"
#else
    (void)a; (void)aq0; (void)aq1; (void)b0; (void)b1; (void)n0; (void)n1; (void)sub;
    out[0] = out[1] = 0;
#endif
}
// (a + b) mod r (sub = 0) / (a - b) mod r (sub = 1) per group of four lanes, operands and result in the lane layout of the
// cooperative product (lane 4v + q: limbs 2q, 2q+1): the later stages of a fused narrow bundle (class C_MULF).  60 issue slots.
__device__ __forceinline__ void fr_addsub_coop4(uint32_t aq0, uint32_t aq1, uint32_t b0, uint32_t b1, uint32_t n0, uint32_t n1, uint32_t sub, uint32_t* out) {
#if defined(__HIP_DEVICE_COMPILE__)
#include "fr_addsub_coop4_gfx950.inc"
#else
    (void)aq0; (void)aq1; (void)b0; (void)b1; (void)n0; (void)n1; (void)sub;
    out[0] = out[1] = 0;
#endif
}
#endif
FRD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }

// canonical -> Montgomery (Fr::new): reduces any x < 2^256
FRD Fr fr_to_mont(const Fr& x) { return fr_mul(x, fr_r2()); }
// Montgomery -> canonical (into_bigint): Montgomery reduction of x alone
FRD Fr fr_from_mont(const Fr& x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CWC_PORTABLE_FR_MUL)
#include "fr_from_mont_gfx950.inc"
#else
    const uint32_t p[8] = {CWC_P0, CWC_P1, CWC_P2, CWC_P3, CWC_P4, CWC_P5, CWC_P6, CWC_P7};
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = x.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t m = t[0] * CWC_INV32;
        uint64_t c = (uint64_t)m * p[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * p[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        t[7] = (uint32_t)c;
    }
    Fr r, s;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    uint32_t br = u256_sub(s, r, fr_p());  // x < r  =>  result < r already; kept for x in [r, 2^256)
    return u256_select(br != 0, r, s);
#endif
}

// a^-1 (Montgomery in, Montgomery out) by Fermat: a^(r-2).  a != 0 expected (0 -> 0).
FRD Fr fr_inv_fermat(const Fr& a) {
    // r - 2, most significant limb first
    const uint32_t e[8] = {CWC_P7, CWC_P6, CWC_P5, CWC_P4, CWC_P3, CWC_P2, CWC_P1, CWC_P0 - 2u};
    Fr acc = fr_one();
    for (int w = 0; w < 8; ++w) {
        for (int bit = 31; bit >= 0; --bit) {
            acc = fr_sqr(acc);
            if ((e[w] >> bit) & 1u) acc = fr_mul(acc, a);
        }
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------------
// Modular inverse by Bernstein-Yang "safegcd" divsteps (delta = 1/2 variant), 30-bit batches on signed
// 30-bit limbs: 20 batches x 30 divsteps = 600 >= the 590 needed for a 256-bit modulus.  Uniform control
// flow (no data-dependent branches), ~20k lane instructions instead of ~200k for a Fermat ladder.
// Replaces the field inversion inside Operation::Div (reference src/graph.rs:109, `a / b`).
// ---------------------------------------------------------------------------------------------------
struct S30 {
    int32_t v[9];  // value = sum v[i] * 2^(30 i)
};
struct Trans2x2 {
    int32_t u, v, q, r;
};

FRD int32_t sgcd_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones iff zeta < 0
        uint32_t c2 = 0u - (g & 1u);           // all ones iff g odd
        uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // conditionally negated f, u, v
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                              // swap iff zeta < 0 and g odd
        zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}

// a*b + c on signed 32-bit factors and a signed 64-bit addend: one v_mad_i64_i32 on the device (hipcc expands the
// C expression into unsigned mads plus sign fix-ups, ~10 instructions)
FRD int64_t mad_i64(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    unsigned long long cy;
    // (volatile: the four accumulator chains of sgcd_update_all stay interleaved as written)
    asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (int64_t)a * b + c;
#endif
}

FRD void sgcd_update_fg(S30& f, S30& g, const Trans2x2& t) {
    const int32_t M30 = 0x3fffffff;
    int64_t cf = mad_i64(t.u, f.v[0], mad_i64(t.v, g.v[0], 0));
    int64_t cg = mad_i64(t.q, f.v[0], mad_i64(t.r, g.v[0], 0));
    cf >>= 30;  // the low 30 bits are zero by construction
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf = mad_i64(t.u, f.v[i], mad_i64(t.v, g.v[i], cf));
        cg = mad_i64(t.q, f.v[i], mad_i64(t.r, g.v[i], cg));
        f.v[i - 1] = (int32_t)cf & M30;
        cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// d, e stay in (-2r, r); every step adds the multiple of r that makes the low 30 bits vanish
FRD void sgcd_update_de(S30& d, S30& e, const Trans2x2& t) {
    const int32_t M30 = 0x3fffffff;
    const int32_t p30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
    const uint32_t pinv30 = 0x10000001u;  // r^-1 mod 2^30
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = mad_i64(t.u, d.v[0], mad_i64(t.v, e.v[0], 0));
    int64_t ce = mad_i64(t.q, d.v[0], mad_i64(t.r, e.v[0], 0));
    md -= (int32_t)((pinv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((pinv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd = mad_i64(p30[0], md, cd);
    ce = mad_i64(p30[0], me, ce);
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cd = mad_i64(t.u, d.v[i], mad_i64(t.v, e.v[i], cd));
        ce = mad_i64(t.q, d.v[i], mad_i64(t.r, e.v[i], ce));
        cd = mad_i64(p30[i], md, cd);
        ce = mad_i64(p30[i], me, ce);
        d.v[i - 1] = (int32_t)cd & M30;
        cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// c >>= 30 on a signed 64-bit accumulator as two 32-bit operations (funnel shift + arithmetic shift of the high word)
FRD int64_t sgcd_sar30(int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CWC_SGCD_SHIFT64)
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)((uint64_t)c >> 32);
    const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 30);
    const int32_t nhi = (int32_t)hi >> 30;
    return (int64_t)(((uint64_t)(uint32_t)nhi << 32) | nlo);
#else
    return c >> 30;
#endif
}
// Both updates of a batch in one pass, the four accumulator chains (d, e, f, g) interleaved limb by limb: a
// v_mad_i64_i32 whose result feeds the next instruction costs a wait state, four independent chains never wait
// (round 2 ran update_de and update_fg one after the other: 75 of their 229 issue slots were s_nop).
FRD void sgcd_update_all(S30& d, S30& e, S30& f, S30& g, const Trans2x2& t) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CWC_SGCD_CXX_UPDATE)
    // one asm block (tools/codegen/gen_sgcd_update.py, emulated against big integers there): statement by statement the
    // compiler pads every v_mad_i64_i32 with a wait state
#if defined(CWC_SGCD_UPDATE_INC)
#include CWC_SGCD_UPDATE_INC
#else
#include "sgcd_update_gfx950.inc"
#endif
    return;
#endif
    const int32_t M30 = 0x3fffffff;
    const int32_t p30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
    const uint32_t pinv30 = 0x10000001u;  // r^-1 mod 2^30
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = mad_i64(t.u, d.v[0], 0);
    int64_t ce = mad_i64(t.q, d.v[0], 0);
    int64_t cf = mad_i64(t.u, f.v[0], 0);
    int64_t cg = mad_i64(t.q, f.v[0], 0);
    cd = mad_i64(t.v, e.v[0], cd);
    ce = mad_i64(t.r, e.v[0], ce);
    cf = mad_i64(t.v, g.v[0], cf);
    cg = mad_i64(t.r, g.v[0], cg);
    md -= (int32_t)((pinv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((pinv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cf = sgcd_sar30(cf);  // the low 30 bits are zero by construction
    cg = sgcd_sar30(cg);
    cd = mad_i64(p30[0], md, cd);
    ce = mad_i64(p30[0], me, ce);
    cd = sgcd_sar30(cd);
    ce = sgcd_sar30(ce);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int32_t di = d.v[i], ei = e.v[i], fi = f.v[i], gi = g.v[i];
        cd = mad_i64(t.u, di, cd);
        ce = mad_i64(t.q, di, ce);
        cf = mad_i64(t.u, fi, cf);
        cg = mad_i64(t.q, fi, cg);
        cd = mad_i64(t.v, ei, cd);
        ce = mad_i64(t.r, ei, ce);
        cf = mad_i64(t.v, gi, cf);
        cg = mad_i64(t.r, gi, cg);
        cd = mad_i64(p30[i], md, cd);
        ce = mad_i64(p30[i], me, ce);
        f.v[i - 1] = (int32_t)cf & M30;
        g.v[i - 1] = (int32_t)cg & M30;
        cf = sgcd_sar30(cf);
        cg = sgcd_sar30(cg);
        d.v[i - 1] = (int32_t)cd & M30;
        e.v[i - 1] = (int32_t)ce & M30;
        cd = sgcd_sar30(cd);
        ce = sgcd_sar30(ce);
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// Variable-time batch of 30 divsteps (delta = 1 convention, eta = -delta): trailing zeros of g are consumed in one
// go and up to 6 low bits of g are cancelled per iteration with w = g * f * (f^2 - 2) (f odd: f*(f^2-2) = -1/f mod 64).
// No secrets here, data-dependent time is fine.
//
// Round 3: one iteration is straight-line code (selects instead of the divergent swap branch: a lone wavefront pays
// ~50 cycles for every taken branch and the old loop had three per iteration), the low-bit products are 24-bit
// multiplications (v_mul_u32_u24 is full rate, v_mul_lo_u32 quarter rate; only the low six bits of them are used), and
// the wave leaves the loop when every lane has used up its 30 steps -- a lane that is done idles exactly: with i == 0
// the sentinel makes `zeros` 0, the swap is masked and the cancelled-bit count is 0.  Two iterations per exit test.
FRD uint32_t sgcd_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;  // (written out: the optimiser turns __umul24 of unmasked operands back into the quarter-rate v_mul_lo_u32)
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return (a & 0xffffffu) * (b & 0xffffffu);
#endif
}
FRD bool sgcd_any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(p) != 0ull;
#else
    return p;
#endif
}
FRD int32_t sgcd_divsteps_30_var(int32_t eta, uint32_t f0, uint32_t g0, Trans2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
    int32_t i = 30;
    do {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            // trailing zeros of g, at most i (sentinel bit)
            const int32_t zeros = (int32_t)__builtin_ctz(g | (0xffffffffu << i));  // (i <= 30: the shift is defined)
            g >>= zeros;
            u <<= zeros;
            v <<= zeros;
            eta -= zeros;
            i -= zeros;
            // swap roles where eta < 0 (and steps are left): (f, g) <- (g, -f), matrix rows alike
            const bool sw = eta < 0 && i != 0;
            const uint32_t nf = 0u - f, nu = 0u - u, nv = 0u - v;
            eta = sw ? -eta : eta;
            const uint32_t f2 = sw ? g : f, g2 = sw ? nf : g;
            const uint32_t u2 = sw ? q : u, q2 = sw ? nu : q;
            const uint32_t v2 = sw ? r : v, r2 = sw ? nv : r;
            f = f2; g = g2; u = u2; q = q2; v = v2; r = r2;
            // cancel min(eta + 1, i, 6) low bits of g
            int32_t limit = eta + 1 < i ? eta + 1 : i;
            limit = limit < 6 ? limit : 6;
            limit = limit < 0 ? 0 : limit;  // (a lane that is done with eta < 0: nothing left to cancel)
            const uint32_t w = sgcd_mul24(sgcd_mul24(g, f), sgcd_mul24(f, f) - 2u) & ((1u << limit) - 1u);
            g += f * w;
            q += u * w;
            r += v * w;
        }
    } while (sgcd_any_lane(i != 0));
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return eta;
}

// canonical integer x in [0, r) -> x^-1 mod r (canonical); 0 -> 0
FRD Fr u256_inv_mod_r(const Fr& x) {
    const int32_t M30 = 0x3fffffff;
    const int32_t p30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
    S30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        d.v[i] = 0;
        e.v[i] = 0;
        f.v[i] = p30[i];
        // limb i = bits [30 i, 30 i + 30) of x
        const int lo = (30 * i) >> 5, sh = (30 * i) & 31;
        uint64_t w = x.v[lo];
        if (lo + 1 < 8) w |= (uint64_t)x.v[lo + 1] << 32;
        g.v[i] = (int32_t)((uint32_t)(w >> sh) & (uint32_t)M30);
    }
    e.v[0] = 1;
#if defined(CWC_CONSTANT_TIME_INVERSE)
    int32_t zeta = -1;
    for (int it = 0; it < 20; ++it) {
        Trans2x2 t;
        zeta = sgcd_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        sgcd_update_de(d, e, t);
        sgcd_update_fg(f, g, t);
    }
#else
    int32_t eta = -1;
    for (int it = 0; it < 25; ++it) {  // 25 x 30 = 750 >= the 735-divstep bound of the delta = 1 variant
        Trans2x2 t;
        eta = sgcd_divsteps_30_var(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        sgcd_update_all(d, e, f, g, t);
        int32_t gz = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) gz |= g.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
        if (__ballot(gz != 0) == 0ull) break;  // every lane of the wave is done (extra batches are harmless: g = 0)
#else
        if (gz == 0) break;
#endif
    }
#endif
    // g == 0 now and f == +-1 (x invertible) ; result = d * sign(f), normalised into [0, r)
    const int32_t sign = f.v[8] >> 31;
    int32_t cond_add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] += p30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] = (d.v[i] ^ sign) - sign;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d.v[i + 1] += d.v[i] >> 30;
        d.v[i] &= M30;
    }
    cond_add = d.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] += p30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d.v[i + 1] += d.v[i] >> 30;
        d.v[i] &= M30;
    }
    // back to 8 x u32
    Fr out;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // bits [32k, 32k+32): from limbs (32k)/30 and the next
        const int li = (32 * k) / 30, off = (32 * k) % 30;
        uint64_t w = (uint64_t)(uint32_t)d.v[li] >> off;
        w |= (uint64_t)(uint32_t)d.v[li + 1] << (30 - off);
        if (li + 2 < 9) w |= (uint64_t)(uint32_t)d.v[li + 2] << (60 - off);
        out.v[k] = (uint32_t)w;
    }
    return out;
}

// Montgomery in (aR), Montgomery out (a^-1 R): (aR)^-1 * R^3 / R
FRD Fr fr_inv(const Fr& a) {
    const Fr r3 = Fr{{0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu}};
    return fr_mul(u256_inv_mod_r(a), r3);
}

// logical right shift of a 256-bit value by n in [0, 255]
FRD Fr u256_shr(const Fr& x, uint32_t n) {
    // (word moves as selections, not branches: n differs between the lanes of a wave wherever a graph shifts by a value, and a
    // divergent branch inside the interpreter's uniform class paths makes the structurizer rewrite those -- DESIGN 5)
    Fr a = x;
    const uint32_t w = n >> 5, s = n & 31;
    const bool w4 = (w & 4u) != 0u, w2 = (w & 2u) != 0u, w1 = (w & 1u) != 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = w4 ? ((i + 4 < 8) ? a.v[i + 4 < 8 ? i + 4 : 0] : 0u) : a.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = w2 ? ((i + 2 < 8) ? a.v[i + 2 < 8 ? i + 2 : 0] : 0u) : a.v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) a.v[i] = w1 ? ((i + 1 < 8) ? a.v[i + 1 < 8 ? i + 1 : 0] : 0u) : a.v[i];
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t pair = ((uint64_t)(i + 1 < 8 ? a.v[i + 1] : 0) << 32) | a.v[i];
        r.v[i] = (uint32_t)(pair >> s);
    }
    return r;
}
// left shift by n in [0, 255], bits past 2^256 dropped (ark-ff BigInt::muln)
FRD Fr u256_shl(const Fr& x, uint32_t n) {
    Fr a = x;
    const uint32_t w = n >> 5, s = n & 31;
    const bool w4 = (w & 4u) != 0u, w2 = (w & 2u) != 0u, w1 = (w & 1u) != 0u;
#pragma unroll
    for (int i = 7; i >= 0; --i) a.v[i] = w4 ? ((i - 4 >= 0) ? a.v[i - 4 >= 0 ? i - 4 : 0] : 0u) : a.v[i];
#pragma unroll
    for (int i = 7; i >= 0; --i) a.v[i] = w2 ? ((i - 2 >= 0) ? a.v[i - 2 >= 0 ? i - 2 : 0] : 0u) : a.v[i];
#pragma unroll
    for (int i = 7; i >= 0; --i) a.v[i] = w1 ? ((i - 1 >= 0) ? a.v[i - 1 >= 0 ? i - 1 : 0] : 0u) : a.v[i];
    Fr r;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        uint64_t pair = ((uint64_t)a.v[i] << 32) | (i - 1 >= 0 ? a.v[i - 1] : 0);
        r.v[i] = (uint32_t)((pair << s) >> 32);
    }
    return r;
}

#if defined(__HIPCC__)
#endif

FRD uint32_t u256_bitlen(const Fr& a) {
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (a.v[i]) n = 32u * i + (32u - (uint32_t)__builtin_clz(a.v[i]));
    return n;
}

// q = a / b, rem = a % b on canonical integers (ruint U256 / and %), b != 0.
// Restoring shift-subtract over the low `top` bits of a, the bits above them pre-loaded into the remainder.
// Callers pass top >= bitlen(a) - bitlen(b) + 1 (clamped at 0; wave-wide maximum so that control flow stays
// uniform): then a >> top < 2^(bitlen(b)-1) <= b, the loop invariant rem < b holds from the start, and only the
// quotient's own length is iterated instead of the whole numerator.
FRD void u256_divrem(Fr& q, Fr& rem, const Fr& a, const Fr& b, uint32_t top) {
    // (rem : num) is one 512-bit shift register.  All indices are static (runtime-indexed register arrays would
    // spill to scratch on the GPU).
    Fr num = top ? u256_shl(a, 256u - top) : fr_zero();
    rem = top >= 256u ? fr_zero() : u256_shr(a, top);
    q = fr_zero();
    for (uint32_t i = 0; i < top; ++i) {
        uint32_t carry = num.v[7] >> 31;
#pragma unroll
        for (int k = 7; k > 0; --k) num.v[k] = (num.v[k] << 1) | (num.v[k - 1] >> 31);
        num.v[0] <<= 1;
        uint32_t ov = rem.v[7] >> 31;
#pragma unroll
        for (int k = 7; k > 0; --k) rem.v[k] = (rem.v[k] << 1) | (rem.v[k - 1] >> 31);
        rem.v[0] = (rem.v[0] << 1) | carry;
        Fr t;
        uint32_t br = u256_sub(t, rem, b);
        bool ge = (br == 0) || ov;  // ov: rem overflowed 2^256 (cannot happen for operands < r)
        rem = u256_select(ge, t, rem);
#pragma unroll
        for (int k = 7; k > 0; --k) q.v[k] = (q.v[k] << 1) | (q.v[k - 1] >> 31);
        q.v[0] = (q.v[0] << 1) | (ge ? 1u : 0u);
    }
}

// Digit-wise division (schoolbook / Knuth D with 32-bit digits): q = floor(a / b), rem = a mod b for b != 0.
// Both operands are shifted so that the divisor's top bit sits at bit 255: every quotient digit is then estimated from
// the same register positions (runtime-indexed register arrays would spill to scratch on the GPU), and the 512-bit
// shifted dividend is consumed 32 bits per step.  `digits` (1..8) = how many low quotient digits can be non-zero:
// ceil((bitlen(a) - bitlen(b) + 1) / 32), or the wave-wide maximum of that so that the loop stays uniform (0: a < b
// everywhere).  A step costs ~80 VALU instructions against ~35 per BIT of the restoring division above.
FRD void u256_divrem_digits(Fr& q, Fr& rem, const Fr& a, const Fr& b, uint32_t digits, uint32_t bitlen_b = 0) {
    const uint32_t L = bitlen_b ? bitlen_b : u256_bitlen(b);   // 1..256 (callers pass b != 0)
    const uint32_t sh = 256u - L;        // divisor shifted left by sh has its top bit set
    const Fr bn = u256_shl(b, sh);  // (a shift by 0 is the value: no branch around the shifts)
    // a << sh as 512 bits: hi:lo (sh = 0: hi = 0)
    Fr lo = u256_shl(a, sh);
    uint32_t R[9];
    {
        const Fr hi = u256_select(sh != 0u, u256_shr(a, L & 255u), fr_zero());  // < 2^(256-L) <= bn
#pragma unroll
        for (int i = 0; i < 8; ++i) R[i] = hi.v[i];
        R[8] = 0;
    }
    q = fr_zero();
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        // R = R * 2^32 + next dividend digit   (R < bn before, < bn * 2^32 after)
#pragma unroll
        for (int i = 8; i > 0; --i) R[i] = R[i - 1];
        R[0] = lo.v[j];
        if ((uint32_t)j >= digits) continue;  // the quotient digit is known to be zero (uniform across the wave)
        // qhat = floor(top 64 bits of R / top 32 bits of bn), capped at 2^32 - 1: a floating-point estimate (within one
        // of the floor) corrected with exact 64-bit arithmetic; the true digit is then within [qhat - 2, qhat] (Knuth D)
        const uint32_t d = bn.v[7];
        uint32_t qhat;
        {
            const uint64_t n64 = ((uint64_t)R[8] << 32) | R[7];
            const double n = (double)R[8] * 4294967296.0 + (double)R[7];
            double e = n / (double)d;
            e = e < 4294967295.0 ? e : 4294967295.0;
            uint32_t est = (uint32_t)e;
#if defined(CWC_TEST_PERTURB_QHAT)
            if ((CWC_TEST_PERTURB_QHAT) > 0 ? est != 0xffffffffu : est != 0u) est += (uint32_t)(CWC_TEST_PERTURB_QHAT);  // (host test: the correction below must absorb an estimate that is off by one)
#endif
            const int64_t r = (int64_t)(n64 - (uint64_t)est * d);
            est = r < 0 ? est - 1u : ((uint64_t)r >= d ? est + 1u : est);
            qhat = R[8] >= d ? 0xffffffffu : est;
        }
        // R -= qhat * bn
        uint32_t P[9];
        {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c += (uint64_t)qhat * bn.v[i];
                P[i] = (uint32_t)c;
                c >>= 32;
            }
            P[8] = (uint32_t)c;
        }
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = sbb32(R[i], P[i], borrow);
        // too large (R negative): add the divisor back, at most twice
#pragma unroll
        for (int fix = 0; fix < 2; ++fix) {
            const bool neg = borrow != 0;
            uint32_t carry = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = adc32(R[i], neg ? (i < 8 ? bn.v[i] : 0u) : 0u, carry);
            borrow = neg && !carry ? 1u : 0u;  // still negative iff the addition did not wrap around
            qhat -= neg ? 1u : 0u;
        }
        q.v[j] = qhat;
    }
    Fr r8;
#pragma unroll
    for (int i = 0; i < 8; ++i) r8.v[i] = R[i];
    rem = u256_shr(r8, sh);
}

// Short division: a < 2^128 by b < 2^64 (b != 0) -- the operand sizes of limb-wise big-integer circuits (64-bit limbs:
// products and remainders below 2^128, divisors of one limb).  Same digit-wise scheme as u256_divrem_digits with a
// two-limb divisor normalised to 64 bits and a 96-bit running remainder; q < 2^128, rem < 2^64.
FRD void u128_divrem_64(Fr& q, Fr& rem, const Fr& a, const Fr& b) {
    const uint64_t bv = ((uint64_t)b.v[1] << 32) | b.v[0];
    const uint32_t s = (uint32_t)(b.v[1] ? __builtin_clz(b.v[1]) : 32 + __builtin_clz(b.v[0] | 1u));  // 0..63 (bv != 0)
    const uint64_t bn = bv << s;                      // top bit set
    const uint32_t bn1 = (uint32_t)(bn >> 32), bn0 = (uint32_t)bn;
    // a << s as six limbs (a < 2^128, s < 64)
    uint32_t A[6];
    {
        const uint64_t a_lo = ((uint64_t)a.v[1] << 32) | a.v[0], a_hi = ((uint64_t)a.v[3] << 32) | a.v[2];
        const uint64_t w0 = a_lo << s;
        const uint64_t w1 = (a_hi << s) | (s ? a_lo >> (64 - s) : 0);
        const uint64_t w2 = s ? a_hi >> (64 - s) : 0;
        A[0] = (uint32_t)w0; A[1] = (uint32_t)(w0 >> 32); A[2] = (uint32_t)w1; A[3] = (uint32_t)(w1 >> 32);
        A[4] = (uint32_t)w2; A[5] = (uint32_t)(w2 >> 32);
    }
    // running remainder R = (R2:R1:R0) < bn * 2^32; start with the top two limbs (A[5]:A[4] < 2^64 / ... < bn)
    uint32_t R2 = 0, R1 = A[5], R0 = A[4];
    q = fr_zero();
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        R2 = R1; R1 = R0; R0 = A[j];
        // qhat = floor((R2:R1) / bn1) capped, corrected exactly (as in u256_divrem_digits)
        uint32_t qhat;
        {
            const uint64_t n64 = ((uint64_t)R2 << 32) | R1;
            const double n = (double)R2 * 4294967296.0 + (double)R1;
            double e = n / (double)bn1;
            e = e < 4294967295.0 ? e : 4294967295.0;
            uint32_t est = (uint32_t)e;
#if defined(CWC_TEST_PERTURB_QHAT)
            if ((CWC_TEST_PERTURB_QHAT) > 0 ? est != 0xffffffffu : est != 0u) est += (uint32_t)(CWC_TEST_PERTURB_QHAT);
#endif
            const int64_t r = (int64_t)(n64 - (uint64_t)est * bn1);
            est = r < 0 ? est - 1u : ((uint64_t)r >= bn1 ? est + 1u : est);
            qhat = R2 >= bn1 ? 0xffffffffu : est;
        }
        // (R2:R1:R0) -= qhat * (bn1:bn0)
        const uint64_t p0 = (uint64_t)qhat * bn0;
        const uint64_t p1 = (uint64_t)qhat * bn1 + (p0 >> 32);
        uint32_t borrow = 0;
        R0 = sbb32(R0, (uint32_t)p0, borrow);
        R1 = sbb32(R1, (uint32_t)p1, borrow);
        R2 = sbb32(R2, (uint32_t)(p1 >> 32), borrow);
#pragma unroll
        for (int fix = 0; fix < 2; ++fix) {  // qhat too large by at most two: add the divisor back
            const bool neg = borrow != 0;
            uint32_t carry = 0;
            R0 = adc32(R0, neg ? bn0 : 0u, carry);
            R1 = adc32(R1, neg ? bn1 : 0u, carry);
            R2 = adc32(R2, 0u, carry);
            borrow = neg && !carry ? 1u : 0u;
            qhat -= neg ? 1u : 0u;
        }
        q.v[j] = qhat;
    }
    // R < bn, i.e. R2 == 0; remainder = (R1:R0) >> s
    const uint64_t r64 = (((uint64_t)R1 << 32) | R0) >> s;
    rem = fr_zero();
    rem.v[0] = (uint32_t)r64;
    rem.v[1] = (uint32_t)(r64 >> 32);
}

// ---- division by an invariant limb (the long-division chains of scan bundles: one divisor for the rounds of a loop) ----------
// Moeller & Granlund, "Improved division by invariant integers" (IEEE TC 2011), algorithms 3 and 4: for a NORMALISED divisor dn
// (top bit set) the 64-bit reciprocal v = floor((2^128 - 1) / dn) - 2^64 turns (u1:u0) / dn with u1 < dn into two
// multiplications and two corrections.
FRD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
FRD uint32_t clz64_nonzero(uint64_t x) {  // x != 0
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    const uint32_t c_hi = (uint32_t)__builtin_clz(hi | 1u), c_lo = 32u + (uint32_t)__builtin_clz(lo | 1u);  // (both computed: a selection, no branch)
    return hi ? c_hi : c_lo;
}
FRD uint64_t recip64(uint64_t dn) {  // floor((2^128 - 1) / dn) - 2^64 for a normalised dn (top bit set); once per divisor
    // Moller & Granlund, "Improved division by invariant integers" (IEEE TC 2011), Algorithm 3: an 11-bit start value (their table
    // entry floor((2^19 - 3 * 2^8) / d9), here one small division in single precision, corrected by comparison), then three
    // Newton steps in 64-bit integer arithmetic -- ~20 multiply-adds instead of the four double divisions of a digit-wise long
    // division.  tests/native/div_recip_test.cc: both ends of the range, powers of two +- 1 and 2 M random divisors against
    // unsigned __int128 (50 M more when the algorithm was transcribed).
    const uint64_t d0 = dn & 1ull, d40 = (dn >> 24) + 1ull, d63 = (dn >> 1) + d0;
    const uint32_t d9 = (uint32_t)(dn >> 55);  // 256 .. 511
    uint32_t v0 = (uint32_t)(523520.0f / (float)d9);
    v0 -= v0 * d9 > 523520u ? 1u : 0u;
    v0 += (v0 + 1u) * d9 <= 523520u ? 1u : 0u;
    const uint64_t v1 = ((uint64_t)v0 << 11) - (((uint64_t)(v0 * v0) * d40) >> 40) - 1ull;
    const uint64_t v2 = (v1 << 13) + ((v1 * ((1ull << 60) - v1 * d40)) >> 47);
    const uint64_t e = (0ull - v2 * d63) + ((v2 >> 1) & (0ull - d0));
    const uint64_t v3 = (v2 << 31) + (mulhi64(v2, e) >> 1);
    // v4 = v3 - floor((v3 + 2^64 + 1) dn / 2^64)
    const uint64_t lo = v3 * dn, hi = mulhi64(v3, dn);
    const uint64_t lo2 = lo + dn;
    return v3 - (hi + dn + (lo2 < lo ? 1ull : 0ull));
}
FRD void div2by1(uint64_t u1, uint64_t u0, uint64_t dn, uint64_t v, uint64_t& q, uint64_t& r) {  // u1 < dn, dn normalised, v = recip64(dn)
    const uint64_t lo = v * u1, hi = mulhi64(v, u1);
    const uint64_t q0 = lo + u0;
    uint64_t q1 = hi + u1 + (q0 < lo ? 1ull : 0ull) + 1ull;
    uint64_t rr = u0 - q1 * dn;
    if (rr > q0) {
        --q1;
        rr += dn;
    }
    if (rr >= dn) {
        ++q1;
        rr -= dn;
    }
    q = q1;
    r = rr;
}
// (th:tl) / d for any th, tl and d != 0 given s = clz(d), dn = d << s, v = recip64(dn): quotient (qh:ql), remainder.
// `high`: whether the high word needs a division of its own (th >= d somewhere; callers test it wave-wide).
FRD void u128_divrem_64_recip(uint64_t th, uint64_t tl, uint64_t d, uint32_t s, uint64_t dn, uint64_t v, bool high, uint64_t& qh, uint64_t& ql, uint64_t& rem) {
    uint64_t rh = th;  // th mod d
    qh = 0;
    if (high) {  // (0:th) << s: the top word (th >> (64 - s)) is below 2^s <= dn
        uint64_t rn;
        div2by1((th >> 1) >> (63u - s), th << s, dn, v, qh, rn);
        rh = rn >> s;
    }
    (void)d;
    uint64_t rn;
    div2by1((rh << s) | ((tl >> 1) >> (63u - s)), tl << s, dn, v, ql, rn);
    rem = rn >> s;
}

// ---- division by a two-word divisor (round 5: the quotient-digit estimates of multi-register long division, a 2n-bit value by an n-bit
// register with 64 < n <= 128: zk-email's 121-bit registers) ------------------------------------------------------------------------
// Moeller & Granlund, "Improved division by invariant integers" (IEEE TC 2011): Algorithm 6 refines the one-word reciprocal of d1 into the
// reciprocal v of the normalised two-word divisor (d1:d0), v = floor((2^192 - 1) / (d1:d0)) - 2^64; Algorithm 5 divides three words by
// two with it: (u2:u1:u0) / (d1:d0) with (u2:u1) < (d1:d0) -> one quotient word and a two-word remainder, two multiplications.
FRD uint64_t recip64_3by2(uint64_t d1, uint64_t d0) {  // d1 normalised (top bit set)
    uint64_t v = recip64(d1);
    uint64_t p = d1 * v + d0;
    // (the algorithm's conditional corrections as selections: no per-lane branch)
    const bool c1 = p < d0, c2 = both(c1, p >= d1);
    v -= (c1 ? 1ull : 0ull) + (c2 ? 1ull : 0ull);
    p -= (c2 ? d1 : 0ull);
    p -= (c1 ? d1 : 0ull);
    const uint64_t t1 = mulhi64(v, d0), t0 = v * d0;
    p += t1;
    const bool c3 = p < t1, c4 = both(c3, either(p > d1, both(p == d1, t0 >= d0)));
    v -= (c3 ? 1ull : 0ull) + (c4 ? 1ull : 0ull);
    return v;
}
FRD void div3by2(uint64_t u2, uint64_t u1, uint64_t u0, uint64_t d1, uint64_t d0, uint64_t v, uint64_t& q, uint64_t& r1, uint64_t& r0) {
    uint64_t q0 = v * u2, q1 = mulhi64(v, u2);
    q0 += u1;
    q1 += u2 + (q0 < u1 ? 1ull : 0ull);
    uint64_t a1 = u1 - q1 * d1;
    const uint64_t t1 = mulhi64(d0, q1), t0 = d0 * q1;
    // (a1:a0) = (a1:u0) - (t1:t0) - (d1:d0)  (mod 2^128)
    uint64_t a0 = u0 - t0;
    a1 -= t1 + (u0 < t0 ? 1ull : 0ull);
    const uint64_t b0 = a0 - d0;
    a1 -= d1 + (a0 < d0 ? 1ull : 0ull);
    a0 = b0;
    ++q1;
    {
        const bool f = a1 >= q0;
        q1 -= f ? 1ull : 0ull;
        const uint64_t c0 = a0 + d0, c1 = a1 + d1 + (c0 < a0 ? 1ull : 0ull);
        a1 = f ? c1 : a1;
        a0 = f ? c0 : a0;
    }
    {
        const bool f = either(a1 > d1, both(a1 == d1, a0 >= d0));
        q1 += f ? 1ull : 0ull;
        const uint64_t c0 = a0 - d0, c1 = a1 - (d1 + (a0 < d0 ? 1ull : 0ull));
        a1 = f ? c1 : a1;
        a0 = f ? c0 : a0;
    }
    q = q1;
    r1 = a1;
    r0 = a0;
}
// q = floor(a / b), rem = a mod b for any a < 2^256 and 2^64 <= b < 2^128 (graph.rs:112-121 on such operands): three quotient words.
FRD void u256_divrem_128(Fr& q, Fr& rem, const Fr& a, const Fr& b) {
    const uint64_t bh = ((uint64_t)b.v[3] << 32) | b.v[2], bl = ((uint64_t)b.v[1] << 32) | b.v[0];
    const uint32_t s = clz64_nonzero(bh);  // bh != 0
    const uint64_t d1 = (bh << s) | ((bl >> 1) >> (63u - s)), d0 = bl << s;
    const uint64_t v = recip64_3by2(d1, d0);
    uint64_t aw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) aw[k] = ((uint64_t)a.v[2 * k + 1] << 32) | a.v[2 * k];
    uint64_t u[5];  // a << s
    u[0] = aw[0] << s;
#pragma unroll
    for (int k = 1; k < 4; ++k) u[k] = (aw[k] << s) | ((aw[k - 1] >> 1) >> (63u - s));
    u[4] = (aw[3] >> 1) >> (63u - s);
    uint64_t qw[3];
#pragma unroll
    for (int j = 2; j >= 0; --j) {
        uint64_t r1, r0;
        div3by2(u[j + 2], u[j + 1], u[j], d1, d0, v, qw[j], r1, r0);
        u[j + 1] = r1;
        u[j] = r0;
    }
    q = fr_zero();
    rem = fr_zero();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        q.v[2 * j] = (uint32_t)qw[j];
        q.v[2 * j + 1] = (uint32_t)(qw[j] >> 32);
    }
    const uint64_t rl = (u[0] >> s) | ((u[1] << 1) << (63u - s)), rh = u[1] >> s;
    rem.v[0] = (uint32_t)rl;
    rem.v[1] = (uint32_t)(rl >> 32);
    rem.v[2] = (uint32_t)rh;
    rem.v[3] = (uint32_t)(rh >> 32);
}

}  // namespace cwc
