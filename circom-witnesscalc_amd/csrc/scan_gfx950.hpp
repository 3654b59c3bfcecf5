// Scan bundles (class C_SCAN, program_dev.h): wave-level helpers of the interpreter's scan path -- the accumulator's move up the
// wave, the limb-sized serial rounds, and the parallel forms of the two recurrences for 64-bit limbs.  Included by kernels.hip
// and by tools/ubench/scan_par_test.hip (the parallel forms against the serial recurrences on one wave).
#pragma once
#include "fr_gfx950.hpp"

namespace cwc {

// ---- scan bundles (class C_SCAN, program_dev.h): helpers -------------------------------------------------------------
// Lane l takes lane l - D's value across the whole wave (DPP wave_shr:1, a gfx9 control; the first D lanes take zero).
// Call it as a statement of its own, never inside the unevaluated arm of `c ? a : wave_shr_lanes(v)`: there the move runs with
// the other lanes switched off, and a DPP move reads zero from a lane that is switched off (tools/ubench/scan_par_test.hip).
template <int D>
__device__ __forceinline__ uint32_t wave_shr_lanes(uint32_t v) {
#pragma unroll
    for (int i = 0; i < D; ++i) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return v;
}
template <int QP>
__device__ __forceinline__ Fr fr_quad_perm(const Fr& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], QP, 0xf, 0xf, false);
    return r;
}
// word k of 2^n - 1 (wave-uniform n)
__device__ __forceinline__ uint32_t mask_word(uint32_t n, uint32_t k) { return n >= 32u * (k + 1u) ? 0xffffffffu : n > 32u * k ? (1u << (n - 32u * k)) - 1u : 0u; }
// The carry chain's limb-sized rounds: x < 2^128 and every accumulator < 2^128 (so t = x + acc < 2^129), 1 <= n <= 128 with
// n = 32 WS + bs.  acc' = t >> n, limb = t & (2^n - 1); the accumulator moves D lanes up the wave between rounds.
template <int WS, int D>
__device__ __forceinline__ void scan_carry_rounds(uint32_t iters, uint32_t bs, const uint32_t (&m)[4], bool start, const uint32_t (&x)[4], const uint32_t (&a0)[4],
                                                  uint32_t (&limb)[4], uint32_t (&carry)[4]) {
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t in[4], t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t sft = wave_shr_lanes<D>(c[k]);
            in[k] = start ? a0[k] : sft;
        }
        uint32_t cy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = adc32(x[k], in[k], cy);
        t[4] = cy;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            limb[k] = t[k] & m[k];
            c[k] = __builtin_amdgcn_alignbit(t[k + WS + 1], t[k + WS], bs);  // (t >> n, word k)
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) carry[k] = c[k];
}

// One-bit recurrence over the pairs of a scan bundle, every segment at once: c_out = gen | (prop & c_in) with c_in = 0 where a
// segment starts (`st`).  Lanes of set t = lane mod T; per pair the OUT lanes carry a gate bit (0 at a segment's start), the ACC
// lanes the generate / propagate bits, and ONE 64-bit integer addition on the scalar unit ripples the carries through them
// (carries into every bit = (a + b) ^ a ^ b).  Returns the carry INTO the lane's pair (the same in its OUT and ACC lanes).
template <int T>
__device__ __forceinline__ uint32_t scan_bit_lookahead(bool st, bool gen, bool prop, uint32_t lane) {
    constexpr uint64_t OUT_LANES = T == 1 ? 0x5555555555555555ull : 0x3333333333333333ull;
    const uint64_t b_gate = __ballot(!st) & OUT_LANES, b_pg = __ballot(gen || prop) & ~OUT_LANES, b_g = __ballot(gen) & ~OUT_LANES;
    uint64_t cbits = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint64_t lt = T == 1 ? ~0ull : (t == 0 ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull);
        const uint64_t a = ((b_gate | b_pg) & lt) | ~lt, bb = b_g & lt;
        cbits |= ((a + bb) ^ a ^ bb) & lt;
    }
    return (uint32_t)(cbits >> (lane | (uint32_t)T)) & 1u;
}
// Parallel forms of the two recurrences for 64-bit limbs (n = k = 64), a bundle's chain segments all at once instead of round by
// round.  `st`: the lane's pair starts a segment (START, or an idle pair); values are the same in both lanes of a pair.
// Carry chain.  X = sum x_p B^p (B = 2^64, a segment's incoming accumulator added to its first x), every x_p < 2^192 as three
// words x0 + x1 B + x2 B^2: the words that meet at position p are s_p = x0_p + x1_(p-1) + x2_(p-2) < 3 B.  Two local rounds bring the
// carries down to one bit per position (s = lo + B ov; lo + ov_(p-1) = lo' + B w; z = lo' + w_(p-1) <= B), the last one is a
// carry-lookahead over the wave: per pair a gate bit (0 at a segment's start) and a generate / propagate bit, one 64-bit integer
// addition on the scalar unit ripples the carries through them.  limb_p = (z_p + c_p) mod B; the carry leaving position p is
// x1_p + x2_p B + x2_(p-1) + ov_p + w_p + c_(p+1) -- the words of x_p and x_(p-1) above position p plus what the positions up to p
// push out.  (tests/test_host_formats.py holds the same algorithm on plain integers against the serial recurrence.)
template <int T>
__device__ __forceinline__ void scan_carry_parallel(bool st, uint32_t lane, const uint32_t (&xp)[6], uint32_t (&limb)[2], uint32_t (&carry)[6]) {
    constexpr int D = 2 * T;
    auto u64 = [](uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; };
    auto shr64 = [&](uint64_t v) {  // the previous pair's value, nothing at a segment's start
        const uint32_t lo = wave_shr_lanes<D>((uint32_t)v), hi = wave_shr_lanes<D>((uint32_t)(v >> 32));
        return st ? 0ull : u64(lo, hi);
    };
    const uint64_t x0 = u64(xp[0], xp[1]), x1 = u64(xp[2], xp[3]), x2 = u64(xp[4], xp[5]);
    const uint64_t y1 = shr64(x1), y2a = shr64(x2), y2 = shr64(y2a);
    uint64_t lo = x0 + y1;
    uint32_t ov = lo < x0 ? 1u : 0u;
    const uint64_t lo_b = lo + y2;
    ov += lo_b < lo ? 1u : 0u;
    const uint32_t ov_sh = wave_shr_lanes<D>(ov), ovp = st ? 0u : ov_sh;
    const uint64_t lo2 = lo_b + ovp;
    const uint32_t w = lo2 < lo_b ? 1u : 0u;
    const uint32_t w_sh = wave_shr_lanes<D>(w), wp = st ? 0u : w_sh;
    const uint64_t z = lo2 + wp;
    const bool gen = z < lo2, prop = z == ~0ull;
    const uint32_t cin = scan_bit_lookahead<T>(st, gen, prop, lane);  // the carry into the pair's ACC lane of this set
    const uint64_t dgt = z + cin;
    const uint32_t cout = (gen || (prop && cin)) ? 1u : 0u;
    limb[0] = (uint32_t)dgt;
    limb[1] = (uint32_t)(dgt >> 32);
    // carry = x1 + x2 B + x2_(p-1) + ov + w + cout  (< 2^129 + ...: three words and a bit)
    const uint32_t small = ov + w + cout;
    uint64_t c0 = x1 + y2a;
    uint64_t k1 = c0 < x1 ? 1ull : 0ull;
    const uint64_t c0b = c0 + small;
    k1 += c0b < c0 ? 1ull : 0ull;
    const uint64_t c1 = x2 + k1;
    carry[0] = (uint32_t)c0b;
    carry[1] = (uint32_t)(c0b >> 32);
    carry[2] = (uint32_t)c1;
    carry[3] = (uint32_t)(c1 >> 32);
    carry[4] = c1 < x2 ? 1u : 0u;
    carry[5] = 0u;
}
// The same for registers of any width n <= 126 bits (round 5: the 121-bit registers of circom-bigint's RSA circuits), B = 2^n.
// X = x + (a segment's incoming accumulator at its first position) < min(B^3, 2^252) everywhere in the wave: three digits d0 + d1 B +
// d2 B^2 below 2^126 each, 128-bit arithmetic on four words; s_p = d0_p + d1_(p-1) + d2_(p-2) < 3 B <= 2^128; the two local rounds,
// the lookahead, limb_p = (z_p + c_p) mod B and the carry leaving position p = d1_p + d2_(p-1) + ov_p + w_p + c_out + d2_p B are
// those of scan_carry_parallel.  Below 2^252 no sum x_p + carry reaches r (carry <= t / 2): the serial recurrence's field
// additions (graph.rs:110) are plain integer additions.
template <int T, int WS>
__device__ __forceinline__ void scan_carry_parallel_wide(bool st, uint32_t lane, uint32_t n, const Fr& xp, Fr& limb, Fr& carry) {
    constexpr int D = 2 * T;
    const uint32_t bs = n & 31u;  // n = 32 WS + bs
    const uint32_t m[4] = {mask_word(n, 0), mask_word(n, 1), mask_word(n, 2), mask_word(n, 3)};
    auto bit_n = [&](const uint32_t (&v)[4]) -> uint32_t { return (v[WS] >> bs) & 1u; };  // bit n of a value below 2^(n + 1) <= 2^127
    auto add4 = [](const uint32_t (&a)[4], const uint32_t (&b)[4], uint32_t (&o)[4]) {
        uint32_t cy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = adc32(a[k], b[k], cy);
    };
    auto add_small = [](const uint32_t (&a)[4], uint32_t b, uint32_t (&o)[4]) {
        uint32_t cy = 0;
        o[0] = adc32(a[0], b, cy);
#pragma unroll
        for (int k = 1; k < 4; ++k) o[k] = adc32(a[k], 0u, cy);
    };
    auto prev4 = [&](const uint32_t (&v)[4], uint32_t (&o)[4]) {  // the previous pair's value, nothing at a segment's start
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t sft = wave_shr_lanes<D>(v[k]);
            o[k] = st ? 0u : sft;
        }
    };
    // t1 = X >> n (its words above 2^(256 - n) are zero), d2 = t1 >> n (below 2^n by the precondition)
    uint32_t t1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t lo_w = k + WS < 8 ? xp.v[k + WS < 8 ? k + WS : 7] : 0u, hi_w = k + WS + 1 < 8 ? xp.v[k + WS + 1 < 8 ? k + WS + 1 : 7] : 0u;
        t1[k] = __builtin_amdgcn_alignbit(hi_w, lo_w, bs);
    }
    uint32_t d0[4], d1[4], d2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d0[k] = xp.v[k] & m[k];
        d1[k] = t1[k] & m[k];
        d2[k] = __builtin_amdgcn_alignbit(t1[k + WS + 1], t1[k + WS], bs);
    }
    uint32_t y1[4], y2a[4], y2[4], s[4], s2[5];
    prev4(d1, y1);
    prev4(d2, y2a);
    prev4(y2a, y2);
    add4(d0, y1, s);
    {
        uint32_t cy = 0;  // (3 B may reach 2^128 at n = 126: keep the carry)
#pragma unroll
        for (int k = 0; k < 4; ++k) s2[k] = adc32(s[k], y2[k], cy);
        s2[4] = cy;
    }
    const uint32_t ov = __builtin_amdgcn_alignbit(s2[WS + 1], s2[WS], bs) & 3u;  // s >> n: 0 .. 2
    uint32_t lo[4], lo2[4], z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lo[k] = s2[k] & m[k];
    const uint32_t ov_sh = wave_shr_lanes<D>(ov), ovp = st ? 0u : ov_sh;
    add_small(lo, ovp, lo2);  // < B + 2
    const uint32_t w = bit_n(lo2);
#pragma unroll
    for (int k = 0; k < 4; ++k) lo2[k] &= m[k];
    const uint32_t w_sh = wave_shr_lanes<D>(w), wp = st ? 0u : w_sh;
    add_small(lo2, wp, z);  // <= B
    const bool gen = bit_n(z) != 0u, prop = ((z[0] ^ m[0]) | (z[1] ^ m[1]) | (z[2] ^ m[2]) | (z[3] ^ m[3])) == 0u;
    const uint32_t cin = scan_bit_lookahead<T>(st, gen, prop, lane);
    uint32_t dg[4];
    add_small(z, cin, dg);
    limb = fr_zero();
#pragma unroll
    for (int k = 0; k < 4; ++k) limb.v[k] = dg[k] & m[k];
    const uint32_t cout = (gen || (prop && cin)) ? 1u : 0u;
    uint32_t c0[4], c1[4];
    add4(d1, y2a, c0);                  // < 2^127
    add_small(c0, ov + w + cout, c1);   // < 2^127 + 4
    // + d2 << n: word j of the shifted digit = (d2[j - WS] << bs) | (d2[j - WS - 1] >> (32 - bs))
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t hi_w = (j - WS >= 0 && j - WS < 4) ? d2[j - WS >= 0 && j - WS < 4 ? j - WS : 0] : 0u;
        const uint32_t lo_w = (j - WS - 1 >= 0 && j - WS - 1 < 4) ? d2[j - WS - 1 >= 0 && j - WS - 1 < 4 ? j - WS - 1 : 0] : 0u;
        const uint32_t sh_w = bs ? __builtin_amdgcn_alignbit(hi_w, lo_w, 32u - bs) : hi_w;
        carry.v[j] = adc32(j < 4 ? c1[j < 4 ? j : 0] : 0u, sh_w, cy);
    }
}
// Long division by one limb d (the same for all steps of a segment, every incoming remainder below it).  A step is the map
// r -> (r B + x) mod d = (r m + v) mod d with m = B mod d, v = x mod d; maps compose ((m1, v1) then (m2, v2) = (m1 m2, v1 m2 + v2)),
// so an inclusive segmented prefix over the pairs (log2 rounds; a segment's first step takes its incoming remainder in: m = 0,
// v = (acc B + x) mod d) leaves every step's outgoing remainder, and the quotient digit is one more two-by-one division of
// (incoming remainder : x).  Products modulo d: a b < d^2 has its high word below d, the precondition of div2by1.
// The two lanes of a pair hold the same values, so they share the work: a round's two products (v' needs pv m, m' needs pm m)
// are ONE product per lane -- the OUT lane's for v, the ACC lane's for m -- exchanged inside the quad afterwards.
template <int T>
__device__ __forceinline__ void scan_div_parallel(bool st, uint32_t lane, uint32_t rounds_cover, uint64_t dv, uint64_t x, uint64_t a0, uint64_t& quo, uint64_t& rem) {
    constexpr int D = 2 * T;
    constexpr int QP_OUT = T == 1 ? 0xA0 /* [0,0,2,2] */ : 0x44 /* [0,1,0,1] */, QP_ACC = T == 1 ? 0xF5 /* [1,1,3,3] */ : 0xEE /* [2,3,2,3] */;
    const bool acc_lane = ((lane / (uint32_t)T) & 1u) != 0;
    auto from_out = [&](uint64_t v) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, QP_OUT, 0xf, 0xf, false), hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), QP_OUT, 0xf, 0xf, false);
        return ((uint64_t)hi << 32) | lo;
    };
    auto from_acc = [&](uint64_t v) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, QP_ACC, 0xf, 0xf, false), hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), QP_ACC, 0xf, 0xf, false);
        return ((uint64_t)hi << 32) | lo;
    };
    const uint32_t s = clz64_nonzero(dv);
    const uint64_t dn = dv << s, rv = recip64(dn);
    auto divrem = [&](uint64_t hi, uint64_t lo, uint64_t& q) -> uint64_t {  // hi < dv
        uint64_t rn;
        div2by1((hi << s) | ((lo >> 1) >> (63u - s)), lo << s, dn, rv, q, rn);
        return rn >> s;
    };
    uint64_t qd;
    // OUT lane: v = ((incoming remainder at a segment's start) B + x) mod d; ACC lane: m = B mod d = (B - d) mod d
    const uint64_t t0 = divrem(acc_lane ? 0ull : (st ? a0 : 0ull), acc_lane ? 0ull - dv : x, qd);
    const uint64_t t0v = from_out(t0), t0m = from_acc(t0);
    uint64_t v = t0v, m = st ? 0ull : t0m;
    bool f = st;
    // (rounds written out: the compiler does not unroll a loop of cross-lane operations by a run-time count, and a lone wave pays ~70 cycles
    // for every taken branch, the loop's back edge included -- five rounds cover a bundle's 32 steps, more stay a loop)
    auto round = [&](uint32_t dl) {
        const int src = (int)((lane - dl * (uint32_t)D) << 2);
        const uint64_t mine = acc_lane ? m : v;  // (the partner pair's lane of the same role holds the same v and m)
        const uint64_t theirs = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(mine >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)mine);
        const bool pf = __builtin_amdgcn_ds_bpermute(src, f ? 1 : 0) != 0;
        // v' = (pv m + v) mod d, m' = (pm m) mod d
        const uint64_t t = divrem(mulhi64(theirs, m), theirs * m, qd);
        const uint64_t t1 = from_out(t), nm = from_acc(t);
        uint64_t nv = t1 + v;
        nv -= (nv < t1 || nv >= dv) ? dv : 0ull;
        v = f ? v : nv;
        m = f ? m : nm;
        f = f || pf;
    };
    if (1u < rounds_cover) round(1u);
    if (2u < rounds_cover) round(2u);
    if (4u < rounds_cover) round(4u);
    if (8u < rounds_cover) round(8u);
    if (16u < rounds_cover) round(16u);
    for (uint32_t dl = 32; dl < rounds_cover; dl <<= 1) round(dl);
    const uint32_t r0 = wave_shr_lanes<D>((uint32_t)v), r1 = wave_shr_lanes<D>((uint32_t)(v >> 32));
    const uint64_t rin = st ? a0 : (((uint64_t)r1 << 32) | r0);
    (void)divrem(rin, x, quo);
    rem = v;
}

// Convolution bundles (HDR_SCAN_CONV): the columns of a k x k limb product, x, y < 2^64.  Lane c of set t (lane = c T + t) holds
// x_c and y_c (y = 0 in the lanes of the columns k and above).  Round i: x_i is read out of its lane, every lane adds
// x_i * (the y it holds) to its 192-bit sum, the y's move one column up the wave: lane c meets y_(c-i) in round i, zero when
// c - i is outside [0, k).
template <int T>
__device__ __forceinline__ void conv_limb_columns(uint32_t k, uint32_t lane, uint64_t x, uint64_t y, uint32_t (&out)[5]) {
    // three 64-bit sums of 32 x 32 products by weight (1: x0 y0; 2^32: x0 y1 + x1 y0; 2^64: x1 y1), each with a counter of its
    // overflows: a round is four v_mad_u64_u32 whose carry-outs (the instruction's scalar destination, which the compiler has no
    // builtin for) go straight into the counters -- 8 instructions instead of 4 multiply-adds + 15 of carry handling
    uint64_t s_lo = 0, s_mid = 0, s_hi = 0;
    uint32_t c_lo = 0, c_mid = 0, c_hi = 0;
    uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
    auto mac = [](uint64_t& sum, uint32_t& overflows, uint32_t xs, uint32_t yv) {  // sum += xs * yv (xs in a scalar register), overflows += carry
        uint64_t cy, unused;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(sum), "=s"(cy) : "s"(xs), "v"(yv));
        asm volatile("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(overflows), "=s"(unused) : "s"(cy));
    };
    auto round = [&](uint32_t i) {
        uint32_t b0, b1;
        if constexpr (T == 1) {
            b0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)i);
            b1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, (int)i);
            mac(s_lo, c_lo, b0, y0);
            mac(s_mid, c_mid, b0, y1);
            mac(s_mid, c_mid, b1, y0);
            mac(s_hi, c_hi, b1, y1);
        } else {  // (two sets side by side: each lane takes its own set's x_i, a vector operand)
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)(2 * i)), e1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, (int)(2 * i));
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, (int)(2 * i + 1)), o1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, (int)(2 * i + 1));
            b0 = (lane & 1u) ? o0 : e0;
            b1 = (lane & 1u) ? o1 : e1;
            auto macv = [](uint64_t& sum, uint32_t& overflows, uint32_t xv, uint32_t yv) {
                uint64_t cy, unused;
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(sum), "=s"(cy) : "v"(xv), "v"(yv));
                asm volatile("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(overflows), "=s"(unused) : "s"(cy));
            };
            macv(s_lo, c_lo, b0, y0);
            macv(s_mid, c_mid, b0, y1);
            macv(s_mid, c_mid, b1, y0);
            macv(s_hi, c_hi, b1, y1);
        }
        y0 = wave_shr_lanes<T>(y0);
        y1 = wave_shr_lanes<T>(y1);
    };
    uint32_t i = 0;
    for (; i + 8u <= k; i += 8u) {  // (eight rounds per back edge: written out, see scan_div_parallel)
        round(i);
        round(i + 1u);
        round(i + 2u);
        round(i + 3u);
        round(i + 4u);
        round(i + 5u);
        round(i + 6u);
        round(i + 7u);
    }
    if (i + 4u <= k) {
        round(i);
        round(i + 1u);
        round(i + 2u);
        round(i + 3u);
        i += 4u;
    }
    for (; i < k; ++i) round(i);
    // total = s_lo + c_lo 2^64 + s_mid 2^32 + c_mid 2^96 + s_hi 2^64 + c_hi 2^128  (< 2^134)
    out[0] = (uint32_t)s_lo;
    const uint64_t t1 = (s_lo >> 32) + (uint32_t)s_mid;
    out[1] = (uint32_t)t1;
    const uint64_t t2 = (t1 >> 32) + c_lo + (s_mid >> 32) + (uint32_t)s_hi;
    out[2] = (uint32_t)t2;
    const uint64_t t3 = (t2 >> 32) + c_mid + (s_hi >> 32);
    out[3] = (uint32_t)t3;
    out[4] = (uint32_t)(t3 >> 32) + c_hi;
}

}  // namespace cwc
