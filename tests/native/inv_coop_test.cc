// Host emulation of the lane-cooperative field inversion (fr_gfx950.hpp u256_inv_mod_r_coop16: sixteen lanes per inversion, lane l
// holds limb l of f, g, d, e; the matrix update of a batch as one product per lane and a TWO-PASS LAZY carry between neighbour lanes)
// against the one-lane inversion and Fermat: the same statements as the device function with the lanes as array indices, the DPP
// row shifts as index shifts and the broadcasts as reads of lane 0 / lane 8.  Checks the arithmetic the device code relies on: the
// limbs stay within 32 bits, the low limb is exact, the exact zero test of g from batch 14 on terminates where the one-lane loop does.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x97531;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 32); }
static Fr rand_fr() {
    for (;;) {
        Fr x;
        for (int i = 0; i < 8; ++i) x.v[i] = rnd();
        x.v[7] &= 0x3fffffffu;
        if (u256_lt(x, fr_p())) return x;
    }
}
static long g_max_batches = 0, g_range_bad = 0;
static Fr inv_coop_emulated(const Fr& x) {
    const int32_t M30 = 0x3fffffff;
    const int32_t p30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
    const uint32_t pinv30 = 0x10000001u;
    int32_t d[16], e[16], f[16], g[16], pl[16];
    for (int l = 0; l < 16; ++l) {
        pl[l] = l < 9 ? p30[l] : 0;
        d[l] = 0;
        e[l] = l == 0 ? 1 : 0;
        f[l] = pl[l];
        g[l] = l < 9 ? (int32_t)(u256_shr(x, 30u * (uint32_t)l).v[0] & (uint32_t)M30) : 0;
    }
    int32_t eta = -1;
    int it = 0;
    for (; it < 25; ++it) {
        Trans2x2 t;
        eta = sgcd_divsteps_30_var(eta, (uint32_t)f[0], (uint32_t)g[0], t);   // (f[0], g[0]: the broadcast of lane 0)
        const int32_t sd = d[8] >> 31, se = e[8] >> 31;                          // (lane 8)
        int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
        const uint32_t cd0 = (uint32_t)t.u * (uint32_t)d[0] + (uint32_t)t.v * (uint32_t)e[0];
        const uint32_t ce0 = (uint32_t)t.q * (uint32_t)d[0] + (uint32_t)t.r * (uint32_t)e[0];
        md -= (int32_t)((pinv30 * cd0 + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((pinv30 * ce0 + (uint32_t)me) & (uint32_t)M30);
        int64_t c[4][16];
        for (int l = 0; l < 16; ++l) {
            c[0][l] = (int64_t)t.u * f[l] + (int64_t)t.v * g[l];
            c[1][l] = (int64_t)t.q * f[l] + (int64_t)t.r * g[l];
            c[2][l] = (int64_t)t.u * d[l] + (int64_t)t.v * e[l] + (int64_t)pl[l] * md;
            c[3][l] = (int64_t)t.q * d[l] + (int64_t)t.r * e[l] + (int64_t)pl[l] * me;
        }
        int32_t* dst[4] = {f, g, d, e};
        for (int k = 0; k < 4; ++k) {
            if ((c[k][0] & M30) != 0) ++g_range_bad;  // the low 30 bits of the sum vanish by construction
            int32_t lo[16], nlo[16], nhi[16];
            int64_t nw[16];
            for (int l = 0; l < 16; ++l) lo[l] = (int32_t)((uint32_t)c[k][l] & (uint32_t)M30);
            for (int l = 0; l < 16; ++l) nw[l] = (c[k][l] >> 30) + (l + 1 < 16 ? lo[l + 1] : 0);   // row_shl:1
            for (int l = 0; l < 16; ++l) {
                nlo[l] = l == 8 ? 0 : (int32_t)((uint32_t)nw[l] & (uint32_t)M30);
                nhi[l] = l == 8 ? 0 : (int32_t)(nw[l] >> 30);
                if (nw[l] > (1ll << 40) || nw[l] < -(1ll << 40)) ++g_range_bad;
            }
            for (int l = 0; l < 16; ++l) {
                const int64_t v = (l == 8 ? nw[l] : (int64_t)nlo[l]) + (l > 0 ? nhi[l - 1] : 0);     // row_shr:1
                if (v > 0x7fffffffll || v < -0x80000000ll) ++g_range_bad;
                if (l > 8 && v != 0) ++g_range_bad;
                dst[k][l] = (int32_t)v;
            }
        }
        if (it >= 13) {  // exact zero test of g: gather, carry from the bottom
            int64_t carry = 0;
            bool zero = true;
            for (int l = 0; l < 9; ++l) {
                const int64_t v = g[l] + carry;
                if (l < 8) { zero = zero && (v & M30) == 0; carry = v >> 30; } else zero = zero && v == 0;
            }
            if (zero) { ++it; break; }
        }
    }
    if (it > g_max_batches) g_max_batches = it;
    // gather and finish as the one-lane function does, on exactly normalised limbs
    S30 dd, ff;
    {
        int64_t cd = 0, cf = 0;
        for (int l = 0; l < 9; ++l) {
            const int64_t vd = d[l] + cd, vf = f[l] + cf;
            if (l < 8) { dd.v[l] = (int32_t)(vd & M30); cd = vd >> 30; ff.v[l] = (int32_t)(vf & M30); cf = vf >> 30; }
            else { dd.v[l] = (int32_t)vd; ff.v[l] = (int32_t)vf; }
        }
    }
    const int32_t sign = ff.v[8] >> 31;
    int32_t cond_add = dd.v[8] >> 31;
    for (int i = 0; i < 9; ++i) dd.v[i] += p30[i] & cond_add;
    for (int i = 0; i < 9; ++i) dd.v[i] = (dd.v[i] ^ sign) - sign;
    for (int i = 0; i < 8; ++i) { dd.v[i + 1] += dd.v[i] >> 30; dd.v[i] &= M30; }
    cond_add = dd.v[8] >> 31;
    for (int i = 0; i < 9; ++i) dd.v[i] += p30[i] & cond_add;
    for (int i = 0; i < 8; ++i) { dd.v[i + 1] += dd.v[i] >> 30; dd.v[i] &= M30; }
    Fr out;
    for (int k = 0; k < 8; ++k) {
        const int li = (32 * k) / 30, off = (32 * k) % 30;
        uint64_t w = (uint64_t)(uint32_t)dd.v[li] >> off;
        w |= (uint64_t)(uint32_t)dd.v[li + 1] << (30 - off);
        if (li + 2 < 9) w |= (uint64_t)(uint32_t)dd.v[li + 2] << (60 - off);
        out.v[k] = (uint32_t)w;
    }
    return out;
}
int main() {
    long bad = 0, n = 0;
    auto check = [&](const Fr& x, const char* what) {
        const Fr a = inv_coop_emulated(x), b = u256_inv_mod_r(x);
        ++n;
        if (memcmp(&a, &b, 32)) { if (bad < 5) printf("mismatch (%s): x.v[0] = %08x\n", what, x.v[0]); ++bad; }
    };
    check(fr_zero(), "zero");
    check(Fr{{1, 0, 0, 0, 0, 0, 0, 0}}, "one");
    Fr pm1;
    u256_sub(pm1, fr_p(), Fr{{1, 0, 0, 0, 0, 0, 0, 0}});
    check(pm1, "r - 1");
    for (int k = 0; k < 254; ++k) {
        Fr x = fr_zero();
        x.v[k / 32] = 1u << (k % 32);
        if (u256_lt(x, fr_p())) check(x, "2^k");
        Fr y;
        u256_sub(y, x, Fr{{1, 0, 0, 0, 0, 0, 0, 0}});
        if (k > 0 && u256_lt(y, fr_p())) check(y, "2^k - 1");
        Fr z;
        u256_sub(z, fr_p(), x);
        if (u256_lt(z, fr_p())) check(z, "r - 2^k");
    }
    for (int iter = 0; iter < 20000; ++iter) {
        Fr x = rand_fr();
        if (iter % 5 == 0) { for (int i = 1 + rnd() % 7; i < 8; ++i) x.v[i] = 0; }
        if (iter % 7 == 0) { for (int i = 0; i < (int)(rnd() % 7); ++i) x.v[i] = 0; if (!u256_lt(x, fr_p())) x = rand_fr(); }
        check(x, "random");
    }
    printf("lane-cooperative inversion (emulated) vs one-lane: %ld mismatches of %ld; %ld range violations; at most %ld batches\n", bad, n, g_range_bad, g_max_batches);
    return bad != 0 || g_range_bad != 0;
}
