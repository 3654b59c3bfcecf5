// Host check of the division by an invariant limb (recip64 / div2by1 / u128_divrem_64_recip: the long-division rounds of the
// scan bundles) against unsigned __int128 arithmetic on random and edge operands; built by tests/test_host_formats.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x1357924680ull;
static uint64_t rnd64() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t a = rng_state >> 32;
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (a << 32) | (rng_state >> 32);
}
static uint64_t pick() {
    switch (rnd64() % 8) {
        case 0: return 0;
        case 1: return ~0ull;
        case 2: return 1ull << (rnd64() % 64);
        case 3: return (1ull << (rnd64() % 64)) - 1;
        case 4: return rnd64() >> (rnd64() % 64);
        default: return rnd64();
    }
}
int main() {
    long bad = 0, n = 0;
    for (int iter = 0; iter < 2000000; ++iter) {
        uint64_t d = pick();
        if (!d) d = 1 + rnd64() % 3;
        uint64_t th = pick(), tl = pick();
        if (iter % 3 == 0) th %= d;            // the loop's steady state: the high word is a remainder
        if (iter % 5 == 0) th = d - 1;
        if (iter % 7 == 0) { th = d; tl = 0; }
        if (iter % 11 == 0) { th = d - 1; tl = ~0ull; }
        const uint32_t s = clz64_nonzero(d);
        const uint64_t dn = d << s, v = recip64(dn);
        const unsigned __int128 vv = (((unsigned __int128)(~dn)) << 64 | ~0ull) / dn;
        if ((uint64_t)vv != v || (vv >> 64) != 0) { if (bad < 5) printf("reciprocal mismatch d=%llx\n", (unsigned long long)d); ++bad; }
        const unsigned __int128 t = ((unsigned __int128)th << 64) | tl;
        const unsigned __int128 q = t / d;
        const uint64_t r = (uint64_t)(t % d);
        for (int high = (th >= d) ? 1 : 0; high < 2; ++high) {
            uint64_t qh, ql, rem;
            u128_divrem_64_recip(th, tl, d, s, dn, v, high != 0, qh, ql, rem);
            ++n;
            if (qh != (uint64_t)(q >> 64) || ql != (uint64_t)q || rem != r) {
                if (bad < 5) printf("mismatch th=%llx tl=%llx d=%llx high=%d\n", (unsigned long long)th, (unsigned long long)tl, (unsigned long long)d, high);
                ++bad;
            }
        }
    }
    // the reciprocal alone on structured divisors: both ends of the normalised range, powers of two +- small, all-ones prefixes
    for (int iter = 0; iter < 6000000; ++iter) {
        uint64_t dn = rnd64() | (1ull << 63);
        if (iter % 7 == 0) dn = (1ull << 63) + rnd64() % 1000;
        if (iter % 11 == 0) dn = ~0ull - rnd64() % 1000;
        if (iter % 13 == 0) dn = (1ull << 63) | (1ull << (rnd64() % 63));
        if (iter % 17 == 0) dn = (1ull << 63) | ((1ull << (rnd64() % 63)) - 1);
        const unsigned __int128 vv = (((unsigned __int128)(~dn)) << 64 | ~0ull) / dn;
        ++n;
        if ((uint64_t)vv != recip64(dn)) { if (bad < 5) printf("reciprocal mismatch dn=%llx\n", (unsigned long long)dn); ++bad; }
    }
    printf("u128_divrem_64_recip vs __int128: %ld mismatches of %ld\n", bad, n);
    return bad != 0;
}
