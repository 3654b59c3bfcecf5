// Host check of the field inversion of the Div bundles (fr_inv / u256_inv_mod_r: safegcd divsteps, reference
// src/graph.rs:109 `a / b`): x * inv(x) == 1 and inv(x) == x^(r-2) (Fermat) on random and edge operands, inv(0) == 0;
// built by tests/test_host_formats.py with the variable-time and the constant-time inner loop.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x13572468;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 32); }
static Fr rand_fr() {
    for (;;) {
        Fr x;
        for (int i = 0; i < 8; ++i) x.v[i] = rnd();
        x.v[7] &= 0x3fffffffu;
        if (u256_lt(x, fr_p())) return x;
    }
}
int main() {
    long bad = 0, n = 0;
    const Fr one = fr_one();  // Montgomery form of 1
    auto check = [&](const Fr& x, const char* what) {
        const Fr inv = fr_inv(x);  // Montgomery in, Montgomery out
        ++n;
        if (u256_is_zero(x)) {
            if (!u256_is_zero(inv)) { if (bad < 5) printf("inv(0) != 0 (%s)\n", what); ++bad; }
            return;
        }
        const Fr prod = fr_mul(x, inv);
        const Fr fermat = fr_inv_fermat(x);
        if (memcmp(&prod, &one, 32) || memcmp(&inv, &fermat, 32) || !u256_lt(inv, fr_p())) {
            if (bad < 5) printf("mismatch (%s): x.v[0] = %08x\n", what, x.v[0]);
            ++bad;
        }
    };
    check(fr_zero(), "zero");
    check(one, "one");
    Fr pm1;
    u256_sub(pm1, fr_p(), Fr{{1, 0, 0, 0, 0, 0, 0, 0}});
    check(pm1, "r - 1 as a Montgomery pattern");
    for (int k = 0; k < 254; ++k) {  // single bits and bit - 1 patterns below r
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
    for (int iter = 0; iter < 3000; ++iter) {
        Fr x = rand_fr();
        if (iter % 5 == 0) { for (int i = 1 + rnd() % 7; i < 8; ++i) x.v[i] = 0; }  // short values
        if (iter % 7 == 0) { for (int i = 0; i < (int)(rnd() % 7); ++i) x.v[i] = 0; if (!u256_lt(x, fr_p())) x = rand_fr(); }  // many trailing zeros
        check(x, "random");
    }
    printf("fr_inv vs Fermat and x * inv(x) == 1: %ld mismatches of %ld\n", bad, n);
    return bad != 0;
}
