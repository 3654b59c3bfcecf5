// Host check of the division by a two-word divisor (recip64_3by2 / div3by2 / u256_divrem_128: the quotient-digit estimates of
// multi-register long division on registers wider than a word) against the bit-serial u256_divrem on random and edge operands;
// built by tests/test_host_formats.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x2468013579ull;
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
static Fr from64(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3) {
    Fr r;
    const uint64_t w[4] = {w0, w1, w2, w3};
    for (int k = 0; k < 4; ++k) { r.v[2 * k] = (uint32_t)w[k]; r.v[2 * k + 1] = (uint32_t)(w[k] >> 32); }
    return r;
}
int main() {
    long bad = 0, n = 0;
    for (int iter = 0; iter < 400000; ++iter) {
        uint64_t bh = pick(), bl = pick();
        if (!bh) bh = 1 + rnd64() % 5;                      // 2^64 <= b < 2^128
        if (iter % 9 == 0) { bh = 1; bl = 0; }              // b = 2^64
        if (iter % 11 == 0) { bh = ~0ull; bl = ~0ull; }     // b = 2^128 - 1
        if (iter % 13 == 0) bh = 1ull << 56;                // a 121-bit register's range
        Fr a = from64(pick(), pick(), pick(), pick());
        if (iter % 4 == 0) a = from64(pick(), pick(), pick(), rnd64() >> 14);   // below 2^242: a product of two 121-bit registers
        if (iter % 5 == 0) a = from64(pick(), pick(), 0, 0);
        if (iter % 7 == 0) a = from64(bl, bh, 0, 0);                           // a == b
        if (iter % 17 == 0) a = from64(bl - 1, bh - (bl == 0), 0, 0);          // a == b - 1
        if (iter % 19 == 0) a = from64(~0ull, ~0ull, ~0ull, ~0ull);
        const Fr b = from64(bl, bh, 0, 0);
        Fr q, r, q2, r2;
        u256_divrem_128(q, r, a, b);
        u256_divrem(q2, r2, a, b, 256);
        ++n;
        if (!u256_eq(q, q2) || !u256_eq(r, r2)) {
            if (bad < 5) printf("mismatch iter %d\n", iter);
            ++bad;
        }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
