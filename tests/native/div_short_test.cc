// Host check of u128_divrem_64 (short division of the Idiv/Mod bundles) against the bit-serial u256_divrem on random
// and edge operands; built three times by tests/test_host_formats.py like div_digits_test.cc.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x9876543;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 32); }
static Fr rand_bits(int bits, int max_limbs) {
    Fr x = fr_zero();
    if (bits <= 0) return x;
    for (int i = 0; i < max_limbs; ++i) x.v[i] = rnd();
    int top = bits - 1;
    for (int i = 0; i < 8; ++i) {
        if (32 * i > top) x.v[i] = 0;
        else if (32 * i + 31 >= top) { x.v[i] &= (top % 32 == 31) ? 0xffffffffu : ((1u << (top % 32 + 1)) - 1u); x.v[i] |= 1u << (top % 32); }
    }
    return x;
}
int main() {
    long bad = 0, n = 0;
    for (int iter = 0; iter < 300000; ++iter) {
        Fr a = rand_bits(rnd() % 129, 4), b = rand_bits(1 + rnd() % 64, 2);
        if (iter % 7 == 0) { b.v[0] = (rnd() & 1) ? 0xffffffffu : rnd(); b.v[1] = (rnd() & 1) ? 0xffffffffu : 0; if (!b.v[0] && !b.v[1]) b.v[0] = 1; }
        if (iter % 11 == 0) { a = b; a.v[2] = rnd() & 1; }
        if (iter % 13 == 0) { for (int i = 0; i < 4; ++i) a.v[i] = 0xffffffffu; }
        if (iter % 17 == 0) { b = fr_zero(); b.v[rnd() % 2] = 1u << (rnd() % 32); }
        if (iter % 19 == 0) { b = fr_zero(); b.v[1] = 0x80000000u; b.v[0] = rnd(); }
        if (iter % 23 == 0) { b = fr_zero(); b.v[0] = 1 + rnd() % 3; }
        const uint32_t la = u256_bitlen(a), lb = u256_bitlen(b);
        Fr q0, r0, q1, r1;
        u256_divrem(q0, r0, a, b, la >= lb ? la - lb + 1 : 0);
        u128_divrem_64(q1, r1, a, b);
        ++n;
        if (memcmp(&q0, &q1, 32) || memcmp(&r0, &r1, 32)) { if (bad < 5) printf("mismatch iter %d bits %u/%u\n", iter, la, lb); ++bad; }
    }
    printf("u128_divrem_64 vs restoring division: %ld mismatches of %ld\n", bad, n);
    return bad != 0;
}
