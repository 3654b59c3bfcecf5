// Host check of u256_divrem_digits (digit-wise division used by the Idiv/Mod bundles) against the bit-serial
// u256_divrem on random and edge operands; built three times by tests/test_host_formats.py: as is, and with the
// quotient-digit estimate forced one too high / one too low (the exact correction must absorb both).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
static uint64_t rng_state = 0x1234567;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 32); }
static Fr rand_bits(int bits) {  // uniform value with exactly `bits` significant bits (0: zero)
    Fr x = fr_zero();
    if (bits <= 0) return x;
    for (int i = 0; i < 8; ++i) x.v[i] = rnd();
    int top = bits - 1;
    for (int i = 0; i < 8; ++i) {
        if (32 * i > top) x.v[i] = 0;
        else if (32 * i + 31 >= top) { x.v[i] &= (top % 32 == 31) ? 0xffffffffu : ((1u << (top % 32 + 1)) - 1u); x.v[i] |= 1u << (top % 32); }
    }
    return x;
}
int main() {
    long bad = 0, n = 0;
    for (int iter = 0; iter < 150000; ++iter) {
        int ba = rnd() % 257, bb = 1 + rnd() % 256;
        Fr a = rand_bits(ba), b = rand_bits(bb);
        if (iter % 7 == 0) { for (int i = 0; i < 8; ++i) if (rnd() & 1) b.v[i] = (rnd() & 1) ? 0xffffffffu : 0; if (u256_is_zero(b)) b.v[0] = 1; }
        if (iter % 11 == 0) { a = b; if (rnd() & 1) a.v[0] ^= 1; }           // near-equal operands
        if (iter % 13 == 0) { for (int i = 0; i < 8; ++i) a.v[i] = 0xffffffffu; }
        if (iter % 17 == 0) { b = fr_zero(); b.v[rnd() % 8] = 1u << (rnd() % 32); } // powers of two
        if (iter % 19 == 0) { b = fr_zero(); b.v[7] = 0x80000000u; b.v[rnd() % 7] = rnd(); }
        const uint32_t la = u256_bitlen(a), lb = u256_bitlen(b);
        const uint32_t top = la >= lb ? la - lb + 1 : 0;
        Fr q0, r0, q1, r1;
        u256_divrem(q0, r0, a, b, top);
        uint32_t digits = (top + 31) / 32;
        if (iter % 3 == 0) digits = 8;           // more digits than needed must be harmless
        else if (iter % 3 == 1 && digits < 8) digits += rnd() % (9 - digits);
        u256_divrem_digits(q1, r1, a, b, digits);
        ++n;
        if (memcmp(&q0, &q1, 32) || memcmp(&r0, &r1, 32)) { if (bad < 5) printf("mismatch iter %d bits %u/%u digits %u\n", iter, la, lb, digits); ++bad; }
    }
    printf("u256_divrem_digits vs restoring division: %ld mismatches of %ld\n", bad, n);
    return bad != 0;
}
