// Microbenchmark: cycles per inversion on one wave (gfx950) -- the one-lane fr_inv with four active lanes (a divider wave's request
// of the division ladder) against the lane-cooperative fr_inv_coop16 (four rows of sixteen lanes, one inversion each), both on a
// dependent chain; and the results of the two compared lane by lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
__global__ void k(uint64_t* out, int iters, int coop) {
    const uint32_t row = threadIdx.x >> 4;
    Fr a = fr_r2();
    a.v[0] ^= (coop ? row : threadIdx.x) * 2654435761u + 12345u;
    a.v[3] ^= (coop ? row : threadIdx.x) * 40503u;
    if (!coop && threadIdx.x >= 4) a = fr_zero();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a = coop ? fr_inv_coop16(a) : fr_inv(a);
        if (coop || threadIdx.x < 4) { a.v[0] ^= 1u; a.v[7] &= 0x0fffffffu; }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a.v[1] == 0x12345) out[1] = a.v[2];
    // parity: the cooperative inversion of this row's operand == the one-lane inversion, and x * inv(x) == 1
    Fr x = fr_r2();
    x.v[0] ^= row * 2654435761u;
    x.v[2] ^= (uint32_t)iters * 97u + row;
    if (row == 3) { x = fr_zero(); x.v[0] = 5u; }           // a short operand
    const Fr ic = fr_inv_coop16(x), i1 = fr_inv(x), p = fr_mul(x, ic), one = fr_one();
    uint32_t bad = 0;
    for (int i = 0; i < 8; ++i) bad |= (p.v[i] ^ one.v[i]) | (ic.v[i] ^ i1.v[i]);
    if (bad) atomicAdd((unsigned long long*)&out[2], 1ull);
    const Fr z = fr_inv_coop16(fr_zero());
    if (!u256_is_zero(z)) atomicAdd((unsigned long long*)&out[2], 1ull);
}
int main() {
    uint64_t* d; hipMalloc(&d, 64); uint64_t h[3];
    const char* names[2] = {"one-lane fr_inv, four active lanes", "fr_inv_coop16, four rows of sixteen lanes"};
    for (int coop = 0; coop < 2; ++coop) {
        hipMemset(d, 0, 64);
        k<<<1, 64>>>(d, 50, coop); hipDeviceSynchronize();
        hipMemset(d, 0, 64);
        k<<<1, 64>>>(d, 200, coop); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-44s %7.0f cycles per inversion (one wave)  mismatching lanes %llu\n", names[coop], h[0] / 200.0, (unsigned long long)h[2]);
    }
}
