// Microbenchmark + parity check (gfx950) of the lane-cooperative Montgomery multipliers (groups of K = 2, 4, 8 lanes,
// tools/codegen/gen_fr_mul_coop.py) against the one-lane fr_mul: results for random and edge operands, and the lone-wave
// cost of a dependent chain of products (s_memtime around the loop; one wave per workgroup).
// Build: hipcc -O3 --offload-arch=gfx950 -o coop_mul coop_mul.hip ; run: ./coop_mul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"

using namespace cwc;

__device__ __forceinline__ void mul_coop8(const Fr& a, uint32_t b0, uint32_t n0, uint32_t* out) {
#include "../../circom-witnesscalc_amd/csrc/fr_mul_coop8_gfx950.inc"
}
__device__ __forceinline__ void mul_coop4(const Fr& a, uint32_t b0, uint32_t b1, uint32_t n0, uint32_t n1, uint32_t* out) {
#include "../../circom-witnesscalc_amd/csrc/fr_mul_coop4_gfx950.inc"
}
__device__ __forceinline__ void mul_coop2(const Fr& a, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t n0, uint32_t n1, uint32_t n2,
                                          uint32_t n3, uint32_t* out) {
#include "../../circom-witnesscalc_amd/csrc/fr_mul_coop2_gfx950.inc"
}

// A[g], B[g]: operands of group g (8 limbs each); out[g]: product limbs; iters > 1: b <- a * b repeatedly (dependent chain)
template <int K>
__global__ void kern(const uint32_t* A, const uint32_t* B, uint32_t* out, int iters, unsigned long long* cycles) {
    constexpr int L = 8 / K;
    const uint32_t lane = threadIdx.x & 63u, g = (blockIdx.x * 64u + lane) / K, k = lane % K;
    const uint32_t p[8] = {CWC_P0, CWC_P1, CWC_P2, CWC_P3, CWC_P4, CWC_P5, CWC_P6, CWC_P7};
    Fr a;
    for (int i = 0; i < 8; ++i) a.v[i] = A[g * 8 + i];
    uint32_t b[L], n[L], r[L];
    for (int j = 0; j < L; ++j) {
        b[j] = B[g * 8 + k * L + j];
        n[j] = 0;
        for (int q = 0; q < 8; ++q) n[j] = (k * L + j == (uint32_t)q) ? p[q] : n[j];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 8) mul_coop8(a, b[0], n[0], r);
        if constexpr (K == 4) mul_coop4(a, b[0], b[1], n[0], n[1], r);
        if constexpr (K == 2) mul_coop2(a, b[0], b[1], b[2], b[3], n[0], n[1], n[2], n[3], r);
        for (int j = 0; j < L; ++j) b[j] = r[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < L; ++j) out[g * 8 + k * L + j] = b[j];
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ void kern1(const uint32_t* A, const uint32_t* B, uint32_t* out, int iters, unsigned long long* cycles) {
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    Fr a, b;
    for (int i = 0; i < 8; ++i) { a.v[i] = A[g * 8 + i]; b.v[i] = B[g * 8 + i]; }
    Fr pv = fr_p();
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(pv.v[i]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) b = fr_mul_wave(a, b, pv);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) out[g * 8 + i] = b.v[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd32() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
static Fr rnd_fr() {
    Fr x;
    for (int i = 0; i < 8; ++i) x.v[i] = rnd32();
    x.v[7] &= 0x1fffffffu;  // < 2^253 < r
    const int kind = rnd32() % 16;
    const Fr p = fr_p();
    if (kind == 0) x = fr_zero();
    if (kind == 1) { x = p; x.v[0] -= 1; }                       // r - 1
    if (kind == 2) { x = fr_zero(); x.v[0] = 1; }
    if (kind == 3) { for (int i = 0; i < 7; ++i) x.v[i] = 0xffffffffu; x.v[7] = 0x0fffffffu; }
    if (kind == 4) { x = p; x.v[0] -= 2; }
    if (kind == 5) { for (int i = 0; i < 8; ++i) x.v[i] = (i & 1) ? 0xffffffffu : 0u; x.v[7] = 0x1fffffffu; }
    return x;
}

template <int K>
static int run(const char* name, int n_groups_total) {
    const int groups_per_block = 64 / K, blocks = n_groups_total / groups_per_block;
    std::vector<uint32_t> A(n_groups_total * 8), B(n_groups_total * 8), O(n_groups_total * 8);
    for (int g = 0; g < n_groups_total; ++g) {
        const Fr a = rnd_fr(), b = rnd_fr();
        for (int i = 0; i < 8; ++i) { A[g * 8 + i] = a.v[i]; B[g * 8 + i] = b.v[i]; }
    }
    uint32_t *dA, *dB, *dO;
    unsigned long long* dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, O.size() * 4); hipMalloc(&dC, 64);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int iters : {1, 3}) {
        if constexpr (K == 1) kern1<<<n_groups_total / 64, 64>>>(dA, dB, dO, iters, dC);
        else kern<K><<<blocks, 64>>>(dA, dB, dO, iters, dC);
        hipDeviceSynchronize();
        hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
        for (int g = 0; g < n_groups_total; ++g) {
            Fr a, b;
            for (int i = 0; i < 8; ++i) { a.v[i] = A[g * 8 + i]; b.v[i] = B[g * 8 + i]; }
            for (int it = 0; it < iters; ++it) b = fr_mul(a, b);
            for (int i = 0; i < 8; ++i)
                if (b.v[i] != O[g * 8 + i]) {
                    if (bad < 5) printf("  %s MISMATCH group %d limb %d iters %d: got %08x want %08x\n", name, g, i, iters, O[g * 8 + i], b.v[i]);
                    ++bad;
                    break;
                }
        }
    }
    // timing: one workgroup of one wave (a lone wave on its SIMD), then one wave on every SIMD
    unsigned long long cyc[2] = {0, 0};
    const int IT = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        const int nb = rep == 0 ? 1 : 1024;
        for (int w = 0; w < 2; ++w) {
            if constexpr (K == 1) kern1<<<nb, 64>>>(dA, dB, dO, IT, dC);
            else kern<K><<<nb, 64>>>(dA, dB, dO, IT, dC);
            hipDeviceSynchronize();
        }
        hipMemcpy(&cyc[rep], dC, 8, hipMemcpyDeviceToHost);
    }
    printf("%-6s parity %s (%d groups x {1,3} products); dependent chain: %.0f cycles per product (1 wave), %.0f (1024 single-wave workgroups)\n", name,
           bad ? "FAILED" : "ok", n_groups_total, (double)cyc[0] / IT, (double)cyc[1] / IT);
    hipFree(dA); hipFree(dB); hipFree(dO); hipFree(dC);
    return bad;
}

int main() {
    int bad = 0;
    // the timing launches read groups beyond the parity set when nb = 1024: size the operand arrays for them
    bad += run<1>("K=1", 1024 * 64);
    bad += run<8>("K=8", 1024 * 8);
    bad += run<4>("K=4", 1024 * 16);
    bad += run<2>("K=2", 1024 * 32);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}
