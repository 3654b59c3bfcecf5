// Microbenchmark (gfx950): issue cost of the integer / fp64 multiply instructions a 256-bit modular
// multiplier can be built from, at 1, 2 and 4 waves per SIMD.  Cycles via s_memtime around an
// unrolled loop; every CU runs the same thing.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"

#define REP 64
#define ITERS 200

template <int KIND, int CHAINS>
__global__ void bench(uint64_t* out, uint32_t seed) {
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    uint64_t acc[8]; double d[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { acc[i] = x + i; d[i] = 1.0 + x * 1e-9 + i; u[i] = x + i * 77; }
    double dx = 1.0000001, dy = 0.9999999;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) acc[c] = (uint64_t)(uint32_t)acc[c] * y + acc[c];           // v_mad_u64_u32
                if (KIND == 1) u[c] = u[c] * y + 1u;                                         // v_mul_lo_u32 (+add)
                if (KIND == 2) u[c] = __umulhi(u[c], y) + u[c];                              // v_mul_hi_u32 (+add)
                if (KIND == 3) d[c] = __builtin_fma(d[c], dx, dy);                           // v_fma_f64
                if (KIND == 4) u[c] = ((u[c] & 0xffffffu) * (y & 0xffffffu)) + 3u;                // v_mad_u32_u24
                if (KIND == 5) u[c] = u[c] + (y ^ u[c]);                                      // plain VALU
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0; double ds = 0;
    for (int i = 0; i < 8; ++i) { s += acc[i] + u[i]; ds += d[i]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 0x1234567 && ds == 1.5) out[1] = s;
}

__global__ void bench_frmul(uint64_t* out, uint32_t seed, int iters) {
    using namespace cwc;
    Fr a = fr_r2(), b = fr_one();
    a.v[0] ^= threadIdx.x + seed;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) { a = fr_mul(a, b); b = fr_mul(b, a); }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a.v[0] == 0x1234567 && b.v[3] == 7) out[1] = a.v[1];
}

template <int KIND, int CHAINS>
void run(const char* name, uint64_t* d_out) {
    for (int wps : {1, 2, 4}) {
        uint64_t h[2] = {0, 0};
        bench<KIND, CHAINS><<<256, 256 * wps>>>(d_out, 1);
        hipDeviceSynchronize();
        bench<KIND, CHAINS><<<256, 256 * wps>>>(d_out, 2);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        double per = (double)h[0] / (ITERS * REP * CHAINS);
        printf("%-14s chains=%d waves/SIMD=%d : %.2f memtime-ticks per wave-instruction (x wps = %.2f per SIMD-instr)\n", name, CHAINS, wps, per, per / wps);
    }
}

int main() {
    uint64_t* d_out; hipMalloc(&d_out, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s clock %d kHz, memtime ticks: s_memtime counts at a fixed 100 MHz? check vs wall below\n", p.name, p.clockRate);
    run<0, 1>("mad_u64_u32", d_out); run<0, 4>("mad_u64_u32", d_out); run<0, 8>("mad_u64_u32", d_out);
    run<1, 1>("mul_lo_u32", d_out); run<1, 8>("mul_lo_u32", d_out);
    run<2, 8>("mul_hi_u32", d_out);
    run<3, 1>("fma_f64", d_out); run<3, 4>("fma_f64", d_out); run<3, 8>("fma_f64", d_out);
    run<4, 8>("mul_u24", d_out);
    run<5, 1>("valu_add_xor", d_out); run<5, 8>("valu_add_xor", d_out);
    for (int wps : {1, 2, 4}) {
        uint64_t h[2];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        bench_frmul<<<256, 256 * wps>>>(d_out, 1, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        bench_frmul<<<256, 256 * wps>>>(d_out, 2, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        double muls = 4000.0;
        printf("fr_mul waves/SIMD=%d : %.0f ticks per fr_mul per wave; wall %.3f ms -> %.3g modmul/s chip-wide (%.1f ns per wave-mul)\n", wps,
               h[0] / muls, ms, muls * 256.0 * 256 * wps / (ms * 1e-3), ms * 1e6 / muls);
    }
    return 0;
}
