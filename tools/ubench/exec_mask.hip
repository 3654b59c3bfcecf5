// Microbenchmark (gfx950): does a VALU instruction of a wave whose EXEC mask covers only the first 16 (or 32) lanes
// issue faster than a full one?  (If the SIMD skipped empty 16-lane passes, narrow bundles could be compacted.)
// Cycles via s_memtime around an unrolled dependent chain.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
#define ITERS 200

template <int KIND>
__global__ void bench(uint64_t* out, uint32_t seed, uint32_t lanes) {
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    uint64_t acc[4]; uint32_t u[4];
    for (int i = 0; i < 4; ++i) { acc[i] = x + i; u[i] = x + i * 77; }
    uint64_t t0 = 0, t1 = 0;
    if ((threadIdx.x & 63) < lanes) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int r = 0; r < REP; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (KIND == 0) acc[c] = (uint64_t)(uint32_t)acc[c] * y + acc[c];  // v_mad_u64_u32
                    if (KIND == 1) u[c] = u[c] + (y ^ u[c]);                            // plain VALU
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i] + u[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 0x1234567) out[1] = s;
}

template <int KIND>
void run(const char* name, uint64_t* d_out) {
    for (uint32_t lanes : {64u, 32u, 16u, 1u}) {
        uint64_t h[2] = {0, 0};
        bench<KIND><<<256, 256>>>(d_out, 1, lanes);
        hipDeviceSynchronize();
        bench<KIND><<<256, 256>>>(d_out, 2, lanes);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-12s active lanes %2u : %.2f memtime ticks per wave instruction\n", name, lanes, (double)h[0] / (ITERS * REP * 4));
    }
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 64);
    run<0>("mad_u64_u32", d_out);
    run<1>("valu_add_xor", d_out);
    return 0;
}
