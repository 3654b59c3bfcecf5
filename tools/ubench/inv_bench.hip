// Microbenchmark: cycles per fr_inv for one wave (gfx950); build twice, with and without -DCWC_CONSTANT_TIME_INVERSE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../circom-witnesscalc_amd/csrc/fr_gfx950.hpp"
using namespace cwc;
__global__ void k(uint64_t* out, int iters, int distinct) {
    Fr a = fr_r2();
    a.v[0] ^= distinct ? threadIdx.x * 2654435761u : 12345u;
    a.v[3] ^= distinct ? threadIdx.x * 40503u : 777u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a = fr_inv(a); a.v[0] ^= 1u; a.v[7] &= 0x0fffffffu; }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a.v[1] == 0x12345) out[1] = a.v[2];
}
int main() {
    uint64_t* d; hipMalloc(&d, 64); uint64_t h[2];
    for (int distinct = 0; distinct < 2; ++distinct) {
        k<<<1, 64>>>(d, 50, distinct); hipDeviceSynchronize();
        k<<<1, 64>>>(d, 200, distinct); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
#ifdef CWC_CONSTANT_TIME_INVERSE
        const char* name = "constant-time (20x30 divsteps)";
#else
        const char* name = "variable-time divsteps";
#endif
        printf("%s, %s operands across lanes: %.0f cycles per fr_inv (one wave)\n", name, distinct ? "distinct" : "identical", h[0] / 200.0);
    }
}
