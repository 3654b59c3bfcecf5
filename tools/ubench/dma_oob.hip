// Micro test (gfx950): what a direct-to-LDS buffer load does for lanes beyond the descriptor's num_records, and whether the
// scalar offset takes part in the range check.  The variable-length record stream would lean on it: lanes behind a
// bundle's last record must land zeros in LDS.
//   case A: num_records = 5 * 16, soffset = 0,      voffset = 16 * lane   -> lanes 0..4 data, the rest ?
//   case B: num_records = 5 * 16, soffset = 16 * 7, voffset = 16 * lane   -> is the check on voffset alone?
//   case C: num_records = (7 + 5) * 16, soffset = 16 * 7                  -> ... or on soffset + voffset?
// LDS is pre-filled with 0xAAAAAAAA so that "not written" and "written with zero" can be told apart.
// Build: hipcc -O3 --offload-arch=gfx950 -o dma_oob dma_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__global__ __launch_bounds__(64) void k(uint32_t* out, const uint32_t* buf, uint32_t num_records, uint32_t soff) {
    __shared__ uint4 lds[64];
    const uint32_t lane = threadIdx.x;
    lds[lane] = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    __syncthreads();
    const uint64_t a = (uint64_t)buf;
    const i32x4 rsrc{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)num_records, 0x00020000};
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)lds;
    dma16(lds0, 16u * lane, rsrc, soff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = lds[lane].x;
}
int main() {
    uint32_t *d_buf, *d_out, h[64], hb[64 * 4 * 2];
    for (int i = 0; i < 64 * 4 * 2; ++i) hb[i] = 0x1000u + (uint32_t)i / 4;  // record r holds 0x1000 + r
    hipMalloc(&d_buf, sizeof hb); hipMalloc(&d_out, 256);
    hipMemcpy(d_buf, hb, sizeof hb, hipMemcpyHostToDevice);
    struct { const char* name; uint32_t nr, so; } cases[] = {{"A num_records=5*16 soffset=0", 80, 0}, {"B num_records=5*16 soffset=7*16", 80, 112}, {"C num_records=12*16 soffset=7*16", 192, 112}};
    for (auto& c : cases) {
        k<<<1, 64>>>(d_out, d_buf, c.nr, c.so);
        hipMemcpy(h, d_out, 256, hipMemcpyDeviceToHost);
        printf("%s:", c.name);
        for (int l = 0; l < 16; ++l) printf(" %x", h[l]);
        printf(" ... lane 63 %x\n", h[63]);
    }
    return 0;
}
