// Microbenchmark (gfx950): what do the five direct-to-LDS staging loads of an interpreter iteration cost a lone wave
// when they are issued back to back, and when they are spread through a block of dependent VALU work?
//   A: 5 x (s_mov m0; s_nop; buffer_load_dwordx4 ... lds) then 320 v_mad_u64_u32      (the interpreter's order)
//   B: the same loads, one every 64 VALU instructions
//   C: the VALU block alone
// Per-lane gather offsets like the interpreter's (a different 32-byte slot per node slot, T = 2).
// Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2000

__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
#define VALU64(acc, y)                                                              \
    _Pragma("unroll") for (int r = 0; r < 64; ++r) {                                 \
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"((uint32_t)acc), "v"(y)); \
    }

template <int MODE>
__global__ __launch_bounds__(64) void bench(uint64_t* out, const char* buf, uint32_t bytes, uint32_t stride) {
    __shared__ uint4 lds[20480 / 16];
    const uint32_t lane = threadIdx.x;
    const uint64_t a = (uint64_t)(buf + (size_t)blockIdx.x * bytes);
    const i32x4 rsrc{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)lds;
    uint32_t voff = ((lane >> 1) * stride + (lane & 1) * 16u) % (bytes - 64u);
    uint64_t acc = lane + 1, cy;
    uint32_t y = lane * 2654435761u + 12345u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
            for (int k = 0; k < 5; ++k) dma16(lds0 + 8192 + k * 1024, voff, rsrc, k * 32);
            for (int k = 0; k < 5; ++k) { VALU64(acc, y) }
        } else if (MODE == 1) {
            for (int k = 0; k < 5; ++k) { dma16(lds0 + 8192 + k * 1024, voff, rsrc, k * 32); VALU64(acc, y) }
        } else if (MODE == 3) {  // one M0 write, the LDS placement in the instruction offset
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen offset:3072 lds\n\t"
                         "buffer_load_dwordx4 %1, %2, 0 offen offset:4080 lds" ::"s"(lds0 + 8192), "v"(voff), "s"(rsrc) : "memory");
            for (int k = 0; k < 5; ++k) { VALU64(acc, y) }
        } else if (MODE == 4) {  // plain loads into registers (no LDS, no M0)
            uint4 q[5];
            for (int k = 0; k < 5; ++k) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(q[k]) : "v"(voff), "s"(rsrc), "s"(k * 32) : "memory");
            for (int k = 0; k < 5; ++k) { VALU64(acc, y) }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int k = 0; k < 5; ++k) acc += q[k].x;
        } else {
            for (int k = 0; k < 5; ++k) { VALU64(acc, y) }
        }
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        voff = (voff + 4096u) % (bytes - 64u) & ~15u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x1234567) out[1] = acc + lds[lane].x;
}

template <int MODE>
double run(uint64_t* d_out, const char* buf, uint32_t bytes, uint32_t stride, int blocks) {
    uint64_t h[2];
    bench<MODE><<<blocks, 64>>>(d_out, buf, bytes, stride);
    hipDeviceSynchronize();
    bench<MODE><<<blocks, 64>>>(d_out, buf, bytes, stride);
    hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    return (double)h[0] / ITERS;
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 64);
    const uint32_t bytes = 4u << 20;
    const int blocks = 1024;  // one wave per SIMD
    char* buf;
    hipMalloc(&buf, (size_t)bytes * blocks);
    hipMemset(buf, 1, (size_t)bytes * blocks);
    for (uint32_t stride : {32u, 4096u, 65536u}) {
        const double a = run<0>(d_out, buf, bytes, stride, blocks), b = run<1>(d_out, buf, bytes, stride, blocks), c = run<2>(d_out, buf, bytes, stride, blocks);
        const double d = run<3>(d_out, buf, bytes, stride, blocks), e = run<4>(d_out, buf, bytes, stride, blocks);
        printf("gather stride %6u B: loads then VALU %.0f | interleaved %.0f | VALU alone %.0f ticks per iteration -> loads cost %.0f back to back, %.0f interleaved, "
               "%.0f with one M0 write, %.0f as plain register loads\n", stride, a, b, c, a - c, b - c, d - c, e - c);
    }
    return 0;
}
