// HIP kernels of the calc-witness hot path for gfx950 (CDNA4, wave64).
//
//   interp_kernel<T>  -- the graph interpreter: the loop of graph::evaluate (reference
//                        src/graph.rs:372-382) with Operation::eval_fr (:102-144),
//                        UnoOperation::eval_fr (:188-197), TresOperation::eval_fr (:221-225).
//   pack_kernel       -- output gather: out[i] = into_bigint(values[outputs[i]]) as 32-byte LE rows
//                        (src/graph.rs:385-388, src/lib.rs:170-173), i.e. the `.wtns` section-2 body.
//
//   fill_consts_kernel -- writes each tile's copy of the constant table (once per workspace).
//
// One wavefront = one tile of T input sets x G = 64/T node slots.  The bundle class is wave-uniform
// (scalar branch); per-lane sub-ops inside a class are resolved with selects.  Values live in HBM as
// [tile][slot][half][T][16 B]: every 16-byte access of a lane group touches T*16 contiguous bytes (1 KiB per
// wave-instruction at T = 64).  Operands reach the arithmetic through LDS only: memory operands by direct-to-LDS
// buffer loads two bundles ahead, recent results through a ring (program_dev.h, format v4).  No MFMA: this is
// 256-bit modular integer arithmetic on v_mad_u64_u32.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "fr_gfx950.hpp"
#include "scan_gfx950.hpp"
#include "program_dev.h"

namespace cwc {

__device__ __forceinline__ Fr fr_from_u4(const uint4& lo, const uint4& hi) {
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}

// wave-wide OR-reduction of a predicate ("does any lane need the slow path")
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// PROF = true is a diagnostic build (gwb_profile_classes): one s_memtime per bundle, summed per bundle class by
// lane 0 of every 64th tile (prof[class*4 + {0: cycles, 3: bundles}]), and for MUL and LIN bundles five sections of
// the iteration (prof[48 + 8*{MUL, LIN} + section]).  No stamp executes in the product kernel.
//
// The program arrays are separate `const __restrict__` kernel arguments (not a by-value struct) so that hipcc can
// prove them read-only: the wave-uniform header stream then becomes scalar loads (s_load).
struct InterpDims {
    uint32_t n_bundles, n_slots, n_inputs, batch, n_const, n_div_requests, trash_off;
    const uint32_t* div_lanes;  // active lanes of each division request (divider programs)
    // streams: interpreter wave w of a workgroup evaluates bundles [stream_first[s], stream_first[s] + stream_count[s]),
    // s = w % n_streams, of tile w / n_streams (program.hpp); with divider waves, divider d serves interpreter wave d
    uint32_t n_streams, stream_first[MAX_STREAMS], stream_count[MAX_STREAMS], stream_div_requests[MAX_STREAMS], stream_cref_first[MAX_STREAMS];
};
static const uint32_t ST_DIVIDER_TIMEOUT = 0x80000000u;  // internal: a mailbox wait gave up (never expected)
static const uint32_t ST_SYNC_TIMEOUT = 0x40000000u;     // internal: the wait for stream 0's post gave up (never expected)

// Mailbox wait of the asynchronous divider protocol: sleeps until the sequence word reaches `need`.  Bounded, so that
// a protocol bug ends the kernel with an error status instead of hanging the device.
__device__ __forceinline__ bool mbox_wait(const volatile uint32_t* seq, uint32_t need) {
    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
        if (*seq >= need) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
// Wait for another stream's post (a word in the tile's sync slot, read past the CU's vector cache).  Bounded like mbox_wait.
__device__ __noinline__ bool post_wait(uint32_t* word, uint32_t need) {
#pragma nounroll
    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
        if (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

// Direct-to-LDS load: 16 bytes per lane from buffer[voff + soff] to LDS[lds_addr + lane * 16].  No VGPR destination,
// so nothing waits on it; completion is counted by hand (s_waitcnt vmcnt) in the interpreter loop.  M0 (the LDS base
// of the load) is written inside the same statement.
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const i32x4& rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// W = interpreter waves that share one divider wave (0: no divider waves).  PACK = such units per workgroup: PACK
// interpreter waves (W = 0), or PACK x (W interpreters + their divider) with the interpreters first.  The waves of a
// workgroup are dealt round the CU's four SIMDs, so four-wave workgroups put one wave on every SIMD where single-wave
// (or two-wave) workgroups land unevenly: 1024 tiles without dividers take 23.2 ms as 256 x 4 waves, 31.4 ms as 1024 x 1.
// MODE 1: the program has fused narrow bundles (class C_MULF).  Their path is compiled into instances of their own: inside the
// one interpreter loop it cost every other program 5 % (register allocation and layout of the hot paths; same-box A/B on
// the authV2-class graph, 1024 sets: 12.55 against 11.93 ms, profiles/r03_regress_ab.txt).
template <int T, bool PROF, int W, int PACK, int MODE = 0>
__global__ __launch_bounds__((W > 0 ? (W + 1) * PACK : PACK) * 64) void interp_kernel(const uint32_t* __restrict__ hdr, const uint4* __restrict__ recs,
                                                    const uint32_t* __restrict__ crefs, InterpDims p, WsTable wst,
                                                    const uint4* __restrict__ inputs, uint32_t* __restrict__ status,
                                                    unsigned long long* __restrict__ prof) {
    constexpr int G = 64 / T;
    // MODE 2: scan / convolution / canonical-product bundles.  MODE 3 (round 5): the same plus the kinds for registers wider than a machine
    // word -- borrow chains, comparisons, parallel carry chains of any width, 128-bit canonical products -- in instances of their own.
    constexpr bool M2 = MODE == 2 || MODE == 3, WIDE = MODE == 3;
    constexpr uint32_t HI = 16u * T;  // byte distance between the two 16-byte halves of a value in a slot
    constexpr bool DIVIDER = W > 0;
    constexpr uint32_t NW = (W > 0 ? (uint32_t)W : 1u) * (uint32_t)PACK;  // interpreter waves per workgroup
    constexpr uint32_t WD = W > 0 ? (uint32_t)W : 1u;                       // interpreters per divider wave
    constexpr uint32_t AREA = lds_area_bytes((uint32_t)W, LDS_BYTES);  // LDS of one interpreter wave
    constexpr uint32_t ML = mbox_lanes((uint32_t)W), MB = mbox_bytes((uint32_t)W);
    const uint32_t batch = p.batch;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_tiles = (batch + T - 1) / T;
    const uint32_t t = lane % T;
    const uint32_t j = (G == 1) ? 0u : lane / T;
    const uint32_t NS = p.n_streams;  // (1, 2 or 4: NW is a multiple)
    const uint32_t stream = wave < NW ? wave % NS : 0u;
    const uint32_t tile_raw = blockIdx.x * (NW / NS) + (wave < NW ? wave : 0u) / NS;
    const uint32_t tile = tile_raw < n_tiles ? tile_raw : n_tiles - 1;  // (divider wave / absent interpreters: any valid tile)
    const uint32_t set = tile * T + t;
    const uint32_t set_c = set < batch ? set : batch - 1;  // padded lanes of the last tile re-evaluate a real set
    // One buffer descriptor per tile: every operand / destination is a 32-bit tile-relative byte offset.
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    const uint32_t chunk = tile / wst.tiles_per_chunk, tile_in_chunk = tile % wst.tiles_per_chunk;
    char* tile_base = reinterpret_cast<char*>(wst.base[chunk]) + (uint64_t)tile_in_chunk * tile_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile_base, 0, (int)(uint32_t)tile_bytes, 0x00020000);
    const i32x4 rsrc_rec = make_rsrc_words(recs, (p.n_bundles + REC_AHEAD) * (uint32_t)G * 16u);  // (incl. the zero padding behind the last bundle)
    uint32_t* const sync_words = reinterpret_cast<uint32_t*>(tile_base + ws_sync_offset(p.n_const, p.n_slots, T));
    static_assert(NW + PACK <= 16, "sequence words: 64 bytes");
    __shared__ uint4 lds[(NW * AREA + (DIVIDER ? NW * MB + 64u : 0u)) / 16];  // the only LDS object: host-computed addresses are offsets into a wave's area
    const uint32_t area = wave * AREA;  // this interpreter wave's LDS area (0 for single-wave workgroups)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)(char*)lds + area;
    const uint32_t t16 = 16u * t, lane16 = 16u * lane, j16 = 16u * j;
    char* const ldsb = reinterpret_cast<char*>(lds) + area;
    char* const mbox_all = reinterpret_cast<char*>(lds) + NW * AREA;        // mailboxes, then the sequence words
    char* const mbox = mbox_all + (wave < NW ? wave : 0u) * MB;             // this interpreter's mailbox
    volatile uint32_t* const seq = reinterpret_cast<volatile uint32_t*>(mbox_all + NW * MB);  // [NW] posted, then [PACK] served
    if (DIVIDER || NW > 1) {
        if (DIVIDER && threadIdx.x < NW + (uint32_t)PACK) seq[threadIdx.x] = 0;  // sequence words start at zero
        // programs of several streams: the posts of each stream so far (HDR_POST / HDR_WAIT) live in the tile's sync slot
        if (NS > 1 && wave < NW && stream == 0 && lane < MAX_STREAMS && tile_raw < n_tiles) {
            __hip_atomic_store(sync_words + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        // this tile's status words (error bits are OR-ed in at the end, by every stream)
        if (wave < NW && stream == 0 && j == 0 && tile_raw < n_tiles && set < batch) status[set] = 0;
        __syncthreads();
    } else {
        if (j == 0 && tile_raw < n_tiles && set < batch) status[set] = 0;
    }
    if (DIVIDER) {
        if (wave >= NW) {
            // ---- divider wave d: serves the division requests of interpreters [d * W, d * W + W) in order
            // (graph.rs:109: b == 0 -> 0).  Request k of every interpreter has the same div_lanes[k] active lanes;
            // they are packed into passes of 64 lanes.
            const uint32_t first_w = (wave - NW) * WD, first_tile = blockIdx.x * (NW / NS) + first_w / NS;
            if (first_tile >= n_tiles) return;
            const uint32_t n_active = NS > 1 ? 1u : n_tiles - first_tile < WD ? n_tiles - first_tile : WD;  // its interpreters with a tile
            char* const mbox_d = mbox_all + first_w * MB;
            const uint32_t n_requests = NS > 1 ? p.stream_div_requests[first_w % NS] : p.n_div_requests;  // (streams: W = 1, its one interpreter's)
            uint32_t div_lanes_base = 0;  // (the requests of the streams in front of this one come first in div_lanes)
            for (uint32_t q = 0; NS > 1 && q < first_w % NS; ++q) div_lanes_base += p.stream_div_requests[q];
            for (uint32_t k = 0; k < n_requests; ++k) {
                bool ok = true;
                for (uint32_t w = 0; w < n_active; ++w) ok = ok && mbox_wait(seq + first_w + w, k + 1);
                if (!ok) {
                    if (first_tile * T + t < batch) atomicOr(&status[first_tile * T + t], ST_DIVIDER_TIMEOUT);  // (its first interpreter's sets)
                    break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t lanes_k = p.div_lanes[div_lanes_base + k];  // (W = 1: from the program as well)
                const uint32_t total = n_active * lanes_k;
                for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
                    const uint32_t g = g0 + lane;
                    const bool valid = g < total;
                    const uint32_t w = valid ? g / lanes_k : 0u, i = valid ? g % lanes_k : 0u;
                    char* mb = mbox_d + w * MB + 16u * i;
                    const uint4* qa = reinterpret_cast<const uint4*>(mb);
                    const uint4* qb = reinterpret_cast<const uint4*>(mb + 32u * ML);
                    Fr a = fr_from_u4(qa[0], qa[ML]), b = fr_from_u4(qb[0], qb[ML]);
                    if (!valid) b = fr_zero();
                    const Fr inv = fr_inv(b);  // safegcd divsteps; inv(0) = 0
                    const Fr r = u256_select(u256_is_zero(b), fr_zero(), fr_mul(a, inv));
                    if (valid) {  // the quotient overwrites operand a
                        uint4* qr = reinterpret_cast<uint4*>(mb);
                        qr[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
                        qr[ML] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) seq[wave] = k + 1;  // (served word of divider d = seq[NW + d])
            }
            return;
        }
    }
    if (tile_raw >= n_tiles) return;  // an interpreter wave without a tile (last workgroup)
    uint32_t div_seq = 0;  // requests posted / collected so far (interpreter wave)
    uint32_t n_posts = 0, n_waits = 0;
    uint32_t cref_row = p.stream_cref_first[stream];  // row of the third-operand / input-index table for the next C_TERN / C_INPUT bundle

    auto ld = [&](uint32_t off) -> Fr {  // synchronous load of a slot (third operands only)
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + (int)HI, 0, 0);
        return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    auto ld_lds = [&](uint32_t addr) -> Fr {  // addr: LDS byte address of the low half
        const uint4* q = reinterpret_cast<const uint4*>(ldsb + addr);
        const uint4 lo = q[0], hi = q[LDS_HALF_BYTES / 16];
        return fr_from_u4(lo, hi);
    };
    auto ld_rec2 = [&](uint32_t bundle, uint32_t half) -> uint2 {  // one half of this lane's staged record
        return *reinterpret_cast<const uint2*>(ldsb + LDS_REC_OFF + (bundle % REC_AHEAD) * REC_BYTES + lane16 + 8u * half);
    };
    const uint32_t B0 = p.stream_first[stream], NBND = B0 + p.stream_count[stream];  // this wave's bundles [B0, NBND)
    if (NBND == B0) return;
    auto clampb = [&](uint32_t b) { return b < NBND ? b : NBND - 1; };
    auto stage_rec = [&](uint32_t bundle) {  // records of `bundle` -> REC ring (the same record for the T lanes of a node slot)
        // (the record array is padded by REC_AHEAD bundles: no clamp)
        dma16(lds0 + LDS_REC_OFF + (bundle % REC_AHEAD) * REC_BYTES, j16, rsrc_rec, bundle * (uint32_t)G * 16u);
    };
    // The four loads of a bundle's memory operands share ONE M0 write (20 instead of 30 cycles per load,
    // tools/ubench/dma_interleave.hip): the instruction offset k * 1 KiB places load k in its quarter of the STAGE cell;
    // it also moves the memory address, which the scalar offset takes back -- through a descriptor that starts
    // DMA_BIAS bytes in front of the tile, so that the scalar offsets stay non-negative.
    constexpr uint32_t DMA_BIAS = 3u * LDS_HALF_BYTES;
    const i32x4 rsrc_dma = make_rsrc_words(tile_base - DMA_BIAS, (uint32_t)tile_bytes + DMA_BIAS);
    const uint32_t dma_soff0 = DMA_BIAS, dma_soff1 = DMA_BIAS + HI - LDS_HALF_BYTES, dma_soff2 = DMA_BIAS - 2u * LDS_HALF_BYTES,
                   dma_soff3 = DMA_BIAS + HI - 3u * LDS_HALF_BYTES;
    static_assert(LDS_HALF_BYTES == 1024, "instruction offsets below are written out");
    auto stage_operands = [&](uint32_t bundle, const uint2& offs) {  // memory operands of `bundle` -> STAGE ring
        const uint32_t s = lds0 + LDS_STAGE_OFF + (bundle % OPND_AHEAD) * STAGE_BYTES;
        const uint32_t ao = offs.x + t16, bo = offs.y + t16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %3, %5 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %2, %3, %6 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %2, %3, %7 offen offset:3072 lds"
                     :: "s"(s), "v"(ao), "v"(bo), "s"(rsrc_dma), "s"(dma_soff0), "s"(dma_soff1), "s"(dma_soff2), "s"(dma_soff3) : "memory");
    };
    uint32_t err_bits = 0;
    Fr pv = fr_p();  // the modulus in VGPRs for fr_add_wave / fr_sub_wave
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(pv.v[i]));
    // Narrow multiplication bundles (C_MULQ): lane 4v + q works on limbs 2q, 2q+1 of value v = t + T * j (cell v of the
    // ring / stage like everywhere else); per-lane constants of that mapping
    constexpr bool COOP = (uint32_t)T <= COOP_MAX_T;
    const uint32_t cq = lane & 3u, cv = lane >> 2;
    const uint32_t t16c = 16u * (cv % T);                                        // set of the lane's value
    const uint32_t coop_chunk = (cq >> 1) * LDS_HALF_BYTES + (cq & 1u) * 8u;     // its 8 bytes inside a [half][cell][16 B] image
    const uint32_t coop_ring_off = 16u * cv + coop_chunk;
    const uint32_t coop_dst_off = t16c + (cq >> 1) * HI + (cq & 1u) * 8u;        // ... inside a slot of the tile
    uint32_t nq0 = cq == 0 ? CWC_P0 : cq == 1 ? CWC_P2 : cq == 2 ? CWC_P4 : CWC_P6;
    uint32_t nq1 = cq == 0 ? CWC_P1 : cq == 1 ? CWC_P3 : cq == 2 ? CWC_P5 : CWC_P7;
    asm volatile("" : "+v"(nq0), "+v"(nq1));
    const uint32_t trash_doff = p.trash_off | t16;  // (OFF_NOWHERE: dropped by the buffer range check)
    constexpr int C_PROF = 12;  // (classes with counters in the diagnostic buffer: all but C_SYNC)
    unsigned long long pf[C_PROF][2], psec[2][7] = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}};  // psec: MUL, LIN
    unsigned long long pf_fused[2] = {0, 0};  // C_MULF (prof[64], prof[67]) / C_SCAN (prof[68], prof[71])
    unsigned long long pf_scan[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};  // scan bundles by kind: carry, division, convolution, borrow, comparison (prof[72 + 4 k], prof[75 + 4 k])
    if (PROF) {
#pragma unroll
        for (int c = 0; c < C_PROF; ++c) pf[c][0] = pf[c][1] = 0;
    }
#define CWC_STAMP(var) unsigned long long var = 0; if (PROF) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
    // The statements every class path of the loop repeats at the same place of its iteration -- written once, expanded textually
    // (the paths keep their own copies in the ISA: a shared tail costs a lone wave a taken branch per bundle).
    // CWC_SHADOW_ISSUE: what is issued in the shadow of an iteration's LDS reads -- the refill of the record ring (REC[b mod 4] held
    // this bundle's record, read two iterations ago) and the two result stores of the previous bundle.
#define CWC_SHADOW_ISSUE()                                                                                                                  \
    stage_rec(b + 4);                                                                                                                       \
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{r_prev.v[0], r_prev.v[1], r_prev.v[2], r_prev.v[3]}, rsrc, (int)doff_prev, 0, 0);          \
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{r_prev.v[4], r_prev.v[5], r_prev.v[6], r_prev.v[7]}, rsrc, (int)doff_prev + (int)HI, 0, 0)
    // CWC_NEXT_HEADER: right behind the wait for the LDS reads -- the header of bundle b + 2 out of its landing register
    // (CWC_HDR_LANDING below), the fetch of bundle b + 3's in the same statement.
// ("=&s": the move writes its destination BEFORE the load reads the header pointer and the offset -- without the early-clobber mark the register
// allocator may give the destination one of their registers, and did in one stamped MODE 3 instance: the load then read from header word : pointer high
// half, a memory fault found by the class profile of the RSA-class graph; tests/test_host_formats.py checks the built kernels for it)
#define CWC_NEXT_HEADER(var)                                                                                                                 \
    asm volatile("s_mov_b32 %0, xnack_mask_lo\n\ts_load_dword xnack_mask_lo, %1, %2" : "=&s"(var) : "s"(hdr), "s"(hdr_off_n2) : "memory"); \
    hdr_off_n2 += 4u

    // Software pipeline, everything through LDS.  While bundle b computes: its operands sit in STAGE[b mod 2] / the
    // RING; the memory operands of bundle b+1 are landing in STAGE[(b+1) mod 2]; those of bundle b+2 are requested as
    // soon as bundle b has read its own (same STAGE cell); records run REC_AHEAD bundles ahead.  The two stores of a
    // bundle's results are issued in the NEXT iteration, behind that iteration's LDS reads (their issue time hides the
    // LDS latency; the ring write stays at the end of the bundle, later bundles read it).  Vector-memory operations per
    // iteration, in issue order: 2 stores (previous bundle), 4 operand loads, 1 record load -- the counted wait at the
    // top of iteration b (vmcnt(7)) retires the operand loads of bundle b and the record load of bundle b+2, both
    // issued in iteration b-2, and leaves the 7 operations of iteration b-1 in flight.
    static_assert(OPND_AHEAD == 2 && REC_AHEAD == 4, "the counted waits below are written for this pipeline depth");
#pragma unroll
    for (uint32_t q = 0; q < REC_AHEAD; ++q) stage_rec(B0 + q);  // (B0 is a multiple of the pipeline depths)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_operands(B0, ld_rec2(B0, 0));
    stage_operands(B0 + 1, ld_rec2(B0 + 1, 0));
    uint2 rec_hi = ld_rec2(B0, 1);  // {dst | ctrl, a_lds | b_lds << 16} of the current bundle
    uint2 rec_hi_n1 = ld_rec2(B0 + 1, 1);  // ... of the next one (the record of bundle b+2 is read whole in iteration b)
    uint32_t h_cur = hdr[B0], h_n1 = hdr[clampb(B0 + 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_wave0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    // The header an iteration needs -- bundle b + 2's -- is fetched by the iteration before it, right behind that iteration's
    // wait for its LDS reads, and retired by this iteration's wait a bundle's arithmetic later: a miss of the scalar cache
    // (one per 64-byte line of headers, ~1 000 cycles when awaited at once) costs nothing.
    // CWC_HDR_LANDING: the load lands in XNACK_MASK_LO, an SGPR the compiler never allocates (the kernels are built without
    // XNACK and the hardware does not use the mask then).  A pending scalar load in a register the compiler knows about does
    // not survive: it copies a loop-carried asm output between registers before the data has arrived (66 of 298 sites when
    // tried).  The iteration moves the header out behind its wait and issues the next fetch in one statement.  The compiler's
    // own counted LDS waits stay safe with one scalar load it does not know of in flight: the true count is never below the
    // one it assumes.  (The header array is padded: no clamp.)
    asm volatile("s_load_dword xnack_mask_lo, %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "s"(hdr), "s"(4u * B0 + 8u) : "memory");
    uint32_t hdr_off_n2 = 4u * B0 + 12u;  // byte offset of the header the first iteration fetches (for the second)
    Fr r_prev = fr_zero();  // results of the previous bundle, stored one iteration late (first iteration: zeros -> trash slot)
    uint32_t doff_prev = trash_doff;
    for (uint32_t b = B0; b < NBND; ++b) {
        CWC_STAMP(st0);
        const uint32_t h = h_cur;
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        CWC_STAMP(st1);
        if constexpr (COOP) {
            uint32_t cls_q = h & HDR_CLASS_MASK;
            asm volatile("" : "+s"(cls_q));
            static_assert(C_MULQ == 11 && C_SYNC == 12 && C_MULF == 13 && C_SCAN == 14 && C_COUNT == 15, "one compare (class >= C_MULQ) leads to the narrow classes");
            // (limb and bit graphs hold few narrow bundles, if any: their instances keep the other classes on the fall-through path)
            const bool narrow_cls = MODE == 1 ? cls_q >= C_MULQ : cls_q == C_MULQ;  // (one compare; C_MULF exists in the MODE 1 instances only)
            if (M2 ? __builtin_expect(narrow_cls, 0) : narrow_cls) {  // (plain instances: the narrow class stays the fall-through path -- out of line it cost the headline 1.3 %, same box)
            if (MODE != 1 || cls_q == C_MULQ) {  // graph.rs:105, four lanes per product: the iteration of the other classes with its own lane mapping
                // (laid out behind the loop's main line: a taken branch costs a lone wave ~50 cycles, and two of three bundles are not narrow)
                const uint32_t la = rec_hi.y + (t16c | (t16c << 16));
                const Fr a_op = ld_lds(la & 0xffffu);
                const uint2 bq = *reinterpret_cast<const uint2*>(ldsb + (la >> 16) + coop_chunk);
                const uint2 aq = *reinterpret_cast<const uint2*>(ldsb + (la & 0xffffu) + coop_chunk);  // (for linear riders)
                const uint4 rec_full_n2 = *reinterpret_cast<const uint4*>(ldsb + LDS_REC_OFF + ((b + 2) % REC_AHEAD) * REC_BYTES + lane16);
                const uint2 rec_n2 = make_uint2(rec_full_n2.x, rec_full_n2.y), rec_hi_n2 = make_uint2(rec_full_n2.z, rec_full_n2.w);
                CWC_SHADOW_ISSUE();
                asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(a_op.v[0]), "v"(a_op.v[4]), "v"(bq.x), "v"(aq.x), "v"(rec_n2.x), "v"(rec_hi_n2.x) : "memory");
                uint32_t h_n2;  // header of bundle b + 2: fetched one iteration ago (CWC_HDR_LANDING above); the next iteration's fetch follows
                CWC_NEXT_HEADER(h_n2);
                CWC_STAMP(st2);
                stage_operands(b + 2, rec_n2);
                CWC_STAMP(st3);
                uint32_t out[2];
                if (h & (HDR_LIN_ADD | HDR_LIN_SUB)) {  // linear nodes ride in groups of their own (ctrl sub-op per record)
                    fr_mul_coop4r(a_op, aq.x, aq.y, bq.x, bq.y, nq0, nq1, rec_hi.x & CTRL_SUB_MASK, out);
                } else {
                    fr_mul_coop4(a_op, bq.x, bq.y, nq0, nq1, out);
                }
                CWC_STAMP(st4);
                // results: the lane's 8 bytes into the ring cell of its value and straight into the value's slot (the
                // delayed stores of the next iteration go to the trash slot); one vector-memory operation more than the
                // other iterations issue, which only makes the counted wait above retire one older store as well
                *reinterpret_cast<uint2*>(ldsb + LDS_RING_OFF + (b % RING_BUNDLES) * RING_SLOT_BYTES + coop_ring_off) = make_uint2(out[0], out[1]);
                {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{out[0], out[1]}, rsrc, (int)((rec_hi.x & ~CTRL_MASK) + coop_dst_off), 0, 0);
                }
                doff_prev = trash_doff;
                rec_hi = rec_hi_n1;
                rec_hi_n1 = rec_hi_n2;
                h_cur = h_n1;
                h_n1 = h_n2;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (PROF) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long t_now = __builtin_amdgcn_s_memtime();
                    pf[C_MULQ][0] += t_now - st0;
                    pf[C_MULQ][1] += 1;
                    (void)st2; (void)st3; (void)st4;
                }
                continue;
            }
            if constexpr (MODE == 1 && (uint32_t)T <= COOP_FUSE_MAX_T) {
            if (cls_q == C_MULF) {  // fused narrow bundle: (a * b) op2 x2 op3 x3 in the registers of the node's four lanes (program_dev.h)
                // lane l holds record l / T: the group's main record (even positions) and extra record (odd positions) by DPP
                constexpr int QP_MAIN = T == 1 ? 0xA0 /* [0,0,2,2] */ : 0x00 /* [0,0,0,0] */, QP_EXTRA = T == 1 ? 0xF5 /* [1,1,3,3] */ : 0xAA /* [2,2,2,2] */;
                const uint32_t mx = (uint32_t)__builtin_amdgcn_mov_dpp((int)rec_hi.x, QP_MAIN, 0xf, 0xf, false);
                const uint32_t my = (uint32_t)__builtin_amdgcn_mov_dpp((int)rec_hi.y, QP_MAIN, 0xf, 0xf, false);
                const uint32_t xx = (uint32_t)__builtin_amdgcn_mov_dpp((int)rec_hi.x, QP_EXTRA, 0xf, 0xf, false);
                const uint32_t xy = (uint32_t)__builtin_amdgcn_mov_dpp((int)rec_hi.y, QP_EXTRA, 0xf, 0xf, false);
                const uint32_t la = my + (t16c | (t16c << 16)), lx = xy + (t16c | (t16c << 16));
                const Fr a_op = ld_lds(la & 0xffffu);
                const uint2 bq = *reinterpret_cast<const uint2*>(ldsb + (la >> 16) + coop_chunk);
                const Fr x2_full = ld_lds(lx & 0xffffu);                                                    // second stage: a product ...
                const uint2 x2q = *reinterpret_cast<const uint2*>(ldsb + (lx & 0xffffu) + coop_chunk);      // ... or an addition
                const uint2 x3q = *reinterpret_cast<const uint2*>(ldsb + (lx >> 16) + coop_chunk);          // third stage: an addition
                const uint4 rec_full_n2 = *reinterpret_cast<const uint4*>(ldsb + LDS_REC_OFF + ((b + 2) % REC_AHEAD) * REC_BYTES + lane16);
                const uint2 rec_n2 = make_uint2(rec_full_n2.x, rec_full_n2.y), rec_hi_n2 = make_uint2(rec_full_n2.z, rec_full_n2.w);
                CWC_SHADOW_ISSUE();
                asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(a_op.v[0]), "v"(a_op.v[4]), "v"(bq.x), "v"(x2_full.v[0]), "v"(x2_full.v[4]), "v"(x2q.x), "v"(x3q.x), "v"(rec_n2.x), "v"(rec_hi_n2.x) : "memory");
                uint32_t h_n2;  // header of bundle b + 2: fetched one iteration ago (CWC_HDR_LANDING above); the next iteration's fetch follows
                CWC_NEXT_HEADER(h_n2);
                stage_operands(b + 2, rec_n2);
                uint32_t out[2];
                fr_mul_coop4(a_op, bq.x, bq.y, nq0, nq1, out);
                const uint32_t op2 = mx & CTRL_SUB_MASK, op3 = xx & CTRL_SUB_MASK;
                if (h & HDR_F_S2MUL) {  // graph.rs:105 on the running value: the other factor in full, the value as the lanes hold it
                    uint32_t o2[2];
                    fr_mul_coop4(x2_full, out[0], out[1], nq0, nq1, o2);
                    out[0] = op2 == FOP_MUL ? o2[0] : out[0];
                    out[1] = op2 == FOP_MUL ? o2[1] : out[1];
                }
                auto lin_stage = [&](uint32_t op, const uint2& xq) {  // graph.rs:110-111: acc + x, acc - x, x - acc
                    const bool rs = op == FOP_RSUB;
                    uint32_t o[2];
                    fr_addsub_coop4(rs ? xq.x : out[0], rs ? xq.y : out[1], rs ? out[0] : xq.x, rs ? out[1] : xq.y, nq0, nq1, op == FOP_ADD ? 0u : 1u, o);
                    out[0] = op >= FOP_ADD ? o[0] : out[0];
                    out[1] = op >= FOP_ADD ? o[1] : out[1];
                };
                if (h & HDR_F_S2LIN) lin_stage(op2, x2q);
                if (h & HDR_F_S3LIN) lin_stage(op3, x3q);
                *reinterpret_cast<uint2*>(ldsb + LDS_RING_OFF + (b % RING_BUNDLES) * RING_SLOT_BYTES + coop_ring_off) = make_uint2(out[0], out[1]);
                {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{out[0], out[1]}, rsrc, (int)((mx & ~CTRL_MASK) + coop_dst_off), 0, 0);
                }
                doff_prev = trash_doff;
                rec_hi = rec_hi_n1;
                rec_hi_n1 = rec_hi_n2;
                h_cur = h_n1;
                h_n1 = h_n2;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (PROF) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long t_now = __builtin_amdgcn_s_memtime();
                    pf_fused[0] += t_now - st0;
                    pf_fused[1] += 1;
                }
                continue;
            }
            }
            }  // (C_SYNC: the main path)
        }
        const uint32_t ctrl = rec_hi.x & CTRL_MASK;
        const uint32_t doff = (rec_hi.x & ~CTRL_MASK) | t16;
        const uint32_t la = rec_hi.y + (t16 | (t16 << 16));
        const Fr a_op = ld_lds(la & 0xffffu), b_op = ld_lds(la >> 16);
        // the record of bundle b+2, whole: {a_off, b_off} for its staging loads now, the other half two iterations on
        const uint4 rec_full_n2 = *reinterpret_cast<const uint4*>(ldsb + LDS_REC_OFF + ((b + 2) % REC_AHEAD) * REC_BYTES + lane16);
        const uint2 rec_n2 = make_uint2(rec_full_n2.x, rec_full_n2.y), rec_hi_n2 = make_uint2(rec_full_n2.z, rec_full_n2.w);
        // header of bundle b+2: a scalar load issued behind the LDS reads and retired with them by the wait below (left
        // to the compiler it lands after the staging loads, and its whole latency in front of the arithmetic: the
        // first use of an LDS-read register waits for lgkmcnt(0), which counts scalar loads too)
        // results of bundle b-1 -> tile (unconditional: values without a slot and inactive node slots go to the tile's
        // trash slot; a fixed number of vector-memory operations per bundle is what makes the counted wait possible)
        CWC_SHADOW_ISSUE();
        // (diagnostic build: the time at which the reads, the record refill and the stores have been ISSUED -- an s_memtime that is not waited
        // for here (its wait would be the wait for the reads): the lgkmcnt(0) below retires it, nothing reads the register before)
        unsigned long long st_iss = 0;
        if (PROF) st_iss = __builtin_amdgcn_s_memtime();
        uint32_t h_n2;  // header of bundle b + 2: fetched one iteration ago (CWC_HDR_LANDING above); the next iteration's fetch follows
        unsigned long long st2 = 0, st3 = 0;
        // the wait for the LDS reads and what must follow it: every read must have completed before the staging loads overwrite
        // STAGE[b mod 2].  A lambda because the linear class runs it behind its own branch (below), the rest in line.
        auto wait_and_stage = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(a_op.v[0]), "v"(a_op.v[4]), "v"(b_op.v[0]), "v"(b_op.v[4]), "v"(rec_n2.x), "v"(rec_hi_n2.x) : "memory");
            CWC_NEXT_HEADER(h_n2);
            if (PROF) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st2)::"memory");
            stage_operands(b + 2, rec_n2);
            if (PROF) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st3)::"memory");
        };

        const uint32_t cls = h & HDR_CLASS_MASK;
        // the two classes that are 97 % of the bundles are tested first, on a copy the compiler cannot fold into the
        // switch below (folded, every bundle walks a binary search of taken branches)
        uint32_t cls_hot = cls, cls_hot2 = cls;  // (two copies: one would be turned into a switch of its own)
        asm volatile("" : "+s"(cls_hot));
        asm volatile("" : "+s"(cls_hot2));
        const bool active = (ctrl & CTRL_ACTIVE) != 0;
        const uint32_t sub = ctrl & CTRL_SUB_MASK;
        // end of an iteration: results into the ring, pipeline registers forward.  Every class path ends with its own
        // copy (`finish(r); continue;`): with one shared tail behind the dispatch the compiler carries "which path was
        // taken" flags through it and the hot paths pay a chain of extra branches.
        auto finish = [&](const Fr& r) {
            CWC_STAMP(st4);
            r_prev = r;
            doff_prev = doff;
            {   // publish the results of this bundle in the ring (read by later bundles of this wave, in order)
                uint4* q = reinterpret_cast<uint4*>(ldsb + LDS_RING_OFF + (b % RING_BUNDLES) * RING_SLOT_BYTES + lane16);
                q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
                q[LDS_HALF_BYTES / 16] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
            }
            rec_hi = rec_hi_n1;
            rec_hi_n1 = rec_hi_n2;
            h_cur = h_n1;
            h_n1 = h_n2;
            // Later bundles read these stores from other lanes of this wave; a wave's vector-memory instructions
            // execute in order, the fence only keeps the compiler from reordering them.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (PROF) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring write belongs to this bundle
                const unsigned long long t_now = __builtin_amdgcn_s_memtime();
    #pragma unroll
                for (int c = 0; c < C_PROF; ++c)
                    if (cls == (uint32_t)c) {
                        pf[c][0] += t_now - st0;  // cycles of this bundle in the pipelined loop (stamp bookkeeping excluded)
                        pf[c][1] += 1;
                    }
                if (M2 && cls == C_SCAN) {
                    pf_fused[0] += t_now - st0;
                    pf_fused[1] += 1;
                    const int kind = (h & HDR_SCAN_CONV) ? 2 : (h & HDR_SCAN_BORROW) ? 3 : (h & HDR_SCAN_LEX) ? 4 : (h & HDR_SCAN_DIV) ? 1 : 0;
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        if (kind == q) {
                            pf_scan[q][0] += t_now - st0;
                            pf_scan[q][1] += 1;
                        }
                }
                if (cls == C_MUL || cls == C_LIN) {
                    unsigned long long* q = psec[cls == C_MUL ? 0 : 1];
                    q[0] += st1 - st0;    // top of the loop + counted wait for the staged operands
                    q[1] += st2 - st1;    // operand / record reads from LDS, the previous bundle's stores issued behind them
                    q[2] += st3 - st2;    // issuing the staging loads
                    q[3] += st4 - st3;    // class dispatch + arithmetic
                    q[4] += t_now - st4;  // ring write
                    q[5] += 1;
                    q[6] += st_iss - st1;  // the issue part of section 1 (the rest of it is the wait for the LDS reads)
                }
            }
        };
        Fr r;
        // booleans in the form the bundle's users read: Montgomery (2^256 mod r) or the canonical integer 1 (evaluated in the
        // classes that produce booleans only: in front of the dispatch it costs every bundle ten issue slots)
        auto one_out = [&]() -> Fr { return (h & HDR_OUT_CANON) ? Fr{{1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}} : fr_one(); };
        // The linear class branches off in front of the wait: its taken branch (~50 cycles for a lone wave) then runs in the
        // shadow of the LDS reads; the multiplier stays the fall-through behind the wait.
        if (cls_hot2 == C_LIN) {
            wait_and_stage();
            // graph.rs:110-111 Add/Sub; Neg (:188-194) arrives as 0 - a (0 - 0 = 0, else r - a).  Most bundles are
            // uniform (header bits); a mixed one computes both and selects per lane.
            if (!(h & HDR_LIN_SUB)) {
                r = fr_add_wave(a_op, b_op, pv);
            } else if (!(h & HDR_LIN_ADD)) {
                r = fr_sub_wave(a_op, b_op, pv);
            } else {
                const unsigned long long subm = __ballot(sub == SUB_SUB);
                r = fr_addsub_wave(a_op, b_op, pv, sub == SUB_SUB ? ~0u : 0u, subm, ~subm);
            }
            finish(r);
            continue;
        }
        wait_and_stage();
        if (__builtin_expect(cls_hot == C_MUL, WIDE ? 0 : 1)) {  // graph.rs:105 (MODE 3: wide-register graphs are scan bundles first, the scan class keeps the fall-through path)
            if constexpr (M2) {
                if (h & HDR_MUL_CC) {  // canonical x canonical -> canonical: limb-sized factors (below 2^64 everywhere in the wave) multiply as integers
                    auto limb_product = [&](const Fr& x, const Fr& y) -> Fr {  // x, y < 2^64
                        const uint64_t p00 = (uint64_t)x.v[0] * y.v[0], p01 = (uint64_t)x.v[0] * y.v[1], p10 = (uint64_t)x.v[1] * y.v[0], p11 = (uint64_t)x.v[1] * y.v[1];
                        const uint64_t m1 = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
                        const uint64_t m2 = (m1 >> 32) + (p01 >> 32) + (p10 >> 32) + (uint32_t)p11;
                        Fr z = fr_zero();
                        z.v[0] = (uint32_t)p00;
                        z.v[1] = (uint32_t)m1;
                        z.v[2] = (uint32_t)m2;
                        z.v[3] = (uint32_t)((m2 >> 32) + (p11 >> 32));
                        return z;
                    };
                    const uint32_t hi_a = a_op.v[2] | a_op.v[3] | a_op.v[4] | a_op.v[5] | a_op.v[6] | a_op.v[7], hi_b = b_op.v[2] | b_op.v[3] | b_op.v[4] | b_op.v[5] | b_op.v[6] | b_op.v[7];
                    if (!wave_any((hi_a | hi_b) != 0u)) {
                        r = limb_product(a_op, b_op);
                    } else if (WIDE && !wave_any(((a_op.v[3] | b_op.v[3]) >> 30 | a_op.v[4] | a_op.v[5] | a_op.v[6] | a_op.v[7] | b_op.v[4] | b_op.v[5] | b_op.v[6] | b_op.v[7]) != 0u)) {
                        // registers wider than a word (round 5: the 121-bit registers of circom-bigint's RSA circuits): factors below 2^126 everywhere in
                        // the wave multiply as integers, sixteen 32 x 32 multiply-adds; the product is below 2^252 < r, the canonical a * b mod r
                        uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            uint32_t carry = 0;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const uint64_t t = (uint64_t)a_op.v[i] * b_op.v[jj] + z[i + jj] + carry;
                                z[i + jj] = (uint32_t)t;
                                carry = (uint32_t)(t >> 32);
                            }
                            z[i + 4] = carry;
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) r.v[k] = z[k];
                    } else {
                        // small NEGATIVE factors (r - y with y < 2^64: the -1 of a bit circuit's 1 - 2b, a difference of bits): (-x) * y = -(x * y)
                        Fr na, nb;
                        (void)u256_sub(na, fr_p(), a_op);
                        (void)u256_sub(nb, fr_p(), b_op);
                        const bool a_neg = hi_a != 0u, b_neg = hi_b != 0u;
                        const Fr ma = u256_select(a_neg, na, a_op), mb = u256_select(b_neg, nb, b_op);
                        if (!wave_any((ma.v[2] | ma.v[3] | ma.v[4] | ma.v[5] | ma.v[6] | ma.v[7] | mb.v[2] | mb.v[3] | mb.v[4] | mb.v[5] | mb.v[6] | mb.v[7]) != 0u)) {
                            const Fr z = limb_product(ma, mb);
                            Fr nz;
                            (void)u256_sub(nz, fr_p(), z);
                            r = u256_select(both(a_neg != b_neg, !u256_is_zero(z)), nz, z);
                        } else {  // any operands: a into Montgomery form, then Montgomery x canonical = the canonical product
                            r = fr_mul_wave(fr_mul_wave(a_op, fr_r2(), pv), b_op, pv);
                        }
                    }
                    finish(r);
                    continue;
                }
            }
            r = fr_mul_wave(a_op, b_op, pv);
            // linear nodes riding in this bundle's free node slots (graph.rs:110-111)
            if ((h & (HDR_LIN_ADD | HDR_LIN_SUB)) == (HDR_LIN_ADD | HDR_LIN_SUB)) {
                const unsigned long long subm = __ballot(sub == SUB_SUB);
                r = u256_select(sub <= SUB_SUB, fr_addsub_wave(a_op, b_op, pv, sub == SUB_SUB ? ~0u : 0u, subm, ~subm), r);
            } else if (h & HDR_LIN_ADD) {
                r = u256_select(sub == SUB_ADD, fr_add_wave(a_op, b_op, pv), r);
            } else if (h & HDR_LIN_SUB) {
                r = u256_select(sub == SUB_SUB, fr_sub_wave(a_op, b_op, pv), r);
            }
            finish(r);
            continue;
        }
        // Programs of the MODE 2 instances (limb and bit graphs): the scan class is tested here, in front of the switch's binary search of
        // taken branches (~50 cycles each for a lone wave) -- config 5 44.2 -> 43.2 ms, and the smaller switch serves the other classes
        // sooner: sha256_512 372 -> 380 k witnesses/s.  (The bit class moved here as well cost both 10-20 %: profiles/r04_dispatch_ab.txt.)
        if constexpr (M2 && (uint32_t)T <= SCAN_MAX_T) {
            uint32_t cls_scan = cls;
            asm volatile("" : "+s"(cls_scan));
            if (WIDE ? __builtin_expect(cls_scan == C_SCAN, 1) : (cls_scan == C_SCAN)) {  // the steps of serial limb recurrences, pair after pair (program_dev.h); graph.rs:105, 110-121, 637-687
                r = fr_zero();
                if constexpr (M2 && (uint32_t)T <= SCAN_MAX_T) {
                    if (h & HDR_SCAN_CONV) {  // the columns of a k x k limb product (program_dev.h): out_c = sum_{i + j = c} x_i y_j; graph.rs:105, 110
                        const uint32_t k = (h >> HDR_SCAN_ITER_SHIFT) + 1u;
                        const bool holds_y = active && lane < k * (uint32_t)T;  // (the columns k and above name a factor only to name one)
                        if (!wave_any(both(active, (a_op.v[2] | a_op.v[3] | a_op.v[4] | a_op.v[5] | a_op.v[6] | a_op.v[7] | b_op.v[2] | b_op.v[3] | b_op.v[4] | b_op.v[5] | b_op.v[6] | b_op.v[7]) != 0u))) {
                            uint32_t col[5];
                            conv_limb_columns<T>(k, lane, ((uint64_t)a_op.v[1] << 32) | a_op.v[0], holds_y ? (((uint64_t)b_op.v[1] << 32) | b_op.v[0]) : 0ull, col);
#pragma unroll
                            for (int w = 0; w < 5; ++w) r.v[w] = col[w];
                        } else {  // factors of any size: the same rounds with field products (x_i into Montgomery form, times the canonical y) and field sums
                            Fr yy = u256_select(holds_y, b_op, fr_zero());
                            for (uint32_t i = 0; i < k; ++i) {
                                Fr xi;
#pragma unroll
                                for (int w = 0; w < 8; ++w) {
                                    if constexpr (T == 1) {
                                        xi.v[w] = (uint32_t)__builtin_amdgcn_readlane((int)a_op.v[w], (int)i);
                                    } else {
                                        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)a_op.v[w], (int)(2 * i)), o = (uint32_t)__builtin_amdgcn_readlane((int)a_op.v[w], (int)(2 * i + 1));
                                        xi.v[w] = (lane & 1u) ? o : e;
                                    }
                                }
                                r = fr_add_wave(r, fr_mul_wave(fr_mul_wave(xi, fr_r2(), pv), yy, pv), pv);
#pragma unroll
                                for (int w = 0; w < 8; ++w) yy.v[w] = wave_shr_lanes<T>(yy.v[w]);
                            }
                        }
                        finish(r);
                        continue;
                    }
                    // lanes of a pair: the OUT record's lanes, then the ACC record's (T each); both compute the whole step
                    constexpr int QP_OUT = T == 1 ? 0xA0 /* [0,0,2,2] */ : 0x44 /* [0,1,0,1] */, QP_ACC = T == 1 ? 0xF5 /* [1,1,3,3] */ : 0xEE /* [2,3,2,3] */;
                    constexpr int D = 2 * T;  // lanes from a pair to the next
                    const uint32_t sh = (h >> HDR_SCAN_SHIFT_SHIFT) & 0xffu, iters = (h >> HDR_SCAN_ITER_SHIFT) + 1u;
                    const bool start = (sub & SCAN_START) != 0, role_acc = (sub & SCAN_ROLE_ACC) != 0;
                    const Fr x = fr_quad_perm<QP_OUT>(a_op), acc0 = fr_quad_perm<QP_OUT>(b_op);
                    // (MODE 3.  Everything below is free of per-lane branches -- both() / either() for && / ||, both arms computed in front of a ?:,
                    // role and kind selections as u256_select: this if / else-if chain is a region of uniform tests, and ONE divergent branch anywhere
                    // below it makes StructurizeCFG rewrite all of them into flag registers and chains of s_cbranch_vcc* -- the 64-bit carry and division
                    // paths included, which is what the round-5 kinds cost programs that never ran them: DESIGN 5, profiles/r05_structurizer_ab.txt.)
                    if (WIDE && (h & (HDR_SCAN_BORROW | HDR_SCAN_LEX))) {
                        // ---- one-bit recurrences (round 5), all steps of the bundle at once: c_out = gen | (prop & c_in), scan_bit_lookahead
                        const Fr y = fr_quad_perm<QP_ACC>(a_op);
                        const bool seg = start || !active;
                        const uint32_t a0 = both(both(start, active), !u256_is_zero(acc0)) ? 1u : 0u;  // the bit coming into a segment (a chain end without one reads 0; a comparison chain's may be a Montgomery-form boolean)
                        auto less = [](const Fr& a, const Fr& b) {  // a < b as the borrow of a - b, the difference kept alive (see C_CMPS below)
                            Fr t;
                            const uint32_t borrow = u256_sub(t, a, b);
                            asm volatile("" ::"v"(t.v[7]));
                            return borrow != 0;
                        };
                        auto signed_lt = [&](const Fr& a, const Fr& b) {  // graph.rs:723-755: negative = above (r - 1) / 2
                            const bool an = less(fr_half(), a), bn = less(fr_half(), b), ab = less(a, b);  // (all three computed: no branch around the third)
                            return an == bn ? ab : an;
                        };
                        // lt / gt of two canonical integers with the reference's signed comparison (graph.rs:723-755); registers below 2^128 everywhere
                        // in the wave are non-negative: one four-word subtraction decides
                        auto compare = [&](const Fr& a, const Fr& b, bool& lt, bool& gt) {
                            if (!wave_any(both(active, (a.v[4] | a.v[5] | a.v[6] | a.v[7] | b.v[4] | b.v[5] | b.v[6] | b.v[7]) != 0u))) {
                                uint32_t d[4], bw = 0;
#pragma unroll
                                for (int k = 0; k < 4; ++k) d[k] = sbb32(a.v[k], b.v[k], bw);
                                lt = bw != 0u;
                                gt = bw == 0u && (d[0] | d[1] | d[2] | d[3]) != 0u;
                            } else {
                                lt = signed_lt(a, b);
                                gt = signed_lt(b, a);
                            }
                        };
                        if ((h & (HDR_SCAN_BORROW | HDR_SCAN_LEX)) == (HDR_SCAN_BORROW | HDR_SCAN_LEX)) {
                            // selections (program_dev.h SelCode): OUT record {a, b} = x, acc0: the comparison; ACC record {p, q} = y, and the second
                            // operand of the ACC record: the arms.  out = a <cmp> b (SEL_NEZ: a != 0), acc = out ? p : q (graph.rs:130-133, 221-225)
                            const Fr q_arm = fr_quad_perm<QP_ACC>(b_op);
                            const uint32_t code = sh & 7u;
                            bool cond;
                            if (code == SEL_NEZ) {
                                cond = !u256_is_zero(x);
                            } else {
                                bool lt, gt;
                                compare(x, acc0, lt, gt);
                                cond = code == SEL_LT ? lt : code == SEL_GT ? gt : code == SEL_LEQ ? !gt : !lt;
                            }
                            const Fr one_out = (sh & SEL_OUT_MONT) ? fr_one() : Fr{{1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
                            r = u256_select(cond, u256_select(role_acc, y, one_out), u256_select(role_acc, q_arm, fr_zero()));
                        } else if (h & HDR_SCAN_LEX) {
                            // acc' = x > y ? KG : x < y ? KL : acc (graph.rs:130-131, 221-225): generate = the registers differ and the winner's
                            // constant is 1, propagate = they are equal
                            const uint32_t kg = (h & HDR_SCAN_KG) ? 1u : 0u, kl = (h & HDR_SCAN_KL) ? 1u : 0u;
                            bool lt, gt;
                            compare(x, y, lt, gt);
                            lt = both(lt, active);
                            gt = both(gt, active);
                            const bool gen = either(both(gt, kg != 0u), both(lt, kl != 0u)), prop = both(active, !either(gt, lt));
                            const uint32_t cin = scan_bit_lookahead<T>(seg, either(gen, both(both(seg, prop), a0 != 0u)), prop, lane);
                            const uint32_t bin = seg ? a0 : cin;
                            const bool res = role_acc ? either(gen, both(prop, bin != 0u)) : bin != 0u;  // (the OUT value is read by nothing)
                            r = u256_select(res, sh ? fr_one() : Fr{{1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}}, fr_zero());  // (shift field 1: Montgomery-form booleans)
                        } else {
                            // borrow chain: s = y + bin; c = x >= s; out = c ? x - y - bin : x - y - bin + 2^n; acc' = c ? 0 : 1 (graph.rs:110-111, 133, 221-225)
                            const uint32_t m[4] = {mask_word(sh, 0), mask_word(sh, 1), mask_word(sh, 2), mask_word(sh, 3)};
                            const uint32_t above = (x.v[0] & ~m[0]) | (x.v[1] & ~m[1]) | (x.v[2] & ~m[2]) | (x.v[3] & ~m[3]) | x.v[4] | x.v[5] | x.v[6] | x.v[7] |
                                                   (y.v[0] & ~m[0]) | (y.v[1] & ~m[1]) | (y.v[2] & ~m[2]) | (y.v[3] & ~m[3]) | y.v[4] | y.v[5] | y.v[6] | y.v[7];
                            if (sh <= 126u && !wave_any(both(active, above != 0u))) {
                                // registers below 2^n everywhere in the wave: everything is a small non-negative integer, the comparison unsigned;
                                // generate = x < y, propagate = x == y
                                uint32_t d[4], bw = 0;
#pragma unroll
                                for (int k = 0; k < 4; ++k) d[k] = sbb32(x.v[k], y.v[k], bw);
                                const bool gen = both(active, bw != 0u), prop = both(active, (d[0] | d[1] | d[2] | d[3]) == 0u);
                                const uint32_t cin = scan_bit_lookahead<T>(seg, either(gen, both(both(seg, prop), a0 != 0u)), prop, lane);
                                const uint32_t bin = seg ? a0 : cin;
                                const bool bout = either(gen, both(prop, bin != 0u));
                                // x - y - bin, + 2^n when a borrow leaves (mod 2^128: the result is in [0, 2^n))
                                uint32_t e[4], b2 = 0;
                                e[0] = sbb32(d[0], bin, b2);
#pragma unroll
                                for (int k = 1; k < 4; ++k) e[k] = sbb32(d[k], 0u, b2);
                                uint32_t cy = 0;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const uint32_t add = (bout && (sh >> 5) == (uint32_t)k) ? (1u << (sh & 31u)) : 0u;
                                    e[k] = adc32(e[k], add, cy);
                                }
#pragma unroll
                                for (int k = 0; k < 4; ++k) r.v[k] = role_acc ? (k == 0 && bout ? 1u : 0u) : e[k];
                            } else {
                                // any operands: the unfused nodes' field arithmetic and signed comparison, round by round
                                Fr pw = fr_zero();
#pragma unroll
                                for (int k = 0; k < 8; ++k) pw.v[k] = (sh >> 5) == (uint32_t)k ? (1u << (sh & 31u)) : 0u;
                                uint32_t bo = 0;
                                Fr diff = fr_zero();
                                const Fr xy = fr_sub_wave(x, y, pv);
                                for (uint32_t it = 0; it < iters; ++it) {
                                    const uint32_t sft = wave_shr_lanes<D>(bo);
                                    Fr in = fr_zero();
                                    in.v[0] = start ? a0 : sft;
                                    const bool c = !signed_lt(x, fr_add_wave(y, in, pv));
                                    const Fr d0 = fr_sub_wave(xy, in, pv);
                                    diff = u256_select(c, d0, fr_add_wave(d0, pw, pv));
                                    bo = c ? 0u : 1u;
                                }
                                r = u256_select(role_acc, Fr{{bo, 0u, 0u, 0u, 0u, 0u, 0u, 0u}}, diff);
                            }
                        }
                    } else if (!(h & HDR_SCAN_DIV)) {
                        // ---- carry chain: t = x + acc; limb = t & (2^n - 1); acc' = t >> n
                        const bool small = !wave_any((x.v[4] | x.v[5] | x.v[6] | x.v[7] | acc0.v[4] | acc0.v[5] | acc0.v[6] | acc0.v[7]) != 0u) && sh >= 1u && sh <= 128u;
                        // 64-bit limbs: every segment of the bundle at once (scan_carry_parallel) when x (+ the accumulator coming in) < 2^192
                        const bool seg = start || !active;
                        uint32_t xp[8];
                        {
                            uint32_t cy = 0;
#pragma unroll
                            for (int k = 0; k < 8; ++k) xp[k] = adc32(active ? x.v[k] : 0u, start && active ? acc0.v[k] : 0u, cy);
                            xp[7] |= cy << 31;  // (x + acc at or above 2^256: outside every parallel form)
                        }
                        // registers of any width up to 126 bits (round 5): x (+ the accumulator coming in) below min(2^(3n), 2^252) everywhere in the wave
                        // (instances for widths of 32 .. 126 bits but 64 -- the 55-, 86-, 100-, 121-bit registers of big-integer libraries; narrower
                        // registers take the serial rounds below)
                        bool wide_ok = WIDE && sh != 64u && sh >= 32u && sh <= 126u && iters > 2u;
                        if (wide_ok) {
                            const Fr xpf = Fr{{xp[0], xp[1], xp[2], xp[3], xp[4], xp[5], xp[6], xp[7]}};
                            if (3u * sh >= 252u) {  // (registers of 84 bits and more: the bound is 2^252)
                                wide_ok = !wave_any((xp[7] >> 28) != 0u);
                            } else {
                                const uint32_t top = 3u * sh;
                                uint32_t hi_or = 0;
#pragma unroll
                                for (int k = 0; k < 8; ++k) hi_or |= xp[k] & ~(top >= 32u * (k + 1) ? 0xffffffffu : top > 32u * k ? (1u << (top - 32u * k)) - 1u : 0u);
                                wide_ok = !wave_any(hi_or != 0u);
                            }
                            if (wide_ok) {
                                Fr limb, carry;
                                // (the word part of the width as a template parameter: shifts with fixed register positions)
                                if (sh >= 96u) scan_carry_parallel_wide<T, 3>(seg, lane, sh, xpf, limb, carry);
                                else if (sh >= 64u) scan_carry_parallel_wide<T, 2>(seg, lane, sh, xpf, limb, carry);
                                else scan_carry_parallel_wide<T, 1>(seg, lane, sh, xpf, limb, carry);
                                r = u256_select(role_acc, carry, limb);
                            }
                        }
                        if (wide_ok) {
                        } else if (sh == 64u && iters > 2u && !wave_any((xp[6] | xp[7]) != 0u)) {
                            const uint32_t xw[6] = {xp[0], xp[1], xp[2], xp[3], xp[4], xp[5]};
                            uint32_t limb[2], carry[6];
                            scan_carry_parallel<T>(seg, lane, xw, limb, carry);
#pragma unroll
                            for (int k = 0; k < 6; ++k) r.v[k] = role_acc ? carry[k] : (k < 2 ? limb[k] : 0u);
                        } else if (small) {
                            const uint32_t m[4] = {mask_word(sh, 0), mask_word(sh, 1), mask_word(sh, 2), mask_word(sh, 3)};
                            const uint32_t xs[4] = {x.v[0], x.v[1], x.v[2], x.v[3]}, as[4] = {acc0.v[0], acc0.v[1], acc0.v[2], acc0.v[3]};
                            uint32_t limb[4], carry[4];
                            const uint32_t bs = sh & 31u;
                            switch (sh >> 5) {
                                case 0: scan_carry_rounds<0, D>(iters, bs, m, start, xs, as, limb, carry); break;
                                case 1: scan_carry_rounds<1, D>(iters, bs, m, start, xs, as, limb, carry); break;
                                case 2: scan_carry_rounds<2, D>(iters, bs, m, start, xs, as, limb, carry); break;
                                case 3: scan_carry_rounds<3, D>(iters, bs, m, start, xs, as, limb, carry); break;
                                default: scan_carry_rounds<4, D>(iters, bs, m, start, xs, as, limb, carry); break;
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) r.v[k] = role_acc ? carry[k] : limb[k];
                        } else {
                            Fr c = fr_zero(), limb = fr_zero();
                            for (uint32_t it = 0; it < iters; ++it) {
                                Fr in;
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    const uint32_t sft = wave_shr_lanes<D>(c.v[k]);
                                    in.v[k] = start ? acc0.v[k] : sft;
                                }
                                const Fr t = fr_add_wave(x, in, pv);
#pragma unroll
                                for (int k = 0; k < 8; ++k) limb.v[k] = t.v[k] & mask_word(sh, (uint32_t)k);
                                c = u256_shr(t, sh);
                            }
                            r = u256_select(role_acc, c, limb);
                        }
                    } else {
                        // ---- long division by one limb: t = acc * 2^k + x; q = t idiv d; acc' = t mod d (d == 0: both 0)
                        const Fr d = fr_quad_perm<QP_ACC>(a_op), bm = fr_quad_perm<QP_ACC>(b_op);
                        const bool dz = u256_is_zero(d);
                        Fr ds = d;
                        ds.v[0] |= dz ? 1u : 0u;
                        const bool small = !wave_any((x.v[2] | x.v[3] | x.v[4] | x.v[5] | x.v[6] | x.v[7] | acc0.v[2] | acc0.v[3] | acc0.v[4] | acc0.v[5] | acc0.v[6] | acc0.v[7] |
                                                      d.v[2] | d.v[3] | d.v[4] | d.v[5] | d.v[6] | d.v[7]) != 0u) && sh >= 1u && sh <= 64u;
                        Fr quo = fr_zero(), rem = fr_zero();
                        if (small) {
                            // every remainder stays below 2^64 (below d, or 0), so t = rem * 2^k + x < 2^128.  The divisor is the
                            // lane's own for all rounds: its normalisation shift and reciprocal are computed once
                            // (fr_gfx950.hpp u128_divrem_64_recip), a round is two multiplications and two corrections.
                            const uint64_t dv = ((uint64_t)ds.v[1] << 32) | ds.v[0];
                            const uint32_t s = clz64_nonzero(dv);
                            const uint64_t dn = dv << s, rv = recip64(dn);
                            const uint64_t xl = ((uint64_t)x.v[1] << 32) | x.v[0];
                            const uint64_t a0 = ((uint64_t)acc0.v[1] << 32) | acc0.v[0];
                            uint64_t r64 = 0, qh = 0, ql = 0;
                            // one divisor per segment, every incoming remainder below it: all segments at once (scan_div_parallel)
                            const bool seg = start || !active;
                            const uint64_t d_prev = ((uint64_t)wave_shr_lanes<D>((uint32_t)(dv >> 32)) << 32) | wave_shr_lanes<D>((uint32_t)dv);
                            if (sh == 64u && iters > 2u && !wave_any(both(active, either(dz, start ? a0 >= dv : dv != d_prev)))) {
                                scan_div_parallel<T>(seg, lane, iters, active ? dv : 1ull, active ? xl : 0ull, active ? a0 : 0ull, ql, r64);
                            } else if (sh == 64u) {
                                // t = rem : x -- the low word is the lane's own x for all rounds: its normalised parts are made once
                                const uint64_t xn = xl << s, xh = (xl >> 1) >> (63u - s);
                                for (uint32_t it = 0; it < iters; ++it) {
                                    const uint32_t s0 = wave_shr_lanes<D>((uint32_t)r64), s1 = wave_shr_lanes<D>((uint32_t)(r64 >> 32));
                                    const uint64_t in = start ? a0 : (((uint64_t)s1 << 32) | s0);
                                    uint64_t q_h = 0, q_l, rr;
                                    if (wave_any(in >= dv)) {  // (a chain's first step with an accumulator that is not below its divisor)
                                        u128_divrem_64_recip(in, xl, dv, s, dn, rv, true, q_h, q_l, rr);
                                    } else {
                                        uint64_t rn;
                                        div2by1((in << s) | xh, xn, dn, rv, q_l, rn);
                                        rr = rn >> s;
                                    }
                                    qh = dz ? 0ull : q_h;
                                    ql = dz ? 0ull : q_l;
                                    r64 = dz ? 0ull : rr;
                                }
                            } else {
                                for (uint32_t it = 0; it < iters; ++it) {
                                    const uint32_t s0 = wave_shr_lanes<D>((uint32_t)r64), s1 = wave_shr_lanes<D>((uint32_t)(r64 >> 32));
                                    const uint64_t in = start ? a0 : (((uint64_t)s1 << 32) | s0);
                                    // in * 2^k (1 <= k < 64) + x
                                    const uint64_t lo = in << sh, hi = (in >> 1) >> (63u - sh);
                                    const uint64_t tl = lo + xl, th = hi + (tl < lo ? 1ull : 0ull);
                                    uint64_t q_h, q_l, rr;
                                    u128_divrem_64_recip(th, tl, dv, s, dn, rv, wave_any(th >= dv), q_h, q_l, rr);
                                    qh = dz ? 0ull : q_h;
                                    ql = dz ? 0ull : q_l;
                                    r64 = dz ? 0ull : rr;
                                }
                            }
                            quo.v[0] = (uint32_t)ql; quo.v[1] = (uint32_t)(ql >> 32); quo.v[2] = (uint32_t)qh; quo.v[3] = (uint32_t)(qh >> 32);
                            rem.v[0] = (uint32_t)r64; rem.v[1] = (uint32_t)(r64 >> 32);
                        } else {
                            for (uint32_t it = 0; it < iters; ++it) {
                                Fr in;
#pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    const uint32_t sft = wave_shr_lanes<D>(rem.v[k]);
                                    in.v[k] = start ? acc0.v[k] : sft;
                                }
                                // (acc canonical) x (2^k in Montgomery form) = acc * 2^k mod r, canonical; any acc < 2^256 is reduced
                                const Fr t = fr_add_wave(fr_mul_wave(in, bm, pv), x, pv);
                                const uint32_t lx = u256_bitlen(t), ly = u256_bitlen(ds);
                                const uint32_t my_dig = lx >= ly ? (lx - ly + 32u) >> 5 : 0u;
                                uint32_t dig = 0;
#pragma unroll
                                for (uint32_t dd = 1; dd <= 8; ++dd) dig = wave_any(my_dig >= dd) ? dd : dig;
                                Fr q1, r1;
                                u256_divrem_digits(q1, r1, t, ds, dig, ly);
                                quo = u256_select(dz, fr_zero(), q1);
                                rem = u256_select(dz, fr_zero(), r1);
                            }
                        }
                        r = u256_select(role_acc, rem, quo);
                    }
                }
                finish(r);
                continue;
            }
        }
        switch (cls) {
            case C_INPUT: {  // graph.rs:376  Fr::new(inputs[i])
                const uint32_t idx = crefs[(size_t)cref_row * G + j];
                ++cref_row;
                const uint4* q = inputs + ((size_t)set_c * p.n_inputs + idx) * 2;
                // (any value below 2^256 is reduced: Fr::new; x * R^2 / R = the Montgomery form, x * R / R = the canonical integer of bit graphs)
                r = fr_mul_wave(fr_from_u4(q[0], q[1]), (h & HDR_OUT_CANON) ? fr_one() : fr_r2(), pv);
                break;
            }
            case C_DIVREQ: {  // hand the operands to the divider wave; this bundle has no result of its own
                if (DIVIDER) {
                    if (lane < ML) {  // (a request has at most ML active lanes: the compiler caps its node count)
                        uint4* qa = reinterpret_cast<uint4*>(mbox + lane16);
                        uint4* qb = reinterpret_cast<uint4*>(mbox + 32u * ML + lane16);
                        qa[0] = make_uint4(a_op.v[0], a_op.v[1], a_op.v[2], a_op.v[3]);
                        qa[ML] = make_uint4(a_op.v[4], a_op.v[5], a_op.v[6], a_op.v[7]);
                        qb[0] = make_uint4(b_op.v[0], b_op.v[1], b_op.v[2], b_op.v[3]);
                        qb[ML] = make_uint4(b_op.v[4], b_op.v[5], b_op.v[6], b_op.v[7]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) seq[wave] = div_seq + 1;
                }
                r = fr_zero();
                break;
            }
            case C_DIVGET: {  // the quotients of the last request (same node slots)
                r = fr_zero();
                if (DIVIDER) {
                    ++div_seq;
                    if (!mbox_wait(seq + NW + wave / WD, div_seq)) err_bits |= ST_DIVIDER_TIMEOUT;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane < ML) {
                        const uint4* qr = reinterpret_cast<const uint4*>(mbox + lane16);
                        r = fr_from_u4(qr[0], qr[ML]);
                    }
                }
                break;
            }
            case C_DIV: {  // graph.rs:109  b == 0 -> 0 else a / b
                const Fr inv = fr_inv(b_op);  // safegcd divsteps; inv(0) = 0
                r = u256_select(u256_is_zero(b_op), fr_zero(), fr_mul(a_op, inv));
                break;
            }
            case C_CMPZ: {  // graph.rs:122-129 Eq/Neq, :134-135 Land/Lor
                const bool az = u256_is_zero(a_op), cz = u256_is_zero(b_op), eq = u256_eq(a_op, b_op);
                const bool v = either(either(both(sub == SUB_EQ, eq), both(sub == SUB_NEQ, !eq)), either(both(sub == SUB_LAND, !either(az, cz)), both(both(sub != SUB_EQ, both(sub != SUB_NEQ, sub != SUB_LAND)), !both(az, cz))));
                r = u256_select(v, one_out(), fr_zero());
                break;
            }
            case C_CMPS: {  // graph.rs:130-133 with u_lt/u_gt/u_lte/u_gte :723-769
                const Fr x = (h & HDR_A_CANON) ? a_op : fr_from_mont(a_op), y = (h & HDR_B_CANON) ? b_op : fr_from_mont(b_op);
                // a < b as the borrow of a - b with the difference kept alive: left to itself the compiler turns a borrow-only chain into
                // eight compare / and / or triples through scalar registers, ~600 cycles of latency per comparison on a lone wave
                // (in the MODE 2 instances only -- limb graphs compare limbs every round; in the plain instances the change costs the hot
                // paths 2-3 % through the loop's register allocation: 81.0 against 83.2 k witnesses/s at the headline)
                auto less = [](const Fr& a, const Fr& b) {
                    Fr t;
                    const uint32_t borrow = u256_sub(t, a, b);
                    if constexpr (M2) asm volatile("" ::"v"(t.v[7]));
                    return borrow != 0;
                };
                bool lt, gt;
                // limb graphs compare registers: below 2^128 everywhere in the wave nothing is negative (graph.rs:723-755: negative = above (r - 1) / 2)
                // and one four-word subtraction decides -- the two comparisons with the constant (r - 1) / 2 are what the compiler lowers to
                // compare / and / or chains through scalar registers, ~1 200 of the bundle's 2 200 cycles on a lone wave
                bool limb_sized = false;
                if constexpr (M2) limb_sized = !wave_any(both(active, (x.v[4] | x.v[5] | x.v[6] | x.v[7] | y.v[4] | y.v[5] | y.v[6] | y.v[7]) != 0u));
                if (M2 && limb_sized) {
                    uint32_t d[4], bw = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = sbb32(x.v[k], y.v[k], bw);
                    lt = bw != 0u;
                    gt = both(bw == 0u, (d[0] | d[1] | d[2] | d[3]) != 0u);
                } else {
                    const bool xn = less(fr_half(), x), yn = less(fr_half(), y);
                    const bool same = xn == yn, xy = less(x, y), yx = less(y, x);  // (both computed: selections, no branch around either -- a divergent branch in a class body makes the structurizer rewrite its uniform ones)
                    lt = same ? xy : xn;
                    gt = same ? yx : yn;
                }
                const bool v = either(either(both(sub == SUB_LT, lt), both(sub == SUB_GT, gt)), either(both(sub == SUB_LEQ, !gt), both(both(sub != SUB_LT, both(sub != SUB_GT, sub != SUB_LEQ)), !lt)));
                r = u256_select(v, one_out(), fr_zero());
                break;
            }
            case C_BIT: {  // graph.rs:621-717
                const Fr x = (h & HDR_A_CANON) ? a_op : fr_from_mont(a_op);
                // compiler-made bit extract (a >> k) & 1 (Band(Shr(a, k), 1), graph.rs:637-672 + :674-687): k rides in
                // the b_lds field, the result is a boolean -- no second conversion, no conversion back
                const uint32_t kx = rec_hi.y >> 20;  // (b_lds = 16 * k)
                const bool is_x = sub == SUB_BITX;
                if (h & HDR_BITX_ALL) {
                    const Fr e = u256_shr(x, kx);
                    r = u256_select((e.v[0] & 1u) != 0u, one_out(), fr_zero());
                    break;
                }
                const Fr y = (h & HDR_B_CANON) ? b_op : fr_from_mont(b_op);  // (canonical values, canonical copies of constants)
                // the canonical result d in the form the users read: canonical, or Montgomery (booleans without a product)
                auto bit_result = [&](const Fr& d) -> Fr {
                    if (h & HDR_OUT_CANON) return d;
                    const bool small = both(d.v[0] < 2u, (d.v[1] | d.v[2] | d.v[3] | d.v[4] | d.v[5] | d.v[6] | d.v[7]) == 0u);
                    if (wave_any(!small)) return fr_mul_wave(d, fr_r2(), pv);
                    return u256_select(d.v[0] != 0u, fr_one(), fr_zero());
                };
                if (h & (HDR_BIT_ALL_SHR | HDR_BIT_ALL_BAND)) {
                    // every node is a Shr or a Band (limb arithmetic: Idiv / Mod by 2^n after the compiler's strength
                    // reduction, masks).  Band: x & y <= min(x, y) < r, nothing to check; Shr: graph.rs:637-672, b >= 254 -> 0
                    Fr d;
#pragma unroll
                    for (int i = 0; i < 8; ++i) d.v[i] = x.v[i] & y.v[i];
                    if (h & HDR_BIT_ALL_SHR) {  // (some or all of them shift)
                        const bool big = either((y.v[1] | y.v[2] | y.v[3] | y.v[4] | y.v[5] | y.v[6] | y.v[7]) != 0u, y.v[0] >= 254u);
                        const Fr sh = u256_select(big, fr_zero(), u256_shr(x, big ? 0u : y.v[0]));
                        d = u256_select(sub == SUB_SHR, sh, d);
                    }
                    r = bit_result(d);
                    break;
                }
                uint32_t hi_or = 0;
#pragma unroll
                for (int i = 1; i < 8; ++i) hi_or |= y.v[i];
                const bool big = either(hi_or != 0u, y.v[0] >= 254u);  // b >= MODULUS_BIT_SIZE -> 0
                const uint32_t n = is_x ? kx : big ? 0u : y.v[0];
                // (mixed bundles: every lane computes the shift and the bitwise forms and selects -- no per-lane branch: DESIGN 5, layout sensitivity)
                const bool shifts = either(either(sub == SUB_SHL, sub == SUB_SHR), is_x);
                Fr d = u256_shr(x, n);
                if (wave_any(sub == SUB_SHL)) d = u256_select(sub == SUB_SHL, u256_shl(x, n), d);  // (left shifts are rare)
                d = u256_select(both(big, !is_x), fr_zero(), d);
                d.v[0] &= is_x ? 1u : 0xffffffffu;
#pragma unroll
                for (int i = 1; i < 8; ++i) d.v[i] = is_x ? 0u : d.v[i];
                {
                    const bool over = both(sub == SUB_SHL, !u256_lt(d, fr_p()));  // graph.rs:634 unwrap on None
                    err_bits |= both(over, active) ? ST_SHL_OVERFLOW : 0u;
                    d = u256_select(over, fr_zero(), d);
                }
                {
                    // and / or / xor = (x | y) & ~(x & y) through masks of the lane's operation
                    const uint32_t m_and = sub == SUB_BAND ? 0xffffffffu : 0u, m_or = sub == SUB_BAND ? 0u : 0xffffffffu, m_xor = (sub == SUB_BAND || sub == SUB_BOR) ? 0u : 0xffffffffu;
                    Fr e;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t a = x.v[i] & y.v[i], o = x.v[i] | y.v[i];
                        e.v[i] = (a & m_and) | ((o & m_or) & ~(a & m_xor));
                    }
                    Fr em;
                    const uint32_t br = u256_sub(em, e, fr_p());  // br == 1 iff e < r; e >= r: one subtraction (e < 2^254 < 2r)
                    err_bits |= both(both(br == 0u, u256_is_zero(em)), both(active, !shifts)) ? ST_BITOP_EQ_R : 0u;  // e == r: reference panics
                    e = u256_select(br == 0u, em, e);
                    d = u256_select(shifts, d, e);
                }
                // boolean-valued results (Num2Bits-style Band(x,1)) skip the Montgomery multiplication
                r = bit_result(d);
                break;
            }
            case C_IDIVMOD: {  // graph.rs:112-121
                const Fr x = (h & HDR_A_CANON) ? a_op : fr_from_mont(a_op), y = (h & HDR_B_CANON) ? b_op : fr_from_mont(b_op);
                const bool yz = u256_is_zero(y);
                Fr ys = y;
                ys.v[0] |= yz ? 1u : 0u;
                Fr q, rem;
                // (round 5, MODE 3) divisors of two 64-bit words everywhere in the wave (or zero: the result is forced below): the quotient-digit
                // estimate of a multi-register long division, a 2n-bit value by an n-bit register (zk-email: n = 121) -- three-by-two division
                // with a reciprocal instead of 32-bit digits (6.3 k -> ~2 k cycles per bundle)
                bool two_words = false;
                if constexpr (WIDE) two_words = !wave_any(both(both(active, !yz), either((ys.v[2] | ys.v[3]) == 0u, (ys.v[4] | ys.v[5] | ys.v[6] | ys.v[7]) != 0u)));
                if (!wave_any((x.v[4] | x.v[5] | x.v[6] | x.v[7] | ys.v[2] | ys.v[3] | ys.v[4] | ys.v[5] | ys.v[6] | ys.v[7]) != 0u)) {
                    u128_divrem_64(q, rem, x, ys);  // limb-sized operands everywhere in the wave (x < 2^128, y < 2^64): short division
                } else if (WIDE && two_words) {
                    const bool sub = either(yz, !active);  // (a lane without a two-word divisor of its own divides by 2^64: its result is unused)
                    Fr yy = ys;
                    yy.v[0] = sub ? 0u : ys.v[0];
                    yy.v[1] = sub ? 0u : ys.v[1];
                    yy.v[2] = sub ? 1u : ys.v[2];
                    yy.v[3] = sub ? 0u : ys.v[3];
#pragma unroll
                    for (int k = 4; k < 8; ++k) yy.v[k] = 0u;
                    u256_divrem_128(q, rem, x, yy);
                } else {
                    // quotient digits (32 bits each) of the longest quotient in the wave: bitlen(x) - bitlen(y) + 1 bits
                    const uint32_t lx = u256_bitlen(x), ly = u256_bitlen(ys);
                    const uint32_t my_dig = lx >= ly ? (lx - ly + 32u) >> 5 : 0u;  // 0..8
                    uint32_t dig = 0;  // wave-wide maximum by ballots (cheaper than a shuffle reduction: the range is tiny)
#pragma unroll
                    for (uint32_t dd = 1; dd <= 8; ++dd) dig = wave_any(my_dig >= dd) ? dd : dig;
                    u256_divrem_digits(q, rem, x, ys, dig, ly);
                }
                const Fr d = u256_select(yz, fr_zero(), u256_select(sub == SUB_IDIV, q, rem));
                r = (h & HDR_OUT_CANON) ? d : fr_mul_wave(d, fr_r2(), pv);
                break;
            }
            case C_TERN: {  // graph.rs:221-225  a == 0 ? c : b ; the third operand is always a memory reference
                const uint32_t cr = crefs[(size_t)cref_row * G + j];
                ++cref_row;
                const Fr y = ld(cr + t16);
                r = u256_select(u256_is_zero(a_op), y, b_op);
                break;
            }
            case C_SYNC: {  // programs of several streams: post and / or wait (a handful of bundles per program)
                r = fr_zero();
                if (h & HDR_POST) {  // the stores of every earlier bundle are in memory: tell the other waves of the tile
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ++n_posts;
                    if (lane == 0) __hip_atomic_store(sync_words + stream, n_posts, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (h & HDR_WAIT) {  // a stream other than 0 waits for stream 0's post (the compiler emits nothing else and the
                                     // validator rejects a wait on stream 0); the loads issued from the next iteration on
                                     // (operands of bundle b + 3, third operands of b + 1) see the prologue's values
                    ++n_waits;
                    if (!post_wait(sync_words, n_waits)) err_bits |= ST_SYNC_TIMEOUT;
                }
                break;
            }
            case C_SCAN: r = fr_zero(); break;  // (MODE 2: handled in front of the switch; other instances never see the class)
            default: r = fr_zero(); break;
        }
        finish(r);
    }
    // the last bundle's results
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{r_prev.v[0], r_prev.v[1], r_prev.v[2], r_prev.v[3]}, rsrc, (int)doff_prev, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{r_prev.v[4], r_prev.v[5], r_prev.v[6], r_prev.v[7]}, rsrc, (int)doff_prev + (int)HI, 0, 0);
    if (PROF && lane == 0) {  // every interpreter wave: longest / shortest / summed run time of the loop
        const unsigned long long cyc = __builtin_amdgcn_s_memtime() - t_wave0;
        atomicMax(&prof[54], cyc);
        atomicMax(&prof[55], (1ull << 40) - cyc);
        atomicAdd(&prof[62], cyc);
        atomicAdd(&prof[63], 1ull);
    }
    if (PROF && lane == 0 && (tile % 64u) == 0u) {
#pragma unroll
        for (int c = 0; c < C_PROF; ++c) {
            atomicAdd(&prof[c * 4 + 0], pf[c][0]);
            atomicAdd(&prof[c * 4 + 3], pf[c][1]);
        }
        atomicAdd(&prof[M2 ? 68 : 64], pf_fused[0]);  // C_MULF / C_SCAN
        atomicAdd(&prof[M2 ? 71 : 67], pf_fused[1]);
        if (M2) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                atomicAdd(&prof[72 + 4 * q], pf_scan[q][0]);
                atomicAdd(&prof[75 + 4 * q], pf_scan[q][1]);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 6; ++q) atomicAdd(&prof[48 + 8 * k + q], psec[k][q]);
        atomicAdd(&prof[92], psec[0][6]);
        atomicAdd(&prof[93], psec[1][6]);
    }
    if (err_bits && set < batch) atomicOr(&status[set], err_bits);
}

// Writes every tile's copy of the constants: the constant area of a tile is n_const slots = n_const*2*T pieces of 16
// bytes, piece e = (constant e / 2T, half (e / T) % 2, set e % T).  Once per (workspace, program).
__global__ __launch_bounds__(256) void fill_consts_kernel(ProgramDev p, WsTable wst, uint32_t n_tiles, uint32_t T) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    const uint32_t pieces = p.n_const * 2u * T;
    if (e >= pieces) return;
    const uint4 v = reinterpret_cast<const uint4*>(p.consts)[(size_t)(e / (2u * T)) * 2u + (e / T) % 2u];
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        char* tb = reinterpret_cast<char*>(wst.base[tile / wst.tiles_per_chunk]) + (uint64_t)(tile % wst.tiles_per_chunk) * tile_bytes;
        reinterpret_cast<uint4*>(tb)[e] = v;
    }
}

// a stored value in the form the output rows carry: canonical integers (MONT = false) or Montgomery form
template <bool MONT>
__device__ __forceinline__ Fr pack_form(const Fr& v, bool canon) {
    if (MONT) return canon ? fr_mul(v, fr_r2()) : v;
    return canon ? v : fr_from_mont(v);
}

// One thread = one witness index of ONE tile, all T sets of it: the thread reads its slot whole ([half][T][16 B] = 32 T
// contiguous bytes, every 64- or 128-byte line of the tile fetched by exactly one wave instruction stream) and writes T
// rows; a wave's 64 consecutive indices make 2 KiB runs in every output row.  (Round 1 split the sets of a slot over
// the waves of a block: twice / four times the read requests for the same lines, 4.6 TB/s; CWC_PACK_V1=1 keeps it for A/B.)
// MONT: the rows keep the interpreter's Montgomery form (x * 2^256 mod r) for a consumer that computes in it.
template <bool MONT, int TT>
__global__ __launch_bounds__(256) void pack_kernel(ProgramDev p, WsTable wst, uint4* __restrict__ out, uint32_t batch, uint32_t T_) {
    const uint32_t T = TT ? (uint32_t)TT : T_;
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= p.n_witness) return;
    const uint32_t ref_raw = p.witness_refs[w];
    const bool canon = (ref_raw & REF_CANON) != 0;  // the slot holds the canonical integer (representation inference, compile.cc)
    const uint32_t ref = ref_raw & ~REF_CANON;
    const uint32_t n_tiles = (batch + T - 1) / T;
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    const uint32_t slot = (ref & REF_CONST) ? (ref & ~REF_CONST) : p.n_const + ref;  // every tile holds the constants too
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const char* tb = reinterpret_cast<const char*>(wst.base[tile / wst.tiles_per_chunk]) + (uint64_t)(tile % wst.tiles_per_chunk) * tile_bytes;
        const uint4* q = reinterpret_cast<const uint4*>(tb) + (size_t)slot * (2 * T);
        if (TT != 0 && TT <= 4) {  // small tiles: all loads of the slot in flight before the first conversion
            uint4 lo[TT ? TT : 1], hi[TT ? TT : 1];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                lo[t] = q[t];
                hi[t] = q[TT + t];
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const uint32_t set = tile * TT + t;
                if (set >= batch) break;
                const Fr c = pack_form<MONT>(fr_from_u4(lo[t], hi[t]), canon);
                uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
                o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
                o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
            }
        } else {
            for (uint32_t t = 0; t < T; ++t) {
                const uint32_t set = tile * T + t;
                if (set >= batch) break;
                const Fr c = pack_form<MONT>(fr_from_u4(q[t], q[T + t]), canon);
                uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
                o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
                o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
            }
        }
    }
}

// Round 3: one thread = one (witness index, set) pair, the T sets of a slot in ADJACENT lanes: a wave's load instruction
// touches 64 / T slots, each in runs of 16 T contiguous bytes (the thread-per-slot shape above makes every lane fetch a
// different line: 64 lines per instruction at T = 4), and a wave's stores are 64 / T consecutive rows of 32 bytes per set.
template <bool MONT, int TT>
__global__ __launch_bounds__(256) void pack_kernel_v3(ProgramDev p, WsTable wst, uint4* __restrict__ out, uint32_t batch) {
    constexpr uint32_t T = TT, WPB = 256u / T;
    const uint32_t t = threadIdx.x % T, w = blockIdx.x * WPB + threadIdx.x / T;
    if (w >= p.n_witness) return;
    const uint32_t ref_raw = p.witness_refs[w];
    const bool canon = (ref_raw & REF_CANON) != 0;
    const uint32_t ref = ref_raw & ~REF_CANON;
    const uint32_t n_tiles = (batch + T - 1) / T;
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    const uint32_t slot = (ref & REF_CONST) ? (ref & ~REF_CONST) : p.n_const + ref;
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const uint32_t set = tile * T + t;
        if (set >= batch) continue;
        const char* tb = reinterpret_cast<const char*>(wst.base[tile / wst.tiles_per_chunk]) + (uint64_t)(tile % wst.tiles_per_chunk) * tile_bytes;
        const uint4* q = reinterpret_cast<const uint4*>(tb) + (size_t)slot * (2 * T);
        const Fr c = pack_form<MONT>(fr_from_u4(q[t], q[T + t]), canon);
        uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
        o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
        o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
    }
}

// (round-1 shape: block = 64 witness indices x min(T, 4) sets of one tile; what tiles wider than 4 sets use)
template <bool MONT>
__global__ __launch_bounds__(256) void pack_kernel_v1(ProgramDev p, WsTable wst, uint4* __restrict__ out, uint32_t batch, uint32_t T) {
    const uint32_t w = blockIdx.x * 64u + threadIdx.x;
    if (w >= p.n_witness) return;
    const uint32_t ref_raw = p.witness_refs[w];
    const bool canon = (ref_raw & REF_CANON) != 0;
    const uint32_t ref = ref_raw & ~REF_CANON;
    const uint32_t n_tiles = (batch + T - 1) / T;
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    const uint32_t slot = (ref & REF_CONST) ? (ref & ~REF_CONST) : p.n_const + ref;
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const char* tb = reinterpret_cast<const char*>(wst.base[tile / wst.tiles_per_chunk]) + (uint64_t)(tile % wst.tiles_per_chunk) * tile_bytes;
        const uint4* q0 = reinterpret_cast<const uint4*>(tb) + (size_t)slot * (2 * T);
        for (uint32_t t = threadIdx.y; t < T; t += blockDim.y) {
            const uint32_t set = tile * T + t;
            if (set >= batch) break;
            const uint4* q = q0 + t;
            const Fr c = pack_form<MONT>(fr_from_u4(q[0], q[T]), canon);
            uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
            o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
            o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
        }
    }
}

// ---- launchers (called from pipeline.cc) -----------------------------------------------------------
hipError_t launch_interp(uint32_t T, uint32_t W, uint32_t pack, uint32_t n_div_requests, const uint32_t* div_lanes, const ProgramDev& p,
                         const WsTable& wst, const void* inputs, uint32_t* status, uint32_t batch, hipStream_t stream, unsigned long long* prof) {
    const uint32_t tiles = (batch + T - 1) / T, nw = (W ? W : 1u) * pack, ns = p.n_streams ? p.n_streams : 1u;
    if (nw == 0 || nw % ns != 0 || (ns > 1 && W > 1)) return hipErrorInvalidValue;
    const uint32_t tiles_per_wg = nw / ns;
    dim3 grid((tiles + tiles_per_wg - 1) / tiles_per_wg), block((W ? (W + 1) * pack : pack) * 64);
    const uint4* in = (const uint4*)inputs;
    InterpDims dims{p.n_bundles, p.n_slots, p.n_inputs, batch, p.n_const, n_div_requests, p.trash_off, div_lanes, ns, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        dims.stream_first[s] = p.stream_first[s];
        dims.stream_count[s] = p.stream_count[s];
        dims.stream_div_requests[s] = p.stream_div_requests[s];
        dims.stream_cref_first[s] = p.stream_cref_first[s];
    }
    if (ns == 1) {
        dims.stream_first[0] = 0;
        dims.stream_count[0] = p.n_bundles;
        dims.stream_div_requests[0] = n_div_requests;
    }
    const uint4* recs = reinterpret_cast<const uint4*>(p.recs);
    const uint32_t mode = p.has_fused;  // 0, 1: fused narrow bundles, 2: scan bundles (validate_program: never both), 3: scan bundles with the wide-register kinds
    static_assert(SCAN_MAX_T == COOP_FUSE_MAX_T, "the instances with a MODE are made for tile widths up to COOP_FUSE_MAX_T");
    // the instances with a MODE (fused narrow bundles; scan / convolution / canonical-product bundles) exist for programs with no or one
    // divider wave per interpreter (compile.cc compiles nothing else, validate_program rejects it)
    if (mode > 3 || (mode && (T > COOP_FUSE_MAX_T || W > 1))) return hipErrorInvalidValue;
#define CWC_LAUNCH3(TT, PP, WW, KK)                                                                                                     \
    do {                                                                                                                                \
        if constexpr ((uint32_t)(TT) <= COOP_FUSE_MAX_T && (WW) <= 1) {                                                                 \
            if (mode == 3) interp_kernel<TT, PP, WW, KK, 3><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof); \
            else if (mode == 2) interp_kernel<TT, PP, WW, KK, 2><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof); \
            else if (mode == 1) interp_kernel<TT, PP, WW, KK, 1><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof); \
            else interp_kernel<TT, PP, WW, KK, 0><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof);     \
        } else {                                                                                                                        \
            interp_kernel<TT, PP, WW, KK, 0><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof);          \
        }                                                                                                                               \
    } while (0)
#define CWC_LAUNCH2(TT, PP)                                  \
    if (W == 0 && pack == 1) CWC_LAUNCH3(TT, PP, 0, 1);      \
    else if (W == 0 && pack == 4) CWC_LAUNCH3(TT, PP, 0, 4); \
    else if (W == 1 && pack == 1) CWC_LAUNCH3(TT, PP, 1, 1); \
    else if (W == 1 && pack == 2) CWC_LAUNCH3(TT, PP, 1, 2); \
    else if (W == 1 && pack == 4) CWC_LAUNCH3(TT, PP, 1, 4); \
    else if (W == 3 && pack == 1) CWC_LAUNCH3(TT, PP, 3, 1); \
    else if (W == 4 && pack == 1) CWC_LAUNCH3(TT, PP, 4, 1); \
    else return hipErrorInvalidValue;
    // The stamped (PROF) instances exist in the diagnostic library only (make diag: -DCWC_DIAG, libcircom_witnesscalc_amd_diag.so,
    // what tools/gpu_classprof.py and tools/gpu_calibrate.py load): they doubled the product's code object and its load time.
#ifdef CWC_DIAG
#define CWC_LAUNCH_PROF(TT, WW, KK) CWC_LAUNCH3(TT, true, WW, KK)
#define CWC_LAUNCH2_PROF(TT) CWC_LAUNCH2(TT, true)
#else
#define CWC_LAUNCH_PROF(TT, WW, KK) return hipErrorNotSupported
#define CWC_LAUNCH2_PROF(TT) return hipErrorNotSupported;
#endif
#define CWC_LAUNCH(TT)                                    \
    case TT:                                              \
        if (prof) { CWC_LAUNCH2_PROF(TT) } else { CWC_LAUNCH2(TT, false) } \
        break;
    switch (T) {
        CWC_LAUNCH(1) CWC_LAUNCH(2) CWC_LAUNCH(4) CWC_LAUNCH(8) CWC_LAUNCH(16) CWC_LAUNCH(32)
        case 64:
            if (W != 0) return hipErrorInvalidValue;
            if (pack == 4) { if (prof) { CWC_LAUNCH_PROF(64, 0, 4); } else CWC_LAUNCH3(64, false, 0, 4); }
            else if (pack == 1) { if (prof) { CWC_LAUNCH_PROF(64, 0, 1); } else CWC_LAUNCH3(64, false, 0, 1); }
            else return hipErrorInvalidValue;
            break;
        default: return hipErrorInvalidValue;
    }
#undef CWC_LAUNCH_PROF
#undef CWC_LAUNCH2_PROF
#undef CWC_LAUNCH
#undef CWC_LAUNCH2
#undef CWC_LAUNCH3
    return hipGetLastError();
}

// The content hash of this file and of what it includes, as the Makefile computed it when this compile started
// (build/kernels.srchash): tests/test_host_formats.py compares it with the sources in the tree -- an object that was built
// from something else (a source restored while its compile was running) does not pass for the current kernels again.
#ifndef CWC_KSRC_HASH
#define CWC_KSRC_HASH "unstamped"
#endif
extern "C" const char* gwb_kernel_source_hash() { return CWC_KSRC_HASH; }
// 1: this code object has the stamped (PROF) interpreter instances -- the diagnostic library (make diag)
extern "C" int gwb_kernels_have_diagnostics() {
#ifdef CWC_DIAG
    return 1;
#else
    return 0;
#endif
}

// An empty kernel of this code object: its first launch makes the runtime load the object (every interpreter instance) --
// the single-shot entry point does that on a thread of its own while the host parses and compiles (pipeline.cc warm_device).
__global__ void warm_kernel() {}
hipError_t launch_warm(hipStream_t stream) {
    warm_kernel<<<1, 64, 0, stream>>>();
    return hipGetLastError();
}

hipError_t launch_fill_consts(uint32_t T, const ProgramDev& p, const WsTable& wst, uint32_t n_tiles, hipStream_t stream) {
    if (n_tiles == 0 || p.n_const == 0) return hipSuccess;
    dim3 grid((p.n_const * 2u * T + 255u) / 256u, n_tiles < 16384u ? n_tiles : 16384u), block(256);
    fill_consts_kernel<<<grid, block, 0, stream>>>(p, wst, n_tiles, T);
    return hipGetLastError();
}

hipError_t launch_pack(uint32_t T, const ProgramDev& p, const WsTable& wst, void* out, uint32_t batch, hipStream_t stream, bool montgomery) {
    if (p.n_witness == 0 || batch == 0) return hipSuccess;
    const uint32_t n_tiles = (batch + T - 1) / T;
    static const bool v1 = getenv("CWC_PACK_V1") != nullptr;
    if (v1 || T > 4) {  // (tiles of 8 sets and more: a thread walking all sets of its slot serialises 8..64 conversions -- 22.6 ms
                        // against 9.5 ms for the 8192-set pack at T = 8; the sets of a slot stay spread over the waves of a block)
        dim3 grid((p.n_witness + 63) / 64, n_tiles < 32768u ? n_tiles : 32768u), block(64, T < 4 ? T : 4);
        if (montgomery) pack_kernel_v1<true><<<grid, block, 0, stream>>>(p, wst, (uint4*)out, batch, T);
        else pack_kernel_v1<false><<<grid, block, 0, stream>>>(p, wst, (uint4*)out, batch, T);
        return hipGetLastError();
    }
    static const int pack_shape = getenv("CWC_PACK") ? atoi(getenv("CWC_PACK")) : 3;  // (2: one thread per slot, the round-2 shape, for A/B)
    if (pack_shape == 3 && T >= 2) {  // (T = 1: the two shapes are the same kernel)
        dim3 grid3((p.n_witness + 256 / T - 1) / (256 / T), n_tiles < 32768u ? n_tiles : 32768u), block3(256);
#define CWC_PACK3(MM, TT) pack_kernel_v3<MM, TT><<<grid3, block3, 0, stream>>>(p, wst, (uint4*)out, batch)
        if (montgomery) {
            if (T == 2) CWC_PACK3(true, 2); else CWC_PACK3(true, 4);
        } else {
            if (T == 2) CWC_PACK3(false, 2); else CWC_PACK3(false, 4);
        }
#undef CWC_PACK3
        return hipGetLastError();
    }
    dim3 grid((p.n_witness + 255) / 256, n_tiles < 32768u ? n_tiles : 32768u), block(256);
#define CWC_PACK(MM, TT) pack_kernel<MM, TT><<<grid, block, 0, stream>>>(p, wst, (uint4*)out, batch, T)
    if (montgomery) {
        if (T == 1) CWC_PACK(true, 1); else if (T == 2) CWC_PACK(true, 2); else CWC_PACK(true, 4);
    } else {
        if (T == 1) CWC_PACK(false, 1); else if (T == 2) CWC_PACK(false, 2); else CWC_PACK(false, 4);
    }
#undef CWC_PACK
    return hipGetLastError();
}

// Diagnostic (bench.py's compute ceiling): every lane of `waves_per_simd` waves on every SIMD runs a dependent chain of
// one-lane Montgomery products; returns nothing, the caller times the launch.
__global__ __launch_bounds__(1024) void modmul_ubench_kernel(uint32_t* sink, uint32_t iters) {
    // (fr_mul: the per-column multiplier, 380 issue slots, no fixed registers -- four waves per SIMD need <= 128 VGPRs;
    // the interpreter's one-block fr_mul_wave, 322 slots, pins v160-v167)
    Fr a = fr_r2(), b = fr_one();
    a.v[0] ^= threadIdx.x + blockIdx.x * 977u;
    for (uint32_t it = 0; it < iters; ++it) {
        a = fr_mul(a, b);
        b = fr_mul(b, a);
    }
    if (a.v[0] == 0x1234567u && b.v[3] == 7u) sink[0] = a.v[1];
}
// The same with the multiplier the interpreter's full-width bundles use (fr_mul_wave: one asm block, 322 issue slots,
// accumulators pinned in v160-v167, so at most two waves per SIMD fit the register file).
__global__ __launch_bounds__(512) void modmul_block_ubench_kernel(uint32_t* sink, uint32_t iters) {
    Fr a = fr_r2(), b = fr_one(), pv = fr_p();
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(pv.v[i]));
    a.v[0] ^= threadIdx.x + blockIdx.x * 977u;
    for (uint32_t it = 0; it < iters; ++it) {
        a = fr_mul_wave(a, b, pv);
        b = fr_mul_wave(b, a, pv);
    }
    if (a.v[0] == 0x1234567u && b.v[3] == 7u) sink[0] = a.v[1];
}
hipError_t launch_modmul_ubench(uint32_t n_cus, uint32_t waves_per_simd, uint32_t iters, uint32_t* sink, hipStream_t stream, bool block_multiplier) {
    if (block_multiplier) {
        if (waves_per_simd > 2) return hipErrorInvalidValue;
        modmul_block_ubench_kernel<<<n_cus, 256 * waves_per_simd, 0, stream>>>(sink, iters);
    } else {
        modmul_ubench_kernel<<<n_cus, 256 * waves_per_simd, 0, stream>>>(sink, iters);
    }
    return hipGetLastError();
}

}  // namespace cwc
