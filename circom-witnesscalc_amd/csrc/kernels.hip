// HIP kernels of the calc-witness hot path for gfx950 (CDNA4, wave64).
//
//   interp_kernel<T>  -- the graph interpreter: the loop of graph::evaluate (reference
//                        src/graph.rs:372-382) with Operation::eval_fr (:102-144),
//                        UnoOperation::eval_fr (:188-197), TresOperation::eval_fr (:221-225).
//   pack_kernel       -- output gather: out[i] = into_bigint(values[outputs[i]]) as 32-byte LE rows
//                        (src/graph.rs:385-388, src/lib.rs:170-173), i.e. the `.wtns` section-2 body.
//
// One wavefront = one tile of T input sets x G = 64/T node slots.  The bundle class is wave-uniform
// (scalar branch); per-lane sub-ops inside a class are resolved with selects.  Values live in HBM as
// [tile][slot][half][T][16 B]: every global_load/store_dwordx4 of a lane group touches T*16 contiguous
// bytes (1 KiB per wave-instruction at T = 64).  No MFMA: this is 256-bit modular integer arithmetic
// on v_mad_u64_u32.
#include <hip/hip_runtime.h>

#include "fr_gfx950.hpp"
#include "program_dev.h"

namespace cwc {

__device__ __forceinline__ Fr fr_from_u4(const uint4& lo, const uint4& hi) {
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}

// wave-wide OR-reduction of a predicate ("does any lane need the slow path")
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// PROF = true is a diagnostic build (gwb_profile_classes): one s_memtime per bundle, summed per bundle class by
// lane 0 of every 64th tile: [cycles, cycles of bundles with a forwarded operand, such bundles, bundles].
// No stamp executes in the product kernel.
//
// The program arrays are separate `const __restrict__` kernel arguments (not a by-value struct) so that hipcc can
// prove them read-only: the wave-uniform header stream then becomes scalar loads (s_load) instead of vector loads
// + v_readfirstlane, whose early s_waitcnt would expose the latency of the operand prefetch.
struct InterpDims {
    uint32_t n_bundles, n_slots, n_inputs, batch, n_const;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int T, bool PROF>
__global__ __launch_bounds__(64) void interp_kernel(const uint32_t* __restrict__ hdr, const uint4* __restrict__ recs,
                                                    const uint32_t* __restrict__ crefs, InterpDims p, WsTable wst,
                                                    const uint4* __restrict__ inputs, uint32_t* __restrict__ status,
                                                    unsigned long long* __restrict__ prof) {
    constexpr int G = 64 / T;
    const uint32_t batch = p.batch;
    const int lane = (int)threadIdx.x;
    const int t = lane % T;
    const int j = (G == 1) ? 0 : lane / T;
    const uint32_t tile = blockIdx.x;
    const uint32_t set = tile * T + (uint32_t)t;
    const uint32_t set_c = set < batch ? set : batch - 1;  // padded lanes of the last tile re-evaluate a real set
    // One buffer descriptor over the launch's workspace [constants | tiles]; all operand / destination addresses are
    // 32-bit byte offsets into it (host-computed, plus this lane's base for tile-relative ones).
    const uint64_t tile_bytes = ws_tile_bytes(p.n_slots, T);
    const uint32_t chunk = tile / wst.tiles_per_chunk, tile_in_chunk = tile % wst.tiles_per_chunk;
    void* ws = wst.base[chunk];
    const uint64_t ws_bytes = ws_const_bytes(p.n_const, T) + (uint64_t)wst.tiles_per_chunk * tile_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)(uint32_t)ws_bytes, 0x00020000);
    const uint32_t lane_base = (uint32_t)(ws_const_bytes(p.n_const, T) + (uint64_t)tile_in_chunk * tile_bytes) + 16u * (uint32_t)t;
    constexpr int HI = 16 * T;  // byte distance between the two 16-byte halves of a value
    // Result ring (one wave per block): slot (bundle mod RING) holds the 64 lane results of that bundle as
    // [half][lane][16 B], so the wave's ds_write_b128 / ds_read_b128 are conflict-free and any lane can read any
    // other lane's recent result (host-computed byte offset + 16*t).
    __shared__ uint4 ring[RING_BUNDLES * 128];
    const uint32_t lds_t = 16u * (uint32_t)t;

    auto ld = [&](uint32_t off) -> Fr {
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + HI, 0, 0);
        return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    auto opnd_off = [&](uint32_t off, uint32_t ctrl, uint32_t tile_bit) -> uint32_t {
        return off + ((ctrl & tile_bit) ? lane_base : 0u);
    };
    auto ld_ring = [&](uint32_t off) -> Fr {  // off: byte offset of the low half inside the ring
        const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(ring) + off);
        const uint4 lo = q[0], hi = q[RING_HALF_BYTES / 16];
        return fr_from_u4(lo, hi);
    };
    uint32_t err_bits = 0;
    unsigned long long pf[C_COUNT][4];
    if (PROF) {
#pragma unroll
        for (int c = 0; c < (int)C_COUNT; ++c) pf[c][0] = pf[c][1] = pf[c][2] = pf[c][3] = 0;
    }
    unsigned long long t_prev = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long probe[4] = {0, 0, 0, 0};

    // Software pipeline.  While bundle b computes: its own memory operands were fetched during bundle b-1, the
    // memory operands of bundle b+1 and the header + records of bundle b+2 are in flight.  Only forwarded operands
    // (register reads, same lane or ds_bpermute) and the arithmetic sit on the chain between consecutive bundles.
#if defined(CWC_EXP_SECTIONS)  // timing experiment only: where inside an iteration does the time go
    unsigned long long sec[4] = {0, 0, 0, 0};
#define CWC_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define CWC_STAMP(var)
#endif
    const uint32_t NBND = p.n_bundles;
    if (NBND == 0) return;
    auto clampb = [&](uint32_t b) { return b < NBND ? b : NBND - 1; };
    uint32_t h_cur = hdr[0], h_n1 = hdr[clampb(1)];
    uint4 rec_cur = recs[j], rec_n1 = recs[(size_t)clampb(1) * G + j];
    Fr ma_cur = ld(opnd_off(rec_cur.z, rec_cur.x, CTRL_A_TILE) & (((h_cur & HDR_CLASS_MASK) == C_INPUT) ? 0u : ~0u));
    Fr mb_cur = ld(opnd_off(rec_cur.w, rec_cur.x, CTRL_B_TILE));
    Fr prev = fr_zero();
    for (uint32_t b = 0; b < NBND; ++b) {
        CWC_STAMP(st0);
        const uint32_t h = h_cur;
        const uint4 rec = rec_cur;
        const uint32_t ctrl = rec.x;
        // prefetch: memory operands of bundle b+1 (INPUT records carry an input index in .z: fetch offset 0 instead),
        // header / records of bundle b+2
        const uint32_t a_n1 = opnd_off(rec_n1.z, rec_n1.x, CTRL_A_TILE) & (((h_n1 & HDR_CLASS_MASK) == C_INPUT) ? 0u : ~0u);
#if defined(CWC_EXP_NOLOAD)  // timing experiment only (wrong results)
        Fr ma_n1 = fr_one(), mb_n1 = fr_one();
        ma_n1.v[0] ^= a_n1;
        mb_n1.v[0] ^= rec_n1.w;
#else
        const Fr ma_n1 = ld(a_n1);
        const Fr mb_n1 = ld(opnd_off(rec_n1.w, rec_n1.x, CTRL_B_TILE));
#endif
        const uint32_t b2 = clampb(b + 2);
        const uint32_t h_n2 = hdr[b2];
        const uint4 rec_n2 = recs[(size_t)b2 * G + j];

        const uint32_t cls = h & HDR_CLASS_MASK;
        const bool active = (ctrl & CTRL_ACTIVE) != 0;
        const uint32_t sub = (ctrl >> CTRL_SUB_SHIFT) & 0xffu;
        // Operand selection: memory operands were prefetched during the previous bundle; PREV lanes take their own
        // last result; LDS lanes read the result ring (wave-uniform header bits skip what no lane needs).
        const uint32_t asrc = (ctrl >> CTRL_ASRC_SHIFT) & 3u, bsrc = (ctrl >> CTRL_BSRC_SHIFT) & 3u;
        Fr a_op = ma_cur, b_op = mb_cur;
        if (h & HDR_A_LDS) a_op = u256_select(asrc == SRC_LDS, ld_ring(rec.z + lds_t), a_op);
        if (h & HDR_B_LDS) b_op = u256_select(bsrc == SRC_LDS, ld_ring(rec.w + lds_t), b_op);
        if (h & HDR_A_PREV) a_op = u256_select(asrc == SRC_PREV, prev, a_op);
        if (h & HDR_B_PREV) b_op = u256_select(bsrc == SRC_PREV, prev, b_op);
        Fr r;
        if (__builtin_expect(cls == C_MUL, 1)) {  // graph.rs:105
#if defined(CWC_EXP_NOMUL)  // timing experiment only (wrong results)
            r = fr_add(a_op, b_op);
#else
            r = fr_mul(a_op, b_op);
#endif
        } else if (__builtin_expect(cls == C_LIN, 1)) {
            // graph.rs:110-111 Add/Sub; Neg (:188-194) arrives as 0 - a.  Branch-free: a + (+-b) with -b = r - b
            // (b = 0 gives a + r, folded by the conditional subtraction of fr_add).
            Fr nb;
            u256_sub(nb, fr_p(), b_op);
            r = fr_add(a_op, u256_select(sub == OP_ADD, b_op, nb));
        } else
        switch (cls) {
            case C_INPUT: {  // graph.rs:376  Fr::new(inputs[i])
                const uint4* q = inputs + ((size_t)set_c * p.n_inputs + rec.z) * 2;
                r = fr_to_mont(fr_from_u4(q[0], q[1]));
                break;
            }
            case C_DIV: {  // graph.rs:109  b == 0 -> 0 else a / b
#if defined(CWC_EXP_NODIV)  // timing experiment only (wrong results)
                r = fr_add(a_op, b_op);
#else
                const Fr inv = fr_inv(b_op);  // safegcd divsteps; inv(0) = 0
                r = u256_select(u256_is_zero(b_op), fr_zero(), fr_mul(a_op, inv));
#endif
                break;
            }
            case C_CMPZ: {  // graph.rs:122-129 Eq/Neq, :134-135 Land/Lor
                const bool az = u256_is_zero(a_op), cz = u256_is_zero(b_op), eq = u256_eq(a_op, b_op);
                const bool v = sub == OP_EQ ? eq : sub == OP_NEQ ? !eq : sub == OP_LAND ? (!az && !cz) : (!az || !cz);
                r = u256_select(v, fr_one(), fr_zero());
                break;
            }
            case C_CMPS: {  // graph.rs:130-133 with u_lt/u_gt/u_lte/u_gte :723-769
                const Fr x = fr_from_mont(a_op), y = fr_from_mont(b_op);
                const bool xn = u256_lt(fr_half(), x), yn = u256_lt(fr_half(), y);
                const bool same = xn == yn;
                const bool lt = same ? u256_lt(x, y) : xn;
                const bool gt = same ? u256_lt(y, x) : yn;
                const bool v = sub == OP_LT ? lt : sub == OP_GT ? gt : sub == OP_LEQ ? !gt : !lt;
                r = u256_select(v, fr_one(), fr_zero());
                break;
            }
            case C_BIT: {  // graph.rs:621-717
                const Fr x = fr_from_mont(a_op), y = fr_from_mont(b_op);
                uint32_t hi_or = 0;
#pragma unroll
                for (int i = 1; i < 8; ++i) hi_or |= y.v[i];
                const bool big = hi_or != 0 || y.v[0] >= 254u;  // b >= MODULUS_BIT_SIZE -> 0
                const uint32_t n = big ? 0u : y.v[0];
                Fr d;
                if (sub == OP_SHL || sub == OP_SHR) {
                    const Fr sl = u256_shl(x, n), sr = u256_shr(x, n);
                    d = u256_select(sub == OP_SHL, sl, sr);
                    d = u256_select(big, fr_zero(), d);
                    if (sub == OP_SHL && !u256_lt(d, fr_p())) {  // graph.rs:634 unwrap on None
                        if (active) err_bits |= ST_SHL_OVERFLOW;
                        d = fr_zero();
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        d.v[i] = sub == OP_BAND ? (x.v[i] & y.v[i]) : sub == OP_BOR ? (x.v[i] | y.v[i]) : (x.v[i] ^ y.v[i]);
                    Fr dm;
                    const uint32_t br = u256_sub(dm, d, fr_p());  // br == 1 iff d < r
                    if (br == 0) {                                // d >= r: one subtraction (d < 2^254 < 2r)
                        if (u256_is_zero(dm) && active) err_bits |= ST_BITOP_EQ_R;  // d == r: reference panics
                        d = dm;
                    }
                }
                // boolean-valued results (Num2Bits-style Band(x,1)) skip the Montgomery multiplication
                const bool small = (d.v[0] < 2u) && ((d.v[1] | d.v[2] | d.v[3] | d.v[4] | d.v[5] | d.v[6] | d.v[7]) == 0u);
                if (wave_any(!small)) {
                    r = fr_to_mont(d);
                } else {
                    r = u256_select(d.v[0] != 0u, fr_one(), fr_zero());
                }
                break;
            }
            case C_IDIVMOD: {  // graph.rs:112-121
                const Fr x = fr_from_mont(a_op), y = fr_from_mont(b_op);
                const bool yz = u256_is_zero(y);
                Fr ys = y;
                ys.v[0] |= yz ? 1u : 0u;
                // trip count = length of the longest quotient in the wave: bitlen(x) - bitlen(y) + 1 (0 when x < y)
                const uint32_t lx = u256_bitlen(x), ly = u256_bitlen(ys);
                uint32_t top = lx >= ly ? lx - ly + 1u : 0u;
#pragma unroll
                for (int off = 32; off; off >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)top, off);
                    top = top > o ? top : o;
                }
                top = (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
                Fr q, rem;
                u256_divrem(q, rem, x, ys, top);
                const Fr d = u256_select(yz, fr_zero(), u256_select(sub == OP_IDIV, q, rem));
                r = fr_to_mont(d);
                break;
            }
            case C_TERN: {  // graph.rs:221-225  a == 0 ? c : b ; the third operand is always a memory reference
                const uint32_t cr = crefs[(size_t)b * G + j];
                const Fr y = ld((cr & ~CREF_TILE) + ((cr & CREF_TILE) ? lane_base : 0u));
                r = u256_select(u256_is_zero(a_op), y, b_op);
                break;
            }
            default: r = fr_zero(); break;
        }
        CWC_STAMP(st2);
        {   // unconditional store (the host points values without a slot and inactive node slots at the tile's trash
            // slot): a fixed number of stores per bundle lets the waitcnt pass count them instead of draining
            const uint32_t doff = rec.y + lane_base;
#if !defined(CWC_EXP_NOSTORE)  // timing experiment only (wrong results)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{r.v[0], r.v[1], r.v[2], r.v[3]}, rsrc, (int)doff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{r.v[4], r.v[5], r.v[6], r.v[7]}, rsrc, (int)doff + HI, 0, 0);
#else
            asm volatile("" ::"v"(doff), "v"(r.v[0]), "v"(r.v[7]));
#endif
        }
        {   // publish the results of this bundle in the ring (read by later bundles of this wave, in order)
            uint4* q = ring + (b & (RING_BUNDLES - 1u)) * 128u + (uint32_t)lane;
            q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
            q[64] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
        }
        prev = r;
        h_cur = h_n1; rec_cur = rec_n1; ma_cur = ma_n1; mb_cur = mb_n1;
        h_n1 = h_n2; rec_n1 = rec_n2;
        // Later bundles read these stores from other lanes of this wave; a wave's vector-memory instructions
        // execute in order, the fence only keeps the compiler from reordering them.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#if defined(CWC_EXP_SECTIONS)
        {
            CWC_STAMP(st3);
            asm volatile("" ::"v"(ma_cur.v[0]), "v"(mb_cur.v[7]), "v"(rec_cur.x), "v"(rec_n1.w), "s"(h_cur), "s"(h_n1));
            sec[0] += st1 - st0; sec[1] += st2 - st1; sec[2] += st3 - st2; sec[3] += 1;
        }
#endif
        if (PROF && (b & 63u) == 63u) {
            // latency probes (diagnostic only): a constant-table line, and the slot this bundle has just stored
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long q0 = __builtin_amdgcn_s_memtime();
            const Fr pc = ld(0);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(pc.v[0]), "v"(pc.v[7]) : "memory");
            const unsigned long long q1 = __builtin_amdgcn_s_memtime();
            const Fr ps = ld(rec.y + lane_base);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(ps.v[0]), "v"(ps.v[7]) : "memory");
            const unsigned long long q2 = __builtin_amdgcn_s_memtime();
            const uint4 pr = recs[(size_t)clampb(b + 40) * G + j];
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(pr.x), "v"(pr.w) : "memory");
            const unsigned long long q3 = __builtin_amdgcn_s_memtime();
            probe[0] += q1 - q0; probe[1] += q2 - q1; probe[2] += q3 - q2; probe[3] += 1;
            t_prev += q3 - q0;  // keep the probes out of the per-class figures
        }
        if (PROF) {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            const bool fwd = (h & (HDR_A_PREV | HDR_A_LDS | HDR_B_PREV | HDR_B_LDS)) != 0;
#pragma unroll
            for (int c = 0; c < (int)C_COUNT; ++c)
                if (cls == (uint32_t)c) {
                    pf[c][0] += t_now - t_prev;              // cycles of this bundle in the pipelined loop
                    pf[c][1] += fwd ? (t_now - t_prev) : 0;  // ... of which bundles with a forwarded operand
                    pf[c][2] += fwd ? 1 : 0;
                    pf[c][3] += 1;
                }
            t_prev = t_now;
        }
    }
    if (PROF && lane == 0 && (tile % 64u) == 0u) {
#pragma unroll
        for (int c = 0; c < (int)C_COUNT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&prof[c * 4 + q], pf[c][q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&prof[36 + q], probe[q]);
    }
#if defined(CWC_EXP_SECTIONS)
    if (tile == 0 && lane == 0) {
        for (int q = 0; q < 4; ++q) { status[2 * q] = (uint32_t)sec[q]; status[2 * q + 1] = (uint32_t)(sec[q] >> 32); }
        return;
    }
#endif
    if (err_bits && set < batch) atomicOr(&status[set], err_bits);
}

// Block = 64 witness indices x min(T, 4) sets of ONE tile: the waves of a block read the same 128-byte lines of the
// tile's slots ([slot][half][T][16 B]) at the same time, so each line comes from HBM once instead of once per set.
__global__ __launch_bounds__(256) void pack_kernel(ProgramDev p, WsTable wst, uint4* __restrict__ out, uint32_t batch, uint32_t T) {
    const uint32_t w = blockIdx.x * 64u + threadIdx.x;
    if (w >= p.n_witness) return;
    const uint32_t ref = p.witness_refs[w];
    const uint32_t n_tiles = (batch + T - 1) / T;
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const uint4* ws = reinterpret_cast<const uint4*>(wst.base[tile / wst.tiles_per_chunk]);
        const uint4* q0 = (ref & REF_CONST)
                              ? ws + (size_t)(ref & ~REF_CONST) * (2 * T)  // constants: head of the chunk's workspace
                              : ws + ws_const_bytes(p.n_const, T) / 16 +
                                    ((size_t)(tile % wst.tiles_per_chunk) * (p.n_slots + 1) + ref) * (2 * T);
        for (uint32_t t = threadIdx.y; t < T; t += blockDim.y) {
            const uint32_t set = tile * T + t;
            if (set >= batch) break;
            const uint4* q = (ref & REF_CONST) ? q0 : q0 + t;
            const Fr c = fr_from_mont(fr_from_u4(q[0], q[T]));
            uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
            o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
            o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
        }
    }
}

// ---- launchers (called from runtime.cc) -----------------------------------------------------------
hipError_t launch_interp(uint32_t T, const ProgramDev& p, const WsTable& wst, const void* inputs, uint32_t* status,
                         uint32_t batch, hipStream_t stream, unsigned long long* prof) {
    const uint32_t tiles = (batch + T - 1) / T;
    dim3 grid(tiles), block(64);
    const uint4* in = (const uint4*)inputs;
    const InterpDims dims{p.n_bundles, p.n_slots, p.n_inputs, batch, p.n_const};
    const uint4* recs = reinterpret_cast<const uint4*>(p.recs);
#define CWC_LAUNCH(TT)                                                                                              \
    case TT:                                                                                                        \
        if (prof) interp_kernel<TT, true><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, prof);    \
        else interp_kernel<TT, false><<<grid, block, 0, stream>>>(p.hdr, recs, p.crefs, dims, wst, in, status, nullptr);   \
        break;
    switch (T) {
        CWC_LAUNCH(1) CWC_LAUNCH(2) CWC_LAUNCH(4) CWC_LAUNCH(8) CWC_LAUNCH(16) CWC_LAUNCH(32) CWC_LAUNCH(64)
        default: return hipErrorInvalidValue;
    }
#undef CWC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_pack(uint32_t T, const ProgramDev& p, const WsTable& wst, void* out, uint32_t batch, hipStream_t stream) {
    if (p.n_witness == 0 || batch == 0) return hipSuccess;
    const uint32_t n_tiles = (batch + T - 1) / T;
    dim3 grid((p.n_witness + 63) / 64, n_tiles < 32768u ? n_tiles : 32768u), block(64, T < 4 ? T : 4);
    pack_kernel<<<grid, block, 0, stream>>>(p, wst, (uint4*)out, batch, T);
    return hipGetLastError();
}

}  // namespace cwc
