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

// PROF = true is a diagnostic build (gwb_profile_classes): s_memtime stamps around the operand loads, the
// arithmetic and the store of every bundle, summed per bundle class by lane 0 of every 64th tile.  Its
// waits serialise the loop, so read its shares, not its length; no stamp executes in the product kernel.
template <int T, bool PROF>
__global__ __launch_bounds__(64) void interp_kernel(ProgramDev p, uint4* vals, const uint4* __restrict__ inputs,
                                                    uint32_t* __restrict__ status, uint32_t batch,
                                                    unsigned long long* __restrict__ prof) {
    constexpr int G = 64 / T;
    const int lane = (int)threadIdx.x;
    const int t = lane % T;
    const int j = (G == 1) ? 0 : lane / T;
    const uint32_t tile = blockIdx.x;
    const uint32_t set = tile * T + (uint32_t)t;
    const uint32_t set_c = set < batch ? set : batch - 1;  // padded lanes of the last tile re-evaluate a real set
    uint4* tv = vals + (size_t)tile * p.n_slots * (2 * T);
    const uint4* consts = reinterpret_cast<const uint4*>(p.consts);
    const uint4* recs = reinterpret_cast<const uint4*>(p.recs);

    auto load = [&](uint32_t ref) -> Fr {
        const bool k = (ref & REF_CONST) != 0;
        const uint32_t idx = ref & ~REF_CONST;
        const uint4* q = k ? consts + (size_t)idx * 2 : tv + (size_t)idx * (2 * T) + t;
        const uint32_t step = k ? 1u : (uint32_t)T;
        const uint4 lo = q[0], hi = q[step];
        return fr_from_u4(lo, hi);
    };
    uint32_t err_bits = 0;
    unsigned long long pf[C_COUNT][4];
    if (PROF) {
#pragma unroll
        for (int c = 0; c < (int)C_COUNT; ++c) pf[c][0] = pf[c][1] = pf[c][2] = pf[c][3] = 0;
    }
    auto stamp = [&]() -> unsigned long long {
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };

    // Software pipeline: the header and the records of bundle b+1 are fetched while bundle b computes, so
    // that only the operand loads (which depend on earlier stores) sit on the critical path.
    uint32_t h_next = p.n_bundles ? p.hdr[0] : 0u;
    uint4 rec_next = p.n_bundles ? recs[j] : make_uint4(0, 0, 0, 0);
    for (uint32_t b = 0; b < p.n_bundles; ++b) {
        const uint32_t h = h_next;
        const uint4 rec = rec_next;
        {
            const uint32_t nb = b + 1 < p.n_bundles ? b + 1 : b;
            h_next = p.hdr[nb];
            rec_next = recs[(size_t)nb * G + j];
        }
        const uint32_t cls = h & 0xffu;
        const uint32_t cnt = h >> 8;
        const bool active = (uint32_t)j < cnt;  // inactive node slots carry a copy of record 0 (valid operands)
        const uint32_t sub = rec.x;
        Fr r;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (PROF) {
            ts0 = stamp();
            // touch the operands so that their loads complete before ts1
            if (cls != C_INPUT) {
                const Fr a0 = load(rec.z), c0 = load(rec.w);
                asm volatile("" ::"v"(a0.v[0]), "v"(c0.v[7]));
            }
            ts1 = stamp();
        }
        switch (cls) {
            case C_INPUT: {  // graph.rs:376  Fr::new(inputs[i])
                const uint4* q = inputs + ((size_t)set_c * p.n_inputs + rec.z) * 2;
                r = fr_to_mont(fr_from_u4(q[0], q[1]));
                break;
            }
            case C_MUL: {  // graph.rs:105
                const Fr a = load(rec.z), c = load(rec.w);
                r = fr_mul(a, c);
                break;
            }
            case C_LIN: {  // graph.rs:110-111 Add/Sub, :188-194 Neg (= 0 - a)
                const Fr a = load(rec.z), c = load(rec.w);  // Neg records carry b = a
                // one modular addition: x + (+-y), with -y = r - y (0 stays 0)
                const bool is_neg = sub == SUB_NEG;
                const Fr x = u256_select(is_neg, fr_zero(), a);
                const Fr y = u256_select(is_neg, a, c);
                r = fr_add(x, u256_select(sub == OP_ADD, y, fr_neg(y)));
                break;
            }
            case C_DIV: {  // graph.rs:109  b == 0 -> 0 else a / b
                const Fr a = load(rec.z), c = load(rec.w);
                const Fr inv = fr_inv(c);  // safegcd divsteps; inv(0) = 0
                r = u256_select(u256_is_zero(c), fr_zero(), fr_mul(a, inv));
                break;
            }
            case C_CMPZ: {  // graph.rs:122-129 Eq/Neq, :134-135 Land/Lor
                const Fr a = load(rec.z), c = load(rec.w);
                const bool az = u256_is_zero(a), cz = u256_is_zero(c), eq = u256_eq(a, c);
                const bool v = sub == OP_EQ ? eq : sub == OP_NEQ ? !eq : sub == OP_LAND ? (!az && !cz) : (!az || !cz);
                r = u256_select(v, fr_one(), fr_zero());
                break;
            }
            case C_CMPS: {  // graph.rs:130-133 with u_lt/u_gt/u_lte/u_gte :723-769
                const Fr x = fr_from_mont(load(rec.z)), y = fr_from_mont(load(rec.w));
                const bool xn = u256_lt(fr_half(), x), yn = u256_lt(fr_half(), y);
                const bool same = xn == yn;
                const bool lt = same ? u256_lt(x, y) : xn;
                const bool gt = same ? u256_lt(y, x) : yn;
                const bool v = sub == OP_LT ? lt : sub == OP_GT ? gt : sub == OP_LEQ ? !gt : !lt;
                r = u256_select(v, fr_one(), fr_zero());
                break;
            }
            case C_BIT: {  // graph.rs:621-717
                const Fr x = fr_from_mont(load(rec.z)), y = fr_from_mont(load(rec.w));
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
                const Fr x = fr_from_mont(load(rec.z)), y = fr_from_mont(load(rec.w));
                const bool yz = u256_is_zero(y);
                Fr ys = y;
                ys.v[0] |= yz ? 1u : 0u;
                uint32_t top = u256_bitlen(x);
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
            case C_TERN: {  // graph.rs:221-225  a == 0 ? c : b
                const Fr a = load(rec.z), x = load(rec.w), y = load(p.crefs[(size_t)b * G + j]);
                r = u256_select(u256_is_zero(a), y, x);
                break;
            }
            default: r = fr_zero(); break;
        }
        if (PROF) {
            asm volatile("" ::"v"(r.v[0]), "v"(r.v[7]));
            ts2 = stamp();
        }
        if (active) {
            uint4* q = tv + (size_t)rec.y * (2 * T) + t;
            q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
            q[T] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
        }
        // Later bundles read these stores from other lanes of this wave; a wave's vector-memory
        // instructions execute in order, the fence only keeps the compiler from reordering them.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (PROF) {
            const unsigned long long ts3 = stamp();
#pragma unroll
            for (int c = 0; c < (int)C_COUNT; ++c)
                if (cls == (uint32_t)c) {
                    pf[c][0] += ts1 - ts0;
                    pf[c][1] += ts2 - ts1;
                    pf[c][2] += ts3 - ts2;
                    pf[c][3] += 1;
                }
        }
    }
    if (PROF && lane == 0 && (tile % 64u) == 0u) {
#pragma unroll
        for (int c = 0; c < (int)C_COUNT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&prof[c * 4 + q], pf[c][q]);
    }
    if (err_bits && set < batch) atomicOr(&status[set], err_bits);
}

__global__ __launch_bounds__(256) void pack_kernel(ProgramDev p, const uint4* __restrict__ vals, uint4* __restrict__ out,
                                                   uint32_t batch, uint32_t T) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= p.n_witness) return;
    const uint32_t ref = p.witness_refs[w];
    const uint4* consts = reinterpret_cast<const uint4*>(p.consts);
    for (uint32_t set = blockIdx.y; set < batch; set += gridDim.y) {
        Fr v;
        if (ref & REF_CONST) {
            const uint4* q = consts + (size_t)(ref & ~REF_CONST) * 2;
            v = fr_from_u4(q[0], q[1]);
        } else {
            const uint32_t tile = set / T, t = set % T;
            const uint4* q = vals + ((size_t)tile * p.n_slots + ref) * (2 * T) + t;
            v = fr_from_u4(q[0], q[T]);
        }
        const Fr c = fr_from_mont(v);
        uint4* o = out + ((size_t)set * p.n_witness + w) * 2;
        o[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
        o[1] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
    }
}

// ---- launchers (called from runtime.cc) -----------------------------------------------------------
hipError_t launch_interp(uint32_t T, const ProgramDev& p, void* vals, const void* inputs, uint32_t* status,
                         uint32_t batch, hipStream_t stream, unsigned long long* prof) {
    const uint32_t tiles = (batch + T - 1) / T;
    dim3 grid(tiles), block(64);
    uint4* v = (uint4*)vals;
    const uint4* in = (const uint4*)inputs;
#define CWC_LAUNCH(TT)                                                                               \
    case TT:                                                                                         \
        if (prof) interp_kernel<TT, true><<<grid, block, 0, stream>>>(p, v, in, status, batch, prof); \
        else interp_kernel<TT, false><<<grid, block, 0, stream>>>(p, v, in, status, batch, nullptr);  \
        break;
    switch (T) {
        CWC_LAUNCH(1) CWC_LAUNCH(2) CWC_LAUNCH(4) CWC_LAUNCH(8) CWC_LAUNCH(16) CWC_LAUNCH(32) CWC_LAUNCH(64)
        default: return hipErrorInvalidValue;
    }
#undef CWC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_pack(uint32_t T, const ProgramDev& p, const void* vals, void* out, uint32_t batch, hipStream_t stream) {
    if (p.n_witness == 0 || batch == 0) return hipSuccess;
    dim3 grid((p.n_witness + 255) / 256, batch < 32768u ? batch : 32768u), block(256);
    pack_kernel<<<grid, block, 0, stream>>>(p, (const uint4*)vals, (uint4*)out, batch, T);
    return hipGetLastError();
}

}  // namespace cwc
