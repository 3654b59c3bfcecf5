// Device-visible part of the compiled program format (shared by the host compiler and the HIP kernels).
#pragma once
#include <stdint.h>

#include "ops.h"

#if defined(__HIPCC__)
#define CWC_HD __host__ __device__ static inline
#define CWC_HDC __host__ __device__ static constexpr
#else
#define CWC_HD static inline
#define CWC_HDC static constexpr
#endif

namespace cwc {

// bundle classes (wave-uniform: one scalar branch per bundle, no divergence on op type)
enum BundleClass : uint32_t {
    C_INPUT = 0,    // dst = to_mont(inputs[set][a])                       graph.rs:376
    C_MUL = 1,      // graph.rs:105; free node slots may carry Add/Sub nodes (header bits say so), see compile.cc
    C_LIN = 2,      // Add / Sub / Neg (as 0 - a)                          graph.rs:110-111, 188-194
    C_DIV = 3,      // graph.rs:109
    C_CMPZ = 4,     // Eq / Neq / Land / Lor (no representation change)    graph.rs:122-129, 134-135
    C_CMPS = 5,     // Lt / Gt / Leq / Geq (signed compare on canonical)   graph.rs:130-133, 720-769
    C_BIT = 6,      // Shl / Shr / Bor / Band / Bxor                       graph.rs:621-717
    C_IDIVMOD = 7,  // Idiv / Mod                                          graph.rs:112-121
    C_TERN = 8,     // TernCond                                            graph.rs:221-225
    // programs compiled for the asynchronous divider (one extra wavefront per tile): a division is split into
    C_DIVREQ = 9,   // ... handing the operands to the divider wave (no result), and
    C_DIVGET = 10,  // ... collecting the quotients a while later (same node slots as the request)
    // narrow multiplication bundle (at most coop_nodes(T) nodes, tile widths up to COOP_MAX_T): four adjacent lanes share
    // one product -- lane 4v + q holds limbs 2q, 2q+1 of b and of the modulus, all of a -- which takes 148 issue slots
    // instead of 322 (fr_mul_coop4_gfx950.inc); value v = t + T * j sits in ring / stage cell v as in every other bundle.
    // The node's record is written four times (record positions 4j .. 4j+3): lane l takes record l / T like everywhere.
    C_MULQ = 11,    // graph.rs:105
    // programs of several streams (wavefronts of one tile with their own bundle sequences): a bundle without nodes in which
    // the wave posts -- every result of its earlier bundles is in memory -- and / or waits: a stream other than 0 for stream
    // 0's next post, stream 0 for the next post of every other stream (header bits HDR_POST / HDR_WAIT)
    C_SYNC = 12,
    // fused narrow bundle (round 3, tile widths up to COOP_FUSE_MAX_T): every node is a short dependent sequence that stays
    // in the registers of its four lanes -- stage 1 a product a * b (lane layout of C_MULQ), stage 2 `op2` with a third
    // operand x2 (another product, or + / - ), stage 3 `op3` with a fourth operand x3 (+ / -): (s * s) * m + c is one node
    // where the unfused graph needs three bundles on its critical chain.  The compiler makes such nodes from Mul / Add / Sub
    // nodes of the graph (exact in the field, graph.rs:105, 110-111; compile.cc fuse_narrow_chains).  Records: positions
    // 4j, 4j+2 = {a_off, b_off, dst | ACTIVE | op2, a_lds | b_lds << 16}, positions 4j+1, 4j+3 = {x2_off, x3_off,
    // trash | op3, x2_lds | x3_lds << 16}; header bits HDR_F_* say which stages any node of the bundle has.  Plain
    // multiplications ride as nodes with op2 = op3 = none.
    C_MULF = 13,
    // scan bundle (round 4, tile widths up to SCAN_MAX_T): the steps of a serial limb recurrence -- the carry chain of a
    // multi-limb sum, the remainder chain of a long division by one limb -- in consecutive PAIRS of node slots, run by a
    // loop of `iterations` rounds inside ONE bundle: every round each pair computes its step from its own operand x and the
    // accumulator that arrives from the pair in front of it (lane shift), so after round s the first s + 1 steps of every
    // chain segment are final (a systolic fixed point: a finished step recomputes the same values).  The unfused graph pays
    // two or three bundles of ~600 cycles of front end for each step.
    //   CARRY  t = x + acc (graph.rs:110);  out = t & (2^n - 1) (Band, :674-687; Mod 2^n, :117-121);  acc' = t >> n (Shr, :637-672; Idiv 2^n, :112-116)
    //   DIV    t = acc * 2^k + x (:105, :110);  out = t idiv d (:112-116);  acc' = t mod d (:117-121)   [d == 0 -> both 0]
    // all on canonical integers with the reference's modular semantics (t is reduced mod r; a wave-uniform test picks the
    // straight limb-sized path or the general 256-bit one).  Position 2p = the step's OUT record {x_off, acc_in_off (chain
    // start only), dst | ACTIVE | START?, x_lds | acc_lds << 16}, position 2p + 1 = its ACC record {d_off, (2^k)_Montgomery_off,
    // dst | ACTIVE | ROLE_ACC | START?, d_lds | B_lds << 16} (CARRY: both unused).  Header: bit 11 kind (0 CARRY, 1 DIV), bits
    // 19-26 the shift n / k (< 254), bits 27-31 iterations - 1 (the longest chain segment of the bundle), count = positions in use.
    //   BORROW (header bit 13, round 5)  the borrow chain of a register-wise subtraction x - y (bigint long_sub): s = y + bin (graph.rs:110);
    //          c = x >= s (:133, :756-769); out = c ? x - y - bin : x - y - bin + 2^n (:110-111, :221-225); acc' = c ? 0 : 1.  ACC record {y_off, -}.
    //   LEX    (header bit 14, round 5)  a comparison decided by the most significant differing register (bigint long_gt):
    //          acc' = x > y ? KG : x < y ? KL : acc (:130-131 with u_gt / u_lt :723-755, :221-225); KG, KL = header bits 15, 16; out unused; shift field 1: the chain's bits are Montgomery-form booleans (0 / 2^256 mod r), 0: the integers 0 / 1.
    //          ACC record {y_off, -}.  The accumulators of both kinds are single bits (a chain starts from 0, 1 or another chain's bit).
    //   CONV (header bit 12, round 4): the 2k - 1 columns of a k x k schoolbook limb product, position c = column c: record
    //   {x_off, y_off, dst | ACTIVE, x_lds | y_lds << 16} names x_c and y_c for c < k (any value above); out_c = sum_{i + j = c}
    //   x_i y_j in the field on canonical integers (graph.rs:105, 110).  Header bits 27-31: k - 1.
    C_SCAN = 14,
    C_COUNT = 15
};
static const uint32_t COOP_LANES = 4, COOP_MAX_T = 4, COOP_FUSE_MAX_T = 2, SCAN_MAX_T = 2;
static const uint32_t HDR_SCAN_DIV = 1u << 11, HDR_SCAN_CONV = 1u << 12;
// one-bit recurrences (round 5): bit 13 BORROW, bit 14 LEX with its two result bits KG (bit 15), KL (bit 16)
// selection bundles (round 5): both of the bits below set; the shift field holds the comparison -- SelCode, + SEL_OUT_MONT when the OUT value
// (the comparison's boolean) is wanted as a Montgomery-form boolean.  OUT record {a, b}: the comparison's operands (SEL_NEZ: a alone, the
// condition); ACC record {p, q}: the arms; out = a <cmp> b (graph.rs:130-133; SEL_NEZ: a != 0), acc = out ? p : q (graph.rs:221-225).
enum SelCode : uint32_t { SEL_LT = 1, SEL_GT = 2, SEL_LEQ = 3, SEL_GEQ = 4, SEL_NEZ = 5, SEL_OUT_MONT = 8 };
static const uint32_t HDR_SCAN_BORROW = 1u << 13, HDR_SCAN_LEX = 1u << 14, HDR_SCAN_KG = 1u << 15, HDR_SCAN_KL = 1u << 16;
static const int HDR_SCAN_SHIFT_SHIFT = 19, HDR_SCAN_ITER_SHIFT = 27;
static const uint32_t SCAN_ROLE_ACC = 1u, SCAN_START = 2u;  // sub-op bits of a scan record
// stage codes of a fused node (C_MULF): op2 in the low three bits of the main record's ctrl, op3 in the extra record's
enum FusedOp : uint32_t { FOP_NONE = 0, FOP_MUL = 1, FOP_ADD = 2, FOP_SUB = 3 /* acc - x */, FOP_RSUB = 4 /* x - acc */ };
static const uint32_t HDR_F_S2MUL = 1u << 11, HDR_F_S2LIN = 1u << 12, HDR_F_S3LIN = 1u << 13;  // C_MULF: stages present in the bundle
CWC_HDC uint32_t coop_nodes(uint32_t T) { return T <= COOP_MAX_T ? 64u / (COOP_LANES * T) : 0u; }  // nodes of a C_MULQ bundle

// Program format v4 -- every operand of a bundle is read from the wave's LDS at a host-computed address, and the
// interpreter spends no vector instruction on decoding or address arithmetic beyond adding the lane's 16*t.
//
// hdr[bundle] (wave-uniform, fetched with scalar loads): bits 0-3 class | bits 4-10 node count |
//   C_LIN and C_MUL: bit 11 some lane subtracts, bit 12 some lane adds (C_LIN: uniform bundles take a shorter path;
//   C_MUL: the bundle also carries linear nodes in otherwise idle node slots)
//
// LDS of a wave (one wave per workgroup), byte addresses:
//   RING   [LDS_RING_OFF  + (bundle mod RING_BUNDLES) * 2 KiB]  results of the last RING_BUNDLES bundles,
//          [half][64 lanes][16 B]: any lane's recent result, whatever lane produced it
//   STAGE  [LDS_STAGE_OFF + (bundle mod OPND_AHEAD) * 4 KiB]    memory operands of a bundle, fetched OPND_AHEAD bundles
//          ahead by direct-to-LDS buffer loads (no VGPRs, no waiting): [a lo | a hi | b lo | b hi][64 lanes][16 B]
//   REC    [LDS_REC_OFF   + (bundle mod REC_AHEAD) * 1 KiB]     the bundle's records, fetched REC_AHEAD bundles ahead
// operand sources: a value produced at most RING_BUNDLES bundles ago comes from the RING, everything else (older
// values, constants, every third operand) from its slot in the tile (MEM, staged through STAGE).
//
// rec[bundle][node slot] = {a_off, b_off, dst | ctrl, a_lds | b_lds << 16}
//   a_off/b_off  tile-relative byte offset of the operand's slot (the lane adds 16*t) for the staging load; operands
//                that come from the RING point at the zero constant (the load still happens, its result is unused)
//   dst          tile-relative byte offset of the destination slot (trash slot when the value needs none);
//                low 4 bits = ctrl: bits 0-2 sub-op within the class (SubOp), bit 3 active
//   a_lds/b_lds  LDS byte address of the operand's low half for t = 0 (the lane adds 16*t): its STAGE cell or a RING cell
// crefs[row][node slot]: TernCond third operand (tile-relative byte offset, always MEM, loaded in place);
//                        INPUT bundles: index into the set's input row.  One row per C_TERN / C_INPUT bundle in bundle
//                        order; a wave counts the rows it has used (its stream starts at stream_cref_first).
// A tile-relative offset beyond every tile (round 4): a staging load from there lands zeros in LDS through the buffer range
// check and moves nothing through the memory system, a store there is dropped.  What the operands of ring-forwarded and
// idle node slots "load" and where results without a slot "go" -- before, a zero constant's slot and the tile's trash slot:
// real requests, 6 KiB per bundle and wave, most of an interpreter launch's L2 traffic.
static const uint32_t OFF_NOWHERE = 0xffff0000u;
static const uint32_t HDR_CLASS_MASK = 0xfu;
static const int HDR_COUNT_SHIFT = 4;
static const uint32_t HDR_LIN_SUB = 1u << 11, HDR_LIN_ADD = 1u << 12, HDR_BITX_ALL = 1u << 13;
// C_MUL (the same bit as HDR_BITX_ALL): every node multiplies two canonical integers and keeps the canonical product
// (a * b mod r, graph.rs:105, without a trip through Montgomery form when the factors are limb-sized); MODE 2 instances only
static const uint32_t HDR_MUL_CC = 1u << 13;
// C_BIT (the same two bits): every node of the bundle is a Shr or a Band (SHR: some shift; BAND alone: none does) -- limb arithmetic (Idiv / Mod by 2^n after the
// compiler's strength reduction, masks) takes a straight path instead of the per-lane select over all five operations
static const uint32_t HDR_BIT_ALL_SHR = 1u << 11, HDR_BIT_ALL_BAND = 1u << 12;
// C_BIT / C_IDIVMOD / C_CMPS (integer operations on canonical values): every lane's first / second operand arrives as a
// canonical integer (a value the compiler keeps in that form, or the canonical copy of a constant) -- the bundle skips
// that operand's conversion out of Montgomery form; OUT: the result stays canonical (these classes and C_CMPZ's booleans).
static const uint32_t HDR_B_CANON = 1u << 14, HDR_A_CANON = 1u << 17, HDR_OUT_CANON = 1u << 18;
// C_SYNC bundles: post and / or wait (see the class).  The loads a wave issues behind its wait are the staging loads of
// the bundle three further on and the third-operand loads of the next bundle.
static const uint32_t HDR_POST = 1u << 15, HDR_WAIT = 1u << 16;
static const uint32_t MAX_STREAMS = 4;
static const uint32_t CTRL_SUB_MASK = 7u, CTRL_ACTIVE = 8u, CTRL_MASK = 15u;
static const uint32_t RING_BUNDLES = 4, OPND_AHEAD = 2, REC_AHEAD = 4;
static const uint32_t RING_SLOT_BYTES = 2048, LDS_HALF_BYTES = 1024, STAGE_BYTES = 4096, REC_BYTES = 1024;
// Asynchronous divider (Program::divider = W > 0): a workgroup is W interpreter wavefronts (one tile each) plus one
// divider wavefront; W = 1 serves the latency regime, W = 4 the throughput regime (the divider packs the requests of
// its four interpreters into the lanes of ONE inversion).  LDS of the workgroup: W interpreter areas, then W mailboxes
// of mbox_lanes(W) lanes -- operand a, operand b as [half][lane][16 B]; the quotient overwrites a -- then the sequence
// words: requests posted by interpreter w (W words), requests served (one word, written by the divider).  One request
// per interpreter is in flight at a time (the compiler emits REQ k, GET k, REQ k+1, ...).
CWC_HDC uint32_t mbox_lanes(uint32_t W) { return W <= 1u ? 64u : 32u; }          // most active lanes of a request
CWC_HDC uint32_t mbox_bytes(uint32_t W) { return mbox_lanes(W) * 64u; }           // per interpreter
CWC_HDC uint32_t lds_area_bytes(uint32_t W, uint32_t lds_bytes) { return (void)W, lds_bytes; }  // per interpreter
static const uint32_t LDS_RING_OFF = 0, LDS_STAGE_OFF = LDS_RING_OFF + RING_BUNDLES * RING_SLOT_BYTES,
                      LDS_REC_OFF = LDS_STAGE_OFF + OPND_AHEAD * STAGE_BYTES, LDS_BYTES = LDS_REC_OFF + REC_AHEAD * REC_BYTES;
// sub-ops inside a class (3 bits)
enum SubOp : uint32_t {
    SUB_ADD = 0, SUB_SUB = 1, SUB_MULT = 2,                        // C_LIN (Neg is 0 - a); C_MUL lanes: SUB_MULT or a linear rider
    SUB_EQ = 0, SUB_NEQ = 1, SUB_LAND = 2, SUB_LOR = 3,            // C_CMPZ
    SUB_LT = 0, SUB_GT = 1, SUB_LEQ = 2, SUB_GEQ = 3,              // C_CMPS
    SUB_SHL = 0, SUB_SHR = 1, SUB_BOR = 2, SUB_BAND = 3, SUB_BXOR = 4,  // C_BIT
    SUB_BITX = 5,  // C_BIT: (a >> k) & 1, k = b_lds / 16 (no second operand is read); header bit 13: every lane is one
    SUB_IDIV = 0, SUB_MOD = 1,                                     // C_IDIVMOD
};

// witness reference (pack kernel): bit 31 set -> constant table index, else value slot of the tile
static const uint32_t REF_CONST = 0x80000000u;
static const uint32_t REF_CANON = 0x40000000u;  // the slot holds the canonical integer, not the Montgomery form

// per-set status bits written by the kernels (the reference panics in these cases)
enum SetStatus : uint32_t {
    ST_SHL_OVERFLOW = 1u,  // Shl result >= r            (graph.rs:634 from_bigint().unwrap())
    ST_BITOP_EQ_R = 2u,    // Bor/Bxor result == r       (graph.rs:701,716)
};

// Device pointers of an uploaded program.
struct ProgramDev {
    const uint32_t* hdr;           // [n_bundles]
    const uint32_t* recs;          // [n_bundles*G*4], 16-byte aligned records
    const uint32_t* crefs;         // [n_cref_rows*G]: a row per C_INPUT / C_TERN bundle
    const uint32_t* consts;        // [n_const*8] Montgomery form (the last entry is a dummy zero)
    const uint32_t* witness_refs;  // [n_witness]
    const uint32_t* div_lanes;     // [n_div_requests] active lanes (node slots x T) of each division request
    uint32_t n_bundles, n_slots, n_inputs, n_witness, n_const;
    uint32_t trash_off;            // where results without a slot go: OFF_NOWHERE, or (programs compiled with CWC_NOWHERE=0) the tile's trash slot
    uint32_t has_fused;            // 1: the program has C_MULF bundles, 2: C_SCAN bundles, 3: C_SCAN bundles of the wide-register kinds (the interpreter instance with their path is launched)
    uint32_t n_streams, stream_first[4], stream_count[4], stream_div_requests[4], stream_cref_first[4];  // (program.hpp; MAX_STREAMS entries)
};

// A launch covers up to WS_MAX_CHUNKS separately allocated workspaces: tile i lives in chunk i / tiles_per_chunk.
static const uint32_t WS_MAX_CHUNKS = 32;
struct WsTable {
    void* base[WS_MAX_CHUNKS];
    uint32_t tiles_per_chunk;
    uint32_t n_chunks;
};

// Tile geometry shared by host and kernels.  A tile holds the values of T input sets:
//   [n_const constant slots | n_slots value slots | two trash slots | sync slot], slot = 32*T bytes = [half][T][16 B].
// Each tile has its own buffer descriptor (base = the tile, 32-bit tile-relative offsets), and its own copy of the
// constants (written once per workspace by fill_consts_kernel), so that every operand is addressed the same way.  The
// sync slot holds the post counters of the tile's streams (programs of several streams).
CWC_HD uint64_t ws_tile_bytes(uint32_t n_const, uint32_t n_slots, uint32_t T) { return ((uint64_t)n_const + n_slots + 3u) * 32u * T; }
CWC_HD uint64_t ws_sync_offset(uint32_t n_const, uint32_t n_slots, uint32_t T) { return ((uint64_t)n_const + n_slots + 2u) * 32u * T; }

}  // namespace cwc
