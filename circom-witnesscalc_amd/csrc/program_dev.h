// Device-visible part of the compiled program format (shared by the host compiler and the HIP kernels).
#pragma once
#include <stdint.h>

#include "ops.h"

#if defined(__HIPCC__)
#define CWC_HD __host__ __device__ static inline
#else
#define CWC_HD static inline
#endif

namespace cwc {

// bundle classes (wave-uniform: one scalar branch per bundle, no divergence on op type)
enum BundleClass : uint32_t {
    C_INPUT = 0,    // dst = to_mont(inputs[set][a])                       graph.rs:376
    C_MUL = 1,      // graph.rs:105
    C_LIN = 2,      // Add / Sub / Neg                                     graph.rs:110-111, 188-194
    C_DIV = 3,      // graph.rs:109
    C_CMPZ = 4,     // Eq / Neq / Land / Lor (no representation change)    graph.rs:122-129, 134-135
    C_CMPS = 5,     // Lt / Gt / Leq / Geq (signed compare on canonical)   graph.rs:130-133, 720-769
    C_BIT = 6,      // Shl / Shr / Bor / Band / Bxor                       graph.rs:621-717
    C_IDIVMOD = 7,  // Idiv / Mod                                          graph.rs:112-121
    C_TERN = 8,     // TernCond                                            graph.rs:221-225
    C_COUNT = 9
};

// Program format v3 -- laid out so that the interpreter spends (almost) no instructions on decoding.
//
// hdr[bundle] (wave-uniform, fetched with scalar loads):
//   bits 0-3 class | bits 4-10 node count | bit 11 some a operand is PREV | bit 12 some a operand is LDS |
//   bit 13 some b operand is PREV | bit 14 some b operand is LDS
// operand sources (per lane, in ctrl):
//   MEM   constant table or value slot in the workspace (global memory, prefetched one bundle ahead)
//   PREV  the lane's own result of the previous bundle (register, no instruction)
//   LDS   the result ring: every bundle writes its 64 lane results into ring slot (bundle mod RING) of the wave's
//         LDS; any value produced at most RING-1 bundles ago is read back from there, whatever lane produced it
//
// rec[bundle][node slot] = {ctrl, dst, a, b}: dst is a BYTE offset (tile-relative) for the store; a/b are byte offsets
// for raw_buffer_load through one descriptor over the workspace [constant table | tile 0 | tile 1 | ...] (MEM; offsets
// flagged tile-relative get the lane's base added) or byte offsets into the wave's LDS ring (LDS; the lane adds
// 16*t).  ctrl: bit0 a tile-relative, bit1 b tile-relative, bits 2-3 a source, bits 4-5 b source, bits 16-23 DuoOp
// code (or SUB_*), bit 24 active.
static const uint32_t HDR_CLASS_MASK = 0xfu;
static const int HDR_COUNT_SHIFT = 4;
static const uint32_t HDR_A_PREV = 1u << 11, HDR_A_LDS = 1u << 12, HDR_B_PREV = 1u << 13, HDR_B_LDS = 1u << 14;
enum OperandSource : uint32_t { SRC_MEM = 0, SRC_PREV = 1, SRC_LDS = 2 };
static const uint32_t CTRL_A_TILE = 1u << 0, CTRL_B_TILE = 1u << 1, CTRL_ACTIVE = 1u << 24;
static const int CTRL_ASRC_SHIFT = 2, CTRL_BSRC_SHIFT = 4, CTRL_SUB_SHIFT = 16;
// result ring in LDS: RING bundles x [half][64 lanes][16 B] = RING * 2 KiB per wave
static const uint32_t RING_BUNDLES = 8;
static const uint32_t RING_SLOT_BYTES = 2048, RING_HALF_BYTES = 1024;
// third operand (TernCond) byte offset: bit 31 = tile-relative (always a memory reference)
static const uint32_t CREF_TILE = 0x80000000u;
enum SubOp : uint32_t { SUB_TERN = 33, SUB_INPUT = 34 };

// witness reference (pack kernel): bit 31 set -> constant table index, else value slot of the tile
static const uint32_t REF_CONST = 0x80000000u;

// per-set status bits written by the kernels (the reference panics in these cases)
enum SetStatus : uint32_t {
    ST_SHL_OVERFLOW = 1u,  // Shl result >= r            (graph.rs:634 from_bigint().unwrap())
    ST_BITOP_EQ_R = 2u,    // Bor/Bxor result == r       (graph.rs:701,716)
};

// Device pointers of an uploaded program.
struct ProgramDev {
    const uint32_t* hdr;           // [n_bundles]
    const uint32_t* recs;          // [n_bundles*G*4], 16-byte aligned records
    const uint32_t* crefs;         // [n_bundles*G]
    const uint32_t* consts;        // [n_const*8] Montgomery form; copied to the head of the workspace per launch
    const uint32_t* witness_refs;  // [n_witness]
    uint32_t n_bundles, n_slots, n_inputs, n_witness, n_const;
};

// A launch covers up to WS_MAX_CHUNKS workspaces of < 4 GiB each (the 32-bit buffer window is per descriptor, not per
// launch): tile i lives in chunk i / tiles_per_chunk, every chunk starts with its own copy of the constant table.
static const uint32_t WS_MAX_CHUNKS = 32;
struct WsTable {
    void* base[WS_MAX_CHUNKS];
    uint32_t tiles_per_chunk;
    uint32_t n_chunks;
};

// workspace geometry shared by host and kernels: [constants, padded to 256 B][tiles]
// tile = (n_slots + 1) slots (the last one is the trash slot) of 32*T bytes: [slot][half][T][16 B].
// A constant uses the same geometry (32*T bytes apart, halves 16*T apart, only the first 16 bytes of each half
// used) so that every operand's high half sits at the same fixed distance from its low half.
CWC_HD uint64_t ws_const_bytes(uint32_t n_const, uint32_t T) { return (((uint64_t)n_const * 32u * T) + 255u) & ~255ull; }
CWC_HD uint64_t ws_tile_bytes(uint32_t n_slots, uint32_t T) { return ((uint64_t)n_slots + 1u) * 32u * T; }

}  // namespace cwc
