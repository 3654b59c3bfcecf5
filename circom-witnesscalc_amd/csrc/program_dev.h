// Device-visible part of the compiled program format (shared by the host compiler and the HIP kernels).
#pragma once
#include <stdint.h>

#include "ops.h"

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

// per-lane sub-op codes inside a record: DuoOp wire codes 0..19 plus
enum SubOp : uint32_t { SUB_NEG = 32, SUB_TERN = 33, SUB_INPUT = 34 };

// operand reference: bit 31 set -> constant table index, else value slot of the tile
static const uint32_t REF_CONST = 0x80000000u;

// per-set status bits written by the kernels (the reference panics in these cases)
enum SetStatus : uint32_t {
    ST_SHL_OVERFLOW = 1u,  // Shl result >= r            (graph.rs:634 from_bigint().unwrap())
    ST_BITOP_EQ_R = 2u,    // Bor/Bxor result == r       (graph.rs:701,716)
};

// Device pointers of an uploaded program (kernel argument).
struct ProgramDev {
    const uint32_t* hdr;           // [n_bundles]
    const uint32_t* recs;          // [n_bundles*G*4], 16-byte aligned records
    const uint32_t* crefs;         // [n_bundles*G]
    const uint32_t* consts;        // [n_const*8], 16-byte aligned halves
    const uint32_t* witness_refs;  // [n_witness]
    uint32_t n_bundles, n_slots, n_inputs, n_witness;
};

}  // namespace cwc
