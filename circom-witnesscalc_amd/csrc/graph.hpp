// Host-side graph IR and file formats of the calc-witness path.
//
// Mirrors the roles of the reference's src/graph.rs:37-59,174-178,209-212,236-245 (Node / Operation
// types), src/storage.rs:137-249 (`wtns.graph.001` container), src/lib.rs:138-247 (inputs JSON,
// inputs buffer) and src/lib.rs:114-123 (`.wtns` framing).  Function names follow the reference.
#pragma once
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "fr_gfx950.hpp"
#include "ops.h"

namespace cwc {

enum NodeKind : uint8_t { N_INPUT = 0, N_CONST = 1, N_UNO = 2, N_DUO = 3, N_TRES = 4,
                          // compiler-internal, never in a file: a fused chain of a product and up to two more steps
                          // (compile.cc fuse_narrow_chains).  op = sq | op2 << 1 | op3 << 4 (FusedOp codes);
                          // sq: (a * a) op2 b op3 c, else (a * b) op2 c
                          N_FUSED = 5,
                          // compiler-internal, never in a file: one output of a step of a serial limb recurrence (compile.cc
                          // detect_scans, class C_SCAN).  op = ScanOp bits: kind (carry chain / long division by one limb) and
                          // role (the step's OUT value: limb / quotient digit; its ACC value: carry / remainder).  Both nodes
                          // of a step name the same operands: a = x, b = the accumulator coming in, c = the divisor (DIV).
                          N_SCAN = 6,
                          // compiler-internal, never in a file: one column sum_{i + j = c} x_i y_j of a schoolbook limb product
                          // (rewrite.cc detect_convolutions; a bundle of class C_SCAN with HDR_SCAN_CONV holds the 2k - 1 columns
                          // of one product).  a = x_c, b = y_c (c < k; the columns above name x_(k-1), y_(k-1)): every factor
                          // is the operand of exactly one column node, the bundle reads them all.
                          N_CONV = 7 };
// (chain ends: a step without an incoming accumulator / without an x reads the constant 0 there; the operand field repeats the other one)
// Round 5, one-bit recurrences of multi-register integers (rewrite.cc detect_bit_scans), operands a = x, b = the bit coming in, c = y:
//   BORROW  the borrow chain of a register-wise subtraction: OUT = x - y - bin (+ 2^n when that is negative), ACC = the borrow going out
//   LEX     a most-significant-difference comparison: ACC = x > y ? KG : x < y ? KL : the bit coming in (KG, KL: the two op bits); OUT unused
enum ScanOp : uint8_t { SCAN_OP_ACC = 1, SCAN_OP_DIV = 2, SCAN_OP_NOACC = 4, SCAN_OP_NOX = 8, SCAN_OP_BORROW = 16, SCAN_OP_LEX = 32, SCAN_OP_KG = 64, SCAN_OP_KL = 128 };
//   SEL     (round 5; BORROW and LEX bits both set: not a recurrence, every step stands alone) a selection through the pair's two records,
//           with its comparison when that is an ordered one read by nothing else: OUT node a = cond (or the comparison's operands a, b),
//           ACC node a = p, b = q; OUT = the comparison's boolean, ACC = cond ? p : q.  scan_imm: the comparison (SelCode, program_dev.h) | 8 when the
//           boolean is wanted in Montgomery form.  The two nodes of a step name DIFFERENT operands (the scheduler waits for both).
static const uint8_t SCAN_OP_SEL = SCAN_OP_BORROW | SCAN_OP_LEX;
static inline bool scan_is_sel(uint8_t op) { return (op & SCAN_OP_SEL) == SCAN_OP_SEL; }
static inline bool scan_is_borrow(uint8_t op) { return (op & SCAN_OP_SEL) == SCAN_OP_BORROW; }
static inline bool scan_is_lex(uint8_t op) { return (op & SCAN_OP_SEL) == SCAN_OP_LEX; }
static inline bool scan_has_third(uint8_t op) { return (op & SCAN_OP_DIV) || scan_is_borrow(op) || scan_is_lex(op); }  // (a divisor / a subtrahend / the other comparand)
static inline uint32_t scan_kind_bits(uint8_t op) { return op & (SCAN_OP_DIV | SCAN_OP_BORROW | SCAN_OP_LEX | SCAN_OP_KG | SCAN_OP_KL); }  // (what the steps of one bundle share, with the shift)

// graph::Node (reference src/graph.rs:236-245).  N_INPUT: a = input index.  N_CONST: a = index into
// Graph::const_values (canonical value, already reduced mod r as storage.rs:28 does on load).
struct Node {
    uint8_t kind;
    uint8_t op;
    uint32_t a, b, c;
};

struct InputSignal {  // InputSignalsInfo entry (reference src/lib.rs:19)
    std::string name;
    uint32_t offset, len;
};

struct Graph {
    std::vector<Node> nodes;
    std::vector<Fr> const_values;            // canonical (non-Montgomery) 256-bit little-endian limbs
    std::vector<uint32_t> witness_signals;   // node index per witness element
    std::vector<InputSignal> inputs;         // in file order
    std::unordered_map<std::string, uint32_t> input_index;  // name -> position in `inputs`
    uint64_t n_op = 0;                       // Uno + Duo + Tres nodes
};

// storage.rs:214-249.  Returns false and sets err on malformed data (the reference panics / returns io::Error).
bool deserialize_witnesscalc_graph(const uint8_t* data, size_t len, Graph& g, std::string& err);
// storage.rs:137-183
std::vector<uint8_t> serialize_witnesscalc_graph(const Graph& g);

// Load-time re-optimiser (optimize.cc; the reference's build-time passes src/graph.rs:358-619 as exact rewrites): constant
// propagation with eval_fr semantics, structural value numbering of all pure operations, removal of unused nodes (Input
// nodes and operations that can fail are kept).  Rewrites g in place; witness values are unchanged.
struct OptimizeStats {
    uint64_t nodes_before = 0, nodes_after = 0, folded = 0, numbered = 0, constants_merged = 0, shaken = 0;
    uint64_t random_constants = 0, random_numbered = 0;  // the probabilistic passes (random_eval_passes)
};
void optimize_loaded_graph(Graph& g, OptimizeStats* stats);
// The reference's probabilistic build-time passes (src/graph.rs:499-583: random_eval, value_numbering, constants), opt-in
// (CWC_RANDOM_EVAL=1): nodes of equal value under random evaluation are merged / folded; run in front of the exact passes.
void random_eval_passes(Graph& g, OptimizeStats* stats);

// lib.rs:138-152
size_t get_inputs_size(const Graph& g);
// Size actually allocated for the inputs buffer: max(get_inputs_size, max(offset+len) over the input map,
// 1 + max Input index anywhere) so that no access is out of range (reference would panic).
size_t inputs_buffer_size(const Graph& g);

typedef std::vector<std::pair<std::string, std::vector<Fr>>> InputList;  // insertion order, unique keys
// lib.rs:195-247.  Error strings follow the reference's Error Debug output.
bool deserialize_inputs(const char* json, size_t len, InputList& out, std::string& err);
// Batched front-end: splits either a top-level JSON array of input objects or NDJSON (one object per
// line, blank lines ignored) into [begin, end) spans, each of which is then parsed by deserialize_inputs.
bool split_inputs_batch(const char* text, size_t len, std::vector<std::pair<size_t, size_t>>& spans, std::string& err);
// lib.rs:177-181 + 154-168: buf = n_inputs x 32-byte canonical LE, buf[0] = 1.
bool populate_inputs(const InputList& inputs, const Graph& g, uint8_t* buf, size_t n_inputs, std::string& err);

// lib.rs:114-123: 76-byte header + 32*n bytes
size_t wtns_size(size_t n_witness);
void wtns_write_header(uint8_t* out, size_t n_witness);
void wtns_from_witness(const uint8_t* witness32, size_t n_witness, uint8_t* out);

// canonical reduction helpers (host)
Fr u256_from_le_bytes_mod_order(const uint8_t* b, size_t n);
bool u256_parse_dec(const std::string& s, Fr& out, std::string& err);

}  // namespace cwc
