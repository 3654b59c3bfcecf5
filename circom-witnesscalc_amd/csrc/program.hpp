// Compiled program: what the host graph compiler hands to the HIP interpreter.
//
// A *tile* is T input sets evaluated by one 64-lane wavefront; the wave's lanes are (node slot j,
// set t) with j in [0, G), G = 64/T, t in [0, T).  The op graph is level-scheduled by dependency
// depth and, inside a level, grouped into same-class *bundles* of up to G independent nodes that one
// wave-instruction stream evaluates at once (T = 64 degenerates to one node per bundle in node order,
// i.e. the reference's sequential loop src/graph.rs:372-382 with lanes = input sets).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "graph.hpp"
#include "program_dev.h"

namespace cwc {

struct ProgramStats {
    uint64_t n_nodes = 0, n_op = 0, n_input_nodes = 0, n_const = 0, n_witness = 0;
    uint64_t depth = 0;                 // dependency levels
    uint64_t class_nodes[C_COUNT] = {0};
    uint64_t class_bundles[C_COUNT] = {0};
    uint64_t n_op_compiled = 0;         // operation nodes after the depth-reducing rewrite (>= n_op is possible)
    uint64_t n_bitx_bundles = 0;        // BIT bundles made of bit-extract nodes only (cheap path)
    uint64_t n_bitx_nodes = 0;          // bit-extract nodes made by the compiler
    uint64_t algorithmic_bytes_per_set = 0;  // 32*[sum_ops(arity+1) + 2*n_input_nodes + 2*W]  
    uint64_t n_coop_rider_bundles = 0;  // narrow multiplication bundles that carry linear riders
    uint64_t n_conversions = 0, n_canonical = 0, form_cycles_saved = 0;  // representation inference: inserted conversions, operations whose value is kept as a canonical integer
    uint64_t n_folded = 0, n_numbered = 0, n_shaken = 0;  // load-time optimiser: operations folded to constants / aliases, nodes merged by value numbering, unused nodes dropped
    uint64_t n_fused_nodes = 0;         // fused narrow chains made by the compiler (class C_MULF)
    uint64_t n_scan_steps = 0;          // steps of serial limb recurrences run inside scan bundles (class C_SCAN)
    uint64_t chain_floor_cycles = 0;    // the compiled graph's longest dependent chain at the best measured per-operation latency on a lone wave (no bundle overhead)
    uint64_t n_conv_products = 0;       // limb products computed inside convolution bundles (rewrite.cc detect_convolutions)
    uint64_t depth_scan = 0;            // estimate of the dependency depth with such steps at a tenth of a level (what bounds the bundle count of a program with scan bundles)
};

struct Program {
    uint32_t T = 0, G = 0;
    uint32_t divider = 0;                // W > 0: compiled for a divider wave shared by W interpreter waves (1 or 4)
    uint32_t n_div_requests = 0;         // C_DIVREQ bundles (what the divider wave serves, in order)
    uint32_t n_bundles = 0, n_slots = 0, n_const = 0, n_inputs = 0, n_witness = 0;
    uint32_t trash_off = 0;              // tile-relative destination of results without a slot (program_dev.h OFF_NOWHERE, or the trash slot)
    std::vector<uint32_t> hdr;           // [n_bundles]      see program_dev.h (format v4)
    std::vector<uint32_t> recs;          // [n_bundles*G*4]  {a_off, b_off, dst | ctrl, a_lds | b_lds << 16}
    std::vector<uint32_t> crefs;         // [n_cref_rows*G]  third operand byte offset (C_TERN) / input index (C_INPUT): one row per such bundle, in bundle order
    std::vector<uint32_t> consts;        // [n_const*8]      Montgomery form
    std::vector<uint32_t> witness_refs;  // [n_witness]      slot or REF_CONST|idx
    std::vector<uint32_t> div_lanes;     // [n_div_requests] active lanes of each division request
    // Streams: the wavefronts of one tile, each with its own bundles [stream_first[s], stream_first[s] + stream_count[s])
    // over the graph's independent parts (compile.cc); stream 0 evaluates the Input nodes for all of them.  With a
    // divider (divider == 1 only) every stream has its own divider wave and serves stream_div_requests[s] requests.
    uint32_t n_streams = 1;
    uint32_t n_cref_rows = 0, stream_cref_first[MAX_STREAMS] = {0, 0, 0, 0};  // rows of crefs; rows in front of each stream's first bundle
    uint32_t stream_first[MAX_STREAMS] = {0, 0, 0, 0}, stream_count[MAX_STREAMS] = {0, 0, 0, 0}, stream_div_requests[MAX_STREAMS] = {0, 0, 0, 0};
    double stream_cycles[MAX_STREAMS] = {0, 0, 0, 0}, stream_cycles_mul_div[MAX_STREAMS] = {0, 0, 0, 0}, stream_chain_cycles[MAX_STREAMS] = {0, 0, 0, 0};  // lone-wave cycles of each stream's bundles
    ProgramStats stats;
};

// Validates the graph (backward references, evaluable ops, index ranges) and compiles it for tile width T
// (power of two, 1..64); divider = W > 0 compiles divisions for a divider wave shared by W interpreter waves (W = 1 or
// 4, T < 64 only).
// streams = 2 or 4: the graph's independent parts are spread over that many wavefronts per tile (fewer when it has fewer
// parts; divider 0 or 1 only).
// quick: one schedule (the base policy) instead of the search over schedule variants: a tenth of the compile time, a few percent
// more cycles.
// `shared`: the rewritten graph (load-time optimiser, bit-extract fusion, tree-height reduction) does not depend on the
// divider mode or the stream count -- candidates of one choice (pick_tile_width) hand the same holder to their compiles,
// the first one through rewrites, the others copy (make_shared_rewrites / free_shared_rewrites; thread-safe).
struct SharedRewrites;
SharedRewrites* make_shared_rewrites();
void free_shared_rewrites(SharedRewrites*);
bool compile_program(const Graph& g, uint32_t T, uint32_t divider, Program& out, std::string& err, uint32_t streams = 1, bool quick = false,
                     SharedRewrites* shared = nullptr);
// Validation and statistics of a loaded graph without compiling a program: out.stats, out.n_inputs, out.n_witness.
bool probe_graph(const Graph& g, Program& out, std::string& err);
// "program key" used by the runtime and the C-ABI wherever a tile width is passed: T | KEY_DIVIDER | KEY_GROUP
static const uint32_t KEY_DIVIDER = 0x100u;  // one divider wave per interpreter wave
static const uint32_t KEY_GROUP = 0x200u;    // one divider wave per four interpreter waves
static const uint32_t KEY_TRIPLE = 0x400u;   // one divider wave per three interpreter waves (a four-wave workgroup: one wave per SIMD)
static const uint32_t KEY_STREAMS2 = 0x800u;  // two wavefronts per tile, each over its share of the graph's independent parts
static const uint32_t KEY_STREAMS4 = 0x1000u; // four
static const uint32_t KEY_MODE_MASK = KEY_DIVIDER | KEY_GROUP | KEY_TRIPLE | KEY_STREAMS2 | KEY_STREAMS4;
static inline uint32_t key_streams(uint32_t key) { return key & KEY_STREAMS4 ? 4u : key & KEY_STREAMS2 ? 2u : 1u; }
static inline uint32_t key_divider_waves(uint32_t key) { return key & KEY_GROUP ? 4u : key & KEY_TRIPLE ? 3u : key & KEY_DIVIDER ? 1u : 0u; }
static inline uint32_t key_mode_of_divider(uint32_t divider) { return divider == 4 ? KEY_GROUP : divider == 3 ? KEY_TRIPLE : divider ? KEY_DIVIDER : 0u; }

// Lone-wave shader cycles of a program's bundles (measured per class on MI355X, profiles/r01_class_profile.txt): what the
// compiler uses to choose between schedule variants and the runtime to choose a program for a batch.
double program_wave_cycles(const Program& p);
double program_wave_cycles_mul_div(const Program& p);
double model_class_cycles(int bundle_class);  // the table behind both (lone-wave shader cycles per bundle of a class)
uint64_t model_table_id();                     // changes with that table (built-in, CWC_MODEL_CYCLES, or the calibration file)

// Structural check of a program from outside compile_program (imported blob): every offset, index and LDS address the
// kernels take from it lies inside the tile / LDS / input geometry.
bool validate_program(const Program& p, std::string& err);

// pointer-free serialisation (what is broadcast between GPUs)
std::vector<uint8_t> program_to_blob(const Program& p);
size_t program_blob_size(const Program& p);
void program_blob_write(const Program& p, uint8_t* dst);  // (program_blob_size(p) bytes at dst)
bool program_from_blob(const uint8_t* data, size_t len, Program& p, std::string& err);

}  // namespace cwc
