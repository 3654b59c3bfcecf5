// Internal header of the graph compiler's translation units: rewrite.cc (exact graph rewrites and the node forms), compile.cc
// (program choice among schedule variants, list scheduler, operand routing, slots, encoding), costmodel.cc (measured cycles
// per bundle class), program_blob.cc (validation and the pointer-free blob).  The compiler's interface is program.hpp.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <chrono>
#include <deque>
#include <mutex>
#include <unordered_map>

#include "flat_map.hpp"
#include "program.hpp"

namespace cwc {

// fused nodes (N_FUSED, made by fuse_narrow_chains below): op = sq | op2 << 1 | op3 << 4
static inline bool fused_sq(uint8_t op) { return (op & 1u) != 0; }
static inline uint32_t fused_op2(uint8_t op) { return (op >> 1) & 7u; }
static inline uint32_t fused_op3(uint8_t op) { return (op >> 4) & 7u; }
static inline uint8_t fused_code(bool sq, uint32_t op2, uint32_t op3) { return (uint8_t)((sq ? 1u : 0u) | (op2 << 1) | (op3 << 4)); }

inline int class_of(const Node& n) {
    switch (n.kind) {
        case N_FUSED: return C_MULF;
        case N_SCAN: return C_SCAN;
        case N_CONV: return C_SCAN;
        case N_INPUT: return C_INPUT;
        case N_UNO: return C_LIN;
        case N_TRES: return C_TERN;
        case N_DUO:
            switch (n.op) {
                case OP_MUL: return C_MUL;
                case OP_DIV: return C_DIV;
                case OP_ADD: case OP_SUB: return C_LIN;
                case OP_EQ: case OP_NEQ: case OP_LAND: case OP_LOR: return C_CMPZ;
                case OP_LT: case OP_GT: case OP_LEQ: case OP_GEQ: return C_CMPS;
                case OP_SHL: case OP_SHR: case OP_BOR: case OP_BAND: case OP_BXOR: case OP_BITX: return C_BIT;
                case OP_IDIV: case OP_MOD: return C_IDIVMOD;
            }
    }
    return -1;
}

// (a fused node's operands in a, b, c: the factor(s) of its product, then the operands of its second and third stage)
inline int arity_of(const Node& n) {
    if (n.kind == N_FUSED) return fused_sq(n.op) ? 1 + (fused_op2(n.op) ? 1 : 0) + (fused_op3(n.op) ? 1 : 0) : 3;
    if (n.kind == N_SCAN) return scan_has_third(n.op) ? 3 : 2;  // x, the accumulator coming in, the divisor / subtrahend / other comparand
    if (n.kind == N_CONV) return 2;                              // x_c, y_c
    return n.kind == N_UNO ? 1 : n.kind == N_DUO ? 2 : n.kind == N_TRES ? 3 : 0;
}

// Relative cost of one bundle of each class (measured on gfx950 for a lone wavefront, shader cycles / 50): the unit
// of the scheduler's critical-path heights and of the tree-height reduction below.
static const uint32_t kClassCost[C_COUNT] = {100, 47, 12, 1470, 25, 100, 110, 175, 38, 22, 22, 26, 14, 46, 80};  // (LIN: 12 measured best of 6..26 on the authV2-class graph)
// The same for graphs whose linear nodes outnumber their multiplications (sha256-like: wide, LIN bundles are half of
// the time): a heavier Add / Sub makes the tree-height reduction rebalance sum chains harder and puts linear chains
// first in the schedule -- sha256_512 at 4096 sets 10.6 -> 9.2 ms; the authV2-class graph (multiplier chains with
// narrow linear steps in between) loses 3 % with it and keeps the measured ratio.
// Tiles of one or two input sets run the critical chain's multiplications in narrow (four-lane) bundles: its steps cost
// what a narrow bundle and a linear bundle cost (1 306 : 706 cycles in the product kernel = 26 : 14; measured best of
// 26..40 : 14..24 on the authV2-class graph: 1024 sets 13.27 -> 12.19 ms, 256 sets 12.27 -> 10.52 ms; wider tiles, whose
// multiplication bundles stay full-width, keep the table above: 8192 sets 32.6 ms with either, 33.4 ms with this one).
// (round 3, same-box A/B of 30 : {16, 20, 24, 30} and neighbours: 30 : 24 is 1 % ahead at 1024 sets -- 80.1-80.2 k against 79.3-79.5 k
// witnesses/s -- and level at 256 / 512 sets, profiles/r03_weights_ab.txt)
static const uint32_t kClassCostNarrow[C_COUNT] = {100, 30, 24, 1470, 25, 100, 110, 175, 38, 22, 22, 26, 24, 46, 80};
static const uint32_t kClassCostLinHeavy[C_COUNT] = {100, 47, 47, 1470, 25, 100, 110, 175, 38, 22, 22, 26, 47, 46, 80};
// What the scheduler's virtual clock advances per bundle (it decides when a division's collect bundle is due; too fast
// a clock collects before the divider wave has answered and the interpreter waits): shader cycles / 50 as measured
// at the end of round 1 (MUL 2 100, LIN 670, request / collect 1 300).
static const uint32_t kClockCost[C_COUNT] = {100, 42, 14, 1470, 25, 100, 110, 175, 38, 26, 26, 24, 14, 44, 80};
// The inversion entries of the three tables follow the cycle table (model_class_cycles(C_DIV) / 50): one number to change
// when the inversion gets faster, and what CWC_MODEL_CYCLES overrides.
uint32_t div_cost50();  // (costmodel.cc)
static inline uint64_t cost_of(const uint32_t* table, int c) { return c == (int)C_DIV ? div_cost50() : table[c]; }
// a fused node costs its bundle's front end and its stages (cycles / 50: 600 + 704 per product + ~280 per addition)
static inline uint64_t fused_cost50(uint8_t op) {
    return 12u + 15u + (fused_op2(op) == FOP_MUL ? 15u : fused_op2(op) ? 6u : 0u) + (fused_op3(op) ? 6u : 0u);
}
// a step of a scan bundle: its share of the bundle's front end and one round of the loop (cycles / 50; kCyclesScan* below)
static inline uint64_t scan_cost50(uint8_t op) { return (op & SCAN_OP_DIV) ? 10u : scan_is_sel(op) ? 38u : (op & (SCAN_OP_BORROW | SCAN_OP_LEX)) ? 1u : 4u; }  // (the one-bit recurrences run all steps at once; a selection stands alone)
static inline uint64_t node_cost(const uint32_t* table, const Node& n) {
    return n.kind == N_FUSED ? fused_cost50(n.op) : n.kind == N_SCAN ? scan_cost50(n.op) : n.kind == N_CONV ? 70u : cost_of(table, class_of(n));  // (a convolution bundle: ~3.5 k cycles)
}

// the list scheduler's "nothing is ready although nodes remain": a grouping (scan chain, convolution block) whose members depend on each other;
// compile_program falls back to the program without such groupings on exactly this error
static const char kErrSchedulerDeadlock[] = "internal error: scheduler found no ready node";

// ---- node forms (rewrite.cc infer_representations) ----
static const uint8_t REP_M = 0, REP_C = 1;
static const uint8_t VF_A_CANON = 1, VF_B_CANON = 2, VF_OUT_CANON = 4;
static const uint8_t VF_MUL_CC = 8;  // a multiplication of two canonical integers that stays canonical (C_MUL bundles with HDR_MUL_CC)
inline bool is_integer_class(int c) { return c == C_BIT || c == C_IDIVMOD || c == C_CMPS; }

// ---- the exact rewrites (rewrite.cc), in pipeline order ----
void rewrite_pow2_divisions(Graph& g);
void fuse_bit_extract(Graph& g);
void reduce_tree_height(Graph& g, size_t kMaxLeaves, const uint32_t* class_cost);
void infer_representations(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, uint64_t& n_conversions, uint64_t& n_canonical, bool all_montgomery,
                           bool allow_cc, uint64_t& n_cc, bool canonical_inputs);
void detect_scans(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner, uint64_t& n_steps);
// borrow chains of register-wise subtractions and most-significant-difference comparisons (SCAN_OP_BORROW / SCAN_OP_LEX)
void detect_bit_scans(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner, uint64_t& n_steps);
// scan_imm[column node] = column | k << 8, scan_partner[column node] = the node of column 0 (the group's name)
void detect_convolutions(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner, uint32_t max_columns,
                         uint64_t& n_products);
void fuse_narrow_chains(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, const uint32_t* class_cost, uint32_t slack_permille, bool two_stage_only,
                        uint64_t& n_fused);

// ---- the cost model's cycle table (costmodel.cc) ----
// Lone-wave shader cycles per bundle class in the product kernel (stamped build minus its five ~40-cycle stamps,
// profiles/r02_class_profile.txt; check: 12 953 MUL + 7 258 LIN + 265 request / collect pairs -> 32.8 M cycles = 13.7 ms
// at 2.4 GHz against 13.6 ms measured for the round-1 program).
// Integer-class bundles (BIT, IDIVMOD, CMPS) are priced with every operand and the result converted (the bigint-class
// profile: BIT 6 585, IDIVMOD 7 054); a bundle whose operands / result stay canonical integers (representation
// inference) saves kCyclesOperandForm per operand and kCyclesResultForm for the result.
static const double kCyclesDefault[C_COUNT] = {4000, 2015, 706, 55000, 1000, 4700, 6400, 6850, 1450, 1490, 3700, 1306, 900, 2400, 16920};
// a scan bundle (C_SCAN) is priced as its front end plus the rounds of its loop (the table entry is 32 rounds of the
// division step, 2 200 + 32 x 460; a bundle books what it costs less): the limb-sized paths, measured on MI355X
// (profiles/r04_class_profile.txt)
static const double kCyclesMulCC = 760;  // a bundle of canonical limb products (HDR_MUL_CC)
// (carry bundles of 32 rounds 6.3 k cycles, division bundles 17 k: 33 and 85 instructions per round on a lone wave, the
// division bundle's reciprocal once per bundle)
static const double kCyclesConvFront = 900, kCyclesConvStep = 125;  // a convolution bundle: k rounds of one 64 x 64 multiply-accumulate per lane (four quarter-rate v_mad_u64_u32 among 14 instructions)
// chains of 64-bit limbs run all segments of a bundle at once (scan_gfx950.hpp): a flat carry-lookahead; log2 rounds of two products modulo d
static const double kCyclesScanParCarry = 700, kCyclesScanParDivRound = 900, kCyclesScanParDivFlat = 900;
static const double kCyclesScanFront = 1000, kCyclesScanFrontDiv = 2200, kCyclesScanStepCarry = 170, kCyclesScanStepDiv = 460;
static const double kCyclesScanBits = 700;  // a bundle of one-bit recurrences (borrow chain, comparison): two comparisons, a carry-lookahead over the wave, one subtraction
// a fused narrow bundle (C_MULF) is priced with all three stages (product, product, addition); what a bundle without
// the second product / without additions saves
static const double kCyclesFusedStageMul = 760, kCyclesFusedStageLin = 300;
// The table above was measured on one box.  Overrides, read once when the library is loaded: CWC_MODEL_CYCLES=
// "class:cycles,..." (what-if runs of the cost model), else the calibration file tools/gpu_calibrate.py --write leaves
// behind after measuring the classes on the machine at hand with the stamped interpreter build -- CWC_MODEL_CYCLES_FILE, or
// model_cycles.txt in the program cache's directory (CWC_PROGRAM_CACHE / XDG_CACHE_HOME / ~/.cache/circom-witnesscalc-amd;
// no directory, no file).  Same "class:cycles,..." text; entries outside [0.25, 4] x the built-in value are ignored.
struct CycleTable {
    double v[C_COUNT];
    bool from_file = false;
    void parse(const char* e, bool bounded) {
        while (*e) {
            char* end = nullptr;
            const long c = strtol(e, &end, 10);
            if (end == e || *end != ':') break;
            const double cyc = strtod(end + 1, &end);
            if (c >= 0 && c < (long)C_COUNT && cyc > 0 && (!bounded || (cyc >= 0.25 * kCyclesDefault[c] && cyc <= 4.0 * kCyclesDefault[c]))) v[c] = cyc;
            while (*end == ' ' || *end == '\n' || *end == '\r') ++end;
            e = *end == ',' ? end + 1 : end;
            if (*end != ',') break;
        }
    }
    CycleTable() {
        for (int c = 0; c < (int)C_COUNT; ++c) v[c] = kCyclesDefault[c];
        if (const char* e = getenv("CWC_MODEL_CYCLES")) {
            parse(e, false);
            return;
        }
        std::string path;
        if (const char* f = getenv("CWC_MODEL_CYCLES_FILE")) {
            path = f;
        } else {
            std::string dir;
            if (const char* e = getenv("CWC_PROGRAM_CACHE")) {
                if (*e && strcmp(e, "0") && strcmp(e, "off")) dir = e;
                else return;
            } else if (const char* x = getenv("XDG_CACHE_HOME")) {
                if (*x) dir = std::string(x) + "/circom-witnesscalc-amd";
            }
            if (dir.empty()) {
                const char* home = getenv("HOME");
                if (!home || !*home) return;
                dir = std::string(home) + "/.cache/circom-witnesscalc-amd";
            }
            path = dir + "/model_cycles.txt";
        }
        if (FILE* f = fopen(path.c_str(), "rb")) {
            char buf[1024];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            buf[n] = 0;
            parse(buf, true);
            from_file = true;
        }
    }
    double operator[](int c) const { return v[c]; }
};extern const CycleTable kCycles;
// (round 2, bigint-class graph with every operand and result canonical: BIT 2 650, IDIVMOD 2 880, CMPS 1 900 net of stamps)
static const double kCyclesBitStraight = 1500;  // what a Shr-only / Band-only bundle saves against the per-lane select over all bit operations
static const double kCyclesBitx = 1300, kCyclesCoopRiders = 60, kCyclesOperandForm = 1200, kCyclesResultForm = 1450, kCyclesBitxOperandForm = 600;
// When a multiplication step becomes a narrow (four lanes per product) bundle: `fill` or more ready multiplications make
// a full-width bundle instead (it costs the same with 10 or 32 nodes); otherwise a narrow one if the multiplications
// within `slack_levels` multiplication levels (in the scheduler's cost units) of the most urgent ready node fit it.  The
// rest stays ready.  fill = 0: never narrow.
struct CoopPolicy {
    uint32_t fill;
    uint32_t slack_levels;  // ~0u: everything ready counts as urgent
    bool all_montgomery = false;  // no representation inference: every value in Montgomery form
    bool witness_slots = false;   // the slots of witness elements in witness order (see the slot allocation)
    bool no_conv = false;         // schoolbook limb products stay unfused (detect_convolutions off): a competitor where many small blocks run side by side
    bool no_scans = false;        // no scan chains at all (detect_scans / detect_bit_scans / detect_convolutions off): the fallback when a fused form cannot be scheduled
    uint32_t fuse = 0;            // fused narrow chains (fuse_narrow_chains): 0 off, else 1 + the slack, in thousandths of the critical path, within which nodes are fused; + 0x10000: product + sum nodes only
};
}  // namespace cwc
