// Host graph compiler: validation, exact rewrites (power-of-two divisions, bit-extract fusion, tree-height reduction with
// shared subexpressions and dead-node elimination), critical-path list scheduling into same-class bundles (linear riders,
// request / collect divisions for the divider waves), operand routing (LDS ring vs. staged memory), liveness-based slot
// allocation, program encoding (format v4) and the pointer-free program blob.  See program.hpp / program_dev.h.
#include <map>

#include "compile_internal.hpp"

namespace cwc {

// The rewritten graph (load-time optimiser, bit-extract fusion, tree-height reduction) depends on the fusion switch and on
// the weight table only: the schedule variants of one compile_program call share it instead of redoing it.
struct RewriteCache {
    struct Entry {
        bool bit_fusion;
        const uint32_t* table;  // kClassCost / kClassCostNarrow (before the linear-heavy switch and the A/B overrides)
        uint32_t G;             // (the tree-height reduction's leaf bound follows it)
        Graph g;
        ProgramStats st;
        const uint32_t* class_cost;
    };
    std::deque<Entry> entries;  // (a deque: entries stay where they are while other threads append)
    std::mutex* lock = nullptr; // set when the compiles of several threads share the cache
};
struct SharedRewrites {  // one cache and one lock per tile width 1, 2, .. 64: compiles of different widths do not wait for each other
    std::mutex m[7];
    RewriteCache cache[7];
};
SharedRewrites* make_shared_rewrites() {
    SharedRewrites* s = new SharedRewrites();
    for (int i = 0; i < 7; ++i) s->cache[i].lock = &s->m[i];
    return s;
}
void free_shared_rewrites(SharedRewrites* s) { delete s; }
static bool compile_variant(const Graph& g_in, uint32_t T, uint32_t divider, bool bit_fusion, const CoopPolicy& policy, Program& out, std::string& err,
                            RewriteCache* cache = nullptr, bool probe_only = false, uint32_t streams = 1);

// Validation and statistics of a loaded graph without compiling a program (what gwb_graph_load needs).
bool probe_graph(const Graph& g, Program& out, std::string& err) { return compile_variant(g, 64, 0, false, CoopPolicy{0, 0}, out, err, nullptr, true); }

// The list scheduler is a heuristic, and exact rewrites and the narrow-bundle policy shift how the chains of a graph line
// up in bundles: the program is compiled with and without the bit-extract fusion, then under a few narrow-bundle
// policies, and the cheapest schedule by the measured cycles per bundle class (program_wave_cycles) is kept -- the
// policies are not fitted to one graph, the cost model picks per graph and tile width.
bool compile_program(const Graph& g, uint32_t T, uint32_t divider, Program& out, std::string& err, uint32_t streams, bool quick, SharedRewrites* shared) {
    const uint32_t G = T ? 64 / T : 1;
    CoopPolicy base{G, ~0u};  // narrow whenever everything ready fits
    if (const char* e = getenv("CWC_COOP_FILL")) base.fill = (uint32_t)atol(e);
    if (const char* e = getenv("CWC_COOP_SLACK")) base.slack_levels = (uint32_t)atol(e);
    const bool forced = getenv("CWC_COOP_FILL") || getenv("CWC_COOP_SLACK");
    if (getenv("CWC_NO_COOP_MUL") || coop_nodes(T) == 0) base.fill = 0;
    if (const char* e = getenv("CWC_WITNESS_SLOTS")) base.witness_slots = atoi(e) != 0;
    RewriteCache own_cache;
    RewriteCache& cache = shared && T >= 1 && T <= 64 && !(T & (T - 1)) ? shared->cache[__builtin_ctz(T)] : own_cache;
    if (!compile_variant(g, T, divider, true, base, out, err, &cache, false, streams)) {
        // A fused form that cannot be scheduled must not fail the graph: the scan / convolution rewrites group nodes into one
        // bundle, and a grouping the detection should have rejected (a member that depends on another member) shows up as a
        // scheduler without ready nodes.  The program without convolution bundles, then without any scan chains, is always there.
        // (the scheduler's deadlock is named by ONE constant, kErrSchedulerDeadlock, here and where it is raised: rewording the message
        // cannot turn the fallback off; when every attempt fails the LAST attempt's error is what the caller sees)
        if (err != kErrSchedulerDeadlock) return false;
        base.no_conv = true;
        std::string err2;
        if (!compile_variant(g, T, divider, true, base, out, err2, &cache, false, streams)) {
            base.no_scans = true;
            if (!compile_variant(g, T, divider, true, base, out, err2, &cache, false, streams)) {
                err = err2;
                return false;
            }
        }
        err.clear();
    }
    if (quick || getenv("CWC_NO_SCHEDULE_VARIANTS")) return true;  // (quick: the first call on a graph runs this one schedule while the search runs in the background)
    // (one after the other: side by side on two threads the two compiles were no faster, 0.55 s either way for the
    // authV2-class graph, and slower for multi-million-node graphs)
    bool fusion = true;
    if (out.stats.n_bitx_nodes != 0 && !getenv("CWC_NO_BIT_FUSION")) {
        Program alt;
        std::string err2;
        if (compile_variant(g, T, divider, false, base, alt, err2, &cache, false, streams) && program_wave_cycles(alt) < program_wave_cycles(out)) {
            out = std::move(alt);
            fusion = false;
        }
    }
    // Representation inference is a heuristic too: where it had to insert conversions, the all-Montgomery program competes
    if (out.stats.n_conversions != 0 && !getenv("CWC_NO_REP_INFERENCE") && g.nodes.size() <= 2000000) {
        CoopPolicy plain = base;
        plain.all_montgomery = true;
        Program alt;
        std::string err2;
        if (compile_variant(g, T, divider, fusion, plain, alt, err2, &cache, false, streams) && program_wave_cycles(alt) < program_wave_cycles(out)) {
            out = std::move(alt);
            base.all_montgomery = true;
        }
    }
    // Convolution bundles hold ONE limb product each: a long product (k = 32: 63 lanes busy for 32 rounds) always pays, many
    // small products that the unfused program runs side by side in a few full bundles may not -- the unfused program competes
    if (out.stats.n_conv_products != 0 && g.nodes.size() <= 2000000 && !getenv("CWC_CONV_ALWAYS")) {
        CoopPolicy plain = base;
        plain.no_conv = true;
        Program alt;
        std::string err2;
        if (compile_variant(g, T, divider, fusion, plain, alt, err2, &cache, false, streams) && program_wave_cycles(alt) < program_wave_cycles(out)) {
            out = std::move(alt);
            base.no_conv = true;
        }
    }
    if (base.fill == 0 || forced || g.nodes.size() > 2000000) return true;  // (huge graphs: one schedule, compile time counts)
    const CoopPolicy more[] = {{0, 0}, {std::max(12u, G * 3 / 8), 0}, {G / 2, 2}, {G * 5 / 8, 2}};
    CoopPolicy kept = base;
    for (CoopPolicy pol : more) {
        pol.all_montgomery = base.all_montgomery;
        pol.witness_slots = base.witness_slots;
        pol.no_conv = base.no_conv;
        pol.no_scans = base.no_scans;
        Program alt;
        std::string err2;
        if (compile_variant(g, T, divider, fusion, pol, alt, err2, &cache, false, streams) && program_wave_cycles(alt) < program_wave_cycles(out)) {
            out = std::move(alt);
            kept = pol;
        }
    }
    // Fused narrow chains (fuse_narrow_chains): how far from the critical path a chain is still fused is a policy too --
    // only the critical chain, chains within a few percent of it, every chain -- and the cost model picks
    // (CWC_FUSE=<thousandths + 1> forces one, CWC_NO_FUSE=1 none).
    if (T <= COOP_FUSE_MAX_T && kept.fill && !getenv("CWC_NO_FUSE")) {
        std::vector<uint32_t> tries = {1, 11, 101, 1001, 0x10001, 0x1000b, 0x10065, 0x103e9};
        if (const char* e = getenv("CWC_FUSE")) tries.assign(1, (uint32_t)atoi(e));
        for (uint32_t f : tries) {
            CoopPolicy pol = kept;
            pol.fuse = f;
            Program alt;
            std::string err2;
            if (compile_variant(g, T, divider, fusion, pol, alt, err2, &cache, false, streams) && (getenv("CWC_FUSE") || program_wave_cycles(alt) < program_wave_cycles(out))) out = std::move(alt);
        }
    }
    return true;
}

static bool compile_variant(const Graph& g_in, uint32_t T, uint32_t divider, bool bit_fusion, const CoopPolicy& policy, Program& out, std::string& err,
                            RewriteCache* cache, bool probe_only, uint32_t streams) {
    if (streams != 1 && streams != 2 && streams != 4) {
        err = "a tile is evaluated by 1, 2 or 4 streams";
        return false;
    }
    if (T == 0 || T > 64 || (T & (T - 1))) {
        err = "tile width must be a power of two in 1..64";
        return false;
    }
    const uint32_t* weight_table = policy.fill && T <= 2 ? kClassCostNarrow : kClassCost;
    const RewriteCache::Entry* hit = nullptr;
    // (a shared cache: whoever comes first rewrites with the lock held, the others wait for the entry)
    std::unique_lock<std::mutex> cache_lock;
    if (cache && cache->lock) cache_lock = std::unique_lock<std::mutex>(*cache->lock);
    if (cache)
        for (const auto& e : cache->entries)
            if (e.bit_fusion == bit_fusion && e.table == weight_table && e.G == 64 / T) hit = &e;
    // validate operand order on the graph as loaded, then work on a rewritten copy
    for (size_t i = 0; !hit && i < g_in.nodes.size(); ++i) {
        const Node& n = g_in.nodes[i];
        const int ar = arity_of(n);
        if ((ar >= 1 && n.a >= i) || (ar >= 2 && n.b >= i) || (ar >= 3 && n.c >= i)) {
            err = "node " + std::to_string(i) + " references a node that is not before it";
            return false;
        }
        if (n.kind == N_CONST && n.a >= g_in.const_values.size()) {
            err = "node " + std::to_string(i) + ": bad constant index";
            return false;
        }
    }
    Graph g = hit ? hit->g : g_in;
    if (hit && cache_lock.owns_lock()) cache_lock.unlock();
    // CWC_DEBUG_COMPILE_TIMES=1: seconds per phase on stderr
    const bool phase_times = getenv("CWC_DEBUG_COMPILE_TIMES") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!phase_times) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "compile T=%u: %-28s %.3f s\n", T, name, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    if (!hit) rewrite_pow2_divisions(g);
    size_t N = g.nodes.size();
    const uint32_t G = 64 / T;
    if (divider != 0 && divider != 1 && divider != 3 && divider != 4) {
        err = "divider waves serve 1, 3 or 4 interpreter waves";
        return false;
    }
    if (G == 1) divider = 0;  // T = 64 keeps the reference's node order, one node per bundle
    out = Program();
    out.T = T;
    out.G = G;
    out.divider = divider;
    ProgramStats& st = out.stats;
    st.n_nodes = g_in.nodes.size();
    st.n_witness = g.witness_signals.size();

    // ---- validate (assert_valid, reference src/graph.rs:343-356; evaluate() itself does not check) ----
    const size_t n_in_buf = inputs_buffer_size(g_in);
    // (the reference sizes the buffer from the leading Input nodes, lib.rs:138-152, and panics on anything beyond; here
    // the buffer covers every Input index and every signal of the input map -- within a sane bound: rows are n x 32 bytes)
    if (n_in_buf > (1u << 27)) {
        err = "inputs buffer of " + std::to_string(n_in_buf) + " elements is too large (an input map entry or Input index beyond 2^27)";
        return false;
    }
    const uint32_t* class_cost = nullptr;
    uint32_t cost_override[C_COUNT];
    if (hit) {
        st = hit->st;
        class_cost = hit->class_cost;
    } else {
    uint64_t arity_sum = 0;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        int ar = arity_of(n);
        if (n.kind == N_DUO && n.op == OP_POW) {  // graph.rs:141-142 unimplemented!
            err = "node " + std::to_string(i) + ": operator Pow not implemented for Montgomery";
            return false;
        }
        if (n.kind == N_UNO && n.op != UOP_NEG) {  // graph.rs:195
            err = "node " + std::to_string(i) + ": uno operator Id not implemented for Montgomery";
            return false;
        }
        if (ar) {
            st.n_op++;
            arity_sum += (uint64_t)ar + 1;
        } else if (n.kind == N_INPUT) {
            st.n_input_nodes++;
        }
    }
    for (uint32_t w : g_in.witness_signals)
        if (w >= g_in.nodes.size()) {
            err = "witness signal references node " + std::to_string(w) + " beyond the graph";
            return false;
        }
    st.algorithmic_bytes_per_set = 32ull * (arity_sum + 2 * st.n_input_nodes + 2 * st.n_witness);

    phase("validate");
    // ---- levels ----
    std::vector<uint32_t> level(N, 0);
    uint32_t depth = 0;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        int ar = arity_of(n);
        if (!ar) continue;
        uint32_t l = level[n.a];
        if (ar >= 2) l = std::max(l, level[n.b]);
        if (ar >= 3) l = std::max(l, level[n.c]);
        level[i] = l + 1;
        depth = std::max(depth, l + 1);
    }
    st.depth = depth;
    // The same with the steps of limb recurrences -- an Idiv / Mod / Shr / Band of a sum, that sum, a product under it -- at a
    // tenth of a level: what the dependency depth comes to once such chains run as scan bundles (tile widths up to
    // SCAN_MAX_T; an estimate, used by the runtime to bound a program's size before anything is compiled).
    {
        std::vector<uint8_t> step(N, 0);
        for (size_t i = N; i-- > 0;) {
            const Node& n = g.nodes[i];
            if (n.kind != N_DUO) continue;
            if ((n.op == OP_IDIV || n.op == OP_MOD || n.op == OP_SHR || n.op == OP_BAND) && g.nodes[n.a].kind == N_DUO && g.nodes[n.a].op == OP_ADD) step[i] = step[n.a] = 1;
            if (n.op == OP_ADD && step[i])
                for (uint32_t o : {n.a, n.b})
                    if (g.nodes[o].kind == N_DUO && g.nodes[o].op == OP_MUL) step[o] = 1;
        }
        // (round 5) the one-bit recurrences of multi-register integers (rewrite.cc detect_bit_scans): a selection on an ordered comparison,
        // the comparison, and the short sums / differences under them (the arms x - y - bin, the comparand y + bin)
        {
            std::vector<uint8_t> reach(N, 0);  // how many more levels of Add / Sub below this node count as part of a step
            auto is_lin = [&](uint32_t o) { return g.nodes[o].kind == N_DUO && (g.nodes[o].op == OP_ADD || g.nodes[o].op == OP_SUB); };
            for (size_t i = N; i-- > 0;) {
                const Node& n = g.nodes[i];
                if (n.kind == N_TRES && g.nodes[n.a].kind == N_DUO && (g.nodes[n.a].op == OP_LT || g.nodes[n.a].op == OP_GT || g.nodes[n.a].op == OP_LEQ || g.nodes[n.a].op == OP_GEQ)) {
                    step[i] = step[n.a] = 1;
                    reach[n.a] = std::max<uint8_t>(reach[n.a], 1);
                    for (uint32_t o : {n.b, n.c})
                        if (is_lin(o)) reach[o] = std::max<uint8_t>(reach[o], 3);
                }
                if (n.kind != N_DUO || !reach[i]) continue;
                if (is_lin((uint32_t)i)) step[i] = 1;
                for (uint32_t o : {n.a, n.b})
                    if (is_lin(o)) reach[o] = std::max<uint8_t>(reach[o], (uint8_t)(is_lin((uint32_t)i) ? reach[i] - 1 : 1));
            }
        }
        std::vector<float> lf(N, 0.0f);
        float deepest = 0.0f;
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            const int ar = arity_of(n);
            if (!ar) continue;
            float l = lf[n.a];
            if (ar >= 2) l = std::max(l, lf[n.b]);
            if (ar >= 3) l = std::max(l, lf[n.c]);
            lf[i] = l + (step[i] ? 0.1f : 1.0f);
            deepest = std::max(deepest, lf[i]);
        }
        st.depth_scan = (uint64_t)deepest + 1;
    }

    phase("levels");
    if (probe_only) {
        for (const Node& n : g.nodes)  // (nodes per class of the graph as loaded: the runtime asks whether there are divisions)
            if (n.kind != N_CONST && class_of(n) >= 0) st.class_nodes[class_of(n)]++;
        out.n_inputs = (uint32_t)n_in_buf;
        out.n_witness = (uint32_t)g.witness_signals.size();
        return true;
    }
    // ---- load-time re-optimiser (the statistics above describe the graph as loaded) ----
    if (!getenv("CWC_NO_LOAD_OPTIMIZE")) {
        OptimizeStats os;
        if (const char* e = getenv("CWC_RANDOM_EVAL"))
            if (atoi(e) != 0) random_eval_passes(g, &os);
        optimize_loaded_graph(g, &os);
        N = g.nodes.size();
        st.n_folded = os.folded + os.random_constants;
        st.n_numbered = os.numbered + os.constants_merged + os.random_numbered;
        st.n_shaken = os.shaken;
        phase("load-time optimiser");
    }
    // ---- exact depth-reducing rewrite ----
    if (bit_fusion && !getenv("CWC_NO_BIT_FUSION")) {
        fuse_bit_extract(g);
        N = g.nodes.size();
        for (const Node& n : g.nodes) st.n_bitx_nodes += n.kind == N_DUO && n.op == OP_BITX;
        phase("bit-extract fusion");
    }
    // scheduling weights by class: linear-heavy graphs (more Add / Sub than Mul nodes) take the heavier linear weight
    class_cost = weight_table;
    {
        size_t n_lin = 0, n_mul = 0;
        for (const Node& n : g.nodes) {
            n_lin += n.kind == N_DUO && (n.op == OP_ADD || n.op == OP_SUB);
            n_mul += n.kind == N_DUO && n.op == OP_MUL;
        }
        if (n_lin > n_mul && !getenv("CWC_NO_LIN_HEAVY_WEIGHTS")) class_cost = kClassCostLinHeavy;
    }
    if (getenv("CWC_SCHED_LIN_COST") || getenv("CWC_SCHED_MUL_COST")) {  // (A/B knobs for the priority weights)
        for (int c = 0; c < (int)C_COUNT; ++c) cost_override[c] = class_cost[c];
        if (const char* e = getenv("CWC_SCHED_LIN_COST")) cost_override[C_LIN] = (uint32_t)atoi(e);
        if (const char* e = getenv("CWC_SCHED_MUL_COST")) cost_override[C_MUL] = (uint32_t)atoi(e);
        class_cost = cost_override;
    }
    if (G > 1 && !getenv("CWC_NO_TREE_REDUCTION")) {
        // whole chains at T = 1 (small batches: depth is everything); at most 8 leaves per tree otherwise, where the
        // extra nodes of wide trees cost lanes and memory traffic (measured on sha256_512: 293 k vs 265 k wit/s at 4096 sets)
        size_t leaves = G >= 64 ? 64 : 8;
        if (const char* e = getenv("CWC_TREE_LEAVES")) leaves = (size_t)std::max(2, atoi(e));  // (A/B knob)
        reduce_tree_height(g, leaves, class_cost);
        N = g.nodes.size();
    }
    for (const Node& n : g.nodes) st.n_op_compiled += arity_of(n) ? 1 : 0;
    if (cache && class_cost != cost_override) cache->entries.push_back(RewriteCache::Entry{bit_fusion, weight_table, G, g, st, class_cost});
    }  // (!hit)
    if (cache_lock.owns_lock()) cache_lock.unlock();

    phase("rewrites");
    // ---- one form per value: Montgomery or canonical (inserts the conversions; see infer_representations) ----
    std::vector<uint8_t> node_rep, node_vflags;
    // (limb-arithmetic graphs -- the probe's scan-aware depth is well below the plain one -- at tile widths with the MODE 2 instances)
    // (The interpreter instances with scan / convolution / canonical-product paths, like the ones with fused narrow bundles, exist for
    // programs with no or one divider wave per interpreter: kernels.hip launch_interp.)
    const bool mode2_ok = T <= SCAN_MAX_T && G >= 2 && divider <= 1;
    const bool limb_graph = mode2_ok && !getenv("CWC_NO_SCAN") && st.depth_scan * 10 < st.depth * 8;
    // bit graphs (sha256-like: one operation in thirty-two or more is a bit extract): canonical inputs, every product canonical
    const bool bit_graph = mode2_ok && !getenv("CWC_NO_BIT_GRAPH") && !policy.all_montgomery && st.n_bitx_nodes * 32 >= st.n_op && st.n_op > 0;
    uint64_t n_mul_cc = 0;
    infer_representations(g, node_rep, node_vflags, st.n_conversions, st.n_canonical, policy.all_montgomery, (limb_graph || bit_graph) && !getenv("CWC_NO_MUL_CC"), n_mul_cc,
                          bit_graph);
    N = g.nodes.size();
    phase("representation inference");
    // ---- scan chains: the steps of serial limb recurrences as pairs of N_SCAN nodes (class C_SCAN) ----
    std::vector<uint32_t> scan_imm, scan_partner;
    if (mode2_ok && !getenv("CWC_NO_SCAN") && !policy.no_scans) {
        detect_scans(g, node_rep, node_vflags, scan_imm, scan_partner, st.n_scan_steps);
        // borrow chains / most-significant-difference comparisons of multi-register integers (limb graphs: the step kinds live in the MODE 2 instances)
        if (limb_graph) detect_bit_scans(g, node_rep, node_vflags, scan_imm, scan_partner, st.n_scan_steps);
        // schoolbook limb products: the column sums of a k x k block as one bundle (2k - 1 columns, one node slot each)
        if (n_mul_cc && !policy.no_conv && !getenv("CWC_NO_CONV")) detect_convolutions(g, node_rep, node_vflags, scan_imm, scan_partner, G, st.n_conv_products);
        N = g.nodes.size();
        phase("scan chains");
    }
    if (policy.fuse && policy.fill && T <= COOP_FUSE_MAX_T && G > 1 && divider <= 1 && st.n_scan_steps == 0 && n_mul_cc == 0) {  // (a program has fused bundles or scan bundles: one interpreter instance each)
        fuse_narrow_chains(g, node_rep, node_vflags, class_cost, (policy.fuse & 0xffffu) - 1, (policy.fuse & 0x10000u) != 0, st.n_fused_nodes);
        N = g.nodes.size();
        phase("fused narrow chains");
    }
    // ---- the floor of this execution model: the graph's longest dependent chain priced at the best measured latency of
    // each operation on a lone wave, as pure arithmetic without any bundle's front end (bench.py roofline.chain) ----
    {
        // shader cycles: the four-lane product 704 (profiles/r03_ubench_coop_mul.txt), an addition 290, the safegcd inversion
        // 52 200 + its product (r03_inv_bench.txt), Fr::new of an input = one full-width product 1 436; integer classes at their
        // arithmetic on canonical operands; a round of a scan loop
        auto floor_cycles = [&](const Node& n) -> double {
            switch (class_of(n)) {
                case C_INPUT: return 1436;
                case C_MUL: return 704;
                case C_LIN: return 290;
                case C_DIV: return 52200 + 704;
                case C_CMPZ: return 100;
                case C_CMPS: return 400;
                case C_BIT: return 300;
                case C_IDIVMOD: return 1500;
                case C_TERN: return 100;
                case C_MULF: return 704.0 * (1 + (fused_op2(n.op) == FOP_MUL ? 1 : 0)) + 290.0 * ((fused_op2(n.op) > FOP_MUL ? 1 : 0) + (fused_op3(n.op) ? 1 : 0));
                case C_SCAN: return n.kind == N_CONV ? kCyclesConvStep * 32 : (n.op & SCAN_OP_DIV) ? kCyclesScanStepDiv : scan_is_sel(n.op) ? 150.0 : (n.op & (SCAN_OP_BORROW | SCAN_OP_LEX)) ? 20.0 : kCyclesScanStepCarry;  // (the serial rounds; chains of 64-bit limbs beat this "floor" with the parallel forms)
                default: return 0;
            }
        };
        std::vector<float> fin(N, 0.0f);
        double longest = 0;
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            if (n.kind == N_CONST) continue;
            const uint32_t ops[3] = {n.a, n.b, n.c};
            float t = 0;
            for (int q = 0; q < arity_of(n); ++q) t = std::max(t, fin[ops[q]]);
            fin[i] = t + (float)floor_cycles(n);
            longest = std::max(longest, (double)fin[i]);
        }
        st.chain_floor_cycles = (uint64_t)longest;
    }
    // ---- constants -> table (Montgomery form), node -> ref ----
    std::vector<uint32_t> ref(N, 0);  // for consts: REF_CONST|idx ; for others: slot (filled later)
    for (size_t i = 0; i < N; ++i)
        if (g.nodes[i].kind == N_CONST) {
            Fr m = fr_to_mont(g.const_values[g.nodes[i].a]);
            ref[i] = REF_CONST | (uint32_t)(out.consts.size() / 8);
            out.consts.insert(out.consts.end(), m.v, m.v + 8);
        }
    st.n_const = out.consts.size() / 8;
    // canonical (non-Montgomery) copies of the constants that are read in canonical form: operands of integer-class
    // nodes (shift amounts, masks, divisors, bounds) and of additions / selections / equality tests of canonical values
    std::unordered_map<uint32_t, uint32_t> canon_const;  // constant node -> table index of its canonical copy
    auto reads_canonical_constants = [&](size_t i, int q) -> bool {  // operand q of node i, a constant: which copy?
        const Node& n = g.nodes[i];
        const int c = class_of(n);
        if (is_integer_class(c)) return q < 2 && !(n.op == OP_BITX && q == 1);
        if (c == C_CMPZ) return (n.op == OP_EQ || n.op == OP_NEQ) && (node_vflags[i] & VF_A_CANON);
        if (c == C_LIN) return node_rep[i] == REP_C;
        if (c == C_MULF) {  // the operand of an addition stage follows the node's form; a factor of a product is a Montgomery constant
            const bool lin_operand = fused_sq(n.op) ? (q == 1 ? fused_op2(n.op) > FOP_MUL : q == 2) : (q == 2 && fused_op2(n.op) > FOP_MUL);
            return lin_operand && node_rep[i] == REP_C;
        }
        if (c == C_TERN) return q >= 1 && node_rep[i] == REP_C;
        if (c == C_SCAN && n.kind == N_SCAN && scan_is_sel(n.op))  // a selection moves words: its arms in the form of its value, a comparison's operands canonical, a condition either way
            return (n.op & SCAN_OP_ACC) ? node_rep[i] == REP_C : true;
        if (c == C_SCAN) return true;  // x, the accumulator, the divisor: canonical integers
        if (c == C_MUL) return (node_vflags[i] & VF_MUL_CC) != 0;  // canonical products (bit graphs: with a constant's canonical copy)
        return false;  // Mul / Div: Montgomery form
    };
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) {
            const uint32_t o = ops[q];
            if (g.nodes[o].kind != N_CONST || canon_const.count(o) || !reads_canonical_constants(i, q)) continue;
            canon_const[o] = (uint32_t)(out.consts.size() / 8);
            const Fr& v = g.const_values[g.nodes[o].a];
            out.consts.insert(out.consts.end(), v.v, v.v + 8);
        }
    }
    const uint32_t zero_const = (uint32_t)(out.consts.size() / 8);  // index of the trailing dummy (value 0)
    out.consts.insert(out.consts.end(), 8, 0u);  // trailing dummy entry: the table is never empty (prefetch target)
    out.n_const = (uint32_t)(out.consts.size() / 8);

    phase("constants");
    // ---- schedule: order of evaluated nodes (inputs + ops) and bundle boundaries ----
    // G == 1: file order (the reference's own loop order; best locality, every bundle is one node anyway).
    // G  > 1: list scheduling.  One bundle = up to G ready nodes of ONE class; a node is ready when all its
    // producers sit in earlier bundles.  The class of the next bundle is that of the ready node with the longest
    // cost-weighted path to a sink (critical path first); nodes with slack wait until their class comes up, so
    // chains that are at different op classes in the same dependency level share bundles across levels instead
    // of costing one bundle per (level, class).
    std::vector<uint32_t> order;
    order.reserve(N);
    std::vector<uint32_t> bundle_of(N, 0xffffffffu);      // bundle that produces the node's value
    std::vector<uint32_t> use_bundle_of(N, 0xffffffffu);  // bundle that reads the node's operands (differs for a
                                                          // division handed to the divider wave: request vs. collect)
    std::vector<uint32_t> bundle_start;  // index into order
    std::vector<uint8_t> bundle_coop;    // 1: narrow multiplication bundle (C_MULQ: four lanes per product), 2: fused narrow bundle
    std::vector<uint32_t> order_pos;     // record position of every entry of `order` inside its bundle
    std::vector<uint32_t> bundle_flags;  // HDR_POST / HDR_WAIT (programs of several streams)
    static const uint32_t REQ_FLAG = 0x80000000u;         // order[] entry: the request half of a division
    // Streams: the graph's independent parts (components that share nothing but Input nodes and constants) can be
    // evaluated by different wavefronts of one tile, each with its own bundle sequence.  Stream 0 also evaluates every
    // Input node first (the prologue) and then posts; the other streams begin with a wait for that post.
    std::vector<uint8_t> stream_of(N, 0);
    uint32_t P = 1;
    uint32_t s_first[MAX_STREAMS] = {0, 0, 0, 0}, s_count[MAX_STREAMS] = {0, 0, 0, 0}, s_div[MAX_STREAMS] = {0, 0, 0, 0};
    uint32_t s_cref[MAX_STREAMS] = {0, 0, 0, 0};  // rows of the third-operand / input-index table in front of each stream
    double s_chain[MAX_STREAMS] = {0, 0, 0, 0};  // longest dependent chain of each stream, lone-wave cycles (divisions at the divider wave's latency)
    if (G == 1) {
        for (size_t i = 0; i < N; ++i)
            if (g.nodes[i].kind != N_CONST) {
                bundle_of[i] = use_bundle_of[i] = (uint32_t)bundle_start.size();
                bundle_start.push_back((uint32_t)order.size());
                bundle_coop.push_back(0);
                bundle_flags.push_back(0);
                order.push_back((uint32_t)i);
                order_pos.push_back(0);
            }
        s_count[0] = (uint32_t)bundle_start.size();
    } else {
        std::vector<uint64_t> height(N, 0);
        std::vector<std::vector<uint32_t>> users;  // adjacency (only non-const producers)
        users.resize(N);
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            const uint32_t ops[3] = {n.a, n.b, n.c};
            uint32_t seen[3];
            int ns = 0;
            for (int q = 0; q < arity_of(n); ++q) {
                const uint32_t o = ops[q];
                if (g.nodes[o].kind == N_CONST) continue;
                bool dup = false;
                for (int z = 0; z < ns; ++z) dup |= seen[z] == o;
                if (dup) continue;
                seen[ns++] = o;
                users[o].push_back((uint32_t)i);
            }
        }
        // (a selection step's ACC value depends on the operands of its OUT node -- the condition, the comparison's operands -- though it does not
        // name them: the OUT node is as urgent as the ACC node, or whatever computes the condition would be scheduled as if nothing waited for it)
        const bool any_sel = !scan_partner.empty();
        for (size_t i = N; i-- > 0;) {
            if (g.nodes[i].kind == N_CONST) continue;
            uint64_t h = 0;
            for (uint32_t u : users[i]) h = std::max(h, height[u]);
            const bool sel = any_sel && g.nodes[i].kind == N_SCAN && scan_is_sel(g.nodes[i].op);
            if (sel && !(g.nodes[i].op & SCAN_OP_ACC)) h = std::max(h, height[i]);  // (OUT in front of its ACC node: pre-set below)
            height[i] = h + node_cost(class_cost, g.nodes[i]);
            if (sel && (g.nodes[i].op & SCAN_OP_ACC)) {
                const uint32_t o = scan_partner[i];
                if (o > i) {  // the OUT node was visited already (a plain selection's: behind the graph's last node); its operands come later
                    height[o] = std::max(height[o], height[i]);
                } else {
                    height[o] = std::max(height[o], h);  // picked up when the loop reaches it
                }
            }
        }
        if (getenv("CWC_DEBUG_CRITICAL_PATH")) {  // diagnostic: class composition of the cost-weighted critical path
            uint32_t cur = 0xffffffffu;
            for (size_t i = 0; i < N; ++i)
                if (g.nodes[i].kind != N_CONST && (cur == 0xffffffffu || height[i] > height[cur])) cur = (uint32_t)i;
            uint64_t cnt[C_COUNT] = {0}, total = height[cur];
            std::string seq;
            while (true) {
                const int c = std::max(0, class_of(g.nodes[cur]));
                cnt[c]++;
                if (seq.size() < 400) seq += "IMLD?????T"[c < 10 ? c : 4];
                uint32_t nxt = 0xffffffffu;
                for (uint32_t u : users[cur])
                    if (nxt == 0xffffffffu || height[u] > height[nxt]) nxt = u;
                if (nxt == 0xffffffffu) break;
                cur = nxt;
            }
            fprintf(stderr, "critical path: cost %llu; nodes by class:", (unsigned long long)total);
            for (int c = 0; c < (int)C_COUNT; ++c)
                if (cnt[c]) fprintf(stderr, " %d:%llu", c, (unsigned long long)cnt[c]);
            fprintf(stderr, "\n  start: %s\n", seq.c_str());
        }
        // operations between a node and the nearest division that depends on it (saturating)
        static const uint32_t kFar = 0xffffu;
        // (measured 3 against 6 and 10: +1.4 % at 1024 sets and +2.6 % at 2048 with divider waves, +1.4 % at 8192 and
        // 16384 sets with inline inversions)
        uint32_t div_wait_ops = 3;
        if (const char* e = getenv("CWC_SCHED_DIV_WAIT")) div_wait_ops = (uint32_t)atoi(e);
        std::vector<uint16_t> dist_to_div(N, (uint16_t)kFar);
        for (size_t i = N; i-- > 0;) {
            if (g.nodes[i].kind == N_CONST) continue;
            if (class_of(g.nodes[i]) == C_DIV) {
                dist_to_div[i] = 0;
                continue;
            }
            uint32_t d = kFar;
            for (uint32_t u : users[i]) d = std::min<uint32_t>(d, dist_to_div[u] + 1u);
            dist_to_div[i] = (uint16_t)std::min<uint32_t>(d, kFar);
        }
        const bool tie_reverse = getenv("CWC_SCHED_TIE_REVERSE") != nullptr;
        const bool ride_along = !getenv("CWC_NO_RIDE_ALONG");
        // Scan chains: a step is scheduled as a unit (its OUT node stands for both), consecutive steps of a chain go into
        // consecutive pairs of ONE bundle.  A bundle's steps share kind and shift: one ready heap per (kind, shift).
        const bool has_scans = st.n_scan_steps != 0 || st.n_conv_products != 0;
        static const uint32_t kConvKey = 1u << 20;   // the heap of convolution groups (a group is named by its column-0 node)
        std::unordered_map<uint32_t, std::vector<uint32_t>> conv_members;  // column-0 node -> the group's nodes in column order
        std::vector<uint32_t> scan_keys;             // distinct (kind << 8 | shift)
        std::vector<uint32_t> scan_next;             // ACC node -> OUT node of the step that continues its chain
        auto scan_shift_of = [&](uint32_t i) -> uint32_t {
            if (!(g.nodes[i].op & SCAN_OP_DIV)) return scan_imm[i];
            const Fr& v = g.const_values[g.nodes[scan_imm[i]].a];  // the constant 2^k
            for (int w = 0; w < 8; ++w)
                if (v.v[w]) return 32u * w + (uint32_t)__builtin_ctz(v.v[w]);
            return 0;
        };
        auto scan_key_of = [&](uint32_t i) -> uint32_t { return (scan_kind_bits(g.nodes[i].op) << 8) | scan_shift_of(i); };  // (kind bits 0x02 .. 0xf0, a shift below 254)
        auto scan_key_index = [&](uint32_t key) -> int {
            for (size_t k = 0; k < scan_keys.size(); ++k)
                if (scan_keys[k] == key) return (int)k;
            return -1;
        };
        if (has_scans) {
            scan_next.assign(N, 0xffffffffu);
            for (size_t i = 0; i < N; ++i) {
                const Node& n = g.nodes[i];
                if (n.kind == N_CONV) {
                    if (scan_key_index(kConvKey) < 0) scan_keys.push_back(kConvKey);
                    std::vector<uint32_t>& m = conv_members[scan_partner[i]];
                    if (m.empty()) m.assign(2 * (scan_imm[i] >> 8) - 1, 0xffffffffu);
                    m[scan_imm[i] & 0xffu] = (uint32_t)i;
                    continue;
                }
                if (n.kind != N_SCAN || (n.op & SCAN_OP_ACC)) continue;
                const uint32_t key = scan_key_of((uint32_t)i);
                if (scan_key_index(key) < 0) scan_keys.push_back(key);
                const Node& pr = g.nodes[n.b];
                if (!(n.op & SCAN_OP_NOACC) && !scan_is_sel(n.op) && pr.kind == N_SCAN && (pr.op & SCAN_OP_ACC) && scan_key_of(n.b) == key && scan_next[n.b] == 0xffffffffu) scan_next[n.b] = (uint32_t)i;  // (a selection stands alone)
            }
        }
        // Narrow multiplication bundles: when no more multiplications are ready than four-lane products fit a wave, the
        // bundle is compiled for the lane-cooperative multiplier (about half the cycles of a full-width multiplication
        // bundle).
        const size_t coop_cap = policy.fill ? coop_nodes(T) : 0;
        const uint64_t coop_slack = policy.slack_levels == ~0u ? ~0ull : (uint64_t)policy.slack_levels * class_cost[C_MUL];
        const size_t coop_fill = policy.fill;

        // Programs of several streams: the prologue -- Input nodes and the operations within a short chain of them, which
        // the graph's parts tend to share (flags, key bits, common subexpressions) -- is evaluated by stream 0 before
        // anything else; it posts behind it, the other streams begin with a wait for that post.
        std::vector<uint8_t> prologue(N, 0);
        static const uint64_t kPrologueBoost = 1ull << 60;
        // One stream's bundle sequence (bundle indices relative to the stream's first bundle).
        struct StreamSched {
            std::vector<uint32_t> order, bundle_start, div_lanes;
            std::vector<uint32_t> order_pos;     // record position of every entry of `order` inside its bundle
            std::vector<uint8_t> bundle_coop;    // 0 full-width, 1 narrow multiplication bundle, 2 fused narrow bundle
            std::vector<uint32_t> bundle_flags;  // HDR_POST / HDR_WAIT: the bundle is a C_SYNC bundle
            uint64_t class_bundles[C_COUNT] = {0};
            uint32_t n_div_requests = 0;
            double cycles() const {
                double c = 0;
                for (int k = 0; k < (int)C_COUNT; ++k) c += kCycles[k] * (double)class_bundles[k];
                return c;
            }
        };
        // `so`: stream of every node; producers in another stream do not gate a node (the streams' phases do).  With
        // record = false nothing outside `ss` is written (pricing a candidate partition).
        auto schedule_stream = [&](uint32_t s, const std::vector<uint8_t>& so, StreamSched& ss, bool record, bool several) -> bool {
            std::vector<uint32_t> indeg(N, 0);
            size_t remaining = 0, prologue_left = 0;
            bool posted = !(several && s == 0);  // stream 0 of several: a post bundle right behind the last prologue node
            for (size_t i = 0; i < N; ++i) {
                if (g.nodes[i].kind == N_CONST) continue;
                for (uint32_t u : users[i])
                    if (so[u] == s && so[i] == s) indeg[u]++;
                remaining += so[i] == s;
                prologue_left += so[i] == s && prologue[i];
            }
            // ready heaps per class, keyed by (height, -index)
            typedef std::pair<uint64_t, uint32_t> Key;  // (height, ~index) so that ties prefer file order
            // (integer-class nodes: one heap per combination of operand / result forms, a bundle's header bits are uniform)
            const int NH = (int)C_COUNT * (17 + (int)scan_keys.size());  // (scan steps: heap C_SCAN + C_COUNT * (17 + key index))
            std::vector<std::vector<Key>> heap(NH);
            std::vector<uint8_t> placed;  // scan nodes that sit in a bundle already (a step's successor inside its own bundle is released with it)
            std::vector<uint8_t> sel_queued;  // selection steps that sit in their ready heap
            if (has_scans) placed.assign(N, 0);
            if (has_scans) sel_queued.assign(N, 0);
            std::unordered_map<uint32_t, uint32_t> conv_ready;  // group -> how many of its nodes have their operands
            auto push = [&](uint32_t i) {
                int hc = class_of(g.nodes[i]);
                if (hc == C_SCAN && g.nodes[i].kind == N_CONV) {  // a group goes into ONE bundle, once the last of its factors is there
                    const uint32_t head = scan_partner[i];
                    const std::vector<uint32_t>& m = conv_members.find(head)->second;
                    if (++conv_ready[head] < m.size()) return;
                    uint64_t hgt = 0;
                    for (uint32_t u : m) hgt = std::max(hgt, height[u]);
                    auto& hs = heap[(int)C_SCAN + (int)C_COUNT * (17 + scan_key_index(kConvKey))];
                    hs.push_back(Key(hgt + (prologue[head] ? kPrologueBoost : 0ull), tie_reverse ? head : ~head));
                    std::push_heap(hs.begin(), hs.end());
                    return;
                }
                if (hc == C_SCAN && scan_is_sel(g.nodes[i].op)) {  // a selection's two nodes name different operands: ready when both are
                    const uint32_t o = (g.nodes[i].op & SCAN_OP_ACC) ? scan_partner[i] : i;
                    if (placed[o] || sel_queued[o] || indeg[o] != 0 || indeg[scan_partner[o]] != 0) return;
                    sel_queued[o] = 1;
                    auto& hs = heap[(int)C_SCAN + (int)C_COUNT * (17 + scan_key_index(scan_key_of(o)))];
                    hs.push_back(Key(std::max(height[o], height[scan_partner[o]]) + (prologue[o] ? kPrologueBoost : 0ull), tie_reverse ? o : ~o));
                    std::push_heap(hs.begin(), hs.end());
                    return;
                }
                if (hc == C_SCAN) {  // the step's OUT node stands for the pair
                    if ((g.nodes[i].op & SCAN_OP_ACC) || placed[i]) return;
                    auto& hs = heap[(int)C_SCAN + (int)C_COUNT * (17 + scan_key_index(scan_key_of(i)))];
                    hs.push_back(Key(std::max(height[i], height[scan_partner[i]]) + (prologue[i] ? kPrologueBoost : 0ull), tie_reverse ? i : ~i));
                    std::push_heap(hs.begin(), hs.end());
                    return;
                }
                if (hc == C_MUL && (node_vflags[i] & VF_MUL_CC)) hc += (int)C_COUNT;  // (canonical products: bundles of their own, never narrow)
                else if (hc == C_BIT) hc += (int)C_COUNT * (1 + node_vflags[i] + 8 * (g.nodes[i].op == OP_SHR || g.nodes[i].op == OP_BAND ? 1 : 0));  // (bundles of Shr / Band nodes take a straight path)
                else if (is_integer_class(hc)) hc += (int)C_COUNT * (1 + node_vflags[i]);
                else if (hc == C_CMPZ) hc += (int)C_COUNT * (1 + (node_vflags[i] & VF_OUT_CANON));
                else if (hc == C_MULF) {  // fused nodes: one heap per combination of stages (a bundle runs every stage one of its nodes has)
                    const uint8_t op = g.nodes[i].op;
                    hc += (int)C_COUNT * (1 + ((fused_op2(op) == FOP_MUL ? 1 : 0) | (fused_op2(op) > FOP_MUL ? 2 : 0) | (fused_op3(op) ? 4 : 0)));
                }
                auto& h = heap[hc];
                h.push_back(Key(height[i] + (prologue[i] ? kPrologueBoost : 0ull), tie_reverse ? i : ~i));
                std::push_heap(h.begin(), h.end());
            };
            for (size_t i = 0; i < N; ++i)
                if (g.nodes[i].kind != N_CONST && so[i] == s && indeg[i] == 0) push((uint32_t)i);
            std::vector<uint32_t> picked;
            // Asynchronous divider: a division bundle is split into a request (operands to the divider wave) and, about
            // one inversion later on the scheduler's clock, a collect bundle with the same nodes in the same node slots;
            // the interpreter runs other ready work in between.  One request is in flight at a time.
            uint64_t clock = 0;
            std::vector<uint32_t> in_flight;  // nodes of the pending request
            uint64_t in_flight_ready = 0;
            // coop: 0 full-width, 1 narrow multiplication bundle (C_MULQ), 2 fused narrow bundle (C_MULF)
            auto emit_bundle = [&](const std::vector<uint32_t>& nodes, bool request, bool collect, int coop = 0, uint32_t sync_flags = 0) {
                const uint32_t b = (uint32_t)ss.bundle_start.size();
                ss.bundle_start.push_back((uint32_t)ss.order.size());
                ss.bundle_coop.push_back((uint8_t)coop);
                ss.bundle_flags.push_back(sync_flags);
                const int cl = sync_flags ? (int)C_SYNC : request ? (int)C_DIVREQ : collect && divider ? (int)C_DIVGET : coop == 2 ? (int)C_MULF : coop ? (int)C_MULQ : nodes.empty() ? (int)C_LIN : class_of(g.nodes[nodes[0]]);
                if ((unsigned)cl < (unsigned)C_COUNT) ss.class_bundles[cl]++;
                (void)b;
                if (record && getenv("CWC_DEBUG_SCHED") && b < (uint32_t)atoi(getenv("CWC_DEBUG_SCHED"))) {  // diagnostic: the first bundles, node by node
                    fprintf(stderr, "bundle %u class %d:", b, cl);
                    for (size_t q = 0; q < nodes.size() && q < 6; ++q) {
                        const Node& dn = g.nodes[nodes[q]];
                        fprintf(stderr, " [%u k%d op%d (%u,%u,%u) h%llu]", nodes[q], dn.kind, dn.op, dn.a, dn.b, dn.c, (unsigned long long)height[nodes[q]]);
                    }
                    fprintf(stderr, "%s\n", nodes.size() > 6 ? " ..." : "");
                }
                for (uint32_t i : nodes) {
                    if (prologue[i] && !request) --prologue_left;
                    if (request) {
                        if (record) use_bundle_of[i] = b;
                        ss.order.push_back(i | REQ_FLAG);
                    } else {
                        if (record) {
                            bundle_of[i] = b;
                            if (!collect) use_bundle_of[i] = b;
                        }
                        ss.order.push_back(i);
                    }
                }
                if (request) return;
                remaining -= nodes.size();
                for (uint32_t i : nodes)  // release users only now: a bundle never reads its own results (but for the steps of a scan bundle: push skips them)
                    for (uint32_t u : users[i])
                        if (so[u] == s && --indeg[u] == 0) push(u);
            };
            if (s != 0) {  // the wait for stream 0's post (the prologue's values), then two idle bundles: the staging loads of
                           // bundles 0 and 1 are issued before the loop and those of bundle 2 in front of the wait
                emit_bundle(picked, false, false, 0, HDR_WAIT);
                emit_bundle(picked, false, false);
                emit_bundle(picked, false, false);
            }
            while (remaining) {
                if (!posted && prologue_left == 0 && in_flight.empty()) {  // (the post's vmcnt(0) covers every store issued so far)
                    emit_bundle(std::vector<uint32_t>(), false, false, 0, HDR_POST);
                    posted = true;
                    continue;
                }
                int best = -1;
                for (int c = 0; c < NH; ++c)
                    if (!heap[c].empty() && (best < 0 || heap[c].front() > heap[best].front())) best = c;
                if (!in_flight.empty()) {
                    // collect when the quotients are due, or when nothing else can run (the interpreter then waits)
                    bool other_ready = false;
                    for (int c = 0; c < NH; ++c) other_ready |= c != C_DIV && !heap[c].empty();
                    if (clock >= in_flight_ready || !other_ready) {
                        emit_bundle(in_flight, false, true);
                        ss.div_lanes.push_back((uint32_t)in_flight.size() * T);
                        in_flight.clear();
                        ss.n_div_requests++;
                        clock += kClockCost[C_DIVGET];
                        continue;
                    }
                    if (best == C_DIV) {  // a second request has to wait for the first one: run the best other class
                        best = -1;
                        for (int c = 0; c < NH; ++c)
                            if (c != C_DIV && !heap[c].empty() && (best < 0 || heap[c].front() > heap[best].front())) best = c;
                    }
                }
                if (best < 0) {
                    err = kErrSchedulerDeadlock;
                    return false;
                }
                // An inversion bundle costs about thirty multiplication bundles however few of its lanes are used, and a
                // wave's time is the sum of its bundles: a ready division waits while another chain is within a few
                // operations of its own division (its ready node goes first), so that sibling chains divide together.
                if (best == C_DIV) {
                    int other = -1;
                    for (int c = 0; c < NH; ++c) {
                        if (c == C_DIV || heap[c].empty()) continue;
                        // the heap top is the class's most urgent node; scan the ready nodes of the class for one that
                        // is about to reach a division
                        bool near = false;
                        for (const Key& k : heap[c]) near |= dist_to_div[tie_reverse ? k.second : ~k.second] <= div_wait_ops;
                        if (near && (other < 0 || heap[c].front() > heap[other].front())) other = c;
                    }
                    if (other >= 0) best = other;
                }
                // A scan bundle costs its front end however few steps it runs, and a wave's time is the sum of its bundles: while the
                // chain of the most urgent ready step goes on with steps whose other operands are not computed yet, anything else
                // that is ready runs first (it has to run anyway), so that chains go into few, full bundles.
                const int conv_heap = scan_key_index(kConvKey) < 0 ? -1 : (int)C_SCAN + (int)C_COUNT * (17 + scan_key_index(kConvKey));
                if (best % (int)C_COUNT == (int)C_SCAN && best >= (int)C_COUNT * 17 && best != conv_heap && !getenv("CWC_SCAN_EAGER")) {
                    const uint32_t head = tie_reverse ? heap[best].front().second : ~heap[best].front().second;
                    size_t len_ready = 1, len_all = 1;
                    bool contiguous = true;
                    for (uint32_t cur = head; len_all < G / 2; ++len_all) {
                        const uint32_t nx = scan_next[scan_partner[cur]];
                        if (nx == 0xffffffffu || so[nx] != s || placed[nx]) break;
                        contiguous = contiguous && indeg[nx] == 1 && indeg[scan_partner[nx]] == 1;  // (one producer left: the accumulator; users[] holds a user once per producer)
                        len_ready += contiguous;
                        cur = nx;
                    }
                    if (len_ready < len_all) {
                        int other = -1;
                        for (int c = 0; c < NH; ++c)
                            if (c % (int)C_COUNT != (int)C_SCAN && !heap[c].empty() && !(c == C_DIV && !in_flight.empty()) && (other < 0 || heap[c].front() > heap[other].front())) other = c;
                        if (other >= 0) best = other;
                    }
                }
                // INPUT nodes first whenever any is ready (they have no producers and feed everything)
                if (!heap[C_INPUT].empty()) best = C_INPUT;
                picked.clear();
                auto& h = heap[best];
                if (best == conv_heap) {  // the columns of one limb product, position c = column c
                    std::pop_heap(h.begin(), h.end());
                    const uint32_t head = tie_reverse ? h.back().second : ~h.back().second;
                    h.pop_back();
                    picked = conv_members.find(head)->second;
                    emit_bundle(picked, false, false);
                    clock += 14 + 70;
                    continue;
                }
                if (best % (int)C_COUNT == (int)C_SCAN && best >= (int)C_COUNT * 17) {
                    // the most urgent ready step and, pair after pair, the steps that continue its chain -- as far as every other
                    // operand of theirs was produced by an earlier bundle --, then the next ready chain of the same kind
                    const size_t cap_steps = G / 2;
                    auto in_bundle = [&](uint32_t x) { return std::find(picked.begin(), picked.end(), x) != picked.end(); };
                    uint32_t longest = 0;
                    while (picked.size() / 2 < cap_steps && !h.empty()) {
                        std::pop_heap(h.begin(), h.end());
                        uint32_t cur = tie_reverse ? h.back().second : ~h.back().second;
                        h.pop_back();
                        uint32_t run = 0;
                        for (;;) {
                            picked.push_back(cur);
                            picked.push_back(scan_partner[cur]);
                            placed[cur] = placed[scan_partner[cur]] = 1;
                            ++run;
                            if (picked.size() / 2 >= cap_steps) break;
                            const uint32_t nx = scan_next[scan_partner[cur]];
                            if (nx == 0xffffffffu || so[nx] != s || placed[nx]) break;
                            const Node& nn = g.nodes[nx];
                            if (indeg[nx] != 1 || indeg[scan_partner[nx]] != 1) break;
                            if ((!(nn.op & SCAN_OP_NOX) && in_bundle(nn.a)) || (scan_has_third(nn.op) && in_bundle(nn.c))) break;  // (x or the divisor / subtrahend / comparand comes out of this very bundle)
                            cur = nx;
                        }
                        longest = std::max(longest, run);
                    }
                    emit_bundle(picked, false, false);
                    clock += 14 + (uint64_t)longest * scan_cost50(g.nodes[picked[0]].op);
                    continue;
                }
                // a request must fit the interpreter's mailbox (mbox_lanes active lanes = node slots x T)
                const bool fused = best >= (int)C_COUNT && best % (int)C_COUNT == (int)C_MULF;  // fused narrow bundle: at most coop_nodes(T) nodes
                const size_t cap = best == C_DIV && divider ? std::max<size_t>(1, std::min<size_t>(G, mbox_lanes(divider) / T)) : fused ? (size_t)coop_nodes(T) : G;
                bool coop = false;
                if (best == C_MUL && coop_cap) {
                    // Narrow or full-width?  The ready multiplications in priority order; the ones within `coop_slack` of the
                    // most urgent node's height cannot wait.  If they fit a narrow bundle it is one (cheapest step for the
                    // critical chain; its free groups take the next most urgent multiplications, then linear riders) and the
                    // rest stays ready: work with slack piles up until it becomes urgent itself and then fills full-width
                    // bundles properly (a full-width bundle costs the same with 10 or 32 nodes).
                    std::vector<Key> cand;
                    while (!h.empty() && cand.size() < G) {
                        std::pop_heap(h.begin(), h.end());
                        cand.push_back(h.back());
                        h.pop_back();
                    }
                    size_t n_urgent = 0;
                    while (n_urgent < cand.size() && (coop_slack >= cand[0].first || cand[n_urgent].first >= cand[0].first - coop_slack)) ++n_urgent;
                    coop = n_urgent <= coop_cap && (cand.size() <= coop_cap || cand.size() < coop_fill);
                    const size_t take = coop ? std::min(coop_cap, cand.size()) : cand.size();
                    for (size_t q = 0; q < cand.size(); ++q) {
                        if (q < take) {
                            picked.push_back(tie_reverse ? cand[q].second : ~cand[q].second);
                        } else {
                            h.push_back(cand[q]);
                            std::push_heap(h.begin(), h.end());
                        }
                    }
                }
                while (!coop && !h.empty() && picked.size() < cap) {
                    std::pop_heap(h.begin(), h.end());
                    picked.push_back(tie_reverse ? h.back().second : ~h.back().second);
                    h.pop_back();
                }
                std::sort(picked.begin(), picked.end());
                if (fused && picked.size() < cap && !heap[C_MUL].empty()) {  // free groups of a fused bundle take ready plain multiplications
                    auto& hm = heap[C_MUL];
                    std::vector<uint32_t> extra;
                    while (!hm.empty() && picked.size() + extra.size() < cap) {
                        std::pop_heap(hm.begin(), hm.end());
                        extra.push_back(tie_reverse ? hm.back().second : ~hm.back().second);
                        hm.pop_back();
                    }
                    std::sort(extra.begin(), extra.end());
                    picked.insert(picked.end(), extra.begin(), extra.end());
                }
                // A wave's time is the sum of its bundles and a multiplication bundle costs the same however few of its
                // node slots are used: ready Add/Sub nodes ride in its free slots (the kernel then also runs the ~40-slot
                // linear body, header bits) instead of asking for a bundle of their own later.
                const size_t slots = coop ? coop_cap : G;  // (a narrow bundle takes riders too: groups of four lanes add / subtract)
                if (best == C_MUL && picked.size() < slots && !heap[C_LIN].empty() && ride_along) {
                    auto& hl = heap[C_LIN];
                    std::vector<uint32_t> riders;
                    while (!hl.empty() && picked.size() + riders.size() < slots) {
                        std::pop_heap(hl.begin(), hl.end());
                        riders.push_back(tie_reverse ? hl.back().second : ~hl.back().second);
                        hl.pop_back();
                    }
                    std::sort(riders.begin(), riders.end());
                    picked.insert(picked.end(), riders.begin(), riders.end());  // multiplications first: they name the class
                }
                if (best == C_DIV && divider) {
                    emit_bundle(picked, true, false);
                    in_flight = picked;
                    clock += kClockCost[C_DIVREQ];
                    in_flight_ready = clock + div_cost50();
                } else {
                    emit_bundle(picked, false, false, fused ? 2 : coop ? 1 : 0);
                    clock += cost_of(kClockCost, coop ? (int)C_MULQ : best % (int)C_COUNT);
                }
            }
            if (!posted) emit_bundle(std::vector<uint32_t>(), false, false, 0, HDR_POST);
            return true;
        };


        // record positions: a bundle's nodes sit at positions 0, 1, .. in the order the scheduler picked them
        auto number_positions = [&](StreamSched& ss) {
            const uint32_t nb = (uint32_t)ss.bundle_start.size();
            ss.order_pos.resize(ss.order.size());
            for (uint32_t b = 0; b < nb; ++b) {
                const uint32_t e = b + 1 < nb ? ss.bundle_start[b + 1] : (uint32_t)ss.order.size();
                for (uint32_t k = ss.bundle_start[b]; k < e; ++k) ss.order_pos[k] = k - ss.bundle_start[b];
            }
        };

        // ---- partition into streams ----
        if (streams > 1 && (divider == 0 || divider == 1)) {
            // components of the operation nodes (edges through Input nodes and constants do not connect)
            std::vector<uint32_t> parent(N);
            for (size_t i = 0; i < N; ++i) parent[i] = (uint32_t)i;
            auto find = [&](uint32_t x) {
                while (parent[x] != x) x = parent[x] = parent[parent[x]];
                return x;
            };
            auto is_op = [&](uint32_t i) { return arity_of(g.nodes[i]) != 0; };
            // per component: the longest dependent chain and the summed work, both in lone-wave cycles (a multiplication
            // on a chain is a narrow bundle where the tile width has them; a division is its request, the inversion and
            // its collect bundle)
            const bool narrow = coop_cap != 0;
            auto node_cycles = [&](int c) -> double {
                if (c == C_MUL) return narrow ? kCycles[C_MULQ] : kCycles[C_MUL];
                if (c == C_MULF) return kCycles[C_MULF];
                if (c == C_SCAN) return 0.5 * (kCyclesScanStepCarry + kCyclesScanStepDiv);  // (a step's round of the loop)
                if (c == C_DIV && divider) return kCycles[C_DIV] + kCycles[C_DIVREQ] + kCycles[C_DIVGET];
                return kCycles[c];
            };
            double theta = 30000;  // (cycles of dependent operations from the inputs that still count as prologue)
            if (const char* e = getenv("CWC_STREAM_PROLOGUE")) theta = atof(e);
            std::vector<double> cp(N, 0);
            for (size_t i = 0; i < N; ++i) {
                const Node& n = g.nodes[i];
                if (n.kind == N_INPUT) prologue[i] = 1;
                if (!is_op((uint32_t)i)) continue;
                const uint32_t ops[3] = {n.a, n.b, n.c};
                double m = 0;
                for (int q = 0; q < arity_of(n); ++q) m = std::max(m, cp[ops[q]]);
                cp[i] = m + node_cycles(class_of(n));
                prologue[i] = cp[i] <= theta;
                // (a selection step's two nodes name different operands and sit in one bundle: the later one brings both to the longer chain --
                // one of them in the prologue and the other in a stream of its own would tear the bundle apart)
                if (n.kind == N_SCAN && scan_is_sel(n.op)) {
                    const uint32_t o = scan_partner[i];
                    if (n.op & SCAN_OP_ACC) {
                        // the ACC node decides for both: ITS users come behind it in node order and take their chain from it, so it must know
                        // the condition's chain now -- the OUT node's operands all precede this node (they were the selection's, or its
                        // comparison's, operands), whether the OUT node itself sits in front of it or behind the graph's last node
                        const Node& on = g.nodes[o];
                        const uint32_t oops[3] = {on.a, on.b, on.c};
                        double mo = 0;
                        for (int q = 0; q < arity_of(on); ++q) mo = std::max(mo, cp[oops[q]]);
                        cp[i] = std::max(cp[i], mo + node_cycles(class_of(n)));
                        prologue[i] = cp[i] <= theta;
                        if (o < i) {
                            cp[o] = cp[i];
                            prologue[o] = prologue[i];
                        }
                    } else if (o < i) {  // (an OUT node behind its ACC node)
                        cp[i] = cp[o];
                        prologue[i] = prologue[o];
                    }
                }
            }
            for (size_t i = 0; i < N; ++i) {
                if (!is_op((uint32_t)i) || prologue[i]) continue;
                const Node& n = g.nodes[i];
                const uint32_t ops[3] = {n.a, n.b, n.c};
                for (int q = 0; q < arity_of(n); ++q)
                    if (is_op(ops[q]) && !prologue[ops[q]]) {
                        const uint32_t ra = find((uint32_t)i), rb = find(ops[q]);
                        if (ra != rb) parent[ra] = rb;
                    }
                if (n.kind == N_SCAN || n.kind == N_CONV) {  // the two nodes of a step / the columns of a product sit in one bundle: one part (their operands may all be prologue values)
                    const uint32_t ra = find((uint32_t)i), rb = find(scan_partner[i]);
                    if (ra != rb) parent[ra] = rb;
                }
            }
            struct Comp { uint32_t root; double cp = 0, work = 0, alone = 0; uint64_t nodes = 0; };
            std::unordered_map<uint32_t, uint32_t> comp_index;
            std::vector<Comp> comps;
            for (size_t i = 0; i < N; ++i) {
                if (!is_op((uint32_t)i) || prologue[i]) continue;
                const Node& n = g.nodes[i];
                const int c = class_of(n);
                const uint32_t r = find((uint32_t)i);
                auto it = comp_index.find(r);
                if (it == comp_index.end()) {
                    it = comp_index.emplace(r, (uint32_t)comps.size()).first;
                    comps.push_back(Comp());
                    comps.back().root = r;
                }
                Comp& co = comps[it->second];
                co.cp = std::max(co.cp, cp[i]);
                const double cap = ((c == C_MUL && narrow) || c == C_MULF) ? (double)std::max<size_t>(1, coop_cap) : c == C_DIV && divider ? std::max(1.0, (double)mbox_lanes(divider) / T) : (double)G;
                co.work += node_cycles(c) / cap;
                co.nodes++;
            }
            for (Comp& co : comps) co.alone = std::max(co.cp, co.work);
            std::vector<uint32_t> by_size(comps.size());
            for (size_t k = 0; k < comps.size(); ++k) by_size[k] = (uint32_t)k;
            std::sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return comps[x].alone > comps[y].alone; });
            // longest first, each to the stream with the least load so far
            std::vector<uint8_t> comp_stream(comps.size(), 0);
            double load[MAX_STREAMS] = {0, 0, 0, 0};
            for (uint32_t k : by_size) {
                uint32_t to = 0;
                for (uint32_t s = 1; s < streams; ++s)
                    if (load[s] < load[to]) to = s;
                comp_stream[k] = (uint8_t)to;
                load[to] += comps[k].alone;
            }
            uint32_t used = 0;
            for (uint32_t s = 0; s < streams; ++s) used += load[s] > 0;
            if (getenv("CWC_DEBUG_STREAMS")) {
                fprintf(stderr, "streams T=%u: %zu components;", T, comps.size());
                for (size_t q = 0; q < by_size.size() && q < 10; ++q) {
                    const Comp& co = comps[by_size[q]];
                    fprintf(stderr, " [%llu nodes, chain %.2f M, work %.2f M -> %u]", (unsigned long long)co.nodes, co.cp / 1e6, co.work / 1e6, comp_stream[by_size[q]]);
                }
                fprintf(stderr, "\n");
            }
            if (used > 1) {
                P = streams;  // (a stream without a part stays empty: its wave ends at once)
                for (size_t i = 0; i < N; ++i)
                    if (is_op((uint32_t)i) && !prologue[i]) {
                        stream_of[i] = comp_stream[comp_index[find((uint32_t)i)]];
                        s_chain[stream_of[i]] = std::max(s_chain[stream_of[i]], cp[i]);
                    }
            } else {
                // one part only: the program would be the one-stream program of the same key.  Among the candidates of a
                // choice (shared rewrites) that sibling is being compiled anyway: refuse, the caller drops this key
                // (10.5 M nodes: half of the choice's compile work).
                if (cache && cache->lock) {
                    err = "the graph has one independent part: a stream program would equal the one-stream program";
                    return false;
                }
                std::fill(prologue.begin(), prologue.end(), 0);
            }
        }

        // ---- schedule every stream; a stream's first bundle index is a multiple of the pipeline depths ----
        for (uint32_t s = 0; s < P; ++s) {
            StreamSched ss;
            bool any = s == 0;
            for (size_t i = 0; i < N && !any; ++i) any = g.nodes[i].kind != N_CONST && stream_of[i] == s;
            if (!any) {
                s_first[s] = (uint32_t)bundle_start.size();
                continue;
            }
            if (!schedule_stream(s, stream_of, ss, true, P > 1)) return false;
            number_positions(ss);
            const uint32_t nb = (uint32_t)ss.bundle_start.size();
            const uint32_t base = (uint32_t)bundle_start.size();
            s_first[s] = base;
            s_count[s] = nb;
            s_div[s] = ss.n_div_requests;
            const uint32_t obase = (uint32_t)order.size();
            for (uint32_t b = 0; b < nb; ++b) {
                bundle_start.push_back(obase + ss.bundle_start[b]);
                bundle_coop.push_back(ss.bundle_coop[b]);
                bundle_flags.push_back(ss.bundle_flags[b]);
            }
            for (uint32_t e : ss.order) {
                const uint32_t i = e & ~REQ_FLAG;
                if (e & REQ_FLAG) {
                    use_bundle_of[i] += base;
                } else {
                    bundle_of[i] += base;
                    if (!(divider && class_of(g.nodes[i]) == C_DIV)) use_bundle_of[i] += base;
                }
            }
            order.insert(order.end(), ss.order.begin(), ss.order.end());
            order_pos.insert(order_pos.end(), ss.order_pos.begin(), ss.order_pos.end());
            out.div_lanes.insert(out.div_lanes.end(), ss.div_lanes.begin(), ss.div_lanes.end());
            out.n_div_requests += ss.n_div_requests;
            while (s + 1 < P && bundle_start.size() % 4 != 0) {  // idle bundles up to the next stream's first one (never executed)
                bundle_start.push_back((uint32_t)order.size());
                bundle_coop.push_back(0);
                bundle_flags.push_back(0);
            }
        }
    }
    const uint32_t NB = (uint32_t)bundle_start.size();
    bundle_start.push_back((uint32_t)order.size());
    out.n_bundles = NB;
    if ((uint64_t)NB * G * 16ull > 0xffffffffull) {  // the record stream is addressed through one 32-bit buffer window
        err = "graph too large: " + std::to_string(NB) + " bundles of " + std::to_string(G) + " records exceed the 4 GiB record window";
        return false;
    }

    phase("schedule");
    if (getenv("CWC_DEBUG_NODE_MIX")) {  // diagnostic: what the scheduled graph is made of -- per (class, operation): nodes, and how their operands were produced
        std::map<std::string, uint64_t> mix;
        static const char* kOps[] = {"Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq", "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor", "BitX"};
        auto name_of = [&](const Node& n) -> std::string {
            switch (n.kind) {
                case N_CONST: return "const";
                case N_INPUT: return "input";
                case N_UNO: return "Neg";
                case N_TRES: return "Tern";
                case N_FUSED: return "fused";
                case N_CONV: return "conv";
                case N_SCAN: return std::string(scan_is_sel(n.op) ? "sel" : (n.op & SCAN_OP_LEX) ? "lex" : (n.op & SCAN_OP_BORROW) ? "borrow" : (n.op & SCAN_OP_DIV) ? "sdiv" : "carry") + ((n.op & SCAN_OP_ACC) ? ".acc" : ".out");
                default: return n.op < sizeof kOps / sizeof *kOps ? kOps[n.op] : "?";
            }
        };
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            if (n.kind == N_CONST) continue;
            std::string key = name_of(n) + "(";
            const uint32_t ops[3] = {n.a, n.b, n.c};
            for (int q = 0; q < arity_of(n); ++q) key += (q ? ", " : "") + name_of(g.nodes[ops[q]]);
            mix[key + ")"]++;
        }
        std::vector<std::pair<uint64_t, std::string>> v;
        for (auto& kv : mix) v.push_back({kv.second, kv.first});
        std::sort(v.rbegin(), v.rend());
        fprintf(stderr, "node mix of the scheduled graph (T = %u):\n", T);
        for (size_t k = 0; k < v.size() && k < 60; ++k) fprintf(stderr, "  %8llu  %s\n", (unsigned long long)v[k].first, v[k].second.c_str());
    }
    // ---- operand routing -------------------------------------------------------------------------------
    // RING: produced at most RING_BUNDLES bundles ago (any node slot) -> read from the wave's result ring in LDS.
    // MEM : everything else (older values, constants, every third operand) -> its slot in the tile, staged into LDS
    //       OPND_AHEAD bundles ahead.  The staging load of bundle b is issued while bundle b - OPND_AHEAD runs, i.e.
    //       before that bundle stores: a MEM operand must be at least OPND_AHEAD + 1 bundles old, which the ring
    //       depth guarantees.  A value that is neither a witness element nor read through MEM is never given a
    //       slot (its store goes to the tile's trash slot).
    static_assert(RING_BUNDLES >= OPND_AHEAD, "values younger than the staging distance must come from the ring");
    std::vector<uint32_t> pos_in_bundle(N, 0);
    for (uint32_t b = 0; b < NB; ++b)
        for (uint32_t k = bundle_start[b]; k < bundle_start[b + 1]; ++k) pos_in_bundle[order[k] & ~REQ_FLAG] = order_pos[k];
    enum { SRC_MEM = 0, SRC_RING = 1 };
    auto route = [&](uint32_t producer, uint32_t consumer, int q) -> uint32_t {
        if ((q >= 2 && g.nodes[consumer].kind != N_FUSED && g.nodes[consumer].kind != N_SCAN) || g.nodes[producer].kind == N_CONST) return SRC_MEM;  // (TernCond reads its third operand in place)
        if (stream_of[producer] != stream_of[consumer]) return SRC_MEM;  // (another wave's ring)
        const uint32_t d = use_bundle_of[consumer] - bundle_of[producer];
        return (d >= 1 && d <= RING_BUNDLES) ? SRC_RING : SRC_MEM;
    };
    std::vector<uint32_t> last_mem_use(N, 0);  // last bundle that reads the value from memory
    std::vector<uint8_t> needs_slot(N, 0);
    for (uint32_t w : g.witness_signals)
        if (g.nodes[w].kind != N_CONST) needs_slot[w] = 2;  // pinned
    auto is_collect = [&](uint32_t entry) {  // the collect half of a division served by the divider wave: no operands
        return divider && !(entry & REQ_FLAG) && class_of(g.nodes[entry]) == C_DIV;
    };
    for (uint32_t e : order) {
        if (is_collect(e)) continue;
        const uint32_t i = e & ~REQ_FLAG;
        const Node& n = g.nodes[i];
        // Neg is encoded as 0 - a: its operand travels in the b position
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) {
            const uint32_t o = ops[q];
            if (n.kind == N_SCAN && q == 1 && g.nodes[o].kind != N_CONST && stream_of[o] == stream_of[i] && bundle_of[o] == use_bundle_of[i]) continue;  // (the accumulator arrives inside the bundle)
            if (g.nodes[o].kind == N_CONST || route(o, i, q) != SRC_MEM) continue;
            if (!needs_slot[o]) needs_slot[o] = 1;
            if (stream_of[o] != stream_of[i]) needs_slot[o] = 2;  // read by another stream: the slot is never reused
            last_mem_use[o] = std::max(last_mem_use[o], use_bundle_of[i]);
        }
    }

    phase("routing");
    // ---- slot allocation (LIFO free list: a just-freed slot is still hot in cache) + encoding ----
    // Slot numbering inside a tile: constants first (index = constant index), then value slots, then the trash slot.
    const uint64_t slot_bytes = 32ull * T;
    const uint32_t NC = out.n_const;
    out.hdr.resize(NB);
    out.recs.assign((size_t)NB * G * 4, 0);
    out.crefs.clear();  // one row of G words per C_INPUT / C_TERN bundle, in bundle order (the interpreter counts rows)
    uint32_t cref_row = 0;
    std::vector<uint32_t> free_slots;
    uint64_t stream_class_bundles[MAX_STREAMS][C_COUNT];
    memset(stream_class_bundles, 0, sizeof stream_class_bundles);
    uint64_t stream_bitx[MAX_STREAMS] = {0, 0, 0, 0}, stream_riders[MAX_STREAMS] = {0, 0, 0, 0};
    double stream_form_saved[MAX_STREAMS] = {0, 0, 0, 0};
    std::vector<uint32_t> dying;  // nodes whose slot is released after the current bundle
    uint32_t n_slots = 0;
    // Witness-ordered slots (policy.witness_slots): the pinned slot of a witness element is its rank among the witness
    // list's distinct nodes, so the output gather (pack kernel) reads consecutive memory and every 128-byte line it
    // fetches is used whole (tiles of one or two sets have 32- / 64-byte slots: with slots in schedule order the two
    // halves of a line are fetched at different times, 1.56 x the algorithmic read volume measured in round 2).  The
    // interpreter's stores / staging loads of such values then scatter: fine where it is bound by instruction issue,
    // 15 % slower on the wide, memory-heavier sha256 graph (round 1) -- hence a policy.
    // Default (round 4): on for tiles of one or two sets of graphs that are not linear-heavy -- measured neutral for the
    // authV2-class interpreter (12.54 ms either way at 1024 sets, profiles/r03_pack_ab.txt) while the pack kernel's reads drop
    // from 1.43 x to 1.0 x the algorithmic volume; CWC_WITNESS_SLOTS=0 / 1 forces either way.
    const bool witness_slots = getenv("CWC_WITNESS_SLOTS") ? policy.witness_slots : (T <= 2 && class_cost != kClassCostLinHeavy);
    std::vector<uint32_t> witness_rank;
    if (witness_slots) {
        witness_rank.assign(N, 0xffffffffu);
        for (uint32_t w : g.witness_signals)
            if (g.nodes[w].kind != N_CONST && witness_rank[w] == 0xffffffffu) witness_rank[w] = n_slots++;
    }
    // (CWC_NOWHERE=0: the zero constant's slot and the trash slot as before round 4, for A/B runs)
    const bool nowhere = !(getenv("CWC_NOWHERE") && atoi(getenv("CWC_NOWHERE")) == 0);
    const uint32_t zero_off = nowhere ? OFF_NOWHERE : (uint32_t)((uint64_t)zero_const * slot_bytes);
    auto mem_off = [&](uint32_t producer) -> uint64_t {
        if (g.nodes[producer].kind == N_CONST) return (uint64_t)(ref[producer] & ~REF_CONST) * slot_bytes;
        return ((uint64_t)NC + ref[producer]) * slot_bytes;
    };
    // (A DIV step names its constant 2^k only through the side table scan_imm -- the Mul node that read it is gone --, and the general path
    // multiplies with that constant's Montgomery form: whatever passes run between the rewrite and here must have kept and renumbered it.)
    for (size_t i = 0; i < N; ++i)
        if (g.nodes[i].kind == N_SCAN && (g.nodes[i].op & SCAN_OP_DIV)) {
            const uint32_t c = i < scan_imm.size() ? scan_imm[i] : 0xffffffffu;
            bool pow2 = c < N && g.nodes[c].kind == N_CONST;
            if (pow2) {
                const Fr& v = g.const_values[g.nodes[c].a];
                int bits = 0;
                for (int w = 0; w < 8; ++w) bits += __builtin_popcount(v.v[w]);
                pow2 = bits == 1;
            }
            if (!pow2) {
                err = "internal error: a division step's base is not a constant power of two";
                return false;
            }
        }
    auto scan_shift_of_node = [&](uint32_t i) -> uint32_t {  // CARRY: n; DIV: k of the constant 2^k
        if (!(g.nodes[i].op & SCAN_OP_DIV)) return scan_imm[i];
        const Fr& v = g.const_values[g.nodes[scan_imm[i]].a];
        for (int w = 0; w < 8; ++w)
            if (v.v[w]) return 32u * w + (uint32_t)__builtin_ctz(v.v[w]);
        return 0;
    };
    auto sub_of = [&](uint8_t op) -> uint32_t {
        switch (op) {
            case OP_ADD: return SUB_ADD;   case OP_SUB: return SUB_SUB;   case OP_MUL: return SUB_MULT;
            case OP_EQ: return SUB_EQ;     case OP_NEQ: return SUB_NEQ;   case OP_LAND: return SUB_LAND; case OP_LOR: return SUB_LOR;
            case OP_LT: return SUB_LT;     case OP_GT: return SUB_GT;     case OP_LEQ: return SUB_LEQ;   case OP_GEQ: return SUB_GEQ;
            case OP_SHL: return SUB_SHL;   case OP_SHR: return SUB_SHR;   case OP_BOR: return SUB_BOR;   case OP_BAND: return SUB_BAND;
            case OP_BXOR: return SUB_BXOR; case OP_IDIV: return SUB_IDIV; case OP_MOD: return SUB_MOD;
            case OP_BITX: return SUB_BITX;
            default: return 0;  // Div: the class says it all
        }
    };
    // first pass: slots bundle by bundle; r = {a_off, b_off, slot id (patched below) , a_lds | b_lds << 16}, ctrl kept aside
    std::vector<uint8_t> ctrl_of((size_t)NB * G, 0);
    for (uint32_t b = 0; b < NB; ++b) {
        const uint32_t k0 = bundle_start[b], k1 = bundle_start[b + 1], cnt = k1 - k0;
        const bool idle = cnt == 0;  // (programs of several streams: padding around the posts and waits; an Add of zeros into the trash slot)
        const bool request = !idle && (order[k0] & REQ_FLAG) != 0, collect = !idle && is_collect(order[k0]);
        const bool coop = bundle_coop[b] == 1 || bundle_coop[b] == 2, fusedb = bundle_coop[b] == 2;
        const uint32_t rep = coop ? COOP_LANES : 1u;  // a C_MULQ / C_MULF node's records take COOP_LANES positions (4j .. 4j+3)
        const int cl = idle ? (bundle_flags[b] ? (int)C_SYNC : (int)C_LIN) : request ? (int)C_DIVREQ : collect ? (int)C_DIVGET : fusedb ? (int)C_MULF : coop ? (int)C_MULQ : class_of(g.nodes[order[k0]]);
        uint32_t stream = 0;
        while (stream + 1 < P && b >= s_first[stream + 1]) ++stream;
        if (b == s_first[stream]) free_slots.clear();  // a slot is reused inside the stream that freed it only (the others run at their own pace)
        if (b < s_first[stream] + s_count[stream]) {  // (not the never-executed padding in front of the next stream)
            st.class_bundles[cl]++;
            st.class_nodes[cl] += cnt;
            stream_class_bundles[stream][cl]++;
        }
        dying.clear();
        const uint32_t stage = LDS_STAGE_OFF + (b % OPND_AHEAD) * STAGE_BYTES;
        if (b == s_first[stream]) s_cref[stream] = cref_row;
        const bool has_crefs = cl == C_INPUT || cl == C_TERN;
        if (has_crefs) out.crefs.resize((size_t)(cref_row + 1) * G, 0);
        // integer-class bundles: which operands arrive as canonical integers, and whether the result stays one
        uint32_t form_bits = 0;
        if (!idle && cl == C_INPUT && (node_vflags[order[k0] & ~REQ_FLAG] & VF_OUT_CANON)) form_bits |= HDR_OUT_CANON;  // (bit graphs: every Input node)
        if (!idle && (is_integer_class(cl) || cl == C_CMPZ)) {
            const uint8_t f0 = node_vflags[order[k0] & ~REQ_FLAG];
            for (uint32_t k = k0; k < k1; ++k) {
                const uint8_t f = node_vflags[order[k] & ~REQ_FLAG];
                if ((cl == C_CMPZ ? (f ^ f0) & VF_OUT_CANON : (f ^ f0)) != 0) {
                    err = "internal error: operand forms differ inside a bundle";
                    return false;
                }
            }
            if (cl != C_CMPZ) form_bits |= (f0 & VF_A_CANON ? HDR_A_CANON : 0u) | (f0 & VF_B_CANON ? HDR_B_CANON : 0u);
            form_bits |= f0 & VF_OUT_CANON ? HDR_OUT_CANON : 0u;
        }
        double form_saved = 0;  // (priced below, once the bundle is known to be all bit extracts or not)
        uint32_t scan_run = 0, scan_longest = 0, scan_bits = 0;  // scan bundles: the current / the longest chain segment, kind and shift
        for (uint32_t k = k0; k < k1; ++k) {
            const uint32_t i = order[k] & ~REQ_FLAG;
            const Node& n = g.nodes[i];
            const uint32_t js = order_pos[k];  // node slot (record position)
            uint32_t slot = 0xffffffffu;
            if (needs_slot[i] && !request) {
                if (witness_slots && witness_rank[i] != 0xffffffffu) {
                    slot = witness_rank[i];
                } else if (!free_slots.empty()) {
                    slot = free_slots.back();
                    free_slots.pop_back();
                } else {
                    slot = n_slots++;
                }
            }
            if (!request) ref[i] = slot;  // 0xffffffff: no slot (every use comes from the ring)
            uint32_t r[4] = {0, 0, slot, 0};
            // default: both operands unused -> staging loads of the zero constant, LDS reads of the own stage cells
            uint32_t off[2] = {zero_off, zero_off};
            uint32_t lds[2] = {stage + js * rep * T * 16u, stage + 2u * LDS_HALF_BYTES + js * rep * T * 16u};  // (C_MULQ: value t + T * js is loaded by lane 4 * T * js + t)
            // operand `producer` (operand number q of the node): its ring cell, or its slot for the staging load into the own stage cell
            auto enc_to = [&](uint32_t producer, int q, uint32_t& off_out, uint32_t& lds_out) {
                if (route(producer, i, q) == SRC_RING) {
                    lds_out = LDS_RING_OFF + (bundle_of[producer] % RING_BUNDLES) * RING_SLOT_BYTES + pos_in_bundle[producer] * T * 16u;
                } else {
                    const uint64_t o = g.nodes[producer].kind == N_CONST && reads_canonical_constants(i, q) ? (uint64_t)canon_const[producer] * slot_bytes : mem_off(producer);
                    off_out = (uint32_t)o;
                }
            };
            auto enc_operand = [&](uint32_t producer, int q) { enc_to(producer, q, off[q], lds[q]); };
            uint32_t ctrl = CTRL_ACTIVE;
            if (fusedb) {
                // main record (positions 4j, 4j+2): the product's factors, destination, op2; extra record (4j+1, 4j+3): the
                // operands of the second and third stage, op3.  A plain multiplication rides with op2 = op3 = none.
                const bool is_f = n.kind == N_FUSED, sq = is_f && fused_sq(n.op);
                const uint32_t op2 = is_f ? fused_op2(n.op) : FOP_NONE, op3 = is_f ? fused_op3(n.op) : FOP_NONE;
                const uint32_t px = js * rep + 1u;  // the extra record's position: its own stage cells
                uint32_t xoff[2] = {zero_off, zero_off};
                uint32_t xlds[2] = {stage + px * T * 16u, stage + 2u * LDS_HALF_BYTES + px * T * 16u};
                enc_to(n.a, 0, off[0], lds[0]);
                if (sq) {
                    enc_to(n.a, 0, off[1], lds[1]);
                    if (op2) enc_to(n.b, 1, xoff[0], xlds[0]);
                    if (op3) enc_to(n.c, 2, xoff[1], xlds[1]);
                } else {
                    enc_to(n.b, 1, off[1], lds[1]);
                    if (op2) enc_to(n.c, 2, xoff[0], xlds[0]);
                }
                const uint32_t rm[4] = {off[0], off[1], slot, lds[0] | (lds[1] << 16)};
                const uint32_t rx[4] = {xoff[0], xoff[1], 0xffffffffu, xlds[0] | (xlds[1] << 16)};
                for (uint32_t x = 0; x < rep; ++x) {
                    memcpy(&out.recs[((size_t)b * G + js * rep + x) * 4], (x & 1u) ? rx : rm, sizeof rm);
                    ctrl_of[(size_t)b * G + js * rep + x] = (uint8_t)(CTRL_ACTIVE | ((x & 1u) ? op3 : op2));
                }
                const uint32_t fops[3] = {n.a, n.b, n.c};
                for (int q = 0; q < arity_of(n); ++q)
                    if (needs_slot[fops[q]] == 1 && last_mem_use[fops[q]] == b) dying.push_back(fops[q]);
                continue;
            }
            if (!collect)
            switch (n.kind) {
                case N_INPUT:
                    if (n.a >= n_in_buf) {
                        err = "Input index out of range";
                        return false;
                    }
                    out.crefs[(size_t)cref_row * G + js] = n.a;  // input index
                    break;
                case N_UNO:  // Neg(a) = 0 - a  (graph.rs:188-194: 0 -> 0, else r - a); a travels in the b position
                    ctrl |= SUB_SUB;
                    enc_operand(n.a, 1);
                    break;
                case N_DUO:
                    ctrl |= sub_of(n.op);
                    enc_operand(n.a, 0);
                    if (n.op == OP_BITX) {  // the shift amount travels in the b_lds field, no second operand is read
                        lds[1] = g.const_values[g.nodes[n.b].a].v[0] * 16u;
                        break;
                    }
                    enc_operand(n.b, 1);
                    break;
                case N_SCAN: {
                    // position 2p: the step's OUT record {x, accumulator at a chain's head}; 2p + 1: its ACC record {divisor, 2^k in Montgomery form} (DIV)
                    const bool is_acc = (n.op & SCAN_OP_ACC) != 0, is_div = (n.op & SCAN_OP_DIV) != 0;
                    if ((js & 1u) != (is_acc ? 1u : 0u)) {
                        err = "internal error: scan records out of place";
                        return false;
                    }
                    const uint32_t pair = js / 2;
                    const bool is_sel = scan_is_sel(n.op);
                    const bool start = pair == 0 || is_sel || (n.op & SCAN_OP_NOACC) || (order[k0 + 2 * pair - 1] & ~REQ_FLAG) != n.b;
                    ctrl |= (is_acc ? SCAN_ROLE_ACC : 0u) | (start ? SCAN_START : 0u);
                    if (!is_acc) {
                        if (!(n.op & SCAN_OP_NOX)) enc_operand(n.a, 0);  // (a chain end without this operand reads 0: the record's default)
                        if (start && !(n.op & SCAN_OP_NOACC)) enc_operand(n.b, 1);
                        scan_run = start ? 1u : scan_run + 1u;
                        scan_longest = std::max(scan_longest, scan_run);
                        scan_bits = (is_div ? HDR_SCAN_DIV : 0u) | ((n.op & SCAN_OP_BORROW) ? HDR_SCAN_BORROW : 0u) | ((n.op & SCAN_OP_LEX) ? HDR_SCAN_LEX : 0u) |
                                    ((n.op & SCAN_OP_KG) ? HDR_SCAN_KG : 0u) | ((n.op & SCAN_OP_KL) ? HDR_SCAN_KL : 0u) | (scan_shift_of_node(i) << HDR_SCAN_SHIFT_SHIFT);
                    } else if (is_sel) {  // the ACC record: the selection's arms p, q
                        enc_operand(n.a, 0);
                        enc_operand(n.b, 1);
                    } else if (is_div) {
                        enc_to(n.c, 2, off[0], lds[0]);
                        off[1] = (uint32_t)mem_off(scan_imm[i]);
                    } else if (scan_has_third(n.op)) {  // BORROW / LEX: y
                        enc_to(n.c, 2, off[0], lds[0]);
                    }
                    break;
                }
                case N_CONV:
                    if (js != (scan_imm[i] & 0xffu)) {
                        err = "internal error: convolution columns out of place";
                        return false;
                    }
                    enc_operand(n.a, 0);
                    enc_operand(n.b, 1);
                    scan_bits = HDR_SCAN_CONV;
                    scan_longest = scan_imm[i] >> 8;  // (k rounds)
                    break;
                case N_TRES:
                    enc_operand(n.a, 0);
                    enc_operand(n.b, 1);
                    out.crefs[(size_t)cref_row * G + js] = g.nodes[n.c].kind == N_CONST && reads_canonical_constants(i, 2) ? (uint32_t)((uint64_t)canon_const[n.c] * slot_bytes)
                                                                                                                      : (uint32_t)mem_off(n.c);  // third operand always through memory
                    break;
            }
            r[0] = off[0];
            r[1] = off[1];
            r[3] = lds[0] | (lds[1] << 16);
            for (uint32_t x = 0; x < rep; ++x) {
                memcpy(&out.recs[((size_t)b * G + js * rep + x) * 4], r, sizeof r);
                ctrl_of[(size_t)b * G + js * rep + x] = (uint8_t)ctrl;
            }
            const uint32_t ops[3] = {n.a, n.b, n.c};
            for (int q = 0; q < (collect ? 0 : arity_of(n)); ++q) {
                uint32_t o = ops[q];
                if (needs_slot[o] == 1 && last_mem_use[o] == b) dying.push_back(o);
            }
        }
        uint32_t lin_bits = 0;
        if (cl == C_MULF) {  // which stages any node of the bundle has (the kernel runs those for every group)
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t op2 = ctrl_of[(size_t)b * G + (k - k0) * rep] & CTRL_SUB_MASK, op3 = ctrl_of[(size_t)b * G + (k - k0) * rep + 1] & CTRL_SUB_MASK;
                lin_bits |= (op2 == FOP_MUL ? HDR_F_S2MUL : op2 ? HDR_F_S2LIN : 0u) | (op3 ? HDR_F_S3LIN : 0u);
            }
            form_saved = (lin_bits & HDR_F_S2MUL ? 0.0 : kCyclesFusedStageMul) + ((lin_bits & (HDR_F_S2LIN | HDR_F_S3LIN)) ? 0.0 : kCyclesFusedStageLin);
        }
        if (cl == C_SCAN) {
            lin_bits = scan_bits | ((scan_longest - 1u) << HDR_SCAN_ITER_SHIFT);
            const bool limbs64 = ((scan_bits >> HDR_SCAN_SHIFT_SHIFT) & 0xffu) == 64u && scan_longest > 2;  // (priced as the parallel forms: what limb-sized operands take)
            uint32_t log_rounds = 0;
            while ((1u << log_rounds) < scan_longest) ++log_rounds;
            const double scan_cycles = (scan_bits & HDR_SCAN_CONV) ? kCyclesConvFront + (double)scan_longest * kCyclesConvStep
                                       : (scan_bits & (HDR_SCAN_BORROW | HDR_SCAN_LEX)) ? kCyclesScanFront + kCyclesScanBits
                                       : (scan_bits & HDR_SCAN_DIV) ? kCyclesScanFrontDiv + (limbs64 ? kCyclesScanParDivFlat + log_rounds * kCyclesScanParDivRound : (double)scan_longest * kCyclesScanStepDiv)
                                                                    : kCyclesScanFront + (limbs64 ? kCyclesScanParCarry : (double)scan_longest * kCyclesScanStepCarry);
            form_saved = kCycles[C_SCAN] - scan_cycles;
        }
        if (cl == C_LIN || cl == C_MUL || cl == C_MULQ)
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t sub = ctrl_of[(size_t)b * G + (k - k0) * rep] & CTRL_SUB_MASK;
                lin_bits |= sub == SUB_SUB ? HDR_LIN_SUB : sub == SUB_ADD ? HDR_LIN_ADD : 0u;
            }
        if (cl == C_MUL && !idle && (node_vflags[order[k0] & ~REQ_FLAG] & VF_MUL_CC)) {  // canonical products (a heap of their own: all or none)
            for (uint32_t k = k0; k < k1; ++k)
                if (!(node_vflags[order[k] & ~REQ_FLAG] & VF_MUL_CC) || class_of(g.nodes[order[k] & ~REQ_FLAG]) != C_MUL) {
                    err = "internal error: canonical and Montgomery products in one bundle";
                    return false;
                }
            lin_bits |= HDR_MUL_CC;
            form_saved = kCycles[C_MUL] - kCyclesMulCC;
        }
        if (cl == C_BIT) {
            bool all = true;
            for (uint32_t k = k0; k < k1; ++k) all = all && (ctrl_of[(size_t)b * G + (k - k0)] & CTRL_SUB_MASK) == SUB_BITX;
            bool limb_ops = true, any_shr = false;  // Shr and Band nodes only: the straight path
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t sub = ctrl_of[(size_t)b * G + (k - k0)] & CTRL_SUB_MASK;
                limb_ops = limb_ops && (sub == SUB_SHR || sub == SUB_BAND);
                any_shr = any_shr || sub == SUB_SHR;
            }
            lin_bits |= !limb_ops ? 0u : any_shr ? HDR_BIT_ALL_SHR : HDR_BIT_ALL_BAND;
            if (all) {
                lin_bits |= HDR_BITX_ALL;
                st.n_bitx_bundles++;
                stream_bitx[stream]++;
                form_saved = form_bits & HDR_A_CANON ? kCyclesBitxOperandForm : 0.0;
            }
        }
        if (cl == C_BIT && (lin_bits & (HDR_BIT_ALL_SHR | HDR_BIT_ALL_BAND)) && !(lin_bits & HDR_BITX_ALL))  // (straight path: measured with every form canonical)
            form_saved = (form_bits & HDR_A_CANON ? kCyclesOperandForm : 0.0) + (form_bits & HDR_B_CANON ? kCyclesOperandForm : 0.0) +
                         (form_bits & HDR_OUT_CANON ? kCyclesResultForm : 0.0) + kCyclesBitStraight;
        else if (is_integer_class(cl) && !(lin_bits & HDR_BITX_ALL))
            form_saved = (form_bits & HDR_A_CANON ? kCyclesOperandForm : 0.0) + (form_bits & HDR_B_CANON ? kCyclesOperandForm : 0.0) +
                         ((form_bits & HDR_OUT_CANON) && cl != C_CMPS ? kCyclesResultForm : 0.0);
        if (b < s_first[stream] + s_count[stream]) {
            st.form_cycles_saved += (uint64_t)form_saved;
            stream_form_saved[stream] += form_saved;
        }
        if (cl == C_MULQ && lin_bits) {
            st.n_coop_rider_bundles++;
            stream_riders[stream]++;
        }
        out.hdr[b] = (uint32_t)cl | (cnt << HDR_COUNT_SHIFT) | lin_bits | form_bits | bundle_flags[b];
        if (has_crefs) {  // (inactive node slots repeat the first word: a valid input index / slot offset)
            for (uint32_t q = cnt; q < G; ++q) out.crefs[(size_t)cref_row * G + q] = out.crefs[(size_t)cref_row * G];
            ++cref_row;
        }
        std::sort(dying.begin(), dying.end());
        dying.erase(std::unique(dying.begin(), dying.end()), dying.end());
        for (uint32_t o : dying) free_slots.push_back(ref[o]);
    }
    n_slots = std::max(n_slots, 1u);
    if (ws_tile_bytes(NC, n_slots, T) >= (uint64_t)OFF_NOWHERE) {
        err = "graph too large: one tile of the value workspace exceeds the 4 GiB buffer range";
        return false;
    }
    // second pass: destination byte offsets (trash slot = n_slots) + ctrl, and inactive padding records
    const uint32_t trash_off = nowhere ? OFF_NOWHERE : (uint32_t)(((uint64_t)NC + n_slots) * slot_bytes);
    out.trash_off = trash_off;
    for (uint32_t b = 0; b < NB; ++b) {
        const uint32_t rep = bundle_coop[b] == 1 || bundle_coop[b] == 2 ? COOP_LANES : 1u;
        const uint32_t cnt = (bundle_start[b + 1] - bundle_start[b]) * rep;  // record positions in use
        const uint32_t stage = LDS_STAGE_OFF + (b % OPND_AHEAD) * STAGE_BYTES;
        for (uint32_t q = 0; q < cnt; ++q) {
            uint32_t* r = &out.recs[((size_t)b * G + q) * 4];
            const uint32_t d = r[2] == 0xffffffffu ? trash_off : (uint32_t)(((uint64_t)NC + r[2]) * slot_bytes);
            r[2] = d | ctrl_of[(size_t)b * G + q];
        }
        for (uint32_t q = cnt; q < G; ++q) {  // inactive node slots: harmless operands, store -> trash, not ACTIVE
            uint32_t* r = &out.recs[((size_t)b * G + q) * 4];
            r[0] = r[1] = zero_off;
            r[2] = trash_off | (bundle_coop[b] == 2 ? 0u : ctrl_of[(size_t)b * G] & CTRL_SUB_MASK);  // (fused bundles: idle groups have no second / third stage)
            const uint32_t cell = (q / rep) * rep * T * 16u;  // (C_MULQ: the four positions of an idle group read one zero cell)
            r[3] = (stage + cell) | ((stage + 2u * LDS_HALF_BYTES + cell) << 16);
        }
    }
    out.n_slots = n_slots;
    out.n_streams = P;
    out.n_cref_rows = cref_row;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        out.stream_first[s] = s_first[s];
        out.stream_count[s] = s_count[s];
        out.stream_div_requests[s] = s_div[s];
        out.stream_cref_first[s] = s < P && s_count[s] ? s_cref[s] : cref_row;
        double c = 0, heavy = 0;
        for (int k = 0; k < (int)C_COUNT; ++k) c += kCycles[k] * (double)stream_class_bundles[s][k];
        c += kCyclesCoopRiders * (double)stream_riders[s] - (kCycles[C_BIT] - kCyclesBitx) * (double)stream_bitx[s] - stream_form_saved[s];
        for (int k : {(int)C_MUL, (int)C_MULQ, (int)C_MULF, (int)C_DIV}) heavy += kCycles[k] * (double)stream_class_bundles[s][k];
        out.stream_cycles[s] = c;
        out.stream_chain_cycles[s] = s_chain[s];
        out.stream_cycles_mul_div[s] = heavy;
    }
    out.n_inputs = (uint32_t)n_in_buf;
    out.n_witness = (uint32_t)g.witness_signals.size();
    out.witness_refs.resize(out.n_witness);
    for (size_t i = 0; i < g.witness_signals.size(); ++i) {
        const uint32_t w = g.witness_signals[i];
        out.witness_refs[i] = ref[w] | (g.nodes[w].kind != N_CONST && node_rep[w] == REP_C ? REF_CANON : 0u);
    }
    phase("slots + encoding");
    return true;
}

}  // namespace cwc
