// The runtime's pipeline: device check, the program choice for a batch (cost model over compiled candidates, quick first
// program + background search), upload, workspaces and the launches of a batch.  Everything numeric runs in the HIP
// kernels of kernels.hip; there is deliberately no CPU evaluation path.
#include "runtime_internal.hpp"

namespace cwcrt {


// prepare_status, reference src/lib.rs:28-38
void set_status(gw_status_t* st, GW_ERROR_CODE code, const std::string& msg) {
    if (!st) return;
    st->code = code;
    if (code == OK && msg.empty()) {
        st->error_msg = nullptr;
        return;
    }
    st->error_msg = (char*)malloc(msg.size() + 1);
    if (st->error_msg) memcpy(st->error_msg, msg.c_str(), msg.size() + 1);
}
int fail(gw_status_t* st, const std::string& msg) {
    set_status(st, ERROR, msg);
    return 1;
}
std::string upload_program(DeviceProgram& dp) {
    const Program& p = dp.host;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // (the interpreter reads the header two bundles ahead without a clamp: 16 bytes of zero padding behind the array)
    const size_t o_hdr = 0, o_recs = o_hdr + al(p.hdr.size() * 4 + 16), o_crefs = o_recs + al(p.recs.size() * 4 + REC_AHEAD * 1024u),  // (records are staged REC_AHEAD bundles ahead, unclamped)
                 o_consts = o_crefs + al(p.crefs.size() * 4 + 256), o_wit = o_consts + al(p.consts.size() * 4 + 32),
                 o_div = o_wit + al(p.witness_refs.size() * 4 + 4), total = o_div + al(p.div_lanes.size() * 4 + 4);
    HIP_TRY(hipMalloc(&dp.d_blob, total));
    char* d = (char*)dp.d_blob;
    HIP_TRY(hipMemset(d, 0, total));
    if (!p.hdr.empty()) HIP_TRY(hipMemcpy(d + o_hdr, p.hdr.data(), p.hdr.size() * 4, hipMemcpyHostToDevice));
    if (!p.recs.empty()) HIP_TRY(hipMemcpy(d + o_recs, p.recs.data(), p.recs.size() * 4, hipMemcpyHostToDevice));
    if (!p.crefs.empty()) HIP_TRY(hipMemcpy(d + o_crefs, p.crefs.data(), p.crefs.size() * 4, hipMemcpyHostToDevice));
    if (!p.consts.empty()) HIP_TRY(hipMemcpy(d + o_consts, p.consts.data(), p.consts.size() * 4, hipMemcpyHostToDevice));
    if (!p.witness_refs.empty()) HIP_TRY(hipMemcpy(d + o_wit, p.witness_refs.data(), p.witness_refs.size() * 4, hipMemcpyHostToDevice));
    dp.dev.hdr = (const uint32_t*)(d + o_hdr);
    dp.dev.recs = (const uint32_t*)(d + o_recs);
    dp.dev.crefs = (const uint32_t*)(d + o_crefs);
    dp.dev.consts = (const uint32_t*)(d + o_consts);
    if (!p.div_lanes.empty()) HIP_TRY(hipMemcpy(d + o_div, p.div_lanes.data(), p.div_lanes.size() * 4, hipMemcpyHostToDevice));
    dp.dev.witness_refs = (const uint32_t*)(d + o_wit);
    dp.dev.div_lanes = (const uint32_t*)(d + o_div);
    dp.dev.n_bundles = p.n_bundles;
    dp.dev.n_slots = p.n_slots;
    dp.dev.n_inputs = p.n_inputs;
    dp.dev.n_witness = p.n_witness;
    dp.dev.n_const = p.n_const;
    dp.dev.trash_off = p.trash_off;
    dp.dev.has_fused = 0;  // (from the bundle headers themselves: an imported program's statistics are not what the kernel runs)
    for (uint32_t h : p.hdr) dp.dev.has_fused |= (h & HDR_CLASS_MASK) == C_MULF ? 1u : ((h & HDR_CLASS_MASK) == C_SCAN || ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC))) ? 2u : 0u;
    // Round 5: the kinds for registers wider than a machine word (borrow chains, comparisons, carry chains of widths other than 64 bits with
    // their parallel form, 128-bit canonical products) live in interpreter instances of their own (MODE 3): a program with a borrow /
    // comparison bundle, or a carry-chain bundle of another width than 64 bits, runs there; every other limb program in the MODE 2
    // instances, whose code these kinds would only push apart (kernels.hip).
    if (dp.dev.has_fused == 2u && getenv("CWC_FORCE_MODE3")) dp.dev.has_fused = 3u;  // (layout experiments: any limb program in the MODE 3 instances)
    if (dp.dev.has_fused == 2u)
        for (uint32_t h : p.hdr)
            if ((h & HDR_CLASS_MASK) == C_SCAN && !(h & HDR_SCAN_CONV) &&
                ((h & (HDR_SCAN_BORROW | HDR_SCAN_LEX)) || (!(h & HDR_SCAN_DIV) && ((h >> HDR_SCAN_SHIFT_SHIFT) & 0xffu) != 64u))) {
                dp.dev.has_fused = 3u;
                break;
            }
    dp.dev.n_streams = p.n_streams;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        dp.dev.stream_first[s] = p.stream_first[s];
        dp.dev.stream_count[s] = p.stream_count[s];
        dp.dev.stream_div_requests[s] = p.stream_div_requests[s];
        dp.dev.stream_cref_first[s] = p.stream_cref_first[s];
    }
    return "";
}

// Interpreter waves per workgroup for programs without a divider wave: workgroups of four deal the waves evenly round
// the four SIMDs of a CU (kernels.hip); below one wave per SIMD of the chip single-wave workgroups spread further.
// CWC_WAVES_PER_WORKGROUP (1 or 4) overrides.
uint32_t waves_per_workgroup(uint32_t divider, uint64_t tiles, uint32_t streams) {
    // programs of several streams: the streams of a tile (and their divider waves) are one workgroup
    if (streams > 1) return divider ? streams : 4u;
    const char* e = getenv("CWC_WAVES_PER_WORKGROUP");
    if (divider == 1) return e ? (atoi(e) >= 4 ? 2u : 1u) : (tiles > 256 ? 2u : 1u);  // units of (interpreter + divider)
    if (divider) return 1;
    if (e) return atoi(e) == 4 ? 4u : 1u;
    return tiles > 512 ? 4u : 1u;
}

uint64_t workspace_budget() {
    const char* e = getenv("CWC_WORKSPACE_GB");
    double gb = e ? atof(e) : 8.0;
    if (gb < 1e-4) gb = 1e-4;  // tiny budgets are allowed (tests use them to force chunking); one tile is the floor
    return (uint64_t)(gb * (double)(1ull << 30));
}
}  // namespace cwcrt

// Program choice (measured on MI355X, profiles/r01_sweep_batch_tile.txt).  A wave's time is the sum of its bundles;
// wider tiles use the lanes better but need more bundles, and the chip holds 2048 of these waves (LDS: 8 per CU).
// Best measured: T = 1 up to 256 sets, 2 up to 1024, 4 up to 8192, then the narrowest tile whose waves are all
// resident at once (16384 sets -> 8, 32768 -> 16, ...).  The asynchronous divider wave (one extra wavefront per tile
// that serves the divisions while the interpreter goes on) pays while the extra waves find free SIMDs: up to 1024 tiles.
// CWC_TARGET_WAVES (default 2048) and CWC_DIVIDER_TILES (default 1024) move the rules; gwb_set_tile_width /
// CWC_TILE_WIDTH override them.
extern "C" uint32_t gwb_pick_tile_width(size_t batch) {
    size_t target = 2048, divider_tiles = 1024;
    if (const char* e = getenv("CWC_TARGET_WAVES")) {
        long v = atol(e);
        if (v > 0) target = (size_t)v;
    }
    if (const char* e = getenv("CWC_DIVIDER_TILES")) divider_tiles = (size_t)atol(e);
    uint32_t t = batch <= 256 ? 1 : batch <= 1024 ? 2 : 4;
    while (t < 64 && (batch + t - 1) / t > target) t *= 2;
    const size_t tiles = (batch + t - 1) / t;
    return t | (tiles <= divider_tiles && t < 64 ? KEY_DIVIDER : 0u);
}

namespace cwcrt {

std::string check_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
               "); this library has no CPU fallback";
    // The interpreter parks a pending scalar load in XNACK_MASK, which the hardware owns when XNACK (retry on page fault) is
    // enabled: such a device is refused instead of risking a corrupted replay.  (Asked once per process.)
    static const std::string xnack_refusal = []() -> std::string {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            (void)hipGetLastError();
            return "";
        }
        return strstr(prop.gcnArchName, "xnack+") ? std::string("device ") + prop.gcnArchName + ": XNACK-enabled devices are not supported (run with HSA_XNACK=0)" : "";
    }();
    return xnack_refusal;
}

// Cost model behind the automatic program choice.  A wave's time is the sum of its bundles (lone-wave shader cycles
// per bundle class, profiles/r01_class_profile.txt).  Launched as four-wave workgroups the waves sit one per SIMD up to
// 1024 of them and cost what a lone wave costs; from there to 2048 some SIMDs hold two and the kernel takes what those
// take: measured x1.3 for the multiplier / inversion bundles (issue-bound, two waves overlap well) and x1.9 for the
// rest (LDS / vector-memory bound), i.e. x1.28-1.36 for the authV2-class graph and x1.75-2.0 for sha256; beyond 2048
// waves they run in rounds.  A divider wave per interpreter counts as a wave of its CU but idles about half of the time
// (x1.33 measured at 1024 pairs); one divider per four interpreters means five-wave workgroups, one per CU (LDS): x1.30
// at 1024 tiles (measured / modelled 1.19-1.26 in rounds 2 and 3 against 1.11-1.25 for the pair programs: at x1.23 the model
// took T = 2 + group divider for 2048 sets, 6 % behind T = 4 + pairs in both rounds' sweeps), rounds of 1024 tiles beyond.
// (profiles/r01_sweep_batch_tile.txt, r03_sweep_batch_tile.txt)
double estimate_cycles(const Program& p, size_t batch) {
    // (tiles of 8 sets and more: their bundles measure ~10 % above the per-class table, which was taken at T = 2 --
    // round 2, authV2-class: 8192 sets T = 4 41.2 ms, T = 8 44.2 ms, T = 8 + group divider 45.0 ms; 16384 sets T = 8 69.2 ms)
    const double wide = p.T >= 8 ? 1.10 : 1.0;
    if (p.n_streams > 1) {
        // Programs of several streams (round 2, authV2-class, profiles/r02_streams_ab.txt): a tile is done when its slowest
        // stream is -- the longer of its bundles' cycles and its longest dependent chain with the divisions at the
        // divider wave's latency; measured / modelled 0.98 at T = 1, 1.08 at T = 2, 1.2 at T = 4.  The streams of a
        // tile and their divider waves are one workgroup: with dividers 98 KiB of LDS for four streams (one workgroup
        // per CU, 256 tiles at a time), 49 KiB for two; without, four waves of 20 KiB.  More live waves than SIMDs
        // (2 x 512 tiles + dividers measured x1.3) slow each other down.
        const double tiles = (double)((batch + p.T - 1) / p.T);
        const double t = program_wave_cycles(p);
        double busy = 0;  // SIMDs' worth of work per tile: every stream and divider wave for the share of t it is busy
        for (uint32_t s = 0; s < p.n_streams; ++s) busy += (p.stream_cycles[s] + model_class_cycles(C_DIV) * p.stream_div_requests[s]) / t;
        const double wgs = p.divider ? tiles : std::ceil(tiles * p.n_streams / 4.0);
        const double wg_per_cu = p.divider ? (p.n_streams == 4 ? 1.0 : 3.0) : 2.0;
        const double rounds = std::max(1.0, wgs / (256.0 * wg_per_cu));
        const double crowd = std::max(1.0, 1.1 * (tiles / rounds) * busy / 1024.0);
        return t * (p.T >= 4 ? 1.2 * wide : p.T == 2 ? 1.08 : 1.0) * crowd * rounds;
    }
    const double per_wave = program_wave_cycles(p) * wide;
    const double heavy = program_wave_cycles_mul_div(p) * wide;
    const double waves = (double)((batch + p.T - 1) / p.T);
    const double two_per_simd = (1.3 * heavy + 1.9 * (per_wave - heavy)) / per_wave;
    if (p.divider == 4) return per_wave * 1.30 * (waves <= 1024 ? 1.0 : waves / 1024);
    // three interpreters + their divider = a four-wave workgroup, one per CU: every wave has its SIMD up to 768 tiles;
    // the shared divider costs 8 % against a divider per interpreter (measured at 512 tiles: 16.8 vs 15.6 ms)
    if (p.divider == 3) return per_wave * 1.08 * (waves <= 768 ? 1.0 : two_per_simd * (waves <= 1536 ? 1.0 : waves / 1536));
    const double resident = waves * (p.divider == 1 ? 2.0 : 1.0);
    double crowd = resident <= 1024 ? 1.0 : two_per_simd;
    if (p.divider == 1 && resident > 1024) crowd = 1.0 + 0.75 * (crowd - 1.0);
    const double rounds = resident <= 2048 ? 1.0 : resident / 2048;
    return per_wave * crowd * rounds;
}

// candidate program keys for a batch (the static rule's tile width and its neighbours, the divider / stream modes that fit)
std::vector<uint32_t> candidate_keys(const ProgramStats& stats, size_t batch, uint32_t rule, uint32_t min_t) {
    size_t divider_tiles = 1024;
    if (const char* e = getenv("CWC_DIVIDER_TILES")) divider_tiles = (size_t)atol(e);
    const bool has_div = stats.class_nodes[C_DIV] > 0;
    const uint32_t t0 = rule & ~KEY_MODE_MASK;
    std::vector<uint32_t> keys;
    for (uint32_t t = std::max(min_t, t0 >= 4 ? t0 / 4 : 1u); t <= t0 * 2 && t <= 32 && (batch >= 64 || t == t0); t *= 2)  // (tiny batches: one tile either way)
        for (uint32_t mode : {0u, KEY_DIVIDER, KEY_TRIPLE, KEY_GROUP}) {
            const size_t tiles = (batch + t - 1) / t;
            if (tiles > 4 * 2048) continue;
            // divider waves: while every pair is resident; one divider per four interpreters: where a five-wave
            // workgroup per CU covers more than half of the batch at once
            const bool divider_fits = has_div && tiles <= divider_tiles;
            if (mode == 0 && divider_fits) continue;  // (measured: with every pair resident the divider program always wins)
            if (mode == KEY_DIVIDER && !divider_fits) continue;
            if (mode == KEY_TRIPLE && !(has_div && tiles > 512 && tiles <= 768 && !getenv("CWC_NO_GROUP_DIVIDER"))) continue;
            if (mode == KEY_GROUP && !(has_div && tiles > 512 && tiles <= 1024 && !getenv("CWC_NO_GROUP_DIVIDER"))) continue;
            keys.push_back(t | mode);
            // the graph's independent parts on wavefronts of their own (streams): while every stream of every tile has
            // a SIMD to itself (small batches, the single-shot entry point)
            if ((mode == 0 || mode == KEY_DIVIDER) && t < 64 && !getenv("CWC_NO_STREAMS")) {
                if (tiles <= 256) keys.push_back(t | mode | KEY_STREAMS4);
                else if (tiles <= 340) keys.push_back(t | mode | KEY_STREAMS2);
            }
        }
    return keys;
}

// the full choice for a batch size, on a thread of its own (reads the graph only): every candidate compiled with the
// search over schedule variants, priced by the cost model
gwb_graph::Refined refine_choice(const Graph& graph, const ProgramStats& stats, size_t batch, uint32_t rule, uint32_t min_t) {
    gwb_graph::Refined r;
    try {
        double best_cost = -1;
        std::unique_ptr<SharedRewrites, void (*)(SharedRewrites*)> rewrites(make_shared_rewrites(), free_shared_rewrites);  // (one rewritten graph per tile width)
        for (uint32_t key : candidate_keys(stats, batch, rule, min_t)) {
            std::unique_ptr<Program> p(new Program());
            std::string err;
            if (!compile_program(graph, key & ~KEY_MODE_MASK, key_divider_waves(key), *p, err, key_streams(key), false, rewrites.get())) continue;
            const double cost = estimate_cycles(*p, batch);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                r.best = key;
            }
            r.programs[key] = std::move(p);
        }
    } catch (...) {
        r.best = 0;
        r.programs.clear();
    }
    return r;
}

// The program key for a batch: forced / environment override, else the static rule's width and its neighbours
// compiled (host only) and priced with the cost model; the choice is remembered per batch size.
// allow_quick = false: the caller wants the searched program now (the program that is exported / broadcast to other ranks:
// imported handles are never refined, a provisional single-schedule program would stay with them for good).
uint32_t pick_tile_width(gwb_graph* g, size_t batch, bool allow_quick) {
    if (!g->has_graph && !g->progs.empty()) return g->progs.begin()->first;  // imported: the one program it has
    if (g->forced_T) return g->forced_T;
    if (const char* e = getenv("CWC_TILE_WIDTH")) {  // width, or width + 256 for the asynchronous divider
        const uint32_t key = (uint32_t)atoi(e), t = key & ~KEY_MODE_MASK;
        if (t >= 1 && t <= 64 && !(t & (t - 1))) return key;
    }
    uint32_t rule = gwb_pick_tile_width(batch);
    // Deep graphs: a program is one header word, G records and G third-operand words per bundle, and a bundle per
    // dependency level at least -- 1 KiB per bundle at T = 1 (1.6 GB for the 10.5 M-node bigint-class graph of
    // BASELINE config 5, depth 1.29 M).  Small batches fill the same number of SIMDs whatever the tile width (every tile
    // is one wave), so the width is raised until the program stream fits CWC_PROGRAM_MB (default 960): 0.85 GB at T = 2.
    uint32_t min_t = 1;
    {
        double budget = 960.0;
        if (const char* e = getenv("CWC_PROGRAM_MB")) budget = atof(e);
        const double per_bundle_t1 = 4.0 + 64.0 * 16.0;
        // (tile widths with scan bundles: limb recurrences take a tenth of their depth in bundles)
        auto levels = [&](uint32_t t) { return (double)(t <= SCAN_MAX_T && !getenv("CWC_NO_SCAN") && g->stats.depth_scan ? g->stats.depth_scan : g->stats.depth); };
        while (min_t < 16 && levels(min_t) * 1.25 * (4.0 + (per_bundle_t1 - 4.0) / min_t) > budget * 1048576.0) min_t *= 2;
        if ((rule & ~KEY_MODE_MASK) < min_t) rule = min_t | ((rule & KEY_MODE_MASK) && min_t < 64 ? (rule & KEY_MODE_MASK) : 0u);
    }
    if (getenv("CWC_STATIC_TILE_RULE") || !g->has_graph) return rule;
    auto hit = g->chosen.find(batch);
    if (hit != g->chosen.end()) return hit->second;
    // ---- small batches: quick program first, the full choice in the background (see gwb_graph::refining) ----
    // (graphs beyond two million nodes have one schedule anyway: nothing for the background to search)
    if (batch < 64 && g->graph.nodes.size() <= 2000000 && !getenv("CWC_NO_QUICK_FIRST_CALL")) {
        auto job = g->refining.find(batch);
        if (job != g->refining.end()) {
            if (!allow_quick) {  // wait for the search that is under way
                if (g->refine_gate) g->refine_gate->store(true);
                job->second.wait();
            }
            if (job->second.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return g->provisional[batch];
            gwb_graph::Refined r = job->second.get();
            g->refining.erase(job);
            const uint32_t quick_key = g->provisional[batch];
            g->provisional.erase(batch);
            if (r.best == 0) {  // (nothing compiled in the background: keep what runs)
                g->chosen[batch] = quick_key;
                return quick_key;
            }
            // the refined programs replace the quick one (also when it was uploaded: its device copy is freed, hipFree waits for the device)
            for (auto& kv : r.programs) {
                if (g->progs.count(kv.first)) {
                    if (g->filled_prog == g->progs[kv.first].get()) g->filled_prog = nullptr;
                    g->progs.erase(kv.first);
                }
                g->compiled[kv.first] = std::move(kv.second);
            }
            g->chosen[batch] = r.best;
            return r.best;
        }
        const bool has_div0 = g->stats.class_nodes[C_DIV] > 0;
        const uint32_t t1 = std::max(1u, min_t);
        const uint32_t quick_key = t1 | (has_div0 && t1 < 64 ? KEY_DIVIDER : 0u) | (t1 < 64 && !getenv("CWC_NO_STREAMS") ? KEY_STREAMS4 : 0u);
        if (allow_quick && !g->progs.count(quick_key) && !g->compiled.count(quick_key)) {
            std::unique_ptr<Program> p(new Program());
            std::string err;
            if (compile_program(g->graph, quick_key & ~KEY_MODE_MASK, key_divider_waves(quick_key), *p, err, key_streams(quick_key), true)) g->compiled[quick_key] = std::move(p);
        }
        if (allow_quick && (g->progs.count(quick_key) || g->compiled.count(quick_key))) {
            const Graph* graph = &g->graph;
            const ProgramStats stats = g->stats;
            g->provisional[batch] = quick_key;
            // (the single-shot entry point's on-disk cache: the task writes the program it settles on, no call waits for the file)
            const std::string cache_file = batch == 1 && !g->cache_written ? g->cache_path : std::string();
            const std::vector<InputSignal> inputs = cache_file.empty() ? std::vector<InputSignal>() : g->inputs;
            if (!cache_file.empty()) g->cache_written = true;
            if (!g->refine_gate) g->refine_gate = std::make_shared<std::atomic<bool>>(false);
            g->refine_gate->store(false);
            std::shared_ptr<std::atomic<bool>> gate = g->refine_gate;
            g->refining[batch] = std::async(std::launch::async, [graph, stats, batch, rule, min_t, cache_file, inputs, gate]() {
                for (int waited = 0; !gate->load() && waited < 2000; ++waited) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                gwb_graph::Refined r = refine_choice(*graph, stats, batch, rule, min_t);
                auto best = r.programs.find(r.best);
                if (!cache_file.empty() && r.best && best != r.programs.end()) {
                    try {
                        const std::vector<uint8_t> b0 = exported_bytes(*best->second, inputs), b = cache_wrap(cache_file, b0.data(), b0.size());
                        write_file_atomically(cache_file, b.data(), b.size());
                        if (getenv("CWC_DEBUG_CACHE")) fprintf(stderr, "program cache: wrote %s (program key %#x, %zu bytes)\n", cache_file.c_str(), r.best, b.size());
                    } catch (...) {
                    }
                }
                return r;
            });
            return quick_key;
        }
    }
    const bool debug = getenv("CWC_DEBUG_COST") != nullptr;
    uint32_t best = rule;
    double best_cost = -1;
    const std::vector<uint32_t> keys = candidate_keys(g->stats, batch, rule, min_t);
    // the candidates that are not compiled yet, each on a thread of its own (the compiler only reads the graph)
    // (candidates of one tile width share the rewritten graph: the first thread through rewrites, the others copy)
    std::unique_ptr<SharedRewrites, void (*)(SharedRewrites*)> rewrites(make_shared_rewrites(), free_shared_rewrites);
    std::vector<std::pair<uint32_t, std::future<std::unique_ptr<Program>>>> jobs;
    for (uint32_t key : keys)
        if (!g->progs.count(key) && !g->compiled.count(key)) {
            const Graph* graph = &g->graph;
            SharedRewrites* shared = rewrites.get();
            jobs.emplace_back(key, std::async(std::launch::async, [graph, key, shared]() {
                                  std::unique_ptr<Program> p(new Program());
                                  std::string err;
                                  if (!compile_program(*graph, key & ~KEY_MODE_MASK, key_divider_waves(key), *p, err, key_streams(key), false, shared)) p.reset();
                                  return p;
                              }));
        }
    for (auto& j : jobs) {
        std::unique_ptr<Program> p = j.second.get();
        if (p) g->compiled[j.first] = std::move(p);
    }
    for (uint32_t key : keys) {
        const Program* p = nullptr;
        auto up = g->progs.find(key);
        auto pre = g->compiled.find(key);
        if (up != g->progs.end()) p = &up->second->host;
        else if (pre != g->compiled.end()) p = pre->second.get();
        else continue;  // (did not compile: not a candidate)
        const double cost = estimate_cycles(*p, batch);
        if (debug) fprintf(stderr, "cost model: batch %zu key %#x -> %.1f Mcycles\n", batch, key, cost / 1e6);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = key;
        }
    }
    g->chosen[batch] = best;
    return best;
}

std::string get_program(gwb_graph* g, uint32_t key, DeviceProgram** out) {
    const uint32_t T = key & ~KEY_MODE_MASK;
    if (T == 64) key = T;  // no divider programs at T = 64
    auto it = g->progs.find(key);
    if (it != g->progs.end()) {
        *out = it->second.get();
        return "";
    }
    if (!g->has_graph) return "imported graph handle has no program for tile width " + std::to_string(T);
    std::unique_ptr<DeviceProgram> dp(new DeviceProgram());
    std::string err;
    auto pre = g->compiled.find(key);
    if (pre != g->compiled.end()) {  // already compiled for the cost model
        dp->host = std::move(*pre->second);
        g->compiled.erase(pre);
    } else if (!compile_program(g->graph, T, key_divider_waves(key), dp->host, err, key_streams(key))) {
        return err;
    }
    err = upload_program(*dp);
    if (!err.empty()) return err;
    *out = dp.get();
    g->progs[key] = std::move(dp);
    return "";
}

std::string run_device(gwb_graph* g, const void* d_inputs, size_t batch, void* d_witness, uint32_t* d_status,
                       hipStream_t stream, bool montgomery, hipEvent_t done_event) {
    if (batch == 0) return "";
    if (batch > 0x7fffffffull) return "batch too large";
    static const bool dbg_steps = getenv("CWC_DEBUG_SINGLE") != nullptr;  // diagnostic: program choice / upload of a call
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = dbg_steps ? now_ms() : 0.0;
    const uint32_t key = pick_tile_width(g, batch);
    struct OpenGate {  // every way out of this call (errors included) lets a waiting background search start
        gwb_graph* g;
        ~OpenGate() {
            if (g->refine_gate) g->refine_gate->store(true);
        }
    } open_gate{g};
    const double t_1 = dbg_steps ? now_ms() : 0.0;
    DeviceProgram* dp = nullptr;
    std::string err = get_program(g, key, &dp);
    if (!err.empty()) return err;
    const double t_2 = dbg_steps ? now_ms() : 0.0;
    g->last_key = (key & ~KEY_MODE_MASK) == 64 ? 64u : key;
    const Program& p = dp->host;
    const uint32_t T = p.T;
    if (!g->last_done) HIP_TRY(hipEventCreateWithFlags(&g->last_done, hipEventDisableTiming));
    if (g->has_last && g->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, g->last_done, 0));
    // From the first enqueue on, every way out (errors included) leaves the ordering state pointing at this stream: the
    // next call on another stream waits for whatever was enqueued here (the workspace and the constant fill are shared).
    struct OrderGuard {
        gwb_graph* g;
        hipStream_t stream;
        ~OrderGuard() {
            if (hipEventRecord(g->last_done, stream) == hipSuccess) {
                g->last_stream = stream;
                g->has_last = true;
            } else {
                (void)hipGetLastError();
                g->has_last = false;
                (void)hipDeviceSynchronize();  // (cannot order by event: nothing of this call is left in flight)
            }
        }
    } order_guard{g, stream};
    // Workspace: tiles of (constants | value slots | trash slot), grouped into separately allocated chunks of at most
    // CWC_WORKSPACE_GB; larger batches than WS_MAX_CHUNKS chunks hold are evaluated in several launches.
    const uint64_t bytes_per_tile = ws_tile_bytes(p.n_const, p.n_slots, T);
    const uint64_t budget = workspace_budget();
    if (bytes_per_tile > 0xffffffffull) return "graph too large for the 4 GiB tile window";
    uint64_t max_tiles = budget / bytes_per_tile;
    if (max_tiles == 0) max_tiles = 1;
    const uint64_t tiles_total = (batch + T - 1) / T;
    const uint64_t chunk_tiles = tiles_total < max_tiles ? tiles_total : max_tiles;
    const size_t need = (size_t)(chunk_tiles * bytes_per_tile);
    const size_t chunk_sets = (size_t)chunk_tiles * T;
    const size_t n_chunks = (batch + chunk_sets - 1) / chunk_sets;
    // chunks per launch: all of them when they fit the table (CWC_STREAMS caps the number); otherwise several
    // launches, one after the other
    size_t per_launch = n_chunks < WS_MAX_CHUNKS ? n_chunks : WS_MAX_CHUNKS;
    if (const char* e = getenv("CWC_STREAMS")) {
        const long v = atol(e);
        if (v >= 1 && (size_t)v < per_launch) per_launch = (size_t)v;
    }
    bool refill = g->filled_prog != dp || g->filled_tiles_per_chunk != chunk_tiles || g->filled_chunks < per_launch;
    for (size_t l = 0; l < per_launch; ++l) {
        if (need > g->vals_bytes[l]) {
            if (g->d_vals[l]) HIP_TRY(hipFree(g->d_vals[l]));  // (hipFree waits for the device: earlier calls are done with it)
            g->d_vals[l] = nullptr;
            g->vals_bytes[l] = 0;
            g->filled_prog = nullptr;
            refill = true;
            HIP_TRY(hipMalloc(&g->d_vals[l], need));
            g->vals_bytes[l] = need;
        }
    }
    if (refill) {  // every tile's copy of the constants (the interpreter never writes there)
        WsTable all;
        memset(&all, 0, sizeof all);
        all.tiles_per_chunk = (uint32_t)chunk_tiles;
        all.n_chunks = (uint32_t)per_launch;
        for (size_t l = 0; l < per_launch; ++l) all.base[l] = g->d_vals[l];
        HIP_TRY(launch_fill_consts(T, dp->dev, all, (uint32_t)(per_launch * chunk_tiles), stream));
        g->filled_prog = dp;
        g->filled_tiles_per_chunk = chunk_tiles;
        g->filled_chunks = per_launch;
    }
    const double t_3 = dbg_steps ? now_ms() : 0.0;
    g->last_call_launches = 0;
    g->timing = gwb_timing_t{};
    g->timing.tile_width = T;
    g->timing.divider = p.divider;
    g->timing.streams = p.n_streams;
    g->timing.n_bundles = p.n_bundles;
    g->timing.n_slots = p.n_slots;
    const size_t launch_sets = per_launch * chunk_sets;
    for (size_t s0 = 0; s0 < batch; s0 += launch_sets) {
        const uint32_t nb = (uint32_t)((batch - s0) < launch_sets ? (batch - s0) : launch_sets);
        WsTable wst;
        memset(&wst, 0, sizeof wst);
        wst.tiles_per_chunk = (uint32_t)chunk_tiles;
        wst.n_chunks = (uint32_t)((nb + chunk_sets - 1) / chunk_sets);
        for (uint32_t l = 0; l < wst.n_chunks; ++l) wst.base[l] = g->d_vals[l];
        hipEvent_t e0, e1, e2;
        if (g->pending.size() >= gwb_graph::kHistory) {  // recycle the oldest launch's events
            e0 = g->pending.front().start, e1 = g->pending.front().after_interp, e2 = g->pending.front().after_pack;
            g->pending.pop_front();
        } else {
            e0 = e1 = e2 = nullptr;
            const bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventCreate(&e2) == hipSuccess;
            if (!ok) {
                for (hipEvent_t e : {e0, e1, e2})
                    if (e) (void)hipEventDestroy(e);
                return "hipEventCreate failed";
            }
        }
        g->pending.push_back(gwb_graph::ChunkEvents{e0, e1, e2});  // (owned by the handle from here on, also on an early return)
        HIP_TRY(hipEventRecord(e0, stream));
        HIP_TRY(launch_interp(T, p.divider, waves_per_workgroup(p.divider, (nb + T - 1) / T, p.n_streams), p.n_div_requests, dp->dev.div_lanes, dp->dev, wst, (const char*)d_inputs + s0 * p.n_inputs * 32, d_status + s0, nb, stream, g->d_prof));
        HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(launch_pack(T, dp->dev, wst, (char*)d_witness + s0 * (size_t)p.n_witness * 32, nb, stream, montgomery));
        HIP_TRY(hipEventRecord(e2, stream));
        g->last_call_launches++;
        g->timing.n_launches++;
    }
    g->timing_pending = true;
    if (done_event) HIP_TRY(hipEventRecord(done_event, stream));
    if (g->refine_gate) g->refine_gate->store(true);  // the first call's work is on the device: the background search may take the host's cores
    if (dbg_steps && now_ms() - t_0 > 20.0)
        fprintf(stderr, "run_device: program choice %.1f ms, program on the device %.1f ms, workspace + constants %.1f ms, launches %.1f ms\n", t_1 - t_0, t_2 - t_1, t_3 - t_2, now_ms() - t_3);
    return "";
}

// Host-buffer entry: rows in, rows out.  The witness rows are the big transfer (authV2-class: 2.4 MB per set), so
// they come back in slices through two pinned staging buffers on a copy stream while worker threads move the previous
// slice into the caller's (pageable) memory; a caller buffer that is already pinned (gwb_host_alloc, hipHostMalloc,
// hipHostRegister) is the copy's destination directly.  Device buffers and staging are kept on the handle.
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

unsigned env_threads(const char* name, unsigned cap) {
    long v = 0;
    if (const char* e = getenv(name)) v = atol(e);
    if (v <= 0) {
        v = (long)std::thread::hardware_concurrency();
        if (cap && v > (long)cap) v = cap;
    }
    return v < 1 ? 1u : (unsigned)v;
}

unsigned copy_threads() {
    long v = 0;
    if (const char* e = getenv("CWC_COPY_THREADS")) v = atol(e);
    if (v <= 0) {
        v = (long)std::thread::hardware_concurrency();
        if (v > 16) v = 16;
    }
    return v < 1 ? 1u : (unsigned)v;
}

std::string device_to_host_rows(gwb_graph* g, void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return "";
    if (!g->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
    if (is_pinned_host(dst)) {
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, g->copy_stream));
        HIP_TRY(hipStreamSynchronize(g->copy_stream));
        return "";
    }
    size_t slice = 32u << 20;
    if (const char* e = getenv("CWC_COPY_SLICE_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 1024) slice = (size_t)v << 20;
    }
    // (pinned memory is slow to get -- two 32 MB buffers were 90 ms of the single-shot entry point's first call: a transfer
    // that fits one slice takes one buffer of its own size)
    if (bytes < slice) slice = std::max<size_t>(g->stage_bytes, (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1));
    const int n_stage = bytes > slice ? 2 : 1;
    if (g->stage_bytes < slice) {
        for (int i = 0; i < 2; ++i) {
            if (g->stage[i]) HIP_TRY(hipHostFree(g->stage[i]));
            g->stage[i] = nullptr;
        }
        g->stage_bytes = 0;
        for (int i = 0; i < n_stage; ++i) HIP_TRY(hipHostMalloc(&g->stage[i], slice, hipHostMallocDefault));
        g->stage_bytes = slice;
    }
    if (n_stage == 2 && !g->stage[1]) HIP_TRY(hipHostMalloc(&g->stage[1], g->stage_bytes, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i)
        if (!g->stage_done[i]) HIP_TRY(hipEventCreateWithFlags(&g->stage_done[i], hipEventDisableTiming));
    const size_t n_slices = (bytes + slice - 1) / slice;
    const unsigned n_workers = bytes < (8u << 20) ? 1u : copy_threads();
    // workers: slice k is theirs once `ready` > k; each takes one stripe of it and counts itself in consumed[k]
    std::atomic<long> ready{0};
    std::atomic<bool> abort{false};
    std::vector<std::atomic<unsigned>> consumed(n_slices);
    for (auto& c : consumed) c.store(0);
    auto stripe_copy = [&](unsigned w, size_t k) {
        const size_t off = k * slice, len = bytes - off < slice ? bytes - off : slice;
        const size_t per = ((len + n_workers - 1) / n_workers + 4095) & ~(size_t)4095;
        const size_t a = (size_t)w * per, b = a + per < len ? a + per : len;
        if (a < b) memcpy((char*)dst + off + a, (const char*)g->stage[k & 1] + a, b - a);
    };
    std::vector<std::thread> workers;
    for (unsigned w = 1; w < n_workers; ++w)
        workers.emplace_back([&, w]() {
            for (size_t k = 0; k < n_slices; ++k) {
                while (ready.load(std::memory_order_acquire) <= (long)k) {
                    if (abort.load()) return;
                    std::this_thread::yield();
                }
                stripe_copy(w, k);
                consumed[k].fetch_add(1, std::memory_order_release);
            }
        });
    std::string err;
    auto issue = [&](size_t k) -> std::string {
        const size_t off = k * slice, len = bytes - off < slice ? bytes - off : slice;
        HIP_TRY(hipMemcpyAsync(g->stage[k & 1], (const char*)d_src + off, len, hipMemcpyDeviceToHost, g->copy_stream));
        HIP_TRY(hipEventRecord(g->stage_done[k & 1], g->copy_stream));
        return "";
    };
    err = issue(0);
    for (size_t k = 0; k < n_slices && err.empty(); ++k) {
        if (k + 1 < n_slices) {
            // buffer (k+1)&1 held slice k-1: every worker must be done with it before the next copy lands there
            if (k >= 1)
                while (consumed[k - 1].load(std::memory_order_acquire) < n_workers) std::this_thread::yield();
            err = issue(k + 1);
            if (!err.empty()) break;
        }
        if (hipEventSynchronize(g->stage_done[k & 1]) != hipSuccess) {
            err = "hipEventSynchronize failed in the witness copy";
            break;
        }
        ready.store((long)k + 1, std::memory_order_release);
        stripe_copy(0, k);
        consumed[k].fetch_add(1, std::memory_order_release);
    }
    if (!err.empty()) abort.store(true);
    for (auto& t : workers) t.join();
    if (!err.empty()) (void)hipStreamSynchronize(g->copy_stream);
    return err;
}

std::string run_host(gwb_graph* g, const void* inputs, size_t batch, void* witness, uint32_t* set_status) {
    if (batch == 0) return "";
    const size_t in_b = batch * (size_t)g->n_inputs * 32, out_b = batch * (size_t)g->n_witness * 32;
    auto grow = [](void*& p, size_t& have, size_t need) -> std::string {
        if (need <= have) return "";
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        have = 0;
        HIP_TRY(hipMalloc(&p, need));
        have = need;
        return "";
    };
    static const bool dbg_steps = getenv("CWC_DEBUG_SINGLE") != nullptr;  // diagnostic: the steps of a host-rows call
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = dbg_steps ? now_ms() : 0.0;
    std::string err = grow(g->h_in, g->h_in_bytes, in_b ? in_b : 32);
    if (err.empty()) err = grow(g->h_out, g->h_out_bytes, out_b ? out_b : 32);
    if (err.empty()) err = grow(g->h_st, g->h_st_bytes, batch * 4);
    if (!err.empty()) return err;
    HIP_TRY(hipMemcpy(g->h_in, inputs, in_b, hipMemcpyHostToDevice));
    const double t1 = dbg_steps ? now_ms() : 0.0;
    err = run_device(g, g->h_in, batch, g->h_out, (uint32_t*)g->h_st, nullptr);
    if (!err.empty()) return err;
    const double t2 = dbg_steps ? now_ms() : 0.0;
    HIP_TRY(hipDeviceSynchronize());
    const double t3 = dbg_steps ? now_ms() : 0.0;
    err = device_to_host_rows(g, witness, g->h_out, out_b);
    if (!err.empty()) return err;
    HIP_TRY(hipMemcpy(set_status, g->h_st, batch * 4, hipMemcpyDeviceToHost));
    if (dbg_steps && now_ms() - t0 > 20.0)
        fprintf(stderr, "run_host: device buffers + rows in %.1f ms, run_device (choice, upload, workspace, launches) %.1f ms, wait for the device %.1f ms, rows out %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
    return "";
}

std::string set_status_text(uint32_t bits) {
    std::string s;
    if (bits & ST_SHL_OVERFLOW) s += "Shl result does not fit the field (reference panics at graph.rs:634)";
    if (bits & 0x80000000u) s += std::string(s.empty() ? "" : "; ") + "internal error: divider mailbox wait timed out";
    if (bits & 0x40000000u) s += std::string(s.empty() ? "" : "; ") + "internal error: wait for another stream's post timed out";
    if (bits & ST_BITOP_EQ_R) s += std::string(s.empty() ? "" : "; ") + "bit operation result equals the modulus (reference panics at graph.rs:686/701/716)";
    return s;
}

// First use of the device in a process: the runtime's initialisation (~60 ms), the device context its first allocation
// makes (~90 ms) and the load of this library's code object are started on a thread of their own by the single-shot entry
// point, beside the host's parsing and compiling of a new graph (~120 ms for the authV2-class graph) -- the calling thread
// does its host work first and touches the device last (it then waits on the runtime's own locks for what is left).
// Errors are left to the calling thread's own checks.
void warm_device() {
    // (joined when the process exits -- an error return may leave the caller free to exit while the runtime is still
    // coming up on this thread; the holder is made on first use, so it is destroyed before the runtime's own statics)
    struct Joined {
        std::thread t;
        ~Joined() {
            if (t.joinable()) t.join();
        }
    };
    static Joined warm;
    static std::once_flag once;
    std::call_once(once, []() {
        if (getenv("CWC_NO_WARM_THREAD")) return;
        warm.t = std::thread([]() {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
                (void)hipGetLastError();
                return;
            }
            void* p = nullptr;  // (the first allocation makes the device context: ~90 ms)
            if (hipMalloc(&p, 4096) == hipSuccess) (void)hipFree(p);
            else (void)hipGetLastError();
            if (launch_warm(nullptr) != hipSuccess) (void)hipGetLastError();
        });
    });
}

}  // namespace cwcrt
