// C-ABI runtime: gw_calc_witness (drop-in for reference src/lib.rs:44-111) and the additive batch API
// (include/graph_witness_batch.h).  Everything numeric runs in the HIP kernels of kernels.hip; this file
// parses, compiles, moves buffers and launches.  There is deliberately no CPU evaluation path.
#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <deque>
#include <future>
#include <string>
#include <thread>
#include <vector>

#define GW_NO_INLINE_FREE_STATUS
#ifndef CWC_TREE_HASH
#define CWC_TREE_HASH "unstamped"
#endif
#include "../../include/graph_witness_batch.h"
#include "graph.hpp"
#include "program.hpp"

namespace cwc {
hipError_t launch_interp(uint32_t T, uint32_t W, uint32_t pack, uint32_t n_div_requests, const uint32_t* div_lanes, const ProgramDev& p,
                         const WsTable& wst, const void* inputs, uint32_t* status, uint32_t batch, hipStream_t stream, unsigned long long* prof);
hipError_t launch_pack(uint32_t T, const ProgramDev& p, const WsTable& wst, void* out, uint32_t batch, hipStream_t stream, bool montgomery);
hipError_t launch_modmul_ubench(uint32_t n_cus, uint32_t waves_per_simd, uint32_t iters, uint32_t* sink, hipStream_t stream, bool block_multiplier);
hipError_t launch_fill_consts(uint32_t T, const ProgramDev& p, const WsTable& wst, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_warm(hipStream_t stream);
}  // namespace cwc

using namespace cwc;

namespace {

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
// No C++ exception may cross the C boundary: allocation failures on huge or hostile inputs become status ERROR.
template <class F>
int guarded(gw_status_t* st, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(st, "out of memory");
    } catch (const std::exception& e) {
        return fail(st, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(st, "internal error");
    }
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e_);    \
    } while (0)

struct DeviceProgram {
    Program host;
    void* d_blob = nullptr;
    ProgramDev dev{};
    DeviceProgram() = default;
    DeviceProgram(const DeviceProgram&) = delete;
    DeviceProgram& operator=(const DeviceProgram&) = delete;
    ~DeviceProgram() {
        if (d_blob) (void)hipFree(d_blob);
    }
};

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
    dp.dev.has_fused = 0;  // (from the bundle headers themselves: an imported program's statistics are not what the kernel runs)
    for (uint32_t h : p.hdr) dp.dev.has_fused |= (h & HDR_CLASS_MASK) == C_MULF ? 1u : ((h & HDR_CLASS_MASK) == C_SCAN || ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC))) ? 2u : 0u;
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
uint32_t waves_per_workgroup(uint32_t divider, uint64_t tiles, uint32_t streams = 1) {
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

}  // namespace

struct gwb_graph {
    Graph graph;
    bool has_graph = false;
    // metadata that exists for loaded and imported handles alike
    std::vector<InputSignal> inputs;
    std::unordered_map<std::string, uint32_t> input_index;
    uint32_t n_inputs = 0, n_witness = 0;
    ProgramStats stats;
    std::map<uint32_t, std::unique_ptr<DeviceProgram>> progs;
    std::map<uint32_t, std::unique_ptr<Program>> compiled;  // compiled for the cost model, not uploaded (yet)
    std::map<size_t, uint32_t> chosen;                       // batch size -> program key picked by the cost model
    uint32_t forced_T = 0;
    uint32_t last_key = 0;  // program key of the last batch call (gwb_program_stats)
    // Small batches (the single-shot entry point above all): the first call compiles ONE program with one schedule and
    // runs it, while a background task does what every other batch size waits for -- all candidate programs, the search
    // over schedule variants, the cost model's choice; the next call that finds the task finished switches over.
    struct Refined {
        uint32_t best = 0;
        std::map<uint32_t, std::unique_ptr<Program>> programs;
    };
    std::map<size_t, std::future<Refined>> refining;  // by batch size
    std::map<size_t, uint32_t> provisional;            // batch size -> the quick program's key while the task runs
    // the task starts compiling once the call that launched it has its kernels enqueued: eight compiler threads beside the first
    // call's allocations, upload and launches cost that call ~100 ms (measured; allocator and page-fault contention)
    std::shared_ptr<std::atomic<bool>> refine_gate;
    bool cache_written = false;                        // single-shot entry point: the refined program went to the on-disk cache
    std::string cache_path;                            // ... to this file (empty: no cache, or the handle came out of it)
    // buffers of the streaming end-to-end entry point (gwb_calc_witness_json_to_wtns), kept between calls: pinned input rows,
    // device rows in / out / status (double-buffered), pinned staging of the witness copy, streams and events
    struct E2eBufs {
        void* h_rows[2] = {nullptr, nullptr};
        void* d_in[2] = {nullptr, nullptr};
        void* d_out[2] = {nullptr, nullptr};
        void* d_st[2] = {nullptr, nullptr};
        void* stage[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
        hipStream_t compute = nullptr, copy[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr}, slice_done[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
        size_t in_bytes = 0, out_bytes = 0, st_bytes = 0, stage_bytes = 0;
        void release() {
            for (int i = 0; i < 2; ++i) {
                if (h_rows[i]) (void)hipHostFree(h_rows[i]);
                if (d_in[i]) (void)hipFree(d_in[i]);
                if (d_out[i]) (void)hipFree(d_out[i]);
                if (d_st[i]) (void)hipFree(d_st[i]);
                h_rows[i] = d_in[i] = d_out[i] = d_st[i] = nullptr;
                for (int b = 0; b < 3; ++b) {
                    if (stage[i][b]) (void)hipHostFree(stage[i][b]);
                    stage[i][b] = nullptr;
                }
            }
            in_bytes = out_bytes = st_bytes = stage_bytes = 0;
        }
    } e2e;
    // value workspaces ("chunks"): separately allocated groups of tiles (CWC_WORKSPACE_GB each), all covered by ONE
    // launch (the kernel picks the chunk per tile; every tile has its own 32-bit buffer window)
    static const int kMaxLanes = (int)WS_MAX_CHUNKS;
    void* d_vals[kMaxLanes] = {nullptr};
    size_t vals_bytes[kMaxLanes] = {0};
    // which constants the tiles of the workspaces currently hold (fill_consts_kernel runs when this changes)
    const DeviceProgram* filled_prog = nullptr;
    uint64_t filled_tiles_per_chunk = 0;
    size_t filled_chunks = 0;
    bool timing_pending = false;
    gwb_timing_t timing{};
    unsigned long long* d_prof = nullptr;  // diagnostic per-class stamps (gwb_profile_classes), else null
    struct ChunkEvents { hipEvent_t start, after_interp, after_pack; };
    // HIP events of the most recent launches, recorded on their launch streams (the last `last_call_launches` of them
    // belong to the last call); the oldest are recycled beyond kHistory launches
    static const size_t kHistory = 256;
    std::deque<ChunkEvents> pending;
    size_t last_call_launches = 0;
    // host-buffer entry point: device rows and the pinned staging of the witness copy, kept between calls
    void *h_in = nullptr, *h_out = nullptr, *h_st = nullptr;
    size_t h_in_bytes = 0, h_out_bytes = 0, h_st_bytes = 0;
    void* stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    // Calls on one handle share the value workspace and the constant fill: work enqueued on a different stream than the
    // previous call's waits for that call's last kernel (an event recorded behind it), so calls execute in enqueue order
    // whatever streams they name.
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    std::mutex mu;

    void drop_events() {
        for (auto& c : pending) { (void)hipEventDestroy(c.start); (void)hipEventDestroy(c.after_interp); (void)hipEventDestroy(c.after_pack); }
        pending.clear();
    }

    ~gwb_graph() {
        progs.clear();  // (DeviceProgram frees its device blob)
        for (int i = 0; i < kMaxLanes; ++i)
            if (d_vals[i]) (void)hipFree(d_vals[i]);
        drop_events();
        for (void* p : {h_in, h_out, h_st})
            if (p) (void)hipFree(p);
        for (int i = 0; i < 2; ++i) {
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (stage_done[i]) (void)hipEventDestroy(stage_done[i]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (last_done) (void)hipEventDestroy(last_done);
        e2e.release();
        for (int i = 0; i < 2; ++i) {
            if (e2e.copy[i]) (void)hipStreamDestroy(e2e.copy[i]);
            if (e2e.done[i]) (void)hipEventDestroy(e2e.done[i]);
            for (int b = 0; b < 3; ++b)
                if (e2e.slice_done[i][b]) (void)hipEventDestroy(e2e.slice_done[i][b]);
        }
        if (e2e.compute) (void)hipStreamDestroy(e2e.compute);
    }
};

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

namespace {

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

uint64_t fnv1a(const uint8_t* p, size_t n);
uint64_t blob_checksum(const uint8_t* p, size_t n);
// What gwb_graph_export hands out / the on-disk cache holds: the program blob, the input map, a checksummed trailer
// (written in place: the image of a multi-million-node graph is most of a gigabyte, every copy of it counts).
size_t exported_size(const Program& p, const std::vector<InputSignal>& inputs) {
    size_t n = (program_blob_size(p) + 7) / 8 * 8 + 4;
    for (const InputSignal& s : inputs) n += 12 + s.name.size();
    return n + 24;
}
void exported_write(const Program& p, const std::vector<InputSignal>& inputs, uint8_t* dst) {
    const size_t exact_len = program_blob_size(p), prog_len = (exact_len + 7) / 8 * 8;
    program_blob_write(p, dst);
    uint8_t* q = dst + exact_len;
    while (q < dst + prog_len) *q++ = 0;
    auto put32 = [&](uint32_t v) { memcpy(q, &v, 4); q += 4; };
    put32((uint32_t)inputs.size());
    for (const InputSignal& s : inputs) {
        put32(s.offset);
        put32(s.len);
        put32((uint32_t)s.name.size());
        if (!s.name.empty()) memcpy(q, s.name.data(), s.name.size());
        q += s.name.size();
    }
    // trailer: exact program length, padded program length (= where the input map starts), checksum of everything before
    uint64_t tr[3] = {(uint64_t)exact_len, (uint64_t)prog_len, 0};
    tr[2] = blob_checksum(dst, (size_t)(q - dst));
    memcpy(q, tr, sizeof tr);
}
std::vector<uint8_t> exported_bytes(const Program& p, const std::vector<InputSignal>& inputs) {
    std::vector<uint8_t> b(exported_size(p, inputs));
    exported_write(p, inputs, b.data());
    return b;
}
void write_file_atomically(const std::string& path, const void* data, size_t n);
std::vector<uint8_t> cache_wrap(const std::string& path, const void* blob, size_t n);

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
uint32_t pick_tile_width(gwb_graph* g, size_t batch, bool allow_quick = true) {
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
                       hipStream_t stream, bool montgomery = false, hipEvent_t done_event = nullptr) {
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

// ---- compiled-graph cache for the single-shot entry point (the reference re-parses per call, lib.rs:129) ----
struct CacheEntry {
    uint64_t hash;
    std::vector<uint8_t> bytes;  // the graph image itself: a hit is a byte-for-byte match, never a hash alone
    std::shared_ptr<gwb_graph> g;
};
std::mutex g_cache_mu;
// (never destroyed: the handles own HIP objects and static destructors run after the HIP runtime may be gone)
std::vector<CacheEntry>& g_cache = *new std::vector<CacheEntry>();

// length + sixteen 64-byte windows spread over the buffer (the graph cache's lookup key; equality is a memcmp)
uint64_t sampled_fingerprint(const uint8_t* p, size_t n);
uint64_t fnv1a(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}
// Checksum of an exported image (format 15; formats up to 14 used the byte-serial FNV-1a above, 1.2 s for the 0.9 GB
// program of a 10.5 M-node graph and a quarter of the cache-hit first call): a position-dependent sum over little-endian
// 64-bit words -- every word is mixed with its index on its own, so the loop has no serial dependency beyond the
// addition; the tail is zero-padded to a word, the length is folded in.  It guards against truncation and corruption in
// transit / on disk, it is no authentication; the structural validation follows it.
uint64_t blob_checksum(const uint8_t* p, size_t n) {
    const uint64_t K1 = 0x9E3779B97F4A7C15ull, K2 = 0xC2B2AE3D27D4EB4Full, K3 = 0x165667B19E3779F9ull;
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    const size_t words = n / 8;
    auto term = [&](uint64_t w, uint64_t i) {
        uint64_t x = (w ^ (i * K1)) * K2;
        return x ^ (x >> 29);
    };
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
        uint64_t w[4];
        memcpy(w, p + 8 * i, 32);
        h0 += term(w[0], i);
        h1 += term(w[1], i + 1);
        h2 += term(w[2], i + 2);
        h3 += term(w[3], i + 3);
    }
    for (; i < words; ++i) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        h0 += term(w, i);
    }
    if (n % 8) {
        uint64_t w = 0;
        memcpy(&w, p + 8 * words, n % 8);
        h0 += term(w, words);
    }
    uint64_t h = h0 + h1 + h2 + h3 + (uint64_t)n * K3;
    h ^= h >> 32;
    h *= K1;
    return h ^ (h >> 29);
}
uint64_t sampled_fingerprint(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    const size_t win = 64, k = 16;
    if (n <= win * k) return fnv1a(p, n) ^ (uint64_t)n;
    for (size_t w = 0; w < k; ++w) {
        const size_t off = (n - win) / (k - 1) * w;
        for (size_t q = 0; q < win; ++q) h = (h ^ p[off + q]) * 1099511628211ull;
    }
    return h;
}

// SHA-256 (FIPS 180-4) of the graph image: the key of the on-disk program cache
std::string sha256_hex(const uint8_t* p, size_t n) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
        0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
        0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
        0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto rotr = [](uint32_t x, int k) { return (x >> k) | (x << (32 - k)); };
    auto block = [&](const uint8_t* b) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & bb) ^ (a & c) ^ (bb & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    };
    size_t i = 0;
    for (; i + 64 <= n; i += 64) block(p + i);
    uint8_t tail[128] = {0};
    const size_t rem = n - i;
    memcpy(tail, p + i, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; ++k) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    block(tail);
    if (tl == 128) block(tail + 64);
    char out[65];
    for (int k = 0; k < 8; ++k) snprintf(out + 8 * k, 9, "%08x", h[k]);
    return std::string(out, 64);
}

// On-disk cache of the single-shot entry point's compiled program: a process that has never seen a graph image finds the
// program an earlier process compiled for it (the first call otherwise parses, compiles and searches schedules).  One file
// per (SHA-256 of the image, library build): <dir>/<sha256>-<build>.cwcprog = the blob of gwb_graph_export (checksummed,
// structurally validated on import: a damaged or stale file is ignored and rewritten).  CWC_PROGRAM_CACHE=<dir> names the
// directory, CWC_PROGRAM_CACHE=0 turns the cache off; default $XDG_CACHE_HOME or ~/.cache, /circom-witnesscalc-amd.
std::string program_cache_file(const void* graph_data, size_t len) {
    std::string dir;
    if (const char* e = getenv("CWC_PROGRAM_CACHE")) {
        if (!*e || !strcmp(e, "0") || !strcmp(e, "off")) return "";
        dir = e;
    } else if (const char* x = getenv("XDG_CACHE_HOME")) {
        if (*x) dir = std::string(x) + "/circom-witnesscalc-amd";
    }
    if (dir.empty()) {
        const char* home = getenv("HOME");
        if (!home || !*home) return "";
        dir = std::string(home) + "/.cache/circom-witnesscalc-amd";
    }
    // (this build -- the content hash of every source under csrc/ as the Makefile stamped it, not a timestamp: the compiler
    // and the kernels that give a program its meaning are compiled separately from this file --, the program format, the
    // cost model's cycle table: a program is chosen under one table)
    static const std::string build = []() {
        const std::string id = std::string(CWC_TREE_HASH " format 16 table ") + std::to_string((unsigned long long)model_table_id());
        return sha256_hex((const uint8_t*)id.data(), id.size()).substr(0, 16);
    }();
    return dir + "/" + sha256_hex((const uint8_t*)graph_data, len) + "-" + build + ".cwcprog";
}
// A cache file = 96 bytes that name what it is for -- "CWCPROG2", then <sha256 of the graph image>-<build> as in its file name,
// zero-padded -- followed by the blob of gwb_graph_export: a file that was renamed or copied over another entry, or written
// by another build under a colliding name, does not pass for this graph's program.
static const size_t kCacheHeader = 96;
std::string cache_entry_name(const std::string& path) {
    const size_t slash = path.rfind('/'), dot = path.rfind(".cwcprog");
    const size_t a = slash == std::string::npos ? 0 : slash + 1;
    return dot == std::string::npos || dot < a ? path.substr(a) : path.substr(a, dot - a);
}
std::vector<uint8_t> cache_wrap(const std::string& path, const void* blob, size_t n) {
    std::vector<uint8_t> out(kCacheHeader + n, 0);
    memcpy(out.data(), "CWCPROG2", 8);
    const std::string name = cache_entry_name(path);
    memcpy(out.data() + 8, name.data(), std::min(name.size(), kCacheHeader - 8));
    memcpy(out.data() + kCacheHeader, blob, n);
    return out;
}
bool cache_unwrap(const std::string& path, const std::vector<uint8_t>& file, const uint8_t** blob, size_t* n) {
    if (file.size() < kCacheHeader || memcmp(file.data(), "CWCPROG2", 8) != 0) return false;
    const std::string name = cache_entry_name(path);
    uint8_t want[kCacheHeader - 8] = {0};
    memcpy(want, name.data(), std::min(name.size(), sizeof want));
    if (name.size() > sizeof want || memcmp(file.data() + 8, want, sizeof want) != 0) return false;
    *blob = file.data() + kCacheHeader;
    *n = file.size() - kCacheHeader;
    return true;
}
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long n = ok ? ftell(f) : -1;
    ok = ok && n >= 0 && n < (1l << 31) && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out.resize((size_t)n);
        ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    return ok;
}
void write_file_atomically(const std::string& path, const void* data, size_t n) {
    const size_t slash = path.rfind('/');
    if (slash != std::string::npos) {  // mkdir -p of the directory (two levels are enough for the default)
        const std::string dir = path.substr(0, slash);
        const size_t up = dir.rfind('/');
        if (up != std::string::npos && up > 0) (void)mkdir(dir.substr(0, up).c_str(), 0700);
        (void)mkdir(dir.c_str(), 0700);
    }
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data, 1, n, f) == n;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

int load_graph(const void* data, size_t len, gwb_graph** out, std::string& err) {
    std::unique_ptr<gwb_graph> g(new gwb_graph());
    if (!deserialize_witnesscalc_graph((const uint8_t*)data, len, g->graph, err)) return 1;
    g->has_graph = true;
    // validation + statistics (bad indices / Pow / Id are caught here); programs are compiled when a batch size is known
    Program probe;
    if (!probe_graph(g->graph, probe, err)) return 1;
    g->stats = probe.stats;
    g->n_inputs = probe.n_inputs;
    g->n_witness = probe.n_witness;
    g->inputs = g->graph.inputs;
    g->input_index = g->graph.input_index;
    *out = g.release();
    return 0;
}

bool quirks() {
    const char* e = getenv("GW_REFERENCE_QUIRKS");
    return e && *e && strcmp(e, "0") != 0;
}

}  // namespace

// =====================================================================================================
// exported C symbols
// =====================================================================================================
extern "C" int gwb_kernels_have_diagnostics();  // kernels.hip

extern "C" {

void gwb_free_status(gw_status_t* status) {
    if (status && status->error_msg) {
        free(status->error_msg);
        status->error_msg = nullptr;
    }
}

int gwb_graph_load(const void* graph_data, size_t len, gwb_graph_t** out, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!graph_data) return fail(status, "graph_data is null");
    if (len == 0) return fail(status, "graph_data_len is 0");
    if (!out) return fail(status, "out is null");
    std::string err;
    if (load_graph(graph_data, len, out, err)) return fail(status, "Failed to load graph: " + err);
    set_status(status, OK, "");
    return 0;
    });
}

void gwb_graph_free(gwb_graph_t* g) { delete g; }

int gwb_graph_info(const gwb_graph_t* g, gwb_graph_info_t* info) {
    if (!g || !info) return 1;
    info->n_nodes = g->stats.n_nodes;
    info->n_op = g->stats.n_op;
    info->n_input_nodes = g->stats.n_input_nodes;
    info->n_const = g->stats.n_const;
    info->n_inputs = g->n_inputs;
    info->n_witness = g->n_witness;
    info->depth = g->stats.depth;
    info->algorithmic_bytes_per_set = g->stats.algorithmic_bytes_per_set;
    return 0;
}

int gwb_graph_serialize(const gwb_graph_t* g, void** out, size_t* out_len, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !out || !out_len) return fail(status, "null argument");
    if (!g->has_graph) return fail(status, "imported handle holds no graph to serialize");
    std::vector<uint8_t> b = serialize_witnesscalc_graph(g->graph);
    *out = malloc(b.size() ? b.size() : 1);
    if (!*out) return fail(status, "out of memory");
    memcpy(*out, b.data(), b.size());
    *out_len = b.size();
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_inputs_from_json(const gwb_graph_t* g, const char* json, void* row, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !json || !row) return fail(status, "null argument");
    InputList list;
    std::string err;
    if (!deserialize_inputs(json, strlen(json), list, err)) return fail(status, "Failed to calculate witness: " + err);
    Graph meta;  // populate_inputs only needs the input map
    meta.inputs = g->inputs;
    meta.input_index = g->input_index;
    if (!populate_inputs(list, meta, (uint8_t*)row, g->n_inputs, err)) return fail(status, "Failed to calculate witness: " + err);
    if (quirks())
        for (const auto& kv : list) {  // lib.rs:162
            const InputSignal& s = g->inputs[g->input_index.at(kv.first)];
            printf("input %s, offset %u, len %u\n", kv.first.c_str(), s.offset, s.len);
        }
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_inputs_from_json_batch(const gwb_graph_t* g, const char* text, size_t text_len, void* rows, size_t max_rows,
                               size_t* n_rows, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !text || !n_rows || (!rows && max_rows)) return fail(status, "null argument");
    std::vector<std::pair<size_t, size_t>> spans;
    std::string err;
    if (!split_inputs_batch(text, text_len, spans, err)) return fail(status, "Failed to calculate witness: " + err);
    *n_rows = spans.size();
    if (spans.size() > max_rows) return fail(status, "rows buffer too small: " + std::to_string(spans.size()) + " input sets");
    Graph meta;
    meta.inputs = g->inputs;
    meta.input_index = g->input_index;
    // the input sets are independent: parsed on CWC_PARSE_THREADS host threads (default: every core), contiguous
    // ranges each; the error of the lowest failing set is reported, as a sequential loop would
    unsigned n_threads = env_threads("CWC_PARSE_THREADS", 0);  // (round 2 capped this at 16 threads: 27 k sets/s on a 256-core host)
    if (n_threads > spans.size() / 16 + 1) n_threads = (unsigned)(spans.size() / 16 + 1);
    if (spans.size() < 64) n_threads = 1;
    if (n_threads > spans.size()) n_threads = (unsigned)spans.size();
    std::vector<std::string> errs(n_threads ? n_threads : 1);
    std::vector<size_t> bad(n_threads ? n_threads : 1, (size_t)-1);
    auto work = [&](unsigned w) {  // (no exception leaves a worker thread: it would end the process)
        const size_t lo = spans.size() * w / n_threads, hi = spans.size() * (w + 1) / n_threads;
        size_t i = lo;
        try {
            InputList list;
            for (; i < hi; ++i) {
                std::string e;
                if (!deserialize_inputs(text + spans[i].first, spans[i].second - spans[i].first, list, e) ||
                    !populate_inputs(list, meta, (uint8_t*)rows + i * (size_t)g->n_inputs * 32, g->n_inputs, e)) {
                    errs[w] = e;
                    bad[w] = i;
                    return;
                }
            }
        } catch (const std::bad_alloc&) {
            errs[w] = "out of memory";
            bad[w] = i;
        } catch (...) {
            errs[w] = "internal error";
            bad[w] = i;
        }
    };
    if (n_threads <= 1) {
        n_threads = 1;
        if (!spans.empty()) work(0);
    } else {
        std::vector<std::thread> th;
        struct Joiner {  // every started worker is joined on every way out (a thread that cannot be started included)
            std::vector<std::thread>& th;
            ~Joiner() {
                for (auto& t : th)
                    if (t.joinable()) t.join();
            }
        } joiner{th};
        th.reserve(n_threads);
        unsigned started = 1;
        try {
            for (; started < n_threads; ++started) th.emplace_back(work, started);
        } catch (...) {  // (thread creation failed: the sets of the missing workers are parsed here)
        }
        work(0);
        for (unsigned w = started; w < n_threads; ++w) work(w);
    }
    for (unsigned w = 0; w < n_threads; ++w)
        if (bad[w] != (size_t)-1) return fail(status, "Failed to calculate witness: input set " + std::to_string(bad[w]) + ": " + errs[w]);
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_wtns_save_batch(const void* witness, size_t n_witness, size_t batch, const char* path_pattern, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    // one `.wtns` file per input set; path_pattern must contain one %zu / %lu-style conversion for the set index
    if ((!witness && batch) || !path_pattern) return fail(status, "null argument");
    std::vector<uint8_t> hdr(76);
    wtns_write_header(hdr.data(), n_witness);
    char path[4096];
    for (size_t i = 0; i < batch; ++i) {
        const int n = snprintf(path, sizeof path, path_pattern, (unsigned long)i);
        if (n <= 0 || (size_t)n >= sizeof path) return fail(status, "bad path pattern");
        FILE* f = fopen(path, "wb");
        if (!f) return fail(status, std::string("cannot open ") + path);
        const bool ok = fwrite(hdr.data(), 1, 76, f) == 76 &&
                        fwrite((const uint8_t*)witness + i * n_witness * 32, 1, n_witness * 32, f) == n_witness * 32;
        if (fclose(f) != 0 || !ok) return fail(status, std::string("short write to ") + path);
    }
    set_status(status, OK, "");
    return 0;
    });
}

// ---- end to end, streaming: JSON text -> rows -> HBM -> kernels -> pinned staging -> `.wtns` files ----------
// Sub-batches of CWC_E2E_SUBBATCH input sets (default 1024) move through a three-stage pipeline: the calling thread parses
// sub-batch k + 1 on the parse threads and enqueues its upload and kernels, while a drain thread copies the witness rows of
// sub-batch k out of HBM in slices of whole sets (copy stream, pinned staging buffers) and a pool of writer threads frames
// every set of a finished slice as its own `.wtns` file (76-byte header + row, lib.rs:114-123).  The interpreter's value
// workspace is shared, the output rows are double-buffered.  The rate is the PCIe link's: 2.4 MB of witness per authV2 set.
namespace {
struct WriterPool {
    struct Task {
        const uint8_t* row;
        size_t index;
        std::atomic<int>* pending;  // of the staging buffer the row lives in
    };
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::deque<Task> q;
    bool stop = false;
    std::string err;
    std::vector<std::thread> th;
    std::string pattern;
    size_t n_witness = 0;
    std::vector<uint8_t> hdr;
    void start(unsigned n, const char* pat, size_t nw) {
        pattern = pat;
        n_witness = nw;
        hdr.resize(76);
        wtns_write_header(hdr.data(), nw);
        for (unsigned i = 0; i < n; ++i) th.emplace_back([this]() { run(); });
    }
    void run() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return stop || !q.empty(); });
                if (q.empty()) return;
                t = q.front();
                q.pop_front();
            }
            char path[4096];
            std::string e;
            const int n = snprintf(path, sizeof path, pattern.c_str(), (unsigned long)t.index);
            if (n <= 0 || (size_t)n >= sizeof path) {
                e = "bad path pattern";
            } else {
                FILE* f = fopen(path, "wb");
                if (!f) {
                    e = std::string("cannot open ") + path;
                } else {
                    const bool ok = fwrite(hdr.data(), 1, 76, f) == 76 && fwrite(t.row, 1, n_witness * 32, f) == n_witness * 32;
                    if (fclose(f) != 0 || !ok) e = std::string("short write to ") + path;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty() && err.empty()) err = e;
                t.pending->fetch_sub(1, std::memory_order_release);
            }
            cv_done.notify_all();
        }
    }
    void push(const Task& t) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(t);
        }
        cv.notify_one();
    }
    void wait_zero(std::atomic<int>& c) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&]() { return c.load(std::memory_order_acquire) == 0; });
    }
    void finish() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
    ~WriterPool() { finish(); }
};
}  // namespace

extern "C" int gwb_calc_witness_json_to_wtns(gwb_graph_t* g, const char* text, size_t text_len, const char* path_pattern, size_t first_index,
                                             size_t* n_sets, uint32_t* set_status_out, size_t max_sets, gwb_e2e_stats_t* stats, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !text || !path_pattern || !n_sets) return fail(status, "null argument");
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    std::vector<std::pair<size_t, size_t>> spans;
    std::string err;
    if (!split_inputs_batch(text, text_len, spans, err)) return fail(status, "Failed to calculate witness: " + err);
    *n_sets = spans.size();
    if (set_status_out && spans.size() > max_sets) return fail(status, "status buffer too small: " + std::to_string(spans.size()) + " input sets");
    if (spans.empty()) {
        set_status(status, OK, "");
        return 0;
    }
    std::lock_guard<std::mutex> lk(g->mu);
    err = check_device();
    if (!err.empty()) return fail(status, err);
    const size_t B = spans.size(), NI = g->n_inputs, NW = g->n_witness, row_b = NW * 32;
    size_t S = 1024;  // (measured on MI355X, authV2-class: 256 -> 10.6 k, 512 -> 14.1 k, 1024 -> 14.3 k witnesses/s; profiles/r03_e2e_ab.txt)
    if (const char* e = getenv("CWC_E2E_SUBBATCH")) {
        const long v = atol(e);
        if (v >= 1) S = (size_t)v;
    }
    if (S > B) S = B;
    const size_t K = (B + S - 1) / S;
    // slices of whole sets, three staging buffers per drain.  CWC_E2E_SLICE_MB (default 96): a device-to-host copy costs
    // ~0.2 ms before it moves anything, so 24 MB slices ran the link at 35 GB/s where one large copy reaches 57
    size_t slice_mb = 96;
    if (const char* e = getenv("CWC_E2E_SLICE_MB")) {
        const long v = atol(e);
        if (v >= 1 && v <= 1024) slice_mb = (size_t)v;
    }
    size_t slice_sets = row_b ? std::max<size_t>(1, (slice_mb << 20) / row_b) : 1;
    if (slice_sets > S) slice_sets = S;
    const int kStage = 3;
    gwb_graph::E2eBufs& bf = g->e2e;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipSuccess;
    };
    struct SyncOnExit {  // nothing of this call is in flight when it returns (the buffers stay on the handle)
        ~SyncOnExit() { (void)hipDeviceSynchronize(); }
    } sync_on_exit;
    bool ok = true;
    if (!bf.compute) ok = hip_ok(hipStreamCreateWithFlags(&bf.compute, hipStreamNonBlocking), "hipStreamCreate");
    for (int i = 0; ok && i < 2; ++i) {
        if (!bf.copy[i]) ok = hip_ok(hipStreamCreateWithFlags(&bf.copy[i], hipStreamNonBlocking), "hipStreamCreate");
        if (ok && !bf.done[i]) ok = hip_ok(hipEventCreateWithFlags(&bf.done[i], hipEventDisableTiming), "hipEventCreate");
        for (int b = 0; ok && b < kStage; ++b)
            if (!bf.slice_done[i][b]) ok = hip_ok(hipEventCreateWithFlags(&bf.slice_done[i][b], hipEventDisableTiming), "hipEventCreate");
    }
    auto& slice_done = bf.slice_done;
    const size_t need_in = std::max<size_t>(32, S * NI * 32), need_out = std::max<size_t>(32, S * row_b), need_st = S * 4, need_stage = std::max<size_t>(32, slice_sets * row_b);
    if (ok && (need_in > bf.in_bytes || need_out > bf.out_bytes || need_st > bf.st_bytes || need_stage > bf.stage_bytes)) {
        (void)hipDeviceSynchronize();
        bf.release();
        for (int i = 0; ok && i < 2; ++i) {
            ok = hip_ok(hipHostMalloc(&bf.h_rows[i], need_in, hipHostMallocDefault), "hipHostMalloc") && hip_ok(hipMalloc(&bf.d_in[i], need_in), "hipMalloc") &&
                 hip_ok(hipMalloc(&bf.d_out[i], need_out), "hipMalloc") && hip_ok(hipMalloc(&bf.d_st[i], need_st), "hipMalloc");
            for (int b = 0; ok && b < kStage; ++b) ok = hip_ok(hipHostMalloc(&bf.stage[i][b], need_stage, hipHostMallocDefault), "hipHostMalloc");
        }
        if (ok) {
            bf.in_bytes = need_in;
            bf.out_bytes = need_out;
            bf.st_bytes = need_st;
            bf.stage_bytes = need_stage;
        } else {
            bf.release();
        }
    }
    if (!ok) return fail(status, err);
    Graph meta;
    meta.inputs = g->inputs;
    meta.input_index = g->input_index;
    const unsigned n_parse = env_threads("CWC_PARSE_THREADS", 0), n_write = env_threads("CWC_WRITE_THREADS", 16)  /* (more writers fight the copy engine for host memory bandwidth: 64 -> 9.4 k, 32 -> 13.9 k, 16 -> 15.5 k witnesses/s, r03_e2e_ab.txt) */;
    WriterPool pool;
    pool.start(n_write, path_pattern, NW);
    std::atomic<int> pending[2][3];
    for (auto& a : pending)
        for (auto& x : a) x.store(0);
    double parse_s = 0;
    std::mutex err_mu;
    std::string drain_err, parse_err;
    size_t parse_bad = (size_t)-1;
    std::vector<uint32_t> st_host(B, 0);
    // (joined on every way out of this function: an exception that unwinds past a joinable std::thread ends the process)
    struct Drains {
        std::thread t[2];
        ~Drains() {
            for (auto& d : t)
                if (d.joinable()) d.join();
        }
    } drain_threads;
    std::thread* const drains = drain_threads.t;
    auto drain = [&](size_t k) {  // witness rows of sub-batch k: HBM -> staging -> files
        const int par = (int)(k & 1);
        const size_t lo = k * S, n = std::min(S, B - lo);
        if (hipEventSynchronize(bf.done[par]) != hipSuccess) {
            std::lock_guard<std::mutex> l2(err_mu);
            if (drain_err.empty()) drain_err = "hipEventSynchronize failed";
            return;
        }
        (void)hipMemcpy(st_host.data() + lo, bf.d_st[par], n * 4, hipMemcpyDeviceToHost);
        // the copy of slice i + 1 is enqueued before the host waits for slice i: the copy engine never idles between slices
        const size_t n_slices = (n + slice_sets - 1) / slice_sets;
        auto issue = [&](size_t i) -> bool {
            const int bb = (int)(i % kStage);
            const size_t s0 = i * slice_sets, m = std::min(slice_sets, n - s0);
            pool.wait_zero(pending[par][bb]);  // the writers are done with what this buffer held
            return hipMemcpyAsync(bf.stage[par][bb], (const char*)bf.d_out[par] + s0 * row_b, m * row_b, hipMemcpyDeviceToHost, bf.copy[par]) == hipSuccess &&
                   hipEventRecord(slice_done[par][bb], bf.copy[par]) == hipSuccess;
        };
        bool okc = issue(0);
        for (size_t i = 0; okc && i < n_slices; ++i) {
            if (i + 1 < n_slices) okc = issue(i + 1);
            const int bb = (int)(i % kStage);
            const size_t s0 = i * slice_sets, m = std::min(slice_sets, n - s0);
            okc = okc && hipEventSynchronize(slice_done[par][bb]) == hipSuccess;
            if (!okc) break;
            // a set whose status word is not zero (the reference panics there: Shl overflow, bit operation == r) gets NO file --
            // a well-formed `.wtns` of a failed evaluation could not be told from a valid one -- and a file of that name left by
            // an earlier run is removed
            size_t m_ok = 0;
            for (size_t q = 0; q < m; ++q) m_ok += st_host[lo + s0 + q] == 0;
            pending[par][bb].store((int)m_ok, std::memory_order_release);
            for (size_t q = 0; q < m; ++q) {
                if (st_host[lo + s0 + q] == 0) {
                    pool.push(WriterPool::Task{(const uint8_t*)bf.stage[par][bb] + q * row_b, first_index + lo + s0 + q, &pending[par][bb]});
                } else {
                    char path[4096];
                    const int pn = snprintf(path, sizeof path, path_pattern, (unsigned long)(first_index + lo + s0 + q));
                    if (pn > 0 && (size_t)pn < sizeof path) (void)remove(path);
                }
            }
        }
        if (!okc) {
            (void)hipStreamSynchronize(bf.copy[par]);
            std::lock_guard<std::mutex> l2(err_mu);
            if (drain_err.empty()) drain_err = "device-to-host copy of the witness rows failed";
            return;
        }
        for (int q = 0; q < kStage; ++q) pool.wait_zero(pending[par][q]);
    };
    double compute_wait_s = 0;
    for (size_t k = 0; k < K && err.empty(); ++k) {
        const int par = (int)(k & 1);
        const size_t lo = k * S, n = std::min(S, B - lo);
        if (drains[par].joinable()) {  // sub-batch k - 2 used these buffers
            const auto t0 = std::chrono::steady_clock::now();
            drains[par].join();
            compute_wait_s += since(t0);
        }
        // parse sub-batch k (contiguous ranges per thread; the lowest failing set is reported)
        const auto tp = std::chrono::steady_clock::now();
        {
            unsigned nt = n_parse;
            if (n < 64) nt = 1;
            if (nt > n) nt = (unsigned)n;
            std::vector<std::string> errs(nt);
            std::vector<size_t> bad(nt, (size_t)-1);
            auto work = [&](unsigned w) {
                const size_t a = n * w / nt, bnd = n * (w + 1) / nt;
                size_t i = a;
                try {
                    InputList list;
                    for (; i < bnd; ++i) {
                        std::string e;
                        const auto& sp = spans[lo + i];
                        if (!deserialize_inputs(text + sp.first, sp.second - sp.first, list, e) ||
                            !populate_inputs(list, meta, (uint8_t*)bf.h_rows[par] + i * NI * 32, NI, e)) {
                            errs[w] = e;
                            bad[w] = lo + i;
                            return;
                        }
                    }
                } catch (...) {
                    errs[w] = "out of memory";
                    bad[w] = lo + i;
                }
            };
            std::vector<std::thread> th;
            unsigned started = 1;
            try {
                for (; started < nt; ++started) th.emplace_back(work, started);
            } catch (...) {
            }
            work(0);
            for (unsigned w = started; w < nt; ++w) work(w);
            for (auto& t : th) t.join();
            for (unsigned w = 0; w < nt; ++w)
                if (bad[w] != (size_t)-1 && bad[w] < parse_bad) {
                    parse_bad = bad[w];
                    parse_err = errs[w];
                }
        }
        parse_s += since(tp);
        if (parse_bad != (size_t)-1) {
            err = "Failed to calculate witness: input set " + std::to_string(parse_bad) + ": " + parse_err;
            break;
        }
        if (hipMemcpyAsync(bf.d_in[par], bf.h_rows[par], n * NI * 32, hipMemcpyHostToDevice, bf.compute) != hipSuccess) {
            err = "host-to-device copy of the input rows failed";
            break;
        }
        err = run_device(g, bf.d_in[par], n, bf.d_out[par], (uint32_t*)bf.d_st[par], bf.compute);
        if (!err.empty()) break;
        if (hipEventRecord(bf.done[par], bf.compute) != hipSuccess) {
            err = "hipEventRecord failed";
            break;
        }
        try {
            drains[par] = std::thread(drain, k);
        } catch (const std::system_error&) {  // no thread to be had: this sub-batch is drained by the calling thread
            drain(k);
        }
    }
    for (int q = 0; q < 2; ++q)
        if (drains[q].joinable()) drains[q].join();
    pool.finish();
    if (err.empty()) err = drain_err;
    if (err.empty()) err = pool.err;
    if (!err.empty()) return fail(status, err);
    if (set_status_out) memcpy(set_status_out, st_host.data(), B * 4);
    size_t n_failed = 0, first_failed = 0;
    for (size_t i = B; i-- > 0;)
        if (st_host[i]) {
            ++n_failed;
            first_failed = i;
        }
    if (stats) stats->failed_sets = n_failed;
    // without a status buffer nobody could tell which sets have no file: the call itself fails (as the reference's single call does)
    if (n_failed && !set_status_out)
        return fail(status, "Failed to calculate witness: input set " + std::to_string(first_failed) + ": " + set_status_text(st_host[first_failed]) + " (" +
                                std::to_string(n_failed) + " of " + std::to_string(B) + " input sets failed; no file was written for them)");
    if (stats) {
        stats->n_sets = B;
        stats->sub_batch = S;
        stats->parse_threads = n_parse;
        stats->write_threads = n_write;
        stats->parse_seconds = parse_s;
        stats->wait_for_drain_seconds = compute_wait_s;
        stats->total_seconds = since(t_start);
        stats->witness_bytes = (uint64_t)B * row_b;
    }
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_set_tile_width(gwb_graph_t* g, uint32_t key) {
    const uint32_t T = key & ~KEY_MODE_MASK;
    const uint32_t mode = key & (KEY_DIVIDER | KEY_GROUP | KEY_TRIPLE), smode = key & (KEY_STREAMS2 | KEY_STREAMS4);
    if (!g || T > 64 || (T & (T - 1)) || ((mode | smode) && T == 0) || (mode & (mode - 1)) || (smode & (smode - 1))) return 1;  // (at most one divider mode, one stream count)
    if (smode && (mode & (KEY_GROUP | KEY_TRIPLE))) return 1;  // (streams have a divider wave each, or none)
    g->forced_T = key;
    return 0;
}

uint32_t gwb_graph_pick_tile_width(gwb_graph_t* g, size_t batch) {
    // the program key the cost model chooses for this graph and batch size (compiles the candidates on the host; no device
    // needed): what rank 0 exports and broadcasts to the other GPUs of a node
    if (!g) return 0;
    try {
        std::lock_guard<std::mutex> lk(g->mu);
        return pick_tile_width(g, batch, false);  // (what is asked for here is exported / broadcast: the searched program, not the quick first one)
    } catch (...) {
        return 0;
    }
}

int gwb_calc_witness_batch_device(gwb_graph_t* g, const void* d_inputs, size_t batch, void* d_witness,
                                  uint32_t* d_set_status, void* hip_stream, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || (batch && (!d_inputs || !d_witness || !d_set_status))) return fail(status, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    std::string err = check_device();
    if (err.empty()) err = run_device(g, d_inputs, batch, d_witness, d_set_status, (hipStream_t)hip_stream);
    if (!err.empty()) return fail(status, err);
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_calc_witness_batch_handoff(gwb_graph_t* g, const void* d_inputs, size_t batch, void* d_witness, uint32_t* d_set_status,
                                   const gwb_handoff_t* h, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !h || (batch && (!d_inputs || !d_witness || !d_set_status))) return fail(status, "null argument");
    if (h->struct_size != sizeof(gwb_handoff_t)) return fail(status, "gwb_handoff_t: struct_size does not match this library");
    if (h->form != GWB_FORM_CANONICAL && h->form != GWB_FORM_MONTGOMERY) return fail(status, "gwb_handoff_t: unknown form");
    std::lock_guard<std::mutex> lk(g->mu);
    std::string err = check_device();
    if (err.empty()) err = run_device(g, d_inputs, batch, d_witness, d_set_status, (hipStream_t)h->hip_stream, h->form == GWB_FORM_MONTGOMERY, (hipEvent_t)h->done_event);
    if (err.empty() && batch == 0 && h->done_event && hipEventRecord((hipEvent_t)h->done_event, (hipStream_t)h->hip_stream) != hipSuccess) err = "hipEventRecord failed";
    if (!err.empty()) return fail(status, err);
    set_status(status, OK, "");
    return 0;
    });
}

static double ubench_modmul(uint32_t waves_per_simd, uint32_t iters, bool block_multiplier);
double gwb_ubench_modmul(uint32_t waves_per_simd, uint32_t iters) { return ubench_modmul(waves_per_simd, iters, false); }
double gwb_ubench_modmul_block(uint32_t waves_per_simd, uint32_t iters) { return ubench_modmul(waves_per_simd, iters, true); }
double gwb_model_class_cycles(uint32_t bundle_class) { return model_class_cycles((int)bundle_class); }
static double ubench_modmul(uint32_t waves_per_simd, uint32_t iters, bool block_multiplier) {
    // chip-wide one-lane Montgomery products per second with `waves_per_simd` waves on every SIMD (bench.py's compute
    // ceiling, measured in the same run); 0 on failure
    try {
        if (!check_device().empty() || waves_per_simd < 1 || waves_per_simd > (block_multiplier ? 2u : 4u) || iters == 0) return 0.0;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0.0;
        uint32_t* sink = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        double rate = 0.0;
        if (hipMalloc(&sink, 64) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            const uint32_t cus = (uint32_t)prop.multiProcessorCount;
            bool ok = launch_modmul_ubench(cus, waves_per_simd, iters, sink, nullptr, block_multiplier) == hipSuccess && hipDeviceSynchronize() == hipSuccess;  // warm-up
            ok = ok && hipEventRecord(e0, nullptr) == hipSuccess && launch_modmul_ubench(cus, waves_per_simd, iters, sink, nullptr, block_multiplier) == hipSuccess &&
                 hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
            float ms = 0.f;
            if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f)
                rate = 2.0 * iters * (double)cus * 256.0 * waves_per_simd / (ms * 1e-3);
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (sink) (void)hipFree(sink);
        return rate;
    } catch (...) {
        return 0.0;
    }
}

int gwb_calc_witness_batch_host(gwb_graph_t* g, const void* inputs, size_t batch, void* witness, uint32_t* set_status_out,
                                gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || (batch && (!inputs || !witness || !set_status_out))) return fail(status, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    std::string err = check_device();
    if (err.empty()) err = run_host(g, inputs, batch, witness, set_status_out);
    if (!err.empty()) return fail(status, err);
    set_status(status, OK, "");
    return 0;
    });
}

void* gwb_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (!check_device().empty() || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void gwb_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int gwb_last_timing(gwb_graph_t* g, gwb_timing_t* t) {
    if (!g || !t) return 1;
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->timing_pending) {
        float interp = 0.f, pack = 0.f;
        for (size_t k = g->pending.size() - g->last_call_launches; k < g->pending.size(); ++k) {
            auto& c = g->pending[k];
            float a = 0.f, b = 0.f;
            if (hipEventSynchronize(c.after_pack) != hipSuccess || hipEventElapsedTime(&a, c.start, c.after_interp) != hipSuccess ||
                hipEventElapsedTime(&b, c.after_interp, c.after_pack) != hipSuccess)
                return 1;
            interp += a;
            pack += b;
        }
        g->timing.interp_ms = interp;
        g->timing.pack_ms = pack;
        g->timing_pending = false;
    }
    *t = g->timing;
    return 0;
}

int gwb_program_stats(gwb_graph_t* g, uint32_t program_key, gwb_program_stats_t* out) {
    // statistics of the compiled program for `program_key` (0: the one the last batch call used): bundles and nodes per
    // class, the cost model's lone-wave cycles, and the mean share of a wave's 64 lanes that hold a node of the graph,
    // weighted by the modelled time of the bundles -- the number behind a low instruction-issue efficiency
    if (!g || !out) return 1;
    try {
        std::lock_guard<std::mutex> lk(g->mu);
        const Program* p = nullptr;
        if (program_key == 0) program_key = g->last_key;
        auto it = g->progs.find(program_key);
        if (it != g->progs.end()) p = &it->second->host;
        auto pre = g->compiled.find(program_key);
        if (!p && pre != g->compiled.end()) p = pre->second.get();
        if (!p) return 1;
        memset(out, 0, sizeof *out);
        out->tile_width = p->T;
        out->divider = p->divider;
        out->streams = p->n_streams;
        out->n_bundles = p->n_bundles;
        out->n_classes = C_COUNT;
        double wsum = 0, lsum = 0, vsum = 0;
        for (uint32_t c = 0; c < C_COUNT && c < 16; ++c) {
            out->class_bundles[c] = p->stats.class_bundles[c];
            out->class_nodes[c] = p->stats.class_nodes[c];
            const double cyc = model_class_cycles((int)c) * (double)p->stats.class_bundles[c];
            const double lanes = (double)p->stats.class_nodes[c] * p->T * ((c == C_MULQ || c == C_MULF) ? (double)COOP_LANES : 1.0);
            wsum += cyc;
            lsum += p->stats.class_bundles[c] ? cyc * lanes / (double)p->stats.class_bundles[c] : 0.0;
            vsum += p->stats.class_bundles[c] ? cyc * (double)p->stats.class_nodes[c] * p->T / (double)p->stats.class_bundles[c] : 0.0;
        }
        out->model_wave_cycles = program_wave_cycles(*p);
        out->lanes_active_mean = wsum > 0 ? lsum / wsum : 0.0;
        out->values_per_bundle_mean = wsum > 0 ? vsum / wsum : 0.0;
        out->n_fused_nodes = p->stats.n_fused_nodes;
        out->chain_floor_cycles = (double)p->stats.chain_floor_cycles;
        out->n_scan_steps = p->stats.n_scan_steps;
        return 0;
    } catch (...) {
        return 1;
    }
}

int gwb_timing_history(gwb_graph_t* g, size_t max_launches, float* interp_ms, float* pack_ms, size_t* n_out) {
    if (!g || !n_out || (max_launches && (!interp_ms || !pack_ms))) return 1;
    std::lock_guard<std::mutex> lk(g->mu);
    const size_t n = g->pending.size() < max_launches ? g->pending.size() : max_launches;
    for (size_t i = 0; i < n; ++i) {  // chronological, ending with the most recent launch
        auto& c = g->pending[g->pending.size() - n + i];
        if (hipEventSynchronize(c.after_pack) != hipSuccess || hipEventElapsedTime(&interp_ms[i], c.start, c.after_interp) != hipSuccess ||
            hipEventElapsedTime(&pack_ms[i], c.after_interp, c.after_pack) != hipSuccess)
            return 1;
    }
    *n_out = n;
    return 0;
}

int gwb_profile_classes(gwb_graph_t* g, const void* d_inputs, size_t batch, void* d_witness, uint32_t* d_set_status,
                        uint64_t* out36, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    // Diagnostic: one batch through the stamped interpreter build; out36[class*4 + {load, compute, store, count}]
    // in shader cycles, summed over the sampled waves (lane 0 of every 64th tile).
    if (!g || !out36) return fail(status, "null argument");
    if (!gwb_kernels_have_diagnostics())
        return fail(status, "class profiling needs the diagnostic library (make -C circom-witnesscalc_amd/csrc diag; load it with CWC_LIB_PATH=<path of "
                            "libcircom_witnesscalc_amd_diag.so>): the product library carries no stamped interpreter instances");
    std::lock_guard<std::mutex> lk(g->mu);
    std::string err = check_device();
    if (!err.empty()) return fail(status, err);
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 96 * 8) != hipSuccess || hipMemset(d, 0, 96 * 8) != hipSuccess) return fail(status, "hipMalloc failed");
    g->d_prof = d;
    err = run_device(g, d_inputs, batch, d_witness, d_set_status, nullptr);
    g->d_prof = nullptr;
    if (err.empty() && hipDeviceSynchronize() != hipSuccess) err = "hipDeviceSynchronize failed";
    if (err.empty() && hipMemcpy(out36, d, 96 * 8, hipMemcpyDeviceToHost) != hipSuccess) err = "hipMemcpy failed";
    (void)hipFree(d);
    if (!err.empty()) return fail(status, err);
    set_status(status, OK, "");
    return 0;
    });
}

size_t gwb_wtns_size(size_t n_witness) { return wtns_size(n_witness); }

int gwb_wtns_from_witness(const void* row, size_t n_witness, void* out) {
    if ((!row && n_witness) || !out) return 1;
    wtns_from_witness((const uint8_t*)row, n_witness, (uint8_t*)out);
    return 0;
}

int gwb_graph_export(gwb_graph_t* g, uint32_t T, void** blob, size_t* blob_len, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!g || !blob || !blob_len) return fail(status, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    Program tmp;
    const Program* p = nullptr;
    if ((T & ~KEY_MODE_MASK) == 64) T = 64;
    auto it = g->progs.find(T);
    std::string err;
    auto pre = g->compiled.find(T);
    if (it != g->progs.end()) {
        p = &it->second->host;
    } else if (pre != g->compiled.end()) {  // compiled for the cost model, not uploaded yet
        p = pre->second.get();
    } else {
        if (!g->has_graph) return fail(status, "imported handle has no program for that tile width");
        if (!compile_program(g->graph, T & ~KEY_MODE_MASK, key_divider_waves(T), tmp, err, key_streams(T))) return fail(status, err);
        p = &tmp;
    }
    const size_t n = exported_size(*p, g->inputs);
    *blob = malloc(n);
    if (!*blob) return fail(status, "out of memory");
    exported_write(*p, g->inputs, (uint8_t*)*blob);
    *blob_len = n;
    set_status(status, OK, "");
    return 0;
    });
}

int gwb_graph_import(const void* blob, size_t len, gwb_graph_t** out, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!blob || !out) return fail(status, "null argument");
    if (len < 24 + 8) return fail(status, "bad blob: too short");
    const uint8_t* b = (const uint8_t*)blob;
    uint64_t tr[3];  // exact program length, padded program length, checksum of everything before the trailer
    memcpy(tr, b + len - 24, 24);
    const size_t body = len - 24;
    if (tr[2] != blob_checksum(b, body)) return fail(status, "bad blob: checksum mismatch (truncated or corrupted)");
    if (tr[0] > tr[1] || tr[1] - tr[0] >= 8 || tr[1] > body || (tr[1] % 8) != 0) return fail(status, "bad blob trailer");
    std::unique_ptr<gwb_graph> g(new gwb_graph());
    std::unique_ptr<DeviceProgram> dp(new DeviceProgram());
    std::string err;
    if (!program_from_blob(b, (size_t)tr[0], dp->host, err)) return fail(status, "bad program blob: " + err);
    if (!validate_program(dp->host, err)) return fail(status, "bad program blob: " + err);
    size_t pos = (size_t)tr[1];
    auto get32 = [&](uint32_t& v) {
        if (pos + 4 > body) return false;
        memcpy(&v, b + pos, 4);
        pos += 4;
        return true;
    };
    uint32_t n;
    if (!get32(n)) return fail(status, "bad blob trailer");
    for (uint32_t i = 0; i < n; ++i) {
        InputSignal s;
        uint32_t nl;
        if (!get32(s.offset) || !get32(s.len) || !get32(nl) || nl > body - pos) return fail(status, "bad blob trailer");
        if ((uint64_t)s.offset + s.len > dp->host.n_inputs) return fail(status, "bad blob: input signal beyond the inputs buffer");
        s.name.assign((const char*)b + pos, nl);
        pos += nl;
        g->input_index[s.name] = (uint32_t)g->inputs.size();
        g->inputs.push_back(s);
    }
    g->stats = dp->host.stats;
    g->n_inputs = dp->host.n_inputs;
    g->n_witness = dp->host.n_witness;
    err = check_device();
    if (err.empty()) err = upload_program(*dp);
    if (!err.empty()) return fail(status, err);
    const uint32_t T = dp->host.T | key_mode_of_divider(dp->host.divider);
    g->progs[T] = std::move(dp);
    *out = g.release();
    set_status(status, OK, "");
    return 0;
    });
}

// One collective in the whole path: the compiled program of rank `root` goes to every GPU of the communicator over RCCL
// (xGMI inside a node).  RCCL's entry points are resolved in the running process (the host program that owns the
// communicator has RCCL loaded; this library does not link it).
int gwb_graph_broadcast(gwb_graph_t* g, uint32_t tile_width, size_t batch_per_rank, int root, int rank, void* nccl_comm, void* hip_stream,
                        gwb_graph_t** out, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!out || !nccl_comm) return fail(status, "null argument");
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_fn)(int);
    bcast_fn bcast = (bcast_fn)dlsym(RTLD_DEFAULT, "ncclBroadcast");
    if (!bcast) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) bcast = (bcast_fn)dlsym(h, "ncclBroadcast");
    }
    if (!bcast) return fail(status, "ncclBroadcast not found: RCCL is not loaded in this process");
    errstr_fn errstr = (errstr_fn)dlsym(RTLD_DEFAULT, "ncclGetErrorString");
    std::string err = check_device();
    if (!err.empty()) return fail(status, err);
    hipStream_t stream = (hipStream_t)hip_stream;
    void* blob = nullptr;
    size_t blob_len = 0;
    // A failure on the root before the first collective must not leave the other ranks blocked in the length broadcast:
    // the root then still broadcasts a length of 0, which every receiver rejects, and all ranks return an error together.
    std::string root_err;
    if (rank == root && !g) root_err = "the root rank needs a loaded graph";
    if (rank == root && g) {
        if (!tile_width) {
            if (!batch_per_rank) root_err = "tile_width = 0 needs the shard size";
            else if (!(tile_width = gwb_graph_pick_tile_width(g, batch_per_rank))) root_err = "no program for that batch size";
        }
        if (root_err.empty()) {
            gw_status_t st2{OK, nullptr};
            if (gwb_graph_export(g, tile_width, &blob, &blob_len, &st2) != 0) {
                root_err = st2.error_msg ? st2.error_msg : "export failed";
                gwb_free_status(&st2);
                free(blob);
                blob = nullptr;
                blob_len = 0;
            }
        }
    }
    struct Bufs {
        void* d_len = nullptr;
        void* d_blob = nullptr;
        void* h_blob = nullptr;
        ~Bufs() {
            if (d_len) (void)hipFree(d_len);
            if (d_blob) (void)hipFree(d_blob);
            free(h_blob);
        }
    } bufs;
    bufs.h_blob = blob;
    auto nccl_fail = [&](int rc, const char* what) { return fail(status, std::string(what) + ": " + (errstr ? errstr(rc) : "RCCL error " + std::to_string(rc))); };
    const int ncclUint8 = 1, ncclUint64 = 5;
    // the root uploads its blob BEFORE the length goes out: an allocation or copy that fails there becomes length 0 too
    if (rank == root && root_err.empty() &&
        (hipMalloc(&bufs.d_blob, blob_len) != hipSuccess || hipMemcpy(bufs.d_blob, blob, blob_len, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipGetLastError();
        root_err = "hipMalloc / hipMemcpy of the program failed on the root rank";
    }
    unsigned long long len64 = root_err.empty() ? blob_len : 0;
    if (hipMalloc(&bufs.d_len, 8) != hipSuccess || hipMemcpy(bufs.d_len, &len64, 8, hipMemcpyHostToDevice) != hipSuccess) return fail(status, "hipMalloc failed");
    int rc = bcast(bufs.d_len, bufs.d_len, 1, ncclUint64, root, nccl_comm, stream);
    if (rc != 0) return nccl_fail(rc, "ncclBroadcast (length)");
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(&len64, bufs.d_len, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(status, "hipMemcpy failed");
    if (!root_err.empty()) return fail(status, root_err);
    if (len64 == 0 || len64 > (1ull << 40)) return fail(status, len64 == 0 ? "the root rank failed before the broadcast (program length 0)" : "bad program length in broadcast");
    if (rank != root && hipMalloc(&bufs.d_blob, (size_t)len64) != hipSuccess) return fail(status, "hipMalloc failed");
    rc = bcast(bufs.d_blob, bufs.d_blob, (size_t)len64, ncclUint8, root, nccl_comm, stream);
    if (rc != 0) return nccl_fail(rc, "ncclBroadcast (program)");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(status, "hipStreamSynchronize failed");
    if (rank == root) {
        *out = g;  // the root keeps its handle (its own program for that key is compiled on first use, or already is)
        set_status(status, OK, "");
        return 0;
    }
    bufs.h_blob = malloc((size_t)len64);
    if (!bufs.h_blob) return fail(status, "out of memory");
    if (hipMemcpy(bufs.h_blob, bufs.d_blob, (size_t)len64, hipMemcpyDeviceToHost) != hipSuccess) return fail(status, "hipMemcpy failed");
    return gwb_graph_import(bufs.h_blob, (size_t)len64, out, status);
    });
}

// ---- the reference's symbol (src/lib.rs:44-111) ----------------------------------------------------
int gw_calc_witness(const char* inputs, const void* graph_data, const size_t graph_data_len, void** wtns_data,
                    size_t* wtns_len, const gw_status_t* status_c) {
    return guarded(const_cast<gw_status_t*>(status_c), [&]() -> int {
    gw_status_t* status = const_cast<gw_status_t*>(status_c);  // the reference writes through it too
    if (!inputs) return fail(status, "inputs is null");                    // lib.rs:51-54
    if (!graph_data) return fail(status, "graph_data is null");            // lib.rs:56-59
    if (graph_data_len == 0) return fail(status, "graph_data_len is 0");   // lib.rs:61-64
    if (!wtns_data || !wtns_len) return fail(status, "wtns_data or wtns_len is null");
    // CStr::to_str UTF-8 check (lib.rs:72-84)
    {
        const unsigned char* s = (const unsigned char*)inputs;
        size_t i = 0, n = strlen(inputs);
        while (i < n) {
            unsigned char c = s[i];
            size_t k = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
            bool ok = k != 0 && i + k <= n;
            for (size_t q = 1; ok && q < k; ++q) ok = (s[i + q] & 0xC0) == 0x80;
            if (ok && k == 2) ok = c >= 0xC2;
            if (ok && k == 3) ok = !(c == 0xE0 && s[i + 1] < 0xA0) && !(c == 0xED && s[i + 1] >= 0xA0);
            if (ok && k == 4) ok = !(c == 0xF0 && s[i + 1] < 0x90) && !(c > 0xF4) && !(c == 0xF4 && s[i + 1] >= 0x90);
            if (!ok) return fail(status, "Failed to parse inputs: invalid utf-8 sequence at byte " + std::to_string(i));
            i += k;
        }
    }
    // calc_witness (lib.rs:125-136): inputs first, then the graph
    InputList list;
    std::string err;
    const bool dbg_single = getenv("CWC_DEBUG_SINGLE") != nullptr;  // diagnostic: where a call's time goes
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now_ms();
    warm_device();
    if (!deserialize_inputs(inputs, strlen(inputs), list, err)) return fail(status, "Failed to calculate witness: " + err);
    const double t_inputs = now_ms();

    std::shared_ptr<gwb_graph> g;
    // (a sampled fingerprint picks the candidate, the byte compare below decides: hashing the whole 3 MB image on every
    // call was 2-3 ms of the single call's 12)
    const uint64_t h = sampled_fingerprint((const uint8_t*)graph_data, graph_data_len);
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto& e : g_cache)
            if (e.hash == h && e.bytes.size() == graph_data_len && memcmp(e.bytes.data(), graph_data, graph_data_len) == 0) g = e.g;
    }
    if (!g) {
        const std::string cf = program_cache_file(graph_data, graph_data_len);  // (SHA-256 of the image: once per graph and process)
        std::vector<uint8_t> blob;
        if (!cf.empty() && read_file(cf, blob)) {  // a program an earlier process compiled for this very image (the import checks for a device behind its host work)
            gwb_graph_t* imported = nullptr;
            gw_status_t st2{OK, nullptr};
            const uint8_t* body = nullptr;
            size_t body_len = 0;
            if (!cache_unwrap(cf, blob, &body, &body_len)) set_status(&st2, ERROR, "not this graph's / this build's cache entry");
            else if (gwb_graph_import(body, body_len, &imported, &st2) == 0) g.reset(imported);
            if (getenv("CWC_DEBUG_CACHE")) fprintf(stderr, "program cache: %s %s%s%s\n", g ? "hit" : "ignored", cf.c_str(), g ? "" : ": ", g ? "" : (st2.error_msg ? st2.error_msg : "?"));
            gwb_free_status(&st2);  // (a damaged / stale file: fall through to the compiler, the file is rewritten)
        }
        if (!g) {
            gwb_graph* raw = nullptr;
            if (load_graph(graph_data, graph_data_len, &raw, err)) return fail(status, "Failed to calculate witness: " + err);
            g.reset(raw);
            g->cache_path = cf;  // where the refined program goes once the background search has finished
        }
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_cache.size() >= 4) g_cache.erase(g_cache.begin());
        g_cache.push_back(CacheEntry{h, std::vector<uint8_t>((const uint8_t*)graph_data, (const uint8_t*)graph_data + graph_data_len), g});
    }
    std::vector<uint8_t> row((size_t)g->n_inputs * 32), wit((size_t)g->n_witness * 32);
    {
        Graph meta;  // populate_inputs only needs the input map (a handle imported from the cache holds no graph)
        meta.inputs = g->inputs;
        meta.input_index = g->input_index;
        if (!populate_inputs(list, meta, row.data(), g->n_inputs, err)) return fail(status, "Failed to calculate witness: " + err);
    }
    if (quirks())
        for (const auto& kv : list) {
            const InputSignal& s = g->inputs[g->input_index.at(kv.first)];
            printf("input %s, offset %u, len %u\n", kv.first.c_str(), s.offset, s.len);
        }
    uint32_t st = 0;
    const double t_graph = now_ms();
    double t_device = t_graph;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        (void)pick_tile_width(g.get(), 1);  // host work first: a new graph's program is compiled while warm_device's thread brings the device up
        err = check_device();
        t_device = now_ms();
        if (err.empty()) err = run_host(g.get(), row.data(), 1, wit.data(), &st);
    }
    if (!err.empty()) return fail(status, "Failed to calculate witness: " + err);
    if (dbg_single) {
        gwb_timing_t tm;
        if (gwb_last_timing(g.get(), &tm) == 0)
            fprintf(stderr, "gw_calc_witness: program key %#x, %llu bundles, interpreter %.2f ms, pack %.2f ms | inputs %.1f ms, graph handle + input row %.1f ms, program choice + device check %.1f ms, upload + run %.1f ms\n",
                    g->last_key, (unsigned long long)tm.n_bundles, tm.interp_ms, tm.pack_ms, t_inputs - t_in, t_graph - t_inputs, t_device - t_graph, now_ms() - t_device);
    }
    // The program for the on-disk cache: the one the background search settled on (the quick first program is not worth
    // keeping).  Written by whichever call first finds the search finished.
    if (g->has_graph && !g->cache_written && !g->cache_path.empty()) {
        const std::string& cf = g->cache_path;
        uint32_t key = 0;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            auto it = g->chosen.find(1);
            if (it != g->chosen.end() && !g->refining.count(1)) key = it->second;
        }
        if (!cf.empty() && key) {
            void* blob = nullptr;
            size_t blob_len = 0;
            gw_status_t st2{OK, nullptr};
            if (gwb_graph_export(g.get(), key, &blob, &blob_len, &st2) == 0) {
                const std::vector<uint8_t> wrapped = cache_wrap(cf, blob, blob_len);
                write_file_atomically(cf, wrapped.data(), wrapped.size());
                if (getenv("CWC_DEBUG_CACHE")) fprintf(stderr, "program cache: wrote %s (program key %#x, %zu bytes)\n", cf.c_str(), key, blob_len);
            }
            gwb_free_status(&st2);
            free(blob);
            g->cache_written = true;
        }
    }
    if (st) return fail(status, "Failed to calculate witness: " + set_status_text(st));
    const size_t n = wtns_size(g->n_witness);
    void* buf = malloc(n);
    if (!buf) return fail(status, "Failed to allocate memory for wtns_data");  // lib.rs:99-102
    wtns_from_witness(wit.data(), g->n_witness, (uint8_t*)buf);
    *wtns_len = n;
    *wtns_data = buf;
    if (quirks()) {
        set_status(status, ERROR, "test error");  // lib.rs:106
        printf("OK\n");                            // lib.rs:108
    } else {
        set_status(status, OK, "");
    }
    return 0;
    });
}

}  // extern "C"
