// Internal header of the C-ABI runtime's translation units (pipeline.cc, capi_batch.cc, capi_single.cc, bcast.cc): the graph
// handle, an uploaded program, and the functions the units share.  Nothing here is part of the library's interface
// (include/graph_witness.h, include/graph_witness_batch.h are).
#pragma once
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

namespace cwcrt {

// prepare_status, reference src/lib.rs:28-38
void set_status(gw_status_t* st, GW_ERROR_CODE code, const std::string& msg);
int fail(gw_status_t* st, const std::string& msg);
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

}  // namespace cwcrt
using namespace cwcrt;

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

namespace cwcrt {
// ---- pipeline.cc: device check, program choice (cost model), upload, the launches of a batch ----
std::string upload_program(DeviceProgram& dp);
uint32_t waves_per_workgroup(uint32_t divider, uint64_t tiles, uint32_t streams = 1);
uint64_t workspace_budget();
std::string check_device();
double estimate_cycles(const Program& p, size_t batch);
uint32_t pick_tile_width(gwb_graph* g, size_t batch, bool allow_quick = true);
std::string get_program(gwb_graph* g, uint32_t key, DeviceProgram** out);
std::string run_device(gwb_graph* g, const void* d_inputs, size_t batch, void* d_witness, uint32_t* d_status, hipStream_t stream, bool montgomery = false,
                       hipEvent_t done_event = nullptr);
std::string run_host(gwb_graph* g, const void* inputs, size_t batch, void* witness, uint32_t* set_status);
unsigned env_threads(const char* name, unsigned cap);
std::string set_status_text(uint32_t bits);
void warm_device();
// ---- bcast.cc: program images (export / import / broadcast) ----
uint64_t fnv1a(const uint8_t* p, size_t n);
uint64_t blob_checksum(const uint8_t* p, size_t n);
uint64_t sampled_fingerprint(const uint8_t* p, size_t n);
size_t exported_size(const Program& p, const std::vector<InputSignal>& inputs);
void exported_write(const Program& p, const std::vector<InputSignal>& inputs, uint8_t* dst);
std::vector<uint8_t> exported_bytes(const Program& p, const std::vector<InputSignal>& inputs);
// ---- capi_single.cc: the on-disk program cache of the single-shot entry point ----
void write_file_atomically(const std::string& path, const void* data, size_t n);
std::vector<uint8_t> cache_wrap(const std::string& path, const void* blob, size_t n);
bool quirks();  // GW_REFERENCE_QUIRKS: the reference's prints and its status quirk (lib.rs:106-108)
// ---- capi_batch.cc ----
int load_graph(const void* data, size_t len, gwb_graph** out, std::string& err);
}  // namespace cwcrt
extern "C" int gwb_kernels_have_diagnostics();  // kernels.hip
