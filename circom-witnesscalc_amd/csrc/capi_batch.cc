// The additive batch API (include/graph_witness_batch.h): handles, inputs front-end, batch entry points (device / host /
// hand-off / streaming JSON -> .wtns), timing and statistics.
#include "runtime_internal.hpp"

namespace cwcrt {

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

}  // namespace cwcrt

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

int gwb_graph_op_histogram(const gwb_graph_t* g, uint64_t* out, size_t n) {
    if (!g || !out || n < 24 || !g->has_graph) return 1;
    memset(out, 0, n * sizeof *out);
    for (const Node& nd : g->graph.nodes) {
        if (nd.kind == N_DUO && nd.op < 20) out[nd.op]++;
        else if (nd.kind == N_UNO) out[20]++;
        else if (nd.kind == N_TRES) out[21]++;
        else if (nd.kind == N_INPUT) out[22]++;
        else if (nd.kind == N_CONST) out[23]++;
    }
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
        // (the status words decide which sets get a file: a failed copy must not read as "every set is fine")
        if (hipMemcpy(st_host.data() + lo, bf.d_st[par], n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            std::lock_guard<std::mutex> l2(err_mu);
            if (drain_err.empty()) drain_err = "hipMemcpy of the status words failed";
            return;
        }
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
        out->n_conv_products = p->stats.n_conv_products;
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

}  // extern "C"
