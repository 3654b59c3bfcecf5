// Program images: what gwb_graph_export hands out, gwb_graph_import takes back and gwb_graph_broadcast moves between the GPUs
// of a node over RCCL (one broadcast of the compiled program, no data-path collective), with their checksums.
#include "runtime_internal.hpp"

namespace cwcrt {

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

}  // namespace cwcrt

extern "C" {

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

// RCCL's entry points are resolved in the running process (the host program has RCCL loaded, e.g. torch's copy; this library
// does not link it), else from the system's librccl.
static void* rccl_symbol(const char* name) {
    void* f = dlsym(RTLD_DEFAULT, name);
    if (!f) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) f = dlsym(h, name);
    }
    return f;
}

// A communicator for gwb_graph_broadcast made through the library itself: a host in a language without a RCCL binding (or
// with one that cannot pass ncclUniqueId by value) draws the 128-byte id on one rank, carries the bytes to the others by
// whatever channel its job has, and every rank joins.  (ncclUniqueId is a 128-byte struct passed BY VALUE to
// ncclCommInitRank: done here in C, where the calling convention is the compiler's business.)
struct GwbUniqueId {
    char internal[GWB_RCCL_UNIQUE_ID_BYTES];
};
int gwb_rccl_unique_id(void* id128, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!id128) return fail(status, "null argument");
    typedef int (*fn_t)(GwbUniqueId*);
    fn_t f = (fn_t)rccl_symbol("ncclGetUniqueId");
    if (!f) return fail(status, "ncclGetUniqueId not found: RCCL is not loaded in this process");
    GwbUniqueId id;
    memset(&id, 0, sizeof id);
    const int rc = f(&id);
    if (rc != 0) return fail(status, "ncclGetUniqueId failed: RCCL error " + std::to_string(rc));
    memcpy(id128, &id, sizeof id);
    set_status(status, OK, "");
    return 0;
    });
}
int gwb_rccl_comm_init(const void* id128, int n_ranks, int rank, void** comm, gw_status_t* status) {
    return guarded(status, [&]() -> int {
    if (!id128 || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(status, "bad argument");
    typedef int (*fn_t)(void**, int, GwbUniqueId, int);
    typedef const char* (*errstr_fn)(int);
    fn_t f = (fn_t)rccl_symbol("ncclCommInitRank");
    if (!f) return fail(status, "ncclCommInitRank not found: RCCL is not loaded in this process");
    std::string err = check_device();
    if (!err.empty()) return fail(status, err);
    GwbUniqueId id;
    memcpy(&id, id128, sizeof id);
    *comm = nullptr;
    const int rc = f(comm, n_ranks, id, rank);
    if (rc != 0) {
        errstr_fn errstr = (errstr_fn)rccl_symbol("ncclGetErrorString");
        return fail(status, std::string("ncclCommInitRank failed: ") + (errstr ? errstr(rc) : "RCCL error"));
    }
    set_status(status, OK, "");
    return 0;
    });
}
int gwb_rccl_comm_ranks(void* comm) {
    typedef int (*fn_t)(void*, int*);
    fn_t f = comm ? (fn_t)rccl_symbol("ncclCommCount") : nullptr;
    int n = 0;
    return f && f(comm, &n) == 0 ? n : -1;
}
void gwb_rccl_comm_destroy(void* comm) {
    typedef int (*fn_t)(void*);
    fn_t f = comm ? (fn_t)rccl_symbol("ncclCommDestroy") : nullptr;
    if (f) (void)f(comm);
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
    bcast_fn bcast = (bcast_fn)rccl_symbol("ncclBroadcast");
    if (!bcast) return fail(status, "ncclBroadcast not found: RCCL is not loaded in this process");
    errstr_fn errstr = (errstr_fn)rccl_symbol("ncclGetErrorString");
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

}  // extern "C"
