// Host graph compiler: validation, level scheduling into same-class bundles, liveness-based slot
// allocation, program encoding.  See program.hpp.
#include <string.h>

#include <algorithm>

#include "program.hpp"

namespace cwc {

static int class_of(const Node& n) {
    switch (n.kind) {
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
                case OP_SHL: case OP_SHR: case OP_BOR: case OP_BAND: case OP_BXOR: return C_BIT;
                case OP_IDIV: case OP_MOD: return C_IDIVMOD;
            }
    }
    return -1;
}

static int arity_of(const Node& n) { return n.kind == N_UNO ? 1 : n.kind == N_DUO ? 2 : n.kind == N_TRES ? 3 : 0; }

bool compile_program(const Graph& g, uint32_t T, Program& out, std::string& err) {
    if (T == 0 || T > 64 || (T & (T - 1))) {
        err = "tile width must be a power of two in 1..64";
        return false;
    }
    const size_t N = g.nodes.size();
    const uint32_t G = 64 / T;
    out = Program();
    out.T = T;
    out.G = G;
    ProgramStats& st = out.stats;
    st.n_nodes = N;
    st.n_witness = g.witness_signals.size();

    // ---- validate (assert_valid, reference src/graph.rs:343-356; evaluate() itself does not check) ----
    const size_t n_in_buf = inputs_buffer_size(g);
    uint64_t arity_sum = 0;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        int ar = arity_of(n);
        if ((ar >= 1 && n.a >= i) || (ar >= 2 && n.b >= i) || (ar >= 3 && n.c >= i)) {
            err = "node " + std::to_string(i) + " references a node that is not before it";
            return false;
        }
        if (n.kind == N_DUO && n.op == OP_POW) {  // graph.rs:141-142 unimplemented!
            err = "node " + std::to_string(i) + ": operator Pow not implemented for Montgomery";
            return false;
        }
        if (n.kind == N_UNO && n.op != UOP_NEG) {  // graph.rs:195
            err = "node " + std::to_string(i) + ": uno operator Id not implemented for Montgomery";
            return false;
        }
        if (n.kind == N_CONST && n.a >= g.const_values.size()) {
            err = "node " + std::to_string(i) + ": bad constant index";
            return false;
        }
        if (ar) {
            st.n_op++;
            arity_sum += (uint64_t)ar + 1;
        } else if (n.kind == N_INPUT) {
            st.n_input_nodes++;
        }
    }
    for (uint32_t w : g.witness_signals)
        if (w >= N) {
            err = "witness signal references node " + std::to_string(w) + " beyond the graph";
            return false;
        }
    st.algorithmic_bytes_per_set = 32ull * (arity_sum + 2 * st.n_input_nodes + 2 * st.n_witness);

    // ---- constants -> table (Montgomery form), node -> ref ----
    std::vector<uint32_t> ref(N, 0);  // for consts: REF_CONST|idx ; for others: slot (filled later)
    for (size_t i = 0; i < N; ++i)
        if (g.nodes[i].kind == N_CONST) {
            Fr m = fr_to_mont(g.const_values[g.nodes[i].a]);
            ref[i] = REF_CONST | (uint32_t)(out.consts.size() / 8);
            out.consts.insert(out.consts.end(), m.v, m.v + 8);
        }
    out.n_const = (uint32_t)(out.consts.size() / 8);
    st.n_const = out.n_const;

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

    // ---- schedule: order of evaluated nodes (inputs + ops) and bundle boundaries ----
    std::vector<uint32_t> order;
    order.reserve(N);
    for (size_t i = 0; i < N; ++i)
        if (g.nodes[i].kind != N_CONST) order.push_back((uint32_t)i);
    if (G > 1) {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            if (level[x] != level[y]) return level[x] < level[y];
            return class_of(g.nodes[x]) < class_of(g.nodes[y]);
        });
    }
    std::vector<uint32_t> bundle_of(N, 0xffffffffu);
    std::vector<uint32_t> bundle_start;  // index into order
    for (size_t k = 0; k < order.size();) {
        size_t e = k + 1;
        if (G > 1) {
            const uint32_t lv = level[order[k]];
            const int cl = class_of(g.nodes[order[k]]);
            while (e < order.size() && e - k < G && level[order[e]] == lv && class_of(g.nodes[order[e]]) == cl) ++e;
        }
        for (size_t q = k; q < e; ++q) bundle_of[order[q]] = (uint32_t)bundle_start.size();
        bundle_start.push_back((uint32_t)k);
        k = e;
    }
    const uint32_t NB = (uint32_t)bundle_start.size();
    bundle_start.push_back((uint32_t)order.size());
    out.n_bundles = NB;

    // ---- liveness: last bundle that reads each node's value; witness nodes are pinned ----
    std::vector<uint32_t> last_use(N, 0);
    std::vector<uint8_t> pinned(N, 0);
    for (uint32_t i : order) last_use[i] = bundle_of[i];
    for (uint32_t i : order) {
        const Node& n = g.nodes[i];
        int ar = arity_of(n);
        uint32_t b = bundle_of[i];
        if (ar >= 1 && g.nodes[n.a].kind != N_CONST) last_use[n.a] = std::max(last_use[n.a], b);
        if (ar >= 2 && g.nodes[n.b].kind != N_CONST) last_use[n.b] = std::max(last_use[n.b], b);
        if (ar >= 3 && g.nodes[n.c].kind != N_CONST) last_use[n.c] = std::max(last_use[n.c], b);
    }
    for (uint32_t w : g.witness_signals) pinned[w] = 1;

    // ---- slot allocation (LIFO free list: a just-freed slot is still hot in cache) + encoding ----
    out.hdr.resize(NB);
    out.recs.assign((size_t)NB * G * 4, 0);
    out.crefs.assign((size_t)NB * G, 0);
    std::vector<uint32_t> free_slots;
    std::vector<uint32_t> dying;  // nodes whose slot is released after the current bundle
    uint32_t n_slots = 0;
    for (uint32_t b = 0; b < NB; ++b) {
        const uint32_t k0 = bundle_start[b], k1 = bundle_start[b + 1];
        const int cl = class_of(g.nodes[order[k0]]);
        out.hdr[b] = (uint32_t)cl | ((k1 - k0) << 8);
        st.class_bundles[cl]++;
        st.class_nodes[cl] += k1 - k0;
        dying.clear();
        for (uint32_t k = k0; k < k1; ++k) {
            const uint32_t i = order[k];
            const Node& n = g.nodes[i];
            uint32_t slot;
            if (!free_slots.empty()) {
                slot = free_slots.back();
                free_slots.pop_back();
            } else {
                slot = n_slots++;
            }
            ref[i] = slot;
            uint32_t* r = &out.recs[((size_t)b * G + (k - k0)) * 4];
            r[1] = slot;
            switch (n.kind) {
                case N_INPUT:
                    if (n.a >= n_in_buf) {
                        err = "Input index out of range";
                        return false;
                    }
                    r[0] = SUB_INPUT;
                    r[2] = n.a;
                    break;
                case N_UNO:
                    r[0] = SUB_NEG;
                    r[2] = ref[n.a];
                    r[3] = ref[n.a];  // kernel convention: Neg carries b = a
                    break;
                case N_DUO:
                    r[0] = n.op;
                    r[2] = ref[n.a];
                    r[3] = ref[n.b];
                    break;
                case N_TRES:
                    r[0] = SUB_TERN;
                    r[2] = ref[n.a];
                    r[3] = ref[n.b];
                    out.crefs[(size_t)b * G + (k - k0)] = ref[n.c];
                    break;
            }
            const uint32_t ops[3] = {n.a, n.b, n.c};
            for (int q = 0; q < arity_of(n); ++q) {
                uint32_t o = ops[q];
                if (g.nodes[o].kind != N_CONST && !pinned[o] && last_use[o] == b) dying.push_back(o);
            }
            if (!pinned[i] && last_use[i] == b) dying.push_back(i);  // dead value: release right away
        }
        for (uint32_t q = k1 - k0; q < G; ++q) {  // inactive node slots: valid operands, store masked off
            memcpy(&out.recs[((size_t)b * G + q) * 4], &out.recs[(size_t)b * G * 4], 16);
            out.crefs[(size_t)b * G + q] = out.crefs[(size_t)b * G];
        }
        std::sort(dying.begin(), dying.end());
        dying.erase(std::unique(dying.begin(), dying.end()), dying.end());
        for (uint32_t o : dying) free_slots.push_back(ref[o]);
    }
    out.n_slots = std::max(n_slots, 1u);
    out.n_inputs = (uint32_t)n_in_buf;
    out.n_witness = (uint32_t)g.witness_signals.size();
    out.witness_refs.resize(out.n_witness);
    for (size_t i = 0; i < g.witness_signals.size(); ++i) out.witness_refs[i] = ref[g.witness_signals[i]];
    return true;
}

// ---- blob ----------------------------------------------------------------------------------------
static const uint32_t kBlobMagic = 0x47505743u;  // "CWPG"
struct BlobHeader {
    uint32_t magic, version, T, G, n_bundles, n_slots, n_const, n_inputs, n_witness, reserved;
    ProgramStats stats;
};

std::vector<uint8_t> program_to_blob(const Program& p) {
    BlobHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.version = 1;
    h.T = p.T; h.G = p.G; h.n_bundles = p.n_bundles; h.n_slots = p.n_slots; h.n_const = p.n_const;
    h.n_inputs = p.n_inputs; h.n_witness = p.n_witness;
    h.stats = p.stats;
    std::vector<uint8_t> out((uint8_t*)&h, (uint8_t*)&h + sizeof h);
    auto put = [&](const std::vector<uint32_t>& v) { out.insert(out.end(), (const uint8_t*)v.data(), (const uint8_t*)(v.data() + v.size())); };
    put(p.hdr); put(p.recs); put(p.crefs); put(p.consts); put(p.witness_refs);
    return out;
}

bool program_from_blob(const uint8_t* data, size_t len, Program& p, std::string& err) {
    BlobHeader h;
    if (len < sizeof h) { err = "program blob too short"; return false; }
    memcpy(&h, data, sizeof h);
    if (h.magic != kBlobMagic || h.version != 1 || h.T == 0 || h.T > 64 || h.G != 64 / h.T) { err = "bad program blob header"; return false; }
    p = Program();
    p.T = h.T; p.G = h.G; p.n_bundles = h.n_bundles; p.n_slots = h.n_slots; p.n_const = h.n_const;
    p.n_inputs = h.n_inputs; p.n_witness = h.n_witness; p.stats = h.stats;
    const size_t n_hdr = p.n_bundles, n_recs = (size_t)p.n_bundles * p.G * 4, n_c = (size_t)p.n_bundles * p.G,
                 n_k = (size_t)p.n_const * 8, n_w = p.n_witness;
    if (len != sizeof h + 4 * (n_hdr + n_recs + n_c + n_k + n_w)) { err = "program blob size mismatch"; return false; }
    const uint32_t* q = (const uint32_t*)(data + sizeof h);
    p.hdr.assign(q, q + n_hdr); q += n_hdr;
    p.recs.assign(q, q + n_recs); q += n_recs;
    p.crefs.assign(q, q + n_c); q += n_c;
    p.consts.assign(q, q + n_k); q += n_k;
    p.witness_refs.assign(q, q + n_w);
    return true;
}

}  // namespace cwc
