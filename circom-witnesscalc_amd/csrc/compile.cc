// Host graph compiler: validation, exact rewrites (power-of-two divisions, bit-extract fusion, tree-height reduction with
// shared subexpressions and dead-node elimination), critical-path list scheduling into same-class bundles (linear riders,
// request / collect divisions for the divider waves), operand routing (LDS ring vs. staged memory), liveness-based slot
// allocation, program encoding (format v4) and the pointer-free program blob.  See program.hpp / program_dev.h.
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

static int class_of(const Node& n) {
    switch (n.kind) {
        case N_FUSED: return C_MULF;
        case N_SCAN: return C_SCAN;
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
static int arity_of(const Node& n) {
    if (n.kind == N_FUSED) return fused_sq(n.op) ? 1 + (fused_op2(n.op) ? 1 : 0) + (fused_op3(n.op) ? 1 : 0) : 3;
    if (n.kind == N_SCAN) return (n.op & SCAN_OP_DIV) ? 3 : 2;  // x, the accumulator coming in, the divisor
    return n.kind == N_UNO ? 1 : n.kind == N_DUO ? 2 : n.kind == N_TRES ? 3 : 0;
}

// Exact strength reduction done before scheduling: Idiv(x, 2^k) == Shr(x, k) and Mod(x, 2^k) == Band(x, 2^k - 1) on the
// canonical integers the reference divides (src/graph.rs:112-121 vs :637-672, :674-687), for every x < r and k <= 253.
// The replacement constants are appended behind the last node (constants have no dependencies).
static void rewrite_pow2_divisions(Graph& g) {
    std::unordered_map<uint32_t, uint32_t> shift_const, mask_const;  // k -> node index
    const size_t N = g.nodes.size();
    for (size_t i = 0; i < N; ++i) {
        Node& n = g.nodes[i];
        if (n.kind != N_DUO || (n.op != OP_IDIV && n.op != OP_MOD)) continue;
        const Node& d = g.nodes[n.b];
        if (d.kind != N_CONST) continue;
        const Fr& v = g.const_values[d.a];
        int k = -1, bits = 0;
        for (int w = 0; w < 8; ++w)
            if (v.v[w]) {
                bits += __builtin_popcount(v.v[w]);
                k = 32 * w + __builtin_ctz(v.v[w]);
            }
        if (bits != 1 || k > 253) continue;
        auto& table = n.op == OP_IDIV ? shift_const : mask_const;
        auto it = table.find((uint32_t)k);
        if (it == table.end()) {
            Fr c = fr_zero();
            if (n.op == OP_IDIV) {
                c.v[0] = (uint32_t)k;
            } else {
                for (int w = 0; w < 8; ++w) c.v[w] = k >= 32 * (w + 1) ? 0xffffffffu : (k > 32 * w ? ((1u << (k - 32 * w)) - 1u) : 0u);
            }
            const uint32_t idx = (uint32_t)g.nodes.size();
            g.nodes.push_back(Node{N_CONST, 0, (uint32_t)g.const_values.size(), 0, 0});
            g.const_values.push_back(c);
            it = table.emplace((uint32_t)k, idx).first;
        }
        Node& n2 = g.nodes[i];  // (push_back may have moved the vector)
        n2.op = n2.op == OP_IDIV ? OP_SHR : OP_BAND;
        n2.b = it->second;
    }
}

// Exact fusion of the bit-decomposition idiom (circomlib Num2Bits: out[i] <-- (in >> i) & 1): Band(Shr(a, k), 1) with a
// constant k < 254 whose Shr has no other user becomes one BITX node.  The pair costs two BIT bundles with four
// conversions out of and one into Montgomery form (graph.rs:637-672 then :674-687); the fused node converts once and
// its result is a boolean.  Shr cannot fail, so dropping the intermediate node loses no error.
static void fuse_bit_extract(Graph& g) {
    const size_t N = g.nodes.size();
    std::vector<uint32_t> uses(N, 0);
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) uses[n.a]++;
        if (ar >= 2) uses[n.b]++;
        if (ar >= 3) uses[n.c]++;
    }
    for (uint32_t w : g.witness_signals) uses[w]++;
    auto small_const = [&](uint32_t idx, uint32_t& value) {
        const Node& c = g.nodes[idx];
        if (c.kind != N_CONST) return false;
        const Fr& v = g.const_values[c.a];
        for (int q = 1; q < 8; ++q)
            if (v.v[q]) return false;
        value = v.v[0];
        return true;
    };
    std::vector<uint8_t> dead(N, 0);
    bool any = false;
    for (size_t i = 0; i < N; ++i) {
        Node& n = g.nodes[i];
        if (n.kind != N_DUO || n.op != OP_BAND) continue;
        for (int side = 0; side < 2; ++side) {
            const uint32_t s = side ? n.b : n.a, c = side ? n.a : n.b;
            uint32_t one = 0, k = 0;
            if (!small_const(c, one) || one != 1u) continue;
            const Node& sh = g.nodes[s];
            if (sh.kind != N_DUO || sh.op != OP_SHR || uses[s] != 1 || !small_const(sh.b, k) || k >= 254u) continue;
            n = Node{N_DUO, OP_BITX, sh.a, sh.b, 0};
            dead[s] = 1;
            any = true;
            break;
        }
    }
    if (!any) return;
    std::vector<uint32_t> pos(N, 0xffffffffu);
    std::vector<Node> kept;
    kept.reserve(N);
    // (constants appended by rewrite_pow2_divisions sit behind their users: number the survivors first)
    uint32_t next = 0;
    for (size_t i = 0; i < N; ++i)
        if (!dead[i]) pos[i] = next++;
    for (size_t i = 0; i < N; ++i) {
        if (dead[i]) continue;
        Node n = g.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) n.a = pos[n.a];
        if (ar >= 2) n.b = pos[n.b];
        if (ar >= 3) n.c = pos[n.c];
        kept.push_back(n);
    }
    for (uint32_t& w : g.witness_signals) w = pos[w];
    g.nodes.swap(kept);
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
static uint32_t div_cost50();
static inline uint64_t cost_of(const uint32_t* table, int c) { return c == (int)C_DIV ? div_cost50() : table[c]; }
// a fused node costs its bundle's front end and its stages (cycles / 50: 600 + 704 per product + ~280 per addition)
static inline uint64_t fused_cost50(uint8_t op) {
    return 12u + 15u + (fused_op2(op) == FOP_MUL ? 15u : fused_op2(op) ? 6u : 0u) + (fused_op3(op) ? 6u : 0u);
}
// a step of a scan bundle: its share of the bundle's front end and one round of the loop (cycles / 50; kCyclesScan* below)
static inline uint64_t scan_cost50(uint8_t op) { return (op & SCAN_OP_DIV) ? 10u : 4u; }
static inline uint64_t node_cost(const uint32_t* table, const Node& n) {
    return n.kind == N_FUSED ? fused_cost50(n.op) : n.kind == N_SCAN ? scan_cost50(n.op) : cost_of(table, class_of(n));
}

// Tree-height reduction, exact in the field: Add and Mul are associative and commutative, so a node at the end of a
// chain of the same operation (a linear combination `lc += c_j * x_j`, or c * (x^4 * x)) may be computed from the
// chain's leaves in any order.  A wave's time is the sum of its bundles and the bundle count follows the longest
// dependency chain, so every node whose own chain is its critical input is rebuilt as a tree over the leaves, cheapest
// and earliest-ready first: sum chains of n terms drop from n-1 to ceil(log2 n) levels, and a constant factor is folded
// into the early part of a product (M_ji * x^5 becomes (M_ji * x) * x^4, one multiplication level less per Poseidon
// round).  The intermediate nodes of the chain are still computed wherever something else (a witness element, another
// node) needs them; common subexpressions are shared; nodes that end up unused are dropped.
// Only Add/Mul nodes are touched, so every operation that can fail (graph.rs:634, :686-716) survives unchanged.
static void reduce_tree_height(Graph& g, size_t kMaxLeaves, const uint32_t* class_cost) {
    // <functional> comparators below
    const size_t N = g.nodes.size();
    Graph h;
    h.const_values = g.const_values;
    std::vector<uint32_t> m(N, 0xffffffffu);  // old index -> new index
    std::vector<uint64_t> rt;                 // earliest finish time of each new node (unbounded width)
    rt.reserve(N + N / 4);
    h.nodes.reserve(N + N / 4);
    // Value numbering of the Add / Mul nodes by operand pair (x <= y).  Not one big hash table: the tables of a
    // multi-million-node graph are far larger than the caches and every probe was a miss (2.4 per node; 10.5 M nodes:
    // 3.7 of the compile's 7.5 s).  Instead every node y heads a list, per operation, of the nodes whose larger operand
    // it is -- y was read a moment ago (its ready time), the list's members were made after it: the lookups stay in
    // the caches.  A list that grows beyond kListMax (one value combined with very many earlier ones) moves into a
    // hash table of its own kind, so the walk stays bounded.
    struct Link { uint32_t head[2], next; };
    const uint32_t NIL = 0xffffffffu, kListMax = 24;
    std::vector<Link> link;
    link.reserve(N + N / 4);
    std::vector<uint8_t> hashed;  // bit 0 / 1: node y's Add / Mul list lives in `overflow`
    hashed.reserve(N + N / 4);
    FlatMap64 overflow[2] = {FlatMap64(1024), FlatMap64(1024)};
    auto emit = [&](const Node& n, uint64_t t) -> uint32_t {
        h.nodes.push_back(n);
        rt.push_back(t);
        link.push_back(Link{{NIL, NIL}, NIL});
        hashed.push_back(0);
        return (uint32_t)(h.nodes.size() - 1);
    };
    for (size_t i = 0; i < N; ++i)  // constants first (rewrite_pow2_divisions appends some behind their users)
        if (g.nodes[i].kind == N_CONST) m[i] = emit(g.nodes[i], 0);
    auto is_ac = [&](uint32_t idx, uint8_t op) { return h.nodes[idx].kind == N_DUO && h.nodes[idx].op == op; };
    auto combine = [&](uint8_t op, uint32_t x, uint32_t y) -> uint32_t {  // shared (op, x, y) node
        if (x > y) std::swap(x, y);
        const int k = op == OP_MUL;
        const uint64_t key = ((uint64_t)x << 32) | y;
        const uint64_t cost = class_cost[k ? C_MUL : C_LIN];
        uint32_t idx;
        if (hashed[y] & (1u << k)) {
            if (overflow[k].find(key, &idx)) return idx;
            idx = emit(Node{N_DUO, op, x, y, 0}, std::max(rt[x], rt[y]) + cost);
            overflow[k].find_or_insert(key, idx, nullptr);
            return idx;
        }
        uint32_t len = 0;
        for (idx = link[y].head[k]; idx != NIL; idx = link[idx].next, ++len)
            if (h.nodes[idx].a == x) return idx;
        idx = emit(Node{N_DUO, op, x, y, 0}, std::max(rt[x], rt[y]) + cost);
        if (len >= kListMax) {  // the list moves into the hash table, this node with it
            for (uint32_t q = link[y].head[k]; q != NIL; q = link[q].next) overflow[k].find_or_insert(((uint64_t)h.nodes[q].a << 32) | y, q, nullptr);
            overflow[k].find_or_insert(key, idx, nullptr);
            hashed[y] |= (uint8_t)(1u << k);
            link[y].head[k] = NIL;
        } else {
            link[idx].next = link[y].head[k];
            link[y].head[k] = idx;
        }
        return idx;
    };
    // A node inside a chain -- its one user is a node of the same operation and it is no witness element -- needs no tree
    // of its own: the chain's end is rebuilt over the leaves and the inner node dies unless something else reads it.
    // (Without this every node of a chain of length L flattened up to kMaxLeaves leaves: most of the compile time of
    // multi-million-node graphs.)
    std::vector<uint8_t> inner(N, 0);
    {
        std::vector<uint32_t> n_users(N, 0), same_op_users(N, 0);
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            const int ar = arity_of(n);
            const uint32_t ops[3] = {n.a, n.b, n.c};
            for (int q = 0; q < ar; ++q) {
                n_users[ops[q]]++;
                const Node& o = g.nodes[ops[q]];
                if (n.kind == N_DUO && o.kind == N_DUO && o.op == n.op && (n.op == OP_ADD || n.op == OP_MUL)) same_op_users[ops[q]]++;
            }
        }
        for (uint32_t w : g.witness_signals) n_users[w] += 2;
        // (only where whole chains are flattened -- T = 1 -- : with the 8-leaf trees of wider tiles the inner nodes' own
        // trees are what keeps a long chain balanced)
        // Used for graphs beyond 16 M nodes only, where the compile time counts: the trees come out the same but are
        // emitted in another order, and the list scheduler then packs the bigint-class graph into more linear bundles
        // (round 3: 12 % more bundles, taken for the 10.5 M-node graph because it saved 4 of 10 compile seconds; round 4: with
        // the limb chains in scan bundles those linear bundles are 34 of 62 bundles per round instead of 19 of 49, 15 % of
        // the run time, and the rest of the compile got cheaper -- 1 M nodes: rewrites 1.17 -> 0.27 s;
        // CWC_TREE_INNER_SKIP=1 / 0 forces either way).
        const char* force = getenv("CWC_TREE_INNER_SKIP");
        if (kMaxLeaves >= 64 && (force ? atoi(force) != 0 : N > 16000000)) {
            for (size_t i = 0; i < N; ++i) inner[i] = n_users[i] == 1 && same_op_users[i] == 1;
            kMaxLeaves = 1u << 16;
        }
    }
    std::vector<uint8_t> inner_new;  // new-graph nodes that are such inner chain nodes
    std::vector<uint32_t> leaves;
    typedef std::pair<uint64_t, uint32_t> LeafKey;  // (ready time, ~position in `leaves`)
    std::vector<LeafKey> latest;
    std::vector<std::pair<uint64_t, uint32_t>> work;
    std::vector<uint64_t> times;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        if (n.kind == N_CONST) continue;
        Node c = n;
        const int ar = arity_of(n);
        if (ar >= 1) c.a = m[n.a];
        if (ar >= 2) c.b = m[n.b];
        if (ar >= 3) c.c = m[n.c];
        if (!(n.kind == N_DUO && (n.op == OP_ADD || n.op == OP_MUL))) {
            uint64_t t = 0;
            if (ar >= 1) t = rt[c.a];
            if (ar >= 2) t = std::max(t, rt[c.b]);
            if (ar >= 3) t = std::max(t, rt[c.c]);
            m[i] = emit(c, t + cost_of(class_cost, class_of(n)));
            continue;
        }
        if (inner[i]) {
            m[i] = combine(n.op, c.a, c.b);
            if (inner_new.size() < h.nodes.size()) inner_new.resize(h.nodes.size() + h.nodes.size() / 2 + 16, 0);
            inner_new[m[i]] = 1;
            continue;
        }
        const uint64_t cost = class_cost[n.op == OP_MUL ? C_MUL : C_LIN];
        const uint64_t direct = std::max(rt[c.a], rt[c.b]) + cost;
        // flatten: keep opening the latest-ready leaf while it is a node of the same operation
        // (a max-heap on (ready time, earliest position in `leaves`): the leaf a linear scan for the first maximum finds)
        leaves.clear();
        leaves.push_back(c.a);
        leaves.push_back(c.b);
        bool opened = false;
        // the chain's own inner nodes (emitted unbalanced above) are opened whatever their ready time ...
        for (size_t q = 0; q < leaves.size() && leaves.size() < kMaxLeaves;) {
            const uint32_t L = leaves[q];
            if (L < inner_new.size() && inner_new[L] && is_ac(L, n.op)) {
                leaves[q] = h.nodes[L].a;
                leaves.push_back(h.nodes[L].b);
                opened = true;
            } else {
                ++q;
            }
        }
        // ... then the latest-ready leaf while it is a node of the same operation
        latest.clear();
        for (size_t q = 0; q < leaves.size(); ++q) latest.push_back(LeafKey(rt[leaves[q]], ~(uint32_t)q));
        std::make_heap(latest.begin(), latest.end());
        while (leaves.size() < kMaxLeaves) {
            const uint32_t worst = ~latest.front().second;
            const uint32_t L = leaves[worst];
            if (!is_ac(L, n.op)) break;
            std::pop_heap(latest.begin(), latest.end());
            latest.pop_back();
            leaves[worst] = h.nodes[L].a;
            latest.push_back(LeafKey(rt[h.nodes[L].a], ~worst));
            std::push_heap(latest.begin(), latest.end());
            latest.push_back(LeafKey(rt[h.nodes[L].b], ~(uint32_t)leaves.size()));
            std::push_heap(latest.begin(), latest.end());
            leaves.push_back(h.nodes[L].b);
            opened = true;
        }
        uint32_t result = 0xffffffffu;
        if (opened) {
            // would the rebuilt tree finish earlier?  (computed on times only, nothing is emitted yet)
            // (min-heaps on (ready time, node): the two earliest are combined until one is left)
            times.clear();
            for (uint32_t L : leaves) times.push_back(rt[L]);
            std::make_heap(times.begin(), times.end(), std::greater<uint64_t>());
            while (times.size() > 1) {
                std::pop_heap(times.begin(), times.end(), std::greater<uint64_t>());
                const uint64_t t0 = times.back();
                times.pop_back();
                std::pop_heap(times.begin(), times.end(), std::greater<uint64_t>());
                const uint64_t t1 = times.back();
                times.back() = std::max(t0, t1) + cost;
                std::push_heap(times.begin(), times.end(), std::greater<uint64_t>());
            }
            if (times[0] < direct) {
                typedef std::pair<uint64_t, uint32_t> W;
                work.clear();
                for (uint32_t L : leaves) work.emplace_back(rt[L], L);
                std::make_heap(work.begin(), work.end(), std::greater<W>());
                while (work.size() > 1) {
                    std::pop_heap(work.begin(), work.end(), std::greater<W>());
                    const uint32_t x = work.back().second;
                    work.pop_back();
                    std::pop_heap(work.begin(), work.end(), std::greater<W>());
                    const uint32_t idx = combine(n.op, x, work.back().second);
                    work.back() = W(rt[idx], idx);
                    std::push_heap(work.begin(), work.end(), std::greater<W>());
                }
                result = work[0].second;
            }
        }
        m[i] = result != 0xffffffffu ? result : combine(n.op, c.a, c.b);
    }
    // drop what nothing needs any more: roots are the witness elements and every node that is not a plain Add/Mul
    const size_t M = h.nodes.size();
    std::vector<uint8_t> live(M, 0);
    for (uint32_t w : g.witness_signals) live[m[w]] = 1;
    for (size_t i = 0; i < M; ++i) {
        const Node& n = h.nodes[i];
        if (n.kind != N_CONST && !(n.kind == N_DUO && (n.op == OP_ADD || n.op == OP_MUL))) live[i] = 1;
    }
    for (size_t i = M; i-- > 0;) {
        if (!live[i]) continue;
        const Node& n = h.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) live[n.a] = 1;
        if (ar >= 2) live[n.b] = 1;
        if (ar >= 3) live[n.c] = 1;
    }
    std::vector<uint32_t> pos(M, 0xffffffffu);
    std::vector<Node> kept;
    kept.reserve(M);
    for (size_t i = 0; i < M; ++i) {
        if (!live[i]) continue;
        Node n = h.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) n.a = pos[n.a];
        if (ar >= 2) n.b = pos[n.b];
        if (ar >= 3) n.c = pos[n.c];
        pos[i] = (uint32_t)kept.size();
        kept.push_back(n);
    }
    for (uint32_t& w : g.witness_signals) w = pos[m[w]];
    g.nodes.swap(kept);
}

// ---- representation inference ------------------------------------------------------------------------------------
// The interpreter keeps field elements in Montgomery form (x * 2^256 mod r); the integer operations of the reference
// (shifts, bit operations, Idiv / Mod, ordered comparisons: src/graph.rs:112-133, 621-769) work on the canonical integer,
// and a bundle of them spends most of its time converting: two operands out of Montgomery form, the result back in
// (three products around a few dozen instructions of integer work).  Graphs that compute on limbs and bits (bigint /
// long-division circuits, range checks) chain such operations through additions and multiplications, none of which
// cares about the form: a + b and a - b hold in either form, and the Montgomery product of a canonical and a Montgomery
// operand IS the canonical product.  So every value gets ONE form, Montgomery (REP_M) or canonical (REP_C):
//   Input -> M.  Add / Sub / Neg / TernCond results: the common form of their operands.  Mul: (M, M) -> M, (M, C) -> C.
//   Integer operations and comparisons read either form (per-bundle header bits say which operands still need the
//   conversion) and write the form their users prefer.  Div: Montgomery operands, Montgomery result.
// Where the forms of two operands do not fit (Add of an M and a C value, Mul of two C values, ...) one of them is
// converted by an inserted multiplication with a constant: x_M * (2^-256)_M = x_C, x_C * (2^256)_M = x_M; a value is
// converted at most once per direction.  Constants serve either form (the table holds canonical copies where needed).
// Graphs without integer chains come out all-Montgomery, as before.
static const uint8_t REP_M = 0, REP_C = 1;
static const uint8_t VF_A_CANON = 1, VF_B_CANON = 2, VF_OUT_CANON = 4;
static const uint8_t VF_MUL_CC = 8;  // a multiplication of two canonical integers that stays canonical (C_MUL bundles with HDR_MUL_CC)
static bool is_integer_class(int c) { return c == C_BIT || c == C_IDIVMOD || c == C_CMPS; }
// allow_cc (limb-arithmetic graphs, tile widths with the MODE 2 interpreter instances): the product of two canonical values
// stays a node of its own kind -- both factors canonical, result canonical (VF_MUL_CC) -- instead of converting one factor:
// limb products are far below r, and the kernel multiplies limb-sized integers directly (general operands: two Montgomery
// products).
static void infer_representations(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, uint64_t& n_conversions, uint64_t& n_canonical, bool all_montgomery,
                                  bool allow_cc, uint64_t& n_cc) {
    const size_t N = g.nodes.size();
    const bool off = all_montgomery || getenv("CWC_NO_REP_INFERENCE") != nullptr;
    // what the users of a value would rather read: > 0 canonical
    std::vector<float> pref(N, 0.0f);
    for (size_t i = N; !off && i-- > 0;) {
        const Node& n = g.nodes[i];
        const int ar = arity_of(n);
        if (!ar) continue;
        const int c = class_of(n);
        float w = 0.0f;
        if (is_integer_class(c)) w = 1.0f;
        else if (c == C_DIV) w = -1.0f;
        else if (c == C_LIN || c == C_MUL || c == C_TERN) w = 0.5f * std::max(-2.0f, std::min(2.0f, pref[i]));
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = (c == C_TERN ? 1 : 0); q < ar; ++q)  // (TernCond tests its first operand for zero: either form)
            if (!(c == C_BIT && n.op == OP_BITX && q == 1) && g.nodes[ops[q]].kind != N_CONST) pref[ops[q]] += w;
    }
    std::vector<Node> out;
    out.reserve(N + N / 8);
    std::vector<uint32_t> at(N, 0xffffffffu);            // old node -> new index
    std::vector<uint32_t> converted(N, 0xffffffffu);     // old node -> new index of its value in the other form
    std::vector<uint8_t> orep(N, REP_M);
    rep.clear();
    vflags.clear();
    uint32_t k_to_c = 0xffffffffu, k_to_m = 0xffffffffu;  // constant nodes 2^-256 and 2^256 mod r
    auto emit = [&](const Node& n, uint8_t r, uint8_t f) -> uint32_t {
        out.push_back(n);
        rep.push_back(r);
        vflags.push_back(f);
        return (uint32_t)out.size() - 1;
    };
    auto konst = [&](bool to_c) -> uint32_t {
        uint32_t& k = to_c ? k_to_c : k_to_m;
        if (k == 0xffffffffu) {
            // 2^256 mod r and its inverse (canonical values; the table holds their Montgomery forms 2^512 mod r and 1)
            const Fr r1 = Fr{{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
            const Fr rinv = fr_from_mont(fr_from_mont(r1));  // ((2^256 * 2^-256) * 2^-256) = 2^-256
            g.const_values.push_back(to_c ? rinv : r1);
            k = emit(Node{N_CONST, 0, (uint32_t)g.const_values.size() - 1, 0, 0}, REP_M, 0);
        }
        return k;
    };
    auto is_const = [&](uint32_t o) { return g.nodes[o].kind == N_CONST; };
    // operand o (old index) in form `want`; constants serve either form
    auto get = [&](uint32_t o, uint8_t want) -> uint32_t {
        if (is_const(o) || orep[o] == want) return at[o];
        if (converted[o] == 0xffffffffu) {
            const uint32_t k = konst(want == REP_C);
            converted[o] = emit(Node{N_DUO, OP_MUL, at[o], k, 0}, want, 0);
            ++n_conversions;
        }
        return converted[o];
    };
    for (size_t i = 0; i < N; ++i)  // constants first: the rewrites append theirs behind their users
        if (g.nodes[i].kind == N_CONST) at[i] = emit(g.nodes[i], REP_M, 0);
    for (size_t i = 0; i < N; ++i) {
        Node n = g.nodes[i];
        if (n.kind == N_CONST) continue;
        const int ar = arity_of(n);
        const int c = class_of(n);
        uint8_t r = REP_M, f = 0;
        if (ar && !off) {
            // (no vote at all -- a value only the witness reads: an integer operation then keeps its canonical result, which saves
            // its bundle the conversion and lets limb recurrences that end in witness elements run as scan bundles)
            const bool want_c = pref[i] > 0.0f || (pref[i] == 0.0f && is_integer_class(c));
            auto form_of = [&](uint32_t o, uint8_t if_const) -> uint8_t { return is_const(o) ? if_const : orep[o]; };
            if (is_integer_class(c) || c == C_CMPZ) {
                if (is_integer_class(c)) {
                    f |= form_of(n.a, REP_C) == REP_C ? VF_A_CANON : 0;
                    f |= (n.op == OP_BITX || form_of(n.b, REP_C) == REP_C) ? VF_B_CANON : 0;
                    n.a = at[n.a];
                    n.b = at[n.b];
                } else if (n.op == OP_EQ || n.op == OP_NEQ) {  // equal forms on both sides (a constant follows the other side)
                    const uint8_t side = is_const(n.a) ? form_of(n.b, REP_M) : orep[n.a];
                    n.a = get(n.a, side);
                    n.b = get(n.b, side);
                    f |= side == REP_C ? VF_A_CANON : 0;  // (not a header bit for this class: which copy of a constant operand is read)
                } else {  // Land / Lor: zero tests
                    n.a = at[n.a];
                    n.b = at[n.b];
                }
                r = want_c ? REP_C : REP_M;
                f |= want_c ? VF_OUT_CANON : 0;
            } else if (c == C_MUL) {
                uint8_t ra = form_of(n.a, REP_M), rb = form_of(n.b, REP_M);
                if (is_const(n.a) != is_const(n.b)) {  // x * constant: the constant in Montgomery form keeps x's form
                    r = is_const(n.a) ? rb : ra;
                    n.a = at[n.a];
                    n.b = at[n.b];
                } else {
                    if (ra == REP_C && rb == REP_C && allow_cc && !is_const(n.a) && !is_const(n.b)) {
                        f |= VF_MUL_CC;
                        ++n_cc;
                    } else if (ra == REP_C && rb == REP_C) {  // one factor into Montgomery form: the one that is already converted, else the second
                        if (!is_const(n.a) && converted[n.a] != 0xffffffffu) ra = REP_M;
                        else rb = REP_M;
                    }
                    n.a = get(n.a, ra);
                    n.b = get(n.b, rb);
                    r = (ra == REP_C || rb == REP_C) ? REP_C : REP_M;
                }
            } else if (c == C_DIV) {
                n.a = get(n.a, REP_M);
                n.b = get(n.b, REP_M);
            } else if (c == C_LIN || c == C_TERN) {
                const uint32_t x = n.kind == N_UNO ? n.a : n.kind == N_TRES ? n.b : n.a, y = n.kind == N_UNO ? n.a : n.kind == N_TRES ? n.c : n.b;
                uint8_t side;
                if (is_const(x) && is_const(y)) side = want_c ? REP_C : REP_M;
                else if (is_const(x)) side = orep[y];
                else if (is_const(y)) side = orep[x];
                else if (orep[x] == orep[y]) side = orep[x];
                else side = want_c ? REP_C : REP_M;
                if (n.kind == N_UNO) {
                    n.a = get(n.a, side);
                } else if (n.kind == N_TRES) {
                    n.a = at[n.a];
                    n.b = get(n.b, side);
                    n.c = get(n.c, side);
                } else {
                    n.a = get(n.a, side);
                    n.b = get(n.b, side);
                }
                r = side;
            }
        } else if (ar) {
            n.a = at[n.a];
            if (ar >= 2) n.b = at[n.b];
            if (ar >= 3) n.c = at[n.c];
            if (is_integer_class(c)) f = (uint8_t)((is_const(g.nodes[i].a) ? VF_A_CANON : 0) | ((n.op == OP_BITX || is_const(g.nodes[i].b)) ? VF_B_CANON : 0));
        }
        orep[i] = r;
        n_canonical += ar && r == REP_C;
        at[i] = emit(n, r, f);
    }
    for (uint32_t& w : g.witness_signals) w = at[w];
    g.nodes.swap(out);
}

// ---- scan chains (round 4) ------------------------------------------------------------------------------------------
// Limb-wise big-integer circuits (RSA / long_div-class: BASELINE config 5) are serial recurrences over canonical integers,
// one step per limb: the carry chain of a multi-limb sum
//     t = x_c + carry_c;  limb_c = t mod 2^n (Band after the strength reduction);  carry_{c+1} = t \ 2^n (Shr)
// and the remainder chain of a long division by one limb
//     t = rem_c * 2^k + x_c;  q_c = t \ d;  rem_{c+1} = t mod d.
// Unfused, a step is two or three bundles on the graph's critical chain (Add, then Band + Shr side by side; Mul, Add, then
// Idiv + Mod), each with ~600 cycles of front end for a few dozen instructions of limb arithmetic: 1.74 M bundles for the
// 10.5 M-node graph.  A step whose inner nodes nothing else reads becomes a PAIR of N_SCAN nodes -- the step's OUT value
// (limb / quotient digit) and its ACC value (carry / remainder), both naming the step's operands (a = x, b = the accumulator
// coming in, c = the divisor) -- and the scheduler places the consecutive steps of a chain in consecutive pairs of node
// slots of ONE bundle (class C_SCAN, program_dev.h), which runs them with a loop inside the bundle.  Exact: the kernel's
// step is the same field addition / product and the same integer operations (graph.rs:105, 110-121, 637-687) on the same
// canonical integers; nothing that can fail is involved (Band with 2^n - 1 stays below 2^253, Shr / Idiv / Mod cannot fail).
// Only values that representation inference keeps canonical are touched.  scan_imm[node]: CARRY the shift n, DIV the node
// index of the constant 2^k (its Montgomery form is what the general path multiplies with).
// scan_partner[node]: the other node of the step.
static void detect_scans(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner,
                         uint64_t& n_steps) {
    const size_t N = g.nodes.size();
    static const uint32_t NONE = 0xffffffffu;
    std::vector<uint32_t> uses(N, 0);
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) uses[ops[q]]++;
    }
    for (uint32_t w : g.witness_signals) uses[w] += 2;  // (a witness element is never an inner node)
    // constants: 2^k -> k, 2^n - 1 -> n, small integers
    auto const_value = [&](uint32_t idx) -> const Fr* { return g.nodes[idx].kind == N_CONST ? &g.const_values[g.nodes[idx].a] : nullptr; };
    auto pow2_of = [&](uint32_t idx) -> int {
        const Fr* v = const_value(idx);
        if (!v) return -1;
        int k = -1, bits = 0;
        for (int w = 0; w < 8; ++w)
            if (v->v[w]) {
                bits += __builtin_popcount(v->v[w]);
                k = 32 * w + __builtin_ctz(v->v[w]);
            }
        return bits == 1 && k >= 1 && k <= 253 ? k : -1;
    };
    auto mask_of = [&](uint32_t idx) -> int {  // 2^n - 1 -> n
        const Fr* v = const_value(idx);
        if (!v) return -1;
        int n = 0;
        bool ended = false;
        for (int w = 0; w < 8; ++w) {
            const uint32_t x = v->v[w];
            if (ended) {
                if (x) return -1;
            } else if (x == 0xffffffffu) {
                n += 32;
            } else {
                if (x & (x + 1u)) return -1;
                n += __builtin_popcount(x);
                ended = true;
            }
        }
        return n >= 1 && n <= 253 ? n : -1;
    };
    auto small_of = [&](uint32_t idx) -> int {  // a shift count
        const Fr* v = const_value(idx);
        if (!v) return -1;
        for (int w = 1; w < 8; ++w)
            if (v->v[w]) return -1;
        return v->v[0] >= 1 && v->v[0] <= 253 ? (int)v->v[0] : -1;
    };
    auto canon = [&](uint32_t o) { return g.nodes[o].kind == N_CONST || rep[o] == REP_C; };
    // the two users of every candidate t: (Band, Shr) or (Idiv, Mod)
    std::vector<uint32_t> user_out(N, NONE), user_acc(N, NONE);
    for (size_t j = 0; j < N; ++j) {
        Node& n = g.nodes[j];
        if (n.kind != N_DUO) continue;
        // (value numbering orders the operands of commutative operations by index: the mask may come first)
        if (n.op == OP_BAND && g.nodes[n.a].kind == N_CONST && g.nodes[n.b].kind != N_CONST) {
            std::swap(n.a, n.b);
            vflags[j] = (uint8_t)((vflags[j] & ~(VF_A_CANON | VF_B_CANON)) | ((vflags[j] & VF_A_CANON) ? VF_B_CANON : 0) | ((vflags[j] & VF_B_CANON) ? VF_A_CANON : 0));
        }
        if (g.nodes[n.a].kind != N_DUO || g.nodes[n.a].op != OP_ADD) continue;
        const uint8_t want = VF_A_CANON | VF_B_CANON | VF_OUT_CANON;
        if ((vflags[j] & want) != want) continue;
        if (n.op == OP_BAND || n.op == OP_IDIV) user_out[n.a] = user_out[n.a] == NONE ? (uint32_t)j : NONE - 1;
        else if (n.op == OP_SHR || n.op == OP_MOD) user_acc[n.a] = user_acc[n.a] == NONE ? (uint32_t)j : NONE - 1;
    }
    struct Step { uint32_t t, out, acc, x, acc_in, d, imm; bool div; };
    std::vector<Step> steps;
    std::vector<uint32_t> step_of_acc(N, NONE);  // ACC node (Shr / Mod) -> step
    for (size_t t = 0; t < N; ++t) {
        const uint32_t o = user_out[t], a = user_acc[t];
        if (o >= NONE - 1 || a >= NONE - 1 || uses[t] != 2 || rep[t] != REP_C) continue;
        const Node& T_ = g.nodes[t];
        const Node &O = g.nodes[o], &A = g.nodes[a];
        if (!canon(T_.a) || !canon(T_.b)) continue;
        if (O.op == OP_BAND && A.op == OP_SHR) {
            const int n = small_of(A.b);
            if (n < 0 || mask_of(O.b) != n) continue;
            steps.push_back(Step{(uint32_t)t, o, a, T_.a, T_.b, 0, (uint32_t)n, false});
        } else if (O.op == OP_IDIV && A.op == OP_MOD && O.b == A.b && canon(O.b)) {
            // t = m + x with m = rem * 2^k read by nothing else
            int side = -1;
            for (int q = 0; q < 2 && side < 0; ++q) {
                const uint32_t m = q ? T_.b : T_.a;
                const Node& M = g.nodes[m];
                if (M.kind != N_DUO || M.op != OP_MUL || uses[m] != 1 || rep[m] != REP_C) continue;
                if ((pow2_of(M.b) >= 0 && canon(M.a) && g.nodes[M.a].kind != N_CONST) || (pow2_of(M.a) >= 0 && canon(M.b) && g.nodes[M.b].kind != N_CONST)) side = q;
            }
            if (side < 0) continue;
            const uint32_t m = side ? T_.b : T_.a, x = side ? T_.a : T_.b;
            const Node& M = g.nodes[m];
            const bool base_b = pow2_of(M.b) >= 0 && g.nodes[M.a].kind != N_CONST;
            steps.push_back(Step{(uint32_t)t, o, a, x, base_b ? M.a : M.b, O.b, base_b ? M.b : M.a, true});
        }
    }
    if (getenv("CWC_DEBUG_SCAN")) {
        size_t n_band = 0, n_shr = 0, pairs = 0, uses_ok = 0, rep_ok = 0, canon_ok = 0;
        for (size_t t = 0; t < N; ++t) {
            n_band += user_out[t] < NONE - 1;
            n_shr += user_acc[t] < NONE - 1;
            if (user_out[t] >= NONE - 1 || user_acc[t] >= NONE - 1) continue;
            ++pairs;
            uses_ok += uses[t] == 2;
            rep_ok += rep[t] == REP_C;
            canon_ok += canon(g.nodes[t].a) && canon(g.nodes[t].b);
        }
        fprintf(stderr, "scan detection: %zu Add nodes with an OUT user, %zu with an ACC user, %zu with both; of those uses == 2: %zu, canonical: %zu, canonical operands: %zu; steps %zu\n",
                n_band, n_shr, pairs, uses_ok, rep_ok, canon_ok, steps.size());
    }
    if (steps.empty()) return;
    for (size_t k = 0; k < steps.size(); ++k) step_of_acc[steps[k].acc] = (uint32_t)k;
    // CARRY steps: the accumulator is the operand that is another step's carry (so that chains link up); either one at a chain's head
    for (Step& st : steps) {
        if (st.div) continue;
        const uint32_t sx = step_of_acc[st.x], sa = step_of_acc[st.acc_in];
        const bool x_links = sx != NONE && !steps[sx].div && steps[sx].imm == st.imm, a_links = sa != NONE && !steps[sa].div && steps[sa].imm == st.imm;
        if (x_links && !a_links) std::swap(st.x, st.acc_in);
    }
    // rewrite: OUT and ACC become N_SCAN nodes on the step's operands, the inner nodes (t, m) lose their users
    std::vector<uint8_t> dead(N, 0);
    for (const Step& st : steps) {
        const uint8_t kind = st.div ? SCAN_OP_DIV : 0;
        g.nodes[st.out] = Node{N_SCAN, kind, st.x, st.acc_in, st.d};
        g.nodes[st.acc] = Node{N_SCAN, (uint8_t)(kind | SCAN_OP_ACC), st.x, st.acc_in, st.d};
        vflags[st.out] = vflags[st.acc] = 0;
        dead[st.t] = 1;
        if (st.div) dead[g.nodes[st.t].a == st.x ? g.nodes[st.t].b : g.nodes[st.t].a] = 1;
    }
    // Node order: a step's nodes sit where Band / Shr (Idiv / Mod) sat, behind t and therefore behind every operand.
    scan_imm.assign(N, 0);
    scan_partner.assign(N, NONE);
    for (const Step& st : steps) {
        scan_imm[st.out] = scan_imm[st.acc] = st.imm;
        scan_partner[st.out] = st.acc;
        scan_partner[st.acc] = st.out;
    }
    n_steps += steps.size();
    // compact (the dead inner nodes would be scheduled)
    std::vector<uint32_t> pos(N, NONE);
    std::vector<Node> kept;
    std::vector<uint8_t> krep, kfl;
    std::vector<uint32_t> kimm, kpart;
    kept.reserve(N);
    krep.reserve(N);
    kfl.reserve(N);
    kimm.reserve(N);
    kpart.reserve(N);
    for (size_t i = 0; i < N; ++i) {
        if (dead[i]) continue;
        Node n = g.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) n.a = pos[n.a];
        if (ar >= 2) n.b = pos[n.b];
        if (ar >= 3) n.c = pos[n.c];
        uint32_t imm = scan_imm[i];
        if (n.kind == N_SCAN && (n.op & SCAN_OP_DIV)) imm = pos[imm];  // (the constant 2^k: a node index)
        pos[i] = (uint32_t)kept.size();
        kept.push_back(n);
        krep.push_back(rep[i]);
        kfl.push_back(vflags[i]);
        kimm.push_back(imm);
        kpart.push_back(scan_partner[i]);  // (old index: renumbered below, the partner may sit behind this node)
    }
    for (uint32_t& x : kpart)
        if (x != NONE) x = pos[x];
    for (uint32_t& w : g.witness_signals) w = pos[w];
    g.nodes.swap(kept);
    rep.swap(krep);
    vflags.swap(kfl);
    scan_imm.swap(kimm);
    scan_partner.swap(kpart);
}

// ---- fused narrow chains (round 3) --------------------------------------------------------------------------------
// A lone wavefront pays ~600 cycles for every bundle before any arithmetic (operand / record reads, staging loads, ring
// write), and the graphs that bound small batches are ONE dependent chain for long stretches: a Poseidon partial round
// of a lone Merkle chain is t -> t^2 -> t^4 -> (M t) t^4 -> + side sum, four bundles of one or two nodes.  A fused node
// keeps such a sequence in the registers of the four lanes that share its product (class C_MULF): (s * s) * m + c, or
// a * b +- c, is one node in one bundle.  Exact in the field (the same products and sums, graph.rs:105, 110-111); only
// Mul / Add / Sub nodes whose values are in one form are touched, nothing that can fail.  The inner nodes stay wherever
// something else -- a witness element, another node -- reads them (the product is then computed twice: once inside the
// fused node on the critical chain, once off it in a lane that would idle), and die otherwise.
// Only nodes within `slack` (scheduler cost units) of the graph's critical path are fused: off the critical path a fused
// node saves nothing and takes one of the few node slots of a narrow bundle.
static void fuse_narrow_chains(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, const uint32_t* class_cost, uint32_t slack_permille,
                               bool two_stage_only, uint64_t& n_fused) {
    const size_t N = g.nodes.size();
    std::vector<uint64_t> rt(N, 0), ht(N, 0);  // earliest finish time / longest path to a sink (own cost included in both)
    std::vector<uint32_t> n_users(N, 0);
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const int ar = arity_of(n);
        if (n.kind == N_CONST) continue;
        const uint32_t ops[3] = {n.a, n.b, n.c};
        uint64_t t = 0;
        for (int q = 0; q < ar; ++q) {
            t = std::max(t, rt[ops[q]]);
            n_users[ops[q]]++;
        }
        rt[i] = t + node_cost(class_cost, n);
    }
    uint64_t cp = 0;
    for (size_t i = N; i-- > 0;) {
        const Node& n = g.nodes[i];
        if (n.kind == N_CONST) continue;
        ht[i] += node_cost(class_cost, n);
        cp = std::max(cp, rt[i] - node_cost(class_cost, n) + ht[i]);
        const int ar = arity_of(n);
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < ar; ++q) ht[ops[q]] = std::max(ht[ops[q]], ht[i]);
    }
    const uint64_t slack = cp / 1000 * slack_permille;
    auto is_mul = [&](uint32_t i) { return g.nodes[i].kind == N_DUO && g.nodes[i].op == OP_MUL; };
    auto critical = [&](uint32_t i) { return rt[i] - node_cost(class_cost, g.nodes[i]) + ht[i] + slack >= cp; };
    bool any = false;
    for (size_t i = 0; i < N; ++i) {
        Node& n = g.nodes[i];
        if (n.kind != N_DUO || (n.op != OP_ADD && n.op != OP_SUB) || !critical((uint32_t)i)) continue;
        // the product side: the later of the two operands if it is a multiplication
        const bool a_mul = is_mul(n.a), b_mul = is_mul(n.b);
        if (!a_mul && !b_mul) continue;
        const bool take_a = a_mul && (!b_mul || rt[n.a] >= rt[n.b]);
        const uint32_t m1 = take_a ? n.a : n.b, c = take_a ? n.b : n.a;
        if (rt[m1] < rt[c]) continue;  // (the sum waits for its other operand: nothing to gain)
        const uint32_t lin = n.op == OP_ADD ? FOP_ADD : take_a ? FOP_SUB : FOP_RSUB;
        const Node& M1 = g.nodes[m1];
        // (s * s) * m + c: the square on the product's later side
        const uint32_t p = rt[M1.a] >= rt[M1.b] ? M1.a : M1.b, q = p == M1.a ? M1.b : M1.a;
        const uint8_t r = rep[i];
        // (the three-stage form wants its last operand before the bundle starts; where that operand is a side sum that is
        // ready only by the time the products are -- Poseidon's partial rounds -- product + sum alone is the better node)
        if (!two_stage_only && M1.a != M1.b && is_mul(p) && g.nodes[p].a == g.nodes[p].b && rt[p] >= rt[q] && rep[p] == rep[g.nodes[p].a] && rep[m1] == r) {
            n = Node{N_FUSED, fused_code(true, FOP_MUL, lin), g.nodes[p].a, q, c};
        } else if (rep[m1] == r) {
            n = Node{N_FUSED, fused_code(false, lin, FOP_NONE), M1.a, M1.b, c};
        } else {
            continue;
        }
        any = true;
        ++n_fused;
    }
    if (!any) return;
    // drop what nothing reads any more (roots: witness elements and everything that is not a plain Add / Mul / fused node)
    std::vector<uint8_t> live(N, 0);
    for (uint32_t w : g.witness_signals) live[w] = 1;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const bool pure = n.kind == N_FUSED || (n.kind == N_DUO && (n.op == OP_ADD || n.op == OP_SUB || n.op == OP_MUL)) || n.kind == N_CONST;
        if (!pure) live[i] = 1;
    }
    for (size_t i = N; i-- > 0;) {
        if (!live[i]) continue;
        const Node& n = g.nodes[i];
        const int ar = arity_of(n);
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < ar; ++q) live[ops[q]] = 1;
    }
    std::vector<uint32_t> pos(N, 0xffffffffu);
    std::vector<Node> kept;
    std::vector<uint8_t> krep, kfl;
    kept.reserve(N);
    for (size_t i = 0; i < N; ++i) {
        if (!live[i]) continue;
        Node n = g.nodes[i];
        const int ar = arity_of(n);
        if (ar >= 1) n.a = pos[n.a];
        if (ar >= 2) n.b = pos[n.b];
        if (ar >= 3) n.c = pos[n.c];
        pos[i] = (uint32_t)kept.size();
        kept.push_back(n);
        krep.push_back(rep[i]);
        kfl.push_back(vflags[i]);
    }
    for (uint32_t& w : g.witness_signals) w = pos[w];
    g.nodes.swap(kept);
    rep.swap(krep);
    vflags.swap(kfl);
}

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
static const double kCyclesScanFront = 1000, kCyclesScanFrontDiv = 2200, kCyclesScanStepCarry = 170, kCyclesScanStepDiv = 460;
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
};
static const CycleTable kCycles;
double model_class_cycles(int c) { return c >= 0 && c < (int)C_COUNT ? kCycles[c] : 0.0; }
// a word that changes with the table: programs are chosen (and cached on disk) under one table
uint64_t model_table_id() {
    uint64_t h = 1469598103934665603ull;
    for (int c = 0; c < (int)C_COUNT; ++c) {
        const uint64_t x = (uint64_t)(kCycles[c] * 16.0);
        h = (h ^ x) * 1099511628211ull;
    }
    return h;
}
static uint32_t div_cost50() { return (uint32_t)(kCycles[C_DIV] / 50.0); }
// (round 2, bigint-class graph with every operand and result canonical: BIT 2 650, IDIVMOD 2 880, CMPS 1 900 net of stamps)
static const double kCyclesBitStraight = 1500;  // what a Shr-only / Band-only bundle saves against the per-lane select over all bit operations
static const double kCyclesBitx = 1300, kCyclesCoopRiders = 60, kCyclesOperandForm = 1200, kCyclesResultForm = 1450, kCyclesBitxOperandForm = 600;
double program_wave_cycles(const Program& p) {
    if (p.n_streams > 1) {  // the tile is done when its slowest stream is
        double m = 0;
        for (uint32_t s = 0; s < p.n_streams; ++s) m = std::max(m, std::max(p.stream_cycles[s], p.stream_chain_cycles[s]));
        return m;
    }
    double c = 0;
    for (int k = 0; k < (int)C_COUNT; ++k) c += kCycles[k] * (double)p.stats.class_bundles[k];
    return c - (kCycles[C_BIT] - kCyclesBitx) * (double)p.stats.n_bitx_bundles + kCyclesCoopRiders * (double)p.stats.n_coop_rider_bundles - (double)p.stats.form_cycles_saved;
}

// the multiplication and inversion bundles' part of it (bundles that are bound by instruction issue)
double program_wave_cycles_mul_div(const Program& p) {
    if (p.n_streams > 1) {
        uint32_t m = 0;
        for (uint32_t s = 1; s < p.n_streams; ++s)
            if (p.stream_cycles[s] > p.stream_cycles[m]) m = s;
        return p.stream_cycles_mul_div[m];
    }
    return kCycles[C_MUL] * (double)p.stats.class_bundles[C_MUL] + kCycles[C_MULQ] * (double)p.stats.class_bundles[C_MULQ] +
           kCycles[C_MULF] * (double)p.stats.class_bundles[C_MULF] + kCycles[C_DIV] * (double)p.stats.class_bundles[C_DIV];
}

// When a multiplication step becomes a narrow (four lanes per product) bundle: `fill` or more ready multiplications make
// a full-width bundle instead (it costs the same with 10 or 32 nodes); otherwise a narrow one if the multiplications
// within `slack_levels` multiplication levels (in the scheduler's cost units) of the most urgent ready node fit it.  The
// rest stays ready.  fill = 0: never narrow.
struct CoopPolicy {
    uint32_t fill;
    uint32_t slack_levels;  // ~0u: everything ready counts as urgent
    bool all_montgomery = false;  // no representation inference: every value in Montgomery form
    bool witness_slots = false;   // the slots of witness elements in witness order (see the slot allocation)
    uint32_t fuse = 0;            // fused narrow chains (fuse_narrow_chains): 0 off, else 1 + the slack, in thousandths of the critical path, within which nodes are fused; + 0x10000: product + sum nodes only
};
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
    if (!compile_variant(g, T, divider, true, base, out, err, &cache, false, streams)) return false;
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
    if (base.fill == 0 || forced || g.nodes.size() > 2000000) return true;  // (huge graphs: one schedule, compile time counts)
    const CoopPolicy more[] = {{0, 0}, {std::max(12u, G * 3 / 8), 0}, {G / 2, 2}, {G * 5 / 8, 2}};
    CoopPolicy kept = base;
    for (CoopPolicy pol : more) {
        pol.all_montgomery = base.all_montgomery;
        pol.witness_slots = base.witness_slots;
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
            if (n.kind != N_CONST) st.class_nodes[class_of(n)]++;
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
    const bool limb_graph = T <= SCAN_MAX_T && G >= 2 && !getenv("CWC_NO_SCAN") && st.depth_scan * 10 < st.depth * 8;
    uint64_t n_mul_cc = 0;
    infer_representations(g, node_rep, node_vflags, st.n_conversions, st.n_canonical, policy.all_montgomery, limb_graph && !getenv("CWC_NO_MUL_CC"), n_mul_cc);
    N = g.nodes.size();
    phase("representation inference");
    // ---- scan chains: the steps of serial limb recurrences as pairs of N_SCAN nodes (class C_SCAN) ----
    std::vector<uint32_t> scan_imm, scan_partner;
    if (T <= SCAN_MAX_T && G >= 2 && !getenv("CWC_NO_SCAN")) {
        detect_scans(g, node_rep, node_vflags, scan_imm, scan_partner, st.n_scan_steps);
        N = g.nodes.size();
        phase("scan chains");
    }
    if (policy.fuse && policy.fill && T <= COOP_FUSE_MAX_T && G > 1 && st.n_scan_steps == 0 && n_mul_cc == 0) {  // (a program has fused bundles or scan bundles: one interpreter instance each)
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
                case C_SCAN: return (n.op & SCAN_OP_DIV) ? kCyclesScanStepDiv : kCyclesScanStepCarry;
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
        if (c == C_SCAN) return true;  // x, the accumulator, the divisor: canonical integers
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
        for (size_t i = N; i-- > 0;) {
            if (g.nodes[i].kind == N_CONST) continue;
            uint64_t h = 0;
            for (uint32_t u : users[i]) h = std::max(h, height[u]);
            height[i] = h + node_cost(class_cost, g.nodes[i]);
        }
        if (getenv("CWC_DEBUG_CRITICAL_PATH")) {  // diagnostic: class composition of the cost-weighted critical path
            uint32_t cur = 0xffffffffu;
            for (size_t i = 0; i < N; ++i)
                if (g.nodes[i].kind != N_CONST && (cur == 0xffffffffu || height[i] > height[cur])) cur = (uint32_t)i;
            uint64_t cnt[C_COUNT] = {0}, total = height[cur];
            std::string seq;
            while (true) {
                const int c = class_of(g.nodes[cur]);
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
        const bool has_scans = st.n_scan_steps != 0;
        std::vector<uint32_t> scan_keys;             // distinct (kind << 8 | shift)
        std::vector<uint32_t> scan_next;             // ACC node -> OUT node of the step that continues its chain
        auto scan_shift_of = [&](uint32_t i) -> uint32_t {
            if (!(g.nodes[i].op & SCAN_OP_DIV)) return scan_imm[i];
            const Fr& v = g.const_values[g.nodes[scan_imm[i]].a];  // the constant 2^k
            for (int w = 0; w < 8; ++w)
                if (v.v[w]) return 32u * w + (uint32_t)__builtin_ctz(v.v[w]);
            return 0;
        };
        auto scan_key_of = [&](uint32_t i) -> uint32_t { return ((g.nodes[i].op & SCAN_OP_DIV) ? 256u : 0u) | scan_shift_of(i); };
        auto scan_key_index = [&](uint32_t key) -> int {
            for (size_t k = 0; k < scan_keys.size(); ++k)
                if (scan_keys[k] == key) return (int)k;
            return -1;
        };
        if (has_scans) {
            scan_next.assign(N, 0xffffffffu);
            for (size_t i = 0; i < N; ++i) {
                const Node& n = g.nodes[i];
                if (n.kind != N_SCAN || (n.op & SCAN_OP_ACC)) continue;
                const uint32_t key = scan_key_of((uint32_t)i);
                if (scan_key_index(key) < 0) scan_keys.push_back(key);
                const Node& pr = g.nodes[n.b];
                if (pr.kind == N_SCAN && (pr.op & SCAN_OP_ACC) && scan_key_of(n.b) == key && scan_next[n.b] == 0xffffffffu) scan_next[n.b] = (uint32_t)i;
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
            if (has_scans) placed.assign(N, 0);
            auto push = [&](uint32_t i) {
                int hc = class_of(g.nodes[i]);
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
                    err = "internal error: scheduler found no ready node";
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
                if (best % (int)C_COUNT == (int)C_SCAN && best >= (int)C_COUNT * 17 && !getenv("CWC_SCAN_EAGER")) {
                    const uint32_t head = tie_reverse ? heap[best].front().second : ~heap[best].front().second;
                    size_t len_ready = 1, len_all = 1;
                    bool contiguous = true;
                    for (uint32_t cur = head; len_all < G / 2; ++len_all) {
                        const uint32_t nx = scan_next[scan_partner[cur]];
                        if (nx == 0xffffffffu || so[nx] != s || placed[nx]) break;
                        contiguous = contiguous && indeg[nx] == 1 && indeg[scan_partner[nx]] == 1;
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
                            if (nx == 0xffffffffu || so[nx] != s || placed[nx] || indeg[nx] != 1 || indeg[scan_partner[nx]] != 1) break;
                            const Node& nn = g.nodes[nx];
                            if (in_bundle(nn.a) || ((nn.op & SCAN_OP_DIV) && in_bundle(nn.c))) break;  // (x or the divisor comes out of this very bundle)
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
                if (n.kind == N_SCAN) {  // the two nodes of a step sit in one bundle: one part (their operands may all be prologue values)
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
    const uint32_t zero_off = (uint32_t)((uint64_t)zero_const * slot_bytes);
    auto mem_off = [&](uint32_t producer) -> uint64_t {
        if (g.nodes[producer].kind == N_CONST) return (uint64_t)(ref[producer] & ~REF_CONST) * slot_bytes;
        return ((uint64_t)NC + ref[producer]) * slot_bytes;
    };
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
                    const bool start = pair == 0 || (order[k0 + 2 * pair - 1] & ~REQ_FLAG) != n.b;
                    ctrl |= (is_acc ? SCAN_ROLE_ACC : 0u) | (start ? SCAN_START : 0u);
                    if (!is_acc) {
                        enc_operand(n.a, 0);
                        if (start) enc_operand(n.b, 1);
                        scan_run = start ? 1u : scan_run + 1u;
                        scan_longest = std::max(scan_longest, scan_run);
                        scan_bits = (is_div ? HDR_SCAN_DIV : 0u) | (scan_shift_of_node(i) << HDR_SCAN_SHIFT_SHIFT);
                    } else if (is_div) {
                        enc_to(n.c, 2, off[0], lds[0]);
                        off[1] = (uint32_t)mem_off(scan_imm[i]);
                    }
                    break;
                }
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
            form_saved = kCycles[C_SCAN] - ((scan_bits & HDR_SCAN_DIV) ? kCyclesScanFrontDiv + (double)scan_longest * kCyclesScanStepDiv : kCyclesScanFront + (double)scan_longest * kCyclesScanStepCarry);
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
    if (ws_tile_bytes(NC, n_slots, T) > 0xffffffffull) {
        err = "graph too large: one tile of the value workspace exceeds the 4 GiB buffer range";
        return false;
    }
    // second pass: destination byte offsets (trash slot = n_slots) + ctrl, and inactive padding records
    const uint32_t trash_off = (uint32_t)(((uint64_t)NC + n_slots) * slot_bytes);
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

// ---- structural validation of a program that did not come out of compile_program (an imported blob) ----------------
// Everything the interpreter and the pack kernel address through the program is checked against the tile and LDS
// geometry: a truncated or corrupted broadcast must fail here, not read or write out of bounds on the device.
bool validate_program(const Program& p, std::string& err) {
    auto bad = [&](const std::string& m) {
        err = "invalid program: " + m;
        return false;
    };
    const uint32_t T = p.T, G = p.G;
    if (T == 0 || T > 64 || (T & (T - 1)) || G != 64 / T) return bad("tile geometry");
    if (p.divider != 0 && p.divider != 1 && p.divider != 3 && p.divider != 4) return bad("divider mode");
    if (p.divider && T == 64) return bad("divider program at tile width 64");
    const uint64_t tile_bytes = ws_tile_bytes(p.n_const, p.n_slots, T);
    if (p.n_const == 0 || p.n_slots == 0 || tile_bytes > 0xffffffffull) return bad("tile size");
    if ((uint64_t)p.n_bundles * G * 16ull > 0xffffffffull) return bad("record stream size");
    if (p.hdr.size() != p.n_bundles || p.recs.size() != (size_t)p.n_bundles * G * 4 || p.crefs.size() != (size_t)p.n_cref_rows * G ||
        p.consts.size() != (size_t)p.n_const * 8 || p.witness_refs.size() != p.n_witness || p.div_lanes.size() != p.n_div_requests)
        return bad("array sizes");
    const uint32_t slot_bytes = 32u * T, HI = 16u * T;
    const uint64_t trash_off = ((uint64_t)p.n_const + p.n_slots) * slot_bytes;
    // streams: consecutive bundle ranges, each starting at a multiple of the pipeline depths; one stream unless the
    // divider mode is none or one divider wave per interpreter
    const uint32_t NS = p.n_streams;
    if (NS != 1 && NS != 2 && NS != 4) return bad("stream count");
    if (NS > 1 && p.divider > 1) return bad("streams with a shared divider wave");
    uint32_t next_first = 0, req_sum = 0;
    for (uint32_t s = 0; s < NS; ++s) {
        if (p.stream_first[s] < next_first || (p.stream_first[s] % 4) != 0 || (uint64_t)p.stream_first[s] + p.stream_count[s] > p.n_bundles) return bad("stream ranges");
        next_first = p.stream_first[s] + p.stream_count[s];
        req_sum += p.stream_div_requests[s];
    }
    if (p.stream_first[0] != 0 || (NS == 1 && p.stream_count[0] != p.n_bundles) || req_sum != p.n_div_requests) return bad("stream ranges");
    uint32_t n_req = 0, n_get = 0, n_posts = 0;
    bool in_flight = false;
    uint32_t stream = 0, stream_req = 0, cref_row = 0;
    for (uint32_t b = 0; b < p.n_bundles; ++b) {
        while (stream + 1 < NS && b >= p.stream_first[stream + 1]) {
            if (in_flight || stream_req != p.stream_div_requests[stream]) return bad("division requests of stream " + std::to_string(stream));
            ++stream;
            stream_req = 0;
        }
        // (checked for the stream b belongs to: the interpreter wave of stream s starts its row counter at stream_cref_first[s])
        if (b == p.stream_first[stream] && p.stream_count[stream] && p.stream_cref_first[stream] != cref_row) return bad("third-operand rows of stream " + std::to_string(stream));
        const bool executed = b < p.stream_first[stream] + p.stream_count[stream];
        const uint32_t h = p.hdr[b], cls = h & HDR_CLASS_MASK, cnt = (h >> HDR_COUNT_SHIFT) & 0x7f;
        if (cls >= C_COUNT || (cls != C_SCAN && (h >> 19) != 0)) return bad("bundle " + std::to_string(b) + ": header");
        if (cls == C_SCAN) {  // pairs of record positions, an iteration count that covers the longest chain segment, a shift below 254
            const uint32_t iters = (h >> HDR_SCAN_ITER_SHIFT) + 1u, sh = (h >> HDR_SCAN_SHIFT_SHIFT) & 0xffu;
            if (T > SCAN_MAX_T || (cnt & 1u) || cnt == 0 || iters > cnt / 2 || sh >= 254u || (h & 0x7f000u)) return bad("bundle " + std::to_string(b) + ": scan bundle");
        }
        // posts and waits are C_SYNC bundles without nodes: stream 0 posts once, every other stream waits in its first bundle
        // (nothing else is compiled)
        if (((h & (HDR_POST | HDR_WAIT)) != 0) != (cls == C_SYNC) || (cls == C_SYNC && cnt != 0)) return bad("bundle " + std::to_string(b) + ": post / wait bits");
        if ((h & HDR_POST) && !(NS > 1 && stream == 0 && executed && n_posts++ == 0)) return bad("bundle " + std::to_string(b) + ": post");
        if (((h & HDR_WAIT) != 0) != (NS > 1 && stream != 0 && executed && b == p.stream_first[stream])) return bad("bundle " + std::to_string(b) + ": wait");
        // the staging loads of the two bundles behind a wait are issued in front of it: they must not read anything
        if (NS > 1 && stream != 0 && executed && (b == p.stream_first[stream] + 1 || b == p.stream_first[stream] + 2) && cnt != 0) return bad("bundle " + std::to_string(b) + ": work right behind a wait");
        if ((h & (HDR_A_CANON | HDR_B_CANON)) && !(cls == C_BIT || cls == C_IDIVMOD || cls == C_CMPS)) return bad("bundle " + std::to_string(b) + ": operand form bits");
        if ((h & HDR_OUT_CANON) && !(cls == C_BIT || cls == C_IDIVMOD || cls == C_CMPS || cls == C_CMPZ)) return bad("bundle " + std::to_string(b) + ": result form bit");
        const uint32_t rep = cls == C_MULQ || cls == C_MULF ? COOP_LANES : 1u;
        if ((cnt == 0 && cls != C_LIN && cls != C_SYNC) || cnt * rep > G) return bad("bundle " + std::to_string(b) + ": node count");
        if (cls == C_MULF && T > COOP_FUSE_MAX_T) return bad("bundle " + std::to_string(b) + ": fused bundle at this tile width");
        if (!executed && cnt != 0) return bad("bundle " + std::to_string(b) + ": outside every stream");
        if (cls == C_MULQ && T > COOP_MAX_T) return bad("bundle " + std::to_string(b) + ": narrow bundle at this tile width");
        if ((cls == C_DIVREQ || cls == C_DIVGET) && !p.divider) return bad("bundle " + std::to_string(b) + ": request / collect without a divider");
        if (cls == C_DIV && p.divider) return bad("bundle " + std::to_string(b) + ": inline division in a divider program");
        if (cls == C_DIVREQ) {
            if (in_flight || n_req >= p.n_div_requests || cnt * T > mbox_lanes(p.divider) || p.div_lanes[n_req] != cnt * T) return bad("bundle " + std::to_string(b) + ": division request");
            in_flight = true;
            ++n_req;
            ++stream_req;
        }
        if (cls == C_DIVGET) {
            if (!in_flight) return bad("bundle " + std::to_string(b) + ": collect without a request");
            in_flight = false;
            ++n_get;
        }
        for (uint32_t q = 0; q < G; ++q) {
            const uint32_t* r = &p.recs[((size_t)b * G + q) * 4];
            // staging loads: 16 bytes per lane at off + 16 t and at off + HI + 16 t
            for (int k = 0; k < 2; ++k)
                if ((r[k] % slot_bytes) != 0 || (uint64_t)r[k] + slot_bytes > tile_bytes) return bad("bundle " + std::to_string(b) + ": operand offset");
            if (cls == C_MULF) {  // stage codes: op2 in the main records (even positions), op3 (additions only) in the extra records
                const uint32_t code = r[2] & CTRL_SUB_MASK;
                if ((q & 1u) ? (code == FOP_MUL || code > FOP_RSUB || (r[2] & ~CTRL_MASK) != trash_off) : code > FOP_RSUB) return bad("bundle " + std::to_string(b) + ": fused stage code");
                if ((code == FOP_MUL && !(h & HDR_F_S2MUL)) || (code > FOP_MUL && !(h & ((q & 1u) ? HDR_F_S3LIN : HDR_F_S2LIN)))) return bad("bundle " + std::to_string(b) + ": fused stage bits");
            }
            if (cls == C_SCAN && q < cnt) {  // position 2p: the step's OUT record, 2p + 1: its ACC record, same START bit; the first pair starts a chain
                const uint32_t sub = r[2] & CTRL_SUB_MASK, sub0 = p.recs[((size_t)b * G + (q & ~1u)) * 4 + 2] & CTRL_SUB_MASK;
                if ((sub & SCAN_ROLE_ACC) != (q & 1u) || (sub & ~(SCAN_ROLE_ACC | SCAN_START)) || ((sub ^ sub0) & SCAN_START) || (q < 2 && !(sub & SCAN_START)) || !(r[2] & CTRL_ACTIVE))
                    return bad("bundle " + std::to_string(b) + ": scan record");
            }
            const uint32_t dst = r[2] & ~CTRL_MASK;
            if ((dst % slot_bytes) != 0 || dst < (uint64_t)p.n_const * slot_bytes || dst > trash_off) return bad("bundle " + std::to_string(b) + ": destination");
            const uint32_t la = r[3] & 0xffffu, lb = r[3] >> 16;
            const bool bitx = cls == C_BIT && (r[2] & CTRL_SUB_MASK) == SUB_BITX;
            if ((la % 16) != 0 || la + 16u * (T - 1) + LDS_HALF_BYTES + 16u > LDS_BYTES) return bad("bundle " + std::to_string(b) + ": LDS address");
            // (a bit-extract lane carries its shift amount there; idle lanes of such a bundle keep a stage address, unused)
            const bool active = (r[2] & CTRL_ACTIVE) != 0;
            if (bitx ? (active && lb / 16 >= 254) : ((lb % 16) != 0 || lb + 16u * (T - 1) + LDS_HALF_BYTES + 16u > LDS_BYTES)) return bad("bundle " + std::to_string(b) + ": LDS address");
            const bool has_row = cls == C_INPUT || cls == C_TERN;
            if (has_row && cref_row >= p.n_cref_rows) return bad("bundle " + std::to_string(b) + ": third-operand row");
            const uint32_t cr = has_row ? p.crefs[(size_t)cref_row * G + q] : 0u;
            if (cls == C_INPUT && cr >= p.n_inputs) return bad("bundle " + std::to_string(b) + ": input index");
            if (cls == C_TERN && ((cr % slot_bytes) != 0 || (uint64_t)cr + slot_bytes > tile_bytes)) return bad("bundle " + std::to_string(b) + ": third operand");
        }
        (void)HI;
        cref_row += cls == C_INPUT || cls == C_TERN;
    }
    if (cref_row != p.n_cref_rows) return bad("third-operand rows");
    bool any_fused = false, any_scan = false;  // (one interpreter instance each: a program has one kind or the other)
    for (uint32_t h : p.hdr) {
        any_fused = any_fused || (h & HDR_CLASS_MASK) == C_MULF;
        any_scan = any_scan || (h & HDR_CLASS_MASK) == C_SCAN || ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC));
        if ((h & HDR_CLASS_MASK) == C_MUL && (h & HDR_MUL_CC) && (T > SCAN_MAX_T || (h & (HDR_LIN_ADD | HDR_LIN_SUB)))) return bad("canonical-product bundle");
    }
    if (any_fused && any_scan) return bad("fused and scan / canonical-product bundles in one program");
    if (in_flight || n_req != p.n_div_requests || n_get != n_req || stream_req != p.stream_div_requests[stream]) return bad("division requests");
    if (NS > 1 && n_posts != 1) return bad("streams without a post");
    for (uint32_t w : p.witness_refs)
        if ((w & REF_CONST) ? (w & ~REF_CONST) >= p.n_const : (w & ~REF_CANON) >= p.n_slots) return bad("witness reference");
    return true;
}

// ---- blob ----------------------------------------------------------------------------------------
static const uint32_t kBlobMagic = 0x47505743u;  // "CWPG"
struct BlobHeader {
    uint32_t magic, version, T, G, n_bundles, n_slots, n_const, n_inputs, n_witness, divider, n_div_requests, n_streams;
    uint32_t stream_first[MAX_STREAMS], stream_count[MAX_STREAMS], stream_div_requests[MAX_STREAMS], stream_cref_first[MAX_STREAMS];
    uint32_t n_cref_rows, reserved;
    double stream_cycles[MAX_STREAMS], stream_cycles_mul_div[MAX_STREAMS], stream_chain_cycles[MAX_STREAMS];
    ProgramStats stats;
};

size_t program_blob_size(const Program& p) {
    return sizeof(BlobHeader) + 4 * (p.hdr.size() + p.recs.size() + p.crefs.size() + p.consts.size() + p.witness_refs.size() + p.div_lanes.size());
}

// the blob at dst (program_blob_size(p) bytes): what gwb_graph_export writes straight into the caller's buffer
void program_blob_write(const Program& p, uint8_t* dst) {
    BlobHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.version = 16;  // (16: scan bundles, class 14, in place of round 3's macro bundles; one more statistics word.  15: blob_checksum in the image's trailer)
    h.T = p.T; h.G = p.G; h.n_bundles = p.n_bundles; h.n_slots = p.n_slots; h.n_const = p.n_const;
    h.n_inputs = p.n_inputs; h.n_witness = p.n_witness;
    h.divider = p.divider; h.n_div_requests = p.n_div_requests;
    h.n_streams = p.n_streams;
    h.n_cref_rows = p.n_cref_rows;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        h.stream_first[s] = p.stream_first[s]; h.stream_count[s] = p.stream_count[s]; h.stream_div_requests[s] = p.stream_div_requests[s]; h.stream_cref_first[s] = p.stream_cref_first[s];
        h.stream_cycles[s] = p.stream_cycles[s]; h.stream_cycles_mul_div[s] = p.stream_cycles_mul_div[s]; h.stream_chain_cycles[s] = p.stream_chain_cycles[s];
    }
    h.stats = p.stats;
    memcpy(dst, &h, sizeof h);
    dst += sizeof h;
    auto put = [&](const std::vector<uint32_t>& v) {
        if (!v.empty()) memcpy(dst, v.data(), 4 * v.size());
        dst += 4 * v.size();
    };
    put(p.hdr); put(p.recs); put(p.crefs); put(p.consts); put(p.witness_refs); put(p.div_lanes);
}

std::vector<uint8_t> program_to_blob(const Program& p) {
    std::vector<uint8_t> out(program_blob_size(p));
    program_blob_write(p, out.data());
    return out;
}

bool program_from_blob(const uint8_t* data, size_t len, Program& p, std::string& err) {
    BlobHeader h;
    if (len < sizeof h) { err = "program blob too short"; return false; }
    memcpy(&h, data, sizeof h);
    if (h.magic != kBlobMagic || h.version != 16 || h.T == 0 || h.T > 64 || h.G != 64 / h.T || (h.divider != 0 && h.divider != 1 && h.divider != 3 && h.divider != 4)) { err = "bad program blob header"; return false; }
    p = Program();
    p.T = h.T; p.G = h.G; p.n_bundles = h.n_bundles; p.n_slots = h.n_slots; p.n_const = h.n_const;
    p.n_inputs = h.n_inputs; p.n_witness = h.n_witness; p.stats = h.stats;
    p.divider = h.divider; p.n_div_requests = h.n_div_requests;
    p.n_streams = h.n_streams;
    p.n_cref_rows = h.n_cref_rows;
    for (uint32_t s = 0; s < MAX_STREAMS; ++s) {
        p.stream_first[s] = h.stream_first[s]; p.stream_count[s] = h.stream_count[s]; p.stream_div_requests[s] = h.stream_div_requests[s]; p.stream_cref_first[s] = h.stream_cref_first[s];
        p.stream_cycles[s] = h.stream_cycles[s]; p.stream_cycles_mul_div[s] = h.stream_cycles_mul_div[s]; p.stream_chain_cycles[s] = h.stream_chain_cycles[s];
    }
    const size_t n_hdr = p.n_bundles, n_recs = (size_t)p.n_bundles * p.G * 4, n_c = (size_t)p.n_cref_rows * p.G,
                 n_k = (size_t)p.n_const * 8, n_w = p.n_witness, n_d = p.n_div_requests;
    if (len != sizeof h + 4 * (n_hdr + n_recs + n_c + n_k + n_w + n_d)) { err = "program blob size mismatch"; return false; }
    const uint32_t* q = (const uint32_t*)(data + sizeof h);
    p.hdr.assign(q, q + n_hdr); q += n_hdr;
    p.recs.assign(q, q + n_recs); q += n_recs;
    p.crefs.assign(q, q + n_c); q += n_c;
    p.consts.assign(q, q + n_k); q += n_k;
    p.witness_refs.assign(q, q + n_w); q += n_w;
    p.div_lanes.assign(q, q + n_d);
    return true;
}

}  // namespace cwc
