// Exact rewrites of a loaded graph, in pipeline order: power-of-two divisions, bit-extract fusion, tree-height
// reduction with shared subexpressions, one form per value (Montgomery / canonical), scan chains (the steps of serial limb
// recurrences), fused narrow chains.  Every pass keeps the witness values; nothing that can fail is dropped or reordered.
#include "compile_internal.hpp"

namespace cwc {

// Exact strength reduction done before scheduling: Idiv(x, 2^k) == Shr(x, k) and Mod(x, 2^k) == Band(x, 2^k - 1) on the
// canonical integers the reference divides (src/graph.rs:112-121 vs :637-672, :674-687), for every x < r and k <= 253.
// The replacement constants are appended behind the last node (constants have no dependencies).
void rewrite_pow2_divisions(Graph& g) {
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
void fuse_bit_extract(Graph& g) {
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

// Tree-height reduction, exact in the field: Add and Mul are associative and commutative, so a node at the end of a
// chain of the same operation (a linear combination `lc += c_j * x_j`, or c * (x^4 * x)) may be computed from the
// chain's leaves in any order.  A wave's time is the sum of its bundles and the bundle count follows the longest
// dependency chain, so every node whose own chain is its critical input is rebuilt as a tree over the leaves, cheapest
// and earliest-ready first: sum chains of n terms drop from n-1 to ceil(log2 n) levels, and a constant factor is folded
// into the early part of a product (M_ji * x^5 becomes (M_ji * x) * x^4, one multiplication level less per Poseidon
// round).  The intermediate nodes of the chain are still computed wherever something else (a witness element, another
// node) needs them; common subexpressions are shared; nodes that end up unused are dropped.
// Only Add/Mul nodes are touched, so every operation that can fail (graph.rs:634, :686-716) survives unchanged.
void reduce_tree_height(Graph& g, size_t kMaxLeaves, const uint32_t* class_cost) {
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
// canonical_inputs (bit graphs: circuits that compute on bits, sha256-like -- a share of their operations are bit extracts):
// the Input bundles keep the canonical integer (x mod r) instead of its Montgomery form, so that everything downstream is
// canonical, and a product with a constant is a canonical product too (the canonical copy of the constant).  Bits and small
// signed combinations of bits multiply as integers in the kernel; other values take the general path.
// allow_cc (limb-arithmetic graphs, tile widths with the MODE 2 interpreter instances): the product of two canonical values
// stays a node of its own kind -- both factors canonical, result canonical (VF_MUL_CC) -- instead of converting one factor:
// limb products are far below r, and the kernel multiplies limb-sized integers directly (general operands: two Montgomery
// products).
void infer_representations(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, uint64_t& n_conversions, uint64_t& n_canonical, bool all_montgomery,
                                  bool allow_cc, uint64_t& n_cc, bool canonical_inputs) {
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
                    if (r == REP_C && allow_cc && canonical_inputs) {  // ... or, in bit graphs, a canonical product with the constant's canonical copy
                        f |= VF_MUL_CC;
                        ++n_cc;
                    }
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
        if (n.kind == N_INPUT && canonical_inputs && !off) {
            r = REP_C;
            f = VF_OUT_CANON;
        }
        orep[i] = r;
        n_canonical += ar && r == REP_C;
        at[i] = emit(n, r, f);
    }
    for (uint32_t& w : g.witness_signals) w = at[w];
    g.nodes.swap(out);
}

// Drops the nodes marked dead and renumbers everything that names a node: operands, the witness list, a DIV step's constant
// 2^k (scan_imm), the partner / group node (scan_partner).
static void compact_dead(Graph& g, const std::vector<uint8_t>& dead, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm,
                         std::vector<uint32_t>& scan_partner) {
    static const uint32_t NONE = 0xffffffffu;
    const size_t N = g.nodes.size();
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
void detect_scans(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner,
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
    std::vector<uint32_t> user_out(N, NONE), user_acc(N, NONE), end_out(N, NONE), end_acc(N, NONE);
    for (size_t j = 0; j < N; ++j) {
        Node& n = g.nodes[j];
        if (n.kind != N_DUO) continue;
        // (value numbering orders the operands of commutative operations by index: the mask may come first)
        if (n.op == OP_BAND && g.nodes[n.a].kind == N_CONST && g.nodes[n.b].kind != N_CONST) {
            std::swap(n.a, n.b);
            vflags[j] = (uint8_t)((vflags[j] & ~(VF_A_CANON | VF_B_CANON)) | ((vflags[j] & VF_A_CANON) ? VF_B_CANON : 0) | ((vflags[j] & VF_B_CANON) ? VF_A_CANON : 0));
        }
        const uint8_t want = VF_A_CANON | VF_B_CANON | VF_OUT_CANON;
        if ((vflags[j] & want) != want) continue;
        if (g.nodes[n.a].kind != N_CONST) {  // (chain ends, below: the pair on any value)
            if (n.op == OP_BAND || n.op == OP_IDIV) end_out[n.a] = end_out[n.a] == NONE ? (uint32_t)j : NONE - 1;
            else if (n.op == OP_SHR || n.op == OP_MOD) end_acc[n.a] = end_acc[n.a] == NONE ? (uint32_t)j : NONE - 1;
        }
        if (g.nodes[n.a].kind != N_DUO || g.nodes[n.a].op != OP_ADD) continue;
        if (n.op == OP_BAND || n.op == OP_IDIV) user_out[n.a] = user_out[n.a] == NONE ? (uint32_t)j : NONE - 1;
        else if (n.op == OP_SHR || n.op == OP_MOD) user_acc[n.a] = user_acc[n.a] == NONE ? (uint32_t)j : NONE - 1;
    }
    struct Step { uint32_t t, out, acc, x, acc_in, d, imm; bool div; uint8_t ends = 0; };
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
    // Chain ends.  The first step of a carry chain whose incoming carry is the constant 0 has no Add node (t = x), nor has the step
    // behind the last column (t = the last carry); the first step of a remainder chain is t = 0 * 2^k + x = x.  Left alone they are
    // two-node bundles of their own (a Band / Shr pair, an Idiv / Mod pair) on the chain's critical path, ~2 k cycles each.  Such a
    // pair on one value becomes a step whose other operand is absent (read as 0) -- exact: x + 0 and 0 * 2^k + x are x -- where it
    // continues a chain or a chain continues it (its accumulator value is an operand of a regular step of the same kind, or its x
    // is a regular step's accumulator value).
    {
        std::vector<uint8_t> is_t(N, 0);
        std::vector<uint32_t> reg_of_acc(N, NONE), reg_of_operand(N, NONE);  // regular steps by their ACC node / by an operand (x or incoming accumulator)
        for (size_t k = 0; k < steps.size(); ++k) {
            is_t[steps[k].t] = 1;
            reg_of_acc[steps[k].acc] = (uint32_t)k;
            reg_of_operand[steps[k].x] = reg_of_operand[steps[k].acc_in] = (uint32_t)k;
        }
        const size_t n_regular = steps.size();
        if (getenv("CWC_DEBUG_SCAN")) {
            size_t pairs = 0, not_t = 0, can = 0, linked = 0;
            for (size_t x = 0; x < N; ++x) {
                if (end_out[x] >= NONE - 1 || end_acc[x] >= NONE - 1) continue;
                ++pairs;
                if (is_t[x]) continue;
                ++not_t;
                can += canon((uint32_t)x);
                linked += reg_of_acc[x] != NONE || reg_of_operand[end_acc[x]] != NONE;
            }
            fprintf(stderr, "chain ends: %zu values with an OUT / ACC pair, %zu besides the regular steps, %zu canonical, %zu linked to a regular step\n", pairs, not_t, can, linked);
        }
        const bool no_ends = getenv("CWC_NO_SCAN_ENDS") != nullptr;
        for (size_t x = 0; x < N && !no_ends; ++x) {
            const uint32_t o = end_out[x], a = end_acc[x];
            if (o >= NONE - 1 || a >= NONE - 1 || is_t[x] || !canon((uint32_t)x)) continue;
            const Node &O = g.nodes[o], &A = g.nodes[a];
            // the regular step this one continues (x is its accumulator value) or that continues this one (reads this step's accumulator value)
            const uint32_t before = reg_of_acc[x] < n_regular ? reg_of_acc[x] : NONE, after = reg_of_operand[a] < n_regular ? reg_of_operand[a] : NONE;
            const uint32_t link = before != NONE ? before : after;
            if (link == NONE) continue;
            if (O.op == OP_BAND && A.op == OP_SHR) {
                const int n = small_of(A.b);
                if (n < 0 || mask_of(O.b) != n || steps[link].div || steps[link].imm != (uint32_t)n) continue;
                steps.push_back(Step{NONE, o, a, (uint32_t)x, (uint32_t)x, 0, (uint32_t)n, false, (uint8_t)(before != NONE ? SCAN_OP_NOX : SCAN_OP_NOACC)});  // (x = the chain's carry: it is the accumulator)
            } else if (O.op == OP_IDIV && A.op == OP_MOD && O.b == A.b && canon(O.b)) {
                if (!steps[link].div || steps[link].d != O.b) continue;
                if (before != NONE) continue;  // (a remainder chain's last step always has its x)
                steps.push_back(Step{NONE, o, a, (uint32_t)x, (uint32_t)x, O.b, steps[link].imm, true, (uint8_t)SCAN_OP_NOACC});
            }
        }
    }
    // Carry-chain tails (round 5): the last step of a chain whose outgoing carry nothing reads has lost its Shr node to the load-time
    // optimiser's sweep -- only limb = (x + carry) & (2^n - 1) is there (the top register of a long_scalar_mult whose carry is dropped).
    // The sum, read by nothing else, becomes the step's ACC node (its value, the carry, is read by nothing): the step joins its chain's
    // bundle instead of costing an Add bundle and a Band bundle on the chain's critical path.
    if (!getenv("CWC_NO_SCAN_ENDS")) {
        std::vector<uint32_t> acc_of_regular(N, NONE);
        for (size_t k = 0; k < steps.size(); ++k)
            if (!steps[k].div) acc_of_regular[steps[k].acc] = (uint32_t)k;
        for (size_t t = 0; t < N; ++t) {
            const uint32_t o = user_out[t];
            if (o >= NONE - 1 || user_acc[t] != NONE || uses[t] != 1 || rep[t] != REP_C) continue;
            const Node& T_ = g.nodes[t];
            const Node& O = g.nodes[o];
            if (O.op != OP_BAND || !canon(T_.a) || !canon(T_.b)) continue;
            const int n = mask_of(O.b);
            // (one operand must be the carry of a regular step of the same width: a lone masked sum is left alone)
            const uint32_t la = acc_of_regular[T_.a], lb = acc_of_regular[T_.b];
            const bool a_is = la != NONE && (int)steps[la].imm == n, b_is = lb != NONE && (int)steps[lb].imm == n;
            if (n < 1 || (!a_is && !b_is)) continue;
            steps.push_back(Step{NONE, o, (uint32_t)t, b_is ? T_.a : T_.b, b_is ? T_.b : T_.a, 0, (uint32_t)n, false});
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
        const uint8_t kind = (uint8_t)((st.div ? SCAN_OP_DIV : 0) | st.ends);
        g.nodes[st.out] = Node{N_SCAN, kind, st.x, st.acc_in, st.d};
        g.nodes[st.acc] = Node{N_SCAN, (uint8_t)(kind | SCAN_OP_ACC), st.x, st.acc_in, st.d};
        vflags[st.out] = vflags[st.acc] = 0;
        if (st.t == NONE) continue;  // (a chain end: no inner nodes)
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
    compact_dead(g, dead, rep, vflags, scan_imm, scan_partner);  // (the dead inner nodes would be scheduled)
}

// ---- one-bit recurrences of multi-register integers (round 5) -------------------------------------------------------------
// The witness hints of big-integer circuits whose registers are wider than a machine word (circom-bigint as zk-email's RSA verifier
// uses it: 121-bit registers x 17; long_div by a k-register divisor) spend most of their dependent chain in two recurrences whose
// state is ONE BIT per register:
//   * the borrow chain of a register-wise subtraction (long_sub), per register, as the function's if / else predicated:
//         s = y + bin;  c = x >= s;  diff = c ? x - y - bin : 2^n + x - y - bin;  bout = c ? 0 : 1
//   * the comparison decided by the most significant differing register (long_gt), least significant register first:
//         res = x > y ? 1 : (x < y ? 0 : res)
// Unfused a register costs three (two) bundles on the graph's critical chain.  A step becomes a PAIR of N_SCAN nodes like the
// steps of detect_scans -- BORROW: OUT = diff, ACC = bout; LEX: ACC = the outer selection, OUT = the inner one (read by nothing) --
// and the kernel runs all steps of a bundle at once: both are carry chains (generate / propagate per register: x < y / x == y;
// x != y with the winning constant / x == y), a carry-lookahead over the wave resolves them (scan_gfx950.hpp scan_bit_lookahead).
// Recognition is by VALUE, not by shape: the arms of the difference are expanded to linear forms over the step's atoms (x, y, bin)
// and compared with x - y - bin and x - y - bin + 2^n, whatever the front-end's association or the load-time optimiser's folding
// (y = 0: `0 + bin`, `x - 0` are gone) made of them.  Exact: the kernel's step is the same field additions / subtractions, the
// same signed comparisons (graph.rs:110-111, 130-133, 723-769) and selections (:221-225) on the same canonical integers;
// nothing that can fail is involved.  Inner nodes that something else reads stay; the rest is removed by a sweep of unused
// pure nodes.  scan_imm[node]: BORROW the register width n, LEX 0.
namespace {
struct LinForm {  // sum of coeff * node + k (field constant), at most 6 terms
    std::pair<uint32_t, int> t[6];
    int n = 0;
    Fr k = fr_zero();
    bool ok = true;
    void add_term(uint32_t node, int c) {
        for (int i = 0; i < n; ++i)
            if (t[i].first == node) {
                t[i].second += c;
                if (!t[i].second) t[i] = t[--n];
                return;
            }
        if (n == 6) { ok = false; return; }
        t[n++] = {node, c};
    }
    void add_const(const Fr& v, int sign) { k = sign > 0 ? fr_add(k, v) : fr_sub(k, v); }
};
}  // namespace

void detect_bit_scans(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner, uint64_t& n_steps) {
    const size_t N = g.nodes.size();
    static const uint32_t NONE = 0xffffffffu;
    if (getenv("CWC_NO_BIT_SCANS")) return;
    if (scan_imm.size() != N) scan_imm.assign(N, 0);
    if (scan_partner.size() != N) scan_partner.assign(N, NONE);
    // users of every comparison node (CSR over the nodes that are candidates for a condition)
    std::vector<uint32_t> uses(N, 0), first_user(N + 1, 0), user_list;
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) uses[ops[q]]++;
    }
    {
        for (size_t i = 0; i < N; ++i) first_user[i + 1] = first_user[i] + uses[i];
        user_list.assign(first_user[N], 0);
        std::vector<uint32_t> fill(first_user.begin(), first_user.end() - 1);
        for (size_t i = 0; i < N; ++i) {
            const Node& n = g.nodes[i];
            const uint32_t ops[3] = {n.a, n.b, n.c};
            for (int q = 0; q < arity_of(n); ++q) user_list[fill[ops[q]]++] = (uint32_t)i;
        }
    }
    std::vector<uint32_t> wit_uses(N, 0);
    for (uint32_t w : g.witness_signals) wit_uses[w]++;
    auto const_value = [&](uint32_t idx) -> const Fr* { return g.nodes[idx].kind == N_CONST ? &g.const_values[g.nodes[idx].a] : nullptr; };
    auto const_small = [&](uint32_t idx) -> int {  // 0 / 1 -> that, else -1
        const Fr* v = const_value(idx);
        if (!v) return -1;
        for (int w = 1; w < 8; ++w)
            if (v->v[w]) return -1;
        return v->v[0] <= 1u ? (int)v->v[0] : -1;
    };
    auto pow2_value = [&](const Fr& v) -> int {
        int k = -1, bits = 0;
        for (int w = 0; w < 8; ++w)
            if (v.v[w]) {
                bits += __builtin_popcount(v.v[w]);
                k = 32 * w + __builtin_ctz(v.v[w]);
            }
        return bits == 1 ? k : -1;
    };
    auto canon = [&](uint32_t o) { return g.nodes[o].kind == N_CONST || rep[o] == REP_C; };
    auto is_cmp = [&](uint32_t i, uint8_t op) { return g.nodes[i].kind == N_DUO && g.nodes[i].op == op; };
    // linear form of `root` over atoms: Add / Sub / Neg nodes are expanded (not the stop nodes, not beyond `budget` nodes)
    auto expand = [&](uint32_t root, uint32_t s0, uint32_t s1, uint32_t s2) -> LinForm {
        LinForm f;
        std::pair<uint32_t, int> stack[24];
        int sp = 0, budget = 16;
        stack[sp++] = {root, 1};
        while (sp && f.ok) {
            const auto [i, sg] = stack[--sp];
            const Node& n = g.nodes[i];
            if (n.kind == N_CONST) {
                f.add_const(g.const_values[n.a], sg);
            } else if (i != s0 && i != s1 && i != s2 && budget > 0 && sp + 2 <= 24 && ((n.kind == N_DUO && (n.op == OP_ADD || n.op == OP_SUB)) || n.kind == N_UNO)) {
                --budget;
                if (n.kind == N_UNO) {
                    if (n.op != 0) { f.ok = false; break; }  // (Neg only)
                    stack[sp++] = {n.a, -sg};
                } else {
                    stack[sp++] = {n.a, sg};
                    stack[sp++] = {n.b, n.op == OP_ADD ? sg : -sg};
                }
            } else {
                f.add_term(i, sg);
            }
        }
        return f;
    };
    // `f` == x - y - bin + extra with extra a constant: returns true and that constant
    auto matches_difference = [&](LinForm f, uint32_t x, uint32_t y, uint32_t bin, Fr& extra) -> bool {
        if (!f.ok) return false;
        auto take = [&](uint32_t node, int sign) {  // f -= sign * node
            if (node == NONE) return;
            if (const Fr* v = const_value(node)) f.add_const(*v, -sign);
            else f.add_term(node, -sign);
        };
        take(x, 1);
        take(y, -1);
        take(bin, -1);
        if (!f.ok || f.n != 0) return false;
        extra = f.k;
        return true;
    };
    struct Step { uint32_t out, acc, x, acc_in, y, imm; uint8_t op; };
    std::vector<Step> steps;
    std::vector<uint8_t> bit_acc(N, 0);  // 1: the ACC node of a BORROW step, 2: of a LEX step (in node order: a step's incoming bit is an earlier step's)
    std::vector<uint8_t> taken(N, 0);
    size_t cand_borrow = 0, cand_lex = 0;
    for (size_t j = 0; j < N; ++j) {
        const Node& n = g.nodes[j];
        if (n.kind != N_TRES || taken[j]) continue;
        const int kb = const_small(n.b), kc = const_small(n.c);
        // ---- BORROW: bout = Tern(x >= s, 0, 1) (or Tern(x < s, 1, 0)) and diff = Tern(the same condition, x - y - bin, x - y - bin + 2^n)
        if (kb >= 0 && kc >= 0 && kb != kc && g.nodes[n.a].kind == N_DUO && rep[j] == REP_C) {
            const Node& C = g.nodes[n.a];
            uint32_t x = NONE, s = NONE;
            bool cond_is_geq = true;  // the condition holds when NO borrow leaves
            if (C.op == OP_GEQ) { x = C.a; s = C.b; }
            else if (C.op == OP_LEQ) { x = C.b; s = C.a; }
            else if (C.op == OP_LT) { x = C.a; s = C.b; cond_is_geq = false; }
            else if (C.op == OP_GT) { x = C.b; s = C.a; cond_is_geq = false; }
            if (x != NONE && (cond_is_geq ? (kb == 0 && kc == 1) : (kb == 1 && kc == 0))) {
                ++cand_borrow;
                // s = y + bin, bin the ACC node of an earlier BORROW step
                uint32_t y = s, bin = NONE;
                const uint32_t zero_node = cond_is_geq ? n.b : n.c;  // (a constant 0 that exists)
                if (bit_acc[s] == 1) { bin = s; y = zero_node; }
                else if (g.nodes[s].kind == N_DUO && g.nodes[s].op == OP_ADD) {
                    if (bit_acc[g.nodes[s].a] == 1) { bin = g.nodes[s].a; y = g.nodes[s].b; }
                    else if (bit_acc[g.nodes[s].b] == 1) { bin = g.nodes[s].b; y = g.nodes[s].a; }
                }
                if (canon(x) && canon(y)) {
                    for (uint32_t q = first_user[n.a]; q < first_user[n.a + 1]; ++q) {
                        const uint32_t u = user_list[q];
                        const Node& D = g.nodes[u];
                        if (u == j || taken[u] || D.kind != N_TRES || D.a != n.a || rep[u] != REP_C) continue;
                        const uint32_t arm_then = cond_is_geq ? D.b : D.c, arm_else = cond_is_geq ? D.c : D.b;
                        Fr e0, e1;
                        if (!matches_difference(expand(arm_then, x, y, bin), x, y, bin, e0) || !u256_is_zero(e0)) continue;
                        if (!matches_difference(expand(arm_else, x, y, bin), x, y, bin, e1)) continue;
                        const int nbits = pow2_value(e1);
                        if (nbits < 1 || nbits > 253) continue;
                        steps.push_back(Step{u, (uint32_t)j, x, bin == NONE ? x : bin, y, (uint32_t)nbits, (uint8_t)(SCAN_OP_BORROW | (bin == NONE ? SCAN_OP_NOACC : 0))});
                        taken[u] = taken[j] = 1;
                        bit_acc[j] = 1;
                        break;
                    }
                }
                if (taken[j]) continue;
            }
        }
        // ---- BORROW, last register: the borrow that leaves is read by nothing, so its selection is gone (the load-time optimiser's sweep) and
        // only diff = Tern(x >= s, x - y - bin, x - y - bin + 2^n) is there.  The comparison, read by nothing else, becomes the step's ACC node.
        if (g.nodes[n.a].kind == N_DUO && rep[j] == REP_C && uses[n.a] == 1 && !wit_uses[n.a] && !taken[n.a]) {
            const Node& C = g.nodes[n.a];
            uint32_t x = NONE, s = NONE;
            bool cond_is_geq = true;
            if (C.op == OP_GEQ) { x = C.a; s = C.b; }
            else if (C.op == OP_LEQ) { x = C.b; s = C.a; }
            else if (C.op == OP_LT) { x = C.a; s = C.b; cond_is_geq = false; }
            else if (C.op == OP_GT) { x = C.b; s = C.a; cond_is_geq = false; }
            if (x != NONE) {
                uint32_t y = s, bin = NONE;
                if (bit_acc[s] == 1) { bin = s; y = NONE; }
                else if (g.nodes[s].kind == N_DUO && g.nodes[s].op == OP_ADD) {
                    if (bit_acc[g.nodes[s].a] == 1) { bin = g.nodes[s].a; y = g.nodes[s].b; }
                    else if (bit_acc[g.nodes[s].b] == 1) { bin = g.nodes[s].b; y = g.nodes[s].a; }
                }
                // (y = 0: the constant 0 is named through the chain's earlier step, whose selection held one -- a chain of one register is left alone)
                if (y == NONE && bin != NONE) {
                    const Node& pb = g.nodes[bin];  // (still the Tern(c, 0, 1) / Tern(c, 1, 0) it was recognised as: rewritten behind this loop)
                    y = const_small(pb.b) == 0 ? pb.b : pb.c;
                }
                if (bin != NONE && y != NONE && canon(x) && canon(y)) {
                    const uint32_t arm_then = cond_is_geq ? n.b : n.c, arm_else = cond_is_geq ? n.c : n.b;
                    Fr e0, e1;
                    if (matches_difference(expand(arm_then, x, y, bin), x, y, bin, e0) && u256_is_zero(e0) && matches_difference(expand(arm_else, x, y, bin), x, y, bin, e1)) {
                        const int nbits = pow2_value(e1);
                        if (nbits >= 1 && nbits <= 253) {
                            ++cand_borrow;
                            steps.push_back(Step{(uint32_t)j, n.a, x, bin, y, (uint32_t)nbits, (uint8_t)SCAN_OP_BORROW});
                            taken[j] = taken[n.a] = 1;
                            bit_acc[n.a] = 1;
                            continue;
                        }
                    }
                }
            }
        }
        // ---- LEX, first register: `x > y ? K1 : K0` is what the load-time optimiser leaves of `x > y ? K1 : (x < y ? K0 : K0)`: a step whose bit
        // coming in is the constant K0.  The comparison, read by nothing else, becomes the step's OUT node (read by nothing).
        if (kb >= 0 && kc >= 0 && (is_cmp(n.a, OP_GT) || is_cmp(n.a, OP_LT)) && uses[n.a] == 1 && !wit_uses[n.a] && !taken[n.a]) {
            const Node& C1 = g.nodes[n.a];
            if (C1.a != C1.b && canon(C1.a) && canon(C1.b)) {
                ++cand_lex;
                const int kg = C1.op == OP_GT ? kb : kc, kl = C1.op == OP_GT ? kc : kb;
                const uint8_t op = (uint8_t)(SCAN_OP_LEX | (kg ? SCAN_OP_KG : 0) | (kl ? SCAN_OP_KL : 0) | (kc == 0 ? SCAN_OP_NOACC : 0));
                steps.push_back(Step{n.a, (uint32_t)j, C1.a, kc == 0 ? C1.a : n.c, C1.b, rep[j] == REP_M ? 1u : 0u, op});
                taken[n.a] = taken[j] = 1;
                bit_acc[j] = 2;
                continue;
            }
        }
        // ---- LEX: outer = Tern(c1, K1, inner), inner = Tern(c2, K2, acc), c1 / c2 the two strict comparisons of one pair (x, y)
        if (kb >= 0 && g.nodes[n.c].kind == N_TRES && !taken[n.c] && uses[n.c] == 1 && !wit_uses[n.c] && rep[n.c] == rep[j]) {
            const Node& I = g.nodes[n.c];
            const int k2 = const_small(I.b);
            const bool c1_ok = is_cmp(n.a, OP_GT) || is_cmp(n.a, OP_LT), c2_ok = is_cmp(I.a, OP_GT) || is_cmp(I.a, OP_LT);
            if (k2 >= 0 && c1_ok && c2_ok) {
                ++cand_lex;
                const Node &C1 = g.nodes[n.a], &C2 = g.nodes[I.a];
                const uint32_t x = C1.a, y = C1.b;
                const int rel1 = C1.op == OP_GT ? 1 : -1;
                int rel2 = 0;
                if (C2.a == x && C2.b == y) rel2 = C2.op == OP_GT ? 1 : -1;
                else if (C2.a == y && C2.b == x) rel2 = C2.op == OP_GT ? -1 : 1;
                const int a0 = const_small(I.c);
                // the bit coming in: a constant 0 / 1, an earlier step's, or any boolean of the chain's form (a comparison's result, a selection
                // between 0 and 1 -- what is left of a chain's first step `x > y ? 1 : (x < y ? 0 : 0)` behind the load-time optimiser)
                auto boolean_node = [&](uint32_t o) {
                    const Node& b = g.nodes[o];
                    if (b.kind == N_TRES) return const_small(b.b) >= 0 && const_small(b.c) >= 0;
                    return b.kind == N_DUO && b.op >= OP_EQ && b.op <= OP_LOR;
                };
                const bool acc_ok = a0 >= 0 || ((bit_acc[I.c] == 2 || boolean_node(I.c)) && rep[I.c] == rep[j]);
                if (rel2 && rel2 != rel1 && x != y && canon(x) && canon(y) && acc_ok) {
                    const int kg = rel1 > 0 ? kb : k2, kl = rel1 > 0 ? k2 : kb;
                    const uint8_t op = (uint8_t)(SCAN_OP_LEX | (kg ? SCAN_OP_KG : 0) | (kl ? SCAN_OP_KL : 0) | (a0 == 0 ? SCAN_OP_NOACC : 0));
                    // (the chain's bits travel in the form representation inference gave the selections: the canonical 0 / 1, or -- selections
                    // between constants default to it -- the Montgomery form 0 / 2^256 mod r; a bundle's steps share it: imm = 1)
                    steps.push_back(Step{n.c, (uint32_t)j, x, a0 == 0 ? x : I.c, y, rep[j] == REP_M ? 1u : 0u, op});
                    taken[n.c] = taken[j] = 1;
                    bit_acc[j] = 2;
                }
            }
        }
    }
    // ---- selections (SCAN_OP_SEL), in programs that hold the kinds above anyway (they run in the MODE 3 interpreter instances): a
    // TernCond whose condition is an ordered comparison that nothing else reads goes through a pair of records with it -- the comparison's
    // operands in the OUT record, the arms in the ACC record, all four staged like any operand: one bundle instead of a comparison bundle
    // and a selection bundle with its two dependent global loads for the third operand.  Every other TernCond as well, its condition tested
    // against zero (CWC_NO_SEL_NEZ=1 leaves those to the TernCond class).
    // Exact: the same comparison (graph.rs:130-133, 723-769) and the same selection (:221-225) on the same values.
    struct Sel { uint32_t out, acc, a, b, p, q, imm; bool nez; };
    std::vector<Sel> sels;
    static const bool sel_always = getenv("CWC_SEL_ALWAYS") && atoi(getenv("CWC_SEL_ALWAYS")) != 0;
    // (knobs read once, not per TernCond node: 1.6 M of them in the ten-million-node RSA graph, each getenv a scan of the environment)
    const bool no_sel_cmp = getenv("CWC_NO_SEL_CMP") != nullptr, no_sel_nez = getenv("CWC_NO_SEL_NEZ") != nullptr;
    if ((!steps.empty() || sel_always) && !getenv("CWC_NO_SEL_SCANS")) {
        for (size_t j = 0; j < N; ++j) {
            const Node& n = g.nodes[j];
            if (n.kind != N_TRES || taken[j]) continue;
            const Node& C = g.nodes[n.a];
            const bool ordered = C.kind == N_DUO && (C.op == OP_LT || C.op == OP_GT || C.op == OP_LEQ || C.op == OP_GEQ);
            const uint8_t want = VF_A_CANON | VF_B_CANON;
            if (ordered && !no_sel_cmp && uses[n.a] == 1 && !wit_uses[n.a] && !taken[n.a] && (vflags[n.a] & want) == want && canon(C.a) && canon(C.b)) {
                const uint32_t code = C.op == OP_LT ? SEL_LT : C.op == OP_GT ? SEL_GT : C.op == OP_LEQ ? SEL_LEQ : SEL_GEQ;
                sels.push_back(Sel{n.a, (uint32_t)j, C.a, C.b, n.b, n.c, code, false});
                taken[n.a] = taken[j] = 1;
            } else if (!no_sel_nez) {
                // (the scheduler's priorities must know that the step's ACC value waits for the OUT node's operands -- compile.cc, the heights
                // of a selection pair -- or whatever computes the condition is scheduled as if nothing waited for it: 562 bundles per
                // multiplication of the RSA-class graph instead of 495)
                // any other condition: its value is tested against zero; the step's OUT node is a new node behind the graph's last one (operands still precede their users)
                sels.push_back(Sel{NONE, (uint32_t)j, n.a, n.a, n.b, n.c, SEL_NEZ, true});
                taken[j] = 1;
            }
        }
    }
    if (getenv("CWC_DEBUG_SCAN")) {
        size_t nb = 0;
        for (const Step& st : steps) nb += (st.op & SCAN_OP_BORROW) != 0;
        fprintf(stderr, "selections with their comparison: %zu\n", sels.size());
        fprintf(stderr, "bit scans: %zu borrow candidates, %zu comparison candidates; steps: %zu borrow, %zu comparison\n", cand_borrow, cand_lex, nb, steps.size() - nb);
    }
    if (steps.empty() && sels.empty()) return;
    for (const Step& st : steps) {
        g.nodes[st.out] = Node{N_SCAN, st.op, st.x, st.acc_in, st.y};
        g.nodes[st.acc] = Node{N_SCAN, (uint8_t)(st.op | SCAN_OP_ACC), st.x, st.acc_in, st.y};
        vflags[st.out] = vflags[st.acc] = 0;
        scan_imm[st.out] = scan_imm[st.acc] = st.imm;
        scan_partner[st.out] = st.acc;
        scan_partner[st.acc] = st.out;
    }
    n_steps += steps.size();
    for (Sel& st : sels) {
        if (st.nez) {
            st.out = (uint32_t)g.nodes.size();
            g.nodes.push_back(Node{N_SCAN, (uint8_t)(SCAN_OP_SEL | SCAN_OP_NOACC), st.a, st.a, st.a});
            rep.push_back(REP_C);
            vflags.push_back(0);
            scan_imm.push_back(0);
            scan_partner.push_back(NONE);
        }
        g.nodes[st.out] = Node{N_SCAN, (uint8_t)(SCAN_OP_SEL | (st.nez ? SCAN_OP_NOACC : 0)), st.a, st.b, st.a};
        g.nodes[st.acc] = Node{N_SCAN, (uint8_t)(SCAN_OP_SEL | SCAN_OP_ACC), st.p, st.q, st.p};
        vflags[st.out] = vflags[st.acc] = 0;
        scan_imm[st.out] = scan_imm[st.acc] = st.imm;
        scan_partner[st.out] = st.acc;
        scan_partner[st.acc] = st.out;
    }
    n_steps += sels.size();
    // sweep: pure nodes that nothing reads any more (the arms, the conditions, the sums under them); a step's nodes stay as a pair
    const size_t N2 = g.nodes.size();  // (with the OUT nodes of the plain selections)
    std::vector<uint32_t> live_uses(N2, 0);
    for (size_t i = 0; i < N2; ++i) {
        const Node& n = g.nodes[i];
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) live_uses[ops[q]]++;
    }
    for (uint32_t w : g.witness_signals) live_uses[w]++;
    std::vector<uint8_t> dead(N2, 0);
    bool any_dead = false;
    for (size_t i = N2; i-- > 0;) {
        const Node& n = g.nodes[i];
        if (live_uses[i]) continue;
        bool pure = false;
        if (n.kind == N_UNO || n.kind == N_TRES) pure = true;
        else if (n.kind == N_DUO) pure = n.op != OP_SHL && n.op != OP_BOR && n.op != OP_BXOR && n.op != OP_BAND;  // (what the load-time optimiser keeps as able to report, optimize.cc can_fail)
        if (!pure) continue;
        dead[i] = 1;
        any_dead = true;
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) live_uses[ops[q]]--;
    }
    if (any_dead) compact_dead(g, dead, rep, vflags, scan_imm, scan_partner);
}

// ---- limb products as convolutions (round 4) -----------------------------------------------------------------------------
// A schoolbook product of two k-limb integers is k^2 limb products and, per column c, the sum of the products with
// i + j = c: in the 10.5 M-node bigint-class graph 16 full bundles of canonical products and a tree of ~19 Add bundles per
// round, a third of its time once the scan chains run in parallel.  Where the products of a complete k x k block are each
// read once, by the Add tree of their column, and nothing else reads the trees' inner nodes, the 2k - 1 column sums become
// N_CONV nodes (graph.hpp) that ONE bundle computes: lane c keeps y_c, x_i is broadcast round after round, the y's move up
// the wave, every lane accumulates x_i y_(c-i) -- k multiply-accumulates per lane instead of k^2 products through memory.
// Exact: the same products and sums in the field on canonical integers (graph.rs:105, 110), in another order.
// Recognition: the leaves of every maximal single-use Add tree over canonical products; the products as edges between their
// factors must form a complete bipartite graph X x Y with |X| = |Y| = k, X and Y disjoint, and the trees must be exactly its
// anti-diagonals: with x_0 y_0 the lone product of a one-leaf tree, j(y) = leaves(tree of x_0 y) - 1 and i(x) likewise, every
// leaf x y of a tree then has the same i(x) + j(y), and the 2k - 1 trees have distinct columns.
void detect_convolutions(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, std::vector<uint32_t>& scan_imm, std::vector<uint32_t>& scan_partner,
                         uint32_t max_columns, uint64_t& n_products) {
    const size_t N = g.nodes.size();
    static const uint32_t NONE = 0xffffffffu;
    if (scan_imm.size() != N) scan_imm.assign(N, 0);
    if (scan_partner.size() != N) scan_partner.assign(N, NONE);
    std::vector<uint32_t> uses(N, 0), user(N, NONE);
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        const uint32_t ops[3] = {n.a, n.b, n.c};
        for (int q = 0; q < arity_of(n); ++q) {
            uses[ops[q]]++;
            user[ops[q]] = (uint32_t)i;
        }
    }
    for (uint32_t w : g.witness_signals) uses[w] += 2;  // (a witness element is never an inner node)
    auto is_product = [&](uint32_t i) {
        const Node& n = g.nodes[i];
        return n.kind == N_DUO && n.op == OP_MUL && (vflags[i] & VF_MUL_CC) && rep[i] == REP_C && g.nodes[n.a].kind != N_CONST && g.nodes[n.b].kind != N_CONST && n.a != n.b;
    };
    auto is_sum = [&](uint32_t i) {
        const Node& n = g.nodes[i];
        return n.kind == N_DUO && n.op == OP_ADD && rep[i] == REP_C;  // (a sum of two canonical values: infer_representations gives a linear node's operands its own form)
    };
    // the tree above every product / sum: a node is inner while it is read once, by a sum
    auto inner = [&](uint32_t i) { return uses[i] == 1 && user[i] != NONE && is_sum(user[i]); };
    auto find_root = [&](uint32_t i) -> uint32_t {  // (the height reduction keeps sum trees shallow; a tree deeper than this is left alone)
        for (int depth = 0; inner(i); ++depth) {
            if (depth > 96) return NONE;
            i = user[i];
        }
        return i;
    };
    std::vector<uint32_t> root_of(N, NONE);  // product -> the root of its tree (itself: a lone product)
    std::vector<uint32_t> leaves(N, 0);      // root -> number of products below it; NONE: the tree has another kind of leaf
    for (size_t i = 0; i < N; ++i) {
        const uint32_t u = (uint32_t)i;
        if (is_product(u)) {
            root_of[u] = find_root(u);
            if (root_of[u] != NONE && leaves[root_of[u]] != NONE) leaves[root_of[u]]++;
        } else if (is_sum(u)) {
            const Node& n = g.nodes[u];
            bool clean = n.a != n.b;
            for (uint32_t o : {n.a, n.b}) clean = clean && (is_product(o) || is_sum(o)) && inner(o);
            if (!clean) {
                const uint32_t r = find_root(u);
                if (r != NONE) leaves[r] = NONE;
            }
        }
    }
    // factor -> its products
    std::unordered_map<uint32_t, std::vector<uint32_t>> adj;
    for (size_t i = 0; i < N; ++i)
        if (is_product((uint32_t)i) && root_of[i] != NONE && leaves[root_of[i]] != NONE) {
            adj[g.nodes[i].a].push_back((uint32_t)i);
            adj[g.nodes[i].b].push_back((uint32_t)i);
        }
    auto other = [&](uint32_t prod, uint32_t f) { return g.nodes[prod].a == f ? g.nodes[prod].b : g.nodes[prod].a; };
    // Only limbs that are known to fit 64 bits: the bundle's multiply-accumulate rounds are for such factors, and its rounds for
    // any factor (field products) cost more than the unfused block's bundles (k x two Montgomery products against k^2 / 64
    // bundles of them).  Known: a Band with a constant, the limb a carry chain's step leaves (t mod 2^n), a constant.
    // CWC_CONV_ANY_WIDTH=1 (tests of the field-arithmetic rounds) lifts the rule.
    const bool any_width = getenv("CWC_CONV_ANY_WIDTH") != nullptr;
    auto const_bits = [&](uint32_t i) -> uint32_t {
        const Fr& v = g.const_values[g.nodes[i].a];
        for (int w = 7; w >= 0; --w)
            if (v.v[w]) return 32u * (uint32_t)w + 32u - (uint32_t)__builtin_clz(v.v[w]);
        return 0;
    };
    auto limb_sized = [&](uint32_t i) -> bool {
        const Node& n = g.nodes[i];
        if (any_width) return true;
        if (n.kind == N_CONST) return const_bits(i) <= 64;
        if (n.kind == N_SCAN) return !(n.op & (SCAN_OP_DIV | SCAN_OP_ACC | SCAN_OP_BORROW | SCAN_OP_LEX)) && scan_imm[i] <= 64;  // (the limb a carry step leaves: t mod 2^n)
        if (n.kind == N_DUO && n.op == OP_BAND) return (g.nodes[n.a].kind == N_CONST && const_bits(n.a) <= 64) || (g.nodes[n.b].kind == N_CONST && const_bits(n.b) <= 64);
        return false;
    };
    std::vector<uint8_t> dead(N, 0), taken(N, 0);
    bool any = false;
    const bool debug = getenv("CWC_DEBUG_CONV") != nullptr;
    const bool skip_dependency_check = getenv("CWC_CONV_SKIP_DEPENDENCY_CHECK") != nullptr;  // (the knob: tests of compile_program's fallback)
    std::vector<uint32_t> walk_epoch;  // the dependency walk's "seen in this block" marks
    uint32_t walk_now = 0;
    size_t n_dependent_blocks = 0;     // complete blocks left unfused because a factor depends on the block (CWC_DEBUG_CONV prints it)
    if (debug) {
        size_t np = 0, nroot = 0, nclean = 0, nseed = 0, ncc = 0, nmul = 0;
        for (size_t i = 0; i < N; ++i) {
            nmul += g.nodes[i].kind == N_DUO && g.nodes[i].op == OP_MUL;
            ncc += g.nodes[i].kind == N_DUO && g.nodes[i].op == OP_MUL && (vflags[i] & VF_MUL_CC);
            if (!is_product((uint32_t)i)) continue;
            ++np;
            nroot += root_of[i] != NONE;
            nclean += root_of[i] != NONE && leaves[root_of[i]] != NONE;
            nseed += root_of[i] == i && leaves[i] == 1;
        }
        fprintf(stderr, "convolution detection: %zu Mul nodes, %zu canonical, %zu candidate products, %zu with a root, %zu in clean trees, %zu lone\n", nmul, ncc, np, nroot, nclean, nseed);
    }
    for (size_t s0 = 0; s0 < N; ++s0) {
        // seed: a lone product that is its own root (column 0)
        const uint32_t seed = (uint32_t)s0;
        if (root_of[seed] != seed || leaves[seed] != 1 || taken[seed]) continue;
        for (int flip = 0; flip < 2; ++flip) {
            const uint32_t x0 = flip ? g.nodes[seed].b : g.nodes[seed].a, y0 = other(seed, x0);
            const std::vector<uint32_t>&px = adj[x0], &py = adj[y0];  // x_0 y for every y; x y_0 for every x
            const size_t k = px.size();
            if (debug) fprintf(stderr, "  seed %u: x0 in %zu products, y0 in %zu\n", seed, px.size(), py.size());
            if (k < 2 || k > 32 || py.size() != k || 2 * k - 1 > max_columns) continue;
            // indices from the sizes of the trees
            std::vector<uint32_t> X(k, NONE), Y(k, NONE);
            std::unordered_map<uint32_t, uint32_t> ix, jy;
            bool ok = true;
            for (uint32_t p : px) {
                const uint32_t y = other(p, x0), r = root_of[p];
                const uint32_t j = leaves[r] == NONE ? NONE : leaves[r] - 1;
                if (j >= k || Y[j] != NONE || taken[p]) { ok = false; break; }
                Y[j] = y;
                jy[y] = j;
            }
            for (uint32_t p : py) {
                if (!ok) break;
                const uint32_t x = other(p, y0), r = root_of[p];
                const uint32_t i = leaves[r] == NONE ? NONE : leaves[r] - 1;
                if (i >= k || X[i] != NONE || taken[p]) { ok = false; break; }
                X[i] = x;
                ix[x] = i;
            }
            if (!ok || X[0] != x0 || Y[0] != y0) continue;
            for (uint32_t x : X) ok = ok && !jy.count(x) && limb_sized(x);  // disjoint; limbs
            for (uint32_t y : Y) ok = ok && limb_sized(y);
            // every x has exactly the products x y_j, each in the tree of column i + j, each tree exactly one anti-diagonal
            std::vector<uint32_t> col_root(2 * k - 1, NONE);
            std::vector<uint32_t> prods;
            for (size_t i = 0; i < k && ok; ++i) {
                const std::vector<uint32_t>& pa = adj[X[i]];
                if (pa.size() != k) { ok = false; break; }
                std::vector<uint8_t> seen(k, 0);
                for (uint32_t p : pa) {
                    auto it = jy.find(other(p, X[i]));
                    if (it == jy.end() || seen[it->second] || taken[p]) { ok = false; break; }
                    seen[it->second] = 1;
                    const uint32_t c = (uint32_t)i + it->second, r = root_of[p];
                    const uint32_t want = (uint32_t)std::min<size_t>(c, 2 * k - 2 - c) + 1;
                    if (leaves[r] != want || (col_root[c] != NONE && col_root[c] != r)) { ok = false; break; }
                    col_root[c] = r;
                    prods.push_back(p);
                }
            }
            for (size_t j = 0; j < k && ok; ++j) ok = adj[Y[j]].size() == k;
            for (size_t c = 0; c < 2 * k - 1 && ok; ++c)
                for (size_t c2 = 0; c2 < c && ok; ++c2) ok = col_root[c] != col_root[c2];
            if (!ok) continue;
            // No factor may depend on the block itself: the bundle needs every x_i, y_j up front, and a factor computed from one of the
            // block's own products or column sums (x_1 = x_0 y_0 mod 2^64, say) would wait for the bundle that waits for it.  Walk up from
            // the factors; only nodes behind the block's first product can depend on it (operands precede their users).
            {
                uint32_t first_prod = NONE;
                for (uint32_t p : prods) first_prod = std::min(first_prod, p);
                std::vector<uint32_t> stack, block;  // the block: its products and the sums above them up to the roots
                for (uint32_t p : prods)
                    for (uint32_t i = p;; i = user[i]) {
                        block.push_back(i);
                        if (i == root_of[p]) break;
                    }
                std::sort(block.begin(), block.end());
                block.erase(std::unique(block.begin(), block.end()), block.end());
                auto in_block = [&](uint32_t i) { return std::binary_search(block.begin(), block.end(), i); };
                for (uint32_t f : X) stack.push_back(f);
                for (uint32_t f : Y) stack.push_back(f);
                // every node is visited once per block (an epoch array over the whole graph: no cap on the ancestry behind first_prod, no
                // linear search -- a block with a long ancestry used to lose its bundle silently once 4 096 nodes had been remembered)
                if (walk_epoch.empty()) walk_epoch.assign(N, 0);
                ++walk_now;
                while (!stack.empty() && ok && !skip_dependency_check) {
                    const uint32_t i = stack.back();
                    stack.pop_back();
                    if (i < first_prod || g.nodes[i].kind == N_CONST || walk_epoch[i] == walk_now) continue;
                    walk_epoch[i] = walk_now;
                    if (in_block(i)) { ok = false; break; }
                    const Node& fn = g.nodes[i];
                    const uint32_t fops[3] = {fn.a, fn.b, fn.c};
                    for (int q = 0; q < arity_of(fn); ++q) stack.push_back(fops[q]);
                }
                if (!ok) ++n_dependent_blocks;
            }
            if (!ok) continue;
            // rewrite: the roots become the column nodes, everything below them dies
            for (uint32_t p : prods) {
                taken[p] = 1;
                for (uint32_t i = p; i != root_of[p]; i = user[i]) dead[i] = 1;
            }
            for (size_t c = 0; c < 2 * k - 1; ++c) {
                const uint32_t r = col_root[c];
                dead[r] = 0;
                taken[r] = 1;
                g.nodes[r] = Node{N_CONV, 0, c < k ? X[c] : X[k - 1], c < k ? Y[c] : Y[k - 1], 0};
                rep[r] = REP_C;
                vflags[r] = 0;
                scan_imm[r] = (uint32_t)c | ((uint32_t)k << 8);
                scan_partner[r] = col_root[0];
            }
            n_products += k * k;
            any = true;
            break;
        }
    }
    if (debug && n_dependent_blocks) fprintf(stderr, "convolutions: %zu complete block(s) left unfused (a factor depends on the block)\n", n_dependent_blocks);
    if (any) compact_dead(g, dead, rep, vflags, scan_imm, scan_partner);
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
void fuse_narrow_chains(Graph& g, std::vector<uint8_t>& rep, std::vector<uint8_t>& vflags, const uint32_t* class_cost, uint32_t slack_permille,
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

}  // namespace cwc
