// Load-time re-optimiser of a loaded graph: the reference's build-time passes (src/graph.rs:358-619:
// tree_shake, propagate, value_numbering, constants) restated as EXACT rewrites that a `.bin` from any producer goes
// through before scheduling.  The default passes are exact (the reference's value_numbering / constants evaluate the graph
// on random field elements, :499-600: those are restated further down, opt-in), and every witness value stays bit-identical:
//
//   propagate        an operation whose operands are all constants becomes a constant, evaluated with the semantics of
//                    Operation::eval_fr / UnoOperation::eval_fr / TresOperation::eval_fr (src/graph.rs:102-144, 188-197,
//                    221-225) -- what evaluate() would have computed at run time; the reference's own pass uses
//                    Operation::eval on U256 (:394-428), whose results differ from eval_fr for Shl and the bit operations,
//                    so it is deliberately NOT followed.  A constant operation that would panic in the reference (Shl
//                    result >= r, bit-op result == r) is left in place: the failure still surfaces at evaluation.
//                    Same-operand comparisons (:404-413: Eq/Leq/Geq -> 1, Neq/Lt/Gt -> 0) and the field identities
//                    x*0, x*1, x+0, x-0, x-x, 0/x are folded too.
//   value numbering  two pure operations with the same operator and the same (already numbered) operands are one node;
//                    Add / Mul / Eq / Neq / Land / Lor / Bor / Band / Bxor commute.  Structural, hence exact.
//   tree shake       nodes nothing depends on are dropped (:431-498) -- except Input nodes (they size the inputs buffer,
//                    src/lib.rs:138-152) and every operation that can fail, with what feeds it: the reference evaluates
//                    the whole graph, so an unused Shl that overflows still aborts there and still reports here.
#include <string.h>

#include <unordered_map>

#include "flat_map.hpp"
#include "graph.hpp"

namespace cwc {

namespace {

bool is_zero(const Fr& x) { return u256_is_zero(x); }
Fr one_canon() {
    Fr o = fr_zero();
    o.v[0] = 1;
    return o;
}
Fr bool_val(bool b) { return b ? one_canon() : fr_zero(); }

// signed comparison of canonical values, src/graph.rs:720-769 (neg(x) := x > halfM)
void cmp_signed(const Fr& a, const Fr& b, bool& lt, bool& gt) {
    const bool an = u256_lt(fr_half(), a), bn = u256_lt(fr_half(), b);
    if (an == bn) {
        lt = u256_lt(a, b);
        gt = u256_lt(b, a);
    } else {
        lt = an;
        gt = bn;
    }
}

// eval_fr on canonical operands -> canonical result; false = the reference would panic (leave the node alone)
bool fold_duo(uint8_t op, const Fr& a, const Fr& b, Fr& out) {
    switch (op) {
        case OP_MUL: out = fr_from_mont(fr_mul(fr_to_mont(a), fr_to_mont(b))); return true;
        case OP_DIV:
            if (is_zero(b)) out = fr_zero();
            else out = fr_from_mont(fr_mul(fr_to_mont(a), fr_inv(fr_to_mont(b))));
            return true;
        case OP_ADD: out = fr_add(a, b); return true;
        case OP_SUB: out = fr_sub(a, b); return true;
        case OP_IDIV:
        case OP_MOD: {
            if (is_zero(b)) {
                out = fr_zero();
                return true;
            }
            Fr q, rem;
            u256_divrem(q, rem, a, b, 256);
            out = op == OP_IDIV ? q : rem;
            return true;
        }
        case OP_EQ: out = bool_val(u256_eq(a, b)); return true;
        case OP_NEQ: out = bool_val(!u256_eq(a, b)); return true;
        case OP_LT: case OP_GT: case OP_LEQ: case OP_GEQ: {
            bool lt, gt;
            cmp_signed(a, b, lt, gt);
            out = bool_val(op == OP_LT ? lt : op == OP_GT ? gt : op == OP_LEQ ? !gt : !lt);
            return true;
        }
        case OP_LAND: out = bool_val(!is_zero(a) && !is_zero(b)); return true;
        case OP_LOR: out = bool_val(!is_zero(a) || !is_zero(b)); return true;
        case OP_SHL:
        case OP_SHR: {  // src/graph.rs:621-672
            if (is_zero(b)) {
                out = a;
                return true;
            }
            uint32_t hi = 0;
            for (int i = 1; i < 8; ++i) hi |= b.v[i];
            if (hi != 0 || b.v[0] >= 254u) {
                out = fr_zero();
                return true;
            }
            out = op == OP_SHL ? u256_shl(a, b.v[0]) : u256_shr(a, b.v[0]);
            return op == OP_SHR || u256_lt(out, fr_p());  // Shl: from_bigint().unwrap() panics on >= r (:634)
        }
        case OP_BOR: case OP_BAND: case OP_BXOR: {  // :674-717
            Fr d;
            for (int i = 0; i < 8; ++i) d.v[i] = op == OP_BAND ? (a.v[i] & b.v[i]) : op == OP_BOR ? (a.v[i] | b.v[i]) : (a.v[i] ^ b.v[i]);
            Fr t;
            if (u256_sub(t, d, fr_p()) == 0) {  // d >= r
                if (is_zero(t)) return false;    // d == r: the reference panics
                d = t;
            }
            out = d;
            return true;
        }
        default: return false;  // Pow: not evaluable (graph.rs:141-142), rejected later
    }
}

bool can_fail(const Node& n) { return n.kind == N_DUO && (n.op == OP_SHL || n.op == OP_BOR || n.op == OP_BXOR || n.op == OP_BAND); }
bool commutes(uint8_t op) {
    return op == OP_MUL || op == OP_ADD || op == OP_EQ || op == OP_NEQ || op == OP_LAND || op == OP_LOR || op == OP_BOR || op == OP_BAND || op == OP_BXOR;
}

struct FrHash {
    size_t operator()(const Fr& x) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 8; ++i) h = (h ^ x.v[i]) * 1099511628211ull;
        return (size_t)h;
    }
};
struct FrEq {
    bool operator()(const Fr& a, const Fr& b) const { return memcmp(a.v, b.v, sizeof a.v) == 0; }
};

}  // namespace

// ---- the reference's PROBABILISTIC passes (src/graph.rs:499-583), opt-in (CWC_RANDOM_EVAL=1) -------------------------------
// random_eval (:500-533): the graph evaluated on random field elements -- Add / Sub / Mul (and Neg) algebraically, every Input
// and every other operation as a random function of its operand VALUES.  By Schwartz-Zippel two nodes with the same value
// are the same polynomial in the inputs and the non-algebraic results (error probability ~ degree / r for the algebraic
// part, ~N^2 / 2^257 for collisions of the random functions: below 2^-200 for any graph that fits a file), which sees what structural value numbering cannot: (a + b) * c against a * c + b * c, sums in
// another association, x - x.
//   value_numbering (:536-562): every reference to a node goes to the FIRST node with its value.
//   constants       (:565-583): a node with the same value under two independent evaluations is a constant.
// A graph from the reference's own build-circuit has been through both; a third-party producer's has not.  Not exact in the
// strict sense, hence not on by default: the exact passes below run either way and clean up behind these (duplicates, dead
// nodes; operations that can fail are random functions here and therefore never folded, and tree shaking keeps them).
// The random functions are keyed hashes (deterministic: every rank of a job compiles the same program).
namespace {
uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// The random function of (tag, operand values): FOUR independently keyed 64-bit lanes absorb every operand word, so two
// different operand tuples collide only if all four lanes do (~N^2 / 2^257 over a graph's nodes; one 64-bit state expanded to
// 256 bits -- the first version -- made unrelated nodes collide with probability N^2 / 2^65, 3e-6 for a 10 M-node graph, and
// value numbering then merged them).  The reference draws fresh 256-bit randoms memoised on the operand tuple (graph.rs:523-532).
Fr prf(uint64_t seed, uint64_t tag, const Fr* a, const Fr* b, const Fr* c) {
    uint64_t h[4];
    for (int k = 0; k < 4; ++k) h[k] = mix64((seed + 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1)) ^ mix64(tag + 0xd1b54a32d192ed03ull * (uint64_t)k));
    for (const Fr* x : {a, b, c}) {
        if (!x) continue;
        for (int i = 0; i < 8; i += 2) {
            const uint64_t w = (uint64_t)x->v[i] | ((uint64_t)x->v[i + 1] << 32);
            for (int k = 0; k < 4; ++k) h[k] = mix64(h[k] ^ w);
        }
        for (int k = 0; k < 4; ++k) h[k] = mix64(h[k] + 0x51);
    }
    uint8_t bytes[32];
    for (int k = 0; k < 4; ++k) {
        const uint64_t w = mix64(h[k] + 0x1000193ull * (uint64_t)(k + 1));
        memcpy(bytes + 8 * k, &w, 8);
    }
    return u256_from_le_bytes_mod_order(bytes, 32);  // a field element, read as a Montgomery residue
}
void random_eval(const Graph& g, uint64_t seed, std::vector<Fr>& val) {
    const size_t N = g.nodes.size();
    val.resize(N);
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        switch (n.kind) {
            case N_CONST: val[i] = fr_to_mont(g.const_values[n.a]); break;
            case N_INPUT: val[i] = prf(seed, 0x100000000ull | n.a, nullptr, nullptr, nullptr); break;
            case N_UNO:
                if (n.op == UOP_NEG) val[i] = fr_neg(val[n.a]);
                else val[i] = prf(seed, 0x200000000ull | n.op, &val[n.a], nullptr, nullptr);
                break;
            case N_DUO:
                if (n.op == OP_ADD) val[i] = fr_add(val[n.a], val[n.b]);
                else if (n.op == OP_SUB) val[i] = fr_sub(val[n.a], val[n.b]);
                else if (n.op == OP_MUL) val[i] = fr_mul(val[n.a], val[n.b]);
                else val[i] = prf(seed, 0x300000000ull | n.op, &val[n.a], &val[n.b], nullptr);
                break;
            default: val[i] = prf(seed, 0x400000000ull | n.op, &val[n.a], &val[n.b], &val[n.c]); break;
        }
    }
}
}  // namespace

void random_eval_passes(Graph& g, OptimizeStats* stats) {
    const size_t N = g.nodes.size();
    std::vector<Fr> va, vb;
    random_eval(g, 0x6a09e667f3bcc908ull, va);
    random_eval(g, 0xbb67ae8584caa73bull, vb);
    OptimizeStats st;
    if (stats) st = *stats;
    // constants (graph.rs:565-583)
    for (size_t i = 0; i < N; ++i) {
        Node& n = g.nodes[i];
        if (n.kind == N_CONST || n.kind == N_INPUT) continue;
        if (memcmp(va[i].v, vb[i].v, sizeof va[i].v) == 0) {
            g.const_values.push_back(fr_from_mont(va[i]));
            n = Node{N_CONST, 0, (uint32_t)(g.const_values.size() - 1), 0, 0};
            st.random_constants++;
        }
    }
    // value numbering (graph.rs:536-562): the first node of every value
    std::unordered_map<Fr, uint32_t, FrHash, FrEq> first;
    first.reserve(N);
    std::vector<uint32_t> renumber(N);
    for (size_t i = 0; i < N; ++i) {
        auto it = first.emplace(va[i], (uint32_t)i);
        renumber[i] = it.first->second;
        st.random_numbered += !it.second && g.nodes[i].kind != N_CONST && g.nodes[i].kind != N_INPUT;
    }
    for (Node& n : g.nodes) {
        const int ar = n.kind == N_UNO ? 1 : n.kind == N_DUO ? 2 : n.kind == N_TRES ? 3 : 0;
        if (ar >= 1) n.a = renumber[n.a];
        if (ar >= 2) n.b = renumber[n.b];
        if (ar >= 3) n.c = renumber[n.c];
    }
    for (uint32_t& w : g.witness_signals) w = renumber[w];
    if (stats) *stats = st;
}

// Rewrites g in place (nodes, constants, witness references); the input map is untouched.  References must be backward
// (validated by the caller).  Returns counts for the statistics line.
void optimize_loaded_graph(Graph& g, OptimizeStats* stats) {
    const size_t N = g.nodes.size();
    std::vector<Node> out;
    out.reserve(N);
    std::vector<uint32_t> m(N, 0xffffffffu);  // old node -> new node
    std::unordered_map<Fr, uint32_t, FrHash, FrEq> const_node;  // canonical value -> new constant node
    FlatMap128 vn(g.n_op ? g.n_op : N);
    OptimizeStats st;
    if (stats) {  // (counters of the probabilistic passes that ran in front)
        st.random_constants = stats->random_constants;
        st.random_numbered = stats->random_numbered;
    }
    st.nodes_before = N;
    auto make_const = [&](const Fr& v) -> uint32_t {
        auto it = const_node.find(v);
        if (it != const_node.end()) return it->second;
        const uint32_t idx = (uint32_t)out.size();
        out.push_back(Node{N_CONST, 0, (uint32_t)g.const_values.size(), 0, 0});
        g.const_values.push_back(v);
        const_node.emplace(v, idx);
        return idx;
    };
    auto const_of = [&](uint32_t idx, Fr& v) -> bool {
        if (out[idx].kind != N_CONST) return false;
        v = g.const_values[out[idx].a];
        return true;
    };
    // Values that are 0 or 1 by construction: the constants 0 / 1, every comparison and logical operation (graph.rs:122-135 produce
    // Fr::zero() / Fr::one()), a selection between two such values.  For them `b == 1` and `b != 0` ARE b (round 5: the predicated
    // `if (long_gt(..) == 1)` of big-integer circuits -- an equality test and a bundle on the critical chain per use).  Memoised per new node.
    std::vector<int8_t> bool_memo;
    auto is_boolean = [&](uint32_t idx) -> bool {
        if (bool_memo.size() < out.size()) bool_memo.resize(out.size(), -1);
        if (bool_memo[idx] >= 0) return bool_memo[idx] != 0;
        // (selections chain: an explicit stack instead of recursion)
        std::vector<uint32_t> stack{idx};
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            if (bool_memo[i] >= 0) { stack.pop_back(); continue; }
            const Node& n = out[i];
            if (n.kind == N_CONST) {
                const Fr& v = g.const_values[n.a];
                bool small = v.v[0] <= 1u;
                for (int w = 1; w < 8; ++w) small = small && v.v[w] == 0u;
                bool_memo[i] = small ? 1 : 0;
            } else if (n.kind == N_DUO) {
                bool_memo[i] = (n.op >= OP_EQ && n.op <= OP_LOR) ? 1 : 0;
            } else if (n.kind == N_TRES) {
                if (bool_memo[n.b] < 0) { stack.push_back(n.b); continue; }
                if (bool_memo[n.c] < 0) { stack.push_back(n.c); continue; }
                bool_memo[i] = (bool_memo[n.b] == 1 && bool_memo[n.c] == 1) ? 1 : 0;
            } else {
                bool_memo[i] = 0;
            }
            stack.pop_back();
        }
        return bool_memo[idx] != 0;
    };
    // Constants of the file sit in front of their users or (appended by earlier rewrites) behind them: number them first.
    for (size_t i = 0; i < N; ++i)
        if (g.nodes[i].kind == N_CONST) {
            const Fr v = g.const_values[g.nodes[i].a];
            auto it = const_node.find(v);
            if (it != const_node.end()) {
                m[i] = it->second;
                st.constants_merged++;
            } else {
                m[i] = (uint32_t)out.size();
                out.push_back(g.nodes[i]);
                const_node.emplace(v, m[i]);
            }
        }
    for (size_t i = 0; i < N; ++i) {
        const Node& n = g.nodes[i];
        if (n.kind == N_CONST) continue;
        if (n.kind == N_INPUT) {
            m[i] = (uint32_t)out.size();
            out.push_back(n);
            continue;
        }
        Node c = n;
        const int ar = n.kind == N_UNO ? 1 : n.kind == N_DUO ? 2 : 3;
        c.a = m[n.a];
        if (ar >= 2) c.b = m[n.b];
        if (ar >= 3) c.c = m[n.c];
        Fr va, vb, vc, r;
        const bool ca = const_of(c.a, va), cb = ar >= 2 && const_of(c.b, vb), cc = ar >= 3 && const_of(c.c, vc);
        uint32_t repl = 0xffffffffu;
        if (n.kind == N_UNO) {
            if (n.op == UOP_NEG && ca) {  // graph.rs:188-194
                repl = make_const(is_zero(va) ? va : fr_sub(fr_zero(), va));
                st.folded++;
            }
        } else if (n.kind == N_TRES) {
            if (ca) {  // TernCond with a constant condition selects one arm (graph.rs:221-225); both arms stay evaluated elsewhere if used
                repl = is_zero(va) ? c.c : c.b;
                st.folded++;
            } else if (c.b == c.c) {
                repl = c.b;
                st.folded++;
            }
            (void)cb; (void)cc; (void)vc;
        } else {
            const uint8_t op = n.op;
            if (ca && cb) {
                if (fold_duo(op, va, vb, r)) {
                    repl = make_const(r);
                    st.folded++;
                }
            } else if (c.a == c.b && (op == OP_EQ || op == OP_LEQ || op == OP_GEQ)) {
                repl = make_const(one_canon());
                st.folded++;
            } else if (c.a == c.b && (op == OP_NEQ || op == OP_LT || op == OP_GT || op == OP_SUB)) {
                repl = make_const(fr_zero());
                st.folded++;
            } else if (op == OP_MUL && ((ca && is_zero(va)) || (cb && is_zero(vb)))) {
                repl = make_const(fr_zero());
                st.folded++;
            } else if (op == OP_MUL && ca && u256_eq(va, one_canon())) {
                repl = c.b;
                st.folded++;
            } else if (op == OP_MUL && cb && u256_eq(vb, one_canon())) {
                repl = c.a;
                st.folded++;
            } else if (op == OP_ADD && ca && is_zero(va)) {
                repl = c.b;
                st.folded++;
            } else if ((op == OP_ADD || op == OP_SUB) && cb && is_zero(vb)) {
                repl = c.a;
                st.folded++;
            } else if (op == OP_DIV && ((ca && is_zero(va)) || (cb && is_zero(vb)))) {  // 0 / x = 0, x / 0 = 0 (graph.rs:109)
                repl = make_const(fr_zero());
                st.folded++;
            } else if (op == OP_DIV && cb && u256_eq(vb, one_canon())) {
                repl = c.a;
                st.folded++;
            } else if (op == OP_EQ && ((cb && u256_eq(vb, one_canon()) && is_boolean(c.a)) || (ca && u256_eq(va, one_canon()) && is_boolean(c.b)))) {
                repl = cb ? c.a : c.b;  // b == 1 for b in {0, 1}: b (graph.rs:122-125 gives Fr::one() / Fr::zero(), the same field elements)
                st.folded++;
            } else if (op == OP_NEQ && ((cb && is_zero(vb) && is_boolean(c.a)) || (ca && is_zero(va) && is_boolean(c.b)))) {
                repl = cb ? c.a : c.b;  // b != 0 for b in {0, 1}: b (graph.rs:126-129)
                st.folded++;
            }
        }
        if (repl == 0xffffffffu) {
            if (n.kind == N_DUO && commutes(n.op) && c.a > c.b) std::swap(c.a, c.b);
            bool fresh = false;
            repl = vn.find_or_insert(((uint64_t)c.kind << 56) | ((uint64_t)c.op << 48) | c.a, ((uint64_t)(ar >= 2 ? c.b : 0u) << 32) | (ar >= 3 ? c.c : 0u),
                                     (uint32_t)out.size(), &fresh);
            if (fresh) out.push_back(c);
            else st.numbered++;
        }
        m[i] = repl;
    }
    // ---- tree shake ------------------------------------------------------------------------------------------------
    const size_t M = out.size();
    std::vector<uint8_t> live(M, 0);
    for (uint32_t w : g.witness_signals) live[m[w]] = 1;
    for (size_t i = 0; i < M; ++i)
        if (out[i].kind == N_INPUT || can_fail(out[i])) live[i] = 1;
    for (size_t i = M; i-- > 0;) {
        if (!live[i]) continue;
        const Node& n = out[i];
        if (n.kind == N_UNO || n.kind == N_DUO || n.kind == N_TRES) live[n.a] = 1;
        if (n.kind == N_DUO || n.kind == N_TRES) live[n.b] = 1;
        if (n.kind == N_TRES) live[n.c] = 1;
    }
    std::vector<uint32_t> pos(M, 0xffffffffu);
    std::vector<Node> kept;
    kept.reserve(M);
    for (size_t i = 0; i < M; ++i)  // constants may sit behind their users (make_const appends): numbered in a first sweep
        if (live[i] && out[i].kind == N_CONST) {
            pos[i] = (uint32_t)kept.size();
            kept.push_back(out[i]);
        }
    for (size_t i = 0; i < M; ++i) {
        if (!live[i] || out[i].kind == N_CONST) continue;
        Node n = out[i];
        if (n.kind == N_UNO || n.kind == N_DUO || n.kind == N_TRES) n.a = pos[n.a];
        if (n.kind == N_DUO || n.kind == N_TRES) n.b = pos[n.b];
        if (n.kind == N_TRES) n.c = pos[n.c];
        pos[i] = (uint32_t)kept.size();
        kept.push_back(n);
    }
    for (uint32_t& w : g.witness_signals) w = pos[m[w]];
    st.nodes_after = kept.size();
    st.shaken = M - kept.size();
    g.nodes.swap(kept);
    if (stats) *stats = st;
}

}  // namespace cwc
