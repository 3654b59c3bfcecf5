// Native generators of BASELINE config 5's two class graphs (the reference front-end cannot compile such circuits, the
// graphs can only be synthetic).  The SPECIFICATION is the Python generator library of the package
// (circom-witnesscalc_amd/graphgen/circuits.py: build_bigint_class, build_rsa_long_div_class -- the latter restates the public
// witness-hint algorithms of circom-bigint as zk-email's RSA verifier uses them); these functions emit the same nodes in the same
// order through the same layout rules (constants first, Input(0), the inputs, then the operations; constants de-duplicated by
// value) and the product's own writer, so the bytes are equal (tests/test_host_formats.py compares them).  They exist because ten
// million nodes through a Python builder take the bench half a minute per graph.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#define GW_NO_INLINE_FREE_STATUS
#include "../../include/graph_witness_batch.h"
#include "graph.hpp"

using namespace cwc;

namespace {

// graphgen/builder.py Builder: symbolic nodes in creation order, laid out in reference order on finish()
struct Sym {
    uint8_t kind, op;  // NodeKind; op / input index for N_INPUT in a
    uint32_t a, b, c;
};
struct FrKey {
    uint32_t v[8];
    bool operator<(const FrKey& o) const { return memcmp(v, o.v, sizeof v) < 0; }
};
struct SymBuilder {
    std::vector<Sym> sym;
    std::vector<Fr> const_of;            // value of sym i (constants only; indexed by position in `consts`)
    std::vector<uint32_t> consts;        // sym ids of the constants, creation order
    std::map<FrKey, uint32_t> by_value;  // value -> sym id
    std::vector<uint32_t> witness;
    std::vector<InputSignal> inputs;
    uint32_t n_in = 1;
    uint32_t one_in;
    SymBuilder() {
        one_in = push(Sym{N_INPUT, 0, 0, 0, 0});
        witness.push_back(one_in);
    }
    uint32_t push(const Sym& s) {
        sym.push_back(s);
        return (uint32_t)(sym.size() - 1);
    }
    uint32_t constant(const Fr& v) {  // (callers pass canonical values below r)
        FrKey k;
        memcpy(k.v, v.v, sizeof k.v);
        auto it = by_value.find(k);
        if (it != by_value.end()) return it->second;
        const uint32_t s = push(Sym{N_CONST, 0, (uint32_t)consts.size(), 0, 0});
        consts.push_back(s);
        const_of.push_back(v);
        by_value[k] = s;
        return s;
    }
    uint32_t small(uint64_t x) {
        Fr v = fr_zero();
        v.v[0] = (uint32_t)x;
        v.v[1] = (uint32_t)(x >> 32);
        return constant(v);
    }
    static Fr pow2(uint32_t n) {
        Fr v = fr_zero();
        v.v[n >> 5] = 1u << (n & 31u);
        return v;
    }
    static Fr ones(uint32_t n) {  // 2^n - 1
        Fr v = fr_zero();
        for (uint32_t w = 0; w < 8; ++w) v.v[w] = n >= 32 * (w + 1) ? 0xffffffffu : n > 32 * w ? (1u << (n - 32 * w)) - 1u : 0u;
        return v;
    }
    std::vector<uint32_t> input(const char* name, uint32_t n) {
        inputs.push_back(InputSignal{name, n_in, n});
        std::vector<uint32_t> h(n);
        for (uint32_t i = 0; i < n; ++i) h[i] = push(Sym{N_INPUT, 0, n_in + i, 0, 0});
        n_in += n;
        return h;
    }
    uint32_t op(uint8_t code, uint32_t a, uint32_t b) { return push(Sym{N_DUO, code, a, b, 0}); }
    uint32_t mul(uint32_t a, uint32_t b) { return op(OP_MUL, a, b); }
    uint32_t add(uint32_t a, uint32_t b) { return op(OP_ADD, a, b); }
    uint32_t sub(uint32_t a, uint32_t b) { return op(OP_SUB, a, b); }
    uint32_t tern(uint32_t c, uint32_t a, uint32_t b) { return push(Sym{N_TRES, 0, c, a, b}); }
    uint32_t signal(uint32_t h) {
        witness.push_back(h);
        return h;
    }
    // Builder.finalize(): constants, inputs, operations -- each group in creation order
    void finish(Graph& g) const {
        const size_t n = sym.size();
        std::vector<uint32_t> remap(n);
        uint32_t pos = 0;
        for (size_t i = 0; i < n; ++i)
            if (sym[i].kind == N_CONST) remap[i] = pos++;
        for (size_t i = 0; i < n; ++i)
            if (sym[i].kind == N_INPUT) remap[i] = pos++;
        for (size_t i = 0; i < n; ++i)
            if (sym[i].kind != N_CONST && sym[i].kind != N_INPUT) remap[i] = pos++;
        g.nodes.assign(n, Node{0, 0, 0, 0, 0});
        g.const_values = const_of;
        g.n_op = 0;
        for (size_t i = 0; i < n; ++i) {
            const Sym& s = sym[i];
            Node& d = g.nodes[remap[i]];
            d.kind = s.kind;
            d.op = s.op;
            if (s.kind == N_CONST || s.kind == N_INPUT) {
                d.a = s.a;
                continue;
            }
            d.a = remap[s.a];
            d.b = s.kind == N_UNO ? 0 : remap[s.b];
            d.c = s.kind == N_TRES ? remap[s.c] : 0;
            g.n_op++;
        }
        g.witness_signals.resize(witness.size());
        for (size_t i = 0; i < witness.size(); ++i) g.witness_signals[i] = remap[witness[i]];
        g.inputs = inputs;
        for (size_t i = 0; i < inputs.size(); ++i) g.input_index[inputs[i].name] = (uint32_t)i;
    }
};

void set_status(gw_status_t* st, GW_ERROR_CODE code, const std::string& msg) {
    if (!st) return;
    st->code = code;
    st->error_msg = nullptr;
    if (code == OK && msg.empty()) return;
    st->error_msg = (char*)malloc(msg.size() + 1);
    if (st->error_msg) memcpy(st->error_msg, msg.c_str(), msg.size() + 1);
}

int emit(const SymBuilder& b, void** out, size_t* out_len, gw_status_t* status) {
    Graph g;
    b.finish(g);
    const std::vector<uint8_t> bytes = serialize_witnesscalc_graph(g);
    *out = malloc(bytes.size() ? bytes.size() : 1);
    if (!*out) {
        set_status(status, ERROR, "out of memory");
        return 1;
    }
    memcpy(*out, bytes.data(), bytes.size());
    *out_len = bytes.size();
    set_status(status, OK, "");
    return 0;
}

// ---- circuits.py build_rsa_long_div_class and its helpers, statement by statement ----
struct Rsa {
    SymBuilder& b;
    uint32_t n, k;
    uint32_t n_base, n_max, zero, one, two;
    typedef std::vector<uint32_t> Regs;
    Regs lsm(uint32_t kk, uint32_t a, const Regs& bb) {  // _lsm
        Regs out(kk + 1, zero);
        for (uint32_t i = 0; i < kk; ++i) {
            const uint32_t temp = b.add(out[i], b.mul(a, bb[i]));
            out[i] = b.op(OP_MOD, temp, n_base);
            out[i + 1] = b.add(out[i + 1], b.op(OP_IDIV, temp, n_base));
        }
        return out;
    }
    uint32_t long_gt(uint32_t kk, const Regs& x, const Regs& y) {  // _long_gt
        uint32_t res = zero;
        for (uint32_t i = 0; i < kk; ++i) {
            const uint32_t gt = b.op(OP_GT, x[i], y[i]);
            const uint32_t lt = b.op(OP_LT, x[i], y[i]);
            res = b.tern(gt, one, b.tern(lt, zero, res));
        }
        return res;
    }
    Regs long_sub(uint32_t kk, const Regs& x, const Regs& y) {  // _long_sub
        Regs diff;
        uint32_t borrow = 0;
        for (uint32_t i = 0; i < kk; ++i) {
            uint32_t c, d_then, d_else;
            if (i == 0) {
                c = b.op(OP_GEQ, x[i], y[i]);
                d_then = b.sub(x[i], y[i]);
                d_else = b.add(b.sub(x[i], y[i]), n_base);
            } else {
                c = b.op(OP_GEQ, x[i], b.add(y[i], borrow));
                d_then = b.sub(b.sub(x[i], y[i]), borrow);
                d_else = b.sub(b.sub(b.add(n_base, x[i]), y[i]), borrow);
            }
            diff.push_back(b.tern(c, d_then, d_else));
            borrow = b.tern(c, zero, one);
        }
        return diff;
    }
    uint32_t short_div_norm(uint32_t kk, const Regs& a, const Regs& bb) {  // _short_div_norm
        uint32_t qhat = b.op(OP_IDIV, b.add(b.mul(a[kk], n_base), a[kk - 1]), bb[kk - 1]);
        {
            const uint32_t gt = b.op(OP_GT, qhat, n_max);
            qhat = b.tern(gt, n_max, qhat);
        }
        const Regs mult = lsm(kk, qhat, bb);
        const uint32_t g1 = long_gt(kk + 1, mult, a);
        Regs bext = bb;
        bext.resize(kk);
        bext.push_back(zero);
        const Regs mult2 = long_sub(kk + 1, mult, bext);
        const uint32_t g2 = long_gt(kk + 1, mult2, a);
        // b.tern(b.op("Eq", g1, one), b.tern(b.op("Eq", g2, one), b.sub(qhat, two), b.sub(qhat, one)), qhat): Python evaluates the
        // arguments left to right -- Eq(g1), then the inner tern's arguments Eq(g2), Sub, Sub, then the inner tern, then the outer
        const uint32_t e1 = b.op(OP_EQ, g1, one);
        const uint32_t e2 = b.op(OP_EQ, g2, one);
        const uint32_t s2 = b.sub(qhat, two);
        const uint32_t s1 = b.sub(qhat, one);
        const uint32_t inner = b.tern(e2, s2, s1);
        return b.tern(e1, inner, qhat);
    }
};

}  // namespace

extern "C" {

int gwb_graphgen_bigint_class(uint32_t k, uint32_t n_bits, uint32_t rounds, void** out, size_t* out_len, gw_status_t* status) {
    try {
        if (!out || !out_len) {
            set_status(status, ERROR, "null argument");
            return 1;
        }
        if (k < 1 || k > 4096 || n_bits < 1 || n_bits > 126 || (uint64_t)rounds * (6ull * k * k + 12ull * k) > 0xf0000000ull) {
            set_status(status, ERROR, "gwb_graphgen_bigint_class: parameters out of range");
            return 1;
        }
        // circuits.py build_bigint_class, statement by statement
        SymBuilder b;
        const std::vector<uint32_t> a_in = b.input("a", k), b_in = b.input("b", k), d_in = b.input("d", 1);
        const uint32_t base = b.constant(SymBuilder::pow2(n_bits)), mask = b.constant(SymBuilder::ones(n_bits));
        const uint32_t one = b.small(1), zero = b.small(0);
        std::vector<uint32_t> x(k), y(k);
        for (uint32_t i = 0; i < k; ++i) x[i] = b.signal(b.op(OP_BAND, a_in[i], mask));
        for (uint32_t i = 0; i < k; ++i) y[i] = b.signal(b.op(OP_BAND, b_in[i], mask));
        uint32_t d = b.signal(b.add(b.op(OP_BAND, d_in[0], mask), one));
        for (uint32_t r = 0; r < rounds; ++r) {
            std::vector<uint32_t> cols(2 * k, 0xffffffffu);
            for (uint32_t i = 0; i < k; ++i)
                for (uint32_t j = 0; j < k; ++j) {
                    const uint32_t pr = b.mul(x[i], y[j]);
                    cols[i + j] = cols[i + j] == 0xffffffffu ? pr : b.add(cols[i + j], pr);
                }
            cols[2 * k - 1] = zero;
            uint32_t carry = zero;
            std::vector<uint32_t> prod;
            for (uint32_t c = 0; c < 2 * k; ++c) {
                const uint32_t t = b.add(cols[c], carry);
                prod.push_back(b.signal(b.op(OP_MOD, t, base)));
                carry = b.signal(b.op(OP_IDIV, t, base));
            }
            uint32_t rem = zero;
            std::vector<uint32_t> quo(2 * k, 0);
            for (uint32_t c = 2 * k; c-- > 0;) {
                const uint32_t t = b.add(b.mul(rem, base), prod[c]);
                quo[c] = b.signal(b.op(OP_IDIV, t, d));
                rem = b.signal(b.op(OP_MOD, t, d));
            }
            std::vector<uint32_t> nx, ny;
            for (uint32_t i = 0; i < k; ++i) {
                const uint32_t lt = b.signal(b.op(OP_LT, quo[i + k], y[i]));
                const uint32_t sel = b.signal(b.tern(lt, quo[i + k], y[i]));
                nx.push_back(b.signal(b.op(OP_BAND, b.add(quo[i], rem), mask)));
                ny.push_back(b.signal(b.op(OP_BAND, b.add(sel, one), mask)));
            }
            x = nx;
            y = ny;
            d = b.signal(b.add(b.op(OP_BAND, b.add(d, rem), mask), one));
        }
        return emit(b, out, out_len, status);
    } catch (...) {
        set_status(status, ERROR, "out of memory");
        return 1;
    }
}

int gwb_graphgen_rsa_long_div_class(uint32_t n, uint32_t k, uint32_t muls, int range_checks, void** out, size_t* out_len, gw_status_t* status) {
    try {
        if (!out || !out_len) {
            set_status(status, ERROR, "null argument");
            return 1;
        }
        if (n < 2 || n > 126 || k < 1 || k > 64 || (uint64_t)muls * ((uint64_t)(k + 1) * (90ull * k + 200) + 2ull * k * (2ull * n + 4)) > 0xf0000000ull) {
            set_status(status, ERROR, "gwb_graphgen_rsa_long_div_class: parameters out of range");
            return 1;
        }
        SymBuilder b;
        const std::vector<uint32_t> x_in = b.input("base", k), p_in = b.input("modulus", k);
        Rsa R{b, n, k, 0, 0, 0, 0, 0};
        R.n_base = b.constant(SymBuilder::pow2(n));
        R.n_max = b.constant(SymBuilder::ones(n));
        const uint32_t mask = b.constant(SymBuilder::ones(n));
        R.zero = b.small(0);
        R.one = b.small(1);
        R.two = b.small(2);
        Rsa::Regs x(k), p;
        for (uint32_t i = 0; i < k; ++i) x[i] = b.signal(b.op(OP_BAND, x_in[i], mask));
        for (uint32_t i = 0; i + 1 < k; ++i) p.push_back(b.signal(b.op(OP_BAND, p_in[i], mask)));
        const uint32_t top_bits = n > 11 ? n - 10 : 1;
        {
            // b.signal(b.add(b.op("Band", p_in[-1], b.const((1 << top_bits) - 1)), b.const(1 << top_bits))): the mask constant, the Band, the other constant, the Add
            const uint32_t m_top = b.constant(SymBuilder::ones(top_bits));
            const uint32_t band = b.op(OP_BAND, p_in[k - 1], m_top);
            const uint32_t c_top = b.constant(SymBuilder::pow2(top_bits));
            p.push_back(b.signal(b.add(band, c_top)));
        }
        const uint32_t scale = b.op(OP_IDIV, R.n_base, b.add(R.one, p[k - 1]));
        const Rsa::Regs norm_b = R.lsm(k, scale, p);
        Rsa::Regs acc = x;
        for (uint32_t s = 0; s < muls; ++s) {
            const Rsa::Regs& u = acc;
            const Rsa::Regs& v = (s % 17 == 16) ? x : acc;
            // fp_mul
            std::vector<uint32_t> cols(2 * k - 1, 0xffffffffu);
            for (uint32_t i = 0; i < k; ++i)
                for (uint32_t j = 0; j < k; ++j) {
                    const uint32_t pr = b.mul(u[i], v[j]);
                    cols[i + j] = cols[i + j] == 0xffffffffu ? pr : b.add(cols[i + j], pr);
                }
            uint32_t carry = R.zero;
            Rsa::Regs prod;
            for (uint32_t c = 0; c + 1 < 2 * k; ++c) {
                const uint32_t t = b.add(cols[c], carry);
                prod.push_back(b.op(OP_MOD, t, R.n_base));
                carry = b.op(OP_IDIV, t, R.n_base);
            }
            prod.push_back(carry);
            // long_div(prod): m = k
            const uint32_t m = k;
            Rsa::Regs rem = prod, quo(m + 1, 0);
            for (uint32_t i = m + 1; i-- > 0;) {
                Rsa::Regs dividend;
                if (i == m) {
                    for (uint32_t j = 0; j < k; ++j) dividend.push_back(rem[j + m]);
                    dividend.push_back(R.zero);
                } else {
                    for (uint32_t j = 0; j <= k; ++j) dividend.push_back(rem[j + i]);
                }
                // short_div(dividend)
                const Rsa::Regs norm_a = R.lsm(k + 1, scale, dividend);
                const uint32_t wide = R.short_div_norm(k + 1, norm_a, norm_b);
                Rsa::Regs nb_k(norm_b.begin(), norm_b.begin() + k);
                const uint32_t narrow = R.short_div_norm(k, norm_a, nb_k);
                quo[i] = b.tern(b.op(OP_NEQ, norm_b[k], R.zero), wide, narrow);
                const Rsa::Regs mult_shift = R.lsm(k, quo[i], p);
                Rsa::Regs subtrahend(m + k, R.zero);
                for (uint32_t j = 0; j <= k; ++j)
                    if (i + j < m + k) subtrahend[i + j] = mult_shift[j];
                rem = R.long_sub(m + k, rem, subtrahend);
            }
            Rsa::Regs out_regs(quo.begin(), quo.begin() + k);
            for (uint32_t i = 0; i < k; ++i) out_regs.push_back(rem[i]);
            for (uint32_t v_ : out_regs) {
                b.signal(v_);
                if (range_checks)
                    for (uint32_t j = 0; j < n; ++j) b.signal(b.op(OP_BAND, b.op(OP_SHR, v_, b.small(j)), R.one));
            }
            acc.assign(rem.begin(), rem.begin() + k);
        }
        return emit(b, out, out_len, status);
    } catch (...) {
        set_status(status, ERROR, "out of memory");
        return 1;
    }
}

}  // extern "C"
