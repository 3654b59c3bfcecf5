/* ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the calc-witness hot path of iden3/circom-witnesscalc
 * (reference @ 2024-10-22, v0.2.0): the `.bin` reader (src/storage.rs:214-249), the sequential
 * evaluator graph::evaluate (src/graph.rs:367-391) with Operation::eval_fr (:102-144),
 * UnoOperation::eval_fr (:188-197), TresOperation::eval_fr (:221-225) and helpers (:621-769),
 * and the `.wtns` framing (src/lib.rs:114-123).  Same algorithm and data layout as the reference:
 * array-of-nodes, one sequential pass, values kept in Montgomery form (4 x u64, R = 2^256), scalar
 * arithmetic, no SIMD, no batching tricks.  It is also the "port" CPU baseline timed by bench.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: the Rust reference is unbuildable here (no cargo/rustc; ark-ff 0.4.2, ark-bn254
 * 0.4.0, ruint 1.12.3, wtns-file 0.1.5, prost 0.13.3 are not vendored).  Pinned by the reference's
 * unit vectors (src/graph.rs:779-883) and the circuit1 fixture in
 * tests/test_oracle_golden.py; otherwise "parity unpinned" (restated from source + the published
 * semantics of the dependencies: canonical residues mod r, integer quotient/remainder).
 */
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t; /* little-endian limbs */

/* src/field.rs:3-4 */
static const fr_t FR_P = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
/* src/field.rs:6 (commented there): INV = -r^-1 mod 2^64 */
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;
/* R = 2^256 mod r (src/field.rs:8), R2 = R^2 mod r */
static const fr_t FR_R = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
static const fr_t FR_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
/* (r-1)/2, src/graph.rs:720 */
static const fr_t FR_HALF = {{0xa1f0fac9f8000000ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}};

static int u256_cmp(const fr_t *a, const fr_t *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] < b->l[i]) return -1;
        if (a->l[i] > b->l[i]) return 1;
    }
    return 0;
}
static int u256_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static uint64_t u256_add(fr_t *r, const fr_t *a, const fr_t *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t u256_sub(fr_t *r, const fr_t *a, const fr_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}

/* Montgomery product a*b*R^-1 mod r (CIOS); valid for any a < 2^256 when b < r (result < r). */
static void fr_mul(fr_t *out, const fr_t *a, const fr_t *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)t[j] + (u128)a->l[j] * b->l[i]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_INV;
        c = (u128)t[0] + (u128)m * FR_P.l[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)t[j] + (u128)m * FR_P.l[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr_t r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || u256_cmp(&r, &FR_P) >= 0) u256_sub(&r, &r, &FR_P);
    *out = r;
}
static void fr_add(fr_t *r, const fr_t *a, const fr_t *b) {
    uint64_t c = u256_add(r, a, b);
    if (c || u256_cmp(r, &FR_P) >= 0) u256_sub(r, r, &FR_P);
}
static void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) {
    if (u256_sub(r, a, b)) u256_add(r, r, &FR_P);
}
/* Fr::new(BigInt): to Montgomery form (reduces any x < 2^256 mod r) */
static void fr_from_u256(fr_t *r, const fr_t *x) { fr_mul(r, x, &FR_R2); }
/* into_bigint(): canonical representative */
static void fr_to_u256(fr_t *r, const fr_t *x) { fr_t one = {{1, 0, 0, 0}}; fr_mul(r, x, &one); }

/* Field inverse of a Montgomery-form element, binary extended Euclid (the algorithm family ark-ff's
 * Field::inverse uses [ext]); input != 0.  With x = aR: returns a^-1 R. */
static void fr_inv(fr_t *out, const fr_t *x) {
    fr_t u = *x, v = FR_P, b = FR_R2, c = {{0, 0, 0, 0}};
    fr_t one = {{1, 0, 0, 0}};
    while (u256_cmp(&u, &one) != 0 && u256_cmp(&v, &one) != 0) {
        while ((u.l[0] & 1) == 0) {
            for (int i = 0; i < 3; ++i) u.l[i] = (u.l[i] >> 1) | (u.l[i + 1] << 63);
            u.l[3] >>= 1;
            uint64_t carry = 0;
            if (b.l[0] & 1) carry = u256_add(&b, &b, &FR_P);
            for (int i = 0; i < 3; ++i) b.l[i] = (b.l[i] >> 1) | (b.l[i + 1] << 63);
            b.l[3] = (b.l[3] >> 1) | (carry << 63);
        }
        while ((v.l[0] & 1) == 0) {
            for (int i = 0; i < 3; ++i) v.l[i] = (v.l[i] >> 1) | (v.l[i + 1] << 63);
            v.l[3] >>= 1;
            uint64_t carry = 0;
            if (c.l[0] & 1) carry = u256_add(&c, &c, &FR_P);
            for (int i = 0; i < 3; ++i) c.l[i] = (c.l[i] >> 1) | (c.l[i + 1] << 63);
            c.l[3] = (c.l[3] >> 1) | (carry << 63);
        }
        if (u256_cmp(&v, &u) < 0) { u256_sub(&u, &u, &v); fr_sub(&b, &b, &c); }
        else { u256_sub(&v, &v, &u); fr_sub(&c, &c, &b); }
    }
    *out = (u256_cmp(&u, &one) == 0) ? b : c;
}

/* U256 / and % (ruint [ext]): Knuth algorithm D on 32-bit digits. b != 0. */
static void u256_divrem(fr_t *q, fr_t *rem, const fr_t *a, const fr_t *b) {
    uint32_t u[9], v[8], qd[8];
    int n = 0, m;
    for (int i = 0; i < 4; ++i) { u[2 * i] = (uint32_t)a->l[i]; u[2 * i + 1] = (uint32_t)(a->l[i] >> 32);
                                  v[2 * i] = (uint32_t)b->l[i]; v[2 * i + 1] = (uint32_t)(b->l[i] >> 32); }
    for (int i = 7; i >= 0; --i) if (v[i]) { n = i + 1; break; }
    memset(qd, 0, sizeof qd);
    if (n == 1) {
        uint64_t r = 0;
        for (int i = 7; i >= 0; --i) { uint64_t cur = (r << 32) | u[i]; qd[i] = (uint32_t)(cur / v[0]); r = cur % v[0]; }
        memset(rem, 0, sizeof *rem); rem->l[0] = r;
    } else {
        int s = __builtin_clz(v[n - 1]);
        uint32_t vn[8], un[9];
        for (int i = n - 1; i > 0; --i) vn[i] = (v[i] << s) | (s ? (v[i - 1] >> (32 - s)) : 0);
        vn[0] = v[0] << s;
        un[8] = s ? (u[7] >> (32 - s)) : 0;
        for (int i = 7; i > 0; --i) un[i] = (u[i] << s) | (s ? (u[i - 1] >> (32 - s)) : 0);
        un[0] = u[0] << s;
        m = 8 - n;
        for (int j = m; j >= 0; --j) {
            uint64_t num = ((uint64_t)un[j + n] << 32) | un[j + n - 1];
            uint64_t qhat = num / vn[n - 1], rhat = num % vn[n - 1];
            while (qhat >= (1ULL << 32) || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
                --qhat; rhat += vn[n - 1];
                if (rhat >= (1ULL << 32)) break;
            }
            int64_t borrow = 0; uint64_t carry = 0;
            for (int i = 0; i < n; ++i) {
                uint64_t p = qhat * vn[i] + carry; carry = p >> 32;
                int64_t t = (int64_t)un[i + j] - borrow - (int64_t)(p & 0xffffffffULL);
                un[i + j] = (uint32_t)t; borrow = (t < 0);
            }
            int64_t t = (int64_t)un[j + n] - borrow - (int64_t)carry;
            un[j + n] = (uint32_t)t;
            if (t < 0) {
                --qhat; uint64_t c2 = 0;
                for (int i = 0; i < n; ++i) { c2 += (uint64_t)un[i + j] + vn[i]; un[i + j] = (uint32_t)c2; c2 >>= 32; }
                un[j + n] += (uint32_t)c2;
            }
            qd[j] = (uint32_t)qhat;
        }
        uint32_t r32[8]; memset(r32, 0, sizeof r32);
        for (int i = 0; i < n; ++i) r32[i] = (un[i] >> s) | ((s && i + 1 <= 8) ? (uint32_t)((uint64_t)un[i + 1] << (32 - s)) : 0);
        for (int i = 0; i < 4; ++i) rem->l[i] = (uint64_t)r32[2 * i] | ((uint64_t)r32[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; ++i) q->l[i] = (uint64_t)qd[2 * i] | ((uint64_t)qd[2 * i + 1] << 32);
}

/* ---- node types (src/graph.rs:236-245; wire codes protos/messages.proto:5-35) ---------------- */
enum { N_INPUT = 0, N_CONST = 1, N_UNO = 2, N_DUO = 3, N_TRES = 4 };
enum { OP_MUL = 0, OP_DIV, OP_ADD, OP_SUB, OP_POW, OP_IDIV, OP_MOD, OP_EQ, OP_NEQ, OP_LT, OP_GT, OP_LEQ, OP_GEQ,
       OP_LAND, OP_LOR, OP_SHL, OP_SHR, OP_BOR, OP_BAND, OP_BXOR };
typedef struct { uint8_t kind, op; uint32_t a, b, c; fr_t k; } node_t; /* k: MontConstant / input idx in a */

/* error codes: where the reference panics */
enum { ORC_OK = 0, ORC_E_SHL_OVERFLOW = 1, ORC_E_BITOP_EQ_R = 2, ORC_E_UNIMPL = 3, ORC_E_FORMAT = 4, ORC_E_INDEX = 5 };

static int is_neg(const fr_t *x) { return u256_cmp(&FR_HALF, x) < 0; } /* src/graph.rs:724 */
static const fr_t FR_ZERO = {{0, 0, 0, 0}};
static void fr_bool(fr_t *r, int v) { *r = v ? FR_R : FR_ZERO; } /* Fr::one()/zero(), Fr::new(0|1) */

/* signed compare family, src/graph.rs:723-769: returns -1/0/1 ordering with sign rule applied */
static int s_cmp(const fr_t *a, const fr_t *b) {
    int an = is_neg(a), bn = is_neg(b);
    if (an == bn) return u256_cmp(a, b);
    return an ? -1 : 1;
}

static int eval_duo(int op, const fr_t *a, const fr_t *b, fr_t *r) {
    fr_t x, y, t;
    switch (op) {
    case OP_MUL: fr_mul(r, a, b); return 0;                                   /* graph.rs:105 */
    case OP_DIV: if (u256_is_zero(b)) { *r = FR_ZERO; return 0; }             /* :109 */
                 fr_inv(&t, b); fr_mul(r, a, &t); return 0;
    case OP_ADD: fr_add(r, a, b); return 0;                                   /* :110 */
    case OP_SUB: fr_sub(r, a, b); return 0;                                   /* :111 */
    case OP_IDIV: case OP_MOD:                                                /* :112-121 */
        if (u256_is_zero(b)) { *r = FR_ZERO; return 0; }
        fr_to_u256(&x, a); fr_to_u256(&y, b);
        { fr_t q, rm; u256_divrem(&q, &rm, &x, &y); fr_from_u256(r, op == OP_IDIV ? &q : &rm); }
        return 0;
    case OP_EQ:  fr_bool(r, u256_cmp(a, b) == 0); return 0;                   /* :122-125 */
    case OP_NEQ: fr_bool(r, u256_cmp(a, b) != 0); return 0;                   /* :126-129 */
    case OP_LT: case OP_GT: case OP_LEQ: case OP_GEQ: {                       /* :130-133 */
        fr_to_u256(&x, a); fr_to_u256(&y, b);
        int c = s_cmp(&x, &y);
        fr_bool(r, op == OP_LT ? c < 0 : op == OP_GT ? c > 0 : op == OP_LEQ ? c <= 0 : c >= 0);
        return 0; }
    case OP_LAND: fr_bool(r, !u256_is_zero(a) && !u256_is_zero(b)); return 0; /* :134 */
    case OP_LOR:  fr_bool(r, !u256_is_zero(a) || !u256_is_zero(b)); return 0; /* :135 */
    case OP_SHL: case OP_SHR: {                                               /* :621-672 */
        if (u256_is_zero(b)) { *r = *a; return 0; }
        fr_to_u256(&y, b);
        if (y.l[1] | y.l[2] | y.l[3] || y.l[0] >= 254) { *r = FR_ZERO; return 0; }
        unsigned n = (unsigned)y.l[0], w = n / 64, s = n % 64;
        fr_to_u256(&x, a);
        memset(&t, 0, sizeof t);
        if (op == OP_SHL) {
            for (int i = 3; i >= (int)w; --i) {
                t.l[i] = x.l[i - w] << s;
                if (s && i - (int)w - 1 >= 0) t.l[i] |= x.l[i - w - 1] >> (64 - s);
            }
            if (u256_cmp(&t, &FR_P) >= 0) return ORC_E_SHL_OVERFLOW;          /* :634 unwrap */
        } else {
            for (int i = 0; i + (int)w < 4; ++i) {
                t.l[i] = x.l[i + w] >> s;
                if (s && i + w + 1 < 4) t.l[i] |= x.l[i + w + 1] << (64 - s);
            }
        }
        fr_from_u256(r, &t); return 0; }
    case OP_BOR: case OP_BAND: case OP_BXOR: {                                /* :674-717 */
        fr_to_u256(&x, a); fr_to_u256(&y, b);
        for (int i = 0; i < 4; ++i)
            t.l[i] = op == OP_BOR ? (x.l[i] | y.l[i]) : op == OP_BAND ? (x.l[i] & y.l[i]) : (x.l[i] ^ y.l[i]);
        if (u256_cmp(&t, &FR_P) > 0) u256_sub(&t, &t, &FR_P);
        if (u256_cmp(&t, &FR_P) >= 0) return ORC_E_BITOP_EQ_R;                /* from_bigint == None */
        fr_from_u256(r, &t); return 0; }
    default: return ORC_E_UNIMPL;                                             /* :141-142 (Pow) */
    }
}

static int eval_uno(int op, const fr_t *a, fr_t *r) {                         /* graph.rs:188-197 */
    if (op != 0) return ORC_E_UNIMPL;
    if (u256_is_zero(a)) { *r = FR_ZERO; return 0; }
    fr_t x, t; fr_to_u256(&x, a); u256_sub(&t, &FR_P, &x); fr_from_u256(r, &t);
    return 0;
}

/* ---- graph handle ----------------------------------------------------------------------------- */
typedef struct {
    node_t *nodes; uint64_t n_nodes;
    uint32_t *witness; uint64_t n_witness;
    uint64_t n_inputs;       /* get_inputs_size, src/lib.rs:138-152 */
    uint64_t n_op;           /* Uno+Duo+Tres nodes */
    char *md; uint64_t md_len; /* raw GraphMetadata bytes (input map decoded by the caller) */
} graph_t;

static int rd_varint(const uint8_t *p, uint64_t len, uint64_t *pos, uint64_t *out) {
    uint64_t v = 0; int sh = 0;
    while (*pos < len) {
        uint8_t b = p[(*pos)++];
        v |= (uint64_t)(b & 0x7f) << sh;
        if (!(b & 0x80)) { *out = v; return 0; }
        sh += 7; if (sh > 63) return -1;
    }
    return -1;
}
/* iterate fields of a message; returns 0 at end, 1 on field, -1 on error */
static int pb_next(const uint8_t *p, uint64_t len, uint64_t *pos, uint32_t *fno, int *wt, uint64_t *ival,
                   const uint8_t **bp, uint64_t *blen) {
    if (*pos >= len) return 0;
    uint64_t key;
    if (rd_varint(p, len, pos, &key)) return -1;
    *fno = (uint32_t)(key >> 3); *wt = (int)(key & 7);
    if (*wt == 0) { if (rd_varint(p, len, pos, ival)) return -1; }
    else if (*wt == 2) { uint64_t l; if (rd_varint(p, len, pos, &l)) return -1; if (*pos + l > len) return -1;
                         *bp = p + *pos; *blen = l; *pos += l; }
    else if (*wt == 1) { if (*pos + 8 > len) return -1; *pos += 8; }
    else if (*wt == 5) { if (*pos + 4 > len) return -1; *pos += 4; }
    else return -1;
    return 1;
}
static int pb_ints(const uint8_t *p, uint64_t len, uint64_t out[5]) {
    uint64_t pos = 0, iv = 0, bl; uint32_t f; int wt, r; const uint8_t *bp;
    memset(out, 0, 5 * sizeof(uint64_t));
    while ((r = pb_next(p, len, &pos, &f, &wt, &iv, &bp, &bl)) == 1) if (wt == 0 && f < 5) out[f] = iv;
    return r;
}
/* Fr::from_le_bytes_mod_order (src/storage.rs:28): arbitrary length, reduced mod r, to Montgomery */
static void const_from_le(fr_t *out, const uint8_t *b, uint64_t n) {
    /* Horner over bytes from the most significant end, in the field: acc = acc*256 + byte */
    fr_t acc = FR_ZERO, c256, t, m256;
    fr_t two56 = {{256, 0, 0, 0}};
    fr_from_u256(&m256, &two56);
    for (uint64_t i = n; i-- > 0;) {
        fr_mul(&t, &acc, &m256);
        fr_t d = {{b[i], 0, 0, 0}};
        fr_from_u256(&c256, &d);
        fr_add(&acc, &t, &c256);
    }
    *out = acc;
}

void orc_graph_free(graph_t *g) { if (g) { free(g->nodes); free(g->witness); free(g->md); free(g); } }

/* deserialize_witnesscalc_graph, src/storage.rs:214-249 */
graph_t *orc_graph_load(const uint8_t *data, uint64_t len, int *err) {
    static const char MAGIC[] = "wtns.graph.001";
    *err = ORC_E_FORMAT;
    if (len < 14 + 8 || memcmp(data, MAGIC, 14)) return NULL;
    uint64_t pos = 14, n = 0;
    memcpy(&n, data + pos, 8); pos += 8;
    if (n > len) return NULL;
    graph_t *g = calloc(1, sizeof *g);
    g->nodes = calloc(n ? n : 1, sizeof(node_t)); g->n_nodes = n;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t ml;
        if (rd_varint(data, len, &pos, &ml) || pos + ml > len) goto bad;
        const uint8_t *mp = data + pos; pos += ml;
        uint64_t p2 = 0, iv = 0, bl = 0; uint32_t f; int wt, r, got = 0; const uint8_t *bp = NULL;
        node_t *nd = &g->nodes[i];
        while ((r = pb_next(mp, ml, &p2, &f, &wt, &iv, &bp, &bl)) == 1) {
            if (wt != 2) continue;
            uint64_t v[5];
            if (f == 1) { if (pb_ints(bp, bl, v) < 0) goto bad; nd->kind = N_INPUT; nd->a = (uint32_t)v[1]; got = 1; }
            else if (f == 2) {
                const uint8_t *vb = NULL; uint64_t vl = 0;
                uint64_t p3 = 0, iv3, bl3; uint32_t f3; int w3; const uint8_t *bp3;
                while (pb_next(bp, bl, &p3, &f3, &w3, &iv3, &bp3, &bl3) == 1)
                    if (f3 == 1 && w3 == 2) {
                        uint64_t p4 = 0, iv4, bl4; uint32_t f4; int w4; const uint8_t *bp4;
                        while (pb_next(bp3, bl3, &p4, &f4, &w4, &iv4, &bp4, &bl4) == 1)
                            if (f4 == 1 && w4 == 2) { vb = bp4; vl = bl4; }
                    }
                nd->kind = N_CONST; const_from_le(&nd->k, vb, vl); got = 1;
            }
            else if (f == 3) { if (pb_ints(bp, bl, v) < 0) goto bad; nd->kind = N_UNO; nd->op = (uint8_t)v[1]; nd->a = (uint32_t)v[2]; got = 1; if (v[1] > 1) goto bad; }
            else if (f == 4) { if (pb_ints(bp, bl, v) < 0) goto bad; nd->kind = N_DUO; nd->op = (uint8_t)v[1]; nd->a = (uint32_t)v[2]; nd->b = (uint32_t)v[3]; got = 1; if (v[1] > 19) goto bad; }
            else if (f == 5) { if (pb_ints(bp, bl, v) < 0) goto bad; nd->kind = N_TRES; nd->op = (uint8_t)v[1]; nd->a = (uint32_t)v[2]; nd->b = (uint32_t)v[3]; nd->c = (uint32_t)v[4]; got = 1; if (v[1] > 0) goto bad; }
        }
        if (r < 0 || !got) goto bad;
        if (nd->kind >= N_UNO) g->n_op++;
    }
    {   /* GraphMetadata */
        uint64_t ml;
        if (rd_varint(data, len, &pos, &ml) || pos + ml > len) goto bad;
        g->md = malloc(ml ? ml : 1); memcpy(g->md, data + pos, ml); g->md_len = ml;
        const uint8_t *mp = data + pos;
        uint64_t cap = 16, p2 = 0, iv = 0, bl = 0; uint32_t f; int wt, r; const uint8_t *bp = NULL;
        g->witness = malloc(cap * sizeof(uint32_t));
        while ((r = pb_next(mp, ml, &p2, &f, &wt, &iv, &bp, &bl)) == 1) {
            if (f != 1) continue;
            if (wt == 0) { if (g->n_witness == cap) g->witness = realloc(g->witness, (cap *= 2) * sizeof(uint32_t)); g->witness[g->n_witness++] = (uint32_t)iv; }
            else if (wt == 2) { uint64_t p3 = 0, x;
                while (p3 < bl) { if (rd_varint(bp, bl, &p3, &x)) goto bad;
                    if (g->n_witness == cap) g->witness = realloc(g->witness, (cap *= 2) * sizeof(uint32_t));
                    g->witness[g->n_witness++] = (uint32_t)x; } }
        }
        if (r < 0) goto bad;
    }
    {   /* get_inputs_size, src/lib.rs:138-152 */
        int start = 0; uint64_t mx = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (g->nodes[i].kind == N_INPUT) { if (g->nodes[i].a > mx) mx = g->nodes[i].a; start = 1; }
            else if (start) break;
        }
        g->n_inputs = mx + 1;
    }
    *err = 0;
    return g;
bad:
    orc_graph_free(g);
    return NULL;
}

void orc_graph_info(const graph_t *g, uint64_t out[4]) { out[0] = g->n_nodes; out[1] = g->n_witness; out[2] = g->n_inputs; out[3] = g->n_op; }
uint64_t orc_graph_metadata(const graph_t *g, const char **p) { *p = g->md; return g->md_len; }
void orc_graph_witness(const graph_t *g, uint32_t *out) { memcpy(out, g->witness, g->n_witness * sizeof(uint32_t)); }

/* graph::evaluate, src/graph.rs:367-391.  inputs: n_inputs x 32 B canonical LE; out: n_witness x 32 B
 * canonical LE.  `values` is caller-provided scratch of n_nodes fr_t (NULL -> malloc per call, as the
 * reference's Vec::with_capacity does). Returns 0 or the first "panic" code (+ node index in *bad). */
int orc_evaluate(const graph_t *g, const uint8_t *inputs, uint64_t n_inputs, uint8_t *out, fr_t *values, uint64_t *bad) {
    int own = 0, rc = 0;
    if (!values) { values = malloc((g->n_nodes ? g->n_nodes : 1) * sizeof(fr_t)); own = 1; }
    for (uint64_t i = 0; i < g->n_nodes; ++i) {
        const node_t *nd = &g->nodes[i];
        switch (nd->kind) {
        case N_CONST: values[i] = nd->k; break;                                       /* :375 */
        case N_INPUT: { fr_t x;                                                       /* :376 */
            if (nd->a >= n_inputs) { rc = ORC_E_INDEX; goto done_bad; }
            memcpy(&x, inputs + 32 * (uint64_t)nd->a, 32); fr_from_u256(&values[i], &x); break; }
        case N_DUO: if (nd->a >= i || nd->b >= i) { rc = ORC_E_INDEX; goto done_bad; }
            rc = eval_duo(nd->op, &values[nd->a], &values[nd->b], &values[i]); if (rc) goto done_bad; break;
        case N_UNO: if (nd->a >= i) { rc = ORC_E_INDEX; goto done_bad; }
            rc = eval_uno(nd->op, &values[nd->a], &values[i]); if (rc) goto done_bad; break;
        case N_TRES: if (nd->a >= i || nd->b >= i || nd->c >= i) { rc = ORC_E_INDEX; goto done_bad; }
            values[i] = u256_is_zero(&values[nd->a]) ? values[nd->c] : values[nd->b]; break; /* :221-225 */
        }
        continue;
    done_bad:
        if (bad) *bad = i;
        if (own) free(values);
        return rc;
    }
    for (uint64_t i = 0; i < g->n_witness; ++i) {                                     /* :385-388 */
        fr_t x;
        if (g->witness[i] >= g->n_nodes) { if (own) free(values); if (bad) *bad = i; return ORC_E_INDEX; }
        fr_to_u256(&x, &values[g->witness[i]]);
        memcpy(out + 32 * i, &x, 32);
    }
    if (own) free(values);
    return 0;
}

/* Evaluate `count` input sets one after another on the calling thread (bench.py cpu_baseline, B1).
 * status[s] = per-set return code. */
int orc_evaluate_batch(const graph_t *g, const uint8_t *inputs, uint64_t n_inputs, uint64_t count, uint8_t *out, int32_t *status) {
    fr_t *values = malloc((g->n_nodes ? g->n_nodes : 1) * sizeof(fr_t));
    int any = 0;
    for (uint64_t s = 0; s < count; ++s) {
        uint64_t bad;
        int rc = orc_evaluate(g, inputs + s * n_inputs * 32, n_inputs, out ? out + s * g->n_witness * 32 : NULL, values, &bad);
        if (status) status[s] = rc;
        any |= rc;
    }
    free(values);
    return any;
}

/* The same loop spread over `n_threads` host threads, contiguous shares of the batch (bench.py cpu_baseline, the
 * all-cores courtesy figure of SURVEY 8(d)(ii); the reference itself is single-threaded). */
typedef struct { const graph_t *g; const uint8_t *inputs; uint64_t n_inputs, lo, hi; uint8_t *out; int32_t *status; int any; } orc_share_t;
static void *orc_share_run(void *arg) {
    orc_share_t *s = (orc_share_t *)arg;
    if (s->hi > s->lo)
        s->any = orc_evaluate_batch(s->g, s->inputs + s->lo * s->n_inputs * 32, s->n_inputs, s->hi - s->lo,
                                    s->out ? s->out + s->lo * s->g->n_witness * 32 : NULL, s->status ? s->status + s->lo : NULL);
    return NULL;
}
int orc_evaluate_batch_threads(const graph_t *g, const uint8_t *inputs, uint64_t n_inputs, uint64_t count, uint8_t *out,
                               int32_t *status, uint32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t *th = malloc(n_threads * sizeof(pthread_t));
    orc_share_t *sh = calloc(n_threads, sizeof(orc_share_t));
    int any = 0;
    for (uint32_t i = 0; i < n_threads; ++i) {
        sh[i] = (orc_share_t){g, inputs, n_inputs, count * i / n_threads, count * (i + 1) / n_threads, out, status, 0};
        if (pthread_create(&th[i], NULL, orc_share_run, &sh[i]) != 0) {  /* run the share here instead */
            orc_share_run(&sh[i]);
            th[i] = 0;
            sh[i].lo = sh[i].hi = 0;
            sh[i].any |= 0x100;
        }
    }
    for (uint32_t i = 0; i < n_threads; ++i) {
        if (!(sh[i].any & 0x100)) pthread_join(th[i], NULL);
        any |= sh[i].any & 0xff;
    }
    free(th);
    free(sh);
    return any;
}

/* single-op known-answer interface: operands and result canonical 32-byte LE; kind: 2 uno, 3 duo, 4 tres */
int orc_eval_op(int kind, int op, const uint8_t *a, const uint8_t *b, const uint8_t *c, uint8_t *out) {
    fr_t x, y, z, ma, mb, mc, r; int rc = 0;
    memcpy(&x, a, 32); fr_from_u256(&ma, &x);
    if (b) { memcpy(&y, b, 32); fr_from_u256(&mb, &y); }
    if (c) { memcpy(&z, c, 32); fr_from_u256(&mc, &z); }
    if (kind == N_UNO) rc = eval_uno(op, &ma, &r);
    else if (kind == N_DUO) rc = eval_duo(op, &ma, &mb, &r);
    else r = u256_is_zero(&ma) ? mc : mb;
    if (rc) return rc;
    fr_to_u256(&x, &r); memcpy(out, &x, 32);
    return 0;
}

/* wtns_from_witness, src/lib.rs:114-123 (+ wtns-file 0.1.5 layout [ext]); out must hold 76 + 32*n bytes */
uint64_t orc_wtns_from_witness(const uint8_t *witness, uint64_t n, uint8_t *out) {
    uint8_t *p = out; uint32_t u; uint64_t q;
    memcpy(p, "wtns", 4); p += 4;
    u = 2; memcpy(p, &u, 4); p += 4;          /* version (forced to 2, lib.rs:118) */
    u = 2; memcpy(p, &u, 4); p += 4;          /* n sections */
    u = 1; memcpy(p, &u, 4); p += 4; q = 40; memcpy(p, &q, 8); p += 8;
    u = 32; memcpy(p, &u, 4); p += 4;         /* n8 */
    memcpy(p, &FR_P, 32); p += 32;            /* prime (lib.rs:117) */
    u = (uint32_t)n; memcpy(p, &u, 4); p += 4;
    u = 2; memcpy(p, &u, 4); p += 4; q = 32 * n; memcpy(p, &q, 8); p += 8;
    memcpy(p, witness, 32 * n); p += 32 * n;
    return (uint64_t)(p - out);
}
