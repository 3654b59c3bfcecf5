// Host-side formats: `.bin` graph container, inputs JSON, `.wtns` (see graph.hpp for reference cites).
#include "graph.hpp"

#include <string.h>

#include <thread>

#include <algorithm>

namespace cwc {

static const char kMagic[] = "wtns.graph.001";  // reference src/storage.rs:16
static const size_t kMagicLen = 14;

// ---------------------------------------------------------------------------------------------
// protobuf wire helpers (schema: reference protos/messages.proto)
// ---------------------------------------------------------------------------------------------
namespace {
struct Cursor {
    const uint8_t* p;
    size_t len, pos;
    bool varint(uint64_t& out) {
        uint64_t v = 0;
        int sh = 0;
        while (pos < len) {
            uint8_t b = p[pos++];
            v |= (uint64_t)(b & 0x7f) << sh;
            if (!(b & 0x80)) {
                out = v;
                return true;
            }
            sh += 7;
            if (sh > 63) return false;
        }
        return false;
    }
};
struct Field {
    uint32_t no;
    int wt;
    uint64_t ival;
    const uint8_t* bp;
    size_t blen;
};
// 1 = field read, 0 = end of message, -1 = malformed
int next_field(Cursor& c, Field& f) {
    if (c.pos >= c.len) return 0;
    uint64_t key;
    if (!c.varint(key)) return -1;
    f.no = (uint32_t)(key >> 3);
    f.wt = (int)(key & 7);
    f.ival = 0;
    f.bp = nullptr;
    f.blen = 0;
    switch (f.wt) {
        case 0: return c.varint(f.ival) ? 1 : -1;
        case 2: {
            uint64_t l;
            if (!c.varint(l) || l > c.len - c.pos) return -1;
            f.bp = c.p + c.pos;
            f.blen = (size_t)l;
            c.pos += (size_t)l;
            return 1;
        }
        case 1:
            if (c.len - c.pos < 8) return -1;
            c.pos += 8;
            return 1;
        case 5:
            if (c.len - c.pos < 4) return -1;
            c.pos += 4;
            return 1;
        default: return -1;
    }
}
bool read_ints(const uint8_t* p, size_t len, uint64_t out[5]) {
    for (int i = 0; i < 5; ++i) out[i] = 0;
    Cursor c{p, len, 0};
    Field f;
    int r;
    while ((r = next_field(c, f)) == 1)
        if (f.wt == 0 && f.no < 5) out[f.no] = f.ival;
    return r == 0;
}
void put_varint(std::vector<uint8_t>& o, uint64_t v) {
    while (v >= 0x80) {
        o.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    o.push_back((uint8_t)v);
}
void put_uint_field(std::vector<uint8_t>& o, uint32_t no, uint64_t v) {
    if (v == 0) return;  // proto3 default elision (as prost)
    put_varint(o, (uint64_t)no << 3);
    put_varint(o, v);
}
void put_bytes_field(std::vector<uint8_t>& o, uint32_t no, const std::vector<uint8_t>& b) {
    put_varint(o, ((uint64_t)no << 3) | 2);
    put_varint(o, b.size());
    o.insert(o.end(), b.begin(), b.end());
}
}  // namespace

Fr u256_from_le_bytes_mod_order(const uint8_t* b, size_t n) {
    // Fr::from_le_bytes_mod_order (reference src/storage.rs:28): any length, value mod r.
    // Horner from the most significant byte: acc = acc*256 + byte (mod r), canonical arithmetic only.
    Fr acc = fr_zero();
    for (size_t i = n; i-- > 0;) {
        for (int k = 0; k < 8; ++k) acc = fr_add(acc, acc);
        Fr d = fr_zero();
        d.v[0] = b[i];
        acc = fr_add(acc, d);
    }
    return acc;
}

bool deserialize_witnesscalc_graph(const uint8_t* data, size_t len, Graph& g, std::string& err) {
    g = Graph();
    if (len < kMagicLen + 8 || memcmp(data, kMagic, kMagicLen) != 0) {
        err = "Invalid magic";
        return false;
    }
    uint64_t n_nodes;
    memcpy(&n_nodes, data + kMagicLen, 8);  // u64 LE (storage.rs:228)
    Cursor c{data, len, kMagicLen + 8};
    if (n_nodes > len) {  // every node record takes at least 2 bytes
        err = "node count exceeds file size";
        return false;
    }
    if (n_nodes > 0xffffffffull) {
        err = "more than 2^32 nodes";
        return false;
    }
    g.nodes.reserve((size_t)n_nodes);
    for (uint64_t i = 0; i < n_nodes; ++i) {
        uint64_t ml;
        if (!c.varint(ml) || ml > c.len - c.pos) {
            err = "Unexpected EOF in node " + std::to_string(i);
            return false;
        }
        Cursor m{data + c.pos, (size_t)ml, 0};
        c.pos += (size_t)ml;
        Field f;
        int r;
        bool got = false;
        Node nd{0, 0, 0, 0, 0};
        uint64_t v[5];
        while ((r = next_field(m, f)) == 1) {
            if (f.wt != 2 || f.no < 1 || f.no > 5) continue;  // unknown fields are skipped
            switch (f.no) {
                case 1:
                    if (!read_ints(f.bp, f.blen, v)) goto malformed;
                    nd = Node{N_INPUT, 0, (uint32_t)v[1], 0, 0};
                    break;
                case 2: {
                    const uint8_t* vb = nullptr;
                    size_t vl = 0;
                    Cursor c2{f.bp, f.blen, 0};
                    Field f2;
                    int r2;
                    while ((r2 = next_field(c2, f2)) == 1)
                        if (f2.no == 1 && f2.wt == 2) {
                            Cursor c3{f2.bp, f2.blen, 0};
                            Field f3;
                            int r3;
                            while ((r3 = next_field(c3, f3)) == 1)
                                if (f3.no == 1 && f3.wt == 2) {
                                    vb = f3.bp;
                                    vl = f3.blen;
                                }
                            if (r3 < 0) goto malformed;
                        }
                    if (r2 < 0) goto malformed;
                    nd = Node{N_CONST, 0, (uint32_t)g.const_values.size(), 0, 0};
                    g.const_values.push_back(u256_from_le_bytes_mod_order(vb, vl));
                    break;
                }
                case 3:
                    if (!read_ints(f.bp, f.blen, v)) goto malformed;
                    if (v[1] > UOP_ID) {
                        err = "unknown UnoOp code in node " + std::to_string(i);
                        return false;
                    }
                    nd = Node{N_UNO, (uint8_t)v[1], (uint32_t)v[2], 0, 0};
                    break;
                case 4:
                    if (!read_ints(f.bp, f.blen, v)) goto malformed;
                    if (v[1] >= OP_DUO_COUNT) {
                        err = "unknown DuoOp code in node " + std::to_string(i);
                        return false;
                    }
                    nd = Node{N_DUO, (uint8_t)v[1], (uint32_t)v[2], (uint32_t)v[3], 0};
                    break;
                case 5:
                    if (!read_ints(f.bp, f.blen, v)) goto malformed;
                    if (v[1] > TOP_TERNCOND) {
                        err = "unknown TresOp code in node " + std::to_string(i);
                        return false;
                    }
                    nd = Node{N_TRES, (uint8_t)v[1], (uint32_t)v[2], (uint32_t)v[3], (uint32_t)v[4]};
                    break;
            }
            got = true;
        }
        if (r < 0) goto malformed;
        if (!got) {
            err = "node " + std::to_string(i) + " has no variant set";  // value.node.unwrap() storage.rs:22
            return false;
        }
        if (nd.kind >= N_UNO) g.n_op++;
        g.nodes.push_back(nd);
        continue;
    malformed:
        err = "malformed protobuf in node " + std::to_string(i);
        return false;
    }
    {   // GraphMetadata (messages.proto:82-85)
        uint64_t ml;
        if (!c.varint(ml) || ml > c.len - c.pos) {
            err = "Unexpected EOF in graph metadata";
            return false;
        }
        Cursor m{data + c.pos, (size_t)ml, 0};
        Field f;
        int r;
        while ((r = next_field(m, f)) == 1) {
            if (f.no == 1 && f.wt == 0) {
                g.witness_signals.push_back((uint32_t)f.ival);
            } else if (f.no == 1 && f.wt == 2) {  // packed
                Cursor pk{f.bp, f.blen, 0};
                while (pk.pos < pk.len) {
                    uint64_t x;
                    if (!pk.varint(x)) {
                        err = "malformed witnessSignals";
                        return false;
                    }
                    g.witness_signals.push_back((uint32_t)x);
                }
            } else if (f.no == 2 && f.wt == 2) {  // map<string, SignalDescription> entry
                InputSignal sig{"", 0, 0};
                Cursor e{f.bp, f.blen, 0};
                Field f2;
                int r2;
                while ((r2 = next_field(e, f2)) == 1) {
                    if (f2.no == 1 && f2.wt == 2) sig.name.assign((const char*)f2.bp, f2.blen);
                    if (f2.no == 2 && f2.wt == 2) {
                        uint64_t v[5];
                        if (!read_ints(f2.bp, f2.blen, v)) {
                            err = "malformed SignalDescription";
                            return false;
                        }
                        sig.offset = (uint32_t)v[1];
                        sig.len = (uint32_t)v[2];
                    }
                }
                if (r2 < 0) {
                    err = "malformed inputs map entry";
                    return false;
                }
                auto it = g.input_index.find(sig.name);
                if (it != g.input_index.end()) {
                    g.inputs[it->second] = sig;  // protobuf map semantics: last entry wins
                } else {
                    g.input_index[sig.name] = (uint32_t)g.inputs.size();
                    g.inputs.push_back(sig);
                }
            }
        }
        if (r < 0) {
            err = "malformed graph metadata";
            return false;
        }
    }
    return true;
}

std::vector<uint8_t> serialize_witnesscalc_graph(const Graph& g) {
    std::vector<uint8_t> out(kMagic, kMagic + kMagicLen);
    uint64_t n = g.nodes.size();
    out.insert(out.end(), (uint8_t*)&n, (uint8_t*)&n + 8);
    std::vector<uint8_t> body, inner, inner2;
    for (const Node& nd : g.nodes) {
        body.clear();
        inner.clear();
        switch (nd.kind) {
            case N_INPUT:
                put_uint_field(inner, 1, nd.a);
                put_bytes_field(body, 1, inner);
                break;
            case N_CONST: {
                const Fr& v = g.const_values[nd.a];
                std::vector<uint8_t> le;
                for (int i = 0; i < 8; ++i)
                    for (int k = 0; k < 4; ++k) le.push_back((uint8_t)(v.v[i] >> (8 * k)));
                while (le.size() > 1 && le.back() == 0) le.pop_back();  // num-bigint to_bytes_le: minimal, zero = [0]
                inner2.clear();
                put_bytes_field(inner2, 1, le);
                put_bytes_field(inner, 1, inner2);
                put_bytes_field(body, 2, inner);
                break;
            }
            case N_UNO:
                put_uint_field(inner, 1, nd.op);
                put_uint_field(inner, 2, nd.a);
                put_bytes_field(body, 3, inner);
                break;
            case N_DUO:
                put_uint_field(inner, 1, nd.op);
                put_uint_field(inner, 2, nd.a);
                put_uint_field(inner, 3, nd.b);
                put_bytes_field(body, 4, inner);
                break;
            case N_TRES:
                put_uint_field(inner, 1, nd.op);
                put_uint_field(inner, 2, nd.a);
                put_uint_field(inner, 3, nd.b);
                put_uint_field(inner, 4, nd.c);
                put_bytes_field(body, 5, inner);
                break;
        }
        put_varint(out, body.size());
        out.insert(out.end(), body.begin(), body.end());
    }
    uint64_t md_off = out.size();
    std::vector<uint8_t> md;
    if (!g.witness_signals.empty()) {
        inner.clear();
        for (uint32_t w : g.witness_signals) put_varint(inner, w);
        put_bytes_field(md, 1, inner);
    }
    for (const InputSignal& s : g.inputs) {
        inner.clear();
        put_bytes_field(inner, 1, std::vector<uint8_t>(s.name.begin(), s.name.end()));
        inner2.clear();
        put_uint_field(inner2, 1, s.offset);
        put_uint_field(inner2, 2, s.len);
        put_bytes_field(inner, 2, inner2);
        put_bytes_field(md, 2, inner);
    }
    put_varint(out, md.size());
    out.insert(out.end(), md.begin(), md.end());
    out.insert(out.end(), (uint8_t*)&md_off, (uint8_t*)&md_off + 8);  // storage.rs:180
    return out;
}

size_t get_inputs_size(const Graph& g) {  // lib.rs:138-152
    bool start = false;
    size_t mx = 0;
    for (const Node& n : g.nodes) {
        if (n.kind == N_INPUT) {
            if (n.a > mx) mx = n.a;
            start = true;
        } else if (start) {
            break;
        }
    }
    return mx + 1;
}

size_t inputs_buffer_size(const Graph& g) {
    size_t n = get_inputs_size(g);
    for (const Node& nd : g.nodes)
        if (nd.kind == N_INPUT) n = std::max(n, (size_t)nd.a + 1);
    for (const InputSignal& s : g.inputs) n = std::max(n, (size_t)s.offset + s.len);
    return n;
}

// ---------------------------------------------------------------------------------------------
// inputs JSON (lib.rs:195-247).  A small strict JSON reader: the reference parses the whole document
// with serde_json first (invalid JSON -> panic there, error here), then classifies the values.
// ---------------------------------------------------------------------------------------------
// U256::from_str_radix(s, 10) [ext: ruint 1.12]: digits (underscores skipped), must fit 256 bits.  Nine digits per
// multiply-add pass over the limbs (the batched JSON front-end parses ~10^5 such numbers per thousand input sets).
bool u256_parse_dec_span(const char* s, size_t n, Fr& out, std::string& err) {
    static const uint32_t kPow10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};
    Fr acc = fr_zero();
    uint32_t chunk = 0, digits = 0;
    auto flush = [&]() -> bool {
        uint64_t c = chunk;
        const uint64_t m = kPow10[digits];
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)acc.v[i] * m;
            acc.v[i] = (uint32_t)c;
            c >>= 32;
        }
        chunk = 0;
        digits = 0;
        return c == 0;
    };
    for (size_t k = 0; k < n; ++k) {
        const char ch = s[k];
        if (ch == '_') continue;
        if (ch < '0' || ch > '9') {
            err = std::string("InputFieldNumberParseError(InvalidDigit('") + ch + "'))";
            return false;
        }
        chunk = chunk * 10u + (uint32_t)(ch - '0');
        if (++digits == 9 && !flush()) {
            // (which of two errors in ONE string is named can differ from a digit-serial parser when an invalid character
            // follows the overflowing digit inside the same group of nine; either way the input is refused)
            err = "InputFieldNumberParseError(BaseOverflow)";
            return false;
        }
    }
    if (digits && !flush()) {
        err = "InputFieldNumberParseError(BaseOverflow)";
        return false;
    }
    out = acc;
    return true;
}
bool u256_parse_dec(const std::string& s, Fr& out, std::string& err) { return u256_parse_dec_span(s.data(), s.size(), out, err); }

namespace {
struct Json {
    const char* p;
    size_t len, pos;
    std::string err;
    void ws() {
        while (pos < len && (p[pos] == ' ' || p[pos] == '\t' || p[pos] == '\n' || p[pos] == '\r')) ++pos;
    }
    bool fail(const char* m) {
        if (err.empty()) err = std::string(m) + " at byte " + std::to_string(pos);
        return false;
    }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(uint32_t& v) {
        if (len - pos < 4) return fail("bad \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char ch = p[pos++];
            v <<= 4;
            if (ch >= '0' && ch <= '9') v |= (uint32_t)(ch - '0');
            else if (ch >= 'a' && ch <= 'f') v |= (uint32_t)(ch - 'a' + 10);
            else if (ch >= 'A' && ch <= 'F') v |= (uint32_t)(ch - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& out) {
        out.clear();
        if (pos >= len || p[pos] != '"') return fail("expected string");
        ++pos;
        while (pos < len) {
            unsigned char ch = (unsigned char)p[pos++];
            if (ch == '"') return true;
            if (ch < 0x20) return fail("control character in string");
            if (ch != '\\') {
                out.push_back((char)ch);
                continue;
            }
            if (pos >= len) break;
            char e = p[pos++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00) {
                        uint32_t lo;
                        if (len - pos < 2 || p[pos] != '\\' || p[pos + 1] != 'u') return fail("lone surrogate");
                        pos += 2;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("lone surrogate");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        return fail("lone surrogate");
                    }
                    utf8(out, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    // Number: is_u64 says whether serde_json's Number::is_u64() would hold (non-negative integer
    // literal without fraction/exponent that fits u64).
    bool number(bool& is_u64, uint64_t& val) {
        size_t s = pos;
        bool neg = false, integral = true;
        if (pos < len && p[pos] == '-') { neg = true; ++pos; }
        if (pos >= len) return fail("bad number");
        if (p[pos] == '0') ++pos;
        else if (p[pos] >= '1' && p[pos] <= '9') while (pos < len && p[pos] >= '0' && p[pos] <= '9') ++pos;
        else return fail("bad number");
        size_t int_end = pos;
        if (pos < len && p[pos] == '.') {
            integral = false;
            ++pos;
            if (pos >= len || p[pos] < '0' || p[pos] > '9') return fail("bad number");
            while (pos < len && p[pos] >= '0' && p[pos] <= '9') ++pos;
        }
        if (pos < len && (p[pos] == 'e' || p[pos] == 'E')) {
            integral = false;
            ++pos;
            if (pos < len && (p[pos] == '+' || p[pos] == '-')) ++pos;
            if (pos >= len || p[pos] < '0' || p[pos] > '9') return fail("bad number");
            while (pos < len && p[pos] >= '0' && p[pos] <= '9') ++pos;
        }
        is_u64 = false;
        val = 0;
        if (integral && !neg) {
            unsigned __int128 v = 0;
            bool ok = true;
            for (size_t i = s; i < int_end; ++i) {
                v = v * 10 + (unsigned)(p[i] - '0');
                if (v > (unsigned __int128)0xffffffffffffffffull) { ok = false; break; }
            }
            if (ok) { is_u64 = true; val = (uint64_t)v; }
        }
        return true;
    }
    bool literal(const char* w) {
        size_t n = strlen(w);
        if (len - pos < n || memcmp(p + pos, w, n) != 0) return fail("bad literal");
        pos += n;
        return true;
    }
    // skip any value (used for kinds the reference rejects after parsing)
    bool skip_value(int depth) {
        if (depth > 128) return fail("recursion limit exceeded");
        ws();
        if (pos >= len) return fail("unexpected end");
        char ch = p[pos];
        std::string tmp;
        if (ch == '"') return string(tmp);
        if (ch == '{') {
            ++pos; ws();
            if (pos < len && p[pos] == '}') { ++pos; return true; }
            for (;;) {
                ws();
                if (!string(tmp)) return false;
                ws();
                if (pos >= len || p[pos] != ':') return fail("expected ':'");
                ++pos;
                if (!skip_value(depth + 1)) return false;
                ws();
                if (pos < len && p[pos] == ',') { ++pos; continue; }
                if (pos < len && p[pos] == '}') { ++pos; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (ch == '[') {
            ++pos; ws();
            if (pos < len && p[pos] == ']') { ++pos; return true; }
            for (;;) {
                if (!skip_value(depth + 1)) return false;
                ws();
                if (pos < len && p[pos] == ',') { ++pos; continue; }
                if (pos < len && p[pos] == ']') { ++pos; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (ch == 't') return literal("true");
        if (ch == 'f') return literal("false");
        if (ch == 'n') return literal("null");
        bool u; uint64_t v;
        return number(u, v);
    }
};
}  // namespace

bool deserialize_inputs(const char* json, size_t len, InputList& out, std::string& err) {
    out.clear();
    Json j{json, len, 0, ""};
    // pass 1: the whole document must be valid JSON (serde_json::from_slice, lib.rs:196)
    if (!j.skip_value(0)) { err = "invalid JSON: " + j.err; return false; }
    j.ws();
    if (j.pos != j.len) { err = "invalid JSON: trailing characters at byte " + std::to_string(j.pos); return false; }
    // pass 2: classify
    j.pos = 0;
    j.ws();
    if (j.p[j.pos] != '{') { err = "InputsUnmarshal(\"inputs must be an object\")"; return false; }  // lib.rs:201
    ++j.pos;
    j.ws();
    if (j.p[j.pos] == '}') return true;
    std::unordered_map<std::string, size_t> seen;
    for (;;) {
        std::string key;
        j.ws();
        j.string(key);
        j.ws();
        ++j.pos;  // ':'
        j.ws();
        std::vector<Fr> vals;
        char ch = j.p[j.pos];
        auto scalar = [&](bool in_array) -> bool {
            char c2 = j.p[j.pos];
            if (c2 == '"') {
                // a string without escapes (every decimal number is one) is parsed where it stands
                size_t e = j.pos + 1;
                while (e < j.len && j.p[e] != '"' && j.p[e] != '\\') ++e;
                Fr v;
                if (e < j.len && j.p[e] == '"') {
                    if (!u256_parse_dec_span(j.p + j.pos + 1, e - j.pos - 1, v, err)) return false;  // lib.rs:208,223
                    j.pos = e + 1;
                } else {
                    std::string s;
                    j.string(s);
                    if (!u256_parse_dec(s, v, err)) return false;
                }
                vals.push_back(v);
                return true;
            }
            if (c2 == '-' || (c2 >= '0' && c2 <= '9')) {
                bool is_u64; uint64_t v;
                j.number(is_u64, v);
                if (!is_u64) { err = "InputsUnmarshal(\"signal value is not a positive integer\")"; return false; }  // :213,:227
                Fr f = fr_zero();
                f.v[0] = (uint32_t)v;
                f.v[1] = (uint32_t)(v >> 32);
                vals.push_back(f);
                return true;
            }
            if (in_array) err = "InputsUnmarshal(\"inputs must be a string: " + key + "\")";  // :232
            else err = "InputsUnmarshal(\"value for key " + key + " must be an a number as a string, as a number of an array of strings of numbers\")";  // :240-242
            return false;
        };
        if (ch == '[') {
            ++j.pos;
            j.ws();
            if (j.p[j.pos] == ']') {
                ++j.pos;
            } else {
                for (;;) {
                    j.ws();
                    if (!scalar(true)) return false;
                    j.ws();
                    if (j.p[j.pos] == ',') { ++j.pos; continue; }
                    ++j.pos;  // ']'
                    break;
                }
            }
        } else {
            if (!scalar(false)) return false;
        }
        auto it = seen.find(key);
        if (it != seen.end()) out[it->second].second = std::move(vals);  // duplicate key: last wins [ext: serde_json]
        else { seen[key] = out.size(); out.emplace_back(key, std::move(vals)); }
        j.ws();
        if (j.p[j.pos] == ',') { ++j.pos; continue; }
        break;  // '}'
    }
    return true;
}

namespace {
// [p, p + n) (no leading / trailing whitespace) is exactly one bracketed JSON value as far as brackets and strings go: the
// depth returns to zero at the last character and not before.  (What is inside is validated by deserialize_inputs.)
bool line_is_one_object(const char* p, size_t n) {
    if (n < 2 || p[0] != '{' || p[n - 1] != '}') return false;
    size_t depth = 0;
    bool in_str = false;
    for (size_t i = 0; i < n; ++i) {
        const char c = p[i];
        if (in_str) {
            if (c == '\\') ++i;
            else if (c == '"') in_str = false;
            continue;
        }
        if (c == '"') in_str = true;
        else if (c == '{' || c == '[') ++depth;
        else if (c == '}' || c == ']') {
            if (depth == 0) return false;
            if (--depth == 0 && i + 1 != n) return false;
        }
    }
    return depth == 0 && !in_str;
}
}  // namespace

bool split_inputs_batch(const char* text, size_t len, std::vector<std::pair<size_t, size_t>>& spans, std::string& err) {
    spans.clear();
    Json j{text, len, 0, ""};
    j.ws();
    // NDJSON as everybody writes it -- one object per line -- is split at the newlines (a raw newline is never inside a JSON
    // string) and the lines are checked in parallel; anything else (objects over several lines, several per line) takes
    // the serial scanner below.  (The serial scan of a 57 MB batch was most of the front-end's time on a 256-core host.)
    if (j.pos < len && text[j.pos] == '{' && len >= (1u << 16)) {
        std::vector<std::pair<size_t, size_t>> lines;
        size_t b = j.pos;
        while (b < len) {
            const char* nl = (const char*)memchr(text + b, '\n', len - b);
            size_t e = nl ? (size_t)(nl - text) : len, lo = b, hi = e;
            while (lo < hi && (text[lo] == ' ' || text[lo] == '\t' || text[lo] == '\r')) ++lo;
            while (hi > lo && (text[hi - 1] == ' ' || text[hi - 1] == '\t' || text[hi - 1] == '\r')) --hi;
            if (hi > lo) lines.emplace_back(lo, hi);
            b = e + 1;
        }
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 64) nt = 64;
        if (nt > lines.size() / 64 + 1) nt = (unsigned)(lines.size() / 64 + 1);
        if (nt < 1) nt = 1;
        std::vector<char> okv(nt, 1);
        auto work = [&](unsigned w) {
            for (size_t i = lines.size() * w / nt; i < lines.size() * (w + 1) / nt; ++i)
                if (!line_is_one_object(text + lines[i].first, lines[i].second - lines[i].first)) {
                    okv[w] = 0;
                    return;
                }
        };
        bool threads_ok = true;
        {
            std::vector<std::thread> th;
            unsigned started = 1;
            try {
                for (; started < nt; ++started) th.emplace_back(work, started);
            } catch (...) {
                threads_ok = true;  // (the ranges of threads that could not be started are checked here)
            }
            work(0);
            for (unsigned w = started; w < nt; ++w) work(w);
            for (auto& t : th) t.join();
        }
        bool all = threads_ok && !lines.empty();
        for (char c : okv) all = all && c;
        if (all) {
            spans.swap(lines);
            return true;
        }
    }
    if (j.pos < len && text[j.pos] == '[') {  // JSON array of objects
        ++j.pos;
        j.ws();
        if (j.pos < len && text[j.pos] == ']') {
            ++j.pos;
        } else {
            for (;;) {
                j.ws();
                const size_t b = j.pos;
                if (!j.skip_value(1)) { err = "invalid JSON: " + j.err; return false; }
                spans.emplace_back(b, j.pos);
                j.ws();
                if (j.pos < len && text[j.pos] == ',') { ++j.pos; continue; }
                if (j.pos < len && text[j.pos] == ']') { ++j.pos; break; }
                err = "invalid JSON: expected ',' or ']' at byte " + std::to_string(j.pos);
                return false;
            }
        }
        j.ws();
        if (j.pos != len) { err = "invalid JSON: trailing characters at byte " + std::to_string(j.pos); return false; }
        return true;
    }
    // NDJSON: a sequence of values separated by whitespace / newlines
    while (j.pos < len) {
        const size_t b = j.pos;
        if (!j.skip_value(0)) { err = "invalid JSON: " + j.err; return false; }
        spans.emplace_back(b, j.pos);
        j.ws();
    }
    return true;
}

bool populate_inputs(const InputList& inputs, const Graph& g, uint8_t* buf, size_t n_inputs, std::string& err) {
    memset(buf, 0, n_inputs * 32);
    if (n_inputs) buf[0] = 1;  // get_inputs_buffer, lib.rs:177-181
    for (const auto& kv : inputs) {
        auto it = g.input_index.find(kv.first);
        if (it == g.input_index.end()) {  // reference: HashMap index panic (lib.rs:158)
            err = "unknown input signal " + kv.first;
            return false;
        }
        const InputSignal& s = g.inputs[it->second];
        if (s.len != kv.second.size()) {  // lib.rs:159-161
            err = "Invalid input length for " + kv.first;
            return false;
        }
        if ((size_t)s.offset + s.len > n_inputs) {
            err = "input " + kv.first + " out of range of the inputs buffer";
            return false;
        }
        for (size_t i = 0; i < kv.second.size(); ++i) memcpy(buf + 32 * ((size_t)s.offset + i), kv.second[i].v, 32);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// .wtns (lib.rs:114-123; iden3 binfile layout of wtns-file 0.1.5 [ext])
// ---------------------------------------------------------------------------------------------
size_t wtns_size(size_t n) { return 76 + 32 * n; }

void wtns_write_header(uint8_t* p, size_t n) {
    uint32_t u;
    uint64_t q;
    memcpy(p, "wtns", 4); p += 4;
    u = 2; memcpy(p, &u, 4); p += 4;   // version forced to 2 (lib.rs:118)
    u = 2; memcpy(p, &u, 4); p += 4;   // sections
    u = 1; memcpy(p, &u, 4); p += 4;
    q = 40; memcpy(p, &q, 8); p += 8;
    u = 32; memcpy(p, &u, 4); p += 4;  // n8
    Fr pr = fr_p();
    memcpy(p, pr.v, 32); p += 32;      // prime = M (lib.rs:117)
    u = (uint32_t)n; memcpy(p, &u, 4); p += 4;
    u = 2; memcpy(p, &u, 4); p += 4;
    q = 32ull * n; memcpy(p, &q, 8);
}

void wtns_from_witness(const uint8_t* w, size_t n, uint8_t* out) {
    wtns_write_header(out, n);
    memcpy(out + 76, w, 32 * n);
}

}  // namespace cwc
