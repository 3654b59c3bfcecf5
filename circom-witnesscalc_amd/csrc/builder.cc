// C-ABI graph builder: the producer side of the `.bin` container.
//
// The reference builds a Vec<graph::Node> (src/graph.rs:236-245), a witness list and an InputSignalsInfo map in
// build-circuit and hands them to serialize_witnesscalc_graph (src/storage.rs:137-183; node encoding :50-91).  This
// file is that producer API for C / FFI callers: nodes are pushed in topological order (every operand is an earlier
// node, what graph.rs:343-356 asserts), constants are canonical field elements, and gwb_builder_finish writes the
// container with the product's own writer (graph.cc) -- the bytes gw_calc_witness and the reference's reader load.
#include <stdlib.h>
#include <string.h>

#include <string>

#define GW_NO_INLINE_FREE_STATUS
#include "../../include/graph_witness_batch.h"
#include "graph.hpp"

using namespace cwc;

struct gwb_builder {
    Graph g;
    std::string err;  // first error (sticky: later pushes return GWB_BUILDER_BAD, finish reports it)
};

namespace {
void set_status(gw_status_t* st, GW_ERROR_CODE code, const std::string& msg) {
    if (!st) return;
    st->code = code;
    st->error_msg = nullptr;
    if (code == OK && msg.empty()) return;
    st->error_msg = (char*)malloc(msg.size() + 1);
    if (st->error_msg) memcpy(st->error_msg, msg.c_str(), msg.size() + 1);
}
uint32_t push(gwb_builder* b, const Node& n, int arity) {
    if (!b) return GWB_BUILDER_BAD;
    if (!b->err.empty()) return GWB_BUILDER_BAD;
    const size_t idx = b->g.nodes.size();
    if (idx >= 0xfffffffeull) {
        b->err = "more than 2^32 - 2 nodes";
        return GWB_BUILDER_BAD;
    }
    const uint32_t ops[3] = {n.a, n.b, n.c};
    for (int q = 0; q < arity; ++q)
        if (ops[q] >= idx) {  // graph.rs:343-356: operands come before their users
            b->err = "node " + std::to_string(idx) + " references node " + std::to_string(ops[q]) + " that is not before it";
            return GWB_BUILDER_BAD;
        }
    b->g.nodes.push_back(n);
    if (arity) b->g.n_op++;
    return (uint32_t)idx;
}
}  // namespace

extern "C" {

gwb_builder_t* gwb_builder_new(void) {
    try {
        return new gwb_builder();
    } catch (...) {
        return nullptr;
    }
}

void gwb_builder_free(gwb_builder_t* b) { delete b; }

uint32_t gwb_builder_input(gwb_builder_t* b, uint32_t input_index) {
    try {
        return push(b, Node{N_INPUT, 0, input_index, 0, 0}, 0);
    } catch (...) {
        return GWB_BUILDER_BAD;
    }
}

uint32_t gwb_builder_constant(gwb_builder_t* b, const void* value_le, size_t len) {
    try {
        if (!b || !b->err.empty()) return GWB_BUILDER_BAD;
        if (!value_le && len) {
            b->err = "constant: null value";
            return GWB_BUILDER_BAD;
        }
        // storage.rs:28: a constant is a field element; longer / larger byte strings are reduced as the reader would
        const Fr v = u256_from_le_bytes_mod_order((const uint8_t*)value_le, len);
        b->g.const_values.push_back(v);
        return push(b, Node{N_CONST, 0, (uint32_t)(b->g.const_values.size() - 1), 0, 0}, 0);
    } catch (...) {
        return GWB_BUILDER_BAD;
    }
}

uint32_t gwb_builder_uno(gwb_builder_t* b, uint32_t op, uint32_t a) {
    try {
        if (b && b->err.empty() && op > UOP_ID) b->err = "unknown UnoOp code " + std::to_string(op);
        return push(b, Node{N_UNO, (uint8_t)op, a, 0, 0}, 1);
    } catch (...) {
        return GWB_BUILDER_BAD;
    }
}

uint32_t gwb_builder_duo(gwb_builder_t* b, uint32_t op, uint32_t a, uint32_t bb) {
    try {
        if (b && b->err.empty() && op >= OP_DUO_COUNT) b->err = "unknown DuoOp code " + std::to_string(op);
        return push(b, Node{N_DUO, (uint8_t)op, a, bb, 0}, 2);
    } catch (...) {
        return GWB_BUILDER_BAD;
    }
}

uint32_t gwb_builder_tres(gwb_builder_t* b, uint32_t op, uint32_t a, uint32_t bb, uint32_t c) {
    try {
        if (b && b->err.empty() && op > TOP_TERNCOND) b->err = "unknown TresOp code " + std::to_string(op);
        return push(b, Node{N_TRES, (uint8_t)op, a, bb, c}, 3);
    } catch (...) {
        return GWB_BUILDER_BAD;
    }
}

int gwb_builder_witness(gwb_builder_t* b, uint32_t node) {
    try {
        if (!b || !b->err.empty()) return 1;
        if (node >= b->g.nodes.size()) {
            b->err = "witness signal references node " + std::to_string(node) + " beyond the graph";
            return 1;
        }
        b->g.witness_signals.push_back(node);
        return 0;
    } catch (...) {
        return 1;
    }
}

int gwb_builder_input_signal(gwb_builder_t* b, const char* name, uint32_t offset, uint32_t len) {
    try {
        if (!b || !b->err.empty()) return 1;
        if (!name) {
            b->err = "input signal: null name";
            return 1;
        }
        const std::string key(name);
        auto it = b->g.input_index.find(key);
        if (it != b->g.input_index.end()) {  // a map: the last entry of a name wins
            b->g.inputs[it->second].offset = offset;
            b->g.inputs[it->second].len = len;
            return 0;
        }
        b->g.input_index[key] = (uint32_t)b->g.inputs.size();
        b->g.inputs.push_back(InputSignal{key, offset, len});
        return 0;
    } catch (...) {
        return 1;
    }
}

uint64_t gwb_builder_node_count(const gwb_builder_t* b) { return b ? b->g.nodes.size() : 0; }

int gwb_builder_finish(const gwb_builder_t* b, void** out, size_t* out_len, gw_status_t* status) {
    try {
        if (!b || !out || !out_len) {
            set_status(status, ERROR, "null argument");
            return 1;
        }
        if (!b->err.empty()) {
            set_status(status, ERROR, "Failed to build graph: " + b->err);
            return 1;
        }
        const std::vector<uint8_t> bytes = serialize_witnesscalc_graph(b->g);
        *out = malloc(bytes.size() ? bytes.size() : 1);
        if (!*out) {
            set_status(status, ERROR, "out of memory");
            return 1;
        }
        memcpy(*out, bytes.data(), bytes.size());
        *out_len = bytes.size();
        set_status(status, OK, "");
        return 0;
    } catch (...) {
        set_status(status, ERROR, "out of memory");
        return 1;
    }
}

}  // extern "C"
