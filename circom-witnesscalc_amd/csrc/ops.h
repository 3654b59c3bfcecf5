// Operation wire codes shared by host and device code.
#pragma once
#include <stdint.h>

namespace cwc {

// wire codes, reference protos/messages.proto:5-35
enum DuoOp : uint8_t {
    OP_MUL = 0, OP_DIV = 1, OP_ADD = 2, OP_SUB = 3, OP_POW = 4, OP_IDIV = 5, OP_MOD = 6, OP_EQ = 7, OP_NEQ = 8,
    OP_LT = 9, OP_GT = 10, OP_LEQ = 11, OP_GEQ = 12, OP_LAND = 13, OP_LOR = 14, OP_SHL = 15, OP_SHR = 16,
    OP_BOR = 17, OP_BAND = 18, OP_BXOR = 19, OP_DUO_COUNT = 20,
    OP_BITX = 20  // (a >> k) & 1 with a constant k: never in a file, made by the compiler from Band(Shr(a, k), 1)
};
enum UnoOp : uint8_t { UOP_NEG = 0, UOP_ID = 1 };
enum TresOp : uint8_t { TOP_TERNCOND = 0 };

}  // namespace cwc
