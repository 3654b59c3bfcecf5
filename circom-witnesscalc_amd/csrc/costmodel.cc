// The cost model's table: lone-wave shader cycles per bundle class as measured on MI355X (built in, CWC_MODEL_CYCLES, or the
// calibration file of tools/gpu_calibrate.py), and a compiled program's price under it.
#include "compile_internal.hpp"

namespace cwc {

const CycleTable kCycles;
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
uint32_t div_cost50() { return (uint32_t)(kCycles[C_DIV] / 50.0); }
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

}  // namespace cwc
