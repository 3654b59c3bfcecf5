// Host-only sanitizer harness for the graph loader and compiler (no HIP): built with -fsanitize=address,undefined by
// tests/test_host_formats.py.  For every .bin given: parse, re-serialize (byte-exact), compile for a set of tile
// widths / divider modes, round-trip the program blob.  Exit code 0 = clean.
#include <stdio.h>
#include <stdlib.h>

#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../circom-witnesscalc_amd/csrc/program.hpp"

using namespace cwc;

int main(int argc, char** argv) {
    int rc = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        Graph g;
        std::string err;
        if (!deserialize_witnesscalc_graph(data.data(), data.size(), g, err)) {
            printf("%s: rejected: %s\n", argv[a], err.c_str());  // malformed inputs must be rejected cleanly
            continue;
        }
        if (serialize_witnesscalc_graph(g) != data) {
            printf("%s: re-serialization differs\n", argv[a]);  // (only canonical writers round-trip byte-exactly)
        }
        const uint32_t widths[] = {1, 2, 8, 64};
        const uint32_t dividers[] = {0, 1, 4};
        for (uint32_t T : widths)
            for (uint32_t W : dividers) {
                Program p, q;
                if (!compile_program(g, T, W, p, err)) {
                    printf("%s: T=%u W=%u: %s\n", argv[a], T, W, err.c_str());
                    continue;
                }
                std::vector<uint8_t> blob = program_to_blob(p);
                if (!program_from_blob(blob.data(), blob.size(), q, err) || q.hdr != p.hdr || q.recs != p.recs ||
                    q.crefs != p.crefs || q.consts != p.consts || q.witness_refs != p.witness_refs || q.div_lanes != p.div_lanes) {
                    printf("%s: T=%u W=%u: blob round trip failed: %s\n", argv[a], T, W, err.c_str());
                    rc = 1;
                }
                // truncated blobs must be rejected, not read out of bounds
                for (size_t cut : {(size_t)0, (size_t)7, blob.size() / 2, blob.size() - 1})
                    if (program_from_blob(blob.data(), cut, q, err)) {
                        printf("%s: truncated blob accepted\n", argv[a]);
                        rc = 1;
                    }
            }
    }
    printf("done rc=%d\n", rc);
    return rc;
}
