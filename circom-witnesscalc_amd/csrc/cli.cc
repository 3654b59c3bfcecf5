// calc-witness CLI twin (reference src/bin/calc-witness.rs:13-49):
//   calc-witness <graph.bin> <inputs.json> <witness.wtns>
// Same positional arguments, usage text, exit codes and progress lines; evaluation goes through
// gw_calc_witness (HIP).  The timing window opens after file reads and closes before the file write,
// like the reference's Instant window (:35-41).
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/graph_witness.h"

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "Usage: %s <graph.bin> <inputs.json> <witness.wtns>\n", argv[0]);
        return 1;
    }
    std::ifstream fi(argv[2], std::ios::binary);
    if (!fi) {
        fprintf(stderr, "Failed to read input file\n");
        return 101;
    }
    std::string inputs((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
    std::ifstream fg(argv[1], std::ios::binary);
    if (!fg) {
        fprintf(stderr, "Failed to read graph file\n");
        return 101;
    }
    std::vector<char> graph((std::istreambuf_iterator<char>(fg)), std::istreambuf_iterator<char>());

    auto t0 = std::chrono::steady_clock::now();
    void* wtns = nullptr;
    size_t wtns_len = 0;
    gw_status_t st;
    int rc = gw_calc_witness(inputs.c_str(), graph.data(), graph.size(), &wtns, &wtns_len, &st);
    auto t1 = std::chrono::steady_clock::now();
    if (rc != 0) {
        fprintf(stderr, "%s\n", st.error_msg ? st.error_msg : "calc_witness failed");
        gw_free_status(&st);
        return 101;  // the reference panics (unwrap) here
    }
    gw_free_status(&st);
    printf("Witness generated in: %.6fms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(wtns, 1, wtns_len, f) != wtns_len) {
        fprintf(stderr, "Failed to write %s\n", argv[3]);
        return 101;
    }
    fclose(f);
    free(wtns);
    printf("witness saved to %s\n", argv[3]);
    return 0;
}
