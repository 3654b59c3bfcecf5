// One wave: the parallel forms of the scan recurrences (csrc/scan_gfx950.hpp) against the serial recurrences computed on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../circom-witnesscalc_amd/csrc scan_par_test.hip -o scan_par_test && ./scan_par_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "scan_gfx950.hpp"
using namespace cwc;

typedef unsigned __int128 u128;
struct Case { uint32_t st[64]; uint32_t xp[64][6]; uint64_t d[64], x[64], a0[64]; uint32_t iters; };
struct Out { uint32_t limb[64][2], carry[64][6]; uint64_t quo[64], rem[64]; };

template <int T>
__global__ void run(const Case* c, Out* o, int n) {
    const uint32_t lane = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        const bool st = c[k].st[lane] != 0;
        const uint32_t xw[6] = {c[k].xp[lane][0], c[k].xp[lane][1], c[k].xp[lane][2], c[k].xp[lane][3], c[k].xp[lane][4], c[k].xp[lane][5]};
        uint32_t limb[2], carry[6];
        scan_carry_parallel<T>(st, lane, xw, limb, carry);
        for (int i = 0; i < 2; ++i) o[k].limb[lane][i] = limb[i];
        for (int i = 0; i < 6; ++i) o[k].carry[lane][i] = carry[i];
        uint64_t q, r;
        scan_div_parallel<T>(st, lane, c[k].iters, c[k].d[lane], c[k].x[lane], c[k].a0[lane], q, r);
        o[k].quo[lane] = q;
        o[k].rem[lane] = r;
    }
}

struct ConvCase { uint64_t x[64], y[64]; uint32_t k; };
struct ConvOut { uint32_t col[64][5]; };
template <int T>
__global__ void run_conv(const ConvCase* c, ConvOut* o, int n) {
    const uint32_t lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        uint32_t col[5];
        conv_limb_columns<T>(c[i].k, lane, c[i].x[lane], lane < c[i].k * T ? c[i].y[lane] : 0ull, col);
        for (int w = 0; w < 5; ++w) o[i].col[lane][w] = col[w];
    }
}

static uint64_t rnd64() { return ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand(); }
static uint64_t pick64() {
    switch (rand() % 8) {
        case 0: return ~0ull;
        case 1: return 0;
        case 2: return ~0ull ^ (uint64_t)(rand() % 4);
        default: return rnd64();
    }
}

template <int T>
static int test(int n) {
    const int D = 2 * T, P = 64 / D;  // lanes per pair, pairs per wave
    std::vector<Case> cs(n);
    for (int k = 0; k < n; ++k) {
        Case& c = cs[k];
        const int mode = rand() % 4;
        for (int t = 0; t < T; ++t) {
            uint64_t d = 1;
            int run = 0, longest = 0;
            for (int p = 0; p < P; ++p) {
                const bool st = p == 0 || rand() % 10 == 0;
                run = st ? 1 : run + 1;
                longest = run > longest ? run : longest;
                if (st) {
                    switch (rand() % 7) { case 0: d = 1; break; case 1: d = 2; break; case 2: d = ~0ull; break; case 3: d = 1ull << 63; break; case 4: d = (1ull << 63) + 1; break;
                                          case 5: d = (uint64_t)(rand() % 1000) + 1; break; default: d = rnd64() | 1; }
                }
                uint64_t w[3] = {mode == 0 ? ~0ull : pick64(), mode == 3 ? 0 : pick64(), mode >= 2 ? 0 : pick64() >> (rand() % 64)};
                const uint64_t a0 = st ? rnd64() % d : 0, x = mode == 0 ? ~0ull : pick64();
                for (int r = 0; r < 2; ++r) {  // OUT lane, ACC lane
                    const int lane = p * D + r * T + t;
                    c.st[lane] = st;
                    for (int i = 0; i < 3; ++i) { c.xp[lane][2 * i] = (uint32_t)w[i]; c.xp[lane][2 * i + 1] = (uint32_t)(w[i] >> 32); }
                    c.d[lane] = d; c.x[lane] = x; c.a0[lane] = a0;
                }
            }
            c.iters = t == 0 || (uint32_t)longest > c.iters ? longest : c.iters;
        }
    }
    Case* dc; Out* dout;
    (void)hipMalloc(&dc, n * sizeof(Case)); (void)hipMalloc(&dout, n * sizeof(Out));
    (void)hipMemcpy(dc, cs.data(), n * sizeof(Case), hipMemcpyHostToDevice);
    run<T><<<1, 64>>>(dc, dout, n);
    std::vector<Out> os(n);
    (void)hipMemcpy(os.data(), dout, n * sizeof(Out), hipMemcpyDeviceToHost);
    int bad_c = 0, bad_d = 0;
    for (int k = 0; k < n; ++k) {
        const Case& c = cs[k];
        for (int t = 0; t < T; ++t) {
            // serial recurrences over the pairs of set t
            uint64_t cw[4] = {0, 0, 0, 0};  // carry (up to 193 bits here)
            uint64_t rem = 0;
            for (int p = 0; p < P; ++p) {
                const int lane = p * D + t;
                if (c.st[lane]) cw[0] = cw[1] = cw[2] = cw[3] = 0;  // (the incoming accumulator is part of xp at a segment's start)
                uint64_t tw[4];
                u128 acc = 0;
                for (int i = 0; i < 4; ++i) {
                    const uint64_t xi = i < 3 ? (((uint64_t)c.xp[lane][2 * i + 1] << 32) | c.xp[lane][2 * i]) : 0;
                    acc += (u128)xi + cw[i];
                    tw[i] = (uint64_t)acc;
                    acc >>= 64;
                }
                const uint64_t limb = tw[0];
                cw[0] = tw[1]; cw[1] = tw[2]; cw[2] = tw[3]; cw[3] = 0;
                const u128 tt = ((u128)(c.st[lane] ? c.a0[lane] : rem) << 64) | c.x[lane];
                const uint64_t q = (uint64_t)(tt / c.d[lane]);
                rem = (uint64_t)(tt % c.d[lane]);
                for (int r = 0; r < 2; ++r) {
                    const int l2 = lane + r * T;
                    const Out& o = os[k];
                    const uint64_t gl = ((uint64_t)o.limb[l2][1] << 32) | o.limb[l2][0];
                    bool ok = gl == limb;
                    for (int i = 0; i < 3; ++i) ok = ok && ((((uint64_t)o.carry[l2][2 * i + 1] << 32) | o.carry[l2][2 * i]) == cw[i]);
                    if (!ok && bad_c++ < 5) printf("T=%d case %d carry: pair %d lane %d limb got %llx want %llx carry0 got %x%08x want %llx\n", T, k, p, l2, (unsigned long long)gl,
                                                   (unsigned long long)limb, o.carry[l2][1], o.carry[l2][0], (unsigned long long)cw[0]);
                    if ((o.quo[l2] != q || o.rem[l2] != rem) && bad_d++ < 5)
                        printf("T=%d case %d div: pair %d lane %d q got %llx want %llx rem got %llx want %llx (d %llx)\n", T, k, p, l2, (unsigned long long)o.quo[l2], (unsigned long long)q,
                               (unsigned long long)o.rem[l2], (unsigned long long)rem, (unsigned long long)c.d[lane]);
                }
            }
        }
    }
    printf("T=%d: %d cases, %d carry mismatches, %d division mismatches\n", T, n, bad_c, bad_d);
    (void)hipFree(dc); (void)hipFree(dout);
    return bad_c + bad_d;
}

// the columns of k x k limb products (conv_limb_columns) against 128-bit host arithmetic
template <int T>
static int test_conv(int n) {
    const int maxk = T == 1 ? 32 : 16;
    std::vector<ConvCase> cs(n);
    for (int i = 0; i < n; ++i) {
        cs[i].k = 2 + rand() % (maxk - 1);
        const int mode = rand() % 3;
        for (int l = 0; l < 64; ++l) {
            cs[i].x[l] = mode == 0 ? ~0ull : pick64();
            cs[i].y[l] = mode == 0 ? ~0ull : pick64();  // (lanes of the columns k and above hold anything: the kernel side masks y)
        }
    }
    ConvCase* dc; ConvOut* dout;
    (void)hipMalloc(&dc, n * sizeof(ConvCase)); (void)hipMalloc(&dout, n * sizeof(ConvOut));
    (void)hipMemcpy(dc, cs.data(), n * sizeof(ConvCase), hipMemcpyHostToDevice);
    run_conv<T><<<1, 64>>>(dc, dout, n);
    std::vector<ConvOut> os(n);
    (void)hipMemcpy(os.data(), dout, n * sizeof(ConvOut), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const int k = (int)cs[i].k;
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < 2 * k - 1; ++c) {
                uint64_t w[3] = {0, 0, 0};  // 192-bit sum
                for (int a = 0; a < k; ++a) {
                    const int b = c - a;
                    if (b < 0 || b >= k) continue;
                    const u128 p = (u128)cs[i].x[a * T + t] * cs[i].y[b * T + t];
                    u128 s = (u128)w[0] + (uint64_t)p;
                    w[0] = (uint64_t)s;
                    s = (u128)w[1] + (uint64_t)(p >> 64) + (uint64_t)(s >> 64);
                    w[1] = (uint64_t)s;
                    w[2] += (uint64_t)(s >> 64);
                }
                const uint32_t* g = os[i].col[c * T + t];
                const bool ok = (((uint64_t)g[1] << 32) | g[0]) == w[0] && (((uint64_t)g[3] << 32) | g[2]) == w[1] && g[4] == (uint32_t)w[2];
                if (!ok && bad++ < 5) printf("T=%d conv case %d k %d column %d: got %x %08x%08x %08x%08x want %llx %016llx %016llx\n", T, i, k, c, g[4], g[3], g[2], g[1], g[0],
                                             (unsigned long long)w[2], (unsigned long long)w[1], (unsigned long long)w[0]);
            }
    }
    printf("T=%d: %d limb products, %d column mismatches\n", T, n, bad);
    (void)hipFree(dc); (void)hipFree(dout);
    return bad;
}

int main() {
    srand(7);
    int bad = test<1>(4000);
    bad += test<2>(4000);
    bad += test_conv<1>(1500);
    bad += test_conv<2>(1500);
    printf(bad ? "FAILED\n" : "scan_par_test: ok\n");
    return bad != 0;
}
