// Validation only: the "fc in parts" flow of the dual path (lstm.hip): the second band layer (here also as a kernel of its own, which the
// library does not ship: band_lstm_h2_kernel<128, ., PART>) writes the two directions'
// shares of the block's fc, time_lstm_h2w_kernel<FUSE, ., PART> adds them and the residual while staging.
//   (1) band layer 1 with PART against the same layer without it + fc(h) evaluated in double; run-to-run bit stability
//   (2) time kernel with PART on (z, part) against the kernel without PART on the pre-added rows (z + pf) + pb: bit-identical;
//       run-to-run bit stability
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o build/band_parts_check tools/band_parts_check.hip
#include "../speechseparation_amd/csrc/lstm.hip"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace bsrnn;
namespace bsrnn { bool force_f32() { return false; } int gemm_mode() { return GEMM_FP16X2; } }
static unsigned g_seed = 777u;
static int lcg() { g_seed = g_seed * 1664525u + 1013904223u; return (int)((g_seed >> 8) & 0x7fffff); }
static float urand() { return lcg() / (float)0x7fffff - 0.5f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double h2d(uint16_t b) { _Float16 v; memcpy(&v, &b, 2); return (double)(float)v; }
static uint16_t small_half() { return (uint16_t)(0x2c00 + (lcg() & 0x3ff) + ((lcg() & 1) << 15)); }   // +-[0.06, 0.12)

static int band(int N, int L, int runs)
{
    const size_t nrow = (size_t)N * L;
    const size_t nw = (size_t)2 * 4 * 6 * 4 * 2 * 64 * 8, nfc = (size_t)4 * 4 * 2 * 64 * 8;
    uint16_t *x, *w, *wfc; float *h, *p, *b, *bfc;
    CK(hipMalloc(&x, nrow * 256 * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&wfc, nfc * 2));
    CK(hipMalloc(&h, nrow * 128 * 4)); CK(hipMalloc(&p, nrow * 128 * 4)); CK(hipMalloc(&b, 512 * 4)); CK(hipMalloc(&bfc, 64 * 4));
    std::vector<uint16_t> hx(nrow * 256), hw(nw), hwfc(nfc);
    for (size_t i = 0; i < nrow; ++i)
        for (int k = 0; k < 128; ++k) {
            const float v = urand() * 1.6f;                       // an h of layer 0: |v| < 1
            _Float16 a = (_Float16)v, c2 = (_Float16)((v - (float)a) * 2048.f);
            memcpy(&hx[i * 256 + k], &a, 2); memcpy(&hx[i * 256 + 128 + k], &c2, 2);
        }
    for (auto& v : hw) v = small_half();
    for (auto& v : hwfc) v = small_half();
    std::vector<float> hb(512), hbfc(64);
    for (auto& v : hb) v = urand() * 0.5f;
    for (auto& v : hbfc) v = urand() * 0.5f;
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfc, hwfc.data(), nfc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bfc, hbfc.data(), 64 * 4, hipMemcpyHostToDevice));
    const dim3 grid((N + 15) / 16, 2), block(256);
    printf("== band layer 1, N=%d sequences x L=%d positions (%d workgroups)\n", N, L, grid.x * 2);
    hipLaunchKernelGGL((band_lstm_h2_kernel<128, false, false>), grid, block, 0, 0, (const float*)x, h, (const uint4*)w, b, N, L, (int*)nullptr, (unsigned long long*)nullptr, (const uint4*)nullptr, (const float*)nullptr);
    CK(hipDeviceSynchronize());
    std::vector<float> hh(nrow * 128), first(nrow * 128), cur(nrow * 128);
    CK(hipMemcpy(hh.data(), h, hh.size() * 4, hipMemcpyDeviceToHost));
    size_t unstable = 0;
    for (int r = 1; r < runs; ++r) {             // the plain launch, run to run
        hipLaunchKernelGGL((band_lstm_h2_kernel<128, false, false>), grid, block, 0, 0, (const float*)x, h, (const uint4*)w, b, N, L, (int*)nullptr, (unsigned long long*)nullptr, (const uint4*)nullptr, (const float*)nullptr);
        CK(hipMemcpy(cur.data(), h, cur.size() * 4, hipMemcpyDeviceToHost));
        size_t d = 0; for (size_t i = 0; i < cur.size(); ++i) d += memcmp(&cur[i], &hh[i], 4) != 0;
        if (d) printf("   plain launch, run %d: %zu words differ from run 0\n", r, d);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < runs; ++r) {
        CK(hipMemset(p, 0xff, nrow * 128 * 4));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((band_lstm_h2_kernel<128, false, true>), grid, block, 0, 0, (const float*)x, p, (const uint4*)w, b, N, L, (int*)nullptr, (unsigned long long*)nullptr, (const uint4*)wfc, bfc);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms * 1e3f);
        CK(hipMemcpy((r ? cur : first).data(), p, cur.size() * 4, hipMemcpyDeviceToHost));
        if (r) {
            size_t d = 0, rows = 0, last = (size_t)-1; int dirs = 0;
            for (size_t i = 0; i < cur.size(); ++i)
                if (memcmp(&cur[i], &first[i], 4) != 0) { ++d; if (i / 128 != last) { ++rows; last = i / 128; } dirs |= 1 << ((i % 128) / 64); }
            unstable += d;
            if (d) {
                printf("   run %d: affected sequences (dir):", r);
                size_t lastseq = (size_t)-1; int shown = 0;
                for (size_t i = 0; i < cur.size() && shown < 24; ++i)
                    if (memcmp(&cur[i], &first[i], 4) != 0) { const size_t sq = (i / 128 / L) * 2 + (i % 128) / 64; if (sq != lastseq) { printf(" %zu(%zu)", sq / 2, sq % 2); lastseq = sq; ++shown; } }
                printf("\n");
                for (size_t i = 0; i < cur.size(); ++i)
                    if (memcmp(&cur[i], &first[i], 4) != 0) {
                        const size_t row = i / 128; const int d2 = (i % 128) / 64;
                        printf("      row %zu (sequence %zu position %zu) dir %d: run 0 / run %d / double:", row, row / L, row % L, d2, r);
                        for (int o = 0; o < 6; ++o) {
                            double acc = d2 == 0 ? hbfc[o] : 0.0;
                            for (int k = 0; k < 64; ++k) {
                                const int tl = o / 16, bk = 2 * d2 + k / 32, ln = (o % 16) + 16 * ((k % 32) / 8), j = k % 8;
                                const size_t base = ((size_t)tl * 4 + bk) * 2 * 64 * 8;
                                acc += (h2d(hwfc[base + (size_t)ln * 8 + j]) + h2d(hwfc[base + 64 * 8 + (size_t)ln * 8 + j]) / 2048.0) * (double)hh[row * 128 + d2 * 64 + k];
                            }
                            printf("  %.5f / %.5f / %.5f", first[row * 128 + d2 * 64 + o], cur[row * 128 + d2 * 64 + o], acc);
                        }
                        printf("\n");
                        break;
                    }
            }
            if (d) printf("   run %d: %zu words differ from run 0 in %zu (sequence, position) rows, directions mask %d; first row: sequence %zu position %zu\n", r, d, rows, dirs,
                          [&] { for (size_t i = 0; i < cur.size(); ++i) if (memcmp(&cur[i], &first[i], 4)) return i / 128 / L; return (size_t)0; }(),
                          [&] { for (size_t i = 0; i < cur.size(); ++i) if (memcmp(&cur[i], &first[i], 4)) return (i / 128) % L; return (size_t)0; }());
        }
    }
    CK(hipGetLastError());
    // against fc of the plain launch's h, in double
    double worst = 0; size_t bad = 0, checked = 0, untouched = 0;
    for (size_t i = 0; i < first.size(); ++i) { uint32_t u; memcpy(&u, &first[i], 4); untouched += u == 0xffffffffu; }
    for (size_t row = 0; row < nrow; row += (nrow > 4096 ? 97 : 1))
        for (int d = 0; d < 2; ++d)
            for (int o = 0; o < 64; ++o) {
                double acc = d == 0 ? hbfc[o] : 0.0;
                for (int k = 0; k < 64; ++k) {
                    const int tl = o / 16, bk = 2 * d + k / 32, ln = (o % 16) + 16 * ((k % 32) / 8), j = k % 8;
                    const size_t base = ((size_t)tl * 4 + bk) * 2 * 64 * 8;
                    const double wk = h2d(hwfc[base + (size_t)ln * 8 + j]) + h2d(hwfc[base + 64 * 8 + (size_t)ln * 8 + j]) / 2048.0;
                    acc += wk * (double)hh[row * 128 + d * 64 + k];
                }
                const double e = fabs(acc - (double)first[row * 128 + d * 64 + o]);
                worst = std::max(worst, e); bad += !(e < 2e-6); ++checked;
            }
    printf("   PART launch %.1f us; fc shares vs double: max %.3e over %zu values (%zu above 2e-6); never written %zu; unstable words over %d runs: %zu\n",
           best, worst, checked, bad, untouched, runs, unstable);
    hipFree(x); hipFree(w); hipFree(wfc); hipFree(h); hipFree(p); hipFree(b); hipFree(bfc);
    return (bad != 0) + (untouched != 0) + (unstable != 0);
}

// (3) both layers in one launch (band_pair_h2_kernel, partner workgroups hand the layer-0 planes over through L2) against layer 0 and
//     layer 1 (with the shares) as two launches: bit for bit, every run
static int pair(int N, int L, int runs)
{
    const size_t nrow = (size_t)N * L;
    const size_t nw0 = (size_t)2 * 4 * 4 * 4 * 2 * 64 * 8, nw1 = (size_t)2 * 4 * 6 * 4 * 2 * 64 * 8, nfc = (size_t)4 * 4 * 2 * 64 * 8;
    float *z, *hb0, *hb0b, *p_ref, *p_pair, *b0, *b1, *bfc; uint16_t *w0, *w1, *wfc; int* flags;
    CK(hipMalloc(&z, nrow * 64 * 4)); CK(hipMalloc(&hb0, nrow * 128 * 4)); CK(hipMalloc(&hb0b, nrow * 128 * 4));
    CK(hipMalloc(&p_ref, nrow * 128 * 4)); CK(hipMalloc(&p_pair, nrow * 128 * 4));
    CK(hipMalloc(&w0, nw0 * 2)); CK(hipMalloc(&w1, nw1 * 2)); CK(hipMalloc(&wfc, nfc * 2));
    CK(hipMalloc(&b0, 512 * 4)); CK(hipMalloc(&b1, 512 * 4)); CK(hipMalloc(&bfc, 64 * 4));
    const size_t nfl = 2 * ((size_t)(N + 15) / 16) + 64;
    CK(hipMalloc(&flags, nfl * 4)); CK(hipMemset(flags, 0, nfl * 4));
    std::vector<float> hz(nrow * 64), hb(512), hbfc(64);
    for (auto& v : hz) v = urand() * 2.f;
    std::vector<uint16_t> hw0(nw0), hw1(nw1), hwfc(nfc);
    for (auto& v : hw0) v = small_half();
    for (auto& v : hw1) v = small_half();
    for (auto& v : hwfc) v = small_half();
    CK(hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w0, hw0.data(), nw0 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw1.data(), nw1 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfc, hwfc.data(), nfc * 2, hipMemcpyHostToDevice));
    for (auto& v : hb) v = urand() * 0.5f;
    CK(hipMemcpy(b0, hb.data(), 512 * 4, hipMemcpyHostToDevice));
    for (auto& v : hb) v = urand() * 0.5f;
    CK(hipMemcpy(b1, hb.data(), 512 * 4, hipMemcpyHostToDevice));
    for (auto& v : hbfc) v = urand() * 0.5f;
    CK(hipMemcpy(bfc, hbfc.data(), 64 * 4, hipMemcpyHostToDevice));
    const dim3 g2((N + 15) / 16, 2), g1((((N + 15) / 16 + 7) / 8) * 16), block(256);
    printf("== band block as one launch, N=%d sequences x L=%d positions (%d workgroups)\n", N, L, g1.x);
    hipLaunchKernelGGL((band_lstm_h2_kernel<64, false, false>), g2, block, 0, 0, z, hb0, (const uint4*)w0, b0, N, L, (int*)nullptr, (unsigned long long*)nullptr, (const uint4*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL((band_lstm_h2_kernel<128, false, true>), g2, block, 0, 0, (const float*)hb0, p_ref, (const uint4*)w1, b1, N, L, (int*)nullptr, (unsigned long long*)nullptr, (const uint4*)wfc, bfc);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> ref(nrow * 128), cur(nrow * 128);
    CK(hipMemcpy(ref.data(), p_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    int* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f; size_t total = 0;
    for (int r = 0; r < runs; ++r) {
        CK(hipMemset(p_pair, 0xff, nrow * 128 * 4)); CK(hipMemset(hb0b, 0xff, nrow * 128 * 4));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((band_pair_h2_kernel<true>), g1, block, 0, 0, (const float*)z, hb0b, p_pair, (const uint4*)w0, b0, (const uint4*)w1, b1, N, L, flag,
                           (const uint4*)wfc, bfc, flags, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms * 1e3f);
        CK(hipMemcpy(cur.data(), p_pair, cur.size() * 4, hipMemcpyDeviceToHost));
        size_t d = 0; for (size_t i = 0; i < cur.size(); ++i) d += cur[i] != ref[i];
        total += d;
        if (d) printf("   run %d: %zu words differ from the two launches\n", r, d);
    }
    int hf = 0; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("   pair launch %.1f us; words differing from layer 0 + layer 1 as two launches over %d runs: %zu; flag %d\n", best, runs, total, hf);
    hipFree(z); hipFree(hb0); hipFree(hb0b); hipFree(p_ref); hipFree(p_pair); hipFree(w0); hipFree(w1); hipFree(wfc); hipFree(b0); hipFree(b1); hipFree(bfc); hipFree(flags); hipFree(flag);
    return total != 0 || hf != 0;
}

static int timek(int R, int T, int K, int runs)
{
    const int N = R * K;
    const size_t nz = (size_t)R * T * K * 64, nst = (size_t)4 * N * 64;
    const size_t nw = (size_t)2 * 4 * 4 * 4 * 2 * 64 * 8, nfc = (size_t)4 * 2 * 2 * 64 * 8;
    float *z, *zs, *part, *o_ref, *o_part, *b, *bfc, *st_in, *so; uint16_t *w, *wfc;
    CK(hipMalloc(&z, nz * 4)); CK(hipMalloc(&zs, nz * 4)); CK(hipMalloc(&part, nz * 8)); CK(hipMalloc(&o_ref, nz * 4)); CK(hipMalloc(&o_part, nz * 4));
    CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&wfc, nfc * 2)); CK(hipMalloc(&b, 512 * 4)); CK(hipMalloc(&bfc, 64 * 4));
    CK(hipMalloc(&st_in, nst * 4)); CK(hipMalloc(&so, nst * 4));
    std::vector<float> hz(nz), hzs(nz), hp(2 * nz), hb(512), hbfc(64), hst(nst);
    for (auto& v : hz) v = urand();
    for (auto& v : hp) v = urand();
    for (size_t i = 0; i < nz; ++i) { const size_t row = i / 64, c = i % 64; hzs[i] = (hz[i] + hp[row * 128 + c]) + hp[row * 128 + 64 + c]; }
    for (auto& v : hb) v = urand() * 0.5f;
    for (auto& v : hbfc) v = urand() * 0.5f;
    for (auto& v : hst) v = urand() * 0.9f;
    std::vector<uint16_t> hw(nw), hwfc(nfc);
    for (auto& v : hw) v = small_half();
    for (auto& v : hwfc) v = small_half();
    CK(hipMemcpy(z, hz.data(), nz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(zs, hzs.data(), nz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(part, hp.data(), nz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfc, hwfc.data(), nfc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bfc, hbfc.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(st_in, hst.data(), nst * 4, hipMemcpyHostToDevice));
    const dim3 grid((N + 3) / 4), block16(1024);
    printf("== time kernel, R=%d T=%d K=%d (%d workgroups)\n", R, T, K, grid.x);
    int* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false, false>), grid, block16, 0, 0, zs, o_ref, (const uint4*)w, b, (const uint4*)wfc, bfc, st_in, so, R, T, K, flag, (unsigned long long*)nullptr, (const float*)nullptr);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> ref(nz), cur(nz);
    CK(hipMemcpy(ref.data(), o_ref, nz * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f; size_t total = 0;
    for (int r = 0; r < runs; ++r) {
        CK(hipMemset(o_part, 0xff, nz * 4));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false, true>), grid, block16, 0, 0, z, o_part, (const uint4*)w, b, (const uint4*)wfc, bfc, st_in, so, R, T, K, flag, (unsigned long long*)nullptr, part);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms * 1e3f);
        CK(hipMemcpy(cur.data(), o_part, nz * 4, hipMemcpyDeviceToHost));
        size_t d = 0, first_t = (size_t)-1, seq = 0;
        for (size_t i = 0; i < nz; ++i)
            if (cur[i] != ref[i]) { ++d; const size_t row = i / 64, t = (row / K) % T; if (t < first_t) { first_t = t; seq = (row / ((size_t)T * K)) * K + row % K; } }
        total += d;
        if (d) printf("   run %d: %zu words differ from the kernel on pre-added rows; earliest step %zu (sequence %zu)\n", r, d, first_t, seq);
    }
    int hf = 0; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("   PART launch %.1f us; words differing over %d runs: %zu; flag %d\n", best, runs, total, hf);
    hipFree(z); hipFree(zs); hipFree(part); hipFree(o_ref); hipFree(o_part); hipFree(w); hipFree(wfc); hipFree(b); hipFree(bfc); hipFree(st_in); hipFree(so); hipFree(flag);
    return total != 0 || hf != 0;
}

int main(int argc, char** argv)
{
    int fails = 0;
    if (argc > 1) { fails += band(atoi(argv[1]), argc > 2 ? atoi(argv[2]) : 12, argc > 3 ? atoi(argv[3]) : 6); printf(fails ? "FAILED\n" : "ok\n"); return fails; }
    fails += band(37, 12, 3);
    fails += band(8064, 12, 4);
    fails += band(32768, 12, 6);
    fails += band(700, 42, 3);
    fails += pair(37, 12, 3);
    fails += pair(8064, 12, 6);
    fails += pair(32768, 12, 6);
    fails += pair(700, 42, 3);
    fails += timek(2, 5, 12, 3);
    fails += timek(3, 9, 5, 3);
    fails += timek(64, 126, 12, 6);
    fails += timek(64, 512, 12, 8);
    printf(fails ? "FAILED (%d)\n" : "all checks passed\n", fails);
    return fails != 0;
}
