// Measurement / validation only: the counter-phase time-axis LSTM kernel (time_lstm_h2w_kernel) against the round-2 kernel
// (time_lstm_h2_kernel): bit-identity of h1 and of the carried state without the fused fc, the fused fc + residual against a
// double-precision evaluation of fc(h1) + b + x, timing, in-kernel phase stamps.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o build/time_lstm_v3_bench tools/time_lstm_v3_bench.hip
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
static unsigned g_seed = 12345u;
static int lcg() { g_seed = g_seed * 1664525u + 1013904223u; return (int)((g_seed >> 8) & 0x7fffff); }
static float urand() { return lcg() / (float)0x7fffff - 0.5f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double h2d(uint16_t b) { _Float16 v; memcpy(&v, &b, 2); return (double)(float)v; }

static int run(int R, int T, int K, bool with_state, int reps)
{
    const int N = R * K;
    const size_t nz = (size_t)R * T * K * 64, nst = (size_t)4 * N * 64;
    const size_t nw = (size_t)2 * 4 * 4 * 4 * 2 * 64 * 8, nfc = (size_t)4 * 2 * 2 * 64 * 8;
    float *z, *h_old, *h_new, *h_fus, *b, *bfc, *st_in, *so_old, *so_new, *so_fus; uint16_t *w, *wfc; unsigned long long* dbg;
    CK(hipMalloc(&z, nz * 4)); CK(hipMalloc(&h_old, nz * 4)); CK(hipMalloc(&h_new, nz * 4)); CK(hipMalloc(&h_fus, nz * 4));
    CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&wfc, nfc * 2)); CK(hipMalloc(&b, 512 * 4)); CK(hipMalloc(&bfc, 64 * 4));
    CK(hipMalloc(&st_in, nst * 4)); CK(hipMalloc(&so_old, nst * 4)); CK(hipMalloc(&so_new, nst * 4)); CK(hipMalloc(&so_fus, nst * 4));
    CK(hipMalloc(&dbg, 4 * 16 * 4 * 8));
    std::vector<float> hz(nz), hb(512), hbfc(64), hst(nst);
    for (auto& v : hz) v = urand();
    for (auto& v : hb) v = urand() * 0.5f;
    for (auto& v : hbfc) v = urand() * 0.5f;
    for (auto& v : hst) v = urand() * 0.9f;
    std::vector<uint16_t> hw(nw), hwfc(nfc);
    for (auto& v : hw) v = (uint16_t)(0x2c00 + (lcg() & 0x3ff) + ((lcg() & 1) << 15));          // fp16 in +-[0.06, 0.12)
    for (auto& v : hwfc) v = (uint16_t)(0x2c00 + (lcg() & 0x3ff) + ((lcg() & 1) << 15));
    CK(hipMemcpy(z, hz.data(), nz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfc, hwfc.data(), nfc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bfc, hbfc.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(st_in, hst.data(), nst * 4, hipMemcpyHostToDevice));
    CK(hipMemset(h_old, 0xff, nz * 4)); CK(hipMemset(h_new, 0xff, nz * 4)); CK(hipMemset(h_fus, 0xff, nz * 4));
    const float* sin_ = with_state ? st_in : nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((N + 3) / 4), block(512), block16(1024);
    printf("== R=%d T=%d K=%d (N=%d sequences, %d workgroups) state_in=%d\n", R, T, K, N, (N + 3) / 4, (int)with_state);
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < reps; ++rep)
        for (int v = 0; v < 3; ++v) {
            CK(hipEventRecord(e0, 0));
            if (v == 0) hipLaunchKernelGGL(time_lstm_h2_kernel<false>, grid, block, 0, 0, z, h_old, (const uint4*)w, b, sin_, so_old, R, T, K, (int*)nullptr, dbg);
            if (v == 1) hipLaunchKernelGGL((time_lstm_h2w_kernel<false, false>), grid, block16, 0, 0, z, h_new, (const uint4*)w, b, (const uint4*)nullptr, (const float*)nullptr, sin_, so_new, R, T, K, (int*)nullptr, dbg);
            if (v == 2) hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false>), grid, block16, 0, 0, z, h_fus, (const uint4*)w, b, (const uint4*)wfc, bfc, sin_, so_fus, R, T, K, (int*)nullptr, dbg);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best[v] = std::min(best[v], ms * 1e3f);
        }
    CK(hipGetLastError());
    printf("   launch time (best of %d): round-2 kernel %.1f us | 16 waves %.1f us | 16 waves + fused fc %.1f us\n", reps - 1, best[0], best[1], best[2]);
    std::vector<uint32_t> a(nz), c(nz);
    std::vector<float> f(nz), ho(nz);
    CK(hipMemcpy(a.data(), h_old, nz * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), h_new, nz * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f.data(), h_fus, nz * 4, hipMemcpyDeviceToHost)); memcpy(ho.data(), a.data(), nz * 4);
    size_t nd = 0;
    for (size_t i = 0; i < nz; ++i) nd += a[i] != c[i];
    std::vector<uint32_t> sa(nst), sb(nst), sc(nst);
    CK(hipMemcpy(sa.data(), so_old, nst * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), so_new, nst * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sc.data(), so_fus, nst * 4, hipMemcpyDeviceToHost));
    size_t nds = 0, ndf = 0;
    for (size_t i = 0; i < nst; ++i) { nds += sa[i] != sb[i]; ndf += sa[i] != sc[i]; }
    printf("   h1: %zu of %zu words differ from the round-2 kernel; state_out: %zu (plain) / %zu (fused) of %zu differ\n", nd, nz, nds, ndf, nst);
    // fused output against fc(h1) + b + x in double (h1 = the round-2 kernel's, which the plain kernel reproduces bit for bit)
    double worst = 0; size_t bad = 0, checked = 0;
    for (int nn = 0; nn < N; nn += (N > 64 ? 5 : 1))
        for (int t = 0; t < T; ++t) {
            const size_t row = ((size_t)(nn / K) * T * K + (size_t)t * K + (nn % K)) * 64;
            for (int o = 0; o < 64; ++o) {
                double acc = hbfc[o];
                for (int k = 0; k < 64; ++k) {
                    const int wv = o / 16, ln = (o % 16) + 16 * ((k % 32) / 8), blk = k / 32, j = k % 8;
                    const size_t base = (((size_t)wv * 2 + blk) * 2) * 64 * 8;
                    const double wk = h2d(hwfc[base + (size_t)ln * 8 + j]) + h2d(hwfc[base + 64 * 8 + (size_t)ln * 8 + j]) / 2048.0;
                    acc += wk * (double)ho[row + k];
                }
                acc += (double)hz[row + o];
                const double d = fabs(acc - (double)f[row + o]);
                worst = std::max(worst, d); bad += !(d < 2e-6); ++checked;
            }
        }
    printf("   fused fc + residual: max |kernel - double| = %.3e over %zu values (%zu above 2e-6)\n", worst, checked, bad);
    size_t untouched = 0;
    for (size_t i = 0; i < nz; ++i) { uint32_t u; memcpy(&u, &f[i], 4); untouched += u == 0xffffffffu; }
    printf("   fused output words never written: %zu\n", untouched);
    size_t nd8 = 0, ndf8 = 0, nds8 = 0, ndsf8 = 0;
    {   // the eight-sequence kernel (time_lstm_h2w8_kernel): h1 / the fused output / the carried state bit for bit against the four-sequence kernel's
        float *h8p, *h8f, *so8, *so8f;
        CK(hipMalloc(&h8p, nz * 4)); CK(hipMalloc(&h8f, nz * 4)); CK(hipMalloc(&so8, nst * 4)); CK(hipMalloc(&so8f, nst * 4));
        CK(hipMemset(h8p, 0xff, nz * 4)); CK(hipMemset(h8f, 0xff, nz * 4));
        const dim3 grid8((N + 7) / 8);
        float best8[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < reps; ++rep)
            for (int v = 0; v < 2; ++v) {
                CK(hipEventRecord(e0, 0));
                if (v == 0) hipLaunchKernelGGL((time_lstm_h2w8_kernel<false, false>), grid8, block16, 0, 0, z, h8p, (const uint4*)w, b, (const uint4*)nullptr, (const float*)nullptr, sin_, so8, R, T, K, (int*)nullptr, dbg);
                else hipLaunchKernelGGL((time_lstm_h2w8_kernel<true, false>), grid8, block16, 0, 0, z, h8f, (const uint4*)w, b, (const uint4*)wfc, bfc, sin_, so8f, R, T, K, (int*)nullptr, dbg);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best8[v] = std::min(best8[v], ms * 1e3f);
            }
        CK(hipGetLastError());
        std::vector<uint32_t> u8(nz), uf8(nz), us8(nst), usf8(nst), uf4(nz);
        CK(hipMemcpy(u8.data(), h8p, nz * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(uf8.data(), h8f, nz * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(us8.data(), so8, nst * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(usf8.data(), so8f, nst * 4, hipMemcpyDeviceToHost));
        memcpy(uf4.data(), f.data(), nz * 4);
        for (size_t i = 0; i < nz; ++i) { nd8 += u8[i] != c[i]; ndf8 += uf8[i] != uf4[i]; }
        for (size_t i = 0; i < nst; ++i) { nds8 += us8[i] != sb[i]; ndsf8 += usf8[i] != sc[i]; }
        printf("   EIGHT sequences per workgroup (%d workgroups): %.1f us | + fused fc %.1f us;  words that differ from the four-sequence kernel: h1 %zu, fused output %zu, state %zu / %zu\n",
               (N + 7) / 8, best8[0], best8[1], nd8, ndf8, nds8, ndsf8);
        hipFree(h8p); hipFree(h8f); hipFree(so8); hipFree(so8f);
    }
    if (reps > 2)            // phase stamps of the 16-wave kernel, without and with the fused fc
        for (int fz = 0; fz < 2; ++fz) {
            if (fz) hipLaunchKernelGGL((time_lstm_h2w_kernel<true, true>), grid, block16, 0, 0, z, h_fus, (const uint4*)w, b, (const uint4*)wfc, bfc, sin_, so_fus, R, T, K, (int*)nullptr, dbg);
            else hipLaunchKernelGGL((time_lstm_h2w_kernel<false, true>), grid, block16, 0, 0, z, h_new, (const uint4*)w, b, (const uint4*)nullptr, (const float*)nullptr, sin_, so_new, R, T, K, (int*)nullptr, dbg);
            CK(hipDeviceSynchronize());
            unsigned long long hd[4 * 16 * 4];
            CK(hipMemcpy(hd, dbg, sizeof hd, hipMemcpyDeviceToHost));
            printf("   stamps, %s:\n", fz ? "fused fc" : "plain");
            for (int wv = 0; wv < 8; wv += 4) {
                const unsigned long long* d = &hd[wv * 4];
                printf("     wave %2d (main %d): per step: fragments + MFMAs + cell %.0f ns, wait for h(t-1) %.0f ns, wait for the group's input half / ring %.0f ns; total %.1f us\n",
                       wv, wv / 4, d[0] * 10.0 / T, d[2] * 10.0 / T, d[3] * 10.0 / T, (d[0] + d[2] + d[3]) / 100.0);
            }
            const int G = (T + 3) / 4;
            printf("     wave  8 (helper 0): per group: staging + waits %.0f ns, input half %.0f ns\n", hd[8 * 4] * 10.0 / G, hd[8 * 4 + 2] * 10.0 / G);
            printf("     wave 12 (helper 1): per group: waits %.0f ns, input half %.0f ns, fc %.0f ns\n", hd[12 * 4] * 10.0 / G, hd[12 * 4 + 2] * 10.0 / G, hd[12 * 4 + 1] * 10.0 / G);
        }
    if (reps > 2)            // phase stamps of the eight-sequence kernel
        for (int fz = 0; fz < 2; ++fz) {
            const dim3 grid8((N + 7) / 8);
            if (fz) hipLaunchKernelGGL((time_lstm_h2w8_kernel<true, true>), grid8, block16, 0, 0, z, h_fus, (const uint4*)w, b, (const uint4*)wfc, bfc, sin_, so_fus, R, T, K, (int*)nullptr, dbg);
            else hipLaunchKernelGGL((time_lstm_h2w8_kernel<false, true>), grid8, block16, 0, 0, z, h_new, (const uint4*)w, b, (const uint4*)nullptr, (const float*)nullptr, sin_, so_new, R, T, K, (int*)nullptr, dbg);
            CK(hipDeviceSynchronize());
            unsigned long long hd[4 * 16 * 4];
            CK(hipMemcpy(hd, dbg, sizeof hd, hipMemcpyDeviceToHost));
            printf("   stamps, EIGHT sequences, %s:\n", fz ? "fused fc" : "plain");
            for (int wv = 0; wv < 8; wv += 4) {
                const unsigned long long* d = &hd[wv * 4];
                printf("     wave %2d (main %d): per step: fragments + MFMAs + cells %.0f ns, wait for h(t-1) %.0f ns, wait for the group's input half / ring %.0f ns; total %.1f us\n",
                       wv, wv / 4, d[0] * 10.0 / T, d[2] * 10.0 / T, d[3] * 10.0 / T, (d[0] + d[2] + d[3]) / 100.0);
            }
            const int G2 = (T + 1) / 2;
            printf("     wave  8 (helper 0): per group of two: staging + waits %.0f ns, input half %.0f ns\n", hd[8 * 4] * 10.0 / G2, hd[8 * 4 + 2] * 10.0 / G2);
            printf("     wave 12 (helper 1): per group of two: waits %.0f ns, input half %.0f ns, fc %.0f ns\n", hd[12 * 4] * 10.0 / G2, hd[12 * 4 + 2] * 10.0 / G2, hd[12 * 4 + 1] * 10.0 / G2);
        }
    const int fail = (nd != 0) + (nds != 0) + (ndf != 0) + (bad != 0) + (untouched != 0) + (nd8 != 0) + (ndf8 != 0) + (nds8 != 0) + (ndsf8 != 0);
    hipFree(z); hipFree(h_old); hipFree(h_new); hipFree(h_fus); hipFree(w); hipFree(wfc); hipFree(b); hipFree(bfc);
    hipFree(st_in); hipFree(so_old); hipFree(so_new); hipFree(so_fus); hipFree(dbg);
    return fail;
}

int main()
{
    int fails = 0;
    fails += run(2, 1, 12, true, 2);
    fails += run(2, 5, 12, true, 2);
    fails += run(3, 9, 5, false, 2);           // N = 15: a partial workgroup
    fails += run(2, 8, 12, true, 2);
    fails += run(1, 26, 42, true, 2);
    fails += run(64, 126, 12, true, 6);
    fails += run(64, 126, 12, false, 3);
    fails += run(32, 376, 42, false, 3);
    printf(fails ? "FAILED (%d)\n" : "all checks passed\n", fails);
    return fails != 0;
}
