// Measurement-only micro-benchmark of the split-precision ("planes") grouped GEMM against the fp32-MFMA kernel
// on the PRE0-shaped job set (per-band square layers, widths rounded so that every segment is 16-byte aligned
// in bf16).  Variants interleaved in one process; prints medians and the max difference to the fp32 kernel.
//   hipcc -O3 --offload-arch=gfx950 -o build/gemm_planes_bench tools/gemm_planes_bench.hip
#include "../speechseparation_amd/csrc/gemm.hip"
#include "../speechseparation_amd/csrc/split_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace bsrnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 8064;
    int widths[11] = {4, 4, 4, 8, 12, 24, 48, 96, 192, 384, 260};
    int nb = 11;
    if (argc > 2) { nb = 1; widths[0] = atoi(argv[2]); }
    const int LD = 2080;
    std::vector<GemmJob> jobs;
    std::vector<int2> tiles;
    size_t wtot = 0;
    int off = 0;
    std::vector<size_t> woff;
    for (int i = 0; i < nb; ++i) {
        GemmJob j = {};
        j.N = j.K = 2 * widths[i];
        if (const char* kc = getenv("KCAP")) j.K = std::min(j.K, atoi(kc));
        j.x_off = j.y_off = 2 * off;
        off += widths[i];
        woff.push_back(wtot);
        wtot += (size_t)j.N * j.K + j.N;
        wtot = (wtot + 7) & ~size_t(7);
        jobs.push_back(j);
    }
    std::vector<int> order(nb);
    for (int i = 0; i < nb; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return jobs[a].K > jobs[b].K; });
    std::vector<int2> tiles128;
    for (int i : order) {
        for (int t = 0; t < (jobs[i].N + 63) / 64; ++t) tiles.push_back(make_int2(i, t));
        for (int t = 0; t < (jobs[i].N + 127) / 128; ++t) tiles128.push_back(make_int2(i, t));
    }
    float *dW, *dX, *dY, *dY2;
    uint16_t *dWp, *dXp, *dYp, *dWp16, *dXp16, *dWq16;
    const size_t xn = (size_t)M * LD;
    CK(hipMalloc(&dW, wtot * 4));
    CK(hipMalloc(&dWp, wtot * 3 * 2));
    CK(hipMalloc(&dX, xn * 4));
    CK(hipMalloc(&dY, xn * 4));
    CK(hipMalloc(&dY2, xn * 4));
    CK(hipMalloc(&dXp, xn * 3 * 2));
    CK(hipMalloc(&dYp, xn * 3 * 2));
    CK(hipMalloc(&dWp16, wtot * 2 * 2));
    CK(hipMalloc(&dXp16, xn * 2 * 2));
    CK(hipMemset(dY, 0, xn * 4)); CK(hipMemset(dY2, 0, xn * 4)); CK(hipMemset(dYp, 0, xn * 6));
    std::vector<float> h(wtot);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(dW, h.data(), wtot * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> hp(wtot * 3);
    for (int i = 0; i < nb; ++i) {
        const size_t n = (size_t)jobs[i].N * jobs[i].K;
        split_planes_host(&h[woff[i]], n, 3, &hp[3 * woff[i]]);
    }
    CK(hipMemcpy(dWp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < nb; ++i) split_planes_host(&h[woff[i]], (size_t)jobs[i].N * jobs[i].K, 2, &hp[2 * woff[i]]);
    CK(hipMemcpy(dWp16, hp.data(), wtot * 2 * 2, hipMemcpyHostToDevice));
    // slab-interleaved fp16x2 weights for the pipelined kernel (rows padded to multiples of 32)
    std::vector<size_t> qoff;
    size_t qtot = 0;
    const int rowpad = getenv("ROWPAD") ? atoi(getenv("ROWPAD")) : -1;     // -1: the library's rule, else extra 16-bit elements per row
    for (int i = 0; i < nb; ++i) {
        const int K32 = (jobs[i].K + 31) & ~31;
        jobs[i].wrow = rowpad < 0 ? h2_row_stride(K32) : 2 * K32 + rowpad;
        qoff.push_back(qtot); qtot += (size_t)jobs[i].N * jobs[i].wrow;
    }
    std::vector<uint16_t> hq(qtot + 8);
    for (int i = 0; i < nb; ++i) pack_h2_slabs_host(&h[woff[i]], jobs[i].N, jobs[i].K, jobs[i].K, (jobs[i].K + 31) & ~31, jobs[i].wrow, &hq[qoff[i]]);
    CK(hipMalloc(&dWq16, hq.size() * 2));
    CK(hipMemcpy(dWq16, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hx(xn);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(dX, hx.data(), xn * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint16_t> xp(xn * 3);
        split_planes_host(hx.data(), xn, 3, xp.data());
        CK(hipMemcpy(dXp, xp.data(), xp.size() * 2, hipMemcpyHostToDevice));
        split_planes_host(hx.data(), xn, 2, xp.data());
        CK(hipMemcpy(dXp16, xp.data(), xn * 2 * 2, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < nb; ++i) {
        jobs[i].W = dW + woff[i]; jobs[i].bias = dW + woff[i] + (size_t)jobs[i].N * jobs[i].K;
        jobs[i].Wp = dWp + 3 * woff[i];
    }
    GemmJob *dJ, *dJ16, *dJq; int2 *dT, *dT128;
    CK(hipMalloc(&dT128, tiles128.size() * sizeof(int2)));
    CK(hipMemcpy(dT128, tiles128.data(), tiles128.size() * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMalloc(&dJ, jobs.size() * sizeof(GemmJob)));
    CK(hipMalloc(&dT, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(dJ, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    for (int i = 0; i < nb; ++i) jobs[i].Wp = dWp16 + 2 * woff[i];
    CK(hipMalloc(&dJ16, jobs.size() * sizeof(GemmJob)));
    CK(hipMemcpy(dJ16, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    for (int i = 0; i < nb; ++i) jobs[i].Wp = dWq16 + qoff[i];
    CK(hipMalloc(&dJq, jobs.size() * sizeof(GemmJob)));
    CK(hipMemcpy(dJq, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    GemmLaunch g = {};
    g.jobs = dJ; g.tiles = dT; g.n_tiles = (int)tiles.size(); g.tile_n = 64;
    g.X = dX; g.ldx = LD; g.Y = dY; g.ldy = LD; g.M = M; g.epilogue = EPI_LEAKY;
    g.Xp = dXp; g.xp_plane = xn; g.Yp = dYp; g.yp_plane = xn; g.out_mode = 1;
    GemmLaunch g1 = g; g1.Y = dY2;                       // 64-wide tiles, fp32 out
    GemmLaunch g2 = g1; g2.tiles = dT128; g2.n_tiles = (int)tiles128.size(); g2.tile_n = 128;
    GemmLaunch g2p = g2; g2p.out_mode = 2;
    GemmLaunch h1 = g1; h1.jobs = dJ16; h1.Xp = dXp16;       // fp16x2 operands
    GemmLaunch h2 = g2; h2.jobs = dJ16; h2.Xp = dXp16;
    GemmLaunch h2p = h2; h2p.out_mode = 2;
    GemmLaunch q1 = h1; q1.jobs = dJq;                       // pipelined kernel: slab-interleaved weights
    GemmLaunch q2 = h2; q2.jobs = dJq;
    double flop = 0;
    for (auto& j : jobs) flop += 2.0 * j.N * j.K * M;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    constexpr int NV = 12;
    const char* names[NV] = {"fp32 mfma (product kernel)", "fp16x2  64: A fly, B planes", "fp16x2 128: A fly, B planes (product)", "fp16x2 128: 8 waves",
                             "fp16x2 128: product, no global stores", "fp16x2 128: product, no epilogue", "fp16x2 128: no gload, no epilogue", "fp16x2 128: no mfma, no epilogue",
                             "fp16x2 128 pipelined", "fp16x2  64 pipelined", "fp16x2 128 pipelined, no gload", "fp16x2 128 pipelined, no split VALU"};
    std::vector<float> t[NV];
    for (int rep = 0; rep < 12; ++rep)
        for (int v = 0; v < NV; ++v) {
            CK(hipEventRecord(a, s));
            switch (v) {
            case 0: launch_gemm_nt<1, 0, 1, 4>(g, s); break;
            case 1: launch_gemm_split<2, 1, 0, 1, 1>(h1, s); break;
            case 2: launch_gemm_split<2, 2, 0, 1, 1>(h2, s); break;
            case 3: launch_gemm_split<2, 2, 0, 1, 1, 1, 0, 8>(h2, s); break;
            case 4: launch_gemm_split<2, 2, 0, 1, 1, 1, 4>(h2, s); break;
            case 5: launch_gemm_split<2, 2, 0, 1, 1, 1, 8>(h2, s); break;
            case 6: launch_gemm_split<2, 2, 0, 1, 1, 1, 9>(h2, s); break;
            case 7: launch_gemm_split<2, 2, 0, 1, 1, 1, 10>(h2, s); break;
            case 8: launch_gemm_h2<2>(q2, s); break;
            case 9: launch_gemm_h2<1>(q1, s); break;
            case 10: launch_gemm_h2<2, 1>(q2, s); break;
            default: launch_gemm_h2<2, 32>(q2, s); }
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2) t[v].push_back(ms);
        }
    printf("M=%d tiles=%d GFLOP=%.2f\n", M, g.n_tiles, flop / 1e9);
    for (int v = 0; v < NV; ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-36s median %.1f us  min %.1f us  -> %.1f TFLOP/s-equivalent\n", names[v], med * 1e3, t[v][0] * 1e3, flop / (med * 1e-3) / 1e12);
    }
    if (getenv("GEMM_TRACE")) {        // phase times of the product variant, summed over all waves of all workgroups
        const int m_tiles = (M + BM - 1) / BM, mc = gemm_mchunk(m_tiles);
        const int nblk = 8 * (((m_tiles + mc - 1) / mc + 7) / 8) * mc * h2.n_tiles;
        unsigned long long* dTr;
        CK(hipMalloc(&dTr, (size_t)nblk * 4 * 64));
        CK(hipMemset(dTr, 0, (size_t)nblk * 4 * 64));
        GemmLaunch gt = h2;
        gt.tap = reinterpret_cast<float*>(dTr);
        launch_gemm_split<2, 2, 0, 1, 1, 1, 16>(gt, s);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> ht((size_t)nblk * 4 * 8);
        CK(hipMemcpy(ht.data(), dTr, ht.size() * 8, hipMemcpyDeviceToHost));
        double sum[5] = {0, 0, 0, 0, 0}, ksteps = 0, cyc = 0, rt = 0;
        for (size_t i = 0; i < (size_t)nblk * 4; ++i)
            if (ht[8 * i + 7]) {
                for (int k = 0; k < 5; ++k) sum[k] += ht[8 * i + k];
                ksteps += ht[8 * i + 5];
                if (ht[8 * i + 5] >= 16) { cyc += ht[8 * i + 6]; rt += ht[8 * i + 7] - 1; }
            }
        printf("in-kernel clock over the main loops (s_memtime / s_memrealtime): %.2f GHz\n", cyc / rt * 0.1);
        const char* nm[5] = {"barrier 1 (wait for the other waves' MFMAs)", "split + ds_write issue", "barrier 2 (LDS writes landed)", "global load issue", "ds_read + MFMA"};
        printf("phase times per wave and k-step (ns), %0.f wave-k-steps:\n", ksteps);
        for (int k = 0; k < 5; ++k) printf("  %-46s %7.0f\n", nm[k], sum[k] * 10.0 / ksteps);
    }
    // numerics against the fp32-MFMA kernel: bf16x3 (fp32 out), fp16x2 (fp32 out), fp16x2 plane output recombined
    launch_gemm_nt<1, 0, 1, 4>(g, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> y0(xn), y1(xn);
    CK(hipMemcpy(y0.data(), dY, xn * 4, hipMemcpyDeviceToHost));
    double mx = 0;
    for (size_t i = 0; i < xn; ++i) mx = std::max(mx, (double)fabsf(y0[i]));
    auto diff = [&](const char* what) {
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(y1.data(), dY2, xn * 4, hipMemcpyDeviceToHost));
        double d = 0, sq = 0;
        for (size_t i = 0; i < xn; ++i) { const double e = (double)y0[i] - y1[i]; d = std::max(d, fabs(e)); sq += e * e; }
        printf("%-44s max|diff| %.3e  rms %.3e   (max|y| %.3f)\n", what, d, sqrt(sq / xn), mx);
    };
    launch_gemm_split<3, 2, 0, 1, 0, 2>(g2, s); diff("bf16x3 128 (1 acc) vs fp32 mfma");
    launch_gemm_split<2, 2, 0, 1, 1, 2>(h2, s); diff("fp16x2 128, A fly, B planes vs fp32 mfma");
    launch_gemm_split<2, 2, 0, 1, 1, 1, 0, 8>(h2, s); diff("fp16x2 128, 8 waves vs fp32 mfma");
    launch_gemm_h2<2>(q2, s); diff("fp16x2 128 pipelined vs fp32 mfma");
    launch_gemm_h2<1>(q1, s); diff("fp16x2 64 pipelined vs fp32 mfma");
    launch_gemm_split<2, 2, 1, 1, 1, 2>(h2p, s);
    CK(hipStreamSynchronize(s));
    {
        std::vector<uint16_t> yp(xn * 2);
        CK(hipMemcpy(yp.data(), dYp, xn * 4, hipMemcpyDeviceToHost));
        double d = 0;
        for (size_t i = 0; i < xn; ++i) d = std::max(d, (double)fabsf(y0[i] - join_planes_host(yp.data(), xn, i, 2)));
        printf("%-44s max|diff| %.3e\n", "fp16x2 plane output recombined vs fp32 mfma", d);
    }
    return 0;
}
