// Measurement-only micro-benchmark of the grouped fp32-MFMA GEMM (not part of the product):
// runs the PRE0-shaped job set (per-band square Linear layers 2w x 2w over M frame rows) with
// ablation variants interleaved in one process (cdna guide rule 24) and prints median times.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/gemm_bench tools/gemm_bench.hip && gpurun_out/gemm_bench
#include "../speechseparation_amd/csrc/gemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdio>
#include <vector>

using namespace bsrnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT, int ABL, int PRIO = 0, int VEC = 2>
static void run(const GemmLaunch& g, hipStream_t s)
{
    launch_gemm_nt<NT, ABL, PRIO, VEC>(g, s);
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 8064;
    int widths[11] = {1, 2, 3, 6, 12, 24, 48, 96, 192, 384, 257};
    int nb = 11;
    if (argc > 2) { nb = 1; widths[0] = atoi(argv[2]); }
    const bool align4 = getenv("ALIGN4") != nullptr;
    if (align4 && nb == 11) { int w4[11] = {2, 2, 4, 6, 12, 24, 48, 96, 192, 384, 258}; for (int i = 0; i < 11; ++i) widths[i] = w4[i]; }
    const int LD = align4 ? 2064 : 2050;     // uniform: a single square job of 2*w columns
    std::vector<GemmJob> jobs;
    std::vector<int2> tiles;
    size_t wtot = 0;
    int off = 0;
    std::vector<size_t> woff;
    for (int i = 0; i < nb; ++i) {
        GemmJob j = {};
        j.N = j.K = 2 * widths[i];
        j.x_off = j.y_off = 2 * off;
        off += widths[i];
        woff.push_back(wtot);
        wtot += (size_t)j.N * j.K + j.N;
        wtot = (wtot + 3) & ~size_t(3);
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
    float *dW, *dX, *dY;
    CK(hipMalloc(&dW, wtot * 4));
    CK(hipMalloc(&dX, (size_t)M * 2064 * 4));
    CK(hipMalloc(&dY, (size_t)M * 2064 * 4));
    std::vector<float> h(wtot);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(dW, h.data(), wtot * 4, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)M * 2064);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(dX, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < nb; ++i) { jobs[i].W = dW + woff[i]; jobs[i].bias = dW + woff[i] + (size_t)jobs[i].N * jobs[i].K; }
    GemmJob* dJ; int2 *dT, *dT128;
    CK(hipMalloc(&dJ, jobs.size() * sizeof(GemmJob)));
    CK(hipMalloc(&dT, tiles.size() * sizeof(int2)));
    CK(hipMalloc(&dT128, tiles128.size() * sizeof(int2)));
    CK(hipMemcpy(dJ, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT128, tiles128.data(), tiles128.size() * sizeof(int2), hipMemcpyHostToDevice));
    GemmLaunch g = {};
    g.jobs = dJ; g.tiles = dT; g.n_tiles = (int)tiles.size(); g.tile_n = 64;
    g.X = dX; g.ldx = LD; g.Y = dY; g.ldy = LD; g.M = M; g.epilogue = EPI_LEAKY;
    GemmLaunch g2 = g;
    g2.tiles = dT128; g2.n_tiles = (int)tiles128.size(); g2.tile_n = 128;
    double flop = 0;
    for (auto& j : jobs) flop += 2.0 * j.N * j.K * M;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[12] = {"64: full", "64: no-gload", "64: no-gload no-barrier/ldswrite", "128: mfma only", "64: mfma only (no gload/barrier/lds)", "64: full prio-mfma VEC4 (needs ALIGN4)", "128: full VEC4 (needs ALIGN4)",
                             "64: no-gload prio-mfma", "64: no-gload static-prio", "128: full", "128: full prio-mfma", "128: full static-prio"};
    constexpr int NV = 12;
    std::vector<float> t[NV];
    for (int rep = 0; rep < 12; ++rep)
        for (int v = 0; v < NV; ++v) {
            CK(hipEventRecord(a, s));
            switch (v) {
            case 0: run<1, 0>(g, s); break; case 1: run<1, 1>(g, s); break; case 2: run<1, 5>(g, s); break;
            case 3: run<2, 13>(g2, s); break; case 4: run<1, 13>(g, s); break;
            case 5: if (align4) run<1, 0, 1, 4>(g, s); else run<1, 0, 1>(g, s); break; case 6: if (align4) run<2, 0, 0, 4>(g2, s); else run<2, 0>(g2, s); break; case 7: run<1, 1, 1>(g, s); break; case 8: run<1, 1, 2>(g, s); break;
            case 9: run<2, 0>(g2, s); break; case 10: run<2, 0, 1>(g2, s); break; default: run<2, 0, 2>(g2, s); }
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2) t[v].push_back(ms);
        }
    printf("M=%d tiles=%d GFLOP=%.2f\n", M, g.n_tiles, flop / 1e9);
    if (getenv("GEMM_TRACE")) {       // placement / timeline of every workgroup (64-wide kernel, full and MFMA-only)
        const int m_tiles = (M + BM - 1) / BM, mc = gemm_mchunk(m_tiles);
        const int nblk = 8 * (((m_tiles + mc - 1) / mc + 7) / 8) * mc * g.n_tiles;
        unsigned long long* dT2;
        CK(hipMalloc(&dT2, (size_t)nblk * 128));
        std::vector<unsigned long long> ht((size_t)nblk * 16);
        for (int var = 0; var < 2; ++var) {
            CK(hipMemset(dT2, 0, (size_t)nblk * 128));
            GemmLaunch gt = g;
            gt.tap = reinterpret_cast<float*>(dT2);
            if (var == 0) run<1, 16>(gt, s); else run<1, 16 + 13>(gt, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(ht.data(), dT2, (size_t)nblk * 128, hipMemcpyDeviceToHost));
            FILE* f = fopen(var == 0 ? "gpurun_out/gemm_trace_full.csv" : "gpurun_out/gemm_trace_mfma.csv", "w");
            fprintf(f, "block,wave,xcc,hw_id,t0,t1\n");
            for (int i = 0; i < nblk * 4; ++i)
                if (ht[4 * i + 3]) fprintf(f, "%d,%d,%llu,%llu,%llu,%llu\n", i / 4, i % 4, ht[4 * i], ht[4 * i + 1], ht[4 * i + 2], ht[4 * i + 3]);
            fclose(f);
        }
    }
    for (int v = 0; v < NV; ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-18s median %.1f us  min %.1f us  -> %.1f TFLOP/s-equivalent\n", names[v], med * 1e3, t[v][0] * 1e3, flop / (med * 1e-3) / 1e12);
    }
    return 0;
}
