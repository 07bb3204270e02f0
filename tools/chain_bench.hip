// Measurement-only probe for the fused chain kernel (speechseparation_amd/csrc/mlp_chain.hip): runs the BandSplit or the
// MaskEstimation chain on synthetic weights for a chosen subset of bands, so that the cost of one band's workgroups can
// be read in isolation (one round of workgroups = the duration of one workgroup).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize [-DCHAIN_PD=8 -DCHAIN_ABL=0] -o build/chain_bench tools/chain_bench.hip
//   build/chain_bench [M=8064] [chain: 0 split | 1 mask] [bands: e.g. 9 or 9,10 or all]
#define CHAIN_PAIR_GEOMETRY 1      // the paired geometry of the widest band exists only in this probe (CHAIN_PAIR=1)
#include "../speechseparation_amd/csrc/mlp_chain.hip"
#include "../speechseparation_amd/csrc/split_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace bsrnn;
namespace bsrnn { int gemm_mode() { return GEMM_FP16X2; } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; }

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 8064;
    const int chain = argc > 2 ? atoi(argv[2]) : 0;
    const std::string sel = argc > 3 ? argv[3] : "all";
    const int widths[12] = {1, 2, 3, 6, 12, 24, 48, 96, 192, 384, 257, 0};
    const int K = 12, H = 64;
    std::vector<int> poff(K);
    int LDP = 0;
    for (int i = 0; i < K; ++i) { poff[i] = LDP; LDP += (2 * widths[i] + 7) & ~7; }
    std::vector<bool> on(K, sel == "all");
    if (sel != "all") { const char* p = sel.c_str(); while (*p) { on[atoi(p)] = true; while (*p && *p != ',') ++p; if (*p) ++p; } }

    struct Built { ChainDesc d; std::vector<uint16_t> w; std::vector<float> b; long cost; };
    std::vector<Built> built;
    unsigned seed = 1;
    double flop = 0;
    size_t wbytes_per_tile = 0;
    for (int i = 0; i < K; ++i) {
        const int a = 2 * widths[i];
        if (!on[i] || a == 0) continue;
        const int m = std::max(a, H), pz = std::max(a, 2 * H);
        const int dims_s[5][2] = {{a, a}, {a, a}, {m, a}, {H, m}, {H, H}}, dims_m[5][2] = {{2 * H, H}, {pz, 2 * H}, {a, pz}, {a, a}, {a, a}};
        Built bu;
        memset(&bu.d, 0, sizeof bu.d);
        ChainDesc& d = bu.d;
        d.p_off = poff[i]; d.a8 = (a + 7) & ~7; d.z_off = i * H;
        int units = 0, maxntl = 0, nbias = 0;
        bu.cost = 0;
        for (int l = 0; l < 5; ++l) {
            const int N = chain ? dims_m[l][0] : dims_s[l][0], Kd = chain ? dims_m[l][1] : dims_s[l][1];
            d.L[l].K16 = (Kd + 15) / 16; d.L[l].NTL = (N + 31) / 32; d.L[l].leaky = l < 4; d.L[l].bias_off = nbias; nbias += 32 * d.L[l].NTL;
            units = std::max(units, 2 * d.L[l].K16);
            if (l < 4) units = std::max(units, 4 * d.L[l].NTL);
            maxntl = std::max(maxntl, d.L[l].NTL);
            bu.cost += (long)d.L[l].K16 * d.L[l].NTL;
            flop += 2.0 * N * Kd * M;
        }
        const int img = 2 * units * 512;
        int RT = 1, GR = 1;
        if (8 * img <= CHAIN_LDS_EX && maxntl <= 4) { RT = 1; GR = 8; }
        else if (4 * img <= CHAIN_LDS_EX && maxntl <= 6) { RT = 1; GR = 4; }
        else if (4 * img <= CHAIN_LDS_EX && maxntl <= 12) { RT = 2; GR = 2; }
        else if (2 * img <= CHAIN_LDS_EX) { RT = 2; GR = 1; }
        if (getenv("CHAIN_GEOM")) sscanf(getenv("CHAIN_GEOM"), "%d,%d", &RT, &GR);      // override (must fit the LDS)
        const bool try48 = RT == 1 && GR == 1 && !getenv("BSRNN_CHAIN_NO48");
        bool try80 = RT == 2 && GR == 1 && !getenv("BSRNN_CHAIN_NO48") && !getenv("BSRNN_CHAIN_NO80");
        bool try64 = false;
        if (try80) {                                    // as api.hip: five row tiles of 16 where they fit and the tiles of 16 are 3 x 8 at most
            int u = 0, maxft = 0; bool whole = true;
            for (int l = 0; l < 5; ++l) {
                const int N = chain ? dims_m[l][0] : dims_s[l][0], Kd = chain ? dims_m[l][1] : dims_s[l][1];
                u = std::max(u, 4 * ((Kd + 31) / 32));
                if (l < 4) u = std::max(u, 2 * ((N + 15) / 16));
                maxft = std::max(maxft, (N + 15) / 16); whole = whole && N % 16 == 0;
            }
            const bool was80 = try80;
            try80 = 2 * u * 80 * 16 <= CHAIN_LDS_EX && maxft <= 24 && whole && maxft % 8 == 0;
            try64 = was80 && !try80 && !getenv("BSRNN_CHAIN_NO64") && 2 * u * 64 * 16 <= CHAIN_LDS_EX && maxft <= 40;      // as api.hip: four row tiles of 16
        }
        const bool pair = getenv("CHAIN_PAIR") && a == 768;        // two workgroups per 80 rows, half of K each (chain_body_pair)
        const bool g48 = try48 || try80 || try64 || pair;
        if (g48) {
            RT = pair ? 5 : (try48 ? 3 : (try64 ? 4 : 5)); GR = 1; units = 0; nbias = 0;
            for (int l = 0; l < 5; ++l) {
                const int N = chain ? dims_m[l][0] : dims_s[l][0], Kd = chain ? dims_m[l][1] : dims_s[l][1];
                d.L[l].K16 = (Kd + 31) / 32; d.L[l].NTL = (N + 15) / 16; d.L[l].bias_off = nbias; nbias += 16 * d.L[l].NTL;
                if (pair) d.L[l].K16 = Kd / 64;
                units = std::max(units, 4 * d.L[l].K16);
                if (l < 4) units = std::max(units, (pair ? 1 : 2) * d.L[l].NTL);
            }
            d.pair = pair; d.pair_base = 0;
        }
        d.RT = RT; d.NW = 8 / GR; d.plane_units = units; d.nbias = nbias;
        d.in_off = chain ? i * H : poff[i]; d.K0 = chain ? H : d.a8;
        if (!getenv("CHAIN_NORAG") && !g48 && RT == 2 && GR == 1 && 2 * img + CHAIN_RAG_LDS <= CHAIN_LDS_EX)
            for (int l = 0; l < 5; ++l) {
                const int N = chain ? dims_m[l][0] : dims_s[l][0];
                if (N % 32 >= 1 && N % 32 <= 4 && d.L[l].NTL >= 2) d.L[l].rag = 1;
            }
        bu.b.assign(nbias, 0.f);
        for (int l = 0; l < 5; ++l) {
            const int N = chain ? dims_m[l][0] : dims_s[l][0], Kd = chain ? dims_m[l][1] : dims_s[l][1];
            std::vector<float> w((size_t)N * Kd);
            for (auto& x : w) x = frand(seed) / sqrtf((float)Kd);
            d.L[l].w_off = (unsigned)(bu.w.size() * 2);
            if (pair) {
                pack_chain_layer16_pair_host(w.data(), N, Kd, Kd, 0, 2, bu.w);
                d.L[l].w_off1 = (unsigned)(bu.w.size() * 2);
                pack_chain_layer16_pair_host(w.data(), N, Kd, Kd, 1, 2, bu.w);
            } else if (g48) pack_chain_layer16_host(w.data(), N, Kd, Kd, 8, 2, bu.w);
            else pack_chain_layer_host(w.data(), N, Kd, Kd, d.NW, 2, bu.w, d.L[l].rag);
            for (int n = 0; n < N; ++n) bu.b[d.L[l].bias_off + n] = 0.1f * frand(seed);
        }
        wbytes_per_tile += pair ? bu.w.size() * 32 / 80 : g48 ? bu.w.size() * 2 * 32 / (16 * d.RT) : bu.w.size() * 2 / d.RT;      // per 32 rows: a wave group streams the weights once for its rows
        printf("band %2d a=%4d  NW=%d RT=%d  units=%3d  weights %.2f MB  cost %ld\n", i, a, d.NW, d.RT, units, bu.w.size() * 2 / 1e6, bu.cost);
        built.push_back(std::move(bu));
    }
    std::stable_sort(built.begin(), built.end(), [](const Built& x, const Built& y) {
        auto cls = [](const ChainDesc& d) { const int r = chain_rows(d); return r <= 48 ? 0 : (r <= 80 ? 1 : (r == 128 ? 2 : 3)); };
        const int cx = cls(x.d), cy = cls(y.d);
        return cx != cy ? cx < cy : x.cost > y.cost;
    });
    ChainLaunch g;
    memset(&g, 0, sizeof g);
    std::vector<ChainDesc> descs;
    for (auto& bu : built) {
        void *dw, *db;
        CK(hipMalloc(&dw, bu.w.size() * 2 + 64)); CK(hipMemcpy(dw, bu.w.data(), bu.w.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&db, bu.b.size() * 4)); CK(hipMemcpy(db, bu.b.data(), bu.b.size() * 4, hipMemcpyHostToDevice));
        bu.d.wstream = dw; bu.d.bias = (const float*)db;
        descs.push_back(bu.d);
    }
    ChainDesc* dd;
    CK(hipMalloc(&dd, descs.size() * sizeof(ChainDesc)));
    CK(hipMemcpy(dd, descs.data(), descs.size() * sizeof(ChainDesc), hipMemcpyHostToDevice));
    const int KH = K * H;
    float *X, *P, *Z, *Y;
    std::vector<float> hx((size_t)M * LDP), hz((size_t)M * KH);
    for (auto& x : hx) x = frand(seed);
    for (auto& x : hz) x = frand(seed);
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&P, hx.size() * 4)); CK(hipMemcpy(P, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&Y, hx.size() * 4));
    CK(hipMalloc(&Z, hz.size() * 4)); CK(hipMemcpy(Z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    // task table: heaviest band first, or (CHAIN_ORDER=mix) the 32-row workgroups interleaved with the others by work
    std::vector<int2> tasks, wide, rest;
    double ww = 0, wr = 0;
    for (size_t di = 0; di < descs.size(); ++di)
        for (int r0 = 0; r0 < M; r0 += chain_rows(descs[di])) {
            const bool w = chain_rows(descs[di]) <= 48 || descs[di].pair;
            (w ? wide : rest).push_back(make_int2((int)di, r0));
            if (descs[di].pair) wide.push_back(make_int2((int)di | 1 << 24, r0));       // the partner: adjacent in the table
            (w ? ww : wr) += (double)built[di].cost * descs[di].RT + 200.0;
        }
    if (getenv("CHAIN_ORDER") && !strcmp(getenv("CHAIN_ORDER"), "mix")) {
        size_t i = 0, j = 0; double aw = 0, ar = 0;
        while (i < wide.size() || j < rest.size()) {
            const bool tw = j >= rest.size() || (i < wide.size() && aw * wr <= ar * ww);
            const int2 t = tw ? wide[i++] : rest[j++];
            (tw ? aw : ar) += (double)built[t.x].cost * descs[t.x].RT + 200.0;
            tasks.push_back(t);
        }
    } else if (getenv("CHAIN_ORDER")) {
        // explicit dispatch order, e.g. CHAIN_ORDER=9,10:88,8,10,7,6 : band[:number of workgroups]; what is left follows in the default order
        std::vector<std::vector<int2>> per(K);
        auto band_of = [&](int di) { return descs[di].z_off / H; };
        for (const int2& t : wide) per[band_of(t.x)].push_back(t);
        for (const int2& t : rest) per[band_of(t.x)].push_back(t);
        std::vector<size_t> cur(K, 0);
        const char* p = getenv("CHAIN_ORDER");
        while (*p) {
            const int b = atoi(p);
            while (*p && *p != ',' && *p != ':') ++p;
            size_t n = per[b].size();
            if (*p == ':') { n = (size_t)atoi(++p); while (*p && *p != ',') ++p; }
            for (size_t k = 0; k < n && cur[b] < per[b].size(); ++k) tasks.push_back(per[b][cur[b]++]);
            if (*p) ++p;
        }
        for (const int2& t : wide) { const int b = band_of(t.x); if (cur[b] < per[b].size() && per[b][cur[b]].y <= t.y) { tasks.push_back(t); ++cur[b]; } }
        for (const int2& t : rest) { const int b = band_of(t.x); if (cur[b] < per[b].size() && per[b][cur[b]].y <= t.y) { tasks.push_back(t); ++cur[b]; } }
        if (tasks.size() != wide.size() + rest.size()) { printf("CHAIN_ORDER: %zu of %zu tasks placed\n", tasks.size(), wide.size() + rest.size()); return 1; }
    } else {
        tasks = wide; tasks.insert(tasks.end(), rest.begin(), rest.end());
    }
    if (!getenv("CHAIN_PAIR_ADJ")) {
        // partners 8 workgroup ids apart (the same XCD): groups of 8 pairs, lower halves then upper halves; a last group of fewer pairs is
        // filled up with other bands' workgroups (none left: the group's pairs sit on different XCDs and the launch says so)
        std::vector<int2> pt, other;
        for (const int2& t : tasks) (descs[t.x & 0xffffff].pair ? pt : other).push_back(t);
        tasks.clear();
        size_t o = 0;
        for (size_t i = 0; i < pt.size(); i += 16) {
            const size_t n = std::min<size_t>(16, pt.size() - i);
            for (size_t k = 0; k < n; k += 2) tasks.push_back(pt[i + k]);
            for (size_t k = n / 2; k < 8 && o < other.size(); ++k) tasks.push_back(other[o++]);
            for (size_t k = 1; k < n; k += 2) tasks.push_back(pt[i + k]);
        }
        tasks.insert(tasks.end(), other.begin() + o, other.end());
    }
    int2* dtasks;
    CK(hipMalloc(&dtasks, tasks.size() * sizeof(int2)));
    CK(hipMemcpy(dtasks, tasks.data(), tasks.size() * sizeof(int2), hipMemcpyHostToDevice));
    g.tasks = dtasks; g.n_tasks = (int)tasks.size();
    g.desc = dd; g.M = M;
    if (chain == 0) { g.Xin = X; g.ldx = LDP; g.P = P; g.ldp = LDP; g.Z = Z; g.ldz = KH; }
    else { g.Xin = Z; g.ldx = KH; g.P = P; g.ldp = LDP; g.Xmul = X; g.ldm = LDP; g.Y = Y; g.ldy = LDP; }
    {
        const int pairs = (M + PAIR_ROWS - 1) / PAIR_ROWS;
        CK(hipMalloc(&g.exch, pairs * PAIR_EXCH_FLOATS * sizeof(float)));
        CK(hipMalloc(&g.pflags, pairs * 16 * sizeof(int))); CK(hipMemset(g.pflags, 0, pairs * 16 * sizeof(int)));
        g.pair_epoch = 0; g.pair_spin = 20000000;
        int* rf; CK(hipMalloc(&rf, 4)); CK(hipMemset(rf, 0, 4)); g.range_flag = rf;
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) { ++g.pair_epoch; launch_mlp_chain(g, chain, s); }
    CK(hipStreamSynchronize(s));
    const int reps = 50;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) { ++g.pair_epoch; launch_mlp_chain(g, chain, s); }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    if (CHAIN_TRACE) {
        // phases of the waves of block 0 and of one late block: stamps are 100 MHz ticks
        unsigned long long* dbg;
        CK(hipMalloc(&dbg, 64 * 8 * 24 * 8)); CK(hipMemset(dbg, 0, 64 * 8 * 24 * 8));
        g.dbg = dbg;
        launch_mlp_chain(g, chain, s);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(64 * 8 * 24);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        const char* names[18] = {"stage-in", "barrier", "L0 tiles", "L0 wait", "L0 write+bar", "L1 tiles", "L1 wait", "L1 write+bar", "L2 tiles", "L2 wait",
                                 "L2 write+bar", "L3 tiles", "L3 wait", "L3 write+bar", "L4 tiles", "", "", ""};
        for (int blk : {0, 37}) {
            printf("block %d: microseconds per phase, waves 0..7\n", blk);
            for (int ph = 0; ph < 15; ++ph) {
                printf("  %-13s", names[ph]);
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* d = &h[((size_t)blk * 8 + w) * 24];
                    printf(" %6.2f", d[ph + 1] >= d[ph] ? (d[ph + 1] - d[ph]) * 0.01 : -1.0);
                }
                printf("\n");
            }
            const unsigned long long* d0 = &h[(size_t)blk * 8 * 24];
            printf("  total (wave 0) %.2f us\n", (d0[15] - d0[0]) * 0.01);
        }
    }
    {
        // checksum of the outputs of the last launch and the guard word (compare CHAIN_PAIR=1 with the unpaired geometry: same sums up to rounding)
        int rfh = 0; CK(hipMemcpy(&rfh, g.range_flag, 4, hipMemcpyDeviceToHost));
        std::vector<float> ho(chain ? hx.size() : hz.size());
        CK(hipMemcpy(ho.data(), chain ? Y : Z, ho.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0, sa = 0; for (float v : ho) { sum += v; sa += fabs(v); }
        printf("guard %d  output sum %.9g  sum|.| %.9g\n", rfh, sum, sa);
        if (getenv("CHAIN_DUMP")) { FILE* f = fopen(getenv("CHAIN_DUMP"), "wb"); fwrite(ho.data(), 4, ho.size(), f); fclose(f); }
    }
    printf("chain %d  M %d  bands %s  PD %d ABL %d:  %d blocks  %.1f us  %.1f TFLOP/s-equivalent  weight stream %.2f GB -> %.2f TB/s\n", chain, M, sel.c_str(),
           CHAIN_PD, CHAIN_ABL, g.n_tasks, us, flop / us / 1e6, wbytes_per_tile * ((M + 31) / 32) / 1e9, wbytes_per_tile * ((M + 31) / 32) / us / 1e6);
    return 0;
}
