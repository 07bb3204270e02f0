// Measurement-only probe: the fp16x2 grouped GEMM with BOTH operands pre-split in slab format and staged by LDS-DMA
// (global_load_lds_dwordx4) instead of global_load -> VALU split -> ds_write, against the product kernel on the
// PRE0-shaped job set.  Same tile geometry, job / tile tables and XCD mapping as gemm_h2_kernel.
//   hipcc -O3 --offload-arch=gfx950 -o build/gemm_glds_bench tools/gemm_glds_bench.hip
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

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// NST = LDS stages (2: one slab in flight, 3: two).  ABL: 1 = no DMA after the prologue, 2 = no MFMA.
// TR = 1: MFMA operands swapped, so the accumulators hold the transposed 32x32 tiles - a lane owns 4 x 4 consecutive
// output columns of one row and stores them with 16-byte global stores straight from registers (no LDS staging, no barriers).
template <int EPI, int NST, int ABL = 0, int BMT = 128, int TR = 0>
__global__ __launch_bounds__(2 * BMT, (BMT == 128 ? 2 : 1)) void gemm_glds_kernel(GemmLaunch g, const _Float16* __restrict__ Xs, int ldxs, const int* __restrict__ xs_off)
{
    typedef _Float16 hT;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int NT = 2, BN = 128, BM = BMT, NTH = 2 * BMT, NWV = NTH / 64;
    constexpr int PLANE = (BM + BN) * 32;           // halves per piece and stage
    constexpr int STAGE = 2 * PLANE;                // 32 KB (48 KB for the 256-row tile)
    __shared__ __attribute__((aligned(16))) hT smemh[NST * STAGE];

    const int m_tiles = (g.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, lidx = blockIdx.x >> 3;
    const int mchunk = g.mchunk;
    const int per_chunk = mchunk * g.n_tiles;
    const int chunk = (lidx / per_chunk) * 8 + xcd;
    const int rem = lidx % per_chunk;
    const int m_tile = chunk * mchunk + rem % mchunk;
    if (m_tile >= m_tiles) return;
    const int2 tj = g.tiles[rem / mchunk];
    const GemmJob job = g.jobs[tj.x];
    const int n0 = tj.y * BN, m0 = m_tile * BM;
    const int N = job.N, K = job.K, M = g.M;
    const int nk = (K + 31) >> 5;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;
    const int wcol = 64 * wn;
    float bias[NT];
    bool live[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nc = n0 + wcol + 32 * j + r32;
        bias[j] = ((gcf)job.bias)[nc < N ? nc : N - 1];
        live[j] = (n0 + wcol + 32 * j) < N;
    }

    // DMA plan of this wave: A rows 32 wave .. +31 and B rows 32 wave .. +31, two groups of 16 rows each, both pieces.
    // One instruction moves 16 rows x 64 bytes of one piece; the LDS image is lane-linear, so the 16-byte unit swizzle of
    // the fragment reads is applied to the SOURCE address (unit ^ ((row >> 2) & 3)).
    const int lrow = lane >> 2, slot = lane & 3;
    constexpr int BG = BN / 16 / NWV;                // 16-row groups of B per wave: 2 (4 waves) or 1 (8 waves)
    unsigned srcA[2], srcB[BG];                      // byte offsets of piece 0, slab 0
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = 32 * wave + 16 * s + lrow;
        const int unit = slot ^ ((r >> 2) & 3);
        int ra = m0 + r; ra = ra < M ? ra : M - 1;
        srcA[s] = ((unsigned)ra * (unsigned)ldxs + (unsigned)xs_off[tj.x] + 8u * unit) * 2u;
    }
#pragma unroll
    for (int s = 0; s < BG; ++s) {
        const int r = 16 * BG * wave + 16 * s + lrow;
        const int unit = slot ^ ((r >> 2) & 3);
        int rb = n0 + r; rb = rb < N ? rb : N - 1;
        srcB[s] = ((unsigned)rb * (unsigned)job.wrow + 8u * unit) * 2u;
    }
    const char __attribute__((address_space(1)))* const Ab = (const char __attribute__((address_space(1)))*)Xs;
    const char __attribute__((address_space(1)))* const Bb = (const char __attribute__((address_space(1)))*)job.Wp;
    auto dma = [&](int ks, hT* st) {
        const unsigned kb = (unsigned)ks * 128u;     // one slab = 64 halves per row
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_global_load_lds((glb_vp)(Ab + (srcA[s] + kb + 64u * p)), (lds_vp)(st + p * PLANE + (32 * wave + 16 * s) * 32), 16, 0, 0);
#pragma unroll
            for (int s = 0; s < BG; ++s)
                __builtin_amdgcn_global_load_lds((glb_vp)(Bb + (srcB[s] + kb + 64u * p)), (lds_vp)(st + p * PLANE + (BM + 16 * BG * wave + 16 * s) * 32), 16, 0, 0);
        }
    };
    constexpr int NDMA = 2 * (2 + BG);               // transfers per wave and slab: 8 or 6
    constexpr int WAIT_NEWEST = 0x0F70 | NDMA;       // s_waitcnt vmcnt(NDMA): everything but the newest slab has landed

    v16f acc[2][NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j][0] = (v16f){0}; acc[i][j][1] = (v16f){0}; }
    const int swz = (r32 >> 2) & 3;
    const int fa = (64 * wm + r32) * 32, fb = (BM + wcol + r32) * 32;
    const int fu[2] = {((0 + half) ^ swz) * 8, ((2 + half) ^ swz) * 8};

    // all 16 fragment reads of a slab are issued up front (64 registers): the second half's reads land behind the
    // first half's MFMAs, so one LDS latency per slab is exposed instead of one per read group
    auto compute = [&](auto fast_tag, const hT* cur) {
        constexpr bool FAST = decltype(fast_tag)::value;
        h8 b[2][NT][2], a[2][2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int j = 0; j < NT; ++j) b[ks][j][pl] = *reinterpret_cast<const h8*>(&cur[pl * PLANE + fb + 32 * j * 32 + fu[ks]]);
#pragma unroll
                for (int i = 0; i < 2; ++i) a[ks][i][pl] = *reinterpret_cast<const h8*>(&cur[pl * PLANE + fa + 32 * i * 32 + fu[ks]]);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ABL & 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { acc[i][0][0][ks] += (float)a[ks][i][0][0] + (float)a[ks][i][1][1]; acc[i][NT - 1][1][ks] += (float)b[ks][NT - 1][0][0] + (float)b[ks][NT - 1][1][1]; }
                continue;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!FAST && !live[j]) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (TR) {
                        acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][j][0], a[ks][i][1], acc[i][j][1], 0, 0, 0);
                        acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][j][1], a[ks][i][0], acc[i][j][1], 0, 0, 0);
                        acc[i][j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][j][0], a[ks][i][0], acc[i][j][0], 0, 0, 0);
                        continue;
                    }
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][1], b[ks][j][0], acc[i][j][1], 0, 0, 0);
                    acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][0], b[ks][j][1], acc[i][j][1], 0, 0, 0);
                    acc[i][j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i][0], b[ks][j][0], acc[i][j][0], 0, 0, 0);
                }
            }
        }
        if (FAST && !(ABL & 2)) {
            // first half's 8 reads, then the second half's reads one behind each of the first 8 MFMAs, then the rest
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    };
    const bool fast = live[NT - 1];

    if (NST == 2) {
        if (nk > 0) dma(0, smemh);
        __syncthreads();                            // waits vmcnt(0): slab 0 landed for every wave
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk && !(ABL & 1)) dma(ks + 1, smemh + ((ks + 1) & 1) * STAGE);
            if (fast) compute(std::true_type(), smemh + (ks & 1) * STAGE); else if (live[0]) compute(std::false_type(), smemh + (ks & 1) * STAGE);
            __syncthreads();
        }
    } else {
        // three stages: slab ks + 2 is requested while slab ks is multiplied; the wait before the barrier leaves the
        // newest slab's 8 transfers of this wave in flight
        if (nk > 0) dma(0, smemh);
        if (nk > 1) dma(1, smemh + STAGE);
        if (nk > 1) __builtin_amdgcn_s_waitcnt(WAIT_NEWEST); else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        int st = 0;
        for (int ks = 0; ks < nk; ++ks) {
            const int st2 = st >= 1 ? st - 1 : 2;     // (ks + 2) % 3
            if (ks + 2 < nk && !(ABL & 1)) dma(ks + 2, smemh + st2 * STAGE);
            if (fast) compute(std::true_type(), smemh + st * STAGE); else if (live[0]) compute(std::false_type(), smemh + st * STAGE);
            if (ks + 2 < nk && !(ABL & 1)) __builtin_amdgcn_s_waitcnt(WAIT_NEWEST); else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            st = st == 2 ? 0 : st + 1;
        }
    }

    typedef v4f __attribute__((address_space(1)))* g4;
    if (TR) {
        // lane (row = lane & 31 of the 32-row block, half) holds columns 8 q + 4 half + 0..3 of the 32-column block in regs 4q..4q+3
        typedef const v4f __attribute__((address_space(1)))* gc4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!live[j]) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wcol + 32 * j + 8 * q + 4 * half;
                if (n >= ((N + 7) & ~7)) continue;
                const v4f bv = *(gc4)((gcf)job.bias + (n + 3 < N ? n : 0));     // (probe: widths are multiples of 8)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + 64 * wm + 32 * i + r32;
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][0][4 * q + e] + (1.f / 2048.f) * acc[i][j][1][4 * q + e] + bv[e];
                        if (EPI == EPI_LEAKY) x = x >= 0.f ? x : 0.01f * x;
                        v[e] = n + e < N ? x : 0.f;
                    }
                    if (m < M) *(g4)((gf)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = v;
                }
            }
        }
        return;
    }
    if (TR == 2) {
        // natural accumulator layout, 4-byte stores straight from registers: a wave instruction writes two rows x 32
        // consecutive columns = two full 128-byte lines; no LDS staging, no barriers
        typedef float __attribute__((address_space(1)))* g1;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!live[j]) continue;
            const int n = n0 + wcol + 32 * j + r32;
            if (n >= ((N + 7) & ~7)) continue;
            const bool in = n < N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0 + 64 * wm + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    float v = acc[i][j][0][reg] + (1.f / 2048.f) * acc[i][j][1][reg] + bias[j];
                    if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                    if (m < M) *((g1)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = in ? v : 0.f;
                }
        }
        return;
    }
    // epilogue through LDS in passes of 128 rows, fp32 output
    float* const sE = reinterpret_cast<float*>(smemh);
    constexpr int ES = BN, UPR4 = BN / 4, NU = 128 * UPR4 / NTH;
#pragma unroll
    for (int pass = 0; pass < BM / 128; ++pass) {
        __syncthreads();
        if ((wm >> 1) == pass) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!live[j]) continue;
                const int col = wcol + 32 * j + r32;
                const bool in = n0 + col < N;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = acc[i][j][0][reg] + (1.f / 2048.f) * acc[i][j][1][reg] + bias[j];
                        if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                        if (!in) v = 0.f;
                        sE[(64 * (wm & 1) + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half) * ES + col] = v;
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + NTH * u;
            const int row = idx / UPR4, c4 = idx % UPR4;
            const int m = m0 + 128 * pass + row, n = n0 + 4 * c4;
            if (m < M && n < ((N + 7) & ~7)) *(g4)((gf)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = *reinterpret_cast<const v4f*>(&sE[row * ES + 4 * c4]);
        }
    }
}

template <int NST, int ABL = 0, int BMT = 128, int TR = 0>
static void launch_glds(const GemmLaunch& g_in, const _Float16* Xs, int ldxs, const int* xs_off, hipStream_t stream)
{
    GemmLaunch g = g_in;
    const int m_tiles = (g.M + BMT - 1) / BMT;
    g.mchunk = BMT == 128 ? gemm_mchunk(m_tiles) : 1;
    const int chunks = (m_tiles + g.mchunk - 1) / g.mchunk;
    dim3 grid(8 * ((chunks + 7) / 8) * g.mchunk * g.n_tiles), block(2 * BMT);
    hipLaunchKernelGGL((gemm_glds_kernel<EPI_LEAKY, NST, ABL, BMT, TR>), grid, block, 0, stream, g, Xs, ldxs, xs_off);
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 8064;
    int widths[11] = {4, 4, 4, 8, 12, 24, 48, 96, 192, 384, 260};
    int nb = 11;
    if (argc > 2) { nb = 1; widths[0] = atoi(argv[2]); }
    const int LD = 2080;
    std::vector<GemmJob> jobs;
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
        wtot = (wtot + 7) & ~size_t(7);
        jobs.push_back(j);
    }
    std::vector<int> order(nb);
    for (int i = 0; i < nb; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return jobs[a].K > jobs[b].K; });
    std::vector<int2> tiles128;
    for (int i : order)
        for (int t = 0; t < (jobs[i].N + 127) / 128; ++t) tiles128.push_back(make_int2(i, t));
    std::vector<float> h(wtot);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    const size_t xn = (size_t)M * LD;
    std::vector<float> hx(xn);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    // slab-format weights and activations
    std::vector<size_t> qoff;
    std::vector<int> xsoff;
    size_t qtot = 0;
    int ldxs = 0;
    for (int i = 0; i < nb; ++i) {
        const int K32 = (jobs[i].K + 31) & ~31;
        jobs[i].wrow = h2_row_stride(K32);
        qoff.push_back(qtot); qtot += (size_t)jobs[i].N * jobs[i].wrow;
        xsoff.push_back(ldxs); ldxs += 2 * K32;
    }
    std::vector<uint16_t> hq(qtot + 8), hxs((size_t)M * ldxs);
    for (int i = 0; i < nb; ++i) {
        const int K32 = (jobs[i].K + 31) & ~31;
        pack_h2_slabs_host(&h[woff[i]], jobs[i].N, jobs[i].K, jobs[i].K, K32, jobs[i].wrow, &hq[qoff[i]]);
        pack_h2_slabs_host(&hx[jobs[i].x_off], M, jobs[i].K, LD, K32, ldxs, &hxs[xsoff[i]]);
    }
    float *dW, *dX, *dY, *dY2;
    uint16_t *dWq, *dXs;
    int* dXoff;
    CK(hipMalloc(&dW, wtot * 4)); CK(hipMalloc(&dX, xn * 4)); CK(hipMalloc(&dY, xn * 4)); CK(hipMalloc(&dY2, xn * 4));
    CK(hipMalloc(&dWq, hq.size() * 2)); CK(hipMalloc(&dXs, hxs.size() * 2)); CK(hipMalloc(&dXoff, nb * 4));
    CK(hipMemset(dY, 0, xn * 4)); CK(hipMemset(dY2, 0, xn * 4));
    CK(hipMemcpy(dW, h.data(), wtot * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hx.data(), xn * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dXs, hxs.data(), hxs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dXoff, xsoff.data(), nb * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < nb; ++i) {
        jobs[i].W = dW + woff[i]; jobs[i].bias = dW + woff[i] + (size_t)jobs[i].N * jobs[i].K;
        jobs[i].Wp = dWq + qoff[i];
    }
    GemmJob* dJ; int2* dT;
    CK(hipMalloc(&dJ, jobs.size() * sizeof(GemmJob))); CK(hipMalloc(&dT, tiles128.size() * sizeof(int2)));
    CK(hipMemcpy(dJ, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, tiles128.data(), tiles128.size() * sizeof(int2), hipMemcpyHostToDevice));
    GemmLaunch g = {};
    g.jobs = dJ; g.tiles = dT; g.n_tiles = (int)tiles128.size(); g.tile_n = 128;
    g.X = dX; g.ldx = LD; g.Y = dY; g.ldy = LD; g.M = M; g.epilogue = EPI_LEAKY;
    GemmLaunch g2 = g; g2.Y = dY2;
    double flop = 0;
    for (auto& j : jobs) flop += 2.0 * j.N * j.K * M;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    constexpr int NV = 12;
    const char* names[NV] = {"product (A split on the fly, register staging)", "LDS-DMA, 2 stages", "LDS-DMA, 3 stages", "LDS-DMA 2 stages, no DMA after prologue",
                             "LDS-DMA 2 stages, no MFMA", "LDS-DMA 3 stages, no MFMA", "LDS-DMA 256x128 tile, 8 waves, 2 stages", "LDS-DMA 256x128 tile, 8 waves, 3 stages",
                             "LDS-DMA 256x128 3 stages, no DMA after prologue", "LDS-DMA 256x128 3 stages, no MFMA", "LDS-DMA 2 stages, transposed acc, direct stores", "LDS-DMA 2 stages, direct 4-byte stores"};
    std::vector<float> t[NV];
    for (int rep = 0; rep < 14; ++rep)
        for (int v = 0; v < NV; ++v) {
            CK(hipEventRecord(a, s));
            switch (v) {
            case 0: launch_gemm_h2<2>(g, s); break;
            case 1: launch_glds<2>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 2: launch_glds<3>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 3: launch_glds<2, 1>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 4: launch_glds<2, 2>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 5: launch_glds<3, 2>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 6: launch_glds<2, 0, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 7: launch_glds<3, 0, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 8: launch_glds<3, 1, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 9: launch_glds<3, 2, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            case 10: launch_glds<2, 0, 128, 1>(g2, (const _Float16*)dXs, ldxs, dXoff, s); break;
            default: launch_glds<2, 0, 128, 2>(g2, (const _Float16*)dXs, ldxs, dXoff, s); }
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2) t[v].push_back(ms);
        }
    printf("M=%d tiles=%d GFLOP=%.2f\n", M, g.n_tiles, flop / 1e9);
    for (int v = 0; v < NV; ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-48s median %.1f us  min %.1f us  -> %.1f TFLOP/s-equivalent\n", names[v], med * 1e3, t[v][0] * 1e3, flop / (med * 1e-3) / 1e12);
    }
    launch_gemm_h2<2>(g, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> y0(xn), y1(xn);
    CK(hipMemcpy(y0.data(), dY, xn * 4, hipMemcpyDeviceToHost));
    for (int v : {1, 2, 6, 7, 10, 11}) {
        CK(hipMemset(dY2, 0, xn * 4));
        if (v == 1) launch_glds<2>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        else if (v == 2) launch_glds<3>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        else if (v == 6) launch_glds<2, 0, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        else if (v == 10) launch_glds<2, 0, 128, 1>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        else if (v == 11) launch_glds<2, 0, 128, 2>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        else launch_glds<3, 0, 256>(g2, (const _Float16*)dXs, ldxs, dXoff, s);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(y1.data(), dY2, xn * 4, hipMemcpyDeviceToHost));
        double d = 0, mx = 0;
        for (size_t i = 0; i < xn; ++i) { d = std::max(d, fabs((double)y0[i] - y1[i])); mx = std::max(mx, (double)fabsf(y0[i])); }
        printf("%-48s max|diff| vs product %.3e (max|y| %.3f)\n", names[v], d, mx);
    }
    return 0;
}
