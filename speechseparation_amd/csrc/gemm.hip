// Grouped fp32 linear layers on the gfx950 matrix cores.
//
// Replaces the ~110 nn.Linear calls of one BSRNN.forward (bsrnn.py:404-412, :422-425, and the
// fc of NormRNNResidual :84) with one launch per "layer slot": a table of (band job, column
// tile) x 128-row tiles of the M = C*T frame rows, walked in an XCD-aware order.
//
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain (no xf32 on
// gfx950), which is what the 1e-4 parity budget against the fp32 reference needs; bf16 MFMA
// would be 16x faster and ~1e-2 wrong.  Roofline for this kernel is therefore the fp32
// matrix peak (157.3 TFLOP/s), see DESIGN.md.
//
// Tile: 128 x (64*NT) x BK per 256-thread workgroup, four waves as 2(M) x 2(N), each wave a
// 64 x (32*NT) patch = 2*NT accumulators of 32x32.  NT = 2 (128-wide) is used for the slots whose
// layers are wide: twice the MFMA work per barrier pair and per LDS byte (measured: the NT = 1
// structure tops out at ~66 % of peak even with global loads removed).  NT = 1 serves the
// 64-column layers.  Operands are K-contiguous in memory for both X [M][ldx] and W [N][K], so A
// and B fragments are read with the same pattern: lane l owns row (l & 31) and the half (l >> 5)
// of the BK-deep K slab, fetched as BK/8 ds_read_b128; MFMA step (j, e) consumes element e of the
// j-th read, i.e. k = (BK/2)*(l>>5) + 4j + e on BOTH operands (the sum over k is order-agnostic as
// long as A and B agree).  LDS rows are padded to BK+4 floats: (BK+4)/4 is odd, so the 16 rows of
// a ds_read_b128 lane group hit 16 distinct 16-byte slots.
#include "kernels.h"

#include <cstdlib>

namespace bsrnn {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef GEMM_BK
#define GEMM_BK 32
#endif
constexpr int BM = 128, BK = GEMM_BK, LDS_STRIDE = BK + 4, MCHUNK_MAX = 8;
// m-tiles per XCD-pinned chunk: 8 when there are enough m-tiles, fewer for small batches so that all
// eight XCDs still get work (with 32 m-tiles, chunks of 8 would leave half of the chip idle)
static inline int gemm_mchunk(int m_tiles) { const int c = (m_tiles + 7) / 8; return c < 1 ? 1 : (c > MCHUNK_MAX ? MCHUNK_MAX : c); }
static_assert(BK == 32 || BK == 64, "BK");
#ifndef GEMM_OCC64
#define GEMM_OCC64 3          // waves per SIMD (= workgroups per CU) requested for the 64-wide kernel
#endif

// explicit global address space: pointers that are loaded from the job table would otherwise be
// generic and compile to flat_load, which also counts in lgkmcnt and so serialises the global
// prefetch behind the LDS-read wait in front of the MFMAs.
typedef const float __attribute__((address_space(1)))* gcf;
typedef float __attribute__((address_space(1)))* gf;
typedef const v2f __attribute__((address_space(1)))* gcf2;

// ABL is a measurement-only switch (tools/gemm_bench.hip): 0 = the product kernel, 1 = no global
// loads after the first K slab, 2 = no MFMAs, 3 = neither, +4 = no barriers / LDS writes after the
// first slab.  PRIO: 0 none, 1 = s_setprio(1) around the MFMA cluster, 2 = static per-workgroup priority.
template <int EPI, int NT, int ABL = 0, int PRIO = 0, int VEC = 2>
__global__ __launch_bounds__(256, (NT == 1 ? GEMM_OCC64 : 2)) void gemm_f32_kernel(GemmLaunch g)
{
    constexpr int BN = 64 * NT;
    typedef float vNf __attribute__((ext_vector_type(VEC)));
    typedef const vNf __attribute__((address_space(1)))* gcfN;
    constexpr int UPR = BK / VEC;              // staging units (VEC floats) per row
    constexpr int RPI = 256 / UPR;             // rows covered by one staging pass of the 256 threads
    constexpr int NA = BM / RPI, NB = BN / RPI; // float2 staging units per thread for the A / B tile
    // one LDS array: A tile, B tile; after the K loop the same memory stages the accumulators for the epilogue
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_STRIDE];
    float* const sA = smem;
    float* const sB = smem + BM * LDS_STRIDE;

    // XCD-aware work mapping (speed only, never correctness).  Workgroups are dealt round-robin
    // over the 8 XCDs, each with a private 4 MiB L2.  m-tiles are grouped in chunks of g.mchunk (<= 8);
    // chunk c is pinned to the blocks with blockIdx % 8 == c % 8, and inside a chunk the column
    // tiles are the OUTER loop (heaviest K first, table order) and the m-tiles the inner one, so
    // the workgroups resident on one XCD share both their X row-tiles and their W column-tiles
    // through that XCD's L2 instead of refetching them from the Infinity Cache per XCD
    // (measured: L2 hit rate 51 % -> 85 %, fetch 255 -> 59 MB per launch).
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
    const int n0 = tj.y * BN;
    const int m0 = m_tile * BM;
    const int N = job.N, K = job.K, M = g.M;
    const gcf X = (gcf)(g.X + job.x_off);
    const gcf W = (gcf)job.W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;
    unsigned long long t_begin = 0, c_begin = 0;
    if (ABL & 16) { t_begin = __builtin_amdgcn_s_memrealtime(); c_begin = __builtin_amdgcn_s_memtime(); }

    // columns of this lane (one per 32-wide sub-tile); biases are fetched before the main loop
    const int ncol0 = n0 + 32 * NT * wn + r32;
    float bias[NT];
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
        const int n = ncol0 + 32 * jn;
        bias[jn] = ((gcf)job.bias)[n < N ? n : N - 1];
    }
    // a wave whose whole column range lies beyond N (narrow jobs in a 128-wide launch) only helps
    // with staging; wave-uniform
    const bool wave_live = (n0 + 32 * NT * wn) < N;

    // staging map: float2 units, 16 per 32-float row; row offsets are loop invariant
    const int s_row = tid / UPR;          // + RPI*i
    const int s_k = (tid % UPR) * VEC;
    // (32-bit element offsets from the wave-uniform bases: half the registers of 64-bit pointers,
    // and the loads take the scalar-base + vector-offset form; all buffers are far below 2^31 floats)
    unsigned oa[NA], ob[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int row = m0 + s_row + RPI * i;
        row = row < M ? row : M - 1;
        oa[i] = (unsigned)row * (unsigned)g.ldx + s_k;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int row = n0 + s_row + RPI * i;
        row = row < N ? row : N - 1;
        ob[i] = (unsigned)row * (unsigned)K + s_k;
    }

    v16f acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = (v16f){0};
    vNf ra[NA], rb[NB];
    v4f fa_keep[2] = {}, fb_keep[NT] = {};   // used by the ABL & 8 measurement variant only

    // K is even, so a float2 is either fully inside or fully outside [0, K)
    auto gload = [&](int k0) {
        if (k0 + BK <= K) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *(gcfN)(X + (oa[i] + k0));
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *(gcfN)(W + (ob[i] + k0));
        } else {
            const bool kin = k0 + s_k < K;
            const int kk = kin ? k0 : -s_k;     // any valid address; the value is zeroed below
            const float zm = kin ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *(gcfN)(X + (oa[i] + kk)) * zm;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *(gcfN)(W + (ob[i] + kk)) * zm;
        }
    };

    if (PRIO == 2) {
        const int pr = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 8) % 3));
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        if (pr == 2) __builtin_amdgcn_s_setprio(2);
    }
    if (K > 0) gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (!(ABL & 4) || k0 == 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *reinterpret_cast<vNf*>(&sA[(s_row + RPI * i) * LDS_STRIDE + s_k]) = ra[i];
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<vNf*>(&sB[(s_row + RPI * i) * LDS_STRIDE + s_k]) = rb[i];
            __syncthreads();
        }
        if (k0 + BK < K && !(ABL & 1)) gload(k0 + BK);
        if (!wave_live) continue;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);

        const float* pa0 = &sA[(64 * wm + r32) * LDS_STRIDE + (BK / 2) * half];
        const float* pb0 = &sB[(32 * NT * wn + r32) * LDS_STRIDE + (BK / 2) * half];
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            v4f a[2], b[NT];
            if ((ABL & 8) && k0 > 0) {      // measurement only: no LDS reads after the first slab
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[i] = fa_keep[i]; asm volatile("" : "+v"(a[i])); }
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) { b[jn] = fb_keep[jn]; asm volatile("" : "+v"(b[jn])); }
            } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const v4f*>(pa0 + 32 * i * LDS_STRIDE + 4 * j);
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) b[jn] = *reinterpret_cast<const v4f*>(pb0 + 32 * jn * LDS_STRIDE + 4 * j);
            if (ABL & 8) { fa_keep[0] = a[0]; fa_keep[1] = a[1]; for (int jn = 0; jn < NT; ++jn) fb_keep[jn] = b[jn]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) {
                        if (ABL & 2)
                            asm volatile("" ::"v"(a[i][e]), "v"(b[jn][e]));
                        else
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[jn][e], acc[i][jn], 0, 0, 0);
                    }
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    }

    if ((ABL & 16) && lane == 0 && g.tap) {     // measurement only: where and when did every wave of this workgroup run
        unsigned long long* d = reinterpret_cast<unsigned long long*>(g.tap) + 4 * ((size_t)blockIdx.x * 4 + wave);
        d[0] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20);      // HW_REG_XCC_ID
        d[1] = (unsigned long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4) | ((__builtin_amdgcn_s_memtime() - c_begin) << 20);   // HW_REG_HW_ID | shader cycles
        d[2] = t_begin;
        d[3] = __builtin_amdgcn_s_memrealtime();
    }
    // Epilogue through LDS.  The C/D layout of the 32x32 MFMA (col = lane & 31, row = (reg & 3) + 8*(reg >> 2)
    // + 4*(lane >> 5)) gives every lane a column, i.e. 4-byte accesses at a row stride; the residual /
    // multiplier loads and the stores are therefore done from a row-major LDS image of the tile with 16 bytes
    // per lane and whole 256-byte row segments per wave (the EPI_MASK launch spent 70 us = 43 % extra in its
    // 4-byte epilogue).  Bias and LeakyReLU are applied on the way into LDS.  Two passes of 64 rows.
    constexpr int ES = BN + 4;                          // row stride of the staged tile (floats)
    static_assert(64 * ES <= (BM + BN) * LDS_STRIDE, "staging tile must fit the operand tiles");
    float* const sE = smem;
    const int my_col = 32 * NT * wn + r32;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        __syncthreads();                                // operand tiles (pass 0) / previous pass fully consumed
        if (wm == hh && wave_live) {
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = acc[i][jn][reg] + bias[jn];
                        if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                        sE[(32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half) * ES + my_col + 32 * jn] = v;
                    }
        }
        __syncthreads();
        constexpr int CPR = BN / 4;                     // float4 chunks per row
#pragma unroll
        for (int u = 0; u < 64 * CPR / 256; ++u) {
            const int idx = tid + 256 * u;
            const int row = idx / CPR, c4 = idx % CPR;
            const int m = m0 + 64 * hh + row, n = n0 + 4 * c4;
            // a chunk that straddles N stays inside the band's 4-aligned segment: its pad columns only need to be finite
            if (m < M && n < N) {
                v4f v = *reinterpret_cast<const v4f*>(&sE[row * ES + 4 * c4]);
                if (EPI == EPI_RES || EPI == EPI_MASK)
                    v += *reinterpret_cast<const v4f __attribute__((address_space(1)))*>((gcf)(g.R + job.r_off + n) + (size_t)m * g.ldr);
                if (EPI == EPI_MASK) {
                    if (g.tap) *reinterpret_cast<v4f __attribute__((address_space(1)))*>((gf)(g.tap + job.m_off + n) + (size_t)m * g.ldt) = v;
                    v *= *reinterpret_cast<const v4f __attribute__((address_space(1)))*>((gcf)(g.Mul + job.m_off + n) + (size_t)m * g.ldm);
                }
                *reinterpret_cast<v4f __attribute__((address_space(1)))*>((gf)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = v;
            }
        }
    }
}

// =====================================================================================
// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3"): every fp32 operand is split exactly into three
// bf16 pieces a = a1 + a2 + a3 (8 significant bits each, residuals are exact in fp32), and a.b is
// evaluated as the six bf16 MFMA terms of weight >= 2^-16,
//     a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2),
// with fp32 accumulation (products of bf16 pairs are exact in fp32).  The dropped terms are <= 2^-23
// relative, i.e. the result is as accurate as an fp32 fma chain, at 6/16 of the fp32-MFMA pipe time
// (v_mfma_f32_32x32x16_bf16: 16x the MACs per cycle of v_mfma_f32_32x32x2_f32).  The large term has its own
// accumulator so the small corrections are summed among themselves before they meet it.
// Same tiling (128 x 64 x 32, 4 waves as 2 x 2), XCD mapping, and LDS-staged epilogue as the fp32 kernel;
// operands are split by the staging threads on their way into LDS (three bf16 planes, rows padded to
// 80 bytes: conflict-free ds_read_b128 fragments).
// =====================================================================================
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const v4f a, bf16x4& p1, bf16x4& p2, bf16x4& p3)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p1[i] = (__bf16)a[i];
        const float r1 = a[i] - (float)p1[i];
        p2[i] = (__bf16)r1;
        const float r2 = r1 - (float)p2[i];
        p3[i] = (__bf16)r2;
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmLaunch g)
{
    constexpr int BN = 64;
    constexpr int PS = 40;                          // plane row stride in bf16 (80 bytes: 32 data + 8 pad)
    constexpr int PLANE = (BM + BN) * PS;           // one plane: A rows then B rows
    typedef const v4f __attribute__((address_space(1)))* gcf4;
    __shared__ __attribute__((aligned(16))) __bf16 smemh[3 * PLANE];
    static_assert(3 * PLANE * 2 >= 64 * (BN + 4) * 4, "epilogue staging must fit");

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
    const int n0 = tj.y * BN;
    const int m0 = m_tile * BM;
    const int N = job.N, K = job.K, M = g.M;
    const gcf X = (gcf)(g.X + job.x_off);
    const gcf W = (gcf)job.W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;
    const int ncol = n0 + 32 * wn + r32;
    const float bias = ((gcf)job.bias)[ncol < N ? ncol : N - 1];
    const bool wave_live = (n0 + 32 * wn) < N;

    // staging: float4 units, 8 per 32-float row, 32 rows per pass
    const int s_row = tid >> 3, s_k = (tid & 7) * 4;
    unsigned oa[4], ob[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = m0 + s_row + 32 * i;
        row = row < M ? row : M - 1;
        oa[i] = (unsigned)row * (unsigned)g.ldx + s_k;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = n0 + s_row + 32 * i;
        row = row < N ? row : N - 1;
        ob[i] = (unsigned)row * (unsigned)K + s_k;
    }
    v4f ra[4], rb[2];
    auto gload = [&](int k0) {
        const bool kin = k0 + s_k < K;              // K is a multiple of 4: a unit is fully in or out
        const int kk = kin ? k0 : -s_k;
        const float zm = kin ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(gcf4)(X + (oa[i] + kk)) * zm;
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *(gcf4)(W + (ob[i] + kk)) * zm;
    };

    v16f hi[2], lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { hi[i] = (v16f){0}; lo[i] = (v16f){0}; }

    if (K > 0) gload(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 p1, p2, p3;
            split3(ra[i], p1, p2, p3);
            const int o = (s_row + 32 * i) * PS + s_k;
            *reinterpret_cast<bf16x4*>(&smemh[o]) = p1;
            *reinterpret_cast<bf16x4*>(&smemh[PLANE + o]) = p2;
            *reinterpret_cast<bf16x4*>(&smemh[2 * PLANE + o]) = p3;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x4 p1, p2, p3;
            split3(rb[i], p1, p2, p3);
            const int o = (BM + s_row + 32 * i) * PS + s_k;
            *reinterpret_cast<bf16x4*>(&smemh[o]) = p1;
            *reinterpret_cast<bf16x4*>(&smemh[PLANE + o]) = p2;
            *reinterpret_cast<bf16x4*>(&smemh[2 * PLANE + o]) = p3;
        }
        __syncthreads();
        if (k0 + 32 < K) gload(k0 + 32);
        if (!wave_live) continue;
        __builtin_amdgcn_s_setprio(1);
        // bf16 32x32x16 operand maps: lane (r = l & 31, h = l >> 5) holds A[r][8h .. 8h+7] / B[8h .. 8h+7][r]
        const int oa0 = (64 * wm + r32) * PS + 8 * half;
        const int ob0 = (BM + 32 * wn + r32) * PS + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 b[3], a[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                b[pl] = *reinterpret_cast<const bf16x8*>(&smemh[pl * PLANE + ob0 + 16 * ks]);
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(&smemh[pl * PLANE + oa0 + 32 * i * PS + 16 * ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], lo[i], 0, 0, 0);
                hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], hi[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // epilogue through LDS, identical to the fp32 kernel
    constexpr int ES = BN + 4;
    float* const sE = reinterpret_cast<float*>(smemh);
    const int my_col = 32 * wn + r32;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        __syncthreads();
        if (wm == hh && wave_live) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float v = (hi[i][reg] + lo[i][reg]) + bias;
                    if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                    sE[(32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half) * ES + my_col] = v;
                }
        }
        __syncthreads();
        constexpr int CPR = BN / 4;
#pragma unroll
        for (int u = 0; u < 64 * CPR / 256; ++u) {
            const int idx = tid + 256 * u;
            const int row = idx / CPR, c4 = idx % CPR;
            const int m = m0 + 64 * hh + row, n = n0 + 4 * c4;
            if (m < M && n < N) {
                v4f v = *reinterpret_cast<const v4f*>(&sE[row * ES + 4 * c4]);
                if (EPI == EPI_RES || EPI == EPI_MASK)
                    v += *reinterpret_cast<const v4f __attribute__((address_space(1)))*>((gcf)(g.R + job.r_off + n) + (size_t)m * g.ldr);
                if (EPI == EPI_MASK) {
                    if (g.tap) *reinterpret_cast<v4f __attribute__((address_space(1)))*>((gf)(g.tap + job.m_off + n) + (size_t)m * g.ldt) = v;
                    v *= *reinterpret_cast<const v4f __attribute__((address_space(1)))*>((gcf)(g.Mul + job.m_off + n) + (size_t)m * g.ldm);
                }
                *reinterpret_cast<v4f __attribute__((address_space(1)))*>((gf)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = v;
            }
        }
    }
}

static void launch_gemm_bf16x3(const GemmLaunch& g_in, hipStream_t stream)
{
    GemmLaunch g = g_in;
    const int m_tiles = (g.M + BM - 1) / BM;
    g.mchunk = gemm_mchunk(m_tiles);
    const int chunks = (m_tiles + g.mchunk - 1) / g.mchunk;
    dim3 grid(8 * ((chunks + 7) / 8) * g.mchunk * g.n_tiles), block(256);
    switch (g.epilogue) {
    case EPI_LINEAR: hipLaunchKernelGGL(gemm_bf16x3_kernel<EPI_LINEAR>, grid, block, 0, stream, g); break;
    case EPI_LEAKY:  hipLaunchKernelGGL(gemm_bf16x3_kernel<EPI_LEAKY>, grid, block, 0, stream, g); break;
    case EPI_RES:    hipLaunchKernelGGL(gemm_bf16x3_kernel<EPI_RES>, grid, block, 0, stream, g); break;
    default:         hipLaunchKernelGGL(gemm_bf16x3_kernel<EPI_MASK>, grid, block, 0, stream, g); break;
    }
}

template <int NT, int ABL, int PRIO = 0, int VEC = 2>
static void launch_gemm_nt(const GemmLaunch& g_in, hipStream_t stream)
{
    GemmLaunch g = g_in;
    const int m_tiles = (g.M + BM - 1) / BM;
    g.mchunk = gemm_mchunk(m_tiles);
    const int chunks = (m_tiles + g.mchunk - 1) / g.mchunk;
    const int chunks_per_xcd = (chunks + 7) / 8;
    dim3 grid(8 * chunks_per_xcd * g.mchunk * g.n_tiles), block(256);
    switch (g.epilogue) {
    case EPI_LINEAR: hipLaunchKernelGGL((gemm_f32_kernel<EPI_LINEAR, NT, ABL, PRIO, VEC>), grid, block, 0, stream, g); break;
    case EPI_LEAKY:  hipLaunchKernelGGL((gemm_f32_kernel<EPI_LEAKY, NT, ABL, PRIO, VEC>), grid, block, 0, stream, g); break;
    case EPI_RES:    hipLaunchKernelGGL((gemm_f32_kernel<EPI_RES, NT, ABL, PRIO, VEC>), grid, block, 0, stream, g); break;
    default:         hipLaunchKernelGGL((gemm_f32_kernel<EPI_MASK, NT, ABL, PRIO, VEC>), grid, block, 0, stream, g); break;
    }
}

void launch_gemm(const GemmLaunch& g, hipStream_t stream)
{
    if (g.M <= 0 || g.n_tiles <= 0) return;
    // PRIO = 1 (s_setprio around the MFMA cluster) measured +3..5 % on the 64-wide kernel, 0 on the 128-wide
    // every job is 16-byte aligned (band-padded layouts, weight rows padded to multiples of 4): dwordx4 loads
    static const int use_split = [] { const char* e = getenv("BSRNN_GEMM_BF16X3"); return e ? atoi(e) : 0; }();
    if (use_split && g.tile_n == 64) { launch_gemm_bf16x3(g, stream); return; }
    if (g.tile_n == 128)
        launch_gemm_nt<2, 0, 0, 4>(g, stream);
    else
        launch_gemm_nt<1, 0, 1, 4>(g, stream);
}

}  // namespace bsrnn
