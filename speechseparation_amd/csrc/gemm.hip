// Grouped per-band linear layers on the gfx950 matrix cores.
//
// Replaces the ~110 nn.Linear calls of one BSRNN.forward (bsrnn.py:404-412, :422-425, and the
// fc of NormRNNResidual :84) with one launch per "layer slot": a table of (band job, column
// tile) x 128-row tiles of the M = C*T frame rows, walked in an XCD-aware order.
//
// Two kernels, selected by launch_gemm() from BSRNN_GEMM (kernels.h, GemmMode):
//   gemm_h2_kernel     (default, "fp16x2"; TERMS = 1: "fp16")  fp32 operands as two fp16 pieces, three MFMA terms on
//                      v_mfma_f32_32x32x16_f16 with fp32 accumulation - fp32-level accuracy at 3/16 of the fp32
//                      matrix-pipe time; pipelined main loop (two LDS stages, one barrier per slab);
//   gemm_f32_kernel    ("f32", and the re-run of a call whose operands left the fp16x2 range)  exact fp32 on
//                      v_mfma_f32_32x32x2_f32, a k-ordered fp32 fma chain (no xf32 on gfx950).
// (The per-band MLP chains themselves run fused, one workgroup per (band, row tile): mlp_chain.hip; these per-layer
// launches carry the residual fc of the dual-path blocks, the unfused A/B flow BSRNN_MLP=layers and the fp32 mode.)
// Both share the job / tile tables, the XCD mapping, the band-padded layouts (every segment 32-byte aligned, pad
// columns zero) and the LDS-staged 16-byte epilogue with the fused bias / LeakyReLU / residual / mask variants.
//
// gemm_f32_kernel: tile 128 x (64*NT) x BK per 256-thread workgroup, four waves as 2(M) x 2(N), each wave a
// 64 x (32*NT) patch = 2*NT accumulators of 32x32.  Operands are K-contiguous in memory for both X [M][ldx] and
// W [N][K], so A and B fragments are read with the same pattern: lane l owns row (l & 31) and the half (l >> 5)
// of the BK-deep K slab, fetched as BK/8 ds_read_b128; MFMA step (j, e) consumes element e of the
// j-th read, i.e. k = (BK/2)*(l>>5) + 4j + e on BOTH operands (the sum over k is order-agnostic as
// long as A and B agree).  LDS rows are padded to BK+4 floats: (BK+4)/4 is odd, so the 16 rows of
// a ds_read_b128 lane group hit 16 distinct 16-byte slots.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace bsrnn {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef GEMM_H2_SCHED
#define GEMM_H2_SCHED 1
#endif
#ifndef GEMM_BK
#define GEMM_BK 32
#endif
constexpr int BM = 128, BK = GEMM_BK, LDS_STRIDE = BK + 4, MCHUNK_MAX = 2;
// m-tiles per XCD-pinned chunk: fewer for small batches so that all eight XCDs still get work.  The cap was 8 for the
// fp32 kernel; with the split-precision kernels (3x less time per tile, same bytes) a chunk of 2 m-tiles keeps its
// activation rows resident in the XCD's L2 across all column tiles and measures 2.5 % faster (sweep 1/2/3/4/8/16).
static inline int gemm_mchunk(int m_tiles)
{
    static const int cap = [] { const char* e = getenv("BSRNN_GEMM_MCHUNK"); const int v = e ? atoi(e) : MCHUNK_MAX; return v < 1 ? 1 : v; }();   // measurement knob
    const int c = (m_tiles + 7) / 8;
    return c < 1 ? 1 : (c > cap ? cap : c);
}
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
                        if (ncol0 + 32 * jn >= N) v = 0.f;          // pad columns of the output segment are exactly zero
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
            // Columns [N, round8(N)) are written too (as zeros): the next layer's K loop runs to round8(K) against
            // zero weights, and a buffer shared by layers of different widths must not leak a stale (possibly
            // non-finite) value of a wider layer into that product.
            if (m < M && n < ((N + 7) & ~7)) {
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

// fp16x2 split of an fp32 value: a ~ a1 + 2^-11 a2 with a1 = fp16(a), a2 = fp16(2^11 (a - a1)) - 22 significant bits, i.e. 2^-23
// relative representation error per operand, the level of fp32 accumulation noise.  Three MFMA terms:
// hi += a1 b1;  lo += a1 b2 + a2 b1;  result = hi + 2^-11 lo  (a2 b2 ~ 2^-22 is dropped).  The 2^11 scaling keeps a2 out of the
// fp16 subnormal range for |a| >= 2^-14.  (Scheme: Ootomo & Yokota, "Recovering single precision accuracy from Tensor
// Cores...", 2022; here the correction terms get their own accumulator.)
template <int NP> struct Piece;
template <> struct Piece<2> {
    typedef _Float16 T;
    typedef _Float16 T4 __attribute__((ext_vector_type(4)));
    typedef _Float16 T8 __attribute__((ext_vector_type(8)));
    // a - a1 and the scaling are exact in fp32, so the fused form fma(-a1, 2048, 2048 a) rounds once, to the same value as
    // convert-back / subtract / multiply / convert - two VALU instructions per element (v_pk_mul_f32 + v_fma_mix{lo,hi}_f16).
    // |a| > 65504 makes a1 infinite: callers track max |a| and raise the range flag (the result is invalid either way).
    static __device__ __forceinline__ void split(v4f a, T4* p)
    {
        asm("" : "+v"(a));       // split the fp32 value itself, whatever produced it (lstm.hip, split_h2)
#pragma unroll
        for (int i = 0; i < 4; ++i) p[0][i] = (_Float16)a[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[1][i] = (_Float16)__builtin_fmaf(-(float)p[0][i], 2048.f, a[i] * 2048.f);
    }
};

// =====================================================================================
// fp16x2 product kernel, pipelined: fp32 activations split on the fly, weights as two fp16 pieces split once on the
// host, hi / lo accumulators.  In-kernel stamps of the first, unpipelined version (round 1, profiles/r01_gemm_split_bench.txt)
// showed a 1.6 us slab of which the MFMAs were 0.37: every wave ran  barrier - split + ds_write - barrier - global-load
// issue - ds_read + MFMA  in sequence.  Here
//   * LDS holds TWO stages; a wave splits slab k+1 into stage (k+1)&1 while the MFMAs of slab k (stage k&1) are
//     in the matrix pipe, and there is ONE barrier per slab;
//   * the split / ds_write / global-load instructions are placed after the MFMAs they should hide behind and
//     interleaved with them by sched_group_barrier (one MFMA, then a few VALU / LDS / VMEM instructions);
//   * rows are 64 bytes without padding, 16-byte units XOR-swizzled with (row >> 2) & 3, which keeps both the
//     ds_read_b128 fragment reads and the stage at 32 KB (two stages + two workgroups per CU fit in 160 KB).
// =====================================================================================
// Epilogue of the fp16x2 kernels: accumulators -> LDS -> 16-byte global stores (whole lines per row), with the fused
// bias / LeakyReLU / residual / mask variants.
template <int EPI, int NT, int TERMS, int SMEM_H>
__device__ __forceinline__ void h2_epilogue(const GemmLaunch& g, const GemmJob& job, v16f (&acc)[2][NT][2], const float (&bias)[NT],
                                            const bool (&live)[NT], _Float16* smemh, const int m0, const int n0, const int tid)
{
    constexpr int BN = 64 * NT;
    constexpr int ES2 = BN + 4;                     // staging row stride (floats) of the two-pass epilogue
    // epilogues without extra operands stage the whole 128-row tile at once when LDS has room (one barrier pair, all four
    // waves write); the residual / mask epilogues keep two 64-row passes (their prefetched rows would not fit in registers)
    constexpr bool ONEPASS = (EPI == EPI_LEAKY || EPI == EPI_LINEAR) && SMEM_H * 2 >= BM * BN * 4;
    constexpr int HP = ONEPASS ? 1 : 2, RPP = BM / HP;
    constexpr int ES = !ONEPASS ? ES2 : (SMEM_H * 2 >= BM * (BN + 4) * 4 ? BN + 4 : BN);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;
    const int wcol = 32 * NT * wn;
    const int N = job.N, M = g.M;
    float* const sE = reinterpret_cast<float*>(smemh);
    typedef const v4f __attribute__((address_space(1)))* gc4;
    typedef v4f __attribute__((address_space(1)))* g4;
    constexpr int UPR4 = BN / 4, NU = RPP * UPR4 / 256;
#pragma unroll
    for (int hh = 0; hh < HP; ++hh) {
        if (hh) __syncthreads();
        // residual / multiplier rows of this half are requested before the accumulators go through LDS, so their
        // latency hides behind the staging and its barrier (the mask launch reads 134 MB this way)
        v4f rv[NU], mv[NU];
        if (EPI == EPI_RES || EPI == EPI_MASK) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = tid + 256 * u;
                const int row = idx / UPR4, c4 = idx % UPR4;
                int m = m0 + RPP * hh + row, n = n0 + 4 * c4;
                m = m < M ? m : M - 1;
                n = n < N ? n : 0;
                rv[u] = *(gc4)((gcf)(g.R + job.r_off + n) + (size_t)m * g.ldr);
                if (EPI == EPI_MASK) mv[u] = *(gc4)((gcf)(g.Mul + job.m_off + n) + (size_t)m * g.ldm);
            }
        }
        if (ONEPASS || wm == hh) {
            const int rb = ONEPASS ? 64 * wm : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!live[j]) continue;
                const int col = wcol + 32 * j + r32;
                const bool in = n0 + col < N;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = (TERMS == 1 ? acc[i][j][0][reg] : acc[i][j][0][reg] + (1.f / 2048.f) * acc[i][j][1][reg]) + bias[j];
                        if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                        if (!in) v = 0.f;
                        sE[(rb + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half) * ES + col] = v;
                    }
            }
        }
        __syncthreads();
        {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = tid + 256 * u;
                const int row = idx / UPR4, c4 = idx % UPR4;
                const int m = m0 + RPP * hh + row, n = n0 + 4 * c4;
                if (m < M && n < ((N + 7) & ~7)) {
                    v4f v = *reinterpret_cast<const v4f*>(&sE[row * ES + 4 * c4]);
                    if (EPI == EPI_RES || EPI == EPI_MASK) v += rv[u];
                    if (EPI == EPI_MASK) {
                        if (g.tap) *(g4)((gf)(g.tap + job.m_off + n) + (size_t)m * g.ldt) = v;
                        v *= mv[u];
                    }
                    *(g4)((gf)(g.Y + job.y_off + n) + (size_t)m * g.ldy) = v;
                }
            }
        }
    }
}

template <int NVALU>
__device__ __forceinline__ void sched_slice()
{
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA, then behind it
    __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);      //   VALU (the split)
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          //   one LDS write
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          //   one global load
}

// TERMS = 3: fp16x2 (hi + two correction terms).  TERMS = 1: plain fp16 operands, one MFMA term (the "16-bit compute"
// configuration of BASELINE.json: ~5e-4 relative error per product) - only the first piece of each operand is staged.
template <int EPI, int NT, int ABL = 0, int TERMS = 3>
__global__ __launch_bounds__(256, ((NT == 1 || TERMS == 1) ? 3 : 2)) void gemm_h2_kernel(GemmLaunch g)
{
    constexpr int NPL = TERMS == 1 ? 1 : 2;         // pieces staged per operand
    typedef _Float16 hT;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef const h8 __attribute__((address_space(1)))* gch8;
    typedef const hT __attribute__((address_space(1)))* gch;
    typedef const v4f __attribute__((address_space(1)))* gcf4;
    constexpr int BN = 64 * NT;
    constexpr int PLANE = (BM + BN) * 32;           // halves per piece and stage: A rows then B rows, 64-byte rows
    constexpr int STAGE = NPL * PLANE;
    constexpr int SMEM_H = 2 * STAGE * 2 >= 64 * (BN + 4) * 4 ? 2 * STAGE : 64 * (BN + 4) * 2;    // two stages, or the epilogue staging tile if larger
    __shared__ __attribute__((aligned(16))) hT smemh[SMEM_H];

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
    const int N = job.N, K = job.K, M = g.M;       // K is a multiple of 8

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;
    const int wcol = 32 * NT * wn;
    float bias[NT];
    bool live[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nc = n0 + wcol + 32 * j + r32;
        bias[j] = ((gcf)job.bias)[nc < N ? nc : N - 1];
        live[j] = (n0 + wcol + 32 * j) < N;
    }

    // staging: 8 threads per row, 32 rows per pass, whole 128-byte lines for both operands: A fp32 in float4 units,
    // B as the 8 units of a weight row's slab (units 0-3 first piece, 4-7 second piece)
    const int s_row = tid >> 3, s_k4 = tid & 7;
    const gch Wp = (gch)job.Wp;
    const gcf X = (gcf)(g.X + job.x_off);
    // weights: [N][K32 / 32][2 pieces][32] fp16 (split_host.h, pack_h2_slabs_host): the two pieces of a slab of a row
    // are one 128-byte line; rows are zero-padded to K32, so the weight side needs no tail masking
    const unsigned wrow = (unsigned)job.wrow;
    // weight staging: 8 units (both pieces) of a row's slab per 8 threads, or the 4 units of the first piece per 4 threads
    constexpr int BPASS = TERMS == 1 ? NT : 2 * NT, BROWS = TERMS == 1 ? 64 : 32;
    const int b_row = TERMS == 1 ? tid >> 2 : tid >> 3, b_u = TERMS == 1 ? tid & 3 : tid & 7;
    unsigned oa[4], obp[BPASS];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int row = m0 + s_row + 32 * i; row = row < M ? row : M - 1; oa[i] = (unsigned)row * (unsigned)g.ldx + 4 * s_k4; }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) { int row = n0 + b_row + BROWS * i; row = row < N ? row : N - 1; obp[i] = (unsigned)row * wrow + 8 * b_u; }
    v4f ra[4];
    h8 rbp[BPASS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BPASS; ++i) rbp[i] = *(gch8)(Wp + (obp[i] + 2 * k0));
        if (k0 + 32 <= K) {          // plain loads: no arithmetic on the registers until they are written to LDS
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *(gcf4)(X + (oa[i] + k0));
            return;
        }
        const bool sin = k0 + 4 * s_k4 < K;
        const int sk = sin ? k0 : -4 * s_k4;
        const float zm = sin ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(gcf4)(X + (oa[i] + sk)) * zm;
    };
    // steady state: wave-uniform base advanced by the slab (scalar) + loop-invariant 32-bit byte offsets, so the
    // loads take the saddr + voffset form and cost no VALU address arithmetic
    unsigned oab[4], obb[BPASS];
#pragma unroll
    for (int i = 0; i < 4; ++i) oab[i] = oa[i] * 4u;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) obb[i] = obp[i] * 2u;
    typedef const char __attribute__((address_space(1)))* gcc;
    auto gload_full = [&](int k0) {
        const gcc wb_ = (gcc)(Wp + 2 * k0);
        const gcc xb_ = (gcc)(X + k0);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) rbp[i] = *(gch8)(wb_ + obb[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(gcf4)(xb_ + oab[i]);
    };
    // LDS offsets (halves) of this thread's staging units; 16-byte unit kq of row r sits at unit kq ^ ((r >> 2) & 3)
    int wa[4], wb[BPASS];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = s_row + 32 * i; wa[i] = r * 32 + (((s_k4 >> 1) ^ ((r >> 2) & 3)) * 8) + (s_k4 & 1) * 4; }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) { const int r = BM + b_row + BROWS * i; wb[i] = (b_u >> 2) * PLANE + r * 32 + (((b_u & 3) ^ ((r >> 2) & 3)) * 8); }
    float amax = 0.f;                             // largest |activation| staged by this thread (range guard)
    auto put_a = [&](hT* st, int i) {
        h4 p[2];
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(ra[i][0])), __builtin_fabsf(ra[i][1]));      // v_max3_f32 with |.| modifiers
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(ra[i][2])), __builtin_fabsf(ra[i][3]));
        if (ABL & 32) {          // measurement only: pretend the activations arrive pre-split (no conversion VALU, same bytes)
            typedef float v2f_ __attribute__((ext_vector_type(2)));
            p[0] = __builtin_bit_cast(h4, (v2f_){ra[i][0], ra[i][1]});
            p[1] = __builtin_bit_cast(h4, (v2f_){ra[i][2], ra[i][3]});
        } else {
            Piece<2>::split(ra[i], p);
        }
        *reinterpret_cast<h4*>(&st[wa[i]]) = p[0];
        if (TERMS != 1) *reinterpret_cast<h4*>(&st[PLANE + wa[i]]) = p[1];
    };
    auto put_b = [&](hT* st, int i) {
        *reinterpret_cast<h8*>(&st[wb[i]]) = rbp[i];
    };

    v16f acc[2][NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j][0] = (v16f){0}; acc[i][j][1] = (v16f){0}; }

    // fragment offsets: lane (r = l & 31, h = l >> 5) holds k = 16 ks + 8 h .. + 7, i.e. unit 2 ks + h
    const int swz = (r32 >> 2) & 3;
    const int fa = (64 * wm + r32) * 32, fb = (BM + wcol + r32) * 32;
    const int fu[2] = {((0 + half) ^ swz) * 8, ((2 + half) ^ swz) * 8};

    if (K > 0) {
        gload(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) put_a(smemh, i);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) put_b(smemh, i);
        if (K > 32) gload(32);
    }
    __syncthreads();
    // One slab.  FAST = steady state (every column tile of the wave live, two more full slabs to come): no branches,
    // so the whole body is one basic block and the staging can be interleaved with the MFMAs; the generic form
    // handles ragged tiles and the last slabs.
    auto step = [&](auto fast_tag, const int k0) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const hT* const cur = smemh + ((k0 >> 5) & 1) * STAGE;
        hT* const nxt = smemh + (((k0 >> 5) + 1) & 1) * STAGE;
        const bool more = FAST || k0 + 32 < K;     // registers hold slab k0 + 32
        if (FAST || live[0]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 b[NT][2], a[2][2];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j][pl] = *reinterpret_cast<const h8*>(&cur[pl * PLANE + fb + 32 * j * 32 + fu[ks]]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i][pl] = *reinterpret_cast<const h8*>(&cur[pl * PLANE + fa + 32 * i * 32 + fu[ks]]);
                }
                if (!(ABL & 2)) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (!FAST && j > 0 && !live[j]) continue;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (TERMS != 1) {
                                acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j][1], 0, 0, 0);
                                acc[i][j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j][1], 0, 0, 0);
                            }
                            acc[i][j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j][0], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) { acc[i][0][0][ks] += (float)a[i][0][0] + (float)a[i][NPL - 1][1]; acc[i][NT - 1][1][ks] += (float)b[NT - 1][0][0] + (float)b[NT - 1][NPL - 1][1]; }
                }
                // the next slab's staging rides behind this half's MFMAs: A rows after ks = 0, B rows and the
                // global loads of the slab after that behind ks = 1
                if (more) {
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) put_a(nxt, i);
                    } else {
#pragma unroll
                        for (int i = 0; i < BPASS; ++i) put_b(nxt, i);
                    }
                }
                if (ks == 1 && !(ABL & 1)) {
                    if (FAST) gload_full(k0 + 64);
                    else if (k0 + 64 < K) gload(k0 + 64);
                }
#if GEMM_H2_SCHED
                if (FAST) {
#pragma unroll
                    for (int q = 0; q < 6 * NT; ++q) {
                        if (ks == 0) sched_slice<8>(); else sched_slice<2>();
                    }
                }
#endif
            }
        } else if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) put_a(nxt, i);
#pragma unroll
            for (int i = 0; i < BPASS; ++i) put_b(nxt, i);
            if (!(ABL & 1) && k0 + 64 < K) gload(k0 + 64);
        }
        __syncthreads();
    };
    {
        int k0 = 0;
        if (live[NT - 1])
            for (; k0 + 96 <= K; k0 += 32) step(std::true_type(), k0);
        for (; k0 < K; k0 += 32) step(std::false_type(), k0);
    }

    h2_epilogue<EPI, NT, TERMS, SMEM_H>(g, job, acc, bias, live, smemh, m0, n0, tid);
    // range guard: a finite operand beyond the fp16 range saturated its first piece.  (NaN operands do not raise it - v_max3
    // drops them - and need not: they come out as NaN exactly as they do from the fp32 reference.)
    if (amax > 65504.f && g.range_flag) *g.range_flag = 1;
}

template <int NT, int ABL = 0, int TERMS = 3>
static void launch_gemm_h2(const GemmLaunch& g_in, hipStream_t stream)
{
    GemmLaunch g = g_in;
    const int m_tiles = (g.M + BM - 1) / BM;
    g.mchunk = gemm_mchunk(m_tiles);
    const int chunks = (m_tiles + g.mchunk - 1) / g.mchunk;
    dim3 grid(8 * ((chunks + 7) / 8) * g.mchunk * g.n_tiles), block(256);
    switch (g.epilogue) {
    case EPI_LINEAR: hipLaunchKernelGGL((gemm_h2_kernel<EPI_LINEAR, NT, ABL, TERMS>), grid, block, 0, stream, g); break;
    case EPI_LEAKY:  hipLaunchKernelGGL((gemm_h2_kernel<EPI_LEAKY, NT, ABL, TERMS>), grid, block, 0, stream, g); break;
    case EPI_RES:    hipLaunchKernelGGL((gemm_h2_kernel<EPI_RES, NT, ABL, TERMS>), grid, block, 0, stream, g); break;
    default:         hipLaunchKernelGGL((gemm_h2_kernel<EPI_MASK, NT, ABL, TERMS>), grid, block, 0, stream, g); break;
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

static thread_local bool tl_force_f32 = false;
void set_force_f32(bool on) { tl_force_f32 = on; }
bool force_f32() { return tl_force_f32; }

int gemm_mode()
{
    static const int mode = [] {
        const char* e = getenv("BSRNN_GEMM");
        if (!e || !*e || !strcmp(e, "fp16x2")) return (int)GEMM_FP16X2;
        if (!strcmp(e, "f32")) return (int)GEMM_F32;
        if (!strcmp(e, "fp16")) return (int)GEMM_FP16;
        if (!strcmp(e, "bf16")) return (int)GEMM_BF16;
        fprintf(stderr, "bsrnn: unknown BSRNN_GEMM='%s' (f32 | fp16x2 | fp16 | bf16), using fp16x2\n", e);
        return (int)GEMM_FP16X2;
    }();
    return mode;
}

void launch_gemm(const GemmLaunch& g, hipStream_t stream)
{
    if (g.M <= 0 || g.n_tiles <= 0) return;
    switch (force_f32() ? (int)GEMM_F32 : gemm_mode()) {
    case GEMM_FP16X2:
    case GEMM_BF16:                 // (bf16 operands exist in the fused chains only, kernels.h)
        if (g.tile_n == 128) launch_gemm_h2<2>(g, stream);
        else launch_gemm_h2<1>(g, stream);
        return;
    case GEMM_FP16:
        if (g.tile_n == 128) launch_gemm_h2<2, 0, 1>(g, stream);
        else launch_gemm_h2<1, 0, 1>(g, stream);
        return;
    default: break;
    }
    // exact fp32 on the fp32 matrix pipe.  PRIO = 1 (s_setprio around the MFMA cluster) measured +3..5 % on the
    // 64-wide kernel, 0 on the 128-wide; every job is 16-byte aligned (band-padded layouts, padded weight rows)
    if (g.tile_n == 128)
        launch_gemm_nt<2, 0, 0, 4>(g, stream);
    else
        launch_gemm_nt<1, 0, 1, 4>(g, stream);
}

}  // namespace bsrnn
