// Small-M path of the grouped per-band Linear layers: a handful of frame rows (the one-frame streaming step of
// BSRNN.forward_recurrent, bsrnn.py:445-510, has M = C = 2 rows).  A 128-row MFMA tile
// would carry 2 useful rows and every launch would pay a tile's whole prologue / K loop / epilogue (19 us per layer
// measured, 10 layers per step); with so few rows the layer is a weight stream (30 MB per step over all layers) and the
// arithmetic is free, so:
//   * exact fp32 on the vector ALU (fma chains; no 16-bit operand pieces, hence no range limit on this path);
//   * one wave computes NF = 4 output features for all rows: lanes split K (16 bytes per lane and pass: every weight row is
//     read as whole 1 KB lines), a butterfly of lane shuffles adds the 64 partial sums;
//   * 16 features per workgroup: a layer spreads over ~130 workgroups, so its weights stream from L2 / HBM on half the chip.
// Same job / tile tables, layouts, epilogues (bias, LeakyReLU, residual, mask tap and multiply, zeroed pad columns) as
// gemm.hip.  Selected by api.hip for calls of at most GEMV_MAX_FRAME_ROWS frame rows (kernels.h).
#include "kernels.h"

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gcf;
typedef float __attribute__((address_space(1)))* gf;
typedef const v4f __attribute__((address_space(1)))* gc4;

template <int EPI, int MR, int NF>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemmLaunch g)
{
    // block -> (column tile of the launch's tile table, 16-feature slice of it); wave -> NF features of the slice
    const int per_tile = g.tile_n / (4 * NF);
    const int2 tj = g.tiles[blockIdx.x / per_tile];
    const GemmJob job = g.jobs[tj.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = tj.y * g.tile_n + (blockIdx.x % per_tile) * (4 * NF) + NF * wave;
    const int N = job.N, K = job.K, M = g.M;
    const int N8 = (N + 7) & ~7;                     // pad columns [N, N8) are written as zeros (see gemm.hip)
    if (n0 >= N8) return;                            // wave-uniform

    const gcf W = (gcf)job.W;
    const gcf X = (gcf)g.X + job.x_off;
    // weight rows beyond N are clamped to the last row (their results are discarded); K is a multiple of 8
    unsigned wrow[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wrow[f] = (unsigned)((n0 + f < N ? n0 + f : N - 1)) * (unsigned)K;
    // rows in chunks of MR (the weights of a later chunk come from L1)
    for (int m0 = 0; m0 < M; m0 += MR) {
    float acc[MR][NF];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[m][f] = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        v4f w[NF], x[MR];
#pragma unroll
        for (int f = 0; f < NF; ++f) w[f] = *(gc4)(W + wrow[f] + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) x[m] = *(gc4)(X + (size_t)(m0 + m < M ? m0 + m : M - 1) * g.ldx + k);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float a = acc[m][f];
                a = __builtin_fmaf(x[m][0], w[f][0], a);
                a = __builtin_fmaf(x[m][1], w[f][1], a);
                a = __builtin_fmaf(x[m][2], w[f][2], a);
                a = __builtin_fmaf(x[m][3], w[f][3], a);
                acc[m][f] = a;
            }
    }
    // sum over the 64 lanes (every lane ends up with the total)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float a = acc[m][f];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
            acc[m][f] = a;
        }
    // lane (m, f) finishes output (row m, feature n0 + f)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (lane != m * NF + f || m0 + m >= M) continue;
            const int n = n0 + f;
            const size_t mr = (size_t)(m0 + m);
            if (n >= N8) continue;
            float v = 0.f;
            if (n < N) {
                v = acc[m][f] + ((gcf)job.bias)[n];
                if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                if (EPI == EPI_RES || EPI == EPI_MASK) v += ((gcf)g.R)[mr * g.ldr + job.r_off + n];
                if (EPI == EPI_MASK) {
                    if (g.tap) ((gf)g.tap)[mr * g.ldt + job.m_off + n] = v;
                    v *= ((gcf)g.Mul)[mr * g.ldm + job.m_off + n];
                }
            } else if (EPI == EPI_MASK && g.tap) {
                ((gf)g.tap)[mr * g.ldt + job.m_off + n] = 0.f;
            }
            ((gf)g.Y)[mr * g.ldy + job.y_off + n] = v;
        }
    }
}

template <int MR, int NF>
static void launch_gemv_mr(const GemmLaunch& g, hipStream_t stream)
{
    dim3 grid(g.n_tiles * (g.tile_n / (4 * NF))), block(256);
    switch (g.epilogue) {
    case EPI_LINEAR: hipLaunchKernelGGL((gemv_rows_kernel<EPI_LINEAR, MR, NF>), grid, block, 0, stream, g); break;
    case EPI_LEAKY:  hipLaunchKernelGGL((gemv_rows_kernel<EPI_LEAKY, MR, NF>), grid, block, 0, stream, g); break;
    case EPI_RES:    hipLaunchKernelGGL((gemv_rows_kernel<EPI_RES, MR, NF>), grid, block, 0, stream, g); break;
    default:         hipLaunchKernelGGL((gemv_rows_kernel<EPI_MASK, MR, NF>), grid, block, 0, stream, g); break;
    }
}

void launch_gemv(const GemmLaunch& g, hipStream_t stream)
{
    if (g.M <= 0 || g.n_tiles <= 0) return;
    if (g.M <= 2) launch_gemv_mr<2, 4>(g, stream);
    else if (g.M <= 4) launch_gemv_mr<4, 4>(g, stream);
    else launch_gemv_mr<8, 4>(g, stream);          // (more rows: in chunks of 8)
}

}  // namespace bsrnn
