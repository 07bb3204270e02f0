// Grouped fp32 linear layers on the gfx950 matrix cores.
//
// Replaces the ~110 nn.Linear calls of one BSRNN.forward (bsrnn.py:404-412, :422-425, and the
// fc of NormRNNResidual :84) with one launch per "layer slot": blockIdx.x walks a table of
// (band job, 64-column tile), blockIdx.y walks 128-row tiles of the M = C*T frame rows.
//
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain (no xf32 on
// gfx950), which is what the 1e-4 parity budget against the fp32 reference needs; bf16 MFMA
// would be 16x faster and ~1e-2 wrong.  Roofline for this kernel is therefore the fp32
// matrix peak (157.3 TFLOP/s), see DESIGN.md.
//
// Tile: 128 x 64 x 32 per 256-thread workgroup, four waves as 2(M) x 2(N), each wave a 64x32
// patch = two 32x32 accumulators.  Operands are K-contiguous in memory for both X [M][ldx]
// and W [N][K], so A and B fragments are read with the same pattern: lane l owns row (l & 31)
// and the 16-float half (l >> 5) of the 32-deep K slab, fetched as four ds_read_b128; MFMA
// step (j, e) consumes element e of the j-th read, i.e. k = 16*(l>>5) + 4j + e on BOTH
// operands (the sum over k is order-agnostic as long as A and B agree).  LDS rows are padded
// to 36 floats: 36/4 = 9 is odd, so the 16 rows of a ds_read_b128 lane group hit 16 distinct
// 16-byte slots (conflict-free, MI355X guide section LDS).
#include "kernels.h"

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 64, BK = 32, LDS_STRIDE = 36;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmLaunch g)
{
    __shared__ __attribute__((aligned(16))) float sA[BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float sB[BN * LDS_STRIDE];

    const int2 tj = g.tiles[blockIdx.x];
    const GemmJob job = g.jobs[tj.x];
    const int n0 = tj.y * BN;
    const int m0 = blockIdx.y * BM;
    const int N = job.N, K = job.K, M = g.M;
    const float* __restrict__ X = g.X + job.x_off;
    const float* __restrict__ W = job.W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, r32 = lane & 31;

    // staging map: float2 units, 16 per 32-float row
    const int s_row = tid >> 4;           // 0..15 (+16*i)
    const int s_k = (tid & 15) * 2;

    v16f acc0 = {0}, acc1 = {0};
    float2 ra[8], rb[4];

    auto gload = [&](int k0) {
        const int k = k0 + s_k;
        const bool kin = k < K;           // K is even, so k < K implies k+1 < K
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int row = m0 + s_row + 16 * i;
            row = row < M ? row : M - 1;
            ra[i] = kin ? *reinterpret_cast<const float2*>(X + (size_t)row * g.ldx + k) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = n0 + s_row + 16 * i;
            row = row < N ? row : N - 1;
            rb[i] = kin ? *reinterpret_cast<const float2*>(W + (size_t)row * K + k) : make_float2(0.f, 0.f);
        }
    };

    if (K > 0) gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<float2*>(&sA[(s_row + 16 * i) * LDS_STRIDE + s_k]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float2*>(&sB[(s_row + 16 * i) * LDS_STRIDE + s_k]) = rb[i];
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);

        const float* pa0 = &sA[(64 * wm + r32) * LDS_STRIDE + 16 * half];
        const float* pa1 = pa0 + 32 * LDS_STRIDE;
        const float* pb = &sB[(32 * wn + r32) * LDS_STRIDE + 16 * half];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v4f a0 = *reinterpret_cast<const v4f*>(pa0 + 4 * j);
            const v4f a1 = *reinterpret_cast<const v4f*>(pa1 + 4 * j);
            const v4f b = *reinterpret_cast<const v4f*>(pb + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], acc1, 0, 0, 0);
            }
        }
    }

    // epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    const int n = n0 + 32 * wn + r32;
    if (n >= N) return;
    const float bias = job.bias[n];
    float* __restrict__ Y = g.Y + job.y_off + n;
    const float* __restrict__ Rp = (EPI == EPI_RES || EPI == EPI_MASK) ? g.R + job.r_off + n : nullptr;
    const float* __restrict__ Mp = (EPI == EPI_MASK) ? g.Mul + job.m_off + n : nullptr;
    float* __restrict__ Tp = (EPI == EPI_MASK && g.tap) ? g.tap + job.m_off + n : nullptr;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = m0 + 64 * wm + 32 * s + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (m < M) {
                float v = (s == 0 ? acc0[reg] : acc1[reg]) + bias;
                if (EPI == EPI_LEAKY) v = v >= 0.f ? v : 0.01f * v;
                if (EPI == EPI_RES) v += Rp[(size_t)m * g.ldr];
                if (EPI == EPI_MASK) {
                    v += Rp[(size_t)m * g.ldr];
                    if (Tp) Tp[(size_t)m * g.ldt] = v;
                    v *= Mp[(size_t)m * g.ldm];
                }
                Y[(size_t)m * g.ldy] = v;
            }
        }
    }
}

void launch_gemm(const GemmLaunch& g, hipStream_t stream)
{
    if (g.M <= 0 || g.n_tiles <= 0) return;
    dim3 grid(g.n_tiles, (g.M + BM - 1) / BM), block(256);
    switch (g.epilogue) {
    case EPI_LINEAR: hipLaunchKernelGGL(gemm_f32_kernel<EPI_LINEAR>, grid, block, 0, stream, g); break;
    case EPI_LEAKY:  hipLaunchKernelGGL(gemm_f32_kernel<EPI_LEAKY>, grid, block, 0, stream, g); break;
    case EPI_RES:    hipLaunchKernelGGL(gemm_f32_kernel<EPI_RES>, grid, block, 0, stream, g); break;
    default:         hipLaunchKernelGGL(gemm_f32_kernel<EPI_MASK>, grid, block, 0, stream, g); break;
    }
}

}  // namespace bsrnn
