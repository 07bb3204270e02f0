// Fused per-band MLP chains on the gfx950 matrix cores.
//
// BandSplit (bsrnn.py:404-415):      x_b -> pre.0 -> pre.2 (= residual P) -> fc.0 -> fc.2 -> fc.4 -> Z[:, :, b, :]
// MaskEstimation (bsrnn.py:420-443): Z[:, :, b, :] -> back.0 -> back.2 -> back.4 -> post.0 -> post.2, + P, * x -> y_b
// One workgroup runs the WHOLE five-layer chain of one band for a tile of frame rows; the intermediates never leave
// the CU (the per-layer launches of gemm.hip wrote and re-read 66-93 MB per layer: 2.07 GB per step at 3.5 TB/s for
// 1.30 GB of algorithmic bytes, and paid ~25 us of fixed cost per launch - profiles/r01k_*).
//
// Arithmetic: the fp16x2 scheme of gemm.hip (operands as two fp16 pieces, hi += w1 x1, lo += w1 x2 + w2 x1, result
// hi + 2^-11 lo, fp32 accumulate), same products in the same k order, so the chain reproduces the unfused flow.
//
// Orientation: the products are computed TRANSPOSED, D^T = W X^T, on v_mfma_f32_32x32x16_f16:
//   A operand = weights   (lane (r, h): W[32 t + r][16 ks + 8 h + j]),  streamed global -> VGPR, never through LDS: every
//               weight byte is used by exactly one wave of the workgroup, and the host packs, per (band, layer, wave),
//               the fragments in the order that wave consumes them: one linear stream, each fragment a coalesced 1 KB;
//   B operand = activations (lane (m, h): x[m][16 ks + 8 h + j]), shared by all waves, in LDS as two fp16 pieces in
//               [k / 8][row m][8] order: every fragment read is one linear ds_read_b128 (address = base + 16 lane);
//   D = [feature n][row m]: lane = activation row, registers = 16 features.  The next layer sums over features, i.e. over
//               REGISTERS of D, so its B fragments need no transpose: a lane splits its own values and stores 4 consecutive
//               features (8 bytes per piece) into the [k / 8][m][8] image - plain ds_write_b64, conflict-free.
// So a layer is: K loop without barriers or staging (weights in flight in registers, activations static in LDS), barrier,
// epilogue (bias, LeakyReLU, split, LDS image of the next layer's input [+ P to HBM]), barrier.
//
// Geometry: 8 waves per workgroup as RT row tiles x NW waves; a row tile is 32 frame rows (the MFMA's N), its NW waves
// own the feature tiles t = wn, wn + NW, ... (at most CT = 3 each).  LDS holds rows x Kmax x 4 bytes, so
//   bands up to 768 columns: NW = 8, RT = 1 (32 rows, 96 KB);  up to 384: NW = 4, RT = 2;  up to 192: NW = 2, RT = 4.
// One launch per chain; tasks (band, row block) in longest-first order.
#include "kernels.h"

#include <type_traits>

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef const float __attribute__((address_space(1)))* gcf;
typedef float __attribute__((address_space(1)))* gf;
typedef const v4f __attribute__((address_space(1)))* gc4;
typedef v4f __attribute__((address_space(1)))* g4;
typedef const h8 __attribute__((address_space(1)))* gch8;
typedef const char __attribute__((address_space(1)))* gcc;

#ifndef CHAIN_PD
#define CHAIN_PD 3                 // k-steps of weight fragments in flight per wave (register sets)
#endif
constexpr int PD = CHAIN_PD;

__device__ __forceinline__ void split4(const v4f a, h4& p0, h4& p1)
{
    // identical to Piece<2>::split of gemm.hip: a1 = fp16(a), a2 = fp16(fma(-a1, 2048, 2048 a))
#pragma unroll
    for (int i = 0; i < 4; ++i) p0[i] = (_Float16)a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) p1[i] = (_Float16)__builtin_fmaf(-(float)p0[i], 2048.f, a[i] * 2048.f);
}

template <int CHAIN, int TERMS>
__global__ __launch_bounds__(512, 2) void mlp_chain_kernel(ChainLaunch g)
{
    constexpr int NPL = TERMS == 1 ? 1 : 2;       // pieces per operand
    __shared__ __attribute__((aligned(16))) char smem[CHAIN_LDS_EX + CHAIN_LDS_BIAS];
    float* const sbias = reinterpret_cast<float*>(smem + CHAIN_LDS_EX);

    // block -> (band, first row): classes of 32 / 64 / 128 / 256 rows per workgroup, class by class, band by band
    int bid = blockIdx.x, di = 0, row0 = 0;
    {
        int rows = 32;
#pragma unroll
        for (int cl = 0; cl < 4; ++cl, rows *= 2) {
            const int nblk = (g.M + rows - 1) / rows, ncl = g.n_cls[cl] * nblk;
            if (bid < ncl || cl == 3) { di += bid / nblk; row0 = (bid - (bid / nblk) * nblk) * rows; break; }
            bid -= ncl; di += g.n_cls[cl];
        }
    }
    const ChainDesc* const dp = g.desc + di;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = dp->NW;
    const int rt = wave / NW, wn = wave - rt * NW;
    const int m = lane & 31, h = lane >> 5;
    const int M = g.M;
    const int row_raw = row0 + 32 * rt + m;
    const bool row_ok = row_raw < M;
    const int row = row_ok ? row_raw : M - 1;

    if (CHAIN == CHAIN_SPLIT && dp->constant) {
        // TrainableConstantModule (bsrnn.py:12-24): the zero-width band's feature is one learned vector for every frame
        const gcf cst = (gcf)dp->bias;
        for (int i = tid; i < 32 * 8 * 16; i += 512) {           // 256 rows x 16 float4
            const int r = row0 + (i >> 4), c4 = i & 15;
            if (r < M) *(g4)((gf)g.Z + (size_t)r * g.ldz + dp->z_off + 4 * c4) = *(gc4)(cst + 4 * c4);
        }
        return;
    }

    const int plane = dp->plane_units * 512;                     // bytes of one piece of one row tile
    char* const ex = smem + rt * NPL * plane;                    // this row tile's activation image
    float amax = 0.f;

    // ---- biases of the five layers -> LDS; input rows -> LDS (split on the way)
    {
        const gcf bsrc = (gcf)dp->bias;
        const int nb = dp->nbias;
        for (int i = tid; i < nb; i += 512) sbias[i] = bsrc[i];
        const gcf xin = (gcf)g.Xin + (size_t)row * g.ldx + dp->in_off;
        const int U0 = 2 * dp->L[0].K16, K0 = dp->K0;
        for (int u = wn; u < U0; u += NW) {
            const int k = 8 * u + 4 * h;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (k < K0) v = *(gc4)(xin + k);
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
            h4 p0, p1;
            split4(v, p0, p1);
            *reinterpret_cast<h4*>(ex + u * 512 + m * 16 + 8 * h) = p0;
            if (NPL == 2) *reinterpret_cast<h4*>(ex + plane + u * 512 + m * 16 + 8 * h) = p1;
        }
    }
    __syncthreads();

    const gcc wbase = (gcc)dp->wstream;
#pragma unroll 1
    for (int l = 0; l < CHAIN_LAYERS; ++l) {
        const int K16 = dp->L[l].K16, NTL = dp->L[l].NTL;
        // tiles of this wave: t = wn + NW c < NTL; its fragment stream starts behind those of the waves before it
        const int full = NTL / NW, rem = NTL - full * NW;
        const int cnt = full + (wn < rem ? 1 : 0);
        const int before = wn * full + (wn < rem ? wn : rem);
        const gcc wp = wbase + dp->L[l].w_off + ((size_t)before * K16 * NPL << 10) + lane * 16;

        v16f hi[CHAIN_CT], lo[CHAIN_CT];
#pragma unroll
        for (int c = 0; c < CHAIN_CT; ++c) { hi[c] = (v16f){0}; lo[c] = (v16f){0}; }

        // ---- K loop: no barriers, no staging.  PD register sets of weight fragments in flight.
        auto kloop = [&](auto cnt_tag) {
            constexpr int CNT = decltype(cnt_tag)::value;
            constexpr int STEP = CNT * NPL * 1024;               // bytes of one k-step of this wave's stream
            h8 w[PD][CNT][NPL];
            auto wload = [&](int set, int ks) {
                const gcc p = wp + (size_t)ks * STEP;
#pragma unroll
                for (int c = 0; c < CNT; ++c)
#pragma unroll
                    for (int pc = 0; pc < NPL; ++pc) w[set][c][pc] = *(gch8)(p + (c * NPL + pc) * 1024);
            };
            auto compute = [&](int set, int ks) {
                const h8 b0 = *reinterpret_cast<const h8*>(ex + ks * 1024 + lane * 16);
                h8 b1 = b0;
                if (NPL == 2) b1 = *reinterpret_cast<const h8*>(ex + plane + ks * 1024 + lane * 16);
#pragma unroll
                for (int c = 0; c < CNT; ++c) {
                    if (NPL == 2) {
                        lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[set][c][0], b1, lo[c], 0, 0, 0);          // w1 x2
                        lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[set][c][NPL - 1], b0, lo[c], 0, 0, 0);    // w2 x1
                    }
                    hi[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[set][c][0], b0, hi[c], 0, 0, 0);              // w1 x1
                }
            };
#pragma unroll
            for (int s = 0; s < PD; ++s)
                if (s < K16) wload(s, s);
            int ks0 = 0;
            for (; ks0 + 2 * PD <= K16; ks0 += PD) {             // steady state: branch-free
#pragma unroll
                for (int s = 0; s < PD; ++s) {
                    // the refill of a register set goes right behind the MFMAs that consumed it (left alone, hipcc sinks all
                    // the loads of an iteration to its end and waits for them at the top of the next: no run-ahead at all)
                    compute(s, ks0 + s);
                    wload(s, ks0 + s + PD);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; ks0 < K16; ks0 += PD) {
#pragma unroll
                for (int s = 0; s < PD; ++s) {
                    if (ks0 + s < K16) {
                        compute(s, ks0 + s);
                        if (ks0 + s + PD < K16) wload(s, ks0 + s + PD);
                    }
                }
            }
        };
        if (cnt == 3) kloop(std::integral_constant<int, 3>());
        else if (cnt == 2) kloop(std::integral_constant<int, 2>());
        else if (cnt == 1) kloop(std::integral_constant<int, 1>());

        __syncthreads();                                         // every wave has read the layer's input image

        // ---- epilogue
        const int boff = dp->L[l].bias_off;
        const bool leaky = dp->L[l].leaky != 0;
        const bool last = l == CHAIN_LAYERS - 1;
        const bool to_p = CHAIN == CHAIN_SPLIT && l == 1;
#pragma unroll
        for (int c = 0; c < CHAIN_CT; ++c) {
            if (c >= cnt) break;
            __builtin_amdgcn_sched_barrier(0);                   // one tile at a time: keeps the epilogue's registers bounded
            const int t = wn + NW * c;
            // the last layer of the mask chain also needs the residual and the spectrum it multiplies: requested up front
            v4f rv[4], mv[4];
            if (CHAIN == CHAIN_MASK && last) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int n0 = 32 * t + 8 * q + 4 * h;
                    n0 = n0 < dp->a8 ? n0 : 0;
                    rv[q] = *(gc4)((gcf)g.P + (size_t)row * g.ldp + dp->p_off + n0);
                    mv[q] = *(gc4)((gcf)g.Xmul + (size_t)row * g.ldm + dp->p_off + n0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 32 * t + 8 * q + 4 * h;           // first of this lane's 4 consecutive features
                const v4f bv = *reinterpret_cast<const v4f*>(&sbias[boff + n0]);
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = TERMS == 1 ? hi[c][4 * q + e] : hi[c][4 * q + e] + (1.f / 2048.f) * lo[c][4 * q + e];
                    v[e] = s + bv[e];
                    if (leaky) v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                }
                if (!last) {
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                    h4 p0, p1;
                    split4(v, p0, p1);
                    const int u = 4 * t + q;
                    *reinterpret_cast<h4*>(ex + u * 512 + m * 16 + 8 * h) = p0;
                    if (NPL == 2) *reinterpret_cast<h4*>(ex + plane + u * 512 + m * 16 + 8 * h) = p1;
                    if (to_p && row_ok && n0 < dp->a8) *(g4)((gf)g.P + (size_t)row * g.ldp + dp->p_off + n0) = v;
                } else if (CHAIN == CHAIN_SPLIT) {
                    if (row_ok && n0 < HID) *(g4)((gf)g.Z + (size_t)row * g.ldz + dp->z_off + n0) = v;
                } else {
                    if (row_ok && n0 < dp->a8) {
                        v += rv[q];                                                  // mask = residual + post(...)   bsrnn.py:425
                        if (g.tap) *(g4)((gf)g.tap + (size_t)row * g.ldt + dp->p_off + n0) = v;
                        *(g4)((gf)g.Y + (size_t)row * g.ldy + dp->p_off + n0) = v * mv[q];   // x * mask          bsrnn.py:441
                    }
                }
            }
        }
        __syncthreads();                                         // the next layer's input image is complete
    }
    // range guard (see gemm.hip): a finite operand beyond the fp16 range saturated its first piece
    if (amax > 65504.f && g.range_flag) *g.range_flag = 1;
}

int chain_blocks(const ChainLaunch& g)
{
    int n = 0, rows = 32;
    for (int cl = 0; cl < 4; ++cl, rows *= 2) n += g.n_cls[cl] * ((g.M + rows - 1) / rows);
    return n;
}

void launch_mlp_chain(const ChainLaunch& g, int chain, hipStream_t stream)
{
    const int nblk = g.M > 0 ? chain_blocks(g) : 0;
    if (nblk <= 0) return;
    dim3 grid(nblk), block(512);
    const bool one = gemm_mode() == GEMM_FP16;
    if (chain == CHAIN_SPLIT) {
        if (one) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_SPLIT, 1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_SPLIT, 3>), grid, block, 0, stream, g);
    } else {
        if (one) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_MASK, 1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_MASK, 3>), grid, block, 0, stream, g);
    }
}

}  // namespace bsrnn
