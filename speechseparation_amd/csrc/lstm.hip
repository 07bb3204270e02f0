// Dual-path recurrent kernels for gfx950: the band-axis BLSTM layer and the causal time-axis
// two-layer LSTM of NormRNNResidual (bsrnn.py:63-98, BandwiseLSTM :131-162, TimewiseLSTM
// :101-128).  torch gate order i,f,g,o:
//     c' = sigmoid(f) c + sigmoid(i) tanh(g),  h' = sigmoid(o) tanh(c').
// Two kernel families, selected by BSRNN_LSTM (kernels.h, LstmMode):
//   *_h2_kernel (default, "fp16x2")  gate products on v_mfma_f32_16x16x32_f16 with both operands as two fp16 pieces
//               (three MFMA terms, fp32 accumulation, fp32-level accuracy); cell state, activations and everything
//               that leaves the kernel are fp32;
//   *_kernel    ("f32")  exact fp32 on v_mfma_f32_16x16x4_f32 / 4x4x1.
// fc_in (Linear 64->64, no activation) is folded into W_ih of layer 0 on the host
// (api.hip: W' = W_ih W_in, b' = W_ih b_in + b_ih + b_hh, in double), the trailing fc + residual
// runs as one grouped-GEMM launch (gemm.hip, EPI_RES).
#include "kernels.h"
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef BAND_NLDS
#define BAND_NLDS 2                 // second-piece weight blocks of the 128-input layer kept in LDS (3: 226 VGPRs, 2: 242, 1: 254)
#endif
#ifndef BAND_NLDS64
#define BAND_NLDS64 0               // measurement: second-piece weight blocks of the 64-input layer kept in LDS (2: <= 168 VGPRs, three workgroups per CU)
#endif
#ifndef BAND_OCC64
#define BAND_OCC64 2                // measurement: waves per SIMD requested for the 64-input layer
#endif
#ifndef PART_DBG
#define PART_DBG 0                  // measurement only (tools/band_parts_check.hip): 1 no fc MFMAs, 2 no fclds array (fragments = garbage), 4 no tail
#endif
#ifndef BAND_NO_PLANES
#define BAND_NO_PLANES 0            // measurement only: 1 = fp32 instead of fp16 planes between the two band layers (A/B, tools/precision_dual_path.py)
#endif
#ifndef OVL_DBG
#define OVL_DBG 0                   // measurement only (overlapped dual path, tools/overlap_variants.sh): 1 a consumer tile sleeps ~20 us behind its wait, 2 the pair hand-over acquires at agent scope, 4 the time kernel's summed rows are stored write-through
#endif
#ifndef BAND_ABL
#define BAND_ABL 0                // measurement only (tools/lstm_h2_trace.hip): bit 1 no x staging, 2 no global h store, 4 no h publish, 8 no step barrier, 16 no MFMAs, 32 no transcendentals
#endif

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_sigmoid(float x)
{
    // v_exp_f32 + v_rcp_f32: ~1 ulp each; saturates correctly (exp -> inf gives 0, exp -> 0 gives 1)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x)
{
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;
}
__device__ __forceinline__ v4f exp2_4(const v4f x)
{
    if (BAND_ABL & 32) return x * 0.5f;
    return (v4f){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1]), __builtin_amdgcn_exp2f(x[2]), __builtin_amdgcn_exp2f(x[3])};
}
__device__ __forceinline__ v4f rcp4(const v4f x)
{
    if (BAND_ABL & 32) return x * 0.25f;
    return (v4f){__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1]), __builtin_amdgcn_rcpf(x[2]), __builtin_amdgcn_rcpf(x[3])};
}

// =====================================================================================
// Band-axis BLSTM layer.  grid = (ceil(N/16), 2 directions), 256 threads.
//
// A workgroup owns 16 sequences (one 16-row MFMA tile) of one direction for all L steps.
// Wave w owns hidden units [16w, 16w+16) for all four gates: four 16x16 accumulators
// (v_mfma_f32_16x16x4_f32), so i,f,g,o of one (sequence, unit) sit in the same lane and
// register index and the cell update needs no cross-lane traffic.  The wave's slice of
// [W_ih | W_hh] (64 gate rows x (IN+64)) stays in VGPRs for the whole launch (B operand);
// x_t and h_{t-1} are the A operand, read from LDS as ds_read_b128 with the k-permutation
// k(step s, quarter q) = 16*(s/4) + 4q + s%4 (host packs W in the same order).  LDS rows
// are padded to stride = 8 (mod 64) floats, which makes the b128 lane groups conflict-free.
// c lives in registers; h_t goes to LDS (next step's A operand) and from there, coalesced,
// to global.  x_{t+1} is prefetched into registers during the MFMAs of step t.
// =====================================================================================
#ifndef BAND_OCC
#define BAND_OCC 2            // waves per SIMD requested for the band kernel (2 workgroups per CU)
#endif
template <int IN>
__global__ __launch_bounds__(256, BAND_OCC) void band_lstm_kernel(const float* __restrict__ xin, float* __restrict__ hout,
                                                        const float* __restrict__ wpk, const float* __restrict__ bias,
                                                        int N, int L)
{
    constexpr int KT = IN + HID, NS = KT / 4;
    constexpr int SX = IN + 8, SH = HID + 8;
    constexpr int XV = IN / 64;                  // float4 per thread per x tile
    __shared__ __attribute__((aligned(16))) float xbuf[2][16 * SX];
    __shared__ __attribute__((aligned(16))) float hbuf[2][16 * SH];
    __shared__ float bias_s[4 * HID];            // this direction's b_ih + b_hh (from LDS every step: the 128-input layer has no registers to spare)

    const int dir = blockIdx.y;
    const int n0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;

    // resident weights: w[s][g] = Wcat[g*64 + 16*wave + l15][k(s, q)]
    float w[NS][4];
    {
        const float* wp = wpk + ((size_t)(dir * 4 + wave) * NS * 4) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) w[s][gte] = wp[(s * 4 + gte) * 64];
    }
    bias_s[tid] = bias[dir * 256 + tid];

    float c[4] = {0.f, 0.f, 0.f, 0.f};

    // staging maps
    const int xr_row[2] = {(tid * XV) / (IN / 4), (tid * XV + 1) / (IN / 4)};
    const int xr_c4[2] = {(tid * XV) % (IN / 4), (tid * XV + 1) % (IN / 4)};
    const int o_row = tid >> 4, o_c4 = tid & 15;

    auto xload = [&](int t, v4f* dst) {      // (native vectors: arrays of HIP's float4 structs stayed in scratch)
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            int row = n0 + xr_row[i];
            row = row < N ? row : N - 1;
            dst[i] = *reinterpret_cast<const v4f*>(xin + ((size_t)row * L + t) * IN + 4 * xr_c4[i]);
        }
    };
    auto xstore = [&](int buf, const v4f* src) {
#pragma unroll
        for (int i = 0; i < XV; ++i)
            *reinterpret_cast<v4f*>(&xbuf[buf][xr_row[i] * SX + 4 * xr_c4[i]]) = src[i];
    };

    {   // prologue: h_{-1} = 0, x of the first step
        *reinterpret_cast<float4*>(&hbuf[0][o_row * SH + 4 * o_c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        v4f x0[XV];
        xload(dir ? L - 1 : 0, x0);
        xstore(0, x0);
    }
    __syncthreads();

    for (int step = 0; step < L; ++step) {
        const int t = dir ? L - 1 - step : step;
        const int cur = step & 1, nxt = cur ^ 1;
        v4f xn[XV];
        const bool more = step + 1 < L;
        if (more) xload(dir ? t - 1 : t + 1, xn);

        v4f acc[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) { const float bsv = bias_s[gte * 64 + 16 * wave + l15]; acc[gte] = (v4f){bsv, bsv, bsv, bsv}; }

        const float* xa = &xbuf[cur][l15 * SX + 4 * q];
        const float* ha = &hbuf[cur][l15 * SH + 4 * q];
#pragma unroll
        for (int j = 0; j < IN / 16; ++j) {
            const v4f a = *reinterpret_cast<const v4f*>(xa + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
                    acc[gte] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[4 * j + e][gte], acc[gte], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < HID / 16; ++j) {
            const v4f a = *reinterpret_cast<const v4f*>(ha + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
                    acc[gte] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[IN / 4 + 4 * j + e][gte], acc[gte], 0, 0, 0);
        }

        // cell update; C/D layout of 16x16 MFMA: col (unit) = lane & 15, row (sequence) = 4*(lane>>4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = fg * c[r] + ig * gg;
            hbuf[nxt][(4 * q + r) * SH + 16 * wave + l15] = og * fast_tanh(c[r]);
        }
        if (more) xstore(nxt, xn);
        __syncthreads();
        if (n0 + o_row < N)
            *reinterpret_cast<float4*>(hout + ((size_t)(n0 + o_row) * L + t) * (2 * HID) + dir * HID + 4 * o_c4) =
                *reinterpret_cast<const float4*>(&hbuf[nxt][o_row * SH + 4 * o_c4]);
    }
}

// =====================================================================================
// Band-axis BLSTM layer, split-precision variant (fp16x2, see gemm.hip): the gate pre-activations are computed
// on the f16 matrix pipe, v_mfma_f32_16x16x32_f16, with both operands as two fp16 pieces (a ~ a1 + 2^-11 a2) and
// three terms  hi += x1 w1,  lo += x1 w2 + x2 w1,  pre = bias + hi + 2^-11 lo  -  fp32-level accuracy at 3/16 of
// the matrix-pipe time of the fp32 kernel above (which is bound by that pipe: 116 of 157 TFLOP/s).
// Same decomposition: 16 sequences x 1 direction per workgroup, wave w owns units [16w, 16w+16) of all four gates,
// so i,f,g,o of one cell share a lane; c and the fp32 arithmetic of the cell update are unchanged.
//   * Weights: B operand, lane (n = l & 15, kb = l >> 4) holds W[gate row of unit 16w+n][32 blk + 8 kb .. +7] of
//     each piece; packed on the host in exactly that order (one 16-byte load per lane and tile).  They stay in
//     VGPRs, except three of the six second-piece blocks of the 128-input layer, which would not fit (192 + 32
//     accumulator registers) and are read from LDS every step (12 conflict-free ds_read_b128 per wave).
//   * Activations: A operand, lane (m = l & 15, kb) holds x[m][32 blk + 8 kb .. +7]; x_t and h_t are kept in LDS
//     as two fp16 planes in [k / 8][sequence][8] order, which makes every fragment read a linear,
//     conflict-free ds_read_b128.  x is split by the staging threads, h by the lane that produced it.
//   * The input half of step t+1 (x_{t+1} W_ih, independent of h_t) is issued after the cell update of step t,
//     in front of the barrier that publishes h_t, so the matrix pipe works through the barrier skew.
// =====================================================================================
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_h2(float v, _Float16& p0, _Float16& p1)
{
    // The value is split AS the fp32 number it is: the empty asm hides its producer from the optimiser.  Without it the
    // compiler fused the producing multiply into ONE of the two conversions (v_fma_mixlo_f16 of the unrounded product for
    // the residual, v_cvt_pk_f16_f32 of the rounded product for the stored piece): in the rare double-rounding cases the
    // two first pieces differ by one fp16 ulp and hi + lo / 2048 is off by 2^-11 |v| (seen as 1e-5 errors of single
    // sequences when the planes went to the next layer; tools/precision_dual_path.py).
    asm("" : "+v"(v));
    p0 = (_Float16)v;
    p1 = (_Float16)__builtin_fmaf(-(float)p0, 2048.f, v * 2048.f);      // = 2048 (v - p0), exact before the conversion: one v_fma_mixlo_f16
}

// PART (layer 1 only): instead of h the launch writes THIS DIRECTION'S SHARE OF THE BLOCK'S fc (bsrnn.py:84: fc(h_fwd | h_bwd) =
// W[:, :64] h_fwd + W[:, 64:] h_bwd + b), hout[n][t][dir * 64 + f] = sum_k W_fc[f][dir * 64 + k] h_dir[n][t][k] (+ b[f] in the
// forward half), and the consumer (the time-axis kernel's staging) adds the two halves and the residual - the grouped-GEMM launch
// of the block's fc, its 100 MB of traffic and its 18 us are gone.  The product is formed where h_{t} is read back as the next
// step's B operand anyway: 6 more MFMAs per wave and step on the fragments already in registers, the fc's A fragments from LDS.
// LDS of one band layer: [x planes 2 slots][h planes 2 slots][second-piece weight blocks][bias][fc fragments]
template <int IN, bool PART>
struct BandLds {
    static constexpr int NLDS = IN == 128 ? BAND_NLDS : BAND_NLDS64;
    static constexpr int XPL = 0;
    static constexpr int HPL = XPL + 2 * 2 * IN * 16 * 2;
    static constexpr int W2 = HPL + 2 * 2 * HID * 16 * 2;
    static constexpr int BIAS = W2 + (NLDS ? 4 * NLDS * 4 * 64 : 1) * 16;
    static constexpr int FC = BIAS + (4 * HID + HID) * 4;          // gate biases [4][64] + the fc bias of the forward share [64]
    static constexpr int BYTES = FC + (PART && !(PART_DBG & 2) ? 4 * 2 * 2 * 64 : 1) * 16;
};

// One layer of one (tile of 16 sequences, direction): the body of band_lstm_h2_kernel and of both phases of band_pair_h2_kernel.
template <int IN, bool TRACE, bool PART>
__device__ __forceinline__ void band_layer_body(char* const lds, const int dir, const int tile,
                                                const float* __restrict__ xin, float* __restrict__ hout,
                                                const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                int N, int L, int* __restrict__ range_flag, unsigned long long* __restrict__ dbg,
                                                const uint4* __restrict__ wfc, const float* __restrict__ bfc)
{
    static_assert(!PART || IN == 2 * HID, "the fc share is formed by the second layer");
    // measurement only (TRACE, tools/lstm_h2_trace.hip): 100 MHz stamps per phase, accumulated per wave
    unsigned long long tp[5] = {0, 0, 0, 0, 0}, tq = 0;
    auto stamp = [&](int k) { if (TRACE) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); tp[k] += now - tq; tq = now; } };
    if (TRACE) tq = __builtin_amdgcn_s_memrealtime();
    constexpr int NBX = IN / 32, NBH = HID / 32, NB = NBX + NBH;
    constexpr bool PLANES_IN = IN == 2 * HID && !BAND_NO_PLANES;    // layer 1: x arrives as the fp16 planes layer 0 wrote
    constexpr bool PLANES_OUT = IN == HID && !BAND_NO_PLANES;       // layer 0: h leaves as fp16 planes (read by layer 1 only)
    // blocks whose second weight piece lives in LDS instead of VGPRs (counted from the last k block): the 128-input
    // layer would need 192 weight + 32 accumulator registers; with the two W_hh blocks in LDS it is 160 + 32 and
    // compiles without scratch at 242 VGPRs (measured with spills: 1.9 us of every 3.5 us step in reloads)
    constexpr int NLDS = IN == 128 ? BAND_NLDS : BAND_NLDS64;
    constexpr int XV = IN / 64;                  // 16-byte units per thread per x tile
    using LD = BandLds<IN, PART>;
    auto& xpl = *reinterpret_cast<_Float16 (*)[2][2][IN * 16]>(lds + LD::XPL);              // [slot][piece][k / 8][seq][8]
    auto& hpl = *reinterpret_cast<_Float16 (*)[2][2][HID * 16]>(lds + LD::HPL);
    uint4* const w2lds = reinterpret_cast<uint4*>(lds + LD::W2);
    float* const bias_lds = reinterpret_cast<float*>(lds + LD::BIAS);                        // this direction's b_ih + b_hh, [gate][unit]
    uint4* const fclds = reinterpret_cast<uint4*>(lds + LD::FC);                             // fc A fragments [wave][k block][piece][lane]
    const int n0 = tile * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;

    // resident weights: w[blk][gate][piece]
    h8v w[NB][4][2];
    {
        const uint4* wp = wpk + ((size_t)(dir * 4 + wave) * NB * 4 * 2) * 64 + lane;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    const uint4 v = wp[((b * 4 + gte) * 2 + pc) * 64];
                    if (pc == 1 && b >= NB - NLDS)
                        w2lds[((wave * NLDS + (b - (NB - NLDS))) * 4 + gte) * 64 + lane] = v;
                    else
                        w[b][gte][pc] = __builtin_bit_cast(h8v, v);
                }
    }
    bias_lds[tid] = bias[dir * 256 + tid];
    if (PART) {
        // W_fc as [4 tile][4 k block][2 piece][64 lane][8] (api.hip): tile = this wave's 16 output features, k blocks 2 dir + b
#pragma unroll
        for (int i = 0; i < 4; ++i) fclds[(PART_DBG & 2) ? 0 : (wave * 4 + i) * 64 + lane] = wfc[((size_t)(wave * 4 + 2 * dir) * 2 + i) * 64 + lane];
    }
    // The loads above are still in flight when the step loop starts, and the compiler's wait-count bookkeeping merges
    // that state into the loop (it waited for vmcnt(0), i.e. for the x prefetch of the same step, in front of the last
    // recurrent MFMA of every step).  A use of every resident register here makes the waits happen once, up front.
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) {
            asm volatile("" : "+v"(w[b][gte][0]));
            if (b < NB - NLDS) asm volatile("" : "+v"(w[b][gte][1]));
        }
    v4f cv = {0.f, 0.f, 0.f, 0.f};               // cell states of this lane's four (unit, sequence) cells

    // x staging, one 16-byte unit per thread and i < XV.
    //   layer 0 (fp32 rows of 64): (row, 4 consecutive columns) -> 8 bytes of each piece, split here;
    //   layer 1 (planes of 2 x 128 halves per (sequence, step)): (row, piece, 8 consecutive k) -> copied as they are.
    int xrow[XV], xcol[XV];                      // row of the tile, 16-byte column inside the row
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int u = tid + 256 * i;
        xrow[i] = u / (IN / 4);                  // IN / 4 units per row in both formats
        xcol[i] = u % (IN / 4);
    }
    auto xload = [&](int t, u4v* dst) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            int row = n0 + xrow[i];
            row = row < N ? row : N - 1;
            dst[i] = *reinterpret_cast<const u4v*>(xin + ((size_t)row * L + t) * IN + 4 * xcol[i]);
        }
    };
    float amax = 0.f;                            // range guard: largest |x| this thread staged (layer 0; |h| < 1)
    auto xstore = [&](int slot, const u4v* src) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            if (PLANES_IN) {
                const int piece = xcol[i] >> 4, k8 = xcol[i] & 15;
                *reinterpret_cast<u4v*>(&xpl[slot][piece][(k8 * 16 + xrow[i]) * 8]) = src[i];
            } else {
                const v4f v = __builtin_bit_cast(v4f, src[i]);
                h4v p0, p1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    amax = __builtin_fmaxf(amax, __builtin_fabsf(v[e]));
                    _Float16 a, b2;
                    split_h2(v[e], a, b2);                 // (no clamp, as in gemm.hip / mlp_chain.hip: the guard reports it)
                    p0[e] = a; p1[e] = b2;
                }
                const int o = ((xcol[i] >> 1) * 16 + xrow[i]) * 8 + (xcol[i] & 1) * 4;
                *reinterpret_cast<h4v*>(&xpl[slot][0][o]) = p0;
                *reinterpret_cast<h4v*>(&xpl[slot][1][o]) = p1;
            }
        }
    };
    auto tmap = [&](int step) { return dir ? L - 1 - step : step; };

    // Gate pre-activations as D^T = W X^T: the weights are the MFMA's A operand (row = unit), the activations its B operand
    // (column = sequence), so accumulator register r of lane (l15, q) is unit 16 wave + 4 q + r of sequence l15: the four
    // values a lane produces per step are CONSECUTIVE units of one sequence - one 8-byte LDS write per piece and one
    // 16-byte global store per step instead of four scattered ones each (the cell update is bound by instruction
    // issue: ~8 clocks per instruction of a lone wave, tools/mfma_valu_overlap.hip).
    // The first k block of a step starts from the bias (C operand read from LDS), the second accumulator from zero.
    v4f hi[4], lo[4];
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    const int frag = (q * 16 + l15) * 8;         // this lane's 16-byte unit inside a 32-deep block of a plane
    const float* const bias_l = &bias_lds[16 * wave + 4 * q];
    auto block_mfma = [&](const int b, const h8v a0, const h8v a1) {
        h8v w2[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte)
            w2[gte] = b >= NB - NLDS ? __builtin_bit_cast(h8v, w2lds[((wave * NLDS + (b - (NB - NLDS))) * 4 + gte) * 64 + lane]) : w[b][gte][1];
        if (BAND_ABL & 16) { asm volatile("" :: "v"(a0), "v"(a1), "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3])); return; }
#pragma unroll
        for (int gte = 0; gte < 4; ++gte)
            hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[b][gte][0], a0, b == 0 ? *reinterpret_cast<const v4f*>(bias_l + gte * HID) : hi[gte], 0, 0, 0);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[b][gte][0], a1, b == 0 ? zero4 : lo[gte], 0, 0, 0);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[gte], a0, lo[gte], 0, 0, 0);
    };
    // fragments one block ahead of the MFMAs; the scheduling fences keep the compiler from hoisting every block's
    // ds_reads to the top (register pressure: the 128-input layer sits at the 256-VGPR limit of 2 waves per SIMD)
    auto x_part = [&](int slot) {
        h8v a0 = *reinterpret_cast<const h8v*>(&xpl[slot][0][frag]);
        h8v a1 = *reinterpret_cast<const h8v*>(&xpl[slot][1][frag]);
#pragma unroll
        for (int b = 0; b < NBX; ++b) {
            h8v n0v = a0, n1v = a1;
            if (b + 1 < NBX) {
                n0v = *reinterpret_cast<const h8v*>(&xpl[slot][0][(b + 1) * 512 + frag]);
                n1v = *reinterpret_cast<const h8v*>(&xpl[slot][1][(b + 1) * 512 + frag]);
            }
            block_mfma(b, a0, a1);
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0v; a1 = n1v;
        }
    };
    // PART: the fc share of the h whose fragments are in registers (features 16 wave + 4 q + r of sequence l15, like the gates)
    v4f fhi = zero4, flo = zero4;
    const float* const fb_l = bias_lds + 4 * HID + 16 * wave + 4 * q;      // the share's bias from LDS (four registers the kernel does not have)
    if (PART && tid < HID) bias_lds[4 * HID + tid] = dir == 0 ? bfc[tid] : 0.f;
    auto fc_mfma = [&](const int b, const h8v a0, const h8v a1) {
        if (PART_DBG & 1) { fhi = *reinterpret_cast<const v4f*>(fb_l); flo = zero4; return; }
        const h8v f1 = __builtin_bit_cast(h8v, (PART_DBG & 2) ? w2lds[(wave * 4 + 2 * b) * 64 + lane] : fclds[(wave * 4 + 2 * b) * 64 + lane]);
        const h8v f2 = __builtin_bit_cast(h8v, (PART_DBG & 2) ? w2lds[(wave * 4 + 2 * b + 1) * 64 + lane] : fclds[(wave * 4 + 2 * b + 1) * 64 + lane]);
        fhi = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, a0, b == 0 ? *reinterpret_cast<const v4f*>(fb_l) : fhi, 0, 0, 0);
        flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, a1, b == 0 ? zero4 : flo, 0, 0, 0);
        flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f2, a0, flo, 0, 0, 0);
    };
    auto h_part = [&](int slot, const bool gates) {
        h8v a0 = *reinterpret_cast<const h8v*>(&hpl[slot][0][frag]);
        h8v a1 = *reinterpret_cast<const h8v*>(&hpl[slot][1][frag]);
#pragma unroll
        for (int b = 0; b < NBH; ++b) {
            h8v n0v = a0, n1v = a1;
            if (b + 1 < NBH) {
                n0v = *reinterpret_cast<const h8v*>(&hpl[slot][0][(b + 1) * 512 + frag]);
                n1v = *reinterpret_cast<const h8v*>(&hpl[slot][1][(b + 1) * 512 + frag]);
            }
            if (gates) block_mfma(NBX + b, a0, a1);
            __builtin_amdgcn_sched_barrier(0);
            if (PART) { fc_mfma(b, a0, a1); __builtin_amdgcn_sched_barrier(0); }
            a0 = n0v; a1 = n1v;
        }
    };

    {   // prologue: h_{-1} = 0 (slot 1), x of the first two steps, input half of step 0
        *reinterpret_cast<uint4*>(&hpl[1][0][0] + tid * 8) = make_uint4(0, 0, 0, 0);      // 2 pieces x 1024 halves = 256 x 16 B
        u4v x0[XV];
        xload(tmap(0), x0);
        xstore(0, x0);
        if (L > 1) { xload(tmap(1), x0); xstore(1, x0); }
    }
    // x two steps ahead travels in registers: loaded at the END of a step (behind that step's h stores in issue order),
    // written to LDS in the next one.  The wait in front of that write is then only ever for a load that is a whole step
    // old; with the load at the top of the step and the write behind the (predicated, i.e. branchy) h stores the compiler
    // had to wait for vmcnt(0), stores included.
    u4v xn[XV];              // (a native vector type: an array of HIP's uint4 structs stayed in scratch)
    if (L > 2) xload(tmap(2), xn);
    __syncthreads();
    x_part(0);
    // Step 0 overwrites slot 0 (x_0 -> x_2) as soon as ITS wave is through h_part and the cell, so every wave must be through
    // x_part(0) first.  (Missing until round 3: a wave that was late by more than a cell update - its weight loads came from HBM
    // instead of the Infinity Cache - read x_2 fragments for x_0; seen as rare wrong sequences, run to run, only in launches whose
    // buffers exceed the cache: 24 576 sequences and more.  tools/band_parts_check.hip.)
    if (L > 2) __syncthreads();
    stamp(0);

    // where this lane's four values of a step go: LDS planes (next step's B operand), global (next layer)
    const int hoff = ((2 * wave + (q >> 1)) * 16 + l15) * 8 + 4 * (q & 1);
    const bool row_ok = n0 + l15 < N;
    const size_t grow = (size_t)(row_ok ? n0 + l15 : 0) * L;
    for (int step = 0; step < L; ++step) {
        const int t = tmap(step);
        const bool more2 = step + 2 < L;

        h_part((step + 1) & 1, true);            // h_{step-1} lives in slot (step - 1) & 1
        if (PART && step > 0 && row_ok && !(BAND_ABL & 2))        // ... and its fc share leaves here
            *reinterpret_cast<v4f*>(hout + (grow + tmap(step - 1)) * (2 * HID) + dir * HID + 16 * wave + 4 * q) = fhi + flo * (1.f / 2048.f);
        stamp(1);

        {   // cell update on 4-vectors (the adds / multiplies become v_pk_*, only the 5 exp + 5 rcp per cell stay scalar)
            const v4f pi = hi[0] + lo[0] * (1.f / 2048.f), pf = hi[1] + lo[1] * (1.f / 2048.f);
            const v4f pg = hi[2] + lo[2] * (1.f / 2048.f), po = hi[3] + lo[3] * (1.f / 2048.f);
            const v4f ig = rcp4(1.0f + exp2_4(pi * -1.44269504f)), fg = rcp4(1.0f + exp2_4(pf * -1.44269504f));
            const v4f gg = 2.0f * rcp4(1.0f + exp2_4(pg * -2.88539008f)) - 1.0f, og = rcp4(1.0f + exp2_4(po * -1.44269504f));
            cv = fg * cv + ig * gg;
            const v4f hv4 = og * (2.0f * rcp4(1.0f + exp2_4(cv * -2.88539008f)) - 1.0f);
            if (more2 && !(BAND_ABL & 1)) xstore(step & 1, xn);         // slot of x_step, whose readers finished before the last barrier
            h4v p0, p1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 a, b2;
                split_h2(hv4[r], a, b2);
                p0[r] = a; p1[r] = b2;
            }
            if (!(BAND_ABL & 4)) {
                *reinterpret_cast<h4v*>(&hpl[step & 1][0][hoff]) = p0;
                *reinterpret_cast<h4v*>(&hpl[step & 1][1][hoff]) = p1;
            }
            if (row_ok && !(BAND_ABL & 2)) {
                if (PLANES_OUT) {
                    _Float16* const hp = reinterpret_cast<_Float16*>(hout) + ((grow + t) * 2) * (2 * HID) + dir * HID + 16 * wave + 4 * q;
                    *reinterpret_cast<h4v*>(hp) = p0;
                    *reinterpret_cast<h4v*>(hp + 2 * HID) = p1;
                } else if (!PART) {
                    *reinterpret_cast<v4f*>(hout + (grow + t) * (2 * HID) + dir * HID + 16 * wave + 4 * q) = hv4;
                }
            }
        }
        if (step + 3 < L) xload(tmap(step + 3), xn);
        stamp(2);
        if (step + 1 < L) x_part((step + 1) & 1);
        stamp(3);
        if (!(BAND_ABL & 8)) __syncthreads();
        stamp(4);
    }
    if (PART && !(PART_DBG & 4)) {               // the last step's h (published before the loop's last barrier)
        h_part((L - 1) & 1, false);
        if (row_ok && !(BAND_ABL & 2))
            *reinterpret_cast<v4f*>(hout + (grow + tmap(L - 1)) * (2 * HID) + dir * HID + 16 * wave + 4 * q) = fhi + flo * (1.f / 2048.f);
    }
    if (!(amax <= 65504.f) && range_flag) *range_flag = 1;
    if (TRACE && lane == 0 && blockIdx.x < 4 && blockIdx.y == 0) {
        unsigned long long* d = dbg + (blockIdx.x * 4 + wave) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = tp[k];
    }
}

// Workgroup -> (tile of 16 sequences, direction).  The two directions of a tile read the same x rows: in the 1-D grid of
// launch_band_lstm they are 8 workgroup ids apart - the same XCD (id % 8), dispatched together - so the second read of a row
// is served by that XCD's L2 instead of a second trip to memory (2-D grids, the measurement tools': direction = blockIdx.y).
__device__ __forceinline__ bool band_tile_of_block(int N, int& dir, int& tile, const int* __restrict__ order = nullptr)
{
    dir = blockIdx.y; tile = blockIdx.x;
    if (gridDim.y == 1) {
        const int w = blockIdx.x & 15;
        dir = w >> 3;
        tile = (blockIdx.x >> 4) * 8 + (w & 7);
    }
    // overlapped dual path: the dispatch ordinal picks its tile from a table sorted by the time the tile's frames leave the time-axis
    // launch running beside this one (the pair of a tile keeps its place: 8 workgroup ids apart); -1 pads the table to whole groups
    if (order) tile = order[tile];
    return tile >= 0 && tile * 16 < N;
}

template <int IN, bool TRACE = false, bool PART = false>
__global__ __launch_bounds__(256, (IN == 64 ? BAND_OCC64 : 2)) void band_lstm_h2_kernel(const float* __restrict__ xin, float* __restrict__ hout,
                                                              const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                              int N, int L, int* __restrict__ range_flag, unsigned long long* __restrict__ dbg,
                                                              const uint4* __restrict__ wfc = nullptr, const float* __restrict__ bfc = nullptr)
{
    __shared__ __attribute__((aligned(16))) char lds[BandLds<IN, PART>::BYTES];
    int dir, tile;
    if (!band_tile_of_block(N, dir, tile)) return;
    band_layer_body<IN, TRACE, PART>(lds, dir, tile, xin, hout, wpk, bias, N, L, range_flag, dbg, wfc, bfc);
}

// Both layers of a band block in ONE launch.  A workgroup runs layer 0 of its (tile, direction), publishes its half of the fp16
// planes and waits for its partner - the other direction of the same tile: 8 workgroup ids away, the same XCD, dispatched in the
// same breath (launch_band_lstm's grid) - then runs layer 1 on both halves.  The hand-over goes through L2: every thread fences its
// plane stores, one thread releases flags[2 tile + dir] = its old value + 1 (the pair's two flags count the launches over this tile in
// lockstep: nothing to clear, nothing passed by value, so the launch can sit in a replayed graph) and polls the partner's flag, bounded; the reads of layer 1 come behind an acquire fence (L1 invalidated).  No deadlock: workgroups
// are dispatched in id order, a waiting workgroup's partner is at most 8 ids behind it, and everything dispatched before a waiting
// pair either is a complete pair or waits for partners that are dispatched before any later workgroup - so slots always free up.
// One launch, one weight prologue (layer 1's 196 KB are requested before the wait) and one tail less per block; the planes are read
// back from L2 while they are hot.
template <bool PART>
__global__ __launch_bounds__(256, 2) void band_pair_h2_kernel(const float* __restrict__ z, float* hb0, float* __restrict__ hb1,
                                                              const uint4* __restrict__ w0, const float* __restrict__ b0,
                                                              const uint4* __restrict__ w1, const float* __restrict__ b1,
                                                              int N, int L, int* __restrict__ range_flag,
                                                              const uint4* __restrict__ wfc, const float* __restrict__ bfc,
                                                              int* flags, int sabotage, OvlConsumer ovl, int* zero_words, int zero_n)
{
    constexpr int B0 = BandLds<HID, false>::BYTES, B1 = BandLds<2 * HID, PART>::BYTES;
    __shared__ __attribute__((aligned(16))) char lds[B0 > B1 ? B0 : B1];
    __shared__ int partner_ok;
    // overlapped dual path: the progress words of the launches BEHIND this one are zeroed here, in stream order in front of their
    // producers and consumers (a memset of their own cost the stream 5 us plus two launch gaps)
    if (zero_words && blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_words[i] = 0;
    int dir, tile;
    if (!band_tile_of_block(N, dir, tile, ovl.order)) return;
    if (ovl.prog) {
        // launched beside the time-axis launch that writes z (kernels.h, OvlConsumer): wait until this tile's 16 frame rows have left it
        if (threadIdx.x == 0) {
            const int m_last = tile * 16 + 15 < N ? tile * 16 + 15 : N - 1;
            partner_ok = ovl_wait_rows(ovl.prog, tile * 16, m_last, ovl.T, L, 0, L - 1, ovl.spin_limit, ovl.base, ovl.wg_shift) ? 1 : 0;
        }
        __syncthreads();
        if (!partner_ok) {                        // (value 5: api.hip runs the call again launch after launch and stops overlapping)
            if (threadIdx.x == 0 && range_flag) *range_flag = 5;
            return;
        }
        __syncthreads();                          // (partner_ok is written again below)
        if (OVL_DBG & 1)
            for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(127);
    }
    band_layer_body<HID, false, false>(lds, dir, tile, z, hb0, w0, b0, N, L, range_flag, nullptr, nullptr, nullptr);
    // Hand-over through the XCD's L2, which both workgroups share: a store is counted out of vmcnt when L2 has it, so vmcnt(0) is
    // all the release this needs (an agent-scope release fence would write the whole L2 back for the sake of other XCDs: measured,
    // 2.6x slower); the flag carries the XCC id so that a placement that breaks the assumption is reported, not computed with.
    __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0): this thread's plane stores are in L2
    __syncthreads();                              // ... every thread's; and nobody reads layer 0's LDS any more
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;      // HW_REG_XCC_ID[3:0]
        // the pair's two flags count the launches that covered this tile, in lockstep: mine + 1 is this launch's number for both
        const int epoch = ((__hip_atomic_load(&flags[2 * tile + dir], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 4) + 1) & 0x7ffffff;
        __hip_atomic_store(&flags[2 * tile + dir], (epoch << 4) | (sabotage ? (xcc ^ 8) : xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, v;
        while (((v = __hip_atomic_load(&flags[2 * tile + (dir ^ 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) break;       // (seconds: the partner never came - reported, not waited for)
        }
        partner_ok = (v >> 4) == epoch && (v & 15) == xcc;
    }
    __syncthreads();
    if (!partner_ok) {                            // (value 4: api.hip runs the call again with one launch per layer and stops pairing)
        if (threadIdx.x == 0 && range_flag) *range_flag = 4;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (orders the plane loads below behind the poll)
    if (OVL_DBG & 2) { asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    else
    asm volatile("buffer_inv sc0" ::: "memory");                 // the partner's planes from L2, not from a line this CU's L1 may hold
    band_layer_body<2 * HID, false, PART>(lds, dir, tile, hb0, hb1, w1, b1, N, L, range_flag, nullptr, wfc, bfc);
}

bool band_pair_enabled()
{
    static const bool on = [] { const char* e = getenv("BSRNN_BAND_PAIR"); return !(e && !strcmp(e, "0")); }();      // A/B: 0 = one launch per layer
    return on && lstm_mode() == LSTM_FP16X2 && !force_f32();
}
// both layers of a band block as one launch (band_pair_h2_kernel); fc16 / fcb as launch_band_lstm's (shares of the fc) or null
void launch_band_pair(const float* z, float* hb0, float* hb1, const void* w0pk16, const float* bias0, const void* w1pk16, const float* bias1,
                      int N, int L, int* range_flag, hipStream_t stream, const void* fc16, const float* fcb, int* flags, const OvlConsumer* ovlp,
                      int* zero_words, int zero_n, hipEvent_t done)
{
    if (N <= 0 || L <= 0) return;
    OvlConsumer ovl = {nullptr, 0, 0, nullptr, 0, 2};
    if (ovlp) ovl = *ovlp;
    const dim3 grid((((N + 15) / 16 + 7) / 8) * 16), block(256);
    // test hook (tests/test_gpu_edges.py): BSRNN_BAND_PAIR=mismatch makes every workgroup publish a wrong XCC id, as if its partner sat on
    // another XCD - the launch reports it (guard value 4) and the context falls back to one launch per layer
    static const int sabotage = [] { const char* e = getenv("BSRNN_BAND_PAIR"); return (e && !strcmp(e, "mismatch")) ? 1 : 0; }();
    // done: an event the launch itself signals when it completes (the completion signal of its own dispatch packet: hipExtLaunchKernel) - a
    // separate hipEventRecord behind the launch is a marker packet of its own and cost the stream a 5-7 us gap (overlapped dual path)
    if (fc16)
        hipExtLaunchKernelGGL((band_pair_h2_kernel<true>), grid, block, 0, stream, nullptr, done, 0, z, hb0, hb1, (const uint4*)w0pk16, bias0, (const uint4*)w1pk16, bias1, N, L,
                              range_flag, (const uint4*)fc16, fcb, flags, sabotage, ovl, zero_words, zero_n);
    else
        hipExtLaunchKernelGGL((band_pair_h2_kernel<false>), grid, block, 0, stream, nullptr, done, 0, z, hb0, hb1, (const uint4*)w0pk16, bias0, (const uint4*)w1pk16, bias1, N, L,
                              range_flag, (const uint4*)nullptr, (const float*)nullptr, flags, sabotage, ovl, zero_words, zero_n);
}

// BSRNN_BAND_FC = part (default: the second band layer writes the two directions' shares of the block's fc, the time-axis launch adds
// them and the residual while it stages its input) | gemm (the block's fc + residual as a grouped-GEMM launch, as in rounds 1-2)
bool band_fc_in_parts()
{
    static const bool on = [] { const char* e = getenv("BSRNN_BAND_FC"); return !(e && !strcmp(e, "gemm")); }();
    // (the shares are formed by the pair launch only: the second layer alone with the shares sits at the edge of 256 VGPRs - as a
    //  kernel of its own it compiled with four spilled registers, inside the pair kernel with none - and is not shipped)
    return on && band_pair_enabled() && time_lstm_fuses_fc();
}

void launch_band_lstm(const float* xin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      int N, int L, int IN, int* range_flag, hipStream_t stream)
{
    if (N <= 0 || L <= 0) return;
    dim3 grid((N + 15) / 16, 2), block(256);
    if (lstm_mode() == LSTM_FP16X2 && !force_f32()) {
        static const bool paired = [] { const char* e = getenv("BSRNN_BAND_GRID"); return !(e && !strcmp(e, "2d")); }();   // A/B: 2d = one direction after the other
        if (paired) grid = dim3((((N + 15) / 16 + 7) / 8) * 16);      // both directions of eight tiles per 16 consecutive workgroups
        if (IN == 64)
            hipLaunchKernelGGL(band_lstm_h2_kernel<64>, grid, block, 0, stream, xin, hout, (const uint4*)wpk16, bias, N, L, range_flag, (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL(band_lstm_h2_kernel<128>, grid, block, 0, stream, xin, hout, (const uint4*)wpk16, bias, N, L, range_flag, (unsigned long long*)nullptr);
        return;
    }
    if (IN == 64)
        hipLaunchKernelGGL(band_lstm_kernel<64>, grid, block, 0, stream, xin, hout, wpk, bias, N, L);
    else
        hipLaunchKernelGGL(band_lstm_kernel<128>, grid, block, 0, stream, xin, hout, wpk, bias, N, L);
}

int lstm_mode()
{
    static const int mode = [] {
        const char* e = getenv("BSRNN_LSTM");
        if (!e || !*e || !strcmp(e, "fp16x2")) return (int)LSTM_FP16X2;
        if (!strcmp(e, "f32")) return (int)LSTM_F32;
        fprintf(stderr, "bsrnn: unknown BSRNN_LSTM='%s' (f32 | fp16x2), using fp16x2\n", e);
        return (int)LSTM_FP16X2;
    }();
    return mode;
}

// =====================================================================================
// Time-axis LSTM (2 layers, unidirectional, causal), state in / state out.
// grid = ceil(R*K / 4), 512 threads: waves 0-3 run layer 0 at step s, waves 4-7 run layer 1
// at step s-1 (software pipeline across layers), one workgroup barrier per step; the two groups
// order their MFMA and cell-update phases differently so that they overlap on the shared SIMDs.
//
// The recurrence is latency bound (T sequential steps), so a workgroup takes only FOUR
// sequences and the gates are computed with v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4x4,
// A = 4 sequences (the same for every block), B = 4 gate columns per block, i.e. 64 gate
// columns per wave = 16 hidden units x {i,f,g,o}.  The WEIGHTS are the MFMA's A operand (lane 4b+i
// holds the row of gate i of unit 16w+b) and the activations its B operand (lane 4b+j supplies
// sequence j), so D[i][j] puts the four gates i,f,g,o of ONE (unit, sequence) into the four
// accumulator registers of ONE lane: the cell update needs no cross-lane traffic and each of
// the 64 lanes updates exactly one cell (a first version with the operands the other way round
// needed 16 DPP broadcasts and 4x redundant cell math: 44 % of every step went into VALU issue,
// which starves next to a stream of 8-cycle MFMAs).  [W_ih | W_hh] (128 k) of the wave's 64 rows
// and the cell state c are register resident.
// =====================================================================================
constexpr int TCH = 8;        // x steps staged per chunk
constexpr int TS = HID + 4;   // LDS row stride (68 floats: 4 rows hit 4 distinct b128 slots)

__global__ __launch_bounds__(512) void time_lstm_kernel(const float* __restrict__ zin, float* __restrict__ hout,
                                                        const float* __restrict__ wpk, const float* __restrict__ bias,
                                                        const float* __restrict__ state_in, float* __restrict__ state_out,
                                                        int R, int T, int K, unsigned long long* __restrict__ dbg)
{
    __shared__ __attribute__((aligned(16))) float xbuf[2][TCH][4 * TS];
    __shared__ __attribute__((aligned(16))) float h0buf[2][4 * TS];
    __shared__ __attribute__((aligned(16))) float h1buf[2][4 * TS];

    const int N = R * K;
    const int n0 = blockIdx.x * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int layer = wave >> 2, w4 = wave & 3;
    const int b = lane >> 2, j = lane & 3;
    const int unit = 16 * w4 + b;
    const size_t tstride = (size_t)K * HID;

    // sequence j of this workgroup (A row / the row this lane stores)
    const int nj_raw = n0 + j;
    const int nj = nj_raw < N ? nj_raw : N - 1;
    const size_t base_j = ((size_t)(nj / K) * T * K + (nj % K)) * HID;

    float w[2 * HID];
    {
        const float* wp = wpk + ((size_t)(layer * 4 + w4) * 2 * HID) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 2 * HID; ++k) w[k] = wp[(size_t)k * 64];
    }
    // this lane's weight row is gate j of `unit` (A operand); its cell is (unit, sequence j)
    v4f bs4;
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) bs4[gte] = bias[layer * 256 + gte * 64 + unit];

    // c of (sequence j, this unit); h_{-1} into LDS slot 1
    float c = state_in ? state_in[((size_t)(2 + layer) * N + nj) * HID + unit] : 0.f;
    {
        const float hinit = state_in ? state_in[((size_t)layer * N + nj) * HID + unit] : 0.f;
        float* hb = layer ? h1buf[1] : h0buf[1];
        hb[j * TS + unit] = hinit;
    }

    // x chunk staging: 8 steps x 4 sequences x 16 float4 = 512 float4, one per thread
    const int xs_t = tid >> 6, xs_i = (tid >> 4) & 3, xs_c4 = tid & 15;
    size_t xs_base;
    {
        int ni = n0 + xs_i; ni = ni < N ? ni : N - 1;
        xs_base = ((size_t)(ni / K) * T * K + (ni % K)) * HID + 4 * xs_c4;
    }
    auto chunk_load = [&](int chunk) -> float4 {
        int t = chunk * TCH + xs_t;
        t = t < T ? t : T - 1;
        return *reinterpret_cast<const float4*>(zin + xs_base + (size_t)t * tstride);
    };
    auto chunk_store = [&](int chunk, float4 v) {
        *reinterpret_cast<float4*>(&xbuf[chunk & 1][xs_t][xs_i * TS + 4 * xs_c4]) = v;
    };
    chunk_store(0, chunk_load(0));
    float4 xnext = make_float4(0.f, 0.f, 0.f, 0.f);
    float hsel = 0.f, csel = 0.f;
    __syncthreads();

    // Gate accumulators (4 independent MFMA chains).  Both layer-groups run the same phase order in
    // lockstep: 128 MFMAs (input half + recurrent half), then the cell.  VALU issue starves next to a
    // stream of 8-cycle MFMAs from the SIMD's other wave (measured: ~20 cycles per VALU instruction), so
    // the two waves of a SIMD should be in their (short) VALU phases at the same time:
    //   layer 0, iteration s:  bias + W_x.x_s      + W_h.h0_{s-1} -> cell(s)   -> publish h0_s
    //   layer 1, iteration s:  bias + W_x.h0_{s-1} + W_h.h1_{s-2} -> cell(s-1) -> publish h1_{s-1}
    // h_t of either layer lives in LDS slot t & 1; one workgroup barrier per iteration.
    v4f a0 = bs4, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
    auto gemv64 = [&](const float* src, const int wofs) {
        // all 16 reads first: a 4x4x1 MFMA lasts 8 cycles, so reads trickled in between groups of
        // four MFMAs leave the chain waiting on LDS latency (measured: MFMA pipe 33 % busy)
        v4f av[HID / 4];
#pragma unroll
        for (int m = 0; m < HID / 4; ++m) av[m] = *reinterpret_cast<const v4f*>(src + 4 * m);
        __builtin_amdgcn_sched_barrier(0);        // keep hipcc from sinking the reads back between the MFMAs
#pragma unroll
        for (int m = 0; m < HID / 4; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[wofs + 4 * m + 0], av[m][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[wofs + 4 * m + 1], av[m][1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[wofs + 4 * m + 2], av[m][2], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[wofs + 4 * m + 3], av[m][3], a3, 0, 0, 0);
        }
    };
    auto reset_acc = [&]() {
        a0 = bs4;
        a1 = a2 = a3 = (v4f){0.f, 0.f, 0.f, 0.f};
    };
    auto cell = [&](int t) {
        const v4f gsum = (a0 + a1) + (a2 + a3);          // i, f, g, o pre-activations of (unit, sequence j)
        const float ig = fast_sigmoid(gsum[0]);
        const float fg = fast_sigmoid(gsum[1]);
        const float gg = fast_tanh(gsum[2]);
        const float og = fast_sigmoid(gsum[3]);
        c = fg * c + ig * gg;
        hsel = og * fast_tanh(c);
        csel = c;
        float* hb = layer ? h1buf[t & 1] : h0buf[t & 1];
        hb[j * TS + unit] = hsel;
        if (layer && nj_raw < N) hout[base_j + (size_t)t * tstride + unit] = hsel;
    };
    // measurement only (dbg != nullptr): 100 MHz stamps per phase, accumulated per wave
    unsigned long long tp[4] = {0, 0, 0, 0}, tq = 0;
    auto stamp = [&](int k) { if (dbg) { const unsigned long long n = __builtin_amdgcn_s_memrealtime(); tp[k] += n - tq; tq = n; } };
    if (dbg) tq = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s <= T; ++s) {
        const int chunk = s / TCH, sin = s % TCH;
        const bool have_next = (chunk + 1) * TCH < T;
        if (sin == 0 && have_next) xnext = chunk_load(chunk + 1);

        stamp(3);                                                         // barrier + loop overhead
        const int t = layer ? s - 1 : s;                                  // the time step this wave computes
        if (layer ? (s >= 1) : (s < T)) {
            reset_acc();
            gemv64(layer ? &h0buf[t & 1][j * TS] : &xbuf[chunk & 1][sin][j * TS], 0);          // input half
            stamp(2);
            gemv64(layer ? &h1buf[(t + 1) & 1][j * TS] : &h0buf[(t + 1) & 1][j * TS], HID);    // recurrent half, h_{t-1}
            stamp(0);
            cell(t);
            stamp(1);
        }
        if (sin == TCH - 1 && have_next) chunk_store(chunk + 1, xnext);
        __syncthreads();
    }
    if (dbg && lane == 0 && blockIdx.x < 4) {
        unsigned long long* d = dbg + (blockIdx.x * 8 + wave) * 4;
        d[0] = tp[0]; d[1] = tp[1]; d[2] = tp[2]; d[3] = tp[3];
    }

    if (state_out && nj_raw < N) {
        state_out[((size_t)layer * N + nj) * HID + unit] = hsel;           // h_{T-1}
        state_out[((size_t)(2 + layer) * N + nj) * HID + unit] = csel;     // c_{T-1}
    }
}

// =====================================================================================
// Time-axis LSTM, split-precision variant (fp16x2 on v_mfma_f32_16x16x32_f16, as the band kernel above).
// The fp32 kernel's step is bound by the 4x4x1 MFMA stream (2 waves x 128 MFMAs x 9.5 cycles per SIMD and
// step).  Same decomposition (4 sequences per workgroup, waves 0-3 layer 0, waves 4-7 layer 1, one barrier per
// step), but the matrix work is organised around the 16-row tile of the f16 MFMA:
//   * recurrent half (h_{t-1} W_hh, the sequential part): A = activations with row 4j = sequence j (the rows in
//     between repeat it and are ignored), B = weights, tile g = gate g of the wave's 16 units (column n = unit
//     16w+n), resident in VGPRs.  Accumulator register 0 of lane (n, q = lane >> 4) is gate g of cell
//     (unit 16w+n, sequence q): every lane owns exactly one cell and its four gates, no cross-lane traffic.
//   * input half (x_t W_ih for layer 0, h0_t W_ih for layer 1) does not depend on the layer's own recurrence,
//     so FOUR time steps are batched into one MFMA group: A row 4j + r = (sequence j, step 4g + r) fills all
//     16 rows, and register r of lane (n, q) is then the input pre-activation of this lane's own cell at step
//     4g + r.  One group of 24 MFMAs per four steps instead of 24 per step; with the two-MFMA recurrent form
//     below: 22 instead of 48 MFMAs per wave and step (matrix-pipe floor 0.34 instead of 0.73 us per step).  Layer 1 therefore runs four steps behind
//     layer 0 (its input h0_{4g..4g+3} is complete after layer 0's step 4g+3); h0 lives in a ring of 8 steps.
// x_t / h_t live in LDS as two fp16 planes in [k / 8][sequence][8] order per step (fragment reads are
// conflict-free ds_read_b128; steps are 1088 bytes apart so that the four steps of a batched read hit
// different banks).
// =====================================================================================
constexpr int TSTEP = 2 * 4 * HID + 32;       // halves per step slot in LDS: two pieces of [8][4][8] + 64 bytes of skew

template <bool TRACE = false>
__global__ __launch_bounds__(512) void time_lstm_h2_kernel(const float* __restrict__ zin, float* __restrict__ hout,
                                                           const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                           const float* __restrict__ state_in, float* __restrict__ state_out,
                                                           int R, int T, int K, int* __restrict__ range_flag, unsigned long long* __restrict__ dbg)
{
    unsigned long long tp[5] = {0, 0, 0, 0, 0}, tq = 0;       // measurement only, as in the band kernel
    auto stamp = [&](int k) { if (TRACE) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); tp[k] += now - tq; tq = now; } };
    __shared__ __attribute__((aligned(16))) _Float16 xpl[2 * TCH * TSTEP];     // [chunk slot][step in chunk]
    __shared__ __attribute__((aligned(16))) _Float16 h0pl[8 * TSTEP];          // ring: h0_t in slot t & 7
    __shared__ __attribute__((aligned(16))) _Float16 h1pl[2 * TSTEP];          // h1_t in slot t & 1

    const int N = R * K;
    const int n0 = blockIdx.x * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int layer = wave >> 2, w4 = wave & 3;
    const int n = lane & 15, q = lane >> 4;
    const int unit = 16 * w4 + n;
    const size_t tstride = (size_t)K * HID;

    // this lane's cell: (unit, sequence q)
    const int nq_raw = n0 + q;
    const int nq = nq_raw < N ? nq_raw : N - 1;
    const size_t base_q = ((size_t)(nq / K) * T * K + (nq % K)) * HID;

    h8v w[4][4][2];                               // [k block: 0,1 input half, 2,3 recurrent half][gate][piece]
    {
        const uint4* wp = wpk + ((size_t)(layer * 4 + w4) * 4 * 4 * 2) * 64 + lane;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) w[b][gte][pc] = __builtin_bit_cast(h8v, wp[((b * 4 + gte) * 2 + pc) * 64]);
    }
    float bs[4];
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) bs[gte] = bias[layer * 256 + gte * 64 + unit];

    const int hoff = ((unit >> 3) * 4 + q) * 8 + (unit & 7);      // where this lane's h goes inside a piece
    float c = state_in ? state_in[((size_t)(2 + layer) * N + nq) * HID + unit] : 0.f;
    {
        const float hinit = state_in ? state_in[((size_t)layer * N + nq) * HID + unit] : 0.f;
        _Float16 p0, p1;
        split_h2(hinit, p0, p1);
        _Float16* hb = layer ? &h1pl[1 * TSTEP] : &h0pl[7 * TSTEP];   // h_{-1}: slot (-1) & 1 = 1, (-1) & 7 = 7
        hb[hoff] = p0;
        hb[4 * HID + hoff] = p1;
    }

    // x chunk staging: 8 steps x 4 sequences x 16 float4 = 512 float4, one per thread, split on the way into LDS
    const int xs_t = tid >> 6, xs_i = (tid >> 4) & 3, xs_c4 = tid & 15;
    size_t xs_base;
    {
        int ni = n0 + xs_i; ni = ni < N ? ni : N - 1;
        xs_base = ((size_t)(ni / K) * T * K + (ni % K)) * HID + 4 * xs_c4;
    }
    auto chunk_load = [&](int chunk) -> float4 {
        int t = chunk * TCH + xs_t;
        t = t < T ? t : T - 1;
        return *reinterpret_cast<const float4*>(zin + xs_base + (size_t)t * tstride);
    };
    float amax = 0.f;                            // range guard
    auto chunk_store = [&](int chunk, float4 v) {
        const float f[4] = {v.x, v.y, v.z, v.w};
        h4v p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
            _Float16 a, b2;
            split_h2(f[e], a, b2);
            p0[e] = a; p1[e] = b2;
        }
        _Float16* dst = &xpl[((chunk & 1) * TCH + xs_t) * TSTEP + ((xs_c4 >> 1) * 4 + xs_i) * 8 + (xs_c4 & 1) * 4];
        *reinterpret_cast<h4v*>(dst) = p0;
        *reinterpret_cast<h4v*>(dst + 4 * HID) = p1;
    };
    chunk_store(0, chunk_load(0));
    float4 xnext = make_float4(0.f, 0.f, 0.f, 0.f);
    float hsel = 0.f, csel = 0.f;
    __syncthreads();

    v4f hi[4], lo[4];
    float pin[4][4];                              // [gate][step in group]: bias + input half of this lane's cell
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    // A fragments inside a 32-deep block of a piece: row l & 15 of the recurrent form carries sequence (l & 15) >> 2;
    // in the batched form it carries (sequence (l & 15) >> 2, step (l & 15) & 3) of the group
    const int afrag = (q * 4 + (n >> 2)) * 8;
    const int bstep = n & 3;
    auto mfma_block = [&](const h8v a0, const h8v a1, const int wb) {
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, w[wb][gte][0], hi[gte], 0, 0, 0);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, w[wb][gte][1], lo[gte], 0, 0, 0);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w[wb][gte][0], lo[gte], 0, 0, 0);
    };
    // recurrent half: src = one step slot.  Only rows 4j of the 16-row tile carry a sequence, so row 4j+1 is given
    // the SECOND piece of the same sequence: one MFMA against w1 (first weight piece) then yields a1 w1 in register
    // 0 and a2 w1 in register 1 of the owning lane, a second MFMA against w2 yields a1 w2 in register 0 - two
    // MFMAs per (block, gate) instead of three, and one fragment read per block instead of two.
    const int rfrag = afrag + ((n & 3) == 1 ? 4 * HID : 0);
    auto recurrent = [&](const _Float16* src) {
        h8v a[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const h8v*>(&src[b * 128 + rfrag]);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) { hi[gte] = zero4; lo[gte] = zero4; }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[2 + b][gte][0], hi[gte], 0, 0, 0);
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[2 + b][gte][1], lo[gte], 0, 0, 0);
        }
    };
    // input half of four consecutive steps; `src` = slot of the group's first step, the lane's step is src + bstep slots
    auto batched_input = [&](const _Float16* src) {
        const _Float16* mine = src + bstep * TSTEP;
        h8v a0[2], a1[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            a0[b] = *reinterpret_cast<const h8v*>(&mine[b * 128 + afrag]);
            a1[b] = *reinterpret_cast<const h8v*>(&mine[4 * HID + b * 128 + afrag]);
        }
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) { hi[gte] = zero4; lo[gte] = zero4; }
#pragma unroll
        for (int b = 0; b < 2; ++b) mfma_block(a0[b], a1[b], b);
#pragma unroll
        for (int gte = 0; gte < 4; ++gte)
#pragma unroll
            for (int r = 0; r < 4; ++r) pin[gte][r] = bs[gte] + (hi[gte][r] + lo[gte][r] * (1.f / 2048.f));
    };
    auto cell = [&](int t, int r) {
        // hi[g] = {a1 w1, a2 w1, -, -}, lo[g] = {a1 w2, -, -, -} of this lane's cell (see recurrent())
        const float ig = fast_sigmoid(pin[0][r] + (hi[0][0] + (hi[0][1] + lo[0][0]) * (1.f / 2048.f)));
        const float fg = fast_sigmoid(pin[1][r] + (hi[1][0] + (hi[1][1] + lo[1][0]) * (1.f / 2048.f)));
        const float gg = fast_tanh(pin[2][r] + (hi[2][0] + (hi[2][1] + lo[2][0]) * (1.f / 2048.f)));
        const float og = fast_sigmoid(pin[3][r] + (hi[3][0] + (hi[3][1] + lo[3][0]) * (1.f / 2048.f)));
        c = fg * c + ig * gg;
        hsel = og * fast_tanh(c);
        csel = c;
        _Float16 p0, p1;
        split_h2(hsel, p0, p1);
        _Float16* hb = layer ? &h1pl[(t & 1) * TSTEP] : &h0pl[(t & 7) * TSTEP];
        hb[hoff] = p0;
        hb[4 * HID + hoff] = p1;
        if (layer && nq_raw < N) hout[base_q + (size_t)t * tstride + unit] = hsel;
    };
    auto xslot = [&](int t) { return &xpl[(((t / TCH) & 1) * TCH + (t % TCH)) * TSTEP]; };

    if (layer == 0) batched_input(xslot(0));      // group 0
    if (TRACE) tq = __builtin_amdgcn_s_memrealtime();
    // iteration s: layer 0 computes step s, layer 1 computes step s - 4
    for (int s4 = 0; s4 < T + 4; s4 += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s4 + r;
            if (s >= T + 4) break;
            const int chunk = s / TCH, sin = s % TCH;
            const bool have_next = (chunk + 1) * TCH < T;
            if (sin == 0 && have_next) xnext = chunk_load(chunk + 1);
            // x of the next chunk goes to LDS six steps after its load, at the TOP of the step: the wait in front of it then only
            // covers operations a whole step old (behind this step's h store the compiler has to wait for vmcnt(0), store included).
            // Its first reader is the batched input at the end of the next step, behind this step's barrier.
            if (sin == TCH - 2 && have_next) chunk_store(chunk + 1, xnext);
            stamp(4);
            if (layer == 0) {
                if (s < T) {
                    recurrent(&h0pl[((s + 7) & 7) * TSTEP]);            // h0_{s-1}
                    stamp(1);
                    cell(s, r);
                    stamp(2);
                }
                if (r == 3 && s + 1 < T) { batched_input(xslot(s + 1)); stamp(0); }     // next group: x_{s+1 .. s+4}
            } else {
                const int t = s - 4;
                if (r == 0 && t >= 0 && t < T) { batched_input(&h0pl[(t & 7) * TSTEP]); stamp(0); }   // h0_{t .. t+3}
                if (t >= 0 && t < T) {
                    recurrent(&h1pl[((t + 1) & 1) * TSTEP]);             // h1_{t-1}
                    stamp(1);
                    cell(t, r);
                    stamp(2);
                }
            }
            __syncthreads();
            stamp(3);
        }
    }
    if (!(amax <= 65504.f) && range_flag) *range_flag = 1;
    if (TRACE && lane == 0 && blockIdx.x < 4) {
        unsigned long long* d = dbg + (blockIdx.x * 8 + wave) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = tp[k];
    }

    if (state_out && nq_raw < N) {
        state_out[((size_t)layer * N + nq) * HID + unit] = hsel;           // h_{T-1}
        state_out[((size_t)(2 + layer) * N + nq) * HID + unit] = csel;     // c_{T-1}
    }
}

// =====================================================================================
// Time-axis LSTM, split precision, SIXTEEN waves per workgroup and NO per-step workgroup barrier (round 3; the kernel
// the fp16x2 mode launches).  Same decomposition and the same arithmetic as time_lstm_h2_kernel above (4 sequences per
// workgroup, one cell per lane, recurrent half with the second piece on row 4j + 1, input half of four steps batched into
// one 16-row tile) - h1 and the carried state are bit-identical to it - but the work is spread over four roles of four
// waves each, one wave of every role on every SIMD:
//     M0, M1  "main" waves of layer 0 / layer 1: ONLY the serial chain of a step - h_{t-1} fragments from LDS, 16
//             recurrent MFMAs against the resident W_hh, the cell, h_t to LDS.  W_hh alone is 64 VGPRs, so a main wave fits
//             the 128 registers of a 1024-thread workgroup.
//     H0, H1  "helper" waves: everything that is NOT on that chain - the batched input half (x W_ih / h0 W_ih of a group of
//             four steps, bias + result to an LDS buffer the main wave reads), the staging of x (global -> fp16 pieces ->
//             LDS), and for layer 1 the block's trailing fc + residual (FUSE, below).
// What was measured on the way (profiles/r03_time_lstm.txt): in the 8-wave kernels every wave's own instruction stream WAS
// the step (a lone wave issues one instruction per 4-6 clocks: ~200 instructions = 0.75 us per step, whether the two layers
// ran their phases in lockstep or in counter-phase); with the roles above but one s_barrier per step the step was still
// 0.71 us, because a barrier makes every step as long as its slowest role and the helpers' MFMAs queue behind the main
// waves'.  So the roles are decoupled: the only synchronisation inside the launch are monotonic counters in LDS
// (sync[]: steps completed per layer, groups published per layer, ...), added to by lane 0 of a wave behind its LDS
// writes (one wave's LDS operations execute in order) and polled with ds_read by whoever needs the data.  The two layers'
// chains then run at their own pace and fill each other's gaps on the shared matrix pipe and vector issue; layer 1 trails
// layer 0 by what the data dependences require (its group's h0 must be complete before its input half can be batched).
// Rings in LDS: h0 and h1 16 steps each, x 2 chunks of 8 steps, input-half buffers 2 groups per layer; every overwrite
// waits on the counter that says its last reader is done.
// FUSE: the block's fc + residual (NormRNNResidual, bsrnn.py:84-86: out = fc(h1) + x) is computed by H1, batched over the
// four steps of a group like the input half (6 MFMAs per group and wave, wave w = output features 16 w .. 16 w + 15): h1
// never goes to HBM and the separate grouped-GEMM launch (15 us, 75 MB per block) is gone; `hout` is then the block's output.
// =====================================================================================
#ifndef TIME_ABL
#define TIME_ABL 0                // measurement only (tools/time_lstm_v3_bench.hip): 1 main waves do not wait for the input half, 2 helpers issue no MFMAs, 4 main waves do not wait for h(t-1), 8 no helper work at all
#endif
#ifndef TIME_OPT
#define TIME_OPT 3                // 1: main waves at raised priority; 2: h(t-1) fragments requested together with the poll of the step counter
#endif
constexpr int H0RING = 16, H1RING = 16;        // steps of h0 / h1 kept in LDS
enum { SY_DONE0 = 0, SY_DONE1, SY_PIN0, SY_PIN1, SY_FC, SY_X, SY_ABORT, SY_COUNT = 8 };
constexpr int SPIN_LIMIT = 1 << 21;            // polls of one wait before the workgroup gives up (seconds; a step takes ~10 polls)

// spin until the LDS counter has reached `target` (monotonic; every wave that waits is in the same workgroup as the waves
// that add, so all of them are resident).  The memory clobbers keep the compiler from moving LDS accesses across the wait.
// Every spin is bounded: a wait that never ends (it cannot, by construction) sets the abort word, which ends every other
// wait of the workgroup as well, and the launch reports it through the range-guard word (value 3) instead of hanging.
// (the poll is an explicit ds_read_b32: through a volatile generic pointer hipcc emitted a FLAT load with sc0 sc1 and a wait for
// vmcnt(0), i.e. for every global load and store the wave had in flight)
__device__ __forceinline__ int lds_peek(const int* p)
{
    int v;
    const unsigned a = (unsigned)(size_t)p;      // low half of the generic address of a __shared__ object = its LDS byte address
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_wait_ge(int* sync, int which, int target)
{
    int spins = 0;
    while (lds_peek(&sync[which]) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) __hip_atomic_store(&sync[SY_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (spins > 64 && lds_peek(&sync[SY_ABORT])) break;
    }
}
// one arrival of this wave: behind everything the wave has written to LDS so far
__device__ __forceinline__ void lds_arrive(int* cnt, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// PART: the block's input is not zin alone but zin + part[.., 0:64] + part[.., 64:128] - the residual and the two directions' shares
// of the preceding band block's fc (band_lstm_h2_kernel<128, ., true> wrote them, [sequence-position][2][64]); the staging wave adds
// them on the way into LDS and the fc wave adds the same three rows as ITS residual, so that block's fc launch does not exist.
template <bool FUSE, bool TRACE = false, bool PART = false>
__global__ __launch_bounds__(1024) void time_lstm_h2w_kernel(const float* __restrict__ zin, float* __restrict__ hout,
                                                             const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                             const uint4* __restrict__ wfc, const float* __restrict__ bfc,
                                                             const float* __restrict__ state_in, float* __restrict__ state_out,
                                                             int R, int T, int K, int* __restrict__ range_flag, unsigned long long* __restrict__ dbg,
                                                             const float* __restrict__ part = nullptr, int* ovl_resident = nullptr, int* ovl_prog = nullptr, int ovl_base = 0)
{
    unsigned long long tp[4] = {0, 0, 0, 0}, tq = 0;          // measurement only
    auto stamp = [&](int k) { if (TRACE) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); tp[k] += now - tq; tq = now; } };
    __shared__ __attribute__((aligned(16))) _Float16 xpl[2 * TCH * TSTEP];     // [chunk slot][step in chunk]
    __shared__ __attribute__((aligned(16))) _Float16 h0pl[H0RING * TSTEP];     // h0_t in slot t & 15
    __shared__ __attribute__((aligned(16))) _Float16 h1pl[H1RING * TSTEP];     // h1_t in slot t & 15
    __shared__ __attribute__((aligned(16))) float pinb[2 * 2 * 4 * 4 * 256];   // [layer][group parity][step in group][gate][cell]: bias + input half
    __shared__ __attribute__((aligned(16))) uint4 wflds[FUSE ? 4 * 4 * 64 : 1];
    __shared__ int sync[SY_COUNT];            // the counters (and the abort word) of lds_wait_ge / lds_arrive
    __shared__ float hb_lds[PART ? 512 : 1];  // PART: the helpers' biases [layer][gate][unit]
    __shared__ int outc[8];                   // overlapped dual path: storing waves that have drained group f, in slot f & 7 (monotonic: 4 per use)

    const int N = R * K;
    const int n0 = blockIdx.x * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                 // wave-uniform: the roles below are scalar branches
    const int role = wave >> 2, w4 = wave & 3;
    const int layer = role & 1;
    const int n = lane & 15, q = lane >> 4;
    const int unit = 16 * w4 + n;
    const int cellid = w4 * 64 + lane;
    const size_t tstride = (size_t)K * HID;
    const int G = (T + 3) >> 2;                  // groups of four steps

    // this lane's cell: (unit, sequence q)
    const int nq_raw = n0 + q;
    const int nq = nq_raw < N ? nq_raw : N - 1;
    const size_t base_q = ((size_t)(nq / K) * T * K + (nq % K)) * HID;
    const int afrag = (q * 4 + (n >> 2)) * 8;    // A fragment inside a 32-deep block of a piece: row l & 15 = (sequence (l & 15) >> 2, ...)
    const int bstep = n & 3;                     // ... batched form: step (l & 15) & 3 of the group
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    if (tid < SY_COUNT) sync[tid] = 0;
    if (tid >= 64 && tid < 72) outc[tid - 64] = 0;
    // overlapped dual path (kernels.h, OvlProducer): this workgroup is on the chip - the launch that consumes its output is let go when all are
    if (PART && tid == 0 && ovl_resident) __hip_atomic_fetch_add(ovl_resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (may be a signal word the command processor polls)
    if (TRACE) tq = __builtin_amdgcn_s_memrealtime();

    // The two role families are laid out as "helpers: ...; return;  main waves: ..." and not as if / else: with a join behind both, the
    // structurizer keeps the values of the path laid out second alive through the first one's loops (a wave runs only one of them, but
    // the compiler sees entry -> helpers -> join -> main as a path) - a dozen registers the 128-VGPR budget of 16 waves per CU does not have.
    auto finish = [&]() {
        if (lds_peek(&sync[SY_ABORT]) && range_flag) *range_flag = 3;
        if (TRACE && lane == 0 && blockIdx.x < 4) {
            unsigned long long* d = dbg + (blockIdx.x * 16 + wave) * 4;
    #pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = tp[k];
        }
    };
    if (role >= 2) {
        // ------------------------------------------------------------------ helper waves: input halves, x staging, fc
        h8v w[2][4][2];                           // W_ih (fc_in folded for layer 0): [k block][gate][piece]
        {
            const uint4* wp = wpk + ((size_t)(layer * 4 + w4) * 4 * 4 * 2) * 64 + lane;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) w[b][gte][pc] = __builtin_bit_cast(h8v, wp[((b * 4 + gte) * 2 + pc) * 64]);
        }
        float bs[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) {
            bs[gte] = bias[layer * 256 + gte * 64 + unit];
            if (PART) hb_lds[layer * 256 + gte * 64 + unit] = bs[gte];      // (read back after the workgroup's first barrier)
        }
        float bf = 0.f;
        if (FUSE && layer) {                      // the fc matrix as B fragments [k block][piece] (column = output feature `unit`): used once per
            const uint4* wp = wfc + ((size_t)w4 * 2 * 2) * 64 + lane;       // group, so it lives in LDS; a wave reads back what it wrote itself
#pragma unroll
            for (int i = 0; i < 4; ++i) wflds[(w4 * 4 + i) * 64 + lane] = wp[i * 64];
            bf = bfc[unit];
        }
        // input half of the four steps of `group` (A rows = (sequence, step), `src` = slot of the group's first step): per gate
        // 6 MFMAs, then bias + result to the LDS buffer the main wave of this cell reads, [step][gate][cell]
        auto input_half = [&](const _Float16* src, int group) {
            if (TIME_ABL & 8) return;
            const _Float16* mine = src + bstep * TSTEP;
            h8v a0[2], a1[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                a0[b] = *reinterpret_cast<const h8v*>(&mine[b * 128 + afrag]);
                a1[b] = *reinterpret_cast<const h8v*>(&mine[4 * HID + b * 128 + afrag]);
            }
            float* dst = pinb + layer * 8192 + (group & 1) * 4096 + cellid;
            float bsg[4];                         // PART: the biases come from LDS per group instead of living in four registers across the
#pragma unroll                                    // loop - the kernel has none to spare (no scratch: see H0)
            for (int gte = 0; gte < 4; ++gte) bsg[gte] = PART ? hb_lds[layer * 256 + gte * 64 + unit] : bs[gte];
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) {
                v4f ghi = zero4, glo = zero4;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (TIME_ABL & 2) { ghi[0] += (float)a0[b][0] * (float)w[b][gte][0][0]; glo[1] += (float)a1[b][1] * (float)w[b][gte][1][1]; continue; }
                    ghi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[b], w[b][gte][0], ghi, 0, 0, 0);
                    glo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[b], w[b][gte][1], glo, 0, 0, 0);
                    glo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[b], w[b][gte][0], glo, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e * 1024 + gte * 256] = bsg[gte] + (ghi[e] + glo[e] * (1.f / 2048.f));
            }
        };
        if (!layer) {
            // ---------------- H0: x staging (its 256 threads: 8 steps x 4 sequences x 16 float4 = 512 float4, two per thread) and
            // the input half of layer 0.  Chunk c (steps 8 c ...) lives in slot c & 1; it is requested a whole chunk ahead,
            // stored once every H0 wave has published the groups that read the slot's previous content, and read once every
            // H0 wave has stored its share.
            const int xs_t = cellid >> 6, xs_i = (cellid >> 4) & 3, xs_c4 = cellid & 15;      // thread -> (step 0-3 [+4], sequence, float4)
            size_t xs_base;
            {
                int ni = n0 + xs_i; ni = ni < N ? ni : N - 1;
                xs_base = ((size_t)(ni / K) * T * K + (ni % K)) * HID + 4 * xs_c4;
            }
            float amax = 0.f;                    // range guard
            struct XRows { float4 z[2]; };
            auto chunk_load = [&](int chunk, XRows& v) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    int t = chunk * TCH + xs_t + 4 * hf;
                    t = t < T ? t : T - 1;
                    v.z[hf] = *reinterpret_cast<const float4*>(zin + xs_base + (size_t)t * tstride);
                }
            };
            auto chunk_store = [&](int chunk, const XRows& v) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float f[4] = {v.z[hf].x, v.z[hf].y, v.z[hf].z, v.z[hf].w};
                    h4v p0, p1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
                        _Float16 a, b2;
                        split_h2(f[e], a, b2);
                        p0[e] = a; p1[e] = b2;
                    }
                    _Float16* dst = &xpl[((chunk & 1) * TCH + xs_t + 4 * hf) * TSTEP + ((xs_c4 >> 1) * 4 + xs_i) * 8 + (xs_c4 & 1) * 4];
                    *reinterpret_cast<h4v*>(dst) = p0;
                    *reinterpret_cast<h4v*>(dst + 4 * HID) = p1;
                }
            };
            auto xslot = [&](int t) { return &xpl[(((t / TCH) & 1) * TCH + (t % TCH)) * TSTEP]; };
            if (PART) {
                // PART stages group by group instead of chunk by chunk: three rows per (sequence, step) - residual and the two fc shares -
                // would be 24 registers held across the input half (the kernel then spills: 1024 threads leave 128 VGPRs per wave);
                // one group's rows are 12.  A group's four steps are requested one group ahead, stored (summed) at the top of the
                // next iteration once every wave has published the group that used the same slots (four groups = 16 steps earlier).
                struct GRows { float4 z, pf, pb; };
                auto rows_of = [&](int g) { int t = 4 * g + xs_t; t = t < T ? t : T - 1; return xs_base + (size_t)t * tstride; };
                auto load_z = [&](int g, GRows& v) { v.z = *reinterpret_cast<const float4*>(zin + rows_of(g)); };
                auto load_shares = [&](int g, GRows& v) {
                    const float* pp = part + 2 * (rows_of(g) - 4 * xs_c4) + 4 * xs_c4;
                    v.pf = *reinterpret_cast<const float4*>(pp);
                    v.pb = *reinterpret_cast<const float4*>(pp + HID);
                };
                auto group_store = [&](int g, const GRows& v) {
                    const float f[4] = {(v.z.x + v.pf.x) + v.pb.x, (v.z.y + v.pf.y) + v.pb.y, (v.z.z + v.pf.z) + v.pb.z, (v.z.w + v.pf.w) + v.pb.w};
                    h4v p0, p1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
                        _Float16 a, b2;
                        split_h2(f[e], a, b2);
                        p0[e] = a; p1[e] = b2;
                    }
                    _Float16* dst = xslot(4 * g + xs_t) + ((xs_c4 >> 1) * 4 + xs_i) * 8 + (xs_c4 & 1) * 4;
                    *reinterpret_cast<h4v*>(dst) = p0;
                    *reinterpret_cast<h4v*>(dst + 4 * HID) = p1;
                    // ... and to the OUTPUT buffer, where the fc wave (H1) picks it up as its residual two groups later and then
                    // overwrites it with the block's result: one read of the three rows instead of two.  Same workgroup, same CU:
                    // the store is complete (vmcnt(0) in front of this wave's PIN0 arrival below) long before the chain H0 -> M0 ->
                    // H1 of LDS counters lets H1 ask for it, and nobody has read that line before (no stale copy in the L1).
                    if (n0 + xs_i < N && 4 * g + xs_t < T) {
                        if (OVL_DBG & 4) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) __hip_atomic_store((float __attribute__((address_space(1)))*)(hout + rows_of(g) + e), f[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else
                        *reinterpret_cast<float4*>(hout + rows_of(g)) = make_float4(f[0], f[1], f[2], f[3]);
                    }
                };
                // Only the residual row (4 registers) is in flight across an input half; the two share rows of the next group are
                // requested BEHIND it and land during the waits at the top of the next iteration: the kernel must stay inside 128
                // VGPRs without scratch (with spills - the weight fragments, reloaded in the loops - a call beside a second
                // process's first kernels came out a few ulp different once in ~50: tools/busy_start_stress.py).
                GRows xr;
                load_z(0, xr); load_shares(0, xr); group_store(0, xr);
                if (G > 1) { load_z(1, xr); load_shares(1, xr); }
                __syncthreads();
                for (int g = 0; g < G; ++g) {
                    if (g) {
                        if (g >= 4) lds_wait_ge(sync, SY_PIN0, 4 * (g - 3));   // group g - 4 (the same slots) is published by every wave
                        group_store(g, xr);
                        lds_arrive(&sync[SY_X], lane);
                        if (g + 1 < G) load_z(g + 1, xr);
                        lds_wait_ge(sync, SY_X, 4 * g);                    // every wave has stored its share of group g
                    }
                    if (g >= 2) lds_wait_ge(sync, SY_DONE0, 16 * (g - 1));   // layer 0 has finished group g - 2 (same buffer)
                    stamp(0);
                    input_half(xslot(4 * g), g);
                    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's row stores of the group are in L2
                    lds_arrive(&sync[SY_PIN0], lane);
                    if (g && g + 1 < G) load_shares(g + 1, xr);
                    stamp(2);
                }
            } else {
            XRows xnext;
            chunk_load(0, xnext); chunk_store(0, xnext);
            if (TCH < T) chunk_load(1, xnext);
            __syncthreads();
            for (int g = 0; g < G; ++g) {
                const int ch = g >> 1;
                if (g && !(g & 1)) {                                  // first group of chunk ch >= 1
                    lds_wait_ge(sync, SY_PIN0, 4 * (2 * ch - 2));    // the groups that read chunk ch - 2 (same slot) are published by every wave
                    chunk_store(ch, xnext);
                    lds_arrive(&sync[SY_X], lane);
                    if ((ch + 1) * TCH < T) chunk_load(ch + 1, xnext);
                    lds_wait_ge(sync, SY_X, 4 * ch);                 // every wave has stored its share of chunk ch
                }
                if (g >= 2) lds_wait_ge(sync, SY_DONE0, 16 * (g - 1));   // layer 0 has finished group g - 2 (same buffer)
                stamp(0);
                input_half(xslot(4 * g), g);
                if (g) lds_wait_ge(sync, SY_PIN0, 4 * g);             // (SY_PIN0 is a sum over the four waves: no wave two groups ahead of another, see H1)
                lds_arrive(&sync[SY_PIN0], lane);
                stamp(2);
            }
            }
            if (!(amax <= 65504.f) && range_flag) *range_flag = 1;
        } else {
            // ---------------- H1: input half of layer 1 (A = h0 of the group, complete once layer 0 has finished its last step)
            // FUSE: fc + residual of a group of layer 1 in two stages, so that H1 never waits for anything but its own gates: the
            // residual rows of group f are REQUESTED in iteration f + 1 (fc_request) and the group is finished in iteration f + 2
            // (fc_finish: A rows = (sequence, step) of h1 as in the batched input half, 6 MFMAs, epilogue), when the gate that
            // iteration waits for anyway - layer 1 has finished group f - says its h1 is complete
            float xres[4] = {0.f, 0.f, 0.f, 0.f};
            const float* const res_src = PART ? hout : zin;     // PART: the staging wave left the summed row (residual + fc shares) in the output buffer
            auto fc_request = [&](int f) {
                const int ffirst = 4 * f, nst = T - ffirst < 4 ? T - ffirst : 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t row = base_q + (size_t)(ffirst + (e < nst ? e : nst - 1)) * tstride;
                    xres[e] = res_src[row + unit];
                }
            };
            // Overlapped dual path: the output rows of group f are in memory (every store of them write-through, this wave's drained by the
            // vmcnt(0) here; the last of the four storing waves to say so for f tells the consumers - each wave works through the groups in
            // order and drains all its older stores with it, so f + 1 published groups mean groups 0 .. f are complete whichever wave
            // published them).  Called one group late, where the wave waits for its residual loads anyway: no extra stall on H1.
            const bool pub = FUSE && PART && ovl_prog != nullptr;       // (only the parts flow overlaps: api.hip)
            // The consumers work in tiles of 16 frames and more, so progress is published every FOURTH group (and at the end): one drain per 16
            // steps instead of one per 4 keeps H1's stalls on its write-through stores off the chain (each drain covers all the wave's older stores).
            auto publish = [&](int f) {
                if ((f & 3) != 3 && f != G - 1) return;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&outc[(f >> 2) & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                // (absolute value, epoch in the upper bits, atomic MAX: the four-group publishes of a workgroup may come from different waves out of order)
                if ((old & 3) == 3 && lane == 0) __hip_atomic_fetch_max(ovl_prog + blockIdx.x, ovl_base + f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            auto fc_finish = [&](int f) {
                if (pub && f >= 1) publish(f - 1);
                const int ffirst = 4 * f, nst = T - ffirst < 4 ? T - ffirst : 4;
                const _Float16* mine = &h1pl[((ffirst & (H1RING - 1)) + bstep) * TSTEP];
                v4f fhi = zero4, flo = zero4;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const h8v f0 = *reinterpret_cast<const h8v*>(&mine[b * 128 + afrag]);
                    const h8v f1 = *reinterpret_cast<const h8v*>(&mine[4 * HID + b * 128 + afrag]);
                    const h8v wf0 = __builtin_bit_cast(h8v, wflds[(w4 * 4 + b * 2 + 0) * 64 + lane]);
                    const h8v wf1 = __builtin_bit_cast(h8v, wflds[(w4 * 4 + b * 2 + 1) * 64 + lane]);
                    fhi = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf0, fhi, 0, 0, 0);
                    flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf1, flo, 0, 0, 0);
                    flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, wf0, flo, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                // one block's fragments at a time (registers)
                }
                lds_arrive(&sync[SY_FC], lane);                       // the group's h1 slots are free again
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nst && nq_raw < N) {
                        const float v = ((fhi[e] + flo[e] * (1.f / 2048.f)) + bf) + xres[e];
                        float* const dst = hout + base_q + (size_t)(ffirst + e) * tstride + unit;
                        if (pub) __hip_atomic_store((float __attribute__((address_space(1)))*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store_dword sc1
                        else *dst = v;
                    }
            };
            __syncthreads();
            for (int g = 0; g < G; ++g) {
                const int last = 4 * g + 4 < T ? 4 * g + 4 : T;
                lds_wait_ge(sync, SY_DONE0, 4 * last);               // h0 of the group complete
                if (g >= 2) lds_wait_ge(sync, SY_DONE1, 16 * (g - 1));   // layer 1 has finished group g - 2 (same buffer; its h1 is complete)
                stamp(0);
                input_half(&h0pl[((4 * g) & (H0RING - 1)) * TSTEP], g);
                // SY_PIN1 and SY_FC are SUMS over the four H1 waves, and their waiters (M1: this group's input half is published; M0 / M1:
                // a ring slot has been read) conclude "every wave has done group g" from "sum >= 4 (g + 1)".  That only holds while no wave
                // is two groups ahead of another - and nothing else ties the H1 waves to each other: one of them stalled for a few
                // microseconds on global memory (its residual loads, its output stores) while the other three went on gave M1 a group
                // whose input half - or an h1 slot whose fc read - was the stalled wave's old one: 16 units of four sequences slightly
                // wrong from a group boundary on.  Seen once in ~100 calls with write-through output stores (overlapped dual path,
                // tools/overlap_probe.py), and the likely cause of round 3's "result depended on a second process starting" (DESIGN 4e).
                // So a wave publishes group g only when all four have published g - 1: skew <= 1 group, for which the sums are exact.
                if (g) lds_wait_ge(sync, SY_PIN1, 4 * g);
                lds_arrive(&sync[SY_PIN1], lane);
                stamp(2);
                if (FUSE) {
                    if (g >= 2) fc_finish(g - 2);
                    if (g >= 1) fc_request(g - 1);
                }
                stamp(1);
            }
            if (FUSE) {                           // the last two groups
                if (G >= 2) { lds_wait_ge(sync, SY_DONE1, 16 * (G - 1)); fc_finish(G - 2); }
                fc_request(G - 1);
                lds_wait_ge(sync, SY_DONE1, 4 * T);
                fc_finish(G - 1);
                if (pub) publish(G - 1);
            }
        }
        finish();
        return;
    }
    {
        // ------------------------------------------------------------------ main waves: the serial chain
        h8v w[2][4][2];                           // W_hh: [k block][gate][piece]
        {
            const uint4* wp = wpk + ((size_t)(layer * 4 + w4) * 4 * 4 * 2) * 64 + lane;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) w[b][gte][pc] = __builtin_bit_cast(h8v, wp[(((2 + b) * 4 + gte) * 2 + pc) * 64]);
        }
        const int hoff = ((unit >> 3) * 4 + q) * 8 + (unit & 7);      // where this lane's h goes inside a piece
        _Float16* const ring = layer ? h1pl : h0pl;
        const int rmask = layer ? H1RING - 1 : H0RING - 1;
        float c = state_in ? state_in[((size_t)(2 + layer) * N + nq) * HID + unit] : 0.f;
        {
            const float hinit = state_in ? state_in[((size_t)layer * N + nq) * HID + unit] : 0.f;
            _Float16 p0, p1;
            split_h2(hinit, p0, p1);
            _Float16* hb = &ring[rmask * TSTEP];                      // h_{-1}
            hb[hoff] = p0;
            hb[4 * HID + hoff] = p1;
        }
        float hsel = 0.f, csel = 0.f;
        // Only rows 4j of the 16-row tile carry a sequence, so row 4j + 1 is given the SECOND piece of the same sequence: one
        // MFMA against w1 yields a1 w1 in register 0 and a2 w1 in register 1 of the owning lane, a second one against w2
        // yields a1 w2 in register 0 - two MFMAs per (block, gate) instead of three, one fragment read per block.
        const int rfrag = afrag + ((n & 3) == 1 ? 4 * HID : 0);
        const float* const pin_l = pinb + layer * 8192 + cellid;
        const int my_done = layer ? SY_DONE1 : SY_DONE0, my_pin = layer ? SY_PIN1 : SY_PIN0;
        // whoever reads this layer's ring besides the layer itself: H1 batches h0 and (fc) h1
        const int reader = layer ? SY_FC : SY_PIN1;
        const bool has_reader = layer ? FUSE : true;
        const int rsteps = layer ? H1RING : H0RING;
        __syncthreads();                          // h_{-1}, counters, x chunk 0 (the helpers publish group 0 behind it)
        if (TIME_OPT & 1) __builtin_amdgcn_s_setprio(2);     // the chain before the helpers wherever both could issue
        for (int t = 0; t < T; ++t) {
            if ((t & 3) == 0) {
                if (!(TIME_ABL & 1)) lds_wait_ge(sync, my_pin, 4 * ((t >> 2) + 1));              // this group's input half is published
                if (has_reader && t >= rsteps) lds_wait_ge(sync, reader, 4 * (((t - rsteps) >> 2) + 1));   // the slots this group overwrites have been read
            }
            stamp(3);
            const _Float16* src = &ring[((t - 1) & rmask) * TSTEP];
            h8v a[2];
            if (TIME_OPT & 2) {
                // h_{t-1} complete (all four waves)?  The counter and the two fragments are requested together - LDS operations of a
                // wave execute in order, so fragments that follow a counter value >= 4 t are complete - and requested again if not:
                // one LDS round trip per step instead of two.
                const unsigned ca = (unsigned)(size_t)&sync[my_done], fa = (unsigned)(size_t)&src[rfrag];
                int spins = 0;
                for (;;) {
                    int cv;
                    u4v f0, f1;
                    asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:256\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(cv), "=&v"(f0), "=&v"(f1) : "v"(ca), "v"(fa) : "memory");
                    a[0] = __builtin_bit_cast(h8v, f0); a[1] = __builtin_bit_cast(h8v, f1);
                    if ((TIME_ABL & 4) || __builtin_amdgcn_readfirstlane(cv) >= 4 * t) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) __hip_atomic_store(&sync[SY_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (spins > 64 && lds_peek(&sync[SY_ABORT])) break;
                }
            } else {
                if (!(TIME_ABL & 4)) lds_wait_ge(sync, my_done, 4 * t);                          // h_{t-1} complete (all four waves)
#pragma unroll
                for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const h8v*>(&src[b * 128 + rfrag]);
            }
            stamp(2);
            const float* const pin_t = pin_l + (((t >> 2) & 1) * 4 + (t & 3)) * 1024;
            const float pin_i = pin_t[0], pin_f = pin_t[256], pin_g = pin_t[512], pin_o = pin_t[768];
            v4f hi[4], lo[4];
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) { hi[gte] = zero4; lo[gte] = zero4; }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][0], hi[gte], 0, 0, 0);
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][1], lo[gte], 0, 0, 0);
            }
            // hi[g] = {a1 w1, a2 w1, -, -}, lo[g] = {a1 w2, -, -, -} of this lane's cell
            const float ig = fast_sigmoid(pin_i + (hi[0][0] + (hi[0][1] + lo[0][0]) * (1.f / 2048.f)));
            const float fg = fast_sigmoid(pin_f + (hi[1][0] + (hi[1][1] + lo[1][0]) * (1.f / 2048.f)));
            const float gg = fast_tanh(pin_g + (hi[2][0] + (hi[2][1] + lo[2][0]) * (1.f / 2048.f)));
            const float og = fast_sigmoid(pin_o + (hi[3][0] + (hi[3][1] + lo[3][0]) * (1.f / 2048.f)));
            c = fg * c + ig * gg;
            hsel = og * fast_tanh(c);
            csel = c;
            _Float16 p0, p1;
            split_h2(hsel, p0, p1);
            _Float16* hb = &ring[(t & rmask) * TSTEP];
            hb[hoff] = p0;
            hb[4 * HID + hoff] = p1;
            lds_arrive(&sync[my_done], lane);
            if (!FUSE && layer && nq_raw < N) hout[base_q + (size_t)t * tstride + unit] = hsel;
            stamp(0);
        }
        if (state_out && nq_raw < N) {
            state_out[((size_t)layer * N + nq) * HID + unit] = hsel;           // h_{T-1}
            state_out[((size_t)(2 + layer) * N + nq) * HID + unit] = csel;     // c_{T-1}
        }
    }
    finish();
}

// =====================================================================================
// The same kernel for EIGHT sequences per workgroup (round 4, experimental: BSRNN_TIME_SEQ8=1).  Why: a time-axis launch is a latency chain that
// holds one CU per workgroup while using a fraction of it - 192 CUs at 64 rows -, and what runs beside it in the overlapped dual path (the second
// band block, the mask chain: api.hip::run_overlapped) is limited by the CUs left, not by its data (profiles/r04c_overlap_timeline.txt).  Half the
// workgroups leave 160 CUs instead of 64.  What changes against time_lstm_h2w_kernel: every row of the 16-row MFMA tiles carries data - recurrent
// form: row 2 j = first piece of sequence j, row 2 j + 1 = its second piece (still two MFMAs per block and gate); batched form (input half, fc):
// row = (sequence r >> 1, step r & 1) - so a group is TWO steps, a lane owns the two cells (unit, sequence 2 q) and (unit, sequence 2 q + 1), the
// step slots in LDS are [k / 8][8 sequences][8] per piece and the rings hold 8 steps (the same four groups).  Counters, roles, the order of every
// cell's arithmetic: unchanged - h and the carried state are bit-identical to the four-sequence kernel (tools/time_lstm_v3_bench.hip).
// =====================================================================================
constexpr int TCH8 = 4, RING8 = 8;
constexpr int TSTEP8 = 2 * 8 * HID + 32;      // halves per step slot: two pieces of [8][8][8] + 64 bytes of skew
template <bool FUSE, bool TRACE = false, bool PART = false>
__global__ __launch_bounds__(1024) void time_lstm_h2w8_kernel(const float* __restrict__ zin, float* __restrict__ hout,
                                                             const uint4* __restrict__ wpk, const float* __restrict__ bias,
                                                             const uint4* __restrict__ wfc, const float* __restrict__ bfc,
                                                             const float* __restrict__ state_in, float* __restrict__ state_out,
                                                             int R, int T, int K, int* __restrict__ range_flag, unsigned long long* __restrict__ dbg,
                                                             const float* __restrict__ part = nullptr, int* ovl_resident = nullptr, int* ovl_prog = nullptr, int ovl_base = 0)
{
    unsigned long long tp[4] = {0, 0, 0, 0}, tq = 0;          // measurement only
    auto stamp = [&](int k) { if (TRACE) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); tp[k] += now - tq; tq = now; } };
    __shared__ __attribute__((aligned(16))) _Float16 xpl[2 * TCH8 * TSTEP8];   // [chunk slot][step in chunk]: a ring of 8 steps
    __shared__ __attribute__((aligned(16))) _Float16 h0pl[RING8 * TSTEP8];     // h0_t in slot t & 7
    __shared__ __attribute__((aligned(16))) _Float16 h1pl[RING8 * TSTEP8];     // h1_t in slot t & 7
    __shared__ __attribute__((aligned(16))) float pinb[2 * 2 * 2 * 4 * 512];   // [layer][group parity][step in group][gate][cell]: bias + input half
    __shared__ __attribute__((aligned(16))) uint4 wflds[FUSE ? 4 * 4 * 64 : 1];
    __shared__ int sync[SY_COUNT];            // the counters (and the abort word) of lds_wait_ge / lds_arrive
    __shared__ float hb_lds[512];            // PART: the helpers' biases [layer][gate][unit]
    __shared__ __attribute__((aligned(16))) uint4 wsp[8 * 64];      // one W_ih fragment of every helper wave (the two cells' index math took its registers)
    __shared__ int outc[8];                   // overlapped dual path: storing waves that have drained group f, in slot f & 7 (monotonic: 4 per use)

    const int N = R * K;
    const int n0 = blockIdx.x * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                 // wave-uniform: the roles below are scalar branches
    const int role = wave >> 2, w4 = wave & 3;
    const int layer = role & 1;
    const int n = lane & 15, q = lane >> 4;
    const int unit = 16 * w4 + n;
    const int cellid = w4 * 64 + lane;
    const size_t tstride = (size_t)K * HID;
    const int G = (T + 1) >> 1;                  // groups of two steps

    // this lane's two cells: (unit, sequence 2 q) and (unit, sequence 2 q + 1)
    // (computed where they are used, not kept: the kernel has no register to spare)
    auto nq_raw_of = [&](int j) { return n0 + 2 * q + j; };
    auto nq_of = [&](int j) { const int v = n0 + 2 * q + j; return v < N ? v : N - 1; };
    auto base_of = [&](int j) { const int v = nq_of(j); return ((size_t)(v / K) * T * K + (v % K)) * HID; };
    const int cell0 = (w4 * 8 + 2 * q) * 16 + n; // index of the first cell in pinb's [cell] axis (the second: + 16)
    const int afrag = (q * 8 + (n >> 1)) * 8;    // A fragment inside a 32-deep block of a piece: row l & 15 = (sequence (l & 15) >> 1, ...)
    const int bstep = n & 1;                     // ... batched form: step (l & 15) & 1 of the group
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    if (tid < SY_COUNT) sync[tid] = 0;
    if (tid >= 64 && tid < 72) outc[tid - 64] = 0;
    // overlapped dual path (kernels.h, OvlProducer): this workgroup is on the chip - the launch that consumes its output is let go when all are
    if (PART && tid == 0 && ovl_resident) __hip_atomic_fetch_add(ovl_resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (may be a signal word the command processor polls)
    if (TRACE) tq = __builtin_amdgcn_s_memrealtime();

    // The two role families are laid out as "helpers: ...; return;  main waves: ..." and not as if / else: with a join behind both, the
    // structurizer keeps the values of the path laid out second alive through the first one's loops (a wave runs only one of them, but
    // the compiler sees entry -> helpers -> join -> main as a path) - a dozen registers the 128-VGPR budget of 16 waves per CU does not have.
    auto finish = [&]() {
        if (lds_peek(&sync[SY_ABORT]) && range_flag) *range_flag = 3;
        if (TRACE && lane == 0 && blockIdx.x < 4) {
            unsigned long long* d = dbg + (blockIdx.x * 16 + wave) * 4;
    #pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = tp[k];
        }
    };
    if (role >= 2) {
        // ------------------------------------------------------------------ helper waves: input halves, x staging, fc
        h8v w[2][4][2];                           // W_ih (fc_in folded for layer 0): [k block][gate][piece]
        {
            const uint4* wp = wpk + ((size_t)(layer * 4 + w4) * 4 * 4 * 2) * 64 + lane;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const uint4 v = wp[((b * 4 + gte) * 2 + pc) * 64];
                        if (b == 0 && gte == 0 && pc == 1) wsp[((role - 2) * 4 + w4) * 64 + lane] = v;      // (read back by the same lane)
                        else w[b][gte][pc] = __builtin_bit_cast(h8v, v);
                    }
        }
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) hb_lds[layer * 256 + gte * 64 + unit] = bias[layer * 256 + gte * 64 + unit];      // (read back after the workgroup's first barrier)
        float bf = 0.f;
        if (FUSE && layer) {                      // the fc matrix as B fragments [k block][piece] (column = output feature `unit`): used once per
            const uint4* wp = wfc + ((size_t)w4 * 2 * 2) * 64 + lane;       // group, so it lives in LDS; a wave reads back what it wrote itself
#pragma unroll
            for (int i = 0; i < 4; ++i) wflds[(w4 * 4 + i) * 64 + lane] = wp[i * 64];
            bf = bfc[unit];
        }
        // input half of the four steps of `group` (A rows = (sequence, step), `src` = slot of the group's first step): per gate
        // 6 MFMAs, then bias + result to the LDS buffer the main wave of this cell reads, [step][gate][cell]
        auto input_half = [&](const _Float16* src, int group) {
            if (TIME_ABL & 8) return;
            const _Float16* mine = src + bstep * TSTEP8;
            h8v a0[2], a1[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                a0[b] = *reinterpret_cast<const h8v*>(&mine[b * 256 + afrag]);
                a1[b] = *reinterpret_cast<const h8v*>(&mine[8 * HID + b * 256 + afrag]);
            }
            float* dst = pinb + layer * 8192 + (group & 1) * 4096 + cell0;
            float bsg[4];                         // PART: the biases come from LDS per group instead of living in four registers across the
#pragma unroll                                    // loop - the kernel has none to spare (no scratch: see H0)
            for (int gte = 0; gte < 4; ++gte) bsg[gte] = hb_lds[layer * 256 + gte * 64 + unit];
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) {
                v4f ghi = zero4, glo = zero4;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (TIME_ABL & 2) { ghi[0] += (float)a0[b][0] * (float)w[b][gte][0][0]; continue; }
                    ghi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[b], w[b][gte][0], ghi, 0, 0, 0);
                    glo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[b], (b == 0 && gte == 0) ? __builtin_bit_cast(h8v, wsp[((role - 2) * 4 + w4) * 64 + lane]) : w[b][gte][1], glo, 0, 0, 0);
                    glo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[b], w[b][gte][0], glo, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)       // row 4 q + e of the tile = (sequence 2 q + (e >> 1), step e & 1)
                    dst[(e & 1) * 2048 + gte * 512 + (e >> 1) * 16] = bsg[gte] + (ghi[e] + glo[e] * (1.f / 2048.f));
            }
        };
        if (!layer) {
            // ---------------- H0: x staging (its 256 threads: 8 steps x 4 sequences x 16 float4 = 512 float4, two per thread) and
            // the input half of layer 0.  Chunk c (steps 8 c ...) lives in slot c & 1; it is requested a whole chunk ahead,
            // stored once every H0 wave has published the groups that read the slot's previous content, and read once every
            // H0 wave has stored its share.
            const int xs_t = cellid >> 7, xs_i = (cellid >> 4) & 7, xs_c4 = cellid & 15;      // thread -> (step 0-1 [+2], sequence, float4)
            size_t xs_base;
            {
                int ni = n0 + xs_i; ni = ni < N ? ni : N - 1;
                xs_base = ((size_t)(ni / K) * T * K + (ni % K)) * HID + 4 * xs_c4;
            }
            float amax = 0.f;                    // range guard
            struct XRows { float4 z[2]; };
            auto chunk_load = [&](int chunk, XRows& v) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    int t = chunk * TCH8 + xs_t + 2 * hf;
                    t = t < T ? t : T - 1;
                    v.z[hf] = *reinterpret_cast<const float4*>(zin + xs_base + (size_t)t * tstride);
                }
            };
            auto chunk_store = [&](int chunk, const XRows& v) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float f[4] = {v.z[hf].x, v.z[hf].y, v.z[hf].z, v.z[hf].w};
                    h4v p0, p1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
                        _Float16 a, b2;
                        split_h2(f[e], a, b2);
                        p0[e] = a; p1[e] = b2;
                    }
                    _Float16* dst = &xpl[((chunk & 1) * TCH8 + xs_t + 2 * hf) * TSTEP8 + ((xs_c4 >> 1) * 8 + xs_i) * 8 + (xs_c4 & 1) * 4];
                    *reinterpret_cast<h4v*>(dst) = p0;
                    *reinterpret_cast<h4v*>(dst + 8 * HID) = p1;
                }
            };
            auto xslot = [&](int t) { return &xpl[(t & (2 * TCH8 - 1)) * TSTEP8]; };
            if (PART) {
                // PART stages group by group instead of chunk by chunk: three rows per (sequence, step) - residual and the two fc shares -
                // would be 24 registers held across the input half (the kernel then spills: 1024 threads leave 128 VGPRs per wave);
                // one group's rows are 12.  A group's four steps are requested one group ahead, stored (summed) at the top of the
                // next iteration once every wave has published the group that used the same slots (four groups = 16 steps earlier).
                struct GRows { float4 z, pf, pb; };
                auto rows_of = [&](int g) { int t = 2 * g + xs_t; t = t < T ? t : T - 1; return xs_base + (size_t)t * tstride; };
                auto load_z = [&](int g, GRows& v) { v.z = *reinterpret_cast<const float4*>(zin + rows_of(g)); };
                auto load_shares = [&](int g, GRows& v) {
                    const float* pp = part + 2 * (rows_of(g) - 4 * xs_c4) + 4 * xs_c4;
                    v.pf = *reinterpret_cast<const float4*>(pp);
                    v.pb = *reinterpret_cast<const float4*>(pp + HID);
                };
                auto group_store = [&](int g, const GRows& v) {
                    const float f[4] = {(v.z.x + v.pf.x) + v.pb.x, (v.z.y + v.pf.y) + v.pb.y, (v.z.z + v.pf.z) + v.pb.z, (v.z.w + v.pf.w) + v.pb.w};
                    h4v p0, p1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
                        _Float16 a, b2;
                        split_h2(f[e], a, b2);
                        p0[e] = a; p1[e] = b2;
                    }
                    _Float16* dst = xslot(2 * g + xs_t) + ((xs_c4 >> 1) * 8 + xs_i) * 8 + (xs_c4 & 1) * 4;
                    *reinterpret_cast<h4v*>(dst) = p0;
                    *reinterpret_cast<h4v*>(dst + 8 * HID) = p1;
                    // ... and to the OUTPUT buffer, where the fc wave (H1) picks it up as its residual two groups later and then
                    // overwrites it with the block's result: one read of the three rows instead of two.  Same workgroup, same CU:
                    // the store is complete (vmcnt(0) in front of this wave's PIN0 arrival below) long before the chain H0 -> M0 ->
                    // H1 of LDS counters lets H1 ask for it, and nobody has read that line before (no stale copy in the L1).
                    if (n0 + xs_i < N && 2 * g + xs_t < T) {
                        if (OVL_DBG & 4) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) __hip_atomic_store((float __attribute__((address_space(1)))*)(hout + rows_of(g) + e), f[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else
                        *reinterpret_cast<float4*>(hout + rows_of(g)) = make_float4(f[0], f[1], f[2], f[3]);
                    }
                };
                // Only the residual row (4 registers) is in flight across an input half; the two share rows of the next group are
                // requested BEHIND it and land during the waits at the top of the next iteration: the kernel must stay inside 128
                // VGPRs without scratch (with spills - the weight fragments, reloaded in the loops - a call beside a second
                // process's first kernels came out a few ulp different once in ~50: tools/busy_start_stress.py).
                GRows xr;
                load_z(0, xr); load_shares(0, xr); group_store(0, xr);
                if (G > 1) { load_z(1, xr); load_shares(1, xr); }
                __syncthreads();
                for (int g = 0; g < G; ++g) {
                    if (g) {
                        if (g >= 4) lds_wait_ge(sync, SY_PIN0, 4 * (g - 3));   // group g - 4 (the same slots) is published by every wave
                        group_store(g, xr);
                        lds_arrive(&sync[SY_X], lane);
                        if (g + 1 < G) load_z(g + 1, xr);
                        lds_wait_ge(sync, SY_X, 4 * g);                    // every wave has stored its share of group g
                    }
                    if (g >= 2) lds_wait_ge(sync, SY_DONE0, 8 * (g - 1));    // layer 0 has finished group g - 2 (same buffer)
                    stamp(0);
                    input_half(xslot(2 * g), g);
                    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's row stores of the group are in L2
                    lds_arrive(&sync[SY_PIN0], lane);
                    if (g && g + 1 < G) load_shares(g + 1, xr);
                    stamp(2);
                }
            } else {
            XRows xnext;
            chunk_load(0, xnext); chunk_store(0, xnext);
            if (TCH8 < T) chunk_load(1, xnext);
            __syncthreads();
            for (int g = 0; g < G; ++g) {
                const int ch = g >> 1;
                if (g && !(g & 1)) {                                  // first group of chunk ch >= 1
                    lds_wait_ge(sync, SY_PIN0, 4 * (2 * ch - 2));    // the groups that read chunk ch - 2 (same slot) are published by every wave
                    chunk_store(ch, xnext);
                    lds_arrive(&sync[SY_X], lane);
                    if ((ch + 1) * TCH8 < T) chunk_load(ch + 1, xnext);
                    lds_wait_ge(sync, SY_X, 4 * ch);                 // every wave has stored its share of chunk ch
                }
                if (g >= 2) lds_wait_ge(sync, SY_DONE0, 8 * (g - 1));    // layer 0 has finished group g - 2 (same buffer)
                stamp(0);
                input_half(xslot(2 * g), g);
                if (g) lds_wait_ge(sync, SY_PIN0, 4 * g);             // (SY_PIN0 is a sum over the four waves: no wave two groups ahead of another, see H1)
                lds_arrive(&sync[SY_PIN0], lane);
                stamp(2);
            }
            }
            if (!(amax <= 65504.f) && range_flag) *range_flag = 1;
        } else {
            // ---------------- H1: input half of layer 1 (A = h0 of the group, complete once layer 0 has finished its last step)
            // FUSE: fc + residual of a group of layer 1 in two stages, so that H1 never waits for anything but its own gates: the
            // residual rows of group f are REQUESTED in iteration f + 1 (fc_request) and the group is finished in iteration f + 2
            // (fc_finish: A rows = (sequence, step) of h1 as in the batched input half, 6 MFMAs, epilogue), when the gate that
            // iteration waits for anyway - layer 1 has finished group f - says its h1 is complete
            float xres[4] = {0.f, 0.f, 0.f, 0.f};  // [e]: (sequence 2 q + (e >> 1), step e & 1) of the group, as the tile's rows
            const float* const res_src = PART ? hout : zin;     // PART: the staging wave left the summed row (residual + fc shares) in the output buffer
            auto fc_request = [&](int f) {
                const int ffirst = 2 * f, nst = T - ffirst < 2 ? T - ffirst : 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t row = base_of(e >> 1) + (size_t)(ffirst + ((e & 1) < nst ? (e & 1) : nst - 1)) * tstride;
                    xres[e] = res_src[row + unit];
                }
            };
            // Overlapped dual path: the output rows of group f are in memory (every store of them write-through, this wave's drained by the
            // vmcnt(0) here; the last of the four storing waves to say so for f tells the consumers - each wave works through the groups in
            // order and drains all its older stores with it, so f + 1 published groups mean groups 0 .. f are complete whichever wave
            // published them).  Called one group late, where the wave waits for its residual loads anyway: no extra stall on H1.
            const bool pub = FUSE && PART && ovl_prog != nullptr;       // (only the parts flow overlaps: api.hip)
            // The consumers work in tiles of 16 frames and more, so progress is published every FOURTH group (and at the end): one drain per 16
            // steps instead of one per 4 keeps H1's stalls on its write-through stores off the chain (each drain covers all the wave's older stores).
            auto publish = [&](int f) {
                if ((f & 7) != 7 && f != G - 1) return;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&outc[(f >> 3) & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                // (absolute value, epoch in the upper bits, atomic MAX: the four-group publishes of a workgroup may come from different waves out of order)
                // (the consumers count in groups of FOUR frames, kernels.h::ovl_wait_rows: frames done = min(2 (f + 1), T))
                const int done4 = f == G - 1 ? (T + 3) >> 2 : (f + 1) >> 1;
                if ((old & 3) == 3 && lane == 0) __hip_atomic_fetch_max(ovl_prog + blockIdx.x, ovl_base + done4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            auto fc_finish = [&](int f) {
                if (pub && f >= 1) publish(f - 1);
                const int ffirst = 2 * f, nst = T - ffirst < 2 ? T - ffirst : 2;
                const _Float16* mine = &h1pl[((ffirst & (RING8 - 1)) + bstep) * TSTEP8];
                v4f fhi = zero4, flo = zero4;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const h8v f0 = *reinterpret_cast<const h8v*>(&mine[b * 256 + afrag]);
                    const h8v f1 = *reinterpret_cast<const h8v*>(&mine[8 * HID + b * 256 + afrag]);
                    const h8v wf0 = __builtin_bit_cast(h8v, wflds[(w4 * 4 + b * 2 + 0) * 64 + lane]);
                    const h8v wf1 = __builtin_bit_cast(h8v, wflds[(w4 * 4 + b * 2 + 1) * 64 + lane]);
                    fhi = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf0, fhi, 0, 0, 0);
                    flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf1, flo, 0, 0, 0);
                    flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, wf0, flo, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                // one block's fragments at a time (registers)
                }
                lds_arrive(&sync[SY_FC], lane);                       // the group's h1 slots are free again
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((e & 1) < nst && nq_raw_of(e >> 1) < N) {
                        const float v = ((fhi[e] + flo[e] * (1.f / 2048.f)) + bf) + xres[e];
                        float* const dst = hout + base_of(e >> 1) + (size_t)(ffirst + (e & 1)) * tstride + unit;
                        if (pub) __hip_atomic_store((float __attribute__((address_space(1)))*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store_dword sc1
                        else *dst = v;
                    }
            };
            __syncthreads();
            for (int g = 0; g < G; ++g) {
                const int last = 2 * g + 2 < T ? 2 * g + 2 : T;
                lds_wait_ge(sync, SY_DONE0, 4 * last);               // h0 of the group complete
                if (g >= 2) lds_wait_ge(sync, SY_DONE1, 8 * (g - 1));    // layer 1 has finished group g - 2 (same buffer; its h1 is complete)
                stamp(0);
                input_half(&h0pl[((2 * g) & (RING8 - 1)) * TSTEP8], g);
                // SY_PIN1 and SY_FC are SUMS over the four H1 waves, and their waiters (M1: this group's input half is published; M0 / M1:
                // a ring slot has been read) conclude "every wave has done group g" from "sum >= 4 (g + 1)".  That only holds while no wave
                // is two groups ahead of another - and nothing else ties the H1 waves to each other: one of them stalled for a few
                // microseconds on global memory (its residual loads, its output stores) while the other three went on gave M1 a group
                // whose input half - or an h1 slot whose fc read - was the stalled wave's old one: 16 units of four sequences slightly
                // wrong from a group boundary on.  Seen once in ~100 calls with write-through output stores (overlapped dual path,
                // tools/overlap_probe.py), and the likely cause of round 3's "result depended on a second process starting" (DESIGN 4e).
                // So a wave publishes group g only when all four have published g - 1: skew <= 1 group, for which the sums are exact.
                if (g) lds_wait_ge(sync, SY_PIN1, 4 * g);
                lds_arrive(&sync[SY_PIN1], lane);
                stamp(2);
                if (FUSE) {
                    if (g >= 2) fc_finish(g - 2);
                    if (g >= 1) fc_request(g - 1);
                }
                stamp(1);
            }
            if (FUSE) {                           // the last two groups
                if (G >= 2) { lds_wait_ge(sync, SY_DONE1, 8 * (G - 1)); fc_finish(G - 2); }
                fc_request(G - 1);
                lds_wait_ge(sync, SY_DONE1, 4 * T);
                fc_finish(G - 1);
                if (pub) publish(G - 1);
            }
        }
        finish();
        return;
    }
    {
        // ------------------------------------------------------------------ main waves: the serial chain
        h8v w[2][4][2];                           // W_hh: [k block][gate][piece]
        {
            const uint4* wp = wpk + ((size_t)(layer * 4 + w4) * 4 * 4 * 2) * 64 + lane;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) w[b][gte][pc] = __builtin_bit_cast(h8v, wp[(((2 + b) * 4 + gte) * 2 + pc) * 64]);
        }
        const int hoff = ((unit >> 3) * 8 + 2 * q) * 8 + (unit & 7);  // where this lane's first cell's h goes inside a piece (the second: + 8)
        _Float16* const ring = layer ? h1pl : h0pl;
        const int rmask = RING8 - 1;
        float c[2], hsel[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            c[j] = state_in ? state_in[((size_t)(2 + layer) * N + nq_of(j)) * HID + unit] : 0.f;
            const float hinit = state_in ? state_in[((size_t)layer * N + nq_of(j)) * HID + unit] : 0.f;
            _Float16 p0, p1;
            split_h2(hinit, p0, p1);
            _Float16* hb = &ring[rmask * TSTEP8];                     // h_{-1}
            hb[hoff + 8 * j] = p0;
            hb[8 * HID + hoff + 8 * j] = p1;
        }
        // Only rows 4j of the 16-row tile carry a sequence, so row 4j + 1 is given the SECOND piece of the same sequence: one
        // MFMA against w1 yields a1 w1 in register 0 and a2 w1 in register 1 of the owning lane, a second one against w2
        // yields a1 w2 in register 0 - two MFMAs per (block, gate) instead of three, one fragment read per block.
        const int rfrag = afrag + ((n & 1) ? 8 * HID : 0);            // (every row of the tile carries a sequence's piece: 2 j = first, 2 j + 1 = second)
        const float* const pin_l = pinb + layer * 8192 + cell0;
        const int my_done = layer ? SY_DONE1 : SY_DONE0, my_pin = layer ? SY_PIN1 : SY_PIN0;
        // whoever reads this layer's ring besides the layer itself: H1 batches h0 and (fc) h1
        const int reader = layer ? SY_FC : SY_PIN1;
        const bool has_reader = layer ? FUSE : true;
        const int rsteps = RING8;
        __syncthreads();                          // h_{-1}, counters, x chunk 0 (the helpers publish group 0 behind it)
        if (TIME_OPT & 1) __builtin_amdgcn_s_setprio(2);     // the chain before the helpers wherever both could issue
        for (int t = 0; t < T; ++t) {
            if ((t & 1) == 0) {
                if (!(TIME_ABL & 1)) lds_wait_ge(sync, my_pin, 4 * ((t >> 1) + 1));              // this group's input half is published
                if (has_reader && t >= rsteps) lds_wait_ge(sync, reader, 4 * (((t - rsteps) >> 1) + 1));   // the slots this group overwrites have been read
            }
            stamp(3);
            const _Float16* src = &ring[((t - 1) & rmask) * TSTEP8];
            h8v a[2];
            if (TIME_OPT & 2) {
                // h_{t-1} complete (all four waves)?  The counter and the two fragments are requested together - LDS operations of a
                // wave execute in order, so fragments that follow a counter value >= 4 t are complete - and requested again if not:
                // one LDS round trip per step instead of two.
                const unsigned ca = (unsigned)(size_t)&sync[my_done], fa = (unsigned)(size_t)&src[rfrag];
                int spins = 0;
                for (;;) {
                    int cv;
                    u4v f0, f1;
                    asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:512\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(cv), "=&v"(f0), "=&v"(f1) : "v"(ca), "v"(fa) : "memory");
                    a[0] = __builtin_bit_cast(h8v, f0); a[1] = __builtin_bit_cast(h8v, f1);
                    if ((TIME_ABL & 4) || __builtin_amdgcn_readfirstlane(cv) >= 4 * t) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) __hip_atomic_store(&sync[SY_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (spins > 64 && lds_peek(&sync[SY_ABORT])) break;
                }
            } else {
                if (!(TIME_ABL & 4)) lds_wait_ge(sync, my_done, 4 * t);                          // h_{t-1} complete (all four waves)
#pragma unroll
                for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const h8v*>(&src[b * 256 + rfrag]);
            }
            stamp(2);
            const float* const pin_t = pin_l + (((t >> 1) & 1) * 2 + (t & 1)) * 2048;
            float pin_i[2], pin_f[2], pin_g[2], pin_o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { pin_i[j] = pin_t[16 * j]; pin_f[j] = pin_t[512 + 16 * j]; pin_g[j] = pin_t[1024 + 16 * j]; pin_o[j] = pin_t[1536 + 16 * j]; }
            v4f hi[4], lo[4];
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) { hi[gte] = zero4; lo[gte] = zero4; }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][0], hi[gte], 0, 0, 0);
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][1], lo[gte], 0, 0, 0);
            }
            // hi[g] = {a1 w1, a2 w1 | the same of the second cell}, lo[g] = {a1 w2, - | a1 w2, -} of this lane's two cells
            _Float16* hb = &ring[(t & rmask) * TSTEP8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float ig = fast_sigmoid(pin_i[j] + (hi[0][2 * j] + (hi[0][2 * j + 1] + lo[0][2 * j]) * (1.f / 2048.f)));
                const float fg = fast_sigmoid(pin_f[j] + (hi[1][2 * j] + (hi[1][2 * j + 1] + lo[1][2 * j]) * (1.f / 2048.f)));
                const float gg = fast_tanh(pin_g[j] + (hi[2][2 * j] + (hi[2][2 * j + 1] + lo[2][2 * j]) * (1.f / 2048.f)));
                const float og = fast_sigmoid(pin_o[j] + (hi[3][2 * j] + (hi[3][2 * j + 1] + lo[3][2 * j]) * (1.f / 2048.f)));
                c[j] = fg * c[j] + ig * gg;
                hsel[j] = og * fast_tanh(c[j]);
                _Float16 p0, p1;
                split_h2(hsel[j], p0, p1);
                hb[hoff + 8 * j] = p0;
                hb[8 * HID + hoff + 8 * j] = p1;
            }
            lds_arrive(&sync[my_done], lane);
            if (!FUSE && layer) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (nq_raw_of(j) < N) hout[base_of(j) + (size_t)t * tstride + unit] = hsel[j];
            }
            stamp(0);
        }
        if (state_out) {
            // (the lane's coordinates again, from the lane counter: nothing of the prologue's index math stays live across the step loop)
            const int le = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int ue = 16 * w4 + (le & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ne = n0 + 2 * (le >> 4) + j;
                if (ne < N) {
                    state_out[((size_t)layer * N + ne) * HID + ue] = hsel[j];           // h_{T-1}
                    state_out[((size_t)(2 + layer) * N + ne) * HID + ue] = c[j];        // c_{T-1}
                }
            }
        }
    }
    finish();
}

// =====================================================================================
// Band-axis block for a HANDFUL of sequences (the one-frame streaming step: N = C frame rows = 2 sequences of K = 12 bands):
// BandwiseLSTM's whole NormRNNResidual (bsrnn.py:138-153, :78-87) - layer 0 in both directions, layer 1 in both directions,
// fc(128 -> 64) + bias + residual - in ONE workgroup per four sequences, instead of two band_lstm_h2 launches (two 4-wave
// workgroups each on an otherwise empty chip, 15 + 18 us of which the 16 x 16 tile with 2 of 16 columns used is the least
// problem: every step pays 48 / 72 MFMAs per wave) and a grouped-GEMM launch.  Same fp16x2 arithmetic.
// Organisation = the time-axis kernel's: activations are the A operand with row 4j = sequence j's first piece and row 4j + 1
// its second piece (one MFMA against w1 gives a1 w1 and a2 w1, one against w2 gives a1 w2: two MFMAs per (gate, k block)
// instead of three), weights the B operand (tile g = gate g of the wave's 16 units) - the packed weights of band_lstm_h2 are
// exactly those fragments (api.hip) - so lane (n, q) owns the cell (unit 16 w + n, sequence q).  Waves 0-3 run the forward
// direction, waves 4-7 the backward one; one barrier per step; all L positions of x, of layer 0's output and of layer 1's
// output stay in LDS as fp16 pieces ([k / 8][sequence][8] per position), the fc at the end batches four positions per tile.
// =====================================================================================
constexpr int BS_MAXL = 16;                     // positions (bands) the LDS images hold; longer band tables take the general kernels
constexpr int BS_XSTEP = TSTEP;                 // halves per position of the 64-wide x image (2 pieces x [8][4][8] + skew)
constexpr int BS_HSTEP = 2 * 4 * 2 * HID + 32;  // halves per position of a 128-wide image (forward | backward halves)

__global__ __launch_bounds__(512) void band_block_small_kernel(const float* __restrict__ zin, float* __restrict__ zout,
                                                               const uint4* __restrict__ w0pk, const float* __restrict__ bias0,
                                                               const uint4* __restrict__ w1pk, const float* __restrict__ bias1,
                                                               const uint4* __restrict__ wfc, const float* __restrict__ bfc,
                                                               int N, int L, int* __restrict__ range_flag)
{
    __shared__ __attribute__((aligned(16))) _Float16 xpl[BS_MAXL * BS_XSTEP];          // x_t, 64 wide
    __shared__ __attribute__((aligned(16))) _Float16 h0pl[(BS_MAXL + 1) * BS_HSTEP];   // layer 0 output, slot L = zeros (h_{-1})
    __shared__ __attribute__((aligned(16))) _Float16 h1pl[(BS_MAXL + 1) * BS_HSTEP];   // layer 1 output, slot L = zeros

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = wave >> 2, w4 = wave & 3;
    const int n = lane & 15, q = lane >> 4;
    const int unit = 16 * w4 + n;
    const int n0 = blockIdx.x * 4;
    const int nq_raw = n0 + q;
    const int nq = nq_raw < N ? nq_raw : N - 1;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- x -> fp16 pieces in LDS (4 sequences x L positions x 16 float4), zero slots for h_{-1}
    float amax = 0.f;
    for (int i = tid; i < 4 * L * 16; i += 512) {
        const int c4 = i & 15, sq = (i >> 4) & 3, t = i >> 6;
        const int ns = n0 + sq < N ? n0 + sq : N - 1;
        const float4 v = *reinterpret_cast<const float4*>(zin + ((size_t)ns * L + t) * HID + 4 * c4);
        const float f[4] = {v.x, v.y, v.z, v.w};
        h4v p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = __builtin_fmaxf(amax, __builtin_fabsf(f[e]));
            _Float16 a, b2;
            split_h2(f[e], a, b2);
            p0[e] = a; p1[e] = b2;
        }
        _Float16* dst = &xpl[t * BS_XSTEP + ((c4 >> 1) * 4 + sq) * 8 + (c4 & 1) * 4];
        *reinterpret_cast<h4v*>(dst) = p0;
        *reinterpret_cast<h4v*>(dst + 4 * HID) = p1;
    }
    for (int i = tid; i < BS_HSTEP / 8; i += 512) {
        *reinterpret_cast<uint4*>(&h0pl[L * BS_HSTEP + 8 * i]) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(&h1pl[L * BS_HSTEP + 8 * i]) = make_uint4(0, 0, 0, 0);
    }

    // A fragment of a 32-deep k block inside one piece of an image: row l & 15 carries sequence (l & 15) >> 2, and rows 4j + 1
    // read the SECOND piece (which lies `pstride` halves behind the first)
    const int afrag = (q * 4 + (n >> 2)) * 8;
    const bool second = (n & 3) == 1;
    const int hoff = ((unit >> 3) * 4 + q) * 8 + (unit & 7);      // this lane's h inside the 64 k of its direction

    // one LSTM layer of this wave's direction: x image `ximg` (NBX k blocks of 32 per position, pieces XP halves apart),
    // output image `himg` (this direction's 64 k start at k-unit 8 dir; pieces 4 * 2 * HID halves apart)
    auto layer = [&](auto nbx_tag, const _Float16* ximg, const int xstep, const int XP, _Float16* himg, const uint4* wpk, const float* bias) {
        constexpr int NBX = decltype(nbx_tag)::value, NB = NBX + 2;
        h8v w[NB][4][2];
        {
            const uint4* wp = wpk + ((size_t)(dir * 4 + w4) * NB * 4 * 2) * 64 + lane;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int gte = 0; gte < 4; ++gte)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) w[b][gte][pc] = __builtin_bit_cast(h8v, wp[((b * 4 + gte) * 2 + pc) * 64]);
        }
        float bs[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) bs[gte] = bias[dir * 256 + gte * 64 + unit];
        float c = 0.f;
        const int HP = 4 * 2 * HID;                              // halves between the two pieces of a 128-wide image
        for (int step = 0; step < L; ++step) {
            const int t = dir ? L - 1 - step : step;
            const int tprev = step == 0 ? L : (dir ? t + 1 : t - 1);      // slot L holds zeros
            h8v a[NB];
            const _Float16* xs = ximg + t * xstep + (second ? XP : 0) + afrag;
#pragma unroll
            for (int b = 0; b < NBX; ++b) a[b] = *reinterpret_cast<const h8v*>(xs + b * 128);
            const _Float16* hs = himg + tprev * BS_HSTEP + (second ? HP : 0) + dir * 256 + afrag;
#pragma unroll
            for (int b = 0; b < 2; ++b) a[NBX + b] = *reinterpret_cast<const h8v*>(hs + b * 128);
            v4f hi[4], lo[4];
#pragma unroll
            for (int gte = 0; gte < 4; ++gte) { hi[gte] = zero4; lo[gte] = zero4; }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) hi[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][0], hi[gte], 0, 0, 0);
#pragma unroll
                for (int gte = 0; gte < 4; ++gte) lo[gte] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[b], w[b][gte][1], lo[gte], 0, 0, 0);
            }
            // hi[g] = {a1 w1, a2 w1, -, -}, lo[g] = {a1 w2, -, -, -} of this lane's cell (unit, sequence q)
            const float ig = fast_sigmoid(bs[0] + (hi[0][0] + (hi[0][1] + lo[0][0]) * (1.f / 2048.f)));
            const float fg = fast_sigmoid(bs[1] + (hi[1][0] + (hi[1][1] + lo[1][0]) * (1.f / 2048.f)));
            const float gg = fast_tanh(bs[2] + (hi[2][0] + (hi[2][1] + lo[2][0]) * (1.f / 2048.f)));
            const float og = fast_sigmoid(bs[3] + (hi[3][0] + (hi[3][1] + lo[3][0]) * (1.f / 2048.f)));
            c = fg * c + ig * gg;
            const float hv = og * fast_tanh(c);
            _Float16 p0, p1;
            split_h2(hv, p0, p1);
            _Float16* hb = himg + t * BS_HSTEP + dir * 256 + hoff;
            hb[0] = p0;
            hb[HP] = p1;
            __syncthreads();
        }
    };
    __syncthreads();
    layer(std::integral_constant<int, 2>(), xpl, BS_XSTEP, 4 * HID, h0pl, w0pk, bias0);          // layer 0: x is 64 wide
    layer(std::integral_constant<int, 4>(), h0pl, BS_HSTEP, 4 * 2 * HID, h1pl, w1pk, bias1);     // layer 1: x = layer 0, both directions

    // ---- fc(128 -> 64) + bias + residual, four positions per 16-row tile: row 4j + r = (sequence j, position 4 g + r); wave =
    // (group of positions, 16 output features) pairs
    const int groups = (L + 3) >> 2;
    const int bstep = n & 3;
    for (int task = wave; task < 4 * groups; task += 8) {
        const int tile = task & 3, g4 = task >> 2;
        int tpos = 4 * g4 + bstep;
        tpos = tpos < L ? tpos : L - 1;
        const _Float16* mine = h1pl + tpos * BS_HSTEP + afrag;
        v4f fhi = zero4, flo = zero4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const h8v f0 = *reinterpret_cast<const h8v*>(mine + b * 128);
            const h8v f1 = *reinterpret_cast<const h8v*>(mine + 4 * 2 * HID + b * 128);
            const h8v wf0 = __builtin_bit_cast(h8v, wfc[((tile * 4 + b) * 2 + 0) * 64 + lane]);
            const h8v wf1 = __builtin_bit_cast(h8v, wfc[((tile * 4 + b) * 2 + 1) * 64 + lane]);
            fhi = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf0, fhi, 0, 0, 0);
            flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, wf1, flo, 0, 0, 0);
            flo = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, wf0, flo, 0, 0, 0);
        }
        const int feat = 16 * tile + n;
        const float bf = bfc[feat];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 4 * g4 + r;
            if (t < L && nq_raw < N) {
                const size_t o = ((size_t)nq * L + t) * HID + feat;
                zout[o] = ((fhi[r] + flo[r] * (1.f / 2048.f)) + bf) + zin[o];
            }
        }
    }
    if (!(amax <= 65504.f) && range_flag) *range_flag = 1;
}

// true when launch_band_block_small() will take the block (fp16x2 modes, few sequences, a band table that fits the LDS images)
bool band_block_is_small(int N, int L)
{
    static const bool on = [] { const char* e = getenv("BSRNN_BAND_SMALL"); return !(e && !strcmp(e, "0")); }();
    return on && lstm_mode() == LSTM_FP16X2 && !force_f32() && gemm_mode() != GEMM_F32 && N >= 1 && N <= 8 && L >= 1 && L <= BS_MAXL;
}
void launch_band_block_small(const float* zin, float* zout, const void* w0pk16, const float* bias0, const void* w1pk16, const float* bias1,
                             const void* fc16, const float* fcb, int N, int L, int* range_flag, hipStream_t stream)
{
    hipLaunchKernelGGL(band_block_small_kernel, dim3((N + 3) / 4), dim3(512), 0, stream, zin, zout, (const uint4*)w0pk16, bias0,
                       (const uint4*)w1pk16, bias1, (const uint4*)fc16, fcb, N, L, range_flag);
}

// The 16-wave kernel computes the block's fc + residual itself when the Linear layers are not asked to be exact fp32
// (BSRNN_GEMM=f32 keeps every nn.Linear on the fp32 matrix kernels) - api.hip then skips the block's grouped-GEMM launch.
// BSRNN_TIME_KERNEL = v2 (round-2 kernel, 8 waves) | v3 (16 waves, separate fc launch) | fused (default).
static int time_kernel_variant()
{
    static const int v = [] {
        const char* e = getenv("BSRNN_TIME_KERNEL");
        if (!e || !*e || !strcmp(e, "fused")) return 2;
        if (!strcmp(e, "v3")) return 1;
        if (!strcmp(e, "v2")) return 0;
        fprintf(stderr, "bsrnn: unknown BSRNN_TIME_KERNEL='%s' (v2 | v3 | fused), using fused\n", e);
        return 2;
    }();
    return v;
}
bool time_lstm_fuses_fc()
{
    return lstm_mode() == LSTM_FP16X2 && !force_f32() && gemm_mode() != GEMM_F32 && time_kernel_variant() == 2;
}

// measurement (BSRNN_BAND_FC=part_add): out = (z + part[.., 0, :]) + part[.., 1, :] as a launch of its own, so that the time kernel
// runs without PART on the same numbers - separates the two halves of the parts flow when something depends on which one runs
__global__ void parts_add_kernel(const float* __restrict__ z, const float* __restrict__ part, float* __restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const size_t row = i / 16, c4 = i % 16;
    const float4 a = reinterpret_cast<const float4*>(z)[i];
    const float4 f = reinterpret_cast<const float4*>(part)[row * 32 + c4], b = reinterpret_cast<const float4*>(part)[row * 32 + 16 + c4];
    reinterpret_cast<float4*>(out)[i] = make_float4((a.x + f.x) + b.x, (a.y + f.y) + b.y, (a.z + f.z) + b.z, (a.w + f.w) + b.w);
}
static bool parts_add_mode()
{
    static const bool on = [] { const char* e = getenv("BSRNN_BAND_FC"); return e && !strcmp(e, "part_add"); }();
    return on;
}

int device_cus()
{
    static const int cus = [] { int d = 0, n = 0; return hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0 ? n : 256; }();
    return cus;
}
int time_lstm_seqs(int N)
{
    static const int seq8 = [] { const char* e = getenv("BSRNN_TIME_SEQ8"); return e ? atoi(e) : -1; }();
    if (lstm_mode() != LSTM_FP16X2 || force_f32() || !time_lstm_fuses_fc()) return 4;
    return seq8 == 1 || (seq8 < 0 && (N + 3) / 4 > device_cus()) ? 8 : 4;
}
void launch_time_lstm(const float* zin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      const float* state_in, float* state_out, int R, int T, int K, int* range_flag, hipStream_t stream,
                      const void* fc16, const float* fcb, const float* part, const OvlProducer* ovl)
{
    const int N = R * K;
    if (N <= 0 || T <= 0) return;
    if (part && parts_add_mode() && time_lstm_fuses_fc() && fc16 && fcb) {
        const size_t n4 = (size_t)R * T * K * 16;          // in place: zin is the block's input from here on (its old content is dead)
        hipLaunchKernelGGL(parts_add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, zin, part, const_cast<float*>(zin), n4);
        part = nullptr;                                    // zin now holds the block's input; the plain fused kernel follows
    }
    dim3 grid((N + 3) / 4), block(512), block16(1024);
    if (lstm_mode() == LSTM_FP16X2 && !force_f32()) {
        // Eight sequences per workgroup where four would need more than one round of workgroups (one per CU): the 41-band table, large batches;
        // BSRNN_TIME_SEQ8 = 0 never / 1 always (A/B, tests).  Results are bit-identical either way.
        if (time_lstm_fuses_fc() && fc16 && fcb && time_lstm_seqs(N) == 8) {
            const dim3 grid8((N + 7) / 8);
            if (part)
                hipLaunchKernelGGL((time_lstm_h2w8_kernel<true, false, true>), grid8, block16, 0, stream, zin, hout, (const uint4*)wpk16, bias, (const uint4*)fc16, fcb,
                                   state_in, state_out, R, T, K, range_flag, (unsigned long long*)nullptr, part, ovl ? ovl->resident : (int*)nullptr,
                                   ovl ? ovl->prog : (int*)nullptr, ovl ? ovl->base : 0);
            else
                hipLaunchKernelGGL((time_lstm_h2w8_kernel<true, false>), grid8, block16, 0, stream, zin, hout, (const uint4*)wpk16, bias, (const uint4*)fc16, fcb,
                                   state_in, state_out, R, T, K, range_flag, (unsigned long long*)nullptr);
            return;
        }
        if (time_lstm_fuses_fc() && fc16 && fcb && part)
            hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false, true>), grid, block16, 0, stream, zin, hout, (const uint4*)wpk16, bias, (const uint4*)fc16, fcb,
                               state_in, state_out, R, T, K, range_flag, (unsigned long long*)nullptr, part, ovl ? ovl->resident : (int*)nullptr,
                               ovl ? ovl->prog : (int*)nullptr, ovl ? ovl->base : 0);
        else if (time_lstm_fuses_fc() && fc16 && fcb)
            hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false>), grid, block16, 0, stream, zin, hout, (const uint4*)wpk16, bias, (const uint4*)fc16, fcb,
                               state_in, state_out, R, T, K, range_flag, (unsigned long long*)nullptr);
        else if (time_kernel_variant() >= 1)
            hipLaunchKernelGGL((time_lstm_h2w_kernel<false, false>), grid, block16, 0, stream, zin, hout, (const uint4*)wpk16, bias, (const uint4*)nullptr,
                               (const float*)nullptr, state_in, state_out, R, T, K, range_flag, (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL(time_lstm_h2_kernel<false>, grid, block, 0, stream, zin, hout, (const uint4*)wpk16, bias, state_in, state_out, R, T, K,
                               range_flag, (unsigned long long*)nullptr);
        return;
    }
    hipLaunchKernelGGL(time_lstm_kernel, grid, block, 0, stream, zin, hout, wpk, bias, state_in, state_out, R, T, K,
                       (unsigned long long*)nullptr);
}

// Gate of the overlapped dual path (kernels.h): ONE wave that leaves when every workgroup of the time-axis launch on the other stream is
// resident; the consumer launch follows it in stream order, so no consumer workgroup can take a CU a producer still needs.
__global__ void ovl_gate_kernel(const int* resident, int target, int* range_flag, int limit)
{
    if (threadIdx.x) return;
    typedef const int __attribute__((address_space(1)))* gci;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load((gci)resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {        // (a running total)
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)limit) { if (range_flag) *range_flag = 5; return; }
    }
}
void launch_ovl_gate(const int* resident, int target, int* range_flag, int spin_limit, hipStream_t stream)
{
    hipLaunchKernelGGL(ovl_gate_kernel, dim3(1), dim3(64), 0, stream, resident, target, range_flag, spin_limit);
}

}  // namespace bsrnn
