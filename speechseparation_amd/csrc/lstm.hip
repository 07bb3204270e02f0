// Dual-path recurrent kernels for gfx950: the band-axis BLSTM layer and the causal time-axis
// two-layer LSTM of NormRNNResidual (bsrnn.py:63-98, BandwiseLSTM :131-162, TimewiseLSTM
// :101-128).  fp32 throughout (exact-fp32 MFMA), torch gate order i,f,g,o:
//     c' = sigmoid(f) c + sigmoid(i) tanh(g),  h' = sigmoid(o) tanh(c').
// fc_in (Linear 64->64, no activation) is folded into W_ih of layer 0 on the host
// (api.hip: W' = W_ih W_in, b' = W_ih b_in + b_ih + b_hh, in double), the trailing fc + residual
// runs as one grouped-GEMM launch (gemm.hip, EPI_RES).
#include "kernels.h"

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
    float bs[4];
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) bs[gte] = bias[dir * 256 + gte * 64 + 16 * wave + l15];

    float c[4] = {0.f, 0.f, 0.f, 0.f};

    // staging maps
    const int xr_row[2] = {(tid * XV) / (IN / 4), (tid * XV + 1) / (IN / 4)};
    const int xr_c4[2] = {(tid * XV) % (IN / 4), (tid * XV + 1) % (IN / 4)};
    const int o_row = tid >> 4, o_c4 = tid & 15;

    auto xload = [&](int t, float4* dst) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            int row = n0 + xr_row[i];
            row = row < N ? row : N - 1;
            dst[i] = *reinterpret_cast<const float4*>(xin + ((size_t)row * L + t) * IN + 4 * xr_c4[i]);
        }
    };
    auto xstore = [&](int buf, const float4* src) {
#pragma unroll
        for (int i = 0; i < XV; ++i)
            *reinterpret_cast<float4*>(&xbuf[buf][xr_row[i] * SX + 4 * xr_c4[i]]) = src[i];
    };

    {   // prologue: h_{-1} = 0, x of the first step
        *reinterpret_cast<float4*>(&hbuf[0][o_row * SH + 4 * o_c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 x0[XV];
        xload(dir ? L - 1 : 0, x0);
        xstore(0, x0);
    }
    __syncthreads();

    for (int step = 0; step < L; ++step) {
        const int t = dir ? L - 1 - step : step;
        const int cur = step & 1, nxt = cur ^ 1;
        float4 xn[XV];
        const bool more = step + 1 < L;
        if (more) xload(dir ? t - 1 : t + 1, xn);

        v4f acc[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) acc[gte] = (v4f){bs[gte], bs[gte], bs[gte], bs[gte]};

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

void launch_band_lstm(const float* xin, float* hout, const float* wpk, const float* bias,
                      int N, int L, int IN, hipStream_t stream)
{
    if (N <= 0 || L <= 0) return;
    dim3 grid((N + 15) / 16, 2), block(256);
    if (IN == 64)
        hipLaunchKernelGGL(band_lstm_kernel<64>, grid, block, 0, stream, xin, hout, wpk, bias, N, L);
    else
        hipLaunchKernelGGL(band_lstm_kernel<128>, grid, block, 0, stream, xin, hout, wpk, bias, N, L);
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

void launch_time_lstm(const float* zin, float* hout, const float* wpk, const float* bias,
                      const float* state_in, float* state_out, int R, int T, int K, hipStream_t stream)
{
    const int N = R * K;
    if (N <= 0 || T <= 0) return;
    dim3 grid((N + 3) / 4), block(512);
    hipLaunchKernelGGL(time_lstm_kernel, grid, block, 0, stream, zin, hout, wpk, bias, state_in, state_out, R, T, K,
                       (unsigned long long*)nullptr);
}

}  // namespace bsrnn
