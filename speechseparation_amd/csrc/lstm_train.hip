// Training step, part 1: the recurrent layers of NormRNNResidual (nn.LSTM, bsrnn.py:66-72) in the form a
// training step needs them - a forward pass that keeps what the backward pass reads, and the backward pass itself
// (back-propagation through time), i.e. what `loss.backward()` does for these layers in the reference's train step
// (train.py:97-115, m_dataset.py:202-226).  Exact fp32 on v_mfma_f32_16x16x4_f32 (bit-for-bit fp32 fma chains): gradients
// are compared with torch.autograd on the CPU restatement (tests/test_gpu_train.py).
//
// One layer, `ndir` directions (direction 1 runs the sequence backwards: torch's `_reverse` weights), N sequences of L
// steps: the band-axis BLSTM (N = C T, L = K, ndir = 2) and the time-axis LSTM (N = C K, L = T, ndir = 1, rows gathered by
// the caller) are the same two kernels.  Weights in torch layout: w_ih [ndir][256][IN], w_hh [ndir][256][64], gate order
// i, f, g, o; bias [ndir][256] = b_ih + b_hh.  Zero initial state (offline training, infer.py:34 / m_dataset.py:191).
//
//   forward    x [N][L][IN]  ->  h [N][L][ndir 64], gates [N][L][ndir][4][64] (after the non-linearity), cells [N][L][ndir][64]
//   backward   dh [N][L][ndir 64] (gradient of the layer output)  ->  dg [N][L][ndir][256] (gradient of the gate
//              pre-activations), the only sequential part: dh_{t-1} += dg_t W_hh
//   then plain matrix products over all (sequence, step) rows:
//              dx = sum_dir dg_dir W_ih_dir        dW_ih = dg^T x        dW_hh = dg^T h_prev        db = column sums of dg
//
// Decomposition as in the inference kernels (lstm.hip): a workgroup owns 16 sequences of one direction, wave w the hidden
// units [16w, 16w + 16) of all four gates, so a lane holds i, f, g, o, c of its four (sequence, unit) cells in forward and
// backward alike; weights stay in VGPRs (MFMA B operand), activations go through LDS with the k-permutation
// k(step s, quarter q) = 16 (s / 4) + 4 q + s % 4, which turns four consecutive K = 4 MFMA steps into one ds_read_b128.
#include "kernels.h"

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float tanh_f(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.88539008f * x)) - 1.0f; }
__device__ __forceinline__ int kperm(int s, int q) { return 16 * (s / 4) + 4 * q + (s % 4); }

// ---------------------------------------------------------------------------------------------- forward, with saves
template <int IN>
__global__ __launch_bounds__(256, IN == 128 ? 1 : 2) void lstm_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                                const float* __restrict__ w_hh, const float* __restrict__ bias,
                                                                float* __restrict__ h, float* __restrict__ gates, float* __restrict__ cells,
                                                                int N, int L, int ndir)
{
    constexpr int KT = IN + HID, NS = KT / 4;
    constexpr int SX = IN + 8, SH = HID + 8;          // row strides = 8 (mod 64) floats: conflict-free ds_read_b128 groups
    constexpr int XV = IN / 64;
    __shared__ __attribute__((aligned(16))) float xbuf[2][16 * SX];
    __shared__ __attribute__((aligned(16))) float hbuf[2][16 * SH];

    const int dir = blockIdx.y;
    const int n0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int unit = 16 * wave + l15;

    // resident weights: w[s][g] = [W_ih | W_hh][g 64 + unit][k(s, q)]
    float w[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = kperm(s, q);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t row = (size_t)dir * 256 + g * 64 + unit;
            w[s][g] = k < IN ? w_ih[row * IN + k] : w_hh[row * HID + (k - IN)];
        }
    }
    float bs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[g] = bias[dir * 256 + g * 64 + unit];
    float c[4] = {0.f, 0.f, 0.f, 0.f};

    int xr_row[XV], xr_c4[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) { xr_row[i] = (tid * XV + i) / (IN / 4); xr_c4[i] = (tid * XV + i) % (IN / 4); }
    const int o_row = tid >> 4, o_c4 = tid & 15;
    auto xload = [&](int t, v4f* dst) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            int row = n0 + xr_row[i];
            row = row < N ? row : N - 1;
            dst[i] = *reinterpret_cast<const v4f*>(x + ((size_t)row * L + t) * IN + 4 * xr_c4[i]);
        }
    };
    auto xstore = [&](int buf, const v4f* src) {
#pragma unroll
        for (int i = 0; i < XV; ++i) *reinterpret_cast<v4f*>(&xbuf[buf][xr_row[i] * SX + 4 * xr_c4[i]]) = src[i];
    };
    {
        *reinterpret_cast<v4f*>(&hbuf[0][o_row * SH + 4 * o_c4]) = (v4f){0.f, 0.f, 0.f, 0.f};
        v4f x0[XV];
        xload(dir ? L - 1 : 0, x0);
        xstore(0, x0);
    }
    __syncthreads();

    const int HO = ndir * HID;
    for (int step = 0; step < L; ++step) {
        const int t = dir ? L - 1 - step : step;
        const int cur = step & 1, nxt = cur ^ 1;
        v4f xn[XV];
        const bool more = step + 1 < L;
        if (more) xload(dir ? t - 1 : t + 1, xn);

        v4f acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (v4f){bs[g], bs[g], bs[g], bs[g]};
        const float* xa = &xbuf[cur][l15 * SX + 4 * q];
        const float* ha = &hbuf[cur][l15 * SH + 4 * q];
#pragma unroll
        for (int j = 0; j < IN / 16; ++j) {
            const v4f a = *reinterpret_cast<const v4f*>(xa + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[4 * j + e][g], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < HID / 16; ++j) {
            const v4f a = *reinterpret_cast<const v4f*>(ha + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[IN / 4 + 4 * j + e][g], acc[g], 0, 0, 0);
        }
        // cell update; C/D layout: col (unit) = lane & 15, row (sequence) = 4 q + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]), gg = tanh_f(acc[2][r]), og = sigmoid_f(acc[3][r]);
            c[r] = fg * c[r] + ig * gg;
            hbuf[nxt][(4 * q + r) * SH + unit] = og * tanh_f(c[r]);
            const int n = n0 + 4 * q + r;
            if (n < N) {
                const size_t rec = ((size_t)n * L + t) * ndir + dir;
                float* gp = gates + rec * 256 + unit;
                gp[0] = ig; gp[64] = fg; gp[128] = gg; gp[192] = og;
                cells[rec * HID + unit] = c[r];
            }
        }
        if (more) xstore(nxt, xn);
        __syncthreads();
        if (n0 + o_row < N)
            *reinterpret_cast<v4f*>(h + ((size_t)(n0 + o_row) * L + t) * HO + dir * HID + 4 * o_c4) =
                *reinterpret_cast<const v4f*>(&hbuf[nxt][o_row * SH + 4 * o_c4]);
    }
}

// ---------------------------------------------------------------------------------------------- backward through time
// dg_t = gate-pre-activation gradients of step t (from dh_t = dh_out_t + dg_{t'} W_hh of the step processed before, and the
// carried dc); the steps run in the reverse of the forward order.  The recurrent product uses the same tile machinery:
// A = dg_t [16 sequences][256] from LDS, B = W_hh [256][units of this wave] resident, D = dh for exactly the cells a lane owns.
__global__ __launch_bounds__(256, 2) void lstm_train_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ cells,
                                                                const float* __restrict__ dh_out, const float* __restrict__ w_hh,
                                                                float* __restrict__ dg, int N, int L, int ndir)
{
    constexpr int SG = 256 + 8;
    __shared__ __attribute__((aligned(16))) float gbuf[16 * SG];

    const int dir = blockIdx.y;
    const int n0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int unit = 16 * wave + l15;
    const int HO = ndir * HID;

    // B operand of dh = dg W_hh: wb[s] = W_hh[dir][k(s, q)][unit]
    float wb[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) wb[s] = w_hh[((size_t)dir * 256 + kperm(s, q)) * HID + unit];

    // What a step reads of the forward pass, per cell of this lane: the four gates, the previous cell state and the output's
    // gradient.  The NEXT step's set is requested while this step's recurrent product runs (a step is otherwise one global
    // round trip long: 2.4 us at the time axis' 126 steps); the cell state of this step is the previous one of the next.
    struct StepIn { float ig[4], fg[4], gg[4], og[4], cp[4], dh[4]; };
    auto tstep = [&](int step) { return dir ? step : L - 1 - step; };      // reverse of the forward order
    auto load_step = [&](int step, StepIn& in) {
        const int t = tstep(step);
        const int tp = dir ? t + 1 : t - 1;                 // the step the forward pass ran before t (its c is c_prev)
        const bool has_prev = tp >= 0 && tp < L;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int n = n0 + 4 * q + r;
            n = n < N ? n : N - 1;
            const size_t rec = ((size_t)n * L + t) * ndir + dir;
            const float* gp = gates + rec * 256 + unit;
            in.ig[r] = gp[0]; in.fg[r] = gp[64]; in.gg[r] = gp[128]; in.og[r] = gp[192];
            in.cp[r] = has_prev ? cells[(((size_t)n * L + tp) * ndir + dir) * HID + unit] : 0.f;
            in.dh[r] = dh_out[((size_t)n * L + t) * HO + dir * HID + unit];
        }
    };
    float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc_next[4] = {0.f, 0.f, 0.f, 0.f}, ct[4];
    StepIn cur, nxt;
    load_step(0, cur);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int n = n0 + 4 * q + r;
        n = n < N ? n : N - 1;
        ct[r] = cells[(((size_t)n * L + tstep(0)) * ndir + dir) * HID + unit];
    }
    for (int step = 0; step < L; ++step) {
        const int t = tstep(step);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * q + r;
            float di = 0.f, df = 0.f, dgg = 0.f, dob = 0.f;
            if (n < N) {
                const float ig = cur.ig[r], fg = cur.fg[r], gg = cur.gg[r], og = cur.og[r];
                const float dh = cur.dh[r] + dh_rec[r];
                const float tc = tanh_f(ct[r]);
                dob = dh * tc * og * (1.0f - og);
                const float dc = dh * og * (1.0f - tc * tc) + dc_next[r];
                di = dc * gg * ig * (1.0f - ig);
                df = dc * cur.cp[r] * fg * (1.0f - fg);
                dgg = dc * ig * (1.0f - gg * gg);
                dc_next[r] = dc * fg;
                float* op = dg + (((size_t)n * L + t) * ndir + dir) * 256 + unit;
                op[0] = di; op[64] = df; op[128] = dgg; op[192] = dob;
            }
            float* lp = &gbuf[(4 * q + r) * SG + unit];
            lp[0] = di; lp[64] = df; lp[128] = dgg; lp[192] = dob;
        }
        if (step + 1 < L) load_step(step + 1, nxt);
        __syncthreads();
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const float* ga = &gbuf[l15 * SG + 4 * q];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const v4f a = *reinterpret_cast<const v4f*>(ga + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wb[4 * j + e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { dh_rec[r] = acc[r]; ct[r] = cur.cp[r]; }
        cur = nxt;
        __syncthreads();                                     // gbuf is rewritten by the next step
    }
}

}  // namespace

size_t lstm_train_scratch_floats(int N, int L, int IN, int ndir)
{
    const int M = N * L;
    const size_t w = sgemm_tn_scratch_floats(M, 256, IN > HID ? IN : HID), w2 = sgemm_tn_scratch_floats(M, 256, HID), b = colsum_scratch_floats(M, ndir * 256);
    return w > w2 ? (w > b ? w : b) : (w2 > b ? w2 : b);
}

void launch_lstm_train_forward(const float* x, const float* w_ih, const float* w_hh, const float* bias, float* h, float* gates,
                               float* cells, int N, int L, int IN, int ndir, hipStream_t s)
{
    if (N <= 0 || L <= 0) return;
    dim3 grid((N + 15) / 16, ndir), block(256);
    if (IN == 64)
        hipLaunchKernelGGL(lstm_train_fwd_kernel<64>, grid, block, 0, s, x, w_ih, w_hh, bias, h, gates, cells, N, L, ndir);
    else
        hipLaunchKernelGGL(lstm_train_fwd_kernel<128>, grid, block, 0, s, x, w_ih, w_hh, bias, h, gates, cells, N, L, ndir);
}

// dg [N][L][ndir][256] and `scratch` (lstm_train_scratch_floats) are workspace; dx may be null (no gradient wanted).
void launch_lstm_train_backward(const float* x, const float* h, const float* gates, const float* cells, const float* dh,
                                const float* w_ih, const float* w_hh, float* dg, float* scratch, float* dx, float* dw_ih,
                                float* dw_hh, float* db, int N, int L, int IN, int ndir, hipStream_t s)
{
    if (N <= 0 || L <= 0) return;
    const int M = N * L, HO = ndir * HID;
    hipLaunchKernelGGL(lstm_train_bwd_kernel, dim3((N + 15) / 16, ndir), dim3(256), 0, s, gates, cells, dh, w_hh, dg, N, L, ndir);
    for (int d = 0; d < ndir; ++d) {
        const float* dgd = dg + (size_t)d * 256;                   // rows of 256 inside records of ndir 256
        const int ldg = ndir * 256;
        if (dx) launch_sgemm(dgd, ldg, w_ih + (size_t)d * 256 * IN, IN, 0, dx, IN, M, IN, 256, d > 0, nullptr, 0, s);
        // dW_ih = dg^T x, and the bias gradient (column sums of dg) from the same pass
        launch_sgemm_tn(dgd, ldg, x, IN, dw_ih + (size_t)d * 256 * IN, db + (size_t)d * 256, scratch, M, 256, IN, L, 0, s);
        // dW_hh = dg^T h_prev: the forward pass of direction 0 read h_{t-1}, direction 1 h_{t+1}
        launch_sgemm_tn(dgd, ldg, h + (size_t)d * HID, HO, dw_hh + (size_t)d * 256 * HID, nullptr, scratch, M, 256, HID, L, d ? 1 : -1, s);
    }
}

}  // namespace bsrnn
