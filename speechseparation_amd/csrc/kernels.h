// Internal launcher interface between api.hip and the gfx950 kernels.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsrnn {

constexpr int HID = 64;       // band_features (bsrnn.py:60)
constexpr int NFFT = 2048;    // infer.py:31
constexpr int HOPS = 1024;
constexpr int NBINS = 1025;
constexpr int F2 = 2050;      // interleaved re/im columns

// ------------------------------------------------------------------ grouped linear layers
// One job = one nn.Linear of one band.  A launch runs every job of one "layer slot" of the
// per-band MLP chains over all M = C*T frame rows.
struct GemmJob {
    const float* W;      // [N][K] row-major (torch Linear layout), device, K padded to a multiple of 8 floats
    const void* Wp;      // the same matrix split into three bf16 planes [3][N][K] (gemm.hip, "planes" kernel)
    const float* bias;   // [N]
    int N, K;            // K may be 0: y = bias (TrainableConstantModule, bsrnn.py:12-24)
    int x_off;           // column offset of the job's input inside an X row
    int y_off;           // column offset of the output inside a Y row
    int r_off;           // column offset inside the residual row (EPI_RES, EPI_MASK)
    int m_off;           // column offset inside the multiplier / mask-tap row (EPI_MASK)
    int wrow;            // fp16x2 slab format: 16-bit elements between consecutive weight rows of Wp
    int xs_off, ys_off;  // slab-format activations (GemmLaunch::Xs / Ys): 16-bit element offset of the job's first slab in a row
};

enum GemmEpilogue {
    EPI_LINEAR = 0,      // y = acc + b
    EPI_LEAKY = 1,       // y = leaky_relu(acc + b, 0.01)
    EPI_RES = 2,         // y = acc + b + R                         (NormRNNResidual, bsrnn.py:84-86)
    EPI_MASK = 3         // mask = acc + b + R ; y = Mul * mask      (bsrnn.py:425, :441)
};

struct GemmLaunch {
    const GemmJob* jobs;     // device array
    const int2* tiles;       // device array [n_tiles]: (job index, column-tile index inside the job)
    int n_tiles;
    int tile_n;              // column-tile width of this launch: 64 or 128
    int mchunk;              // m-tiles per XCD-pinned chunk (filled in by launch_gemm)
    const float* X; int ldx;
    float* Y; int ldy;
    const float* R; int ldr;
    const float* Mul; int ldm;
    float* tap; int ldt;     // optional mask tap (EPI_MASK), may be null
    int M;
    int epilogue;
    // split-precision ("planes") kernel only: activations as three bf16 planes, plane p of an [M][ld] matrix
    // starts p * plane elements after the first; columns use the same offsets and ld as X / Y
    const void* Xp; size_t xp_plane;   // input planes; null: X is fp32 and is split on the fly
    void* Yp; size_t yp_plane;         // output planes (written when out_mode & 2)
    int out_mode;                      // bit 0: write fp32 Y, bit 1: write planes Yp, bit 2: write slabs Ys
    // fp16x2 mode, activations pre-split by the producing layer's epilogue in the weights' slab format
    // [row][K32 / 32][2 pieces][32] fp16 (both pieces of a 32-deep slab of a row share one 128-byte line; columns from
    // N up to the next multiple of 32 are written as zeros): the consumer copies them global -> LDS by LDS-DMA.
    const void* Xs; int ldxs;          // input slabs (null: X is fp32 and is split on the fly); ld in 16-bit elements
    void* Ys; int ldys;                // output slabs (written when out_mode & 4)
    int* range_flag;                   // fp16x2: set to 1 when an operand exceeded the fp16 range (may be null)
};
void launch_gemm(const GemmLaunch& g, hipStream_t stream);
// How the Linear layers are evaluated (environment BSRNN_GEMM = f32 | fp16x2 | bf16x3 | fp16, read once per process).
// fp16 = plain 16-bit operands, one MFMA term, fp32 accumulate (the reduced-precision configuration, not the default).
// f32: no 16-bit weights; fp16x2 / fp16: slab-interleaved fp16 pieces; bf16x3: three bf16 planes.
enum GemmMode { GEMM_F32 = 0, GEMM_FP16 = 1, GEMM_FP16X2 = 2, GEMM_BF16X3 = 3 };
int gemm_mode();
constexpr int GEMM_BM = 128;

// ------------------------------------------------------------------ dual-path LSTM kernels
// Band-axis BLSTM layer (both directions in one launch): N sequences of length L.
//   xin  [N][L][IN]           IN = 64 (layer 0, fc_in folded into W_ih) or 128 (layer 1)
//   hout [N][L][128]          forward half at [0,64), backward half at [64,128)
//   wpk  packed [2 dir][4 wave][(IN+64)/4 step][4 gate][64 lane], bias [2][256]
//   wpk16 (LSTM_FP16X2): the same matrix as two fp16 pieces in the f16 MFMA's B-operand order,
//         [2 dir][4 wave][(IN+64)/32 blk][4 gate][2 piece][64 lane][8]
void launch_band_lstm(const float* xin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      int N, int L, int IN, int* range_flag, hipStream_t stream);
// How the recurrent layers evaluate their gate products (environment BSRNN_LSTM = f32 | fp16x2, read once).
enum LstmMode { LSTM_F32 = 0, LSTM_FP16X2 = 2 };
int lstm_mode();
// Time-axis LSTM, both layers pipelined in one launch, causal with state carry.
//   zin/hout [R][T][K][64]; sequences n = r*K + k;  wpk packed [2 layer][4 wave][128 k][64 lane]
//   state_in/out [2 (h,c)][2 layer][R*K][64] or null
//   wpk16 (LSTM_FP16X2): two fp16 pieces in B-operand order, [2 layer][4 wave][4 blk][4 gate][2 piece][64 lane][8]
void launch_time_lstm(const float* zin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      const float* state_in, float* state_out, int R, int T, int K, int* range_flag, hipStream_t stream);

// ------------------------------------------------------------------ STFT / iSTFT / layout
struct FftTables {           // device tables, built once per context (double precision on host)
    const float2* tw1024;    // exp(-2 pi i k / 1024), k < 1024
    const float2* tw2048;    // exp(-2 pi i k / 2048), k <= 1024
    const float* hann;       // periodic Hann(2048)
    const float* inv_env;    // 1 / (w^2[i] + w^2[i+1024]), i < 1024   (torch.istft envelope)
    const float* inv_wsum;   // 1 / (w[i] + w[i+1024]),   i < 1024   (infer-streaming.py:145)
    // band-padded spectrum layout: bin k's (re, im) sit at columns colmap[k], colmap[k]+1 of a row of ld floats;
    // every band starts at a multiple of 4 floats so that the first Linear layer can use 16-byte loads
    const int* colmap;       // [1025]
    int ld;
};
// wave [R][n] -> X frame-major [R*T][ld] (re/im interleaved, band-padded columns), reflect padding, Hann.
void launch_stft(const FftTables& tb, const float* wave, float* X, int R, int64_t n, int T, hipStream_t s);
// Y frame-major [R*T][ld] -> wave_out [R][(T-1)*1024]: inverse real FFT, synthesis window, overlap-add / envelope, fused
void launch_istft(const FftTables& tb, const float* Y, float* out, int R, int T, hipStream_t s);
// [C][2050][T] (reference layout) <-> [C*T][ld] (band-padded)
void launch_to_frame_major(const FftTables& tb, const float* x, float* xf, int C, int T, hipStream_t s);
void launch_from_frame_major(const FftTables& tb, const float* yf, float* y, int C, int T, hipStream_t s);
// streaming DSP, one frame per row (infer-streaming.py:116-145)
//   analysis: buf [C][2048] slides by 1024, appends chunk [C][1024]; X [C][ld] = rfft(buf*hann)
//   synthesis: s = irfft(mix(Y, X)); out = (s[0:1024] + prev[1024:2048]) * inv_wsum; prev = s
void launch_stream_analysis(const FftTables& tb, float* buf, const float* chunk, float* X, int C, hipStream_t s);
void launch_stream_synthesis(const FftTables& tb, const float* Y, const float* X, const float* mix_dev,
                             float* prev, float* out, int C, hipStream_t s);

// ------------------------------------------------------------------ validation metrics (metrics.hip)
// Per-workgroup partial sums in double; the host adds them.  est [R][n_est]; speech, mix [R][n_in] (first n_est used).
constexpr int METRIC_TIME_Q = 7;     // sum s^2, (x-s)^2, x*s, x^2, |x-s|, m^2, (m-x)^2
int metric_time_chunks(int64_t n_est);                                   // workgroups per row of the two row-wise kernels
void launch_metric_time(const float* est, const float* speech, const float* mix, int R, int64_t n_est, int64_t n_in,
                        double* part /* [R][chunks][7] */, hipStream_t s);
void launch_metric_sisdr(const float* est, const float* speech, const float* alpha /* [R] */, int R, int64_t n_est, int64_t n_in,
                         double* part /* [R][chunks][2] */, hipStream_t s);
void launch_metric_input_sdr(const float* speech, const float* mix, int R, int64_t n, double* part /* [blocks] */, int blocks, hipStream_t s);
void launch_metric_freq(const FftTables& tb, const float* Yf, const float* Sf, int M, double* part /* [blocks][2] */, int blocks, hipStream_t s);

}  // namespace bsrnn
