/*
 * bsrnn_hip.h -- C ABI of the MI355X-native BSRNN separation path (libbsrnn_hip.so).
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no torch / C++ types.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repo phhusson/SpeechSeparation).  Host bindings (ctypes, the LADSPA plugin,
 * or a libtorch/pybind stub in the reference itself) bind exactly these symbols; see
 * INTEGRATION.md for the reference-side stubs.
 *
 * Conventions
 *   - all tensors are contiguous float32;
 *   - "dev" pointers are device (HIP) pointers on the context's device, "host" pointers are
 *     ordinary host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls enqueue work
 *     and return without synchronising unless stated otherwise;
 *   - return value 0 = success, otherwise a BSRNN_E* code; bsrnn_last_error() gives text;
 *   - nothing here ever falls back to a CPU implementation.
 *
 * Concurrency contract
 *   - ONE call at a time per context.  A context owns one workspace (activations of the call in flight) and one weight
 *     arena; every compute entry point, bsrnn_commit_params and the stream calls of its bsrnn_stream objects count as
 *     calls on it.  A call that arrives from a second host thread while another is inside the library is refused with
 *     BSRNN_ESTATE (it never races).  Use one context per host thread for concurrent work; contexts share nothing.
 *   - Calls enqueue on the caller's HIP stream.  Consecutive calls on the SAME stream are ordered by the stream.  A call
 *     on a DIFFERENT stream than the previous call first waits, on the host, for that previous stream to drain (the two
 *     would otherwise overlap on the workspace): correct, but it serialises - keep a context on one stream.
 *   - A larger call may regrow the workspace and bsrnn_commit_params rebuilds the arena; both wait for the device first,
 *     and bsrnn_stream objects notice (a generation counter) and re-capture their hipGraph at their next step.
 *   - bsrnn_destroy with bsrnn_stream objects still alive only retires the context (further calls: BSRNN_ESTATE); the
 *     memory is released when the last of its streams is destroyed.
 *   - C = rows of dim 0 (utterance-channels), T / L = STFT frames, F2 = 2050 interleaved
 *     re/im columns, K = number of bands including the zero-width band, H = 64.
 */
#ifndef BSRNN_HIP_H
#define BSRNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BSRNN_ABI_VERSION 2      /* 2: range policy (bsrnn_set_range_policy), training entry points, bsrnn_forward_chunk refuses aliased state */

#define BSRNN_OK          0
#define BSRNN_EARG        1   /* bad argument / shape */
#define BSRNN_ESTATE      2   /* call out of order (e.g. compute before commit) */
#define BSRNN_EHIP        3   /* HIP runtime error */
#define BSRNN_EIO         4   /* weight file problem */
#define BSRNN_ENOKEY      5   /* unknown parameter key */
#define BSRNN_ERANGE      6   /* an activation left the range of the split-precision mode (|a| > 65504), see below */

typedef struct bsrnn_ctx bsrnn_ctx;
typedef struct bsrnn_stream bsrnn_stream;

int         bsrnn_abi_version(void);
const char* bsrnn_last_error(void);
/* How the matrix products are evaluated, e.g. "gemm=fp16x2 lstm=fp16x2" (measurement / logging only).
 * Inputs, outputs, state and accumulation are float32 in every mode; the modes differ in which matrix pipe
 * carries the products: "f32" = v_mfma_f32_*_f32 (exact fp32 fma chains), "fp16x2" = fp32 operands
 * split into 2 fp16 pieces and multiplied on the 16-bit matrix cores with fp32 accumulation
 * (error at the level of fp32 rounding noise).  Selected once per process by the environment
 * variables BSRNN_GEMM (f32 | fp16x2 | fp16) and BSRNN_LSTM (f32 | fp16x2); default fp16x2 for both.
 * BSRNN_GEMM=fp16 is the one REDUCED-precision mode (plain fp16 operands in the Linear layers, one MFMA term,
 * fp32 accumulation: ~1e-3 of the output range); it exists for the "16-bit compute" benchmark configuration.
 * Range: the fp16x2 mode represents operands up to |a| = 65504 (spectra of audio in [-1, 1] stay below 1024).  A larger
 * finite activation cannot be represented; the kernels notice (a host-visible guard word).  What happens then is the
 * context's RANGE POLICY (bsrnn_set_range_policy):
 *   - BSRNN_RANGE_EXACT (default): every model entry point (bsrnn_forward, _forward_chunk, _forward_recurrent, _dual_path,
 *     _separate, _stream_step, _evaluate) waits for its own kernels before it returns, looks at the guard and, if it is set,
 *     runs the same call again on the library's exact-fp32 kernels (fp32 weights are always resident; no range limit) from
 *     the same inputs / the same starting state: rc 0 always comes with correct numbers, as from the reference.  The price
 *     is one stream synchronisation per call.
 *   - BSRNN_RANGE_DEFERRED: the entry points return without waiting (launch pipelines, the benchmark loop).  A violation is
 *     reported late: the NEXT call on the context, bsrnn_sync or bsrnn_stream_get_state fails with BSRNN_ERANGE, once, and
 *     the results of the call that caused it are invalid (not-a-number or nonsense, never silently plausible saturated
 *     values: nothing is clamped).  The caller repeats the work under the default policy.
 *   bsrnn_stream_step_host always behaves as under BSRNN_RANGE_EXACT (it waits for its kernels anyway).
 * NaN / Inf inputs are not range errors: they come out as NaN, as they do from the reference. */
const char* bsrnn_compute_mode(void);

/* ---- construction -------------------------------------------------------------------
 * Replaces `BSRNN()` (bsrnn.py:328-376).  `widths` are band widths in bins, in order,
 * including the trailing zero-width band (generate_bandsplits()[0], bsrnn.py:247-326);
 * sum(widths) must be 1025.  The band table is data: pass a different one for the
 * 41-band variant.  `device` is the HIP device ordinal, or -1 for a HOST-ONLY context: parameter inventory,
 * bsrnn_set_param / bsrnn_get_param and bsrnn_load_weights_file (as a validator) work without a GPU, every compute
 * entry point returns BSRNN_ESTATE (tools/convert_weights.py, CPU tests). */
int  bsrnn_create(int device, const int32_t* widths, int32_t n_bands, bsrnn_ctx** out);
void bsrnn_destroy(bsrnn_ctx* ctx);
int  bsrnn_n_bands(const bsrnn_ctx* ctx);
int  bsrnn_device(const bsrnn_ctx* ctx);
/* Range policy of the split-precision modes (see bsrnn_compute_mode above); default BSRNN_RANGE_EXACT. */
#define BSRNN_RANGE_DEFERRED 0
#define BSRNN_RANGE_EXACT    1
int  bsrnn_set_range_policy(bsrnn_ctx* ctx, int32_t policy);
int  bsrnn_get_range_policy(const bsrnn_ctx* ctx);
/* 1 when the committed context runs the per-band MLP chains (bsrnn.py:404-415, :420-443) as fused launches (one workgroup =
 * one band's five Linear layers, intermediates in LDS; the default), 0 when it runs one grouped launch per layer: the
 * exact-fp32 mode, BSRNN_MLP=layers (A/B), or a band table with a band too wide for the fused kernel's LDS image
 * (more than 768 columns).  Same arithmetic, bit-identical results either way. */
int  bsrnn_mlp_fused(const bsrnn_ctx* ctx);
/* How this context runs the dual path of large calls (bsrnn.py:352-356, the four recurrent blocks): 1 = overlapped - the second band
 * block is launched beside the first (causal) time-axis launch and the mask MLPs beside the second, on an auxiliary stream of the
 * context, each workgroup waiting for the frames of its own rows (results are bit-identical to 0); 0 = one launch after the other
 * (environment BSRNN_OVERLAP=0); 2 = switched off for good after a consumer's bounded wait expired once (that call was run again launch
 * after launch before it returned; under the 'deferred' range policy it is reported by the next call instead).  Calls of fewer than
 * 32 frames or 8 rows x 12 bands never overlap.  Measurement / test support. */
int  bsrnn_overlap_state(const bsrnn_ctx* ctx);
/* Measurement / debugging: copy the first `nfloats` floats of one of the context's internal activation buffers of the LAST call to the
 * host (synchronises the device).  which: 0 = Z0, 1 = Z1 (the dual-path tensor's two buffers, [rows*frames][K][64]), 2 = HB1 (the
 * band blocks' fc shares, [rows*frames][K][2][64]), 3 = P, 4 = Yf (band-padded rows of LDP floats). */
int  bsrnn_debug_peek(bsrnn_ctx* ctx, int32_t which, float* host_out, int64_t nfloats);
/* Process-wide counts of what the library has done that does not belong on a real-time thread: which = 0 device / pinned / stream
 * allocations, 1 stream captures begun, 2 graph instantiations, 3 graph launches.  A bsrnn_stream does all of the first three in
 * bsrnn_stream_create (as the reference's plugin does in its constructor, speech-ladspa-onnx.cpp:55-120); its step calls leave them
 * unchanged (tests/test_gpu_entrypoints.py drives the LADSPA plugin's run() and checks). */
long long bsrnn_debug_counter(int32_t which);

/* ---- parameters ---------------------------------------------------------------------
 * Replaces `load_state_dict` (infer.py:19, infer-streaming.py:46): parameters are named
 * by the reference's state_dict keys and given as host float32
 * in torch layout (Linear weight [out,in]).  bsrnn_commit_params() folds/packs/uploads
 * and must be called after the last bsrnn_set_param() and before any compute call. */
int  bsrnn_param_count(const bsrnn_ctx* ctx);
int  bsrnn_param_info(const bsrnn_ctx* ctx, int32_t index, const char** key,
                      int64_t* dim0, int64_t* dim1, int32_t* ndim);
int  bsrnn_set_param(bsrnn_ctx* ctx, const char* key, const float* host_data, int64_t numel);
int  bsrnn_get_param(const bsrnn_ctx* ctx, const char* key, float* host_out, int64_t numel);
int  bsrnn_commit_params(bsrnn_ctx* ctx);
/* Flat weight file (speechseparation_amd/weights.py) -- replaces the ONNX file that
 * speech-ladspa-onnx.cpp:73 opens.  Does set_param for every tensor, then commit.  Every tensor must carry a key of
 * the inventory and exactly its rank and dimensions (BSRNN_ENOKEY / BSRNN_EIO otherwise; sizes are never taken from
 * the file, and nothing throws across this boundary). */
int  bsrnn_load_weights_file(bsrnn_ctx* ctx, const char* path);

/* I/O signature of the one-frame model, as the reference's exported ONNX file declares it (infer-streaming.py:74
 * `torch.onnx.export(..., (x, state))`; read back by the plugin, speech-ladspa-onnx.cpp:82-111, which sizes its state
 * buffers from the shape of the input named "state.0"): index 0 "x.0" [C, 2050] and 1 "state.0" [4, 2, C*K, 64] are
 * inputs, 2 "y.0" [C, 2050] and 3 "new_state.0" [4, 2, C*K, 64] outputs.  Hosts written against that session API can
 * size and name their tensors from here instead of from a model file; C is the caller's row count (2 in the plugin). */
int  bsrnn_io_count(void);
int  bsrnn_io_info(const bsrnn_ctx* ctx, int32_t index, int32_t C, const char** name, int32_t* is_input,
                   int64_t dims[4], int32_t* ndim);

/* ---- model entry points ---------------------------------------------------------------
 * bsrnn_forward            = BSRNN.forward            (bsrnn.py:385-443)
 *     x_dev [C, 2050, T] -> y_dev [C, 2050, T]   (y = x * mask; x is not modified)
 * bsrnn_forward_recurrent  = BSRNN.forward_recurrent  (bsrnn.py:445-510)
 *     x_dev [C, 2050], state_in_dev [4, 2, C*K, 64] -> y_dev [C, 2050], state_out_dev (same
 *     shape; a DIFFERENT buffer under BSRNN_RANGE_EXACT - BSRNN_EARG otherwise: a call that left the
 *     fp16 range is run again from state_in_dev -, may alias it under BSRNN_RANGE_DEFERRED; the same
 *     holds for bsrnn_forward_chunk and bsrnn_dual_path).  State slabs: 0/1 = h/c of lstms.1, 2/3 = h/c of
 *     lstms.3; dim 1 = LSTM layer; dim 2 = c*K + k.
 * bsrnn_forward_chunk      = L consecutive forward_recurrent steps in one call (BASELINE.json
 *     config 3): x_dev [C, 2050, L], state carried causally; L = 1 equals forward_recurrent.
 * mask_dev (optional, may be NULL): receives the mask [C, 2050, T] (bsrnn.py:425-432). */
int  bsrnn_forward(bsrnn_ctx* ctx, const float* x_dev, float* y_dev, float* mask_dev,
                   int32_t C, int32_t T, void* stream);
int  bsrnn_forward_recurrent(bsrnn_ctx* ctx, const float* x_dev, const float* state_in_dev,
                             float* y_dev, float* state_out_dev, int32_t C, void* stream);
int  bsrnn_forward_chunk(bsrnn_ctx* ctx, const float* x_dev, const float* state_in_dev,
                         float* y_dev, float* state_out_dev, int32_t C, int32_t L, void* stream);

/* `self.lstms(z)` alone (bsrnn.py:352-356, :417): z_dev [C, T, K, 64] -> z_out_dev.  state
 * pointers may be NULL (zero initial state, final state discarded). */
int  bsrnn_dual_path(bsrnn_ctx* ctx, const float* z_dev, float* z_out_dev,
                     const float* state_in_dev, float* state_out_dev,
                     int32_t C, int32_t T, void* stream);

/* ---- training step, part 1: the recurrent layers ------------------------------------------
 * The reference trains with torch.autograd (train.py:97-115: `loss.backward()`, `optimizer.step()`); a replacement has to
 * provide the backward pass itself.  These two calls are one `nn.LSTM` layer of NormRNNResidual (bsrnn.py:66-72) with
 * `ndir` directions (2 = bidirectional: direction 1 uses torch's `_reverse` weights and runs the sequence backwards) over
 * N sequences of L steps, zero initial state, exact fp32:
 *   the band-axis BLSTM is N = C*T, L = K, ndir = 2 (bsrnn.py:144-150), the time-axis LSTM N = C*K, L = T, ndir = 1 with the
 *   rows gathered as in bsrnn.py:110-113.
 * All pointers are device memory.  Weights in torch layout: w_ih [ndir][256][IN] (IN = 64 | 128), w_hh [ndir][256][64],
 * gate order i,f,g,o; bias [ndir][256] = b_ih + b_hh.
 *   forward : x [N][L][IN] -> h [N][L][ndir*64]; gates [N][L][ndir][256] (after sigmoid / tanh) and cells [N][L][ndir][64]
 *             are what backward reads.
 *   backward: dh [N][L][ndir*64] = gradient of the loss w.r.t. h  ->  dx [N][L][IN] (NULL: not wanted), dw_ih, dw_hh in the
 *             weights' layouts, db [ndir][256] (= the gradient of b_ih and of b_hh).  Gradients are overwritten, not
 *             accumulated; sums run in a fixed order (bit-reproducible).  Workspace: a grow-only buffer of the context. */
int  bsrnn_lstm_train_forward(bsrnn_ctx* ctx, const float* x_dev, const float* w_ih_dev, const float* w_hh_dev,
                              const float* bias_dev, float* h_dev, float* gates_dev, float* cells_dev,
                              int32_t N, int32_t L, int32_t IN, int32_t ndir, void* stream);
int  bsrnn_lstm_train_backward(bsrnn_ctx* ctx, const float* x_dev, const float* h_dev, const float* gates_dev,
                               const float* cells_dev, const float* dh_dev, const float* w_ih_dev, const float* w_hh_dev,
                               float* dx_dev, float* dw_ih_dev, float* dw_hh_dev, float* db_dev,
                               int32_t N, int32_t L, int32_t IN, int32_t ndir, void* stream);

/* nn.Linear (+ LeakyReLU(0.01) when leaky != 0) of the per-band MLPs and of fc / fc_in (bsrnn.py:333-376, :69, :74) for the
 * training step: y = act(x W^T + b), x [M, K] with row stride ldx (a band is a column block of a wider row), W [N, K] torch
 * layout, y [M, N] with row stride ldy.  Backward: from dy (row stride lddy; and y when leaky) -> dx [M, K] row stride lddx
 * (NULL: not wanted), dw [N, K], db [N]; overwritten, bit-reproducible.  All device pointers, exact fp32. */
int  bsrnn_linear_train_forward(bsrnn_ctx* ctx, const float* x_dev, int32_t ldx, const float* w_dev, const float* b_dev,
                                float* y_dev, int32_t ldy, int32_t M, int32_t K, int32_t N, int32_t leaky, void* stream);
int  bsrnn_linear_train_backward(bsrnn_ctx* ctx, const float* x_dev, int32_t ldx, const float* w_dev, const float* y_dev,
                                 int32_t ldy, const float* dy_dev, int32_t lddy, float* dx_dev, int32_t lddx,
                                 float* dw_dev, float* db_dev, int32_t M, int32_t K, int32_t N, int32_t leaky, void* stream);

/* The same for n layers that share the row count M - the same Linear layer of all bands (bsrnn.py:406-411, :423-425 loop over
 * the bands) - in grouped launches: host arrays [n] of device pointers / leading dimensions / K / N.  dx_dev[i] may be NULL. */
int  bsrnn_linear_group_train_forward(bsrnn_ctx* ctx, int32_t n, const float* const* x_dev, const int32_t* ldx,
                                      const float* const* w_dev, const float* const* b_dev, float* const* y_dev,
                                      const int32_t* ldy, const int32_t* K, const int32_t* N, int32_t M, int32_t leaky, void* stream);
int  bsrnn_linear_group_train_backward(bsrnn_ctx* ctx, int32_t n, const float* const* x_dev, const int32_t* ldx,
                                       const float* const* w_dev, const float* const* y_dev, const int32_t* ldy,
                                       const float* const* dy_dev, const int32_t* lddy, float* const* dx_dev, const int32_t* lddx,
                                       float* const* dw_dev, float* const* db_dev, const int32_t* K, const int32_t* N,
                                       int32_t M, int32_t leaky, void* stream);

/* torch.optim.AdamW(model.parameters(), lr, weight_decay) of train.py:50, one tensor per call: p, m (exp_avg), v (exp_avg_sq)
 * updated in place from the gradient g; `step` counts from 1 (bias correction).  n floats each, device pointers. */
int  bsrnn_adamw_step(bsrnn_ctx* ctx, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* The same update for many parameter tensors at once (80 per launch, pointers passed in the kernel arguments): host arrays
 * [n_tensors] of device pointers and of element counts. */
int  bsrnn_adamw_step_multi(bsrnn_ctx* ctx, float* const* p_dev, const float* const* g_dev, float* const* m_dev, float* const* v_dev,
                            const int64_t* sizes, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, void* stream);

/* The same with the step count and the learning rate in DEVICE memory: state_dev = {float lr, float bc1, float bc2s, int32 step}
 * (16 bytes; the caller initialises lr and step - the number of steps taken so far -, the call advances step by one in stream
 * order and derives the bias corrections from it on the device).  Nothing that changes from iteration to iteration travels by
 * value, so a whole training iteration (train.py:97-115: forward, backward, this call) can be captured once into a hipGraph and
 * replayed; the learning rate of a schedule is a 4-byte copy into state_dev between replays. */
int  bsrnn_adamw_step_multi_dev(bsrnn_ctx* ctx, float* const* p_dev, const float* const* g_dev, float* const* m_dev, float* const* v_dev,
                                const int64_t* sizes, int32_t n_tensors, float* state_dev, float beta1, float beta2, float eps,
                                float weight_decay, void* stream);

/* ---- the STFT sandwich of the callers --------------------------------------------------
 * bsrnn_stft   = infer.py:29-33 (dup. m_dataset.py:187-190): wave_dev [R, n] ->
 *                x_dev [R, 2050, T], T = 1 + n/1024; periodic Hann 2048, hop 1024,
 *                center/reflect, onesided, re/im interleaved.  n must be > 1024.
 * bsrnn_istft  = infer.py:35-37: y_dev [R, 2050, T] -> wave_out_dev [R, (T-1)*1024].
 * bsrnn_separate = the whole sandwich fused on the device (frame-major internally, no
 *                [C,2050,T] round trips): wave_dev [R, n] -> wave_out_dev [R, (T-1)*1024]. */
int  bsrnn_stft(bsrnn_ctx* ctx, const float* wave_dev, float* x_dev, int32_t R, int64_t n, void* stream);
int  bsrnn_istft(bsrnn_ctx* ctx, const float* y_dev, float* wave_out_dev, int32_t R, int32_t T, void* stream);
/* Backward of bsrnn_istft for the training step (the loss of m_dataset.py:211-216 has a waveform term): dwave_dev
 * [R, (T-1)*1024] = gradient w.r.t. the waveform -> dy_dev [R, 2050, T] = gradient w.r.t. the spectrum (the transpose of
 * torch.istft: zero-padded windowed FFT of dwave / envelope, one-sided bins weighted 1, 2, ..., 2, 1 over 2048, no gradient
 * for the imaginary parts of bins 0 and 1024). */
int  bsrnn_istft_backward(bsrnn_ctx* ctx, const float* dwave_dev, float* dy_dev, int32_t R, int32_t T, void* stream);
int  bsrnn_separate(bsrnn_ctx* ctx, const float* wave_dev, float* wave_out_dev, int32_t R, int64_t n, void* stream);

/* ---- validation metrics (m_dataset.py:182-226 `infer` + `train_infer`, infer.py:44-47) ------
 * bsrnn_evaluate: mix_dev [R, n] (the mixture, sample[0]) and speech_dev [R, n] (the clean target,
 * sample[1]) -> separates the mixture (bsrnn_separate) and returns, in metrics_host[BSRNN_N_METRICS]:
 *   LOSS          L1(x_time, s_time) + L1(Re X, Re S) + L1(Im X, Im S), each a mean over its elements
 *                 (train.py:54 L1Loss; no discriminator term), S = STFT of the clean signal;
 *   SDR           mean over rows of 10 log10((sum s^2 + 1e-9) / (sum (x - s)^2 + 1e-9)), s cut to len(x);
 *   INPUT_SDR     the reference's `sdr2` exactly as written: its sample tensors are [1, R, n], so the sums
 *                 run over the R rows and the mean over the n sample positions (m_dataset.py:219-222);
 *   SISDR         scale-invariant SDR of torchmetrics (zero_mean = False, eps = float32 epsilon), mean over rows;
 *   L1_TIME / L1_RE / L1_IM   the three loss terms;
 *   SEPARATION_DB 10 ln(sum mix^2 / sum (mix - x)^2) over all rows (natural log, as infer.py:47 prints it).
 * est_out_dev (optional) receives the separated signal [R, (T-1)*1024].  Synchronous: returns when the
 * numbers are on the host.  Reductions accumulate in double on the device; rows are independent, so any R
 * works (the reference's un-interleave hard-codes 2 rows, m_dataset.py:192). */
#define BSRNN_M_LOSS          0
#define BSRNN_M_SDR           1
#define BSRNN_M_INPUT_SDR     2
#define BSRNN_M_SISDR         3
#define BSRNN_M_L1_TIME       4
#define BSRNN_M_L1_RE         5
#define BSRNN_M_L1_IM         6
#define BSRNN_M_SEPARATION_DB 7
#define BSRNN_N_METRICS       8
int  bsrnn_evaluate(bsrnn_ctx* ctx, const float* mix_dev, const float* speech_dev, int32_t R, int64_t n,
                    float* est_out_dev, double* metrics_host, void* stream);

/* ---- streaming (infer-streaming.py:84-147; speech-ladspa-onnx.cpp:171-267) -------------
 * A bsrnn_stream owns, on the device, the sliding 2048-sample analysis buffer, the LSTM
 * state [4,2,C*K,64] and the previous synthesis frame for C rows.
 * bsrnn_stream_create does ALL first-use work (allocations, loading every kernel, and - with BSRNN_STREAM_GRAPH=1 - capturing
 * and instantiating the step's hipGraphs: two throw-away steps on zero input, carry zeroed again afterwards), so that the first
 * bsrnn_stream_step* call costs what every later one costs.
 * bsrnn_stream_step: chunk_dev [C, 1024] -> out_dev [C, 1024] (delayed by one chunk):
 *     rfft(buf*hann) -> forward_recurrent -> irfft -> 2-slot overlap-add / sum(window).
 * bsrnn_stream_step_host: same with host buffers, synchronous (used by the LADSPA plugin);
 *     `mix` applies the plugin's wet/dry control on the spectrum (speech-ladspa-onnx.cpp:
 *     215-226): mix >= 0: mix*y + (1-mix)*x ; mix < 0: x + mix*y.  Use mix = 1 for
 *     infer-streaming.py semantics. */
int  bsrnn_stream_create(bsrnn_ctx* ctx, int32_t C, bsrnn_stream** out);
void bsrnn_stream_destroy(bsrnn_stream* s);
int  bsrnn_stream_reset(bsrnn_stream* s, void* stream);
int  bsrnn_stream_step(bsrnn_stream* s, const float* chunk_dev, float* out_dev, float mix, void* stream);
int  bsrnn_stream_step_host(bsrnn_stream* s, const float* chunk_host, float* out_host, float mix);
int  bsrnn_stream_get_state(bsrnn_stream* s, float* state_host /* [4,2,C*K,64] */);

/* ---- measurement support ---------------------------------------------------------------
 * bsrnn_set_profiling(ctx, mask): bit i of `mask` brackets stage i of every compute call with
 * hipEvents on the call's stream (mask < 0 = all stages, 0 = off; each bracket costs ~10 us of
 * stream time, so time-critical runs enable only the stage they report);
 * bsrnn_stage_times() synchronises on the last call and returns per-stage elapsed
 * milliseconds (accumulated since the last reset) and launch counts.  Stage names:
 * bsrnn_stage_name(i).  Used by bench.py for the live roofline figure. */
int         bsrnn_set_profiling(bsrnn_ctx* ctx, int32_t on);
int         bsrnn_stage_count(void);
const char* bsrnn_stage_name(int32_t i);
int         bsrnn_stage_times(bsrnn_ctx* ctx, double* ms_out, int64_t* launches_out, int32_t reset);

/* ---- device memory helpers for hosts without a HIP binding (the plugin, C tests) -------- */
int  bsrnn_dev_alloc(bsrnn_ctx* ctx, int64_t nbytes, void** out);
int  bsrnn_dev_free(bsrnn_ctx* ctx, void* p);
int  bsrnn_copy_h2d(bsrnn_ctx* ctx, void* dst_dev, const void* src_host, int64_t nbytes);
int  bsrnn_copy_d2h(bsrnn_ctx* ctx, void* dst_host, const void* src_dev, int64_t nbytes);
int  bsrnn_sync(bsrnn_ctx* ctx, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BSRNN_HIP_H */
