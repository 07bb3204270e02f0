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
    const void* Wp;      // the same matrix as two fp16 pieces, slab-interleaved [N][K32 / 32][2][32] (split_host.h)
    const float* bias;   // [N]
    int N, K;            // K may be 0: y = bias (TrainableConstantModule, bsrnn.py:12-24)
    int x_off;           // column offset of the job's input inside an X row
    int y_off;           // column offset of the output inside a Y row
    int r_off;           // column offset inside the residual row (EPI_RES, EPI_MASK)
    int m_off;           // column offset inside the multiplier / mask-tap row (EPI_MASK)
    int wrow;            // 16-bit elements between consecutive weight rows of Wp
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
    int* range_flag;                   // fp16x2: set to 1 when an operand exceeded the fp16 range (may be null)
};
void launch_gemm(const GemmLaunch& g, hipStream_t stream);
// Small-M path (gemv.hip): the same launch on the vector ALU in exact fp32, for calls of a few frame rows (the one-frame
// streaming step: C rows).  api.hip uses it for every per-layer launch of a call with C * L <= GEMV_MAX_FRAME_ROWS.
constexpr int GEMV_MAX_FRAME_ROWS = 4;
void launch_gemv(const GemmLaunch& g, hipStream_t stream);
// How the Linear layers are evaluated (environment BSRNN_GEMM = f32 | fp16x2 | fp16, read once per process).
// fp16 = plain 16-bit operands, one MFMA term, fp32 accumulate (the reduced-precision configuration, not the default).
// The fp32 weights are always resident beside the fp16 pieces: a call whose operands left the fp16x2 range is re-run on
// the exact-fp32 kernels (set_force_f32, per host thread) by the synchronous entry points of api.hip.
// bf16 = plain bf16 operands, one MFMA term, in the fused MLP chains (BASELINE config 2 as it is named; no range limit, 8 significant
// bits); the few launches outside the chains (a band too wide for the LDS image, the block fc of the BSRNN_BAND_FC=gemm flow) then run fp16x2.
enum GemmMode { GEMM_F32 = 0, GEMM_FP16 = 1, GEMM_FP16X2 = 2, GEMM_BF16 = 3 };
int gemm_mode();
void set_force_f32(bool on);
bool force_f32();
constexpr int GEMM_BM = 128;

// ------------------------------------------------------------------ fused per-band MLP chains (mlp_chain.hip)
// One workgroup = one band x one block of frame rows, all five Linear layers of BandSplit (bsrnn.py:404-415) or of
// MaskEstimation (bsrnn.py:420-443); intermediates stay in LDS as fp16x2 pieces.
constexpr int CHAIN_LAYERS = 5;
constexpr int CHAIN_CT = 3;                   // feature tiles (32 wide) per wave and layer, at most
constexpr int CHAIN_LDS_EX = 144 * 1024;      // activation images of the workgroup's row tiles
constexpr int CHAIN_LDS_BIAS = 13 * 1024;     // the chain's biases (both together: 157 of the CU's 160 KB)
enum { CHAIN_SPLIT = 0, CHAIN_MASK = 1 };
struct ChainLayer {
    int K16;             // k-steps of 16 (input width rounded up)
    int NTL;             // feature tiles of 32 (output width rounded up); weights and biases beyond N are zero
    int bias_off;        // first bias of the layer inside the chain's bias block (floats)
    int leaky;           // LeakyReLU(0.01) after the layer
    unsigned w_off;      // byte offset of the layer's fragment streams inside ChainDesc::wstream
    int rag;             // 1: the last of the NTL feature tiles (<= 4 real features) is split over the k-steps of all waves (split_host.h)
#ifdef CHAIN_PAIR_GEOMETRY
    unsigned w_off1;     // paired geometry (ChainDesc::pair): the fragment streams of the workgroup that holds the upper half of K (w_off: the lower)
#endif
};
constexpr int CHAIN_RAG_LDS = 8 * 1024;       // LDS behind the activation images that the partial sums of such a tile need
struct ChainDesc {
    ChainLayer L[CHAIN_LAYERS];
    const void* wstream; // per layer, per wave wn: for tile t = wn + NW c, for ks, for piece: 64 lanes x 8 fp16 (split_host.h)
    const float* bias;   // the five bias vectors, each padded with zeros to NTL * 32 (constant band: the constant itself)
    int nbias;
    int NW, RT;          // groups of NW waves share the feature tiles of their RT row tiles (of 32 rows): mlp_chain.hip.
                         // RT = 3: the 48-row geometry on 16 x 16 x 32 MFMAs (K16 then counts k-steps of 32, NTL tiles of 16)
    int plane_units;     // 512-byte units of one piece of one row tile's activation image: max(2 K16, 4 NTL) over the layers
    int in_off;          // first column of the band inside an input row (SPLIT: spectrum row, MASK: b * 64 of a Z row)
    int K0;              // valid input columns, a multiple of 8 (beyond: zeros)
    int p_off;           // first column of the band in the band-padded rows (P, spectrum, output)
    int a8;              // band width in columns rounded up to 8: what is written to P / Y (pad columns exactly zero)
    int z_off;           // first column of the band inside a Z row
    int constant;        // zero-width band (TrainableConstantModule, bsrnn.py:12-24): Z[:, z_off .. +64) = bias[0 .. 64)
#ifdef CHAIN_PAIR_GEOMETRY
    // Paired geometry (mlp_chain.hip::chain_body_pair; measured and NOT adopted, DESIGN.md section 4c: built only into tools/chain_bench.hip):
    // TWO workgroups share 80 frame rows of the band; each keeps one half of K of every layer's input in LDS, multiplies it with ALL the
    // layer's output features and hands the partial sums of the partner's half of the outputs over through L2 - the band's weights are
    // streamed once per 80 rows instead of once per 48.  Then L[l].K16 = k-steps of 32 of HALF the layer's K, L[l].NTL = feature tiles of
    // 16 of the whole layer; every width is a multiple of 64.
    int pair;            // 1: this geometry; the task's x carries the half in bit 24
    int pair_base;       // first pair index of this band inside ChainLaunch::exch / pflags (pair = pair_base + row0 / 80)
#endif
};
#ifdef CHAIN_PAIR_GEOMETRY
constexpr int PAIR_ROWS = 80, PAIR_NH = 384;           // rows of a pair, features of half an output at most (a band of 768 columns)
constexpr size_t PAIR_EXCH_FLOATS = (size_t)2 * 2 * PAIR_ROWS * PAIR_NH;     // per pair: [sender][layer parity][row][feature of the receiver's half]
#endif
struct ChainLaunch {
    const ChainDesc* desc;   // device array
    const int2* tasks;       // device array [n_tasks]: (descriptor index, first frame row) of every workgroup, in dispatch order
    int n_tasks;             //   (built per M by chain_tasks(): the fill-bound wide bands interleaved with the others)
    int M;
    const float* Xin; int ldx;       // SPLIT: spectrum rows; MASK: Z rows
    float* P; int ldp;               // SPLIT: written (bandFCs_pre output = the mask's residual); MASK: read
    float* Z; int ldz;               // SPLIT: written
    const float* Xmul; int ldm;      // MASK: the spectrum the mask multiplies
    float* Y; int ldy;               // MASK: x * mask
    float* tap; int ldt;             // MASK: the mask itself (optional)
    int* range_flag;
    unsigned long long* dbg;         // measurement only (CHAIN_TRACE builds of tools/chain_bench.hip): per-wave phase stamps
    // MASK chain launched BESIDE the time-axis launch that produces its input (api.hip, overlapped dual path): every workgroup first
    // waits until the frames of its rows have left that launch (OvlConsumer below); null = the input is complete at launch
    const int* ovl_prog; int ovl_T, ovl_K, ovl_spin, ovl_base, ovl_wg_shift;
#ifdef CHAIN_PAIR_GEOMETRY
    // the exchange buffers and the flags of the pairs (flag [pair][sender][wave] = (pair_epoch * 16 + layers handed over) << 4 | XCC id)
    float* exch; int* pflags; int pair_epoch, pair_spin;
#endif
};
// rows per workgroup of a descriptor (32 RT GR; 256 for a constant band)
__host__ __device__ inline int chain_rows(const ChainDesc& d) { return d.constant ? 256 : (d.RT >= 3 ? 16 * d.RT : 32 * d.RT * (8 / d.NW)); }   // RT >= 3: row tiles of 16 (16 x 16 x 32 geometry: 48 or 80 rows)
void launch_mlp_chain(const ChainLaunch& g, int chain, hipStream_t stream);

// ------------------------------------------------------------------ overlapped dual path (api.hip: run_overlapped)
// A time-axis launch is causal and the launch behind it (the next band block, the mask chain) works frame by frame, so that launch is
// started on a second stream BESIDE the time-axis launch - on the 64 CUs its 192 workgroups leave idle and on every CU one of them
// frees - and each of its workgroups waits until the frames of its own rows have left the time-axis launch:
//   producer (time_lstm_h2w_kernel): resident[0] += 1 per workgroup at its start (the consumer launch is gated on ALL of them being
//       resident, so a waiting consumer can never keep a producer off the chip); every output row is stored write-through (sc1);
//       prog[wg] += 1 for every finished group of four steps, by the last of the four storing waves behind each one's vmcnt(0);
//   consumer: one lane polls prog[] of the time workgroups that own its rows (relaxed agent-scope loads, bounded), one agent-scope
//       acquire, workgroup barrier, then plain loads (cdna_hip_programming.md, Guideline 16).  A wait that expires is REPORTED (range
//       flag value 5: api.hip runs the call again launch after launch and stops overlapping), never computed with.
// All words are zeroed in stream order before the producers of a call start; nothing travels by value that changes from call to call.
// Progress words carry the CALL'S EPOCH in their upper bits (api.hip counts overlapped calls per context): prog[wg] = epoch << 12 | groups done,
// raised with an atomic max; the resident counters only ever grow and a gate waits for the context's running total.  So nothing is zeroed per
// call and nothing orders the consumer's stream behind the producer's except the words themselves (an overlapped call is never captured into
// a graph - api.hip - so by-value epochs cannot go stale; every 2^18 calls the host drains the device and starts the epochs again).
struct OvlProducer { int* resident; int* prog; int base; };                          // base = epoch << OVL_EPOCH_SHIFT
struct OvlConsumer { const int* prog; int T; int spin_limit; const int* order; int base; int wg_shift; };   // spin_limit: 100 MHz ticks a wait may last; wg_shift: log2 of the producer's sequences per workgroup
constexpr int OVL_EPOCH_SHIFT = 12;            // groups of four steps per launch < 4096 (frames < 16 384: api.hip checks)   // order (band launch): dispatch ordinal -> tile of 16 sequences, by the time its frames are ready
constexpr int OVL_SPIN_LIMIT = 20000000;       // 100 MHz ticks (s_memrealtime) before a wait gives up: 200 ms - a healthy wait lasts as long as a
                                               // time-axis launch (0.1 ... a few ms); under a tool that serialises kernels (rocprofv3 --pmc) the
                                               // producer never runs beside the consumer: the call falls back after this long, once per context
void launch_ovl_gate(const int* resident, int target, int* range_flag, int spin_limit, hipStream_t stream);
// sequences per workgroup of the time-axis launch over N sequences: 4 (time_lstm_h2w_kernel), or 8 (time_lstm_h2w8_kernel) where four would need more than one
// round of workgroups or BSRNN_TIME_SEQ8=1 asks for it; the launcher and whoever sizes things by that launch's workgroups (api.hip: overlap) ask here
int time_lstm_seqs(int N);
int device_cus();                                  // CUs of the current device (256 on MI355X), read once
#if defined(__HIPCC__)
// frame rows m_first .. m_last (m = batch row * T + frame) of bands k_first .. k_last: wait until every time-axis workgroup that owns one
// of those sequences (n = batch row * K + band, 1 << wg_shift per workgroup) has published the groups that cover the frames.  ONE lane calls this.
__device__ __forceinline__ bool ovl_wait_rows(const int* prog, int m_first, int m_last, int T, int K, int k_first, int k_last, int limit, int base, int wg_shift)
{
    typedef const int __attribute__((address_space(1)))* gci;
    const gci pg = (gci)prog;
    const int r_a = m_first / T, r_b = m_last / T;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = r_a; r <= r_b; ++r) {
        const int t_last = r < r_b ? T - 1 : m_last - r * T;
        const int need = base + (t_last >> 2) + 1;
        for (int wg = (r * K + k_first) >> wg_shift; wg <= (r * K + k_last) >> wg_shift; ++wg)
            while (__hip_atomic_load(pg + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)limit) return false;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // this CU's L1 holds nothing older than the poll
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... once the invalidate has completed (the caller's barrier follows)
    return true;
}
#endif

// ------------------------------------------------------------------ dual-path LSTM kernels
// Band-axis BLSTM layer (both directions in one launch): N sequences of length L.
//   xin  [N][L][IN]           IN = 64 (layer 0, fc_in folded into W_ih) or 128 (layer 1)
//   hout [N][L][128]          forward half at [0,64), backward half at [64,128)
//        In the fp16x2 mode the IN = 64 launch (layer 0) writes its output as the two fp16 planes it already has for its own
//        recurrence - per (sequence, step): [piece 0: 128 halves][piece 1: 128 halves], the same 512 bytes - and the IN = 128
//        launch (layer 1) expects exactly that as xin: the pair is always launched back to back on the same buffer (api.hip);
//        the exact-fp32 kernels (BSRNN_LSTM=f32, range-guard re-run) exchange plain fp32.
//   wpk  packed [2 dir][4 wave][(IN+64)/4 step][4 gate][64 lane], bias [2][256]
//   wpk16 (LSTM_FP16X2): the same matrix as two fp16 pieces in the f16 MFMA's B-operand order,
//         [2 dir][4 wave][(IN+64)/32 blk][4 gate][2 piece][64 lane][8]
void launch_band_lstm(const float* xin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      int N, int L, int IN, int* range_flag, hipStream_t stream);
// The block's fc in parts (BSRNN_BAND_FC): with fc16 / fcb the pair launch below writes to hb1, instead of h, the two directions' SHARES
// of fc(h): hb1[n][t][dir * 64 + f] = sum_k W_fc[f][dir * 64 + k] h_dir[n][t][k] (+ b[f] in the forward half); the time-axis launch
// adds the halves and the residual (`part` of launch_time_lstm), so the block's fc launch disappears.
bool band_fc_in_parts();
// Both layers of a band block in one launch (lstm.hip::band_pair_h2_kernel): flags = 2 ints per tile of 16 sequences, zero before the
// first launch and otherwise only touched by these launches.
bool band_pair_enabled();
void launch_band_pair(const float* z, float* hb0, float* hb1, const void* w0pk16, const float* bias0, const void* w1pk16, const float* bias1,
                      int N, int L, int* range_flag, hipStream_t stream, const void* fc16, const float* fcb, int* flags,
                      const OvlConsumer* ovl = nullptr, int* zero_words = nullptr, int zero_n = 0, hipEvent_t done = nullptr);   // zero_words: progress words to clear (overlapped dual path)
// The whole band-axis block (both layers, both directions, fc + residual) of a few sequences in one workgroup: the streaming
// step's N = C frame rows.  w0pk16 / w1pk16 / bias0 / bias1 are launch_band_lstm's arguments of the two layers; fc16 the block's
// fc (128 -> 64) as fp16x2 B fragments [4 tile][4 blk][2 piece][64 lane][8], fcb its bias.  zout = fc(h1) + b + zin.
bool band_block_is_small(int N, int L);
void launch_band_block_small(const float* zin, float* zout, const void* w0pk16, const float* bias0, const void* w1pk16, const float* bias1,
                             const void* fc16, const float* fcb, int N, int L, int* range_flag, hipStream_t stream);
// How the recurrent layers evaluate their gate products (environment BSRNN_LSTM = f32 | fp16x2, read once).
enum LstmMode { LSTM_F32 = 0, LSTM_FP16X2 = 2 };
int lstm_mode();
// Time-axis LSTM, both layers pipelined in one launch, causal with state carry.
//   zin/hout [R][T][K][64]; sequences n = r*K + k;  wpk packed [2 layer][4 wave][128 k][64 lane]
//   state_in/out [2 (h,c)][2 layer][R*K][64] or null
//   wpk16 (LSTM_FP16X2): two fp16 pieces in B-operand order, [2 layer][4 wave][4 blk][4 gate][2 piece][64 lane][8]
//   fc16 / fcb (LSTM_FP16X2, optional): the block's trailing fc (bsrnn.py:84) as two fp16 pieces in B-operand order,
//         [4 wave][2 blk][2 piece][64 lane][8], and its bias [64].  When time_lstm_fuses_fc() and both are given, the launch
//         writes the BLOCK's output fc(h1) + b + zin to hout (h1 never leaves the chip); otherwise hout = h1.
//   part (with fc16 / fcb only, optional) [R][T][K][2][64]: the block's input is zin + part[.., 0, :] + part[.., 1, :] instead of zin
//         (the shares of the preceding band block's fc, see launch_band_lstm); hout must then be a different buffer than zin.
void launch_time_lstm(const float* zin, float* hout, const float* wpk, const void* wpk16, const float* bias,
                      const float* state_in, float* state_out, int R, int T, int K, int* range_flag, hipStream_t stream,
                      const void* fc16 = nullptr, const float* fcb = nullptr, const float* part = nullptr,
                      const OvlProducer* ovl = nullptr);   // ovl: with fc16 / fcb / part only (the fused launch of the parts flow)
bool time_lstm_fuses_fc();

// ------------------------------------------------------------------ training step, part 1: recurrent layers (lstm_train.hip)
// One nn.LSTM layer (bsrnn.py:66-72) with ndir directions over N sequences of L steps, exact fp32; weights in torch layout
// (w_ih [ndir][256][IN], w_hh [ndir][256][64], bias [ndir][256] = b_ih + b_hh), zero initial state.
//   forward:  x [N][L][IN] -> h [N][L][ndir 64], gates [N][L][ndir][256] (after the non-linearities), cells [N][L][ndir][64]
//   backward: dh [N][L][ndir 64] -> dx [N][L][IN] (or null), dw_ih, dw_hh, db (= db_ih = db_hh) in the weights' layouts;
//             dg [N][L][ndir][256] and scratch [lstm_train_scratch_floats] are workspace
constexpr int LSTM_TRAIN_CHUNKS = 64;        // row chunks of the weight-gradient reductions (summed in a fixed order): ~512 rows each
constexpr int LSTM_TRAIN_MAX_CHUNKS = 256;   // ... more, up to this, when the gradient has only a few 64 x 64 tiles
// generic fp32 products of the training step (train_ops.hip): C (+)= A op(B) (+ bias) (LeakyReLU); out = A^T B with an optional
// step shift of B's rows (h_prev); column sums.  Scratch = partial sums of the row chunks.
// a group of products in one launch (the same layer of all bands); passed by value in the kernel arguments
constexpr int GEMM_GROUP = 12;
struct TrainGemmJob {
    const float* A; const float* B; float* C; const float* bias;
    int lda, ldb, ldc;
    int M, N, K;
    int tiles_x, tiles_y, rpc;      // filled in by the launchers
};
struct TrainGemmGroup { TrainGemmJob j[GEMM_GROUP]; int first[GEMM_GROUP + 1]; int count; };
struct ReduceJob { const float* part; float* out1; float* out2; int n1, n2, chunks; };
struct ReduceGroup { ReduceJob j[GEMM_GROUP]; int first[GEMM_GROUP + 1]; int count; };
void launch_sgemm_group(TrainGemmGroup& g, int trans_b, int accumulate, int leaky, hipStream_t stream);      // C_i (+)= A_i op(B_i) (+ bias_i)
size_t tn_group_scratch_floats(const TrainGemmGroup& g);
void launch_sgemm_tn_group(const TrainGemmGroup& g, float* scratch, int L, int shift, hipStream_t stream);   // C_i = A_i^T B_i, bias_i = column sums of A_i
size_t sgemm_tn_scratch_floats(int M, int N1, int N2);
size_t colsum_scratch_floats(int M, int cols);
void launch_sgemm(const float* A, int lda, const float* B, int ldb, int trans_b, float* C, int ldc, int M, int N, int K,
                  int accumulate, const float* bias, int leaky, hipStream_t stream);
void launch_sgemm_tn(const float* A, int lda, const float* B, int ldb, float* out, float* colsum, float* scratch, int M, int N1, int N2,
                     int L, int shift, hipStream_t stream);
void launch_colsum(const float* A, int lda, float* out, float* scratch, int M, int cols, hipStream_t stream);
// one AdamW update of n parameters (torch.optim.AdamW semantics; bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t))
void launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                  float bc1, float bc2s, hipStream_t stream);
// the same update for up to ADAM_GROUP tensors in one launch; everything travels by value in the kernel arguments
constexpr int ADAM_GROUP = 80;
struct AdamGroup {
    float* p[ADAM_GROUP]; const float* g[ADAM_GROUP]; float* m[ADAM_GROUP]; float* v[ADAM_GROUP];
    int n[ADAM_GROUP];                  // elements per tensor (< 2^31)
    int first_block[ADAM_GROUP + 1];    // prefix sums of ceil(n / 1024)
    int count;
};
// state != nullptr: {lr, bc1, bc2s, step(int32)} in device memory override the by-value arguments (launch_adamw_tick advances it)
void launch_adamw_group(const AdamGroup& a, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s, hipStream_t stream,
                        const float* state = nullptr);
void launch_adamw_tick(float* state, float b1, float b2, hipStream_t stream);
// nn.Linear (+ LeakyReLU(0.01)) for a group of layers that share the row count M (the same layer of all bands):
struct LinearJob {
    const float* x; int ldx;        // [M][K], row stride ldx
    const float* w; const float* b; // W [N][K] (torch layout), b [N]
    float* y; int ldy;              // forward output / backward input (read when leaky)
    const float* dy; int lddy;      // backward: gradient of y
    float* dx; int lddx;            // backward: gradient of x (null: not wanted)
    float* dw; float* db;           // backward: gradients of W and b
    int K, N;
};
void launch_linear_group_forward(const LinearJob* jobs, int n, int M, int leaky, hipStream_t stream);
size_t linear_group_scratch_floats(const LinearJob* jobs, int n, int M, int leaky);
void launch_linear_group_backward(const LinearJob* jobs, int n, int M, int leaky, float* scratch, hipStream_t stream);
// nn.Linear (+ LeakyReLU(0.01)): y = act(x W^T + b), W [N][K]; backward: dx (may be null), dW, db from dy (and y when leaky)
size_t linear_train_scratch_floats(int M, int K, int N, int leaky);
void launch_linear_train_forward(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N,
                                 int leaky, hipStream_t stream);
void launch_linear_train_backward(const float* x, int ldx, const float* w, const float* y, int ldy, const float* dy, int lddy,
                                  float* dx, int lddx, float* dw, float* db, float* scratch, int M, int K, int N, int leaky,
                                  hipStream_t stream);
size_t lstm_train_scratch_floats(int N, int L, int IN, int ndir);
void launch_lstm_train_forward(const float* x, const float* w_ih, const float* w_hh, const float* bias, float* h, float* gates,
                               float* cells, int N, int L, int IN, int ndir, hipStream_t stream);
void launch_lstm_train_backward(const float* x, const float* h, const float* gates, const float* cells, const float* dh,
                                const float* w_ih, const float* w_hh, float* dg, float* scratch, float* dx, float* dw_ih,
                                float* dw_hh, float* db, int N, int L, int IN, int ndir, hipStream_t stream);

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
// gradient of launch_istft's output w.r.t. its input (training step): dwave [R][(T-1)*1024] -> dY frame-major [R*T][ld];
// scratch [R][(T-1)*1024]
void launch_istft_backward(const FftTables& tb, const float* dwave, float* scratch, float* dY, int R, int T, hipStream_t s);
// [C][2050][T] (reference layout) <-> [C*T][ld] (band-padded)
void launch_to_frame_major(const FftTables& tb, const float* x, float* xf, int C, int T, hipStream_t s);
void launch_from_frame_major(const FftTables& tb, const float* yf, float* y, int C, int T, hipStream_t s);
// streaming DSP, one frame per row (infer-streaming.py:116-145).  What a step carries to the next one is read from one buffer and
// written to another (the stream object alternates two sets): a step can be run again from its unchanged starting point.
//   analysis: buf_out [C][2048] = [buf_in[:, 1024:], chunk [C][1024]]; X [C][ld] = rfft(buf_out * hann)
//   synthesis: s = irfft(mix(Y, X)); out = (s[0:1024] + prev_in[1024:2048]) * inv_wsum; prev_out = s
void launch_stream_analysis(const FftTables& tb, const float* buf_in, float* buf_out, const float* chunk, float* X, int C, hipStream_t s);
void launch_stream_synthesis(const FftTables& tb, const float* Y, const float* X, float mix, const float* prev_in, float* prev_out, float* out,
                             int C, hipStream_t s);

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
