// C ABI of libbsrnn_hip.so (include/bsrnn_hip.h): context, parameter intake by the reference's
// state_dict key names, host-side folding/packing, workspace, and the launch sequence of
// BSRNN.forward / forward_recurrent (bsrnn.py:385-510) and of the callers' STFT sandwich.
// There is no CPU compute path here: every entry point either enqueues HIP kernels or fails.
#include "../../include/bsrnn_hip.h"
#include "kernels.h"
#include "split_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace bsrnn;

// --------------------------------------------------------------------------- first-use accounting (bsrnn_debug_counter)
// What the library has done that does not belong on a real-time thread: device / pinned allocations, stream captures, graph
// instantiations.  Process-wide; tests read them around the LADSPA plugin's run() (tests/test_gpu_entrypoints.py).
static std::atomic<long long> g_dbg[4];
enum { DBG_ALLOC = 0, DBG_CAPTURE = 1, DBG_INSTANTIATE = 2, DBG_GRAPH_LAUNCH = 3 };

// --------------------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(BSRNN_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// --------------------------------------------------------------------------- context
namespace {

struct Param {
    std::string key;
    int64_t d0 = 0, d1 = 0;
    int ndim = 1;
    std::vector<float> data;
    bool set = false;
    int64_t numel() const { return ndim == 2 ? d0 * d1 : d0; }
};

enum Slot { PRE0, PRE2, FC0, FC2, FC4, BACK0, BACK2, BACK4, POST0, POST2, BLK_FC0, BLK_FC1, BLK_FC2, BLK_FC3, NSLOT };

enum Stage { ST_LAYOUT, ST_STFT, ST_BANDSPLIT, ST_BAND_LSTM, ST_BAND_FC, ST_TIME_LSTM, ST_TIME_FC, ST_MASK, ST_ISTFT, ST_STREAM_DSP, NSTAGE };
const char* kStageNames[NSTAGE] = {"layout", "stft", "bandsplit_mlp", "band_lstm", "band_fc", "time_lstm", "time_fc",
                                   "mask_mlp", "istft", "stream_dsp"};

struct EvRec { hipEvent_t a, b; int stage; };

}  // namespace

struct bsrnn_ctx {
    int device = 0;                 // HIP ordinal; -1 = host-only context (parameter staging / file validation, no compute)
    // Concurrency contract (include/bsrnn_hip.h): one call at a time per context.  `busy` turns an overlapping call from a
    // second host thread into BSRNN_ESTATE; a call on a different HIP stream than the previous one first waits for that
    // stream on the host (the context has ONE workspace); `gen` counts reallocations of anything a captured streaming
    // graph may point at (workspace, tap buffer, weight arena) so that the graph is re-captured instead of replayed.
    std::atomic<int> busy{0};
    int range_policy = BSRNN_RANGE_EXACT;      // what a model entry point does about the fp16x2 range guard (bsrnn_set_range_policy)
    unsigned gen = 1;
    int live_streams = 0;           // bsrnn_stream objects that point at this context
    bool zombie = false;            // bsrnn_destroy() was called while streams were alive: freed with the last stream
    bool have_last = false;
    std::vector<int> widths, off;   // bins per band, start bin
    int K = 0;
    std::vector<Param> params;
    std::map<std::string, int> index;
    bool committed = false;

    // activation column layout
    std::vector<int> aoff, poff;
    int LDA = 0, LDP = 0;

    // device-resident weights and tables
    float* d_arena = nullptr;
    GemmJob* d_jobs = nullptr;
    int2* d_tiles = nullptr;
    int job0[NSLOT], njobs[NSLOT], tile0[NSLOT], ntiles[NSLOT], tile_n[NSLOT];

    // fused per-band MLP chains (mlp_chain.hip): device descriptor arrays, grouped by class (kernels.h, ChainLaunch)
    bool stage_error = false;       // run_stage() found no task table for its row count (cannot happen: ensure_tasks runs first); reported by the entry point
    bool small_rows = false;        // the call in flight has <= GEMV_MAX_FRAME_ROWS frame rows: per-layer GEMV launches (gemv.hip)
    bool fused = false;             // false: per-layer launches (BSRNN_MLP=layers, fp32 mode, or a band too wide for the LDS image)
    ChainDesc* d_chain[2] = {nullptr, nullptr};
    std::vector<ChainDesc> h_chain[2];          // host copies (geometry and cost per band: the task tables are made from them)
    std::vector<long> chain_cost[2];
    struct TaskTable { int2* d[2]; int n[2]; };
    std::map<int, TaskTable> chain_tasks;       // per row count M: device task tables of the two chains

    const float *bandW[2][2], *bandB[2][2], *timeW[2], *timeB[2];
    const void *bandW16[2][2], *timeW16[2];
    const void* timeFc16[2] = {nullptr, nullptr};   // the time blocks' fc as fp16x2 B fragments (fused into the time-axis launch, lstm.hip)
    const float* timeFcB[2] = {nullptr, nullptr};
    const void* bandFc16[2] = {nullptr, nullptr};   // the band blocks' fc (128 -> 64) likewise, for the few-sequence kernel (band_block_small_kernel)
    const float* bandFcB[2] = {nullptr, nullptr};
    int *h_range = nullptr, *d_range = nullptr;    // range guard of the fp16x2 kernels: host-mapped word the kernels set                     // fp16x2 pieces in MFMA operand order (lstm.hip)
    float* d_tables = nullptr;
    float* d_train_ws = nullptr;       // grow-only scratch of the training entry points (stream-ordered reuse: one call at a time)
    size_t train_ws_floats = 0;
    std::vector<float*> train_ws_retired;   // outgrown scratch buffers: a captured training graph (train.GraphedTrainStep) may still point at
                                            // them, so they live until the context goes (growth is geometric: at most ~4x the final size in all)
    std::vector<void*> retired;             // likewise the outgrown workspaces / tap buffers / progress words (bsrnn_stft, _istft, _istft_backward and
                                            // every model entry point put workspace addresses into captured kernel nodes)
    int* d_colmap = nullptr;
    FftTables tb;

    // workspace (grow-only)
    size_t cap_rows = 0;
    float* d_ws = nullptr;
    float *Xf, *Yf, *A1, *A2, *P, *Z0, *Z1, *HB0, *HB1, *H1;
    int* band_flags = nullptr;      // band_pair_h2_kernel's hand-over flags (2 per tile of 16 frame rows + slack per row block), zero at allocation
    bool band_pair_off = false;     // a pair launch reported that its partner workgroups did not meet (value 4): one launch per layer from then on
    size_t tap_rows = 0;
    float* d_tap = nullptr;

    // profiling
    unsigned prof = 0;            // bitmask of stages bracketed by events
    std::vector<EvRec> pool;
    size_t pool_used = 0;
    double acc_ms[NSTAGE];
    int64_t acc_n[NSTAGE];
    hipStream_t last_stream = nullptr;

    // Overlapped dual path (run_overlapped below; kernels.h, OvlProducer / OvlConsumer): the second band block runs beside the first
    // time-axis launch and the mask chain beside the second, on the context's first auxiliary stream.
    bool overlap_env = true;        // BSRNN_OVERLAP=0: one launch after the other on the caller's stream (A/B; bit-identical results)
    int overlap_mode = 3;           // bit 0: the second band block beside the first time-axis launch; bit 1: the mask chain beside the second
                                    // (BSRNN_OVERLAP=band | mask | 1 = both); bit 2 (BSRNN_OVERLAP=pub, measurement): the publishing / waiting
                                    // launches one after the other on the caller's stream
    int overlap_sabotage = 0;       // BSRNN_OVERLAP=timeout (test hook): the producers publish nothing, the consumers give up after ~2 ms
    bool overlap_off = false;       // a consumer's wait expired once (range flag value 5): this context runs launch after launch from then on
    int* d_ovl = nullptr;           // [2 time blocks][OVL_HEAD ints: resident counter | progress word per time-axis workgroup]
    int ovl_stride = 0;             // ints per block
    struct OvlTable { int2* mask_tasks; int n_mask; int* band_order; int n_ord; };
    std::map<std::pair<int, int>, OvlTable> ovl_tables;       // per (rows C, frames T): consumer dispatch orders by readiness
    hipEvent_t ev_ovl_fork = nullptr, ev_ovl_join = nullptr;
    int ovl_epoch = 0;              // overlapped calls so far on this context (upper bits of the progress words, kernels.h); reset every 2^18
    int ovl_epoch_period = 1 << 18; // (test hook BSRNN_OVL_EPOCHS: a short period exercises the reset)
    int ovl_resident_total[2] = {0, 0};      // time-axis workgroups the two resident counters have been promised so far (the gates' targets)
    bool ovl_unjoined = false;      // the auxiliary stream may still be draining the last overlapped call (a host-side join follows where one is needed)
    // How a consumer launch is held back until every workgroup of its producer is resident: a one-wave gate kernel that spins on the
    // resident counter (default).  A resident foreign wave costs a band launch its pairing (its partners sit 8 ids apart and complete inside
    // one round of 512 slots; with 511 the last eight pairs straddle two rounds: +18 us), so the gates are kept off the band launches'
    // dispatch by events (fork in front of gate 0, a mid event between band 1 and gate 1).  BSRNN_OVL_GATE=cp (measured and rejected,
    // profiles/r04_gate_kernel_vs_cp.txt): hipStreamWaitValue32 on 8-byte signal words - on this runtime not a command-processor wait but
    // a blit kernel (__amd_rocclr_streamOpsWait) that spins so hard that the time-axis launch beside it takes 3x as long (1.37 ms per step).
    int* ovl_sig[2] = {nullptr, nullptr};
    hipEvent_t ev_ovl_mid = nullptr;

    // concurrent row blocks of one call (bsrnn_separate)
    // BSRNN_PARTS / BSRNN_PART_LAG.  0 = automatic: two row blocks on two streams once the batch's time-axis launch no longer fits one round of workgroups
    // (bsrnn_separate: from 171 rows on at K = 12; until round 4, with four sequences per workgroup only, from 128 rows on), the
    // second one stage behind the first: one block's matrix work fills the other's latency-bound time-axis LSTM (192 of 256
    // CUs, serial chain) and the ramps / tails of the fused chain launches (+8-12 % at 128 rows).  At the benchmark's 64 rows
    // two blocks of 32 gain 3 % (1.157 -> 1.119 ms per step, BSRNN_PARTS=2) but every kernel then runs beside another
    // block's kernels: the per-kernel durations (and with them the roofline figure of bench.py) stop describing the kernel,
    // so one block there.  Rows are independent; tests/test_gpu_edges.py checks 64- and 130-row calls bit for bit against
    // their row blocks.  (That check first failed: see the note at the top of fft.hip.)
    int n_parts = 0, part_lag = 1;
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
};

constexpr int MAX_PARTS = 4;
constexpr int OVL_HEAD = 16;          // ints in front of a block's progress words (the resident counter on a line of its own)

struct bsrnn_stream {
    bsrnn_ctx* ctx;
    int C;
    // What a step carries to the next one - the sliding analysis buffer, the previous synthesis frame and the LSTM state - exists
    // twice: step k reads set k & 1 and writes the other one, so a step whose operands leave the fp16 range can be run again,
    // exactly, from its untouched starting point (range policy, finish_call), without a copy per step.
    float *base = nullptr;             // the one allocation
    float *buf[2], *prev[2], *state[2];
    int cur = 0;                       // the set the NEXT step reads
    float *X, *Y, *chunk, *out;
    float *h_in = nullptr, *h_out = nullptr;   // pinned staging for the host-buffer entry point
    // One step = analysis, ~18 launches of the model, synthesis.  The model part works on fixed buffers (X -> Y, state set p -> set
    // 1 - p), so it is captured once per parity into a hipGraph and replayed (launch-bound inner loop); the two DSP kernels are
    // launched around it with the caller's own chunk / output pointers and the wet/dry control as a kernel argument (no staging copies).
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipStream_t cap = nullptr;
    bool use_graph = true;
    unsigned gen = 0;                  // context generation the graphs were captured against
};

namespace {

struct StageScope {   // brackets one stage of a call with events when profiling is on
    bsrnn_ctx* c; hipStream_t s; EvRec* r = nullptr;
    StageScope(bsrnn_ctx* c_, int stage, hipStream_t s_) : c(c_), s(s_)
    {
        if (((c->prof >> stage) & 1u) && c->pool_used < c->pool.size()) {
            r = &c->pool[c->pool_used++];
            r->stage = stage;
            (void)hipEventRecord(r->a, s);
        }
    }
    ~StageScope() { if (r) (void)hipEventRecord(r->b, s); }
};

int add_param(bsrnn_ctx* c, const std::string& key, int64_t d0, int64_t d1, int ndim)
{
    Param p;
    p.key = key; p.d0 = d0; p.d1 = d1; p.ndim = ndim;
    c->index[key] = (int)c->params.size();
    c->params.push_back(p);
    return 0;
}
void add_linear(bsrnn_ctx* c, const std::string& prefix, int n_out, int n_in)
{
    add_param(c, prefix + ".weight", n_out, n_in, 2);
    add_param(c, prefix + ".bias", n_out, 0, 1);
}
int imax(int a, int b) { return a > b ? a : b; }
int round8(int a) { return (a + 7) & ~7; }

// parameter inventory in the reference's state_dict order (bsrnn.py:329-376)
void build_inventory(bsrnn_ctx* c)
{
    char b[128];
    const int H = HID;
    for (int i = 0; i < c->K; ++i) {
        const int a = 2 * c->widths[i];
        if (a > 0) {
            snprintf(b, sizeof b, "bandFCs_pre.%d.0", i); add_linear(c, b, a, a);
            snprintf(b, sizeof b, "bandFCs_pre.%d.2", i); add_linear(c, b, a, a);
        } else { snprintf(b, sizeof b, "bandFCs_pre.%d.0.trainable_constant", i); add_param(c, b, 0, 0, 1); }
    }
    for (int i = 0; i < c->K; ++i) {
        const int a = 2 * c->widths[i], m = imax(a, H);
        if (a > 0) {
            snprintf(b, sizeof b, "bandFCs.%d.0", i); add_linear(c, b, m, a);
            snprintf(b, sizeof b, "bandFCs.%d.2", i); add_linear(c, b, H, m);
            snprintf(b, sizeof b, "bandFCs.%d.4", i); add_linear(c, b, H, H);
        } else { snprintf(b, sizeof b, "bandFCs.%d.0.trainable_constant", i); add_param(c, b, H, 0, 1); }
    }
    for (int j = 0; j < 4; ++j) {
        const bool bidir = (j % 2 == 0);
        snprintf(b, sizeof b, "lstms.%d.m.fc_in", j); add_linear(c, b, H, H);
        for (int layer = 0; layer < 2; ++layer) {
            const int n_in = layer == 0 ? H : (bidir ? 2 * H : H);
            for (int d = 0; d < (bidir ? 2 : 1); ++d) {
                const char* sfx = d ? "_reverse" : "";
                snprintf(b, sizeof b, "lstms.%d.m.rnn.weight_ih_l%d%s", j, layer, sfx); add_param(c, b, 4 * H, n_in, 2);
                snprintf(b, sizeof b, "lstms.%d.m.rnn.weight_hh_l%d%s", j, layer, sfx); add_param(c, b, 4 * H, H, 2);
                snprintf(b, sizeof b, "lstms.%d.m.rnn.bias_ih_l%d%s", j, layer, sfx); add_param(c, b, 4 * H, 0, 1);
                snprintf(b, sizeof b, "lstms.%d.m.rnn.bias_hh_l%d%s", j, layer, sfx); add_param(c, b, 4 * H, 0, 1);
            }
        }
        snprintf(b, sizeof b, "lstms.%d.m.fc", j); add_linear(c, b, H, bidir ? 2 * H : H);
    }
    for (int i = 0; i < c->K; ++i) {
        const int a = 2 * c->widths[i], pz = imax(a, 2 * H);
        if (a > 0) {
            snprintf(b, sizeof b, "bandFCs_back.%d.0", i); add_linear(c, b, 2 * H, H);
            snprintf(b, sizeof b, "bandFCs_back.%d.2", i); add_linear(c, b, pz, 2 * H);
            snprintf(b, sizeof b, "bandFCs_back.%d.4", i); add_linear(c, b, a, pz);
        } else { snprintf(b, sizeof b, "bandFCs_back.%d.0.trainable_constant", i); add_param(c, b, 0, 0, 1); }
    }
    for (int i = 0; i < c->K; ++i) {
        const int a = 2 * c->widths[i];
        if (a > 0) {
            snprintf(b, sizeof b, "bandFCs_back_post.%d.0", i); add_linear(c, b, a, a);
            snprintf(b, sizeof b, "bandFCs_back_post.%d.2", i); add_linear(c, b, a, a);
        } else { snprintf(b, sizeof b, "bandFCs_back_post.%d.0.trainable_constant", i); add_param(c, b, 0, 0, 1); }
    }
}

const Param& P_(const bsrnn_ctx* c, const std::string& key) { return c->params[c->index.at(key)]; }

// host-side arena builder (16-byte aligned segments)
struct Arena {
    std::vector<float> h;
    size_t put(const float* p, size_t n)
    {
        size_t o = (h.size() + 3) & ~size_t(3);
        h.resize(o + n);
        if (n) memcpy(&h[o], p, n * sizeof(float));
        return o;
    }
    size_t put(const std::vector<float>& v) { return put(v.data(), v.size()); }
};

// [W_ih (fc_in folded for layer 0) | W_hh] and the summed bias of one LSTM layer/direction
void lstm_cat(const bsrnn_ctx* c, int j, int layer, const char* sfx, int n_in, std::vector<double>& wcat, std::vector<double>& bsum)
{
    char b[128];
    const int H = HID, KT = n_in + H;
    snprintf(b, sizeof b, "lstms.%d.m.rnn.weight_ih_l%d%s", j, layer, sfx); const Param& wih = P_(c, b);
    snprintf(b, sizeof b, "lstms.%d.m.rnn.weight_hh_l%d%s", j, layer, sfx); const Param& whh = P_(c, b);
    snprintf(b, sizeof b, "lstms.%d.m.rnn.bias_ih_l%d%s", j, layer, sfx); const Param& bih = P_(c, b);
    snprintf(b, sizeof b, "lstms.%d.m.rnn.bias_hh_l%d%s", j, layer, sfx); const Param& bhh = P_(c, b);
    wcat.assign((size_t)4 * H * KT, 0.0);
    bsum.assign(4 * H, 0.0);
    for (int r = 0; r < 4 * H; ++r) bsum[r] = (double)bih.data[r] + (double)bhh.data[r];
    if (layer == 0) {
        // fc_in folded: W' = W_ih W_in, b' += W_ih b_in   (bsrnn.py:82-83: rnn(fc_in(x)), no activation between)
        snprintf(b, sizeof b, "lstms.%d.m.fc_in.weight", j); const Param& win = P_(c, b);
        snprintf(b, sizeof b, "lstms.%d.m.fc_in.bias", j); const Param& bin = P_(c, b);
        for (int r = 0; r < 4 * H; ++r) {
            for (int k = 0; k < H; ++k) {
                double s = 0;
                for (int u = 0; u < H; ++u) s += (double)wih.data[(size_t)r * H + u] * (double)win.data[(size_t)u * H + k];
                wcat[(size_t)r * KT + k] = s;
            }
            double sb = 0;
            for (int u = 0; u < H; ++u) sb += (double)wih.data[(size_t)r * H + u] * (double)bin.data[u];
            bsum[r] += sb;
        }
    } else {
        for (int r = 0; r < 4 * H; ++r)
            for (int k = 0; k < n_in; ++k) wcat[(size_t)r * KT + k] = wih.data[(size_t)r * n_in + k];
    }
    for (int r = 0; r < 4 * H; ++r)
        for (int k = 0; k < H; ++k) wcat[(size_t)r * KT + n_in + k] = whh.data[(size_t)r * H + k];
}

int ensure_ws(bsrnn_ctx* c, size_t rows)
{
    if (rows <= c->cap_rows) return 0;
    // Grow-only, geometrically, and the outgrown buffer is RETIRED, not freed: a hipGraph captured by the caller (train.GraphedTrainStep keeps
    // one per clip length; a user's torch.cuda.graph around forward) holds workspace addresses in its kernel nodes, and work of the previous
    // call may still be running in it.  It stays valid scratch for whoever knows it; everything new uses the new one (streaming graphs of
    // this library re-capture: generation counter).  At most ~3x the final size in all, released with the context.
    if (c->cap_rows) rows = std::max(rows, c->cap_rows + c->cap_rows / 2);
    const size_t KH = (size_t)c->K * HID;
    auto seg = [](size_t n) { return (n + 63) & ~size_t(63); };
    const size_t sizes[11] = {seg(rows * c->LDP), seg(rows * c->LDP), seg(rows * c->LDA), seg(rows * c->LDA), seg(rows * c->LDP),
                              seg(rows * KH), seg(rows * KH), seg(rows * KH * 2), seg(rows * KH * 2), seg(rows * KH),
                              seg(rows / 8 + 2 * MAX_PARTS + 64)};
    size_t total = 0;
    for (size_t s : sizes) total += s;
    float* fresh = nullptr;
    ++g_dbg[DBG_ALLOC];
    HIP_TRY(hipMalloc((void**)&fresh, total * sizeof(float)));
    // pad columns (band segments are 16-byte aligned, rows padded) are read as K-padding by the GEMM and are
    // never written afterwards: they must be finite, so the whole workspace starts at zero
    HIP_TRY(hipMemset(fresh, 0, total * sizeof(float)));
    if (c->d_ws) c->retired.push_back(c->d_ws);
    c->d_ws = fresh;
    float* p = c->d_ws;
    float** dst[10] = {&c->Xf, &c->Yf, &c->A1, &c->A2, &c->P, &c->Z0, &c->Z1, &c->HB0, &c->HB1, &c->H1};
    for (int i = 0; i < 10; ++i) { *dst[i] = p; p += sizes[i]; }
    c->band_flags = reinterpret_cast<int*>(p);
    c->cap_rows = rows;
    ++c->gen;
    return 0;
}
int ensure_tap(bsrnn_ctx* c, size_t rows)
{
    if (rows <= c->tap_rows) return 0;
    if (c->tap_rows) rows = std::max(rows, c->tap_rows + c->tap_rows / 2);
    float* fresh = nullptr;
    ++g_dbg[DBG_ALLOC];
    HIP_TRY(hipMalloc((void**)&fresh, rows * c->LDP * sizeof(float)));
    if (c->d_tap) c->retired.push_back(c->d_tap);          // (retired like the workspace, see ensure_ws)
    c->d_tap = fresh;
    c->tap_rows = rows;
    ++c->gen;
    return 0;
}

int ensure_streams(bsrnn_ctx* c, int parts)
{
    if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int j = 0; j < parts; ++j) {
        if (!c->aux[j]) { ++g_dbg[DBG_ALLOC]; HIP_TRY(hipStreamCreateWithFlags(&c->aux[j], hipStreamNonBlocking)); }
        if (!c->ev_join[j]) HIP_TRY(hipEventCreateWithFlags(&c->ev_join[j], hipEventDisableTiming));
    }
    return 0;
}

// Task table of a fused chain launch for M frame rows: one entry (descriptor, first row) per workgroup, in dispatch order:
// longest workgroups first (the descriptors are sorted by class and cost at commit), all row blocks of a band together
// (they share its weight stream through L2).  Measured and dropped: interleaving the fill-bound 768-wide band with the
// others (its workgroups take 97 us on half the CUs against 129 us on all of them, tools/chain_bench.hip) - the late starts
// of the long workgroups cost more than the contention saves (310 vs 250 us per chain).
void build_chain_tasks(const bsrnn_ctx* c, int ch, int M, std::vector<int2>& out)
{
    out.clear();
    const auto& ds = c->h_chain[ch];
    for (size_t di = 0; di < ds.size(); ++di)
        for (int r0 = 0; r0 < M; r0 += chain_rows(ds[di])) out.push_back(make_int2((int)di, r0));
}

// Task tables for the row counts of ONE call (several when the call runs as concurrent row blocks).  The cache is bounded; when it
// is full it is dropped once, before any of this call's tables is made, so a call never evicts a table it is about to use
// (captured streaming graphs notice through ctx->gen and re-capture).
int ensure_tasks(bsrnn_ctx* c, const int* Ms, int n)
{
    if (!c->fused) return 0;
    int missing = 0;
    for (int i = 0; i < n; ++i) missing += c->chain_tasks.find(Ms[i]) == c->chain_tasks.end();
    if (!missing) return 0;
    if (c->chain_tasks.size() + missing > 64) {
        HIP_TRY(hipDeviceSynchronize());
        for (auto& kv : c->chain_tasks)
            for (int ch = 0; ch < 2; ++ch) (void)hipFree(kv.second.d[ch]);
        c->chain_tasks.clear();
        ++c->gen;
    }
    for (int i = 0; i < n; ++i) {
        if (c->chain_tasks.find(Ms[i]) != c->chain_tasks.end()) continue;
        bsrnn_ctx::TaskTable t;
        t.d[0] = t.d[1] = nullptr;
        std::vector<int2> h;
        for (int ch = 0; ch < 2; ++ch) {
            build_chain_tasks(c, ch, Ms[i], h);
            t.n[ch] = (int)h.size();
            ++g_dbg[DBG_ALLOC];
            hipError_t e = hipMalloc((void**)&t.d[ch], (h.size() + 1) * sizeof(int2));
            if (e == hipSuccess) e = hipMemcpy(t.d[ch], h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                for (int k = 0; k <= ch; ++k) if (t.d[k]) (void)hipFree(t.d[k]);
                return fail(BSRNN_EHIP, "task table: %s", hipGetErrorString(e));
            }
        }
        c->chain_tasks.emplace(Ms[i], t);
    }
    return 0;
}
int ensure_tasks(bsrnn_ctx* c, int M) { return ensure_tasks(c, &M, 1); }

void gemm_slot(bsrnn_ctx* c, int slot, const float* X, int ldx, float* Y, int ldy, const float* R, int ldr,
               const float* Mul, int ldm, float* tap, int M, int epi, hipStream_t s)
{
    GemmLaunch g;
    memset(&g, 0, sizeof g);
    g.range_flag = c->d_range;
    g.jobs = c->d_jobs + c->job0[slot];
    g.tiles = c->d_tiles + c->tile0[slot];
    g.n_tiles = c->ntiles[slot];
    g.tile_n = c->tile_n[slot];
    g.X = X; g.ldx = ldx; g.Y = Y; g.ldy = ldy; g.R = R; g.ldr = ldr; g.Mul = Mul; g.ldm = ldm;
    g.tap = tap; g.ldt = c->LDP; g.M = M; g.epilogue = epi;
    // a call of a few frame rows: its per-band layers (M = C rows) run as exact-fp32 GEMV launches instead of 128-row MFMA
    // tiles; the block fc layers (M K rows against a 64 x 128 matrix) stay on the MFMA kernel (measured: 5.7 vs 28 us)
    if (c->small_rows && M <= 2 * GEMV_MAX_FRAME_ROWS) launch_gemv(g, s);
    else launch_gemm(g, s);
}

// A contiguous block of rows (utterance-channels) of one call, with its slice of the workspace
// and the stream it runs on.  Rows are independent (bsrnn.py:394-395), so a call can be cut into
// several such parts that run concurrently on separate HIP streams.
struct Part {
    int C, T;                       // rows and frames of this part
    hipStream_t s;
    const float* Xf; float* Yf; float* tap;              // [C*T][LDP], band-padded spectrum layout
    float *A1, *A2, *P, *Z0, *Z1, *HB0, *HB1, *H1;
    int* band_flags;                                     // this part's hand-over flags of the band-pair launch
    const float* state_in; float* state_out;             // [4][2][C_total*K][64] slabs already offset to this part's first row
    size_t state_slab;                                   // floats between the two Time blocks' slabs (uses C_total)
    const float* wave; float* wave_out; int64_t n;       // only for the fused sandwich
    const bsrnn_ctx::OvlTable* ovl;                      // non-null: the overlapped flow (run_overlapped) - producers publish, consumers wait
    int ovl_mode;                                        // ... which of the two hand-overs (bsrnn_ctx::overlap_mode)
    int ovl_base;                                        // this call's epoch << OVL_EPOCH_SHIFT
    hipEvent_t band_done[2];                             // events the two band launches signal themselves when they complete (or null)
};

Part make_part(bsrnn_ctx* c, int row0, int C, int T, hipStream_t s, int j = 0)
{
    Part p;
    memset(&p, 0, sizeof p);
    const size_t m0 = (size_t)row0 * T, KH = (size_t)c->K * HID;
    p.C = C; p.T = T; p.s = s;
    p.Xf = c->Xf + m0 * c->LDP; p.Yf = c->Yf + m0 * c->LDP;
    p.A1 = c->A1 + m0 * c->LDA; p.A2 = c->A2 + m0 * c->LDA; p.P = c->P + m0 * c->LDP;
    p.Z0 = c->Z0 + m0 * KH; p.Z1 = c->Z1 + m0 * KH; p.H1 = c->H1 + m0 * KH;
    p.HB0 = c->HB0 + m0 * KH * 2; p.HB1 = c->HB1 + m0 * KH * 2;
    // Row block j starts j pairs behind its first tile's natural place: block j - 1 ends at most at floor(m0 / 16) + 1 + (j - 1), so the
    // pair ranges of concurrent blocks are disjoint for any row split (odd R, 3 or 4 blocks included; checked in bsrnn_separate)
    p.band_flags = c->band_flags + 2 * (m0 / 16 + j);
    return p;
}

// the band blocks' fc in parts needs the pair launch (lstm.hip); a context whose pair launch once reported that its partner workgroups
// did not meet runs the round-2 flow from then on
static bool ctx_parts(const bsrnn_ctx* c) { return band_fc_in_parts() && !c->band_pair_off; }
static bool ctx_pair(const bsrnn_ctx* c) { return band_pair_enabled() && !c->band_pair_off; }

enum { MS_STFT, MS_BANDSPLIT, MS_BAND0, MS_BANDFC0, MS_TIME0, MS_TIMEFC0, MS_BAND1, MS_BANDFC1, MS_TIME1, MS_TIMEFC1, MS_MASK, MS_ISTFT, MS_COUNT };

// One stage of the model for one part.  Xf [M][2050] -> Yf [M][2050], M = C*T, row = c*T + t.
void run_stage(bsrnn_ctx* c, const Part& p, int stage)
{
    const int M = p.C * p.T, K = c->K, KH = K * HID;
    hipStream_t s = p.s;
    switch (stage) {
    case MS_STFT:
        if (p.wave) { StageScope sc(c, ST_STFT, s); launch_stft(c->tb, p.wave, const_cast<float*>(p.Xf), p.C, p.n, p.T, s); }
        break;
    case MS_BANDSPLIT: {   // bandFCs_pre (2 linears) -> residual P; bandFCs (3 linears) -> Z0   bsrnn.py:404-415
        StageScope sc(c, ST_BANDSPLIT, s);
        if (c->fused && !force_f32() && !c->small_rows) {      // all five layers of every band in one launch, intermediates in LDS
            ChainLaunch g;
            memset(&g, 0, sizeof g);
            auto tti = c->chain_tasks.find(M);                           // made by ensure_tasks() before any launch (and outside graph capture)
            if (tti == c->chain_tasks.end()) { c->stage_error = true; break; }
            const bsrnn_ctx::TaskTable& tt = tti->second;
            g.desc = c->d_chain[CHAIN_SPLIT]; g.tasks = tt.d[CHAIN_SPLIT]; g.n_tasks = tt.n[CHAIN_SPLIT];
            g.M = M; g.Xin = p.Xf; g.ldx = c->LDP; g.P = p.P; g.ldp = c->LDP; g.Z = p.Z0; g.ldz = KH; g.range_flag = c->d_range;
            launch_mlp_chain(g, CHAIN_SPLIT, s);
            break;
        }
        gemm_slot(c, PRE0, p.Xf, c->LDP, p.A1, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, PRE2, p.A1, c->LDA, p.P, c->LDP, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, FC0, p.P, c->LDP, p.A1, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, FC2, p.A1, c->LDA, p.A2, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, FC4, p.A2, c->LDA, p.Z0, KH, nullptr, 0, nullptr, 0, nullptr, M, EPI_LINEAR, s);
        break;
    }
    case MS_BAND0: case MS_BAND1: {   // BandwiseLSTM: N = M sequences of length K   bsrnn.py:138-153
        const int blk = stage == MS_BAND1;
        StageScope sc(c, ST_BAND_LSTM, s);
        if (band_block_is_small(M, K)) {          // a few frame rows (streaming): the whole block, fc + residual included, in one launch
            launch_band_block_small(p.Z0, p.Z1, c->bandW16[blk][0], c->bandB[blk][0], c->bandW16[blk][1], c->bandB[blk][1],
                                    c->bandFc16[blk], c->bandFcB[blk], M, K, c->d_range, s);
            break;
        }
        // parts flow (band_fc_in_parts()): block 0 reads Z0 and its time block writes Z1, block 1 reads Z1 and its time block writes
        // Z0 - the block's fc + residual are formed inside the two launches around them (kernels.h), MS_BANDFC does not exist
        const bool parts = ctx_parts(c);
        const float* zi = parts && blk ? p.Z1 : p.Z0;
        if (ctx_pair(c)) {                        // both layers in one launch (A/B: BSRNN_BAND_PAIR=0)
            OvlConsumer oc = {nullptr, 0, 0, nullptr, 0, 2};
            const bool cons = p.ovl && blk && (p.ovl_mode & 1);
            if (cons)                             // beside the first time-axis launch: tiles in the order their frames leave it
                oc = OvlConsumer{c->d_ovl + OVL_HEAD, p.T, c->overlap_sabotage ? 200000 : OVL_SPIN_LIMIT, p.ovl->band_order, p.ovl_base, time_lstm_seqs(p.C * K) == 8 ? 3 : 2};
            launch_band_pair(zi, p.HB0, p.HB1, c->bandW16[blk][0], c->bandB[blk][0], c->bandW16[blk][1], c->bandB[blk][1], M, K, c->d_range, s,
                             parts ? c->bandFc16[blk] : nullptr, parts ? c->bandFcB[blk] : nullptr, p.band_flags, cons ? &oc : nullptr, nullptr, 0,
                             p.ovl ? p.band_done[blk] : nullptr);
            break;
        }
        launch_band_lstm(zi, p.HB0, c->bandW[blk][0], c->bandW16[blk][0], c->bandB[blk][0], M, K, 64, c->d_range, s);
        launch_band_lstm(p.HB0, p.HB1, c->bandW[blk][1], c->bandW16[blk][1], c->bandB[blk][1], M, K, 128, c->d_range, s);
        break;
    }
    case MS_BANDFC0: case MS_BANDFC1: {
        const int blk = stage == MS_BANDFC1;
        if (band_block_is_small(M, K) || ctx_parts(c)) break;     // done inside the band launch / inside the launches around it
        StageScope sc(c, ST_BAND_FC, s);
        gemm_slot(c, BLK_FC0 + 2 * blk, p.HB1, 2 * HID, p.Z1, HID, p.Z0, HID, nullptr, 0, nullptr, M * K, EPI_RES, s);
        break;
    }
    case MS_TIME0: case MS_TIME1: {   // TimewiseLSTM: N = C*K sequences of length T, causal, state carry   bsrnn.py:106-128
        const int blk = stage == MS_TIME1;
        StageScope sc(c, ST_TIME_LSTM, s);
        // fp16x2 mode: the launch also computes the block's fc + residual (out = fc(h1) + Z1 -> Z0); otherwise it writes h1
        const bool fused = time_lstm_fuses_fc();
        if (ctx_parts(c) && !band_block_is_small(M, K)) {
            OvlProducer op = {nullptr, nullptr, 0};
            const bool prod = p.ovl && (p.ovl_mode & (blk ? 2 : 1));
            if (prod) op = OvlProducer{c->ovl_sig[blk] ? c->ovl_sig[blk] : c->d_ovl + blk * c->ovl_stride, c->overlap_sabotage ? nullptr : c->d_ovl + blk * c->ovl_stride + OVL_HEAD, p.ovl_base};
            launch_time_lstm(blk ? p.Z1 : p.Z0, blk ? p.Z0 : p.Z1, c->timeW[blk], c->timeW16[blk], c->timeB[blk],
                             p.state_in ? p.state_in + blk * p.state_slab : nullptr,
                             p.state_out ? p.state_out + blk * p.state_slab : nullptr, p.C, p.T, K, c->d_range, s,
                             c->timeFc16[blk], c->timeFcB[blk], p.HB1, prod ? &op : nullptr);
            break;
        }
        launch_time_lstm(p.Z1, fused ? p.Z0 : p.H1, c->timeW[blk], c->timeW16[blk], c->timeB[blk],
                         p.state_in ? p.state_in + blk * p.state_slab : nullptr,
                         p.state_out ? p.state_out + blk * p.state_slab : nullptr, p.C, p.T, K, c->d_range, s,
                         c->timeFc16[blk], c->timeFcB[blk]);
        break;
    }
    case MS_TIMEFC0: case MS_TIMEFC1: {
        const int blk = stage == MS_TIMEFC1;
        if (time_lstm_fuses_fc()) break;          // done inside the time-axis launch
        StageScope sc(c, ST_TIME_FC, s);
        gemm_slot(c, BLK_FC1 + 2 * blk, p.H1, HID, p.Z0, HID, p.Z1, HID, nullptr, 0, nullptr, M * K, EPI_RES, s);
        break;
    }
    case MS_MASK: {   // bandFCs_back (3) + bandFCs_back_post (2) + skip + x*mask   bsrnn.py:420-443
        StageScope sc(c, ST_MASK, s);
        if (c->fused && !force_f32() && !c->small_rows) {
            ChainLaunch g;
            memset(&g, 0, sizeof g);
            auto tti = c->chain_tasks.find(M);
            if (tti == c->chain_tasks.end()) { c->stage_error = true; break; }
            const bsrnn_ctx::TaskTable& tt = tti->second;
            g.desc = c->d_chain[CHAIN_MASK]; g.tasks = tt.d[CHAIN_MASK]; g.n_tasks = tt.n[CHAIN_MASK];
            g.M = M; g.Xin = p.Z0; g.ldx = KH; g.P = p.P; g.ldp = c->LDP; g.Xmul = p.Xf; g.ldm = c->LDP;
            g.Y = p.Yf; g.ldy = c->LDP; g.tap = p.tap; g.ldt = c->LDP; g.range_flag = c->d_range;
            if (p.ovl && (p.ovl_mode & 2)) {      // beside the second time-axis launch: the earliest-ready heavy workgroups first
                g.tasks = p.ovl->mask_tasks; g.n_tasks = p.ovl->n_mask;
                g.ovl_prog = c->d_ovl + c->ovl_stride + OVL_HEAD; g.ovl_T = p.T; g.ovl_K = K;
                g.ovl_spin = c->overlap_sabotage ? 200000 : OVL_SPIN_LIMIT;
                g.ovl_base = p.ovl_base; g.ovl_wg_shift = time_lstm_seqs(p.C * K) == 8 ? 3 : 2;
            }
            launch_mlp_chain(g, CHAIN_MASK, s);
            break;
        }
        gemm_slot(c, BACK0, p.Z0, KH, p.A1, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, BACK2, p.A1, c->LDA, p.A2, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, BACK4, p.A2, c->LDA, p.A1, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, POST0, p.A1, c->LDA, p.A2, c->LDA, nullptr, 0, nullptr, 0, nullptr, M, EPI_LEAKY, s);
        gemm_slot(c, POST2, p.A2, c->LDA, p.Yf, c->LDP, p.P, c->LDP, p.Xf, c->LDP, p.tap, M, EPI_MASK, s);
        break;
    }
    case MS_ISTFT:
        if (p.wave_out) {
            StageScope sc(c, ST_ISTFT, s);
            launch_istft(c->tb, p.Yf, p.wave_out, p.C, p.T, s);
        }
        break;
    }
}

// ---- overlapped dual path ------------------------------------------------------------------------------------------------------
// Whether a call of C rows x T frames runs overlapped: the fp16x2 parts flow with fused chains (the launches that know how to publish /
// wait), a time-axis launch that leaves CUs free (<= 224 of 256 workgroups) and enough frames for a head start to exist.
static bool overlap_wanted(const bsrnn_ctx* c, int C, int T)
{
    if (!c->overlap_env || c->overlap_off || !ctx_parts(c) || !c->fused || force_f32() || gemm_mode() == GEMM_F32) return false;
    const int S = time_lstm_seqs(C * c->K), nwg = (C * c->K + S - 1) / S;      // workgroups of the time-axis launch
    return !band_block_is_small(C * T, c->K) && nwg >= 32 && nwg <= 224 && T >= 32 && T < (4 << OVL_EPOCH_SHIFT) - 8 && C * T > GEMV_MAX_FRAME_ROWS;
}
static void free_ovl_tables(bsrnn_ctx* c)
{
    for (auto& kv : c->ovl_tables) { (void)hipFree(kv.second.mask_tasks); (void)hipFree(kv.second.band_order); }
    c->ovl_tables.clear();
}
// Progress words and the consumers' dispatch orders for a call of C rows x T frames (made outside any capture, like the task tables).
int ensure_ovl(bsrnn_ctx* c, int C, int T)
{
    if (!overlap_wanted(c, C, T)) return 0;
    int rc = ensure_streams(c, 1);
    if (rc) return rc;
    if (!c->ev_ovl_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_ovl_fork, hipEventDisableTiming | hipEventDisableSystemFence));      // (device-side ordering only: no system-scope release on the record)
    if (!c->ev_ovl_join) HIP_TRY(hipEventCreateWithFlags(&c->ev_ovl_join, hipEventDisableTiming | hipEventDisableSystemFence));
    if (!c->ev_ovl_mid) HIP_TRY(hipEventCreateWithFlags(&c->ev_ovl_mid, hipEventDisableTiming | hipEventDisableSystemFence));
    static const bool want_cp = [] { const char* e = getenv("BSRNN_OVL_GATE"); return e && !strcmp(e, "cp"); }();
    if (want_cp && !c->ovl_sig[0]) {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess) { can = 0; (void)hipGetLastError(); }
        for (int b = 0; b < 2 && can; ++b) {
            ++g_dbg[DBG_ALLOC];
            if (hipExtMallocWithFlags((void**)&c->ovl_sig[b], 8, hipMallocSignalMemory) != hipSuccess || hipMemset(c->ovl_sig[b], 0, 8) != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k <= b; ++k) if (c->ovl_sig[k]) { (void)hipFree(c->ovl_sig[k]); c->ovl_sig[k] = nullptr; }
                can = 0;                          // (the kernel gates take over)
            }
        }
        if (c->ovl_sig[0]) c->ovl_resident_total[0] = c->ovl_resident_total[1] = 0;
    }
    const int M = C * T, K = c->K, S = time_lstm_seqs(C * K), nwg = (C * K + S - 1) / S;
    const int stride = OVL_HEAD + ((nwg + 15) & ~15);
    if (stride > c->ovl_stride) {
        if (c->d_ovl) { c->retired.push_back(c->d_ovl); c->d_ovl = nullptr; }      // (retired like the workspace, see ensure_ws)
        ++g_dbg[DBG_ALLOC];
        HIP_TRY(hipMalloc((void**)&c->d_ovl, (size_t)2 * stride * sizeof(int)));
        HIP_TRY(hipMemset(c->d_ovl, 0, (size_t)2 * stride * sizeof(int)));
        c->ovl_stride = stride;
        if (!c->ovl_sig[0]) c->ovl_resident_total[0] = c->ovl_resident_total[1] = 0;       // (fresh counters; the epochs go on: fresh words are below every base)
        ++c->gen;
    }
    const auto key = std::make_pair(C, T);
    if (c->ovl_tables.find(key) != c->ovl_tables.end()) return 0;
    if (c->ovl_tables.size() >= 16) { HIP_TRY(hipDeviceSynchronize()); free_ovl_tables(c); ++c->gen; }
    // band block: tiles of 16 frame rows m = row * T + frame, sorted by the last frame the tile needs (a tile that straddles two batch
    // rows needs the first one's last frame); padded with -1 to the launch's whole groups of eight tiles
    const int tiles = (M + 15) / 16, n_ord = ((tiles + 7) / 8) * 8;
    std::vector<int> order(tiles), ready(tiles);
    for (int t = 0; t < tiles; ++t) {
        const int m0 = 16 * t, m1 = std::min(M - 1, m0 + 15);
        order[t] = t;
        ready[t] = m0 / T != m1 / T ? T - 1 : m1 % T;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ready[a] < ready[b]; });
    order.resize(n_ord, -1);
    // mask chain: its task table (longest workgroups first) with the workgroups that can START while the time-axis launch still runs
    // in front: as many as that launch leaves CUs free (each runs longer than the rest of it), the earliest-ready of the heavy classes
    // (<= 80 rows per workgroup: the widest bands, which also end the launch when they start late)
    std::vector<int2> tasks;
    build_chain_tasks(c, CHAIN_MASK, M, tasks);
    const auto& ds = c->h_chain[CHAIN_MASK];
    std::vector<int> cand;
    auto task_ready = [&](const int2& t) {
        const int m1 = std::min(M - 1, t.y + chain_rows(ds[t.x]) - 1);
        return t.y / T != m1 / T ? T - 1 : m1 % T;
    };
    for (int i = 0; i < (int)tasks.size(); ++i)
        if (!ds[tasks[i].x].constant && chain_rows(ds[tasks[i].x]) <= 80 && task_ready(tasks[i]) < T - 1) cand.push_back(i);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return task_ready(tasks[a]) < task_ready(tasks[b]); });
    const int n_early = std::min((int)cand.size(), std::max(0, 256 - nwg));
    std::vector<char> early(tasks.size(), 0);
    std::vector<int2> mt;
    for (int i = 0; i < n_early; ++i) { mt.push_back(tasks[cand[i]]); early[cand[i]] = 1; }
    for (int i = 0; i < (int)tasks.size(); ++i)
        if (!early[i]) mt.push_back(tasks[i]);
    bsrnn_ctx::OvlTable tb;
    tb.mask_tasks = nullptr; tb.band_order = nullptr; tb.n_mask = (int)mt.size(); tb.n_ord = n_ord;
    hipError_t e = hipMalloc((void**)&tb.mask_tasks, (mt.size() + 1) * sizeof(int2));
    if (e == hipSuccess) e = hipMemcpy(tb.mask_tasks, mt.data(), mt.size() * sizeof(int2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&tb.band_order, (size_t)n_ord * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(tb.band_order, order.data(), (size_t)n_ord * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (tb.mask_tasks) (void)hipFree(tb.mask_tasks);
        if (tb.band_order) (void)hipFree(tb.band_order);
        return fail(BSRNN_EHIP, "overlap tables: %s", hipGetErrorString(e));
    }
    c->ovl_tables.emplace(key, tb);
    return 0;
}

// The stages of one part with the dual path overlapped (kernels.h): caller's stream A, the context's auxiliary stream B.
//   A: [STFT] BandSplit, zero the progress words, band block 0, TIME 0 ............ gate 1, MASK chain (waits per workgroup), [iSTFT]
//   B:                                        (fork) gate 0, BAND block 1 (waits per tile), TIME 1 ......................... (join)
// gate b = one wave that leaves when every workgroup of time launch b is resident: no consumer workgroup is dispatched before that, so a
// waiting consumer never holds a CU a producer needs; producers wait for nobody.  Same kernels, same arithmetic, other dispatch order:
// bit-identical to the serial flow (tests/test_gpu_overlap.py).
void run_overlapped(bsrnn_ctx* c, Part p, const bsrnn_ctx::OvlTable* tb, int first, int last)
{
    hipStream_t A = p.s, B = c->aux[0];
    p.ovl = tb;
    p.ovl_mode = c->overlap_mode;
    const bool band = p.ovl_mode & 1, mask = p.ovl_mode & 2, serial = p.ovl_mode & 4;
    Part pb = p;
    if (!serial) pb.s = B;
    const int S = time_lstm_seqs(p.C * c->K), nwg = (p.C * c->K + S - 1) / S, limit = OVL_SPIN_LIMIT;
    // this call's epoch (upper bits of every progress word it publishes or waits for) and the gates' targets (running totals)
    if (++c->ovl_epoch >= c->ovl_epoch_period || c->ovl_resident_total[0] > (1 << 30) || c->ovl_resident_total[1] > (1 << 30)) {   // start again: nothing in flight, every word zero
        (void)hipDeviceSynchronize();
        (void)hipMemset(c->d_ovl, 0, (size_t)2 * c->ovl_stride * sizeof(int));
        for (int b = 0; b < 2; ++b) if (c->ovl_sig[b]) (void)hipMemset(c->ovl_sig[b], 0, 8);
        c->ovl_epoch = 1; c->ovl_resident_total[0] = c->ovl_resident_total[1] = 0;
    }
    p.ovl_base = pb.ovl_base = c->ovl_epoch << OVL_EPOCH_SHIFT;
    if (band) c->ovl_resident_total[0] += nwg;
    if (mask) c->ovl_resident_total[1] += nwg;
    const bool cp = c->ovl_sig[0] != nullptr && !serial;
    // hold stream `s` until every workgroup of time launch `blk` of THIS call is resident
    auto gate = [&](int blk, hipStream_t s) {
        if (cp) (void)hipStreamWaitValue32(s, c->ovl_sig[blk], (uint32_t)c->ovl_resident_total[blk], hipStreamWaitValueGte, 0xffffffffu);
        else launch_ovl_gate(c->d_ovl + blk * c->ovl_stride, c->ovl_resident_total[blk], c->d_range, limit, s);
    };
    const bool ev_in_launch = band && !cp && !serial;
    p.band_done[0] = pb.band_done[0] = ev_in_launch ? c->ev_ovl_fork : nullptr;
    p.band_done[1] = pb.band_done[1] = ev_in_launch && mask ? c->ev_ovl_mid : nullptr;
    for (int st = first; st <= MS_BANDSPLIT; ++st) run_stage(c, p, st);
    run_stage(c, p, MS_BAND0);
    // The auxiliary stream is joined on the host where something needs it (finish_call under the default range policy, bsrnn_sync, a stream
    // switch) or by an event for a call that returns LSTM state; a join event in front of the iSTFT cost the caller's stream a 6 us gap.
    auto join = [&]() {
        if (serial) return;
        if (p.state_out) { (void)hipEventRecord(c->ev_ovl_join, B); (void)hipStreamWaitEvent(A, c->ev_ovl_join, 0); }
        else c->ovl_unjoined = true;
    };
    if (band) {
        // command-processor gates: the auxiliary stream is ordered behind the caller's by the words alone (no fork).  Kernel gates: the
        // fork event keeps gate 0 off band 0's dispatch, and gate 1 waits (event) until band 1 has been dispatched completely.
        if (!cp && !serial) (void)hipStreamWaitEvent(B, c->ev_ovl_fork, 0);          // (signalled by band 0's own dispatch: p.band_done)
        run_stage(c, p, MS_TIME0);
        gate(0, pb.s);
        run_stage(c, pb, MS_BAND1);                                                  // (... and the mid event by band 1's)
        run_stage(c, pb, MS_TIME1);
        if (mask) {
            if (!cp && !serial) (void)hipStreamWaitEvent(A, c->ev_ovl_mid, 0);
            gate(1, A);
        } else if (!serial) { (void)hipEventRecord(c->ev_ovl_join, B); (void)hipStreamWaitEvent(A, c->ev_ovl_join, 0); }      // (the mask chain does not wait per workgroup)
        run_stage(c, p, MS_MASK);
        if (mask) join();
    } else {                                      // the mask chain alone beside the second time-axis launch
        run_stage(c, p, MS_TIME0);
        run_stage(c, p, MS_BAND1);
        if (!serial) { (void)hipEventRecord(c->ev_ovl_fork, A); (void)hipStreamWaitEvent(B, c->ev_ovl_fork, 0); }           // (time 1 reads what band 1 wrote)
        run_stage(c, pb, MS_TIME1);
        gate(1, A);
        run_stage(c, p, MS_MASK);
        join();
    }
    for (int st = MS_MASK + 1; st <= last; ++st) run_stage(c, p, st);
    // A command-processor wait has no time-out of its own: if a launch of this call failed, its producer may never arrive - release the waits
    // from the host (the caller gets the launch error through hipGetLastError at the end of the entry point).
    if (cp && hipPeekAtLastError() != hipSuccess)
        for (int b = 0; b < 2; ++b) *(volatile int*)c->ovl_sig[b] = c->ovl_resident_total[b];      // (signal memory is host-visible)
}
// host-side join of the auxiliary stream with the last overlapped call (see run_overlapped)
static int ovl_join_host(bsrnn_ctx* c)
{
    if (c->ovl_unjoined && c->aux[0]) HIP_TRY(hipStreamSynchronize(c->aux[0]));
    c->ovl_unjoined = false;
    return 0;
}
static const bsrnn_ctx::OvlTable* ovl_table(const bsrnn_ctx* c, int C, int T, hipStream_t s)
{
    if (!overlap_wanted(c, C, T) || !c->d_ovl) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }      // (the legacy default stream cannot be captured)
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    auto it = c->ovl_tables.find(std::make_pair(C, T));
    return it == c->ovl_tables.end() ? nullptr : &it->second;
}

// The model proper for a single part on one stream.
int run_model(bsrnn_ctx* c, const float* Xf, float* Yf, float* tap, int C, int T,
              const float* state_in, float* state_out, hipStream_t s)
{
    static const bool gemv_on = [] { const char* e = getenv("BSRNN_GEMV"); return !(e && !strcmp(e, "0")); }();
    c->small_rows = gemv_on && C * T <= GEMV_MAX_FRAME_ROWS;
    Part p = make_part(c, 0, C, T, s);
    p.Xf = Xf; p.Yf = Yf; p.tap = tap;
    p.state_in = state_in; p.state_out = state_out;
    p.state_slab = (size_t)2 * 2 * C * c->K * HID;     // one Time block's (h,c) x 2 layers
    const bsrnn_ctx::OvlTable* tb = c->small_rows ? nullptr : ovl_table(c, C, T, s);
    if (tb) run_overlapped(c, p, tb, MS_BANDSPLIT, MS_MASK);
    else
    for (int st = MS_BANDSPLIT; st <= MS_MASK; ++st) run_stage(c, p, st);
    c->small_rows = false;
    if (c->stage_error) { c->stage_error = false; return fail(BSRNN_ESTATE, "no task table for %d frame rows (internal error)", C * T); }
    HIP_TRY(hipGetLastError());
    return 0;
}

int check_range(bsrnn_ctx* c)
{
    if (c->h_range && *(volatile int*)c->h_range) {
        const int v = *(volatile int*)c->h_range;
        *(volatile int*)c->h_range = 0;
        if (v == 3) return fail(BSRNN_EHIP, "an earlier time-axis LSTM launch gave up waiting on its own workgroup-local counters (internal error)");
        if (v == 4) {
            c->band_pair_off = true;
            ++c->gen;                                 // captured streaming graphs contain the pair launch: re-capture with one launch per layer
            return fail(BSRNN_EHIP, "an earlier band-axis launch (both layers in one launch, range policy 'deferred') did not find its partner workgroups on the "
                                    "same XCD in time; its results are invalid - repeat the call (the context now runs one launch per layer)");
        }
        if (v == 5) {
            c->overlap_off = true;
            ++c->gen;
            return fail(BSRNN_EHIP, "an earlier call (range policy 'deferred') ran its dual path overlapped and a consumer workgroup gave up waiting for the "
                                    "time-axis launch beside it; its results are invalid - repeat the call (the context now runs launch after launch)");
        }
        return fail(BSRNN_ERANGE, "an earlier call (range policy 'deferred') fed the fp16x2 matrix path an activation beyond +-65504; its "
                                  "results are invalid - repeat it under the default policy, rescale the input or set BSRNN_GEMM=f32 BSRNN_LSTM=f32");
    }
    return 0;
}
// End of a model entry point under the default range policy (include/bsrnn_hip.h): wait for the call's own kernels, look at the
// guard word the fp16x2 kernels set when an operand left the fp16 range, and if it is set run the call again on the library's
// exact-fp32 kernels (fp32 weights are always resident; no range limit) before returning - never wrong numbers with rc 0.
template <class F>
int finish_call(bsrnn_ctx* c, hipStream_t s, F&& rerun)
{
    if (c->range_policy != BSRNN_RANGE_EXACT || !c->h_range || force_f32()) return 0;
    if (gemm_mode() == GEMM_F32 && lstm_mode() == LSTM_F32) return 0;
    HIP_TRY(hipStreamSynchronize(s));
    if (int rcj = ovl_join_host(c)) return rcj;      // (the second time-axis launch of an overlapped call ran on the auxiliary stream: its guard word too)
    const int v = *(volatile int*)c->h_range;
    if (!v) return 0;
    *(volatile int*)c->h_range = 0;
    if (v == 3) return fail(BSRNN_EHIP, "the time-axis LSTM launch gave up waiting on its own workgroup-local counters (internal error)");
    // Structural fall-backs first (same arithmetic, fewer assumptions about dispatch): 5 = a consumer of the overlapped dual path gave up
    // waiting for the time-axis launch beside it -> launch after launch; 4 = the band-pair launch's partners did not meet (placement /
    // dispatch order not as assumed) -> one launch per layer.  Each sticks to the context; at most one re-run per kind.
    int vv = v;
    for (int tries = 0; tries < 2 && (vv == 4 || vv == 5); ++tries) {
        if (vv == 4) c->band_pair_off = true; else c->overlap_off = true;
        ++c->gen;                                 // (a streaming graph captured with the old flow is re-captured by the re-run)
        if (int rc4 = rerun()) return rc4;
        HIP_TRY(hipStreamSynchronize(s));
        if (int rcj = ovl_join_host(c)) return rcj;
        vv = *(volatile int*)c->h_range;
        if (!vv) return 0;
        *(volatile int*)c->h_range = 0;
    }
    if (vv == 3 || vv == 4 || vv == 5) return fail(BSRNN_EHIP, "a recurrent launch gave up waiting on its partners (internal error, guard value %d)", vv);
    set_force_f32(true);
    const int rc = rerun();
    set_force_f32(false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
int check_ready(bsrnn_ctx* c)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0) return fail(BSRNN_ESTATE, "host-only context (device -1): no compute entry points");
    if (c->zombie) return fail(BSRNN_ESTATE, "context was destroyed");
    if (!c->committed) return fail(BSRNN_ESTATE, "bsrnn_commit_params() has not been called");
    HIP_TRY(hipSetDevice(c->device));
    return check_range(c);
}

// One call at a time per context: an overlapping call from another host thread is refused instead of racing on the
// workspace.  (Re-entrant on the same thread: bsrnn_evaluate -> bsrnn_separate, stream_step_host -> stream_step.)
thread_local bsrnn_ctx* tl_owner = nullptr;
struct CallGuard {
    bsrnn_ctx* c; bool ok, outer;
    explicit CallGuard(bsrnn_ctx* c_) : c(c_), ok(true), outer(false)
    {
        if (!c || tl_owner == c) return;
        int expect = 0;
        ok = c->busy.compare_exchange_strong(expect, 1);
        if (ok) { outer = true; tl_owner = c; }
    }
    ~CallGuard() { if (outer) { tl_owner = nullptr; c->busy.store(0); } }
    int refuse() const { return fail(BSRNN_ESTATE, "context is in use by a call from another host thread (one call at a time per context)"); }
};
// The context has one workspace: work enqueued by the previous call on ANOTHER stream must have finished before this
// call's kernels may touch it.  Same stream (the normal case): stream order does it, nothing to do here.
int order_after_last(bsrnn_ctx* c, hipStream_t s)
{
    if (c->have_last && c->last_stream != s) { HIP_TRY(hipStreamSynchronize(c->last_stream)); if (int rcj = ovl_join_host(c)) return rcj; }
    c->last_stream = s; c->have_last = true;
    return 0;
}
#define ENTER_CALL(c, s)                                   \
    CallGuard guard_(c);                                   \
    if (!guard_.ok) return guard_.refuse();                \
    { int rc_ = order_after_last(c, s); if (rc_) return rc_; }

}  // namespace

// =========================================================================== ABI
extern "C" {

int bsrnn_abi_version(void) { return BSRNN_ABI_VERSION; }
const char* bsrnn_last_error(void) { return g_err; }
const char* bsrnn_compute_mode(void)
{
    static char buf[64];
    const int g = gemm_mode(), l = lstm_mode();
    snprintf(buf, sizeof buf, "gemm=%s lstm=%s", g == GEMM_F32 ? "f32" : (g == GEMM_FP16X2 ? "fp16x2" : (g == GEMM_BF16 ? "bf16" : "fp16")),
             l == LSTM_F32 ? "f32" : "fp16x2");
    return buf;
}

int bsrnn_create(int device, const int32_t* widths, int32_t n_bands, bsrnn_ctx** out)
{
    if (!out || !widths || n_bands < 1 || n_bands > 256) return fail(BSRNN_EARG, "bad band table");
    int sum = 0;
    for (int i = 0; i < n_bands; ++i) {
        if (widths[i] < 0) return fail(BSRNN_EARG, "negative band width");
        sum += widths[i];
    }
    if (sum != NBINS) return fail(BSRNN_EARG, "band widths sum to %d, expected %d", sum, NBINS);
    if (device != -1) {
        int ndev = 0;
        HIP_TRY(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) return fail(BSRNN_EARG, "device %d out of range (%d devices)", device, ndev);
        HIP_TRY(hipSetDevice(device));
    }
    bsrnn_ctx* c = new bsrnn_ctx();
    c->device = device;
    c->K = n_bands;
    c->widths.assign(widths, widths + n_bands);
    int pos = 0, ao = 0, po = 0;
    for (int i = 0; i < n_bands; ++i) {
        c->off.push_back(pos); pos += widths[i];
        // 32-column granularity: the same per-band offsets (x 2, in 16-bit elements) address the slab-format activations,
        // whose bands are padded to whole 32-deep slabs
        c->aoff.push_back(ao); ao += (imax(2 * widths[i], 2 * HID) + 31) & ~31;
        c->poff.push_back(po); po += round8(2 * widths[i]);
    }
    c->LDA = ao; c->LDP = imax(po, 8);
    build_inventory(c);
    memset(c->acc_ms, 0, sizeof c->acc_ms);
    memset(c->acc_n, 0, sizeof c->acc_n);
    if (device == -1) { *out = c; return 0; }     // host-only: inventory, set / get, weight-file validation
    {   // range-guard word: pinned host memory the fp16x2 kernels can set and the host can read without a sync
        hipError_t e = hipHostMalloc((void**)&c->h_range, sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) { *c->h_range = 0; e = hipHostGetDevicePointer((void**)&c->d_range, c->h_range, 0); }
        if (e != hipSuccess) { c->h_range = nullptr; c->d_range = nullptr; (void)hipGetLastError(); }
    }
    if (const char* e = getenv("BSRNN_OVERLAP")) {
        c->overlap_env = strcmp(e, "0") != 0; c->overlap_sabotage = !strcmp(e, "timeout");
        if (!strcmp(e, "band")) c->overlap_mode = 1;
        if (!strcmp(e, "mask")) c->overlap_mode = 2;
        if (!strcmp(e, "pub")) c->overlap_mode = 3 | 4;
    }
    if (const char* e = getenv("BSRNN_OVL_EPOCHS")) c->ovl_epoch_period = std::max(2, std::min(1 << 18, atoi(e)));
    if (const char* e = getenv("BSRNN_PARTS")) c->n_parts = std::max(0, std::min(MAX_PARTS, atoi(e)));
    if (const char* e = getenv("BSRNN_PART_LAG")) c->part_lag = std::max(0, std::min((int)MS_COUNT, atoi(e)));

    // FFT tables in double precision
    std::vector<float> t(2 * 1024 + 2 * 1025 + 2048 + 1024 + 1024 + 16);
    size_t o = 0;
    const double PI = 3.14159265358979323846;
    size_t o_tw1 = o;
    for (int k = 0; k < 1024; ++k) { t[o++] = (float)cos(2 * PI * k / 1024); t[o++] = (float)(-sin(2 * PI * k / 1024)); }
    size_t o_tw2 = o;
    for (int k = 0; k <= 1024; ++k) { t[o++] = (float)cos(2 * PI * k / 2048); t[o++] = (float)(-sin(2 * PI * k / 2048)); }
    o = (o + 3) & ~size_t(3);
    size_t o_hann = o;
    std::vector<float> w(2048);
    for (int i = 0; i < 2048; ++i) { w[i] = (float)(0.5 - 0.5 * cos(2 * PI * i / 2048)); t[o++] = w[i]; }
    size_t o_env = o;
    for (int i = 0; i < 1024; ++i) {     // torch.istft envelope: overlap-add of window^2 (float), inverted
        const float e = w[i] * w[i] + w[i + 1024] * w[i + 1024];
        t[o++] = 1.0f / e;
    }
    size_t o_ws = o;
    for (int i = 0; i < 1024; ++i) t[o++] = 1.0f / (w[i] + w[i + 1024]);
    hipError_t e = hipMalloc((void**)&c->d_tables, o * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(c->d_tables, t.data(), o * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete c; return fail(BSRNN_EHIP, "table upload: %s", hipGetErrorString(e)); }
    c->tb.tw1024 = (const float2*)(c->d_tables + o_tw1);
    c->tb.tw2048 = (const float2*)(c->d_tables + o_tw2);
    c->tb.hann = c->d_tables + o_hann;
    c->tb.inv_env = c->d_tables + o_env;
    c->tb.inv_wsum = c->d_tables + o_ws;
    {   // bin -> column of its real part in the band-padded spectrum layout
        std::vector<int> cm(NBINS);
        for (int i = 0; i < n_bands; ++i)
            for (int k = 0; k < widths[i]; ++k) cm[c->off[i] + k] = c->poff[i] + 2 * k;
        e = hipMalloc((void**)&c->d_colmap, NBINS * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(c->d_colmap, cm.data(), NBINS * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete c; return fail(BSRNN_EHIP, "table upload: %s", hipGetErrorString(e)); }
        c->tb.colmap = c->d_colmap;
        c->tb.ld = c->LDP;
    }
    *out = c;
    return 0;
}

static void destroy_now(bsrnn_ctx* c)
{
    if (c->device < 0) { delete c; return; }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& r : c->pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (int j = 0; j < MAX_PARTS; ++j) {
        if (c->aux[j]) (void)hipStreamDestroy(c->aux[j]);
        if (c->ev_join[j]) (void)hipEventDestroy(c->ev_join[j]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_ovl_fork) (void)hipEventDestroy(c->ev_ovl_fork);
    if (c->ev_ovl_join) (void)hipEventDestroy(c->ev_ovl_join);
    free_ovl_tables(c);
    if (c->d_ovl) (void)hipFree(c->d_ovl);
    for (int b = 0; b < 2; ++b) if (c->ovl_sig[b]) (void)hipFree(c->ovl_sig[b]);
    if (c->ev_ovl_mid) (void)hipEventDestroy(c->ev_ovl_mid);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_tap) (void)hipFree(c->d_tap);
    if (c->d_arena) (void)hipFree(c->d_arena);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    for (int ch = 0; ch < 2; ++ch)
        if (c->d_chain[ch]) (void)hipFree(c->d_chain[ch]);
    for (auto& kv : c->chain_tasks)
        for (int ch = 0; ch < 2; ++ch) (void)hipFree(kv.second.d[ch]);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_train_ws) (void)hipFree(c->d_train_ws);
    for (float* p : c->train_ws_retired) (void)hipFree(p);
    for (void* p : c->retired) (void)hipFree(p);
    if (c->d_colmap) (void)hipFree(c->d_colmap);
    if (c->h_range) (void)hipHostFree(c->h_range);
    delete c;
}

void bsrnn_destroy(bsrnn_ctx* c)
{
    if (!c) return;
    // streams hold a pointer to their context (and a captured graph holds the context's buffers): with streams alive the
    // context only stops accepting calls here and is freed when the last of them is destroyed
    if (c->live_streams > 0) { c->zombie = true; return; }
    destroy_now(c);
}

int bsrnn_n_bands(const bsrnn_ctx* c) { return c ? c->K : -1; }
int bsrnn_mlp_fused(const bsrnn_ctx* c) { return c ? (c->fused ? 1 : 0) : -1; }
int bsrnn_device(const bsrnn_ctx* c) { return c ? c->device : -1; }
int bsrnn_debug_peek(bsrnn_ctx* c, int32_t which, float* host_out, int64_t nfloats)
{
    if (!c || !host_out || nfloats < 0 || which < 0 || which > 4) return fail(BSRNN_EARG, "bsrnn_debug_peek: bad arguments");
    if (c->device < 0 || !c->d_ws) return fail(BSRNN_ESTATE, "bsrnn_debug_peek: no workspace yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t KH = (size_t)c->K * HID;
    const float* src[5] = {c->Z0, c->Z1, c->HB1, c->P, c->Yf};
    const size_t cap[5] = {c->cap_rows * KH, c->cap_rows * KH, c->cap_rows * KH * 2, c->cap_rows * c->LDP, c->cap_rows * c->LDP};
    if ((size_t)nfloats > cap[which]) return fail(BSRNN_EARG, "bsrnn_debug_peek: the buffer holds %zu floats", cap[which]);
    HIP_TRY(hipMemcpy(host_out, src[which], (size_t)nfloats * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
long long bsrnn_debug_counter(int32_t which) { return which >= 0 && which < 4 ? g_dbg[which].load() : -1; }
int bsrnn_overlap_state(const bsrnn_ctx* c) { return !c ? -1 : (c->overlap_off ? 2 : (c->overlap_env ? 1 : 0)); }
int bsrnn_set_range_policy(bsrnn_ctx* c, int32_t policy)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (policy != BSRNN_RANGE_EXACT && policy != BSRNN_RANGE_DEFERRED) return fail(BSRNN_EARG, "bsrnn_set_range_policy: unknown policy %d", policy);
    c->range_policy = policy;
    return 0;
}
int bsrnn_get_range_policy(const bsrnn_ctx* c) { return c ? c->range_policy : -1; }
int bsrnn_param_count(const bsrnn_ctx* c) { return c ? (int)c->params.size() : -1; }

int bsrnn_param_info(const bsrnn_ctx* c, int32_t i, const char** key, int64_t* d0, int64_t* d1, int32_t* ndim)
{
    if (!c || i < 0 || i >= (int)c->params.size()) return fail(BSRNN_EARG, "param index out of range");
    const Param& p = c->params[i];
    if (key) *key = p.key.c_str();
    if (d0) *d0 = p.d0;
    if (d1) *d1 = p.d1;
    if (ndim) *ndim = p.ndim;
    return 0;
}

int bsrnn_set_param(bsrnn_ctx* c, const char* key, const float* host, int64_t numel)
{
    if (!c || !key) return fail(BSRNN_EARG, "null argument");
    auto it = c->index.find(key);
    if (it == c->index.end()) return fail(BSRNN_ENOKEY, "unexpected key '%s'", key);
    Param& p = c->params[it->second];
    if (numel != p.numel()) return fail(BSRNN_EARG, "size mismatch for %s: got %lld, expected %lld", key, (long long)numel, (long long)p.numel());
    if (numel > 0 && !host) return fail(BSRNN_EARG, "null data for %s", key);
    p.data.assign(host, host + numel);
    p.set = true;
    c->committed = false;
    return 0;
}

int bsrnn_get_param(const bsrnn_ctx* c, const char* key, float* out, int64_t numel)
{
    if (!c || !key) return fail(BSRNN_EARG, "null argument");
    auto it = c->index.find(key);
    if (it == c->index.end()) return fail(BSRNN_ENOKEY, "unexpected key '%s'", key);
    const Param& p = c->params[it->second];
    if (!p.set) return fail(BSRNN_ESTATE, "%s was never set", key);
    if (numel != p.numel()) return fail(BSRNN_EARG, "size mismatch for %s", key);
    if (numel) memcpy(out, p.data.data(), numel * sizeof(float));
    return 0;
}

int bsrnn_commit_params(bsrnn_ctx* c)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    for (const Param& p : c->params)
        if (!p.set) return fail(BSRNN_ESTATE, "missing key '%s'", p.key.c_str());
    if (c->device < 0) return 0;          // host-only context: the inventory is complete, there is nothing to upload
    CallGuard guard_(c);
    if (!guard_.ok) return guard_.refuse();
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    ++c->gen;                             // the arena is rebuilt: captured streaming graphs hold pointers into the old one

    Arena ar;
    std::vector<GemmJob> jobs;
    std::vector<size_t> jw, jb, jwp;     // arena offsets, patched to pointers after upload
    // fp16x2 / fp16 modes: the same matrix as two fp16 pieces, slab-interleaved, rows padded to a multiple of 32
    // (gemm_h2_kernel), packed into arena floats
    const int gmode = gemm_mode();
    auto put_planes = [&](const std::vector<float>& wp, int N, int Kp) -> size_t {
        if (gmode == GEMM_F32 || wp.empty()) return 0;
        std::vector<uint16_t> pl;
        const int K32 = (Kp + 31) & ~31, wrow = h2_row_stride(K32);
        pl.assign((size_t)N * wrow + 1, 0);
        pack_h2_slabs_host(wp.data(), N, Kp, Kp, K32, wrow, pl.data());
        std::vector<float> packed(pl.size() / 2 + 1);
        memcpy(packed.data(), pl.data(), pl.size() * sizeof(uint16_t));
        return ar.put(packed);
    };
    std::vector<int2> tiles;
    const int H = HID, K = c->K;
    char b[128];

    auto add_job = [&](const char* prefix, int N, int Kd, int x_off, int y_off, int r_off, int m_off) {
        GemmJob j;
        memset(&j, 0, sizeof j);
        j.N = N; j.K = Kd; j.x_off = x_off; j.y_off = y_off; j.r_off = r_off; j.m_off = m_off;
        const Param& w = P_(c, std::string(prefix) + ".weight");
        const Param& bi = P_(c, std::string(prefix) + ".bias");
        // weight rows padded with zeros to a multiple of 8: every row is 16-byte aligned in fp32 and in the 16-bit
        // planes, and the kernels' K loops run over whole 16-byte units (the matching input pad columns are zero,
        // see ensure_ws and the GEMM epilogue)
        const int Kp = round8(Kd);
        j.K = Kp;
        std::vector<float> wp((size_t)N * Kp, 0.f);
        for (int r = 0; r < N; ++r) memcpy(&wp[(size_t)r * Kp], &w.data[(size_t)r * Kd], Kd * sizeof(float));
        jw.push_back(ar.put(wp));
        jb.push_back(ar.put(bi.data));
        jwp.push_back(put_planes(wp, N, Kp));
        j.wrow = h2_row_stride((Kp + 31) & ~31);
        jobs.push_back(j);
    };
    auto begin_slot = [&](int slot) { c->job0[slot] = (int)jobs.size(); c->tile0[slot] = (int)tiles.size(); };
    // column tiles of a slot: 128 wide when any layer of the slot is wider than 64 columns (the
    // 128 x 128 kernel does twice the MFMA work per barrier), else 64; heaviest K first so the
    // tail of a launch is made of cheap tiles.  tiles[].x is relative to the slot's first job.
    auto end_slot = [&](int slot) {
        const int j0 = c->job0[slot];
        c->njobs[slot] = (int)jobs.size() - j0;
        int maxn = 0;
        for (int ji = j0; ji < (int)jobs.size(); ++ji) maxn = imax(maxn, jobs[ji].N);
        // 128-wide tiles pay off only when the launch has many more workgroups than CU slots (uniform
        // large GEMMs: 114 vs 99 TFLOP/s); at M = C*T ~ 8k rows the 64-wide tiling balances the ragged
        // per-band costs better (measured 2.75 vs 2.79 ms per step), so it is the default.
        // The split-precision kernels do 2-3x less matrix-pipe work per tile and are bound by the CU's load
        // path instead: there the 128-wide tile (2/3 of the bytes per flop) wins (tools/gemm_planes_bench.hip).
        const int tn = (maxn > 64 && (gmode != GEMM_F32 || getenv("BSRNN_GEMM_TILE128"))) ? 128 : 64;
        c->tile_n[slot] = tn;
        std::vector<int> order;
        for (int ji = j0; ji < (int)jobs.size(); ++ji) order.push_back(ji);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return jobs[a].K > jobs[b].K; });
        for (int ji : order)
            for (int t = 0; t < (jobs[ji].N + tn - 1) / tn; ++t) tiles.push_back(make_int2(ji - j0, t));
        c->ntiles[slot] = (int)tiles.size() - c->tile0[slot];
    };

    // per-band MLP chains; tiles[].x is relative to the slot's first job
    struct Def { int slot; const char* fmt; };
    for (int slot = PRE0; slot <= POST2; ++slot) {
        begin_slot(slot);
        for (int i = 0; i < K; ++i) {
            const int a = 2 * c->widths[i];
            const int xin = c->poff[i];            // column of the band in the band-padded spectrum layout (16-byte aligned)
            const int m = imax(a, H), pz = imax(a, 2 * H);
            if (a == 0) {
                if (slot == FC4) {                 // TrainableConstantModule -> Z[:, :, i, :] = constant
                    GemmJob j;
                    memset(&j, 0, sizeof j);
                    j.N = H; j.K = 0; j.y_off = i * H;
                    snprintf(b, sizeof b, "bandFCs.%d.0.trainable_constant", i);
                    const Param& cst = P_(c, b);
                    jw.push_back(ar.put(cst.data)); jb.push_back(ar.put(cst.data)); jwp.push_back(0);
                    jobs.push_back(j);
                }
                continue;
            }
            switch (slot) {
            case PRE0:  snprintf(b, sizeof b, "bandFCs_pre.%d.0", i); add_job(b, a, a, xin, c->aoff[i], 0, 0); break;
            case PRE2:  snprintf(b, sizeof b, "bandFCs_pre.%d.2", i); add_job(b, a, a, c->aoff[i], c->poff[i], 0, 0); break;
            case FC0:   snprintf(b, sizeof b, "bandFCs.%d.0", i); add_job(b, m, a, c->poff[i], c->aoff[i], 0, 0); break;
            case FC2:   snprintf(b, sizeof b, "bandFCs.%d.2", i); add_job(b, H, m, c->aoff[i], c->aoff[i], 0, 0); break;
            case FC4:   snprintf(b, sizeof b, "bandFCs.%d.4", i); add_job(b, H, H, c->aoff[i], i * H, 0, 0); break;
            case BACK0: snprintf(b, sizeof b, "bandFCs_back.%d.0", i); add_job(b, 2 * H, H, i * H, c->aoff[i], 0, 0); break;
            case BACK2: snprintf(b, sizeof b, "bandFCs_back.%d.2", i); add_job(b, pz, 2 * H, c->aoff[i], c->aoff[i], 0, 0); break;
            case BACK4: snprintf(b, sizeof b, "bandFCs_back.%d.4", i); add_job(b, a, pz, c->aoff[i], c->aoff[i], 0, 0); break;
            case POST0: snprintf(b, sizeof b, "bandFCs_back_post.%d.0", i); add_job(b, a, a, c->aoff[i], c->aoff[i], 0, 0); break;
            case POST2: snprintf(b, sizeof b, "bandFCs_back_post.%d.2", i); add_job(b, a, a, c->aoff[i], xin, c->poff[i], xin); break;
            }
        }
        end_slot(slot);
    }
    // fc of the four NormRNNResidual blocks (bsrnn.py:84), one job each over M*K rows
    for (int j = 0; j < 4; ++j) {
        const int slot = BLK_FC0 + j;
        begin_slot(slot);
        snprintf(b, sizeof b, "lstms.%d.m.fc", j);
        add_job(b, H, (j % 2 == 0) ? 2 * H : H, 0, 0, 0, 0);
        end_slot(slot);
    }

    // fused chains (mlp_chain.hip): per band and chain the five layers' fragment streams, the concatenated biases and a
    // descriptor; classes by rows per workgroup (the activation image of a row tile must fit 96 KB / RT of LDS)
    std::vector<ChainDesc> chains[2];
    std::vector<size_t> ch_w[2], ch_b[2];           // arena offsets, patched to pointers after upload
    bool fused = gmode != GEMM_F32;
    if (const char* e = getenv("BSRNN_MLP")) fused = fused && strcmp(e, "layers") != 0;
    for (int ch = 0; ch < 2 && fused; ++ch) {
        struct Built { ChainDesc d; size_t w, b; long cost; };
        std::vector<Built> built;
        for (int i = 0; i < K && fused; ++i) {
            const int a = 2 * c->widths[i], m = imax(a, H), pz = imax(a, 2 * H);
            Built bu;
            memset(&bu.d, 0, sizeof bu.d);
            ChainDesc& d = bu.d;
            d.p_off = c->poff[i]; d.a8 = round8(a); d.z_off = i * H;
            if (a == 0) {
                if (ch != CHAIN_SPLIT) continue;
                snprintf(b, sizeof b, "bandFCs.%d.0.trainable_constant", i);
                d.constant = 1; d.NW = 1; d.RT = 1; d.nbias = H;     // (a 256-row class member, like the GR = 8 geometry)
                bu.w = 0; bu.b = ar.put(P_(c, b).data); bu.cost = -1;
                built.push_back(bu);
                continue;
            }
            struct LD { const char* fmt; int N, Kd, leaky; };
            const LD split_l[5] = {{"bandFCs_pre.%d.0", a, a, 1}, {"bandFCs_pre.%d.2", a, a, 1}, {"bandFCs.%d.0", m, a, 1},
                                   {"bandFCs.%d.2", H, m, 1}, {"bandFCs.%d.4", H, H, 0}};
            const LD mask_l[5] = {{"bandFCs_back.%d.0", 2 * H, H, 1}, {"bandFCs_back.%d.2", pz, 2 * H, 1}, {"bandFCs_back.%d.4", a, pz, 1},
                                  {"bandFCs_back_post.%d.0", a, a, 1}, {"bandFCs_back_post.%d.2", a, a, 0}};
            const LD* ld = ch == CHAIN_SPLIT ? split_l : mask_l;
            int units = 0, maxntl = 0, nbias = 0;
            long cost = 0;
            for (int l = 0; l < CHAIN_LAYERS; ++l) {
                d.L[l].K16 = (ld[l].Kd + 15) / 16; d.L[l].NTL = (ld[l].N + 31) / 32; d.L[l].leaky = ld[l].leaky;
                d.L[l].bias_off = nbias; nbias += 32 * d.L[l].NTL;
                units = imax(units, 2 * d.L[l].K16);
                if (l + 1 < CHAIN_LAYERS) units = imax(units, 4 * d.L[l].NTL);
                maxntl = imax(maxntl, d.L[l].NTL);
                cost += (long)d.L[l].K16 * d.L[l].NTL;
            }
            const int img = 2 * units * 512;                                  // bytes of one row tile's image (both pieces)
            // geometry (mlp_chain.hip): RT row tiles per wave group (each weight fragment is used for all of them), GR groups
            int RT = 0, GR = 1;
            if (8 * img <= CHAIN_LDS_EX && maxntl <= 4) { RT = 1; GR = 8; }          // narrowest: every wave a chain of its own
            else if (4 * img <= CHAIN_LDS_EX && maxntl <= 6) { RT = 1; GR = 4; }     // narrow: four groups of two waves
            else if (4 * img <= CHAIN_LDS_EX && maxntl <= 12) { RT = 2; GR = 2; }    // two groups of four waves, two row tiles each
            else if (2 * img <= CHAIN_LDS_EX) { RT = 2; GR = 1; }
            else if (img <= CHAIN_LDS_EX) { RT = 1; GR = 1; }
            // the widest bands (32 rows would be all the LDS holds): 48 rows on 16 x 16 x 32 MFMAs instead (chain_body48); there the
            // layer fields count k-steps of 32 and feature tiles of 16
            // ... and bands of the 64-row class whose image leaves room for FIVE row tiles of 16 and whose feature tiles of 16 are at most
            // three per wave (the 384-wide band: 24 tiles = 3 x 8 where the 32 x 32 geometry has twelve tiles for eight waves): 80 rows
            // per weight fragment instead of 64, every wave busy (BSRNN_CHAIN_NO80=1 keeps them on the 32 x 32 geometry)
            bool g48 = false;
            const bool no48 = getenv("BSRNN_CHAIN_NO48") != nullptr;
            const bool try48 = RT == 1 && GR == 1 && !no48;
            const bool try80 = RT == 2 && GR == 1 && !no48 && !getenv("BSRNN_CHAIN_NO80");
            // ... and the other bands of the 64-row class (the 514-wide band: 33 feature tiles of 16, ragged) on FOUR row tiles of 16: the same 64
            // rows, but two feature tiles' fragments per k-step and four k-steps in flight per wave (128 KB per CU instead of the 64 KB the
            // two-row-tile 32 x 32 body has registers for, which held its K loops at 48 GB/s per CU against the 70 the fill path gives:
            // profiles/r03_chain_trace.txt); BSRNN_CHAIN_NO64=1 keeps them on the 32 x 32 geometry
            bool try64 = false;
            if (try48 || try80) {
                int rt16 = try48 ? 3 : 5, ctr = try48 ? 6 : 3;
                int u48 = 0, maxft = 0, nb48 = 0;
                bool whole = true;                                   // every layer's width a multiple of 16 (no ragged tile of 16)
                for (int l = 0; l < CHAIN_LAYERS; ++l) {
                    const int K32 = (ld[l].Kd + 31) / 32, FT = (ld[l].N + 15) / 16;
                    u48 = imax(u48, 4 * K32);
                    if (l + 1 < CHAIN_LAYERS) u48 = imax(u48, 2 * FT);
                    maxft = imax(maxft, FT); nb48 += 16 * FT;
                    whole = whole && ld[l].N % 16 == 0;
                }
                if (try80 && !(whole && maxft % 8 == 0 && maxft <= 8 * ctr && 2 * u48 * (16 * rt16) * 16 <= CHAIN_LDS_EX) && !getenv("BSRNN_CHAIN_NO64")) {
                    try64 = true; rt16 = 4; ctr = 5;
                }
                if (2 * u48 * (16 * rt16) * 16 <= CHAIN_LDS_EX && maxft <= 8 * ctr && nb48 * 4 <= CHAIN_LDS_BIAS && (try48 || try64 || (whole && maxft % 8 == 0))) {
                    g48 = true; RT = rt16; GR = 1; units = u48; nbias = 0; cost = 0;
                    for (int l = 0; l < CHAIN_LAYERS; ++l) {
                        d.L[l].K16 = (ld[l].Kd + 31) / 32; d.L[l].NTL = (ld[l].N + 15) / 16;
                        d.L[l].bias_off = nbias; nbias += 16 * d.L[l].NTL;
                        cost += (long)d.L[l].K16 * d.L[l].NTL;          // (half the MACs of a 32 x 32 x 16 unit each: same scale per row)
                    }
                }
            }
            const int ct_max = GR == 8 ? 4 : CHAIN_CT;                                // feature tiles per wave the geometry's body holds
            if (!g48 && (RT < 1 || (8 / GR) * ct_max < maxntl || nbias * 4 > CHAIN_LDS_BIAS)) { fused = false; break; }   // a band too wide for the fused kernel: per-layer flow
            d.RT = RT; d.NW = 8 / GR; d.plane_units = units; d.nbias = nbias;
            d.in_off = ch == CHAIN_SPLIT ? c->poff[i] : i * H;
            d.K0 = ch == CHAIN_SPLIT ? round8(a) : H;
            // a last feature tile with at most 4 real features (514 columns = 16 tiles + 2) is split over the k-steps of all eight
            // waves instead of costing one wave a whole tile (mlp_chain.hip, split_host.h): the geometry that implements it is
            // RT 2 / GR 1, and the partial sums need 8 KB of LDS behind the two activation images
            static const bool rag_on = [] { const char* e = getenv("BSRNN_CHAIN_RAG"); return !(e && !strcmp(e, "0")); }();
            if (rag_on && !g48 && RT == 2 && GR == 1 && 2 * img + CHAIN_RAG_LDS <= CHAIN_LDS_EX)
                for (int l = 0; l < CHAIN_LAYERS; ++l) {
                    const int tail = ld[l].N % 32;
                    if (tail >= 1 && tail <= 4 && d.L[l].NTL >= 2) {
                        d.L[l].rag = 1;
                        cost -= (long)d.L[l].K16 - d.L[l].K16 / 8;
                    }
                }
            std::vector<uint16_t> stream;
            std::vector<float> biases(nbias, 0.f);
            for (int l = 0; l < CHAIN_LAYERS; ++l) {
                snprintf(b, sizeof b, ld[l].fmt, i);
                const Param& w = P_(c, std::string(b) + ".weight");
                const Param& bi = P_(c, std::string(b) + ".bias");
                d.L[l].w_off = (unsigned)(stream.size() * sizeof(uint16_t));
                const int npl = (gmode == GEMM_FP16 || gmode == GEMM_BF16) ? 1 : 2;
                if (g48) pack_chain_layer16_host(w.data.data(), ld[l].N, ld[l].Kd, ld[l].Kd, 8, npl, stream, gmode == GEMM_BF16);
                else pack_chain_layer_host(w.data.data(), ld[l].N, ld[l].Kd, ld[l].Kd, d.NW, npl, stream, d.L[l].rag, gmode == GEMM_BF16);
                memcpy(&biases[d.L[l].bias_off], bi.data.data(), ld[l].N * sizeof(float));
            }
            stream.resize((stream.size() + 7) & ~size_t(7), 0);
            bu.w = ar.put(reinterpret_cast<const float*>(stream.data()), stream.size() / 2);
            bu.b = ar.put(biases);
            bu.cost = cost;
            built.push_back(bu);
        }
        if (!fused) break;
        // class = rows per workgroup (RT = 1, 2, 4, constant bands), heaviest band first inside a class
        std::stable_sort(built.begin(), built.end(), [](const Built& x, const Built& y) {
            auto cls = [](const ChainDesc& d) { const int rows = chain_rows(d); return d.constant ? 4 : (rows <= 48 ? 0 : (rows <= 80 ? 1 : (rows == 128 ? 2 : 3))); };
            const int cx = cls(x.d), cy = cls(y.d);
            return cx != cy ? cx < cy : x.cost > y.cost;
        });
        c->chain_cost[ch].clear();
        for (const Built& bu : built) {
            chains[ch].push_back(bu.d); ch_w[ch].push_back(bu.w); ch_b[ch].push_back(bu.b);
            c->chain_cost[ch].push_back(bu.cost);
        }
    }

    // LSTM weights, folded and packed in the kernels' register order (lstm.hip)
    size_t o_bandW[2][2], o_bandW16[2][2], o_bandB[2][2], o_timeW[2], o_timeW16[2], o_timeB[2], o_timeFc16[2], o_timeFcB[2], o_bandFc16[2], o_bandFcB[2];
    std::vector<double> wcat, bsum;
    for (int blk = 0; blk < 2; ++blk) {
        const int jb_ = 2 * blk;                               // lstms.0 / lstms.2: bidirectional over bands
        for (int layer = 0; layer < 2; ++layer) {
            const int IN = layer == 0 ? H : 2 * H, KT = IN + H, NS = KT / 4;
            std::vector<float> pk((size_t)2 * 4 * NS * 4 * 64), pb(2 * 256);
            const int NB = KT / 32;
            std::vector<uint16_t> pk16((size_t)2 * 4 * NB * 4 * 2 * 64 * 8);
            for (int d = 0; d < 2; ++d) {
                lstm_cat(c, jb_, layer, d ? "_reverse" : "", IN, wcat, bsum);
                for (int wv = 0; wv < 4; ++wv)
                    for (int bk = 0; bk < NB; ++bk)
                        for (int g = 0; g < 4; ++g)
                            for (int ln = 0; ln < 64; ++ln)
                                for (int e = 0; e < 8; ++e) {
                                    const int row = g * 64 + 16 * wv + (ln & 15);
                                    const int k = 32 * bk + 8 * (ln >> 4) + e;
                                    const float v = (float)wcat[(size_t)row * KT + k];
                                    uint16_t pc[2];
                                    split_planes_host(&v, 1, 2, pc);
                                    const size_t base = ((((size_t)d * 4 + wv) * NB + bk) * 4 + g) * 2;
                                    pk16[((base + 0) * 64 + ln) * 8 + e] = pc[0];
                                    pk16[((base + 1) * 64 + ln) * 8 + e] = pc[1];
                                }
                for (int wv = 0; wv < 4; ++wv)
                    for (int s = 0; s < NS; ++s)
                        for (int g = 0; g < 4; ++g)
                            for (int ln = 0; ln < 64; ++ln) {
                                const int row = g * 64 + 16 * wv + (ln & 15);
                                const int k = 16 * (s / 4) + 4 * (ln >> 4) + (s % 4);
                                pk[((((size_t)d * 4 + wv) * NS + s) * 4 + g) * 64 + ln] = (float)wcat[(size_t)row * KT + k];
                            }
                for (int r = 0; r < 256; ++r) pb[d * 256 + r] = (float)bsum[r];
            }
            o_bandW[blk][layer] = ar.put(pk);
            o_bandB[blk][layer] = ar.put(pb);
            o_bandW16[blk][layer] = ar.put(reinterpret_cast<const float*>(pk16.data()), pk16.size() / 2);
        }
        {   // the block's fc (128 -> 64, bsrnn.py:84) as fp16x2 B fragments for band_block_small_kernel: [4 tile][4 blk][2 piece][64 lane][8],
            // lane (n = l & 15, kb = l >> 4) holds W_fc[16 tile + n][32 blk + 8 kb .. + 7]
            snprintf(b, sizeof b, "lstms.%d.m.fc.weight", jb_); const Param& wfc = P_(c, b);
            snprintf(b, sizeof b, "lstms.%d.m.fc.bias", jb_); const Param& bfc = P_(c, b);
            std::vector<uint16_t> f16((size_t)4 * 4 * 2 * 64 * 8);
            for (int tl = 0; tl < 4; ++tl)
                for (int bk = 0; bk < 4; ++bk)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const float v = wfc.data[(size_t)(16 * tl + (ln & 15)) * 2 * H + 32 * bk + 8 * (ln >> 4) + e];
                            uint16_t pc[2];
                            split_planes_host(&v, 1, 2, pc);
                            const size_t base = ((size_t)tl * 4 + bk) * 2;
                            f16[((base + 0) * 64 + ln) * 8 + e] = pc[0];
                            f16[((base + 1) * 64 + ln) * 8 + e] = pc[1];
                        }
            o_bandFc16[blk] = ar.put(reinterpret_cast<const float*>(f16.data()), f16.size() / 2);
            o_bandFcB[blk] = ar.put(bfc.data);
        }
        const int jt = 2 * blk + 1;                            // lstms.1 / lstms.3: causal over time
        std::vector<float> pk((size_t)2 * 4 * 128 * 64), pb(2 * 256);
        std::vector<uint16_t> pk16((size_t)2 * 4 * 4 * 4 * 2 * 64 * 8);
        for (int layer = 0; layer < 2; ++layer) {
            lstm_cat(c, jt, layer, "", H, wcat, bsum);
            for (int wv = 0; wv < 4; ++wv)
                for (int bk = 0; bk < 4; ++bk)
                    for (int g = 0; g < 4; ++g)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int e = 0; e < 8; ++e) {
                                const int row = g * 64 + 16 * wv + (ln & 15);
                                const int k = 32 * bk + 8 * (ln >> 4) + e;
                                const float v = (float)wcat[(size_t)row * 128 + k];
                                uint16_t pc[2];
                                split_planes_host(&v, 1, 2, pc);
                                const size_t base = ((((size_t)layer * 4 + wv) * 4 + bk) * 4 + g) * 2;
                                pk16[((base + 0) * 64 + ln) * 8 + e] = pc[0];
                                pk16[((base + 1) * 64 + ln) * 8 + e] = pc[1];
                            }
            for (int wv = 0; wv < 4; ++wv)
                for (int k = 0; k < 128; ++k)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int row = (ln & 3) * 64 + 16 * wv + (ln >> 2);
                        pk[(((size_t)layer * 4 + wv) * 128 + k) * 64 + ln] = (float)wcat[(size_t)row * 128 + k];
                    }
            for (int r = 0; r < 256; ++r) pb[layer * 256 + r] = (float)bsum[r];
        }
        o_timeW[blk] = ar.put(pk);
        o_timeW16[blk] = ar.put(reinterpret_cast<const float*>(pk16.data()), pk16.size() / 2);
        o_timeB[blk] = ar.put(pb);
        {   // the block's fc (64 -> 64, bsrnn.py:84) in the f16 MFMA's B-operand order: [4 wave][2 blk][2 piece][64 lane][8],
            // lane (n = l & 15, kb = l >> 4) holds W_fc[16 wave + n][32 blk + 8 kb .. + 7]
            snprintf(b, sizeof b, "lstms.%d.m.fc.weight", jt); const Param& wfc = P_(c, b);
            snprintf(b, sizeof b, "lstms.%d.m.fc.bias", jt); const Param& bfc = P_(c, b);
            std::vector<uint16_t> f16((size_t)4 * 2 * 2 * 64 * 8);
            for (int wv = 0; wv < 4; ++wv)
                for (int bk = 0; bk < 2; ++bk)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const float v = wfc.data[(size_t)(16 * wv + (ln & 15)) * H + 32 * bk + 8 * (ln >> 4) + e];
                            uint16_t pc[2];
                            split_planes_host(&v, 1, 2, pc);
                            const size_t base = ((size_t)wv * 2 + bk) * 2;
                            f16[((base + 0) * 64 + ln) * 8 + e] = pc[0];
                            f16[((base + 1) * 64 + ln) * 8 + e] = pc[1];
                        }
            o_timeFc16[blk] = ar.put(reinterpret_cast<const float*>(f16.data()), f16.size() / 2);
            o_timeFcB[blk] = ar.put(bfc.data);
        }
    }

    // upload
    if (c->d_arena) { HIP_TRY(hipFree(c->d_arena)); c->d_arena = nullptr; }
    if (c->d_jobs) { HIP_TRY(hipFree(c->d_jobs)); c->d_jobs = nullptr; }
    if (c->d_tiles) { HIP_TRY(hipFree(c->d_tiles)); c->d_tiles = nullptr; }
    HIP_TRY(hipMalloc((void**)&c->d_arena, (ar.h.size() + 4) * sizeof(float)));
    HIP_TRY(hipMemcpy(c->d_arena, ar.h.data(), ar.h.size() * sizeof(float), hipMemcpyHostToDevice));
    for (size_t i = 0; i < jobs.size(); ++i) {
        jobs[i].W = c->d_arena + jw[i]; jobs[i].bias = c->d_arena + jb[i];
        jobs[i].Wp = c->d_arena + jwp[i];
    }
    HIP_TRY(hipMalloc((void**)&c->d_jobs, jobs.size() * sizeof(GemmJob)));
    HIP_TRY(hipMemcpy(c->d_jobs, jobs.data(), jobs.size() * sizeof(GemmJob), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&c->d_tiles, tiles.size() * sizeof(int2)));
    HIP_TRY(hipMemcpy(c->d_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    c->fused = false;
    for (auto& kv : c->chain_tasks)
        for (int ch = 0; ch < 2; ++ch) (void)hipFree(kv.second.d[ch]);
    c->chain_tasks.clear();
    for (int ch = 0; ch < 2; ++ch) {
        c->h_chain[ch].clear();
        if (c->d_chain[ch]) { HIP_TRY(hipFree(c->d_chain[ch])); c->d_chain[ch] = nullptr; }
        if (!fused) continue;
        for (size_t i = 0; i < chains[ch].size(); ++i) {
            chains[ch][i].wstream = c->d_arena + ch_w[ch][i];
            chains[ch][i].bias = c->d_arena + ch_b[ch][i];
        }
        HIP_TRY(hipMalloc((void**)&c->d_chain[ch], chains[ch].size() * sizeof(ChainDesc)));
        HIP_TRY(hipMemcpy(c->d_chain[ch], chains[ch].data(), chains[ch].size() * sizeof(ChainDesc), hipMemcpyHostToDevice));
        c->h_chain[ch] = chains[ch];
    }
    c->fused = fused;
    for (int blk = 0; blk < 2; ++blk) {
        for (int layer = 0; layer < 2; ++layer) {
            c->bandW[blk][layer] = c->d_arena + o_bandW[blk][layer];
            c->bandB[blk][layer] = c->d_arena + o_bandB[blk][layer];
            c->bandW16[blk][layer] = c->d_arena + o_bandW16[blk][layer];
        }
        c->timeW[blk] = c->d_arena + o_timeW[blk];
        c->timeW16[blk] = c->d_arena + o_timeW16[blk];
        c->timeB[blk] = c->d_arena + o_timeB[blk];
        c->bandFc16[blk] = c->d_arena + o_bandFc16[blk];
        c->bandFcB[blk] = c->d_arena + o_bandFcB[blk];
        c->timeFc16[blk] = c->d_arena + o_timeFc16[blk];
        c->timeFcB[blk] = c->d_arena + o_timeFcB[blk];
    }
    c->committed = true;
    return 0;
}

int bsrnn_load_weights_file(bsrnn_ctx* c, const char* path)
{
    if (!c || !path) return fail(BSRNN_EARG, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(BSRNN_EIO, "cannot open %s", path);
    auto bad = [&](const char* why) { fclose(f); return fail(BSRNN_EIO, "%s: %s", path, why); };
    // Nothing in the file is trusted: every tensor is looked up in the inventory FIRST and must have exactly the
    // inventory's rank and dimensions (not just the element count: a transposed matrix has the same count) before a byte
    // of it is read, so sizes never come from the file; nothing may throw across the C ABI (the LADSPA host would terminate).
    try {
        char magic[8];
        uint32_t nb = 0, nt = 0;
        if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BSRNNW01", 8)) return bad("bad magic");
        if (fread(&nb, 4, 1, f) != 1 || (int)nb != c->K) return bad("band count differs from the context's table");
        for (uint32_t i = 0; i < nb; ++i) {
            uint32_t w;
            if (fread(&w, 4, 1, f) != 1 || (int)w != c->widths[i]) return bad("band table differs from the context's table");
        }
        if (fread(&nt, 4, 1, f) != 1) return bad("truncated");
        if (nt > c->params.size()) return bad("more tensors than the model has parameters");
        std::vector<float> buf;
        std::string key;
        for (uint32_t t = 0; t < nt; ++t) {
            uint32_t kl = 0, nd = 0;
            if (fread(&kl, 4, 1, f) != 1 || kl > 256) return bad("bad key length");
            key.resize(kl);
            if (kl && fread(&key[0], 1, kl, f) != kl) return bad("truncated key");
            auto it = c->index.find(key);
            if (it == c->index.end()) { fclose(f); return fail(BSRNN_ENOKEY, "%s: unexpected key '%s'", path, key.c_str()); }
            const Param& p = c->params[it->second];
            if (fread(&nd, 4, 1, f) != 1 || (int)nd != p.ndim) return bad(("rank of '" + key + "' differs from the model's").c_str());
            uint64_t dims[2] = {0, 0};
            for (uint32_t d = 0; d < nd; ++d)
                if (fread(&dims[d], 8, 1, f) != 1) return bad("truncated dims");
            if (dims[0] != (uint64_t)p.d0 || (nd == 2 && dims[1] != (uint64_t)p.d1))
                return bad(("shape of '" + key + "' differs from the model's").c_str());
            const size_t n = (size_t)p.numel();
            buf.resize(n);
            if (n && fread(buf.data(), 4, n, f) != n) return bad("truncated data");
            int rc = bsrnn_set_param(c, key.c_str(), buf.data(), (int64_t)n);
            if (rc) { fclose(f); return rc; }
        }
    } catch (...) {
        return bad("out of memory or malformed file");
    }
    fclose(f);
    return bsrnn_commit_params(c);
}

// --------------------------------------------------------------------------- I/O signature (the exported ONNX file's)
int bsrnn_io_count(void) { return 4; }
int bsrnn_io_info(const bsrnn_ctx* c, int32_t index, int32_t C, const char** name, int32_t* is_input, int64_t dims[4], int32_t* ndim)
{
    static const char* const kNames[4] = {"x.0", "state.0", "y.0", "new_state.0"};      // infer-streaming.py:74 (torch.onnx.export naming)
    if (!c || index < 0 || index >= 4 || C < 1 || !dims || !ndim) return fail(BSRNN_EARG, "bsrnn_io_info: bad arguments");
    if (name) *name = kNames[index];
    if (is_input) *is_input = index < 2;
    if (index & 1) { dims[0] = 4; dims[1] = 2; dims[2] = (int64_t)C * c->K; dims[3] = HID; *ndim = 4; }
    else { dims[0] = C; dims[1] = F2; dims[2] = dims[3] = 0; *ndim = 2; }
    return 0;
}

// --------------------------------------------------------------------------- model entry points
int bsrnn_forward(bsrnn_ctx* c, const float* x, float* y, float* mask, int32_t C, int32_t T, void* stream)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!x || !y || C < 1 || T < 1) return fail(BSRNN_EARG, "bsrnn_forward: bad arguments (C=%d, T=%d)", C, T);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const size_t M = (size_t)C * T;
    if ((rc = ensure_ws(c, M)) || (rc = ensure_tasks(c, (int)M)) || (rc = ensure_ovl(c, C, T))) return rc;
    if (mask && (rc = ensure_tap(c, M))) return rc;
    auto run = [&]() -> int {
        { StageScope sc(c, ST_LAYOUT, s); launch_to_frame_major(c->tb, x, c->Xf, C, T, s); }
        if (int rc2 = run_model(c, c->Xf, c->Yf, mask ? c->d_tap : nullptr, C, T, nullptr, nullptr, s)) return rc2;
        {
            StageScope sc(c, ST_LAYOUT, s);
            launch_from_frame_major(c->tb, c->Yf, y, C, T, s);
            if (mask) launch_from_frame_major(c->tb, c->d_tap, mask, C, T, s);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if ((rc = run())) return rc;
    return finish_call(c, s, run);
}

int bsrnn_forward_chunk(bsrnn_ctx* c, const float* x, const float* state_in, float* y, float* state_out,
                        int32_t C, int32_t L, void* stream)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!x || !y || !state_in || !state_out || C < 1 || L < 1) return fail(BSRNN_EARG, "bsrnn_forward_chunk: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const size_t M = (size_t)C * L;
    if ((rc = ensure_ws(c, M)) || (rc = ensure_tasks(c, (int)M)) || (rc = ensure_ovl(c, C, L))) return rc;
    if (state_in == state_out && c->range_policy == BSRNN_RANGE_EXACT)
        return fail(BSRNN_EARG, "bsrnn_forward_chunk: state_in and state_out must be different buffers (a call that leaves the fp16 range is run again from state_in)");
    auto run = [&]() -> int {
        { StageScope sc(c, ST_LAYOUT, s); launch_to_frame_major(c->tb, x, c->Xf, C, L, s); }
        if (int rc2 = run_model(c, c->Xf, c->Yf, nullptr, C, L, state_in, state_out, s)) return rc2;
        { StageScope sc(c, ST_LAYOUT, s); launch_from_frame_major(c->tb, c->Yf, y, C, L, s); }
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if ((rc = run())) return rc;
    return finish_call(c, s, run);
}

int bsrnn_forward_recurrent(bsrnn_ctx* c, const float* x, const float* state_in, float* y, float* state_out,
                            int32_t C, void* stream)
{
    return bsrnn_forward_chunk(c, x, state_in, y, state_out, C, 1, stream);
}

int bsrnn_dual_path(bsrnn_ctx* c, const float* z, float* z_out, const float* state_in, float* state_out,
                    int32_t C, int32_t T, void* stream)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!z || !z_out || C < 1 || T < 1) return fail(BSRNN_EARG, "bsrnn_dual_path: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const int M = C * T, K = c->K;
    if ((rc = ensure_ws(c, M))) return rc;
    const size_t nz = (size_t)M * K * HID;
    if (state_in && state_in == state_out && c->range_policy == BSRNN_RANGE_EXACT)
        return fail(BSRNN_EARG, "bsrnn_dual_path: state_in and state_out must be different buffers");
    auto run = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(c->Z0, z, nz * sizeof(float), hipMemcpyDeviceToDevice, s));
    const size_t slab = (size_t)2 * 2 * C * K * HID;
    for (int blk = 0; blk < 2; ++blk) {
        if (band_block_is_small(M, K)) {
            launch_band_block_small(c->Z0, c->Z1, c->bandW16[blk][0], c->bandB[blk][0], c->bandW16[blk][1], c->bandB[blk][1],
                                    c->bandFc16[blk], c->bandFcB[blk], M, K, c->d_range, s);
        } else if (ctx_parts(c)) {                 // the block's fc + residual inside the launches around it (run_stage, kernels.h)
            float* zi = blk ? c->Z1 : c->Z0;
            float* zo = blk ? c->Z0 : c->Z1;
            launch_band_pair(zi, c->HB0, c->HB1, c->bandW16[blk][0], c->bandB[blk][0], c->bandW16[blk][1], c->bandB[blk][1], M, K, c->d_range, s,
                             c->bandFc16[blk], c->bandFcB[blk], c->band_flags);
            launch_time_lstm(zi, zo, c->timeW[blk], c->timeW16[blk], c->timeB[blk], state_in ? state_in + blk * slab : nullptr,
                             state_out ? state_out + blk * slab : nullptr, C, T, K, c->d_range, s, c->timeFc16[blk], c->timeFcB[blk], c->HB1);
            continue;
        } else {
            launch_band_lstm(c->Z0, c->HB0, c->bandW[blk][0], c->bandW16[blk][0], c->bandB[blk][0], M, K, 64, c->d_range, s);
            launch_band_lstm(c->HB0, c->HB1, c->bandW[blk][1], c->bandW16[blk][1], c->bandB[blk][1], M, K, 128, c->d_range, s);
            gemm_slot(c, BLK_FC0 + 2 * blk, c->HB1, 2 * HID, c->Z1, HID, c->Z0, HID, nullptr, 0, nullptr, M * K, EPI_RES, s);
        }
        const bool fused = time_lstm_fuses_fc();
        launch_time_lstm(c->Z1, fused ? c->Z0 : c->H1, c->timeW[blk], c->timeW16[blk], c->timeB[blk], state_in ? state_in + blk * slab : nullptr,
                         state_out ? state_out + blk * slab : nullptr, C, T, K, c->d_range, s, c->timeFc16[blk], c->timeFcB[blk]);
        if (!fused) gemm_slot(c, BLK_FC1 + 2 * blk, c->H1, HID, c->Z0, HID, c->Z1, HID, nullptr, 0, nullptr, M * K, EPI_RES, s);
    }
    HIP_TRY(hipMemcpyAsync(z_out, c->Z0, nz * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return 0;
    };
    if ((rc = run())) return rc;
    return finish_call(c, s, run);
}

// --------------------------------------------------------------------------- training step, part 1: recurrent layers
// nn.LSTM of NormRNNResidual (bsrnn.py:66-72) forward with saves and backward through time (lstm_train.hip); what
// loss.backward() runs for these layers in train.py:97-115.  Operands are the caller's device buffers (torch layouts);
// nothing of the committed inference weights is used.
// Scratch of the training entry points: one grow-only buffer per context.  Calls on a context are serialised (ENTER_CALL) and a
// stream switch waits for the previous stream, so reuse in stream order is safe; growing synchronises the device once.
static float* train_scratch(bsrnn_ctx* c, size_t floats)
{
    if (floats <= c->train_ws_floats) return c->d_train_ws;
    // Grow: the old buffer is RETIRED, not freed.  Kernel nodes of a captured training iteration hold its address; a later, larger
    // clip (its eager warm-up or an eager fall-back clip) must not turn those graphs into writers of freed memory.  A retired
    // buffer stays valid scratch for the graphs that know it (they are replayed one at a time, in stream order, like every call).
    float* fresh = nullptr;
    const size_t want = floats + floats / 4;
    if (hipMalloc((void**)&fresh, want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (c->d_train_ws) c->train_ws_retired.push_back(c->d_train_ws);
    c->d_train_ws = fresh;
    c->train_ws_floats = want;
    return c->d_train_ws;
}

static int train_args_ok(bsrnn_ctx* c, int32_t N, int32_t L, int32_t IN, int32_t ndir, const char* who)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (N < 1 || L < 1 || (IN != HID && IN != 2 * HID) || ndir < 1 || ndir > 2 || (int64_t)N * L > (int64_t)1 << 30)
        return fail(BSRNN_EARG, "%s: need N, L >= 1, IN = 64 | 128, ndir = 1 | 2 (got N=%d L=%d IN=%d ndir=%d)", who, N, L, IN, ndir);
    return 0;
}

int bsrnn_lstm_train_forward(bsrnn_ctx* c, const float* x, const float* w_ih, const float* w_hh, const float* bias, float* h,
                             float* gates, float* cells, int32_t N, int32_t L, int32_t IN, int32_t ndir, void* stream)
{
    int rc = train_args_ok(c, N, L, IN, ndir, "bsrnn_lstm_train_forward");
    if (rc) return rc;
    if (!x || !w_ih || !w_hh || !bias || !h || !gates || !cells) return fail(BSRNN_EARG, "bsrnn_lstm_train_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    launch_lstm_train_forward(x, w_ih, w_hh, bias, h, gates, cells, N, L, IN, ndir, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_lstm_train_backward(bsrnn_ctx* c, const float* x, const float* h, const float* gates, const float* cells, const float* dh,
                              const float* w_ih, const float* w_hh, float* dx, float* dw_ih, float* dw_hh, float* db,
                              int32_t N, int32_t L, int32_t IN, int32_t ndir, void* stream)
{
    int rc = train_args_ok(c, N, L, IN, ndir, "bsrnn_lstm_train_backward");
    if (rc) return rc;
    if (!x || !h || !gates || !cells || !dh || !w_ih || !w_hh || !dw_ih || !dw_hh || !db)
        return fail(BSRNN_EARG, "bsrnn_lstm_train_backward: null argument (only dx may be null)");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    // workspace in stream order: gate gradients of every (sequence, step, direction) + the partial sums of the reductions
    const size_t n_dg = (size_t)N * L * ndir * 256, n_scr = lstm_train_scratch_floats(N, L, IN, ndir);
    float* ws = train_scratch(c, n_dg + n_scr);
    if (!ws) return fail(BSRNN_EHIP, "bsrnn_lstm_train_backward: out of device memory (%zu MB of workspace)", (n_dg + n_scr) * sizeof(float) >> 20);
    launch_lstm_train_backward(x, h, gates, cells, dh, w_ih, w_hh, ws, ws + n_dg, dx, dw_ih, dw_hh, db, N, L, IN, ndir, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_adamw_step(bsrnn_ctx* c, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int32_t step, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!p || !g || !m || !v || n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
        return fail(BSRNN_EARG, "bsrnn_adamw_step: bad arguments (n=%lld step=%d)", (long long)n, step);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    launch_adamw(p, g, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), s);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int adamw_multi(bsrnn_ctx* c, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* sizes,
                       int32_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, float* state,
                       void* stream, const char* who)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!p || !g || !m || !v || !sizes || n_tensors < 1 || (!state && step < 1) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
        return fail(BSRNN_EARG, "%s: bad arguments (n_tensors=%d step=%d)", who, n_tensors, step);
    for (int i = 0; i < n_tensors; ++i)
        if (!p[i] || !g[i] || !m[i] || !v[i] || sizes[i] < 0) return fail(BSRNN_EARG, "%s: tensor %d: null pointer or negative size", who, i);
    for (int i = 0; i < n_tensors; ++i)
        if (sizes[i] >= ((int64_t)1 << 31) - 1024) return fail(BSRNN_EARG, "%s: tensor %d has %lld elements (limit 2^31)", who, i, (long long)sizes[i]);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    double bc1 = 1.0, bc2 = 1.0;
    if (state) launch_adamw_tick(state, beta1, beta2, s);
    else { bc1 = 1.0 - pow((double)beta1, step); bc2 = 1.0 - pow((double)beta2, step); }
    for (int i0 = 0; i0 < n_tensors; i0 += ADAM_GROUP) {
        AdamGroup a;
        a.count = std::min(ADAM_GROUP, n_tensors - i0);
        a.first_block[0] = 0;
        for (int j = 0; j < a.count; ++j) {
            a.p[j] = p[i0 + j]; a.g[j] = g[i0 + j]; a.m[j] = m[i0 + j]; a.v[j] = v[i0 + j];
            a.n[j] = (int)sizes[i0 + j];
            a.first_block[j + 1] = a.first_block[j] + (a.n[j] + 1023) / 1024;
        }
        launch_adamw_group(a, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), s, state);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_adamw_step_multi(bsrnn_ctx* c, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* sizes,
                           int32_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream)
{
    return adamw_multi(c, p, g, m, v, sizes, n_tensors, lr, beta1, beta2, eps, weight_decay, step, nullptr, stream, "bsrnn_adamw_step_multi");
}

int bsrnn_adamw_step_multi_dev(bsrnn_ctx* c, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* sizes,
                               int32_t n_tensors, float* state_dev, float beta1, float beta2, float eps, float weight_decay, void* stream)
{
    if (!state_dev) return fail(BSRNN_EARG, "bsrnn_adamw_step_multi_dev: null optimizer state");
    return adamw_multi(c, p, g, m, v, sizes, n_tensors, 0.f, beta1, beta2, eps, weight_decay, 0, state_dev, stream, "bsrnn_adamw_step_multi_dev");
}

int bsrnn_linear_train_forward(bsrnn_ctx* c, const float* x, int32_t ldx, const float* w, const float* b, float* y, int32_t ldy,
                               int32_t M, int32_t K, int32_t N, int32_t leaky, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!x || !w || !b || !y || M < 1 || K < 1 || N < 1 || ldx < K || ldy < N)
        return fail(BSRNN_EARG, "bsrnn_linear_train_forward: bad arguments (M=%d K=%d N=%d ldx=%d ldy=%d)", M, K, N, ldx, ldy);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    launch_linear_train_forward(x, ldx, w, b, y, ldy, M, K, N, leaky != 0, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_linear_train_backward(bsrnn_ctx* c, const float* x, int32_t ldx, const float* w, const float* y, int32_t ldy,
                                const float* dy, int32_t lddy, float* dx, int32_t lddx, float* dw, float* db,
                                int32_t M, int32_t K, int32_t N, int32_t leaky, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!x || !w || !dy || !dw || !db || (leaky && !y) || M < 1 || K < 1 || N < 1 || ldx < K || lddy < N || (leaky && ldy < N) || (dx && lddx < K))
        return fail(BSRNN_EARG, "bsrnn_linear_train_backward: bad arguments (M=%d K=%d N=%d)", M, K, N);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const size_t n_scr = linear_train_scratch_floats(M, K, N, leaky != 0);
    float* ws = train_scratch(c, n_scr);
    if (!ws) return fail(BSRNN_EHIP, "bsrnn_linear_train_backward: out of device memory (%zu MB of workspace)", n_scr * sizeof(float) >> 20);
    launch_linear_train_backward(x, ldx, w, y, ldy, dy, lddy, dx, lddx, dw, db, ws, M, K, N, leaky != 0, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The same Linear layer of several bands (or any layers that share the row count M) in grouped launches: arrays [n] per field.
int bsrnn_linear_group_train_forward(bsrnn_ctx* c, int32_t n, const float* const* x, const int32_t* ldx, const float* const* w,
                                     const float* const* b, float* const* y, const int32_t* ldy, const int32_t* K, const int32_t* N,
                                     int32_t M, int32_t leaky, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 1 || n > 4096 || !x || !ldx || !w || !b || !y || !ldy || !K || !N || M < 1) return fail(BSRNN_EARG, "bsrnn_linear_group_train_forward: bad arguments");
    std::vector<LinearJob> jobs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !w[i] || !b[i] || !y[i] || K[i] < 1 || N[i] < 1 || ldx[i] < K[i] || ldy[i] < N[i])
            return fail(BSRNN_EARG, "bsrnn_linear_group_train_forward: job %d: bad arguments (K=%d N=%d ldx=%d ldy=%d)", i, K[i], N[i], ldx[i], ldy[i]);
        LinearJob& j = jobs[i];
        memset(&j, 0, sizeof j);
        j.x = x[i]; j.ldx = ldx[i]; j.w = w[i]; j.b = b[i]; j.y = y[i]; j.ldy = ldy[i]; j.K = K[i]; j.N = N[i];
    }
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    launch_linear_group_forward(jobs.data(), n, M, leaky != 0, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_linear_group_train_backward(bsrnn_ctx* c, int32_t n, const float* const* x, const int32_t* ldx, const float* const* w,
                                      const float* const* y, const int32_t* ldy, const float* const* dy, const int32_t* lddy,
                                      float* const* dx, const int32_t* lddx, float* const* dw, float* const* db,
                                      const int32_t* K, const int32_t* N, int32_t M, int32_t leaky, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (n < 1 || n > 4096 || !x || !ldx || !w || !dy || !lddy || !dx || !lddx || !dw || !db || !K || !N || M < 1 || (leaky && (!y || !ldy)))
        return fail(BSRNN_EARG, "bsrnn_linear_group_train_backward: bad arguments");
    std::vector<LinearJob> jobs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !w[i] || !dy[i] || !dw[i] || !db[i] || K[i] < 1 || N[i] < 1 || ldx[i] < K[i] || lddy[i] < N[i] || (dx[i] && lddx[i] < K[i]) ||
            (leaky && (!y[i] || ldy[i] < N[i])))
            return fail(BSRNN_EARG, "bsrnn_linear_group_train_backward: job %d: bad arguments (K=%d N=%d)", i, K[i], N[i]);
        LinearJob& j = jobs[i];
        memset(&j, 0, sizeof j);
        j.x = x[i]; j.ldx = ldx[i]; j.w = w[i]; j.y = leaky ? const_cast<float*>(y[i]) : nullptr; j.ldy = leaky ? ldy[i] : 0;
        j.dy = dy[i]; j.lddy = lddy[i]; j.dx = dx[i]; j.lddx = lddx[i]; j.dw = dw[i]; j.db = db[i]; j.K = K[i]; j.N = N[i];
    }
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const size_t n_scr = linear_group_scratch_floats(jobs.data(), n, M, leaky != 0);
    float* ws = train_scratch(c, n_scr);
    if (!ws) return fail(BSRNN_EHIP, "bsrnn_linear_group_train_backward: out of device memory (%zu MB of workspace)", n_scr * sizeof(float) >> 20);
    launch_linear_group_backward(jobs.data(), n, M, leaky != 0, ws, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------- STFT sandwich
int bsrnn_stft(bsrnn_ctx* c, const float* wave, float* x, int32_t R, int64_t n, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!wave || !x || R < 1 || n <= NFFT / 2) return fail(BSRNN_EARG, "bsrnn_stft: need n > 1024 samples (reflect padding), got %lld", (long long)n);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const int T = 1 + (int)(n / HOPS);
    int rc = ensure_ws(c, (size_t)R * T);
    if (rc) return rc;
    { StageScope sc(c, ST_STFT, s); launch_stft(c->tb, wave, c->Xf, R, n, T, s); }
    { StageScope sc(c, ST_LAYOUT, s); launch_from_frame_major(c->tb, c->Xf, x, R, T, s); }
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_istft(bsrnn_ctx* c, const float* y, float* wave_out, int32_t R, int32_t T, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!y || !wave_out || R < 1 || T < 2) return fail(BSRNN_EARG, "bsrnn_istft: need T >= 2 frames");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    int rc = ensure_ws(c, (size_t)R * T);
    if (rc) return rc;
    { StageScope sc(c, ST_LAYOUT, s); launch_to_frame_major(c->tb, y, c->Yf, R, T, s); }
    {
        StageScope sc(c, ST_ISTFT, s);
        launch_istft(c->tb, c->Yf, wave_out, R, T, s);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_istft_backward(bsrnn_ctx* c, const float* dwave, float* dy, int32_t R, int32_t T, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    if (c->device < 0 || c->zombie) return fail(BSRNN_ESTATE, "context cannot compute (host-only or destroyed)");
    HIP_TRY(hipSetDevice(c->device));
    if (!dwave || !dy || R < 1 || T < 2) return fail(BSRNN_EARG, "bsrnn_istft_backward: need T >= 2 frames");
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    int rc = ensure_ws(c, (size_t)R * T);
    if (rc) return rc;
    float* ws = train_scratch(c, (size_t)R * (T - 1) * HOPS);
    if (!ws) return fail(BSRNN_EHIP, "bsrnn_istft_backward: out of device memory");
    launch_istft_backward(c->tb, dwave, ws, c->Yf, R, T, s);
    launch_from_frame_major(c->tb, c->Yf, dy, R, T, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_separate(bsrnn_ctx* c, const float* wave, float* wave_out, int32_t R, int64_t n, void* stream)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!wave || !wave_out || R < 1 || n <= NFFT / 2) return fail(BSRNN_EARG, "bsrnn_separate: need n > 1024 samples, got %lld", (long long)n);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    const int T = 1 + (int)(n / HOPS);
    if ((rc = ensure_ws(c, (size_t)R * T))) return rc;
    const int64_t out_len = (int64_t)(T - 1) * HOPS;

    // Rows are independent, so the batch is cut into `parts` contiguous row blocks that run the whole
    // stage sequence concurrently on separate streams: the ramps, tails and latency-bound stages of
    // one block (e.g. the time-axis LSTM occupies 192 of 256 CUs) overlap matrix work of the other.
    // Part j starts `lag` stages behind part j-1 so that they sit in different stages.
    // (automatic: one block while the time-axis launch of the whole batch is one round of workgroups - eight sequences each from 1 024 sequences on -,
    //  two from there: 128 / 160 rows 1.75 / 2.26 -> 1.73 / 2.18 ms with one block, 192 / 256 rows 2.59 / 3.43 ms with two against 2.72 / 3.47)
    int parts = c->n_parts > 0 ? c->n_parts : (R >= 128 && ((int64_t)R * c->K + time_lstm_seqs(R * c->K) - 1) / time_lstm_seqs(R * c->K) > device_cus() ? 2 : 1);
    if (R < 2 * parts || (int64_t)R * T < 2048) parts = 1;
    if (parts > 1 && (rc = ensure_streams(c, parts))) return rc;
    Part pt[MAX_PARTS];
    {   // task tables of every row block of this call, made before any launch: one call never evicts a table it needs itself
        int ms[MAX_PARTS];
        for (int j = 0; j < parts; ++j) ms[j] = ((int)((int64_t)R * (j + 1) / parts) - (int)((int64_t)R * j / parts)) * T;
        if ((rc = ensure_tasks(c, ms, parts))) return rc;
        if (parts == 1 && (rc = ensure_ovl(c, R, T))) return rc;
    }
    for (int j = 0; j < parts; ++j) {
        const int r0 = (int)((int64_t)R * j / parts), r1 = (int)((int64_t)R * (j + 1) / parts);
        pt[j] = make_part(c, r0, r1 - r0, T, parts > 1 ? c->aux[j] : s, j);
        if (j && pt[j].band_flags < pt[j - 1].band_flags + 2 * (((size_t)pt[j - 1].C * T + 15) / 16))
            return fail(BSRNN_ESTATE, "row blocks %d and %d would share a hand-over flag pair (internal error)", j - 1, j);
        pt[j].wave = wave + (size_t)r0 * n; pt[j].n = n;
        pt[j].wave_out = wave_out + (size_t)r0 * out_len;
    }
    auto run = [&]() -> int {
        if (parts > 1) {
            HIP_TRY(hipEventRecord(c->ev_fork, s));
            for (int j = 0; j < parts; ++j) HIP_TRY(hipStreamWaitEvent(c->aux[j], c->ev_fork, 0));
        }
        const bsrnn_ctx::OvlTable* tb = parts == 1 ? ovl_table(c, R, T, s) : nullptr;
        if (tb) run_overlapped(c, pt[0], tb, MS_STFT, MS_ISTFT);      // the dual path overlapped on the context's auxiliary stream
        const int lag = c->part_lag;
        for (int step = 0; !tb && step < MS_COUNT + lag * (parts - 1); ++step)
            for (int j = 0; j < parts; ++j) {
                const int st = step - lag * j;
                if (st >= 0 && st < MS_COUNT) run_stage(c, pt[j], st);
            }
        if (parts > 1)
            for (int j = 0; j < parts; ++j) {
                HIP_TRY(hipEventRecord(c->ev_join[j], c->aux[j]));
                HIP_TRY(hipStreamWaitEvent(s, c->ev_join[j], 0));
            }
        if (c->stage_error) { c->stage_error = false; return fail(BSRNN_ESTATE, "no task table for a row block of this call (internal error)"); }
        HIP_TRY(hipGetLastError());
        return 0;
    };
    if ((rc = run())) return rc;
    return finish_call(c, s, run);
}

// --------------------------------------------------------------------------- validation metrics
// m_dataset.py:182-226 (`infer` + `train_infer` without the discriminator) and infer.py:44-47.
int bsrnn_evaluate(bsrnn_ctx* c, const float* mix, const float* speech, int32_t R, int64_t n, float* est_out,
                   double* metrics, void* stream)
{
    if (!metrics || !speech) return fail(BSRNN_EARG, "bsrnn_evaluate: null argument");
    float* est = est_out;
    double* d_part = nullptr;
    float* d_alpha = nullptr;
    const int T = 1 + (int)(n / HOPS);
    const int64_t n_est = (int64_t)(T - 1) * HOPS;
    int rc = 0;
    auto cleanup = [&](int code) {
        if (!est_out && est) (void)hipFree(est);
        if (d_part) (void)hipFree(d_part);
        if (d_alpha) (void)hipFree(d_alpha);
        return code;
    };
    if ((rc = check_ready(c))) return rc;
    if (!mix || R < 1 || n <= NFFT / 2) return fail(BSRNN_EARG, "bsrnn_evaluate: need n > 1024 samples, got %lld", (long long)n);
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    if (!est_out && hipMalloc((void**)&est, (size_t)R * n_est * sizeof(float)) != hipSuccess) {
        est = nullptr;
        return fail(BSRNN_EHIP, "bsrnn_evaluate: out of device memory");
    }
    auto run = [&]() -> int {
    // x_time and the estimate's spectrum (left in Yf, frame-major)                               m_dataset.py:186-195
    if ((rc = bsrnn_separate(c, mix, est, R, n, stream))) return rc;
    // waveform_speech_freq: the clean signal through the same analysis (Xf is free once the mask launch ran)   :196
    launch_stft(c->tb, speech, c->Xf, R, n, T, s);

    const int chunks = metric_time_chunks(n_est), FB = 512, IB = 256;
    const size_t n_time = (size_t)R * chunks * METRIC_TIME_Q, n_si = (size_t)R * chunks * 2, n_freq = (size_t)FB * 2, n_in = IB;
    const size_t n_part = n_time + n_si + n_freq + n_in;
    if ((!d_part && hipMalloc((void**)&d_part, n_part * sizeof(double)) != hipSuccess) || (!d_alpha && hipMalloc((void**)&d_alpha, R * sizeof(float)) != hipSuccess))
        return fail(BSRNN_EHIP, "bsrnn_evaluate: out of device memory");
    double *p_time = d_part, *p_si = p_time + n_time, *p_freq = p_si + n_si, *p_in = p_freq + n_freq;
    launch_metric_time(est, speech, mix, R, n_est, n, p_time, s);
    launch_metric_freq(c->tb, c->Yf, c->Xf, R * T, p_freq, FB, s);
    launch_metric_input_sdr(speech, mix, R, n, p_in, IB, s);
    std::vector<double> h(n_part);
    if (hipMemcpyAsync(h.data(), d_part, (n_time) * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(BSRNN_EHIP, "bsrnn_evaluate: %s", hipGetErrorString(hipGetLastError()));
    std::vector<double> q((size_t)R * METRIC_TIME_Q, 0.0);
    for (int r = 0; r < R; ++r)
        for (int ch = 0; ch < chunks; ++ch)
            for (int i = 0; i < METRIC_TIME_Q; ++i) q[(size_t)r * METRIC_TIME_Q + i] += h[((size_t)r * chunks + ch) * METRIC_TIME_Q + i];
    // SI-SDR: alpha = (<x, s> + eps) / (<s, s> + eps) in fp32, eps = float32 machine epsilon
    const float eps = 1.1920928955078125e-07f;
    std::vector<float> alpha(R);
    for (int r = 0; r < R; ++r) alpha[r] = ((float)q[(size_t)r * METRIC_TIME_Q + 2] + eps) / ((float)q[(size_t)r * METRIC_TIME_Q + 0] + eps);
    if (hipMemcpyAsync(d_alpha, alpha.data(), R * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)
        return fail(BSRNN_EHIP, "bsrnn_evaluate: %s", hipGetErrorString(hipGetLastError()));
    launch_metric_sisdr(est, speech, d_alpha, R, n_est, n, p_si, s);
    if (hipMemcpyAsync(h.data() + n_time, p_si, (n_si + n_freq + n_in) * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail(BSRNN_EHIP, "bsrnn_evaluate: %s", hipGetErrorString(hipGetLastError()));

    double sdr = 0, sisdr = 0, l1_time = 0, m2 = 0, md2 = 0;
    for (int r = 0; r < R; ++r) {
        const double* qr = &q[(size_t)r * METRIC_TIME_Q];
        sdr += 10.0 * log10((qr[0] + 1e-9) / (qr[1] + 1e-9));                                   // m_dataset.py:214-217
        double ts2 = 0, nz2 = 0;
        for (int ch = 0; ch < chunks; ++ch) { ts2 += h[n_time + ((size_t)r * chunks + ch) * 2]; nz2 += h[n_time + ((size_t)r * chunks + ch) * 2 + 1]; }
        sisdr += 10.0 * log10((ts2 + (double)eps) / (nz2 + (double)eps));
        l1_time += qr[4]; m2 += qr[5]; md2 += qr[6];
    }
    double l1_re = 0, l1_im = 0, in_sdr = 0;
    for (int b = 0; b < FB; ++b) { l1_re += h[n_time + n_si + 2 * b]; l1_im += h[n_time + n_si + 2 * b + 1]; }
    for (int b = 0; b < IB; ++b) in_sdr += h[n_time + n_si + n_freq + b];
    l1_time /= (double)R * (double)n_est;                                                      // L1Loss(reduction='mean'), train.py:54
    l1_re /= (double)R * NBINS * T;
    l1_im /= (double)R * NBINS * T;
    metrics[BSRNN_M_LOSS] = l1_time + l1_re + l1_im;                                           // m_dataset.py:211-213
    metrics[BSRNN_M_SDR] = sdr / R;
    metrics[BSRNN_M_INPUT_SDR] = in_sdr / (double)n;
    metrics[BSRNN_M_SISDR] = sisdr / R;
    metrics[BSRNN_M_L1_TIME] = l1_time;
    metrics[BSRNN_M_L1_RE] = l1_re;
    metrics[BSRNN_M_L1_IM] = l1_im;
    metrics[BSRNN_M_SEPARATION_DB] = 10.0 * log(m2 / md2);                                     // natural log, infer.py:47
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(BSRNN_EHIP, "bsrnn_evaluate: %s", hipGetErrorString(e));
    };
    rc = run();
    // Synchronous entry point: every kernel of the call has finished, so the range guard of the fp16x2 kernels is final.
    // If an operand left the fp16 range, the numbers above are saturated: run the call again on the exact-fp32 kernels of
    // this library (same weights, no range limit) instead of returning them.
    if (rc == 0 && c->h_range && *(volatile int*)c->h_range && !force_f32()) {
        *(volatile int*)c->h_range = 0;
        set_force_f32(true);
        rc = run();
        set_force_f32(false);
    }
    return cleanup(rc);
}


// --------------------------------------------------------------------------- streaming
// Device layout of a stream object: two carry sets [buf | prev | state] (see bsrnn_stream), then the per-step scratch [X | Y | chunk | out].
static size_t stream_carry_floats(const bsrnn_ctx* c, int C) { return (size_t)C * NFFT * 2 + (size_t)4 * 2 * C * c->K * HID; }
static size_t stream_total_floats(const bsrnn_ctx* c, int C)
{
    return 2 * stream_carry_floats(c, C) + (size_t)C * ((size_t)c->LDP * 2 + HOPS * 2) + 4;
}

int bsrnn_stream_create(bsrnn_ctx* c, int32_t C, bsrnn_stream** out)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!out || C < 1) return fail(BSRNN_EARG, "bsrnn_stream_create: bad arguments");
    CallGuard guard_(c);
    if (!guard_.ok) return guard_.refuse();
    bsrnn_stream* st = new bsrnn_stream();
    st->ctx = c; st->C = C;
    const size_t nstate = (size_t)4 * 2 * C * c->K * HID;
    const size_t total = stream_total_floats(c, C);
    float* p = nullptr;
    ++g_dbg[DBG_ALLOC];
    hipError_t e = hipMalloc((void**)&p, total * sizeof(float));
    if (e != hipSuccess) { delete st; return fail(BSRNN_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
    st->base = p;
    for (int k = 0; k < 2; ++k) {
        st->buf[k] = p; p += (size_t)C * NFFT;
        st->prev[k] = p; p += (size_t)C * NFFT;
        st->state[k] = p; p += nstate;
    }
    st->X = p; p += (size_t)C * c->LDP;
    st->Y = p; p += (size_t)C * c->LDP;
    st->chunk = p; p += (size_t)C * HOPS;
    st->out = p; p += (size_t)C * HOPS;
    // The model part of a step as plain launches (default) or as one hipGraph per carry parity (BSRNN_STREAM_GRAPH=1).  Measured from C
    // (tools/stream_cloop.cpp, profiles/r04_stream_kernels.txt): the graph launch of the 14 dependent kernel nodes costs the host MORE than
    // 14 plain launches (78 vs 68 us inside the call) and the chunk 143.2 vs 139.5 us - the gaps between dependent kernels are the same
    // inside a replayed graph - so the cheaper form is the default; the graph path stays for A/B (BSRNN_NO_GRAPH is accepted and means the default).
    st->use_graph = getenv("BSRNN_STREAM_GRAPH") != nullptr && getenv("BSRNN_NO_GRAPH") == nullptr;
    if (hipHostMalloc((void**)&st->h_in, (size_t)C * HOPS * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&st->h_out, (size_t)C * HOPS * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(st->base); delete st; return fail(BSRNN_EHIP, "hipHostMalloc failed");
    }
    e = hipMemset(st->base, 0, total * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(st->base); delete st; return fail(BSRNN_EHIP, "hipMemset: %s", hipGetErrorString(e)); }
    rc = ensure_ws(c, C);
    if (!rc) rc = ensure_tasks(c, C);
    if (rc) { (void)hipHostFree(st->h_in); (void)hipHostFree(st->h_out); (void)hipFree(st->base); delete st; return rc; }
    ++c->live_streams;
    // Everything a first step would otherwise do on the caller's (audio) thread happens here, as the reference does in its constructor
    // (speech-ladspa-onnx.cpp:55-120: session, FFT plans, state): one throw-away step per carry parity through the host-buffer entry
    // point - it loads every kernel's code object, captures and instantiates both parity graphs and touches the pinned staging
    // buffers - then the carry sets are zeroed again.  bsrnn_stream_step / _step_host allocate, capture and instantiate nothing
    // afterwards unless the context's workspace or weights change under the stream (generation counter).
    {
        std::vector<float> zero((size_t)C * HOPS, 0.f), sink((size_t)C * HOPS);
        for (int k = 0; k < 2 && !rc; ++k) rc = bsrnn_stream_step_host(st, zero.data(), sink.data(), 1.0f);
        if (!rc && hipMemset(st->base, 0, total * sizeof(float)) != hipSuccess) rc = fail(BSRNN_EHIP, "hipMemset failed");
        if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(BSRNN_EHIP, "hipDeviceSynchronize failed");
        st->cur = 0;
        if (rc) { bsrnn_stream_destroy(st); return rc; }
    }
    *out = st;
    return 0;
}

static void stream_drop_graph(bsrnn_stream* st)
{
    for (int k = 0; k < 2; ++k) {
        if (st->exec[k]) { (void)hipGraphExecDestroy(st->exec[k]); st->exec[k] = nullptr; }
        if (st->graph[k]) { (void)hipGraphDestroy(st->graph[k]); st->graph[k] = nullptr; }
    }
}

void bsrnn_stream_destroy(bsrnn_stream* st)
{
    if (!st) return;
    bsrnn_ctx* c = st->ctx;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    stream_drop_graph(st);
    if (st->cap) (void)hipStreamDestroy(st->cap);
    if (st->h_in) (void)hipHostFree(st->h_in);
    if (st->h_out) (void)hipHostFree(st->h_out);
    (void)hipFree(st->base);
    delete st;
    if (--c->live_streams == 0 && c->zombie) destroy_now(c);     // bsrnn_destroy() came first: the context goes with its last stream
}

int bsrnn_stream_reset(bsrnn_stream* st, void* stream)
{
    if (!st) return fail(BSRNN_EARG, "null stream");
    HIP_TRY(hipSetDevice(st->ctx->device));
    HIP_TRY(hipMemsetAsync(st->base, 0, stream_total_floats(st->ctx, st->C) * sizeof(float), (hipStream_t)stream));
    st->cur = 0;
    return 0;
}

// One step from carry set p = st->cur into set 1 - p: chunk -> out (device pointers of the caller or the object's own buffers).
// The model part is a graph replay when possible.  Does NOT flip st->cur: the caller does, once the step is known to be good.
static int stream_step_run(bsrnn_stream* st, const float* chunk, float* out, float mix, hipStream_t s)
{
    bsrnn_ctx* c = st->ctx;
    const int p = st->cur, q = p ^ 1;
    int rc;
    { StageScope sc(c, ST_STREAM_DSP, s); launch_stream_analysis(c->tb, st->buf[p], st->buf[q], chunk, st->X, st->C, s); }
    if (st->use_graph && c->prof == 0 && !force_f32()) {
        // The captured launches hold the context's workspace and weight-arena pointers.  A larger call on the context (workspace
        // regrown) or a re-commit of the parameters (arena rebuilt) since the capture changes ctx->gen: capture again
        // instead of replaying launches that point into freed memory.
        if ((st->exec[0] || st->exec[1]) && st->gen != c->gen) {
            HIP_TRY(hipDeviceSynchronize());
            stream_drop_graph(st);
        }
        if (!st->exec[p]) {
            if (!st->cap) { ++g_dbg[DBG_ALLOC]; HIP_TRY(hipStreamCreateWithFlags(&st->cap, hipStreamNonBlocking)); }
            ++g_dbg[DBG_CAPTURE];
            HIP_TRY(hipStreamBeginCapture(st->cap, hipStreamCaptureModeThreadLocal));
            rc = run_model(c, st->X, st->Y, nullptr, st->C, 1, st->state[p], st->state[q], st->cap);
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamEndCapture(st->cap, &g);           // (always ended, whatever run_model said: the stream must leave capture mode)
            if (rc || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);                       // a failed capture leaves nothing behind
                (void)hipGetLastError();
                if (rc) return rc;
                return fail(BSRNN_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            }
            ++g_dbg[DBG_INSTANTIATE];
            e = hipGraphInstantiate(&st->exec[p], g, nullptr, nullptr, 0);
            if (e != hipSuccess) {
                (void)hipGraphDestroy(g);
                st->exec[p] = nullptr;
                return fail(BSRNN_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
            }
            st->graph[p] = g;
            st->gen = c->gen;
        }
        ++g_dbg[DBG_GRAPH_LAUNCH];
        HIP_TRY(hipGraphLaunch(st->exec[p], s));
    } else if ((rc = run_model(c, st->X, st->Y, nullptr, st->C, 1, st->state[p], st->state[q], s))) {
        return rc;
    }
    { StageScope sc(c, ST_STREAM_DSP, s); launch_stream_synthesis(c->tb, st->Y, st->X, mix, st->prev[p], st->prev[q], out, st->C, s); }
    HIP_TRY(hipGetLastError());
    return 0;
}

int bsrnn_stream_step(bsrnn_stream* st, const float* chunk, float* out, float mix, void* stream)
{
    if (!st || !chunk || !out) return fail(BSRNN_EARG, "bsrnn_stream_step: null argument");
    bsrnn_ctx* c = st->ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    ENTER_CALL(c, s);
    if ((rc = ensure_ws(c, st->C)) || (rc = ensure_tasks(c, st->C))) return rc;
    const float* src = chunk;
    if (chunk == out && c->range_policy == BSRNN_RANGE_EXACT) {
        // in place: a re-run (range policy) must still see the input, so it is kept aside first
        HIP_TRY(hipMemcpyAsync(st->chunk, chunk, (size_t)st->C * HOPS * sizeof(float), hipMemcpyDeviceToDevice, s));
        src = st->chunk;
    }
    if ((rc = stream_step_run(st, src, out, mix, s))) return rc;
    // default range policy: wait, look at the guard, and if an operand left the fp16 range run the step again on the exact-fp32
    // kernels from the same (untouched) carry set
    if ((rc = finish_call(c, s, [&]() -> int { return stream_step_run(st, src, out, mix, s); }))) return rc;
    st->cur ^= 1;
    return 0;
}

int bsrnn_stream_step_host(bsrnn_stream* st, const float* chunk_host, float* out_host, float mix)
{
    if (!st || !chunk_host || !out_host) return fail(BSRNN_EARG, "bsrnn_stream_step_host: null argument");
    bsrnn_ctx* c = st->ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    ENTER_CALL(c, (hipStream_t) nullptr);
    const size_t nb = (size_t)st->C * HOPS * sizeof(float);
    memcpy(st->h_in, chunk_host, nb);
    HIP_TRY(hipMemcpyAsync(st->chunk, st->h_in, nb, hipMemcpyHostToDevice, nullptr));
    // This entry point waits for its own kernels anyway, so it repairs a range violation whatever the policy says
    const int keep = c->range_policy;
    c->range_policy = BSRNN_RANGE_EXACT;
    rc = bsrnn_stream_step(st, st->chunk, st->out, mix, nullptr);
    c->range_policy = keep;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(st->h_out, st->out, nb, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(out_host, st->h_out, nb);
    return 0;
}

int bsrnn_stream_get_state(bsrnn_stream* st, float* state_host)
{
    if (!st || !state_host) return fail(BSRNN_EARG, "null argument");
    HIP_TRY(hipSetDevice(st->ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t nstate = (size_t)4 * 2 * st->C * st->ctx->K * HID;
    HIP_TRY(hipMemcpy(state_host, st->state[st->cur], nstate * sizeof(float), hipMemcpyDeviceToHost));
    return check_range(st->ctx);          // steps made under the 'deferred' policy report a range violation here (or at the next call / bsrnn_sync)
}

// --------------------------------------------------------------------------- measurement
int bsrnn_set_profiling(bsrnn_ctx* c, int32_t on)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (on && c->pool.empty()) {
        c->pool.resize(8192);
        // timing-only events: without the system-scope release fence a record costs the stream ~1 us instead of ~10
        for (auto& r : c->pool) {
            HIP_TRY(hipEventCreateWithFlags(&r.a, hipEventDisableSystemFence));
            HIP_TRY(hipEventCreateWithFlags(&r.b, hipEventDisableSystemFence));
        }
    }
    c->prof = on < 0 ? 0xffffffffu : (unsigned)on;   // bit i enables stage i; negative = all stages
    return 0;
}
int bsrnn_stage_count(void) { return NSTAGE; }
const char* bsrnn_stage_name(int32_t i) { return (i >= 0 && i < NSTAGE) ? kStageNames[i] : ""; }

int bsrnn_stage_times(bsrnn_ctx* c, double* ms_out, int64_t* n_out, int32_t reset)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < c->pool_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->pool[i].a, c->pool[i].b) == hipSuccess) {
            c->acc_ms[c->pool[i].stage] += ms;
            c->acc_n[c->pool[i].stage] += 1;
        }
    }
    c->pool_used = 0;
    for (int i = 0; i < NSTAGE; ++i) {
        if (ms_out) ms_out[i] = c->acc_ms[i];
        if (n_out) n_out[i] = c->acc_n[i];
    }
    if (reset) { memset(c->acc_ms, 0, sizeof c->acc_ms); memset(c->acc_n, 0, sizeof c->acc_n); }
    return 0;
}

// --------------------------------------------------------------------------- device memory helpers
int bsrnn_dev_alloc(bsrnn_ctx* c, int64_t nbytes, void** out)
{
    if (!c || !out || nbytes < 0) return fail(BSRNN_EARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(out, (size_t)nbytes));
    return 0;
}
int bsrnn_dev_free(bsrnn_ctx* c, void* p)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(p));
    return 0;
}
int bsrnn_copy_h2d(bsrnn_ctx* c, void* dst, const void* src, int64_t nbytes)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(dst, src, (size_t)nbytes, hipMemcpyHostToDevice));
    return 0;
}
int bsrnn_copy_d2h(bsrnn_ctx* c, void* dst, const void* src, int64_t nbytes)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(dst, src, (size_t)nbytes, hipMemcpyDeviceToHost));
    return 0;
}
int bsrnn_sync(bsrnn_ctx* c, void* stream)
{
    if (!c) return fail(BSRNN_EARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (int rcj = ovl_join_host(c)) return rcj;
    return check_range(c);
}

}  // extern "C"
