// STFT / iSTFT / layout kernels of the callers' sandwich (infer.py:29-37, m_dataset.py:187-195,
// infer-streaming.py:116-145, speech-ladspa-onnx.cpp:191-261).
//
// One 256-thread workgroup transforms one frame.  The real 2048-point transform is done as a
// complex 1024-point radix-4 Stockham FFT in LDS (5 passes, one radix-4 butterfly per thread
// per pass) plus the real-FFT split/merge step; twiddles and windows come from tables built
// in double precision on the host.  These stages are HBM-bound (4 KiB in, 8.2 KiB out per
// frame); their loads and stores are coalesced along the frame.
#include "kernels.h"
#include <type_traits>

#include <cstdlib>

namespace bsrnn {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// In-LDS complex FFT of 1024 points, Stockham autosort radix-4.  z0 holds the input, the
// result ends in the returned buffer.  INV = true computes the unnormalised inverse.
// Twiddles of the four non-trivial passes, fetched once per thread BEFORE the passes start (they
// depend only on the thread index): a global load inside each pass would put an L2 round trip into
// the dependent chain of every pass.
struct Twiddles { float2 t[4][3]; };
template <bool INV>
__device__ __forceinline__ Twiddles load_twiddles(const float2* __restrict__ tw, int tid)
{
    Twiddles r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = 4 << (2 * q);                 // 4, 16, 64, 256
        const int k = tid & (p - 1);
        const int step = 256 / p;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float2 t = tw[(m + 1) * k * step];
            if (INV) t = cconj(t);
            r.t[q][m] = t;
        }
    }
    return r;
}

template <bool INV>
__device__ __forceinline__ float2* fft1024(float2* z0, float2* z1, const Twiddles& twd, int tid)
{
    float2* src = z0;
    float2* dst = z1;
#ifdef FFT_ABL_NOPASSES
    __syncthreads();
    return src;
#endif
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int p = 1 << (2 * q);
        const int k = tid & (p - 1);
        const int jo = ((tid - k) << 2) + k;
        float2 u0 = src[tid], u1 = src[tid + 256], u2 = src[tid + 512], u3 = src[tid + 768];
        if (q > 0) {
            u1 = cmul(u1, twd.t[q - 1][0]); u2 = cmul(u2, twd.t[q - 1][1]); u3 = cmul(u3, twd.t[q - 1][2]);
        }
        const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), d = csub(u1, u3);
        const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);   // (+i or -i) * d
        dst[jo] = cadd(v0, v2);
        dst[jo + p] = cadd(v1, v3);
        dst[jo + 2 * p] = csub(v0, v2);
        dst[jo + 3 * p] = csub(v1, v3);
        __syncthreads();
        float2* tmp = src; src = dst; dst = tmp;
    }
    return src;
}

// Per-thread slice of the real-FFT split/merge step: bins k = tid + 256 i.  The column map and the
// 2048-point twiddles are fetched up front (before the FFT passes) so that the bin loops below contain no
// dependent global load -> global access chains.
struct SplitCtx {
    int col[5];          // column of bin tid + 256 i   (i = 4 only for tid == 0: bin 1024)
    int colr[4];         // column of bin 1024 - (tid + 256 i)
    float2 tw[5];
};
__device__ __forceinline__ SplitCtx load_split(const FftTables& tb, int tid, bool want_reverse)
{
    SplitCtx c;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = i < 4 ? tid + 256 * i : 1024;          // fifth slot: bin 1024, computed and stored by every thread (see rfft_split_store)
        c.col[i] = tb.colmap ? tb.colmap[k] : 2 * k;
        c.tw[i] = tb.tw2048[k];
        if (i < 4) c.colr[i] = want_reverse ? (tb.colmap ? tb.colmap[1024 - k] : 2 * (1024 - k)) : 0;
    }
    return c;
}

// real spectrum X[0..1024] from Z = FFT1024(x[2n] + i x[2n+1]); writes interleaved re/im
__device__ __forceinline__ void rfft_split_store(const float2* Z, const SplitCtx& sc, float* __restrict__ out, int tid)
{
#pragma unroll
    // No branch around a store: with the fifth (Nyquist) store conditional the compiler cannot count the stores in flight and
    // waits for vmcnt(0) - the previous frame's stores included - in front of the next frame's samples.  Every thread stores
    // bin 1024 instead (the same value to the same address).
    for (int i = 0; i < 5; ++i) {
        const int k = i < 4 ? tid + 256 * i : 1024;
        const float2 zk = Z[k & 1023];
        const float2 zc = cconj(Z[(1024 - k) & 1023]);
        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
        const float2 dd = csub(zk, zc);                               // (zk - zc) / (2i)
        const float2 o = make_float2(0.5f * dd.y, -0.5f * dd.x);
        float2 x = cadd(e, cmul(sc.tw[i], o));
        if (k == 0 || k == 1024) x.y = 0.f;                           // exactly real for real input
        *reinterpret_cast<float2*>(out + sc.col[i]) = x;
    }
}

// Z[k] = E[k] + i O[k] for the inverse; imaginary parts of DC / Nyquist are ignored like c2r does.
// `lin` = true: Y is a plain interleaved [2050] row (LDS copy), else columns come from sc.
struct MergeRegs { float2 xk[4], xc[4]; };
template <bool LIN>
__device__ __forceinline__ void irfft_load(const float* __restrict__ Y, const SplitCtx& sc, MergeRegs& r, int tid)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        r.xk[i] = *reinterpret_cast<const float2*>(Y + (LIN ? 2 * k : sc.col[i]));
        r.xc[i] = *reinterpret_cast<const float2*>(Y + (LIN ? 2 * (1024 - k) : sc.colr[i]));
    }
}
__device__ __forceinline__ void irfft_store(const MergeRegs& r, const SplitCtx& sc, float2* z, int tid)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        float2 a = r.xk[i], b = r.xc[i];
        if (k == 0) { a.y = 0.f; b.y = 0.f; }
        b = cconj(b);
        const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 dd = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 o = cmul(dd, cconj(sc.tw[i]));
        z[k] = make_float2(e.x - o.y, e.y + o.x);                     // e + i*o
    }
}
template <bool LIN>
__device__ __forceinline__ void irfft_merge(const float* __restrict__ Y, const SplitCtx& sc, float2* z, int tid)
{
    MergeRegs r;
    irfft_load<LIN>(Y, sc, r, tid);                                   // all loads first
    irfft_store(r, sc, z, tid);
}

// NOTE: the library is compiled with -fno-slp-vectorize (csrc/Makefile) because of this file.  With the SLP vectorizer the complex butterflies
// become packed-fp32 (v_pk_*) code, and the STFT / iSTFT kernels then returned garbage in whole frames whenever kernels of
// another process or of another stream of this process shared the GPU - always right when alone.  Found with
// tools/row_block_check.py and tests/coresident_check.py; bisected to the FFT passes (not the barriers, the twiddle loads,
// the LDS neighbours or the counted waits) and to the vectorizer (-O1 and -O3 -fno-slp-vectorize are clean, -O2 / -O3 are
// not).  The scalar code is as fast.
//
// Frames per workgroup of the two offline kernels.  All workgroups of a launch cost the same, so a grid slightly larger
// than the chip's resident capacity (CUs x workgroups per CU) runs as two rounds with the second nearly empty: at
// R = 64, T = 126 six frames per workgroup gave 1344 workgroups for 1024 (STFT) / 768 (iSTFT) slots.  The chunk length
// is chosen per launch to minimise rounds x (frames walked per workgroup); small inputs get short chunks (more
// workgroups), `extra` = frames a chunk recomputes (the iSTFT's overlap frame).
static int resident_slots(const void* kernel)
{
    int dev = 0, cus = 256, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    return cus * per_cu;
}
static int frames_per_workgroup(int units, int rows, int slots, int extra)
{
    static const int forced = [] { const char* e = getenv("BSRNN_FFT_RUN"); return e ? atoi(e) : 0; }();      // measurement / debugging knob
    if (forced > 0) return forced;
    int best = 4;
    long best_cost = -1;
    for (int L = 16; L >= 4; --L) {
        const long wgs = (long)((units + L - 1) / L) * rows;
        const long cost = ((wgs + slots - 1) / slots) * (L + extra);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = L; }
    }
    return best;
}

// ------------------------------------------------------------------------------ offline STFT
// One workgroup transforms `sch` consecutive frames of one row.  Frames overlap by half: the thread that owns complex
// samples c + 512, c + 768 of frame t owns c, c + 256 of frame t + 1, so only the new half is loaded per frame
// (requested before the FFT passes of the current frame) and the raw samples stay in registers.
#ifndef FFT_OCC_STFT
#define FFT_OCC_STFT 4            // waves per SIMD (= workgroups per CU) the offline STFT is compiled for (A/B: tools/fft_variants.sh)
#endif
#ifndef FFT_OCC_ISTFT
#define FFT_OCC_ISTFT 4
#endif
template <bool ZERO_PAD>      // ZERO_PAD: samples outside [0, n) are zeros instead of reflections (the adjoint of the iSTFT, below)
__global__ __launch_bounds__(256, FFT_OCC_STFT) void stft_kernel(FftTables tb, const float* __restrict__ wave, float* __restrict__ X,
                                                   int64_t n, int T, int sch)
{
    __shared__ __attribute__((aligned(16))) float2 z0[1024], z1[1024];
#ifdef FFT_EXCLUSIVE_LDS
    // measurement only (tools/coresident_variants.sh): the workgroup claims the CU's whole LDS, so no other kernel's waves share its CU
    __shared__ float lds_pad[(160 * 1024 - 2 * 1024 * 8) / 4 - 64];
    { volatile float* vp = lds_pad; float t_ = vp[threadIdx.x]; asm volatile("" :: "v"(t_)); }
#endif
    const int tid = threadIdx.x;
    const Twiddles twd = load_twiddles<false>(tb.tw1024, tid);
    const SplitCtx spl = load_split(tb, tid, false);
    const int r = blockIdx.y;
    const int t0 = blockIdx.x * sch;
    const int t1 = (t0 + sch < T) ? t0 + sch : T;
    const float* src = wave + (size_t)r * n;
    float2 win[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) win[k] = make_float2(tb.hann[2 * (tid + 256 * k)], tb.hann[2 * (tid + 256 * k) + 1]);
    // padded frame sample i (0..2047) of frame t is original index t*1024 + i - 1024, reflected at both ends
    auto sample2 = [&](int t, int c) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int64_t idx = (int64_t)t * HOPS + 2 * c + e - NFFT / 2;
            if (ZERO_PAD) {
                v[e] = (idx >= 0 && idx < n) ? src[idx] : 0.f;
                continue;
            }
            if (idx < 0) idx = -idx;
            if (idx >= n) idx = 2 * (n - 1) - idx;
            v[e] = src[idx];
        }
        return make_float2(v[0], v[1]);
    };
    float2 raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = sample2(t0, tid + 256 * k);
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z0[tid + 256 * k] = make_float2(raw[k].x * win[k].x, raw[k].y * win[k].y);
        __syncthreads();
        raw[0] = raw[2]; raw[1] = raw[3];
        { const int tn = t + 1 < t1 ? t + 1 : t; raw[2] = sample2(tn, tid + 512); raw[3] = sample2(tn, tid + 768); }      // (no branch: after the last frame a dummy reload)
        const float2* Z = fft1024<false>(z0, z1, twd, tid);
        rfft_split_store(Z, spl, X + ((size_t)r * T + t) * tb.ld, tid);
        __syncthreads();                          // Z (= z1) is overwritten by the next frame's first pass
    }
}

void launch_stft(const FftTables& tb, const float* wave, float* X, int R, int64_t n, int T, hipStream_t s)
{
    static const int slots = resident_slots((const void*)stft_kernel<false>);
    const int sch = frames_per_workgroup(T, R, slots, 0);
    dim3 grid((unsigned)((T + sch - 1) / sch), R);
    hipLaunchKernelGGL(stft_kernel<false>, grid, dim3(256), 0, s, tb, wave, X, n, T, sch);
}

// ------------------------------------------------------------------------------ backward of the offline iSTFT (training step)
// torch.istft (infer.py:35-37 / m_dataset.py:192-195) is linear in the spectrum: wave[n] = (1 / env[n]) sum_t w[j] irfft(Y_t)[j],
// j = n + 1024 - 1024 t, env = sum of squared windows (two frames cover every kept sample).  Its transpose, applied to the
// loss gradient g of the waveform, is an STFT of g / env with ZERO padding (the trimmed ends carry no gradient) whose bins
// are scaled by c_k / 2048, c_0 = c_1024 = 1, else 2 (a one-sided bin stands for itself and its mirror), and the imaginary
// parts of bins 0 and 1024 (which irfft ignores) get no gradient.  dwave [R][(T-1) 1024] -> dY frame-major [R T][ld].
__global__ void istft_bwd_prescale_kernel(FftTables tb, const float* __restrict__ dwave, float* __restrict__ g, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % HOPS);                       // rows are multiples of 1024 long
    const float w0 = tb.hann[j], w1 = tb.hann[j + HOPS];
    g[i] = dwave[i] / (w0 * w0 + w1 * w1);
}
__global__ void istft_bwd_postscale_kernel(FftTables tb, float* __restrict__ X, size_t rows)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * NBINS) return;
    const size_t row = i / NBINS;
    const int k = (int)(i % NBINS);
    float* p = X + row * tb.ld + tb.colmap[k];
    const bool edge = k == 0 || k == NBINS - 1;
    const float sc = (edge ? 1.0f : 2.0f) / (float)NFFT;
    p[0] *= sc;
    p[1] = edge ? 0.f : p[1] * sc;
}
void launch_istft_backward(const FftTables& tb, const float* dwave, float* scratch, float* dY, int R, int T, hipStream_t s)
{
    if (T < 2) return;
    const int64_t n = (int64_t)(T - 1) * HOPS;
    const size_t total = (size_t)R * n;
    hipLaunchKernelGGL(istft_bwd_prescale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tb, dwave, scratch, total);
    static const int slots = resident_slots((const void*)stft_kernel<true>);
    const int sch = frames_per_workgroup(T, R, slots, 0);
    hipLaunchKernelGGL(stft_kernel<true>, dim3((unsigned)((T + sch - 1) / sch), R), dim3(256), 0, s, tb, scratch, dY, n, T, sch);
    const size_t rows = (size_t)R * T;
    hipLaunchKernelGGL(istft_bwd_postscale_kernel, dim3((unsigned)((rows * NBINS + 255) / 256)), dim3(256), 0, s, tb, dY, rows);
}

// ------------------------------------------------------------------------------ offline iSTFT
// One workgroup produces `ich` consecutive output hops of one row: output hop b = first half of synthesis frame
// b + 1 + second half of frame b, divided by the window envelope (torch.istft).  The workgroup walks frames
// b0 .. b0 + ich, keeps the windowed second half of the previous frame in registers (the thread that owns complex
// samples c, c + 256 of a frame's first half also owns c + 512, c + 768 of the second half) and recomputes one
// frame per chunk - no [M][2048] frame buffer in HBM and no separate overlap-add launch (was 132 MB + 15 us).
// The spectrum of frame t + 1 is requested before the FFT passes of frame t.
__global__ __launch_bounds__(256, FFT_OCC_ISTFT) void istft_fused_kernel(FftTables tb, const float* __restrict__ Y, float* __restrict__ out, int T, int ich)
{
    __shared__ __attribute__((aligned(16))) float2 z0[1024], z1[1024];
#ifdef FFT_EXCLUSIVE_LDS
    // measurement only (tools/coresident_variants.sh): the workgroup claims the CU's whole LDS, so no other kernel's waves share its CU
    __shared__ float lds_pad[(160 * 1024 - 2 * 1024 * 8) / 4 - 64];
    { volatile float* vp = lds_pad; float t_ = vp[threadIdx.x]; asm volatile("" :: "v"(t_)); }
#endif
    const int tid = threadIdx.x;
    const Twiddles twd = load_twiddles<true>(tb.tw1024, tid);
    const SplitCtx spl = load_split(tb, tid, true);
    const int r = blockIdx.y;
    const int b0 = blockIdx.x * ich;
    const int b1 = (b0 + ich < T - 1) ? b0 + ich : T - 1;          // output hops [b0, b1) <- frames b0 .. b1
    const size_t len = (size_t)(T - 1) * HOPS;
    const float sc = 1.0f / 1024.0f;
    float2 wlo[2], whi[2], env[2], carry[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        wlo[k] = make_float2(tb.hann[2 * c] * sc, tb.hann[2 * c + 1] * sc);
        whi[k] = make_float2(tb.hann[2 * c + HOPS] * sc, tb.hann[2 * c + 1 + HOPS] * sc);
        env[k] = make_float2(tb.inv_env[2 * c], tb.inv_env[2 * c + 1]);
        carry[k] = make_float2(0.f, 0.f);
    }
    const float* Yr = Y + (size_t)r * T * tb.ld;
    MergeRegs mr;
    irfft_load<false>(Yr + (size_t)b0 * tb.ld, spl, mr, tid);
    // The first frame of the chunk only fills the carry; it is peeled so that the loop body has no branch around its loads and
    // stores (the compiler then counts them and waits for the next spectrum with vmcnt(2) instead of vmcnt(0), stores included).
    auto frame = [&](int t, auto first) {
        irfft_store(mr, spl, z0, tid);
        __syncthreads();
        irfft_load<false>(Yr + (size_t)(t < b1 ? t + 1 : t) * tb.ld, spl, mr, tid);      // (after the last frame: a dummy reload)
        const float2* z = fft1024<true>(z0, z1, twd, tid);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = tid + 256 * k;
            const float2 a = z[c], b = z[c + 512];
            if (!decltype(first)::value) {
                // same operation order as the two-pass version: (frame * 1/1024 * window) summed, then / envelope
                const float2 v = make_float2((a.x * wlo[k].x + carry[k].x) * env[k].x, (a.y * wlo[k].y + carry[k].y) * env[k].y);
                *reinterpret_cast<float2*>(out + (size_t)r * len + (size_t)(t - 1) * HOPS + 2 * c) = v;
            }
            carry[k] = make_float2(b.x * whi[k].x, b.y * whi[k].y);
        }
        __syncthreads();                          // z (= z1) is overwritten by the next frame's first pass
    };
    frame(b0, std::true_type());
    for (int t = b0 + 1; t <= b1; ++t) frame(t, std::false_type());
}

void launch_istft(const FftTables& tb, const float* Y, float* out, int R, int T, hipStream_t s)
{
    if (T < 2) return;
    static const int slots = resident_slots((const void*)istft_fused_kernel);
    const int ich = frames_per_workgroup(T - 1, R, slots, 1);
    dim3 grid((unsigned)((T - 1 + ich - 1) / ich), R);
    hipLaunchKernelGGL(istft_fused_kernel, grid, dim3(256), 0, s, tb, Y, out, T, ich);
}

// ------------------------------------------------------------------------------ [C][2050][T] <-> [C*T][ld]
// The reference boundary is [C][2050][T] (T innermost, bsrnn.py:385); inside the library rows are
// frames and columns follow the band-padded map.  32x32 tiles through LDS, both sides coalesced.
template <bool TO_FRAME_MAJOR>
__global__ __launch_bounds__(256) void layout_kernel(FftTables tb, const float* __restrict__ src, float* __restrict__ dst, int T)
{
    __shared__ float tile[32][33];
    const int c = blockIdx.z;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;      // f = interleaved column 0..2049
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const float* ref = TO_FRAME_MAJOR ? src : nullptr;
    if (TO_FRAME_MAJOR) {
#pragma unroll
        for (int i = 0; i < 32; i += 8) {                      // read [f][t], t contiguous
            const int f = f0 + ty + i, t = t0 + tx;
            if (f < F2 && t < T) tile[ty + i][tx] = ref[((size_t)c * F2 + f) * T + t];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i += 8) {                      // write [t][col(f)], f contiguous
            const int t = t0 + ty + i, f = f0 + tx;
            if (f < F2 && t < T) dst[((size_t)c * T + t) * tb.ld + tb.colmap[f >> 1] + (f & 1)] = tile[tx][ty + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 32; i += 8) {                      // read [t][col(f)]
            const int t = t0 + ty + i, f = f0 + tx;
            if (f < F2 && t < T) tile[ty + i][tx] = src[((size_t)c * T + t) * tb.ld + tb.colmap[f >> 1] + (f & 1)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i += 8) {                      // write [f][t]
            const int f = f0 + ty + i, t = t0 + tx;
            if (f < F2 && t < T) dst[((size_t)c * F2 + f) * T + t] = tile[tx][ty + i];
        }
    }
}

// The same transpose for calls of many frames (T >= 32): a workgroup moves 64 interleaved columns x up to 128 frames.  On the
// [C][2050][T] side the 64 rows of a tile are ONE contiguous range when the tile spans all T frames (T <= 128: the offline sizes),
// read / written with 16-byte accesses where T allows; on the frame-major side a wave moves the 64 consecutive columns of one frame
// (256 bytes) per instruction.  32 KB in and out per workgroup instead of 4 KB: 47-49 -> 3x us per launch at R = 64, T = 126.
constexpr int LF = 64, LT = 128;
template <bool TO_FRAME_MAJOR>
__global__ __launch_bounds__(256) void layout_wide_kernel(FftTables tb, const float* __restrict__ src, float* __restrict__ dst, int T)
{
    __shared__ float tile[LF][LT + 1];                         // [column][frame], + 1: conflict-free in both directions
    const int c = blockIdx.z, f0 = blockIdx.y * LF, t0 = blockIdx.x * LT;
    const int nf = F2 - f0 < LF ? F2 - f0 : LF, nt = T - t0 < LT ? T - t0 : LT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t cm0 = ((size_t)c * F2 + f0) * T + t0;         // first element of the tile on that side (row stride T)
    const bool whole = nt == T;                                // the tile's rows are adjacent in memory
    const float* const cmp = TO_FRAME_MAJOR ? src : dst;       // the [C][2050][T] side
    const bool vec4 = whole && ((nf * T) & 3) == 0 && ((cm0 & 3) == 0) && ((reinterpret_cast<size_t>(cmp) & 15) == 0);
    // frame-major side: lane = column of the tile, one frame per wave instruction
    const int fcol = f0 + lane;
    const bool col_ok = lane < nf;
    const int dcol = col_ok ? tb.colmap[fcol >> 1] + (fcol & 1) : 0;
    if (TO_FRAME_MAJOR) {
        if (vec4) {
            const int n4 = nf * T / 4;
            for (int i = tid; i < n4; i += 256) {
                const float4 v = *reinterpret_cast<const float4*>(src + cm0 + 4 * (size_t)i);
                int f = (4 * i) / T, t = 4 * i - f * T;
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { tile[f][t] = e[k]; if (++t == T) { t = 0; ++f; } }
            }
        } else {
            for (int f = wave; f < nf; f += 4)
                for (int t = lane; t < nt; t += 64) tile[f][t] = src[cm0 + (size_t)f * T + t];
        }
        __syncthreads();
        if (col_ok)
            for (int t = wave; t < nt; t += 4) dst[((size_t)c * T + t0 + t) * tb.ld + dcol] = tile[lane][t];
    } else {
        if (col_ok)
            for (int t = wave; t < nt; t += 4) tile[lane][t] = src[((size_t)c * T + t0 + t) * tb.ld + dcol];
        __syncthreads();
        if (vec4) {
            const int n4 = nf * T / 4;
            for (int i = tid; i < n4; i += 256) {
                int f = (4 * i) / T, t = 4 * i - f * T;
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[k] = tile[f][t]; if (++t == T) { t = 0; ++f; } }
                *reinterpret_cast<float4*>(dst + cm0 + 4 * (size_t)i) = make_float4(e[0], e[1], e[2], e[3]);
            }
        } else {
            for (int f = wave; f < nf; f += 4)
                for (int t = lane; t < nt; t += 64) dst[cm0 + (size_t)f * T + t] = tile[f][t];
        }
    }
}

void launch_to_frame_major(const FftTables& tb, const float* x, float* xf, int C, int T, hipStream_t s)
{
    if (T >= 32) {
        hipLaunchKernelGGL(layout_wide_kernel<true>, dim3((T + LT - 1) / LT, (F2 + LF - 1) / LF, C), dim3(256), 0, s, tb, x, xf, T);
        return;
    }
    dim3 grid((T + 31) / 32, (F2 + 31) / 32, C);
    hipLaunchKernelGGL(layout_kernel<true>, grid, dim3(256), 0, s, tb, x, xf, T);
}
void launch_from_frame_major(const FftTables& tb, const float* yf, float* y, int C, int T, hipStream_t s)
{
    if (T >= 32) {
        hipLaunchKernelGGL(layout_wide_kernel<false>, dim3((T + LT - 1) / LT, (F2 + LF - 1) / LF, C), dim3(256), 0, s, tb, yf, y, T);
        return;
    }
    dim3 grid((T + 31) / 32, (F2 + 31) / 32, C);
    hipLaunchKernelGGL(layout_kernel<false>, grid, dim3(256), 0, s, tb, yf, y, T);
}

// ------------------------------------------------------------------------------ streaming DSP
__global__ __launch_bounds__(256) void stream_analysis_kernel(FftTables tb, const float* __restrict__ buf_in, float* __restrict__ buf,
                                                              const float* __restrict__ chunk, float* __restrict__ X)
{
    __shared__ __attribute__((aligned(16))) float2 z0[1024], z1[1024];
    const int tid = threadIdx.x;
    const Twiddles twd = load_twiddles<false>(tb.tw1024, tid);
    const SplitCtx spl = load_split(tb, tid, false);
    const int c = blockIdx.x;
    float* b = buf + (size_t)c * NFFT;
    const float* ch = chunk + (size_t)c * HOPS;
    // new buffer = [old[1024:2048], chunk]   (infer-streaming.py:116)
    float2 keep[2], fresh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cc = tid + 256 * i;                 // complex index 0..511 of each half
        keep[i] = *reinterpret_cast<const float2*>(buf_in + (size_t)c * NFFT + HOPS + 2 * cc);
        fresh[i] = *reinterpret_cast<const float2*>(ch + 2 * cc);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cc = tid + 256 * i;
        *reinterpret_cast<float2*>(b + 2 * cc) = keep[i];
        *reinterpret_cast<float2*>(b + HOPS + 2 * cc) = fresh[i];
        z0[cc] = make_float2(keep[i].x * tb.hann[2 * cc], keep[i].y * tb.hann[2 * cc + 1]);
        z0[512 + cc] = make_float2(fresh[i].x * tb.hann[HOPS + 2 * cc], fresh[i].y * tb.hann[HOPS + 2 * cc + 1]);
    }
    __syncthreads();
    const float2* Z = fft1024<false>(z0, z1, twd, tid);
    rfft_split_store(Z, spl, X + (size_t)c * tb.ld, tid);
}

__global__ __launch_bounds__(256) void stream_synthesis_kernel(FftTables tb, const float* __restrict__ Y, const float* __restrict__ X,
                                                               const float mix, const float* __restrict__ prev_in, float* __restrict__ prev,
                                                               float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float2 z0[1024], z1[1024];
    __shared__ __attribute__((aligned(16))) float spec[F2 + 2];
    const int tid = threadIdx.x;
    const Twiddles twd = load_twiddles<true>(tb.tw1024, tid);
    const SplitCtx spl = load_split(tb, tid, false);
    const int c = blockIdx.x;
    const float* y = Y + (size_t)c * tb.ld;
    const float* x = X + (size_t)c * tb.ld;
    // wet/dry on the spectrum (speech-ladspa-onnx.cpp:215-226); mix = 1 is the plain model output
    const float dry = mix >= 0.f ? 1.f - mix : 1.f;
    for (int i = tid; i < F2; i += 256) {
        const int col = tb.colmap[i >> 1] + (i & 1);
        spec[i] = (mix == 1.f) ? y[col] : mix * y[col] + dry * x[col];
    }
    __syncthreads();
    irfft_merge<true>(spec, spl, z0, tid);
    __syncthreads();
    const float2* z = fft1024<true>(z0, z1, twd, tid);
    float* pv = prev + (size_t)c * NFFT;
    float* o = out + (size_t)c * HOPS;
    const float sc = 1.0f / 1024.0f;
    // out = (s_now[0:1024] + s_prev[1024:2048]) / (w[0:1024] + w[1024:2048]); prev = s_now
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cc = tid + 256 * i;                 // samples 2cc, 2cc+1 of the first half
        const float2 now = make_float2(z[cc].x * sc, z[cc].y * sc);
        const float2 old = *reinterpret_cast<const float2*>(prev_in + (size_t)c * NFFT + HOPS + 2 * cc);
        *reinterpret_cast<float2*>(o + 2 * cc) =
            make_float2((now.x + old.x) * tb.inv_wsum[2 * cc], (now.y + old.y) * tb.inv_wsum[2 * cc + 1]);
    }
    __syncthreads();
    for (int cc = tid; cc < 1024; cc += 256)
        *reinterpret_cast<float2*>(pv + 2 * cc) = make_float2(z[cc].x * sc, z[cc].y * sc);
}

void launch_stream_analysis(const FftTables& tb, const float* buf_in, float* buf_out, const float* chunk, float* X, int C, hipStream_t s)
{
    hipLaunchKernelGGL(stream_analysis_kernel, dim3(C), dim3(256), 0, s, tb, buf_in, buf_out, chunk, X);
}
void launch_stream_synthesis(const FftTables& tb, const float* Y, const float* X, float mix, const float* prev_in, float* prev_out, float* out,
                             int C, hipStream_t s)
{
    hipLaunchKernelGGL(stream_synthesis_kernel, dim3(C), dim3(256), 0, s, tb, Y, X, mix, prev_in, prev_out, out);
}

}  // namespace bsrnn
