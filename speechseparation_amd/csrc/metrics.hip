// Validation arithmetic of the reference's `train_infer` / `infer.py` report on device-resident signals:
// L1 tri-loss, SDR, the reference's "input SDR", SI-SDR and the "Separation dB" figure
// (m_dataset.py:202-226, infer.py:44-47).  Pure HBM-bound reductions: every kernel reads its operands once
// with 16-byte loads, accumulates the fp32 products in double per thread, reduces wave-wide with DPP shuffles
// and leaves one partial per workgroup; the host adds the (few hundred) partials in a fixed order, so the
// result does not depend on the launch.
#include "kernels.h"

namespace bsrnn {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NQ>
__device__ __forceinline__ void block_reduce_store(double (&q)[NQ], double* dst)
{
    __shared__ double sh[4][NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q[i] += __shfl_down(q[i], off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NQ; ++i) sh[wave][i] = q[i];
    __syncthreads();
    if (threadIdx.x < NQ) dst[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// grid (chunks, R).  est [R][n_est]; speech, mix [R][n_in], only their first n_est samples are used (m_dataset.py:198).
// part [R][chunks][7]: sum s^2, sum (x-s)^2, sum x*s, sum x^2, sum |x-s|, sum m^2, sum (m-x)^2
__global__ __launch_bounds__(256) void metric_time_kernel(const float* __restrict__ est, const float* __restrict__ speech,
                                                          const float* __restrict__ mix, int64_t n_est, int64_t n_in,
                                                          int vec, double* __restrict__ part)
{
    const int r = blockIdx.y;
    const float* x = est + (size_t)r * n_est;
    const float* s = speech + (size_t)r * n_in;
    const float* m = mix + (size_t)r * n_in;
    double q[METRIC_TIME_Q] = {0, 0, 0, 0, 0, 0, 0};
    auto one = [&](float xv, float sv, float mv) {
        const float d = xv - sv, e = mv - xv;          // fp32 differences, as the reference forms them
        q[0] += (double)sv * sv; q[1] += (double)d * d; q[2] += (double)xv * sv; q[3] += (double)xv * xv;
        q[4] += (double)__builtin_fabsf(d); q[5] += (double)mv * mv; q[6] += (double)e * e;
    };
    // vec: every row of the three signals starts 16-byte aligned (the host checks bases and n_in % 4; n_est is
    // a multiple of 1024), otherwise scalar loads
    const int64_t n4 = vec ? n_est / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const v4f xv = *reinterpret_cast<const v4f*>(x + 4 * i);
        const v4f sv = *reinterpret_cast<const v4f*>(s + 4 * i);
        const v4f mv = *reinterpret_cast<const v4f*>(m + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) one(xv[j], sv[j], mv[j]);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_est; i += (int64_t)gridDim.x * 256) one(x[i], s[i], m[i]);
    block_reduce_store<METRIC_TIME_Q>(q, part + ((size_t)r * gridDim.x + blockIdx.x) * METRIC_TIME_Q);
}

// SI-SDR second pass (torchmetrics scale_invariant_signal_distortion_ratio, zero_mean = False): with the row's
// alpha, sum (alpha s)^2 and sum (alpha s - x)^2, the products formed in fp32 like the reference's tensors.
// part [R][chunks][2]
__global__ __launch_bounds__(256) void metric_sisdr_kernel(const float* __restrict__ est, const float* __restrict__ speech,
                                                           const float* __restrict__ alpha, int64_t n_est, int64_t n_in,
                                                           double* __restrict__ part)
{
    const int r = blockIdx.y;
    const float* x = est + (size_t)r * n_est;
    const float* s = speech + (size_t)r * n_in;
    const float a = alpha[r];
    double q[2] = {0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_est; i += (int64_t)gridDim.x * 256) {
        const float ts = a * s[i], nz = ts - x[i];
        q[0] += (double)ts * ts; q[1] += (double)nz * nz;
    }
    block_reduce_store<2>(q, part + ((size_t)r * gridDim.x + blockIdx.x) * 2);
}

// The reference's `sdr2` (m_dataset.py:219-222): its sample tensors still carry the DataLoader's batch dimension
// ([1, rows, n]), so `dim=1` sums over the ROWS, not over time: one ratio per sample position, averaged over
// the n positions.  Kept as written.  part [gridDim.x]
__global__ __launch_bounds__(256) void metric_input_sdr_kernel(const float* __restrict__ speech, const float* __restrict__ mix,
                                                               int R, int64_t n, double* __restrict__ part)
{
    double q[1] = {0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < R; ++r) {
            const float sv = speech[(size_t)r * n + i], d = sv - mix[(size_t)r * n + i];
            a += sv * sv; b += d * d;
        }
        q[0] += 10.0 * log10((double)(a + 1e-9f) / (double)(b + 1e-9f));
    }
    block_reduce_store<1>(q, part + blockIdx.x);
}

// Spectral L1 terms: Yf, Sf frame-major band-padded spectra [M][ld]; bin k at columns colmap[k] (re), +1 (im).
// part [gridDim.x][2]: sum |re diff|, sum |im diff|
__global__ __launch_bounds__(256) void metric_freq_kernel(const float* __restrict__ Yf, const float* __restrict__ Sf,
                                                          const int* __restrict__ colmap, int ld, int M, double* __restrict__ part)
{
    double q[2] = {0, 0};
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const float* y = Yf + (size_t)m * ld;
        const float* s = Sf + (size_t)m * ld;
        for (int k = threadIdx.x; k < NBINS; k += 256) {
            const int c = colmap[k];
            const float2 yv = *reinterpret_cast<const float2*>(y + c), sv = *reinterpret_cast<const float2*>(s + c);
            q[0] += (double)__builtin_fabsf(yv.x - sv.x);
            q[1] += (double)__builtin_fabsf(yv.y - sv.y);
        }
    }
    block_reduce_store<2>(q, part + (size_t)blockIdx.x * 2);
}

}  // namespace

int metric_time_chunks(int64_t n_est) { int64_t c = (n_est + 16383) / 16384; return (int)(c < 1 ? 1 : (c > 256 ? 256 : c)); }

void launch_metric_time(const float* est, const float* speech, const float* mix, int R, int64_t n_est, int64_t n_in,
                        double* part, hipStream_t s)
{
    const int vec = (n_in % 4 == 0) && (((uintptr_t)est | (uintptr_t)speech | (uintptr_t)mix) & 15) == 0;
    hipLaunchKernelGGL(metric_time_kernel, dim3(metric_time_chunks(n_est), R), dim3(256), 0, s, est, speech, mix, n_est, n_in, vec, part);
}
void launch_metric_sisdr(const float* est, const float* speech, const float* alpha, int R, int64_t n_est, int64_t n_in,
                         double* part, hipStream_t s)
{
    hipLaunchKernelGGL(metric_sisdr_kernel, dim3(metric_time_chunks(n_est), R), dim3(256), 0, s, est, speech, alpha, n_est, n_in, part);
}
void launch_metric_input_sdr(const float* speech, const float* mix, int R, int64_t n, double* part, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(metric_input_sdr_kernel, dim3(blocks), dim3(256), 0, s, speech, mix, R, n, part);
}
void launch_metric_freq(const FftTables& tb, const float* Yf, const float* Sf, int M, double* part, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(metric_freq_kernel, dim3(blocks), dim3(256), 0, s, Yf, Sf, tb.colmap, tb.ld, M, part);
}

}  // namespace bsrnn
