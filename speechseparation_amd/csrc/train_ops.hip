// Training step: the plain matrix products around the recurrent kernels (lstm_train.hip) and the per-band Linear layers
// (nn.Linear + LeakyReLU of bandFCs_pre / bandFCs / bandFCs_back / bandFCs_back_post and the fc / fc_in of NormRNNResidual,
// bsrnn.py:333-376, :69, :74) as forward and backward, i.e. what torch.autograd runs for them in train.py:97-115.
// Exact fp32 on v_mfma_f32_16x16x4_f32; first versions (64 x 64 tiles through LDS, one launch per band and layer; 60 TF on
// the 768 x 768 layers at 8 064 rows, tools/train_gemm_bench.hip - a 128 x 64 tile with transposed LDS operands and 16-byte
// LDS reads measured SLOWER, 23 - 41 TF: bank conflicts of the transposing stores): the inference path's grouped / fused
// kernels are the model for the next pass.
//   forward   y = act(x W^T + b)                                   x [M][K], W [N][K] (torch layout), y [M][N]
//   backward  dp = dy * act'(y);  dx = dp W;  dW = dp^T x;  db = column sums of dp
// Weight-gradient reductions over the M rows run in fixed row chunks (about 512 rows, at most LSTM_TRAIN_CHUNKS) that are then
// added in order: gradients are bit-reproducible (no atomics).
#include "kernels.h"

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

// A launch runs a GROUP of products (the same layer of all bands: one launch instead of eleven); the jobs travel by value in
// the kernel arguments, workgroup b belongs to the job with first[j] <= b < first[j + 1].
__device__ __forceinline__ int find_job(const TrainGemmGroup& g, int b)
{
    int j = 0;
    while (j + 1 < g.count && b >= g.first[j + 1]) ++j;
    return j;
}

// C [M][N] (+)= A [M][K] op(B) (+ bias[N]) (LeakyReLU), op(B) = B [K][N] or (TRANS_B) B^T with B [N][K]; row-major with
// leading dimensions.  64 x 64 tile per workgroup, wave w the 16-row strip w, K in slabs of 16 through LDS.
template <bool TRANS_B>
__global__ __launch_bounds__(256) void sgemm_kernel(TrainGemmGroup g, int accumulate, int leaky)
{
    __shared__ float sa[64][17], sb[16][65];
    const int ji = find_job(g, blockIdx.x);
    const TrainGemmJob& jb = g.j[ji];
    const int local = blockIdx.x - g.first[ji];
    const float* __restrict__ A = jb.A; const float* __restrict__ B = jb.B; float* __restrict__ C = jb.C;
    const float* __restrict__ bias = jb.bias;
    const int lda = jb.lda, ldb = jb.ldb, ldc = jb.ldc, M = jb.M, N = jb.N, K = jb.K;
    const int m0 = (local % jb.tiles_x) * 64, n0 = (local / jb.tiles_x) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    v4f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    float ra[4], rb[4];                                       // the next slab, in flight while the current one is multiplied
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int ar = idx >> 4, ac = idx & 15;           // A slab: 64 rows x 16 k
            const int m = m0 + ar, k = k0 + ac;
            ra[i] = (m < M && k < K) ? A[(size_t)m * lda + k] : 0.f;
            if (TRANS_B) {                                    // B^T slab from B [N][K]: k contiguous
                const int bn = idx >> 4, bk = idx & 15;
                const int n = n0 + bn, kk = k0 + bk;
                rb[i] = (n < N && kk < K) ? B[(size_t)n * ldb + kk] : 0.f;
            } else {                                          // B slab: 16 k x 64 columns
                const int br = idx >> 6, bc = idx & 63;
                const int kk = k0 + br, n = n0 + bc;
                rb[i] = (kk < K && n < N) ? B[(size_t)kk * ldb + n] : 0.f;
            }
        }
    };
    if (K > 0) fetch(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            sa[idx >> 4][idx & 15] = ra[i];
            if (TRANS_B) sb[idx & 15][idx >> 4] = rb[i];
            else sb[idx >> 6][idx & 63] = rb[i];
        }
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a = sa[16 * wave + l15][4 * s + q];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sb[4 * s + q][16 * j + l15], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * q + r, n = n0 + 16 * j + l15;
            if (m < M && n < N) {
                float* p = C + (size_t)m * ldc + n;
                float v = acc[j][r];
                if (bias) v += bias[n];
                if (accumulate) v += *p;
                if (leaky) v = v >= 0.f ? v : 0.01f * v;
                *p = v;
            }
        }
}

// Partial sums of C [N1][N2] = A^T B over the row chunk blockIdx.z: A [M][N1] (lda), B [M][N2] (ldb), both row-major.
// `shift`: B's row for A's row m = (n, t) is (n, t + shift), zero where t + shift leaves [0, L) (h_prev of dW_hh); 0: same row.
// TN_KS rows of both operands per barrier pair (16 measured 11 TF on the tall-skinny LSTM weight gradients: barrier-bound).
#ifndef TN_KS
#define TN_KS 32
#endif
__global__ __launch_bounds__(256) void sgemm_tn_partial_kernel(TrainGemmGroup g, int L, int shift)
{
    // a job here: C = part (its chunk records), N = N1, K = N2, rpc rows per chunk, tiles_x x tiles_y tiles per chunk;
    // bias != null: the chunk's record is [N1 x N2 product | N1 column sums of A] (the bias gradient rides along: the A slab
    // is in LDS anyway); the workgroups of the first column tile add them up
    __shared__ float sa[TN_KS][65], sb[TN_KS][65];
    const int ji = find_job(g, blockIdx.x);
    const TrainGemmJob& jb = g.j[ji];
    const int local = blockIdx.x - g.first[ji];
    const float* __restrict__ A = jb.A; const float* __restrict__ B = jb.B; float* __restrict__ part = jb.C;
    const int lda = jb.lda, ldb = jb.ldb, M = jb.M, N1 = jb.N, N2 = jb.K, rows_per_chunk = jb.rpc;
    const bool with_colsum = jb.bias != nullptr;
    const int per_chunk = jb.tiles_x * jb.tiles_y, bz = local / per_chunk, bxy = local % per_chunk;
    const int bx = bxy % jb.tiles_x, by = bxy / jb.tiles_x;
    const int i0 = bx * 64, j0 = by * 64;
    const int r_lo = bz * rows_per_chunk, r_hi = min(M, r_lo + rows_per_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    v4f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    // the next slab's elements travel in registers while the current one is multiplied (global latency behind the MFMAs)
    float ra[TN_KS / 4], rb[TN_KS / 4];
    float csum = 0.f;
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < TN_KS / 4; ++i) {
            const int idx = tid + 256 * i;
            const int rr = idx >> 6, cc = idx & 63;
            const int m = m0 + rr;
            ra[i] = (m < r_hi && i0 + cc < N1) ? A[(size_t)m * lda + i0 + cc] : 0.f;
            float bv = 0.f;
            if (m < r_hi && j0 + cc < N2) {
                const int t = m % L + shift;
                if (t >= 0 && t < L) bv = B[(size_t)(m + shift) * ldb + j0 + cc];
            }
            rb[i] = bv;
        }
    };
    if (r_lo < r_hi) fetch(r_lo);
    for (int m0 = r_lo; m0 < r_hi; m0 += TN_KS) {
#pragma unroll
        for (int i = 0; i < TN_KS / 4; ++i) {
            const int idx = tid + 256 * i;
            sa[idx >> 6][idx & 63] = ra[i];
            sb[idx >> 6][idx & 63] = rb[i];
        }
        __syncthreads();
        if (m0 + TN_KS < r_hi) fetch(m0 + TN_KS);
#pragma unroll
        for (int s = 0; s < TN_KS / 4; ++s) {
            const float a = sa[4 * s + q][16 * wave + l15];       // A^T: tile row = column of A
            csum += a;                                            // (rows 4 s + q of column 16 wave + l15; unused unless with_colsum)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sb[4 * s + q][16 * j + l15], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    const size_t rec = (size_t)N1 * N2 + (with_colsum ? N1 : 0);
    float* out = part + (size_t)bz * rec;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * wave + 4 * q + r, jj = j0 + 16 * j + l15;
            if (i < N1 && jj < N2) out[(size_t)i * N2 + jj] = acc[j][r];
        }
    if (with_colsum && by == 0) {                // the four row quarters of a column, added in a fixed order
        sa[q][16 * wave + l15] = csum;
        __syncthreads();
        if (tid < 64 && i0 + tid < N1) out[(size_t)N1 * N2 + i0 + tid] = (sa[0][tid] + sa[1][tid]) + (sa[2][tid] + sa[3][tid]);
    }
}

// out[i] = sum over the chunks of part[c][i], i < n1 + n2 (records of n1 + n2 floats; the first n1 go to out1, the rest to out2),
// for every job of the group (256 elements per workgroup).  Four partial sums over c = 0, 1, 2, 3 (mod 4), combined at the
// end: a fixed order, four loads in flight.
__global__ void reduce_partials_kernel(ReduceGroup g)
{
    int ji = 0;
    while (ji + 1 < g.count && (int)blockIdx.x >= g.first[ji + 1]) ++ji;
    const ReduceJob& jb = g.j[ji];
    const int i = ((int)blockIdx.x - g.first[ji]) * blockDim.x + threadIdx.x;
    const int n = jb.n1 + jb.n2, chunks = jb.chunks;
    if (i >= n) return;
    const float* __restrict__ part = jb.part;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= chunks; c += 4) {
        s0 += part[(size_t)c * n + i];
        s1 += part[(size_t)(c + 1) * n + i];
        s2 += part[(size_t)(c + 2) * n + i];
        s3 += part[(size_t)(c + 3) * n + i];
    }
    for (; c < chunks; ++c) s0 += part[(size_t)c * n + i];
    const float r = (s0 + s1) + (s2 + s3);
    if (i < jb.n1) jb.out1[i] = r;
    else jb.out2[i - jb.n1] = r;
}

// column sums of A [M][cols] (lda) over row chunks (bias gradients); partials [chunks][cols]
__global__ void colsum_partial_kernel(const float* __restrict__ A, int lda, float* __restrict__ part, int M, int cols, int rows_per_chunk)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    const int r_lo = blockIdx.y * rows_per_chunk, r_hi = min(M, r_lo + rows_per_chunk);
    float s = 0.f;
    for (int m = r_lo; m < r_hi; ++m) s += A[(size_t)m * lda + col];
    part[(size_t)blockIdx.y * cols + col] = s;
}

// dp = dy * (y >= 0 ? 1 : 0.01): the derivative of LeakyReLU(0.01) read off its output (0.01 > 0 keeps the sign); jobs as above
// with A = dy (lda), B = y (ldb), C = dp [M][N] dense, 1024 elements per workgroup
__global__ __launch_bounds__(256) void leaky_bwd_kernel(TrainGemmGroup g)
{
    const int ji = find_job(g, blockIdx.x);
    const TrainGemmJob& jb = g.j[ji];
    const size_t base = (size_t)(blockIdx.x - g.first[ji]) * 1024;
    const size_t total = (size_t)jb.M * jb.N;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = base + threadIdx.x + 256 * k;
        if (i >= total) break;
        const int m = (int)(i / jb.N), n = (int)(i % jb.N);
        const float gr = jb.A[(size_t)m * jb.lda + n];
        jb.C[i] = jb.B[(size_t)m * jb.ldb + n] > 0.f ? gr : 0.01f * gr;       // (torch's leaky_relu_backward: slope 0.01 at exactly 0)
    }
}

// Row chunks of a reduction over M rows into `tiles` output tiles: about 512 rows each, more when few tiles would leave the
// chip empty (a 64 x 64 weight gradient is one tile), at most LSTM_TRAIN_CHUNKS and never under 32 rows; a function of the
// shape only, so the order of the additions - and with it every bit of a gradient - is the same from run to run.
int chunk_count(int M, int tiles)
{
    int c = (M + 511) / 512;
    if (c > LSTM_TRAIN_CHUNKS) c = LSTM_TRAIN_CHUNKS;
    const int fill = (1024 + tiles - 1) / tiles;          // few output tiles (a 256 x 64 LSTM gradient has 4): four workgroups per CU
    if (c < fill) c = fill;
    if (c > (M + 63) / 64) c = (M + 63) / 64;
    return c < 1 ? 1 : (c > LSTM_TRAIN_MAX_CHUNKS ? LSTM_TRAIN_MAX_CHUNKS : c);
}
int rows_per_chunk(int M, int tiles)
{
    const int c = chunk_count(M, tiles);
    return ((M + c - 1) / c + 15) / 16 * 16;
}

}  // namespace

size_t sgemm_tn_scratch_floats(int M, int N1, int N2) { return (size_t)chunk_count(M, ((N1 + 63) / 64) * ((N2 + 63) / 64)) * ((size_t)N1 * N2 + N1); }
size_t colsum_scratch_floats(int M, int cols) { return (size_t)chunk_count(M, (cols + 255) / 256) * cols; }

void launch_sgemm_group(TrainGemmGroup& g, int trans_b, int accumulate, int leaky, hipStream_t s)
{
    g.first[0] = 0;
    for (int i = 0; i < g.count; ++i) {
        TrainGemmJob& j = g.j[i];
        j.tiles_x = (j.M + 63) / 64; j.tiles_y = (j.N + 63) / 64;
        g.first[i + 1] = g.first[i] + ((j.M > 0 && j.N > 0) ? j.tiles_x * j.tiles_y : 0);
    }
    if (g.first[g.count] <= 0) return;
    if (trans_b) hipLaunchKernelGGL(sgemm_kernel<true>, dim3(g.first[g.count]), dim3(256), 0, s, g, accumulate, leaky);
    else hipLaunchKernelGGL(sgemm_kernel<false>, dim3(g.first[g.count]), dim3(256), 0, s, g, accumulate, leaky);
}

void launch_sgemm(const float* A, int lda, const float* B, int ldb, int trans_b, float* C, int ldc, int M, int N, int K,
                  int accumulate, const float* bias, int leaky, hipStream_t s)
{
    TrainGemmGroup g;
    g.count = 1;
    g.j[0] = TrainGemmJob{A, B, C, bias, lda, ldb, ldc, M, N, K, 0, 0, 0};
    launch_sgemm_group(g, trans_b, accumulate, leaky, s);
}

// Weight (and bias) gradients of a group: job i has A = dp_i [M][N1_i] (lda), B = x_i [M][N2_i] (ldb), N = N1, K = N2,
// C = dW_i [N1][N2], bias = db_i (or null: no column sums).  Partials go to `scratch` (tn_group_scratch_floats).
size_t tn_group_scratch_floats(const TrainGemmGroup& g)
{
    size_t tot = 0;
    for (int i = 0; i < g.count; ++i) tot += sgemm_tn_scratch_floats(g.j[i].M, g.j[i].N, g.j[i].K);
    return tot;
}
void launch_sgemm_tn_group(const TrainGemmGroup& in, float* scratch, int L, int shift, hipStream_t s)
{
    TrainGemmGroup g = in;
    ReduceGroup rg;
    rg.count = 0; rg.first[0] = 0;
    g.first[0] = 0;
    size_t off = 0;
    for (int i = 0; i < g.count; ++i) {
        TrainGemmJob& j = g.j[i];
        const bool live = j.M > 0 && j.N > 0 && j.K > 0;
        j.tiles_x = (j.N + 63) / 64; j.tiles_y = (j.K + 63) / 64;
        j.rpc = live ? rows_per_chunk(j.M, j.tiles_x * j.tiles_y) : 1;
        const int chunks = live ? (j.M + j.rpc - 1) / j.rpc : 0;
        g.first[i + 1] = g.first[i] + j.tiles_x * j.tiles_y * chunks;
        if (live) {
            ReduceJob& r = rg.j[rg.count];
            r.part = scratch + off; r.out1 = in.j[i].C; r.n1 = j.N * j.K; r.out2 = const_cast<float*>(in.j[i].bias); r.n2 = in.j[i].bias ? j.N : 0;
            r.chunks = chunks;
            rg.first[rg.count + 1] = rg.first[rg.count] + (r.n1 + r.n2 + 255) / 256;
            ++rg.count;
            j.C = scratch + off;
            off += (size_t)chunks * ((size_t)r.n1 + r.n2);
        }
    }
    if (g.first[g.count] <= 0) return;
    hipLaunchKernelGGL(sgemm_tn_partial_kernel, dim3(g.first[g.count]), dim3(256), 0, s, g, L > 0 ? L : 1, shift);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(rg.first[rg.count]), dim3(256), 0, s, rg);
}

// out = A^T B; colsum (may be null) = column sums of A: both from one pass over the rows
void launch_sgemm_tn(const float* A, int lda, const float* B, int ldb, float* out, float* colsum, float* scratch, int M, int N1, int N2, int L,
                     int shift, hipStream_t s)
{
    if (N1 <= 0 || N2 <= 0) return;
    TrainGemmGroup g;
    g.count = 1;
    g.j[0] = TrainGemmJob{A, B, out, colsum, lda, ldb, 0, M, N1, N2, 0, 0, 0};
    launch_sgemm_tn_group(g, scratch, L, shift, s);
}

void launch_colsum(const float* A, int lda, float* out, float* scratch, int M, int cols, hipStream_t s)
{
    if (cols <= 0) return;
    const int rpc = rows_per_chunk(M, (cols + 255) / 256), chunks = (M + rpc - 1) / rpc;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, chunks), dim3(256), 0, s, A, lda, scratch, M, cols, rpc);
    ReduceGroup rg;
    rg.count = 1; rg.first[0] = 0; rg.first[1] = (cols + 255) / 256;
    rg.j[0] = ReduceJob{scratch, out, nullptr, cols, 0, chunks};
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(rg.first[1]), dim3(256), 0, s, rg);
}

// ---------------------------------------------------------------------------------------------- AdamW (train.py:50)
// torch.optim.AdamW(lr, betas, eps, weight_decay) in torch's own operation order (decoupled decay first, then
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)); bc1 = 1 - b1^t and bc2s = sqrt(1 - b2^t) come from the host.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);           // lerp, as torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const float denom = __builtin_sqrtf(vi) / bc2s + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
}
// many tensors per launch: the pointers travel in the kernel arguments (copied at launch: nothing to keep alive, no table to
// update in stream order); block b belongs to the tensor t with first_block[t] <= b < first_block[t + 1], 1024 elements per block
// state (optional): {lr, bc1, bc2s, step} in device memory, written by adamw_tick_kernel in stream order - a captured launch
// (hipGraph replay of a whole training iteration) then sees the step count and the learning rate of the replay, not of the capture.
__global__ __launch_bounds__(256) void adamw_group_kernel(AdamGroup a, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                          const float* __restrict__ state)
{
    if (state) { lr = state[0]; bc1 = state[1]; bc2s = state[2]; }
    int ti = 0;
    while (ti + 1 < a.count && (int)blockIdx.x >= a.first_block[ti + 1]) ++ti;
    float* p = a.p[ti]; const float* g = a.g[ti]; float* m = a.m[ti]; float* v = a.v[ti];
    const int n = a.n[ti], base = ((int)blockIdx.x - a.first_block[ti]) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + threadIdx.x + 256 * k;
        if (i >= n) break;
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        pi -= (lr / bc1) * (mi / (__builtin_sqrtf(vi) / bc2s + eps));
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}
void launch_adamw_group(const AdamGroup& a, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s, hipStream_t s,
                        const float* state)
{
    if (a.count <= 0 || a.first_block[a.count] <= 0) return;
    hipLaunchKernelGGL(adamw_group_kernel, dim3(a.first_block[a.count]), dim3(256), 0, s, a, lr, b1, b2, eps, wd, bc1, bc2s, state);
}
// step += 1; bc1 = 1 - b1^step, bc2s = sqrt(1 - b2^step) (double, as the host path computes them)
__global__ void adamw_tick_kernel(float* state, float b1, float b2)
{
    if (threadIdx.x || blockIdx.x) return;
    int* step = reinterpret_cast<int*>(state + 3);
    const int t = *step + 1;
    *step = t;
    state[1] = (float)(1.0 - pow((double)b1, (double)t));
    state[2] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
void launch_adamw_tick(float* state, float b1, float b2, hipStream_t s)
{
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, s, state, b1, b2);
}

void launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                  float bc1, float bc2s, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s);
}

// ---------------------------------------------------------------------------------------------- nn.Linear (+ LeakyReLU)
// A group = the same layer of several bands (or one layer): LinearJob i is y_i = act(x_i W_i^T + b_i) on M rows.
void launch_linear_group_forward(const LinearJob* jobs, int n, int M, int leaky, hipStream_t s)
{
    for (int i0 = 0; i0 < n; i0 += GEMM_GROUP) {
        TrainGemmGroup g;
        g.count = n - i0 < GEMM_GROUP ? n - i0 : GEMM_GROUP;
        for (int i = 0; i < g.count; ++i) {
            const LinearJob& L = jobs[i0 + i];
            g.j[i] = TrainGemmJob{L.x, L.w, L.y, L.b, L.ldx, L.K, L.ldy, M, L.N, L.K, 0, 0, 0};
        }
        launch_sgemm_group(g, 1, 0, leaky, s);
    }
}

static size_t round4(size_t v) { return (v + 3) & ~(size_t)3; }
size_t linear_group_scratch_floats(const LinearJob* jobs, int n, int M, int leaky)
{
    size_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const size_t r = sgemm_tn_scratch_floats(M, jobs[i].N, jobs[i].K > 0 ? jobs[i].K : 1), c = colsum_scratch_floats(M, jobs[i].N);
        tot += round4(r > c ? r : c) + (leaky ? round4((size_t)M * jobs[i].N) : 0);
    }
    return tot;
}

// Backward of the group: dp = dy * act'(y) (when leaky), dx = dp W (where wanted), dW = dp^T x and db = column sums of dp in one
// pass.  scratch: linear_group_scratch_floats.
void launch_linear_group_backward(const LinearJob* jobs, int n, int M, int leaky, float* scratch, hipStream_t s)
{
    if (M <= 0) return;
    for (int i0 = 0; i0 < n; i0 += GEMM_GROUP) {
        const int cnt = n - i0 < GEMM_GROUP ? n - i0 : GEMM_GROUP;
        const float* dp[GEMM_GROUP];
        int lddp[GEMM_GROUP];
        TrainGemmGroup ew, gx, gw;
        ew.count = gx.count = gw.count = 0;
        ew.first[0] = 0;
        // scratch of this chunk of jobs: first the dp buffers (when leaky), then ONE region for the reductions' partial records
        for (int i = 0; i < cnt; ++i) {
            const LinearJob& L = jobs[i0 + i];
            dp[i] = L.dy; lddp[i] = L.lddy;
            if (leaky) {
                float* buf = scratch;
                scratch += round4((size_t)M * L.N);
                ew.j[ew.count] = TrainGemmJob{L.dy, L.y, buf, nullptr, L.lddy, L.ldy, L.N, M, L.N, 0, 0, 0, 0};
                ew.first[ew.count + 1] = ew.first[ew.count] + (int)(((size_t)M * L.N + 1023) / 1024);
                ++ew.count;
                dp[i] = buf; lddp[i] = L.N;
            }
        }
        float* const red = scratch;
        for (int i = 0; i < cnt; ++i) {
            const LinearJob& L = jobs[i0 + i];
            const size_t r = sgemm_tn_scratch_floats(M, L.N, L.K > 0 ? L.K : 1), c = colsum_scratch_floats(M, L.N);
            scratch += round4(r > c ? r : c);
        }
        if (ew.count && ew.first[ew.count] > 0) hipLaunchKernelGGL(leaky_bwd_kernel, dim3(ew.first[ew.count]), dim3(256), 0, s, ew);
        for (int i = 0; i < cnt; ++i) {
            const LinearJob& L = jobs[i0 + i];
            if (L.dx && L.K > 0) gx.j[gx.count++] = TrainGemmJob{dp[i], L.w, L.dx, nullptr, lddp[i], L.K, L.lddx, M, L.K, L.N, 0, 0, 0};      // dx = dp W
            if (L.K > 0) gw.j[gw.count++] = TrainGemmJob{dp[i], L.x, L.dw, L.db, lddp[i], L.ldx, 0, M, L.N, L.K, 0, 0, 0};              // dW = dp^T x, db
            else launch_colsum(dp[i], lddp[i], L.db, red, M, L.N, s);       // (a layer without inputs: bias only; runs before the group below)
        }
        if (gx.count) launch_sgemm_group(gx, 0, 0, 0, s);
        if (gw.count) {
            launch_sgemm_tn_group(gw, red, 1, 0, s);
        }
    }
}

void launch_linear_train_forward(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N,
                                 int leaky, hipStream_t s)
{
    LinearJob j{};
    j.x = x; j.ldx = ldx; j.w = w; j.b = b; j.y = y; j.ldy = ldy; j.K = K; j.N = N;
    launch_linear_group_forward(&j, 1, M, leaky, s);
}

size_t linear_train_scratch_floats(int M, int K, int N, int leaky)
{
    LinearJob j{};
    j.K = K; j.N = N;
    return linear_group_scratch_floats(&j, 1, M, leaky);
}

// dx may be null; y is only read when leaky.  scratch: linear_train_scratch_floats.
void launch_linear_train_backward(const float* x, int ldx, const float* w, const float* y, int ldy, const float* dy, int lddy,
                                  float* dx, int lddx, float* dw, float* db, float* scratch, int M, int K, int N, int leaky,
                                  hipStream_t s)
{
    if (M <= 0 || N <= 0) return;
    LinearJob j{};
    j.x = x; j.ldx = ldx; j.w = w; j.y = const_cast<float*>(y); j.ldy = ldy; j.dy = dy; j.lddy = lddy; j.dx = dx; j.lddx = lddx; j.dw = dw; j.db = db; j.K = K; j.N = N;
    launch_linear_group_backward(&j, 1, M, leaky, scratch, s);
}

}  // namespace bsrnn
