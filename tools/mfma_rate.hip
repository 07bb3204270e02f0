// Measurement-only: issue rate of the f32-input MFMA shapes in dependent-chain patterns like the
// ones the LSTM / GEMM kernels use.  Prints shader cycles per MFMA (s_memtime) for 1 and 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ float g_rand[4096];

template <int SHAPE, int CHAINS, int RANDOM = 0>
__global__ void k(float* out, unsigned long long* cyc, int iters)
{
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    if (RANDOM && SHAPE == 2) {       // random per-lane operands held in registers: realistic toggling, no memory traffic
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { av[u] = g_rand[(threadIdx.x * 16 + u) & 4095]; bv[u] = g_rand[(threadIdx.x * 16 + u + 2048 + 7 * blockIdx.x) & 4095]; }
        v16f c[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], c[u % CHAINS], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(av[u]), "+v"(bv[u]));
        }
        out[threadIdx.x + blockIdx.x * blockDim.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
        return;
    }
    unsigned long long t0 = 0, t1 = 0;
    if (SHAPE == 0) {
        v4f c[4] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u % CHAINS], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[threadIdx.x + blockIdx.x * blockDim.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if (SHAPE == 1) {
        v4f c[4] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u % CHAINS], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[threadIdx.x + blockIdx.x * blockDim.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        v16f c[4] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[u % CHAINS], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[threadIdx.x + blockIdx.x * blockDim.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int SHAPE, int CHAINS, int RANDOM = 0>
int run(const char* name, int threads)
{
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = RANDOM ? 4000 : 2000;
    CK(hipMalloc(&out, blocks * threads * 4));
    CK(hipMalloc(&cyc, blocks * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SHAPE, CHAINS, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<SHAPE, CHAINS, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    unsigned long long h[16];
    CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    const double flop_per = SHAPE == 0 ? 512.0 : SHAPE == 1 ? 2048.0 : 4096.0;
    const double flops = flop_per * iters * 16.0 * blocks * (threads / 64);
    printf("%-8s chains=%d waves/SIMD=%d : %.2f cyc/MFMA/wave | wall %.3f ms -> %.1f TFLOP/s | wave0 cycles %llu -> implied clock %.2f GHz\n", name, CHAINS, threads / 256,
           (double)h[0] / (iters * 16.0), ms, flops / (ms * 1e-3) / 1e12, h[0], h[0] / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main()
{
    {
        float h[4096];
        for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rand), h, sizeof h) != hipSuccess) return 1;
    }
    run<2, 4, 1>("32x32x2 RANDOM-REG", 256); run<2, 4, 1>("32x32x2 RANDOM-REG", 512); run<2, 4, 1>("32x32x2 RANDOM-REG", 768);
    run<0, 4>("4x4x1", 256); run<0, 4>("4x4x1", 512); run<0, 2>("4x4x1", 256); run<0, 1>("4x4x1", 256); run<0, 1>("4x4x1", 512);
    run<1, 4>("16x16x4", 256); run<1, 4>("16x16x4", 512); run<1, 1>("16x16x4", 256);
    run<2, 4>("32x32x2", 256); run<2, 2>("32x32x2", 256); run<2, 2>("32x32x2", 512); run<2, 2>("32x32x2", 768); run<2, 1>("32x32x2", 256);
    return 0;
}
