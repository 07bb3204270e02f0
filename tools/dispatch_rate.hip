// Measurement-only: cost of dispatching many short 256-thread workgroups (with the GEMM's LDS footprint).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int* tab, const float* src)
{
    __shared__ float lds[27648 / 4];
    if (MODE == 0) { if (threadIdx.x == 0) lds[0] = 1.f; __syncthreads(); if (threadIdx.x == 1) out[blockIdx.x] = lds[0]; return; }
    // MODE 1: dependent chain like the GEMM prologue: table -> descriptor -> data -> LDS -> barrier
    const int a = tab[blockIdx.x & 1023];
    const int b = tab[1024 + (a & 1023)];
    const float v = src[(size_t)(b & 1023) * 2050 + threadIdx.x];
    lds[threadIdx.x] = v;
    __syncthreads();
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x];
}
int main()
{
    float *out, *src; int* tab;
    CK(hipMalloc(&out, (size_t)16384 * 256 * 4)); CK(hipMalloc(&src, (size_t)1024 * 2050 * 4)); CK(hipMalloc(&tab, 2048 * 4));
    CK(hipMemset(tab, 0, 2048 * 4)); CK(hipMemset(src, 0, (size_t)1024 * 2050 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int n : {256, 1024, 4096, 16384}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n), dim3(256), 0, 0, out, tab, src);
                else hipLaunchKernelGGL(k<1>, dim3(n), dim3(256), 0, 0, out, tab, src);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("mode %d  %6d WGs: %.1f us  (%.3f us per WG per CU)\n", mode, n, best * 1e3, best * 1e3 / (n / 256.0));
        }
    return 0;
}
