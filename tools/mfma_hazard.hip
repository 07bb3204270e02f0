// Measurement-only: does rewriting the A/B source registers between f32 MFMAs (as any real GEMM loop
// must) cost issue rate?  3 waves per SIMD (768 threads, 1 block per CU), 32x32x2, 2 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[768 * 16];
    for (int i = threadIdx.x; i < 768 * 16; i += 768) lds[i] = (i % 97) * 0.01f;
    __syncthreads();
    v4f a0 = {1.f, 2.f, 3.f, 4.f}, a1 = {.5f, .25f, .125f, 2.f}, b = {1.f, -1.f, .5f, 3.f};
    v4f ka0 = a0 * 1.5f, ka1 = a1 * 1.5f, kb = b * 0.75f;
    v16f c0 = {}, c1 = {};
    const float* p = &lds[threadIdx.x * 16];
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {                       // VALU rewrites the operand registers (like a register-staged pipeline)
            a0 = ka0; a1 = ka1; b = kb;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b));
        } else if (MODE == 2) {                // LDS reads rewrite them (like the real loop)
            a0 = *reinterpret_cast<const v4f*>(p);
            a1 = *reinterpret_cast<const v4f*>(p + 4);
            b = *reinterpret_cast<const v4f*>(p + 8);
        } else if (MODE == 3) {                // LDS reads into a second register set, ping-pong by unrolling
            ka0 = *reinterpret_cast<const v4f*>(p);
            ka1 = *reinterpret_cast<const v4f*>(p + 4);
            kb = *reinterpret_cast<const v4f*>(p + 8);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], c1, 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka0[e], kb[e], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka1[e], kb[e], c1, 0, 0, 0);
            }
            a0 = *reinterpret_cast<const v4f*>(p + 12);
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = c0[0] + c1[5] + a0[0];
}

template <int MODE>
int run(const char* name)
{
    float* out;
    const int iters = 4000;
    CK(hipMalloc(&out, 256 * 768 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double n_mfma = (MODE == 3 ? 16.0 : 8.0) * iters * 256 * 12;
    printf("%-44s %.3f ms -> %.1f TFLOP/s\n", name, ms, n_mfma * 4096 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
    return 0;
}
int main()
{
    run<0>("constant operands");
    run<1>("operands rewritten by v_mov every 8 MFMAs");
    run<2>("operands rewritten by ds_read_b128 every 8");
    run<3>("ds_read into alternate register set");
    return 0;
}
