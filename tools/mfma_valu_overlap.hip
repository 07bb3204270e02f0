// Measurement-only probe: do matrix-pipe instructions (v_mfma_f32_16x16x32_f16) overlap with vector-ALU work on gfx950,
// (a) inside one wave (independent instructions interleaved in program order) and (b) between the two waves of a SIMD?
// The recurrent kernels (lstm.hip) are planned around the answer.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/mfma_valu_overlap tools/mfma_valu_overlap.hip
// One workgroup; waves 0-3 (one per SIMD) run stream `a`, waves 4-7 (the SIMDs' second waves, if launched) run stream `b`.
// Streams: 0 idle, 1 MFMA only, 2 v_exp only, 3 v_fma only, 4 MFMA+v_exp alternating, 5 MFMA + 4 v_fma alternating,
// 6 v_exp + 4 v_fma, 7 MFMA + v_exp + 3 v_fma.  Output: shader clocks per loop iteration (16 slices) per wave group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void stream(int iters, h8v a, h8v b, v4f* acc, float* t, float* f)
{
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            constexpr bool M = MODE == 1 || MODE == 4 || MODE == 5 || MODE == 7;
            constexpr bool T = MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7;
            constexpr int NV = MODE == 3 ? 4 : MODE == 5 ? 4 : MODE == 6 ? 4 : MODE == 7 ? 3 : 0;
            if (M) {
                asm volatile("" : "+v"(a));
                acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[s & 3], 0, 0, 0);
            }
            if (T) {
                asm volatile("" : "+v"(t[s & 7]));
                t[s & 7] = __builtin_amdgcn_exp2f(t[s & 7]);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                asm volatile("" : "+v"(f[(4 * s + v) & 7]));
                f[(4 * s + v) & 7] = __builtin_fmaf(f[(4 * s + v) & 7], 0.999f, 0.001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(512) void probe(int mode_a, int mode_b, int iters, long long* out, float* sink)
{
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? mode_a : mode_b;
    h8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float t[8], f[8];
    for (int i = 0; i < 8; ++i) { t[i] = -0.01f * (threadIdx.x + i); f[i] = 0.5f + i; }
    __syncthreads();
    const long long t0 = clock64();
    switch (mode) {
    case 1: stream<1>(iters, a, b, acc, t, f); break;
    case 2: stream<2>(iters, a, b, acc, t, f); break;
    case 3: stream<3>(iters, a, b, acc, t, f); break;
    case 4: stream<4>(iters, a, b, acc, t, f); break;
    case 5: stream<5>(iters, a, b, acc, t, f); break;
    case 6: stream<6>(iters, a, b, acc, t, f); break;
    case 7: stream<7>(iters, a, b, acc, t, f); break;
    default: break;
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += t[i] + f[i];
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

int main()
{
    long long* out; float* sink;
    CK(hipMalloc(&out, 8 * 8)); CK(hipMalloc(&sink, 512 * 4));
    const int iters = 200;
    const char* nm[8] = {"idle", "MFMA", "v_exp", "v_fma x4", "MFMA+v_exp", "MFMA+4 v_fma", "v_exp+4 v_fma", "MFMA+v_exp+3 v_fma"};
    printf("shader clocks per 16-slice iteration (one slice = the named group), first wave of each group\n");
    const int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {5, 0}, {6, 0}, {7, 0},
                            {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {1, 6}, {4, 4}, {5, 5}, {7, 7}, {4, 6}};
    for (auto& p : pairs) {
        const int threads = p[1] ? 512 : 256;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, p[0], p[1], iters, out, sink);
            CK(hipDeviceSynchronize());
        }
        long long h[8];
        CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        printf("  waves 0-3: %-20s %7.1f clk   waves 4-7: %-20s %7.1f clk\n", nm[p[0]], (double)h[0] / iters, nm[p[1]], p[1] ? (double)h[4] / iters : 0.0);
    }
    return 0;
}
