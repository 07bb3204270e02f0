// Stand-alone probe for the co-residency hazard of round 1: do packed-fp32 instructions with operand
// modifiers (the forms the SLP vectorizer emitted for the FFT butterflies) give wrong results while waves of ANOTHER
// kernel issue MFMAs on the same SIMD?  The victim kernel evaluates chains of v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
// with neg / op_sel modifiers (inline assembly) next to the same arithmetic in scalar instructions and counts lanes where
// the two disagree; it runs alone, then beside an MFMA-only kernel on a second stream.
//   hipcc -O3 --offload-arch=gfx950 -o build/vpk_hazard tools/vpk_hazard.hip && ./build/vpk_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(unsigned long long* bad, int iters, int use_lds)
{
    __shared__ v2f buf[2][1024];
    const int tid = threadIdx.x;
    v2f a = {1.0f + 0.001f * tid, 0.5f - 0.002f * tid}, b = {0.25f + 0.003f * tid, -0.75f + 0.001f * tid};
    v2f sa = a, sb = b;                 // scalar twin
    unsigned long long mism = 0;
    const v2f sc = {__builtin_amdgcn_readfirstlane(iters) > 0 ? 0.5f : 0.25f, -0.5f};     // wave-uniform: lives in an SGPR pair
    for (int it = 0; it < iters; ++it) {
        v2f d, s, m, f;
        // packed: d = a - b (neg on src1); s = (a.y + b.x, a.x + b.y) via op_sel; m = a * b; f = fma(a, b, -d)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(s) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(f) : "v"(a), "v"(b), "v"(d));
        // the vectorised FFT code also feeds packed ops from SGPR pairs (0.5 scale factors) and shuffles halves with v_pk_mov_b32
        v2f g, h;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(g) : "v"(m), "s"(sc));
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(h) : "v"(g), "v"(f));
        mism += (g.x != m.x * sc.x) + (g.y != m.y * sc.y) + (h.x != g.y) + (h.y != f.x);
        // scalar twins (same IEEE operations)
        const v2f sd = {sa.x - sb.x, sa.y - sb.y};
        const v2f ss = {sa.y + sb.x, sa.x + sb.y};
        const v2f sm = {sa.x * sb.x, sa.y * sb.y};
        const v2f sf = {__builtin_fmaf(sa.x, sb.x, -sd.x), __builtin_fmaf(sa.y, sb.y, -sd.y)};
        mism += (d.x != sd.x) + (d.y != sd.y) + (s.x != ss.x) + (s.y != ss.y) + (m.x != sm.x) + (m.y != sm.y) + (f.x != sf.x) + (f.y != sf.y);
        d.x += 0.125f * h.x; s.y -= 0.125f * h.y;           // feed the shuffled values back into the chain
        // next operands: a bounded mix, optionally through LDS like an FFT pass (write, barrier, read a permuted slot)
        v2f na = {0.5f * (d.x + s.y), 0.5f * (m.x - f.y) + 0.1f}, nb = {0.5f * (s.x - d.y), 0.25f * (f.x + m.y) - 0.2f};
        if (use_lds) {
            buf[it & 1][tid] = na; buf[it & 1][tid + 256] = nb;
            __syncthreads();
            na = buf[it & 1][(tid * 5 + 1) & 255]; nb = buf[it & 1][256 + ((tid * 3 + 7) & 255)];
        }
        a = na; b = nb; sa = na; sb = nb;
        a.x = __builtin_fminf(__builtin_fmaxf(a.x, -4.f), 4.f); a.y = __builtin_fminf(__builtin_fmaxf(a.y, -4.f), 4.f);
        b.x = __builtin_fminf(__builtin_fmaxf(b.x, -4.f), 4.f); b.y = __builtin_fminf(__builtin_fmaxf(b.y, -4.f), 4.f);
        sa = a; sb = b;
    }
    if (mism) atomicAdd(bad, mism);
}

// Hypothesis H2 (round 2): in the vectorised FFT passes an LDS store is followed IMMEDIATELY by a packed-fp32 instruction
// that overwrites part of the store's data registers (a write-after-read on registers the LDS unit is still fetching):
//     ds_write_b128 v104, v[110:113]            ds_write2st64_b64 v18, v[92:93], v[94:95] offset1:4
//     v_pk_add_f32  v[112:113], ...             v_pk_mul_f32      v[92:93], ...
// The scalar build has the same adjacency with v_sub / v_mul (47 places) and is clean, so the hardware interlocks scalar
// VALU writes against an LDS store's pending data fetch; is a packed write interlocked as well when another kernel's MFMAs
// compete for the VGPR read ports?  The victim stores a known 16-byte value, overwrites half of its registers with poison
// by the very next instruction (MODE 0 / 2: v_pk_add_f32, MODE 1 / 3: two scalar v_add_f32 - the control), and reads the
// LDS back: poison in LDS = the store fetched its data after the overwrite.
template <int MODE>
__global__ __launch_bounds__(256) void victim_war(unsigned long long* bad, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[2][256 * 4 + 16 * 256];
    const int tid = threadIdx.x;
    unsigned long long mism = 0;
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    for (int it = 0; it < iters; ++it) {
        const float base = (float)(it & 1023) + 0.001f * tid;
        v4f_ d = {base, base + 1.f, base + 2.f, base + 3.f};
        const v2f px = {1e30f, 1e30f}, py = {1e30f, 1e30f};
        float* slot = &buf[it & 1][MODE < 2 ? 4 * tid : 2 * tid];      // (st64 form: second 8 bytes land 2048 bytes further)
        const unsigned addr = (unsigned)(size_t)slot;              // LDS byte address (the low 32 bits of the generic pointer)
        if (MODE == 0)
            asm volatile("ds_write_b128 %1, v[20:23]\n\tv_pk_add_f32 v[22:23], %2, %3" : "+{v[20:23]}"(d) : "v"(addr), "v"(px), "v"(py) : "memory");
        else if (MODE == 1)
            asm volatile("ds_write_b128 %1, v[20:23]\n\tv_add_f32 v22, %2, %3\n\tv_add_f32 v23, %2, %3" : "+{v[20:23]}"(d) : "v"(addr), "v"(px.x), "v"(py.x) : "memory");
        else if (MODE == 2)
            asm volatile("ds_write2st64_b64 %1, v[20:21], v[22:23] offset1:4\n\tv_pk_mul_f32 v[20:21], %2, %3" : "+{v[20:23]}"(d) : "v"(addr), "v"(px), "v"(py) : "memory");
        else
            asm volatile("ds_write2st64_b64 %1, v[20:21], v[22:23] offset1:4\n\tv_mul_f32 v20, %2, %3\n\tv_mul_f32 v21, %2, %3" : "+{v[20:23]}"(d) : "v"(addr), "v"(px.x), "v"(py.x) : "memory");
        __syncthreads();
        float got[4];
        if (MODE < 2) { for (int e = 0; e < 4; ++e) got[e] = slot[e]; }
        else { got[0] = slot[0]; got[1] = slot[1]; got[2] = slot[512]; got[3] = slot[513]; }    // offset1:4 = 4 * 64 * 8 bytes = 512 floats further
        for (int e = 0; e < 4; ++e) mism += got[e] != base + (float)e;
        asm volatile("" :: "v"(d));
        __syncthreads();
    }
    if (mism) atomicAdd(bad, mism);
}

__global__ __launch_bounds__(256) void aggressor(float* sink, int iters)
{
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.01f * (threadIdx.x + i)); y[i] = (_Float16)(0.02f * (i + 1)); }
    v16f acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, acc3, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.f) sink[0] = 1.f;
}

int main()
{
    unsigned long long* d_bad;
    float* d_sink;
    CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_sink, 4));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int use_lds = 0; use_lds < 2; ++use_lds)
        for (int with_aggr = 0; with_aggr < 2; ++with_aggr) {
            unsigned long long total = 0;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipMemsetAsync(d_bad, 0, 8, sb));
                CK(hipStreamSynchronize(sb));
                if (with_aggr) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, d_sink, 40000);
                hipLaunchKernelGGL(victim, dim3(2048), dim3(256), 0, sb, d_bad, 20000, use_lds);
                CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
                unsigned long long h = 0;
                CK(hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost));
                total += h;
            }
            printf("victim %s LDS exchange, %s: %llu packed results differ from their scalar twins (20 launches x 2048 x 256 lanes x 20000 x 12)\n",
                   use_lds ? "with" : "without", with_aggr ? "beside an MFMA kernel" : "alone", total);
        }
    // H2: write-after-read between an LDS store's data registers and the next (packed / scalar) instruction
    const char* names[4] = {"ds_write_b128 + v_pk_add_f32", "ds_write_b128 + 2 x v_add_f32 (control)", "ds_write2st64_b64 + v_pk_mul_f32",
                            "ds_write2st64_b64 + 2 x v_mul_f32 (control)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int with_aggr = 0; with_aggr < 2; ++with_aggr) {
            unsigned long long total = 0;
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipMemsetAsync(d_bad, 0, 8, sb));
                CK(hipStreamSynchronize(sb));
                if (with_aggr) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, d_sink, 40000);
                if (mode == 0) hipLaunchKernelGGL(victim_war<0>, dim3(2048), dim3(256), 0, sb, d_bad, 4000);
                else if (mode == 1) hipLaunchKernelGGL(victim_war<1>, dim3(2048), dim3(256), 0, sb, d_bad, 4000);
                else if (mode == 2) hipLaunchKernelGGL(victim_war<2>, dim3(2048), dim3(256), 0, sb, d_bad, 4000);
                else hipLaunchKernelGGL(victim_war<3>, dim3(2048), dim3(256), 0, sb, d_bad, 4000);
                CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
                unsigned long long h = 0;
                CK(hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost));
                total += h;
            }
            printf("H2 %-46s %-22s: %llu of %.3g stored words read back wrong\n", names[mode], with_aggr ? "beside an MFMA kernel" : "alone", total,
                   10.0 * 2048 * 256 * 4000 * 4);
        }
    return 0;
}
