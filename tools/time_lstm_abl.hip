// timing-only ablation runner for time_lstm_h2w_kernel (results are wrong by construction under TIME_ABL != 0)
#include "../speechseparation_amd/csrc/lstm.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace bsrnn;
namespace bsrnn { bool force_f32() { return false; } int gemm_mode() { return GEMM_FP16X2; } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int R = 64, T = 126, K = 12, N = R * K;
    const size_t nz = (size_t)R * T * K * 64;
    float *z, *h, *b, *bfc; uint16_t *w, *wfc; int* flag;
    CK(hipMalloc(&z, nz * 4)); CK(hipMalloc(&h, nz * 4)); CK(hipMalloc(&w, 2 * 4 * 4 * 4 * 2 * 64 * 8 * 2)); CK(hipMalloc(&wfc, 4 * 2 * 2 * 64 * 8 * 2));
    CK(hipMalloc(&b, 512 * 4)); CK(hipMalloc(&bfc, 256)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    std::vector<float> hz(nz); unsigned s = 1; for (auto& v : hz) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    std::vector<uint16_t> hw(2 * 4 * 4 * 4 * 2 * 64 * 8); for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x2c00 + ((s >> 8) & 0x3ff) + (((s >> 20) & 1) << 15)); }
    CK(hipMemcpy(z, hz.data(), nz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfc, hw.data(), 4 * 2 * 2 * 64 * 8 * 2, hipMemcpyHostToDevice)); CK(hipMemset(b, 0, 512 * 4)); CK(hipMemset(bfc, 0, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) for (int v = 0; v < 2; ++v) {
        CK(hipEventRecord(e0, 0));
        if (v == 0) hipLaunchKernelGGL((time_lstm_h2w_kernel<false, false>), dim3(N / 4), dim3(1024), 0, 0, z, h, (const uint4*)w, b, (const uint4*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, R, T, K, flag, (unsigned long long*)nullptr);
        else hipLaunchKernelGGL((time_lstm_h2w_kernel<true, false>), dim3(N / 4), dim3(1024), 0, 0, z, h, (const uint4*)w, b, (const uint4*)wfc, bfc, (const float*)nullptr, (float*)nullptr, R, T, K, flag, (unsigned long long*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best[v] = std::min(best[v], ms * 1e3f);
    }
    int hf = 0; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("TIME_ABL=%d: plain %.1f us, fused %.1f us (flag %d)\n", TIME_ABL, best[0], best[1], hf);
    return 0;
}
