// Measurement-only: phase breakdown of the time-axis LSTM kernel (100 MHz stamps inside the kernel).
#include "../speechseparation_amd/csrc/lstm.hip"
#include <cstdio>
#include <vector>
using namespace bsrnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const int R = 64, T = 126, K = 12;
    const size_t nz = (size_t)R * T * K * 64;
    float *z, *h, *w, *b; unsigned long long* dbg;
    CK(hipMalloc(&z, nz * 4)); CK(hipMalloc(&h, nz * 4)); CK(hipMalloc(&w, 2 * 4 * 128 * 64 * 4)); CK(hipMalloc(&b, 512 * 4));
    CK(hipMalloc(&dbg, 4 * 8 * 4 * 8));
    std::vector<float> hz(nz), hw(2 * 4 * 128 * 64);
    for (auto& v : hz) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    CK(hipMemcpy(z, hz.data(), nz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b, 0, 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(time_lstm_kernel, dim3(R * K / 4), dim3(512), 0, 0, z, h, w, b, (const float*)nullptr, (float*)nullptr, R, T, K, rep == 2 ? dbg : nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("time_lstm launch %d: %.1f us\n", rep, ms * 1e3);
    }
    unsigned long long hd[4 * 8 * 4];
    CK(hipMemcpy(hd, dbg, sizeof hd, hipMemcpyDeviceToHost));
    const char* nm[4] = {"recurrent gemv", "cell", "input gemv", "barrier+misc"};
    for (int blk = 0; blk < 2; ++blk)
        for (int wv = 0; wv < 8; wv += 4) {
            printf("block %d wave %d (layer %d):", blk, wv, wv / 4);
            double tot = 0;
            for (int k = 0; k < 4; ++k) tot += hd[(blk * 8 + wv) * 4 + k];
            for (int k = 0; k < 4; ++k) printf("  %s %.1f us (%.0f%%, %.0f ns/step)", nm[k], hd[(blk * 8 + wv) * 4 + k] / 100.0, 100.0 * hd[(blk * 8 + wv) * 4 + k] / tot, hd[(blk * 8 + wv) * 4 + k] * 10.0 / T);
            printf("  total %.1f us\n", tot / 100.0);
        }
    return 0;
}
