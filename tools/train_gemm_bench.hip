// Measurement-only: the training step's generic fp32 products (speechseparation_amd/csrc/train_ops.hip) on representative shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -o build/train_gemm_bench tools/train_gemm_bench.hip
#ifndef TRAIN_OPS_SRC
#define TRAIN_OPS_SRC "../speechseparation_amd/csrc/train_ops.hip"
#endif
#include TRAIN_OPS_SRC
#include <cstdio>
#include <vector>
using namespace bsrnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int shapes[][3] = {{8064, 768, 768}, {8064, 514, 514}, {8064, 192, 192}, {8064, 64, 64}, {8064, 128, 64}, {96768, 256, 64}, {96768, 256, 128}, {252, 768, 768}};
    float *A, *B, *C, *S;
    const size_t big = (size_t)96768 * 768;
    CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&B, big * 4)); CK(hipMalloc(&C, big * 4)); CK(hipMalloc(&S, (size_t)64 * 768 * 768 * 4));
    CK(hipMemset(A, 0, big * 4)); CK(hipMemset(B, 0, big * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        const int M = sh[0], K = sh[1], N = sh[2];
        float ms[3];
        for (int kind = 0; kind < 3; ++kind) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (kind == 0) launch_sgemm(A, K, B, K, 1, C, N, M, N, K, 0, nullptr, 0, 0);          // y = x W^T
                else if (kind == 1) launch_sgemm(A, N, B, K, 0, C, K, M, K, N, 0, nullptr, 0, 0);     // dx = dp W
                else launch_sgemm_tn(A, N, B, K, C, nullptr, S, M, N, K, 1, 0, 0);                             // dW = dp^T x
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[kind], e0, e1));
            }
        }
        const double gf = 2.0 * M * K * N / 1e9;
        printf("M %6d K %4d N %4d:  x W^T %8.1f us %6.1f TF   dp W %8.1f us %6.1f TF   dp^T x %8.1f us %6.1f TF\n", M, K, N,
               ms[0] * 1e3, gf / ms[0], ms[1] * 1e3, gf / ms[1], ms[2] * 1e3, gf / ms[2]);
    }
    return 0;
}
