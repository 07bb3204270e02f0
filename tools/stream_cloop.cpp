// Wall time of the one-frame streaming step measured from C, with nothing but the C ABI between the loop and the GPU (VERDICT r02 #6:
// host time and device time separated; the Python wrapper's share is measured by tools/stream_probe.py).
//   g++ -O2 -std=c++17 -I include -o build/stream_cloop tools/stream_cloop.cpp -Lspeechseparation_amd/lib -lbsrnn_hip -Wl,-rpath,$PWD/speechseparation_amd/lib
//   build/stream_cloop [C=2] [steps=2000]
#include "bsrnn_hip.h"

#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %d %s\n", #x, rc_, bsrnn_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 2, steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int32_t widths[12] = {1, 2, 3, 6, 12, 24, 48, 96, 192, 384, 257, 0};          // generate_bandsplits(), bsrnn.py:247-326
    bsrnn_ctx* ctx = nullptr;
    CK(bsrnn_create(0, widths, 12, &ctx));
    unsigned seed = 1;
    for (int i = 0; i < bsrnn_param_count(ctx); ++i) {                                   // random weights of the default-init scale
        const char* key; int64_t d0, d1; int32_t nd;
        CK(bsrnn_param_info(ctx, i, &key, &d0, &d1, &nd));
        const int64_t n = nd == 2 ? d0 * d1 : d0;
        std::vector<float> v((size_t)n);
        const float sc = nd == 2 ? 1.f / sqrtf((float)d1) : 0.05f;
        for (auto& x : v) { seed = seed * 1664525u + 1013904223u; x = (((seed >> 8) & 0xffff) / 32768.f - 1.f) * sc; }
        CK(bsrnn_set_param(ctx, key, v.data(), n));
    }
    CK(bsrnn_commit_params(ctx));
    bsrnn_stream* st = nullptr;
    CK(bsrnn_stream_create(ctx, C, &st));
    void *d_in = nullptr, *d_out = nullptr;
    CK(bsrnn_dev_alloc(ctx, (int64_t)C * 1024 * 4, &d_in));
    CK(bsrnn_dev_alloc(ctx, (int64_t)C * 1024 * 4, &d_out));
    std::vector<float> h((size_t)C * 1024), ho((size_t)C * 1024);
    for (auto& x : h) { seed = seed * 1664525u + 1013904223u; x = (((seed >> 8) & 0xffff) / 32768.f - 1.f) * 0.1f; }
    CK(bsrnn_copy_h2d(ctx, d_in, h.data(), (int64_t)C * 1024 * 4));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("compute mode %s, C = %d rows, %d steps each\n", bsrnn_compute_mode(), C, steps);
    for (int pol = 0; pol < 2; ++pol) {
        CK(bsrnn_set_range_policy(ctx, pol ? BSRNN_RANGE_EXACT : BSRNN_RANGE_DEFERRED));
        for (int i = 0; i < 50; ++i) CK(bsrnn_stream_step(st, (const float*)d_in, (float*)d_out, 1.0f, nullptr));
        CK(bsrnn_sync(ctx, nullptr));
        double host = 0;
        const auto t0 = now();
        for (int i = 0; i < steps; ++i) {
            const auto a = now();
            CK(bsrnn_stream_step(st, (const float*)d_in, (float*)d_out, 1.0f, nullptr));
            host += us(a, now());
        }
        CK(bsrnn_sync(ctx, nullptr));
        const double wall = us(t0, now()) / steps;
        printf("bsrnn_stream_step, device buffers, range policy %-8s: %7.1f us per chunk wall, %6.1f us of it inside the call on the host\n",
               pol ? "exact" : "deferred", wall, host / steps);
    }
    {
        for (int i = 0; i < 50; ++i) CK(bsrnn_stream_step_host(st, h.data(), ho.data(), 1.0f));
        const auto t0 = now();
        for (int i = 0; i < steps; ++i) CK(bsrnn_stream_step_host(st, h.data(), ho.data(), 1.0f));
        printf("bsrnn_stream_step_host (the LADSPA plugin's call: H2D, step, D2H, synchronous): %7.1f us per chunk\n", us(t0, now()) / steps);
    }
    bsrnn_stream_destroy(st);
    bsrnn_dev_free(ctx, d_in); bsrnn_dev_free(ctx, d_out);
    bsrnn_destroy(ctx);
    return 0;
}
