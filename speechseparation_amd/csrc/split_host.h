// Host-side counterparts of the split-precision operand formats of gemm.hip (weights are split once at commit
// time; tools and tests use the same routines to build and to recombine planes).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace bsrnn {

inline uint16_t bf16_from_float(float f)                 // round to nearest even
{
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

inline uint16_t f16_from_float(float f)                  // round to nearest even, subnormals kept, |f| <= 65504
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    const uint32_t ab = x & 0x7fffffffu;
    if (ab >= 0x47800000u) return sign | 0x7c00;          // >= 65536: infinity (callers clamp first)
    if (ab < 0x38800000u) {                               // below 2^-14: subnormal grid of 2^-24
        float a; memcpy(&a, &ab, 4);
        return sign | (uint16_t)std::nearbyintf(a * 16777216.f);
    }
    const uint32_t mant = ab & 0x7fffffu;
    uint16_t h = (uint16_t)((((ab >> 23) - 112) << 10) | (mant >> 13));
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return sign | h;
}
inline float f16_to_float(uint16_t h)
{
    const int e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? std::ldexp((float)m, -24) : (e == 31 ? INFINITY : std::ldexp((float)(1024 + m), e - 25));
    return (h & 0x8000) ? -v : v;
}

// np = 3: bf16 pieces a = p0 + p1 + p2;  np = 2: fp16 pieces a ~ p0 + 2^-11 p1 (see gemm.hip).  planes: [np][n]
inline void split_planes_host(const float* src, size_t n, int np, uint16_t* planes)
{
    for (size_t i = 0; i < n; ++i) {
        if (np == 3) {
            const float a = src[i];
            const uint16_t p0 = bf16_from_float(a); const float r1 = a - bf16_to_float(p0);
            const uint16_t p1 = bf16_from_float(r1); const float r2 = r1 - bf16_to_float(p1);
            planes[i] = p0; planes[n + i] = p1; planes[2 * n + i] = bf16_from_float(r2);
        } else {
            const float a = std::fmin(std::fmax(src[i], -65504.f), 65504.f);
            const uint16_t p0 = f16_from_float(a);
            planes[i] = p0; planes[n + i] = f16_from_float((a - f16_to_float(p0)) * 2048.f);
        }
    }
}
// fp16x2 weights for the pipelined GEMM kernel: [N][K32 / 32][2 pieces][32] - both pieces of a 32-deep slab of a row
// share one 128-byte line, so a k-step fetches exactly one line per weight row.  w is [N][ldw] fp32 with zero padding
// up to K32 (a multiple of 32) provided by the caller through `K` (columns >= K read as zero).
// Rows are `wrow` 16-bit elements apart (>= 2 K32; see h2_row_stride).
inline int h2_row_stride(int K32)
{
    return 2 * K32;     // (an extra line per row against L2 channel camping was measured: no effect on gfx950)
}
inline void pack_h2_slabs_host(const float* w, int N, int K, int ldw, int K32, int wrow, uint16_t* out)
{
    for (int r = 0; r < N; ++r)
        for (int k = 0; k < K32; ++k) {
            const float v = k < K ? w[(size_t)r * ldw + k] : 0.f;
            uint16_t pc[2];
            split_planes_host(&v, 1, 2, pc);
            uint16_t* o = out + (size_t)r * wrow + (k / 32) * 64 + (k % 32);
            o[0] = pc[0];
            o[32] = pc[1];
        }
}

// Fragment streams of one Linear layer for the fused chain kernel (mlp_chain.hip): the weights are the MFMA's A operand
// (v_mfma_f32_32x32x16_f16: lane (r = l & 31, h = l >> 5) holds W[32 t + r][16 ks + 8 h + j], j < 8) and every wave of a
// row tile walks its own linear stream: for wave wn of NW, for its tiles t = wn, wn + NW, ..., for k-step ks, for piece
// (npl = 2: fp16x2, 1: plain fp16): 64 lanes x 8 halves = 1 KB.  Rows >= N and columns >= K are zeros.  Appends to `out`.
// `rag` = 1: the last feature tile (which holds at most 4 real features: N % 32 in 1..4) is not given to one wave as a
// whole tile but split over the k-steps: wave wn streams, BEHIND all the whole tiles of all waves, the fragments of its
// k-slice [chain_rag_first(K16, NW, wn), chain_rag_first(K16, NW, wn + 1)) of that tile (the waves' partial sums are added
// through LDS, mlp_chain.hip).  A band of 514 columns has 17 tiles for 8 waves: without this one wave runs 3 tiles where
// the others run 2 and every layer takes 3 tile times instead of 2.1.
inline int chain_rag_first(int K16, int NW, int wn)
{
    const int base = K16 / NW, rem = K16 - base * NW;
    return wn * base + (wn < rem ? wn : rem);
}
// bf = true (npl = 1 only): the one piece is bf16 instead of fp16 (BSRNN_GEMM=bf16)
inline void pack_chain_layer_host(const float* w, int N, int K, int ldw, int NW, int npl, std::vector<uint16_t>& out, int rag = 0, bool bf = false)
{
    const int K16 = (K + 15) / 16, NTL = (N + 31) / 32 - rag;
    auto frag = [&](int t, int ks, int pc) {
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = 32 * t + (l & 31), k = 16 * ks + 8 * (l >> 5) + j;
                const float v = (n < N && k < K) ? w[(size_t)n * ldw + k] : 0.f;
                uint16_t p[2];
                split_planes_host(&v, 1, 2, p);
                out.push_back(bf ? bf16_from_float(v) : p[pc]);
            }
    };
    for (int wn = 0; wn < NW; ++wn)
        for (int t = wn; t < NTL; t += NW)
            for (int ks = 0; ks < K16; ++ks)
                for (int pc = 0; pc < npl; ++pc) frag(t, ks, pc);
    if (rag)
        for (int wn = 0; wn < NW; ++wn)
            for (int ks = chain_rag_first(K16, NW, wn); ks < chain_rag_first(K16, NW, wn + 1); ++ks)
                for (int pc = 0; pc < npl; ++pc) frag(NTL, ks, pc);
}

// The same for the 48-row geometry on v_mfma_f32_16x16x32_f16 (feature tiles of 16, k-steps of 32): lane (n = l & 15,
// kb = l >> 4) holds W[16 t + n][32 ks + 8 kb + j].
inline void pack_chain_layer16_host(const float* w, int N, int K, int ldw, int NW, int npl, std::vector<uint16_t>& out, bool bf = false)
{
    const int K32 = (K + 31) / 32, FT = (N + 15) / 16;
    for (int wn = 0; wn < NW; ++wn)
        for (int t = wn; t < FT; t += NW)
            for (int ks = 0; ks < K32; ++ks)
                for (int pc = 0; pc < npl; ++pc)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int n = 16 * t + (l & 15), k = 32 * ks + 8 * (l >> 4) + j;
                            const float v = (n < N && k < K) ? w[(size_t)n * ldw + k] : 0.f;
                            uint16_t p[2];
                            split_planes_host(&v, 1, 2, p);
                            out.push_back(bf ? bf16_from_float(v) : p[pc]);
                        }
}

// Paired geometry (mlp_chain.hip::chain_body_pair, probe only: tools/chain_bench.hip): the stream of the workgroup that holds K-half `half` of the layer input.  Wave wn of 8
// walks, k-step by k-step over ITS half of K, first the feature tiles of the PARTNER's half of the outputs ((1 - half) FT / 2 + wn + 8 j),
// then the tiles of its own half (half FT / 2 + wn + 8 j).  N and K multiples of 64.
inline void pack_chain_layer16_pair_host(const float* w, int N, int K, int ldw, int half, int npl, std::vector<uint16_t>& out, bool bf = false)
{
    const int K32h = K / 64, FT = N / 16, FTh = FT / 2, k0 = half * (K / 2);
    for (int wn = 0; wn < 8; ++wn)
        for (int own = 0; own < 2; ++own)
            for (int j = 0; wn + 8 * j < FTh; ++j) {
                const int t = (own ? half : 1 - half) * FTh + wn + 8 * j;
                for (int ks = 0; ks < K32h; ++ks)
                    for (int pc = 0; pc < npl; ++pc)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int n = 16 * t + (l & 15), k = k0 + 32 * ks + 8 * (l >> 4) + e;
                                const float v = w[(size_t)n * ldw + k];
                                uint16_t p[2];
                                split_planes_host(&v, 1, 2, p);
                                out.push_back(bf ? bf16_from_float(v) : p[pc]);
                            }
            }
}

inline float join_planes_host(const uint16_t* planes, size_t n, size_t i, int np)
{
    if (np == 3) return (bf16_to_float(planes[i]) + bf16_to_float(planes[n + i])) + bf16_to_float(planes[2 * n + i]);
    return f16_to_float(planes[i]) + f16_to_float(planes[n + i]) * (1.f / 2048.f);
}

}  // namespace bsrnn
