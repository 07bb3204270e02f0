// Fused per-band MLP chains on the gfx950 matrix cores.
//
// BandSplit (bsrnn.py:404-415):      x_b -> pre.0 -> pre.2 (= residual P) -> fc.0 -> fc.2 -> fc.4 -> Z[:, :, b, :]
// MaskEstimation (bsrnn.py:420-443): Z[:, :, b, :] -> back.0 -> back.2 -> back.4 -> post.0 -> post.2, + P, * x -> y_b
// One workgroup runs the WHOLE five-layer chain of one band for a tile of frame rows; the intermediates never leave
// the CU (the per-layer launches of gemm.hip wrote and re-read 66-93 MB per layer: 2.07 GB per step at 3.5 TB/s for
// 1.30 GB of algorithmic bytes, and paid ~25 us of fixed cost per launch - profiles/r01k_*).
//
// Arithmetic: the fp16x2 scheme of gemm.hip (operands as two fp16 pieces, hi += w1 x1, lo += w1 x2 + w2 x1, result
// hi + 2^-11 lo, fp32 accumulate), same products in the same k order, so the chain reproduces the unfused flow.
//
// Orientation: the products are computed TRANSPOSED, D^T = W X^T, on v_mfma_f32_32x32x16_f16:
//   A operand = weights   (lane (r, h): W[32 t + r][16 ks + 8 h + j]),  streamed global -> VGPR, never through LDS: every
//               weight byte is used by exactly one wave of the workgroup, and the host packs, per (band, layer, wave),
//               the fragments in the order that wave consumes them: one linear stream, each fragment a coalesced 1 KB;
//   B operand = activations (lane (m, h): x[m][16 ks + 8 h + j]), shared by all waves, in LDS as two fp16 pieces in
//               [k / 8][row m][8] order: every fragment read is one linear ds_read_b128 (address = base + 16 lane);
//   D = [feature n][row m]: lane = activation row, registers = 16 features.  The next layer sums over features, i.e. over
//               REGISTERS of D, so its B fragments need no transpose: a lane splits its own values and stores 4 consecutive
//               features (8 bytes per piece) into the [k / 8][m][8] image - plain ds_write_b64, conflict-free.
// So a layer is: K loop without barriers or staging (weights in flight in registers, activations static in LDS), barrier,
// epilogue (bias, LeakyReLU, split, LDS image of the next layer's input [+ P to HBM]), barrier.
//
// Geometry: a workgroup = 8 waves in GR groups of NW = 8 / GR; a group owns RT row tiles of 32 frame rows (the MFMA's N),
// its waves share the feature tiles (t = wn, wn + NW, ...: at most CT = 3 each, worked through one after the other) and
// each wave multiplies every weight fragment it loads with all RT row tiles of its group.  The K loop of a wide band is
// bound by what a CU can pull from L2 (~70 GB/s, measured), so rows per weight byte is the lever there, and LDS capacity
// (rows x Kmax x 4 bytes of activation image) sets it; the narrow bands are latency bound and want their eight waves busy:
//   bands up to 768 columns: RT 1, GR 1 ( 32 rows, 96 KB image);   up to 544: RT 2, GR 1 ( 64 rows, shared weights);
//   up to 288: RT 2, GR 2 (128 rows, two groups of 4 waves);        up to 192: RT 1, GR 4 (128 rows, four groups of 2 waves);
//   up to 128: RT 1, GR 8 (256 rows: every wave runs the whole chain of its own row tile, no barriers).
// One launch per chain over a task table (band, row block), longest workgroups first (api.hip, build_chain_tasks).
#include "kernels.h"

#include <type_traits>

namespace bsrnn {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef const float __attribute__((address_space(1)))* gcf;
typedef float __attribute__((address_space(1)))* gf;
typedef const v4f __attribute__((address_space(1)))* gc4;
typedef v4f __attribute__((address_space(1)))* g4;
typedef const h8 __attribute__((address_space(1)))* gch8;
typedef const char __attribute__((address_space(1)))* gcc;
typedef unsigned u4x __attribute__((ext_vector_type(4)));

#ifndef CHAIN_NO_RAG_CODE
#define CHAIN_NO_RAG_CODE 0     // measurement only: compile the ragged-tile k-split out (register pressure probe)
#endif
#ifndef CHAIN_PD
#define CHAIN_PD 8                 // k-steps of weight fragments in flight per wave (register sets of 2 fragments)
#endif
constexpr int PD = CHAIN_PD;
// measurement only (tools/ab_chain_pd.sh): 1 = no MFMAs (operands kept alive), 2 = no weight loads after a tile's first PD steps,
// 4 = no activation fragment reads after a tile's first, 8 = all waves read the same weight stream; results are wrong by construction
#ifndef CHAIN_ABL
#define CHAIN_ABL 0
#endif
// measurement only (tools/chain_bench.hip): 100 MHz stamps at the phase boundaries of every wave of the first workgroups
#ifndef CHAIN_TRACE
#define CHAIN_TRACE 0
#endif
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 0
#endif
#ifndef CHAIN_PD64
#define CHAIN_PD64 3               // prefetch depth of the 64-row geometry on 16 x 16 tiles (four row tiles)
#endif
#ifndef CHAIN_TP64
#define CHAIN_TP64 2               // ... and its feature tiles per pass over K
#endif
#ifndef CHAIN_PD48
#define CHAIN_PD48 4               // prefetch depth of the 48-row geometry (its k-steps carry two tiles' fragments)
#endif

// One-term bf16 mode (BSRNN_GEMM=bf16, TERMS = -1 in the templates below): the 16-bit containers of the LDS images and of the weight
// fragments hold bf16 instead of fp16 (same size, same layouts); activations are rounded to nearest even (v_cvt_pk_bf16_f32) and the
// products run on v_mfma_f32_*_bf16.  No range limit, 8 significant bits: BASELINE config 2 as it is named.
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ h4 to_bf16_bits(v4f a)
{
    return __builtin_bit_cast(h4, __builtin_convertvector(a, bf4));
}
template <int TERMS>
__device__ __forceinline__ v16f mfma32(const h8 a, const h8 b, const v16f c)
{
    if (TERMS == -1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <int TERMS>
__device__ __forceinline__ v4f mfma16(const h8 a, const h8 b, const v4f c)
{
    if (TERMS == -1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int TERMS> __device__ __forceinline__ void split4t(v4f a, h4& p0, h4& p1);
__device__ __forceinline__ void split4(v4f a, h4& p0, h4& p1)
{
    // identical to Piece<2>::split of gemm.hip: a1 = fp16(a), a2 = fp16(fma(-a1, 2048, 2048 a)); the empty asm keeps the
    // compiler from fusing a's producer into one of the two conversions (lstm.hip, split_h2)
    asm("" : "+v"(a));
#pragma unroll
    for (int i = 0; i < 4; ++i) p0[i] = (_Float16)a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) p1[i] = (_Float16)__builtin_fmaf(-(float)p0[i], 2048.f, a[i] * 2048.f);
}
template <int TERMS> __device__ __forceinline__ void split4t(v4f a, h4& p0, h4& p1)
{
    if (TERMS == -1) { asm("" : "+v"(a)); p0 = to_bf16_bits(a); p1 = p0; }
    else split4(a, p0, p1);
}

// The body of one workgroup: GR groups of NW = 8 / GR waves; a group owns RT row tiles (compile-time: they size the
// accumulators) and its waves share the feature tiles.  Rows per workgroup = 32 RT GR.
template <int CHAIN, int TERMS, int RT, int GR>
__device__ __forceinline__ void chain_body(const ChainLaunch& g, const ChainDesc* const dp, const int row0, char* const smem_all)
{
    constexpr int NPL = (TERMS == 1 || TERMS == -1) ? 1 : 2;       // pieces per operand
    constexpr int NW = 8 / GR;
    // prefetch depth: register sets of weight fragments in flight (fewer where two row tiles' accumulators take the room)
    constexpr int PDR = RT == 1 ? PD : (PD < 4 ? PD : 4);
    constexpr int CTR = GR == 8 ? 4 : CHAIN_CT;   // feature tiles per wave and layer, at most
    // GR = 8: every wave is a group of its own (one row tile, all feature tiles of the band, layers up to 128 wide): its
    // activation image is private, so the layer barriers are not needed (a wave's LDS accesses are performed in order)
    constexpr bool SOLO = GR == 8;
    constexpr bool RAG = RT == 2 && GR == 1 && !CHAIN_NO_RAG_CODE;      // the geometry that knows how to split a ragged last tile over the k-steps (host: ChainLayer::rag)
    auto group_barrier = [&]() { if (!SOLO) __syncthreads(); };
    float* const sbias = reinterpret_cast<float*>(smem_all + CHAIN_LDS_EX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / NW, wn = wave - grp * NW;
    const int m = lane & 31, h = lane >> 5;
    const int M = g.M;
    int row[RT];
    bool row_ok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rr = row0 + 32 * (grp * RT + r) + m;
        row_ok[r] = rr < M;
        row[r] = row_ok[r] ? rr : M - 1;
    }

    unsigned long long tstamp[4 + 3 * CHAIN_LAYERS];
    int nstamp = 0;
    auto stamp = [&]() { if (CHAIN_TRACE) tstamp[nstamp++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();
    const int plane = dp->plane_units * 512;                     // bytes of one piece of one row tile
    const int img = NPL * plane;                                 // bytes of one row tile's activation image
    char* const smem = smem_all + grp * RT * img;                // this group's row tiles
    float amax = 0.f;

    // weight fragments in flight: register set s holds a k-step of the wave's current tile.  The sets are filled for a layer's
    // first steps while the previous layer is still in its epilogue and barriers (and for layer 0 before the input is staged)
    constexpr int STEP = NPL * 1024;                             // bytes of one k-step of one tile: a fragment per piece
    const gcc wbase = (gcc)dp->wstream;
    const unsigned lane16 = lane * 16;
    h8 w[PDR][NPL];
    auto wload = [&](int set, gcc p) {
#pragma unroll
        for (int pc = 0; pc < NPL; ++pc) w[set][pc] = *(gch8)(p + pc * 1024 + lane16);
    };
    // tiles of this wave in layer l: t = wn + NW c < NTL; its fragment stream (tile by tile, k-step by k-step) starts behind
    // those of the waves before it.  (Wave-uniform base + 32-bit lane offset: saddr + voffset loads.)
    auto layer_stream = [&](int l, int& K16, int& cnt, gcc& wp) {
        K16 = dp->L[l].K16;
        const int NTL = dp->L[l].NTL - dp->L[l].rag;            // whole tiles (a ragged last tile is split over the k-steps, below)
        const int full = NTL / NW, rem = NTL - full * NW;
        cnt = full + (wn < rem ? 1 : 0);
        int before = wn * full + (wn < rem ? wn : rem);
        if (CHAIN_ABL & 8) before = 0;             // measurement only: every wave streams wave 0's fragments (L1-hot for 7 of 8)
        wp = wbase + dp->L[l].w_off + (size_t)before * K16 * STEP;
    };
    // the wave's k-slice [kr0, kr0 + kr) of the layer's ragged last tile and its fragment stream (behind all whole tiles)
    auto rag_stream = [&](int l, int& kr0, int& kr, gcc& rp) {
        const int K16 = dp->L[l].K16, base = K16 / NW, rem = K16 - base * NW;
        kr0 = wn * base + (wn < rem ? wn : rem);
        kr = base + (wn < rem ? 1 : 0);
        rp = wbase + dp->L[l].w_off + ((size_t)(dp->L[l].NTL - 1) * K16 + kr0) * STEP;
    };
    auto prefetch_layer = [&](int l) {
        int K16, cnt; gcc wp;
        layer_stream(l, K16, cnt, wp);
        // a layer with a ragged last tile starts with the wave's k-slice of it (set s: step s of the slice), then its whole tiles
        int kr0 = 0, kr = 0; gcc rp = wbase;
        if (RAG && dp->L[l].rag) rag_stream(l, kr0, kr, rp);
#pragma unroll
        for (int s = 0; s < PDR; ++s) {
            if (s < kr) wload(s, rp + (size_t)s * STEP);
            else if (cnt > 0 && s < K16) wload(s, wp + (size_t)s * STEP);
        }
    };
    prefetch_layer(0);
#if CHAIN_PRIO
    // waves 4-7 are dispatched second and lose every arbitration against their SIMD partner (measured: 33 vs 28 us per
    // layer of the 768-wide band): one static priority raise for that half (cdna guide, T5 static form)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    // ---- biases of the five layers -> LDS; input rows -> LDS (split on the way), several loads in flight per lane
    {
        const gcf bsrc = (gcf)dp->bias;
        const int nb = dp->nbias;
        for (int i = tid; i < nb; i += 512) sbias[i] = bsrc[i];
        const int U0 = 2 * dp->L[0].K16, K0 = dp->K0;
        constexpr int IB = 8 / RT;                               // units per batch and row tile: 8 float4 in flight per lane
        for (int u0 = wn; u0 < U0; u0 += NW * IB) {      // (the group's waves share the units)
            v4f v[RT][IB];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const gcf xin = (gcf)g.Xin + (size_t)row[r] * g.ldx + dp->in_off;
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int k = 8 * (u0 + NW * i) + 4 * h;
                    v[r][i] = (v4f){0.f, 0.f, 0.f, 0.f};
                    if (k < K0) v[r][i] = *(gc4)(xin + k);
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int u = u0 + NW * i;
                    if (u >= U0) continue;
                    const v4f x = v[r][i];
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
                    h4 p0, p1;
                    split4t<TERMS>(x, p0, p1);
                    *reinterpret_cast<h4*>(smem + r * img + u * 512 + m * 16 + 8 * h) = p0;
                    if (NPL == 2) *reinterpret_cast<h4*>(smem + r * img + plane + u * 512 + m * 16 + 8 * h) = p1;
                }
        }
    }
    stamp();
    __syncthreads();
    stamp();

    // one layer; LAST (the chain's fifth layer: global stores instead of an LDS image) is a compile-time property so that the
    // registers of the two kinds of epilogue (held tiles / prefetched residual and multiplier rows) never coexist
    auto layer = [&](auto last_tag, const int l) {
        constexpr bool last = decltype(last_tag)::value;
        // the rows of this workgroup again, behind an opaque asm: the global addresses of the epilogue below are then formed
        // here, per layer - as loop invariants the compiler computed all of them (14 64-bit pointers) once, in front of the
        // layer loop, and kept them in scratch (the only scratch use of the kernel, reloaded every layer)
        int rowl[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) { rowl[r] = row[r]; asm volatile("" : "+v"(rowl[r])); }
        int K16, cnt; gcc wp;
        layer_stream(l, K16, cnt, wp);
        const int boff = dp->L[l].bias_off;
        const bool leaky = dp->L[l].leaky != 0;
        const bool to_p = CHAIN == CHAIN_SPLIT && l == 1;
        const bool rag = RAG && dp->L[l].rag != 0;              // wave-uniform, the same for all waves of the workgroup
        int kr0 = 0, kr = 0; gcc rp = wbase;
        if (rag) rag_stream(l, kr0, kr, rp);

        // The K loop is bound by the CU's fill rate from L2 (~70 GB/s per CU, measured: tools/chain_bench.hip), so every
        // weight fragment a wave loads is multiplied with all RT row tiles of its group.  The wave works through its tiles one after the other: one accumulator pair per row tile, and PDR k-steps of
        // weight fragments in flight.  A finished tile cannot go to the LDS image yet (other waves are still reading the
        // layer's input from it), so it is kept split, as fp16 pieces, until the barrier.  Register set s holds k-step
        // ks0 + s; it is refilled right behind the MFMAs that consumed it, with the step PDR further on, or with the NEXT
        // tile's step s when this tile has no such step.
        v4f* const rscr = reinterpret_cast<v4f*>(smem_all + CHAIN_LDS_EX - CHAIN_RAG_LDS);
        // ---- the ragged last tile (<= 4 real features, e.g. columns 512, 513 of a 514-wide band: 17 tiles for 8 waves): every
        // wave multiplies its k-slice of it (K16 / NW steps instead of one wave running a whole tile of K16) FIRST, while no finished
        // tile occupies registers; the partial sums of the lanes that hold its first four features go to LDS and wave 0 adds them
        // behind the layer's barrier.
        if (rag) {
            v16f hi[RT], lo[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) { hi[r] = (v16f){0}; lo[r] = (v16f){0}; }
            for (int s0 = 0; s0 < kr; s0 += PDR) {
#pragma unroll
                for (int sx = 0; sx < PDR; ++sx) {
                    if (s0 + sx < kr) {
                        const int ks = kr0 + s0 + sx;
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            h8 b[NPL];
#pragma unroll
                            for (int pc = 0; pc < NPL; ++pc) b[pc] = *reinterpret_cast<const h8*>(smem + r * img + pc * plane + ks * 1024 + lane * 16);
                            if (NPL == 2) {
                                lo[r] = mfma32<TERMS>(w[sx][0], b[NPL - 1], lo[r]);
                                lo[r] = mfma32<TERMS>(w[sx][NPL - 1], b[0], lo[r]);
                            }
                            hi[r] = mfma32<TERMS>(w[sx][0], b[0], hi[r]);
                        }
                        if (s0 + sx + PDR < kr) wload(sx, rp + (size_t)(s0 + sx + PDR) * STEP);
                        else if (cnt > 0 && sx < K16) wload(sx, wp + (size_t)sx * STEP);            // the first whole tile comes next
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    v4f pv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[e] = NPL == 1 ? hi[r][e] : hi[r][e] + (1.f / 2048.f) * lo[r][e];
                    rscr[(wn * RT + r) * 32 + m] = pv;
                }
            }
        }
        h4 held[last ? 1 : CTR][RT][4][NPL];
#pragma unroll
        for (int c = 0; c < CTR; ++c) {
            if (c >= cnt) break;
            const int t = wn + NW * c;
            const gcc tp = wp + (size_t)c * K16 * STEP;
            const bool more = c + 1 < cnt;
            // the last layer of the mask chain also needs the residual and the spectrum it multiplies: in flight during the K
            // loop where the registers allow it (RT <= 2)
            constexpr bool PRE_RM = CHAIN == CHAIN_MASK && last && RT <= 2;     // (no held tiles in the last layer)
            v4f rv[PRE_RM ? RT : 1][4], mv[PRE_RM ? RT : 1][4];
            if (PRE_RM) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int n0 = 32 * t + 8 * q + 4 * h;
                        n0 = n0 < dp->a8 ? n0 : 0;
                        rv[r][q] = *(gc4)((gcf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0);
                        mv[r][q] = *(gc4)((gcf)g.Xmul + (size_t)rowl[r] * g.ldm + dp->p_off + n0);
                    }
            }
            v16f hi[RT], lo[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) { hi[r] = (v16f){0}; lo[r] = (v16f){0}; }
            // activation fragments one k-step ahead of the MFMAs where the registers allow it (the fences below would
            // otherwise leave every ds_read right in front of its consumer: one exposed LDS latency per MFMA)
            constexpr bool BAHEAD = RT == 1;
            h8 bn[RT][NPL];
            auto bload = [&](int ks) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int pc = 0; pc < NPL; ++pc) bn[r][pc] = *reinterpret_cast<const h8*>(smem + r * img + pc * plane + ks * 1024 + lane * 16);
            };
            if (BAHEAD) bload(0);
            auto compute = [&](int set, int ks) {
                h8 b[RT][NPL];
                if (BAHEAD) {
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int pc = 0; pc < NPL; ++pc) b[r][pc] = bn[r][pc];
                    if (!(CHAIN_ABL & 4)) bload(ks + 1 < K16 ? ks + 1 : ks);
                } else {
                    // two row tiles' accumulators leave no room for fragments in flight: read them one row tile at a time, right
                    // in front of their MFMAs (the SIMD's other wave covers the LDS latency)
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
#pragma unroll
                        for (int pc = 0; pc < NPL; ++pc) b[r][pc] = *reinterpret_cast<const h8*>(smem + r * img + pc * plane + ks * 1024 + lane * 16);
                        if (r + 1 < RT) __builtin_amdgcn_sched_barrier(0);
                        if (NPL == 2) {
                            lo[r] = mfma32<TERMS>(w[set][0], b[r][NPL - 1], lo[r]);
                            lo[r] = mfma32<TERMS>(w[set][NPL - 1], b[r][0], lo[r]);
                        }
                        hi[r] = mfma32<TERMS>(w[set][0], b[r][0], hi[r]);
                        if (r + 1 < RT) __builtin_amdgcn_sched_barrier(0);
                    }
                    return;
                }
                if (CHAIN_ABL & 1) {
                    asm volatile("" ::"v"(w[set][0]), "v"(w[set][NPL - 1]), "v"(b[0][0]), "v"(b[RT - 1][NPL - 1]));
                    return;
                }
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (NPL == 2) {
                        lo[r] = mfma32<TERMS>(w[set][0], b[r][NPL - 1], lo[r]);      // w1 x2
                        lo[r] = mfma32<TERMS>(w[set][NPL - 1], b[r][0], lo[r]);      // w2 x1
                    }
                    hi[r] = mfma32<TERMS>(w[set][0], b[r][0], hi[r]);                // w1 x1
                }
            };
            int ks0 = 0;
            for (; ks0 + 2 * PDR <= K16; ks0 += PDR) {           // steady state: branch-free
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    // (left alone, hipcc sinks all the loads of an iteration to its end and waits for them at the top of the
                    // next: no run-ahead at all; the fence keeps each refill behind its own step)
                    compute(s, ks0 + s);
                    if (!(CHAIN_ABL & 2)) wload(s, tp + (size_t)(ks0 + s + PDR) * STEP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; ks0 < K16; ks0 += PDR) {                      // the last groups of the tile; refills cross into the next tile
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    const int ks = ks0 + s;
                    if (ks < K16) {
                        compute(s, ks);
                        if (ks + PDR < K16) wload(s, tp + (size_t)(ks + PDR) * STEP);
                        else if (more && s < K16) wload(s, tp + (size_t)(K16 + s) * STEP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }

            // ---- the tile's epilogue: bias, LeakyReLU, then either global stores (last layer; P) or the split pieces
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = 32 * t + 8 * q + 4 * h;       // first of this lane's 4 consecutive features
                    const v4f bv = *reinterpret_cast<const v4f*>(&sbias[boff + n0]);
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sum = NPL == 1 ? hi[r][4 * q + e] : hi[r][4 * q + e] + (1.f / 2048.f) * lo[r][4 * q + e];
                        v[e] = sum + bv[e];
                        if (leaky) v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                    }
                    if (!last) {
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                        h4 p0, p1;
                        split4t<TERMS>(v, p0, p1);
                        held[last ? 0 : c][r][q][0] = p0;
                        held[last ? 0 : c][r][q][NPL - 1] = NPL == 2 ? p1 : p0;
                        if (to_p && row_ok[r] && n0 < dp->a8) *(g4)((gf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0) = v;
                    } else if (CHAIN == CHAIN_SPLIT) {
                        if (row_ok[r] && n0 < HID) *(g4)((gf)g.Z + (size_t)rowl[r] * g.ldz + dp->z_off + n0) = v;
                    } else {
                        if (row_ok[r] && n0 < dp->a8) {
                            const v4f rr = PRE_RM ? rv[PRE_RM ? r : 0][q] : *(gc4)((gcf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0);
                            const v4f mm = PRE_RM ? mv[PRE_RM ? r : 0][q] : *(gc4)((gcf)g.Xmul + (size_t)rowl[r] * g.ldm + dp->p_off + n0);
                            v += rr;                                                 // mask = residual + post(...)   bsrnn.py:425
                            if (g.tap) *(g4)((gf)g.tap + (size_t)rowl[r] * g.ldt + dp->p_off + n0) = v;
                            *(g4)((gf)g.Y + (size_t)rowl[r] * g.ldy + dp->p_off + n0) = v * mm;   // x * mask      bsrnn.py:441
                        }
                    }
                }
        }
        // wave 0 of the group: the ragged tile's features n0 .. n0 + 7 (lanes h = 0: the sum of the partials, h = 1: padding) through
        // the same epilogue as a whole tile's first register quad; the padding up to the next layer's k range is written as zeros
        auto rag_finish = [&]() {
            const int t = dp->L[l].NTL - 1;
            const int n0 = 32 * t + 4 * h;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                v4f v = (v4f){0.f, 0.f, 0.f, 0.f};
                if (h == 0) {
                    for (int w8 = 0; w8 < NW; ++w8) v += rscr[(w8 * RT + r) * 32 + m];
                    const v4f bv = *reinterpret_cast<const v4f*>(&sbias[boff + n0]);
                    v += bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (leaky) v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                }
                if (!last) {
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                    h4 p0, p1;
                    split4t<TERMS>(v, p0, p1);
                    const h4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    char* const d = smem + r * img + (4 * t) * 512 + m * 16 + 8 * h;
                    *reinterpret_cast<h4*>(d) = p0;                       // k-unit 4 t: features n0 .. n0 + 7
                    *reinterpret_cast<h4*>(d + 512) = z4;                 // k-unit 4 t + 1: padding
                    if (NPL == 2) { *reinterpret_cast<h4*>(d + plane) = p1; *reinterpret_cast<h4*>(d + plane + 512) = z4; }
                    if (to_p && row_ok[r] && n0 < dp->a8) *(g4)((gf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0) = v;
                } else if (CHAIN == CHAIN_SPLIT) {
                    if (row_ok[r] && n0 < HID) *(g4)((gf)g.Z + (size_t)rowl[r] * g.ldz + dp->z_off + n0) = v;
                } else {
                    if (row_ok[r] && n0 < dp->a8) {
                        v += *(gc4)((gcf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0);
                        if (g.tap) *(g4)((gf)g.tap + (size_t)rowl[r] * g.ldt + dp->p_off + n0) = v;
                        *(g4)((gf)g.Y + (size_t)rowl[r] * g.ldy + dp->p_off + n0) = v * *(gc4)((gcf)g.Xmul + (size_t)rowl[r] * g.ldm + dp->p_off + n0);
                    }
                }
            }
        };
        // the next layer's first fragments: in flight across the barriers; where two row tiles' held pieces fill the
        // registers (RT = 2), only once those have gone to LDS
        constexpr bool XPRE = RT == 1;
        if (!last && XPRE) prefetch_layer(l + 1);
        stamp();
        if (last) {
            if (rag) {
                __syncthreads();
                if (wn == 0) rag_finish();
            }
            return;
        }
        group_barrier();                                         // every wave has read the layer's input image
        stamp();
        if (rag && wn == 0) rag_finish();
#pragma unroll
        for (int c = 0; c < CTR; ++c) {
            if (c >= cnt) break;
            const int t = wn + NW * c;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int u = 4 * t + q;
                    *reinterpret_cast<h4*>(smem + r * img + u * 512 + m * 16 + 8 * h) = held[last ? 0 : c][r][q][0];
                    if (NPL == 2) *reinterpret_cast<h4*>(smem + r * img + plane + u * 512 + m * 16 + 8 * h) = held[last ? 0 : c][r][q][NPL - 1];
                }
        }
        if (!XPRE) prefetch_layer(l + 1);
        group_barrier();                                         // the next layer's input image is complete
        stamp();
    };
#pragma unroll 1
    for (int l = 0; l < CHAIN_LAYERS - 1; ++l) layer(std::false_type(), l);
    layer(std::true_type(), CHAIN_LAYERS - 1);

    if (CHAIN_TRACE && g.dbg && lane == 0 && blockIdx.x < 64) {
        unsigned long long* d = g.dbg + ((size_t)blockIdx.x * 8 + wave) * 24;
        for (int i = 0; i < nstamp && i < 24; ++i) d[i] = tstamp[i];
    }
    // range guard (see gemm.hip): a finite operand beyond the fp16 range saturated its first piece
    if (TERMS != -1 && amax > 65504.f && g.range_flag) *g.range_flag = 1;      // (bf16 has the range of fp32)
}

// The widest bands (more than 576 columns: the 768-wide band of the 12-band table) on v_mfma_f32_16x16x32_f16: 48 frame
// rows per workgroup (three row tiles of 16) instead of 32 - all the 144 KB of LDS hold of a 768-column image.  These bands
// are bound by the CU's fill rate from L2, every weight fragment is loaded once per workgroup, so rows per workgroup is what
// counts: 1.5x the rows per weight byte, and 168 instead of 252 workgroups of that band (all resident at once on 256 CUs).
//   A = weights:      lane (n = l & 15, kb = l >> 4) holds W[16 t + n][32 ks + 8 kb + j]     (feature tiles of 16)
//   B = activations:  lane (m = l & 15, kb)          holds x[m][32 ks + 8 kb + j]: LDS image [k / 8][48 rows][8], per piece
//   D: lane (m, g = l >> 4), register e: feature 16 t + 4 g + e of row m - four consecutive features, as in chain_body.
// In this geometry ChainLayer::K16 counts k-steps of 32 and ChainLayer::NTL feature tiles of 16.
// RT row tiles of 16 (48 rows: the 768-wide band; 80 rows: bands whose image leaves room for five - the 384-wide band, 24 feature
// tiles = three per wave where the 32 x 32 geometry had twelve tiles for eight waves and 64 rows per weight fragment), CTR = feature
// tiles per wave the held-tile registers are sized for, PDR = k-steps of weight fragments in flight.
// ZPAD: the band's widths are not multiples of 16 / 32 (the 514-wide band on four row tiles): the k-units of the image that no layer's
// output covers but a later layer's K loop reads (against zero weights) are zeroed once - uninitialised LDS may hold NaN patterns.
template <int CHAIN, int TERMS, int RT = 3, int CTR = 6, int PDR = CHAIN_PD48, int TP = 2, bool ZPAD = false>
__device__ __forceinline__ void chain_body48(const ChainLaunch& g, const ChainDesc* const dp, const int row0, char* const smem)
{
    constexpr int NPL = (TERMS == 1 || TERMS == -1) ? 1 : 2;
    constexpr int NW = 8, ROWS = 16 * RT, UB = ROWS * 16;      // UB: bytes of one k-unit (8 k) of one piece
                                                  // (PDR k-steps in flight; a step is TWO feature tiles' fragments, 4 KB per wave)
    // TP: feature tiles per pass over K: each activation fragment read from LDS feeds TP tiles (LDS read bandwidth is the
    // co-bottleneck of 16 x 16 tiles)
    float* const sbias = reinterpret_cast<float*>(smem + CHAIN_LDS_EX);
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));            // this body's own copy (a shared one lives - and at 256 VGPRs spills - across all bodies)
    const int tid = tid_, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, gq = lane >> 4;
    const int M = g.M;
    int row[RT];
    bool row_ok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rr = row0 + 16 * r + m;
        row_ok[r] = rr < M;
        row[r] = row_ok[r] ? rr : M - 1;
    }
    const int plane = dp->plane_units * UB;
    float amax = 0.f;

    constexpr int STEP = NPL * 1024;
    const gcc wbase = (gcc)dp->wstream;
    const unsigned lane16 = lane * 16;
    h8 w[PDR][TP][NPL];
    // k-step ks of the pass whose first tile's stream starts at p; `two`: the pass has a second tile (K32 steps further on)
    auto wload = [&](int set, gcc p, int ks, int K32, bool two) {
#pragma unroll
        for (int tt = 0; tt < TP; ++tt) {
            if (tt == 1 && !two) break;
#pragma unroll
            for (int pc = 0; pc < NPL; ++pc) w[set][tt][pc] = *(gch8)(p + (size_t)(tt * K32 + ks) * STEP + pc * 1024 + lane16);
        }
    };
    auto layer_stream = [&](int l, int& K32, int& cnt, gcc& wp) {
        K32 = dp->L[l].K16;
        const int FT = dp->L[l].NTL;
        const int full = FT / NW, rem = FT - full * NW;
        cnt = full + (wn < rem ? 1 : 0);
        const int before = wn * full + (wn < rem ? wn : rem);
        wp = wbase + dp->L[l].w_off + (size_t)before * K32 * STEP;
    };
    auto prefetch_layer = [&](int l) {
        int K32, cnt; gcc wp;
        layer_stream(l, K32, cnt, wp);
        if (cnt > 0) {
#pragma unroll
            for (int s = 0; s < PDR; ++s)
                if (s < K32) wload(s, wp, s, K32, cnt > 1);
        }
    };
    prefetch_layer(0);

    // ---- biases -> LDS; the 48 input rows -> LDS, split on the way.  A wave instruction covers 16 rows x 2 k-units.
    {
        const gcf bsrc = (gcf)dp->bias;
        const int nb = dp->nbias;
        for (int i = tid; i < nb; i += 512) sbias[i] = bsrc[i];
        const int U0 = 4 * dp->L[0].K16, K0 = dp->K0;             // k-units of the first layer's input
        for (int u0 = 2 * wn; u0 < U0; u0 += 2 * NW * 2) {
            v4f v[RT][2];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const gcf xin = (gcf)g.Xin + (size_t)row[r] * g.ldx + dp->in_off;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = 8 * (u0 + 2 * NW * i + (gq >> 1)) + 4 * (gq & 1);
                    v[r][i] = (v4f){0.f, 0.f, 0.f, 0.f};
                    if (k < K0) v[r][i] = *(gc4)(xin + k);
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int u = u0 + 2 * NW * i + (gq >> 1);
                    if (u >= U0) continue;
                    const v4f x = v[r][i];
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
                    h4 p0, p1;
                    split4t<TERMS>(x, p0, p1);
                    char* const d = smem + u * UB + (16 * r + m) * 16 + 8 * (gq & 1);
                    *reinterpret_cast<h4*>(d) = p0;
                    if (NPL == 2) *reinterpret_cast<h4*>(d + plane) = p1;
                }
        }
        if (ZPAD) {
            const int tail = (dp->plane_units - U0) * UB;                     // bytes behind the staged input, per piece
            for (int o = tid * 16; o < tail; o += 512 * 16) {
                *reinterpret_cast<uint4*>(smem + U0 * UB + o) = make_uint4(0, 0, 0, 0);
                if (NPL == 2) *reinterpret_cast<uint4*>(smem + plane + U0 * UB + o) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    __syncthreads();

    auto layer = [&](auto last_tag, const int l) {
        constexpr bool last = decltype(last_tag)::value;
        // the rows of this workgroup again, behind an opaque asm: the global addresses of the epilogue below are then formed
        // here, per layer - as loop invariants the compiler computed all of them (14 64-bit pointers) once, in front of the
        // layer loop, and kept them in scratch (the only scratch use of the kernel, reloaded every layer)
        // (and recomputed from row0, so that neither the row indices nor their validity live in registers across the layers)
        int rowl[RT];
        bool rokl[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            int rr = row0 + 16 * r + m;
            asm volatile("" : "+v"(rr));
            rokl[r] = rr < M;
            rowl[r] = rokl[r] ? rr : M - 1;
        }
        int K32, cnt; gcc wp;
        layer_stream(l, K32, cnt, wp);
        const int boff = dp->L[l].bias_off;
        const bool leaky = dp->L[l].leaky != 0;
        const bool to_p = CHAIN == CHAIN_SPLIT && l == 1;
        h4 held[last ? 1 : CTR][RT][NPL];
#pragma unroll
        for (int c = 0; c < CTR; c += TP) {
            if (c >= cnt) break;
            const bool two = c + 1 < cnt;                         // wave-uniform: the pass has a second tile
            const gcc tp = wp + (size_t)c * K32 * STEP;
            const bool more = c + TP < cnt, more_two = c + TP + 1 < cnt;
            // the mask chain's last layer adds the residual P and multiplies the spectrum: with three row tiles the rows are requested
            // in front of the K loop; with five their 80 registers do not fit beside the accumulators and the fragments, so they are
            // requested when the K loop is through (its fragment registers are free then)
            constexpr bool PRE_RM = RT <= 3;
            v4f rv[TP][RT], mv[TP][RT];
            auto load_rm = [&]() {
#pragma unroll
                for (int tt = 0; tt < TP; ++tt) {
                    const int n0 = 16 * (wn + NW * (c + tt)) + 4 * gq;
                    const int nn = (n0 < dp->a8 && (tt == 0 || two)) ? n0 : 0;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        rv[tt][r] = *(gc4)((gcf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + nn);
                        mv[tt][r] = *(gc4)((gcf)g.Xmul + (size_t)rowl[r] * g.ldm + dp->p_off + nn);
                    }
                }
            };
            if (CHAIN == CHAIN_MASK && last && PRE_RM) load_rm();
            v4f hi[TP][RT], lo[TP][RT];
#pragma unroll
            for (int tt = 0; tt < TP; ++tt)
#pragma unroll
                for (int r = 0; r < RT; ++r) { hi[tt][r] = (v4f){0.f, 0.f, 0.f, 0.f}; lo[tt][r] = hi[tt][r]; }
            h8 bn[RT][NPL];                                       // activation fragments, one k-step ahead of the MFMAs
            auto bload = [&](int ks) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int pc = 0; pc < NPL; ++pc)
                        bn[r][pc] = *reinterpret_cast<const h8*>(smem + pc * plane + (4 * ks + gq) * UB + (16 * r + m) * 16);
            };
            bload(0);
            auto compute = [&](int set, int ks) {
                h8 (&b)[RT][NPL] = bn;                            // (the next step's fragments are requested behind this step's MFMAs:
                                                                  //  a second register set for them does not fit beside two tiles)
                if (CHAIN_ABL & 1) {
                    asm volatile("" ::"v"(w[set][0][0]), "v"(w[set][TP - 1][NPL - 1]), "v"(b[0][0]), "v"(b[RT - 1][NPL - 1]));
                    if (!(CHAIN_ABL & 4)) bload(ks + 1 < K32 ? ks + 1 : ks);
                    return;
                }
#pragma unroll
                for (int tt = 0; tt < TP; ++tt) {
                    if (tt == 1 && !two) break;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (NPL == 2) {
                            lo[tt][r] = mfma16<TERMS>(w[set][tt][0], b[r][NPL - 1], lo[tt][r]);      // w1 x2
                            lo[tt][r] = mfma16<TERMS>(w[set][tt][NPL - 1], b[r][0], lo[tt][r]);      // w2 x1
                        }
                        hi[tt][r] = mfma16<TERMS>(w[set][tt][0], b[r][0], hi[tt][r]);                // w1 x1
                    }
                }
                if (!(CHAIN_ABL & 4)) bload(ks + 1 < K32 ? ks + 1 : ks);
            };
            int ks0 = 0;
            for (; ks0 + 2 * PDR <= K32; ks0 += PDR) {
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    compute(s, ks0 + s);
                    if (!(CHAIN_ABL & 2)) wload(s, tp, ks0 + s + PDR, K32, two);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; ks0 < K32; ks0 += PDR) {
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    const int ks = ks0 + s;
                    if (ks < K32) {
                        compute(s, ks);
                        if (ks + PDR < K32) wload(s, tp, ks + PDR, K32, two);
                        else if (more && s < K32) wload(s, tp + (size_t)TP * K32 * STEP, s, K32, more_two);      // the next pass
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- the tiles' epilogue
            if (CHAIN == CHAIN_MASK && last && !PRE_RM) load_rm();
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
            if (tt == 1 && !two) break;
            const int n0 = 16 * (wn + NW * (c + tt)) + 4 * gq;    // first of this lane's 4 consecutive features
            const v4f bv = *reinterpret_cast<const v4f*>(&sbias[boff + n0]);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sum = NPL == 1 ? hi[tt][r][e] : hi[tt][r][e] + (1.f / 2048.f) * lo[tt][r][e];
                    v[e] = sum + bv[e];
                    if (leaky) v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                }
                if (!last) {
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                    h4 p0, p1;
                    split4t<TERMS>(v, p0, p1);
                    held[last ? 0 : c + tt][r][0] = p0;
                    held[last ? 0 : c + tt][r][NPL - 1] = NPL == 2 ? p1 : p0;
                    if (to_p && rokl[r] && n0 < dp->a8) *(g4)((gf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0) = v;
                } else if (CHAIN == CHAIN_SPLIT) {
                    if (rokl[r] && n0 < HID) *(g4)((gf)g.Z + (size_t)rowl[r] * g.ldz + dp->z_off + n0) = v;
                } else {
                    if (rokl[r] && n0 < dp->a8) {
                        v += rv[tt][r];                                              // mask = residual + post(...)   bsrnn.py:425
                        if (g.tap) *(g4)((gf)g.tap + (size_t)rowl[r] * g.ldt + dp->p_off + n0) = v;
                        *(g4)((gf)g.Y + (size_t)rowl[r] * g.ldy + dp->p_off + n0) = v * mv[tt][r];   // x * mask   bsrnn.py:441
                    }
                }
            }
            }
        }
        if (last) return;
        prefetch_layer(l + 1);                                   // in flight across the barriers
        __syncthreads();                                         // every wave has read the layer's input image
#pragma unroll
        for (int c = 0; c < CTR; ++c) {
            if (c >= cnt) break;
            const int t = wn + NW * c;
            // features 16 t + 4 gq .. + 3: k-unit 2 t + (gq >> 1), half (gq & 1)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                char* const d = smem + (2 * t + (gq >> 1)) * UB + (16 * r + m) * 16 + 8 * (gq & 1);
                *reinterpret_cast<h4*>(d) = held[last ? 0 : c][r][0];
                if (NPL == 2) *reinterpret_cast<h4*>(d + plane) = held[last ? 0 : c][r][NPL - 1];
            }
        }
        __syncthreads();                                         // the next layer's input image is complete
    };
#pragma unroll 1
    for (int l = 0; l < CHAIN_LAYERS - 1; ++l) layer(std::false_type(), l);
    layer(std::true_type(), CHAIN_LAYERS - 1);
    if (TERMS != -1 && amax > 65504.f && g.range_flag) *g.range_flag = 1;      // (bf16 has the range of fp32)
}

#ifdef CHAIN_PAIR_GEOMETRY
// =====================================================================================
// The widest band as PAIRS of workgroups (round 4): MEASURED AND NOT ADOPTED - compiled only with -DCHAIN_PAIR_GEOMETRY, which only
// tools/chain_bench.hip does (tools/pair_probe.sh; numbers in profiles/r04_pair_geometry.txt, the reasons in DESIGN.md section 4c).  The fused chains are bound by the rate at which the chip moves weight fragments from
// L2 into the CUs (9.3 TB/s aggregate, counter-backed: profiles/r04b_summary.md), and the 768-wide band streams 57 % of them because LDS
// holds the fp16x2 image of only 48 of its rows.  Here two workgroups share 80 rows: workgroup `half` keeps half `half` of K of every
// layer's input (80 x 384 x 4 B = 120 KB), multiplies it with ALL output features - its fragment stream is half the layer's weights -
// and hands the partial sums of the PARTNER's half of the outputs over through L2; the partial sums of its own half it completes with
// the partner's, and their epilogue (bias, LeakyReLU, split) writes exactly the half of K it holds of the next layer.  Weight bytes per row
// of the band: / (80 / 48) -> - 40 %; + 2 x 120 KB of partial sums written and read per layer and pair.
// Hand-over (cdna_hip_programming.md, Guideline 16): partial sums stored write-through (buffer_store_dwordx4 ... sc1), every wave drains,
// workgroup barrier, one lane stores the sender's flag (pair_epoch * 16 + layers handed over, sc1); a receiving wave polls that flag with
// relaxed agent-scope loads (bounded), one agent-scope acquire per wave and layer (the exchange buffers are reused every second layer: this
// CU's L1 may hold their old lines), plain 16-byte loads.  The two workgroups of a pair are adjacent in the task table, so they are
// dispatched together; a partner that never arrives is REPORTED (guard value 6: api.hip runs the call again with the unpaired geometry).
// Geometry: five row tiles of 16 on v_mfma_f32_16x16x32_f16 (the 80-row body's), a wave's tiles = wn + 8 j of each half (three of each
// for 768 features), the partner's half first.
// =====================================================================================
#ifndef CHAIN_PAIR_ABL
#define CHAIN_PAIR_ABL 0      // measurement only: 1 no partial-sum stores, 2 no wait for / loads of the partner's, 4 no fragment loads in the K loop, 8 no LDS reads in it
#endif
#ifndef CHAIN_PAIR_LATE
#define CHAIN_PAIR_LATE 1
#endif
#ifndef CHAIN_PAIR_TPS
#define CHAIN_PAIR_TPS 1      // feature tiles per pass of the BandSplit chain in the paired geometry
#endif
#ifndef CHAIN_PAIR_PD
#define CHAIN_PAIR_PD 2
#endif
#ifndef CHAIN_PAIR_TPM
#define CHAIN_PAIR_TPM 1      // ... of the MaskEstimation chain
#endif
#ifndef CHAIN_PAIR_PDM
#define CHAIN_PAIR_PDM 2
#endif
template <int CHAIN, int TERMS, int PDR = (CHAIN == CHAIN_SPLIT ? CHAIN_PAIR_PD : CHAIN_PAIR_PDM), int TP = (CHAIN == CHAIN_SPLIT ? CHAIN_PAIR_TPS : CHAIN_PAIR_TPM)>
__device__ __forceinline__ void chain_body_pair(const ChainLaunch& g, const ChainDesc* const dp, const int row0, const int half, char* const smem)
{
    constexpr int RT = 5, CTR = 3;
    constexpr int NPL = (TERMS == 1 || TERMS == -1) ? 1 : 2;
    constexpr int NW = 8, ROWS = 16 * RT, UB = ROWS * 16;
    float* const sbias = reinterpret_cast<float*>(smem + CHAIN_LDS_EX);
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, gq = lane >> 4;
    const int M = g.M;
    const int plane = dp->plane_units * UB;
    float amax = 0.f;
    // the pair's exchange buffers and flags
    const int pid = dp->pair_base + row0 / ROWS;
    float* const ex_send = g.exch + (size_t)(pid * 2 + half) * 2 * (ROWS * PAIR_NH);
    float* const ex_recv = g.exch + (size_t)(pid * 2 + (half ^ 1)) * 2 * (ROWS * PAIR_NH);
    typedef int __attribute__((address_space(1)))* gi;
    const gi flag_send = (gi)(g.pflags + (pid * 2 + half) * NW + wn), flag_recv = (gi)(g.pflags + (pid * 2 + (half ^ 1)) * NW + wn);      // one per wave
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;               // HW_REG_XCC_ID[3:0]
    const __amdgpu_buffer_rsrc_t send_rsrc = __builtin_amdgcn_make_buffer_rsrc(ex_send, 0, (int)(2 * ROWS * PAIR_NH * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t recv_rsrc = __builtin_amdgcn_make_buffer_rsrc(ex_recv, 0, (int)(2 * ROWS * PAIR_NH * sizeof(float)), 0x00020000);
    bool timed_out = false, misplaced = false;

    constexpr int STEP = NPL * 1024;
    const gcc wbase = (gcc)dp->wstream;
    const unsigned lane16 = lane * 16;
    h8 w[PDR][TP][NPL];
    auto wload = [&](int set, gcc p, int ks, int K32, bool two) {
#pragma unroll
        for (int tt = 0; tt < TP; ++tt) {
            if (tt == 1 && !two) break;
#pragma unroll
            for (int pc = 0; pc < NPL; ++pc) w[set][tt][pc] = *(gch8)(p + (size_t)(tt * K32 + ks) * STEP + pc * 1024 + lane16);
        }
    };
    // this wave's tiles of a layer: cnt of the partner's half, then cnt of its own; its stream starts behind those of the waves before it
    auto layer_stream = [&](int l, int& K32, int& cnt, gcc& wp) {
        K32 = dp->L[l].K16;
        const int FTh = dp->L[l].NTL >> 1;
        cnt = wn < FTh ? (FTh - wn + NW - 1) / NW : 0;
        int before = 0;
        for (int w8 = 0; w8 < wn; ++w8) before += w8 < FTh ? 2 * ((FTh - w8 + NW - 1) / NW) : 0;
        wp = wbase + (half ? dp->L[l].w_off1 : dp->L[l].w_off) + (size_t)before * K32 * STEP;
    };
    auto prefetch_layer = [&](int l) {
        int K32, cnt; gcc wp;
        layer_stream(l, K32, cnt, wp);
        if (cnt > 0) {
#pragma unroll
            for (int s = 0; s < PDR; ++s)
                if (s < K32) wload(s, wp, s, K32, TP > 1 && cnt > 1);
        }
    };
    prefetch_layer(0);

    // ---- biases -> LDS; this workgroup's half of the input columns of the 80 rows -> LDS, split on the way
    {
        const gcf bsrc = (gcf)dp->bias;
        const int nb = dp->nbias;
        for (int i = tid; i < nb; i += 512) sbias[i] = bsrc[i];
        const int U0 = 4 * dp->L[0].K16, K0h = dp->K0 >> 1;
        for (int u0 = 2 * wn; u0 < U0; u0 += 2 * NW * 2) {
            v4f v[RT][2];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                int rr = row0 + 16 * r + m;
                rr = rr < M ? rr : M - 1;
                const gcf xin = (gcf)g.Xin + (size_t)rr * g.ldx + dp->in_off + half * K0h;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = 8 * (u0 + 2 * NW * i + (gq >> 1)) + 4 * (gq & 1);
                    v[r][i] = (v4f){0.f, 0.f, 0.f, 0.f};
                    if (k < K0h) v[r][i] = *(gc4)(xin + k);
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int u = u0 + 2 * NW * i + (gq >> 1);
                    if (u >= U0) continue;
                    const v4f x = v[r][i];
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
                    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
                    h4 p0, p1;
                    split4t<TERMS>(x, p0, p1);
                    char* const d = smem + u * UB + (16 * r + m) * 16 + 8 * (gq & 1);
                    *reinterpret_cast<h4*>(d) = p0;
                    if (NPL == 2) *reinterpret_cast<h4*>(d + plane) = p1;
                }
        }
    }
    __syncthreads();

    auto layer = [&](auto last_tag, const int l) {
        constexpr bool last = decltype(last_tag)::value;
        int rowl[RT];
        bool rokl[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            int rr = row0 + 16 * r + m;
            asm volatile("" : "+v"(rr));
            rokl[r] = rr < M;
            rowl[r] = rokl[r] ? rr : M - 1;
        }
        int K32, cnt; gcc wp;
        layer_stream(l, K32, cnt, wp);
        const int FTh = dp->L[l].NTL >> 1;
        const int boff = dp->L[l].bias_off;
        const bool leaky = dp->L[l].leaky != 0;
        const bool to_p = CHAIN == CHAIN_SPLIT && l == 1;
        const int par = l & 1;
        int mx = m, gx = gq;                                     // (opaque per layer: the exchange / epilogue offsets are not worth registers across the K loops)
        asm volatile("" : "+v"(mx), "+v"(gx));
        h4 held[last ? 1 : CTR][RT][NPL];
        v4f hi[TP][RT], lo[TP][RT];
        // one pass over this half of K for the wave's tiles c, c + 1 of its stream (c counts the partner's tiles first, then its own)
        auto kloop = [&](const int c, const bool two, const bool more, const bool more_two) {
            const gcc tp = wp + (size_t)c * K32 * STEP;
#pragma unroll
            for (int tt = 0; tt < TP; ++tt)
#pragma unroll
                for (int r = 0; r < RT; ++r) { hi[tt][r] = (v4f){0.f, 0.f, 0.f, 0.f}; lo[tt][r] = hi[tt][r]; }
            h8 bn[RT][NPL];
            auto bload = [&](int ks) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int pc = 0; pc < NPL; ++pc)
                        bn[r][pc] = *reinterpret_cast<const h8*>(smem + pc * plane + (4 * ks + gq) * UB + (16 * r + m) * 16);
            };
            bload(0);
            auto compute = [&](int set, int ks) {
#pragma unroll
                for (int tt = 0; tt < TP; ++tt) {
                    if (tt == 1 && !two) break;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (NPL == 2) {
                            lo[tt][r] = mfma16<TERMS>(w[set][tt][0], bn[r][NPL - 1], lo[tt][r]);
                            lo[tt][r] = mfma16<TERMS>(w[set][tt][NPL - 1], bn[r][0], lo[tt][r]);
                        }
                        hi[tt][r] = mfma16<TERMS>(w[set][tt][0], bn[r][0], hi[tt][r]);
                    }
                }
                if (!(CHAIN_PAIR_ABL & 8)) bload(ks + 1 < K32 ? ks + 1 : ks);
            };
            int ks0 = 0;
            for (; ks0 + 2 * PDR <= K32; ks0 += PDR) {
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    compute(s, ks0 + s);
                    if (!(CHAIN_PAIR_ABL & 4)) wload(s, tp, ks0 + s + PDR, K32, two);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; ks0 < K32; ks0 += PDR) {
#pragma unroll
                for (int s = 0; s < PDR; ++s) {
                    const int ks = ks0 + s;
                    if (ks < K32) {
                        compute(s, ks);
                        if (ks + PDR < K32) wload(s, tp, ks + PDR, K32, two);
                        else if (more && s < K32) wload(s, tp + (size_t)(two ? 2 : 1) * K32 * STEP, s, K32, more_two);      // the next pass
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (NPL == 2) {
#pragma unroll
                for (int tt = 0; tt < TP; ++tt)
#pragma unroll
                    for (int r = 0; r < RT; ++r) hi[tt][r] += (1.f / 2048.f) * lo[tt][r];
            }
        };
        // ---- the partner's half of the outputs: raw partial sums -> the exchange buffer (write-through), rows as they come
#pragma unroll 1
        for (int c = 0; c < cnt; c += TP) {
            const bool two = TP > 1 && c + 1 < cnt;
            const int cn = c + (two ? 2 : 1);                                    // the next pass: more of the partner's tiles, or the first of its own
            kloop(c, two, cnt > 0, TP > 1 && cn + 1 < 2 * cnt && (cn + 1 < cnt || cn >= cnt));
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                if (tt == 1 && !two) break;
                const int fl = 16 * (wn + NW * (c + tt)) + 4 * gx;               // feature inside the receiver's half
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (!(CHAIN_PAIR_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4x, hi[tt][r]), send_rsrc, ((par * ROWS + 16 * r + mx) * PAIR_NH + fl) * 4, 0, 0);
                }
            }
        }
        // wave wn's partial sums are what the partner's wave wn completes: the hand-over is wave to wave.  Through the XCD's L2, which both
        // workgroups share (lstm.hip::band_pair_h2_kernel): a store is counted out of vmcnt when L2 has it; the flag carries the XCC id, so
        // that a placement which breaks the assumption is reported.
        auto hand_over = [&]() {
        if (cnt > 0 && !(CHAIN_PAIR_ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flag_send, ((g.pair_epoch * 16 + l + 1) << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = g.pair_epoch * 16 + l + 1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int v;
            while (((v = __hip_atomic_load(flag_recv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) - want < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)g.pair_spin) { timed_out = true; break; }
            }
            if (!timed_out && (v & 15) != xcc) misplaced = true;                  // the partner sits on another XCD
            asm volatile("buffer_inv sc0" ::: "memory");                         // the partner's sums from L2, not from a line this CU's L1 may hold (the buffers are reused every second layer)
        }
        };
        if (!CHAIN_PAIR_LATE) hand_over();
        // ---- its own half: complete the partial sums with the partner's (requested in front of the K loop), then the usual epilogue
#pragma unroll
        for (int c0 = 0; c0 < CTR; c0 += TP) {
            if (c0 >= cnt) break;
            const int c = cnt + c0;
            const bool two = TP > 1 && c0 + 1 < cnt;
            const bool more = c0 + (two ? 2 : 1) < cnt;
            v4f pp[TP][RT];
            auto load_pp = [&]() {
#pragma unroll
                for (int tt = 0; tt < TP; ++tt) {
                    if (tt == 1 && !two) break;
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        pp[tt][r] = (CHAIN_PAIR_ABL & 2) ? (v4f){0.f, 0.f, 0.f, 0.f}
                            : __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(recv_rsrc, ((par * ROWS + 16 * r + mx) * PAIR_NH + 16 * (wn + NW * (c0 + tt)) + 4 * gx) * 4, 0, 0));
                }
            };
            // CHAIN_PAIR_LATE: the hand-over (drain, flag, wait for the partner's) behind the K loop of the first pass of the own half - the
            // stores have long been counted out then and the partner's flag is there; its sums for this pass are requested late, once
            if (!CHAIN_PAIR_LATE || c0 > 0) load_pp();
            kloop(c, two, more, TP > 1 && c0 + (two ? 2 : 1) + 1 < cnt);
            if (CHAIN_PAIR_LATE && c0 == 0) { hand_over(); load_pp(); }
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                if (tt == 1 && !two) break;
                const int tl = wn + NW * (c0 + tt);                              // tile inside this half
                const int n0 = 16 * (half * FTh + tl) + 4 * gx;                  // first of this lane's 4 consecutive features
                const v4f bv = *reinterpret_cast<const v4f*>(&sbias[boff + n0]);
                v4f rv[RT], mv[RT];                                              // the mask chain's last layer: residual P and the spectrum
                if (CHAIN == CHAIN_MASK && last) {
                    const int nn = n0 < dp->a8 ? n0 : 0;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        rv[r] = *(gc4)((gcf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + nn);
                        mv[r] = *(gc4)((gcf)g.Xmul + (size_t)rowl[r] * g.ldm + dp->p_off + nn);
                    }
                }
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float own = hi[tt][r][e];
                        // (lower half of K + upper half of K, whichever workgroup adds: both halves of a pair produce the same bits)
                        v[e] = (half ? pp[tt][r][e] + own : own + pp[tt][r][e]) + bv[e];
                        if (leaky) v[e] = v[e] >= 0.f ? v[e] : 0.01f * v[e];
                    }
                    if (!last) {
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
                        h4 p0, p1;
                        split4t<TERMS>(v, p0, p1);
                        held[last ? 0 : c0 + tt][r][0] = p0;
                        held[last ? 0 : c0 + tt][r][NPL - 1] = NPL == 2 ? p1 : p0;
                        if (to_p && rokl[r] && n0 < dp->a8) *(g4)((gf)g.P + (size_t)rowl[r] * g.ldp + dp->p_off + n0) = v;
                    } else if (CHAIN == CHAIN_SPLIT) {
                        if (rokl[r] && n0 < HID) *(g4)((gf)g.Z + (size_t)rowl[r] * g.ldz + dp->z_off + n0) = v;
                    } else {
                        if (rokl[r] && n0 < dp->a8) {
                            v += rv[r];
                            if (g.tap) *(g4)((gf)g.tap + (size_t)rowl[r] * g.ldt + dp->p_off + n0) = v;
                            *(g4)((gf)g.Y + (size_t)rowl[r] * g.ldy + dp->p_off + n0) = v * mv[r];
                        }
                    }
                }
            }
        }
        if (last) return;
        prefetch_layer(l + 1);
        __syncthreads();                                         // every wave has read the layer's input image
#pragma unroll
        for (int c = 0; c < CTR; ++c) {
            if (c >= cnt) break;
            const int tl = wn + NW * c;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                char* const d = smem + (2 * tl + (gx >> 1)) * UB + (16 * r + mx) * 16 + 8 * (gx & 1);
                *reinterpret_cast<h4*>(d) = held[last ? 0 : c][r][0];
                if (NPL == 2) *reinterpret_cast<h4*>(d + plane) = held[last ? 0 : c][r][NPL - 1];
            }
        }
        __syncthreads();                                         // the next layer's input image (this half of its K) is complete
    };
#pragma unroll 1
    for (int l = 0; l < CHAIN_LAYERS - 1; ++l) layer(std::false_type(), l);
    layer(std::true_type(), CHAIN_LAYERS - 1);
    if (TERMS != -1 && amax > 65504.f && g.range_flag) *g.range_flag = 1;
    if ((timed_out || misplaced) && g.range_flag) *g.range_flag = timed_out ? 6 : 7;      // the partner never arrived / sits on another XCD: reported, the numbers are not used
}

#endif      // CHAIN_PAIR_GEOMETRY

template <int CHAIN, int TERMS>
__global__ __launch_bounds__(512, 2) void mlp_chain_kernel(ChainLaunch g)
{
    __shared__ __attribute__((aligned(16))) char smem[CHAIN_LDS_EX + CHAIN_LDS_BIAS];

    const int2 task = g.tasks[blockIdx.x];
    const int di = task.x & 0xffffff, row0 = task.y;      // (bits 24+: the probe-only paired geometry's half)
    const ChainDesc* const dp = g.desc + di;
    if (CHAIN == CHAIN_MASK && g.ovl_prog) {
        // launched beside the time-axis launch that writes the chain's input (kernels.h, OvlConsumer): wait until the frames of this
        // workgroup's rows of this band have left it
        __shared__ int ovl_ok;
        if (threadIdx.x == 0) {
            const int rows = chain_rows(*dp), m_last = row0 + rows - 1 < g.M ? row0 + rows - 1 : g.M - 1, band = dp->z_off / HID;
            ovl_ok = ovl_wait_rows(g.ovl_prog, row0, m_last, g.ovl_T, g.ovl_K, band, band, g.ovl_spin, g.ovl_base, g.ovl_wg_shift) ? 1 : 0;
        }
        __syncthreads();
        if (!ovl_ok) {                            // (value 5: api.hip runs the call again launch after launch and stops overlapping)
            if (threadIdx.x == 0 && g.range_flag) *g.range_flag = 5;
            return;
        }
    }
    if (dp->constant) {
        // TrainableConstantModule (bsrnn.py:12-24): the zero-width band's feature is one learned vector for every frame
        if (CHAIN == CHAIN_SPLIT) {
            const gcf cst = (gcf)dp->bias;
            for (int i = threadIdx.x; i < 256 * 16; i += 512) {  // 256 rows x 16 float4
                const int r = row0 + (i >> 4), c4 = i & 15;
                if (r < g.M) *(g4)((gf)g.Z + (size_t)r * g.ldz + dp->z_off + 4 * c4) = *(gc4)(cst + 4 * c4);
            }
        }
        return;
    }
    // geometry of the workgroup (wave-uniform): 48 rows on 16 x 16 tiles, or rows = 32 RT GR on 32 x 32 tiles
    const int RT = dp->RT, GR = 8 / dp->NW;
#ifdef CHAIN_ONLY_BODY            // measurement only: compile ONE geometry (register / scratch use per body: tools/kernel_resources.py)
#ifdef CHAIN_PAIR_GEOMETRY
    if (CHAIN_ONLY_BODY == 7) { chain_body_pair<CHAIN, TERMS>(g, dp, row0, (task.x >> 24) & 1, smem); return; }
#endif
    if (CHAIN_ONLY_BODY == 6) chain_body48<CHAIN, TERMS, 4, 5, CHAIN_PD64, CHAIN_TP64, true>(g, dp, row0, smem);
    else if (CHAIN_ONLY_BODY == 0) chain_body48<CHAIN, TERMS>(g, dp, row0, smem);
    else if (CHAIN_ONLY_BODY == 1) chain_body<CHAIN, TERMS, 1, 8>(g, dp, row0, smem);
    else if (CHAIN_ONLY_BODY == 2) chain_body<CHAIN, TERMS, 1, 1>(g, dp, row0, smem);
    else if (CHAIN_ONLY_BODY == 3) chain_body<CHAIN, TERMS, 2, 1>(g, dp, row0, smem);
    else if (CHAIN_ONLY_BODY == 4) chain_body<CHAIN, TERMS, 2, 2>(g, dp, row0, smem);
    else chain_body<CHAIN, TERMS, 1, 4>(g, dp, row0, smem);
    (void)RT; (void)GR;
    return;
#endif
    // (every geometry ends in `return`: with an else-if chain and one join the structurizer lays the bodies out one behind the other
    //  and keeps values of the later ones - the thread index, for one - alive through the earlier ones' loops: spills at 256 VGPRs)
#ifdef CHAIN_PAIR_GEOMETRY
    if (dp->pair) { chain_body_pair<CHAIN, TERMS>(g, dp, row0, (task.x >> 24) & 1, smem); return; }
#endif
    if (RT == 3) { chain_body48<CHAIN, TERMS>(g, dp, row0, smem); return; }
    if (RT == 5) { chain_body48<CHAIN, TERMS, 5, 3, 2, (CHAIN == CHAIN_SPLIT ? 1 : 2)>(g, dp, row0, smem); return; }
    if (RT == 4) { chain_body48<CHAIN, TERMS, 4, 5, CHAIN_PD64, CHAIN_TP64, true>(g, dp, row0, smem); return; }
    if (GR == 8) { chain_body<CHAIN, TERMS, 1, 8>(g, dp, row0, smem); return; }
    if (RT == 1 && GR == 1) { chain_body<CHAIN, TERMS, 1, 1>(g, dp, row0, smem); return; }
    if (RT == 2 && GR == 1) { chain_body<CHAIN, TERMS, 2, 1>(g, dp, row0, smem); return; }
    if (RT == 2 && GR == 2) { chain_body<CHAIN, TERMS, 2, 2>(g, dp, row0, smem); return; }
    chain_body<CHAIN, TERMS, 1, 4>(g, dp, row0, smem);
}

void launch_mlp_chain(const ChainLaunch& g, int chain, hipStream_t stream)
{
    if (g.M <= 0 || g.n_tasks <= 0) return;
    dim3 grid(g.n_tasks), block(512);
    const int mode = gemm_mode();
    if (chain == CHAIN_SPLIT) {
        if (mode == GEMM_FP16) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_SPLIT, 1>), grid, block, 0, stream, g);
        else if (mode == GEMM_BF16) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_SPLIT, -1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_SPLIT, 3>), grid, block, 0, stream, g);
    } else {
        if (mode == GEMM_FP16) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_MASK, 1>), grid, block, 0, stream, g);
        else if (mode == GEMM_BF16) hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_MASK, -1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((mlp_chain_kernel<CHAIN_MASK, 3>), grid, block, 0, stream, g);
    }
}

}  // namespace bsrnn
