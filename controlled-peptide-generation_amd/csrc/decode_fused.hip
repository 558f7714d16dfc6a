// Whole-loop decoding for small decoders (hidden <= 128, vocab <= 32) as ONE persistent launch:
//   cpg_decode_greedy_fused   RNN_VAE.sample_G(sample_mode='greedy')   models/model.py:225-385
//   cpg_decode_beam_fused     RNN_VAE.sample_G(sample_mode='beam')     models/model.py:258-276,314-328,364-376,387-404;
//                             Beam.advance                             models/Beam.py:56-105
// both over GRUDecoder.forward_sample, models/decoder.py:86-109.
//
// The reference's default decoder (h = z+c = 102, vocab 24) is far too small for one-launch-per-step to be efficient:
// W_hh is 125 KB, a step over N rows re-reads h [N,H] and the constant input term rowc [N,3H] from HBM and writes h back,
// 25 times.  Here a workgroup owns a tile of 64 decoder rows for all T steps and nothing but the chosen tokens (and the
// beam back-pointers) leaves the CU:
//   * W_hh lives in REGISTERS as f32 MFMA B-fragments: wave w owns hidden units [32w, 32w+32) of all three gates
//     (6 column tiles x K/4 k-steps = 156 VGPRs at H=102), so the r/z/n pre-activations of a unit meet in one lane;
//   * the hidden state tile [64][H], the per-sample constant term rowc, the token table tab[V][3H] (= W_ih[:, :E] . emb,
//     models/decoder.py:67,92) and fc [V][H] are staged in LDS once per tile;
//   * per step: h . W_hh^T on the matrix cores (A fragments read 16 B per lane from LDS with the contraction index
//     permuted as in gemm_core.h), the GRU cell in the accumulator layout, h' back to LDS, the vocabulary projection as
//     a second small MFMA product, then the token selection:
//       greedy - first-max argmax + finished/EOS bookkeeping, 16 lanes per wave;
//       beam   - a tile holds whole sentences (rows = sentence-major, K beams each, so rowc is stored once per sentence):
//                one lane per row does log_softmax + its row's K best, one lane per sentence merges them (Beam.advance),
//                and the hidden rows are re-gathered by back-pointer inside LDS.
// One workgroup (4 waves, 1 per SIMD) per CU; the launch is persistent over tiles.
#include "cpg_internal.h"
#include "gemm_core.h"
#include <stdio.h>

// Product engine of the two fused kernels.  1 (default, round 4): f32-grade on f16 pairs (gemm_core.h: split2h_pair) - W_hh lives in
// registers as f16-pair MFMA B-fragments times a power of two chosen from the wave's own rows (the register count of the f32 fragments it replaces), the A fragments (state
// rows from LDS) and the fc rows are split when they are read, THREE v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block in place of
// eight v_mfma_f32_16x16x4_f32 (48-61 matrix-pipe cycles against 268; products exact in the f32 accumulator, dropped lo x lo term
// <= 2^-22 of a product: closer to the exact sums than a k-ordered f32 fma chain, tests/test_gpu_persistent.py).  0: exact-f32 MFMA.
#ifndef CPG_FUSED_PAIR
#define CPG_FUSED_PAIR 1
#endif

// Diagnostic builds of the beam kernel (results wrong): 1 no product, 2 no cell, 4 no vocabulary projection, 8 no stage 1 (row
// lists), 16 no stage 2 (sentence merge), 32 no re-gather
#ifndef CPG_BEAM_ABLATE
#define CPG_BEAM_ABLATE 0
#endif

namespace {

constexpr int RM = 64;  // decoder rows per workgroup tile
constexpr int MT = RM / 16;
constexpr int LGS = 33;  // logits row stride in LDS (floats)
constexpr int MAXK = 8;  // beam width limit (CPG_MAX_BEAM of decode.hip)

// K = 16*G + (up to 4*R) contraction steps: G full 16-deep groups (one ds_read_b128 per lane feeds four MFMA k-steps,
// lane group q supplies k = 16g + 4q + j at step j) + R plain k-steps for the tail (k = 16G + 4r + q).
template <int G, int R>
struct FusedCfg {
    static constexpr int KSTEPS = 4 * G + R;
    static constexpr int KP = 16 * G + 4 * R;
    // f16-pair product: KB 32-deep k-blocks (v_mfma_f32_16x16x32_f16) + KT 16-deep ones for the tail (v_mfma_f32_16x16x16_f16: half
    // the fragment registers of a padded 32-deep block - at H = 102 the tail is 8 k); k >= KP reads as zero
    static constexpr int KB = KP / 32;
    static constexpr int KT = (KP % 32 + 15) / 16;
    // row stride = 4 * odd: the 8 rows one ds_read_b128 lane group touches land in disjoint bank quads
    static constexpr int LDH = ((KP / 4 + 1) % 2 == 1) ? KP + 4 : KP + 8;
    // hidden units per wave: wave w owns [UPW*w, UPW*(w+1)); column tile `sub` holds its units 16*sub .. 16*sub+15, lanes
    // past UPW are idle (their W_hh fragments are zero and nothing is stored), so every guard is a lane predicate
    static constexpr int UPW = KP / 4;
};

struct DecoderWeights {
    const float* tab;   // [Vt,3H] W_ih[:, :E] . emb[tok]
    const float* w_hh;  // [3H,H]
    const float* b_hh;  // [3H]
    const float* fc_w;  // [V,H]
    const float* fc_b;  // [V]
    int H, V, Vt;
};

template <int G, int R>
__device__ __forceinline__ int k_of_step(int s, int lq) {
    return s < 4 * G ? 16 * (s >> 2) + 4 * lq + (s & 3) : 16 * G + 4 * (s - 4 * G) + lq;
}

// Per-lane constants of one wave: its W_hh fragments and biases.
template <int G, int R>
struct WaveWeights {
#if CPG_FUSED_PAIR
    cpg_f16x8 Bh[FusedCfg<G, R>::KB][6], Bl[FusedCfg<G, R>::KB][6];   // W_hh x 2^e_w as f16 pairs: lane (unit l15, k = 32 kb + 8 lq + i)
    cpg_f16x4 Bth[FusedCfg<G, R>::KT > 0 ? FusedCfg<G, R>::KT : 1][6], Btl[FusedCfg<G, R>::KT > 0 ? FusedCfg<G, R>::KT : 1][6];   // tail: k = 32 KB + 16 kt + 4 lq + i
#else
    float Bf[FusedCfg<G, R>::KSTEPS][6];
#endif
    float bh[6];
    float ownf[2];  // 1 where this lane's unit of column tile `sub` is a real hidden unit, else 0
    int ucl[2];     // that unit clamped into [0,H): the lane computes a throw-away duplicate instead of branching
    int col[2];     // LDS column h' goes to: the unit, or one of the row's padding columns [KP, LDH) for idle lanes
    int vclamp0, vclamp1;
    float fcb0, fcb1;
    // f16-pair build: powers of two of the weights' images, chosen from their largest magnitudes (gemm_core.h: weight_exp_of - any
    // finite weight is covered; rounds 4-5 used a fixed 2^8).  wback = 2^-e of THIS wave's W_hh rows (its own accumulator columns);
    // fsc / fback = 2^(+-e) of the vocabulary projection's weights, which stage_tables applies when it fills the LDS copy.
    float wback, fsc, fback;

    __device__ __forceinline__ void load(const DecoderWeights& w) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
        using C = FusedCfg<G, R>;
        wback = fsc = fback = 1.f;
#if CPG_FUSED_PAIR
        float wsc = 1.f;
        {
            float m = 0.f;
            for (int g = 0; g < 3; ++g)
                for (int u = 0; u < C::UPW; ++u) {
                    const int unit = C::UPW * wave + u;
                    if (unit >= w.H) break;
                    const float* row = w.w_hh + (size_t)(g * w.H + unit) * w.H;
                    for (int k = lane; k < w.H; k += 64) m = fmaxf(m, fabsf(row[k]));
                }
            const int e = weight_exp_of(wave_max(m));
            wsc = pair_pow2(e);
            wback = pair_pow2(-e);
            if (w.fc_w && w.V > 0) {
                float fm = 0.f;
                for (int i = lane; i < w.V * w.H; i += 64) fm = fmaxf(fm, fabsf(w.fc_w[i]));
                const int ef = weight_exp_of(wave_max(fm));
                fsc = pair_pow2(ef);
                fback = pair_pow2(-ef);
            }
        }
#endif
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const int g = nt >> 1, ul = 16 * (nt & 1) + l15, unit = C::UPW * wave + ul;
            const bool own = ul < C::UPW && unit < w.H;
            bh[nt] = own ? w.b_hh[g * w.H + unit] : 0.f;
#if CPG_FUSED_PAIR
#pragma unroll
            for (int kb = 0; kb < C::KB; ++kb) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 32 * kb + 8 * lq + 2 * i;
                    const float x0 = (own && k < w.H) ? w.w_hh[(size_t)(g * w.H + unit) * w.H + k] * wsc : 0.f;
                    const float x1 = (own && k + 1 < w.H) ? w.w_hh[(size_t)(g * w.H + unit) * w.H + k + 1] * wsc : 0.f;
                    split2h_pair(x0, x1, hi[i], lo[i]);
                }
                Bh[kb][nt] = __builtin_bit_cast(cpg_f16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
                Bl[kb][nt] = __builtin_bit_cast(cpg_f16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
            }
#pragma unroll
            for (int kt = 0; kt < C::KT; ++kt) {
                uint32_t hi[2], lo[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = 32 * C::KB + 16 * kt + 4 * lq + 2 * i;
                    const float x0 = (own && k < w.H) ? w.w_hh[(size_t)(g * w.H + unit) * w.H + k] * wsc : 0.f;
                    const float x1 = (own && k + 1 < w.H) ? w.w_hh[(size_t)(g * w.H + unit) * w.H + k + 1] * wsc : 0.f;
                    split2h_pair(x0, x1, hi[i], lo[i]);
                }
                Bth[kt][nt] = __builtin_bit_cast(cpg_f16x4, make_uint2(hi[0], hi[1]));
                Btl[kt][nt] = __builtin_bit_cast(cpg_f16x4, make_uint2(lo[0], lo[1]));
            }
#else
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s) {
                const int k = k_of_step<G, R>(s, lq);
                Bf[s][nt] = (own && k < w.H) ? w.w_hh[(size_t)(g * w.H + unit) * w.H + k] : 0.f;
            }
#endif
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ul = 16 * sub + l15, unit = C::UPW * wave + ul;
            ownf[sub] = (ul < C::UPW && unit < w.H) ? 1.f : 0.f;
            ucl[sub] = min(unit, w.H - 1);
            col[sub] = ul < C::UPW ? unit : C::KP + (l15 & 3);
        }
        vclamp0 = min(l15, w.V - 1);
        vclamp1 = min(16 + l15, w.V - 1);
        fcb0 = l15 < w.V ? w.fc_b[l15] : 0.f;
        fcb1 = 16 + l15 < w.V ? w.fc_b[16 + l15] : 0.f;
    }
};

// tab -> LDS, fc_w x fsc -> LDS rows of stride LDH (zero padded in k; fsc = WaveWeights::fsc, the power of two of the f16-pair build)
template <int G, int R>
__device__ __forceinline__ void stage_tables(const DecoderWeights& w, float* tab_l, float* fc_l, float fsc) {
    using C = FusedCfg<G, R>;
    for (int i = threadIdx.x; i < w.Vt * 3 * w.H; i += 256) tab_l[i] = w.tab[i];
    for (int i = threadIdx.x; i < w.V * C::LDH; i += 256) {
        const int v = i / C::LDH, k = i - v * C::LDH;
        fc_l[i] = k < w.H ? w.fc_w[(size_t)v * w.H + k] * fsc : 0.f;
    }
}

// Eight consecutive k (32 kb + 8 lq ..) of one LDS row of stride LDH as an f16-pair fragment; k >= KP reads as zero (the row's
// padding columns [KP, LDH) hold zeros, so a chunk that straddles KP is covered; whole chunks past it are masked).  scale: a power of two.
template <int KP>
__device__ __forceinline__ void pair_frag(const float* row, int kb, int lq, float scale, cpg_f16x8& hi, cpg_f16x8& lo) {
    const int k0 = 32 * kb + 8 * lq;
    const bool in = k0 < KP;
    const float* p = row + (in ? k0 : 0);
    f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4);
    const float f = in ? scale : 0.f;
    x0 *= f;
    x1 *= f;
    uint32_t h[4], l[4];
    split2h_pair(x0[0], x0[1], h[0], l[0]);
    split2h_pair(x0[2], x0[3], h[1], l[1]);
    split2h_pair(x1[0], x1[1], h[2], l[2]);
    split2h_pair(x1[2], x1[3], h[3], l[3]);
    hi = __builtin_bit_cast(cpg_f16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(cpg_f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// the same for a 16-deep tail block: four consecutive k (k0 = 32 KB + 16 kt + 4 lq)
template <int KP>
__device__ __forceinline__ void pair_frag16(const float* row, int k0, float scale, cpg_f16x4& hi, cpg_f16x4& lo) {
    const bool in = k0 < KP;
    f32x4 x = *reinterpret_cast<const f32x4*>(row + (in ? k0 : 0));
    x *= in ? scale : 0.f;
    uint32_t h[2], l[2];
    split2h_pair(x[0], x[1], h[0], l[0]);
    split2h_pair(x[2], x[3], h[1], l[1]);
    hi = __builtin_bit_cast(cpg_f16x4, make_uint2(h[0], h[1]));
    lo = __builtin_bit_cast(cpg_f16x4, make_uint2(l[0], l[1]));
}

// acc[mt][gate*2+sub] = h_src[rows of m-tiles MT0..MT0+NMT-1] . W_hh^T for this wave's hidden units
template <int G, int R, int MT0, int NMT>
__device__ __forceinline__ void gru_product(const WaveWeights<G, R>& ww, const float* h_src, f32x4 (&acc)[NMT][6]) {
    using C = FusedCfg<G, R>;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if CPG_FUSED_PAIR
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb) {
        cpg_f16x8 ah[NMT], al[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) pair_frag<C::KP>(h_src + ((MT0 + mt) * 16 + l15) * C::LDH, kb, lq, 1.f, ah[mt], al[mt]);
#pragma unroll
        for (int t = 0; t < 3; ++t)   // low x high, high x low, high x high
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t == 0 ? al[mt] : ah[mt], t == 1 ? ww.Bl[kb][nt] : ww.Bh[kb][nt],
                                                                         acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
        cpg_f16x4 ah[NMT], al[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
            pair_frag16<C::KP>(h_src + ((MT0 + mt) * 16 + l15) * C::LDH, 32 * C::KB + 16 * kt + 4 * lq, 1.f, ah[mt], al[mt]);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(t == 0 ? al[mt] : ah[mt], t == 1 ? ww.Btl[kt][nt] : ww.Bth[kt][nt],
                                                                        acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] *= ww.wback;   // the weights' power of two back out (exact)
    return;
#else
#pragma unroll
    for (int g = 0; g < G; ++g) {
        f32x4 af[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
            af[mt] = *reinterpret_cast<const f32x4*>(&h_src[((MT0 + mt) * 16 + l15) * C::LDH + 16 * g + 4 * lq]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], ww.Bf[4 * g + j][nt], acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float at[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) at[mt] = h_src[((MT0 + mt) * 16 + l15) * C::LDH + 16 * G + 4 * r + lq];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[mt], ww.Bf[4 * G + r][nt], acc[mt][nt], 0, 0, 0);
    }
#endif
}

// Cell nonlinearities.  CPG_FUSED_FAST_CELL = 1 (default since round 4): hardware exp2 / reciprocal forms (v_exp_f32 on x log2(e),
// v_rcp_f32) - ~2 ulp against ~1 ulp of the library expf + IEEE division (0: -DCPG_FUSED_FAST_CELL=0), a third of the instructions.
// With the f16-pair product the fused kernels are bound by their VALU work, not by the matrix pipe: beam-5 over 1 M z 163 -> 132 ms,
// greedy 27.3 -> 21.1 ms.  Every decode parity test (golden ids bit-exact, oracle ids exact outside f32 ties of margin < 1e-5, tie
// counts bounded) passes with either form, with the same tie counts (profiles/r04_*_tie_report.json).
#ifndef CPG_FUSED_FAST_CELL
#define CPG_FUSED_FAST_CELL 1
#endif
__device__ __forceinline__ float sigmoid_c(float x) {
#if CPG_FUSED_FAST_CELL
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#else
    return sigmoidf_(x);
#endif
}

// tanh without control flow (the cell must stay one basic block so the scheduler can slide it under MFMAs):
// |x| < 0.625: x + x^3 P(x^2), P fitted to <= 0.8 ulp; else 1 - 2/(e^{2|x|}+1) (<= 1.7 ulp); both evaluated, one selected.
__device__ __forceinline__ float tanh_bf(float x) {
    const float a = fabsf(x), t = a * a;
    float p = 0.0023131025955080986f;
    p = fmaf(p, t, -0.008364741690456867f);
    p = fmaf(p, t, 0.021776489913463593f);
    p = fmaf(p, t, -0.05396042391657829f);
    p = fmaf(p, t, 0.13333310186862946f);
    p = fmaf(p, t, -0.3333333432674408f);
    const float small = fmaf(a * t, p, a);
#if CPG_FUSED_FAST_CELL
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * a) + 1.f);
#else
    const float big = 1.f - 2.f / (expf(2.f * a) + 1.f);
#endif
    return copysignf(a < 0.625f ? small : big, x);
}

// GRU cell in the accumulator layout (gate order r,z,n; n = tanh(gi_n + r*(W_hn h + b_hn)); h' = (1-z)n + z h).
// gi = tab[tok[row]] + rowc[rcrow[row]].  Reads the old state from h_src, writes h' to this wave's own columns of h_dst.
template <int G, int R, int MT0, int NMT>
__device__ __forceinline__ void gru_cell(const WaveWeights<G, R>& ww, const f32x4 (&acc)[NMT][6], int H, const float* tab_l,
                                         const float* rowc_l, const int* tok_l, const int* rcrow_l, const float* h_src,
                                         float* h_dst) {
    using C = FusedCfg<G, R>;
    const int lane = threadIdx.x & 63, lq = lane >> 4;
    const int H3 = 3 * H;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = (MT0 + mt) * 16 + 4 * lq + jj;
            const float* tr = tab_l + tok_l[row] * H3;
            const float* rc = rowc_l + (rcrow_l ? rcrow_l[row] : row) * H3;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int uc = ww.ucl[sub];
                const float gi_r = tr[uc] + rc[uc];
                const float gi_z = tr[H + uc] + rc[H + uc];
                const float gi_n = tr[2 * H + uc] + rc[2 * H + uc];
                const float hn = acc[mt][4 + sub][jj] + ww.bh[4 + sub];
                const float rg = sigmoid_c(gi_r + (acc[mt][sub][jj] + ww.bh[sub]));
                const float zg = sigmoid_c(gi_z + (acc[mt][2 + sub][jj] + ww.bh[2 + sub]));
                const float ng = tanh_bf(gi_n + rg * hn);
                const float hold = h_src[row * C::LDH + uc];
                // multiply (not select) by the ownership mask: a select lets the compiler sink the whole cell under a branch
                h_dst[row * C::LDH + ww.col[sub]] = ((1.f - zg) * ng + zg * hold) * ww.ownf[sub];
            }
        }
}

// logits[16 rows of this wave][V] = h fc_w^T + fc_b  -> logit_l
template <int G, int R>
__device__ __forceinline__ void vocab_logits(const WaveWeights<G, R>& ww, const float* h, const float* fc_l, float* logit_l) {
    using C = FusedCfg<G, R>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    f32x4 lg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#if CPG_FUSED_PAIR
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb) {
        cpg_f16x8 ah, al, bh0, bl0, bh1, bl1;
        pair_frag<C::KP>(h + (wave * 16 + l15) * C::LDH, kb, lq, 1.f, ah, al);
        pair_frag<C::KP>(fc_l + ww.vclamp0 * C::LDH, kb, lq, 1.f, bh0, bl0);
        pair_frag<C::KP>(fc_l + ww.vclamp1 * C::LDH, kb, lq, 1.f, bh1, bl1);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, lg[1], 0, 0, 0);
    }
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
        const int k0 = 32 * C::KB + 16 * kt + 4 * lq;
        cpg_f16x4 ah, al, bh0, bl0, bh1, bl1;
        pair_frag16<C::KP>(h + (wave * 16 + l15) * C::LDH, k0, 1.f, ah, al);
        pair_frag16<C::KP>(fc_l + ww.vclamp0 * C::LDH, k0, 1.f, bh0, bl0);
        pair_frag16<C::KP>(fc_l + ww.vclamp1 * C::LDH, k0, 1.f, bh1, bl1);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh1, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl1, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh0, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh1, lg[1], 0, 0, 0);
    }
    lg[0] *= ww.fback;   // the LDS copy of fc_w carries 2^e_fc (stage_tables)
    lg[1] *= ww.fback;
#else
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&h[(wave * 16 + l15) * C::LDH + 16 * g + 4 * lq]);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&fc_l[ww.vclamp0 * C::LDH + 16 * g + 4 * lq]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&fc_l[ww.vclamp1 * C::LDH + 16 * g + 4 * lq]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], b0[j], lg[0], 0, 0, 0);
            lg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], b1[j], lg[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float at = h[(wave * 16 + l15) * C::LDH + 16 * G + 4 * r + lq];
        lg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, fc_l[ww.vclamp0 * C::LDH + 16 * G + 4 * r + lq], lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, fc_l[ww.vclamp1 * C::LDH + 16 * G + 4 * r + lq], lg[1], 0, 0, 0);
    }
#endif
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int row = wave * 16 + 4 * lq + jj;
        logit_l[row * LGS + l15] = lg[0][jj] + ww.fcb0;
        logit_l[row * LGS + 16 + l15] = lg[1][jj] + ww.fcb1;
    }
}

// ------------------------------------------------------------------------------------------------ greedy
struct GreedyArgs {
    DecoderWeights w;
    const float* h0;    // [N,H]   = [z;c]
    const float* rowc;  // [N,3H]  W_ih[:, E:] . [z;c] + b_ih
    int64_t* ids;       // [N,ld_ids]; this kernel writes columns 1..T
    int* unfinished;    // [T] += rows still running after each step
    int N, T, ld_ids, start, pad, eos, ntiles;
};

// Scheduling directive for a barrier-to-barrier region that holds NMFMA matrix instructions and an independent body of
// VALU / LDS work: emit them as {1 MFMA, PER others} groups.  An f32 16x16x4 MFMA occupies the matrix pipe for 32 cycles
// (8 issue slots); left alone, hipcc emits all MFMAs first and the VALU work after them, i.e. serialises the two pipes.
#ifndef CPG_FUSED_PER
#define CPG_FUSED_PER 0
#endif
template <int NMFMA>
__device__ __forceinline__ void interleave_mfma_with_rest() {
#if CPG_FUSED_PER > 0
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
        __builtin_amdgcn_sched_group_barrier(0x086, CPG_FUSED_PER, 0);  // VALU | SALU | DS
    }
#endif
}

// One quarter of a half-tile's vocabulary projection per wave: m-tile MT0 + (wave>>1), vocabulary columns 16*(wave&1)..+15.
template <int G, int R, int MT0>
__device__ __forceinline__ void half_logits(const float* h, const float* fc_l, const float* fc_b, int V, float* logit_l, float fback) {
    using C = FusedCfg<G, R>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int mrow = (MT0 + (wave >> 1)) * 16, v = 16 * (wave & 1) + l15, vc = min(v, V - 1);
    f32x4 lg = f32x4{0.f, 0.f, 0.f, 0.f};
#if CPG_FUSED_PAIR
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb) {
        cpg_f16x8 ah, al, bh, bl;
        pair_frag<C::KP>(h + (mrow + l15) * C::LDH, kb, lq, 1.f, ah, al);
        pair_frag<C::KP>(fc_l + vc * C::LDH, kb, lq, 1.f, bh, bl);
        lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, lg, 0, 0, 0);
    }
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
        const int k0 = 32 * C::KB + 16 * kt + 4 * lq;
        cpg_f16x4 ah, al, bh, bl;
        pair_frag16<C::KP>(h + (mrow + l15) * C::LDH, k0, 1.f, ah, al);
        pair_frag16<C::KP>(fc_l + vc * C::LDH, k0, 1.f, bh, bl);
        lg = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, lg, 0, 0, 0);
        lg = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, lg, 0, 0, 0);
    }
    lg *= fback;
#else
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&h[(mrow + l15) * C::LDH + 16 * g + 4 * lq]);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(&fc_l[vc * C::LDH + 16 * g + 4 * lq]);
#pragma unroll
        for (int j = 0; j < 4; ++j) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], lg, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        lg = __builtin_amdgcn_mfma_f32_16x16x4f32(h[(mrow + l15) * C::LDH + 16 * G + 4 * r + lq],
                                                  fc_l[vc * C::LDH + 16 * G + 4 * r + lq], lg, 0, 0, 0);
#endif
    const float bias = fc_b[vc];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) logit_l[(mrow + 4 * lq + jj) * LGS + v] = lg[jj] + bias;
}

// Greedy selection of one half-tile, done redundantly by EVERY wave (lanes 0..31 = the half's rows) so that the chosen
// tokens reach the wave's own next cell pass through its private LDS copy without a workgroup barrier.  Returns the
// number of rows still running (identical in all waves).  Wave 0 alone writes the ids and the step's running count.
__device__ __forceinline__ int greedy_select_half(const GreedyArgs& a, const float* logit_l, int half, int row0, int step,
                                                  int& fin, int* tokw_l) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int live = 0;
    if (lane < 32) {
        const int row = half * 32 + lane;
        const float* l = logit_l + row * LGS;
        float best = -INFINITY;
        int arg = 0;
        for (int v = 0; v < a.w.V; ++v) {
            const float x = l[v];
            if (x > best) {
                best = x;
                arg = v;
            }
        }
        const int t = fin ? a.pad : arg;
        live = (fin || t == a.eos) ? 0 : 1;
        if (t == a.eos) fin = 1;
        tokw_l[row] = t;
        if (wave == 0 && row0 + row < a.N) a.ids[(size_t)(row0 + row) * a.ld_ids + 1 + step] = t;
    }
    const int nl = __popcll(__ballot(live));
    if (threadIdx.x == 0 && nl) atomicAdd(&a.unfinished[step], nl);
    return nl;
}

// The tile is run as two half-tiles P (rows 0..31) and Q (rows 32..63) half a step out of phase, so that each
// barrier-to-barrier region holds the matrix work of one half (its h . W_hh^T, plus the other half's vocabulary product)
// and the VALU work of the other half's GRU cell: one basic block, which the scheduler interleaves (1 wave per SIMD:
// nothing else can hide the cell's exp/div/tanh arithmetic behind the MFMA pipe).
//   region 1 of step s:  cell(P,s) -> hP(s+1)   |  product(Q,s)   | logits(Q,s-1)      then select(Q,s-1)
//   region 2 of step s:  cell(Q,s) -> hQ(s+1)   |  product(P,s+1) | logits(P,s)        then select(P,s)
template <int G, int R>
__global__ __launch_bounds__(256, 1) void decode_greedy_fused_kernel(GreedyArgs a) {
    using C = FusedCfg<G, R>;
    extern __shared__ float4 cpg_fused_smem[];
    const int H = a.w.H, H3 = 3 * H, V = a.w.V;
    float* h_l = reinterpret_cast<float*>(cpg_fused_smem);     // [RM][LDH]
    float* fc_l = h_l + RM * C::LDH;                            // [V][LDH]
    float* rowc_l = fc_l + V * C::LDH;                          // [RM][3H]
    float* tab_l = rowc_l + RM * H3;                            // [Vt][3H]
    float* logit_l = tab_l + a.w.Vt * H3;                       // [RM][LGS]
    float* fcb_l = logit_l + RM * LGS;                          // [32]
    int* tokw_l = reinterpret_cast<int*>(fcb_l + 32) + (threadIdx.x >> 6) * RM;  // [4][RM]: this wave's copy of the tokens

    const int tid = threadIdx.x, lane = tid & 63;
    WaveWeights<G, R> ww;
    ww.load(a.w);
    stage_tables<G, R>(a.w, tab_l, fc_l, ww.fsc);
    if (tid < 32) fcb_l[tid] = a.w.fc_b[min(tid, V - 1)];

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * RM, nrows = min(RM, a.N - row0);
        __syncthreads();  // previous tile fully retired before its LDS state is replaced
        for (int i = tid; i < RM * C::LDH; i += 256) {
            const int r = i / C::LDH, k = i - r * C::LDH;
            h_l[i] = (r < nrows && k < H) ? a.h0[(size_t)(row0 + r) * H + k] : 0.f;
        }
        {
            const float* src = a.rowc + (size_t)row0 * H3;
            const int n = nrows * H3;
            for (int i = tid; i < RM * H3; i += 256) rowc_l[i] = i < n ? src[i] : 0.f;
        }
        tokw_l[lane] = a.start;
        int finP = lane < 32 && lane >= nrows, finQ = lane < 32 && 32 + lane >= nrows;  // rows past N never run
        int liveP = 1, liveQ = 1;
        __syncthreads();

        f32x4 accP[2][6], accQ[2][6];
        gru_product<G, R, 0, 2>(ww, h_l, accP);
        for (int step = 0; step < a.T; ++step) {
            // region 1
            gru_product<G, R, 2, 2>(ww, h_l, accQ);
            half_logits<G, R, 2>(h_l, fc_l, fcb_l, V, logit_l, ww.fback);  // step 0: hQ(0), result unused
            gru_cell<G, R, 0, 2>(ww, accP, H, tab_l, rowc_l, tokw_l, nullptr, h_l, h_l);
            interleave_mfma_with_rest<13 * C::KSTEPS>();
            __syncthreads();
            if (step > 0) liveQ = greedy_select_half(a, logit_l, 1, row0, step - 1, finQ, tokw_l);
            // region 2
            gru_product<G, R, 0, 2>(ww, h_l, accP);  // last step: unused
            half_logits<G, R, 0>(h_l, fc_l, fcb_l, V, logit_l, ww.fback);
            gru_cell<G, R, 2, 2>(ww, accQ, H, tab_l, rowc_l, tokw_l, nullptr, h_l, h_l);
            interleave_mfma_with_rest<13 * C::KSTEPS>();
            __syncthreads();
            liveP = greedy_select_half(a, logit_l, 0, row0, step, finP, tokw_l);
            if (liveP + liveQ == 0) break;  // whole tile finished (identical decision in every wave): the rest stays <pad>
        }
        if (liveP + liveQ != 0) {  // drain: Q's last step
            half_logits<G, R, 2>(h_l, fc_l, fcb_l, V, logit_l, ww.fback);
            __syncthreads();
            greedy_select_half(a, logit_l, 1, row0, a.T - 1, finQ, tokw_l);
        }
    }
}

size_t greedy_lds_bytes(int ldh, int H, int V, int Vt) {
    return ((size_t)RM * ldh + (size_t)V * ldh + (size_t)RM * 3 * H + (size_t)Vt * 3 * H + RM * LGS + 32) * 4 + 4 * RM * sizeof(int);
}

template <int G, int R>
int launch_greedy(const GreedyArgs& a, int device_cus, hipStream_t s) {
    const size_t bytes = greedy_lds_bytes(FusedCfg<G, R>::LDH, a.w.H, a.w.V, a.w.Vt);
    auto kern = decode_greedy_fused_kernel<G, R>;
    CPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int grid = a.ntiles < device_cus ? a.ntiles : device_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ beam
// lane exchanges of the selection stages as DPP moves (one VALU instruction; __shfl_xor compiles to ds_bpermute_b32, an LDS
// crossbar round trip per exchange in chains of 20-70 of them): 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm
// [2,3,0,1] (lane ^ 2), 0x141 = row_half_mirror (lane i <-> 7 - i of each 8: pairs the two quads of a group of eight)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true);
}

struct BeamArgs {
    DecoderWeights w;
    const float* h0;      // [N,H]  one row per sentence
    const float* rowc;    // [N,3H] one row per sentence
    int32_t* hist_tok;    // [T,N,K] pre-filled with -1: steps a sentence is not advanced on keep -1
    int32_t* hist_prev;   // [T,N,K]
    float* hist_score;    // [T,N,K]
    int N, T, K, n_best, min_length, bos, eos, S, ntiles;  // S = sentences per tile = RM / K
};

// KT: beam width as a compile-time constant (0 = a.K at run time): the selection stages are loops over K with K-dependent
// bounds and divisions by K; the reference's default width 5 at its default decoder size gets its own instantiation.
template <int G, int R, int KT = 0>
__global__ __launch_bounds__(256, 1) void decode_beam_fused_kernel(BeamArgs a) {
    using C = FusedCfg<G, R>;
    extern __shared__ float4 cpg_fused_smem[];
    const int H = a.w.H, H3 = 3 * H, V = a.w.V, K = KT ? KT : a.K, S = KT ? RM / KT : a.S;
    float* hx_l = reinterpret_cast<float*>(cpg_fused_smem);  // [RM][LDH] state the product reads (rows in beam order)
    float* hy_l = hx_l + RM * C::LDH;                         // [RM][LDH] state after the cell, before the re-gather
    float* fc_l = hy_l + RM * C::LDH;                         // [V][LDH]
    float* rowc_l = fc_l + V * C::LDH;                        // [S][3H]
    float* tab_l = rowc_l + S * H3;                           // [Vt][3H]
    float* logit_l = tab_l + a.w.Vt * H3;                     // [RM][LGS]
    float* cs_l = logit_l + RM * LGS;                         // [RM][MAXK] each row's K best scores ...
    float* sc_l = cs_l + RM * MAXK;                           // [RM] beam scores (row = sentence*K + beam)
    int* cv_l = reinterpret_cast<int*>(sc_l + RM);            // [RM][MAXK] ... and their tokens
    int* tok_l = cv_l + RM * MAXK;                            // [RM] token each row consumes next
    int* rcrow_l = tok_l + RM;                                // [RM] sentence of a row
    int* org_l = rcrow_l + RM;                                // [RM] tile row a row's new state comes from
    int* nfin_l = org_l + RM;                                 // [S]
    int* done_l = nfin_l + RM;                                // [S]

    const int tid = threadIdx.x;
    WaveWeights<G, R> ww;
    ww.load(a.w);
    stage_tables<G, R>(a.w, tab_l, fc_l, ww.fsc);
    if (tid < RM) rcrow_l[tid] = min(tid / K, S - 1);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int s0 = tile * S, nsent = min(S, a.N - s0);
        __syncthreads();
        for (int i = tid; i < RM * C::LDH; i += 256) {
            const int r = i / C::LDH, k = i - r * C::LDH, sl = r / K;
            hx_l[i] = (sl < nsent && k < H) ? a.h0[(size_t)(s0 + sl) * H + k] : 0.f;
        }
        {
            const float* src = a.rowc + (size_t)s0 * H3;
            const int n = nsent * H3;
            for (int i = tid; i < S * H3; i += 256) rowc_l[i] = i < n ? src[i] : 0.f;
        }
        if (tid < RM) {
            tok_l[tid] = a.bos;
            sc_l[tid] = 0.f;
            org_l[tid] = tid;
            nfin_l[tid] = 0;
            done_l[tid] = 0;
        }
        __syncthreads();

        for (int step = 0; step < a.T; ++step) {
            f32x4 acc[MT][6];
            if (!(CPG_BEAM_ABLATE & 1)) gru_product<G, R, 0, MT>(ww, hx_l, acc);
            else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
            }
            if (!(CPG_BEAM_ABLATE & 2)) gru_cell<G, R, 0, MT>(ww, acc, H, tab_l, rowc_l, tok_l, rcrow_l, hx_l, hy_l);
            __syncthreads();
            if (!(CPG_BEAM_ABLATE & 4)) vocab_logits<G, R>(ww, hy_l, fc_l, logit_l);
            __syncthreads();

            // ---- Beam.advance, stage 1: FOUR lanes per row (all 256 threads): log_softmax over the row, each lane's own sorted
            // K best of its quarter of the vocabulary (tokens q, q+4, ...), then a K-round merge across the quad.  Order of a
            // row's list: score descending, ties lower token first (Beam.py's topk over the flattened scores).
            // (One lane per row ran this as a serial chain of LDS reads, 24 expf and a data-dependent insertion on ONE wave while
            // three waves idled: 45 % of the kernel - tools/ablate_beam.sh.)
            if (!(CPG_BEAM_ABLATE & 8)) {
                const int r1 = tid >> 2, q = tid & 3;
                const int sl = r1 / K, kb = r1 - sl * K;
                const bool act = sl < nsent && !done_l[min(sl, S - 1)] && (step > 0 || kb == 0);   // uniform within the quad
                constexpr int VQ = 8;   // V <= 32
                const float* l = logit_l + r1 * LGS;
                float lv[VQ];
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < VQ; ++i) {
                    const int v = q + 4 * i;
                    lv[i] = v < V ? l[v] : -INFINITY;
                    m = fmaxf(m, lv[i]);
                }
                m = fmaxf(m, dpp_f<0xB1>(m));
                m = fmaxf(m, dpp_f<0x4E>(m));
                float se = 0.f;
#pragma unroll
                for (int i = 0; i < VQ; ++i)
                    if (q + 4 * i < V) se += expf(lv[i] - m);
                se += dpp_f<0xB1>(se);
                se += dpp_f<0x4E>(se);
                const float lse = m + logf(se);
                const bool parent_eos = step > 0 && tok_l[r1] == a.eos;
                const float base = sc_l[r1];
                float bs[MAXK];
                int bv[MAXK];
#pragma unroll
                for (int p = 0; p < MAXK; ++p) {
                    bs[p] = -INFINITY;
                    bv[p] = 0;
                }
#pragma unroll
                for (int i = 0; i < VQ; ++i) {
                    const int v = q + 4 * i;
                    if (v >= V) continue;
                    float lp = lv[i] - lse;
                    if (step + 1 < a.min_length && v == a.eos) lp = -1e20f;
                    if (v == a.bos) lp = -1e20f;
                    float cs = step > 0 ? lp + base : lp;
                    if (parent_eos) cs = -1e20f;
                    int cc = v;
                    bool carried = false;
#pragma unroll
                    for (int p = 0; p < MAXK; ++p) {
                        if (p >= K) break;
                        if (carried ? (cs >= bs[p]) : (cs > bs[p])) {
                            const float fs = bs[p];
                            const int fv = bv[p];
                            bs[p] = cs;
                            bv[p] = cc;
                            cs = fs;
                            cc = fv;
                            carried = true;
                        }
                    }
                }
#pragma unroll
                for (int sel = 0; sel < MAXK; ++sel) {
                    if (sel >= K) break;
                    float ws = bs[0];
                    int wv = bv[0];
                    {
                        const float os = dpp_f<0xB1>(ws);
                        const int ov = dpp_i<0xB1>(wv);
                        const bool take = os > ws || (os == ws && ov < wv);
                        ws = take ? os : ws;
                        wv = take ? ov : wv;
                    }
                    {
                        const float os = dpp_f<0x4E>(ws);
                        const int ov = dpp_i<0x4E>(wv);
                        const bool take = os > ws || (os == ws && ov < wv);
                        ws = take ? os : ws;
                        wv = take ? ov : wv;
                    }
                    if (bs[0] == ws && bv[0] == wv) {   // this lane's head won (tokens are unique across the quad): pop it
#pragma unroll
                        for (int p = 0; p + 1 < MAXK; ++p) {
                            bs[p] = bs[p + 1];
                            bv[p] = bv[p + 1];
                        }
                        bs[MAXK - 1] = -INFINITY;
                        bv[MAXK - 1] = 0;
                    }
                    if (act && q == 0) {
                        cs_l[r1 * MAXK + sel] = ws;
                        cv_l[r1 * MAXK + sel] = wv;
                    }
                }
            }
            __syncthreads();

            // ---- stage 2: EIGHT lanes per sentence, lane k holding beam row k's list: K rounds of an 8-lane arg-max
            // (ties: lower beam first = lower flat index), the winner pops, lane `sel` keeps the sel-th selection and then
            // installs it as the new state of row `sel` (stage 2 reads only the candidate lists: the per-row state can be
            // replaced right away)
            int active = 0;
            if ((CPG_BEAM_ABLATE & 16) && tid < nsent) active = 1;
            if (tid < 8 * S && !(CPG_BEAM_ABLATE & 16)) {
                const int sl = tid >> 3, k8 = tid & 7;
                const bool live = sl < nsent && !done_l[sl];   // uniform within the group of 8
                const int kmax = step == 0 ? 1 : K;
                float bs[MAXK];
                int bv[MAXK];
#pragma unroll
                for (int p = 0; p < MAXK; ++p) {
                    const bool have = live && k8 < kmax && p < K;
                    const int idx = (min(sl * K + k8, RM - 1)) * MAXK + p;
                    bs[p] = have ? cs_l[idx] : -INFINITY;
                    bv[p] = have ? cv_l[idx] : 0;
                }
                float my_sc = 0.f;
                int my_tok = 0, my_org = 0, nf_add = 0, tok0 = 0;
#pragma unroll
                for (int sel = 0; sel < MAXK; ++sel) {
                    if (sel >= K) break;
                    float ws = bs[0];
                    int wv = bv[0], wk = k8;
#define CPG_BEAM_MERGE8(CTRL)                                                  \
    {                                                                          \
        const float os = dpp_f<CTRL>(ws);                                      \
        const int ov = dpp_i<CTRL>(wv), ok = dpp_i<CTRL>(wk);                  \
        const bool take = os > ws || (os == ws && ok < wk);                    \
        ws = take ? os : ws;                                                   \
        wv = take ? ov : wv;                                                   \
        wk = take ? ok : wk;                                                   \
    }
                    CPG_BEAM_MERGE8(0xB1)
                    CPG_BEAM_MERGE8(0x4E)
                    CPG_BEAM_MERGE8(0x141)
#undef CPG_BEAM_MERGE8
                    if (k8 == wk) {
#pragma unroll
                        for (int p = 0; p + 1 < MAXK; ++p) {
                            bs[p] = bs[p + 1];
                            bv[p] = bv[p + 1];
                        }
                        bs[MAXK - 1] = -INFINITY;
                        bv[MAXK - 1] = 0;
                    }
                    if (k8 == sel) {
                        my_sc = ws;
                        my_tok = wv;
                        my_org = wk;
                    }
                    if (sel == 0) tok0 = wv;
                    nf_add += wv == a.eos;
                }
                if (live) {
                    const int nf = nfin_l[sl] + nf_add;   // every lane of the group reads the old count before lane 0 updates it
                    if (k8 < K) {
                        const int row = sl * K + k8;
                        sc_l[row] = my_sc;
                        tok_l[row] = my_tok;
                        org_l[row] = sl * K + my_org;
                        const size_t h = ((size_t)step * a.N + (s0 + sl)) * K + k8;
                        a.hist_tok[h] = my_tok;
                        a.hist_prev[h] = my_org;
                        a.hist_score[h] = my_sc;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (k8 == 0) {
                        nfin_l[sl] = nf;
                        if (tok0 == a.eos && nf >= a.n_best) done_l[sl] = 1; else active = 1;
                    }
                }
            }
            const int nact = __syncthreads_count(active);
            if (nact == 0) break;  // every sentence of the tile is done (model.py:364-366)

            // ---- re-gather the hidden rows by back-pointer (model.py:387-404): hx[row] = hy[org[row]]
            for (int i = tid; i < RM * (C::KP / 4) && !(CPG_BEAM_ABLATE & 32); i += 256) {
                const int r = i / (C::KP / 4), c = i - r * (C::KP / 4);
                *reinterpret_cast<f32x4*>(&hx_l[r * C::LDH + 4 * c]) = *reinterpret_cast<const f32x4*>(&hy_l[org_l[r] * C::LDH + 4 * c]);
            }
            __syncthreads();
        }
    }
}

size_t beam_lds_bytes(int ldh, int H, int V, int Vt, int K) {
    const int S = RM / K;
    return ((size_t)2 * RM * ldh + (size_t)V * ldh + (size_t)S * 3 * H + (size_t)Vt * 3 * H + RM * LGS + RM * MAXK + RM) * 4 +
           ((size_t)RM * MAXK + 5 * RM) * sizeof(int);
}

template <int G, int R>
int launch_beam(const BeamArgs& a, int device_cus, hipStream_t s) {
    const size_t bytes = beam_lds_bytes(FusedCfg<G, R>::LDH, a.w.H, a.w.V, a.w.Vt, a.K);
    auto kern = decode_beam_fused_kernel<G, R, 0>;
    if constexpr (G == 6 && R == 2) {
        if (a.K == 5) kern = decode_beam_fused_kernel<G, R, 5>;
    }
    CPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int grid = a.ntiles < device_cus ? a.ntiles : device_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// contraction depth the dispatch pads H to, and the matching LDS row stride
int fused_kp(int H) {
    if (H / 16 == 6 && (H % 16 + 3) / 4 == 2) return 104;
    return H <= 32 ? 32 : H <= 64 ? 64 : H <= 96 ? 96 : 128;
}
int fused_ldh(int H) {
    const int kp = fused_kp(H);
    return ((kp / 4 + 1) % 2 == 1) ? kp + 4 : kp + 8;
}

int device_limits(int* cus, int* lds) {
    int dev = 0;
    CPG_HIP(hipGetDevice(&dev));
    CPG_HIP(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    CPG_HIP(hipDeviceGetAttribute(lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    return 0;
}

#define CPG_FUSED_DISPATCH(LAUNCH, H, ...)                                     \
    do {                                                                       \
        if (fused_kp(H) == 104) return LAUNCH<6, 2>(__VA_ARGS__);              \
        if ((H) <= 32) return LAUNCH<2, 0>(__VA_ARGS__);                       \
        if ((H) <= 64) return LAUNCH<4, 0>(__VA_ARGS__);                       \
        if ((H) <= 96) return LAUNCH<6, 0>(__VA_ARGS__);                       \
        return LAUNCH<8, 0>(__VA_ARGS__);                                      \
    } while (0)

// ================================================================================================ training: whole sequences
// The same residency for TRAINING at these sizes (cpg_gru_seq_fwd / _bwd, _biseq_* dispatch here: cpg_gru_small_seq_ok).  A workgroup
// owns a tile of 32 batch rows of one direction for all T steps:
//   forward  - W_hh in registers as above (gru_product), the state double-buffered in LDS, the token of a row read per step (teacher
//              forcing), h_t and the saved gates (r, z, n, W_hn h + b_hn) written to the slabs cpg_gru_seq_bwd reads;
//   backward - W_hh once more in registers, now as exact-f32 MFMA B fragments of the OTHER orientation (dH = dG_rec W_hh contracts
//              over the 3H gate rows); the recurrent gate gradients of the tile live in LDS ([32][3H]), the carry z (.) dH and the
//              cell's backward run in the accumulator layout on the wave's own hidden units: nothing but dG / dh0 leaves the CU.
// Same arithmetic as the per-step kernels (csrc/gru.hip: the staged exact-f32 backward step; forward on f16 pairs as the decode kernels).
// rows per workgroup tile: RS = 32, or 16 when that still fits one round of workgroups (a tile's step is bound by the instruction
// stream of its four waves - one per SIMD - so half the rows are ~0.55 of the time: small batches spread over twice the CUs)

struct SmallFwdArgs {
    CpgSmallFwdDir d[2];
    int B, H, T, ntiles, rs;
};

template <int G, int R, int RS>
__global__ __launch_bounds__(256, 1) void gru_seq_small_fwd_kernel(SmallFwdArgs a) {
    using C = FusedCfg<G, R>;
    extern __shared__ float4 cpg_fused_smem[];
    const CpgSmallFwdDir& d = a.d[blockIdx.y];
    const int H = a.H, H3 = 3 * H, B = a.B, T = a.T;
    const size_t BH = (size_t)B * H;
    float* h_a = reinterpret_cast<float*>(cpg_fused_smem);   // [RS][LDH]
    float* h_b = h_a + RS * C::LDH;
    float* rowc_l = h_b + RS * C::LDH;                        // [RS][3H]
    int* tok_l = reinterpret_cast<int*>(rowc_l + RS * H3);    // [T][RS]: the tile's tokens of every step
    const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4;
    WaveWeights<G, R> ww;
    ww.load(DecoderWeights{d.tab, d.w_hh, d.b_hh, nullptr, nullptr, H, 0, 1});
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * RS, nrows = min(RS, B - row0);
        __syncthreads();
        const float* h0 = d.hs + (size_t)(d.reverse ? T : 0) * BH;   // the caller put the initial state into its slot
        for (int i = tid; i < RS * C::LDH; i += 256) {
            const int r = i / C::LDH, k = i - r * C::LDH;
            h_a[i] = (r < nrows && k < H) ? h0[(size_t)(row0 + r) * H + k] : 0.f;
            h_b[i] = 0.f;
        }
        for (int i = tid; i < RS * H3; i += 256) {
            const int r = i / H3;
            rowc_l[i] = (d.rowc && r < nrows) ? d.rowc[(size_t)row0 * H3 + i] : 0.f;
        }
        for (int i = tid; i < T * RS; i += 256) {
            const int t = i / RS, r = i - t * RS;
            tok_l[i] = r < nrows ? d.tok[(size_t)t * B + row0 + r] : 0;
        }
        __syncthreads();
        // input-side pre-activations gi = tab[tok] + rowc of this lane's (row, unit) pairs, fetched ONE STEP AHEAD: the token-table rows
        // come from global memory (L2), and their latency runs under the previous step's product and cell
        float gi[RS / 16][4][2][3];
        auto fetch = [&](int t) {
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = mt * 16 + 4 * lq + jj;
                    const float* tr = d.tab + (size_t)tok_l[t * RS + row] * H3;
                    const float* rc = rowc_l + row * H3;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const int uc = ww.ucl[sub];
#pragma unroll
                        for (int q = 0; q < 3; ++q) gi[mt][jj][sub][q] = tr[q * H + uc] + rc[q * H + uc];
                    }
                }
        };
        fetch(d.reverse ? T - 1 : 0);
        float* hs_src = h_a;
        float* hs_dst = h_b;
        for (int p = 0; p < T; ++p) {
            const int t = d.reverse ? T - 1 - p : p;
            f32x4 acc[RS / 16][6];
            gru_product<G, R, 0, RS / 16>(ww, hs_src, acc);
            float* hs_out = d.hs + (size_t)(d.reverse ? t : t + 1) * BH + (size_t)row0 * H;
            float* g_out = d.gates ? d.gates + (size_t)t * 4 * BH + (size_t)row0 * H : nullptr;
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = mt * 16 + 4 * lq + jj;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const int uc = ww.ucl[sub];
                        const float hn = acc[mt][4 + sub][jj] + ww.bh[4 + sub];
                        const float rg = sigmoid_c(gi[mt][jj][sub][0] + (acc[mt][sub][jj] + ww.bh[sub]));
                        const float zg = sigmoid_c(gi[mt][jj][sub][1] + (acc[mt][2 + sub][jj] + ww.bh[2 + sub]));
                        const float ng = tanh_bf(gi[mt][jj][sub][2] + rg * hn);
                        const float hold = hs_src[row * C::LDH + uc];
                        const float hnew = (1.f - zg) * ng + zg * hold;
                        hs_dst[row * C::LDH + ww.col[sub]] = hnew * ww.ownf[sub];
                        if (ww.ownf[sub] != 0.f && row < nrows) {
                            const size_t o = (size_t)row * H + uc;
                            hs_out[o] = hnew;
                            if (g_out) {
                                g_out[o] = rg;
                                g_out[BH + o] = zg;
                                g_out[2 * BH + o] = ng;
                                g_out[3 * BH + o] = hn;
                            }
                        }
                    }
                }
            if (p + 1 < T) fetch(d.reverse ? t - 1 : t + 1);
            __syncthreads();   // the new state is complete (and the old one no longer read)
            float* sw = hs_src;
            hs_src = hs_dst;
            hs_dst = sw;
        }
    }
}

static size_t small_fwd_lds(int RS, int ldh, int H, int T) { return ((size_t)2 * RS * ldh + (size_t)RS * 3 * H) * 4 + (size_t)T * RS * sizeof(int); }

template <int G, int R>
int launch_small_fwd(const SmallFwdArgs& a, int ndir, int cus, hipStream_t s) {
    const size_t bytes = small_fwd_lds(a.rs, FusedCfg<G, R>::LDH, a.H, a.T);
    auto kern = a.rs == 16 ? gru_seq_small_fwd_kernel<G, R, 16> : gru_seq_small_fwd_kernel<G, R, 32>;
    int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(kern), (int)bytes);
    if (rc) return rc;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid, ndir), dim3(256), bytes, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ---- backward.  G3 = 16-deep groups of the contraction over the 3H recurrent gate rows (zero padded): k of MFMA step s, lane group q
// is 16 (s >> 2) + 4 q + (s & 3) as in the exact-f32 form above.
struct SmallBwdArgs {
    CpgSmallBwdDir d[2];
    int B, H, T, ntiles, upw, rs;   // upw: hidden units per wave (= fused_kp(H) / 4, the forward kernels' split)
};

template <int G3, int RS>
__global__ __launch_bounds__(256, 1) void gru_seq_small_bwd_kernel(SmallBwdArgs a) {
    constexpr int KP3 = 16 * G3, KS3 = 4 * G3;
    constexpr int LDG = ((KP3 / 4 + 1) % 2 == 1) ? KP3 + 4 : KP3 + 8;   // row stride = 4 * odd (ds_read_b128 lane groups on disjoint banks)
    extern __shared__ float4 cpg_fused_smem[];
    float* dg_l = reinterpret_cast<float*>(cpg_fused_smem);   // [RS][LDG]: dr_pre | dz_pre | r (.) dn_pre of the step just processed
    const CpgSmallBwdDir& d = a.d[blockIdx.y];
    const int H = a.H, B = a.B, T = a.T;
    const size_t BH = (size_t)B * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    // this wave's hidden units (two 16-column tiles) and W_hh[k][unit] fragments
    int unit[2];
    bool own[2];
    float Bf[KS3][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int ul = 16 * sub + l15;
        unit[sub] = a.upw * wave + ul;
        own[sub] = ul < a.upw && unit[sub] < H;
        const int uc = min(unit[sub], H - 1);
#pragma unroll
        for (int s = 0; s < KS3; ++s) {
            const int k = 16 * (s >> 2) + 4 * lq + (s & 3);
            Bf[s][sub] = (own[sub] && k < 3 * H) ? d.w_hh[(size_t)k * H + uc] : 0.f;
        }
        unit[sub] = uc;
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * RS, nrows = min(RS, B - row0);
        __syncthreads();
        for (int i = tid; i < RS * LDG; i += 256) dg_l[i] = 0.f;
        float carry[RS / 16][2][4], prod[RS / 16][2][4];
#pragma unroll
        for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) carry[mt][sub][jj] = prod[mt][sub][jj] = 0.f;
        __syncthreads();
        // epilogue operands of a step (saved gates, h_prev, the external gradients) for this lane's (row, unit) pairs: loaded one step
        // AHEAD, before the product of the step in front of it - their latency then runs under the MFMAs instead of in front of the cell
        struct Ep { float rg, zg, ng, hn, hp, ex; };
        Ep ep[RS / 16][2][4];
        auto fetch = [&](int p) {
            const int t = d.reverse ? T - 1 - p : p;
            const float* gt = d.gates + (size_t)t * 4 * BH + (size_t)row0 * H;
            const float* hp = d.hs + (size_t)(d.reverse ? t + 1 : t) * BH + (size_t)row0 * H;
            const float* ex = d.dhs_ext ? d.dhs_ext + (size_t)t * BH + (size_t)row0 * H : nullptr;
            const float* ex2 = (p == T - 1 && d.dh_last) ? d.dh_last + (size_t)row0 * H : nullptr;
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = min(mt * 16 + 4 * lq + jj, nrows - 1);   // (clamped: branch-free loads; the stores below are guarded)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const size_t o = (size_t)row * H + unit[sub];
                        Ep& e = ep[mt][sub][jj];
                        e.rg = gt[o]; e.zg = gt[BH + o]; e.ng = gt[2 * BH + o]; e.hn = gt[3 * BH + o]; e.hp = hp[o];
                        e.ex = (ex ? ex[o] : 0.f) + (ex2 ? ex2[o] : 0.f);
                    }
                }
        };
        fetch(T - 1);
        for (int p = T - 1; p >= 0; --p) {
            const int t = d.reverse ? T - 1 - p : p;
            float* dg = d.dG + ((size_t)t * B + row0) * 4 * H;
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = mt * 16 + 4 * lq + jj;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        if (!(own[sub] && row < nrows)) continue;
                        const int u = unit[sub];
                        const Ep e = ep[mt][sub][jj];
                        const float dh = prod[mt][sub][jj] + carry[mt][sub][jj] + e.ex;
                        const float rg = e.rg, zg = e.zg, ng = e.ng, hn = e.hn, hprev = e.hp;
                        const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
                        const float dz_pre = dh * (hprev - ng) * zg * (1.f - zg);
                        const float dr_pre = dn_pre * hn * rg * (1.f - rg);
                        carry[mt][sub][jj] = dh * zg;
                        float* o4 = dg + (size_t)row * 4 * H + u;
                        o4[0] = dr_pre;
                        o4[H] = dz_pre;
                        o4[2 * H] = dn_pre * rg;
                        o4[3 * H] = dn_pre;
                        float* l = dg_l + row * LDG + u;
                        l[0] = dr_pre;
                        l[H] = dz_pre;
                        l[2 * H] = dn_pre * rg;
                    }
                }
            if (p > 0) fetch(p - 1);
            __syncthreads();   // the tile's recurrent gate gradients are complete
            f32x4 acc[RS / 16][2];
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) acc[mt][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < G3; ++g) {
                f32x4 af[RS / 16];
#pragma unroll
                for (int mt = 0; mt < RS / 16; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(&dg_l[(mt * 16 + l15) * LDG + 16 * g + 4 * lq]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
                            acc[mt][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], Bf[4 * g + j][sub], acc[mt][sub], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) prod[mt][sub][jj] = acc[mt][sub][jj];
            __syncthreads();   // every wave has read the tile before the next step overwrites it
        }
        if (d.dh0) {
#pragma unroll
            for (int mt = 0; mt < RS / 16; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = mt * 16 + 4 * lq + jj;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
                        if (own[sub] && row < nrows) d.dh0[(size_t)(row0 + row) * H + unit[sub]] = prod[mt][sub][jj] + carry[mt][sub][jj];
                }
        }
    }
}

template <int G3>
int launch_small_bwd(const SmallBwdArgs& a, int ndir, int cus, hipStream_t s) {
    constexpr int KP3 = 16 * G3;
    constexpr int LDG = ((KP3 / 4 + 1) % 2 == 1) ? KP3 + 4 : KP3 + 8;
    const size_t bytes = (size_t)a.rs * LDG * 4;
    auto kern = a.rs == 16 ? gru_seq_small_bwd_kernel<G3, 16> : gru_seq_small_bwd_kernel<G3, 32>;
    int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(kern), (int)bytes);
    if (rc) return rc;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid, ndir), dim3(256), bytes, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- whole-sequence training launches for small GRU recurrences (cpg_internal.h)
// rows per tile: 16 while every 16-row tile of the launch still gets its own CU, else 32 (option small_seq_rows = 16 | 32 overrides)
static int small_rows(int B, int ndir, int cus) {
    const CpgOptVal o = cpg_opt(OPT_SMALL_SEQ_ROWS);
    if (o.set && (o.i == 16 || o.i == 32)) return o.i;
    return (long)cdiv(B, 16) * ndir <= cus ? 16 : 32;
}
bool cpg_gru_small_seq_ok(int B, int H) {
    const CpgOptVal o = cpg_opt(OPT_GRU_SMALL_SEQ);
    if (o.set && o.i == 0) return false;
    return B > 0 && H > 0 && H <= 128 && cpg_compute_mode_get() != 1;
}
int cpg_gru_small_seq_fwd(int T, int B, int H, int ndir, const CpgSmallFwdDir* d, hipStream_t s) {
    SmallFwdArgs a;
    for (int i = 0; i < 2; ++i) a.d[i] = d[i < ndir ? i : 0];
    const int cus = cpg_device_cus();
    a.rs = small_rows(B, ndir, cus);
    a.B = B; a.H = H; a.T = T; a.ntiles = cdiv(B, a.rs);
    CPG_FUSED_DISPATCH(launch_small_fwd, H, a, ndir, cus, s);
}
int cpg_gru_small_seq_bwd(int T, int B, int H, int ndir, const CpgSmallBwdDir* d, hipStream_t s) {
    SmallBwdArgs a;
    for (int i = 0; i < 2; ++i) a.d[i] = d[i < ndir ? i : 0];
    const int cus = cpg_device_cus();
    a.rs = small_rows(B, ndir, cus);
    a.B = B; a.H = H; a.T = T; a.ntiles = cdiv(B, a.rs); a.upw = fused_kp(H) / 4;
    if (3 * H <= 96) return launch_small_bwd<6>(a, ndir, cus, s);
    if (3 * H <= 192) return launch_small_bwd<12>(a, ndir, cus, s);
    if (3 * H <= 288) return launch_small_bwd<18>(a, ndir, cus, s);
    if (3 * H <= 320) return launch_small_bwd<20>(a, ndir, cus, s);
    return launch_small_bwd<24>(a, ndir, cus, s);
}

CPG_EXPORT size_t cpg_decode_greedy_fused_lds_bytes(int H, int V, int Vt) {
    if (H <= 0 || H > 128 || V <= 0 || V > 32 || Vt <= 0) return 0;
    return greedy_lds_bytes(fused_ldh(H), H, V, Vt);
}

CPG_EXPORT int cpg_decode_greedy_fused(const float* h0, const float* rowc, const float* tab, int Vt, const float* w_hh,
                                       const float* b_hh, const float* fc_w, const float* fc_b, int N, int H, int V, int T,
                                       int start, int pad, int eos, int64_t* ids, int ld_ids, int* unfinished, void* stream) {
    CPG_CHECK_ARG(h0 && rowc && tab && w_hh && b_hh && fc_w && fc_b && ids && unfinished);
    CPG_CHECK_ARG(N > 0 && T > 0 && ld_ids >= T + 1 && H > 0 && H <= 128 && V > 0 && V <= 32 && Vt > 0);
    CPG_CHECK_ARG(start >= 0 && start < Vt && pad >= 0 && pad < Vt && eos >= 0 && V <= Vt);
    int cus = 0, lds = 0;
    if (int rc = device_limits(&cus, &lds)) return rc;
    const size_t need = cpg_decode_greedy_fused_lds_bytes(H, V, Vt);
    if (need > (size_t)lds) {
        cpg_set_error("cpg_decode_greedy_fused: needs %zu bytes of LDS per workgroup, device offers %d", need, lds);
        return -3;
    }
    GreedyArgs a{{tab, w_hh, b_hh, fc_w, fc_b, H, V, Vt}, h0, rowc, ids, unfinished, N, T, ld_ids, start, pad, eos, cdiv(N, RM)};
    hipStream_t s = (hipStream_t)stream;
    CPG_FUSED_DISPATCH(launch_greedy, H, a, cus, s);
}

// Launcher introspection (bench.py labels the CLaSS roofline with it): the whole-loop decode kernel a launch with hidden size H
// (and beam width K) runs, named as rocprofv3 prints it.  kind 0 greedy, 1 beam.  Returns the length written (0: not covered).
CPG_EXPORT int cpg_decode_fused_kernel_name(int kind, int H, int K, char* buf, int n) {
    if (H <= 0 || H > 128 || kind < 0 || kind > 1) return 0;
    int g, r = 0;
    if (fused_kp(H) == 104) { g = 6; r = 2; }
    else g = H <= 32 ? 2 : H <= 64 ? 4 : H <= 96 ? 6 : 8;
    if (kind == 0) return snprintf(buf, n, "decode_greedy_fused_kernel<%d, %d>", g, r);
    return snprintf(buf, n, "decode_beam_fused_kernel<%d, %d, %d>", g, r, (g == 6 && r == 2 && K == 5) ? 5 : 0);
}

CPG_EXPORT size_t cpg_decode_beam_fused_lds_bytes(int H, int V, int Vt, int K) {
    if (H <= 0 || H > 128 || V <= 0 || V > 32 || Vt <= 0 || K <= 0 || K > MAXK || K > V) return 0;
    return beam_lds_bytes(fused_ldh(H), H, V, Vt, K);
}

CPG_EXPORT int cpg_decode_beam_fused(const float* h0, const float* rowc, const float* tab, int Vt, const float* w_hh,
                                     const float* b_hh, const float* fc_w, const float* fc_b, int N, int H, int V, int T, int K,
                                     int n_best, int min_length, int bos, int eos, int32_t* hist_tok, int32_t* hist_prev,
                                     float* hist_score, void* stream) {
    CPG_CHECK_ARG(h0 && rowc && tab && w_hh && b_hh && fc_w && fc_b && hist_tok && hist_prev && hist_score);
    CPG_CHECK_ARG(N > 0 && T > 0 && H > 0 && H <= 128 && V > 0 && V <= 32 && Vt >= V && K > 0 && K <= MAXK && K <= V);
    CPG_CHECK_ARG(n_best > 0 && n_best <= K && bos >= 0 && bos < Vt && eos >= 0);
    int cus = 0, lds = 0;
    if (int rc = device_limits(&cus, &lds)) return rc;
    const size_t need = cpg_decode_beam_fused_lds_bytes(H, V, Vt, K);
    if (need > (size_t)lds) {
        cpg_set_error("cpg_decode_beam_fused: needs %zu bytes of LDS per workgroup, device offers %d", need, lds);
        return -3;
    }
    const int S = RM / K;
    BeamArgs a{{tab, w_hh, b_hh, fc_w, fc_b, H, V, Vt}, h0, rowc, hist_tok, hist_prev, hist_score, N, T, K, n_best, min_length,
               bos, eos, S, cdiv(N, S)};
    hipStream_t s = (hipStream_t)stream;
    CPG_FUSED_DISPATCH(launch_beam, H, a, cus, s);
}
