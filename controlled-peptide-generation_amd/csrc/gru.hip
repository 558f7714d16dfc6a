// GRU sequence forward / backward for gfx950: one fused launch per time step.
//
// Replaces torch.nn.GRU as the reference drives it (models/encoder.py:25-30,42; models/decoder.py:40-41,77,98).
// Gate row order r,z,n;  n = tanh(gi_n + r*(W_hn h + b_hn));  h' = (1-z)*n + z*h.
//
// Forward step  (gru_step_fwd_kernel):  [B,H] x [H,3H] recurrent product on the f32 MFMA engine with the r/z/n rows of the
//   same hidden units in one tile, and the whole cell (input-side gather, sigmoid/tanh, blend, save-for-backward) as the
//   epilogue - the [B,T,3H] input pre-activation tensor is never materialised: it is rebuilt per element from
//     tab[tok[b]]  (token table  emb @ W_ih[:, :E]^T + b_ih, V rows)   +   rowc[b]  (constant over time: [z;c] @ W_ih[:, E:]^T)
//     + dense[b]   (upper encoder layers).
// Backward step (gru_step_bwd_kernel):  dH_s = ext_s + z_{s+1}*dH_{s+1} + dgh_{s+1} W_hh  ([B,3H] x [3H,H] product), and the
//   cell backward as the epilogue, writing dG_s = [dr_pre, dz_pre, dhn, dn_pre] (dgh = first 3H columns, contiguous).
//
// State slab hs[(T+1),B,H]:  forward direction: hs[0]=h0, h_t at hs[t+1];  reverse: hs[T]=h0, h_t at hs[t].
// So h_{prev}(t) is one contiguous [T*B,H] block in both directions (offset 0 / B*H) for the dW_hh product.
#include "gemm_core.h"
#include "cpg_internal.h"
#include "pair_engine.h"
#include <stdlib.h>
#include <limits.h>
#include <string.h>

#ifndef CPG_FWD_PREFETCH
#define CPG_FWD_PREFETCH 0   // measured: fetching the epilogue operands ahead of the MFMA loop costs registers (occupancy) for no gain
#endif

typedef uint32_t pk_u32x2 __attribute__((ext_vector_type(2)));

constexpr int DM_ROWS_C = 128, DM_VMAX_C = 31;   // batch rows per workgroup / largest token table of dgi_mfma_kernel (asserted there)

struct GruFwdArgs {
    const float* h_prev;
    const float* w_hh;
    const float* b_hh;
    const int32_t* tok;  // [B] ids of this step, or null
    const float* tab;    // [V,3H]
    const float* rowc;   // [B,3H]
    const float* dense;  // [B,3H] of this step
    float* h_out;        // [B,H]
    float* gates;        // [4,B,H] of this step (r,z,n,hn), or null; bf16 elements when gates_bf16 (bf16 compute mode only)
    int gates_bf16;
    int B, H;
    int row0, row1;      // this launch covers batch rows [row0,row1): rows are independent recurrences, so row groups
                         // can run as separate launch chains on separate streams, out of phase with each other
    const int32_t* nrows;  // device scalar or null: only rows < *nrows are live at this step (length-sorted batches:
                           // rows whose remaining targets are all <pad> need no state) - read on the device, no host sync
    const int* wx = nullptr;   // exponent record of w_hh (cpg_weight_exp): the f16-pair engine multiplies W_hh by 2^e_w before its split
                               // (gemm_core.h: weight_exp_from_parts); null: the launch runs the bf16x3 engine, which needs none
};

// Up to two independent sequences (the two directions of a biGRU layer) share one launch: gridDim.z selects the
// argument set.  Twice the work per launch amortises the launch ramp / first-slab / epilogue phases, which are a fixed
// ~30 % of a single-direction launch at B=2048,H=512 (plain product: 59 TFLOP/s at M=2048, 91 at M=8192).
struct GruFwdPair {
    GruFwdArgs d[2];
};

// h . W_hh^T of the forward step: exact-f32 MFMA (0) or six bf16 MFMAs on operands split when the slab is stored (7)
#ifndef CPG_STEP_FWD_SPLIT
#define CPG_STEP_FWD_SPLIT 7
#endif
// PREC: 7 = f32-grade (three planes, six MFMAs), 1 = bf16 compute mode (one plane, one MFMA; cpg_set_compute_mode(1))
template <class TC, bool VEC, int PREC = 7>
using FwdLoop = MainLoop<TC, true, true, VEC, VEC, false, (CPG_STEP_FWD_SPLIT == 7 && TC::BK == 32) ? PREC : 0>;   // PREC 7 | 8 | 1

template <class TC, bool VEC, int PREC>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruFwdPair pr) {
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const GruFwdArgs& g = pr.d[bz];
    const int H = g.H, B = g.nrows ? min(g.row1, *g.nrows) : g.row1;  // row bound of this launch
    const int m0 = g.row0 + by * TC::BM, j0 = bx * (TC::BN / 3);
    if (m0 >= B) return;  // tile entirely past the live rows (uniform for the workgroup)
    static_assert(TC::NI % 3 == 0, "wave tile holds r,z,n blocks");
    constexpr int NJ = TC::NI / 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % TC::WN;

    // Epilogue operands (input-side pre-activations gathered from the token table / row constant / dense term, and
    // h_prev) do not depend on the matrix product: fetch them FIRST so their two dependent global-load latencies
    // (tok -> table row) run under the MFMA loop instead of after it.
    float gi[NJ][TC::MI][4][3], hp[NJ][TC::MI][4];
    auto fetch = [&]() {
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
        const int j = j0 + (wn * NJ + jb) * 16 + (lane & 15);
        const bool jok = j < H;
        const int jc = jok ? j : 0;
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                const int rc = (row < B) ? row : 0;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                if (g.tok) {
                    const float* t = g.tab + (size_t)g.tok[rc] * 3 * H;
                    a0 += t[jc]; a1 += t[H + jc]; a2 += t[2 * H + jc];
                }
                if (g.rowc) {
                    const float* t = g.rowc + (size_t)rc * 3 * H;
                    a0 += t[jc]; a1 += t[H + jc]; a2 += t[2 * H + jc];
                }
                if (g.dense) {
                    const float* t = g.dense + (size_t)rc * 3 * H;
                    a0 += t[jc]; a1 += t[H + jc]; a2 += t[2 * H + jc];
                }
                gi[jb][mi][r][0] = a0; gi[jb][mi][r][1] = a1; gi[jb][mi][r][2] = a2;
                hp[jb][mi][r] = g.h_prev[(size_t)rc * H + jc];
            }
    }
    };
#if CPG_FWD_PREFETCH
    fetch();
#endif

    OpA a{g.h_prev, H, m0, B, nullptr, 1.f};
    OpB b{g.w_hh, H, j0, H, H, nullptr, 1.f};
    constexpr bool PAIR = FwdLoop<TC, VEC, PREC>::NP == 2;   // PREC 8 on the plane engine: f16 pairs, W_hh times 2^e_w (gemm_core.h)
    float w_back = 1.f;
    if constexpr (PAIR) {
        const int e = weight_exp_from_parts(g.wx);
        b.pscale = pair_pow2(e);
        w_back = pair_pow2(-e);
    }
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    FwdLoop<TC, VEC, PREC>::run(a, b, H, acc);
    if constexpr (PAIR) {
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] *= w_back;
    }
#if !CPG_FWD_PREFETCH
    fetch();
#endif

    const size_t BH = (size_t)g.B * H;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
        const int j = j0 + (wn * NJ + jb) * 16 + (lane & 15);
        if (j >= H) continue;
        const float bh_r = g.b_hh[j], bh_z = g.b_hh[H + j], bh_n = g.b_hh[2 * H + j];
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                if (row >= B) continue;
                const float hn = acc[mi][jb * 3 + 2][r] + bh_n;
                const float rg = sigmoidf_(gi[jb][mi][r][0] + (acc[mi][jb * 3 + 0][r] + bh_r));
                const float zg = sigmoidf_(gi[jb][mi][r][1] + (acc[mi][jb * 3 + 1][r] + bh_z));
                const float ng = tanhf(gi[jb][mi][r][2] + rg * hn);
                const size_t o = (size_t)row * H + j;
                g.h_out[o] = (1.f - zg) * ng + zg * hp[jb][mi][r];
                if constexpr (PREC == 1) {
                    if (g.gates && g.gates_bf16) {   // bf16 compute mode: the four saved values of an element as one 8-byte group
                        __builtin_nontemporal_store(pk_u32x2{cvt_pk_bf16(rg, zg), cvt_pk_bf16(ng, hn)},
                                                    reinterpret_cast<pk_u32x2*>(reinterpret_cast<uint16_t*>(g.gates) + 4 * o));
                        continue;
                    }
                }
                if (g.gates) {  // written once, read once by the backward pass much later: keep them out of the L2
                    __builtin_nontemporal_store(rg, g.gates + o);
                    __builtin_nontemporal_store(zg, g.gates + BH + o);
                    __builtin_nontemporal_store(ng, g.gates + 2 * BH + o);
                    __builtin_nontemporal_store(hn, g.gates + 3 * BH + o);
                }
            }
    }
}

struct GruBwdArgs {
    const float* dG_next;  // [B,4H] of the step processed just before this one (s+1), null on the first launch
    const float* w_hh;     // [3H,H]
    const float* w_hhT;    // [H,3H] = w_hh^T for the direct-to-LDS kernel (both operands K-contiguous), or null
    const float* dH_next;  // [B,H] z_{s+1} .* (total gradient of h_{s+1}): what step s+1's launch left in dH_out; null on the first launch
    const float* ext;      // [B,H] external gradient on h_s (time-aligned slice) or null
    const float* ext2;     // [B,H] second external gradient (final-state gradient on the first launch) or null
    const float* gates;    // [4,B,H] of step s; null on the closing launch that only emits dh0
    const float* h_prev;   // [B,H] h_{s-1}
    float* dH_out;         // [B,H] z_s .* (total gradient of h_s): the carry term of step s-1, pre-multiplied here where z_s is in
                           // registers anyway (one operand stream less per step); closing launch (no gates): dh0 itself
    float* dG_out;         // [B,4H]
    int B, H;
    int row0, row1;
    const int32_t* nrows;       // device scalar or null: rows live at step s (see GruFwdArgs)
    const int32_t* nrows_next;  // rows live at step s+1: beyond them dG_next / dH_next were never written
    int ep_step;                // (even) slab spacing of the staggered epilogue-operand fetch; 0: every workgroup ahead of slab 0
    int gates_bf16;             // gates hold bf16 elements (bf16 compute mode, direct-to-LDS kernels only)
    int dg_bf16;                // bf16 gradient storage (implies gates_bf16): dG_next / dG_out hold bf16 [B,4H], w_hhT bf16 [H,3H]
    // f16-pair engine of the direct-to-LDS step (PREC 3, below): the three recurrent blocks of dG once more, as the NEXT launch's
    // A operand - [B][6H] f16, k-groups of 32 in the order (column group, block), each 128 bytes = [32 hi | 32 lo] of the values
    // times 2^e, e = ex[(row / 32) * (H / 32) + column group] (INT_MAX: the 32 x 32 x 3 values are all zero).  w_hhT then holds the
    // same layout of W_hh^T times 2^e_w (pair_engine.h: weight_exp_from_parts over ex_min + H/32).
    const uint16_t* pp_next;
    const int* ex_next;
    uint16_t* pp_out;
    int* ex_out;
    int* ex_min;   // [H/32] smallest exponent of every column group over the launches of the sequence so far (pair_w_kernel resets it):
                   // the column scale of the dW_hh product on f16 pairs (cpg_gru_wgrad_hh)
    // all-T planes form (pair_engine.h: ApScratch): pp_out / ex_out are step s's OWN images (kept), dG_out receives ONLY the
    // input-side n-gate block dn_pre as [B,H] f32, and the step also leaves h_prev as unscaled f16-pair planes [B][2H]
    uint16_t* hp_out = nullptr;
    // bf16 compute mode with bf16 gradient storage, all-T form: the step also leaves h_prev rounded to bf16 [B][H] (the rounding the
    // mode's dW_hh product applied when it staged the f32 states) - the B operand of its conversion-free product (pair_tn.h, NP = 1)
    uint16_t* hb_out = nullptr;
};

struct GruBwdPair {
    GruBwdArgs d[2];
};

// dgh . W_hh of the register-staged backward step: W_hh [3H,H] as stored is the transposed-use (XC) operand.  Its split
// staging works on k-row pairs (tiles at least 64 columns wide: six bf16 MFMAs on operands split when the slab is stored);
// narrower tiles run the exact-f32 MFMA (v_mfma_f32_16x16x4_f32).  Full tiles of dense batches run gru_step_bwd_dl_kernel.
template <class TC, bool VEC>
using BwdLoop = MainLoop<TC, true, false, VEC, VEC, false, (TC::BK == 32 && TC::BV % 2 == 0) ? 7 : 0>;

template <class TC, bool VEC>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruBwdPair pr) {
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const GruBwdArgs& g = pr.d[bz];
    const int H = g.H, B = g.nrows ? min(g.row1, *g.nrows) : g.row1;  // row bound of this launch
    const int Bn = g.nrows_next ? min(B, *g.nrows_next) : B;            // rows that carry a gradient from step s+1
    const int m0 = g.row0 + by * TC::BM, j0 = bx * TC::BN;
    if (m0 >= B) return;
    const size_t BH = (size_t)g.B * H;
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (VEC) {
        // ---- row-layout epilogue (H % 4 == 0, 16-byte aligned operands): every prologue / epilogue access is a 16-byte
        // load or store of four consecutive columns of one row - a quarter of the memory instructions of the accumulator
        // layout (one dword per lane, 64 B per row segment), which is what a step launch is bound by
        extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float* const tb = cpg_smem + BwdLoop<TC, VEC>::smem_bytes() / sizeof(float) + wave * 256;
        const int rb0 = m0 + (wave / TC::WN) * TC::WTM + (lane >> 2), cb0 = j0 + (wave % TC::WN) * TC::WTN + 4 * (lane & 3);
        f32x4 pre[TC::MI][TC::NI], sv[TC::MI][TC::NI][5];
        // The epilogue operands (36 B per element: saved gates, h_prev, z*dH of step s+1, external gradients) are fetched
        // from INSIDE the slab loop, at a slab that differs between the workgroups sharing a CU: issued ahead of the loop by
        // every workgroup at once they are one 36 MB burst the whole chip waits out before its first slab.
        auto load_ep = [&]() {
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC::NI; ++ni) {
                    const int row = rb0 + mi * 16, col = cb0 + ni * 16;
                    const size_t o = (size_t)((row < B) ? row : 0) * H + ((col < H) ? col : 0);
                    f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (g.dH_next && row < Bn) p += *reinterpret_cast<const f32x4*>(g.dH_next + o);
                    if (g.ext) p += *reinterpret_cast<const f32x4*>(g.ext + o);
                    if (g.ext2) p += *reinterpret_cast<const f32x4*>(g.ext2 + o);
                    pre[mi][ni] = p;
                    if (g.gates) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) sv[mi][ni][q] = *reinterpret_cast<const f32x4*>(g.gates + q * BH + o);
                        sv[mi][ni][4] = *reinterpret_cast<const f32x4*>(g.h_prev + o);
                    }
                }
        };
        if (g.dG_next) {
            OpA a{g.dG_next, 4 * H, m0, Bn, nullptr, 1.f};  // rows past Bn read as zero
            OpB b{g.w_hh, H, j0, H, 0, nullptr, 1.f};
            const int hb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);  // dispatch order: XCD = hb % 8
            const int phase = ((hb >> 3) + (hb >> 8)) & 3, last = ((3 * H + TC::BK - 1) / TC::BK - 1) & ~1;
            BwdLoop<TC, VEC>::run(a, b, 3 * H, acc, min(phase * g.ep_step, last), load_ep);
        } else {
            load_ep();
        }
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TC::NI; ++ni) {
                const f32x4 dh = acc_block_to_rows(tb, acc[mi][ni], lane) + pre[mi][ni];
                const int row = rb0 + mi * 16, col = cb0 + ni * 16;
                if (row >= B || col >= H) continue;
                const size_t o = (size_t)row * H + col;
                *reinterpret_cast<f32x4*>(g.dH_out + o) = g.gates ? dh * sv[mi][ni][1] : dh;
                if (!g.gates) continue;
                const f32x4 rg = sv[mi][ni][0], zg = sv[mi][ni][1], ng = sv[mi][ni][2], hn = sv[mi][ni][3], hp = sv[mi][ni][4];
                const f32x4 dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
                const f32x4 dz_pre = dh * (hp - ng) * zg * (1.f - zg);
                const f32x4 dr_pre = dn_pre * hn * rg * (1.f - rg);
                float* d = g.dG_out + (size_t)row * 4 * H + col;
                *reinterpret_cast<f32x4*>(d) = dr_pre;
                *reinterpret_cast<f32x4*>(d + H) = dz_pre;
                *reinterpret_cast<f32x4*>(d + 2 * H) = dn_pre * rg;
                *reinterpret_cast<f32x4*>(d + 3 * H) = dn_pre;
            }
        return;
    }

    // ---- accumulator-layout epilogue (any H / alignment)
    // epilogue operands first (see the forward kernel): saved gates, h_prev and the non-GEMM part of dH
    float pre[TC::NI][TC::MI][4], sv[TC::NI][TC::MI][4][5];
#pragma unroll
    for (int ni = 0; ni < TC::NI; ++ni) {
        const int j = j0 + acc_col<TC>(ni);
        const int jc = (j < H) ? j : 0;
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                const size_t o = (size_t)((row < B) ? row : 0) * H + jc;
                float p = 0.f;
                if (g.dH_next && row < Bn) p += g.dH_next[o];
                if (g.ext) p += g.ext[o];
                if (g.ext2) p += g.ext2[o];
                pre[ni][mi][r] = p;
                if (g.gates) {
                    sv[ni][mi][r][0] = g.gates[o];
                    sv[ni][mi][r][1] = g.gates[BH + o];
                    sv[ni][mi][r][2] = g.gates[2 * BH + o];
                    sv[ni][mi][r][3] = g.gates[3 * BH + o];
                    sv[ni][mi][r][4] = g.h_prev[o];
                }
            }
    }
    if (g.dG_next) {
        OpA a{g.dG_next, 4 * H, m0, Bn, nullptr, 1.f};  // rows past Bn read as zero
        OpB b{g.w_hh, H, j0, H, 0, nullptr, 1.f};
        BwdLoop<TC, VEC>::run(a, b, 3 * H, acc);
    }
#pragma unroll
    for (int ni = 0; ni < TC::NI; ++ni) {
        const int j = j0 + acc_col<TC>(ni);
        if (j >= H) continue;
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                if (row >= B) continue;
                const size_t o = (size_t)row * H + j;
                const float dh = acc[mi][ni][r] + pre[ni][mi][r];
                g.dH_out[o] = g.gates ? dh * sv[ni][mi][r][1] : dh;
                if (!g.gates) continue;
                const float rg = sv[ni][mi][r][0], zg = sv[ni][mi][r][1], ng = sv[ni][mi][r][2], hn = sv[ni][mi][r][3];
                const float hp = sv[ni][mi][r][4];
                const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
                const float dz_pre = dh * (hp - ng) * zg * (1.f - zg);
                const float dr_pre = dn_pre * hn * rg * (1.f - rg);
                float* d = g.dG_out + (size_t)row * 4 * H;
                d[j] = dr_pre;
                d[H + j] = dz_pre;
                d[2 * H + j] = dn_pre * rg;
                d[3 * H + j] = dn_pre;
            }
    }
}

using GF128 = TileCfg<128, 96, 32, 2, 2, 3>;
using GF64 = TileCfg<64, 96, 32, 2, 2, 3>;
using GF32 = TileCfg<32, 96, 32, 2, 2, 3>;
// register-staged backward step (every shape the direct-to-LDS kernel does not cover: partial tiles, ragged batches, H % 32 != 0)
using GB64 = TileCfg<64, 32, 32, 4, 1, 1>;
using GB32 = TileCfg<32, 64, 32, 2, 2, 1>;
using GB32N = TileCfg<32, 32, 32, 2, 2, 1>;

// ---- backward step with a direct-to-LDS main loop ("DL"): exact-f32 MFMA on 32 x 32 tiles like gru_step_bwd_kernel<GB32N>, but
// both operands K-contiguous (dgh rows, W_hh^T rows) and staged by `global_load_lds_dwordx4` into a THREE-stage LDS ring, two
// slabs ahead of the product: no staging registers, no ds_write pass, four ds_read_b128 per slab instead of two + eight dword
// reads, and almost no VALU work in the slab loop (the exact-f32 MFMA shares its lanes with the VALU: every address / mask
// instruction of the register-staged loop is paid in matrix-pipe time).  An LDS-DMA lane writes to
// base + 16 * lane, so the image is unpadded ([row][32 floats]); bank conflicts of the fragment reads are avoided by a
// source-side swizzle instead: lane (row, slot s) loads k-chunk s ^ f(row), f(row) = (row >> 1) & 7, and the reader of k-chunk q
// of a row reads slot q ^ f(row).  Contraction order = the register-staged kernel's (k = 16h + 4q + j): bit-identical sums.
// Covers dense launches with B % 32 == 0, H % 32 == 0 and 16-byte aligned operands; everything else runs gru_step_bwd_kernel.
// BM x BN tile, 2 x 2 waves (wave tile BM/2 x BN/2 = MI x NI blocks of 16 x 16); main loop: DlLoop (gemm_core.h)
#ifndef CPG_DL_DELAY
#define CPG_DL_DELAY 0
#endif
#ifndef CPG_DL_DELAY2
#define CPG_DL_DELAY2 0
#endif
#if CPG_DL_DELAY2
__device__ unsigned cpg_dl_tickets[4096];
#endif
#ifndef CPG_BWD_DL_NS
#define CPG_BWD_DL_NS 3   // stages of the backward step's LDS ring (2: 44.4 / 33.7 us paired / single at config B, 3: 40.3 / 28.0, 4: EXPERIMENTS R6.9)
#endif
#define CPG_STR_(x) #x
#define CPG_STR(x) CPG_STR_(x)
#ifndef CPG_DL_ABLATE
#define CPG_DL_ABLATE 0   // diagnostic builds (tools/variant_build.sh): 1 no epilogue loads, 2 no stores, 4 no main loop
#endif
// Saved gates r, z, n, hn of four consecutive elements (offset o of a [B,H] plane) for the direct-to-LDS backward kernels.
// f32: four planes [4][B,H], one 16-byte load each.  bf16 (PREC 1 = bf16 compute mode only, gates_bf16): [B,H][4] bf16 - the four
// values of an element are one 8-byte group, so a lane's four elements are 32 contiguous bytes: two loads instead of four.
template <int PREC>
__device__ __forceinline__ void ld_gates4(const float* gates, size_t BH, size_t o, int bf, f32x4* sv) {
    if constexpr (PREC >= 1) {
        if (bf) {
            const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(gates) + 4 * o);
            const uint4 w0 = p[0], w1 = p[1];   // (r|z, n|hn) of elements 0,1 and 2,3
            auto lo = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
            auto hi = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); };
            sv[0] = f32x4{lo(w0.x), lo(w0.z), lo(w1.x), lo(w1.z)};
            sv[1] = f32x4{hi(w0.x), hi(w0.z), hi(w1.x), hi(w1.z)};
            sv[2] = f32x4{lo(w0.y), lo(w0.w), lo(w1.y), lo(w1.w)};
            sv[3] = f32x4{hi(w0.y), hi(w0.w), hi(w1.y), hi(w1.w)};
            return;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const f32x4*>(gates + q * BH + o);
}

// dG row segment of a lane: four consecutive columns of each of the four gate-gradient blocks.  PREC 2 (bf16 gradient storage): the
// values are rounded to bf16 (RNE) HERE - the rounding the bf16 mode's consumers applied to the f32 values when they read them (the
// next step's fragment read, the dW_hh product's LDS store): the BPTT chain and dW_hh see the same operands as with f32 storage.
template <int PREC>
__device__ __forceinline__ void st_dg4(float* dG_out, size_t row, int H, int col, const f32x4 v0, const f32x4 v1, const f32x4 v2, const f32x4 v3,
                                       bool ap = false) {
    if (PREC == 3 && ap) {   // all-T planes form: the three recurrent blocks live in the kept plane images only; dn_pre as [B,H]
        *reinterpret_cast<f32x4*>(dG_out + row * H + col) = v3;
        return;
    }
    if constexpr (PREC == 2) {
        uint16_t* d = reinterpret_cast<uint16_t*>(dG_out) + row * 4 * H + col;
        *reinterpret_cast<uint2*>(d) = make_uint2(cvt_pk_bf16(v0[0], v0[1]), cvt_pk_bf16(v0[2], v0[3]));
        *reinterpret_cast<uint2*>(d + H) = make_uint2(cvt_pk_bf16(v1[0], v1[1]), cvt_pk_bf16(v1[2], v1[3]));
        *reinterpret_cast<uint2*>(d + 2 * H) = make_uint2(cvt_pk_bf16(v2[0], v2[1]), cvt_pk_bf16(v2[2], v2[3]));
        *reinterpret_cast<uint2*>(d + 3 * H) = make_uint2(cvt_pk_bf16(v3[0], v3[1]), cvt_pk_bf16(v3[2], v3[3]));
    } else {
        float* d = dG_out + row * 4 * H + col;
#ifndef CPG_DIAG_SKIP_REC_F32   // diagnostic build (results wrong downstream): what the f32 copy of the recurrent blocks costs the PREC 3 step
        *reinterpret_cast<f32x4*>(d) = v0;
        *reinterpret_cast<f32x4*>(d + H) = v1;
        *reinterpret_cast<f32x4*>(d + 2 * H) = v2;
#endif
        *reinterpret_cast<f32x4*>(d + 3 * H) = v3;
    }
}

// PREC 1: bf16 compute mode (operands rounded at the fragment read, one bf16 MFMA per block and slab); PREC 2: the same arithmetic on
// bf16 gradient storage (dG and W_hh^T are bf16 in memory: DlLoop's 64-deep slabs)
// WR = 4 (round 6, PREC 3): one 512-thread workgroup on a 128 x 64 tile where two 64 x 64 workgroups of a CU each fetched the same
// W_hh^T tile - 3/4 of the operand bytes through the L2; every wave still owns 32 rows x BN/2 columns
// WC = 4 (round 6, PREC 3): the 64 x 64 tile on EIGHT waves of 32 x 16 - the kernel's three phases (epilogue-operand loads, main loop,
// cell arithmetic + stores) are one serial chain per wave (EXPERIMENTS R6.9: they add up), so half the epilogue state per wave and four
// waves per SIMD (<= 128 VGPRs) shorten the chain at the same LDS and L2 traffic
template <int BM, int BN, int NS, int PREC = 0, int WR = 2, int WC = 2>
__global__ __launch_bounds__(64 * WR * WC, WC == 4 ? 2 : 1) void gru_step_bwd_dl_kernel(GruBwdPair pr) {
    using DL = DlLoop<BM, BN, NS, PREC, WR, WC>;
    static_assert(PREC != 3 || BM / WR == 32, "f16-pair step: a wave owns 32 rows - one exponent row of the pair planes");
    constexpr int MI = DL::MI, NI = DL::NI;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const GruBwdArgs& g = pr.d[bz];
    const int H = g.H;
    const int m0 = g.row0 + by * BM, j0 = bx * BN;
    const size_t BH = (size_t)g.B * H;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tb = cpg_smem + DL::smem_floats() + wave * 256;
    const int wm = wave / WC, wn = wave % WC;
    const int rb0 = m0 + wm * (BM / WR) + (lane >> 2), cb0 = j0 + wn * (BN / WC) + 4 * (lane & 3);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pre[MI][NI], sv[MI][NI][5];
    auto load_ep = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const size_t o = (size_t)(rb0 + 16 * mi) * H + cb0 + 16 * ni;
                f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
                if (CPG_DL_ABLATE & 1) {   // diagnostic build: no epilogue-operand loads (results wrong)
                    pre[mi][ni] = p;
#pragma unroll
                    for (int q = 0; q < 5; ++q) sv[mi][ni][q] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
                    continue;
                }
                if (g.dH_next) p += *reinterpret_cast<const f32x4*>(g.dH_next + o);
                if (g.ext) p += *reinterpret_cast<const f32x4*>(g.ext + o);
                if (g.ext2) p += *reinterpret_cast<const f32x4*>(g.ext2 + o);
                pre[mi][ni] = p;
                if (g.gates) {
                    ld_gates4<PREC>(g.gates, BH, o, g.gates_bf16, sv[mi][ni]);
                    sv[mi][ni][4] = *reinterpret_cast<const f32x4*>(g.h_prev + o);
                }
            }
    };
#if CPG_DL_DELAY2  // diagnostic builds: the workgroup that arrives SECOND on its CU (ticket per hardware CU id) starts CPG_DL_DELAY2 ticks late
    if (gridDim.x * gridDim.y * gridDim.z == 512) {
        __shared__ unsigned tk_;
        if (threadIdx.x == 0) {
            const unsigned cu = (unsigned)__builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4) | ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8);
            tk_ = atomicAdd(&cpg_dl_tickets[cu & 4095], 1u);
        }
        __syncthreads();
        if (tk_ & 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)CPG_DL_DELAY2) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
#if CPG_DL_DELAY   // diagnostic builds: the second workgroup of every CU (dispatch order) starts CPG_DL_DELAY 10-ns ticks late
    if (((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)CPG_DL_DELAY) __builtin_amdgcn_s_sleep(8);
    }
#endif
    PairConsumer<MI, NI, 3> pc;   // PREC 3 (pair_engine.h)
    const int w_exp = (PREC == 3 && g.dG_next) ? weight_exp_from_parts(g.ex_min + H / 32) : 0;   // power of two of the W_hh^T image
    if (g.dG_next && !(CPG_DL_ABLATE & 4)) {
        const int hb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int phase = ((hb >> 3) + (hb >> 8)) & 3;
        if constexpr (PREC == 3) {
            pc.init(g.ex_next + (size_t)((m0 + wm * 32) / 32) * (H / 32), H / 32, lane);
            DL::run(g.pp_next + (size_t)m0 * 6 * H, (size_t)6 * H, reinterpret_cast<const uint16_t*>(g.w_hhT) + (size_t)j0 * 6 * H,
                    (size_t)6 * H, 6 * H, cpg_smem, acc, min(phase * g.ep_step, 3 * H / 32 - 1), load_ep, [&](int kt) { return pc.pre(kt, acc); });
            pc.finish(acc, w_exp);
        } else if constexpr (PREC == 2)
            DL::run(reinterpret_cast<const uint16_t*>(g.dG_next) + (size_t)m0 * 4 * H, (size_t)4 * H,
                    reinterpret_cast<const uint16_t*>(g.w_hhT) + (size_t)j0 * 3 * H, (size_t)3 * H, 3 * H, cpg_smem, acc,
                    min(phase * (g.ep_step / 2), 3 * H / 64 - 1), load_ep);
        else
            DL::run(g.dG_next + (size_t)m0 * 4 * H, (size_t)4 * H, g.w_hhT + (size_t)j0 * 3 * H, (size_t)3 * H, 3 * H, cpg_smem, acc,
                    min(phase * g.ep_step, 3 * H / 32 - 1), load_ep);
    } else {
        load_ep();
    }
    f32x4 pv[PREC == 3 ? MI : 1][PREC == 3 ? NI : 1][3];   // PREC 3: the three recurrent blocks of dG, kept for the pair planes
    float vmax = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const f32x4 dh = acc_block_to_rows(tb, acc[mi][ni], lane) + pre[mi][ni];
            const int row = rb0 + 16 * mi, col = cb0 + 16 * ni;
            const size_t o = (size_t)row * H + col;
            if ((CPG_DL_ABLATE & 2) && dh[0] != 12345.f) continue;   // diagnostic build: no result stores
            *reinterpret_cast<f32x4*>(g.dH_out + o) = g.gates ? dh * sv[mi][ni][1] : dh;
            if (!g.gates) continue;
            const f32x4 rg = sv[mi][ni][0], zg = sv[mi][ni][1], ng = sv[mi][ni][2], hn = sv[mi][ni][3], hp = sv[mi][ni][4];
            const f32x4 dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
            const f32x4 dz_pre = dh * (hp - ng) * zg * (1.f - zg);
            const f32x4 dr_pre = dn_pre * hn * rg * (1.f - rg);
            st_dg4<PREC>(g.dG_out, (size_t)row, H, col, dr_pre, dz_pre, dn_pre * rg, dn_pre, g.hp_out != nullptr);
            if constexpr (PREC == 2) {
                if (g.hb_out) *reinterpret_cast<uint2*>(g.hb_out + o) = make_uint2(cvt_pk_bf16(hp[0], hp[1]), cvt_pk_bf16(hp[2], hp[3]));
            }
            if constexpr (PREC == 3) {
                if (g.hp_out) pair_store4<1>(g.hp_out, (size_t)row, H, col, 0, hp);   // h_prev: the dW_hh product's B operand
                pv[mi][ni][0] = dr_pre; pv[mi][ni][1] = dz_pre; pv[mi][ni][2] = dn_pre * rg;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(pv[mi][ni][q][j]));   // (fmaxf drops a NaN: see below)
            }
        }
    if constexpr (PREC == 3) {
        if (!g.gates || !g.pp_out) return;   // (block-uniform)
        // ---- the next launch's A operand (pair_engine.h): this wave's 32 rows x BN/2 columns x 3 blocks
        const int grp = (j0 + wn * (BN / WC)) / 32;
        const int e = pair_group_exponent<BN / WC>(vmax, cpg_smem, wave, lane, g.ex_out + (size_t)((m0 + wm * 32) / 32) * (H / 32) + grp,
                                                   g.ex_min + grp);
        if (e != INT_MAX || g.hp_out) {   // kept images (all-T form) are read by consumers that do not look at the table first: zeros
            const float sc = e == INT_MAX ? 1.f : pair_pow2(e);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 3; ++q) pair_store4<3>(g.pp_out, (size_t)(rb0 + 16 * mi), H, cb0 + 16 * ni, q, pv[mi][ni][q] * sc);
        }
    }
}


// ---- the same step for launches that have only ~256 tiles of 64 x 64 (one direction at B=2048, H=512): ONE 512-thread
// workgroup per tile whose two 256-thread halves each run the direct-to-LDS loop over HALF of K on their own LDS ring (the
// operand traffic per MFMA of the 64 x 64 tile with the eight waves per CU of two 64 x 32 workgroups), then swap partial
// blocks through LDS so that each half finishes - adds, cell backward, stores - one 16-row block row of every wave tile.
// The two half sums are added once (a + b): NOT the slab-by-slab order of the other kernels, equal to them within f32 rounding.
template <int PREC>
__global__ __launch_bounds__(512) void gru_step_bwd_dl2_kernel(GruBwdPair pr) {
    using DL = DlLoop<64, 64, 3, PREC>;
    constexpr int NI = DL::NI;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const GruBwdArgs& g = pr.d[bz];
    const int H = g.H;
    const int m0 = g.row0 + by * 64, j0 = bx * 64;
    const size_t BH = (size_t)g.B * H;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave8 >> 2, wave = wave8 & 3;
    float* const ring = cpg_smem + kg * DL::smem_floats();
    float* const tb = cpg_smem + 2 * DL::smem_floats() + wave8 * 256;
    const int wm = wave >> 1, wn = wave & 1;
    // after the swap this wave owns block row kg of its wave tile
    const int rb = m0 + wm * 32 + 16 * kg + (lane >> 2), cb0 = j0 + wn * 32 + 4 * (lane & 3);
    f32x4 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pre[NI], sv[NI][5];
    auto load_ep = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const size_t o = (size_t)rb * H + cb0 + 16 * ni;
            f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.dH_next) p += *reinterpret_cast<const f32x4*>(g.dH_next + o);
            if (g.ext) p += *reinterpret_cast<const f32x4*>(g.ext + o);
            if (g.ext2) p += *reinterpret_cast<const f32x4*>(g.ext2 + o);
            pre[ni] = p;
            if (g.gates) {
                ld_gates4<PREC>(g.gates, BH, o, g.gates_bf16, sv[ni]);
                sv[ni][4] = *reinterpret_cast<const f32x4*>(g.h_prev + o);
            }
        }
    };
    f32x4 mine[NI];
    if (g.dG_next) {
        const int Kh = 3 * H / 2;
        const int hb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int phase = ((hb >> 3) + (hb >> 8)) & 3;
        if constexpr (PREC == 2)
            DL::run(reinterpret_cast<const uint16_t*>(g.dG_next) + (size_t)m0 * 4 * H + kg * Kh, (size_t)4 * H,
                    reinterpret_cast<const uint16_t*>(g.w_hhT) + (size_t)j0 * 3 * H + kg * Kh, (size_t)3 * H, Kh, ring, acc,
                    min(phase * (g.ep_step / 4), Kh / 64 - 1), load_ep);
        else
            DL::run(g.dG_next + (size_t)m0 * 4 * H + kg * Kh, (size_t)4 * H, g.w_hhT + (size_t)j0 * 3 * H + kg * Kh, (size_t)3 * H, Kh,
                    ring, acc, min(phase * (g.ep_step / 2), Kh / 32 - 1), load_ep);
        __syncthreads();   // every fragment read of the rings is done: their first 16 KB carry the swap
        f32x4* const xb = reinterpret_cast<f32x4*>(cpg_smem);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) xb[(((1 - kg) * 4 + wave) * NI + ni) * 64 + lane] = kg ? acc[0][ni] : acc[1][ni];
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mine[ni] = (kg ? acc[1][ni] : acc[0][ni]) + xb[((kg * 4 + wave) * NI + ni) * 64 + lane];
    } else {
        load_ep();
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mine[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const f32x4 dh = acc_block_to_rows(tb, mine[ni], lane) + pre[ni];
        const int col = cb0 + 16 * ni;
        const size_t o = (size_t)rb * H + col;
        *reinterpret_cast<f32x4*>(g.dH_out + o) = g.gates ? dh * sv[ni][1] : dh;
        if (!g.gates) continue;
        const f32x4 rg = sv[ni][0], zg = sv[ni][1], ng = sv[ni][2], hn = sv[ni][3], hp = sv[ni][4];
        const f32x4 dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
        const f32x4 dz_pre = dh * (hp - ng) * zg * (1.f - zg);
        const f32x4 dr_pre = dn_pre * hn * rg * (1.f - rg);
        st_dg4<PREC>(g.dG_out, (size_t)rb, H, col, dr_pre, dz_pre, dn_pre * rg, dn_pre);
        if constexpr (PREC == 2) {
            if (g.hb_out) *reinterpret_cast<uint2*>(g.hb_out + o) = make_uint2(cvt_pk_bf16(hp[0], hp[1]), cvt_pk_bf16(hp[2], hp[3]));
        }
    }
}


template <class TC, int PREC>
static int launch_fwd_p(const GruFwdPair& pr, int nd, bool vec, hipStream_t s) {
    const GruFwdArgs& a = pr.d[0];
    dim3 grid(cdiv(a.H, TC::BN / 3), cdiv(a.row1 - a.row0, TC::BM), nd);
    const size_t smem = FwdLoop<TC, true, PREC>::smem_bytes();
    const void* k = vec ? reinterpret_cast<const void*>(gru_step_fwd_kernel<TC, true, PREC>)
                        : reinterpret_cast<const void*>(gru_step_fwd_kernel<TC, false, PREC>);
    if (smem > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per (instantiation, device)
        const int rc = cpg_allow_big_lds(k, (int)smem);
        if (rc) return rc;
    }
    if (vec)
        hipLaunchKernelGGL((gru_step_fwd_kernel<TC, true, PREC>), grid, dim3(256), smem, s, pr);
    else
        hipLaunchKernelGGL((gru_step_fwd_kernel<TC, false, PREC>), grid, dim3(256), smem, s, pr);
    return 0;
}

template <class TC>
static int launch_fwd(const GruFwdPair& pr, int nd, bool vec, hipStream_t s) {
    // f32-grade mode: the engine of the persistent forward kernels (option f32_engine: f16 pairs by default, three bf16 planes)
    const int np = cpg_persist_planes();
    bool have_wx = true;   // the f16-pair engine needs the exponent record of every direction's W_hh; without it: bf16x3 (no range to guard)
    for (int d = 0; d < nd; ++d) have_wx = have_wx && pr.d[d].wx != nullptr;
    return np == 1 ? launch_fwd_p<TC, 1>(pr, nd, vec, s) : (np == 2 && have_wx) ? launch_fwd_p<TC, 8>(pr, nd, vec, s) : launch_fwd_p<TC, 7>(pr, nd, vec, s);
}

template <class TC>
static int launch_bwd(const GruBwdPair& pr, int nd, bool vec, hipStream_t s) {
    const GruBwdArgs& a = pr.d[0];
    dim3 grid(cdiv(a.H, TC::BN), cdiv(a.row1 - a.row0, TC::BM), nd);
    const size_t smem = BwdLoop<TC, true>::smem_bytes() + 4 * 256 * sizeof(float);  // + per-wave transposition buffers
    if (smem > 64 * 1024) {
        const void* k = vec ? reinterpret_cast<const void*>(gru_step_bwd_kernel<TC, true>)
                            : reinterpret_cast<const void*>(gru_step_bwd_kernel<TC, false>);
        const int rc = cpg_allow_big_lds(k, (int)smem);
        if (rc) return rc;
    }
    if (vec)
        hipLaunchKernelGGL((gru_step_bwd_kernel<TC, true>), grid, dim3(256), smem, s, pr);
    else
        hipLaunchKernelGGL((gru_step_bwd_kernel<TC, false>), grid, dim3(256), smem, s, pr);
    return 0;
}

// Row-tile height of the per-step forward kernel (option gru_fwd_bm = 32 | 64 | 128 forces).  Measured on MI355X at B=2048,
// H=512: 64-row tiles (2 workgroups per CU) beat 128-row tiles by 20-25 % - the second resident workgroup covers the other's
// staging waits and epilogue traffic; 32-row tiles lose on the split engine (every row tile converts the whole W_hh slab).
static int fwd_bm_choice(int rows, int H, int nd) {
    const CpgOptVal& o = cpg_opt(OPT_GRU_FWD_BM);
    if (o.set && (o.i == 128 || o.i == 64 || o.i == 32)) return (int)o.i;
    return ((long)cdiv(rows, 64) * cdiv(H, 32) >= 256 || rows > 32) ? 64 : 32;
}

static int gru_fwd_launch(const GruFwdPair& pr, int nd, hipStream_t s) {
    const GruFwdArgs& a = pr.d[0];
    bool vec = a.H % 4 == 0;
    for (int d = 0; d < nd; ++d) vec = vec && aligned16(pr.d[d].h_prev) && aligned16(pr.d[d].w_hh);
    const int bm = fwd_bm_choice(a.row1 - a.row0, a.H, nd);
    const int rc = bm == 128 ? launch_fwd<GF128>(pr, nd, vec, s) : bm == 64 ? launch_fwd<GF64>(pr, nd, vec, s) : launch_fwd<GF32>(pr, nd, vec, s);
    if (rc) return rc;
    CPG_LAUNCH_CHECK();
    return 0;
}

int cpg_gru_step_fwd_launch(const GruFwdArgs& a, hipStream_t s) {
    GruFwdPair pr;
    pr.d[0] = a;
    pr.d[1] = a;
    return gru_fwd_launch(pr, 1, s);
}

// ---- backward-step launch policy.  Option gru_bwd_tile ("32x32" | "64x32" | "32x64" | "64x64") forces the tile of either
// kernel; gru_bwd_dl = 0 keeps the register-staged kernel; gru_bwd_dl2 = 0 / 1 disables / forces the two-K-halves form.
struct BwdTile {
    int bm, bn;
    int waves = 4;   // 8: the f16-pair step's eight-wave forms (128 x 64: 4 x 2 waves; 64 x 64: 2 x 4 waves of 32 x 16) - "64x64x8"
};
static bool parse_tile(const CpgOptVal& o, BwdTile& t) {
    if (!o.set) return false;
    int bm = 0, bn = 0, w = 4;
    if (sscanf(o.s, "%dx%dx%d", &bm, &bn, &w) < 2) return false;
    if (!(bm == 128 && bn == 64) && ((bm != 32 && bm != 64) || (bn != 32 && bn != 64))) return false;   // 128x64: the f16-pair step only
    if (bm == 128) w = 8;
    if (w != 4 && !(w == 8 && bm >= 64 && bn == 64)) return false;
    t = {bm, bn, w};
    return true;
}

// Register-staged exact-f32 kernel: 32 x 32 tiles when they give >= 1024 workgroups (measured 51.4 us against 53.7 for 64 x 32 at
// B=2048, H=512), 64-row tiles for big batches otherwise, 32 x 64 for small ones.
static BwdTile staged_tile(int rows, int H, int nd) {
    BwdTile t;
    if (parse_tile(cpg_opt(OPT_GRU_BWD_TILE), t) && !(t.bm >= 64 && t.bn == 64)) return t;
    if ((long)cdiv(rows, 32) * cdiv(H, 32) * nd >= 1024) return {32, 32};
    return ((long)cdiv(rows, 64) * cdiv(H, 32) >= 256 || rows > 32) ? BwdTile{64, 32} : BwdTile{32, 64};
}

// Slab spacing of the staggered epilogue-operand fetch of the register-staged kernel: a quarter of the slab count by default
// (the four workgroups a CU holds fetch ahead of slabs 0, KT/4, KT/2, 3KT/4); option gru_bwd_stagger overrides, 0 = all ahead.
static int bwd_ep_step(int H) {
    const CpgOptVal& o = cpg_opt(OPT_GRU_BWD_STAGGER);
    const int kt = cdiv(3 * H, 32);
    return (int)(o.set && o.i >= 0 ? o.i : kt / 4) & ~1;
}

// The direct-to-LDS backward step (gru_step_bwd_dl_kernel) covers full 32 x 32 tiles of dense batches.
static bool bwd_dl_shape_ok(int row0, int row1, int H) {
    const CpgOptVal& o = cpg_opt(OPT_GRU_BWD_DL);
    if (o.set && o.i == 0) return false;
    return row0 % 32 == 0 && (row1 - row0) % 32 == 0 && row1 > row0 && H % 32 == 0;
}
// W_hh^T is needed by the direct-to-LDS kernel only
static bool bwd_wants_wt(int rows, int H, int row0, bool dense) { return dense && H % 4 == 0 && bwd_dl_shape_ok(row0, row0 + rows, H); }

template <int BM, int BN, int PREC, int WR = 2, int WC = 2>
static int launch_dl(const GruBwdPair& pr, int nd, hipStream_t s) {
    const GruBwdArgs& a = pr.d[0];
    dim3 grid(a.H / BN, (a.row1 - a.row0) / BM, nd);
    const size_t smem = (DlLoop<BM, BN, CPG_BWD_DL_NS, PREC, WR, WC>::smem_floats() + WR * WC * 256) * sizeof(float);
    if (smem > 64 * 1024) {
        const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(gru_step_bwd_dl_kernel<BM, BN, CPG_BWD_DL_NS, PREC, WR, WC>), (int)smem);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((gru_step_bwd_dl_kernel<BM, BN, CPG_BWD_DL_NS, PREC, WR, WC>), grid, dim3(64 * WR * WC), smem, s, pr);
    return 0;
}

template <int PREC>
static int launch_dl2(const GruBwdPair& pr, int nd, hipStream_t s) {
    const GruBwdArgs& a = pr.d[0];
    dim3 grid(a.H / 64, (a.row1 - a.row0) / 64, nd);
    const size_t smem = (2 * DlLoop<64, 64, 3, PREC>::smem_floats() + 8 * 256) * sizeof(float);
    const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(gru_step_bwd_dl2_kernel<PREC>), (int)smem);
    if (rc) return rc;
    hipLaunchKernelGGL((gru_step_bwd_dl2_kernel<PREC>), grid, dim3(512), smem, s, pr);
    return 0;
}
// 64 x 64 tiles, two K-halves per workgroup: launches with 128 <= tiles < 512 (fewer than two 64 x 64 workgroups per CU, the
// decoder's single direction at B=2048, H=512) IN THE bf16 COMPUTE MODE - measured at that shape, us per launch against the
// 64 x 32 direct-to-LDS kernel: bf16 mode 21.1 vs 24.7, f32-grade 38.9 vs 37.1 (its main loop is matrix-pipe-bound either way
// and the eight-wave barrier costs more than the halved operand traffic returns).
static bool bwd_dl2_wanted(int rows, int H, int nd, bool bf16) {
    const CpgOptVal& o = cpg_opt(OPT_GRU_BWD_DL2);
    if (o.set && o.i == 0) return false;
    if (rows % 64 != 0 || H % 64 != 0) return false;
    if (o.set && o.i == 1) return true;
    const long wg64 = (long)(rows / 64) * (H / 64) * nd;
    return bf16 && wg64 >= 128 && wg64 < 512;
}

// Tile of the direct-to-LDS kernel (tools/kb.py, B=2048, H=512, us per launch; 32x32 / 64x32 / 32x64 / 64x64): single direction
// 38.9 / 35.5 / 35.0 / 38.2, paired directions 71.6 / 66.9 / 68.0 / 60.7 - larger tiles halve the operand traffic per MFMA, as
// long as at least two workgroups per CU remain.
static BwdTile dl_tile(int rows, int H, int nd) {
    const bool r64 = rows % 64 == 0, h64 = H % 64 == 0;
    const long wg64 = (long)(rows / 64) * (H / 64) * nd;   // 64 x 64 tiles of the launch
    BwdTile t{32, 32};
    if (!parse_tile(cpg_opt(OPT_GRU_BWD_TILE), t)) {
        if (r64 && h64 && wg64 >= 512) t = {64, 64};
        else if (r64 && wg64 >= 256) t = {64, 32};
    }
    if (t.bm == 64 && !r64) t.bm = 32;
    if (t.bn == 64 && !h64) t.bn = 32;
    return t;
}

// Saved gates stored as bf16: bf16 compute mode (option bf16_store = 0 keeps f32), and only where every backward step of the sequence
// runs a direct-to-LDS kernel (dense batch, shape covered) - the register-staged kernels read f32 gates.
bool cpg_gru_store_bf16(int B, int H, bool dense) {
    if (cpg_compute_mode_get() != 1) return false;
    const CpgOptVal& o = cpg_opt(OPT_BF16_STORE);
    if (o.set && o.i == 0) return false;
    return dense && H % 4 == 0 && bwd_dl_shape_ok(0, B, H);
}
// Gate GRADIENTS (dG) stored as bf16 as well - "bf16 gradient storage": where the saved gates are bf16 (above), the width suits the
// 64-deep slabs of the bf16-operand loop (both halves of the two-K-halves kernel: H % 128 == 0) and every consumer of dG has a bf16
// form: the next step's operand (DlLoop PREC 2), the dW_hh product (bf16 A operand), the one-pass input-side reduction
// (dgi_mfma_kernel: token table of V <= 31 rows, 128-row chunks).  Option bf16_dg = 0 keeps f32.  PMC of the f32-storage form
// (profiles/r04_bf16_summary.md): the paired BPTT launch moves 137 MB at 0.52 of the HBM peak with the matrix pipe 7 % busy.
bool cpg_gru_dg_store_bf16(int B, int H, bool dense, int V) {
    if (!cpg_gru_store_bf16(B, H, dense)) return false;
    const CpgOptVal& o = cpg_opt(OPT_BF16_DG);
    if (o.set && o.i == 0) return false;
    return H % 128 == 0 && B % DM_ROWS_C == 0 && V > 0 && V <= DM_VMAX_C;
}
// element e of a saved-gates / gate-gradient buffer
static inline float* gate_at(const float* gates, size_t e, bool bf) {
    return const_cast<float*>(bf ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(gates) + e) : gates + e);
}

enum BwdKernelKind { BK_STAGED, BK_DL, BK_DL2 };
struct BwdPlan {
    BwdKernelKind kind;
    BwdTile tile;
    bool bf16;
    bool pair_ok;   // the f16-pair form of the direct-to-LDS step covers this launch shape (f32-grade mode, 64-row tiles)
};
// f16-pair engine of the backward step (PREC 3): option gru_bwd_engine = "exact" keeps the exact-f32 MFMA
static bool bwd_pair_enabled() {
    if (cpg_compute_mode_get() == 1) return false;
    const CpgOptVal o = cpg_opt(OPT_GRU_BWD_ENGINE);
    return !(o.set && strcmp(o.s, "exact") == 0);
}
static BwdPlan bwd_plan(int rows, int H, int nd, int row0, bool vec, bool have_wt, bool dense) {
    const bool bf16 = cpg_compute_mode_get() == 1;
    if (vec && have_wt && dense && bwd_dl_shape_ok(row0, row0 + rows, H)) {
        if (!cpg_opt(OPT_GRU_BWD_TILE).set && bwd_dl2_wanted(rows, H, nd, bf16)) return {BK_DL2, {64, 64}, bf16, false};
        BwdTile t = dl_tile(rows, H, nd);
        const bool pair = t.bm >= 64 && H <= 2048 && bwd_pair_enabled();
        if (t.bm == 128 && !(pair && rows % 128 == 0 && H % 64 == 0)) t.bm = 64, t.waves = 4;   // 128 x 64 exists for the f16-pair step on whole tiles only
        if (t.waves == 8 && !((pair || bf16) && t.bm == 64 && t.bn == 64 && H % 64 == 0 && rows % 64 == 0)) t.waves = 4;
        // f16-pair step, launcher's own choice (round 6): 64 x 64 tiles on EIGHT waves of 32 x 16 from 256 tiles up - 116 VGPRs, four waves
        // per SIMD at two workgroups per CU (paired directions: 42.9 -> 41.6 us) or eight waves on every CU (one direction at B=2048,
        // H=512: 28.9 on 64 x 32 -> 27.8)
        if (pair && !cpg_opt(OPT_GRU_BWD_TILE).set && rows % 64 == 0 && H % 64 == 0 && (long)(rows / 64) * (H / 64) * nd >= 256) t = {64, 64, 8};
        // bf16 compute mode: the same from 512 tiles up (paired directions at config B: 24.2 -> 22.8 us; one direction stays on the
        // two-K-halves kernel above: 15.4 against 16.5)
        if (bf16 && !cpg_opt(OPT_GRU_BWD_TILE).set && t.bm == 64 && t.bn == 64 && rows % 64 == 0 && (long)(rows / 64) * (H / 64) * nd >= 512) t.waves = 8;
        return {BK_DL, t, bf16, pair};
    }
    return {BK_STAGED, staged_tile(rows, H, nd), false, false};   // exact f32 in either compute mode
}

static int gru_bwd_launch(const GruBwdPair& pr_in, int nd, hipStream_t s) {
    GruBwdPair pr = pr_in;
    for (int d = 0; d < 2; ++d) pr.d[d].ep_step = bwd_ep_step(pr.d[d].H);
    const GruBwdArgs& a = pr.d[0];
    bool vec = a.H % 4 == 0, have_wt = true, dense = true;
    for (int d = 0; d < nd; ++d) {
        vec = vec && aligned16(pr.d[d].w_hh) && (!pr.d[d].dG_next || aligned16(pr.d[d].dG_next));
        // the row-layout epilogue moves four columns per lane: every state / gate / gradient base 16-byte aligned
        const void* ptrs[] = {pr.d[d].dH_next, pr.d[d].ext, pr.d[d].ext2, pr.d[d].gates, pr.d[d].h_prev,
                              pr.d[d].dH_out, pr.d[d].dG_out};
        for (const void* q : ptrs) vec = vec && (!q || aligned16(q));
        have_wt = have_wt && pr.d[d].w_hhT != nullptr && aligned16(pr.d[d].w_hhT);
        dense = dense && !pr.d[d].nrows && !pr.d[d].nrows_next;
    }
    const BwdPlan pl = bwd_plan(a.row1 - a.row0, a.H, nd, a.row0, vec, have_wt, dense);
    if (a.gates_bf16 && (pl.kind == BK_STAGED || !pl.bf16)) {
        cpg_set_error("gru backward: gates saved as bf16 need the bf16 compute mode and the direct-to-LDS step (transposed-weight "
                      "scratch, aligned operands, option gru_bwd_dl unchanged since the forward pass)");
        return -4;
    }
    const bool pair = a.pp_next != nullptr || a.pp_out != nullptr;
    if (pair && !(pl.kind == BK_DL && pl.pair_ok)) {
        cpg_set_error("gru backward: the f16-pair step was prepared for this sequence but a launch of it is not covered (options changed "
                      "between the launches of one sequence?)");
        return -4;
    }
    const bool dgb = a.dg_bf16 != 0;   // bf16 gradient storage: PREC 2 kernels (64-deep slabs: 3H, and 3H/2 for the two-halves form, % 64)
    if (dgb && (!a.gates_bf16 || (3 * a.H) % 64 != 0 || (pl.kind == BK_DL2 && (3 * a.H / 2) % 64 != 0))) {
        cpg_set_error("gru backward: bf16 gradient storage needs bf16 saved gates and a width the 64-deep slabs divide (H %% 128 == 0)");
        return -4;
    }
    int rc = 0;
    if (pl.kind == BK_DL2) {
        rc = dgb ? launch_dl2<2>(pr, nd, s) : pl.bf16 ? launch_dl2<1>(pr, nd, s) : launch_dl2<0>(pr, nd, s);
    } else if (pl.kind == BK_DL) {
#define CPG_DL_PICK(BM, BN) (dgb ? launch_dl<BM, BN, 2>(pr, nd, s) : pl.bf16 ? launch_dl<BM, BN, 1>(pr, nd, s) : launch_dl<BM, BN, 0>(pr, nd, s))
        if (pair) rc = pl.tile.bm == 128 ? launch_dl<128, 64, 3, 4>(pr, nd, s) : pl.tile.waves == 8 ? launch_dl<64, 64, 3, 2, 4>(pr, nd, s)
                     : pl.tile.bn == 64 ? launch_dl<64, 64, 3>(pr, nd, s) : launch_dl<64, 32, 3>(pr, nd, s);
        else if (pl.tile.bm == 64 && pl.tile.bn == 64 && pl.tile.waves == 8 && pl.bf16)   // bf16 compute mode on eight waves
            rc = dgb ? launch_dl<64, 64, 2, 2, 4>(pr, nd, s) : launch_dl<64, 64, 1, 2, 4>(pr, nd, s);
        else if (pl.tile.bm == 64 && pl.tile.bn == 64) rc = CPG_DL_PICK(64, 64);
        else if (pl.tile.bm == 64) rc = CPG_DL_PICK(64, 32);
        else if (pl.tile.bn == 64) rc = CPG_DL_PICK(32, 64);
        else rc = CPG_DL_PICK(32, 32);
#undef CPG_DL_PICK
    } else if (pl.tile.bm == 64) {
        rc = launch_bwd<GB64>(pr, nd, vec, s);
    } else if (pl.tile.bn == 64) {
        rc = launch_bwd<GB32>(pr, nd, vec, s);
    } else {
        rc = launch_bwd<GB32N>(pr, nd, vec, s);
    }
    if (rc) return rc;
    CPG_LAUNCH_CHECK();
    return 0;
}

// W_hh^T as f16 pairs for the PREC 3 backward steps (pair_engine.h): out[j][2 G H] f16, k-groups of 32 in the order (column group,
// block) as the dG planes, each [32 hi | 32 lo] of W_hh[k][j] x 2^e_w.  e_w follows the matrix' largest magnitude (pair_engine.h:
// weight_exp_from_parts over wx = ex_min + H/32, filled by cpg_weight_absmax just before): max|W| 2^e_w in [2^13, 2^14), so any finite
// weight has an image (rounds 4-5 scaled by a fixed 2^8: a weight of 256 or more became an f16 infinity).  grid (H/32, G H/32), block (32, 8).
__global__ void pair_w_kernel(const float* w, int G, int H, uint16_t* out, int* ex_min) {
    __shared__ float tile[32][33];
    const float sc = pair_pow2(weight_exp_from_parts(ex_min + H / 32));
    if (blockIdx.x == 0 && blockIdx.y == 0) {   // a new sequence: no exponent seen yet
        for (int i = threadIdx.y * 32 + threadIdx.x; i < H / 32; i += 256) ex_min[i] = INT_MAX;
    }
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;   // tile[k - r0][j - c0]
    for (int i = threadIdx.y; i < 32; i += 8) tile[i][threadIdx.x] = w[(size_t)(r0 + i) * H + c0 + threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.y * 32 + threadIdx.x, jj = tid >> 3, q = tid & 7, c8 = (q & 3) * 8;
    const int blk = r0 / H, grp = (r0 - blk * H) / 32;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2h_pair(tile[c8 + 2 * i][jj] * sc, tile[c8 + 2 * i + 1][jj] * sc, hi[i], lo[i]);
    uint16_t* const d = out + (size_t)(c0 + jj) * 2 * G * H + (size_t)(G * grp + blk) * 64 + (q >> 2) * 32 + c8;
    *reinterpret_cast<uint4*>(d) = (q >> 2) ? make_uint4(lo[0], lo[1], lo[2], lo[3]) : make_uint4(hi[0], hi[1], hi[2], hi[3]);
}
int cpg_pair_w(const float* w_hh, int G, int H, uint16_t* out, int* ex_min, hipStream_t s) {
    if (!ex_min) { cpg_set_error("cpg_pair_w: the f16-pair image of W_hh needs the scratch that carries its exponent"); return -4; }
    const int rc = cpg_weight_absmax(w_hh, G * H, H, H, ex_min + H / 32, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_w_kernel, dim3(H / 32, G * H / 32), dim3(32, 8), 0, s, w_hh, G, H, out, ex_min);
    CPG_LAUNCH_CHECK();
    return 0;
}

// out[H,3H] = w[3H,H]^T
__global__ void transpose_w_kernel(const float* w, int R, int C, float* out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[i][threadIdx.x] = w[(size_t)r * C + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) out[(size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

// the same, rounded to bf16 (RNE): the B operand of the bf16-storage backward step
__global__ void transpose_w_bf16_kernel(const float* w, int R, int C, uint16_t* out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[i][threadIdx.x] = w[(size_t)r * C + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) out[(size_t)c * R + r] = (uint16_t)(cvt_pk_bf16(tile[threadIdx.x][i], 0.f) & 0xffffu);
    }
}

static int transpose_w(const float* w_hh, int H, float* wT, hipStream_t s, bool bf16 = false, bool pair = false, int* ex_min = nullptr) {
    if (pair) return cpg_pair_w(w_hh, 3, H, reinterpret_cast<uint16_t*>(wT), ex_min, s);
    if (bf16) hipLaunchKernelGGL(transpose_w_bf16_kernel, dim3(cdiv(H, 32), cdiv(3 * H, 32)), dim3(32, 8), 0, s, w_hh, 3 * H, H, reinterpret_cast<uint16_t*>(wT));
    else hipLaunchKernelGGL(transpose_w_kernel, dim3(cdiv(H, 32), cdiv(3 * H, 32)), dim3(32, 8), 0, s, w_hh, 3 * H, H, wT);
    CPG_LAUNCH_CHECK();
    return 0;
}

static int gru_step_bwd_launch(const GruBwdArgs& a, hipStream_t s) {
    GruBwdPair pr;
    pr.d[0] = a;
    pr.d[1] = a;
    return gru_bwd_launch(pr, 1, s);
}

// ------------------------------------------------------------------------------------------ reductions over dG
// dgi column c in [0,3H) lives at dG column c (c < 2H) or 3H + (c - 2H).
__device__ __forceinline__ int dgi_col(int c, int H, int lstm) { return (lstm || c < 2 * H) ? c : c + H; }

// part[chunk][v][c] = sum over rows (t,b) of the chunk with tok == v of dgi[row][c].   block: 64 columns x RL row lanes.
__global__ void dgi_by_token_kernel(const float* dG, const int32_t* tok, int rows, int H, int V, int rows_per_chunk,
                                    float* part, int lstm) {
    const int NC = lstm ? 4 * H : 3 * H;
    extern __shared__ float accs[];  // [RL][V][64]
    const int RL = blockDim.y;
    const int c = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    float* mine = accs + (size_t)ty * V * 64;
    for (int v = 0; v < V; ++v) mine[v * 64 + threadIdx.x] = 0.f;
    const int rb = blockIdx.y * rows_per_chunk, re = min(rows, rb + rows_per_chunk);
    if (c < NC) {
        const int gc = dgi_col(c, H, lstm);
        for (int row = rb + ty; row < re; row += RL) {
            const int v = tok[row];
            mine[v * 64 + threadIdx.x] += dG[(size_t)row * 4 * H + gc];
        }
    }
    __syncthreads();
    if (c < NC)
        for (int v = ty; v < V; v += RL) {
            float s = 0.f;
            for (int q = 0; q < RL; ++q) s += accs[((size_t)q * V + v) * 64 + threadIdx.x];
            part[((size_t)blockIdx.y * V + v) * NC + c] = s;
        }
}

__global__ void chunk_reduce_kernel(const float* part, int chunks, size_t n, float* out, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)c * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

// One-hot image of the step tokens with a column of ones appended: oh[row][v] = (tok[row] == v), oh[row][V] = 1.
// dG^T . oh on the matrix cores then yields the token-grouped sums AND the plain column sums in one pass over dG.
constexpr int OH_LD = 64;  // 64 columns (V+1 used): wide enough for the 64-column tiles the split-operand engine needs
__global__ void onehot_kernel(const int32_t* tok, int rows, int V, float* oh) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * OH_LD) return;
    const int row = i / OH_LD, c = i - (size_t)row * OH_LD;
    oh[i] = (c == tok[row] || c == V) ? 1.f : 0.f;
}

// R [4H, OH_LD] = dG^T . oh  ->  dtab[v][c] (+)= R[dgi_col(c)][v]; dsum[c4] (+)= R[c4][V]
__global__ void dgi_scatter_kernel(const float* R, int H, int V, int lstm, float* dtab, float* dsum, int accumulate) {
    const int NC = lstm ? 4 * H : 3 * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dtab && i < V * NC) {
        const int v = i / NC, c = i - v * NC;
        const float x = R[(size_t)dgi_col(c, H, lstm) * OH_LD + v];
        dtab[i] = accumulate ? dtab[i] + x : x;
    }
    if (dsum && i < 4 * H) {
        const float x = R[(size_t)i * OH_LD + V];
        dsum[i] = accumulate ? dsum[i] + x : x;
    }
}

// drowc[b][c] (+)= sum_t dgi[t][b][c]
__global__ void dgi_over_time_kernel(const float* dG, int T, int B, int H, float* out, int accumulate, int lstm) {
    const int NC = lstm ? 4 * H : 3 * H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * NC) return;
    const int b = i / NC, c = i % NC;
    const int gc = dgi_col(c, H, lstm);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += dG[((size_t)t * B + b) * 4 * H + gc];
    out[i] = accumulate ? out[i] + s : s;
}

// same, four columns per lane (H % 4 == 0: a 4-column group never straddles the dhn gap), 5 time steps in flight
__global__ void dgi_over_time_vec_kernel(const float* dG, int T, int B, int H, float* out, int accumulate, int lstm) {
    const int NC4 = (lstm ? 4 * H : 3 * H) / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * NC4) return;
    const int b = i / NC4, c = (i % NC4) * 4;
    const int gc = dgi_col(c, H, lstm);
    const f32x4* src = reinterpret_cast<const f32x4*>(dG + (size_t)b * 4 * H + gc);
    const size_t step = (size_t)B * H;  // in f32x4 units: one time step = B*4H floats
    f32x4 p[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) p[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (; t + 5 <= T; t += 5) {
#pragma unroll
        for (int u = 0; u < 5; ++u) p[u] += src[(size_t)(t + u) * step];
    }
    for (; t < T; ++t) p[0] += src[(size_t)t * step];
    f32x4 s = ((p[0] + p[1]) + (p[2] + p[3])) + p[4];
    f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)b * 4 * NC4 + c);
    if (accumulate) s += *dst;
    *dst = s;
}

// ------------------------------------------------------------------------------------------ launcher introspection
// Names of the kernels the launchers above would pick, in the form rocprofv3 prints them (without "void " and the argument
// list): bench.py labels its roofline object with them instead of carrying literals that a tile-policy change would
// silently desynchronise from the profile.
template <class TC>
static int tc_name(char* b, int n) {
    return snprintf(b, n, "TileCfg<%d, %d, %d, %d, %d, %d, %d>", TC::BM, TC::BN, TC::BK, TC::WM, TC::WN, TC::NSEG, TC::NT);
}

// kind 0: forward step, 1: backward step.  ndir 1 | 2 (paired biGRU launches).  have_wt: W_hh^T handed to the backward.
// Returns the length written (0 on a bad kind).
CPG_EXPORT int cpg_gru_step_kernel_name(int kind, int B, int H, int ndir, int have_wt, char* buf, int n) {
    char tc[96];
    const bool vec = H % 4 == 0;
    if (kind == 0) {
        const int bm = fwd_bm_choice(B, H, ndir);
        if (bm == 128) tc_name<GF128>(tc, sizeof tc);
        else if (bm == 64) tc_name<GF64>(tc, sizeof tc);
        else tc_name<GF32>(tc, sizeof tc);
        const int np = cpg_persist_planes();
        return snprintf(buf, n, "gru_step_fwd_kernel<%s, %s, %d>", tc, vec ? "true" : "false", np == 1 ? 1 : np == 2 ? 8 : 7);
    }
    if (kind == 1) {
        const BwdPlan pl = bwd_plan(B, H, ndir, 0, vec, have_wt != 0, true);
        if (pl.kind == BK_DL2) return snprintf(buf, n, "gru_step_bwd_dl2_kernel<%d>", pl.bf16 ? 1 : 0);
        if (pl.kind == BK_DL) return snprintf(buf, n, "gru_step_bwd_dl_kernel<%d, %d, " CPG_STR(CPG_BWD_DL_NS) ", %d, %d, %d>", pl.tile.bm, pl.tile.bn, pl.bf16 ? 1 : pl.pair_ok ? 3 : 0, pl.tile.bm == 128 ? 4 : 2, (pl.tile.waves == 8 && pl.tile.bm == 64) ? 4 : 2);
        if (pl.tile.bm == 64) tc_name<GB64>(tc, sizeof tc);
        else if (pl.tile.bn == 64) tc_name<GB32>(tc, sizeof tc);
        else tc_name<GB32N>(tc, sizeof tc);
        return snprintf(buf, n, "gru_step_bwd_kernel<%s, %s>", tc, vec ? "true" : "false");
    }
    return 0;
}

// Product form of the named step kernel: 0 exact-f32 MFMA, 1 split-bf16 engine (six bf16 MFMAs per block), 2 one bf16 MFMA
// per block (bf16 compute mode), 3 f16 pairs (three f16 MFMAs per block).
CPG_EXPORT int cpg_gru_step_kernel_is_split(int kind, int B, int H, int ndir, int have_wt) {
    if (kind == 0) { const int np = cpg_persist_planes(); return np == 1 ? 2 : np == 2 ? 3 : 1; }
    const BwdPlan pl = bwd_plan(B, H, ndir, 0, H % 4 == 0, have_wt != 0, true);
    if (pl.kind != BK_STAGED) return pl.bf16 ? 2 : (pl.kind == BK_DL && pl.pair_ok) ? 3 : 0;
    return pl.tile.bn == 64 ? 1 : 0;   // XC k-row pairs: 64-column tiles run the split engine
}

// ------------------------------------------------------------------------------------------ C ABI
// 1 when the saved gates of a [T,4,B,H] GRU sequence are bf16 elements (half the buffer): bf16 compute mode, dense batch (no
// step_rows), shape covered by the direct-to-LDS backward step.  The caller sizes / types the gates buffer by this answer and keeps
// the compute mode and options unchanged between the forward and the backward pass of a sequence (the backward refuses otherwise).
CPG_EXPORT int cpg_gru_gates_bf16(int B, int H, int ragged) { return cpg_gru_store_bf16(B, H, !ragged) ? 1 : 0; }
CPG_EXPORT int cpg_gru_dg_bf16(int B, int H, int ragged, int V) { return cpg_gru_dg_store_bf16(B, H, !ragged, V) ? 1 : 0; }

CPG_EXPORT int cpg_gru_seq_fwd(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh, const int32_t* tok,
                               const float* tab, const float* rowc, const float* dense, float* hs, float* gates,
                               int row_begin, int row_end, const int32_t* step_rows, const void* wx, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && b_hh && hs && 0 <= row_begin && row_begin < row_end && row_end <= B);
    CPG_CHECK_ARG((tok == nullptr) == (tab == nullptr));
    const size_t BH = (size_t)B * H;
    const bool gbf = gates && cpg_gru_store_bf16(B, H, step_rows == nullptr);
    if (tok && !dense && !step_rows && row_begin == 0 && row_end == B && !gbf && T <= 256 && cpg_gru_small_seq_ok(B, H)) {
        // small recurrence: the whole sequence in one launch (csrc/decode_fused.hip: gru_seq_small_fwd_kernel)
        const CpgSmallFwdDir d{w_hh, b_hh, tok, tab, rowc, hs, gates, reverse};
        return cpg_gru_small_seq_fwd(T, B, H, 1, &d, (hipStream_t)stream);
    }
    for (int p = 0; p < T; ++p) {
        const int t = reverse ? T - 1 - p : p;
        GruFwdArgs a;
        a.h_prev = reverse ? hs + (size_t)(t + 1) * BH : hs + (size_t)t * BH;
        a.h_out = reverse ? hs + (size_t)t * BH : hs + (size_t)(t + 1) * BH;
        a.w_hh = w_hh;
        a.b_hh = b_hh;
        a.tok = tok ? tok + (size_t)t * B : nullptr;
        a.tab = tab;
        a.rowc = rowc;
        a.dense = dense ? dense + (size_t)t * B * 3 * H : nullptr;
        a.gates = gates ? gate_at(gates, (size_t)t * 4 * BH, gbf) : nullptr;
        a.gates_bf16 = gbf;
        a.B = B;
        a.H = H;
        a.row0 = row_begin;
        a.row1 = row_end;
        a.nrows = step_rows ? step_rows + t : nullptr;
        a.wx = (const int*)wx;
        int rc = cpg_gru_step_fwd_launch(a, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

// One decode step (GRUDecoder.forward_sample, models/decoder.py:86-109, without the vocab projection).
CPG_EXPORT int cpg_gru_step_fwd(int B, int H, const float* w_hh, const float* b_hh, const int32_t* tok, const float* tab,
                                const float* rowc, const float* h_prev, float* h_out, const void* wx, void* stream) {
    CPG_CHECK_ARG(B > 0 && H > 0 && w_hh && b_hh && h_prev && h_out && h_prev != h_out);
    GruFwdArgs a{h_prev, w_hh, b_hh, tok, tab, rowc, nullptr, h_out, nullptr, 0, B, H, 0, B, nullptr, (const int*)wx};
    return cpg_gru_step_fwd_launch(a, (hipStream_t)stream);
}

// Scratch of the f16-pair backward step, per direction: two [B][6H] f16 plane images (ping-pong over the steps) and their two
// [B/32][H/32] exponent tables.  0: the pair step does not cover this shape / mode (pass null).
CPG_EXPORT size_t cpg_gru_bwd_pair_bytes(int rows, int H, int ndir) {
    if (rows <= 0 || H <= 0 || H % 32 != 0 || rows % 64 != 0) return 0;
    const BwdPlan pl = bwd_plan(rows, H, ndir, 0, true, true, true);
    if (!(pl.kind == BK_DL && pl.pair_ok)) return 0;
    return pair_scratch_bytes(rows, H, 3);
}

// dhs_ext: [T,B,H] time-aligned external gradients on every step's output (or null); dh_last: gradient on the final state.
// dG out [T,B,4H]; dH_scratch [2,B,H]; dh0 [B,H] (or null when the initial state needs no gradient).
// All-T planes form of the BPTT chain (pair_engine.h: ApScratch): covered where the f16-pair backward step is (whole dense batches,
// f32-grade mode) AND every consumer of the kept images has its form - the dW_hh product on pair_tn_kernel (128 x 128 tiles: H % 128),
// the one-pass input-side reduction (B % 128, H % 64).  Option gru_ap = 0 keeps the f32 gate gradients.
static bool gru_ap_ok(int B, int H, int ndir) {
    const CpgOptVal o = cpg_opt(OPT_GRU_AP);
    if (o.set && o.i == 0) return false;
    if (B <= 0 || H <= 0 || H % 128 != 0 || B % 128 != 0) return false;
    // bf16 compute mode: the form is the bf16 state copy beside the mode's bf16 gate gradients (cpg_gru_dg_bf16 with a token table)
    if (cpg_compute_mode_get() == 1) return cpg_gru_dg_store_bf16(B, H, true, 1);
    return cpg_gru_bwd_pair_bytes(B, H, ndir) > 0;
}
CPG_EXPORT size_t cpg_gru_ap_bytes(int T, int B, int H, int ndir) {
    if (T <= 0 || !gru_ap_ok(B, H, ndir)) return 0;
    if (cpg_compute_mode_get() == 1) return (size_t)T * B * H * sizeof(uint16_t);   // the bf16 copy of h_prev of every step
    return ap_scratch_bytes(T, B, H, 3);
}

static int gru_seq_bwd_impl(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                            const float* dhs_ext, const float* dh_last, float* dG, float* dH_scratch, float* dh0,
                            int row_begin, int row_end, const int32_t* step_rows, float* w_hhT_scratch, void* pair_scratch,
                            int dg_bf16, void* ap_scratch, void* stream);

CPG_EXPORT int cpg_gru_seq_bwd(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                               const float* dhs_ext, const float* dh_last, float* dG, float* dH_scratch, float* dh0,
                               int row_begin, int row_end, const int32_t* step_rows, float* w_hhT_scratch, void* pair_scratch,
                               int dg_bf16, void* stream) {
    return gru_seq_bwd_impl(T, B, H, reverse, w_hh, hs, gates, dhs_ext, dh_last, dG, dH_scratch, dh0, row_begin, row_end, step_rows,
                            w_hhT_scratch, pair_scratch, dg_bf16, nullptr, stream);
}
CPG_EXPORT int cpg_gru_seq_bwd_ap(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                                  const float* dhs_ext, const float* dh_last, float* dN, float* dH_scratch, float* dh0,
                                  float* w_hhT_scratch, void* ap, void* stream) {
    CPG_CHECK_ARG(ap && w_hhT_scratch && aligned16(ap));
    if (cpg_gru_ap_bytes(T, B, H, 1) == 0) {
        cpg_set_error("cpg_gru_seq_bwd_ap: shape / mode not covered (cpg_gru_ap_bytes answers 0: f32-grade mode, H %% 128 == 0, B %% 128 == 0)");
        return -4;
    }
    return gru_seq_bwd_impl(T, B, H, reverse, w_hh, hs, gates, dhs_ext, dh_last, dN, dH_scratch, dh0, 0, B, nullptr, w_hhT_scratch,
                            nullptr, cpg_compute_mode_get() == 1 ? 1 : 0, ap, stream);
}

static int gru_seq_bwd_impl(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                            const float* dhs_ext, const float* dh_last, float* dG, float* dH_scratch, float* dh0,
                            int row_begin, int row_end, const int32_t* step_rows, float* w_hhT_scratch, void* pair_scratch,
                            int dg_bf16, void* ap_scratch, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && hs && gates && dG && dH_scratch);
    CPG_CHECK_ARG(0 <= row_begin && row_begin < row_end && row_end <= B);
    const size_t BH = (size_t)B * H;
    if (w_hhT_scratch && !bwd_wants_wt(row_end - row_begin, H, row_begin, step_rows == nullptr)) w_hhT_scratch = nullptr;  // W_hh as stored
    const bool gbf = cpg_gru_store_bf16(B, H, step_rows == nullptr);
    const bool dgb = dg_bf16 != 0;
    CPG_CHECK_ARG(!dgb || (gbf && w_hhT_scratch && row_begin == 0 && row_end == B));   // bf16 gradient storage: whole dense batches on the direct-to-LDS step
    // f16-pair step: every launch of the sequence has the same shape, so the plan of one decides for all
    const bool allb = ap_scratch != nullptr && dgb;   // bf16 mode's all-T form: bf16 dG as ever + the bf16 state copy in ap_scratch
    const bool allt = ap_scratch != nullptr && !dgb;  // all-T planes: the caller asked cpg_gru_ap_bytes
    const bool pair = allt || (pair_scratch && w_hhT_scratch && !dgb && cpg_gru_bwd_pair_bytes(row_end - row_begin, H, 1) > 0 && row_begin % 64 == 0);
    if (!pair && !allb && !dgb && !gbf && !step_rows && row_begin == 0 && row_end == B && !w_hhT_scratch && cpg_gru_small_seq_ok(B, H)) {
        // small recurrence the direct-to-LDS step does not cover: the whole BPTT chain in one launch (gru_seq_small_bwd_kernel)
        const CpgSmallBwdDir d{w_hh, hs, gates, dhs_ext, dh_last, dG, dh0, reverse};
        return cpg_gru_small_seq_bwd(T, B, H, 1, &d, (hipStream_t)stream);
    }
    uint16_t* PP[2] = {nullptr, nullptr};
    int* EX[2] = {nullptr, nullptr};
    int* EMIN = nullptr;
    ApScratch AP{};
    const size_t ppt = (size_t)B * 6 * H, ext = (size_t)(B / 32) * (H / 32);   // per step: plane image, exponent table
    if (allt) { AP = ap_split(ap_scratch, T, B, H, 3); EMIN = AP.ex_min; }
    else if (pair) pair_split(pair_scratch, B, H, 3, PP, EX, EMIN);
    if (w_hhT_scratch) {
        int rc = transpose_w(w_hh, H, w_hhT_scratch, (hipStream_t)stream, dgb, pair, EMIN);
        if (rc) return rc;
    }
    int prev_t = -1;
    for (int p = T - 1; p >= -1; --p) {  // p = processing index of the step whose dH we form; p=-1 closes with dh0
        if (p < 0 && !dh0) break;
        const int t = p < 0 ? -1 : (reverse ? T - 1 - p : p);
        GruBwdArgs a;
        a.B = B;
        a.H = H;
        a.row0 = row_begin;
        a.row1 = row_end;
        a.w_hh = w_hh;
        a.w_hhT = w_hhT_scratch;
        a.nrows = (step_rows && t >= 0) ? step_rows + t : nullptr;
        a.nrows_next = (step_rows && prev_t >= 0) ? step_rows + prev_t : nullptr;
        a.gates_bf16 = gbf;
        a.dg_bf16 = dgb;
        const int cur = (p + 2) & 1;
        a.pp_next = nullptr; a.ex_next = nullptr; a.pp_out = nullptr; a.ex_out = nullptr; a.ex_min = EMIN;
        if (prev_t >= 0) {
            a.dG_next = allt ? dG /* non-null marker: the operand is pp_next */ : gate_at(dG, (size_t)prev_t * B * 4 * H, dgb);
            a.dH_next = dH_scratch + (size_t)(cur ^ 1) * BH;
            if (allt) { a.pp_next = AP.planes + prev_t * ppt; a.ex_next = AP.ex + prev_t * ext; }
            else if (pair) { a.pp_next = PP[cur ^ 1]; a.ex_next = EX[cur ^ 1]; }
        } else {
            a.dG_next = nullptr;
            a.dH_next = nullptr;
        }
        if (allt && p >= 0) { a.pp_out = AP.planes + (size_t)t * ppt; a.ex_out = AP.ex + (size_t)t * ext; a.hp_out = AP.hplanes + (size_t)t * B * 2 * H; }
        else if (pair && p >= 0) { a.pp_out = PP[cur]; a.ex_out = EX[cur]; }
        if (allb && p >= 0) a.hb_out = (uint16_t*)ap_scratch + (size_t)t * BH;
        a.ext2 = (p == T - 1) ? dh_last : nullptr;
        if (p >= 0) {
            a.ext = dhs_ext ? dhs_ext + (size_t)t * BH : nullptr;
            a.gates = gate_at(gates, (size_t)t * 4 * BH, gbf);
            a.h_prev = reverse ? hs + (size_t)(t + 1) * BH : hs + (size_t)t * BH;
            a.dH_out = dH_scratch + (size_t)cur * BH;
            a.dG_out = allt ? dG + (size_t)t * BH : gate_at(dG, (size_t)t * B * 4 * H, dgb);
        } else {
            a.ext = nullptr;
            a.gates = nullptr;
            a.h_prev = nullptr;
            a.dH_out = dh0;
            a.dG_out = nullptr;
        }
        int rc = gru_step_bwd_launch(a, (hipStream_t)stream);
        if (rc) return rc;
        prev_t = t;
    }
    return 0;
}

static size_t dgi_mm_workspace(int T, int B, int H) {
    return ((size_t)T * B * OH_LD + (size_t)OH_LD * 4 * H) * sizeof(float) + cpg_gemm_tn_workspace(T * B, 4 * H, OH_LD);
}

CPG_EXPORT size_t cpg_gru_wgrad_workspace(int T, int B, int H, int V) {
    size_t a = cpg_gemm_tn_workspace(T * B, 4 * H, H), a3 = cpg_gemm_tn_workspace(T * B, 3 * H, H);   // LSTM / GRU gate widths: the split-K factors differ
    if (a3 > a) a = a3;
    size_t b = cpg_colsum_workspace(T * B, 4 * H);
    int chunks = cdiv(T * B, 512);
    if (chunks > 256) chunks = 256;
    size_t c = (size_t)chunks * (V > 0 ? V : 1) * 4 * H * sizeof(float);  // sized for the 4-gate (LSTM) case too
    size_t d = dgi_mm_workspace(T, B, H);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    if (H % 128 == 0) {   // the all-T planes products (pair_tn.h): one round of 128 x 128 tiles, split over the rows
        const size_t e = cpg_pair_tn_workspace(4 * H, H, T * B), e3 = cpg_pair_tn_workspace(3 * H, H, T * B);
        m = m > e ? m : e;
        m = m > e3 ? m : e3;
    }
    return (m > d ? m : d) + 256;
}

// dW_hh[3H,H] (+)= sum_t dgh_t^T h_prev(t) ; db_hh[3H] (+)= sum dgh.
// pair_scratch (optional): the scratch the sequence's cpg_gru_seq_bwd / _biseq_bwd call was given (same B, H; that call enqueued
// earlier on a stream this one is ordered after) - the product then runs on f16 pairs, three MFMAs per block, with the gate-gradient
// columns scaled by the power of two the backward steps recorded per 32-column group (their largest magnitude over the sequence
// lands in [2^13, 2^14)) and the rows of dW_hh scaled back.
CPG_EXPORT int cpg_gru_wgrad_hh(int T, int B, int H, int reverse, const float* dG, const float* hs, float* dw_hh,
                                float* db_hh, int accumulate, void* workspace, size_t workspace_bytes, const void* pair_scratch,
                                int dg_bf16, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && dG && hs && dw_hh && workspace);
    CPG_CHECK_ARG(!dg_bf16 || !db_hh);   // bf16 gate gradients: the bias gradient comes out of cpg_gru_dgi_reduce's column sums
    const float* hprev = reverse ? hs + (size_t)B * H : hs;
    const int* exps = nullptr;
    if (pair_scratch && !dg_bf16 && cpg_compute_mode_get() != 1) {
        uint16_t* pp[2];
        int* ex[2];
        int* emin = nullptr;
        pair_split(const_cast<void*>(pair_scratch), B, H, 3, pp, ex, emin);
        exps = emin;
    }
    int rc = cpg_gemm_tn(dG, 4 * H, hprev, H, nullptr, 1.f, dw_hh, H, T * B, 3 * H, H, accumulate, (float*)workspace,
                         workspace_bytes, (hipStream_t)stream, dg_bf16, exps, H);
    if (rc || !db_hh) return rc;  // db_hh null: the caller derives it (shared r,z columns come from the token-table gradient)
    return cpg_colsum(dG, 4 * H, T * B, 3 * H, db_hh, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// All-T planes form: dW_hh[3H,H] (+)= sum over (t, b) of the kept gate-gradient planes^T x the state planes, both read as they were
// written by the sequence's cpg_gru_seq_bwd_ap / _biseq_bwd_ap call (pair_tn.h: no conversion in the loop).  The bias gradient comes
// out of cpg_gru_dgi_reduce_ap's column sums.
CPG_EXPORT int cpg_gru_wgrad_hh_ap(int T, int B, int H, const void* ap, const void* dG_bf16, float* dw_hh, int accumulate,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && ap && dw_hh && workspace);
    if (cpg_gru_ap_bytes(T, B, H, 1) == 0) {
        cpg_set_error("cpg_gru_wgrad_hh_ap: shape / mode not covered (cpg_gru_ap_bytes answers 0)");
        return -4;
    }
    if (cpg_compute_mode_get() == 1) {   // bf16 mode: A = the bf16 gate gradients [T B, 4H] (columns 0 .. 3H), B = the bf16 state copy
        CPG_CHECK_ARG(dG_bf16 != nullptr);
        return cpg_pair_tn_bf16((const uint16_t*)dG_bf16, (size_t)4 * H, (const uint16_t*)ap, (size_t)H, dw_hh, H, 3 * H, H, T * B, accumulate,
                                (float*)workspace, workspace_bytes, (hipStream_t)stream);
    }
    const ApScratch a = ap_split(const_cast<void*>(ap), T, B, H, 3);
    return cpg_pair_tn(a.planes, (size_t)6 * H, a.ex, a.ex_min, H / 32, 3, a.hplanes, (size_t)2 * H, dw_hh, H, 3 * H, H, T * B, accumulate,
                       (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// final reduction of the per-chunk partials of dgi_mfma_kernel (B/DM_ROWS chunks, fixed order)
__global__ void dgi_fused_final_kernel(const float* part_tab, const float* part_sum, int chunks, int H, int V, int lstm, float* dtab,
                                       float* dsum, int accumulate) {
    const int NC = lstm ? 4 * H : 3 * H, C4 = 4 * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dtab && i < V * NC) {
        const int v = i / NC, c = i - v * NC;
        const int gc = dgi_col(c, H, lstm);
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < chunks; ++k) s += part_tab[((size_t)k * V + v) * C4 + gc];
        dtab[i] = accumulate ? dtab[i] + s : s;
    }
    if (dsum && i < C4) {
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += part_sum[(size_t)k * C4 + i];
        dsum[i] = accumulate ? dsum[i] + s : s;
    }
}
// ---- the three input-side reductions (token-grouped sums, column sums, sums over time) with the token-grouped sums on the matrix cores: R[v][c] = sum_rows onehot[row][v] dG[row][c] is
// a product with an exact operand (0 / 1), so the exact-f32 MFMA gives f32 sums without any operand split, the one-hot operand is
// built in registers from the token ids (two compares per lane and k-step), and row V of the operand is all ones: the column
// sums come out of the same product.  Workgroup = 64 dG columns x 128 batch rows x all T steps, four waves of 32 rows; a lane
// owns four columns of the rows bw + 8 lq + ks (ks = k-step 0..7): one 16-byte load per k-step feeds the four column sets of
// the product (block column n <-> dG column 4 n + j) and the lane's running sum over time (drowc, complete rows: no partials).
// dG is read once, 1 KB per wave and k-step against 8 MFMAs: the matrix pipe can take ~9.8 TB/s of it, HBM delivers ~5.
constexpr int DM_RW = 32, DM_ROWS = 4 * DM_RW, DM_VMAX = 31;
static_assert(DM_ROWS == DM_ROWS_C && DM_VMAX == DM_VMAX_C, "cpg_gru_dg_store_bf16 states the shape limits of dgi_mfma_kernel");
// DGBF: dG holds bf16 elements (bf16 gradient storage): a lane's four columns are one 8-byte load, widened exactly to f32
// DGAP (all-T planes form, GRU): column blocks of the three recurrent gate-gradient blocks read the kept f16-pair plane images
// (ap_planes [T][B][6H], exponents ap_ex [T][B/32][H/32]: value = (hi + lo) 2^-e, exact in f32), the dn_pre block reads dG = dN [T,B,H]
template <bool ROWC, bool DGBF = false, bool DGAP = false>
__global__ __launch_bounds__(256) void dgi_mfma_kernel(const float* dG, const int32_t* tok, int T, int B, int H, int V, int lstm,
                                                        float* part_tab, float* part_sum, float* drowc, int accumulate,
                                                        const uint16_t* ap_planes = nullptr, const int* ap_ex = nullptr) {
    __shared__ f32x4 dm_red[3][2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int C4 = 4 * H, col = blockIdx.x * 64 + 4 * l15;
    const int bw = blockIdx.y * DM_ROWS + wave * DM_RW + 8 * lq;
    const bool two = V + 1 > 16;   // token rows 16..31 of the one-hot operand in use
    f32x4 acc[2][4], racc[8], x[8], xn[8];
    int tk[8], tkn[8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) racc[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int t, f32x4 (&xv)[8], int (&tv)[8]) {
        const int4 t0 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw), t1 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw + 4);
        tv[0] = t0.x; tv[1] = t0.y; tv[2] = t0.z; tv[3] = t0.w; tv[4] = t1.x; tv[5] = t1.y; tv[6] = t1.z; tv[7] = t1.w;
        if constexpr (DGAP) {
            const int q = col / H, c = col - q * H;   // (block-uniform q: 64-column blocks never straddle a gate block, H % 64 == 0)
            const int G = lstm ? 4 : 3;                // LSTM: all four blocks are kept images (dG may be null)
            if (q < G) {
                const int e = ap_ex[((size_t)t * (B / 32) + bw / 32) * (H / 32) + c / 32];
                const float sc = __builtin_bit_cast(float, (unsigned)(127 - (e == INT_MAX ? 0 : e)) << 23);
                const size_t ldp = (size_t)2 * G * H;
                const uint16_t* base = ap_planes + ((size_t)t * B + bw) * ldp + (size_t)(G * (c / 32) + q) * 64 + (c & 31);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const uint2 hi = *reinterpret_cast<const uint2*>(base + (size_t)ks * ldp), lo = *reinterpret_cast<const uint2*>(base + (size_t)ks * ldp + 32);
                    const cpg_f16x2 h0 = __builtin_bit_cast(cpg_f16x2, hi.x), h1 = __builtin_bit_cast(cpg_f16x2, hi.y);
                    const cpg_f16x2 l0 = __builtin_bit_cast(cpg_f16x2, lo.x), l1 = __builtin_bit_cast(cpg_f16x2, lo.y);
                    xv[ks] = f32x4{((float)h0[0] + (float)l0[0]) * sc, ((float)h0[1] + (float)l0[1]) * sc,
                                   ((float)h1[0] + (float)l1[0]) * sc, ((float)h1[1] + (float)l1[1]) * sc};
                }
            } else {
                const float* base = dG + ((size_t)t * B + bw) * H + c;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) xv[ks] = *reinterpret_cast<const f32x4*>(base + (size_t)ks * H);
            }
        } else if constexpr (DGBF) {
            const uint16_t* base = reinterpret_cast<const uint16_t*>(dG) + ((size_t)t * B + bw) * C4 + col;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint2 w = *reinterpret_cast<const uint2*>(base + (size_t)ks * C4);
                xv[ks] = f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                               __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
            }
        } else {
            const float* base = dG + ((size_t)t * B + bw) * C4 + col;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) xv[ks] = *reinterpret_cast<const f32x4*>(base + (size_t)ks * C4);
        }
    };
    if constexpr (DGAP) {
        // ---- all-T planes form: the operand IS f16 pairs, so the token-grouped sums run on the f16 matrix pipe - per 32 rows, token tile
        // and column set two v_mfma_f32_16x16x32_f16 (one-hot x high halves, one-hot x low halves: exact products, f32 sums) in place
        // of eight v_mfma_f32_16x16x4_f32, the step's power of two applied to the step's block sums.  (The exact-f32 form kept the
        // matrix pipe busy for half of the launch: 64 MFMAs of 32 cycles per wave and step, PMC in profiles/r05_pmc.json.)  The f32
        // dn_pre block of the GRU (dG = dN) is split here, with one power of two per wave and step (its 32 x 64 values).
        struct Raw {
            uint2 hi[8], lo[8];   // row 8 lq + ks: the four columns' high / low halves
            float sc;             // value = (hi + lo) * sc
        };
        const int qb = col / H, cb = col - qb * H;   // (block-uniform)
        const int G = lstm ? 4 : 3;
        auto fetch_raw = [&](int t, Raw& r, int (&tv)[8]) {
            const int4 t0 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw), t1 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw + 4);
            tv[0] = t0.x; tv[1] = t0.y; tv[2] = t0.z; tv[3] = t0.w; tv[4] = t1.x; tv[5] = t1.y; tv[6] = t1.z; tv[7] = t1.w;
            if (qb < G) {
                const int e = ap_ex[((size_t)t * (B / 32) + bw / 32) * (H / 32) + cb / 32];
                r.sc = __builtin_bit_cast(float, (unsigned)(127 - (e == INT_MAX ? 0 : e)) << 23);
                const size_t ldp = (size_t)2 * G * H;
                const uint16_t* base = ap_planes + ((size_t)t * B + bw) * ldp + (size_t)(G * (cb / 32) + qb) * 64 + (cb & 31);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    r.hi[ks] = *reinterpret_cast<const uint2*>(base + (size_t)ks * ldp);
                    r.lo[ks] = *reinterpret_cast<const uint2*>(base + (size_t)ks * ldp + 32);
                }
            } else {
                const float* base = dG + ((size_t)t * B + bw) * H + cb;
                f32x4 v[8];
                float vmax = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    v[ks] = *reinterpret_cast<const f32x4*>(base + (size_t)ks * H);
#pragma unroll
                    for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(v[ks][j]));
                }
                vmax = wave_max(vmax);
                // largest value -> [2^13, 2^14): high halves normal down to 2^-27 of it, nothing overflows; all-zero block: scale 1
                const int ex = vmax > 0.f ? 14 - (int)((__builtin_bit_cast(unsigned, vmax) >> 23) & 0xff) + 126 : 0;
                const int exc = max(-100, min(100, ex));
                const float up = __builtin_bit_cast(float, (unsigned)(127 + exc) << 23);
                r.sc = __builtin_bit_cast(float, (unsigned)(127 - exc) << 23);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    uint32_t h0, l0, h1, l1;
                    split2h_pair(v[ks][0] * up, v[ks][1] * up, h0, l0);
                    split2h_pair(v[ks][2] * up, v[ks][3] * up, h1, l1);
                    r.hi[ks] = make_uint2(h0, h1);
                    r.lo[ks] = make_uint2(l0, l1);
                }
            }
        };
        // column j of the lane's 8 rows x 4 columns as an MFMA B fragment (k = 8 lq + i <-> row 8 lq + i)
        auto colfrag = [&](const uint2 (&w)[8], int j) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = (j & 2) ? w[2 * i].y : w[2 * i].x, b = (j & 2) ? w[2 * i + 1].y : w[2 * i + 1].x;
                o[i] = (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            return __builtin_bit_cast(cpg_f16x8, make_uint4(o[0], o[1], o[2], o[3]));
        };
        Raw ra, rb;
        fetch_raw(0, ra, tk);
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) fetch_raw(t + 1, rb, tkn);
            if (ROWC) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const cpg_f16x2 h0 = __builtin_bit_cast(cpg_f16x2, ra.hi[ks].x), h1 = __builtin_bit_cast(cpg_f16x2, ra.hi[ks].y);
                    const cpg_f16x2 l0 = __builtin_bit_cast(cpg_f16x2, ra.lo[ks].x), l1 = __builtin_bit_cast(cpg_f16x2, ra.lo[ks].y);
                    racc[ks] += f32x4{((float)h0[0] + (float)l0[0]) * ra.sc, ((float)h0[1] + (float)l0[1]) * ra.sc,
                                      ((float)h1[0] + (float)l1[0]) * ra.sc, ((float)h1[1] + (float)l1[1]) * ra.sc};
                }
            }
            cpg_f16x8 oh[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int tokm = 16 * m + l15;
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o[i] = ((tk[2 * i] == tokm || tokm == V) ? 0x3C00u : 0u) | ((tk[2 * i + 1] == tokm || tokm == V) ? 0x3C000000u : 0u);
                oh[m] = __builtin_bit_cast(cpg_f16x8, make_uint4(o[0], o[1], o[2], o[3]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cpg_f16x8 bh = colfrag(ra.hi, j), bl = colfrag(ra.lo, j);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (m == 1 && !two) break;
                    f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[m], bl, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[m], bh, st, 0, 0, 0);
                    acc[m][j] += st * ra.sc;
                }
            }
            ra = rb;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) tk[ks] = tkn[ks];
        }
    } else if constexpr (DGBF) {
        // ---- bf16 gradient storage (bf16 compute mode): the operand IS bf16, the one-hot operand is exact in bf16 - ONE v_mfma_f32_16x16x32_bf16
        // per 32 rows, token tile and column set in place of eight exact-f32 MFMAs on widened values (same products, f32 sums)
        uint2 wa[8], wb[8];
        auto fetch_bf = [&](int t, uint2 (&w)[8], int (&tv)[8]) {
            const int4 t0 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw), t1 = *reinterpret_cast<const int4*>(tok + (size_t)t * B + bw + 4);
            tv[0] = t0.x; tv[1] = t0.y; tv[2] = t0.z; tv[3] = t0.w; tv[4] = t1.x; tv[5] = t1.y; tv[6] = t1.z; tv[7] = t1.w;
            const uint16_t* base = reinterpret_cast<const uint16_t*>(dG) + ((size_t)t * B + bw) * C4 + col;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w[ks] = *reinterpret_cast<const uint2*>(base + (size_t)ks * C4);
        };
        auto colfrag = [&](const uint2 (&w)[8], int j) {   // column j of the lane's 8 rows x 4 columns: k = 8 lq + i <-> row 8 lq + i
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = (j & 2) ? w[2 * i].y : w[2 * i].x, b = (j & 2) ? w[2 * i + 1].y : w[2 * i + 1].x;
                o[i] = (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            return __builtin_bit_cast(cpg_bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
        };
        fetch_bf(0, wa, tk);
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) fetch_bf(t + 1, wb, tkn);
            if (ROWC) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    racc[ks] += f32x4{__builtin_bit_cast(float, wa[ks].x << 16), __builtin_bit_cast(float, wa[ks].x & 0xffff0000u),
                                      __builtin_bit_cast(float, wa[ks].y << 16), __builtin_bit_cast(float, wa[ks].y & 0xffff0000u)};
            }
            cpg_bf16x8 oh[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int tokm = 16 * m + l15;
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o[i] = ((tk[2 * i] == tokm || tokm == V) ? 0x3F80u : 0u) | ((tk[2 * i + 1] == tokm || tokm == V) ? 0x3F800000u : 0u);
                oh[m] = __builtin_bit_cast(cpg_bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cpg_bf16x8 bj = colfrag(wa, j);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oh[0], bj, acc[0][j], 0, 0, 0);
                if (two) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oh[1], bj, acc[1][j], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                wa[ks] = wb[ks];
                tk[ks] = tkn[ks];
            }
        }
    } else {
    fetch(0, x, tk);
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) fetch(t + 1, xn, tkn);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ROWC) racc[ks] += x[ks];
            const float a0 = (tk[ks] == l15 || l15 == V) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, x[ks][j], acc[0][j], 0, 0, 0);
            if (two) {
                const float a1 = (tk[ks] == 16 + l15 || 16 + l15 == V) ? 1.f : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, x[ks][j], acc[1][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            x[ks] = xn[ks];
            tk[ks] = tkn[ks];
        }
    }
    }
    if (ROWC) {   // sums over time (the GRU's dhn block, columns [2H,3H) of dG, is not an input-side gradient)
        const bool is_dgi = lstm || col < 2 * H || col >= 3 * H;
        const int NC = lstm ? 4 * H : 3 * H, dcol = (lstm || col < 2 * H) ? col : col - H;
        if (is_dgi) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f32x4* o = reinterpret_cast<f32x4*>(drowc + (size_t)(bw + ks) * NC + dcol);
                *o = accumulate ? *o + racc[ks] : racc[ks];
            }
        }
    }
    // waves 1..3 hand their block sums to wave 0, which adds them in a fixed order and writes the workgroup's partials
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) dm_red[wave - 1][m][j][lane] = acc[m][j];
    }
    __syncthreads();
    if (wave == 0) {
        const size_t chunk = blockIdx.y;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m == 1 && !two) break;
            f32x4 sacc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc[j] = ((acc[m][j] + dm_red[0][m][j][lane]) + dm_red[1][m][j][lane]) + dm_red[2][m][j][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = 16 * m + 4 * lq + r;
                const f32x4 val = f32x4{sacc[0][r], sacc[1][r], sacc[2][r], sacc[3][r]};
                if (v < V) *reinterpret_cast<f32x4*>(part_tab + (chunk * V + v) * C4 + col) = val;
                else if (v == V) *reinterpret_cast<f32x4*>(part_sum + chunk * C4 + col) = val;
            }
        }
    }
}
static size_t dgi_mfma_workspace(int B, int H, int V) { return (size_t)(B / DM_ROWS) * (V + 1) * 4 * H * sizeof(float); }
// option dgi_mode = "gemm" keeps the one-hot product + over-time pass (the form of every other shape).
static bool dgi_mfma_ok(int B, int H, int V, const float* dG, const int32_t* tok, const float* drowc, size_t ws_bytes) {
    const CpgOptVal& o = cpg_opt(OPT_DGI_MODE);
    if (o.set && !strcmp(o.s, "gemm")) return false;
    return tok && V > 0 && V <= DM_VMAX && H % 64 == 0 && B % DM_ROWS == 0 && aligned16(dG) && aligned16(tok) &&
           (!drowc || aligned16(drowc)) && ws_bytes >= dgi_mfma_workspace(B, H, V);
}

// Input-side reductions of dgi = [dr_pre, dz_pre, dn_pre]:
//   dtab[V,3H]  (+)= sum over (t,b) with tok[t,b]==v     (gradient of the token table; null to skip)
//   drowc[B,3H] (+)= sum over t                          (gradient of the constant-over-time term; null to skip)
int cpg_dgi_reduce_impl(int T, int B, int H, int lstm, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                        float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream, int dg_bf16) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && dG);
    const int NC = lstm ? 4 * H : 3 * H;
    hipStream_t s = (hipStream_t)stream;
    const int rows = T * B;
    if (dg_bf16 && !((dtab || dsum) && dgi_mfma_ok(B, H, V, dG, tok, drowc, workspace_bytes))) {
        cpg_set_error("cpg_gru_dgi_reduce: bf16 gate gradients are read by the one-pass matrix-core reduction only (token table of <= %d "
                      "rows, batch %% %d == 0, H %% 64 == 0, option dgi_mode unset) - cpg_gru_dg_bf16 states the shapes", DM_VMAX, DM_ROWS);
        return -4;
    }
    if ((dtab || dsum) && dgi_mfma_ok(B, H, V, dG, tok, drowc, workspace_bytes)) {
        CPG_CHECK_ARG(workspace);
        const int chunks = B / DM_ROWS;
        float* part_tab = (float*)workspace;
        float* part_sum = part_tab + (size_t)chunks * V * 4 * H;
        const dim3 grid(4 * H / 64, chunks);
        if (dg_bf16) {
            if (drowc) hipLaunchKernelGGL((dgi_mfma_kernel<true, true>), grid, dim3(256), 0, s, dG, tok, T, B, H, V, lstm, part_tab, part_sum, drowc, accumulate);
            else hipLaunchKernelGGL((dgi_mfma_kernel<false, true>), grid, dim3(256), 0, s, dG, tok, T, B, H, V, lstm, part_tab, part_sum, drowc, accumulate);
        } else if (drowc) hipLaunchKernelGGL(dgi_mfma_kernel<true>, grid, dim3(256), 0, s, dG, tok, T, B, H, V, lstm, part_tab, part_sum, drowc, accumulate);
        else hipLaunchKernelGGL(dgi_mfma_kernel<false>, grid, dim3(256), 0, s, dG, tok, T, B, H, V, lstm, part_tab, part_sum, drowc, accumulate);
        CPG_LAUNCH_CHECK();
        const int m = V * NC > 4 * H ? V * NC : 4 * H;
        hipLaunchKernelGGL(dgi_fused_final_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, (const float*)part_tab, (const float*)part_sum,
                           chunks, H, V, lstm, dtab, dsum, accumulate);
        CPG_LAUNCH_CHECK();
        return 0;
    }
    if ((dtab || dsum) && V + 1 <= OH_LD && workspace_bytes >= dgi_mm_workspace(T, B, H)) {
        // R[4H,OH_LD] = dG^T . onehot1 : token-grouped sums (columns 0..V-1) and column sums of dG (column V), dG read once
        CPG_CHECK_ARG(tok && V > 0 && workspace);
        float* oh = (float*)workspace;
        float* R = oh + (size_t)rows * OH_LD;
        float* gws = R + (size_t)OH_LD * 4 * H;
        const size_t n = (size_t)rows * OH_LD;
        hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tok, rows, V, oh);
        CPG_LAUNCH_CHECK();
        int rc = cpg_gemm_tn(dG, 4 * H, oh, OH_LD, nullptr, 1.f, R, OH_LD, rows, 4 * H, OH_LD, 0, gws,
                             workspace_bytes - ((char*)gws - (char*)workspace), s);
        if (rc) return rc;
        const int m = V * NC > 4 * H ? V * NC : 4 * H;
        hipLaunchKernelGGL(dgi_scatter_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, R, H, V, lstm, dtab, dsum, accumulate);
        CPG_LAUNCH_CHECK();
    } else {
        if (dtab) {
            CPG_CHECK_ARG(tok && V > 0 && workspace);
            int chunks = cdiv(rows, 512);
            if (chunks > 256) chunks = 256;
            const int rpc = cdiv(rows, chunks);
            chunks = cdiv(rows, rpc);
            const size_t n = (size_t)V * NC;
            if (workspace_bytes < n * chunks * sizeof(float)) {
                cpg_set_error("cpg_gru_dgi_reduce: workspace too small");
                return -3;
            }
            int RL = 4;
            while (RL > 1 && (size_t)RL * V * 64 * sizeof(float) > 96 * 1024) RL >>= 1;
            const size_t smem = (size_t)RL * V * 64 * sizeof(float);
            if (smem > 150 * 1024) {
                cpg_set_error("cpg_gru_dgi_reduce: vocabulary of %d rows does not fit the LDS accumulators", V);
                return -4;
            }
            hipLaunchKernelGGL(dgi_by_token_kernel, dim3(cdiv(NC, 64), chunks), dim3(64, RL), smem, s, dG, tok, rows, H, V, rpc,
                               (float*)workspace, lstm);
            CPG_LAUNCH_CHECK();
            hipLaunchKernelGGL(chunk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace,
                               chunks, n, dtab, accumulate);
            CPG_LAUNCH_CHECK();
        }
        if (dsum) {
            int rc = cpg_colsum(dG, 4 * H, rows, 4 * H, dsum, accumulate, (float*)workspace, workspace_bytes, s);
            if (rc) return rc;
        }
    }
    if (drowc) {
        const size_t n = (size_t)B * NC;
        if (H % 4 == 0 && aligned16(dG) && aligned16(drowc))
            hipLaunchKernelGGL(dgi_over_time_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, dG, T, B, H,
                               drowc, accumulate, lstm);
        else
            hipLaunchKernelGGL(dgi_over_time_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dG, T, B, H, drowc,
                               accumulate, lstm);
        CPG_LAUNCH_CHECK();
    }
    return 0;
}

// LSTM extension, all-T planes form: all four gate-gradient blocks are read from the kept images (ap: cpg_lstm_ap_bytes); results as
// cpg_lstm_dgi_reduce.
extern "C" size_t cpg_lstm_ap_bytes(int T, int B, int H);
CPG_EXPORT int cpg_lstm_dgi_reduce_ap(int T, int B, int H, const void* ap, const int32_t* tok, int V, float* dtab, float* dsum, float* drowc,
                                      int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && ap && tok && workspace && (dtab || dsum));
    if (cpg_lstm_ap_bytes(T, B, H) == 0 || !(V > 0 && V <= DM_VMAX) || !aligned16(tok) || (drowc && !aligned16(drowc)) ||
        workspace_bytes < dgi_mfma_workspace(B, H, V)) {
        cpg_set_error("cpg_lstm_dgi_reduce_ap: not covered (cpg_lstm_ap_bytes, token table of 1..%d rows, aligned operands, workspace of "
                      "cpg_gru_wgrad_workspace bytes)", DM_VMAX);
        return -4;
    }
    const ApScratch a = ap_split(const_cast<void*>(ap), T, B, H, 4);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = B / DM_ROWS;
    float* part_tab = (float*)workspace;
    float* part_sum = part_tab + (size_t)chunks * V * 4 * H;
    const dim3 grid(4 * H / 64, chunks);
    if (drowc) hipLaunchKernelGGL((dgi_mfma_kernel<true, false, true>), grid, dim3(256), 0, s, (const float*)nullptr, tok, T, B, H, V, 1, part_tab, part_sum, drowc, accumulate, a.planes, a.ex);
    else hipLaunchKernelGGL((dgi_mfma_kernel<false, false, true>), grid, dim3(256), 0, s, (const float*)nullptr, tok, T, B, H, V, 1, part_tab, part_sum, drowc, accumulate, a.planes, a.ex);
    CPG_LAUNCH_CHECK();
    const int m = V * 4 * H;
    hipLaunchKernelGGL(dgi_fused_final_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, (const float*)part_tab, (const float*)part_sum, chunks, H, V, 1,
                       dtab, dsum, accumulate);
    CPG_LAUNCH_CHECK();
    return 0;
}

// All-T planes form of the input-side reductions: the three recurrent gate-gradient blocks are read from the kept plane images of
// `ap`, the n-gate's input-side block from dN [T,B,H]; results as cpg_gru_dgi_reduce (dsum[4H] = column sums of dr, dz, dhn, dn).
CPG_EXPORT int cpg_gru_dgi_reduce_ap(int T, int B, int H, const void* ap, const float* dN, const int32_t* tok, int V, float* dtab,
                                     float* dsum, float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && ap && dN && tok && workspace && (dtab || dsum));
    if (cpg_gru_ap_bytes(T, B, H, 1) == 0 || !(V > 0 && V <= DM_VMAX) || !aligned16(dN) || !aligned16(tok) || (drowc && !aligned16(drowc)) ||
        workspace_bytes < dgi_mfma_workspace(B, H, V)) {
        cpg_set_error("cpg_gru_dgi_reduce_ap: not covered (cpg_gru_ap_bytes, token table of 1..%d rows, aligned operands, workspace of "
                      "cpg_gru_wgrad_workspace bytes)", DM_VMAX);
        return -4;
    }
    const ApScratch a = ap_split(const_cast<void*>(ap), T, B, H, 3);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = B / DM_ROWS;
    float* part_tab = (float*)workspace;
    float* part_sum = part_tab + (size_t)chunks * V * 4 * H;
    const dim3 grid(4 * H / 64, chunks);
    if (drowc) hipLaunchKernelGGL((dgi_mfma_kernel<true, false, true>), grid, dim3(256), 0, s, dN, tok, T, B, H, V, 0, part_tab, part_sum, drowc, accumulate, a.planes, a.ex);
    else hipLaunchKernelGGL((dgi_mfma_kernel<false, false, true>), grid, dim3(256), 0, s, dN, tok, T, B, H, V, 0, part_tab, part_sum, drowc, accumulate, a.planes, a.ex);
    CPG_LAUNCH_CHECK();
    const int m = V * 3 * H > 4 * H ? V * 3 * H : 4 * H;
    hipLaunchKernelGGL(dgi_fused_final_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, (const float*)part_tab, (const float*)part_sum, chunks, H, V, 0,
                       dtab, dsum, accumulate);
    CPG_LAUNCH_CHECK();
    return 0;
}

CPG_EXPORT int cpg_gru_dgi_reduce(int T, int B, int H, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                                  float* drowc, int accumulate, void* workspace, size_t workspace_bytes, int dg_bf16, void* stream) {
    return cpg_dgi_reduce_impl(T, B, H, 0, dG, tok, V, dtab, dsum, drowc, accumulate, workspace, workspace_bytes, stream, dg_bf16);
}

static void fill_fwd(GruFwdArgs& a, int t, int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                     const int32_t* tok, const float* tab, const float* dense, float* hs, float* gates) {
    const size_t BH = (size_t)B * H;
    a.h_prev = reverse ? hs + (size_t)(t + 1) * BH : hs + (size_t)t * BH;
    a.h_out = reverse ? hs + (size_t)t * BH : hs + (size_t)(t + 1) * BH;
    a.w_hh = w_hh;
    a.b_hh = b_hh;
    a.tok = tok ? tok + (size_t)t * B : nullptr;
    a.tab = tab;
    a.rowc = nullptr;
    a.dense = dense ? dense + (size_t)t * B * 3 * H : nullptr;
    const bool gbf = gates && cpg_gru_store_bf16(B, H, true);
    a.gates = gates ? gate_at(gates, (size_t)t * 4 * BH, gbf) : nullptr;
    a.gates_bf16 = gbf;
    a.B = B;
    a.H = H;
    a.row0 = 0;
    a.row1 = B;
    a.nrows = nullptr;
}

// Both directions of one biGRU layer (models/encoder.py:25-30,42) in lock step: launch p runs time p of the forward
// direction and time T-1-p of the reverse direction.  *_f / *_r: per-direction arguments as in cpg_gru_seq_fwd.
CPG_EXPORT int cpg_gru_biseq_fwd(int T, int B, int H, const float* w_hh_f, const float* b_hh_f, const float* w_hh_r,
                                 const float* b_hh_r, const int32_t* tok, const float* tab_f, const float* tab_r,
                                 const float* dense_f, const float* dense_r, float* hs_f, float* hs_r, float* gates_f,
                                 float* gates_r, const void* wx_f, const void* wx_r, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh_f && b_hh_f && w_hh_r && b_hh_r && hs_f && hs_r);
    CPG_CHECK_ARG((tok == nullptr) == (tab_f == nullptr) && (tab_f == nullptr) == (tab_r == nullptr));
    CPG_CHECK_ARG((dense_f == nullptr) == (dense_r == nullptr) && (gates_f == nullptr) == (gates_r == nullptr));
    if (tok && !dense_f && !(gates_f && cpg_gru_store_bf16(B, H, true)) && T <= 256 && cpg_gru_small_seq_ok(B, H)) {   // small recurrence: one launch for both directions
        const CpgSmallFwdDir d[2] = {{w_hh_f, b_hh_f, tok, tab_f, nullptr, hs_f, gates_f, 0}, {w_hh_r, b_hh_r, tok, tab_r, nullptr, hs_r, gates_r, 1}};
        return cpg_gru_small_seq_fwd(T, B, H, 2, d, (hipStream_t)stream);
    }
    for (int p = 0; p < T; ++p) {
        GruFwdPair pr;
        fill_fwd(pr.d[0], p, T, B, H, 0, w_hh_f, b_hh_f, tok, tab_f, dense_f, hs_f, gates_f);
        fill_fwd(pr.d[1], T - 1 - p, T, B, H, 1, w_hh_r, b_hh_r, tok, tab_r, dense_r, hs_r, gates_r);
        pr.d[0].wx = (const int*)wx_f;
        pr.d[1].wx = (const int*)wx_r;
        int rc = gru_fwd_launch(pr, 2, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

// BPTT of both directions in lock step (no initial-state gradient: the encoder starts from h0 = 0).
// dhs_ext_* [T,B,H] time-aligned (null = zeros); dh_last_* [B,H] gradient on the direction's final state (null = zeros);
// dG_* [T,B,4H]; scratch_* [2,B,H].
static int gru_biseq_bwd_impl(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                              const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                              const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f,
                              float* dG_r, float* scratch_f, float* scratch_r, float* w_hhT_scratch_f,
                              float* w_hhT_scratch_r, void* pair_scratch_f, void* pair_scratch_r, int dg_bf16, void* ap_f, void* ap_r,
                              void* stream);
CPG_EXPORT int cpg_gru_biseq_bwd(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                                 const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                 const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f,
                                 float* dG_r, float* scratch_f, float* scratch_r, float* w_hhT_scratch_f,
                                 float* w_hhT_scratch_r, void* pair_scratch_f, void* pair_scratch_r, int dg_bf16, void* stream) {
    return gru_biseq_bwd_impl(T, B, H, w_hh_f, w_hh_r, hs_f, hs_r, gates_f, gates_r, dhs_ext_f, dhs_ext_r, dh_last_f, dh_last_r, dG_f, dG_r,
                              scratch_f, scratch_r, w_hhT_scratch_f, w_hhT_scratch_r, pair_scratch_f, pair_scratch_r, dg_bf16, nullptr,
                              nullptr, stream);
}
CPG_EXPORT int cpg_gru_biseq_bwd_ap(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                                    const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                    const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dN_f,
                                    float* dN_r, float* scratch_f, float* scratch_r, float* w_hhT_scratch_f,
                                    float* w_hhT_scratch_r, void* ap_f, void* ap_r, void* stream) {
    CPG_CHECK_ARG(ap_f && ap_r && w_hhT_scratch_f && w_hhT_scratch_r && aligned16(ap_f) && aligned16(ap_r));
    if (cpg_gru_ap_bytes(T, B, H, 2) == 0) {
        cpg_set_error("cpg_gru_biseq_bwd_ap: shape / mode not covered (cpg_gru_ap_bytes answers 0)");
        return -4;
    }
    return gru_biseq_bwd_impl(T, B, H, w_hh_f, w_hh_r, hs_f, hs_r, gates_f, gates_r, dhs_ext_f, dhs_ext_r, dh_last_f, dh_last_r, dN_f, dN_r,
                              scratch_f, scratch_r, w_hhT_scratch_f, w_hhT_scratch_r, nullptr, nullptr, cpg_compute_mode_get() == 1 ? 1 : 0,
                              ap_f, ap_r, stream);
}
static int gru_biseq_bwd_impl(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                              const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                              const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f,
                              float* dG_r, float* scratch_f, float* scratch_r, float* w_hhT_scratch_f,
                              float* w_hhT_scratch_r, void* pair_scratch_f, void* pair_scratch_r, int dg_bf16, void* ap_f, void* ap_r,
                              void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh_f && w_hh_r && hs_f && hs_r && gates_f && gates_r && dG_f && dG_r);
    CPG_CHECK_ARG(scratch_f && scratch_r && (w_hhT_scratch_f == nullptr) == (w_hhT_scratch_r == nullptr));
    if (w_hhT_scratch_f && !bwd_wants_wt(B, H, 0, true)) w_hhT_scratch_f = w_hhT_scratch_r = nullptr;  // W_hh as stored
    const bool dgb = dg_bf16 != 0;
    CPG_CHECK_ARG(!dgb || (cpg_gru_store_bf16(B, H, true) && w_hhT_scratch_f));
    const bool allb = ap_f != nullptr && ap_r != nullptr && dgb;   // bf16 mode: bf16 state copies
    const bool allt = ap_f != nullptr && ap_r != nullptr && !dgb;
    const bool pair = allt || (pair_scratch_f && pair_scratch_r && w_hhT_scratch_f && !dgb && cpg_gru_bwd_pair_bytes(B, H, 2) > 0);
    if (!pair && !allb && !dgb && !cpg_gru_store_bf16(B, H, true) && !w_hhT_scratch_f && cpg_gru_small_seq_ok(B, H)) {
        // small recurrence: both directions' BPTT chains in one launch (csrc/decode_fused.hip: gru_seq_small_bwd_kernel)
        const CpgSmallBwdDir d[2] = {{w_hh_f, hs_f, gates_f, dhs_ext_f, dh_last_f, dG_f, nullptr, 0},
                                     {w_hh_r, hs_r, gates_r, dhs_ext_r, dh_last_r, dG_r, nullptr, 1}};
        return cpg_gru_small_seq_bwd(T, B, H, 2, d, (hipStream_t)stream);
    }
    uint16_t* PP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int* EXP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int* EMIN[2] = {nullptr, nullptr};
    ApScratch AP[2] = {};
    const size_t ppt = (size_t)B * 6 * H, ext = (size_t)(B / 32) * (H / 32);
    if (allt) {
        AP[0] = ap_split(ap_f, T, B, H, 3);
        AP[1] = ap_split(ap_r, T, B, H, 3);
        EMIN[0] = AP[0].ex_min;
        EMIN[1] = AP[1].ex_min;
    } else if (pair) {
        pair_split(pair_scratch_f, B, H, 3, PP[0], EXP[0], EMIN[0]);
        pair_split(pair_scratch_r, B, H, 3, PP[1], EXP[1], EMIN[1]);
    }
    if (w_hhT_scratch_f) {
        int rc = transpose_w(w_hh_f, H, w_hhT_scratch_f, (hipStream_t)stream, dgb, pair, EMIN[0]);
        if (!rc) rc = transpose_w(w_hh_r, H, w_hhT_scratch_r, (hipStream_t)stream, dgb, pair, EMIN[1]);
        if (rc) return rc;
    }
    const float* WT[2] = {w_hhT_scratch_f, w_hhT_scratch_r};
    const size_t BH = (size_t)B * H;
    const float* W[2] = {w_hh_f, w_hh_r};
    const float* HS[2] = {hs_f, hs_r};
    const float* GT[2] = {gates_f, gates_r};
    const float* EX[2] = {dhs_ext_f, dhs_ext_r};
    const float* LAST[2] = {dh_last_f, dh_last_r};
    float* DG[2] = {dG_f, dG_r};
    float* SC[2] = {scratch_f, scratch_r};
    const bool gbf = cpg_gru_store_bf16(B, H, true);
    int prev_t[2] = {-1, -1};
    for (int p = T - 1; p >= 0; --p) {
        GruBwdPair pr;
        const int cur = (p + 2) & 1;
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - p : p;
            GruBwdArgs& a = pr.d[d];
            a.nrows = nullptr;
            a.nrows_next = nullptr;
            a.gates_bf16 = gbf;
            a.dg_bf16 = dgb;
            a.B = B;
            a.H = H;
            a.row0 = 0;
            a.row1 = B;
            a.w_hh = W[d];
            a.w_hhT = WT[d];
            a.pp_next = nullptr; a.ex_next = nullptr; a.ex_min = EMIN[d];
            a.pp_out = allt ? AP[d].planes + (size_t)t * ppt : pair ? PP[d][cur] : nullptr;
            a.ex_out = allt ? AP[d].ex + (size_t)t * ext : pair ? EXP[d][cur] : nullptr;
            a.hp_out = allt ? AP[d].hplanes + (size_t)t * B * 2 * H : nullptr;
            a.hb_out = allb ? (uint16_t*)(d ? ap_r : ap_f) + (size_t)t * BH : nullptr;
            if (prev_t[d] >= 0) {
                a.dG_next = allt ? DG[d] : gate_at(DG[d], (size_t)prev_t[d] * B * 4 * H, dgb);
                a.dH_next = SC[d] + (size_t)(cur ^ 1) * BH;
                if (allt) { a.pp_next = AP[d].planes + (size_t)prev_t[d] * ppt; a.ex_next = AP[d].ex + (size_t)prev_t[d] * ext; }
                else if (pair) { a.pp_next = PP[d][cur ^ 1]; a.ex_next = EXP[d][cur ^ 1]; }
            } else {
                a.dG_next = nullptr;
                a.dH_next = nullptr;
            }
            a.ext = EX[d] ? EX[d] + (size_t)t * BH : nullptr;
            a.ext2 = (p == T - 1) ? LAST[d] : nullptr;   // gradient on the direction's final state enters at its last step
            a.gates = gate_at(GT[d], (size_t)t * 4 * BH, gbf);
            a.h_prev = d ? HS[d] + (size_t)(t + 1) * BH : HS[d] + (size_t)t * BH;
            a.dH_out = SC[d] + (size_t)cur * BH;
            a.dG_out = allt ? DG[d] + (size_t)t * BH : gate_at(DG[d], (size_t)t * B * 4 * H, dgb);
            prev_t[d] = t;
        }
        int rc = gru_bwd_launch(pr, 2, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

