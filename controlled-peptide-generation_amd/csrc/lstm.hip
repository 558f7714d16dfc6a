// LSTM sequence forward / backward for gfx950 - same structure as gru.hip (one fused launch per time step, the
// [B,T,4H] input pre-activation rebuilt per element from token table + row constant + dense term).
//
// NOT a reference component: IBM/controlled-peptide-generation has no LSTM.  BASELINE.json's configs name an
// LSTM cell, so the build offers one; its semantics are torch.nn.LSTM's (gate row order i,f,g,o;
// c' = f*c + i*g ; h' = o*tanh(c')) and it is pinned to torch.nn.LSTM only (oracle/lstm.py, tests/test_lstm.py).
//
// State slabs hs, cs [(T+1),B,H] (layout as in gru.hip).  gates [T,4,B,H] = i,f,g,o.  dG [T,B,4H] = pre-activation
// gradients (identical for the input side and the hidden side).
#include "gemm_core.h"
#include "cpg_internal.h"
#include "pair_engine.h"
#include <string.h>
#include <stdlib.h>

struct LstmFwdArgs {
    const float* h_prev;
    const float* c_prev;
    const float* w_hh;   // [4H,H]
    const float* b_hh;   // [4H]
    const int32_t* tok;
    const float* tab;    // [V,4H]
    const float* rowc;   // [B,4H]
    const float* dense;  // [B,4H]
    float* h_out;
    float* c_out;
    float* gates;        // [4,B,H] or null
    int B, H;
};

// h . W_hh^T of the forward step on the split-operand bf16 engine (gemm_core.h) for tiles of at least 64 rows, like the GRU
// forward step; smaller tiles keep the exact-f32 MFMA.
template <class TC, bool VEC>
using LstmFwdLoop = MainLoop<TC, true, true, VEC, VEC, false, (TC::BM >= 64 && TC::BK == 32) ? 7 : 0>;

template <class TC, bool VEC>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmFwdArgs g) {
    const int H = g.H, B = g.B;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int m0 = by * TC::BM, j0 = bx * (TC::BN / 4);
    static_assert(TC::NI % 4 == 0, "wave tile holds i,f,g,o blocks");
    constexpr int NJ = TC::NI / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % TC::WN;
    float gi[NJ][TC::MI][4][4], cp[NJ][TC::MI][4];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
        const int j = j0 + (wn * NJ + jb) * 16 + (lane & 15);
        const int jc = (j < H) ? j : 0;
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                const int rc = (row < B) ? row : 0;
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (g.tok) {
                    const float* t = g.tab + (size_t)g.tok[rc] * 4 * H;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] += t[q * H + jc];
                }
                if (g.rowc) {
                    const float* t = g.rowc + (size_t)rc * 4 * H;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] += t[q * H + jc];
                }
                if (g.dense) {
                    const float* t = g.dense + (size_t)rc * 4 * H;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] += t[q * H + jc];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) gi[jb][mi][r][q] = a[q];
                cp[jb][mi][r] = g.c_prev[(size_t)rc * H + jc];
            }
    }
    OpA a{g.h_prev, H, m0, B, nullptr, 1.f};
    OpB b{g.w_hh, H, j0, H, H, nullptr, 1.f};
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    LstmFwdLoop<TC, VEC>::run(a, b, H, acc);
    const size_t BH = (size_t)B * H;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
        const int j = j0 + (wn * NJ + jb) * 16 + (lane & 15);
        if (j >= H) continue;
        const float b_i = g.b_hh[j], b_f = g.b_hh[H + j], b_g = g.b_hh[2 * H + j], b_o = g.b_hh[3 * H + j];
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                if (row >= B) continue;
                const float ig = sigmoidf_(gi[jb][mi][r][0] + (acc[mi][jb * 4 + 0][r] + b_i));
                const float fg = sigmoidf_(gi[jb][mi][r][1] + (acc[mi][jb * 4 + 1][r] + b_f));
                const float gg = tanhf(gi[jb][mi][r][2] + (acc[mi][jb * 4 + 2][r] + b_g));
                const float og = sigmoidf_(gi[jb][mi][r][3] + (acc[mi][jb * 4 + 3][r] + b_o));
                const float cn = fg * cp[jb][mi][r] + ig * gg;
                const size_t o = (size_t)row * H + j;
                g.c_out[o] = cn;
                g.h_out[o] = og * tanhf(cn);
                if (g.gates) {
                    __builtin_nontemporal_store(ig, g.gates + o);
                    __builtin_nontemporal_store(fg, g.gates + BH + o);
                    __builtin_nontemporal_store(gg, g.gates + 2 * BH + o);
                    __builtin_nontemporal_store(og, g.gates + 3 * BH + o);
                }
            }
    }
}

struct LstmBwdArgs {
    const float* dG_next;  // [B,4H] of the step processed just before (s+1), null on the first launch
    const float* w_hh;     // [4H,H]
    const float* w_hhT;    // [H,4H] = w_hh^T, or null (direct-to-LDS kernel only)
    const float* dC_next;  // [B,H] carried cell gradient dc_{s+1} * f_{s+1}, null on the first launch
    const float* ext;      // [B,H] external gradient on h_s
    const float* ext2;     // [B,H] second external gradient (the final-state gradient, on the direction's first launch) or null
    const float* gates;    // [4,B,H] of step s; null on the closing launch (emits dh0 / dc0)
    const float* c_prev;   // [B,H] c_{s-1}
    const float* c_cur;    // [B,H] c_s
    float* dH_out;         // closing launch: dh0
    float* dC_out;         // [B,H] dc_s * f_s   (closing launch: dc0 = dC_next passthrough)
    float* dG_out;         // [B,4H]
    int B, H;
    // f16-pair hand-off of the direct-to-LDS step (PREC 3; pair_engine.h): dG once more as [B][8H] f16 pairs x 2^e per 32 x 32 group
    const uint16_t* pp_next;
    const int* ex_next;
    uint16_t* pp_out;
    int* ex_out;
    int* ex_min;
    int ap;                // all-T planes form: the image is read later by the dW_hh product - all-zero groups store zeros too
};

// Both directions of a bidirectional layer share ONE launch per step (blockIdx.z picks the direction), as the GRU pairs of
// csrc/gru.hip: twice the work per launch over the same fixed per-launch phases.
struct LstmBwdPair {
    LstmBwdArgs d[2];
};

template <class TC, bool VEC>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmBwdPair pr) {
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const LstmBwdArgs& g = pr.d[bz];
    const int H = g.H, B = g.B;
    const int m0 = by * TC::BM, j0 = bx * TC::BN;
    const size_t BH = (size_t)B * H;
    float pre[TC::NI][TC::MI][4], sv[TC::NI][TC::MI][4][7];
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
                pre[ni][mi][r] = (g.ext ? g.ext[o] : 0.f) + (g.ext2 ? g.ext2[o] : 0.f);
                sv[ni][mi][r][6] = g.dC_next ? g.dC_next[o] : 0.f;
                if (g.gates) {
                    sv[ni][mi][r][0] = g.gates[o];
                    sv[ni][mi][r][1] = g.gates[BH + o];
                    sv[ni][mi][r][2] = g.gates[2 * BH + o];
                    sv[ni][mi][r][3] = g.gates[3 * BH + o];
                    sv[ni][mi][r][4] = g.c_prev[o];
                    sv[ni][mi][r][5] = g.c_cur[o];
                }
            }
    }
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.dG_next) {
        OpA a{g.dG_next, 4 * H, m0, B, nullptr, 1.f};
        OpB b{g.w_hh, H, j0, H, 0, nullptr, 1.f};
        MainLoop<TC, true, false, VEC, VEC>::run(a, b, 4 * H, acc);
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
                const float dcn = sv[ni][mi][r][6];
                if (!g.gates) {
                    g.dH_out[o] = dh;
                    g.dC_out[o] = dcn;
                    continue;
                }
                const float ig = sv[ni][mi][r][0], fg = sv[ni][mi][r][1], gg = sv[ni][mi][r][2], og = sv[ni][mi][r][3];
                const float cpv = sv[ni][mi][r][4], tc = tanhf(sv[ni][mi][r][5]);
                const float dc = dcn + dh * og * (1.f - tc * tc);
                g.dC_out[o] = dc * fg;
                float* d = g.dG_out + (size_t)row * 4 * H;
                d[j] = dc * gg * ig * (1.f - ig);
                d[H + j] = dc * cpv * fg * (1.f - fg);
                d[2 * H + j] = dc * ig * (1.f - gg * gg);
                d[3 * H + j] = dh * tc * og * (1.f - og);
            }
    }
}

// Backward step with the direct-to-LDS main loop (DlLoop, gemm_core.h): dG_next [B,4H] . W_hh^T rows, both K-contiguous, exact-f32
// MFMA, 16-byte row-layout epilogue.  Same sums as lstm_step_bwd_kernel (same contraction order); dense full tiles only.
// PREC 1: bf16 compute mode - operands rounded to bf16 at the fragment read, one bf16 MFMA per block and slab (as the GRU kernel)
// PREC 3: f32-grade on f16 pairs handed from launch to launch (pair_engine.h; as gru_step_bwd_dl_kernel) - three f16 MFMAs per block
// WR x WC: wave grid (gemm_core.h) - 2 x 2 waves of BM/2 x BN/2, or (round 6, PREC 3) 2 x 4 waves of 32 x 16 on the 64 x 64 tile
template <int BM, int BN, int PREC = 0, int WR = 2, int WC = 2>
__global__ __launch_bounds__(64 * WR * WC, WC == 4 ? 2 : 1) void lstm_step_bwd_dl_kernel(LstmBwdPair pr) {
    using DL = DlLoop<BM, BN, 3, PREC, WR, WC>;
    constexpr int MI = DL::MI, NI = DL::NI;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const LstmBwdArgs& g = pr.d[bz];
    const int H = g.H;
    const int m0 = by * BM, j0 = bx * BN;
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
    f32x4 pre[MI][NI], sv[MI][NI][7];
    auto load_ep = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const size_t o = (size_t)(rb0 + 16 * mi) * H + cb0 + 16 * ni;
                const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                pre[mi][ni] = g.ext ? *reinterpret_cast<const f32x4*>(g.ext + o) : zero;
                if (g.ext2) pre[mi][ni] += *reinterpret_cast<const f32x4*>(g.ext2 + o);
                sv[mi][ni][6] = g.dC_next ? *reinterpret_cast<const f32x4*>(g.dC_next + o) : zero;
                if (g.gates) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[mi][ni][q] = *reinterpret_cast<const f32x4*>(g.gates + q * BH + o);
                    sv[mi][ni][4] = *reinterpret_cast<const f32x4*>(g.c_prev + o);
                    sv[mi][ni][5] = *reinterpret_cast<const f32x4*>(g.c_cur + o);
                }
            }
    };
    PairConsumer<MI, NI, 4> pc;
    const int w_exp = (PREC == 3 && g.pp_next) ? weight_exp_from_parts(g.ex_min + H / 32) : 0;   // power of two of the W_hh^T image (pair_engine.h)
    if (PREC == 3 ? g.pp_next != nullptr : g.dG_next != nullptr) {   // (the all-T planes form has no f32 dG)
        const int hb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int phase = ((hb >> 3) + (hb >> 8)) & 3, KT = 4 * H / 32;
        if constexpr (PREC == 3) {
            pc.init(g.ex_next + (size_t)((m0 + wm * 32) / 32) * (H / 32), H / 32, lane);
            DL::run(g.pp_next + (size_t)m0 * 8 * H, (size_t)8 * H, reinterpret_cast<const uint16_t*>(g.w_hhT) + (size_t)j0 * 8 * H,
                    (size_t)8 * H, 8 * H, cpg_smem, acc, min(phase * ((KT / 4) & ~1), KT - 1), load_ep, [&](int kt) { return pc.pre(kt, acc); });
            pc.finish(acc, w_exp);
        } else {
            DL::run(g.dG_next + (size_t)m0 * 4 * H, (size_t)4 * H, g.w_hhT + (size_t)j0 * 4 * H, (size_t)4 * H, 4 * H, cpg_smem, acc,
                    min(phase * ((KT / 4) & ~1), KT - 1), load_ep);
        }
    } else {
        load_ep();
    }
    f32x4 pv[PREC == 3 ? MI : 1][PREC == 3 ? NI : 1][4];   // PREC 3: the four blocks of dG, kept for the pair planes
    float vmax = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const f32x4 dh = acc_block_to_rows(tb, acc[mi][ni], lane) + pre[mi][ni];
            const size_t o = (size_t)(rb0 + 16 * mi) * H + cb0 + 16 * ni;
            const f32x4 dcn = sv[mi][ni][6];
            if (!g.gates) {
                *reinterpret_cast<f32x4*>(g.dH_out + o) = dh;
                *reinterpret_cast<f32x4*>(g.dC_out + o) = dcn;
                continue;
            }
            const f32x4 ig = sv[mi][ni][0], fg = sv[mi][ni][1], gg = sv[mi][ni][2], og = sv[mi][ni][3], cpv = sv[mi][ni][4];
            f32x4 tc;
#pragma unroll
            for (int e = 0; e < 4; ++e) tc[e] = tanhf(sv[mi][ni][5][e]);
            const f32x4 dc = dcn + dh * og * (1.f - tc * tc);
            *reinterpret_cast<f32x4*>(g.dC_out + o) = dc * fg;
            const f32x4 d0 = dc * gg * ig * (1.f - ig), d1 = dc * cpv * fg * (1.f - fg), d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
            if (g.dG_out) {   // (null in the all-T planes form: the kept images are the only copy)
                float* d = g.dG_out + (size_t)(rb0 + 16 * mi) * 4 * H + cb0 + 16 * ni;
                *reinterpret_cast<f32x4*>(d) = d0;
                *reinterpret_cast<f32x4*>(d + H) = d1;
                *reinterpret_cast<f32x4*>(d + 2 * H) = d2;
                *reinterpret_cast<f32x4*>(d + 3 * H) = d3;
            }
            if constexpr (PREC == 3) {
                pv[mi][ni][0] = d0; pv[mi][ni][1] = d1; pv[mi][ni][2] = d2; pv[mi][ni][3] = d3;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(pv[mi][ni][q][j]));
            }
        }
    if constexpr (PREC == 3) {
        if (!g.gates || !g.pp_out) return;   // (block-uniform)
        const int grp = (j0 + wn * (BN / WC)) / 32;
        const int e = pair_group_exponent<BN / WC>(vmax, cpg_smem, wave, lane, g.ex_out + (size_t)((m0 + wm * 32) / 32) * (H / 32) + grp,
                                              g.ex_min + grp);
        if (e != INT_MAX || g.ap) {   // (the chain's own consumer looks at the table first; the all-T form's dW_hh product does not)
            const float sc = e != INT_MAX ? pair_pow2(e) : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) pair_store4<4>(g.pp_out, (size_t)(rb0 + 16 * mi), H, cb0 + 16 * ni, q, pv[mi][ni][q] * sc);
        }
    }
}

using LF64 = TileCfg<64, 128, 32, 2, 2, 4>;
using LF32 = TileCfg<32, 128, 32, 2, 2, 4>;
using LF64S = TileCfg<64, 64, 32, 4, 1, 4>;   // 64 rows x (4 gates x 16 units): 61 KB of bf16 planes, two workgroups per CU
using LB64 = TileCfg<64, 32, 32, 4, 1, 1>;
using LB32 = TileCfg<32, 64, 32, 2, 2, 1>;
using LB32N = TileCfg<32, 32, 32, 2, 2, 1>;

static bool lstm_dl_ok(int B, int H);
// f16-pair form of the direct-to-LDS backward step: f32-grade mode, option gru_bwd_engine = exact switches it off (as csrc/gru.hip)
static bool lstm_dl_w8(int B, int H, int nd);
static bool lstm_pair_enabled(int H) {
    if (cpg_compute_mode_get() == 1 || H > 2048) return false;
    const CpgOptVal o = cpg_opt(OPT_GRU_BWD_ENGINE);
    return !(o.set && strcmp(o.s, "exact") == 0);
}
// tile choices of the step launches (shared by the launchers and the introspection entry point below)
static int lstm_fwd_choice(int B, int H) { return B >= 64 ? 0 : ((B > 32 && (long)cdiv(B, 32) * cdiv(H, 32) < 1024) ? 1 : 2); }  // LF64S | LF64 | LF32
static int lstm_bwd_choice(int B, int H) { return (long)cdiv(B, 32) * cdiv(H, 32) >= 1024 ? 0 : (B > 32 ? 1 : 2); }               // LB32N | LB64 | LB32

template <class TC>
static int lstm_tc_name(char* b, int n) {
    return snprintf(b, n, "TileCfg<%d, %d, %d, %d, %d, %d, %d>", TC::BM, TC::BN, TC::BK, TC::WM, TC::WN, TC::NSEG, TC::NT);
}
// Launcher introspection for bench.py (as cpg_gru_step_kernel_name): the kernel an LSTM step launch runs, named as rocprofv3
// prints it.  kind 0 forward, 1 backward.  cpg_lstm_step_kernel_is_split: 1 = six bf16 MFMAs on split operands, 0 = exact f32.
CPG_EXPORT int cpg_lstm_step_kernel_name(int kind, int B, int H, char* buf, int n) {
    char tc[96];
    const char* vec = H % 4 == 0 ? "true" : "false";
    if (kind == 0) {
        const int c = lstm_fwd_choice(B, H);
        if (c == 0) lstm_tc_name<LF64S>(tc, sizeof tc);
        else if (c == 1) lstm_tc_name<LF64>(tc, sizeof tc);
        else lstm_tc_name<LF32>(tc, sizeof tc);
        return snprintf(buf, n, "lstm_step_fwd_kernel<%s, %s>", tc, vec);
    }
    if (kind == 1) {
        if (lstm_dl_ok(B, H) && H % 4 == 0)
        {   // (B: rows of the launch - both directions' rows for a paired one)
            const int prec = cpg_compute_mode_get() == 1 ? 1 : lstm_pair_enabled(H) ? 3 : 0;
            const bool w8 = prec == 3 && lstm_dl_w8(B, H, 1);
            return snprintf(buf, n, "lstm_step_bwd_dl_kernel<64, %d, %d, 2, %d>", (w8 || (H % 64 == 0 && (long)(B / 64) * (H / 64) >= 512)) ? 64 : 32,
                            prec, w8 ? 4 : 2);
        }
        const int c = lstm_bwd_choice(B, H);
        if (c == 0) lstm_tc_name<LB32N>(tc, sizeof tc);
        else if (c == 1) lstm_tc_name<LB64>(tc, sizeof tc);
        else lstm_tc_name<LB32>(tc, sizeof tc);
        return snprintf(buf, n, "lstm_step_bwd_kernel<%s, %s>", tc, vec);
    }
    return 0;
}
CPG_EXPORT int cpg_lstm_step_kernel_is_split(int kind, int B, int H) {
    if (kind == 0) return lstm_fwd_choice(B, H) != 2;   // 64-row tiles run the plane engine (LstmFwdLoop)
    // backward: exact-f32 MFMA; on the direct-to-LDS loop in the bf16 compute mode one bf16 MFMA per block (2, as the GRU reports it)
    if (!(lstm_dl_ok(B, H) && H % 4 == 0)) return 0;
    return cpg_compute_mode_get() == 1 ? 2 : lstm_pair_enabled(H) ? 3 : 0;
}

static int lstm_fwd_launch(const LstmFwdArgs& a, hipStream_t s) {
    const bool vec = a.H % 4 == 0 && aligned16(a.h_prev) && aligned16(a.w_hh);
    const int choice = lstm_fwd_choice(a.B, a.H);
    if (choice == 0) {  // measured at B=2048, H=512: 13.76 -> 13.56 ms per training step against 32x128 exact-f32 tiles
        dim3 grid(cdiv(a.H, LF64S::BN / 4), cdiv(a.B, LF64S::BM));
        const size_t smem = LstmFwdLoop<LF64S, true>::smem_bytes();
        if (vec) hipLaunchKernelGGL((lstm_step_fwd_kernel<LF64S, true>), grid, dim3(256), smem, s, a);
        else hipLaunchKernelGGL((lstm_step_fwd_kernel<LF64S, false>), grid, dim3(256), smem, s, a);
    } else if (choice == 1) {
        // 32-row tiles once they give >= 1024 workgroups
        dim3 grid(cdiv(a.H, LF64::BN / 4), cdiv(a.B, LF64::BM));
        const size_t smem = LstmFwdLoop<LF64, true>::smem_bytes();
        const int rc = cpg_allow_big_lds(vec ? reinterpret_cast<const void*>(lstm_step_fwd_kernel<LF64, true>)
                                             : reinterpret_cast<const void*>(lstm_step_fwd_kernel<LF64, false>), (int)smem);
        if (rc) return rc;
        if (vec) hipLaunchKernelGGL((lstm_step_fwd_kernel<LF64, true>), grid, dim3(256), smem, s, a);
        else hipLaunchKernelGGL((lstm_step_fwd_kernel<LF64, false>), grid, dim3(256), smem, s, a);
    } else {
        dim3 grid(cdiv(a.H, LF32::BN / 4), cdiv(a.B, LF32::BM));
        const size_t smem = LstmFwdLoop<LF32, true>::smem_bytes();
        if (vec) hipLaunchKernelGGL((lstm_step_fwd_kernel<LF32, true>), grid, dim3(256), smem, s, a);
        else hipLaunchKernelGGL((lstm_step_fwd_kernel<LF32, false>), grid, dim3(256), smem, s, a);
    }
    CPG_LAUNCH_CHECK();
    return 0;
}

// out[C,R] = w[R,C]^T
__global__ void lstm_transpose_w_kernel(const float* w, int R, int C, float* out) {
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

// direct-to-LDS backward step: dense full tiles, W_hh^T handed over.  Option lstm_bwd_dl = 0 keeps the register-staged kernel.
static bool lstm_dl_ok(int B, int H) {
    const CpgOptVal& o = cpg_opt(OPT_LSTM_BWD_DL);
    if (o.set && o.i == 0) return false;
    return B % 64 == 0 && H % 32 == 0;
}
template <int BM, int BN, int PREC, int WR = 2, int WC = 2>
static int lstm_launch_dl_p(const LstmBwdPair& pr, int nd, hipStream_t s) {
    const LstmBwdArgs& a = pr.d[0];
    const size_t smem = (DlLoop<BM, BN, 3, PREC, WR, WC>::smem_floats() + WR * WC * 256) * sizeof(float);
    if (smem > 64 * 1024) {
        const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(lstm_step_bwd_dl_kernel<BM, BN, PREC, WR, WC>), (int)smem);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((lstm_step_bwd_dl_kernel<BM, BN, PREC, WR, WC>), dim3(a.H / BN, a.B / BM, nd), dim3(64 * WR * WC), smem, s, pr);
    return 0;
}
// f16-pair step on eight waves of 32 x 16 (as csrc/gru.hip): 64 x 64 tiles from 256 tiles up
static bool lstm_dl_w8(int B, int H, int nd) { return H % 64 == 0 && (long)(B / 64) * (H / 64) * nd >= 256; }
// bf16 compute mode (cpg_set_compute_mode(1)): the step product with bf16-rounded operands, like every other recurrent product of the mode
template <int BM, int BN>
static int lstm_launch_dl(const LstmBwdPair& pr, int nd, hipStream_t s) {
    if (pr.d[0].pp_next || pr.d[0].pp_out) return lstm_launch_dl_p<BM, BN, 3>(pr, nd, s);   // (the eight-wave form is picked by the caller)
    return cpg_compute_mode_get() == 1 ? lstm_launch_dl_p<BM, BN, 1>(pr, nd, s) : lstm_launch_dl_p<BM, BN, 0>(pr, nd, s);
}

// nd = 1: one direction (pr.d[0]); nd = 2: both directions of a bidirectional layer in one launch (same B, H; the two
// argument sets must agree on which optional operands are present, which the sequence launchers guarantee)
static int lstm_bwd_launch(const LstmBwdPair& pr, int nd, hipStream_t s) {
    const LstmBwdArgs& a = pr.d[0];
    if (a.w_hhT && lstm_dl_ok(a.B, a.H)) {
        bool al = true;
        for (int d = 0; d < nd; ++d) {
            const LstmBwdArgs& q = pr.d[d];
            const void* ptrs[] = {q.dG_next, q.w_hhT, q.dC_next, q.ext, q.ext2, q.gates, q.c_prev, q.c_cur, q.dH_out, q.dC_out, q.dG_out};
            for (const void* v : ptrs) al = al && (!v || aligned16(v));
        }
        if (al) {
            // 64 x 64 tiles once they give two workgroups per CU, else 64 x 32 (as the GRU kernel: csrc/gru.hip)
            const int rc = ((a.pp_next || a.pp_out) && lstm_dl_w8(a.B, a.H, nd)) ? lstm_launch_dl_p<64, 64, 3, 2, 4>(pr, nd, s)
                           : (a.H % 64 == 0 && (long)(a.B / 64) * (a.H / 64) * nd >= 512) ? lstm_launch_dl<64, 64>(pr, nd, s)
                                                                                          : lstm_launch_dl<64, 32>(pr, nd, s);
            if (rc) return rc;
            CPG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (a.pp_next || a.pp_out) {
        cpg_set_error("lstm backward: the f16-pair step was prepared for this sequence but a launch of it is not covered (unaligned operand?)");
        return -4;
    }
    bool vec = a.H % 4 == 0;
    for (int d = 0; d < nd; ++d) vec = vec && aligned16(pr.d[d].w_hh) && (!pr.d[d].dG_next || aligned16(pr.d[d].dG_next));
    const int choice = lstm_bwd_choice(a.B * nd, a.H);
    if (choice == 0) {
        dim3 grid(cdiv(a.H, LB32N::BN), cdiv(a.B, LB32N::BM), nd);
        const size_t smem = LB32N::smem_floats<true, false>() * sizeof(float);
        if (vec) hipLaunchKernelGGL((lstm_step_bwd_kernel<LB32N, true>), grid, dim3(256), smem, s, pr);
        else hipLaunchKernelGGL((lstm_step_bwd_kernel<LB32N, false>), grid, dim3(256), smem, s, pr);
    } else if (choice == 1) {
        dim3 grid(cdiv(a.H, LB64::BN), cdiv(a.B, LB64::BM), nd);
        const size_t smem = LB64::smem_floats<true, false>() * sizeof(float);
        if (vec) hipLaunchKernelGGL((lstm_step_bwd_kernel<LB64, true>), grid, dim3(256), smem, s, pr);
        else hipLaunchKernelGGL((lstm_step_bwd_kernel<LB64, false>), grid, dim3(256), smem, s, pr);
    } else {
        dim3 grid(cdiv(a.H, LB32::BN), cdiv(a.B, LB32::BM), nd);
        const size_t smem = LB32::smem_floats<true, false>() * sizeof(float);
        if (vec) hipLaunchKernelGGL((lstm_step_bwd_kernel<LB32, true>), grid, dim3(256), smem, s, pr);
        else hipLaunchKernelGGL((lstm_step_bwd_kernel<LB32, false>), grid, dim3(256), smem, s, pr);
    }
    CPG_LAUNCH_CHECK();
    return 0;
}

// hs, cs: state slabs [(T+1),B,H] with h0 / c0 in slot 0 (forward) or T (reverse); gates [T,4,B,H] or null.
CPG_EXPORT int cpg_lstm_seq_fwd(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh, const int32_t* tok,
                                const float* tab, const float* rowc, const float* dense, float* hs, float* cs, float* gates,
                                void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && b_hh && hs && cs);
    CPG_CHECK_ARG((tok == nullptr) == (tab == nullptr));
    const size_t BH = (size_t)B * H;
    for (int p = 0; p < T; ++p) {
        const int t = reverse ? T - 1 - p : p;
        const size_t ip = reverse ? (size_t)(t + 1) * BH : (size_t)t * BH;
        const size_t io = reverse ? (size_t)t * BH : (size_t)(t + 1) * BH;
        LstmFwdArgs a{hs + ip, cs + ip, w_hh, b_hh, tok ? tok + (size_t)t * B : nullptr, tab, rowc,
                      dense ? dense + (size_t)t * B * 4 * H : nullptr, hs + io, cs + io,
                      gates ? gates + (size_t)t * 4 * BH : nullptr, B, H};
        int rc = lstm_fwd_launch(a, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

CPG_EXPORT int cpg_lstm_step_fwd(int B, int H, const float* w_hh, const float* b_hh, const int32_t* tok, const float* tab,
                                 const float* rowc, const float* h_prev, const float* c_prev, float* h_out, float* c_out,
                                 void* stream) {
    CPG_CHECK_ARG(B > 0 && H > 0 && w_hh && b_hh && h_prev && c_prev && h_out && c_out && h_prev != h_out && c_prev != c_out);
    LstmFwdArgs a{h_prev, c_prev, w_hh, b_hh, tok, tab, rowc, nullptr, h_out, c_out, nullptr, B, H};
    return lstm_fwd_launch(a, (hipStream_t)stream);
}

// dhs_ext [T,B,H] (time-aligned, may be null); dG out [T,B,4H]; scratch [2,B,H] (carried cell gradient);
// dh0, dc0 [B,H] (both or neither).
// Scratch of the f16-pair backward step per direction (pair_engine.h); 0: not covered (pass null)
CPG_EXPORT size_t cpg_lstm_bwd_pair_bytes(int B, int H) {
    if (B <= 0 || H <= 0 || !lstm_dl_ok(B, H) || H % 4 != 0 || !lstm_pair_enabled(H)) return 0;
    return pair_scratch_bytes(B, H, 4);
}

// ---- all-T planes form (round 5, as csrc/gru.hip's): the f16-pair images of dG the backward steps hand to each other are KEPT for all T
// ([T][B][8H] f16 + exponents, pair_engine.h ApScratch with G = 4) and are the A operand of the dW_hh product (pair_tn.h: LDS-DMA,
// transposing reads, no conversion in its loop); the B operand is the unscaled image of h_prev that cpg_lstm_wgrad_hh_ap makes in one
// pass over the state slab.  With an LSTM the input-side gradient IS dG: the token-table / row-constant reductions read the images too
// (cpg_lstm_dgi_reduce_ap) and dG may be null - the images are then the only copy (16 B per state element either way; what goes away
// is the in-loop conversion of the dW_hh product and the cache-resident ping-pong images).  Layers with a dense input term pass dG as
// well (their nn.Linear-shaped products read f32).  Coverage: f32-grade mode, the f16-pair backward step, B and H multiples of
// 128; option gru_ap = 0 switches the form off for both cells.
static bool lstm_ap_ok(int B, int H) {
    const CpgOptVal o = cpg_opt(OPT_GRU_AP);
    if (o.set && o.i == 0) return false;
    if (B <= 0 || H <= 0 || H % 128 != 0 || B % 128 != 0 || cpg_compute_mode_get() == 1) return false;
    return cpg_lstm_bwd_pair_bytes(B, H) > 0;
}
CPG_EXPORT size_t cpg_lstm_ap_bytes(int T, int B, int H) { return (T > 0 && lstm_ap_ok(B, H)) ? ap_scratch_bytes(T, B, H, 4) : 0; }

static int lstm_seq_bwd_impl(int T, int B, int H, int reverse, const float* w_hh, const float* cs, const float* gates,
                             const float* dhs_ext, float* dG, float* scratch, float* dh0, float* dc0, float* w_hhT_scratch,
                             void* pair_scratch, void* ap_scratch, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && cs && gates && (dG || ap_scratch) && scratch && ((dh0 == nullptr) == (dc0 == nullptr)));
    const size_t BH = (size_t)B * H;
    if (w_hhT_scratch && !lstm_dl_ok(B, H)) w_hhT_scratch = nullptr;
    const bool allt = ap_scratch != nullptr;
    const bool pair = (allt || pair_scratch) && w_hhT_scratch && cpg_lstm_bwd_pair_bytes(B, H) > 0;
    uint16_t* PP[2] = {nullptr, nullptr};
    int* EX[2] = {nullptr, nullptr};
    int* EMIN = nullptr;
    ApScratch A{nullptr, nullptr, nullptr, nullptr};
    if (allt) {
        if (!pair) return -4;
        A = ap_split(ap_scratch, T, B, H, 4);
        EMIN = A.ex_min;
    } else if (pair) {
        pair_split(pair_scratch, B, H, 4, PP, EX, EMIN);
    }
    const size_t plane = (size_t)B * 8 * H, extab = (size_t)(B / 32) * (H / 32);
    if (pair) {
        int rc = cpg_pair_w(w_hh, 4, H, reinterpret_cast<uint16_t*>(w_hhT_scratch), EMIN, (hipStream_t)stream);
        if (rc) return rc;
    } else if (w_hhT_scratch) {   // W_hh^T [H,4H] once per sequence: the direct-to-LDS kernel wants both operands K-contiguous
        hipLaunchKernelGGL(lstm_transpose_w_kernel, dim3(cdiv(H, 32), cdiv(4 * H, 32)), dim3(32, 8), 0, (hipStream_t)stream, w_hh,
                           4 * H, H, w_hhT_scratch);
        CPG_LAUNCH_CHECK();
    }
    int prev_t = -1;
    for (int p = T - 1; p >= -1; --p) {
        if (p < 0 && !dh0) break;
        const int t = p < 0 ? -1 : (reverse ? T - 1 - p : p);
        const int cur = (p + 2) & 1;
        LstmBwdPair pr;
        LstmBwdArgs& a = pr.d[0];
        a.B = B;
        a.H = H;
        a.w_hh = w_hh;
        a.w_hhT = w_hhT_scratch;
        a.ext2 = nullptr;
        a.dG_next = (prev_t >= 0 && dG) ? dG + (size_t)prev_t * B * 4 * H : nullptr;
        a.dC_next = prev_t >= 0 ? scratch + (size_t)(cur ^ 1) * BH : nullptr;
        a.pp_next = (pair && prev_t >= 0) ? (allt ? A.planes + (size_t)prev_t * plane : PP[cur ^ 1]) : nullptr;
        a.ex_next = (pair && prev_t >= 0) ? (allt ? A.ex + (size_t)prev_t * extab : EX[cur ^ 1]) : nullptr;
        a.pp_out = (pair && p >= 0) ? (allt ? A.planes + (size_t)t * plane : PP[cur]) : nullptr;
        a.ex_out = (pair && p >= 0) ? (allt ? A.ex + (size_t)t * extab : EX[cur]) : nullptr;
        a.ex_min = EMIN;
        a.ap = allt ? 1 : 0;
        if (p >= 0) {
            a.ext = dhs_ext ? dhs_ext + (size_t)t * BH : nullptr;
            a.gates = gates + (size_t)t * 4 * BH;
            a.c_prev = reverse ? cs + (size_t)(t + 1) * BH : cs + (size_t)t * BH;
            a.c_cur = reverse ? cs + (size_t)t * BH : cs + (size_t)(t + 1) * BH;
            a.dH_out = nullptr;
            a.dC_out = scratch + (size_t)cur * BH;
            a.dG_out = dG ? dG + (size_t)t * B * 4 * H : nullptr;
        } else {
            a.ext = nullptr;
            a.gates = nullptr;
            a.c_prev = a.c_cur = nullptr;
            a.dH_out = dh0;
            a.dC_out = dc0;
            a.dG_out = nullptr;
        }
        pr.d[1] = a;
        int rc = lstm_bwd_launch(pr, 1, (hipStream_t)stream);
        if (rc) return rc;
        prev_t = t;
    }
    return 0;
}

CPG_EXPORT int cpg_lstm_seq_bwd(int T, int B, int H, int reverse, const float* w_hh, const float* cs, const float* gates,
                                const float* dhs_ext, float* dG, float* scratch, float* dh0, float* dc0, float* w_hhT_scratch,
                                void* pair_scratch, void* stream) {
    return lstm_seq_bwd_impl(T, B, H, reverse, w_hh, cs, gates, dhs_ext, dG, scratch, dh0, dc0, w_hhT_scratch, pair_scratch, nullptr, stream);
}
CPG_EXPORT int cpg_lstm_seq_bwd_ap(int T, int B, int H, int reverse, const float* w_hh, const float* cs, const float* gates,
                                   const float* dhs_ext, float* dG, float* scratch, float* dh0, float* dc0, float* w_hhT_scratch,
                                   void* ap, void* stream) {
    CPG_CHECK_ARG(ap && w_hhT_scratch && aligned16(ap));
    if (cpg_lstm_ap_bytes(T, B, H) == 0) {
        cpg_set_error("cpg_lstm_seq_bwd_ap: shape / mode not covered (cpg_lstm_ap_bytes answers 0: f32-grade mode, H %% 128 == 0, B %% 128 == 0)");
        return -4;
    }
    return lstm_seq_bwd_impl(T, B, H, reverse, w_hh, cs, gates, dhs_ext, dG, scratch, dh0, dc0, w_hhT_scratch, nullptr, ap, stream);
}

// Both directions of one biLSTM layer, ONE launch per step for the pair (launch p: time p of the forward direction, time T-1-p
// of the reverse one).  Arguments as cpg_lstm_seq_bwd per direction (_f forward, _r reverse); dh_last_* [B,H] (optional): the
// gradient on a direction's final HIDDEN state enters at its last step; no initial-state gradients (the encoder starts from
// h0 = c0 = 0).  (Extension: the reference has no LSTM - SURVEY F2; the pairing mirrors cpg_gru_biseq_bwd.)
static int lstm_biseq_bwd_impl(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* cs_f,
                               const float* cs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                               const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f, float* dG_r,
                               float* scratch_f, float* scratch_r, float* w_hhT_scratch_f, float* w_hhT_scratch_r,
                               void* pair_scratch_f, void* pair_scratch_r, void* ap_f, void* ap_r, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh_f && w_hh_r && cs_f && cs_r && gates_f && gates_r && ((dG_f && dG_r) || (ap_f && ap_r && !dG_f && !dG_r)));
    CPG_CHECK_ARG(scratch_f && scratch_r && (w_hhT_scratch_f == nullptr) == (w_hhT_scratch_r == nullptr));
    CPG_CHECK_ARG((dhs_ext_f == nullptr) == (dhs_ext_r == nullptr) && (dh_last_f == nullptr) == (dh_last_r == nullptr));
    if (w_hhT_scratch_f && !lstm_dl_ok(B, H)) w_hhT_scratch_f = w_hhT_scratch_r = nullptr;
    const float* W[2] = {w_hh_f, w_hh_r};
    float* WT[2] = {w_hhT_scratch_f, w_hhT_scratch_r};
    const bool allt = ap_f && ap_r;
    const bool pair = ((pair_scratch_f && pair_scratch_r) || allt) && w_hhT_scratch_f && cpg_lstm_bwd_pair_bytes(B, H) > 0;
    uint16_t* PP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int* EXP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int* EMIN[2] = {nullptr, nullptr};
    ApScratch A[2] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    if (allt) {
        if (!pair) return -4;
        A[0] = ap_split(ap_f, T, B, H, 4);
        A[1] = ap_split(ap_r, T, B, H, 4);
        EMIN[0] = A[0].ex_min;
        EMIN[1] = A[1].ex_min;
    } else if (pair) {
        pair_split(pair_scratch_f, B, H, 4, PP[0], EXP[0], EMIN[0]);
        pair_split(pair_scratch_r, B, H, 4, PP[1], EXP[1], EMIN[1]);
    }
    const size_t plane = (size_t)B * 8 * H, extab = (size_t)(B / 32) * (H / 32);
    for (int d = 0; d < 2 && WT[d]; ++d) {
        if (pair) {
            int rc = cpg_pair_w(W[d], 4, H, reinterpret_cast<uint16_t*>(WT[d]), EMIN[d], (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        hipLaunchKernelGGL(lstm_transpose_w_kernel, dim3(cdiv(H, 32), cdiv(4 * H, 32)), dim3(32, 8), 0, (hipStream_t)stream, W[d],
                           4 * H, H, WT[d]);
        CPG_LAUNCH_CHECK();
    }
    const size_t BH = (size_t)B * H;
    const float* CS[2] = {cs_f, cs_r};
    const float* GT[2] = {gates_f, gates_r};
    const float* EX[2] = {dhs_ext_f, dhs_ext_r};
    const float* LAST[2] = {dh_last_f, dh_last_r};
    float* DG[2] = {dG_f, dG_r};
    float* SC[2] = {scratch_f, scratch_r};
    int prev_t[2] = {-1, -1};
    for (int p = T - 1; p >= 0; --p) {
        LstmBwdPair pr;
        const int cur = (p + 2) & 1;
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - p : p;
            LstmBwdArgs& a = pr.d[d];
            a.B = B;
            a.H = H;
            a.w_hh = W[d];
            a.w_hhT = WT[d];
            a.dG_next = (prev_t[d] >= 0 && DG[d]) ? DG[d] + (size_t)prev_t[d] * B * 4 * H : nullptr;
            a.dC_next = prev_t[d] >= 0 ? SC[d] + (size_t)(cur ^ 1) * BH : nullptr;
            a.pp_next = (pair && prev_t[d] >= 0) ? (allt ? A[d].planes + (size_t)prev_t[d] * plane : PP[d][cur ^ 1]) : nullptr;
            a.ex_next = (pair && prev_t[d] >= 0) ? (allt ? A[d].ex + (size_t)prev_t[d] * extab : EXP[d][cur ^ 1]) : nullptr;
            a.pp_out = pair ? (allt ? A[d].planes + (size_t)t * plane : PP[d][cur]) : nullptr;
            a.ex_out = pair ? (allt ? A[d].ex + (size_t)t * extab : EXP[d][cur]) : nullptr;
            a.ex_min = EMIN[d];
            a.ap = allt ? 1 : 0;
            a.ext = EX[d] ? EX[d] + (size_t)t * BH : nullptr;
            a.ext2 = (p == T - 1) ? LAST[d] : nullptr;
            a.gates = GT[d] + (size_t)t * 4 * BH;
            a.c_prev = d ? CS[d] + (size_t)(t + 1) * BH : CS[d] + (size_t)t * BH;
            a.c_cur = d ? CS[d] + (size_t)t * BH : CS[d] + (size_t)(t + 1) * BH;
            a.dH_out = nullptr;
            a.dC_out = SC[d] + (size_t)cur * BH;
            a.dG_out = DG[d] ? DG[d] + (size_t)t * B * 4 * H : nullptr;
            prev_t[d] = t;
        }
        int rc = lstm_bwd_launch(pr, 2, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

CPG_EXPORT int cpg_lstm_biseq_bwd(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* cs_f,
                                  const float* cs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                  const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f, float* dG_r,
                                  float* scratch_f, float* scratch_r, float* w_hhT_scratch_f, float* w_hhT_scratch_r,
                                  void* pair_scratch_f, void* pair_scratch_r, void* stream) {
    return lstm_biseq_bwd_impl(T, B, H, w_hh_f, w_hh_r, cs_f, cs_r, gates_f, gates_r, dhs_ext_f, dhs_ext_r, dh_last_f, dh_last_r, dG_f, dG_r,
                               scratch_f, scratch_r, w_hhT_scratch_f, w_hhT_scratch_r, pair_scratch_f, pair_scratch_r, nullptr, nullptr, stream);
}
CPG_EXPORT int cpg_lstm_biseq_bwd_ap(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* cs_f,
                                     const float* cs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                     const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f, float* dG_r,
                                     float* scratch_f, float* scratch_r, float* w_hhT_scratch_f, float* w_hhT_scratch_r,
                                     void* ap_f, void* ap_r, void* stream) {
    CPG_CHECK_ARG(ap_f && ap_r && w_hhT_scratch_f && w_hhT_scratch_r && aligned16(ap_f) && aligned16(ap_r));
    if (cpg_lstm_ap_bytes(T, B, H) == 0) {
        cpg_set_error("cpg_lstm_biseq_bwd_ap: shape / mode not covered (cpg_lstm_ap_bytes answers 0)");
        return -4;
    }
    return lstm_biseq_bwd_impl(T, B, H, w_hh_f, w_hh_r, cs_f, cs_r, gates_f, gates_r, dhs_ext_f, dhs_ext_r, dh_last_f, dh_last_r, dG_f, dG_r,
                               scratch_f, scratch_r, w_hhT_scratch_f, w_hhT_scratch_r, nullptr, nullptr, ap_f, ap_r, stream);
}

// dw_hh[4H,H] (+)= sum_t dG_t^T h_prev(t) ; db_hh[4H] (+)= sum dG.   workspace: cpg_gru_wgrad_workspace(T,B,H,V)
// pair_scratch (optional): the scratch of the sequence's backward call - the product then runs on f16 pairs (as cpg_gru_wgrad_hh)
CPG_EXPORT int cpg_lstm_wgrad_hh(int T, int B, int H, int reverse, const float* dG, const float* hs, float* dw_hh,
                                 float* db_hh, int accumulate, void* workspace, size_t workspace_bytes, const void* pair_scratch,
                                 void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && dG && hs && dw_hh && workspace);
    const float* hprev = reverse ? hs + (size_t)B * H : hs;
    const int* exps = nullptr;
    if (pair_scratch && cpg_lstm_bwd_pair_bytes(B, H) > 0) {
        uint16_t* pp[2];
        int* ex[2];
        int* emin = nullptr;
        pair_split(const_cast<void*>(pair_scratch), B, H, 4, pp, ex, emin);
        exps = emin;
    }
    int rc = cpg_gemm_tn(dG, 4 * H, hprev, H, nullptr, 1.f, dw_hh, H, T * B, 4 * H, H, accumulate, (float*)workspace,
                         workspace_bytes, (hipStream_t)stream, 0, exps, H);
    if (rc || !db_hh) return rc;  // db_hh null: the caller takes it from cpg_lstm_dgi_reduce's column sums
    return cpg_colsum(dG, 4 * H, T * B, 4 * H, db_hh, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// All-T planes form: dw_hh[4H,H] (+)= kept gate-gradient images^T x the image of h_prev (made here: one pass over the state slab
// hs [T+1,B,H] - slots 0..T-1 forward, 1..T reverse; |h| < 1: unscaled).  The bias gradient comes out of cpg_lstm_dgi_reduce's column
// sums (token-table layers) - this form covers those layers only.
CPG_EXPORT int cpg_lstm_wgrad_hh_ap(int T, int B, int H, int reverse, void* ap, const float* hs, float* dw_hh, int accumulate,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && ap && hs && dw_hh && workspace && aligned16(hs));
    if (cpg_lstm_ap_bytes(T, B, H) == 0) {
        cpg_set_error("cpg_lstm_wgrad_hh_ap: shape / mode not covered (cpg_lstm_ap_bytes answers 0)");
        return -4;
    }
    const ApScratch a = ap_split(ap, T, B, H, 4);
    int rc = cpg_pair_rows(reverse ? hs + (size_t)B * H : hs, H, H, nullptr, 0, 0, T * B, a.hplanes, stream);
    if (rc) return rc;
    return cpg_pair_tn(a.planes, (size_t)8 * H, a.ex, a.ex_min, H / 32, 4, a.hplanes, (size_t)2 * H, dw_hh, H, 4 * H, H, T * B, accumulate,
                       (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

CPG_EXPORT int cpg_lstm_dgi_reduce(int T, int B, int H, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                                   float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return cpg_dgi_reduce_impl(T, B, H, 1, dG, tok, V, dtab, dsum, drowc, accumulate, workspace, workspace_bytes, stream);
}
