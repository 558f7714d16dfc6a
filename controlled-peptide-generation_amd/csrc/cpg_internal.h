// Internal (non-exported) entry points shared between the .hip translation units of libcpg_hip.so.
#pragma once
#include "cpg_common.h"

// Y[M,N] (+)= (X .* xmask*xms)[M,K] W[N,K]^T + bias
int cpg_gemm_nt(const float* X, int ldx, const uint8_t* xmask, float xms, const float* W, int ldw, const float* bias,
                float* Y, int ldy, int M, int N, int K, int accumulate, hipStream_t s);
// Y[M,N] (+)= X[M,K] B[K,N]; optional keep-mask on the stored result
int cpg_gemm_nn(const float* X, int ldx, const float* Bm, int ldb, float* Y, int ldy, int M, int N, int K, int accumulate,
                const uint8_t* cmask, float cms, hipStream_t s);
// dW[N,Kd] (+)= dY[Mr,N]^T (X .* xmask*xms)[Mr,Kd]
// dy_bf16: dY holds bf16 elements (lddy in elements): the dW_hh product on bf16 gate gradients (bf16 compute mode)
int cpg_gemm_tn(const float* dY, int lddy, const float* X, int ldx, const uint8_t* xmask, float xms, float* dW, int lddw,
                int Mr, int N, int Kd, int accumulate, float* ws, size_t ws_bytes, hipStream_t s, int dy_bf16 = 0,
                const int* dy_exps = nullptr /* f16-pair form: exponent per 32 columns of dY, see GemmArgs::a_exps */, int dy_exps_mod = 1);
size_t cpg_gemm_tn_workspace(int Mr, int N, int Kd);
// dW[M,N] (+)= A^T B over R rows, both operands as f16-pair plane images (csrc/pair_tn.h): row r of an image = columns/32 segments of
// [32 hi | 32 lo] f16; A's segments carry 2^a_ex[(r/32) * a_groups + seg / a_seg_per_group], brought to 2^a_emin[group] in the
// kernel and taken back out of the output rows; with G = a_seg_per_group > 1 the image interleaves G blocks of M / G columns
int cpg_pair_tn(const uint16_t* A, size_t lda, const int* a_ex, const int* a_emin, int a_groups, int a_seg_per_group, const uint16_t* B,
                size_t ldb, float* dW, int lddw, int M, int N, int R, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
size_t cpg_pair_tn_workspace(int M, int N, int R);
int cpg_pair_tn_bf16(const uint16_t* A, size_t lda, const uint16_t* B, size_t ldb, float* dW, int lddw, int M, int N, int R, int accumulate,
                     float* ws, size_t ws_bytes, hipStream_t s);
// x = [x1 | x2] (x2 optional) f32 -> unscaled f16-pair image [R][2 (C1 + C2)] (csrc/planes.hip; exported: include/cpg_api.h)
extern "C" int cpg_pair_rows(const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int R, void* img, void* stream);
int cpg_colsum(const float* X, int ld, int M, int N, float* out, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
size_t cpg_colsum_workspace(int M, int N);

// token-grouped / over-time reductions of the input-side gate gradients (+ dsum[4H] = column sums of dG);
// lstm=1: 4H identity-mapped columns
int cpg_dgi_reduce_impl(int T, int B, int H, int lstm, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                        float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream, int dg_bf16 = 0);

// 0 = f32-grade products, 1 = bf16 recurrent products (cpg_set_compute_mode, api.hip)
int cpg_compute_mode_get();
// saved gates of a GRU sequence stored as bf16 (gru.hip): bf16 compute mode, dense batches on shapes the direct-to-LDS backward covers
bool cpg_gru_store_bf16(int B, int H, bool dense);
// ... and the gate gradients dG too (bf16 gradient storage; V = rows of the sequence's token table, 0 = none)
bool cpg_gru_dg_store_bf16(int B, int H, bool dense, int V);

// ABI version: bumped whenever an exported signature changes (cpg/_lib.py refuses a library whose version differs)
#define CPG_ABI_VERSION 323

// ---- option table (api.hip): tuning knobs of the launch policy, read from the environment ONCE and set through
// cpg_set_option afterwards.  Unset = the built-in policy (the measured best at the bench configuration).
enum CpgOpt {
    OPT_GRU_PERSIST,      // 0: per-step launches instead of the whole-sequence persistent forward
    OPT_LSTM_PERSIST,     // same for the LSTM extension
    OPT_F32_ENGINE,       // "f16x2" (default) | "bf16x3": split of the f32-grade persistent forward kernels (gemm_core.h)
    OPT_LSTM_PERSIST_NG,  // 1 | 2 | 4: upper bound on the 8-unit groups one workgroup of the persistent LSTM forward holds (default: 4)
    OPT_GRU_FWD_BM,       // 32 | 64 | 128: row-tile height of the per-step forward kernel
    OPT_GRU_BWD_DL,       // 0: register-staged exact-f32 backward step instead of the direct-to-LDS loop
    OPT_GRU_BWD_TILE,     // "32x32" | "64x32" | "32x64" | "64x64": tile of the backward step
    OPT_GRU_BWD_DL2,      // 0 never / 1 whenever tiles are full: the 512-thread two-K-halves backward step
    OPT_GRU_BWD_STAGGER,  // slabs between the staggered epilogue-operand fetches of the register-staged backward step
    OPT_GRU_BWD_ENGINE,   // "f16x2" (default) | "exact": product of the direct-to-LDS backward step in the f32-grade mode
    OPT_LSTM_BWD_DL,      // 0: register-staged LSTM backward step
    OPT_TN_TILE,          // tile of the dW (transposed-use) products, e.g. "256x128"
    OPT_TN_SPLIT,         // split-K factor of those products
    OPT_GEMM_TILE,        // tile of the nn.Linear-shaped products
    OPT_DGI_MODE,         // input-side reductions: "mfma" (default) | "fused" | "gemm"
    OPT_MMD_DL,           // 0: register-staged Gram launch of the full-kernel MMD
    OPT_BF16_STORE,       // 0: f32 saved gates in the bf16 compute mode too (default there: bf16, see cpg_gru_gates_bf16)
    OPT_BF16_DG,          // 0: f32 gate gradients in the bf16 compute mode too (default there: bf16 where covered, see cpg_gru_dg_bf16)
    OPT_GRU_AP,           // 0: f32 gate gradients + ping-pong planes instead of the all-T planes form of the f16-pair BPTT chain (cpg_gru_ap_bytes)
    OPT_GRU_SMALL_SEQ,    // 0: per-step launches for small GRU recurrences too (default: whole-sequence launches, csrc/decode_fused.hip gru_seq_small_*)
    OPT_SMALL_SEQ_ROWS,   // 16 | 32: batch rows per workgroup of those launches (default: 16 while every tile still gets its own CU)
    OPT__COUNT
};
struct CpgOptVal {
    bool set;
    long i;       // atol of the text
    char s[24];   // the text
};
CpgOptVal cpg_opt(CpgOpt o);   // a copy, taken under the table's mutex (api.hip)
int cpg_persist_planes();      // operand planes of the persistent forward kernels now: 1 bf16 compute mode, 2 f16 pair, 3 bf16 triple
// Whole-sequence launches for SMALL GRU recurrences in training (csrc/decode_fused.hip): hidden <= 128, token-table input (+ optional
// per-row constant), dense batch, f32-grade mode.  One workgroup keeps a 32-row tile's state on its CU for all T steps with W_hh in
// registers (16-row tiles while those still get a CU each) - the reference's default sizes (h = 80 / 102, batch 32) are otherwise one 8-20 us launch per time step and direction.
struct CpgSmallFwdDir { const float* w_hh; const float* b_hh; const int32_t* tok; const float* tab; const float* rowc; float* hs; float* gates; int reverse; };
struct CpgSmallBwdDir { const float* w_hh; const float* hs; const float* gates; const float* dhs_ext; const float* dh_last; float* dG; float* dh0; int reverse; };
bool cpg_gru_small_seq_ok(int B, int H);
int cpg_gru_small_seq_fwd(int T, int B, int H, int ndir, const CpgSmallFwdDir* d, hipStream_t s);
int cpg_gru_small_seq_bwd(int T, int B, int H, int ndir, const CpgSmallBwdDir* d, hipStream_t s);
int cpg_device_cus();                                   // CUs of the current device (cached per device)
int cpg_allow_big_lds(const void* kernel, int bytes);   // opt a kernel into > 64 KB dynamic LDS, once per (kernel, device)
