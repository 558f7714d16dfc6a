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
int cpg_gemm_tn(const float* dY, int lddy, const float* X, int ldx, const uint8_t* xmask, float xms, float* dW, int lddw,
                int Mr, int N, int Kd, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
size_t cpg_gemm_tn_workspace(int Mr, int N, int Kd);
int cpg_colsum(const float* X, int ld, int M, int N, float* out, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
size_t cpg_colsum_workspace(int M, int N);

// token-grouped / over-time reductions of the input-side gate gradients (+ dsum[4H] = column sums of dG);
// lstm=1: 4H identity-mapped columns
int cpg_dgi_reduce_impl(int T, int B, int H, int lstm, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                        float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

// 0 = f32-grade products, 1 = bf16 recurrent products (cpg_set_compute_mode, api.hip)
int cpg_compute_mode_get();
