// CNN sequence classifier forward and backward (models/classifier.py:39-60; reached through q_c='classifier',
// models/model.py:186-188, where the reference lets gradients flow through c = softmax(classifier(x)) although it never trains
// the module itself).  ADJACENT row.
//
// With token inputs the convolution collapses to table look-ups, like the GRU input projection: for a filter of width w
//   conv[b,f,p] = bias[f] + sum_dw  tab[dw][tok[b,p+dw]][f],   tab[dw] = emb @ W[:,0,dw,:]^T   ([V,F], built by cpg_linear_fwd)
// followed by ReLU and max over p (F.max_pool1d over the whole length).  The final Linear runs on the MFMA engine.
#include "cpg_internal.h"

// tabs: filters of widths min_w .. min_w+nconv-1 back to back, layer l occupying w_l*V*F floats ([dw][v][f]).
__global__ void cnn_pool_kernel(const int64_t* ids, int B, int T, int V, int F, int min_w, int nconv, const float* tabs,
                                const float* bias, float* pooled, int16_t* argpos) {
    const int b = blockIdx.x, f = threadIdx.x;
    if (f >= F) return;
    const int64_t* row = ids + (size_t)b * T;
    size_t base = 0;
    for (int l = 0; l < nconv; ++l) {
        const int w = min_w + l;
        float best = 0.f;  // ReLU output is >= 0 and there is at least one position
        int arg = -1;      // first position whose pre-activation is the positive maximum; -1: ReLU is flat there (no gradient)
        for (int p = 0; p + w <= T; ++p) {
            float s = bias[l * F + f];
            for (int dw = 0; dw < w; ++dw) s += tabs[(base + (size_t)dw * V + (size_t)row[p + dw]) * F + f];
            if (s > best) {
                best = s;
                arg = p;
            }
        }
        pooled[(size_t)b * nconv * F + l * F + f] = best;
        if (argpos) argpos[(size_t)b * nconv * F + l * F + f] = (int16_t)arg;
        base += (size_t)w * V;
    }
}

// Backward of the pooled features: the gradient of pooled[b, l, f] goes to the ONE position its maximum came from (max_pool1d;
// none when the ReLU was flat), i.e. to the bias and to the w table rows that position read.  One thread per (layer, filter)
// walks the batch in order and accumulates its [w, V] slice in LDS: fixed summation order, no atomics.
__global__ void cnn_pool_bwd_kernel(const int64_t* ids, const int16_t* argpos, const float* dpooled, int B, int T, int V, int F,
                                    int min_w, int nconv, float* dtabs, float* dbias) {
    extern __shared__ float acc[];   // [blockDim.x][w * V]
    const int l = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = min_w + l;
    size_t base = 0;
    for (int k = 0; k < l; ++k) base += (size_t)(min_w + k) * V;
    float* a = acc + (size_t)threadIdx.x * w * V;
    for (int i = 0; i < w * V; ++i) a[i] = 0.f;
    if (f >= F) return;
    float db = 0.f;
    for (int b = 0; b < B; ++b) {
        const int p = argpos[(size_t)b * nconv * F + l * F + f];
        if (p < 0) continue;
        const float g = dpooled[(size_t)b * nconv * F + l * F + f];
        db += g;
        for (int dw = 0; dw < w; ++dw) a[dw * V + (int)ids[(size_t)b * T + p + dw]] += g;
    }
    dbias[l * F + f] = db;
    for (int i = 0; i < w * V; ++i) dtabs[(base + i) * F + f] = a[i];
}

CPG_EXPORT int cpg_cnn_classifier_pool_bwd(const int64_t* ids, const int16_t* argpos, const float* dpooled, int B, int T, int V,
                                           int F, int min_width, int nconv, float* dtabs, float* dbias, void* stream) {
    CPG_CHECK_ARG(ids && argpos && dpooled && dtabs && dbias && B > 0 && V > 0 && F > 0 && min_width > 0 && nconv > 0);
    const int wmax = min_width + nconv - 1;
    const size_t smem = (size_t)64 * wmax * V * sizeof(float);
    CPG_CHECK_ARG(smem <= 64 * 1024);
    hipLaunchKernelGGL(cnn_pool_bwd_kernel, dim3((F + 63) / 64, nconv), dim3(64), smem, (hipStream_t)stream, ids, argpos, dpooled,
                       B, T, V, F, min_width, nconv, dtabs, dbias);
    CPG_LAUNCH_CHECK();
    return 0;
}

CPG_EXPORT int cpg_cnn_classifier_pool(const int64_t* ids, int B, int T, int V, int F, int min_width, int nconv,
                                       const float* tabs, const float* bias, float* pooled, int16_t* argpos, void* stream) {
    CPG_CHECK_ARG(ids && tabs && bias && pooled && B > 0 && T >= min_width + nconv - 1 && V > 0 && F > 0 && F <= 1024);
    CPG_CHECK_ARG(min_width > 0 && nconv > 0);
    hipLaunchKernelGGL(cnn_pool_kernel, dim3(B), dim3(((F + 63) / 64) * 64), 0, (hipStream_t)stream, ids, B, T, V, F, min_width,
                       nconv, tabs, bias, pooled, argpos);
    CPG_LAUNCH_CHECK();
    return 0;
}
