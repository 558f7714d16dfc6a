// CNN sequence classifier forward (models/classifier.py:39-60; reached through q_c='classifier', models/model.py:186-188).
// ADJACENT row, inference only - the reference never trains this module.
//
// With token inputs the convolution collapses to table look-ups, like the GRU input projection: for a filter of width w
//   conv[b,f,p] = bias[f] + sum_dw  tab[dw][tok[b,p+dw]][f],   tab[dw] = emb @ W[:,0,dw,:]^T   ([V,F], built by cpg_linear_fwd)
// followed by ReLU and max over p (F.max_pool1d over the whole length).  The final Linear runs on the MFMA engine.
#include "cpg_internal.h"

// tabs: filters of widths min_w .. min_w+nconv-1 back to back, layer l occupying w_l*V*F floats ([dw][v][f]).
__global__ void cnn_pool_kernel(const int64_t* ids, int B, int T, int V, int F, int min_w, int nconv, const float* tabs,
                                const float* bias, float* pooled) {
    const int b = blockIdx.x, f = threadIdx.x;
    if (f >= F) return;
    const int64_t* row = ids + (size_t)b * T;
    size_t base = 0;
    for (int l = 0; l < nconv; ++l) {
        const int w = min_w + l;
        float best = 0.f;  // ReLU output is >= 0 and there is at least one position
        for (int p = 0; p + w <= T; ++p) {
            float s = bias[l * F + f];
            for (int dw = 0; dw < w; ++dw) s += tabs[(base + (size_t)dw * V + (size_t)row[p + dw]) * F + f];
            best = fmaxf(best, s);
        }
        pooled[(size_t)b * nconv * F + l * F + f] = best;
        base += (size_t)w * V;
    }
}

CPG_EXPORT int cpg_cnn_classifier_pool(const int64_t* ids, int B, int T, int V, int F, int min_width, int nconv,
                                       const float* tabs, const float* bias, float* pooled, void* stream) {
    CPG_CHECK_ARG(ids && tabs && bias && pooled && B > 0 && T >= min_width + nconv - 1 && V > 0 && F > 0 && F <= 1024);
    CPG_CHECK_ARG(min_width > 0 && nconv > 0);
    hipLaunchKernelGGL(cnn_pool_kernel, dim3(B), dim3(((F + 63) / 64) * 64), 0, (hipStream_t)stream, ids, B, T, V, F, min_width,
                       nconv, tabs, bias, pooled);
    CPG_LAUNCH_CHECK();
    return 0;
}
