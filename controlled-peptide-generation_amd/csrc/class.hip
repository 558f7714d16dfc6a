// CLaSS proposal sampling and z-space rejection (density_modeling.py:43-60,79-80) on the device.
//   cpg_gmm_sample        sklearn GaussianMixture.sample, diag covariance:  z = mean_k + normal * sqrt(cov_k)  (f64 -> f32)
//   cpg_lr_score_accept   binary LogisticRegression.predict_proba[:, target] per attribute, product, accept = U < product
// All draws (component of each row, normals, uniforms) are inputs.
#include "cpg_internal.h"

__global__ void gmm_sample_kernel(const double* means, const double* covars, const int32_t* comp, const double* normals,
                                  int n, int D, float* z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * D) return;
    const int r = i / D, d = i % D;
    const int k = comp[r];
    z[i] = (float)(means[(size_t)k * D + d] + normals[i] * sqrt(covars[(size_t)k * D + d]));
}

CPG_EXPORT int cpg_gmm_sample(const double* means, const double* covars, const int32_t* comp, const double* normals, int n,
                              int D, float* z, void* stream) {
    CPG_CHECK_ARG(means && covars && comp && normals && z && n > 0 && D > 0);
    const size_t tot = (size_t)n * D;
    hipLaunchKernelGGL(gmm_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, means,
                       covars, comp, normals, n, D, z);
    CPG_LAUNCH_CHECK();
    return 0;
}

// One wave per row: f64 dot products against up to CPG_MAX_ATTR classifier directions, k-ordered per lane then a
// butterfly reduction (fixed order => deterministic).
#define CPG_MAX_ATTR 8
__global__ void lr_score_accept_kernel(const float* z, int n, int D, const double* coef, const double* intercept,
                                       const int32_t* target, int A, const double* uniforms, double* probs, double* accum,
                                       uint8_t* accepted) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= n) return;
    double dot[CPG_MAX_ATTR];
#pragma unroll
    for (int a = 0; a < CPG_MAX_ATTR; ++a) dot[a] = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double x = (double)z[(size_t)row * D + d];
#pragma unroll
        for (int a = 0; a < CPG_MAX_ATTR; ++a)
            if (a < A) dot[a] += x * coef[(size_t)a * D + d];
    }
    double acc = 1.0;
#pragma unroll
    for (int a = 0; a < CPG_MAX_ATTR; ++a) {
        if (a >= A) break;
        const double s = wave_sum_d(dot[a]) + intercept[a];
        const double p1 = 1.0 / (1.0 + exp(-s));
        const double p = target[a] == 1 ? p1 : 1.0 - p1;
        if (lane == 0) probs[(size_t)a * n + row] = p;
        acc *= p;
    }
    if (lane == 0) {
        accum[row] = acc;
        accepted[row] = uniforms[row] < acc ? 1 : 0;
    }
}

CPG_EXPORT int cpg_lr_score_accept(const float* z, int n, int D, const double* coef, const double* intercept,
                                   const int32_t* target, int A, const double* uniforms, double* probs, double* accum,
                                   uint8_t* accepted, void* stream) {
    CPG_CHECK_ARG(z && coef && intercept && target && uniforms && probs && accum && accepted);
    CPG_CHECK_ARG(n > 0 && D > 0 && A > 0 && A <= CPG_MAX_ATTR);
    hipLaunchKernelGGL(lr_score_accept_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, z, n, D, coef, intercept,
                       target, A, uniforms, probs, accum, accepted);
    CPG_LAUNCH_CHECK();
    return 0;
}


// ---- residue rows of decoded ids (the compaction idx2sentences(..., print_special_tokens=False) does per row in python,
// data_processing/dataset.py:285-300 via sample_pipeline.py:129-139): ids int16 [n, L] (< first_residue = special or padding) ->
// letters uint8 [n, L] (the ids >= first_residue of the row, left-aligned, zero-filled) and their count.  One thread per row,
// 2 L bytes in and L bytes out per row: HBM-bound and tiny (the tensor-op form - int64 cast, cumsum, scatter - took 20 ms per
// million rows).
__global__ void residue_rows_kernel(const int16_t* ids, size_t n, int L, int first_residue, uint8_t* letters, int32_t* counts) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int16_t* r = ids + row * L;
    uint8_t* o = letters + row * L;
    int k = 0;
    for (int j = 0; j < L; ++j) {
        const int v = r[j];
        if (v >= first_residue) o[k++] = (uint8_t)v;
    }
    counts[row] = k;
    for (; k < L; ++k) o[k] = 0;
}
CPG_EXPORT int cpg_residue_rows(const int16_t* ids, size_t n, int L, int first_residue, uint8_t* letters, int32_t* counts,
                                void* stream) {
    CPG_CHECK_ARG(ids && letters && counts && n > 0 && L > 0 && first_residue >= 0);
    hipLaunchKernelGGL(residue_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, n, L,
                       first_residue, letters, counts);
    CPG_LAUNCH_CHECK();
    return 0;
}
