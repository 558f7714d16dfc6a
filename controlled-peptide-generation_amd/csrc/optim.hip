// Global-norm clipping + Adam on flat f32 buffers (train_vae.py:15,39-42: Adam(lr) ; clip_grad_norm_(clip) ; step()).
// The clip coefficient stays on the device (no host sync): coef = min(max_norm / (sqrt(sumsq) + 1e-6), 1).
// Duplicate-parameter semantics of the reference (word_emb.weight is yielded twice by vae_params()) are
// expressed by the caller: the embedding segment is added to the norm twice (mult=2), its gradient is scaled by coef^2
// (coef_pow=2) and cpg_adam_step is applied to it twice with consecutive step numbers.
#include "cpg_internal.h"

#define SUMSQ_BLOCKS 512

__global__ void sumsq_partial_kernel(const float* x, size_t n, float* part) {
    __shared__ float red[4];
    float s = 0.f;
    if ((((uintptr_t)x) & 15) == 0) {   // 16-byte loads, four running sums per lane, two loads in flight; the tail as scalars
        const size_t n4 = n / 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        const size_t st = (size_t)SUMSQ_BLOCKS * 256;
        for (; i + st < n4; i += 2 * st) {
            const float4 u = reinterpret_cast<const float4*>(x)[i], w = reinterpret_cast<const float4*>(x)[i + st];
            a.x += u.x * u.x + w.x * w.x;
            a.y += u.y * u.y + w.y * w.y;
            a.z += u.z * u.z + w.z * w.z;
            a.w += u.w * u.w + w.w * w.w;
        }
        if (i < n4) {
            const float4 u = reinterpret_cast<const float4*>(x)[i];
            a.x += u.x * u.x;
            a.y += u.y * u.y;
            a.z += u.z * u.z;
            a.w += u.w * u.w;
        }
        s = (a.x + a.y) + (a.z + a.w);
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += x[4 * n4 + threadIdx.x] * x[4 * n4 + threadIdx.x];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)SUMSQ_BLOCKS * 256) s += x[i] * x[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sumsq_final_kernel(const float* part, float mult, int accumulate, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = ((red[0] + red[1]) + (red[2] + red[3])) * mult;
        out[0] = accumulate ? out[0] + t : t;
    }
}

CPG_EXPORT size_t cpg_sumsq_workspace(void) { return SUMSQ_BLOCKS * sizeof(float); }

// out[0] (+)= mult * sum x^2
CPG_EXPORT int cpg_sumsq(const float* x, size_t n, float mult, int accumulate, float* out, float* workspace, void* stream) {
    CPG_CHECK_ARG(x && out && workspace && n > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, x, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, mult, accumulate, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

template <bool VEC>
__global__ void adam_step_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                                 float eps, float bc1, float bc2_sqrt, const float* sumsq, float max_norm, int coef_pow,
                                 float gscale, int step, const int32_t* iter, int step_mult) {
    __shared__ float s_bc[2];
    if (iter) {   // step number formed on the device (captured training steps): the bias corrections once per block, in double
        if (threadIdx.x == 0) {
            const double st = (double)(step_mult * iter[0] + step);
            s_bc[0] = (float)(1.0 - pow((double)b1, st));
            s_bc[1] = (float)sqrt(1.0 - pow((double)b2, st));
        }
        __syncthreads();
        bc1 = s_bc[0];
        bc2_sqrt = s_bc[1];
    }
    float coef = 1.f;
    if (sumsq) {
        const float c = fminf(max_norm / (sqrtf(sumsq[0]) * gscale + 1e-6f), 1.f);  // norm of the scaled gradient
        coef = c;
        for (int k = 1; k < coef_pow; ++k) coef *= c;
    }
    auto upd = [&](float gx, float& mx, float& vx, float& px) {
        const float gi = gx * gscale * coef;
        mx = b1 * mx + (1.f - b1) * gi;
        vx = b2 * vx + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vx) / bc2_sqrt + eps;
        px = px - (lr / bc1) * (mx / denom);
    };
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {   // four elements per thread, 16-byte accesses (the launcher checked the alignment): same arithmetic per element
        const size_t i = 4 * q;
        if (i + 3 < n) {
            float4 gv = *reinterpret_cast<const float4*>(g + i), mv = *reinterpret_cast<const float4*>(m + i);
            float4 vv = *reinterpret_cast<const float4*>(v + i), pv = *reinterpret_cast<const float4*>(p + i);
            upd(gv.x, mv.x, vv.x, pv.x);
            upd(gv.y, mv.y, vv.y, pv.y);
            upd(gv.z, mv.z, vv.z, pv.z);
            upd(gv.w, mv.w, vv.w, pv.w);
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
            *reinterpret_cast<float4*>(p + i) = pv;
        } else {
            for (size_t j = i; j < n; ++j) upd(g[j], m[j], v[j], p[j]);
        }
        return;
    }
    if (q >= n) return;
    upd(g[q], m[q], v[q], p[q]);
}

// One Adam update (torch.optim.Adam defaults: no weight decay, no amsgrad) of p[0..n) with step number `step` (1-based) - or,
// with `iter` (device int32, completed optimiser iterations), step number step_mult * iter[0] + step formed on the device.
// gscale multiplies the raw gradient first (1/world_size after a SUM all-reduce).  sumsq may be null (no clipping).
CPG_EXPORT int cpg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                             float eps, int step, const float* sumsq, float max_norm, int coef_pow, float gscale,
                             const int32_t* iter, int step_mult, void* stream) {
    CPG_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && coef_pow >= 1);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    if (n >= 4096 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v))
        hipLaunchKernelGGL(adam_step_kernel<true>, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                           lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), sumsq, max_norm, coef_pow, gscale, step, iter, step_mult);
    else
        hipLaunchKernelGGL(adam_step_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                           lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), sumsq, max_norm, coef_pow, gscale, step, iter, step_mult);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ one-launch forms (round 6)
// clip_grad_norm_ + Adam.step of the whole flat buffer as TWO launches (seven before: two partial + two final sum-of-squares launches,
// three Adam launches, + the iteration counter): cpg_sumsq_segs leaves SUMSQ_BLOCKS partials of sum_i w_i g_i^2, w_i = the multiplicity
// of the parameter element i belongs to (a parameter listed m times in the optimiser's list counts m times in the norm: SURVEY F6);
// cpg_adam_step_segs reduces the partials in every block (same order everywhere: one coefficient), then updates every element - an
// element of a parameter listed m times takes m consecutive Adam steps with its gradient scaled by coef^m, exactly what the
// per-segment launches did (/root/reference/train_vae.py:15,39-42 with the duplicate list entry of models/model.py:88-94).
struct DupSegs {
    unsigned long long off[2], end[2];   // [off, end) of the flat buffer, multiples of 4 (the caller passes the PADDED segments: padding holds zero gradients)
    int mult[2];
    int n;
};
__device__ __forceinline__ int seg_mult(const DupSegs& d, size_t i) {
    int m = 1;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (k < d.n && i >= d.off[k] && i < d.end[k]) m = d.mult[k];
    return m;
}
__global__ void sumsq_segs_kernel(const float* __restrict__ x, size_t n, DupSegs d, float* __restrict__ part) {
    __shared__ float red[4];
    const size_t n4 = n / 4;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)SUMSQ_BLOCKS * 256) {
        const float4 u = reinterpret_cast<const float4*>(x)[i];
        const float w = (float)seg_mult(d, 4 * i);
        s += w * ((u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = 4 * n4 + threadIdx.x;
        s += (float)seg_mult(d, i) * x[i] * x[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void adam_segs_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        size_t n, float lr, float b1, float b2, float eps, const float* __restrict__ part,
                                                        float* __restrict__ sumsq_out, float max_norm, float gscale, const int32_t* __restrict__ iter,
                                                        DupSegs d) {
    __shared__ float red[4];
    __shared__ float s_bc[5][2];   // [0]: parameters listed once (step iter + 1); [1 + j]: step mult * iter + j + 1 of a parameter listed mult times
    __shared__ float s_c;
    const int mm = d.n ? d.mult[0] : 1;
    if (threadIdx.x < 5) {
        const int j = threadIdx.x;
        const double st = j == 0 ? (double)iter[0] + 1.0 : (double)mm * (double)iter[0] + (double)j;
        s_bc[j][0] = (float)(1.0 - pow((double)b1, st));
        s_bc[j][1] = (float)sqrt(1.0 - pow((double)b2, st));
    }
    float coef = 1.f;
    if (part) {
        float s = part[threadIdx.x] + part[threadIdx.x + 256];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (red[0] + red[1]) + (red[2] + red[3]);
            s_c = fminf(max_norm / (sqrtf(t) * gscale + 1e-6f), 1.f);   // norm of the scaled gradient
            if (blockIdx.x == 0 && sumsq_out) sumsq_out[0] = t;
        }
    }
    __syncthreads();
    if (part) coef = s_c;
    auto upd = [&](float gi, float& mx, float& vx, float& px, float bc1, float bc2s) {
        mx = b1 * mx + (1.f - b1) * gi;
        vx = b2 * vx + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vx) / bc2s + eps;
        px = px - (lr / bc1) * (mx / denom);
    };
    const size_t i = 4 * ((size_t)blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const int mult = seg_mult(d, i);
    float cf = coef;
    for (int k = 1; k < mult; ++k) cf *= coef;
    auto elem = [&](float gx, float& mx, float& vx, float& px) {
        const float gi = gx * gscale * cf;
        if (mult == 1) upd(gi, mx, vx, px, s_bc[0][0], s_bc[0][1]);
        else
            for (int j = 1; j <= mult; ++j) upd(gi, mx, vx, px, s_bc[j][0], s_bc[j][1]);
    };
    if (i + 3 < n) {
        float4 gv = *reinterpret_cast<const float4*>(g + i), mv = *reinterpret_cast<const float4*>(m + i);
        float4 vv = *reinterpret_cast<const float4*>(v + i), pv = *reinterpret_cast<const float4*>(p + i);
        elem(gv.x, mv.x, vv.x, pv.x);
        elem(gv.y, mv.y, vv.y, pv.y);
        elem(gv.z, mv.z, vv.z, pv.z);
        elem(gv.w, mv.w, vv.w, pv.w);
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
        *reinterpret_cast<float4*>(p + i) = pv;
    } else {
        for (size_t j = i; j < n; ++j) elem(g[j], m[j], v[j], p[j]);
    }
}
static int fill_dups(DupSegs& d, int ndup, const unsigned long long* off, const unsigned long long* len, const int* mult) {
    d.n = ndup;
    for (int k = 0; k < 2; ++k) { d.off[k] = d.end[k] = 0; d.mult[k] = 1; }
    for (int k = 0; k < ndup; ++k) {
        if (off[k] % 4 || len[k] % 4 || mult[k] < 1 || mult[k] > 4 || (k > 0 && mult[k] != mult[0])) return -2;
        d.off[k] = off[k];
        d.end[k] = off[k] + len[k];
        d.mult[k] = mult[k];
    }
    return 0;
}
// workspace: cpg_sumsq_workspace() bytes.  dup_off / dup_len (elements, multiples of 4: the padded segments) / dup_mult: ndup <= 2
// parameters listed dup_mult (<= 4, the same for both) times.
CPG_EXPORT int cpg_sumsq_segs(const float* g, size_t n, int ndup, const unsigned long long* dup_off, const unsigned long long* dup_len,
                              const int* dup_mult, float* workspace, void* stream) {
    CPG_CHECK_ARG(g && workspace && n > 0 && ndup >= 0 && ndup <= 2 && aligned16(g) && (ndup == 0 || (dup_off && dup_len && dup_mult)));
    DupSegs d;
    CPG_CHECK_ARG(fill_dups(d, ndup, dup_off, dup_len, dup_mult) == 0);
    hipLaunchKernelGGL(sumsq_segs_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, d, workspace);
    CPG_LAUNCH_CHECK();
    return 0;
}
// One Adam iteration of the whole flat buffer (step numbers from the device counter `iter` = completed iterations, which the caller
// advances afterwards).  partials: cpg_sumsq_segs's workspace, or null (no clipping); sumsq_out (optional) receives the clipped norm's square.
CPG_EXPORT int cpg_adam_step_segs(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                  const float* partials, float* sumsq_out, float max_norm, float gscale, const int32_t* iter, int ndup,
                                  const unsigned long long* dup_off, const unsigned long long* dup_len, const int* dup_mult, void* stream) {
    CPG_CHECK_ARG(p && g && m && v && iter && n > 0 && ndup >= 0 && ndup <= 2 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v));
    DupSegs d;
    CPG_CHECK_ARG(fill_dups(d, ndup, dup_off, dup_len, dup_mult) == 0);
    hipLaunchKernelGGL(adam_segs_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                       beta1, beta2, eps, partials, sumsq_out, max_norm, gscale, iter, d);
    CPG_LAUNCH_CHECK();
    return 0;
}
// the two device-side step counters of a training step in one launch: *rng_base += rng_by (either may be null), *iter += iter_by
__global__ void step_counters_kernel(uint64_t* rng_base, uint64_t rng_by, int32_t* iter, int32_t iter_by) {
    if (rng_base) rng_base[0] += rng_by;
    if (iter) iter[0] += iter_by;
}
CPG_EXPORT int cpg_step_counters_add(uint64_t* rng_base, uint64_t rng_by, int32_t* iter, int32_t iter_by, void* stream) {
    CPG_CHECK_ARG(rng_base || iter);
    hipLaunchKernelGGL(step_counters_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_base, rng_by, iter, iter_by);
    CPG_LAUNCH_CHECK();
    return 0;
}
