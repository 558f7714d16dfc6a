// Loss terms of the WAE/VAE training step (reference losses.py, train_vae.py:27-37) and their gradients.
// Every reduction is two-stage with a fixed partition (block partials in a fixed order, then one block) so results are
// run-to-run deterministic; sums and counts are returned separately so a data-parallel caller can all-reduce them and
// still obtain the single-device value.
#include "gemm_core.h"
#include "cpg_internal.h"
#include "rng_core.h"

#define RED_BLOCKS 256

// block-wide sum of NV values per thread; result valid in thread 0. 256 threads.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /*[4*NV]*/) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[w * NV + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = (red[k] + red[NV + k]) + (red[2 * NV + k] + red[3 * NV + k]);
}

template <int NV>
__global__ void final_sum_kernel(const float* part, int nblocks, float* out) {
    __shared__ float red[4 * NV];
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += part[(size_t)i * NV + k];
    block_sum<NV>(v, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = v[k];
}

// ------------------------------------------------------------------------------------------ reconstruction CE
// losses.recon_dec (losses.py:18-31): targets = cat(ids[:,1:], PAD); mean NLL over non-PAD targets of the whole batch.
__global__ void recon_ce_partial_kernel(const int64_t* ids, const float* logits, int B, int T, int V, int pad, float* part) {
    __shared__ float red[8];
    float v[2] = {0.f, 0.f};
    const int rows = B * T;
    for (int row = blockIdx.x * 256 + threadIdx.x; row < rows; row += RED_BLOCKS * 256) {
        const int b = row / T, t = row % T;
        const int tgt = (t + 1 < T) ? (int)ids[(size_t)b * T + t + 1] : pad;
        if (tgt == pad) continue;
        const float* l = logits + (size_t)row * V;
        float m = -INFINITY;
        for (int k = 0; k < V; ++k) m = fmaxf(m, l[k]);
        float se = 0.f;
        for (int k = 0; k < V; ++k) se += expf(l[k] - m);
        v[0] += (m + logf(se)) - l[tgt];
        v[1] += 1.f;
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = v[0];
        part[blockIdx.x * 2 + 1] = v[1];
    }
}

// out[0] = sum of NLL over valid targets, out[1] = number of valid targets
CPG_EXPORT int cpg_recon_ce_fwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, float* out,
                                float* workspace, void* stream) {
    CPG_CHECK_ARG(ids && logits && out && workspace && B > 0 && T > 0 && V > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(recon_ce_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, ids, logits, B, T, V, pad, workspace);
    hipLaunchKernelGGL(final_sum_kernel<2>, dim3(1), dim3(256), 0, s, (const float*)workspace, RED_BLOCKS, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

// out[0], out[1] as cpg_recon_ce_fwd; out[2] = out[0] / max(out[1], 1): the loss itself, formed in the reduction's last block
// (the same division the host-side wrapper issued as two more launches)
__global__ void recon_ce_final3_kernel(const float* part, int blocks, float* out) {
    __shared__ float red[8];
    float v[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < blocks; i += 256) {
        v[0] += part[2 * i];
        v[1] += part[2 * i + 1];
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        out[0] = v[0];
        out[1] = v[1];
        out[2] = v[0] / fmaxf(v[1], 1.f);
    }
}
CPG_EXPORT int cpg_recon_ce_loss_fwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, float* out,
                                     float* workspace, void* stream) {
    CPG_CHECK_ARG(ids && logits && out && workspace && B > 0 && T > 0 && V > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(recon_ce_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, ids, logits, B, T, V, pad, workspace);
    hipLaunchKernelGGL(recon_ce_final3_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, RED_BLOCKS, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

// The trainer's form (round 6): cross-entropy of TIME-MAJOR logits [T B, V] (what the vocabulary projection writes: no transposition
// to [B,T,V] and back) that also leaves the UNSCALED logit gradient softmax - onehot of every scored row (zeros on <pad> targets) -
// the backward of the vocabulary projection multiplies it by gout / count on the way in (cpg_vocab_fc_bwd's g / count), so the
// forward needs neither the count nor the upstream gradient.  Same arithmetic per row as recon_ce_partial_kernel / recon_ce_bwd_kernel.
__global__ void recon_ce_tm_kernel(const int64_t* __restrict__ ids, const float* __restrict__ logits, int B, int T, int V, int pad,
                                   float* __restrict__ part, float* __restrict__ dl) {
    __shared__ float red[8];
    float v[2] = {0.f, 0.f};
    const int rows = B * T;
    const bool v4 = V % 4 == 0 && V <= 32 && ((((uintptr_t)logits) | ((uintptr_t)dl)) & 15) == 0;
    for (int row = blockIdx.x * 256 + threadIdx.x; row < rows; row += RED_BLOCKS * 256) {
        const int t = row / B, b = row - t * B;
        const int tgt = (t + 1 < T) ? (int)ids[(size_t)b * T + t + 1] : pad;
        float* d = dl + (size_t)row * V;
        const float* l = logits + (size_t)row * V;
        if (tgt == pad) {
            if (v4)
                for (int k = 0; k < V; k += 4) *reinterpret_cast<float4*>(d + k) = make_float4(0.f, 0.f, 0.f, 0.f);
            else
                for (int k = 0; k < V; ++k) d[k] = 0.f;
            continue;
        }
        if (v4) {
            float x[32];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (4 * q < V) {
                    const float4 w = *reinterpret_cast<const float4*>(l + 4 * q);
                    x[4 * q] = w.x; x[4 * q + 1] = w.y; x[4 * q + 2] = w.z; x[4 * q + 3] = w.w;
                }
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < V) m = fmaxf(m, x[k]);
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < V) se += expf(x[k] - m);
            const float lse = m + logf(se);
            float xt = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < V && k == tgt) xt = x[k];
            v[0] += lse - xt;
            v[1] += 1.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (4 * q < V) {
                    float4 o;
                    o.x = expf(x[4 * q] - lse) - (4 * q == tgt ? 1.f : 0.f);
                    o.y = expf(x[4 * q + 1] - lse) - (4 * q + 1 == tgt ? 1.f : 0.f);
                    o.z = expf(x[4 * q + 2] - lse) - (4 * q + 2 == tgt ? 1.f : 0.f);
                    o.w = expf(x[4 * q + 3] - lse) - (4 * q + 3 == tgt ? 1.f : 0.f);
                    *reinterpret_cast<float4*>(d + 4 * q) = o;
                }
            continue;
        }
        float m = -INFINITY;
        for (int k = 0; k < V; ++k) m = fmaxf(m, l[k]);
        float se = 0.f;
        for (int k = 0; k < V; ++k) se += expf(l[k] - m);
        const float lse = m + logf(se);
        v[0] += lse - l[tgt];
        v[1] += 1.f;
        for (int k = 0; k < V; ++k) d[k] = expf(l[k] - lse) - (k == tgt ? 1.f : 0.f);
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = v[0];
        part[blockIdx.x * 2 + 1] = v[1];
    }
}
// out[0] = sum of NLL, out[1] = number of scored targets, out[2] = out[0] / max(out[1], 1); dl [T B, V] = softmax - onehot (unscaled)
CPG_EXPORT int cpg_recon_ce_tm_fwd(const int64_t* ids, const float* logits_tm, int B, int T, int V, int pad, float* out, float* dl,
                                   float* workspace, void* stream) {
    CPG_CHECK_ARG(ids && logits_tm && out && dl && workspace && B > 0 && T > 0 && V > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(recon_ce_tm_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, ids, logits_tm, B, T, V, pad, workspace, dl);
    hipLaunchKernelGGL(recon_ce_final3_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, RED_BLOCKS, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ---- loss = t0 + w1 t1 + w2 t2 + w3 t3 (train_vae.py:35-37) as ONE launch, and its gradient fan-out as one: the host-side
// expression costs three multiplies, three adds and three backward multiplies of one element each.  Products and sums are
// rounded one by one, left to right, exactly as the separate element-wise kernels round them.
struct WSum4 {
    const float* t[4];
    float w[4];
};
__global__ void weighted_sum4_kernel(WSum4 a, const float* wdev, float* out) {
    float s = 0.f;
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!a.t[i]) continue;
        const float w = wdev ? wdev[i] : a.w[i];
        const float p = w == 1.f ? a.t[i][0] : __fmul_rn(w, a.t[i][0]);
        s = any ? __fadd_rn(s, p) : p;
        any = true;
    }
    out[0] = s;
}
__global__ void scale_fanout4_kernel(const float* g, WSum4 a, const float* wdev, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = __fmul_rn(g[0], wdev ? wdev[i] : a.w[i]);
}
// out[0] = sum_i w_i * t_i[0] over the non-null terms (device scalars), in order.  wdev (optional, device float[4]) replaces
// the host weights: a captured training step reads the annealed beta from memory the host updates between replays.
CPG_EXPORT int cpg_weighted_sum4(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1,
                                 float w2, float w3, const float* wdev, float* out, void* stream) {
    CPG_CHECK_ARG(out && (t0 || t1 || t2 || t3));
    WSum4 a{{t0, t1, t2, t3}, {w0, w1, w2, w3}};
    hipLaunchKernelGGL(weighted_sum4_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, wdev, out);
    CPG_LAUNCH_CHECK();
    return 0;
}
// out[i] = g[0] * w_i, i = 0..3
CPG_EXPORT int cpg_scale_fanout4(const float* g, float w0, float w1, float w2, float w3, const float* wdev, float* out,
                                 void* stream) {
    CPG_CHECK_ARG(g && out);
    WSum4 a{{nullptr, nullptr, nullptr, nullptr}, {w0, w1, w2, w3}};
    hipLaunchKernelGGL(scale_fanout4_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, a, wdev, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

// dlogits = gout * (softmax - onehot) / count on valid rows, 0 elsewhere.  gout and count are device scalars (no host sync).
__global__ void recon_ce_bwd_kernel(const int64_t* ids, const float* logits, int B, int T, int V, int pad, const float* gout,
                                    const float* count, float* dlogits) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    const int tgt = (t + 1 < T) ? (int)ids[(size_t)b * T + t + 1] : pad;
    float* d = dlogits + (size_t)row * V;
    if (tgt == pad) {
        if (V % 4 == 0 && (((uintptr_t)dlogits) & 15) == 0)
            for (int k = 0; k < V; k += 4) *reinterpret_cast<float4*>(d + k) = make_float4(0.f, 0.f, 0.f, 0.f);
        else
            for (int k = 0; k < V; ++k) d[k] = 0.f;
        return;
    }
    const float* l = logits + (size_t)row * V;
    const float sc = gout[0] / fmaxf(count[0], 1.f);
    if (V % 4 == 0 && V <= 32 && ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0) {
        // the row in registers: 16-byte loads / stores (a thread's row is V contiguous floats), the same operations in the same order
        float x[32];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (4 * q < V) {
                const float4 v = *reinterpret_cast<const float4*>(l + 4 * q);
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < V) m = fmaxf(m, x[k]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (k < V) se += expf(x[k] - m);
        const float lse = m + logf(se);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (4 * q < V) {
                float4 o;
                o.x = (expf(x[4 * q] - lse) - (4 * q == tgt ? 1.f : 0.f)) * sc;
                o.y = (expf(x[4 * q + 1] - lse) - (4 * q + 1 == tgt ? 1.f : 0.f)) * sc;
                o.z = (expf(x[4 * q + 2] - lse) - (4 * q + 2 == tgt ? 1.f : 0.f)) * sc;
                o.w = (expf(x[4 * q + 3] - lse) - (4 * q + 3 == tgt ? 1.f : 0.f)) * sc;
                *reinterpret_cast<float4*>(d + 4 * q) = o;
            }
        return;
    }
    float m = -INFINITY;
    for (int k = 0; k < V; ++k) m = fmaxf(m, l[k]);
    float se = 0.f;
    for (int k = 0; k < V; ++k) se += expf(l[k] - m);
    const float lse = m + logf(se);
    for (int k = 0; k < V; ++k) d[k] = (expf(l[k] - lse) - (k == tgt ? 1.f : 0.f)) * sc;
}

CPG_EXPORT int cpg_recon_ce_bwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, const float* gout,
                                const float* count, float* dlogits, void* stream) {
    CPG_CHECK_ARG(ids && logits && gout && count && dlogits && B > 0 && T > 0 && V > 0);
    hipLaunchKernelGGL(recon_ce_bwd_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, (hipStream_t)stream, ids, logits, B, T, V,
                       pad, gout, count, dlogits);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ reparameterisation
// RNN_VAE.sample_z (models/model.py:107-112): z = mu + exp(logvar/2) * eps
__global__ void reparam_fwd_kernel(const float* mu, const float* lv, const float* eps, float* z, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = mu[i] + expf(lv[i] * 0.5f) * eps[i];
}
__global__ void reparam_bwd_kernel(const float* dz, const float* lv, const float* eps, float* dmu, float* dlv, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dmu[i] = dz[i];
        dlv[i] = dz[i] * eps[i] * 0.5f * expf(lv[i] * 0.5f);
    }
}
CPG_EXPORT int cpg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n, void* stream) {
    CPG_CHECK_ARG(mu && logvar && eps && z && n > 0);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar,
                       eps, z, n);
    CPG_LAUNCH_CHECK();
    return 0;
}
CPG_EXPORT int cpg_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, size_t n,
                               void* stream) {
    CPG_CHECK_ARG(dz && logvar && eps && dmu && dlogvar && n > 0);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz, logvar,
                       eps, dmu, dlogvar, n);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ latent statistics
// out[0] = sum 0.5(e^lv + mu^2 - 1 - lv)   (kl_gaussianprior * B,      losses.py:8-10)
// out[1] = sum 0.5(e^lv - 1 - lv)          (kl_gaussian_sharedmu * B,  losses.py:13-15)
// out[2] = sum |lv|                        (z_logvar_L1 * B,           train_vae.py:33)
// out[3] = sum |mu| ; out[4] = sum lv      (logged means,              train_vae.py:44-45)
__global__ void latent_stats_partial_kernel(const float* mu, const float* lv, size_t n, float* part) {
    __shared__ float red[20];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RED_BLOCKS * 256) {
        const float m = mu[i], l = lv[i], e = expf(l);
        v[0] += 0.5f * (e + m * m - 1.f - l);
        v[1] += 0.5f * (e - 1.f - l);
        v[2] += fabsf(l);
        v[3] += fabsf(m);
        v[4] += l;
    }
    block_sum<5>(v, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) part[blockIdx.x * 5 + k] = v[k];
}
CPG_EXPORT int cpg_latent_stats_fwd(const float* mu, const float* logvar, size_t n, float* out, float* workspace,
                                    void* stream) {
    CPG_CHECK_ARG(mu && logvar && out && workspace && n > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(latent_stats_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, mu, logvar, n, workspace);
    hipLaunchKernelGGL(final_sum_kernel<5>, dim3(1), dim3(256), 0, s, (const float*)workspace, RED_BLOCKS, out);
    CPG_LAUNCH_CHECK();
    return 0;
}
// dmu (+)= g_kl*mu/B ; dlv (+)= (g_kl+g_klmu)*0.5(e^lv-1)/B + g_l1*sign(lv)/B.  g_* are device scalars or null.
__global__ void latent_stats_bwd_kernel(const float* mu, const float* lv, size_t n, float invB, const float* g_kl,
                                        const float* g_klmu, const float* g_l1, float* dmu, float* dlv, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = g_kl ? g_kl[0] : 0.f, b = g_klmu ? g_klmu[0] : 0.f, c = g_l1 ? g_l1[0] : 0.f;
    const float l = lv[i];
    const float sgn = (l > 0.f) ? 1.f : ((l < 0.f) ? -1.f : 0.f);
    const float gm = a * mu[i] * invB;
    const float gl = ((a + b) * 0.5f * (expf(l) - 1.f) + c * sgn) * invB;
    dmu[i] = accumulate ? dmu[i] + gm : gm;
    dlv[i] = accumulate ? dlv[i] + gl : gl;
}
CPG_EXPORT int cpg_latent_stats_bwd(const float* mu, const float* logvar, size_t n, int B, const float* g_kl,
                                    const float* g_klmu, const float* g_l1, float* dmu, float* dlogvar, int accumulate,
                                    void* stream) {
    CPG_CHECK_ARG(mu && logvar && dmu && dlogvar && n > 0 && B > 0);
    hipLaunchKernelGGL(latent_stats_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu,
                       logvar, n, 1.f / (float)B, g_kl, g_klmu, g_l1, dmu, dlogvar, accumulate);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ the latent block of a training step, fused
// RNN_VAE.sample_z + sample_c_prior + GRUDecoder.init_hidden + the three analytic latent penalties (models/model.py:107-126,
// models/decoder.py:53-54, losses.py:8-15, train_vae.py:33) as ONE elementwise launch (+ the 5-value final sum):
//   eps ~ N(0,1) drawn here when not injected (the numbers cpg_rng_normal would have written: rng_core.h), c ~ Cat(.5,.5) likewise
//   z = mu + exp(logvar / 2) eps  ->  z [B,Z] AND the decoder's initial state / constant input zc = [z ; c] [B, Z + C] (no torch.cat)
//   partial sums of the five latent statistics (latent_stats_partial_kernel's) per block
// and its backward as one launch: dmu, dlogvar from the gradients on z, on zc[:, :Z] and on the three penalties.
struct LatentFwdArgs {
    const float* mu; const float* lv;
    const float* eps_in;            // [B,Z] or null: draw (seed, off_eps, base)
    const float* c_in;              // [B,C] or null: draw one-hot rows of Bernoulli(p_one) (seed, off_c, base), C == 2
    float* eps_out;                 // [B,Z] or null (kept for the backward pass when eps is drawn here)
    float* z; float* zc; float* c_out;   // [B,Z], [B,Z+C], [B,C]
    float* part;                    // [RED_BLOCKS][5]
    int B, Z, C;
    uint64_t seed, off_eps, off_c;
    const uint64_t* base;
    float p_one;
};
__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentFwdArgs a) {
    __shared__ float red[20];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t n = (size_t)a.B * a.Z, nq = (n + 3) / 4;
    const uint64_t base = a.base ? *a.base : 0;
    const int ZC = a.Z + a.C;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)RED_BLOCKS * 256) {
        float e4[4];
        if (!a.eps_in) philox_normal4(a.seed, a.off_eps + base + q, e4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = 4 * q + k;
            if (i >= n) break;
            const float m = a.mu[i], l = a.lv[i];
            const float ep = a.eps_in ? a.eps_in[i] : e4[k];
            const float zz = m + expf(l * 0.5f) * ep;
            const size_t b = i / a.Z;
            const int j = (int)(i - b * a.Z);
            a.z[i] = zz;
            a.zc[b * ZC + j] = zz;
            if (a.eps_out) a.eps_out[i] = ep;
            const float ex = expf(l);
            v[0] += 0.5f * (ex + m * m - 1.f - l);
            v[1] += 0.5f * (ex - 1.f - l);
            v[2] += fabsf(l);
            v[3] += fabsf(m);
            v[4] += l;
        }
    }
    // the class block: rows of c, copied into zc[:, Z:]
    if (a.c_in) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)a.B * a.C; i += (size_t)RED_BLOCKS * 256) {
            const size_t b = i / a.C;
            const int j = (int)(i - b * a.C);
            const float cv = a.c_in[i];
            a.zc[b * ZC + a.Z + j] = cv;
            if (a.c_out) a.c_out[i] = cv;
        }
    } else if (a.C == 2) {
        const size_t nb4 = ((size_t)a.B + 3) / 4;
        for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nb4; q += (size_t)RED_BLOCKS * 256) {
            uint32_t r[4];
            philox4x32(a.seed, a.off_c + base + q, 2u, r);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t b = 4 * q + k;
                if (b >= (size_t)a.B) break;
                const bool one = u01(r[k]) < a.p_one;
                const float c0 = one ? 0.f : 1.f, c1 = one ? 1.f : 0.f;
                a.zc[b * ZC + a.Z] = c0;
                a.zc[b * ZC + a.Z + 1] = c1;
                a.c_out[2 * b] = c0;
                a.c_out[2 * b + 1] = c1;
            }
        }
    }
    block_sum<5>(v, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) a.part[blockIdx.x * 5 + k] = v[k];
}
// out[0..2] = the three penalties (sums / B), out[3..4] = sum |mu|, sum logvar (logged means)
__global__ void latent_final_kernel(const float* part, float invB, float* out) {
    __shared__ float red[20];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256)
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += part[(size_t)i * 5 + k];
    block_sum<5>(v, red);
    if (threadIdx.x == 0) {
        out[0] = v[0] * invB; out[1] = v[1] * invB; out[2] = v[2] * invB;
        out[3] = v[3]; out[4] = v[4];
    }
}
CPG_EXPORT size_t cpg_latent_fused_workspace(void) { return (size_t)RED_BLOCKS * 5 * sizeof(float); }
CPG_EXPORT int cpg_latent_fused_fwd(const float* mu, const float* logvar, const float* eps_in, const float* c_in, int B, int Z, int C,
                                    uint64_t seed, uint64_t off_eps, uint64_t off_c, const uint64_t* base, float p_one, float* eps_out,
                                    float* z, float* zc, float* c_out, float* out5, float* workspace, void* stream) {
    CPG_CHECK_ARG(mu && logvar && z && zc && out5 && workspace && B > 0 && Z > 0 && C >= 0 && (c_in || C == 0 || (C == 2 && c_out)));
    LatentFwdArgs a{mu, logvar, eps_in, c_in, eps_out, z, zc, c_out, workspace, B, Z, C, seed, off_eps, off_c, base, p_one};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, a);
    hipLaunchKernelGGL(latent_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, 1.f / (float)B, out5);
    CPG_LAUNCH_CHECK();
    return 0;
}
// dmu = dzt + g_kl mu / B ;  dlogvar = dzt eps exp(logvar / 2) / 2 + ((g_kl + g_klmu) (e^logvar - 1) / 2 + g_l1 sign(logvar)) / B
// with dzt = dz + dzc[:, :Z] (either may be null); g_* device scalars or null.
// dmu / dlv rows have stride ldo >= Zp and are written for Zp >= Z columns, zeros from column Z on (Zp > Z: the padded pair
// [dmu | dlv] the encoder heads' backward products read as whole 32-deep slabs)
__global__ void latent_fused_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ dzc, int ldzc, const float* __restrict__ mu,
                                        const float* __restrict__ lv, const float* __restrict__ eps, int B, int Z, float invB,
                                        const float* g_kl, const float* g_klmu, const float* g_l1, float* __restrict__ dmu,
                                        float* __restrict__ dlv, int ldo, int Zp) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * Zp) return;
    const size_t row = t / Zp;
    const int j = (int)(t - row * Zp);
    const size_t o = row * ldo + j;
    if (j >= Z) {
        dmu[o] = 0.f;
        dlv[o] = 0.f;
        return;
    }
    const size_t i = row * Z + j;
    const float a = g_kl ? g_kl[0] : 0.f, b = g_klmu ? g_klmu[0] : 0.f, c = g_l1 ? g_l1[0] : 0.f;
    float d = dz ? dz[i] : 0.f;
    if (dzc) d += dzc[row * ldzc + j];
    const float l = lv[i];
    const float sgn = (l > 0.f) ? 1.f : ((l < 0.f) ? -1.f : 0.f);
    dmu[o] = d + a * mu[i] * invB;
    dlv[o] = d * eps[i] * 0.5f * expf(l * 0.5f) + ((a + b) * 0.5f * (expf(l) - 1.f) + c * sgn) * invB;
}
CPG_EXPORT int cpg_latent_fused_bwd(const float* dz, const float* dzc, int ldzc, const float* mu, const float* logvar, const float* eps,
                                    int B, int Z, const float* g_kl, const float* g_klmu, const float* g_l1, float* dmu, float* dlogvar,
                                    int ldo, int Zp, void* stream) {
    CPG_CHECK_ARG(mu && logvar && eps && dmu && dlogvar && B > 0 && Z > 0 && (!dzc || ldzc >= Z) && Zp >= Z && ldo >= Zp);
    const size_t n = (size_t)B * Zp;
    hipLaunchKernelGGL(latent_fused_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz, dzc, ldzc, mu, logvar,
                       eps, B, Z, 1.f / (float)B, g_kl, g_klmu, g_l1, dmu, dlogvar, ldo, Zp);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ MMD, random Fourier features
// losses.mmd_rf / compute_gaussian_rf (losses.py:59-93): phi(x) = cos(x W / sigma + b) * sqrt(2/R); loss = |mean phi(z1) - mean phi(z2)|^2
// raw = z @ rf_w comes from the MFMA engine (cpg_matmul_nn); this kernel adds the phase, takes cos and column-sums.
__global__ void rf_colsum_partial_kernel(const float* raw, const float* rf_b, int Bn, int R, float inv_sigma, float amp,
                                         int rows_per_chunk, float* part) {
    __shared__ float red[4][64];
    const int r = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    const int mb = blockIdx.y * rows_per_chunk, me = min(Bn, mb + rows_per_chunk);
    float s = 0.f;
    if (r < R) {
        const float ph = rf_b[r];
        for (int m = mb + ty; m < me; m += 4) s += cosf(raw[(size_t)m * R + r] * inv_sigma + ph) * amp;
    }
    red[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && r < R)
        part[(size_t)blockIdx.y * R + r] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void rf_colsum_final_kernel(const float* part, int chunks, int R, float* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= chunks; c += 8) {   // eight loads in flight, added in chunk order (the sums of the plain loop)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * R + r];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < chunks; ++c) s += part[(size_t)c * R + r];
    out[r] = s;
}
// feature column sums: sums[R] = sum_b phi(z_b)   (raw [B,R] = z @ rf_w computed by the caller)
CPG_EXPORT int cpg_rf_feature_sums(const float* raw, const float* rf_b, int Bn, int R, float sigma, float* sums,
                                   float* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(raw && rf_b && sums && workspace && Bn > 0 && R > 0 && sigma > 0.f);
    hipStream_t s = (hipStream_t)stream;
    int chunks = cdiv(Bn, 64);
    if (chunks > 128) chunks = 128;
    const int rpc = cdiv(Bn, chunks);
    chunks = cdiv(Bn, rpc);
    CPG_CHECK_ARG(workspace_bytes >= (size_t)chunks * R * sizeof(float));
    const float amp = sqrtf(2.f / (float)R);
    hipLaunchKernelGGL(rf_colsum_partial_kernel, dim3(cdiv(R, 64), chunks), dim3(64, 4), 0, s, raw, rf_b, Bn, R, 1.f / sigma,
                       amp, rpc, workspace);
    hipLaunchKernelGGL(rf_colsum_final_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, (const float*)workspace, chunks, R, sums);
    CPG_LAUNCH_CHECK();
    return 0;
}
// loss[0] = sum_r ((s1[r]-s2[r])*invB)^2 ; diff[r] = (s1[r]-s2[r])*invB
__global__ void rf_loss_kernel(const float* s1, const float* s2, int R, float invB, float* loss, float* diff) {
    __shared__ float red[4];
    float v[1] = {0.f};
    for (int r = threadIdx.x; r < R; r += 256) {
        const float d = (s1[r] - s2[r]) * invB;
        diff[r] = d;
        v[0] += d * d;
    }
    block_sum<1>(v, red);
    if (threadIdx.x == 0) loss[0] = v[0];
}
CPG_EXPORT int cpg_rf_loss(const float* sums1, const float* sums2, int R, int B_global, float* loss, float* diff,
                           void* stream) {
    CPG_CHECK_ARG(sums1 && sums2 && loss && diff && R > 0 && B_global > 0);
    hipLaunchKernelGGL(rf_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums1, sums2, R, 1.f / (float)B_global, loss,
                       diff);
    CPG_LAUNCH_CHECK();
    return 0;
}
// the tail of the random-feature term in ONE launch (single rank): s1 / s2 = sums over the chunks of cpg_rf_features' partials (chunk
// order), diff = (s1 - s2) / B, loss = sum diff^2  (rf_colsum_final_kernel x 2 + rf_loss_kernel; one block)
__global__ void rf_sums_loss_kernel(const float* __restrict__ part, int chunks, int R, float invB, float* __restrict__ s1, float* __restrict__ s2,
                                    float* __restrict__ loss, float* __restrict__ diff) {
    __shared__ float red[4];
    float v[1] = {0.f};
    const float* p1 = part;
    const float* p2 = part + (size_t)chunks * R;
    for (int r = threadIdx.x; r < R; r += 256) {
        float a = 0.f, b = 0.f;
        for (int c = 0; c < chunks; ++c) {
            a += p1[(size_t)c * R + r];
            b += p2[(size_t)c * R + r];
        }
        s1[r] = a;
        s2[r] = b;
        if (loss) {
            const float d = (a - b) * invB;
            diff[r] = d;
            v[0] += d * d;
        }
    }
    if (loss) {
        block_sum<1>(v, red);
        if (threadIdx.x == 0) loss[0] = v[0];
    }
}
// loss / diff null: only the two sum vectors (a data-parallel caller all-reduces them, then calls cpg_rf_loss)
CPG_EXPORT int cpg_rf_sums_loss(const float* part, int chunks, int R, int B_global, float* sums1, float* sums2, float* loss, float* diff,
                                void* stream) {
    CPG_CHECK_ARG(part && sums1 && sums2 && chunks > 0 && R > 0 && B_global > 0 && ((loss == nullptr) == (diff == nullptr)));
    hipLaunchKernelGGL(rf_sums_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, chunks, R, 1.f / (float)B_global, sums1, sums2,
                       loss, diff);
    CPG_LAUNCH_CHECK();
    return 0;
}
// dpre[b,r] = gout * (2*diff[r]/B_global) * (-sin(raw/sigma + b) * sqrt(2/R)) / sigma    (then dz1 = dpre @ rf_w^T)
__global__ void rf_bwd_kernel(const float* raw, const float* rf_b, const float* diff, const float* gout, int Bn, int R,
                              float inv_sigma, float amp, float invB, float* dpre) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Bn * R) return;
    const int r = i % R;
    dpre[i] = gout[0] * 2.f * diff[r] * invB * (-sinf(raw[i] * inv_sigma + rf_b[r]) * amp) * inv_sigma;
}
CPG_EXPORT int cpg_rf_bwd(const float* raw, const float* rf_b, const float* diff, const float* gout, int Bn, int R,
                          float sigma, int B_global, float* dpre, void* stream) {
    CPG_CHECK_ARG(raw && rf_b && diff && gout && dpre && Bn > 0 && R > 0 && sigma > 0.f && B_global > 0);
    const size_t n = (size_t)Bn * R;
    hipLaunchKernelGGL(rf_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, rf_b, diff,
                       gout, Bn, R, 1.f / sigma, sqrtf(2.f / (float)R), 1.f / (float)B_global, dpre);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ MMD, full Gaussian kernel
// losses.mmd_full_kernel (losses.py:47-56,96-108), Gram form: |x-y|^2 = |x|^2 + |y|^2 - 2 x.y with the three Gram matrices
// from the MFMA engine; squared norms are read off the Gram diagonals so K11_ii = K22_ii = 1 exactly, as in the reference.
// Quirk kept: `H - torch.diag(H)` broadcasts the diagonal VECTOR over rows, so
//     loss = (sum_ij H_ij - N * sum_j H_jj) / (N (N-1)),  H = K11 + K22 - 2 K12,  K = exp(-d/sigma^2).
// Optional outputs for the backward pass: P = 2*coef.*W11, Q = -2*coef.*W12 with coef_ij = (1 - N[i==j])/(N(N-1)) and
// W = -dK/dd up to the constant the backward applies (Gaussian: W = K, constant 1/sigma^2; the others: constant 1).
// KIND (compute_mmd_kernel, losses.py:102-107): 0 gaussian exp(-d/s^2), 1 laplace exp(-sqrt(d+s^2)), 2 energy (d+s^2)^-1/4.
template <int KIND>
__device__ __forceinline__ void mmd_kern(float d, float inv_s2, float s2, float& k, float& w) {
    if (KIND == 0) {
        k = w = expf(-d * inv_s2);
    } else if (KIND == 1) {
        const float r = sqrtf(d + s2);
        k = expf(-r);
        w = k / (2.f * r);
    } else {
        const float u = d + s2, q = rsqrtf(sqrtf(u));  // u^-1/4
        k = q;
        w = 0.25f * q / u;
    }
}
// ---- fused form: the three Gram products as ONE launch on the f32 MFMA tile engine (blockIdx.z: K11, K22, K12) whose epilogue
// applies the kernel and reduces the workgroup's tile - the Gram matrices are never written (they were 48 MB of round trip
// at N = 2048 plus a 77 us element-wise pass), K11 / K22 tiles below the diagonal are skipped (their mirror images count twice).
// Squared norms come from a row-norm pre-pass; d_ii is exactly 0 as in the reference's broadcast form.
#ifndef CPG_MMD_SPLIT
#define CPG_MMD_SPLIT 0   // 0: exact-f32 MFMA; 7: six bf16 MFMAs on 3-way split operands (f32-grade)
#endif
using MmdTile = TileCfg<128, 64, 32, 2, 2, 1>;
struct MmdArgs {
    const float* z1;
    const float* z2;
    const float* n1;   // [N] squared row norms
    const float* n2;
    float* part;       // [3][tiles][2] per-workgroup (sum K, sum diag K), zero for skipped tiles
    float* P;          // [N,N] or null
    float* Q;
    int N, D, pairs;
    float inv_s2, s2;
};
__global__ void rownorm2_kernel(const float* z1, const float* z2, int N, int D, float* n1, float* n2) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= 2 * N) return;
    const float* r = (wave < N ? z1 : z2) + (size_t)(wave % N) * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += r[k] * r[k];
    s = wave_sum(s);
    if (lane == 0) (wave < N ? n1 : n2)[wave % N] = s;
}
template <bool VEC, int KIND>
__global__ __launch_bounds__(256) void mmd_gram_fused_kernel(MmdArgs g) {
    using TC = MmdTile;
    using Loop = MainLoop<TC, true, true, VEC, VEC, false, CPG_MMD_SPLIT>;
    __shared__ float red[8];
    const int z = blockIdx.z, N = g.N;
    const int m0 = blockIdx.y * TC::BM, n0 = blockIdx.x * TC::BN;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x, tiles = gridDim.x * gridDim.y;
    const bool sym = z < 2;                           // K11, K22: symmetric
    const bool need_all = (z == 0 && g.P != nullptr); // P wants every element of K11
    float v[2] = {0.f, 0.f};
    if (!(sym && !need_all && n0 + TC::BN - 1 < m0)) {   // workgroup-uniform
        const float* A = z == 1 ? g.z2 : g.z1;
        const float* Bm = z == 0 ? g.z1 : g.z2;
        const float* na = z == 1 ? g.n2 : g.n1;
        const float* nb = z == 0 ? g.n1 : g.n2;
        f32x4 acc[TC::MI][TC::NI];
#pragma unroll
        for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        OpA a{A, g.D, m0, N, nullptr, 1.f, g.pairs};
        OpB b{Bm, g.D, n0, N, 0, nullptr, 1.f, g.pairs};
        Loop::run(a, b, g.D, acc);
        const float cf = 1.f / ((float)N * (float)(N - 1));
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) {
            const int j = n0 + acc_col<TC>(ni);
            if (j >= N) continue;
            const float bj = nb[j];
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = m0 + acc_row<TC>(mi, r);
                    if (i >= N) continue;
                    const bool diag = sym && i == j;
                    float k, w;
                    mmd_kern<KIND>(diag ? 0.f : fmaxf(na[i] + bj - 2.f * acc[mi][ni][r], 0.f), g.inv_s2, g.s2, k, w);
                    const float wt = (sym && !need_all) ? (j > i ? 2.f : (j == i ? 1.f : 0.f)) : 1.f;
                    v[0] += wt * k;
                    if (i == j) v[1] += k;
                    if (z != 1 && g.P) {
                        const float coef = (i == j) ? cf * (1.f - (float)N) : cf;
                        if (z == 0) g.P[(size_t)i * N + j] = 2.f * coef * w;
                        else g.Q[(size_t)i * N + j] = -2.f * coef * w;
                    }
                }
        }
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        g.part[((size_t)z * tiles + tile) * 2] = v[0];
        g.part[((size_t)z * tiles + tile) * 2 + 1] = v[1];
    }
}
// ---- the same launch on the direct-to-LDS loop (gemm_core.h DlLoop, 64x64 tiles): needs K-contiguous operands whose row length
// is a multiple of 32, so the row-norm pre-pass also writes zero-padded copies zp [2N, Dp] (8 MB at 2048 x 510, L2-resident);
// the padding adds exact zeros to every dot product.  N % 64 == 0.
// CPG_MMD_PAIR = 1 (round 4): the Gram products on f16 pairs (gemm_core.h, DlLoop PREC 3) - the padded operand copy that
// rownorm2_pad_kernel writes anyway holds [32 hi | 32 lo] f16 per 32 k instead of 32 floats (same bytes), three f16 MFMAs per block and
// slab in place of eight f32 ones.  z and the prior samples are O(1): no scale.  0: exact-f32 MFMA on the f32 copy.
#ifndef CPG_MMD_PAIR
#define CPG_MMD_PAIR 1
#endif
using MmdDl = DlLoop<64, 64, 3, CPG_MMD_PAIR ? 3 : 0>;
__global__ void rownorm2_pad_kernel(const float* z1, const float* z2, int N, int D, int Dp, float* n1, float* n2, float* zp) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= 2 * N) return;
    const float* r = (wave < N ? z1 : z2) + (size_t)(wave % N) * D;
    float* o = zp + (size_t)wave * Dp;
    float s = 0.f;
#if CPG_MMD_PAIR
    uint32_t* const op = reinterpret_cast<uint32_t*>(o);   // per 32 k: 16 words of high halves, 16 words of low halves
    for (int kp = lane; kp < Dp / 2; kp += 64) {
        const int k = 2 * kp;
        const float x0 = k < D ? r[k] : 0.f, x1 = k + 1 < D ? r[k + 1] : 0.f;
        uint32_t hi, lo;
        split2h_pair(x0, x1, hi, lo);
        op[(kp >> 4) * 32 + (kp & 15)] = hi;
        op[(kp >> 4) * 32 + 16 + (kp & 15)] = lo;
        s += x0 * x0 + x1 * x1;
    }
#else
    for (int k = lane; k < Dp; k += 64) {
        const float x = k < D ? r[k] : 0.f;
        o[k] = x;
        s += x * x;
    }
#endif
    s = wave_sum(s);
    if (lane == 0) (wave < N ? n1 : n2)[wave % N] = s;
}
template <int KIND>
__global__ __launch_bounds__(256) void mmd_gram_dl_kernel(MmdArgs g, const float* zp, int Dp) {
    extern __shared__ float dl_smem[];
    __shared__ float red[8];
    const int z = blockIdx.z, N = g.N;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x, tiles = gridDim.x * gridDim.y;
    const bool sym = z < 2;
    const bool need_all = (z == 0 && g.P != nullptr);
    float v[2] = {0.f, 0.f};
    if (!(sym && !need_all && n0 + 63 < m0)) {   // workgroup-uniform
        const float* A = zp + (size_t)(z == 1 ? N : 0) * Dp;
        const float* Bm = zp + (size_t)(z == 0 ? 0 : N) * Dp;
        const float* na = z == 1 ? g.n2 : g.n1;
        const float* nb = z == 0 ? g.n1 : g.n2;
        f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#if CPG_MMD_PAIR
        MmdDl::run(reinterpret_cast<const uint16_t*>(A + (size_t)m0 * Dp), (size_t)2 * Dp, reinterpret_cast<const uint16_t*>(Bm + (size_t)n0 * Dp),
                   (size_t)2 * Dp, 2 * Dp, dl_smem, acc, -1, [] {}, [](int) { return true; });
#else
        MmdDl::run(A + (size_t)m0 * Dp, (size_t)Dp, Bm + (size_t)n0 * Dp, (size_t)Dp, Dp, dl_smem, acc, -1, [] {});
#endif
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
        const float cf = 1.f / ((float)N * (float)(N - 1));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int j = n0 + (wave & 1) * 32 + ni * 16 + l15;
            const float bj = nb[j];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = m0 + (wave >> 1) * 32 + mi * 16 + 4 * lq + r;
                    const bool diag = sym && i == j;
                    float k, w;
                    mmd_kern<KIND>(diag ? 0.f : fmaxf(na[i] + bj - 2.f * acc[mi][ni][r], 0.f), g.inv_s2, g.s2, k, w);
                    const float wt = (sym && !need_all) ? (j > i ? 2.f : (j == i ? 1.f : 0.f)) : 1.f;
                    v[0] += wt * k;
                    if (i == j) v[1] += k;
                    if (z != 1 && g.P) {
                        const float coef = (i == j) ? cf * (1.f - (float)N) : cf;
                        if (z == 0) g.P[(size_t)i * N + j] = 2.f * coef * w;
                        else g.Q[(size_t)i * N + j] = -2.f * coef * w;
                    }
                }
        }
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        g.part[((size_t)z * tiles + tile) * 2] = v[0];
        g.part[((size_t)z * tiles + tile) * 2 + 1] = v[1];
    }
}
static bool mmd_dl_ok(int N) {
    const CpgOptVal& o = cpg_opt(OPT_MMD_DL);
    const bool off = o.set && o.i == 0;
    return !off && N % 64 == 0;
}
static int mmd_dp(int D) { return (D + 31) / 32 * 32; }
// out[0] = (sum H - N tr H) / (N (N-1)),  H = K11 + K22 - 2 K12;  out[1] = sum H, out[2] = tr H.  One block, fixed order.
__global__ void mmd_fused_final_kernel(const float* part, int tiles, int N, float* out) {
    __shared__ float red[24];
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < tiles; i += 256)
#pragma unroll
        for (int z = 0; z < 3; ++z) {
            v[2 * z] += part[((size_t)z * tiles + i) * 2];
            v[2 * z + 1] += part[((size_t)z * tiles + i) * 2 + 1];
        }
    block_sum<6>(v, red);
    if (threadIdx.x == 0) {
        const float sum = (v[0] + v[2]) - 2.f * v[4], tr = (v[1] + v[3]) - 2.f * v[5];
        out[0] = (sum - (float)N * tr) / ((float)N * (float)(N - 1));
        out[1] = sum;
        out[2] = tr;
    }
}
static size_t mmd_tiles(int N) { return (size_t)cdiv(N, MmdTile::BM) * cdiv(N, MmdTile::BN); }
static size_t mmd_tiles_max(int N) { return (size_t)cdiv(N, 64) * cdiv(N, 64); }
// [n1 | n2 | per-tile partials | zero-padded operand copies for the direct-to-LDS form]
CPG_EXPORT size_t cpg_mmd_full_workspace(int N, int D) {
    return ((size_t)2 * N + 6 * mmd_tiles_max(N) + (size_t)2 * N * mmd_dp(D)) * sizeof(float) + 256;
}

template <bool VEC, int KIND>
static void mmd_launch(const MmdArgs& g, hipStream_t s) {
    const size_t smem = MainLoop<MmdTile, true, true, VEC, VEC, false, CPG_MMD_SPLIT>::smem_bytes();
    hipLaunchKernelGGL((mmd_gram_fused_kernel<VEC, KIND>), dim3(cdiv(g.N, MmdTile::BN), cdiv(g.N, MmdTile::BM), 3), dim3(256), smem, s, g);
}

// z1,z2 [N,D].  out[0] = loss.  P,Q [N,N] optional (null when no gradient is needed).  workspace: cpg_mmd_full_workspace(N).
// kernel: 0 gaussian, 1 laplace, 2 energy (cfg.losses.wae_mmd.kernel, cfg.py:250).
CPG_EXPORT int cpg_mmd_full_fwd(const float* z1, const float* z2, int N, int D, float sigma, int kernel, float* out, float* P,
                                float* Q, void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(z1 && z2 && out && workspace && N > 1 && D > 0 && sigma > 0.f && ((P == nullptr) == (Q == nullptr)));
    CPG_CHECK_ARG(kernel >= 0 && kernel <= 2);
    CPG_CHECK_ARG(workspace_bytes >= cpg_mmd_full_workspace(N, D));
    hipStream_t s = (hipStream_t)stream;
    MmdArgs g;
    g.z1 = z1; g.z2 = z2; g.N = N; g.D = D; g.P = P; g.Q = Q;
    float* n1 = (float*)workspace;
    g.n1 = n1; g.n2 = n1 + N; g.part = n1 + 2 * (size_t)N;
    g.s2 = sigma * sigma;
    g.inv_s2 = 1.f / g.s2;
    if (mmd_dl_ok(N)) {
        const int Dp = mmd_dp(D);
        float* zp = g.part + 6 * mmd_tiles_max(N);
        zp += (64 - (((uintptr_t)zp >> 2) & 63)) & 63;   // 256-byte aligned rows
        g.pairs = 0;
        hipLaunchKernelGGL(rownorm2_pad_kernel, dim3(cdiv(2 * N, 4)), dim3(256), 0, s, z1, z2, N, D, Dp, n1, n1 + N, zp);
        const size_t smem = MmdDl::smem_floats() * sizeof(float);
        const dim3 grid(N / 64, N / 64, 3);
        if (kernel == 0) hipLaunchKernelGGL(mmd_gram_dl_kernel<0>, grid, dim3(256), smem, s, g, (const float*)zp, Dp);
        else if (kernel == 1) hipLaunchKernelGGL(mmd_gram_dl_kernel<1>, grid, dim3(256), smem, s, g, (const float*)zp, Dp);
        else hipLaunchKernelGGL(mmd_gram_dl_kernel<2>, grid, dim3(256), smem, s, g, (const float*)zp, Dp);
        hipLaunchKernelGGL(mmd_fused_final_kernel, dim3(1), dim3(256), 0, s, (const float*)g.part, (int)mmd_tiles_max(N), N, out);
        CPG_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(rownorm2_kernel, dim3(cdiv(2 * N, 4)), dim3(256), 0, s, z1, z2, N, D, n1, n1 + N);
    const bool vec = D % 4 == 0 && aligned16(z1) && aligned16(z2);
    g.pairs = D % 2 == 0 && (((uintptr_t)z1) & 7) == 0 && (((uintptr_t)z2) & 7) == 0;
#define CPG_MMD_LAUNCH(KIND)                 \
    if (vec) mmd_launch<true, KIND>(g, s);   \
    else mmd_launch<false, KIND>(g, s)
    if (kernel == 0) { CPG_MMD_LAUNCH(0); }
    else if (kernel == 1) { CPG_MMD_LAUNCH(1); }
    else { CPG_MMD_LAUNCH(2); }
#undef CPG_MMD_LAUNCH
    hipLaunchKernelGGL(mmd_fused_final_kernel, dim3(1), dim3(256), 0, s, (const float*)g.part, (int)mmd_tiles(N), N, out);
    CPG_LAUNCH_CHECK();
    return 0;
}

// dz1 = gout * c * ( (rowsum(P)+rowsum(Q)) .* z1 - P z1 - Q z2 ),  c = -2/sigma^2 (gaussian) or -2 (laplace, energy)
__global__ void rowsum2_kernel(const float* P, const float* Q, int N, float* rs) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N) return;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += P[(size_t)wave * N + j] + Q[(size_t)wave * N + j];
    s = wave_sum(s);
    if (lane == 0) rs[wave] = s;
}
__global__ void mmd_full_bwd_combine_kernel(const float* z1, const float* rs, const float* Mz, const float* gout, int N, int D,
                                            float c, float* dz1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * D) return;
    const int r = i / D;
    dz1[i] = gout[0] * c * (rs[r] * z1[i] - Mz[i]);
}
CPG_EXPORT int cpg_mmd_full_bwd(const float* z1, const float* z2, const float* P, const float* Q, const float* gout, int N,
                                int D, float sigma, int kernel, float* dz1, void* workspace, size_t workspace_bytes,
                                void* stream) {
    CPG_CHECK_ARG(z1 && z2 && P && Q && gout && dz1 && workspace && N > 1 && D > 0 && kernel >= 0 && kernel <= 2);
    CPG_CHECK_ARG(workspace_bytes >= ((size_t)N * D + N) * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* Mz = (float*)workspace;
    float* rs = Mz + (size_t)N * D;
    int rc = cpg_gemm_nn(P, N, z1, D, Mz, D, N, D, N, 0, nullptr, 1.f, s);
    if (!rc) rc = cpg_gemm_nn(Q, N, z2, D, Mz, D, N, D, N, 1, nullptr, 1.f, s);
    if (rc) return rc;
    hipLaunchKernelGGL(rowsum2_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, P, Q, N, rs);
    const size_t n = (size_t)N * D;
    hipLaunchKernelGGL(mmd_full_bwd_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z1, (const float*)rs,
                       (const float*)Mz, gout, N, D, kernel == 0 ? -2.f / (sigma * sigma) : -2.f, dz1);
    CPG_LAUNCH_CHECK();
    return 0;
}
