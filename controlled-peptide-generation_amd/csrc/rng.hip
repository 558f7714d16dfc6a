// Counter-based random streams for the training step's stochastic inputs when the caller does not inject them:
//   eps / z_prior ~ N(0,1)      (models/model.py:111,118 ; losses.py:37)
//   word-dropout / out-dropout Bernoulli masks  (models/decoder.py:117-133 ; nn.Dropout at decoder.py:44)
//   uniforms for the CLaSS accept test          (density_modeling.py:58)
// Philox4x32-10 keyed by (seed), counter = (offset + element/4, stream id): reproducible for a given (seed, offset)
// independent of launch geometry.  `base` (optional device uint64): added to `offset` ON THE DEVICE - a training step whose
// launches are replayed from a captured hipGraph keeps its host-side offsets (relative to the step) and advances *base once per
// step (cpg_counter_add_u64), so every replay draws fresh numbers.  These are NEW streams (the reference mixes torch and numpy generators); parity tests
// inject the reference's captured draws instead.
#include "cpg_internal.h"
#include "rng_core.h"

__global__ void rng_normal_kernel(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // 4 outputs per thread
    if (q * 4 >= n) return;
    float v[4];
    philox_normal4(seed, offset + (base ? *base : 0) + q, v);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) out[q * 4 + k] = v[k];
}

__global__ void rng_uniform_kernel(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    uint32_t r[4];
    philox4x32(seed, offset + (base ? *base : 0) + q, 1u, r);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) out[q * 4 + k] = u01(r[k]);
}

__global__ void rng_uniform_f64_kernel(double* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // 2 outputs per thread
    if (q * 2 >= n) return;
    uint32_t r[4];
    philox4x32(seed, offset + (base ? *base : 0) + q, 3u, r);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (q * 2 + k < n) {
            const uint64_t bits = ((uint64_t)r[2 * k] << 21) ^ (uint64_t)(r[2 * k + 1] >> 11);  // 53 random bits
            out[q * 2 + k] = (double)(bits & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
        }
}

// out[i] = 1 with probability p_one
__global__ void rng_bernoulli_kernel(uint8_t* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    uint32_t r[4];
    philox4x32(seed, offset + (base ? *base : 0) + q, 2u, r);
    if (q * 4 + 3 < n && ((((uintptr_t)out) & 3) == 0)) {   // the four flags as one 32-bit store
        const uint32_t w = (u01(r[0]) < p_one ? 1u : 0u) | (u01(r[1]) < p_one ? 0x100u : 0u) | (u01(r[2]) < p_one ? 0x10000u : 0u) |
                           (u01(r[3]) < p_one ? 0x1000000u : 0u);
        reinterpret_cast<uint32_t*>(out)[q] = w;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) out[q * 4 + k] = u01(r[k]) < p_one ? 1 : 0;
}

// out[i][:] = one-hot of the SAME draw rng_bernoulli_kernel makes for element i: c ~ Cat([1 - p_one, p_one]) as float rows
// (RNN_VAE.sample_c_prior, models/model.py:122-126, in one launch instead of draw + cast + fill + scatter)
__global__ void rng_onehot2_kernel(float* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    uint32_t r[4];
    philox4x32(seed, offset + (base ? *base : 0) + q, 2u, r);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) {
            const bool one = u01(r[k]) < p_one;
            *reinterpret_cast<float2*>(out + 2 * (q * 4 + k)) = make_float2(one ? 0.f : 1.f, one ? 1.f : 0.f);
        }
}

#define RNG_LAUNCH(kern, per, ...)                                                                                  \
    do {                                                                                                            \
        const size_t nq = (n + (per) - 1) / (per);                                                                  \
        hipLaunchKernelGGL(kern, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        CPG_LAUNCH_CHECK();                                                                                         \
    } while (0)

CPG_EXPORT int cpg_rng_normal(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream) {
    CPG_CHECK_ARG(out && n > 0);
    RNG_LAUNCH(rng_normal_kernel, 4, out, n, seed, offset, base);
    return 0;
}
CPG_EXPORT int cpg_rng_uniform(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream) {
    CPG_CHECK_ARG(out && n > 0);
    RNG_LAUNCH(rng_uniform_kernel, 4, out, n, seed, offset, base);
    return 0;
}
CPG_EXPORT int cpg_rng_uniform_f64(double* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream) {
    CPG_CHECK_ARG(out && n > 0);
    RNG_LAUNCH(rng_uniform_f64_kernel, 2, out, n, seed, offset, base);
    return 0;
}
CPG_EXPORT int cpg_rng_bernoulli_u8(uint8_t* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base,
                                    void* stream) {
    CPG_CHECK_ARG(out && n > 0 && p_one >= 0.f && p_one <= 1.f);
    RNG_LAUNCH(rng_bernoulli_kernel, 4, out, n, p_one, seed, offset, base);
    return 0;
}

CPG_EXPORT int cpg_rng_onehot2(float* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream) {
    CPG_CHECK_ARG(out && n > 0 && p_one >= 0.f && p_one <= 1.f);
    RNG_LAUNCH(rng_onehot2_kernel, 4, out, n, p_one, seed, offset, base);
    return 0;
}

// Device-side step counters (one thread): *p += by.  Part of a captured training step: the Philox base and the optimiser's
// iteration count advance on the device, so a hipGraph replay is a NEW step.
__global__ void counter_add_u64_kernel(uint64_t* p, uint64_t by) { p[0] += by; }
__global__ void counter_add_i32_kernel(int32_t* p, int32_t by) { p[0] += by; }
CPG_EXPORT int cpg_counter_add_u64(uint64_t* p, uint64_t by, void* stream) {
    CPG_CHECK_ARG(p);
    hipLaunchKernelGGL(counter_add_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, by);
    CPG_LAUNCH_CHECK();
    return 0;
}
CPG_EXPORT int cpg_counter_add_i32(int32_t* p, int32_t by, void* stream) {
    CPG_CHECK_ARG(p);
    hipLaunchKernelGGL(counter_add_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, by);
    CPG_LAUNCH_CHECK();
    return 0;
}
