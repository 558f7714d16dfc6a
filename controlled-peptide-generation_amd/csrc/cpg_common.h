// Shared helpers for the gfx950 kernels of libcpg_hip.so (CDNA4: wave64, f32 MFMA, 160 KiB LDS/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CPG_EXPORT extern "C" __attribute__((visibility("default")))

// Ablation macros compile phases of a kernel out (results WRONG by construction: "where does the time go" builds).  They are
// honoured in diagnostic builds only (tools/variant_build.sh passes -DCPG_DIAG); the product library cannot be built with them.
#if !defined(CPG_DIAG) && (defined(CPG_ABLATE) || defined(CPG_DL_ABLATE) || defined(CPG_PERSIST_ABLATE) || defined(CPG_BEAM_ABLATE))
#error "CPG_*_ABLATE needs -DCPG_DIAG: these macros build kernels whose results are wrong by construction"
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing -------------------------------------------------------------------------
void cpg_set_error(const char* fmt, ...);

#define CPG_CHECK_ARG(cond)                                                           \
    do {                                                                              \
        if (!(cond)) {                                                                \
            cpg_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);      \
            return -2;                                                                \
        }                                                                             \
    } while (0)

#define CPG_LAUNCH_CHECK()                                                                      \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) {                                                                \
            cpg_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                                    \
        }                                                                                       \
    } while (0)

#define CPG_HIP(call)                                                                           \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            cpg_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));  \
            return (int)e__;                                                                    \
        }                                                                                       \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- device math (accurate forms: parity with the CPU reference is the first gate) ------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Cell nonlinearities of the persistent sequence kernels (round 6): hardware exp2 / rcp forms, 4 and 16 VALU instructions against the
// ~35 / ~50 of the library forms (a wave64 VALU instruction occupies its SIMD for 4 cycles; the cell was 2.1 of the 14.6 us of a
// step and ran against the partner wave's MFMAs).  Accuracy: sigmoid within 2 ulp of 1 in absolute terms (v_exp_f32 and v_rcp_f32 are
// 1 ulp each; the argument product rounds once); tanh within 3 ulp of the result: odd series through a^9 below 0.25 (next term 2e-9),
// (1 - e) / (1 + e) with e = exp(-2|x|) above.  Saturates to 0 / 1 / +-1, NaN stays NaN.
__device__ __forceinline__ float cell_sigmoidf(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float cell_tanhf(float x) {
    const float a = fabsf(x), a2 = a * a;
    const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * a);
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    float p = fmaf(a2, 62.f / 2835.f, -17.f / 315.f);
    p = fmaf(a2, p, 2.f / 15.f);
    p = fmaf(a2, p, -1.f / 3.f);
    const float small = fmaf(a * a2, p, a);
    return copysignf(a < 0.25f ? small : big, x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
