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
