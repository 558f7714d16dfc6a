// Library-wide plumbing of libcpg_hip.so: version, thread-local error text, small layout kernels.
#include <stdarg.h>
#include "cpg_common.h"

static thread_local char g_err[512] = "";

void cpg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

CPG_EXPORT const char* cpg_last_error(void) { return g_err; }
CPG_EXPORT int cpg_version(void) { return 100; }  // 0.1.0

// Compute mode of the recurrent products (forward / backward step products and dW_hh = dG^T h): 0 = f32-grade (default:
// exact-f32 MFMA or six bf16 MFMAs on 3-way split operands), 1 = bf16 (operands rounded to bf16 when staged, ONE bf16 MFMA
// per block, f32 accumulation; BASELINE.json configs[1]/[4] "bf16").  Process-wide, read at launch time.
static int g_compute_mode = 0;
int cpg_compute_mode_get() { return g_compute_mode; }
CPG_EXPORT int cpg_set_compute_mode(int mode) {
    if (mode != 0 && mode != 1) {
        cpg_set_error("cpg_set_compute_mode: mode %d (0 = f32-grade, 1 = bf16 recurrent products)", mode);
        return -2;
    }
    g_compute_mode = mode;
    return 0;
}
CPG_EXPORT int cpg_get_compute_mode(void) { return g_compute_mode; }

// Number of visible gfx950 devices (0 on a host without a GPU).  The Python host refuses to run without one.
CPG_EXPORT int cpg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ids int64 [B,T] (batch-major, as the reference's loader hands them over, data_processing/dataset.py:242-244)
//   -> tok int32 [T,B] time-major, with WordDropout applied when a mask is given (models/decoder.py:117-133:
//      masked positions become <unk>, no exemption for <start>/<pad>).
__global__ void tokens_prepare_kernel(const int64_t* ids, const uint8_t* wd_mask, int B, int T, int unk, int32_t* tok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int t = i / B, b = i % B;
    const size_t src = (size_t)b * T + t;
    int v = (int)ids[src];
    if (wd_mask && wd_mask[src]) v = unk;
    tok[i] = v;
}

CPG_EXPORT int cpg_tokens_prepare(const int64_t* ids, const uint8_t* wd_mask, int B, int T, int unk, int32_t* tok,
                                  void* stream) {
    CPG_CHECK_ARG(ids && tok && B > 0 && T > 0);
    hipLaunchKernelGGL(tokens_prepare_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, (hipStream_t)stream, ids, wd_mask, B, T,
                       unk, tok);
    CPG_LAUNCH_CHECK();
    return 0;
}

// dst[d1][d0][inner] = src[d0][d1][inner]
template <typename Tp>
__global__ void transpose01_kernel(const Tp* src, int d0, int d1, int inner, Tp* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)d0 * d1 * inner;
    if (i >= n) return;
    const int k = i % inner;
    const size_t r = i / inner;
    const int a = r % d0, b = r / d0;  // dst index (b, a, k)
    dst[i] = src[((size_t)a * d1 + b) * inner + k];
}

CPG_EXPORT int cpg_transpose01_f32(const float* src, int d0, int d1, int inner, float* dst, void* stream) {
    CPG_CHECK_ARG(src && dst && d0 > 0 && d1 > 0 && inner > 0);
    const size_t n = (size_t)d0 * d1 * inner;
    hipLaunchKernelGGL(transpose01_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       d0, d1, inner, dst);
    CPG_LAUNCH_CHECK();
    return 0;
}

CPG_EXPORT int cpg_transpose01_u8(const uint8_t* src, int d0, int d1, int inner, uint8_t* dst, void* stream) {
    CPG_CHECK_ARG(src && dst && d0 > 0 && d1 > 0 && inner > 0);
    const size_t n = (size_t)d0 * d1 * inner;
    hipLaunchKernelGGL(transpose01_kernel<uint8_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       d0, d1, inner, dst);
    CPG_LAUNCH_CHECK();
    return 0;
}
