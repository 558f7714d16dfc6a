// Library-wide plumbing of libcpg_hip.so: version, thread-local error text, the option table, small layout kernels.
#include <string.h>
#include <stdarg.h>
#include <stdlib.h>
#include <mutex>
#include "cpg_internal.h"

static thread_local char g_err[512] = "";

void cpg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

CPG_EXPORT const char* cpg_last_error(void) { return g_err; }
CPG_EXPORT int cpg_version(void) { return CPG_ABI_VERSION; }

// ---- option table: the launch policy's tuning knobs, ONE process-wide struct.  Filled once, on first use, from the
// environment (CPG_<NAME IN CAPITALS>), afterwards changed only through cpg_set_option: no launch path calls getenv.
static const char* const g_opt_names[OPT__COUNT] = {
    "gru_persist", "lstm_persist", "f32_engine", "lstm_persist_groups", "gru_fwd_bm", "gru_bwd_dl", "gru_bwd_tile", "gru_bwd_dl2", "gru_bwd_stagger", "gru_bwd_engine", "lstm_bwd_dl",
    "tn_tile", "tn_split", "gemm_tile", "dgi_mode", "mmd_dl", "bf16_store", "bf16_dg", "gru_ap", "gru_small_seq", "small_seq_rows"};
static CpgOptVal g_opts[OPT__COUNT];
static std::once_flag g_opts_once;
static std::mutex g_opts_mu;   // cpg_set_option may run on one thread while launch paths on another (autograd's backward thread) read

static void opt_assign(CpgOptVal& v, const char* text) {
    v.set = text && text[0];
    v.i = v.set ? atol(text) : 0;
    snprintf(v.s, sizeof(v.s), "%s", v.set ? text : "");
}

static void opts_from_env() {
    for (int o = 0; o < OPT__COUNT; ++o) {
        char key[48] = "CPG_";
        size_t k = 4;
        for (const char* c = g_opt_names[o]; *c && k + 1 < sizeof(key); ++c) key[k++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        key[k] = 0;
        opt_assign(g_opts[o], getenv(key));
    }
}

// Returns a COPY taken under the table's mutex: a reader never sees a half-written entry (flag, number and text of one
// cpg_set_option call arrive together).  Options must still not be CHANGED between the forward and the backward pass of a
// sequence where they select a storage format (bf16_store): the Python binding checks the saved gates' dtype for that.
CpgOptVal cpg_opt(CpgOpt o) {
    std::call_once(g_opts_once, opts_from_env);
    std::lock_guard<std::mutex> lk(g_opts_mu);
    return g_opts[o];
}

static int opt_index(const char* name) {
    for (int o = 0; name && o < OPT__COUNT; ++o)
        if (!strcmp(name, g_opt_names[o])) return o;
    return -1;
}

// value: decimal number or short token ("64x32"); null / "" = back to the built-in policy
CPG_EXPORT int cpg_set_option(const char* name, const char* value) {
    const int o = opt_index(name);
    if (o < 0) {
        cpg_set_error("cpg_set_option: unknown option '%s'", name ? name : "(null)");
        return -2;
    }
    std::call_once(g_opts_once, opts_from_env);
    CpgOptVal v;
    opt_assign(v, value);
    std::lock_guard<std::mutex> lk(g_opts_mu);
    g_opts[o] = v;
    return 0;
}

// Copies the option's current text ("" when unset) into buf; returns 1 when set, 0 when unset, -2 for an unknown name.
CPG_EXPORT int cpg_get_option(const char* name, char* buf, int n) {
    const int o = opt_index(name);
    if (o < 0) return -2;
    const CpgOptVal v = cpg_opt((CpgOpt)o);
    if (buf && n > 0) snprintf(buf, (size_t)n, "%s", v.s);
    return v.set ? 1 : 0;
}

// Per-device facts the launchers need (CU count), cached per device index.
int cpg_device_cus() {
    static int cus[64];
    static std::once_flag once[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::call_once(once[dev], [dev] {
        hipDeviceProp_t pr;
        cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 0;
    });
    return cus[dev];
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); the return code is the caller's to check.
int cpg_allow_big_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static struct { const void* k; int dev; int bytes; } seen[512];
    static int nseen = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < nseen; ++i)
        if (seen[i].k == kernel && seen[i].dev == dev && seen[i].bytes >= bytes) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        cpg_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
        return (int)e;
    }
    if (nseen < 512) seen[nseen++] = {kernel, dev, bytes};
    return 0;
}

// Compute mode of the recurrent products (forward / backward step products and dW_hh = dG^T h): 0 = f32-grade (default:
// exact-f32 MFMA or six bf16 MFMAs on 3-way split operands), 1 = bf16 (operands rounded to bf16 when staged, ONE bf16 MFMA
// per block, f32 accumulation; BASELINE.json configs[1]/[4] "bf16").  Process-wide like the option table, read at launch time.
static int g_compute_mode = 0;
int cpg_compute_mode_get() { return g_compute_mode; }
int cpg_persist_planes() {
    if (g_compute_mode == 1) return 1;
    const CpgOptVal o = cpg_opt(OPT_F32_ENGINE);
    return (o.set && strcmp(o.s, "bf16x3") == 0) ? 3 : 2;
}
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
