// Does VALU work issued between MFMAs of the SAME wave (1 wave per SIMD) hide behind them?
// f32 MFMA (v_mfma_f32_16x16x4_f32) vs bf16 MFMA (v_mfma_f32_16x16x32_bf16), each alone, the VALU filler alone, and
// interleaved {1 MFMA, NV v_fma_f32}.  Prints cycles per iteration (s_memtime) and the overlap ratio.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int KIND, int NV>  // MODE 1: mfma, 2: valu, 3: both ; KIND 0: f32 16x16x4, 1: bf16 16x16x32
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 1) {
                if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[u], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q % 8]) : "v"(b), "v"(a));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int KIND, int NV>
double run(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((probe<MODE, KIND, NV>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<MODE, KIND, NV>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    return s / 256 / iters / 8;  // s_memtime ticks per {MFMA + NV VALU} group
}

template <int KIND, int NV>
void report(float* out, long long* cyc, const char* name) {
    const int iters = 4000;
    const double m = run<1, KIND, NV>(out, cyc, iters), v = run<2, KIND, NV>(out, cyc, iters), b = run<3, KIND, NV>(out, cyc, iters);
    printf("%-28s NV=%d  mfma %.1f  valu %.1f  both %.1f  ticks/group   (sum %.1f, max %.1f)\n", name, NV, m, v, b, m + v,
           m > v ? m : v);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    report<0, 2>(out, cyc, "f32 16x16x4");
    report<0, 4>(out, cyc, "f32 16x16x4");
    report<0, 6>(out, cyc, "f32 16x16x4");
    report<1, 2>(out, cyc, "bf16 16x16x32");
    report<1, 4>(out, cyc, "bf16 16x16x32");
    report<1, 6>(out, cyc, "bf16 16x16x32");
    return 0;
}
