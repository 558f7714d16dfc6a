// Micro-benchmark: C[R, N] = A[R, K] . B[N, K]^T with both operands as K-contiguous f16-pair plane images ([32 hi | 32 lo] per 32 k),
// DlLoop<BM, BN, NS, 3> (LDS-DMA ring, ds_read_b128 fragments, three f16 MFMAs per block) - the loop of an nn.Linear-shaped product
// on pre-split planes (config C's upper-layer input products).  No exponents, plain f32 stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlled-peptide-generation_amd/csrc tools/micro/pair_nt_bench.hip -o build_variants/pair_nt_bench
#include "gemm_core.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int NS>
__global__ __launch_bounds__(256) void pair_nt_kernel(const uint16_t* A, const uint16_t* B, float* C, int R, int N, int K2 /* 2 K */) {
    using DL = DlLoop<BM, BN, NS, 3>;
    constexpr int MI = DL::MI, NI = DL::NI;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int m0 = by * BM, n0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    DL::run(A + (size_t)m0 * K2, (size_t)K2, B + (size_t)n0 * K2, (size_t)K2, K2, cpg_smem, acc, -1, []() {});
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(size_t)(m0 + wm * (BM / 2) + mi * 16 + 4 * lq + r) * N + n0 + wn * (BN / 2) + ni * 16 + l15] = acc[mi][ni][r];
}

template <int BM, int BN, int NS>
static void run(const uint16_t* A, const uint16_t* B, float* C, int R, int N, int K, const char* tag) {
    using DL = DlLoop<BM, BN, NS, 3>;
    const size_t smem = DL::smem_floats() * 4;
    CK(hipFuncSetAttribute((const void*)pair_nt_kernel<BM, BN, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(N / BN, R / BM, 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pair_nt_kernel<BM, BN, NS>), grid, dim3(256), smem, 0, A, B, C, R, N, 2 * K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 5;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((pair_nt_kernel<BM, BN, NS>), grid, dim3(256), smem, 0, A, B, C, R, N, 2 * K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it, fl = 2.0 * R * N * (double)K;
    printf("%-22s grid %5d smem %6zu  %9.1f us  %6.1f TFLOP/s (%.3f of 833)\n", tag, grid.x * grid.y, smem, us, fl / us * 1e-6, fl / us * 1e-6 / 833.3);
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 51200, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 3072;
    printf("C[%d,%d] = A[%d,%d] B[%d,%d]^T on f16-pair planes\n", R, N, R, K, N, K);
    std::vector<uint16_t> hA((size_t)R * 2 * K), hB((size_t)N * 2 * K);
    srand(2);
    for (auto& x : hA) x = 0x3000 + (rand() & 0x3ff);
    for (auto& x : hB) x = 0x3000 + (rand() & 0x3ff);
    uint16_t *dA, *dB; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)R * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    run<64, 64, 3>(dA, dB, dC, R, N, K, "64x64 NS=3");
    run<128, 64, 3>(dA, dB, dC, R, N, K, "128x64 NS=3");
    run<128, 128, 2>(dA, dB, dC, R, N, K, "128x128 NS=2");
    run<128, 128, 3>(dA, dB, dC, R, N, K, "128x128 NS=3");
    run<128, 128, 4>(dA, dB, dC, R, N, K, "128x128 NS=4");
    return 0;
}
