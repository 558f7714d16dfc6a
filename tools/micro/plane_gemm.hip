// Micro-benchmark: C[M,N] = A[M,K] . B[N,K]^T with BOTH operands handed over as three pre-split bf16 planes ([3][rows][K],
// K-contiguous), six bf16 MFMAs per 16x16x32 block (f32-grade), K-split across blockIdx.z into partial slabs - the main
// loop a backward step on pre-split dgh / W_hh^T planes would run.  No conversion work, no epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlled-peptide-generation_amd/csrc tools/micro/plane_gemm.hip -o build_variants/plane_gemm
#include "gemm_core.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// ABL (diagnostic, wrong results): 1 no global loads after slab 0, 2 no LDS stores after slab 0, 4 no MFMA slab
template <class TC, int ABL>
__global__ __launch_bounds__(256) void plane_gemm_kernel(const uint16_t* Ap, const uint16_t* Bp, float* C, int M, int N, int K, int kchunk) {
    using ML = MainLoop<TC, true, true, true, true, false, 7>;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    uint32_t* const base = reinterpret_cast<uint32_t*>(cpg_smem);
    uint32_t* const A0 = base;
    uint32_t* const A1 = base + ML::ASZ7;
    uint32_t* const B0 = base + 2 * ML::ASZ7;
    uint32_t* const B1 = base + 2 * ML::ASZ7 + ML::BSZ7;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int m0 = by * TC::BM, n0 = bx * TC::BN, kb = bz * kchunk;
    const int tid = threadIdx.x;
    constexpr int ACH = TC::BM * 4 * 3 / 256, BCH = TC::BN * 4 * 3 / 256;   // 16-byte chunks per thread and slab
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[ACH], rb[BCH];
    const uint16_t* ga[ACH];
    const uint16_t* gb[BCH];
    int la[ACH], lb[BCH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + i * 256, pl = c / (TC::BM * 4), r = (c % (TC::BM * 4)) / 4, q = c % 4;
        ga[i] = Ap + ((size_t)pl * M + m0 + r) * K + kb + q * 8;
        la[i] = pl * ML::APL + r * ML::KCW + q * 4;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int c = tid + i * 256, pl = c / (TC::BN * 4), r = (c % (TC::BN * 4)) / 4, q = c % 4;
        gb[i] = Bp + ((size_t)pl * N + n0 + r) * K + kb + q * 8;
        lb[i] = pl * ML::BPL + r * ML::KCW + q * 4;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ga[i] + k0);
#pragma unroll
        for (int i = 0; i < BCH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(gb[i] + k0);
    };
    auto sstore = [&](uint32_t* As, uint32_t* Bs) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) *reinterpret_cast<u32x4*>(As + la[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i) *reinterpret_cast<u32x4*>(Bs + lb[i]) = rb[i];
    };
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    OpA oa{nullptr, 0, 0, 0, nullptr, 1.f};
    OpB ob{nullptr, 0, 0, 0, 0, nullptr, 1.f};
    typename ML::Stage st;
    const int KT = kchunk / 32;
    gload(0);
    sstore(A0, B0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        if (kt + 1 < KT && !(ABL & 1)) gload((kt + 1) * 32);
        if (!(ABL & 4)) ML::template slab7<false>(oa, ob, A0, B0, A1, B1, st, acc);
        if (kt + 1 < KT && !(ABL & 2)) sstore(A1, B1);
        __syncthreads();
        if (kt + 1 >= KT) break;
        if (kt + 2 < KT && !(ABL & 1)) gload((kt + 2) * 32);
        if (!(ABL & 4)) ML::template slab7<false>(oa, ob, A1, B1, A0, B0, st, acc);
        if (kt + 2 < KT && !(ABL & 2)) sstore(A0, B0);
        __syncthreads();
    }
    float* Cz = C + (size_t)bz * M * N;
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cz[(size_t)(m0 + acc_row<TC>(mi, r)) * N + n0 + acc_col<TC>(ni)] = acc[mi][ni][r];
}

static uint16_t f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static float bf2f(uint16_t h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <class TC, int ABL = 0>
static void run(const char* name, int M, int N, int K, int KS, const uint16_t* dA, const uint16_t* dB, float* dC, const std::vector<float>& ref) {
    using ML = MainLoop<TC, true, true, true, true, false, 7>;
    const size_t smem = ML::smem_bytes();
    hipFuncSetAttribute(reinterpret_cast<const void*>(plane_gemm_kernel<TC, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    dim3 grid(N / TC::BN, M / TC::BM, KS);
    const int kchunk = K / KS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((plane_gemm_kernel<TC, ABL>), grid, dim3(256), smem, 0, dA, dB, dC, M, N, K, kchunk);
    hipEventRecord(e0);
    const int it = 50;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((plane_gemm_kernel<TC, ABL>), grid, dim3(256), smem, 0, dA, dB, dC, M, N, K, kchunk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h((size_t)KS * M * N);
    hipMemcpy(h.data(), dC, h.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int z = 0; z < KS; ++z) s += h[((size_t)z * M + i) * N + j];
            err = fmax(err, fabs(s - ref[(size_t)i * N + j]));
        }
    const double us = ms / it * 1e3;
    printf("%-10s grid %4d x %3d x %d (%5d WGs, %5.1f KB LDS): %7.1f us  %6.1f TFLOP/s f32-grade  max|err| (64 rows) %.2e\n", name, grid.x, grid.y,
           grid.z, grid.x * grid.y * grid.z, smem / 1024.0, us, 2.0 * M * N * K / us / 1e6, err);
}

int main() {
    const int M = 2048, N = 512, K = 1536;
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    srand(1);
    for (auto& x : A) x = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    std::vector<uint16_t> Ap((size_t)3 * M * K), Bp((size_t)3 * N * K);
    auto split = [](const std::vector<float>& X, std::vector<uint16_t>& P, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            float r = X[i];
            for (int pl = 0; pl < 3; ++pl) {
                const uint16_t h = f2bf(r);
                P[pl * n + i] = h;
                r -= bf2f(h);
            }
        }
    };
    split(A, Ap, (size_t)M * K);
    split(B, Bp, (size_t)N * K);
    std::vector<float> ref((size_t)64 * N);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * B[(size_t)j * K + k];
            ref[(size_t)i * N + j] = (float)s;
        }
    uint16_t *dA, *dB;
    float* dC;
    hipMalloc(&dA, Ap.size() * 2);
    hipMalloc(&dB, Bp.size() * 2);
    hipMalloc(&dC, (size_t)8 * M * N * 4);
    hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice);
    run<TileCfg<64, 64, 32, 2, 2, 1>>("64x64 k4", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 1>("  no gload", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 3>("  no ld/st", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 4>("  no mfma", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 7>("  nothing", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 1>("k1 no gload", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 4>("k1 no mfma", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>, 7>("k1 nothing", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>>("64x64 k2", M, N, K, 2, dA, dB, dC, ref);
    run<TileCfg<64, 64, 32, 2, 2, 1>>("64x64 k1", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<64, 32, 32, 4, 1, 1>>("64x32 k2", M, N, K, 2, dA, dB, dC, ref);
    run<TileCfg<64, 32, 32, 4, 1, 1>>("64x32 k1", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<32, 32, 32, 2, 2, 1>>("32x32 k1", M, N, K, 1, dA, dB, dC, ref);
    run<TileCfg<128, 64, 32, 2, 2, 1>>("128x64 k4", M, N, K, 4, dA, dB, dC, ref);
    run<TileCfg<128, 64, 32, 2, 2, 1>>("128x64 k8", M, N, K, 8, dA, dB, dC, ref);
    run<TileCfg<128, 128, 32, 2, 2, 1>>("128x128 k8", M, N, K, 8, dA, dB, dC, ref);
    return 0;
}
