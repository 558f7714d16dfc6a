// Micro-benchmark + check of pair_tn_kernel (csrc/pair_tn.h): dW[3H, H] = dG^T h over T*B rows with both operands as f16-pair planes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlled-peptide-generation_amd/csrc tools/micro/wgrad_planes.hip -o build_variants/wgrad_planes
//   build_variants/wgrad_planes [H=512] [R=51200] [check_rows=4096]
#include "pair_tn.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t h_bits(float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; }
static float h_val(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }

template <int WM, int WN, int NS, int ABL = 0, int PIPE = 0>
static float run(const PairTnArgs& a0, int S, float* ws, float* out, int iters, const char* tag, double flops) {
    using P = PairTn<WM, WN, NS>;
    PairTnArgs a = a0;
    a.r_chunk = ((a.R + S - 1) / S + 31) / 32 * 32;
    S = (a.R + a.r_chunk - 1) / a.r_chunk;
    a.C = S > 1 ? ws : out;
    a.ldc = a.N;
    a.slab_stride = (size_t)a.M * a.N;
    const size_t smem = P::smem_bytes(a.r_chunk / 32);
    CK(hipFuncSetAttribute((const void*)pair_tn_kernel<WM, WN, NS, ABL, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(a.N / P::BN, a.M / P::BM, S);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pair_tn_kernel<WM, WN, NS, ABL, PIPE>), grid, dim3(P::NT), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pair_tn_kernel<WM, WN, NS, ABL, PIPE>), grid, dim3(P::NT), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / iters;
    printf("%-28s S=%2d grid=%4d smem=%6zu  %8.1f us  %6.1f TFLOP/s algorithmic (%.3f of 833)\n", tag, S, grid.x * grid.y * grid.z, smem, us,
           flops / us * 1e-6, flops / us * 1e-6 / 833.3);
    return us;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 512;
    const int R = argc > 2 ? atoi(argv[2]) : 51200;
    const int RC = argc > 3 ? atoi(argv[3]) : 4096;
    const int M = 3 * H, N = H, G = 3, NG = H / 32;
    printf("dW[%d,%d] over %d rows; check on the first %d rows\n", M, N, R, RC);
    // synthetic gate gradients with magnitudes spread over row blocks and column groups, states in [-1, 1]
    std::vector<uint16_t> A((size_t)R * 2 * M), Bp((size_t)R * 2 * N);
    std::vector<int> ex((size_t)(R / 32) * NG), emin(NG, INT_MAX);
    std::vector<float> av((size_t)RC * M), bv((size_t)RC * N);   // dequantised values of the checked rows (image column order)
    srand(1);
    auto rnd = []() { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int rb = 0; rb < R / 32; ++rb)
        for (int cg = 0; cg < NG; ++cg) {
            const float mag = ldexpf(1.f, -((rb * 7 + cg * 3) % 23) - 3);
            std::vector<float> v(32 * 96);
            float vmax = 0.f;
            for (auto& x : v) { x = rnd() * mag; vmax = fmaxf(vmax, fabsf(x)); }
            int fe; frexpf(vmax, &fe);
            const int e = 14 - fe;
            ex[(size_t)rb * NG + cg] = e;
            emin[cg] = e < emin[cg] ? e : emin[cg];
            for (int r = 0; r < 32; ++r)
                for (int q = 0; q < 3; ++q)
                    for (int x = 0; x < 32; ++x) {
                        const float s = ldexpf(v[(r * 3 + q) * 32 + x], e);
                        const uint16_t hi = h_bits(s), lo = h_bits(s - h_val(hi));
                        const size_t row = (size_t)rb * 32 + r;
                        uint16_t* d = &A[row * 2 * M + (size_t)(3 * cg + q) * 64];
                        d[x] = hi; d[32 + x] = lo;
                        if (row < (size_t)RC) av[row * M + (3 * cg + q) * 32 + x] = ldexpf(h_val(hi) + h_val(lo), -e);
                    }
        }
    for (size_t row = 0; row < (size_t)R; ++row)
        for (int sg = 0; sg < N / 32; ++sg)
            for (int x = 0; x < 32; ++x) {
                const float s = rnd();
                const uint16_t hi = h_bits(s), lo = h_bits(s - h_val(hi));
                Bp[row * 2 * N + sg * 64 + x] = hi; Bp[row * 2 * N + sg * 64 + 32 + x] = lo;
                if (row < (size_t)RC) bv[row * N + sg * 32 + x] = h_val(hi) + h_val(lo);
            }
    uint16_t *dA, *dB; int *dex, *demin; float *ws, *out;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, Bp.size() * 2)); CK(hipMalloc(&dex, ex.size() * 4)); CK(hipMalloc(&demin, NG * 4));
    CK(hipMalloc(&ws, (size_t)64 * M * N * 4)); CK(hipMalloc(&out, (size_t)M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dex, ex.data(), ex.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(demin, emin.data(), NG * 4, hipMemcpyHostToDevice));
    PairTnArgs a{dA, (size_t)2 * M, dex, demin, NG, G, dB, (size_t)2 * N, out, N, 0, M, N, RC, 0, 0};
    // ---- check (no split, first RC rows) against f64 sums of the dequantised operands
    {
        CK(hipMemset(out, 0xff, (size_t)M * N * 4));
        a.R = RC; a.r_chunk = RC;
        using P = PairTn<2, 2, 2>;
        const size_t smem = P::smem_bytes(RC / 32);
        CK(hipFuncSetAttribute((const void*)pair_tn_kernel<2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((pair_tn_kernel<2, 2, 2>), dim3(N / 128, M / 128, 1), dim3(256), smem, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<float> got((size_t)M * N);
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        int nbad = 0;
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 5) {
                double s = 0, sa = 0;
                for (int r = 0; r < RC; ++r) { const double t = (double)av[(size_t)r * M + m] * bv[(size_t)r * N + n]; s += t; sa += fabs(t); }
                const int seg = m / 32, orow = (seg % G) * (M / G) + 32 * (seg / G) + m % 32;
                const double err = fabs(got[(size_t)orow * N + n] - s) / (sa + 1e-300);
                if (err > worst) worst = err;
                if (!(err < 2e-6) && nbad++ < 5) printf("  bad m=%d n=%d got %g want %g\n", m, n, got[(size_t)orow * N + n], s);
            }
        printf("check: worst |err| / sum|terms| = %.3g  (%s)\n", worst, (nbad == 0 && worst < 2e-6) ? "OK" : "FAIL");
    }
    const double flops = 2.0 * M * N * (double)R;
    a.R = R;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int it = argc > 5 ? atoi(argv[5]) : 20;
    if (mode == 0 || mode == 1) run<2, 2, 2>(a, 10, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 2) run<2, 2, 2, 4>(a, 10, ws, out, it, "128x128 NS=2 noDMA", flops);
    if (mode == 0 || mode == 7) run<2, 2, 2, 0, 1>(a, 10, ws, out, it, "128x128 NS=2 pipe", flops);
    if (mode == 0 || mode == 3) run<4, 2, 2>(a, 10, ws, out, it, "256x128 NS=2", flops);
    if (mode == 0 || mode == 4) run<4, 2, 3>(a, 10, ws, out, it, "256x128 NS=3", flops);
    if (mode == 0 || mode == 5) run<2, 2, 2>(a, 8, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 5) run<2, 2, 2>(a, 12, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 5) run<2, 2, 2>(a, 16, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 5) run<2, 2, 2>(a, 21, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 5) run<2, 2, 2>(a, 32, ws, out, it, "128x128 NS=2", flops);
    if (mode == 0 || mode == 6) run<2, 2, 3>(a, 10, ws, out, it, "128x128 NS=3", flops);
    if (mode == 0 || mode == 1) run<2, 2, 2>(a, 10, ws, out, it, "128x128 NS=2 (again)", flops);
    return 0;
}
