// Dense f32 products behind nn.Linear / `@` on the hot path, on the f32 MFMA tile engine (gemm_core.h).
//   cpg_linear_fwd         Y = X W^T + b          (models/encoder.py:35-36,50-51; models/decoder.py:43-45,83)
//   cpg_linear_bwd_input   dX = dY W
//   cpg_linear_bwd_weight  dW = dY^T X, db = colsum(dY)   (split-K over the batch dimension, deterministic slab reduce)
//   cpg_matmul_nn          Y = X B                (losses.py:85  z @ rf_w)
#include "gemm_core.h"
#include "cpg_internal.h"
#include "pair_tn.h"
#ifndef CPG_TN_PRODUCT_SPLIT
#define CPG_TN_PRODUCT_SPLIT 7  // 0: exact-f32 MFMA; 7: six bf16 MFMAs on 3-way split operands, f32-grade (gemm_core.h)
#endif
#include <stdlib.h>
#include <string.h>
#ifndef CPG_PAIR_TN_WGS
#define CPG_PAIR_TN_WGS 512   // workgroups the f16-pair 128 x 128 dW_hh launch aims at (its 51 KB of LDS and 136 registers allow three per CU)
#endif
#ifndef CPG_PAIR_TN_128
#define CPG_PAIR_TN_128 1   // the f16-pair dW_hh product on the single-buffered 128 x 128 tile (two workgroups per CU) instead of 256 x 128
#endif

struct GemmArgs {
    const float* A; int lda; int M;
    const float* B; int ldb; int N;
    int K;
    float* C; int ldc;
    const float* bias;
    int accumulate;
    const uint8_t* a_mask; float a_mscale;   // multiplies A on load
    const uint8_t* b_mask; float b_mscale;   // multiplies B on load
    const uint8_t* c_mask; float c_mscale;   // multiplies C on store (same indexing as C)
    int k_chunk;        // split-K: blockIdx.z handles [z*k_chunk, min(K,(z+1)*k_chunk)), writes slab z
    size_t slab_stride; // floats between slabs (0 when not split)
    int pairs_a = 0, pairs_b = 0;  // 8-byte pairs of the scalar staging path are safe (OpA::pairs), set by launch_gemm
    int a_bf16 = 0;                // A holds bf16 elements (OpA::bf16): the dW_hh product on bf16 gate gradients
    const int* a_exps = nullptr;   // PREC 8 (f16 pairs): power-of-two exponent per 32-column group of A's M axis (OpA::exps); the kernel
    int a_exps_mod = 1;            // multiplies the columns by 2^e on the way in and the output rows by 2^-e on the way out
    const int* b_wx = nullptr;     // PREC 8, nn.Linear-shaped form: partial maxima of B (the weights; cpg_weight_absmax) - B goes in times 2^e_w
                                   // (gemm_core.h: weight_exp_from_parts), the result comes out times its inverse
};

// PREC: 7 = f32-grade (three bf16 planes), 8 = f32-grade on f16 pairs (the dW_hh product behind the f16-pair BPTT), 1 = bf16 compute
// mode (cpg_set_compute_mode(1)); only the transposed-use (dW = dY^T X) products run on the plane engine, everything else is the
// exact-f32 MFMA whatever PREC says
template <class TC, bool A_KC, bool B_KC, int PREC>
constexpr int gemm_split() {
    if (PREC == 8 && A_KC && B_KC && TC::BK == 32) return 8;   // y = x W^T on f16 pairs: only where the caller vouches for x (cpg_linear_fwd_pairs)
    return (!A_KC && !B_KC && CPG_TN_PRODUCT_SPLIT == 7) ? PREC : 0;
}
template <class TC, bool A_KC, bool B_KC, bool VEC, bool MASKS, int PREC = 7, bool A_BF16 = false>
using GemmLoop = MainLoop<TC, A_KC, B_KC, VEC, VEC, MASKS, gemm_split<TC, A_KC, B_KC, PREC>(), A_BF16>;

template <class TC, bool A_KC, bool B_KC, bool VEC, bool MASKS, int PREC, bool A_BF16 = false>
__global__ __launch_bounds__(TC::NT) void gemm_kernel(GemmArgs g) {
    int bx, by, bz;
    xcd_tile_order<PREC == 1>(bx, by, bz);
    const int m0 = by * TC::BM, n0 = bx * TC::BN;
    int kb = 0, K = g.K;
    if (g.k_chunk > 0) {
        kb = bz * g.k_chunk;
        K = min(g.K - kb, g.k_chunk);
    }
    const size_t aoff = A_KC ? (size_t)kb : (size_t)kb * g.lda;
    const size_t boff = B_KC ? (size_t)kb : (size_t)kb * g.ldb;
    OpA a{g.a_bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(g.A) + aoff) : g.A + aoff, g.lda, m0, g.M,
          g.a_mask ? g.a_mask + aoff : nullptr, g.a_mscale, g.pairs_a, g.a_bf16, g.a_exps, g.a_exps_mod};
    OpB b{g.B + boff, g.ldb, n0, g.N, 0, g.b_mask ? g.b_mask + boff : nullptr, g.b_mscale, g.pairs_b};
    float b_back = 1.f;
    if constexpr (PREC == 8 && A_KC && B_KC) {
        const int e = weight_exp_from_parts(g.b_wx);
        b.pscale = pair_pow2(e);
        b_back = pair_pow2(-e);
    }
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposed-operand products (dW = dY^T X: the 80 GFLOP dW_hh product) run on split bf16 operands, see gemm_core.h
    GemmLoop<TC, A_KC, B_KC, VEC, MASKS, PREC, A_BF16>::run(a, b, K, acc);
    float* C = g.C + (size_t)bz * g.slab_stride;
    const bool plain = g.k_chunk == 0;
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) {
            const int col = n0 + acc_col<TC>(ni);
            if (col >= g.N) continue;
            const float bv = (plain && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                if (row >= g.M) continue;
                const size_t o = (size_t)row * g.ldc + col;
                if constexpr (PREC == 8 && A_KC && B_KC) {
                    acc[mi][ni][r] *= b_back;   // the weights' power of two back out (exact)
                } else if constexpr (PREC == 8) {   // take the column scale of the A operand back out (exact)
                    const int e = g.a_exps ? g.a_exps[(row % g.a_exps_mod) / 32] : 0;
                    acc[mi][ni][r] *= __builtin_bit_cast(float, (unsigned)(127 - (e == INT_MAX ? 0 : e)) << 23);
                }
                float v = acc[mi][ni][r] + bv;
                if (plain) {
                    if (g.accumulate) v += C[o];
                    if (g.c_mask) v = g.c_mask[o] ? v * g.c_mscale : 0.f;
                }
                C[o] = v;
            }
        }
}

// out[m,n] (+)= sum_z slab[z][m,n]
__global__ void slab_reduce_kernel(const float* slabs, size_t slab_stride, int S, float* C, int ldc, int M, int N,
                                   int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = i / N, n = i % N;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[z * slab_stride + (size_t)m * N + n];
    const size_t o = (size_t)m * ldc + n;
    C[o] = accumulate ? C[o] + s : s;
}
// the same sums (same order over z), four columns per thread with 16-byte loads, the slabs' loads of a thread all in flight:
// N % 4 == 0, ldc % 4 == 0, 16-byte aligned bases
__global__ void slab_reduce4_kernel(const float* __restrict__ slabs, size_t slab_stride, int S, float* __restrict__ C, int ldc, int M, int N,
                                    int accumulate) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = N / 4;
    if (q >= (size_t)M * n4) return;
    const int m = q / n4, n = (int)(q - (size_t)m * n4) * 4;
    const float* src = slabs + (size_t)m * N + n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 8 <= S; z += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(z + u) * slab_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s.x += v[u].x;
            s.y += v[u].y;
            s.z += v[u].z;
            s.w += v[u].w;
        }
    }
    for (; z < S; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)z * slab_stride);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(C + (size_t)m * ldc + n);
    if (accumulate) {
        const float4 c = *o;
        s.x += c.x;
        s.y += c.y;
        s.z += c.z;
        s.w += c.w;
    }
    *o = s;
}
static void slab_reduce(const float* slabs, size_t slab_stride, int S, float* C, int ldc, int M, int N, int accumulate, hipStream_t s) {
    if (N % 4 == 0 && ldc % 4 == 0 && slab_stride % 4 == 0 && aligned16(slabs) && aligned16(C)) {
        const size_t n = (size_t)M * (N / 4);
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, slab_stride, S, C, ldc, M, N, accumulate);
    } else {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, slab_stride, S, C, ldc, M, N, accumulate);
    }
}

// part[chunk][n] = sum over rows of the chunk of X[m, n]; 64 columns x 4 row lanes per block.
__global__ void colsum_partial_kernel(const float* X, int ld, int M, int N, int rows_per_chunk, float* part) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    const int mb = blockIdx.y * rows_per_chunk, me = min(M, mb + rows_per_chunk);
    float s = 0.f;
    if (n < N)
        for (int m = mb + ty; m < me; m += 4) s += X[(size_t)m * ld + n];
    red[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && n < N) part[(size_t)blockIdx.y * N + n] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                              red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[n] = sum over chunks of part[chunk][n]; 64 columns x 16 chunk lanes per block (fixed order: deterministic).
__global__ void colsum_final_kernel(const float* part, int chunks, int N, float* out, int accumulate) {
    __shared__ float red[16][64];
    const int n = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    float s = 0.f;
    if (n < N)
        for (int c = ty; c < chunks; c += 16) s += part[(size_t)c * N + n];
    red[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][threadIdx.x];
        out[n] = accumulate ? out[n] + t : t;
    }
}

template <class TC, bool A_KC, bool B_KC, int PREC>
static int launch_tc_p(const GemmArgs& g, int zdim, bool vec, hipStream_t s) {
    dim3 grid(cdiv(g.N, TC::BN), cdiv(g.M, TC::BM), zdim);
    const size_t smem = GemmLoop<TC, A_KC, B_KC, true, false, PREC>::smem_bytes();
    const bool masks = g.a_mask || g.b_mask;
    if (smem > 64 * 1024) {  // more than the default dynamic-LDS limit: opt in once per instantiation
        const void* ks[4] = {reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, true, true, PREC>),
                             reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, true, false, PREC>),
                             reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, false, true, PREC>),
                             reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, false, false, PREC>)};
        const int rc = cpg_allow_big_lds(ks[(vec ? 0 : 2) + (masks ? 0 : 1)], (int)smem);
        if (rc) return rc;
    }
    if constexpr (!A_KC && !B_KC) {
        if (g.a_bf16) {   // bf16 gate gradients as the dY operand (cpg_gemm_tn checked the alignment conditions of the 16-byte path)
            if (!vec || masks) {
                cpg_set_error("gemm: a bf16 dY operand needs the 16-byte staging path and no keep-mask");
                return -2;
            }
            if (smem > 64 * 1024) {
                const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, true, false, PREC, true>), (int)smem);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, true, false, PREC, true>), grid, dim3(TC::NT), smem, s, g);
            CPG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (vec && masks)
        hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, true, true, PREC>), grid, dim3(TC::NT), smem, s, g);
    else if (vec)
        hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, true, false, PREC>), grid, dim3(TC::NT), smem, s, g);
    else if (masks)
        hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, false, true, PREC>), grid, dim3(TC::NT), smem, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, false, false, PREC>), grid, dim3(TC::NT), smem, s, g);
    CPG_LAUNCH_CHECK();
    return 0;
}

// bf16 compute mode covers the big recurrent weight-gradient product only (the 256x128 tile: dW_hh); the small nn.Linear
// weight gradients stay f32-grade
template <class TC, bool A_KC, bool B_KC>
static int launch_tc(const GemmArgs& g, int zdim, bool vec, hipStream_t s) {
    if constexpr (!A_KC && !B_KC && (TC::NT == 512 || (TC::BM == 128 && TC::BN == 128))) {
        if (cpg_compute_mode_get() == 1) return launch_tc_p<TC, A_KC, B_KC, 1>(g, zdim, vec, s);
    }
    if constexpr (!A_KC && !B_KC && (TC::NT == 512 || (TC::BM == 128 && TC::BN == 128))) {   // f16 pairs: the big tiles of the dW_hh product, 16-byte staging path, no masks
        if (g.a_exps && vec && !g.a_mask && !g.b_mask && !g.a_bf16) {
            const size_t smem = GemmLoop<TC, A_KC, B_KC, true, false, 8>::smem_bytes();
            if (smem > 64 * 1024) {
                const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(gemm_kernel<TC, A_KC, B_KC, true, false, 8>), (int)smem);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((gemm_kernel<TC, A_KC, B_KC, true, false, 8>), dim3(cdiv(g.N, TC::BN), cdiv(g.M, TC::BM), zdim), dim3(TC::NT),
                               smem, s, g);
            CPG_LAUNCH_CHECK();
            return 0;
        }
    }
    return launch_tc_p<TC, A_KC, B_KC, 7>(g, zdim, vec, s);
}

using T128x64 = TileCfg<128, 64, 32, 2, 2, 1>;
using T128x32 = TileCfg<128, 32, 32, 4, 1, 1>;
using T32x128 = TileCfg<32, 128, 32, 1, 4, 1>;
using T64x64 = TileCfg<64, 64, 32, 2, 2, 1>;
using T64x32 = TileCfg<64, 32, 32, 4, 1, 1>;   // small tiles for products with few 64x64 tiles (exact-f32 engine only)
using T32x64 = TileCfg<32, 64, 32, 2, 2, 1>;
using T32x32 = TileCfg<32, 32, 32, 2, 2, 1>;
using T128x128 = TileCfg<128, 128, 32, 2, 2, 1>;
// 512-thread workgroup (two waves per SIMD from ONE workgroup), each wave a 64x64 block: the dW_hh product.  One such
// workgroup per CU (150 KB of split-plane LDS images); operand traffic per MAC is 2/3 of the 128x64 tile's.
using T256x128 = TileCfg<256, 128, 32, 4, 2, 1, 512>;
// 192 x 128: the dW_hh product (M = 3H = 1536, N = H = 512) is then 8 x 4 = 32 tiles per K-chunk, i.e. with split-K 8 ONE
// K-chunk per XCD (xcd_tile_order: 32 workgroups = the XCD's 32 CUs) - every A / B panel row is fetched into exactly one L2.
using T192x128 = TileCfg<192, 128, 32, 4, 2, 1, 512>;

enum TnTile { TN_AUTO = 0, TN_256x128, TN_192x128, TN_128x128, TN_128x64, TN_64x64, TN_128x32, TN_32x128 };

// Option tn_tile forces the tile of the transposed-use (dW = dY^T X) products: tools/kb.py and tests/test_gpu_tiles.py
// (every instantiation against the golden vectors).
static TnTile tn_tile_knob() {
    const CpgOptVal& o = cpg_opt(OPT_TN_TILE);
    if (!o.set) return TN_AUTO;
    const char* e = o.s;
    if (!strcmp(e, "256x128")) return TN_256x128;
    if (!strcmp(e, "192x128")) return TN_192x128;
    if (!strcmp(e, "128x128")) return TN_128x128;
    if (!strcmp(e, "128x64")) return TN_128x64;
    if (!strcmp(e, "64x64")) return TN_64x64;
    if (!strcmp(e, "128x32")) return TN_128x32;
    if (!strcmp(e, "32x128")) return TN_32x128;
    return TN_AUTO;
}

template <bool A_KC, bool B_KC>
static int launch_gemm(const GemmArgs& g_in, int zdim, hipStream_t s, TnTile force = TN_AUTO) {
    GemmArgs g = g_in;
    // rows that are only 8-byte aligned (z_dim = 510): the scalar staging path may move pairs (gemm_core.h, fetch)
    const bool even_k = g.K % 2 == 0 && g.k_chunk % 2 == 0;
    g.pairs_a = (((uintptr_t)g.A) & 7) == 0 && g.lda % 2 == 0 && (A_KC ? even_k : g.M % 2 == 0);
    g.pairs_b = (((uintptr_t)g.B) & 7) == 0 && g.ldb % 2 == 0 && (B_KC ? even_k : g.N % 2 == 0);
    // 16-byte operand loads need aligned bases / leading dimensions and vectors that never straddle a bound
    const bool vec = aligned16(g.A) && aligned16(g.B) && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
                     (!g.a_mask || (((uintptr_t)g.a_mask) & 3) == 0) && (!g.b_mask || (((uintptr_t)g.b_mask) & 3) == 0) &&
                     (g.k_chunk % 4 == 0) && ((A_KC || B_KC) ? g.K % 4 == 0 : true) && (A_KC || g.M % 4 == 0) &&
                     (B_KC || g.N % 4 == 0);
    if constexpr (!A_KC && !B_KC) {
        switch (force) {
            case TN_256x128: return launch_tc<T256x128, A_KC, B_KC>(g, zdim, vec, s);
            case TN_192x128: return launch_tc<T192x128, A_KC, B_KC>(g, zdim, vec, s);
            case TN_128x128: return launch_tc<T128x128, A_KC, B_KC>(g, zdim, vec, s);
            case TN_128x64: return launch_tc<T128x64, A_KC, B_KC>(g, zdim, vec, s);
            case TN_64x64: return launch_tc<T64x64, A_KC, B_KC>(g, zdim, vec, s);
            case TN_128x32: return launch_tc<T128x32, A_KC, B_KC>(g, zdim, vec, s);
            case TN_32x128: return launch_tc<T32x128, A_KC, B_KC>(g, zdim, vec, s);
            default: break;
        }
    }
    if constexpr (A_KC) {  // nn.Linear forward / input-gradient products (exact-f32 engine): option gemm_tile forces a tile
        if (const CpgOptVal& o = cpg_opt(OPT_GEMM_TILE); o.set) {
            const char* e = o.s;
            if (!strcmp(e, "128x64")) return launch_tc<T128x64, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "64x64")) return launch_tc<T64x64, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "64x32")) return launch_tc<T64x32, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "32x64")) return launch_tc<T32x64, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "32x32")) return launch_tc<T32x32, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "128x32")) return launch_tc<T128x32, A_KC, B_KC>(g, zdim, vec, s);
            if (!strcmp(e, "32x128")) return launch_tc<T32x128, A_KC, B_KC>(g, zdim, vec, s);
        }
    }
    if (g.M <= 32) return launch_tc<T32x128, A_KC, B_KC>(g, zdim, vec, s);
    if (g.N <= 32) return launch_tc<T128x32, A_KC, B_KC>(g, zdim, vec, s);
    const long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 64) * zdim;
    // tools/gbench.py on MI355X: below two 128x64 tiles per CU, and for very short contractions (the launch is all epilogue),
    // 64x64 tiles are 20-25 % faster (rowc product 50 -> 40 us, d logits -> d out at K=24 55 -> 41 us)
    // tools/lin_sweep.py: fewer than two 64x64 tiles per CU AND a long contraction (the launch is one latency chain of >= 32 slabs
    // per workgroup) - 64x32 tiles double the workgroups: [z;c] input gradient 76.6 -> 59.4 us, encoder heads 32.7 -> 28.4
    if (A_KC && g.K >= 1024 && (long)cdiv(g.M, 64) * cdiv(g.N, 64) * zdim < 512) return launch_tc<T64x32, A_KC, B_KC>(g, zdim, vec, s);
    if (A_KC && (tiles128 < 512 || g.K <= 64)) return launch_tc<T64x64, A_KC, B_KC>(g, zdim, vec, s);
    if (tiles128 < 256) return launch_tc<T64x64, A_KC, B_KC>(g, zdim, vec, s);
    return launch_tc<T128x64, A_KC, B_KC>(g, zdim, vec, s);
}

int cpg_gemm_nt(const float* X, int ldx, const uint8_t* xmask, float xms, const float* W, int ldw, const float* bias,
                float* Y, int ldy, int M, int N, int K, int accumulate, hipStream_t s) {
    GemmArgs g{X, ldx, M, W, ldw, N, K, Y, ldy, bias, accumulate, xmask, xms, nullptr, 1.f, nullptr, 1.f, 0, 0};
    return launch_gemm<true, true>(g, 1, s);
}

// ------------------------------------------------------------------------------------------ grouped small products (round 6)
// A training step issues ~20 independent nn.Linear-shaped products of a few GFLOP at most - the three token tables, the two encoder heads
// and their gradients, ... (models/encoder.py:35-36,50-51, models/decoder.py:70-77) - each of which fills a fraction of the chip when it runs
// alone, one after the other.  cpg_gemm_group runs up to GEMM_GROUP_MAX problems of one form in ONE launch (a flat grid over all
// their tiles), and a problem may chain TWO (A, B, K) segments into the same accumulators:
//   * an input that is the concatenation of two tensors (the encoder's two final states) needs no torch.cat,
//   * dX = dY1 W1 + dY2 W2 (the two heads' input gradients) is one product instead of two products and an add,
//   * columns of the result can go to two destinations (n_split): the gradient of a concatenation needs no slicing copies.
// Engines: forms NT / NN run the exact-f32 MFMA, TN (weight gradients) the three-plane bf16 split, exactly as the single-problem
// launchers above - the sums of a problem are those cpg_linear_* would form (a chained problem adds its second segment's slabs
// behind the first's).
constexpr int GEMM_GROUP_MAX = 6;
struct GemmProb {            // mirrors CpgGemmProb of include/cpg_api.h field for field
    const float* A[2];
    const float* B[2];
    int lda[2], ldb[2], K[2];
    int M, N;
    float* C;
    float* C2;
    int ldc, ldc2, n_split;
    int accumulate;
    const float* bias;
    const int* wx_a;         // exponent records (cpg_weight_exp) of the A / B operands of BOTH segments, or null (2^0): a group whose every
    const int* wx_b;         // problem carries one for each side that needs it runs the NT form on f16 pairs (below)
    int pairs;               // 1: the caller vouches for f16-pair products of this problem (form 0 only)
    int pad_;
};
struct GemmGroup {
    GemmProb p[GEMM_GROUP_MAX];
    int tile0[GEMM_GROUP_MAX + 1];   // first flat tile of each problem
    int tiles_n[GEMM_GROUP_MAX];
    int pairs_a[GEMM_GROUP_MAX][2], pairs_b[GEMM_GROUP_MAX][2];
    int n;
};

// EPI 1 (random Fourier features, losses.py:84-88): the product raw = x W is turned into phi = cos(raw / sigma + phase[col]) * amp in the
// epilogue and summed over the tile's rows: part[problem][row tile][N] (fixed order: lanes, row blocks, then the two wave rows through
// LDS); raw itself is stored only where C is given.  The [B, R] feature matrix never exists.
struct RfEpi {
    const float* phase;
    float inv_sigma, amp;
    float* part;            // [nprob][max row tiles][N]
    int row_tiles;          // stride (in row tiles) between problems
};
template <class TC, bool A_KC, bool B_KC, bool VEC, int EPI = 0>
__global__ __launch_bounds__(TC::NT) void gemm_group_kernel(GemmGroup G, RfEpi E) {
    using Loop = GemmLoop<TC, A_KC, B_KC, VEC, false, 7>;
    const int t = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_GROUP_MAX; ++i)
        if (i < G.n && t >= G.tile0[i]) pi = i;
    const GemmProb& g = G.p[pi];
    const int tl = t - G.tile0[pi];
    const int by = tl / G.tiles_n[pi], bx = tl - by * G.tiles_n[pi];
    const int m0 = by * TC::BM, n0 = bx * TC::BN;
    f32x4 acc[TC::MI][TC::NI];
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sg = 0; sg < 2; ++sg) {
        if (g.K[sg] <= 0) break;
        if (sg) __syncthreads();
        OpA a{g.A[sg], g.lda[sg], m0, g.M, nullptr, 1.f, G.pairs_a[pi][sg], 0, nullptr, 1};
        OpB b{g.B[sg], g.ldb[sg], n0, g.N, 0, nullptr, 1.f, G.pairs_b[pi][sg]};
        Loop::run(a, b, g.K[sg], acc);
    }
    if constexpr (EPI == 1) {
        static_assert(TC::WM == 2 && TC::NT == 256, "two wave rows per tile");
        extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / TC::WN;
        float* red = cpg_smem;   // [WM][BN]; the main loop ended on a barrier
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) {
            const int cl = acc_col<TC>(ni), col = n0 + cl;
            const float ph = col < g.N ? E.phase[col] : 0.f;
            float cs = 0.f;
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + acc_row<TC>(mi, r);
                    const float raw = acc[mi][ni][r];
                    if (row < g.M && col < g.N) {
                        if (g.C) g.C[(size_t)row * g.ldc + col] = raw;
                        cs += cosf(raw * E.inv_sigma + ph) * E.amp;
                    }
                }
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
            if (lane < 16) red[wm * TC::BN + cl] = cs;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < TC::BN; c += TC::NT)
            if (n0 + c < g.N) E.part[((size_t)pi * E.row_tiles + by) * g.N + n0 + c] = red[c] + red[TC::BN + c];
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC::NI; ++ni) {
            const int col = n0 + acc_col<TC>(ni);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
            float* dst = g.C;
            int ld = g.ldc, c = col;
            if (g.C2 && col >= g.n_split) { dst = g.C2; ld = g.ldc2; c = col - g.n_split; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + acc_row<TC>(mi, r);
                if (row >= g.M) continue;
                const size_t o = (size_t)row * ld + c;
                float v = acc[mi][ni][r] + bv;
                if (g.accumulate) v += dst[o];
                dst[o] = v;
            }
        }
}

// ---- the NT form on the direct-to-LDS loop (gemm_core.h: DlLoop<64, 64, 3, 0>, the BPTT step's main loop): both operands go global -> LDS by
// LDS-DMA three slabs deep, no staging registers - a slab costs ~0.3 us against 1.5-3 us on the register-staged loop, which is what
// bounds these mid-sized products (16-64 slabs per workgroup, one round of workgroups: a latency chain).  Exact-f32 MFMA in MainLoop's
// contraction order: the sums are gemm_group_kernel's.  Needs 16-byte aligned rows and contractions that are multiples of 32 (the
// launcher checks); M / N tails are handled by clamping the rows the LDS-DMA reads and masking the stores.
// PREC 4: the same loop with the f32 slab values split into f16 pairs where the fragments are read (three f16 MFMAs per block in place
// of eight f32 ones: the exact form is bound by the f32 matrix pipe - 48 us for the encoder heads' 4.3 GFLOP).  Each operand is
// multiplied by a power of two from its exponent record first (its largest magnitude -> [2^13, 2^14): weights AND gradient matrices;
// values more than 2^16 below an operand's largest keep an absolute precision of 2^-39 of it), the result by the inverse.
template <int BM, int BN, int PREC>
__global__ __launch_bounds__(256) void gemm_group_dl_kernel(GemmGroup G) {
    using DL = DlLoop<BM, BN, 3, PREC>;
    constexpr int MI = DL::MI, NI = DL::NI;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    const int t = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_GROUP_MAX; ++i)
        if (i < G.n && t >= G.tile0[i]) pi = i;
    const GemmProb& g = G.p[pi];
    const int tl = t - G.tile0[pi];
    const int by = tl / G.tiles_n[pi], bx = tl - by * G.tiles_n[pi];
    const int m0 = by * BM, n0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int M = g.M, N = g.N;
    float back = 1.f;
    DlPairScale ps{1.f, 1.f};
    if constexpr (PREC == 4) {
        const int ea = g.wx_a ? weight_exp_from_parts(g.wx_a) : 0, eb = g.wx_b ? weight_exp_from_parts(g.wx_b) : 0;
        ps.a = pair_pow2(ea);
        ps.b = pair_pow2(eb);
        back = pair_pow2(-ea) * pair_pow2(-eb);
    }
    for (int sg = 0; sg < 2; ++sg) {
        const int K = g.K[sg];
        if (K <= 0) break;
        if (sg) __syncthreads();      // the ring is reused: every fragment read of the first segment is done
        const float* A = g.A[sg];
        const float* B = g.B[sg];
        const size_t lda = (size_t)g.lda[sg], ldb = (size_t)g.ldb[sg];
        auto ar = [&](int i, int r) { return A + (size_t)min(m0 + 32 * i + r, M - 1) * lda; };
        auto br = [&](int i, int r) { return B + (size_t)min(n0 + 32 * i + r, N - 1) * ldb; };
        if constexpr (PREC == 4) DL::run(A, lda, B, ldb, K, cpg_smem, acc, -1, []() {}, [](int) { return true; }, ps, ar, br);
        else DL::run(A, lda, B, ldb, K, cpg_smem, acc, -1, []() {}, [](int) { return true; }, DlNoScale{}, ar, br);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 16 + l15;
            if (col >= N) continue;
            const float bv = g.bias ? g.bias[col] : 0.f;
            float* dst = g.C;
            int ld = g.ldc, c = col;
            if (g.C2 && col >= g.n_split) { dst = g.C2; ld = g.ldc2; c = col - g.n_split; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 16 + 4 * lq + r;
                if (row >= M) continue;
                const size_t o = (size_t)row * ld + c;
                float v = acc[mi][ni][r] * back + bv;
                if (g.accumulate) v += dst[o];
                dst[o] = v;
            }
        }
}
// the direct-to-LDS form covers a group whose every segment has 16-byte aligned rows and a contraction of whole 32-deep slabs
static bool group_dl_ok(const GemmGroup& G) {
    for (int i = 0; i < G.n; ++i) {
        const GemmProb& g = G.p[i];
        if (g.M < 32 || g.N < 32) return false;
        for (int sg = 0; sg < 2 && g.K[sg] > 0; ++sg)
            if (g.K[sg] % 32 || g.lda[sg] % 4 || g.ldb[sg] % 4 || !aligned16(g.A[sg]) || !aligned16(g.B[sg])) return false;
    }
    return true;
}
static int launch_group_dl(GemmGroup& G, hipStream_t s) {
    constexpr int BM = 64, BN = 64;
    int tiles = 0;
    for (int i = 0; i < G.n; ++i) {
        G.tile0[i] = tiles;
        G.tiles_n[i] = cdiv(G.p[i].N, BN);
        tiles += G.tiles_n[i] * cdiv(G.p[i].M, BM);
    }
    G.tile0[G.n] = tiles;
    const size_t smem = DlLoop<BM, BN, 3, 0>::smem_floats() * sizeof(float);
    bool pairs = true;
    for (int i = 0; i < G.n; ++i) pairs = pairs && G.p[i].pairs != 0;
    if (pairs) hipLaunchKernelGGL((gemm_group_dl_kernel<BM, BN, 4>), dim3(tiles), dim3(256), smem, s, G);
    else hipLaunchKernelGGL((gemm_group_dl_kernel<BM, BN, 0>), dim3(tiles), dim3(256), smem, s, G);
    CPG_LAUNCH_CHECK();
    return 0;
}

template <class TC, bool A_KC, bool B_KC, int EPI = 0>
static int launch_group(GemmGroup& G, bool vec, hipStream_t s, RfEpi E = RfEpi{}) {
    int tiles = 0;
    for (int i = 0; i < G.n; ++i) {
        G.tile0[i] = tiles;
        G.tiles_n[i] = cdiv(G.p[i].N, TC::BN);
        tiles += G.tiles_n[i] * cdiv(G.p[i].M, TC::BM);
    }
    G.tile0[G.n] = tiles;
    const size_t smem = GemmLoop<TC, A_KC, B_KC, true, false, 7>::smem_bytes();
    const void* k = vec ? reinterpret_cast<const void*>(gemm_group_kernel<TC, A_KC, B_KC, true, EPI>)
                        : reinterpret_cast<const void*>(gemm_group_kernel<TC, A_KC, B_KC, false, EPI>);
    if (smem > 64 * 1024) {
        const int rc = cpg_allow_big_lds(k, (int)smem);
        if (rc) return rc;
    }
    if (vec) hipLaunchKernelGGL((gemm_group_kernel<TC, A_KC, B_KC, true, EPI>), dim3(tiles), dim3(TC::NT), smem, s, G, E);
    else hipLaunchKernelGGL((gemm_group_kernel<TC, A_KC, B_KC, false, EPI>), dim3(tiles), dim3(TC::NT), smem, s, G, E);
    CPG_LAUNCH_CHECK();
    return 0;
}

template <bool A_KC, bool B_KC>
static bool group_flags(GemmGroup& G) {
    bool vec = true;
    for (int i = 0; i < G.n; ++i) {
        const GemmProb& g = G.p[i];
        for (int sg = 0; sg < 2 && g.K[sg] > 0; ++sg) {
            const int K = g.K[sg];
            const bool even_k = K % 2 == 0;
            G.pairs_a[i][sg] = (((uintptr_t)g.A[sg]) & 7) == 0 && g.lda[sg] % 2 == 0 && (A_KC ? even_k : g.M % 2 == 0);
            G.pairs_b[i][sg] = (((uintptr_t)g.B[sg]) & 7) == 0 && g.ldb[sg] % 2 == 0 && (B_KC ? even_k : g.N % 2 == 0);
            vec = vec && aligned16(g.A[sg]) && aligned16(g.B[sg]) && g.lda[sg] % 4 == 0 && g.ldb[sg] % 4 == 0 &&
                  ((A_KC || B_KC) ? K % 4 == 0 : true) && (A_KC || g.M % 4 == 0) && (B_KC || g.N % 4 == 0);
        }
    }
    return vec;
}
// phi-sums of the random-feature MMD term (losses.compute_gaussian_rf, losses.py:84-93) for nx <= 2 inputs x_i [Bn, Z] against one basis
// rf_w [Z, R] in ONE launch: part[i][chunk][R] = sums over the chunk's 64 rows of cos(x_i rf_w / sigma + rf_b) * sqrt(2 / R); raw0 (optional)
// receives x_0 rf_w (what the backward pass needs).  *chunks_out = number of 64-row chunks.  part: nx * chunks * R floats.
CPG_EXPORT size_t cpg_rf_features_workspace(int nx, int Bn, int R) { return (size_t)nx * cdiv(Bn, 64) * R * sizeof(float); }
CPG_EXPORT int cpg_rf_features(int nx, const float* x0, const float* x1, int ldx, int Bn, int Z, const float* rf_w, int R, const float* rf_b,
                               float sigma, float* raw0, float* part, size_t part_bytes, void* stream) {
    CPG_CHECK_ARG(nx >= 1 && nx <= 2 && x0 && (nx == 1 || x1) && rf_w && rf_b && part && Bn > 0 && Z > 0 && R > 0 && sigma > 0.f && ldx >= Z);
    CPG_CHECK_ARG(part_bytes >= cpg_rf_features_workspace(nx, Bn, R));
    GemmGroup G;
    memset(&G, 0, sizeof(G));
    G.n = nx;
    for (int i = 0; i < nx; ++i) {
        GemmProb& g = G.p[i];
        g.A[0] = i ? x1 : x0; g.lda[0] = ldx; g.B[0] = rf_w; g.ldb[0] = R; g.K[0] = Z;
        g.M = Bn; g.N = R;
        g.C = i ? nullptr : raw0; g.ldc = R;
    }
    const bool vec = group_flags<true, false>(G);
    RfEpi E{rf_b, 1.f / sigma, sqrtf(2.f / (float)R), part, cdiv(Bn, 64)};
    return launch_group<T64x64, true, false, 1>(G, vec, (hipStream_t)stream, E);
}

template <bool A_KC, bool B_KC>
static int gemm_group(GemmGroup& G, hipStream_t s) {
    bool vec = group_flags<A_KC, B_KC>(G);
    int maxM = 0, maxN = 0, maxK = 0;
    long t64 = 0;
    for (int i = 0; i < G.n; ++i) {
        const GemmProb& g = G.p[i];
        maxM = g.M > maxM ? g.M : maxM;
        maxN = g.N > maxN ? g.N : maxN;
        t64 += (long)cdiv(g.M, 64) * cdiv(g.N, 64);
        for (int sg = 0; sg < 2 && g.K[sg] > 0; ++sg) maxK = g.K[sg] > maxK ? g.K[sg] : maxK;
    }
    if constexpr (A_KC && B_KC) {
        if (group_dl_ok(G) && !(cpg_opt(OPT_GEMM_TILE).set)) return launch_group_dl(G, s);
    }
    // tiles as launch_gemm picks them for one problem, on the group's totals (tools/gbench.py, tools/lin_sweep.py)
    if (maxM <= 32) return launch_group<T32x128, A_KC, B_KC>(G, vec, s);
    if (maxN <= 32) return launch_group<T128x32, A_KC, B_KC>(G, vec, s);
    if (A_KC && maxK >= 1024 && t64 < 512) return launch_group<T64x32, A_KC, B_KC>(G, vec, s);
    return launch_group<T64x64, A_KC, B_KC>(G, vec, s);
}

// form: 0 = NT  C[M,N] (+)= sum_s A_s[M,K_s] B_s[N,K_s]^T (+ bias)     nn.Linear forward
//       1 = NN  C[M,N] (+)= sum_s A_s[M,K_s] B_s[K_s,N]   (+ bias)     input gradient dX = dY W
//       2 = TN  C[M,N] (+)= sum_s A_s[K_s,M]^T B_s[K_s,N]              weight gradient dW = dY^T X
CPG_EXPORT int cpg_gemm_group_prob_bytes(void) { return (int)sizeof(GemmProb); }
CPG_EXPORT int cpg_gemm_group(int form, int nprob, const void* probs, void* stream) {
    CPG_CHECK_ARG(form >= 0 && form <= 2 && nprob >= 1 && nprob <= GEMM_GROUP_MAX && probs);
    GemmGroup G;
    memset(&G, 0, sizeof(G));
    G.n = nprob;
    memcpy(G.p, probs, sizeof(GemmProb) * nprob);
    for (int i = 0; i < nprob; ++i) {
        const GemmProb& g = G.p[i];
        CPG_CHECK_ARG(g.A[0] && g.B[0] && g.C && g.M > 0 && g.N > 0 && g.K[0] > 0 && g.K[1] >= 0 && (g.K[1] == 0 || (g.A[1] && g.B[1])));
        CPG_CHECK_ARG(!g.C2 || (g.n_split > 0 && g.n_split < g.N));
        CPG_CHECK_ARG(form != 2 || !g.bias);
    }
    hipStream_t s = (hipStream_t)stream;
    return form == 0 ? gemm_group<true, true>(G, s) : form == 1 ? gemm_group<true, false>(G, s) : gemm_group<false, false>(G, s);
}

// ---- backward of n <= 4 token tables tab_i = emb W_i^T + b_i (ops.TokenTablesFn), two launches for everything:
// stage 1, one workgroup per (table, chunk of TB_J = 16 gate rows j), all operands of the chunk in LDS:
//   dW_i[j][e]  (+)= sum_v dtab_i[v][j] emb[v][e]                 the chunk's rows of the weight gradient (contraction over V <= 32)
//   db_i[j]     (+)= sum_v dtab_i[v][j]
//   part[i][chunk][v][e] = sum_{j in chunk} dtab_i[v][j] W_i[j][e]   the chunk's share of the embedding gradient
// stage 2: demb[v][e] (+)= the shares summed in (table, chunk) order; row skip_row untouched when accumulating (nn.Embedding(padding_idx))
// / zero otherwise.  Plain f32 FMAs in a fixed order.  (Round 6, first version: one thread block per 64 embedding columns walked all
// n G gate rows - 72 workgroups, 66-79 us per call; and the weight gradient was a separate grouped launch whose 24-deep contraction kept
// the tile engine at 57 us.)
constexpr int TB_J = 16, TB_VMAX = 32, TB_EMAX = 256;   // (64-row chunks: 24-48 workgroups, each a 100-us chain of dependent loads; 16: short chains on 96-192 CUs)
struct TabBwdArgs {
    const float* dtab[4];
    const float* W[4];
    float* dW[4];
    float* db[4];
    const float* emb;
    float* part;
    int ldw[4], lddw[4];
    int n, V, G, E, lde, acc_w, acc_db, chunks;
};
__global__ __launch_bounds__(256) void token_tables_bwd_kernel(TabBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    const int i = blockIdx.y, ch = blockIdx.x, j0 = ch * TB_J, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V, E = a.E, EP = E + 1, nj = min(TB_J, a.G - j0);
    constexpr int EC = TB_EMAX / 64;         // column slots per lane: e = lane + 64 c
    float* dt_l = cpg_smem;                  // [V][TB_J + 1]
    float* emb_l = dt_l + V * (TB_J + 1);    // [V][EP]
    float* w_l = emb_l + V * EP;             // [TB_J][EP]
    // staging: rows over waves, columns over lanes - no index arithmetic, a wave's loads of a row group all in flight
    const float* const dtab = a.dtab[i];
    const float* const Wi = a.W[i];
    float* const dWi = a.dW[i];
    const int ldw = a.ldw[i], lddw = a.lddw[i];
    for (int x = tid; x < V * TB_J; x += 256) {
        const int v = x / TB_J, j = x - v * TB_J;
        dt_l[v * (TB_J + 1) + j] = j < nj ? dtab[(size_t)v * a.G + j0 + j] : 0.f;
    }
    for (int v0 = wave; v0 < V; v0 += 16) {
        float t[4][EC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const int v = v0 + 4 * u, e = lane + 64 * c;
                t[u][c] = (v < V && e < E) ? a.emb[(size_t)v * a.lde + e] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const int v = v0 + 4 * u, e = lane + 64 * c;
                if (v < V && e < E) emb_l[v * EP + e] = t[u][c];
            }
    }
    for (int r0 = wave; r0 < TB_J; r0 += 16) {
        float t[4][EC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const int j = r0 + 4 * u, e = lane + 64 * c;
                t[u][c] = (j < nj && e < E) ? Wi[(size_t)(j0 + j) * ldw + e] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const int j = r0 + 4 * u, e = lane + 64 * c;
                if (e < E) w_l[j * EP + e] = t[u][c];
            }
    }
    __syncthreads();
    // dW rows of the chunk: wave -> rows j = wave, wave + 4, ..., lane -> columns; V-deep
    if (dWi)
        for (int r0 = wave; r0 < nj; r0 += 16) {
            float s[4][EC], old[4][EC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int j = r0 + 4 * u, e = lane + 64 * c;
                    s[u][c] = 0.f;
                    old[u][c] = (a.acc_w && j < nj && e < E) ? dWi[(size_t)(j0 + j) * lddw + e] : 0.f;
                }
            for (int v = 0; v < V; ++v) {
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = dt_l[v * (TB_J + 1) + min(r0 + 4 * u, TB_J - 1)];
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const float ev = emb_l[v * EP + min(lane + 64 * c, E - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s[u][c] = fmaf(d[u], ev, s[u][c]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int j = r0 + 4 * u, e = lane + 64 * c;
                    if (j < nj && e < E) dWi[(size_t)(j0 + j) * lddw + e] = old[u][c] + s[u][c];
                }
        }
    if (a.db[i] && tid < nj) {
        float sb = 0.f;
        for (int v = 0; v < V; ++v) sb += dt_l[v * (TB_J + 1) + tid];
        float* o = a.db[i] + j0 + tid;
        *o = a.acc_db ? *o + sb : sb;
    }
    // the chunk's share of demb: wave -> rows v = wave, wave + 4, ..., lane -> columns; TB_J-deep
    if (a.part) {
        float* p = a.part + ((size_t)i * a.chunks + ch) * V * E;
        for (int v0 = wave; v0 < V; v0 += 16) {
            float s[4][EC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < EC; ++c) s[u][c] = 0.f;
            for (int j = 0; j < TB_J; ++j) {
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = dt_l[min(v0 + 4 * u, V - 1) * (TB_J + 1) + j];
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const float wv = w_l[j * EP + min(lane + 64 * c, E - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s[u][c] = fmaf(d[u], wv, s[u][c]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int v = v0 + 4 * u, e = lane + 64 * c;
                    if (v < V && e < E) p[v * E + e] = s[u][c];
                }
        }
    }
}
// 16 lanes per output: lane l sums shares l, l + 16, ... (all loads in flight at once), the 16 sums are combined in a fixed tree
__global__ void token_tables_demb_kernel(const float* __restrict__ part, int nparts, int V, int E, float* __restrict__ demb, int lde, int accumulate,
                                         int skip_row) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, x = t >> 4, l = t & 15;
    const bool ok = x < V * E;
    float s = 0.f;
    if (ok)
        for (int c = l; c < nparts; c += 16) s += part[(size_t)c * V * E + x];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (!ok || l) return;
    const int v = x / E, e = x - v * E;
    float* o = demb + (size_t)v * lde + e;
    if (v == skip_row) { if (!accumulate) *o = 0.f; }
    else *o = accumulate ? *o + s : s;
}
// dtab / W / dW / db: HOST arrays of n device pointers (W_i, dW_i: the [G, E] column block of the layer's W_ih / of its gradient that
// the embedding multiplies, row strides ldw[i] / lddw[i]; dW / db entries may be null).  emb [V, lde_in]; demb [V, lde] or null.
// workspace: cpg_token_tables_bwd_workspace(n, V, G, E) bytes.
CPG_EXPORT size_t cpg_token_tables_bwd_workspace(int n, int V, int G, int E) { return (size_t)n * cdiv(G, TB_J) * V * E * sizeof(float); }
CPG_EXPORT int cpg_token_tables_bwd(int n, int V, int G, int E, const void* const* dtab, const void* const* W, const int* ldw,
                                    const float* emb, int lde_in, void* const* dW, const int* lddw, int accumulate_w, void* const* db,
                                    int accumulate_db, float* demb, int lde, int accumulate_emb, int skip_row, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(n >= 1 && n <= 4 && V > 0 && V <= TB_VMAX && G > 0 && E > 0 && E <= TB_EMAX && dtab && W && ldw && emb && lde_in >= E && dW &&
                  lddw && db && (!demb || (lde >= E && workspace && workspace_bytes >= cpg_token_tables_bwd_workspace(n, V, G, E))));
    TabBwdArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        CPG_CHECK_ARG(dtab[i] && W[i] && ldw[i] >= E && (!dW[i] || lddw[i] >= E));
        a.dtab[i] = (const float*)dtab[i];
        a.W[i] = (const float*)W[i];
        a.dW[i] = (float*)dW[i];
        a.db[i] = (float*)db[i];
        a.ldw[i] = ldw[i];
        a.lddw[i] = lddw[i];
    }
    a.emb = emb; a.lde = lde_in; a.part = demb ? (float*)workspace : nullptr;
    a.n = n; a.V = V; a.G = G; a.E = E; a.acc_w = accumulate_w; a.acc_db = accumulate_db; a.chunks = cdiv(G, TB_J);
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = ((size_t)V * (TB_J + 1) + (size_t)V * (E + 1) + (size_t)TB_J * (E + 1)) * sizeof(float);
    if (smem > 64 * 1024) {
        const int rc = cpg_allow_big_lds((const void*)token_tables_bwd_kernel, (int)smem);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(token_tables_bwd_kernel, dim3(a.chunks, n), dim3(256), smem, s, a);
    CPG_LAUNCH_CHECK();
    if (demb) {
        hipLaunchKernelGGL(token_tables_demb_kernel, dim3(cdiv(V * E * 16, 256)), dim3(256), 0, s, (const float*)workspace, n * a.chunks, V, E, demb, lde,
                           accumulate_emb, skip_row);
        CPG_LAUNCH_CHECK();
    }
    return 0;
}

// ---- column sums of up to four [M, N] matrices in ONE single-stage launch (bias gradients of a group of heads: M = batch rows):
// out_i[n] (+)= sum_m X_i[m][n]; 16 columns x 64 row lanes per block (many blocks, short row loops), lanes reduced through LDS in lane order.
struct ColsumMultiArgs {
    const float* X[4];
    float* out[4];
    int ld[4];
    int M, N, accumulate;
};
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumMultiArgs a) {
    __shared__ float red[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, i = blockIdx.y, n = blockIdx.x * 16 + tx;
    float s = 0.f;
    if (n < a.N) {
        const float* x = a.X[i] + n;
        for (int m = ty; m < a.M; m += 64) s += x[(size_t)m * a.ld[i]];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < a.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][tx];
        a.out[i][n] = a.accumulate ? a.out[i][n] + t : t;
    }
}
CPG_EXPORT int cpg_colsum_multi(int nmat, const void* const* X, const int* ld, int M, int N, void* const* out, int accumulate, void* stream) {
    CPG_CHECK_ARG(nmat >= 1 && nmat <= 4 && X && ld && out && M > 0 && N > 0);
    ColsumMultiArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < nmat; ++i) {
        CPG_CHECK_ARG(X[i] && out[i] && ld[i] >= N);
        a.X[i] = (const float*)X[i];
        a.out[i] = (float*)out[i];
        a.ld[i] = ld[i];
    }
    a.M = M; a.N = N; a.accumulate = accumulate;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(cdiv(N, 16), nmat), dim3(1024), 0, (hipStream_t)stream, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ---- dst[c][r] = src[r][c] (r < R, c < C), zeros for R <= r < Rpad: operands of the NN / TN forms turned into K-contiguous rows
// (padded to whole 32-deep slabs) for the direct-to-LDS NT loop.  32 x 32 tiles through LDS.
__global__ void transpose_pad_kernel(const float* __restrict__ src, int lds, int R, int C, float* __restrict__ dst, int ldd, int Rpad) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < C && r < Rpad) dst[(size_t)c * ldd + r] = tile[threadIdx.x][i];
    }
}
CPG_EXPORT int cpg_transpose_pad(const float* src, int lds, int R, int C, float* dst, int ldd, int Rpad, void* stream) {
    CPG_CHECK_ARG(src && dst && R > 0 && C > 0 && lds >= C && Rpad >= R && ldd >= Rpad);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(cdiv(C, 32), cdiv(Rpad, 32)), dim3(32, 8), 0, (hipStream_t)stream, src, lds, R, C, dst, ldd, Rpad);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ---- largest magnitude of a weight matrix, for the engines that split weights into f16 pairs (gemm_core.h: weight_exp_from_parts).
// Block b takes rows b, b + WX_PARTS, ...; 16-byte loads where the rows allow them.  wx[b] = float bits of the block's maximum.
__device__ __forceinline__ float absmax_part(const float* __restrict__ w, int rows, int cols, int ld, int vec);
__global__ __launch_bounds__(1024) void weight_absmax2_kernel(const float* __restrict__ w1, int rows1, int cols1, int ld1, int vec1,
                                                               const float* __restrict__ w2, int rows2, int cols2, int ld2, int vec2,
                                                               int* __restrict__ wx) {
    __shared__ float red[16];
    float m = absmax_part(w1, rows1, cols1, ld1, vec1);
    if (w2) m = fmaxf(m, absmax_part(w2, rows2, cols2, ld2, vec2));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = fmaxf(t, red[k]);
        wx[blockIdx.x] = __builtin_bit_cast(int, t);
    }
}
__device__ __forceinline__ float absmax_part(const float* __restrict__ w, int rows, int cols, int ld, int vec) {
    float m = 0.f;
    if (vec) {
        const int q4 = cols >> 2;
        const int nr = (rows - (int)blockIdx.x + WX_PARTS - 1) / WX_PARTS;
        for (int i = threadIdx.x; i < nr * q4; i += 1024) {
            const int rr = i / q4, c = i - rr * q4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)(blockIdx.x + rr * WX_PARTS) * ld + 4 * c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        for (int r = blockIdx.x; r < rows; r += WX_PARTS)
            for (int c = threadIdx.x; c < cols; c += 1024) m = fmaxf(m, fabsf(w[(size_t)r * ld + c]));
    }
    return m;
}
__global__ __launch_bounds__(1024) void weight_absmax_kernel(const float* __restrict__ w, int rows, int cols, int ld, int vec, int* __restrict__ wx) {
    __shared__ float red[16];
    float m = 0.f;
    if (vec) {   // the block's rows as one index space of 16-byte quads: every thread has several independent loads in flight
        const int q4 = cols >> 2;
        const int nr = (rows - (int)blockIdx.x + WX_PARTS - 1) / WX_PARTS;   // rows blockIdx.x, + WX_PARTS, ...
        for (int i = threadIdx.x; i < nr * q4; i += 1024) {
            const int rr = i / q4, c = i - rr * q4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)(blockIdx.x + rr * WX_PARTS) * ld + 4 * c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        for (int r = blockIdx.x; r < rows; r += WX_PARTS)
            for (int c = threadIdx.x; c < cols; c += 1024) m = fmaxf(m, fabsf(w[(size_t)r * ld + c]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = fmaxf(t, red[k]);
        wx[blockIdx.x] = __builtin_bit_cast(int, t);
    }
}
int cpg_weight_absmax(const float* w, int rows, int cols, int ld, int* wx, hipStream_t s) {
    if (!w || !wx || rows <= 0 || cols <= 0 || ld < cols) { cpg_set_error("cpg_weight_absmax: bad argument"); return -2; }
    const int vec = (cols % 4 == 0 && ld % 4 == 0 && aligned16(w)) ? 1 : 0;
    hipLaunchKernelGGL(weight_absmax_kernel, dim3(WX_PARTS), dim3(1024), 0, s, w, rows, cols, ld, vec, wx);
    CPG_LAUNCH_CHECK();
    return 0;
}
// C ABI: the exponent record (cpg_weight_exp_bytes() bytes, device memory) of a weight matrix [rows, cols] (row stride ld) - what the
// entry points that split weights into f16 pairs on the fly take as `wx` (cpg_gru_step_fwd, cpg_gru_seq_fwd, cpg_linear_fwd_pairs)
CPG_EXPORT size_t cpg_weight_exp_bytes(void) { return WX_PARTS * sizeof(int); }
// ONE record for two matrices (their joint largest magnitude): operands that enter the same accumulators as two chained segments
CPG_EXPORT int cpg_weight_exp2(const float* w1, int rows1, int cols1, int ld1, const float* w2, int rows2, int cols2, int ld2, void* wx,
                               void* stream) {
    CPG_CHECK_ARG(w1 && w2 && wx && rows1 > 0 && cols1 > 0 && ld1 >= cols1 && rows2 > 0 && cols2 > 0 && ld2 >= cols2);
    const int v1 = (cols1 % 4 == 0 && ld1 % 4 == 0 && aligned16(w1)) ? 1 : 0, v2 = (cols2 % 4 == 0 && ld2 % 4 == 0 && aligned16(w2)) ? 1 : 0;
    hipLaunchKernelGGL(weight_absmax2_kernel, dim3(WX_PARTS), dim3(1024), 0, (hipStream_t)stream, w1, rows1, cols1, ld1, v1, w2, rows2, cols2, ld2,
                       v2, (int*)wx);
    CPG_LAUNCH_CHECK();
    return 0;
}
CPG_EXPORT int cpg_weight_exp(const float* w, int rows, int cols, int ld, void* wx, void* stream) {
    return cpg_weight_absmax(w, rows, cols, ld, (int*)wx, (hipStream_t)stream);
}

// y = x W^T + b on f16 pairs (three f16 MFMAs per block, gemm_core.h) for inputs the CALLER vouches for: magnitudes O(1) (|x| < 65504,
// absolute precision 2^-25 below 2^-14) - recurrent states.  W goes in times 2^e_w, e_w from the matrix' own largest magnitude (wx:
// cpg_weight_exp over W - any finite weight is covered; without wx the product runs the exact-f32 engine).
// Large products only (the 128 x 64 tile, 16-byte staging path); everything else runs cpg_gemm_nt.
int cpg_gemm_nt_pairs(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
                      int accumulate, const int* wx, hipStream_t s) {
    const bool vec = aligned16(X) && aligned16(W) && ldx % 4 == 0 && ldw % 4 == 0 && K % 4 == 0;
    if (!wx || !vec || cpg_compute_mode_get() == 1 || (long)cdiv(M, 128) * cdiv(N, 64) < 512 || K < 256)
        return cpg_gemm_nt(X, ldx, nullptr, 1.f, W, ldw, bias, Y, ldy, M, N, K, accumulate, s);
    GemmArgs g{X, ldx, M, W, ldw, N, K, Y, ldy, bias, accumulate, nullptr, 1.f, nullptr, 1.f, nullptr, 1.f, 0, 0};
    g.b_wx = wx;
    using TC = TileCfg<128, 64, 32, 2, 2, 1>;
    const size_t smem = GemmLoop<TC, true, true, true, false, 8>::smem_bytes();
    if (smem > 64 * 1024) {
        const int rc = cpg_allow_big_lds(reinterpret_cast<const void*>(gemm_kernel<TC, true, true, true, false, 8>), (int)smem);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((gemm_kernel<TC, true, true, true, false, 8>), dim3(cdiv(N, TC::BN), cdiv(M, TC::BM), 1), dim3(TC::NT), smem, s, g);
    CPG_LAUNCH_CHECK();
    return 0;
}

// Y[m,n] (+)= sum_k X[m,k] B[k,n] for a handful of rows (M <= 32) and long K: one (64-column, row) block with 16 K-lanes
// and a fixed-order LDS reduction.  The tile engine would put such a problem on 1-2 workgroups walking K serially.
__global__ void skinny_nn_kernel(const float* X, int ldx, const float* Bm, int ldb, float* Y, int ldy, int N, int K,
                                 int accumulate) {
    __shared__ float red[16][64];
    const int n = blockIdx.x * 64 + threadIdx.x, m = blockIdx.y, ty = threadIdx.y;
    float s = 0.f;
    if (n < N) {
        // 8 independent loads in flight per lane (the loop is latency-bound otherwise: 96 dependent trips at K=1536)
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = ty;
        for (; k + 16 * 7 < K; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] += X[(size_t)m * ldx + k + 16 * u] * Bm[(size_t)(k + 16 * u) * ldb + n];
        }
        for (; k < K; k += 16) p[0] += X[(size_t)m * ldx + k] * Bm[(size_t)k * ldb + n];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    red[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && n < N) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][threadIdx.x];
        const size_t o = (size_t)m * ldy + n;
        Y[o] = accumulate ? Y[o] + v : v;
    }
}

int cpg_gemm_nn(const float* X, int ldx, const float* Bm, int ldb, float* Y, int ldy, int M, int N, int K, int accumulate,
                const uint8_t* cmask, float cms, hipStream_t s) {
    if (M <= 32 && K >= 256 && !cmask) {
        hipLaunchKernelGGL(skinny_nn_kernel, dim3(cdiv(N, 64), M), dim3(64, 16), 0, s, X, ldx, Bm, ldb, Y, ldy, N, K, accumulate);
        CPG_LAUNCH_CHECK();
        return 0;
    }
    GemmArgs g{X, ldx, M, Bm, ldb, N, K, Y, ldy, nullptr, accumulate, nullptr, 1.f, nullptr, 1.f, cmask, cms, 0, 0};
    return launch_gemm<true, false>(g, 1, s);
}

// Plan of a dW[M,N] = dY^T X product contracting K rows: tile and split-K factor.
struct TnPlan {
    TnTile tile;
    int S, k_chunk;
};
static TnPlan tn_plan(int M, int N, int K, bool pairs = false) {
    TnPlan p{tn_tile_knob(), 1, 0};
    const CpgOptVal& split = cpg_opt(OPT_TN_SPLIT);
    // Large products (the dW_hh product: M=3H, N=H, K=T*B): 256x128 tiles, ONE 512-thread workgroup per CU, split-K chosen
    // so that a single round of <= 256 workgroups covers the problem.
    // (bf16 compute mode: the single-buffered 128x128 tile, two workgroups per CU - 347 us against 365 at the dW_hh shape incl.
    // the reductions; in f32-grade mode it is the slower one, 604 against 564)
    if (p.tile == TN_AUTO && M >= 512 && N >= 256 && K >= 8192 && M % 4 == 0 && N % 4 == 0)
        p.tile = ((cpg_compute_mode_get() == 1 || (pairs && CPG_PAIR_TN_128)) && M % 128 == 0 && N % 128 == 0) ? TN_128x128 : TN_256x128;
    long want;
    bool small_out = false;
    if (p.tile == TN_256x128) {
        const long tiles = (long)cdiv(M, 256) * cdiv(N, 128);
        want = 256 / tiles;
    } else if (p.tile == TN_192x128) {
        const long tiles = (long)cdiv(M, 192) * cdiv(N, 128);
        want = 256 / tiles;
    } else if (p.tile == TN_128x128) {   // single-buffered plane images: two workgroups per CU, one round
        const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
        want = ((pairs && cpg_compute_mode_get() != 1) ? CPG_PAIR_TN_WGS : 512) / tiles;
    } else {
        // Two 128x64 workgroups are resident per CU (57 KB LDS each): aim at ~3 full rounds of 512 workgroups so the
        // last round is not half empty (measured at M=1536,N=512,K=51200: S=4 (384 WGs) 1356 us, S=16 (1536 WGs) 1084 us).
        const long tiles = (long)cdiv(M, 128) * cdiv(N, 64);
        want = (1536 + tiles - 1) / tiles;
        small_out = tiles <= 16;
    }
    if (split.set && split.i > 0) want = split.i;
    if (want < 1) want = 1;
    long maxs = K / 512;  // at least 16 slabs per workgroup: shorter chunks are all prologue (measured: K=2048 split 16 ways
                          // ran a 3.2 GFLOP product in 0.46 ms)
    // ... except for outputs of a few tiles (the reference's default sizes: dW_hh [306,102] over K = T*B = 800 rows): there ONE
    // workgroup per tile walks all of K slab by slab (~1.1 us per slab, 28-69 us per product on 2-8 CUs); 4-slab chunks instead
    if (small_out) maxs = K / 128;
    if (maxs < 1) maxs = 1;
    if (!(split.set && split.i > 0) && want > maxs) want = maxs;
    if (want > 64) want = 64;
    p.k_chunk = cdiv(cdiv(K, (int)want), 32) * 32;
    p.S = cdiv(K, p.k_chunk);
    return p;
}

// C[N,Kd] (+)= A^T B where A = dY[Mr, N] (ld lddy), B = X[Mr, Kd] (ld ldx); contraction over the Mr rows.
int cpg_gemm_tn(const float* dY, int lddy, const float* X, int ldx, const uint8_t* xmask, float xms, float* dW, int lddw,
                int Mr, int N, int Kd, int accumulate, float* ws, size_t ws_bytes, hipStream_t s, int dy_bf16, const int* dy_exps,
                int dy_exps_mod) {
    TnPlan p = tn_plan(N, Kd, Mr, dy_exps != nullptr);
    int S = p.S;
    const size_t slab = (size_t)N * Kd;
    if (S > 1 && ws_bytes < slab * S * sizeof(float)) {
        S = 1;
    }
    if (dy_bf16 && !(lddy % 4 == 0 && N % 4 == 0 && (((uintptr_t)dY) & 15) == 0 && ldx % 4 == 0 && Kd % 4 == 0 && aligned16(X))) {
        cpg_set_error("cpg_gemm_tn: bf16 dY needs the 16-byte staging path (aligned bases, leading dimensions and widths multiples of 4)");
        return -2;
    }
    if (S <= 1) {
        GemmArgs g{dY, lddy, N, X, ldx, Kd, Mr, dW, lddw, nullptr, accumulate, nullptr, 1.f, xmask, xms, nullptr, 1.f, 0, 0};
        g.a_bf16 = dy_bf16;
        g.a_exps = dy_exps; g.a_exps_mod = dy_exps_mod;
        return launch_gemm<false, false>(g, 1, s, p.tile);
    }
    GemmArgs g{dY, lddy, N, X, ldx, Kd, Mr, ws, Kd, nullptr, 0, nullptr, 1.f, xmask, xms, nullptr, 1.f, p.k_chunk, slab};
    g.a_bf16 = dy_bf16;
    g.a_exps = dy_exps; g.a_exps_mod = dy_exps_mod;
    int rc = launch_gemm<false, false>(g, S, s, p.tile);
    if (rc) return rc;
    slab_reduce(ws, slab, S, dW, lddw, N, Kd, accumulate, s);
    CPG_LAUNCH_CHECK();
    return 0;
}

// Name (as rocprofv3 prints it, without "void " and the argument list) and split-K factor of the kernel cpg_gemm_tn picks
// for dW[N,Kd] = dY[Mr,N]^T X[Mr,Kd] with 16-byte aligned operands - for bench.py's roofline object.
// dy_pairs: the call hands column exponents (the dW_hh product behind the f16-pair BPTT, cpg_gru_wgrad_hh with its pair scratch)
CPG_EXPORT int cpg_gemm_tn_kernel_name(int Mr, int N, int Kd, int dy_pairs, char* buf, int n) {
    const TnPlan p = tn_plan(N, Kd, Mr, dy_pairs != 0);
    TnTile t = p.tile;
    if (t == TN_AUTO) {
        if (N <= 32) t = TN_32x128;
        else if (Kd <= 32) t = TN_128x32;
        else t = ((long)cdiv(N, 128) * cdiv(Kd, 64) * p.S < 256) ? TN_64x64 : TN_128x64;
    }
    const char* tc = t == TN_256x128 ? "256, 128, 32, 4, 2, 1, 512" : t == TN_192x128 ? "192, 128, 32, 4, 2, 1, 512" : t == TN_128x128 ? "128, 128, 32, 2, 2, 1, 256" :
                     t == TN_128x64 ? "128, 64, 32, 2, 2, 1, 256" : t == TN_64x64 ? "64, 64, 32, 2, 2, 1, 256" :
                     t == TN_128x32 ? "128, 32, 32, 4, 1, 1, 256" : "32, 128, 32, 1, 4, 1, 256";
    const bool vec = N % 4 == 0 && Kd % 4 == 0 && p.k_chunk % 4 == 0;
    const bool bf = (t == TN_256x128 || t == TN_192x128 || t == TN_128x128) && cpg_compute_mode_get() == 1;
    return snprintf(buf, n, "gemm_kernel<TileCfg<%s>, false, false, %s, false, %d>", tc, vec ? "true" : "false",
                    bf ? 1 : (dy_pairs && vec && (t == TN_256x128 || t == TN_192x128 || t == TN_128x128)) ? 8 : 7);
}
CPG_EXPORT int cpg_gemm_tn_split(int Mr, int N, int Kd, int dy_pairs) { return tn_plan(N, Kd, Mr, dy_pairs != 0).S; }

// ---- dW[M, N] (+)= A^T B with both operands as f16-pair plane images in memory (pair_tn.h): the dW_hh product of the all-T planes
// form.  128 x 128 tiles (two 68 KB workgroups per CU), split over the R rows so that ONE round of workgroups covers the chip
// (option tn_split overrides), partial slabs reduced in a fixed order.  Needs M % 128 == 0, N % 128 == 0, R % 32 == 0.
static int pair_tn_split(int M, int N, int R, size_t ws_bytes) {
    const long tiles = (long)(M / 128) * (N / 128);
    long S = (2L * cpg_device_cus()) / tiles;
    const CpgOptVal split = cpg_opt(OPT_TN_SPLIT);
    if (split.set && split.i > 0) S = split.i;
    const long maxs = R / 512 > 0 ? R / 512 : 1;   // at least 16 slabs per workgroup
    if (S > maxs) S = maxs;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    while (S > 1 && (size_t)M * N * S * sizeof(float) > ws_bytes) --S;
    // the exponent-factor table of a workgroup's slabs sits behind the LDS ring: keep two workgroups per CU where the problem allows
    while (S < maxs && PairTn<2, 2, 2>::smem_bytes(cdiv(cdiv(R, (int)S), 32)) > 80 * 1024 && (size_t)M * N * (S + 1) * sizeof(float) <= ws_bytes) ++S;
    return (int)S;
}
int cpg_pair_tn(const uint16_t* A, size_t lda, const int* a_ex, const int* a_emin, int a_groups, int a_seg_per_group, const uint16_t* B,
                size_t ldb, float* dW, int lddw, int M, int N, int R, int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
    if (!(M > 0 && N > 0 && R > 0 && M % 128 == 0 && N % 128 == 0 && R % 32 == 0 && aligned16(A) && aligned16(B) && lda % 8 == 0 && ldb % 8 == 0)) {
        cpg_set_error("cpg_pair_tn: needs M %% 128 == 0, N %% 128 == 0, rows %% 32 == 0 and 16-byte aligned plane images");
        return -2;
    }
    using P = PairTn<2, 2, 2>;
    int S = pair_tn_split(M, N, R, ws_bytes);
    PairTnArgs g{A, lda, a_ex, a_emin, a_groups, a_seg_per_group, B, ldb, nullptr, 0, 0, M, N, R, 0, 0};
    g.r_chunk = cdiv(cdiv(R, S), 32) * 32;
    S = cdiv(R, g.r_chunk);
    const size_t smem = P::smem_bytes(g.r_chunk / 32);
    if (smem > 160 * 1024) {
        cpg_set_error("cpg_pair_tn: workspace too small for a split that fits the exponent table into LDS");
        return -3;
    }
    int rc = cpg_allow_big_lds((const void*)pair_tn_kernel<2, 2, 2, 0, 0>, (int)smem);
    if (rc) return rc;
    if (S > 1) {
        g.C = ws; g.ldc = N; g.slab_stride = (size_t)M * N; g.accumulate = 0;
    } else {
        g.C = dW; g.ldc = lddw; g.slab_stride = 0; g.accumulate = accumulate;
    }
    hipLaunchKernelGGL((pair_tn_kernel<2, 2, 2, 0, 0>), dim3(N / P::BN, M / P::BM, S), dim3(P::NT), smem, s, g);
    CPG_LAUNCH_CHECK();
    if (S > 1) {
        slab_reduce(ws, (size_t)M * N, S, dW, lddw, M, N, accumulate, s);
        CPG_LAUNCH_CHECK();
    }
    return 0;
}
// the same product on ONE bf16 plane per operand (pair_tn.h, NP = 1): A [R, lda] bf16 (M leading columns used), B [R, ldb] bf16
int cpg_pair_tn_bf16(const uint16_t* A, size_t lda, const uint16_t* B, size_t ldb, float* dW, int lddw, int M, int N, int R, int accumulate,
                     float* ws, size_t ws_bytes, hipStream_t s) {
    if (!(M > 0 && N > 0 && R > 0 && M % 128 == 0 && N % 128 == 0 && R % 32 == 0 && aligned16(A) && aligned16(B) && lda % 8 == 0 && ldb % 8 == 0)) {
        cpg_set_error("cpg_pair_tn_bf16: needs M %% 128 == 0, N %% 128 == 0, rows %% 32 == 0 and 16-byte aligned operands");
        return -2;
    }
    using P = PairTn<2, 2, 2, 1>;
    int S = pair_tn_split(M, N, R, ws_bytes);
    PairTnArgs g{A, lda, nullptr, nullptr, 1, 1, B, ldb, nullptr, 0, 0, M, N, R, 0, 0};
    g.r_chunk = cdiv(cdiv(R, S), 32) * 32;
    S = cdiv(R, g.r_chunk);
    const size_t smem = P::smem_bytes(0);
    int rc = cpg_allow_big_lds((const void*)pair_tn_kernel<2, 2, 2, 0, 0, 1>, (int)smem);
    if (rc) return rc;
    if (S > 1) {
        g.C = ws; g.ldc = N; g.slab_stride = (size_t)M * N; g.accumulate = 0;
    } else {
        g.C = dW; g.ldc = lddw; g.slab_stride = 0; g.accumulate = accumulate;
    }
    hipLaunchKernelGGL((pair_tn_kernel<2, 2, 2, 0, 0, 1>), dim3(N / P::BN, M / P::BM, S), dim3(P::NT), smem, s, g);
    CPG_LAUNCH_CHECK();
    if (S > 1) {
        slab_reduce(ws, (size_t)M * N, S, dW, lddw, M, N, accumulate, s);
        CPG_LAUNCH_CHECK();
    }
    return 0;
}
size_t cpg_pair_tn_workspace(int M, int N, int R) {
    const long tiles = (long)(M / 128) * (N / 128);
    long S = tiles > 0 ? (2L * cpg_device_cus()) / tiles : 1;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    return (size_t)M * N * S * sizeof(float) + 256;
}
CPG_EXPORT int cpg_pair_tn_split(int M, int N, int R) { return pair_tn_split(M, N, R, (size_t)1 << 40); }

size_t cpg_gemm_tn_workspace(int Mr, int N, int Kd) {
    const TnPlan p = tn_plan(N, Kd, Mr), q = tn_plan(N, Kd, Mr, true);   // either form of the call (with / without column exponents)
    return (size_t)N * Kd * (p.S > q.S ? p.S : q.S) * sizeof(float) + 256;
}

// Row chunks of the column sum: 256-row chunks (at most 96) for wide matrices; narrow ones (few 64-column blocks) get more,
// shorter chunks so that the partial pass still launches ~1024 workgroups.
static int colsum_chunks(int M, int N) {
    int chunks = cdiv(M, 256);
    if (chunks > 96) chunks = 96;
    const int want = cdiv(1024, cdiv(N, 64)), most = cdiv(M, 16);
    if (chunks < want) chunks = want < most ? want : most;
    if (chunks < 1) chunks = 1;
    return chunks;
}

int cpg_colsum(const float* X, int ld, int M, int N, float* out, int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
    const int chunks = colsum_chunks(M, N);
    if ((size_t)chunks * N * sizeof(float) > ws_bytes) {
        cpg_set_error("cpg_colsum: workspace too small (%zu < %zu)", ws_bytes, (size_t)chunks * N * sizeof(float));
        return -3;
    }
    const int rows = cdiv(M, chunks);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), chunks), dim3(64, 4), 0, s, X, ld, M, N, rows, ws);
    CPG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 64)), dim3(64, 16), 0, s, ws, chunks, N, out, accumulate);
    CPG_LAUNCH_CHECK();
    return 0;
}

size_t cpg_colsum_workspace(int M, int N) { return (size_t)colsum_chunks(M, N) * N * sizeof(float); }

// ------------------------------------------------------------------------------------------ C ABI
CPG_EXPORT int cpg_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int M,
                              int N, int K, int accumulate, void* stream) {
    CPG_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    return cpg_gemm_nt(X, ldx, nullptr, 1.f, W, ldw, bias, Y, ldy, M, N, K, accumulate, (hipStream_t)stream);
}

// cpg_linear_fwd for an input of O(1) magnitudes (recurrent states: the input projection of an upper encoder layer) - large products
// then run on f16 pairs, see cpg_gemm_nt_pairs; same results within f32 rounding
CPG_EXPORT int cpg_linear_fwd_pairs(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int M,
                                    int N, int K, int accumulate, const void* wx, void* stream) {
    CPG_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    return cpg_gemm_nt_pairs(X, ldx, W, ldw, bias, Y, ldy, M, N, K, accumulate, (const int*)wx, (hipStream_t)stream);
}

CPG_EXPORT int cpg_linear_bwd_input(const float* dY, int lddy, const float* W, int ldw, float* dX, int lddx, int M, int N,
                                    int K, int accumulate, void* stream) {
    CPG_CHECK_ARG(dY && W && dX && M > 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K);
    // dX[M,K] = dY[M,N] * W[N,K]   (contraction over N; W rows are the contraction index: "XC" operand)
    return cpg_gemm_nn(dY, lddy, W, ldw, dX, lddx, M, K, N, accumulate, nullptr, 1.f, (hipStream_t)stream);
}

CPG_EXPORT size_t cpg_linear_bwd_weight_workspace(int M, int N, int K) {
    size_t a = cpg_gemm_tn_workspace(M, N, K), b = cpg_colsum_workspace(M, N);
    return (a > b ? a : b) + 256;
}

CPG_EXPORT int cpg_linear_bwd_weight(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw, float* db,
                                     int M, int N, int K, int accumulate, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    CPG_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K);
    int rc = cpg_gemm_tn(dY, lddy, X, ldx, nullptr, 1.f, dW, lddw, M, N, K, accumulate, (float*)workspace, workspace_bytes,
                         (hipStream_t)stream);
    if (rc) return rc;
    if (db) rc = cpg_colsum(dY, lddy, M, N, db, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    return rc;
}

// nn.Dropout in front of nn.Linear: the inter-layer dropout of nn.GRU(dropout=p_dropout) (models/encoder.py:25-30: the output of
// every encoder layer but the last, in train mode) feeding the next layer's W_ih product.  keep: uint8 0/1 with X's indexing.
CPG_EXPORT int cpg_linear_masked_fwd(const float* X, int ldx, const uint8_t* keep, float scale, const float* W, int ldw,
                                     const float* bias, float* Y, int ldy, int M, int N, int K, int accumulate, void* stream) {
    CPG_CHECK_ARG(X && keep && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    return cpg_gemm_nt(X, ldx, keep, scale, W, ldw, bias, Y, ldy, M, N, K, accumulate, (hipStream_t)stream);
}

// dX[M,K] = (dY[M,N] W[N,K]) .* keep*scale   (keep with dX's indexing)
CPG_EXPORT int cpg_linear_masked_bwd_input(const float* dY, int lddy, const float* W, int ldw, const uint8_t* keep, float scale,
                                           float* dX, int lddx, int M, int N, int K, void* stream) {
    CPG_CHECK_ARG(dY && W && keep && dX && M > 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K);
    return cpg_gemm_nn(dY, lddy, W, ldw, dX, lddx, M, K, N, 0, keep, scale, (hipStream_t)stream);
}

// dW[N,K] (+)= dY[M,N]^T (X .* keep*scale)[M,K] ; db[N] (+)= column sums of dY (db may be null); workspace as cpg_linear_bwd_weight
CPG_EXPORT int cpg_linear_masked_bwd_weight(const float* dY, int lddy, const float* X, int ldx, const uint8_t* keep, float scale,
                                            float* dW, int lddw, float* db, int M, int N, int K, int accumulate, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(dY && X && keep && dW && M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K);
    int rc = cpg_gemm_tn(dY, lddy, X, ldx, keep, scale, dW, lddw, M, N, K, accumulate, (float*)workspace, workspace_bytes,
                         (hipStream_t)stream);
    if (rc) return rc;
    if (db) rc = cpg_colsum(dY, lddy, M, N, db, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    return rc;
}

CPG_EXPORT size_t cpg_colsum_workspace_bytes(int M, int N) { return cpg_colsum_workspace(M, N) + 256; }

// out[N] (+)= column sums of X[M,N] (bias gradients; fixed two-stage partition)
CPG_EXPORT int cpg_colsum_f32(const float* X, int ld, int M, int N, float* out, int accumulate, void* workspace,
                              size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(X && out && workspace && M > 0 && N > 0 && ld >= N);
    return cpg_colsum(X, ld, M, N, out, accumulate, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

CPG_EXPORT int cpg_matmul_nn(const float* X, int ldx, const float* Bm, int ldb, float* Y, int ldy, int M, int N, int K,
                             int accumulate, void* stream) {
    CPG_CHECK_ARG(X && Bm && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldb >= N && ldy >= N);
    return cpg_gemm_nn(X, ldx, Bm, ldb, Y, ldy, M, N, K, accumulate, nullptr, 1.f, (hipStream_t)stream);
}
