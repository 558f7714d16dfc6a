// nn.Linear-shaped products over MANY rows on f16-pair plane images (round 5): the input projection of an upper encoder layer
// (models/encoder.py:25-30: nn.GRU(num_layers > 1) feeds layer l the concatenated outputs of layer l-1 - [T B, 2 He] x [2 He, 3 He] per
// direction) and its two gradients.  At BASELINE.json configs[4] dimensions (T B = 51 200 rows, He = 1024) these three products were
// 27 of the step's 58 ms on engines that convert in their loops (exact-f32 MFMA 114 TFLOP/s, in-loop f16 split 154, bf16 triple 171).
// Here every operand is turned into a plane image ONCE by a bandwidth-bound pass and the products run conversion-free:
//   y  = x W^T + b        DlLoop<128,128,2,3>: both images K-contiguous, LDS-DMA ring, ds_read_b128 fragments (409 TFLOP/s measured)
//   dx = dy W             the same loop; dy's image carries one power of two per (32 rows x 32-unit group): every A fragment is brought to
//                         its row block's common unit on the way into the MFMAs (DlLoop's fscale hook), W^T is imaged in dy's k order
//   dW = dy^T x           pair_tn_kernel (pair_tn.h): LDS-DMA + transposing reads over the same two images
// Image formats: pair_engine.h / pair_tn.h (a row = 128-byte segments of [32 hi | 32 lo] f16).
#include "pair_engine.h"
#include "pair_tn.h"
#include "cpg_internal.h"

namespace {

// ---- x [R, C1 (+ C2)] f32 -> unscaled image [R][2 (C1 + C2)] (|x| <= 65504; states: |x| <= 1).  One thread = 4 columns of one row.
// wx (weights only): the matrix' exponent record - the image holds x 2^e_w (gemm_core.h: weight_exp_from_parts); null: unscaled (states).
__global__ void pair_rows_kernel(const float* __restrict__ x1, int ld1, int C1, const float* __restrict__ x2, int ld2, int C2, int R,
                                 uint16_t* __restrict__ img, const int* __restrict__ wx) {
    const float scale = wx ? pair_pow2(weight_exp_from_parts(wx)) : 1.f;
    const int C = C1 + C2, q4 = C / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * q4) return;
    const size_t row = i / q4;
    const int col = (int)(i - row * q4) * 4;
    const f32x4 v = col < C1 ? *reinterpret_cast<const f32x4*>(x1 + row * ld1 + col) : *reinterpret_cast<const f32x4*>(x2 + row * ld2 + (col - C1));
    pair_store4<1>(img, row, C, col, 0, v * scale);
}

// ---- W [G Hk, Nc] (row-major, ld) -> image of W^T: out[n][2 G Hk], k-segments in the order (32-unit group, block) - the k order of a
// gradient image (below) - times 2^e_w (wx: the matrix' exponent record).  grid (Nc/32, G Hk/32), block (32, 8): a 32 x 32 tile through LDS.
__global__ void pair_wT_kernel(const float* __restrict__ w, int ld, int G, int Hk, int Nc, uint16_t* __restrict__ out, const int* __restrict__ wx) {
    __shared__ float tile[32][33];
    const float sc = pair_pow2(weight_exp_from_parts(wx));
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;   // tile[k - r0][n - c0]
    for (int i = threadIdx.y; i < 32; i += 8) tile[i][threadIdx.x] = w[(size_t)(r0 + i) * ld + c0 + threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.y * 32 + threadIdx.x, jj = tid >> 3, q = tid & 7, c8 = (q & 3) * 8;
    const int blk = r0 / Hk, grp = (r0 - blk * Hk) / 32;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2h_pair(tile[c8 + 2 * i][jj] * sc, tile[c8 + 2 * i + 1][jj] * sc, hi[i], lo[i]);
    uint16_t* const d = out + (size_t)(c0 + jj) * 2 * G * Hk + (size_t)(G * grp + blk) * 64 + (q >> 2) * 32 + c8;
    *reinterpret_cast<uint4*>(d) = (q >> 2) ? make_uint4(lo[0], lo[1], lo[2], lo[3]) : make_uint4(hi[0], hi[1], hi[2], hi[3]);
}

// ---- gate gradients dG [R, ldg] f32, G blocks of H columns at column offsets off[q] -> image [R][2 G H] in the order (32-unit
// group, block), times 2^e with ONE e per (32 rows x group) over its G blocks (largest magnitude -> [2^13, 2^14); INT_MAX: all zero,
// zeros stored), ex[R/32][H/32], emin[H/32] (atomicMin; preset to INT_MAX by the caller's memset pattern 0x7f... see the launcher).
// One workgroup = 32 rows x one group: thread (row = tid / 8, 4 columns).
struct GradPlanesArgs {
    const float* dG; size_t ldg; int R, H, G; int off[4];
    uint16_t* img; int* ex; int* emin;
};
template <int G>
__global__ __launch_bounds__(256) void grad_planes_kernel(GradPlanesArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x, rb = blockIdx.y;
    const size_t row = (size_t)rb * 32 + (tid >> 3);
    const int col = grp * 32 + (tid & 7) * 4;
    f32x4 v[G];
    float vmax = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) {
        v[q] = *reinterpret_cast<const f32x4*>(a.dG + row * a.ldg + a.off[q] + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(v[q][j]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) red[wave] = vmax;
    __syncthreads();
    vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = INT_MAX;
    if (vmax > 0.f) {
        int fe = 0;
        if (vmax < 3.0e38f) { (void)frexpf(vmax, &fe); e = max(-100, min(100, 14 - fe)); }
        else e = 0;
    }
    if (tid == 0) {
        a.ex[(size_t)rb * (a.H / 32) + grp] = e;
        if (e != INT_MAX) atomicMin(a.emin + grp, e);
    }
    const float sc = e == INT_MAX ? 1.f : pair_pow2(e);
#pragma unroll
    for (int q = 0; q < G; ++q) pair_store4<G>(a.img, row, a.H, col, q, v[q] * sc);
}
__global__ void fill_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- C[R, N] (+)= A B^T on K-contiguous images.  A: [R][K2] with optional exponents (one per 32-row block and group of G consecutive
// 32-k segments: a gradient image), B: [N][K2] times 2^e_w of its exponent record b_wx.  128 x 128 tiles, 2 x 2 waves of 64 x 64.
struct PairNtArgs {
    const uint16_t* A; size_t lda;
    const int* a_ex; int a_groups;     // null: unscaled A
    const uint16_t* B; size_t ldb;
    int K2;                            // elements of a row of either image (2 x logical K)
    float* C; size_t ldc;
    const float* bias;
    int accumulate;
    const int* b_wx;
};
template <int G>
__global__ __launch_bounds__(256) void pair_nt_kernel(PairNtArgs g) {
    using DL = DlLoop<128, 128, 2, 3>;
    constexpr int MI = DL::MI, NI = DL::NI;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int m0 = by * 128, n0 = bx * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float back = pair_pow2(-weight_exp_from_parts(g.b_wx));
    float f0 = 1.f, f1 = 1.f;   // per 32-row block of the wave's 64 rows: 2^-e_ref takes the block's unit back out
    if (g.a_ex) {
        // A wave's 64 rows are TWO 32-row blocks with their own exponents per group (accumulator rows mi 0,1 / mi 2,3).  Every segment
        // is brought to its block's common unit 2^e_ref (e_ref = the block's smallest exponent = its largest-magnitude group) where the
        // fragment is read: the factor 2^(e_ref - e) <= 1 is exact while the product stays a normal f16 and flushes gracefully below
        // (2^-38 of the block's largest value) - no accumulator rescaling, no per-block liveness in the slab loop.
        const int rb = (m0 + wm * 64) / 32;
        int ev0 = lane < g.a_groups ? g.a_ex[(size_t)rb * g.a_groups + lane] : INT_MAX;
        int ev1 = lane < g.a_groups ? g.a_ex[(size_t)(rb + 1) * g.a_groups + lane] : INT_MAX;
        int er0 = ev0, er1 = ev1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            er0 = min(er0, __shfl_xor(er0, o));
            er1 = min(er1, __shfl_xor(er1, o));
        }
        er0 = __builtin_amdgcn_readfirstlane(er0);
        er1 = __builtin_amdgcn_readfirstlane(er1);
        f0 = er0 == INT_MAX ? 0.f : pair_pow2(-er0);
        f1 = er1 == INT_MAX ? 0.f : pair_pow2(-er1);
        auto fac = [&](int ev, int eref, int gi) -> uint32_t {
            const int e = __builtin_amdgcn_readlane(ev, gi);
            if (e == INT_MAX) return 0u;                       // all-zero segment (zeros stored)
            const int d = max(eref - e, -30);                  // <= 0
            const uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)__builtin_bit_cast(float, (unsigned)(127 + d) << 23));
            return bits | (bits << 16);
        };
        DL::run(g.A + (size_t)m0 * g.lda, g.lda, g.B + (size_t)n0 * g.ldb, g.ldb, g.K2, cpg_smem, acc, -1, []() {}, [](int) { return true; },
                [&](int kt, int mi) { return mi < MI / 2 ? fac(ev0, er0, kt / G) : fac(ev1, er1, kt / G); });
    } else {
        DL::run(g.A + (size_t)m0 * g.lda, g.lda, g.B + (size_t)n0 * g.ldb, g.ldb, g.K2, cpg_smem, acc, -1, []() {});
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = acc[mi][ni] * (mi < MI / 2 ? f0 : f1);
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * 64 + ni * 16 + l15;
            const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(m0 + wm * 64 + mi * 16 + 4 * lq + r) * g.ldc + col;
                float v = acc[mi][ni][r] * back + bv;
                if (g.accumulate) v += g.C[o];
                g.C[o] = v;
            }
        }
}

// ---- one GRU decode step on plane images (GRUDecoder.forward_sample's recurrent part, models/decoder.py:86-99, for the per-step
// decode chain of decoders too wide for the whole-loop kernels: CLaSS at config-B / C width).  h_{t-1} arrives as its f16-pair image
// (written by the previous step), W_hh as an image whose rows are ordered so that a 96-row tile = r, z, n of 32 hidden units
// (16 per wave); the product is DlLoop<128, 96, 2, 3> - no conversion in the loop (the round-4 step kernel splits both operands
// in every tile: 0.30 of the pair roof) -, the cell runs in the accumulator layout and the new state goes out as f32 AND as the next
// step's image.  Arithmetic of the cell = gru_step_fwd_kernel's.
struct StepPlanesArgs {
    const uint16_t* hp_in; uint16_t* hp_out;
    const float* h_prev; float* h_out;
    const uint16_t* wimg; const float* b_hh;
    const int32_t* tok; const float* tab; const float* rowc;
    int N, H;
    int rowc_rows;            // rowc holds this many rows, row r reads rowc[r % rowc_rows] (beam-major rows k N + i share sentence i's term)
    const int32_t* origin;    // beam search: [nsent][K] back-pointers - row k nsent + i takes its previous state from row origin[i][k] nsent + i
    int nsent, K;             // (the re-gather of _update_hidden, models/model.py:378-385, folded into the operand loads); null: identity
    const int* wx;            // exponent record of W_hh (behind its image): the image holds W_hh 2^e_w
};
__global__ __launch_bounds__(256) void gru_step_fwd_planes_kernel(StepPlanesArgs g) {
    using DL = DlLoop<128, 96, 2, 3>;
    constexpr int MI = DL::MI, NI = DL::NI;
    static_assert(NI == 3, "a wave holds the r, z, n blocks of its 16 hidden units");
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int H = g.H, m0 = by * 128, j0 = bx * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    float* const tb = cpg_smem + DL::smem_floats() + wave * 256;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rows are beam-major (row = k nsent + i): sentence and beam of the tile's 128 consecutive rows without a division per element
    // (64-bit div / mod per gathered element doubled the launch time) - one scalar division per workgroup, then a conditional subtract
    // (nsent >= 128 here: cpg_gru_step_fwd_planes checks it); ns = the period of rowc as well (rowc_rows; N when rowc has a row per row)
    const unsigned ns = g.origin ? (unsigned)g.nsent : (unsigned)g.rowc_rows;
    const unsigned k0 = (unsigned)m0 / ns, i0 = (unsigned)m0 - k0 * ns;
    auto sent_beam = [&](unsigned d, unsigned& i, unsigned& k) {   // row m0 + d, d < 128
        i = i0 + d;
        k = k0;
        if (i >= ns) { i -= ns; ++k; }
    };
    auto src_row = [&](unsigned d) -> size_t {
        if (!g.origin) return (size_t)m0 + d;
        unsigned i, k;
        sent_beam(d, i, k);
        return (size_t)g.origin[(size_t)i * g.K + k] * ns + i;
    };
    if (g.origin)
        DL::run(g.hp_in, (size_t)2 * H, g.wimg + (size_t)bx * 96 * 2 * H, (size_t)2 * H, 2 * H, cpg_smem, acc, -1, []() {}, [](int) { return true; },
                DlNoScale{}, [&](int i, int r) { return g.hp_in + src_row((unsigned)(32 * i + r)) * 2 * H; });
    else
        DL::run(g.hp_in + (size_t)m0 * 2 * H, (size_t)2 * H, g.wimg + (size_t)bx * 96 * 2 * H, (size_t)2 * H, 2 * H, cpg_smem, acc, -1, []() {});
    const float back = pair_pow2(-weight_exp_from_parts(g.wx));
    const int u = j0 + wn * 16 + l15;
    const float bh_r = g.b_hh[u], bh_z = g.b_hh[H + u], bh_n = g.b_hh[2 * H + u];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        f32x4 hnew;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned d = (unsigned)(wm * 64 + mi * 16 + 4 * lq + r);
            const size_t row = (size_t)m0 + d;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (g.tok) {
                const float* t = g.tab + (size_t)g.tok[row] * 3 * H;
                a0 += t[u]; a1 += t[H + u]; a2 += t[2 * H + u];
            }
            unsigned si, sk;
            sent_beam(d, si, sk);
            if (g.rowc) {
                const float* t = g.rowc + (size_t)(g.rowc_rows == g.N ? row : si) * 3 * H;
                a0 += t[u]; a1 += t[H + u]; a2 += t[2 * H + u];
            }
            const float hp = g.h_prev[(g.origin ? (size_t)g.origin[(size_t)si * g.K + sk] * ns + si : row) * H + u];
            const float hn = acc[mi][2][r] * back + bh_n;
            const float rg = sigmoidf_(a0 + (acc[mi][0][r] * back + bh_r));
            const float zg = sigmoidf_(a1 + (acc[mi][1][r] * back + bh_z));
            const float ng = tanhf(a2 + rg * hn);
            hnew[r] = (1.f - zg) * ng + zg * hp;
            g.h_out[row * H + u] = hnew[r];
        }
        // the next step's A operand: the 16 x 16 block in row layout (lane -> row lane / 4, four consecutive units)
        const f32x4 v = acc_block_to_rows(tb, hnew, lane);
        pair_store4<1>(g.hp_out, (size_t)m0 + wm * 64 + mi * 16 + (lane >> 2), H, j0 + wn * 16 + 4 * (lane & 3), 0, v);
    }
}
// W_hh [3H, H] -> image [3H][2H] x 2^e_w with rows in tile order: image row 96 t + 48 w + 16 gate + i = W_hh row gate H + 32 t + 16 w + i
__global__ void gru_step_w_image_kernel(const float* __restrict__ w, int H, uint16_t* __restrict__ img, const int* __restrict__ wx) {
    const float sc = pair_pow2(weight_exp_from_parts(wx));
    const int q4 = H / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)3 * H * q4) return;
    const int ir = (int)(i / q4), col = (int)(i - (size_t)ir * q4) * 4;
    const int t = ir / 96, rr = ir - 96 * t, wv = rr / 48, gate = (rr - 48 * wv) / 16, ii = rr & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)(gate * H + 32 * t + 16 * wv + ii) * H + col);
    pair_store4<1>(img, (size_t)ir, H, col, 0, v * sc);
}

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ entry points
// Shapes covered by the plane products: whole 128-row / 128-column tiles and 32-unit groups.
CPG_EXPORT int cpg_planes_ok(int R, int K, int N) {
    return (cpg_compute_mode_get() != 1 && R > 0 && R % 128 == 0 && K > 0 && K % 128 == 0 && N > 0 && N % 128 == 0 && (long)R * N >= (1L << 22)) ? 1 : 0;
}
// image of x = [x1 | x2] (x2 optional): R rows of 2 (C1 + C2) f16
CPG_EXPORT size_t cpg_pair_rows_bytes(int R, int C) { return (size_t)R * 2 * C * sizeof(uint16_t); }
// image of a WEIGHT matrix [R, C]: the image + its exponent record (cpg_weight_exp) behind it - the scratch the plane products and
// the plane decode step ask for
CPG_EXPORT size_t cpg_weight_image_bytes(int R, int C) { return align256(cpg_pair_rows_bytes(R, C)) + WX_PARTS * sizeof(int); }
static int* weight_image_wx(void* img, int R, int C) { return (int*)((char*)img + align256(cpg_pair_rows_bytes(R, C))); }
CPG_EXPORT int cpg_pair_rows(const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int R, void* img, void* stream) {
    CPG_CHECK_ARG(x1 && img && R > 0 && C1 > 0 && C1 % 32 == 0 && C2 >= 0 && C2 % 32 == 0 && (C2 == 0 || x2) && ld1 % 4 == 0 && (C2 == 0 || ld2 % 4 == 0));
    CPG_CHECK_ARG(aligned16(x1) && (!x2 || aligned16(x2)) && aligned16(img));
    const size_t n = (size_t)R * ((C1 + C2) / 4);
    hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x1, ld1, C1, x2, ld2, C2, R,
                       (uint16_t*)img, (const int*)nullptr);
    CPG_LAUNCH_CHECK();
    return 0;
}
// y [R, N] (+)= ximg [R][2K] . W[N, K]^T + bias; scratch: cpg_weight_image_bytes(N, K) bytes for W's image and exponent record
CPG_EXPORT int cpg_linear_fwd_planes(const void* ximg, int R, int K, const float* W, int ldw, const float* bias, float* Y, int ldy, int N,
                                     int accumulate, void* scratch, size_t scratch_bytes, void* stream) {
    CPG_CHECK_ARG(ximg && W && Y && scratch && cpg_planes_ok(R, K, N) && ldw % 4 == 0 && aligned16(W) && aligned16(scratch));
    CPG_CHECK_ARG(scratch_bytes >= cpg_weight_image_bytes(N, K));
    hipStream_t s = (hipStream_t)stream;
    int* const wx = weight_image_wx(scratch, N, K);
    int rc = cpg_weight_absmax(W, N, K, ldw, wx, s);
    if (rc) return rc;
    const size_t n = (size_t)N * (K / 4);
    hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ldw, K, (const float*)nullptr, 0, 0, N,
                       (uint16_t*)scratch, (const int*)wx);
    CPG_LAUNCH_CHECK();
    PairNtArgs g{(const uint16_t*)ximg, (size_t)2 * K, nullptr, 0, (const uint16_t*)scratch, (size_t)2 * K, 2 * K, Y, (size_t)ldy, bias, accumulate,
                 wx};
    const size_t smem = DlLoop<128, 128, 2, 3>::smem_floats() * sizeof(float);
    rc = cpg_allow_big_lds((const void*)pair_nt_kernel<3>, (int)smem);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_nt_kernel<3>, dim3(N / 128, R / 128), dim3(256), smem, s, g);
    CPG_LAUNCH_CHECK();
    return 0;
}
// gradient image of dG's G blocks (column offsets off[0..G-1], H columns each): planes [R][2 G H], ex [R/32][H/32], emin [H/32]
CPG_EXPORT size_t cpg_grad_planes_bytes(int R, int H, int G) {
    return align256((size_t)R * 2 * G * H * sizeof(uint16_t)) + align256((size_t)(R / 32) * (H / 32) * sizeof(int)) + align256((size_t)(H / 32) * sizeof(int));
}
static void grad_planes_split(void* gp, int R, int H, int G, uint16_t*& img, int*& ex, int*& emin) {
    img = (uint16_t*)gp;
    ex = (int*)((char*)gp + align256((size_t)R * 2 * G * H * sizeof(uint16_t)));
    emin = (int*)((char*)ex + align256((size_t)(R / 32) * (H / 32) * sizeof(int)));
}
CPG_EXPORT int cpg_grad_planes(const float* dG, int ldg, int R, int H, int G, const int* off, void* gp, void* stream) {
    CPG_CHECK_ARG(dG && gp && off && R > 0 && R % 32 == 0 && H > 0 && H % 32 == 0 && (G == 3 || G == 4) && ldg % 4 == 0 && aligned16(dG) && aligned16(gp));
    GradPlanesArgs a{dG, (size_t)ldg, R, H, G, {0, 0, 0, 0}, nullptr, nullptr, nullptr};
    for (int q = 0; q < G; ++q) {
        CPG_CHECK_ARG(off[q] >= 0 && off[q] % 4 == 0);
        a.off[q] = off[q];
    }
    grad_planes_split(gp, R, H, G, a.img, a.ex, a.emin);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(H / 32, 64)), dim3(64), 0, s, a.emin, H / 32, INT_MAX);
    CPG_LAUNCH_CHECK();
    if (G == 3) hipLaunchKernelGGL(grad_planes_kernel<3>, dim3(H / 32, R / 32), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(grad_planes_kernel<4>, dim3(H / 32, R / 32), dim3(256), 0, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}
// dx [R, In] (+)= dGin . W   (W [G H, In] row-major); scratch: cpg_weight_image_bytes(In, G H) bytes for the image of W^T and W's exponent record
CPG_EXPORT int cpg_linear_bwd_input_planes(const void* gp, int R, int H, int G, const float* W, int ldw, float* dX, int lddx, int In,
                                           int accumulate, void* scratch, size_t scratch_bytes, void* stream) {
    CPG_CHECK_ARG(gp && W && dX && scratch && (G == 3 || G == 4) && cpg_planes_ok(R, G * H, In) && H % 32 == 0 && H / 32 <= 64 && aligned16(scratch));
    CPG_CHECK_ARG(scratch_bytes >= cpg_weight_image_bytes(In, G * H));
    hipStream_t s = (hipStream_t)stream;
    int* const wx = weight_image_wx(scratch, In, G * H);
    int rc = cpg_weight_absmax(W, G * H, In, ldw, wx, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_wT_kernel, dim3(In / 32, G * H / 32), dim3(32, 8), 0, s, W, ldw, G, H, In, (uint16_t*)scratch, (const int*)wx);
    CPG_LAUNCH_CHECK();
    uint16_t* img; int* ex; int* emin;
    grad_planes_split(const_cast<void*>(gp), R, H, G, img, ex, emin);
    PairNtArgs g{img, (size_t)2 * G * H, ex, H / 32, (const uint16_t*)scratch, (size_t)2 * G * H, 2 * G * H, dX, (size_t)lddx, nullptr, accumulate, wx};
    const size_t smem = DlLoop<128, 128, 2, 3>::smem_floats() * sizeof(float);
    const void* k = G == 3 ? (const void*)pair_nt_kernel<3> : (const void*)pair_nt_kernel<4>;
    rc = cpg_allow_big_lds(k, (int)smem);
    if (rc) return rc;
    if (G == 3) hipLaunchKernelGGL(pair_nt_kernel<3>, dim3(In / 128, R / 128), dim3(256), smem, s, g);
    else hipLaunchKernelGGL(pair_nt_kernel<4>, dim3(In / 128, R / 128), dim3(256), smem, s, g);
    CPG_LAUNCH_CHECK();
    return 0;
}
// dW [G H, In] (+)= dGin^T . x   (x as its image [R][2 In]); workspace: cpg_pair_tn_workspace bytes (cpg_linear_bwd_weight_planes_workspace)
CPG_EXPORT size_t cpg_linear_bwd_weight_planes_workspace(int R, int H, int G, int In) { return cpg_pair_tn_workspace(G * H, In, R); }
CPG_EXPORT int cpg_linear_bwd_weight_planes(const void* gp, int R, int H, int G, const void* ximg, int In, float* dW, int lddw, int accumulate,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(gp && ximg && dW && workspace && (G == 3 || G == 4) && cpg_planes_ok(R, G * H, In));
    uint16_t* img; int* ex; int* emin;
    grad_planes_split(const_cast<void*>(gp), R, H, G, img, ex, emin);
    return cpg_pair_tn(img, (size_t)2 * G * H, ex, emin, H / 32, G, (const uint16_t*)ximg, (size_t)2 * In, dW, lddw, G * H, In, R, accumulate,
                       (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- GRU decode step on plane images (see gru_step_fwd_planes_kernel).  cpg_gru_step_planes_ok: 1 where the form covers N rows at width H
// (f32-grade mode, N % 128 == 0, H % 128 == 0, enough rows to amortise the images).  The W_hh image (cpg_weight_image_bytes(3H, H) bytes: image + exponent record) is
// built once per decode by cpg_gru_step_w_image; the state travels as (h f32 [N,H], image [N][2H]) - cpg_pair_rows makes the first image.
CPG_EXPORT int cpg_gru_step_planes_ok(int N, int H) {
    return (cpg_compute_mode_get() != 1 && N >= 1024 && N % 128 == 0 && H >= 128 && H % 128 == 0) ? 1 : 0;
}
CPG_EXPORT int cpg_gru_step_w_image(const float* w_hh, int H, void* wimg, void* stream) {
    CPG_CHECK_ARG(w_hh && wimg && H > 0 && H % 128 == 0 && aligned16(w_hh) && aligned16(wimg));
    int* const wx = weight_image_wx(wimg, 3 * H, H);
    const int rc = cpg_weight_absmax(w_hh, 3 * H, H, H, wx, (hipStream_t)stream);
    if (rc) return rc;
    const size_t n = (size_t)3 * H * (H / 4);
    hipLaunchKernelGGL(gru_step_w_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_hh, H, (uint16_t*)wimg,
                       (const int*)wx);
    CPG_LAUNCH_CHECK();
    return 0;
}
CPG_EXPORT int cpg_gru_step_fwd_planes(int N, int H, const void* wimg, const float* b_hh, const int32_t* tok, const float* tab, const float* rowc,
                                       int rowc_rows, const float* h_prev, const void* hp_in, const int32_t* origin, int nsent, int K,
                                       float* h_out, void* hp_out, void* stream) {
    CPG_CHECK_ARG(wimg && b_hh && h_prev && hp_in && h_out && hp_out && h_prev != h_out && hp_in != hp_out && cpg_gru_step_planes_ok(N, H));
    CPG_CHECK_ARG((!rowc || (rowc_rows >= 128 && N % rowc_rows == 0)) && (!origin || (nsent >= 128 && K > 0 && (long)nsent * K == N)));
    CPG_CHECK_ARG(!origin || !rowc || rowc_rows == nsent || rowc_rows == N);
    StepPlanesArgs g{(const uint16_t*)hp_in, (uint16_t*)hp_out, h_prev, h_out, (const uint16_t*)wimg, b_hh, tok, tab, rowc, N, H,
                     rowc ? rowc_rows : N, origin, nsent, K, weight_image_wx(const_cast<void*>(wimg), 3 * H, H)};
    const size_t smem = (DlLoop<128, 96, 2, 3>::smem_floats() + 4 * 256) * sizeof(float);
    int rc = cpg_allow_big_lds((const void*)gru_step_fwd_planes_kernel, (int)smem);
    if (rc) return rc;
    hipLaunchKernelGGL(gru_step_fwd_planes_kernel, dim3(H / 32, N / 128), dim3(256), smem, (hipStream_t)stream, g);
    CPG_LAUNCH_CHECK();
    return 0;
}
