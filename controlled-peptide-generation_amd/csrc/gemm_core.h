// MFMA tile engine for gfx950, f32 in / f32 out.  Two product forms share the staging, tiling and epilogue conventions:
//   SPLIT = 0  v_mfma_f32_16x16x4_f32 (exact f32, 64 FLOP/clk/SIMD) on f32 LDS images - described first;
//   SPLIT = 7  six v_mfma_f32_16x16x32_bf16 on operands split three ways into bf16 when a slab is stored to LDS
//              (f32-grade results, 2.2x less matrix-pipe time, and - unlike the f32 MFMA - able to overlap VALU work);
//              see the comment above MainLoop.
//
// One 256-thread workgroup (4 wave64) computes a BM x BN tile of  C = A * B  with the contraction
// dimension K streamed through LDS in BK-deep slabs, double buffered (global -> registers while the
// MFMAs of the current slab run, registers -> the other LDS buffer, one barrier per slab).
//
// Each operand can be stored either with K contiguous ("KC": rows of h / rows of W, the natural torch
// layouts) or with its free dimension contiguous ("XC": transposed use, e.g. dW = dY^T X).  The LDS image
// is chosen per layout so that the per-lane fragment reads (ds_read_b32) are bank-conflict free:
//   KC -> [X][BK+4]   rows 16-byte aligned: each lane fetches 4 consecutive k of its row with ONE ds_read_b128 (the
//                     contraction index is permuted accordingly, see MainLoop::PERM), staged with ds_write_b128
//   XC -> [BK][X+16]  ds_read_b32: lanes 0..15 consecutive banks, lanes 16..31 shifted by 16
// MFMA fragment mapping (cdna guide section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C/D: col = l&15, row = (l>>4)*4 + reg.
//
// The N side supports a "segmented" row map so one tile can hold the r,z,n gate rows of the same hidden
// units: n_local -> jblk = n_local/(16*NSEG), seg = (n_local/16)%NSEG, j = j0 + jblk*16 + n_local%16,
// global row/col = seg*seg_stride + j, valid iff j < seg_len.  NSEG=1 is the plain map.
#pragma once
#include "cpg_common.h"
#include <limits.h>

// Diagnostic builds only (tools/ablate.sh): -DCPG_ABLATE=<mask> removes one phase of the slab loop after the first slab so its
// cost can be measured in isolation.  1: no global loads / LDS writes   2: no LDS fragment reads   4: no barrier
// 8: no MFMA (one VALU fma per fragment pair instead).
// Results of such builds are wrong by construction; the shipped library is built without the macro.
#ifndef CPG_ABLATE
#define CPG_ABLATE 0
#endif
#ifndef CPG_SCHED
#define CPG_SCHED 0
#endif
#ifndef CPG_KC_PAD
#define CPG_KC_PAD 8
#endif
#ifndef CPG_XC_PERM_PAD
#define CPG_XC_PERM_PAD 4   // pad of a transposed-use image whose partner operand is K-contiguous (see TileCfg::xc_pad)
#endif
#ifndef CPG_LOOP_UNROLL2
#define CPG_LOOP_UNROLL2 1
#endif

struct OpA {
    const float* p;
    int ld;
    int m0;               // first global M index of this tile
    int M;                // bound on the M index
    const uint8_t* mask;  // optional keep-mask with the same indexing as p (value = p * (mask ? mscale : 0))
    float mscale;
    int pairs = 0;        // 1: 8-byte pairs of the scalar (non-16-byte) staging path are aligned and never straddle a bound
    int bf16 = 0;         // 1: p points at bf16 elements (ld, offsets in elements) - transposed-use operands on the 16-byte staging path
                          // only (the dW_hh product on bf16 gate gradients): four elements = one 8-byte load, widened in registers
    const int* exps = nullptr;   // SPLIT 8 (f16 pairs): power-of-two exponent of M column m = exps[(m % exps_mod) / 32]; INT_MAX: 0
    int exps_mod = 1;
};

struct OpB {
    const float* p;
    int ld;
    int j0;          // first j of this tile
    int seg_len;     // bound on j
    int seg_stride;  // row/col offset between segments
    const uint8_t* mask;
    float mscale;
    int pairs = 0;   // as OpA::pairs
    float pscale = 1.f;   // SPLIT 8 (f16 pairs): power of two the operand is multiplied by before the split (weights: 2^e_w from
                          // weight_exp_from_parts below); the caller takes it back out of the accumulators
};

template <int BM_, int BN_, int BK_, int WM_, int WN_, int NSEG_, int NT_ = 256>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NSEG = NSEG_;
    static constexpr int NT = NT_;  // 256 (one wave per SIMD per workgroup) or 512 (two: the 256x128 dW tiles)
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int MI = WTM / 16, NI = WTN / 16;
    static constexpr int AV = BM * BK / 4 / NT;  // float4 staging registers per thread
    static constexpr int BV = BN * BK / 4 / NT;
    static_assert(WM * WN * 64 == NT, "one 16x16-block grid position per wave");
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tile must be a multiple of 16");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging must divide evenly");
    static_assert(WTN % (16 * NSEG) == 0, "a wave must own whole segment groups");
    // K-contiguous images are read 16 bytes per lane (lane (x = l&15, q = l>>4) -> row x, words 4q..4q+3): ds_read_b128 is
    // served in four fixed 16-lane groups mixing q and x ({0-3,12-15,20-27}, ...), conflict-free iff the row stride is
    // 8, 24, 40 or 56 words modulo 64 (MI355X_MICROARCH.md LDS table; PMC: BK + 4 = 36 words cost 38 % of the backward step's
    // LDS cycles in bank conflicts, BK + 8 = 40 none)
    static constexpr int KPAD = CPG_KC_PAD;
    // Transposed-use ("XC", [k][x]) images are read one dword per lane: lane (x = l&15, q = l>>4) reads row k(q, j).  With both
    // operands XC the four lane groups read CONSECUTIVE k rows (k = 16h + 4j + q) and a row stride of X + 16 words puts them on
    // distinct bank quarters; when the OTHER operand is K-contiguous the contraction order is permuted (k = 16h + 4q + j, see
    // MainLoop::PERM) and the groups read rows 4 apart: stride X + 16 then maps all four onto the SAME 16 banks (4-way conflict
    // on every fragment read - PMC: 29 % of the backward step's LDS cycles), stride X + 4 (4 * stride = 16 mod 64) spreads them.
    template <bool OTHER_KC>
    static constexpr int xc_pad() { return OTHER_KC ? CPG_XC_PERM_PAD : 16; }
    template <bool KC, bool OTHER_KC>
    static constexpr int lda() { return KC ? BK + KPAD : BM + xc_pad<OTHER_KC>(); }
    template <bool KC, bool OTHER_KC>
    static constexpr int ldb() { return KC ? BK + KPAD : BN + xc_pad<OTHER_KC>(); }
    template <bool KC, bool OTHER_KC>
    static constexpr int a_elems() { return KC ? BM * (BK + KPAD) : BK * (BM + xc_pad<OTHER_KC>()); }
    template <bool KC, bool OTHER_KC>
    static constexpr int b_elems() { return KC ? BN * (BK + KPAD) : BK * (BN + xc_pad<OTHER_KC>()); }
    template <bool AKC, bool BKC>
    static constexpr int smem_floats() { return 2 * (a_elems<AKC, BKC>() + b_elems<BKC, AKC>()); }
};

// XCD-aware tile order (MI355X: 8 XCDs with private 4 MiB L2s; workgroup `id` is observed to run on XCD id % 8 - used for
// speed only, any placement is correct).  The default x-fastest order puts every N-tile column on its own XCD, so each
// XCD streams the WHOLE M-side operand (measured on the dW_hh product: FETCH_SIZE 1.29 GB raw vs 0.52 GB algorithmic).
// Remapped order: the gx workgroups that share one (y,z) - the same M-side rows / K-chunk - run back to back on ONE XCD,
// and each XCD owns a contiguous 1/8 of the (y,z) combinations.
template <bool SEQ8 = false>
__device__ __forceinline__ void xcd_tile_order(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int combos = gy * gz;
    if (SEQ8 && combos % 8 != 0 && (gx * combos) % 8 == 0) {
        // each XCD takes a contiguous 1/8 of the x-fastest tile sequence (the bf16-mode dW_hh product, which is bound by its
        // operand loads: 284 -> 249 us; see the note below for the f32-grade launch)
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int t = (lin & 7) * (gx * combos / 8) + (lin >> 3);
        bx = t % gx;
        by = (t / gx) % gy;
        bz = (t / gx) / gy;
        return;
    }
    if (combos % 8 != 0) {
        // (Giving each XCD a contiguous 1/8 of the x-fastest tile sequence here as well was measured on the 4 x 6 x 10 grid of the
        // dW_hh product: L2 misses drop 3x, the launch takes 583 us instead of 430 - the four workgroups that share an A panel
        // then ask ONE L2 for the same lines at the same time.)
        bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
        return;
    }
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = lin & 7, idx = lin >> 3;
    bx = idx % gx;
    const int combo = xcd * (combos / 8) + idx / gx;
    by = combo % gy;
    bz = combo / gy;
}

template <int NSEG>
__device__ __forceinline__ void bmap(const OpB& b, int n_local, int& idx, int& j) {
    const int jblk = n_local / (16 * NSEG);
    const int seg = (n_local / 16) % NSEG;
    j = b.j0 + jblk * 16 + (n_local & 15);
    idx = seg * b.seg_stride + j;
}

__device__ __forceinline__ float4 finish4(float4 v, unsigned okbits, bool has_mask, uchar4 k, float ms) {
    if (has_mask) {
        v.x = k.x ? v.x * ms : 0.f;
        v.y = k.y ? v.y * ms : 0.f;
        v.z = k.z ? v.z * ms : 0.f;
        v.w = k.w ? v.w * ms : 0.f;
    }
    if (okbits != 0xFu) {
        v.x = (okbits & 1u) ? v.x : 0.f;
        v.y = (okbits & 2u) ? v.y : 0.f;
        v.z = (okbits & 4u) ? v.z : 0.f;
        v.w = (okbits & 8u) ? v.w : 0.f;
    }
    return v;
}

// bf16 split-operand products (SPLIT = 7).  Measured on MI355X (tools/micro/mfma_valu_overlap.hip): the f32 MFMA
// (v_mfma_f32_16x16x4_f32, 32 cycles) cannot overlap with VALU work at all - {1 MFMA + n VALU} costs the SUM of the two,
// it runs on the same f32 lanes - while v_mfma_f32_16x16x32_bf16 (20.5 cycles for 8x the contraction depth) does.
// x = x0 + x1 + x2 with x_i bf16 (residual <= 2^-27 |x|); a.b ~= a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 (dropped terms
// <= 2^-26 |a.b|), products exact in f32, accumulated in the MFMA's f32 accumulator: f32-grade results from six bf16
// MFMAs (123 cycles) in place of eight f32 MFMAs (268 cycles) per 16x16x32 block, and the conversion VALU work overlaps.
typedef __bf16 cpg_bf16x8 __attribute__((ext_vector_type(8)));

// one (even k, odd k) pair -> one 32-bit word per plane (low half = even k): 3 packed converts + 4 unpacks + 4 subtractions.
// (Written with the instruction itself: from scalar __bf16 casts hipcc emits one v_cvt_pk per ELEMENT plus moves - measured
// 312 VALU instructions per slab and wave on the dW_hh product, VALU-bound.)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint16_t bf16_bits(float x) { return (uint16_t)(cvt_pk_bf16(x, 0.f) & 0xffffu); }   // RNE
__device__ __forceinline__ void split3_pair(float lo, float hi, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    w0 = cvt_pk_bf16(lo, hi);
    const float l1 = lo - __uint_as_float(w0 << 16), h1 = hi - __uint_as_float(w0 & 0xffff0000u);
    w1 = cvt_pk_bf16(l1, h1);
    const float l2 = l1 - __uint_as_float(w1 << 16), h2 = h1 - __uint_as_float(w1 & 0xffff0000u);
    w2 = cvt_pk_bf16(l2, h2);
}

// f16 pair split (the persistent forward kernels' f32-grade engine since round 4): x = x0 + x1 with x_i f16 (11 significand
// bits each, round to nearest even; residual <= 2^-23 |x| while x1 is a normal f16, <= 2^-25 absolute below that - the MFMA keeps
// f16 subnormal inputs, measured on MI355X: tools/micro/f16_subnormal.hip); a.b ~= a1b0 + a0b1 + a0b0 (dropped a1b1 <= 2^-22
// |a.b|): THREE v_mfma_f32_16x16x32_f16 per block and 4 bytes per element in place of six MFMAs and 6 bytes.  f16 has five
// exponent bits: an operand whose magnitudes are not O(1) (weights) is multiplied by a power of two first (exact), the
// accumulator by its inverse.
typedef _Float16 cpg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cpg_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 cpg_f16x2 __attribute__((ext_vector_type(2)));
typedef float cpg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pair(float lo, float hi, uint32_t& w0, uint32_t& w1) {
    const cpg_f16x2 a = __builtin_convertvector(cpg_f32x2{lo, hi}, cpg_f16x2);   // v_cvt_pk_f16_f32
    const cpg_f32x2 b = __builtin_convertvector(a, cpg_f32x2);
    const cpg_f16x2 c = __builtin_convertvector(cpg_f32x2{lo - b[0], hi - b[1]}, cpg_f16x2);
    w0 = __builtin_bit_cast(uint32_t, a);
    w1 = __builtin_bit_cast(uint32_t, c);
}

__device__ __forceinline__ float pair_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }   // |e| <= 126

// ---- the power of two a WEIGHT matrix is multiplied by before it is split into f16 pairs (round 6; rounds 4-5 used a fixed 2^8, which
// turned a weight of magnitude >= 256 into an f16 infinity).  Chosen per matrix from its largest magnitude, on the device:
// cpg_weight_absmax (gemm.hip) leaves WX_PARTS partial maxima (float bits) in `wx`; every kernel that images or consumes the matrix
// derives the same exponent e from them - max|W| 2^e in [2^13, 2^14), the convention of the gradient images and of the persistent
// forward's W_hh slices - so any finite f32 weight is representable.  The pair keeps 22 significand bits for weights down to 2^-16 of
// the matrix' largest magnitude (the low half stays a normal f16) and an absolute precision of 2^-25 2^-e = 2^-39 max|W| below that -
// more than f32 needs for any matrix whose entries span less than five decimal orders (rounds 4-5: 2^-11 .. 255 at the fixed scale).  An all-zero matrix takes e = 0; an infinity among the weights goes in unscaled and reaches
// the result as it is (as in f32 arithmetic).
constexpr int WX_PARTS = 32;
__device__ __forceinline__ int weight_exp_of(float vmax) {
    if (!(vmax > 0.f) || vmax >= 3.0e38f) return 0;
    int fe = 0;
    (void)frexpf(vmax, &fe);
    return max(-100, min(100, 14 - fe));
}
// every lane of a wave gets the matrix' exponent (wave-uniform by construction: all lanes reduce the same WX_PARTS values)
__device__ __forceinline__ int weight_exp_from_parts(const int* __restrict__ wx) {
    float v = __builtin_bit_cast(float, wx[__lane_id() & (WX_PARTS - 1)]);
#pragma unroll
    for (int o = WX_PARTS / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(weight_exp_of(v));
}
// wx[0 .. WX_PARTS) = float bits of partial maxima of |w| over a [rows, cols] matrix of row stride ld (one launch)
int cpg_weight_absmax(const float* w, int rows, int cols, int ld, int* wx, hipStream_t s);


// SPLIT = 0: exact-f32 MFMA on f32 LDS images (the layouts described at the top of this file).
// (Splitting per wave at fragment-read time on the f32 LDS images was tried first: the conversion is then repeated by
//  every wave that shares an operand - VALU-bound; not kept.)
// SPLIT = 7: operands are split ONCE, when a slab is stored to LDS.  LDS then holds three bf16 planes per operand:
//              KC -> plane[X][16 data + 4 pad words]   a lane's 8 consecutive k = one ds_read_b128
//              XC -> plane[BK/2][X + 4 words]          word = (k even, k odd) of one x; a lane's 8 k = four ds_read_b32;
//                                                      a thread stages rows k and k+1 of the same 4 columns and writes
//                                                      the four packed words with one ds_write_b128 per plane
//            MFMA slot i of lane group q contracts k = 8q + i for both layouts.
// (Pre-splitting the weight operand once per call instead of in every tile was tried and measured: no gain - 37.5 vs 37.1 us
// per forward step - the conversion VALU work is not what bounds the small-tile step kernels.)
// A_BF16: the A operand holds bf16 elements in memory (transposed-use operands on the 16-byte staging path only: the dW_hh product
// on bf16 gate gradients) - four elements are one 8-byte load, widened exactly in registers.  Compile-time: a run-time test in
// front of every staging load cost the bf16-mode dW_hh launch 200 us (round 4).
template <class TC, bool A_KC, bool B_KC, bool AVEC, bool BVEC, bool MASKS = false, int SPLIT = 0, bool A_BF16 = false>
struct MainLoop {
    static_assert(!A_BF16 || (!A_KC && AVEC && !MASKS), "bf16 A operands: transposed use, 16-byte staging path, no mask");
    static constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK;
    static constexpr int LDA = TC::template lda<A_KC, B_KC>();
    static constexpr int LDB = TC::template ldb<B_KC, A_KC>();
    static constexpr int ASZ = TC::template a_elems<A_KC, B_KC>();
    static constexpr int BSZ = TC::template b_elems<B_KC, A_KC>();
    // K-index of MFMA k-step s (0..BK/4-1), lane group q (0..3).  When an operand is K-contiguous its fragments are read
    // 16 bytes at a time (4 consecutive k per lane), so within each 16-deep half the contraction index is permuted:
    // step j of half h contracts k = 16h + 4q + j.  Any bijection works as long as A and B agree.
    static constexpr bool PERM = A_KC || B_KC;
    static_assert(BK % 16 == 0, "slab depth must be a multiple of 16");
    static constexpr int NH = BK / 16;  // 16-deep halves per slab
    // staging-time split (SPLIT == 7): plane geometry in 32-bit words
    static constexpr int KCW = 20;
    static constexpr int SXA = BM + 4, SXB = BN + 4;
    // TRX: both operands transposed-use (dW = dY^T X).  Each plane is then a row of [32 k][16 x] bf16 subtiles (1 KB + 32 B
    // pad): the staging thread writes the four columns of one k it loaded as ONE ds_write_b64 per plane, and a fragment is
    // two ds_read_b64_tr_b16 (the LDS transposes 4 k x 16 x blocks on the way out; tools/micro/tr_read_probe.hip pins the
    // lane/slot mapping) over a contiguous subtile - instead of four ds_read_b32 per plane from a [k-pair][x] image, which
    // at two waves per SIMD ran the LDS at a fraction of its rate and bounded the dW_hh product.
    // Slot i of lane group q contracts k = 4q + i (i < 4) / 16 + 4q + (i - 4): any bijection works when A and B agree.
    static constexpr bool TRX = SPLIT != 0 && !A_KC && !B_KC;
    // SPLIT = 1 ("bf16 compute mode", cfg.hw.dtype = 'bf16'): the same plane engine with ONE plane - operands rounded to
    // bf16 (round to nearest even) when the slab is stored, one MFMA per block, f32 accumulation.  NOT f32-grade: its own
    // tests state agreement thresholds instead of the 1e-4 bars.
    // SPLIT = 8: f32-grade on f16 PAIRS split at LDS-store time (two planes, three v_mfma_f32_16x16x32_f16 per block; the
    // transposed-use dW_hh product whose dY columns carry a power-of-two scale - OpA::exps - that the caller takes back out)
    static constexpr int NP = SPLIT == 7 ? 3 : SPLIT == 8 ? 2 : 1;
    static constexpr int TRW = 264;  // words per subtile: 32 rows x 8 words + 8 pad (consecutive subtiles on distinct banks)
    static constexpr int APL = A_KC ? BM * KCW : (TRX ? (BM / 16) * TRW : (BK / 2) * SXA);
    static constexpr int BPL = B_KC ? BN * KCW : (TRX ? (BN / 16) * TRW : (BK / 2) * SXB);
    static constexpr int ASZ7 = NP * APL, BSZ7 = NP * BPL;
    static_assert(SPLIT == 0 || SPLIT == 7 || SPLIT == 1 || SPLIT == 8,
                  "0: exact f32; 7: f32-grade, three bf16 planes split at LDS-store time, six MFMAs; 1: bf16 compute mode, one plane, one MFMA; 8: f16 pairs");
    static_assert(SPLIT != 8 || !MASKS, "f16 pairs: no keep-masks");
    static_assert(SPLIT != 8 || !TRX || AVEC, "f16 pairs, transposed use: the 16-byte staging path (column exponents per vector)");
    static_assert(SPLIT == 0 || BK == 32, "plane products are written for 32-deep slabs");
    static_assert(SPLIT == 0 || TRX || ((A_KC || TC::AV % 2 == 0) && (B_KC || TC::BV % 2 == 0)), "XC staging works on k-row pairs");
    // SB ("single buffer", the 128 x 128 transposed-use tile on the plane engine): ONE LDS image per operand, two barriers per
    // slab (product | conversion + store), so that TWO workgroups fit a CU (48 KB each) and run those phases against each other -
    // the double-buffered 256 x 128 tile has one workgroup per CU whose eight waves all stall on the same loads.
    static constexpr bool SB = SPLIT != 0 && !A_KC && !B_KC && TC::BM == 128 && TC::BN == 128;
    static constexpr size_t smem_bytes() {
        return SPLIT != 0 ? (size_t)(SB ? 1 : 2) * (ASZ7 + BSZ7) * 4 : (size_t)2 * (ASZ + BSZ) * sizeof(float);
    }
    // staging vector i of this thread -> (k row inside the slab, column quad) for an XC operand X columns wide
    template <int BX>
    __device__ static __forceinline__ void xc_index(int i, int& kk, int& xq) {
        if (SPLIT != 0 && !TRX) {
            const int u = threadIdx.x + (i >> 1) * TC::NT;
            xq = u % (BX / 4);
            kk = 2 * (u / (BX / 4)) + (i & 1);
        } else {
            const int v = threadIdx.x + i * TC::NT;
            kk = v / (BX / 4);
            xq = v % (BX / 4);
        }
    }

    struct Stage {  // one K-slab of both operands in flight in registers
        float4 a[TC::AV], b[TC::BV];
        uchar4 am[TC::AV], bm[TC::BV];
        unsigned aok[TC::AV], bok[TC::BV];
    };
    struct Plan {  // per-thread, slab-invariant part of the staging addresses (hoisted out of the slab loop)
        size_t a[TC::AV], b[TC::BV];  // element offset of the vector at k0 = 0 (clamped to a valid row / column)
        int an[TC::AV], bn[TC::BV];   // KC: 4*kq (k offset inside the slab) ; XC: number of valid columns (0..4)
        bool aok[TC::AV], bok[TC::BV];  // KC: row in range ; XC: unused
        float asc[SPLIT == 8 ? TC::AV : 1];   // SPLIT 8: power-of-two scale of this thread's four A columns (one 32-column group)
    };

    __device__ static __forceinline__ void plan(const OpA& a, const OpB& b, Plan& pl) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            if (A_KC) {
                const int row = v / (BK / 4), kq = v % (BK / 4);
                const int gm = a.m0 + row;
                pl.aok[i] = gm < a.M;
                pl.an[i] = 4 * kq;
                pl.a[i] = (size_t)(pl.aok[i] ? gm : 0) * a.ld + 4 * kq;
            } else {
                int kk, mq;
                xc_index<BM>(i, kk, mq);
                const int gm = a.m0 + 4 * mq;
                const int nc = min(max(a.M - gm, 0), 4);
                pl.aok[i] = true;
                pl.an[i] = nc;
                pl.a[i] = (size_t)kk * a.ld + (nc > 0 ? gm : 0);
                if constexpr (SPLIT == 8) {
                    const int e = (a.exps && nc > 0) ? a.exps[(gm % a.exps_mod) / 32] : 0;
                    pl.asc[i] = __builtin_bit_cast(float, (unsigned)(127 + (e == INT_MAX ? 0 : e)) << 23);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            int idx, j;
            if (B_KC) {
                const int nl = v / (BK / 4), kq = v % (BK / 4);
                bmap<TC::NSEG>(b, nl, idx, j);
                pl.bok[i] = j < b.seg_len;
                pl.bn[i] = 4 * kq;
                pl.b[i] = (size_t)(pl.bok[i] ? idx : 0) * b.ld + 4 * kq;
            } else {
                int kk, nq;
                xc_index<BN>(i, kk, nq);
                bmap<TC::NSEG>(b, 4 * nq, idx, j);
                const int nc = min(max(b.seg_len - j, 0), 4);
                pl.bok[i] = true;
                pl.bn[i] = nc;
                pl.b[i] = (size_t)kk * b.ld + (nc > 0 ? idx : 0);
            }
        }
    }

    // Unconditional (clamped) loads; validity bits are applied at LDS-store time, after the MFMA phase they overlap with.
    template <bool KC, bool VEC>
    // (pairs: operands whose rows are only 8-byte aligned - z_dim = 510 - go as two 8-byte loads instead of four dwords with
    //  four predicates: half the memory instructions and a third of the VALU work of the scalar path, which the exact-f32
    //  MFMA cannot overlap with)
    __device__ static __forceinline__ float4 fetch(const float* p, const uint8_t* mask, size_t base, int n, bool rok, int ld,
                                                   int k0, int K, int kk_xc, unsigned& okbits, uchar4& mk, bool pairs, bool src_bf16 = false,
                                                   bool raw = false) {
        if (KC) {
            const int nk = K - (k0 + n);  // n = 4*kq
            if (VEC) {
                const bool ok = rok && nk >= 4;
                okbits = ok ? 0xFu : 0u;
                const size_t o = ok ? base + k0 : 0;
                if (MASKS && mask) mk = *reinterpret_cast<const uchar4*>(mask + o);
                return *reinterpret_cast<const float4*>(p + o);
            }
            if (pairs && !(MASKS && mask)) {
                const bool ok0 = rok && nk >= 2, ok1 = rok && nk >= 4;
                okbits = (ok0 ? 3u : 0u) | (ok1 ? 12u : 0u);
                const float2 lo = *reinterpret_cast<const float2*>(p + (ok0 ? base + k0 : 0));
                const float2 hi = *reinterpret_cast<const float2*>(p + (ok1 ? base + k0 + 2 : 0));
                mk = make_uchar4(1, 1, 1, 1);
                return make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            float t[4];
            unsigned char m4[4] = {1, 1, 1, 1};
            okbits = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = rok && c < nk;
                okbits |= ok ? (1u << c) : 0u;
                const size_t o = ok ? base + k0 + c : 0;
                t[c] = p[o];
                if (MASKS && mask) m4[c] = mask[o];
            }
            mk = make_uchar4(m4[0], m4[1], m4[2], m4[3]);
            return make_float4(t[0], t[1], t[2], t[3]);
        } else {
            const bool kok = (k0 + kk_xc) < K;  // n = number of valid columns
            if (VEC) {
                const bool ok = kok && n >= 4;
                okbits = ok ? 0xFu : 0u;
                const size_t o = ok ? base + (size_t)k0 * ld : 0;
                if (MASKS && mask) mk = *reinterpret_cast<const uchar4*>(mask + o);
                if (src_bf16) {   // four bf16 = 8 bytes; widening is exact, and the bf16-mode store rounds them back to themselves
                    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + o);
                    // raw: the one-plane (bf16 mode) store takes the two packed pairs as they are - no widening, no re-rounding
                    if (raw) return make_float4(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, w.y), 0.f, 0.f);
                    return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
                }
                return *reinterpret_cast<const float4*>(p + o);
            }
            if (pairs && !(MASKS && mask)) {
                const bool ok0 = kok && n >= 2, ok1 = kok && n >= 4;
                okbits = (ok0 ? 3u : 0u) | (ok1 ? 12u : 0u);
                const size_t o = base + (size_t)k0 * ld;
                const float2 lo = *reinterpret_cast<const float2*>(p + (ok0 ? o : 0));
                const float2 hi = *reinterpret_cast<const float2*>(p + (ok1 ? o + 2 : 0));
                mk = make_uchar4(1, 1, 1, 1);
                return make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            float t[4];
            unsigned char m4[4] = {1, 1, 1, 1};
            okbits = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = kok && c < n;
                okbits |= ok ? (1u << c) : 0u;
                const size_t o = ok ? base + (size_t)k0 * ld + c : 0;
                t[c] = p[o];
                if (MASKS && mask) m4[c] = mask[o];
            }
            mk = make_uchar4(m4[0], m4[1], m4[2], m4[3]);
            return make_float4(t[0], t[1], t[2], t[3]);
        }
    }

    __device__ static __forceinline__ void gload(const OpA& a, const OpB& b, const Plan& pl, int k0, int K, Stage& st) {
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            int kk, xq_;
            xc_index<BM>(i, kk, xq_);
            st.a[i] = fetch<A_KC, AVEC>(a.p, a.mask, pl.a[i], pl.an[i], pl.aok[i], a.ld, k0, K, kk, st.aok[i], st.am[i], a.pairs != 0, A_BF16, A_BF16 && NP == 1);
        }
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            int kk, xq_;
            xc_index<BN>(i, kk, xq_);
            st.b[i] = fetch<B_KC, BVEC>(b.p, b.mask, pl.b[i], pl.bn[i], pl.bok[i], b.ld, k0, K, kk, st.bok[i], st.bm[i], b.pairs != 0);
        }
    }

    __device__ static __forceinline__ void sstore(const OpA& a, const OpB& b, float* As, float* Bs, const Stage& st) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            const float4 r = finish4(st.a[i], st.aok[i], MASKS && a.mask != nullptr, st.am[i], a.mscale);
            const int off = A_KC ? (v / (BK / 4)) * LDA + 4 * (v % (BK / 4)) : (v / (BM / 4)) * LDA + 4 * (v % (BM / 4));
            *reinterpret_cast<float4*>(As + off) = r;
        }
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            const float4 r = finish4(st.b[i], st.bok[i], MASKS && b.mask != nullptr, st.bm[i], b.mscale);
            const int off = B_KC ? (v / (BK / 4)) * LDB + 4 * (v % (BK / 4)) : (v / (BN / 4)) * LDB + 4 * (v % (BN / 4));
            *reinterpret_cast<float4*>(Bs + off) = r;
        }
    }

    // ---- SPLIT != 0: split / round at LDS-store time ------------------------------------------------------------------
    __device__ static __forceinline__ void split_pair(float lo, float hi, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
        if (NP == 3) {
            split3_pair(lo, hi, w0, w1, w2);
        } else if (NP == 2) {
            split2h_pair(lo, hi, w0, w1);
            w2 = 0u;
        } else {
            w0 = cvt_pk_bf16(lo, hi);
            w1 = w2 = 0u;
        }
    }
    // RAW: r[i].x / .y already ARE the two packed bf16 pairs of the four elements (bf16 operand in memory, one-plane mode)
    template <bool KC, int BX, int NV, int PLW, int SX, bool RAW = false>
    __device__ static __forceinline__ void sstore7_op(uint32_t* dst, const float4 (&r)[NV]) {
        const int tid = threadIdx.x;
        if (KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + i * TC::NT, row = v / (BK / 4), kq = v % (BK / 4);
                uint32_t a0, a1, a2, b0, b1, b2;
                split_pair(r[i].x, r[i].y, a0, a1, a2);
                split_pair(r[i].z, r[i].w, b0, b1, b2);
                uint32_t* q = dst + row * KCW + 2 * kq;
                *reinterpret_cast<uint2*>(q) = make_uint2(a0, b0);
                if (NP >= 2) *reinterpret_cast<uint2*>(q + PLW) = make_uint2(a1, b1);
                if (NP == 3) *reinterpret_cast<uint2*>(q + 2 * PLW) = make_uint2(a2, b2);
            }
        } else if (TRX) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + i * TC::NT, kk = v / (BX / 4), xq = v % (BX / 4);
                uint32_t a0, a1 = 0, a2 = 0, b0, b1 = 0, b2 = 0;
                if constexpr (RAW) {
                    a0 = __builtin_bit_cast(uint32_t, r[i].x);
                    b0 = __builtin_bit_cast(uint32_t, r[i].y);
                } else {
                    split_pair(r[i].x, r[i].y, a0, a1, a2);
                    split_pair(r[i].z, r[i].w, b0, b1, b2);
                }
                uint32_t* q = dst + (xq >> 2) * TRW + kk * 8 + (xq & 3) * 2;
                *reinterpret_cast<uint2*>(q) = make_uint2(a0, b0);
                if (NP >= 2) *reinterpret_cast<uint2*>(q + PLW) = make_uint2(a1, b1);
                if (NP == 3) *reinterpret_cast<uint2*>(q + 2 * PLW) = make_uint2(a2, b2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                const int u = tid + j * TC::NT, xq = u % (BX / 4), p = u / (BX / 4);
                const float4 e = r[2 * j], o = r[2 * j + 1];
                uint32_t w0[4], w1[4], w2[4];
                split_pair(e.x, o.x, w0[0], w1[0], w2[0]);
                split_pair(e.y, o.y, w0[1], w1[1], w2[1]);
                split_pair(e.z, o.z, w0[2], w1[2], w2[2]);
                split_pair(e.w, o.w, w0[3], w1[3], w2[3]);
                uint32_t* q = dst + p * SX + 4 * xq;
                *reinterpret_cast<uint4*>(q) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
                if (NP >= 2) *reinterpret_cast<uint4*>(q + PLW) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
                if (NP == 3) *reinterpret_cast<uint4*>(q + 2 * PLW) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
            }
        }
    }

    __device__ static __forceinline__ void sstore7(const OpA& a, const OpB& b, uint32_t* As, uint32_t* Bs, const Stage& st, const Plan& pl) {
        float4 ra[TC::AV], rb[TC::BV];
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            ra[i] = finish4(st.a[i], st.aok[i], MASKS && a.mask != nullptr, st.am[i], a.mscale);
            if constexpr (SPLIT == 8 && !A_KC) {   // column exponents of a transposed-use A operand (plan())
                const float f = pl.asc[i];
                ra[i] = make_float4(ra[i].x * f, ra[i].y * f, ra[i].z * f, ra[i].w * f);
            }
        }
        sstore7_op<A_KC, BM, TC::AV, APL, SXA, A_BF16 && NP == 1>(As, ra);
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            rb[i] = finish4(st.b[i], st.bok[i], MASKS && b.mask != nullptr, st.bm[i], b.mscale);
            if constexpr (SPLIT == 8) {
                const float f = b.pscale;
                rb[i] = make_float4(rb[i].x * f, rb[i].y * f, rb[i].z * f, rb[i].w * f);
            }
        }
        sstore7_op<B_KC, BN, TC::BV, BPL, SXB>(Bs, rb);
    }

    template <bool KC, int PLW, int SX>
    __device__ static __forceinline__ cpg_bf16x8 read7(const uint32_t* plane0, int pl, int x, int lq) {
        const uint32_t* P = plane0 + pl * PLW;
        if (KC) return *reinterpret_cast<const cpg_bf16x8*>(P + x * KCW + 4 * lq);
        if (TRX) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
            const int s = x & 15;
            const uint32_t* c = P + (x >> 4) * TRW + (4 * lq + (s >> 2)) * 8 + (s & 3) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(c));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(c + 16 * 8));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(cpg_bf16x8, w);
        }
        const uint4 w = make_uint4(P[(4 * lq + 0) * SX + x], P[(4 * lq + 1) * SX + x], P[(4 * lq + 2) * SX + x], P[(4 * lq + 3) * SX + x]);
        return __builtin_bit_cast(cpg_bf16x8, w);
    }

    // One slab: for every column block read its three B planes once, then walk the row blocks (A planes re-read per column
    // block: LDS reads are cheap here, registers are not - holding all planes of a 128x64 tile costs a resident wave), six
    // bf16 MFMAs per 16x16 block; the next slab's conversion + LDS writes are left free to interleave with them.
    // Column blocks are walked NG at a time: their B planes are read once and kept while the row blocks stream past (A planes
    // re-read once per group).  NG = 1 at 128 registers per wave; NG = 2 for the 512-thread workgroups (256 registers per
    // wave), which halves the LDS fragment traffic of the wide tiles (the XC image costs four ds_read_b32 per plane).
    static constexpr int NG = (TC::NT == 512 && TC::NI % 2 == 0) ? 2 : 1;
    static constexpr int MG = (TC::MI % 2 == 0) ? 2 : 1;   // row blocks walked together (independent accumulators)
    template <bool STORE>
    __device__ static __forceinline__ void slab7(const OpA& a, const OpB& b, const uint32_t* Ac, const uint32_t* Bc, uint32_t* An,
                                                 uint32_t* Bn, const Stage& st, f32x4 (&acc)[TC::MI][TC::NI], const Plan& pl) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / TC::WN, wn = wave % TC::WN;
        const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int n0 = 0; n0 < TC::NI; n0 += NG) {
            cpg_bf16x8 fb[NG][3];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    if (CPG_ABLATE & 2) fb[g][pl] = __builtin_bit_cast(cpg_bf16x8, acc[0][n0 + g]);
                    else fb[g][pl] = read7<B_KC, BPL, SXB>(Bc, pl, wn * TC::WTN + (n0 + g) * 16 + l15, lq);
                }
#pragma unroll
            for (int m0 = 0; m0 < TC::MI; m0 += MG) {
                cpg_bf16x8 fa[MG][3];
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        if (CPG_ABLATE & 2) fa[m][pl] = __builtin_bit_cast(cpg_bf16x8, acc[m0 + m][0]);
                        else fa[m][pl] = read7<A_KC, APL, SXA>(Ac, pl, wm * TC::WTM + (m0 + m) * 16 + l15, lq);
                    }
                if (CPG_ABLATE & 8) {
#pragma unroll
                    for (int m = 0; m < MG; ++m)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[m0 + m][n0 + g] += __builtin_bit_cast(f32x4, fa[m][0]) * __builtin_bit_cast(f32x4, fb[g][0]);
                    continue;
                }
                // The six products of a block go into ONE accumulator in a fixed order (the result depends on it); a
                // dependent MFMA issues only when its predecessor has left the pipe (8 passes for 16x16x32 against a 4-pass
                // issue slot), so the MG x NG independent blocks are walked term by term - same sums, no dependency stalls.
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
                if constexpr (SPLIT == 8) {   // f16 pairs: low x high, high x low, high x high
                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int m = 0; m < MG; ++m)
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                acc[m0 + m][n0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cpg_f16x8, fa[m][HA[t]]),
                                                                                             __builtin_bit_cast(cpg_f16x8, fb[g][HB[t]]),
                                                                                             acc[m0 + m][n0 + g], 0, 0, 0);
                    continue;
                }
#pragma unroll
                for (int t = (NP == 3 ? 0 : 5); t < 6; ++t)
#pragma unroll
                    for (int m = 0; m < MG; ++m)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[m0 + m][n0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m][TA[t]], fb[g][TB[t]], acc[m0 + m][n0 + g], 0, 0, 0);
            }
        }
        if (STORE && !(CPG_ABLATE & 1)) sstore7(a, b, An, Bn, st, pl);
    }

    template <class Hook>
    __device__ static __forceinline__ void run7(const OpA& a, const OpB& b, int K, f32x4 (&acc)[TC::MI][TC::NI], int hook_kt,
                                                Hook&& hook) {
        extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
        uint32_t* const base = reinterpret_cast<uint32_t*>(cpg_smem);
        uint32_t* const A0 = base;
        uint32_t* const A1 = base + ASZ7;
        uint32_t* const B0 = base + 2 * ASZ7;
        uint32_t* const B1 = base + 2 * ASZ7 + BSZ7;
        Stage st;
        Plan pl;
        plan(a, b, pl);
        const int KT = (K + BK - 1) / BK;
        gload(a, b, pl, 0, K, st);
        if constexpr (SB) {
            uint32_t* const Bs = base + ASZ7;
            sstore7(a, b, A0, Bs, st, pl);
            __syncthreads();
            for (int kt = 0; kt < KT; ++kt) {
                if (kt == hook_kt) hook();
                if (kt + 1 < KT) gload(a, b, pl, (kt + 1) * BK, K, st);
                slab7<false>(a, b, A0, Bs, A0, Bs, st, acc, pl);
                __syncthreads();
                if (kt + 1 < KT) {
                    sstore7(a, b, A0, Bs, st, pl);
                    __syncthreads();
                }
            }
            return;
        }
        sstore7(a, b, A0, B0, st, pl);
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt == hook_kt) hook();
            if (kt + 1 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, pl, (kt + 1) * BK, K, st);
                slab7<true>(a, b, A0, B0, A1, B1, st, acc, pl);
            } else {
                slab7<false>(a, b, A0, B0, A1, B1, st, acc, pl);
            }
            __syncthreads();
            if (kt + 1 >= KT) break;
            if (kt + 2 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, pl, (kt + 2) * BK, K, st);
                slab7<true>(a, b, A1, B1, A0, B0, st, acc, pl);
            } else {
                slab7<false>(a, b, A1, B1, A0, B0, st, acc, pl);
            }
            __syncthreads();
        }
    }

    struct Frag {  // fragments of one slab: [half][block][k-step inside the half]
        float a[NH][TC::MI][4], b[NH][TC::NI][4];
    };

    __device__ static __forceinline__ void read_frags(const float* Ac, const float* Bc, Frag& f) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / TC::WN, wn = wave % TC::WN;
        const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi) {
                const int x = wm * TC::WTM + mi * 16 + l15;
                if (A_KC) {  // one 16-byte read: k = 16h + 4*lq + {0,1,2,3}
                    const float4 v = *reinterpret_cast<const float4*>(Ac + x * LDA + 16 * h + 4 * lq);
                    f.a[h][mi][0] = v.x; f.a[h][mi][1] = v.y; f.a[h][mi][2] = v.z; f.a[h][mi][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = PERM ? 16 * h + 4 * lq + j : 16 * h + 4 * j + lq;
                        f.a[h][mi][j] = Ac[k * LDA + x];
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < TC::NI; ++ni) {
                const int x = wn * TC::WTN + ni * 16 + l15;
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(Bc + x * LDB + 16 * h + 4 * lq);
                    f.b[h][ni][0] = v.x; f.b[h][ni][1] = v.y; f.b[h][ni][2] = v.z; f.b[h][ni][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = PERM ? 16 * h + 4 * lq + j : 16 * h + 4 * j + lq;
                        f.b[h][ni][j] = Bc[k * LDB + x];
                    }
                }
            }
        }
    }

    template <int H0, int H1>
    __device__ static __forceinline__ void mfmas(const Frag& f, f32x4 (&acc)[TC::MI][TC::NI]) {
#pragma unroll
        for (int h = H0; h < H1; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TC::NI; ++ni) {
                        if (CPG_ABLATE & 8) {  // no matrix instructions: keep the operands live with one VALU op instead
                            acc[mi][ni][0] += f.a[h][mi][j] * f.b[h][ni][j];
                        } else {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[h][mi][j], f.b[h][ni][j], acc[mi][ni], 0, 0, 0);
                        }
                    }
    }

    // One slab: [fragment reads of the whole slab] | MFMAs first half | LDS writes of the next slab | MFMAs second half.
    // The sched_barriers pin that order: left alone, hipcc sinks every ds_read next to its first use and reuses the
    // same VGPRs (read -> lgkmcnt(0) -> 4 MFMAs -> read ...), exposing the LDS latency once per k-step.
    template <bool STORE>
    __device__ static __forceinline__ void slab(const OpA& a, const OpB& b, const float* Ac, const float* Bc, float* An,
                                                float* Bn, const Stage& st, f32x4 (&acc)[TC::MI][TC::NI]) {
        Frag f;
        if (CPG_ABLATE & 2) {  // keep the registers live and opaque, but do not touch the LDS
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < TC::MI; ++mi) f.a[h][mi][j] = acc[mi][0][j];
#pragma unroll
                    for (int ni = 0; ni < TC::NI; ++ni) f.b[h][ni][j] = acc[0][ni][(j + 1) & 3];
                }
        } else {
            read_frags(Ac, Bc, f);
        }
        __builtin_amdgcn_sched_barrier(0);
#if CPG_SCHED == 0
        mfmas<0, NH / 2>(f, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (STORE && !(CPG_ABLATE & 1)) sstore(a, b, An, Bn, st);
        __builtin_amdgcn_sched_barrier(0);
        mfmas<NH / 2, NH>(f, acc);
#elif CPG_SCHED == 1
        // free interleaving of the LDS stores with the matrix instructions
        mfmas<0, NH / 2>(f, acc);
        if (STORE && !(CPG_ABLATE & 1)) sstore(a, b, An, Bn, st);
        mfmas<NH / 2, NH>(f, acc);
#else
        // explicit pattern: one MFMA, then up to CPG_SCHED-1 VALU/DS-write instructions of the staging code, repeated
        mfmas<0, NH>(f, acc);
        if (STORE && !(CPG_ABLATE & 1)) sstore(a, b, An, Bn, st);
#pragma unroll
        for (int i = 0; i < NH * 4 * TC::MI * TC::NI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, CPG_SCHED - 1, 0);  // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              // DS write
        }
#endif
    }

    // acc[mi][ni] += A_tile * B_tile over the whole K range.  Uses TC::smem_floats<A_KC,B_KC>() floats of dynamic LDS.
    // The two LDS buffers are addressed with compile-time offsets from the __shared__ symbol itself (2x unrolled slab
    // loop): runtime-selected buffer pointers degrade to flat_* accesses whose waits also drain the global prefetch.
    // `hook` runs once, ahead of slab `hook_kt` (even; < 0: never): the caller's own global loads, issued from inside the loop
    // so that their latency and their HBM burst sit under the matrix work instead of ahead of it.
    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    __device__ static __forceinline__ void run(const OpA& a, const OpB& b, int K, f32x4 (&acc)[TC::MI][TC::NI]) {
        run(a, b, K, acc, -1, NoHook{});
    }
    template <class Hook>
    __device__ static __forceinline__ void run(const OpA& a, const OpB& b, int K, f32x4 (&acc)[TC::MI][TC::NI], int hook_kt,
                                               Hook&& hook) {
        if (SPLIT != 0) {
            run7(a, b, K, acc, hook_kt, hook);
            return;
        }
        extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
        float* const A0 = cpg_smem;
        float* const A1 = cpg_smem + ASZ;
        float* const B0 = cpg_smem + 2 * ASZ;
        float* const B1 = cpg_smem + 2 * ASZ + BSZ;
        Stage st;
        Plan pl;
        plan(a, b, pl);
        const int KT = (K + BK - 1) / BK;
        gload(a, b, pl, 0, K, st);
        sstore(a, b, A0, B0, st);
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt == hook_kt) hook();
            if (kt + 1 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, pl, (kt + 1) * BK, K, st);
                slab<true>(a, b, A0, B0, A1, B1, st, acc);
            } else {
                slab<false>(a, b, A0, B0, A1, B1, st, acc);
            }
            if (!(CPG_ABLATE & 4)) __syncthreads();
            if (kt + 1 >= KT) break;
            if (kt + 2 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, pl, (kt + 2) * BK, K, st);
                slab<true>(a, b, A1, B1, A0, B0, st, acc);
            } else {
                slab<false>(a, b, A1, B1, A0, B0, st, acc);
            }
            if (!(CPG_ABLATE & 4)) __syncthreads();
        }
    }
};

// Coordinates of accumulator element (mi, ni, reg) held by this lane.
template <class TC>
__device__ __forceinline__ int acc_row(int mi, int reg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / TC::WN) * TC::WTM + mi * 16 + (lane >> 4) * 4 + reg;
}
template <class TC>
__device__ __forceinline__ int acc_col(int ni) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % TC::WN) * TC::WTN + ni * 16 + (lane & 15);
}

// 16x16 accumulator block (lane (u = l&15, rq = l>>4), reg -> row 4rq+reg, col u) -> one f32x4 per lane in row layout
// (lane -> row l>>2, cols 4(l&3)..+3) through a 1 KB per-wave LDS buffer no other wave touches
__device__ __forceinline__ f32x4 acc_block_to_rows(float* tb, const f32x4 v, int lane) {
    const int u = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[(4 * rq + r) * 16 + u] = v[r];
    return *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * 16 + 4 * (lane & 3));
}


// ---- direct-to-LDS main loop for products with BOTH operands K-contiguous: acc[mi][ni] += A[rows, K] . B[cols, K]^T on the
// exact-f32 MFMA, BM x BN tile, 2 x 2 waves (wave tile BM/2 x BN/2 = MI x NI blocks of 16 x 16).  Operands are staged by
// `global_load_lds_dwordx4` into an NS-stage LDS ring, NS - 1 slabs ahead of the product: no staging registers, no ds_write
// pass, one ds_read_b128 per fragment half, and almost no VALU work in the slab loop (the exact-f32 MFMA shares its lanes with
// the VALU: every address / mask instruction of the register-staged loop is paid in matrix-pipe time).  An LDS-DMA lane writes
// to base + 16 * lane, so the image is unpadded ([row][32 floats]); bank conflicts of the fragment reads are avoided by a
// source-side swizzle instead: lane (row, slot s) loads k-chunk s ^ f(row), f(row) = (row >> 1) & 7, and the reader of k-chunk q
// of a row reads slot q ^ f(row) (PMC: 1 % of the LDS cycles in conflicts).  Contraction order = MainLoop's permuted order
// (k = 16h + 4q + j): sums are bit-identical to the register-staged exact-f32 kernels.
// Requirements: full tiles (no bound masks), K % 32 == 0, 16-byte aligned rows (lda, ldb multiples of 4, aligned bases).
// `hook` runs once ahead of slab hook_kt (the caller's own global loads; < 0: never).  smem: smem_floats() floats.
// PREC = 1 ("bf16 compute mode"): each lane's eight slab values per block row are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) after
// the fragment read and ONE v_mfma_f32_16x16x32_bf16 per block consumes the slab - same LDS image and reads as PREC = 0.
// PREC = 2 (bf16 compute mode on operands that ARE bf16 in memory: the bf16 gradient storage of round 4): a slab is 64 k deep, so a
// row is again 128 bytes = eight 16-byte chunks - the LDS image, the LDS-DMA pattern and the swizzle are those of the f32 slab;
// chunk q of a row now holds k = 8q .. 8q+7, which is exactly the A / B fragment of v_mfma_f32_16x16x32_bf16 for lane group
// q & 3 of k-block q >> 2: the two ds_read_b128 of a block row feed two MFMAs, no conversion, half the bytes per k.
// PREC = 3 (f32-grade on f16 PAIRS that are pairs in memory, round 4): a 128-byte slab row holds 32 k as [32 hi | 32 lo] f16, so
// chunks 0..3 are the high halves of k = 8q .. 8q+7 and chunks 4..7 the low halves of the same k - the reads of PREC 2, feeding
// THREE v_mfma_f32_16x16x32_f16 per block and slab (lo x hi, hi x lo, hi x hi): the bytes per k of the f32 slab, 32 k per slab, and
// 48 matrix-pipe cycles per block and slab in place of the 268 of eight f32 MFMAs.  `pre(kt)` runs ahead of slab kt's products
// (the caller rescales its accumulators when the operands' power-of-two scale changes along k) and returns false to skip them.
struct DlNoScale {};   // DlLoop::run without a fragment-scale hook
struct DlPairScale {   // PREC 4: powers of two the A / B operand values are multiplied by before their f16-pair split (passed in the fscale slot)
    float a, b;
};
// WR: wave rows of the loop's WR x 2 wave grid - 2: four waves (256 threads), wave tile BM/2 x BN/2; 4 (round 6: the 128 x 64 tile of the
// paired BPTT launch, one 512-thread workgroup where two 64 x 64 ones each fetched the same W_hh^T tile): eight waves, wave tile BM/4 x BN/2
// WC: wave columns - 2, or 4 (round 6: 64 x 64 tile on eight waves of 32 x 16 - half the epilogue state per wave, four waves per SIMD)
template <int BM, int BN, int NS = 3, int PREC = 0, int WR = 2, int WC = 2>
struct DlLoop {
    static constexpr int NT = 64 * WR * WC;                          // threads that run one loop
    static constexpr int PR = NT / 8;                                // rows one pass of those threads moves (16 bytes per thread, 128 per row)
    static constexpr int MI = BM / (16 * WR), NI = BN / (16 * WC);   // 16 x 16 blocks of a wave tile
    static constexpr int PA = BM / PR, PB = BN / PR;                 // passes per slab: LDS-DMA instructions per thread
    static_assert(BM % PR == 0 && BN % PR == 0 && BM % (16 * WR) == 0 && BN % (16 * WC) == 0, "whole passes, whole blocks");
    static constexpr int AF = BM * 32, BF = BN * 32, SF = AF + BF;   // floats per operand slab / per stage
    static constexpr int LPS = PA + PB;                              // LDS-DMA instructions per thread and slab
    static constexpr int AHEAD = NS - 1;
    static constexpr size_t smem_floats() { return (size_t)NS * SF; }

    // E: element type of the operands in memory - float (PREC 0 / 1) or uint16_t = bf16 (PREC 2); K, lda, ldb count elements
    template <class Hook, class E>
    __device__ static __forceinline__ void run(const E* A, size_t lda, const E* Bt, size_t ldb, int K, float* smem,
                                               f32x4 (&acc)[MI][NI], int hook_kt, Hook&& hook) {
        run(A, lda, Bt, ldb, K, smem, acc, hook_kt, hook, [](int) { return true; });
    }
    template <class Hook, class E, class Pre>
    __device__ static __forceinline__ void run(const E* A, size_t lda, const E* Bt, size_t ldb, int K, float* smem,
                                               f32x4 (&acc)[MI][NI], int hook_kt, Hook&& hook, Pre&& pre) {
        run(A, lda, Bt, ldb, K, smem, acc, hook_kt, hook, pre, DlNoScale{});
    }
    // fscale (PREC 3 only): uint32 fscale(kt, mi) = a power of two <= 1 as two packed f16, multiplied into block row mi's A fragments
    // of slab kt (v_pk_mul_f16) - A images whose k-segments carry different power-of-two scales are brought to ONE unit on the way into
    // the MFMAs, so the accumulators never need rescaling (planes.hip: gradient images of two 32-row blocks per wave)
    template <class Hook, class E, class Pre, class FS>
    __device__ static __forceinline__ void run(const E* A, size_t lda, const E* Bt, size_t ldb, int K, float* smem,
                                               f32x4 (&acc)[MI][NI], int hook_kt, Hook&& hook, Pre&& pre, FS&& fscale) {
        run(A, lda, Bt, ldb, K, smem, acc, hook_kt, hook, pre, fscale, DlNoScale{});
    }
    // arow (optional): const E* arow(i, r) = start of the A row that tile row 32 i + r (r = this thread's tid / 8) is GATHERED from -
    // an LDS-DMA lane supplies its own global address, so a row-gathered A operand (beam search: states re-gathered by back-pointer,
    // planes.hip) costs nothing in the loop
    template <class Hook, class E, class Pre, class FS, class AR>
    __device__ static __forceinline__ void run(const E* A, size_t lda, const E* Bt, size_t ldb, int K, float* smem,
                                               f32x4 (&acc)[MI][NI], int hook_kt, Hook&& hook, Pre&& pre, FS&& fscale, AR&& arow) {
        run(A, lda, Bt, ldb, K, smem, acc, hook_kt, hook, pre, fscale, arow, DlNoScale{});
    }
    // brow (optional): the same for the B operand's rows (tile column 32 i + r) - products whose N is no multiple of the tile clamp
    // the rows past it onto the last valid one (round 6: the grouped nn.Linear-shaped launches)
    template <class Hook, class E, class Pre, class FS, class AR, class BR>
    __device__ static __forceinline__ void run(const E* A, size_t lda, const E* Bt, size_t ldb, int K, float* smem,
                                               f32x4 (&acc)[MI][NI], int hook_kt, Hook&& hook, Pre&& pre, FS&& fscale, AR&& arow, BR&& brow) {
        static_assert((PREC == 2 || PREC == 3) == (sizeof(E) == 2), "PREC 2 / 3 <-> 16-bit operands in memory");
        float pscale_a = 1.f, pscale_b = 1.f;
        if constexpr (__is_same(__remove_cvref(FS), DlPairScale)) { pscale_a = fscale.a; pscale_b = fscale.b; }
        constexpr int EPC = 16 / (int)sizeof(E);   // elements per 16-byte chunk
        constexpr int BKE = 8 * EPC;               // elements per slab (a row of a slab is always 128 bytes)
        // (a 512-thread workgroup runs two of these loops side by side - its two 256-thread halves, each on its own ring and its
        // own half of K: gru_step_bwd_dl2_kernel; the slab barrier is the workgroup's)
        const int tid = threadIdx.x & (NT - 1), lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / WC, wn = wave % WC, l15 = lane & 15, lq = lane >> 4;
        const int KT = K / BKE;
        // this thread's 16-byte pieces of a slab: piece i covers row = PR i + tid / 8, slot = tid % 8 holds k-chunk slot ^ f(row)
        const int srow = tid >> 3, sch = (tid & 7) ^ ((srow >> 1) & 7);   // f(PR i + r) = f(r): PR is a multiple of 16
        const E* ga = A + (size_t)srow * lda + EPC * sch;
        const E* gb = Bt + (size_t)srow * ldb + EPC * sch;
        const size_t gaP = PR * lda, gbP = PR * ldb;
        const E* gar[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if constexpr (__is_same(__remove_cvref(AR), DlNoScale)) gar[i] = ga + i * gaP;
            else gar[i] = arow(i, srow) + EPC * sch;
        }
        const E* gbr[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if constexpr (__is_same(__remove_cvref(BR), DlNoScale)) gbr[i] = gb + i * gbP;
            else gbr[i] = brow(i, srow) + EPC * sch;
        }
        auto issue = [&](int kt, float* stage) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds(gar[i] + kt * BKE, (__attribute__((address_space(3))) void*)(stage + i * (PR * 32) + wave * 256), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_global_load_lds(gbr[i] + kt * BKE, (__attribute__((address_space(3))) void*)(stage + AF + i * (PR * 32) + wave * 256), 16, 0, 0);
        };
        // fragment word offsets inside a stage (slab-invariant): row r of block mi / ni, k-chunk q = 4h + lq -> slot q ^ f(r)
        const int ra = wm * (BM / WR) + l15, rbn = wn * (BN / WC) + l15;
        const int fa = (ra >> 1) & 7, fb = (rbn >> 1) & 7;   // f is the same for rows 16 apart
        const int oa0 = ra * 32 + 4 * (lq ^ fa), oa1 = ra * 32 + 4 * ((4 + lq) ^ fa);
        const int ob0 = AF + rbn * 32 + 4 * (lq ^ fb), ob1 = AF + rbn * 32 + 4 * ((4 + lq) ^ fb);
        // one slab: slab kt has landed once only the loads of the slabs after it are outstanding; the barrier then makes every
        // wave's piece visible and retires every wave's fragment reads of slab kt-1, whose stage is refilled right after it
        auto slab = [&](int kt, const float* cur, float* refill) {
            if (kt == hook_kt) hook();
            if (NS == 3) {   // the shipped form: one compare per slab
                if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                const int left = KT - 1 - kt;
                if (left >= AHEAD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (AHEAD - 1)) : "memory");
                else if (AHEAD > 2 && left == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * 2) : "memory");
                else if (AHEAD > 1 && left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + AHEAD < KT) issue(kt + AHEAD, refill);
            if constexpr (PREC == 3) {
                if (pre(kt)) {
                    cpg_f16x8 fa[2][MI], fb[2][NI];   // [0] high halves, [1] low halves
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        fa[0][mi] = *reinterpret_cast<const cpg_f16x8*>(cur + oa0 + mi * 512);
                        fa[1][mi] = *reinterpret_cast<const cpg_f16x8*>(cur + oa1 + mi * 512);
                        if constexpr (!__is_same(__remove_cvref(FS), DlNoScale)) {
                            const cpg_f16x2 f = __builtin_bit_cast(cpg_f16x2, (uint32_t)fscale(kt, mi));
#pragma unroll
                            for (int p = 0; p < 2; ++p) {
                                uint4 w = __builtin_bit_cast(uint4, fa[p][mi]);
                                w.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(cpg_f16x2, w.x) * f);
                                w.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(cpg_f16x2, w.y) * f);
                                w.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(cpg_f16x2, w.z) * f);
                                w.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(cpg_f16x2, w.w) * f);
                                fa[p][mi] = __builtin_bit_cast(cpg_f16x8, w);
                            }
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        fb[0][ni] = *reinterpret_cast<const cpg_f16x8*>(cur + ob0 + ni * 512);
                        fb[1][ni] = *reinterpret_cast<const cpg_f16x8*>(cur + ob1 + ni * 512);
                    }
                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[HA[t]][mi], fb[HB[t]][ni], acc[mi][ni], 0, 0, 0);
                }
            } else if constexpr (PREC == 2) {
                cpg_bf16x8 fa[2][MI], fb[2][NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    fa[0][mi] = *reinterpret_cast<const cpg_bf16x8*>(cur + oa0 + mi * 512);
                    fa[1][mi] = *reinterpret_cast<const cpg_bf16x8*>(cur + oa1 + mi * 512);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    fb[0][ni] = *reinterpret_cast<const cpg_bf16x8*>(cur + ob0 + ni * 512);
                    fb[1][ni] = *reinterpret_cast<const cpg_bf16x8*>(cur + ob1 + ni * 512);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][mi], fb[h][ni], acc[mi][ni], 0, 0, 0);
            } else if constexpr (PREC == 1) {
                cpg_bf16x8 fa[MI], fb[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cur + oa0 + mi * 512), hi = *reinterpret_cast<const f32x4*>(cur + oa1 + mi * 512);
                    fa[mi] = __builtin_bit_cast(cpg_bf16x8, make_uint4(cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]), cvt_pk_bf16(hi[0], hi[1]), cvt_pk_bf16(hi[2], hi[3])));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cur + ob0 + ni * 512), hi = *reinterpret_cast<const f32x4*>(cur + ob1 + ni * 512);
                    fb[ni] = __builtin_bit_cast(cpg_bf16x8, make_uint4(cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]), cvt_pk_bf16(hi[0], hi[1]), cvt_pk_bf16(hi[2], hi[3])));
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
            } else if constexpr (PREC == 4) {
                // f32 operands in LDS, split into f16 pairs where the fragments are read: lane group lq takes k = 4 lq .. + 3 and
                // 16 + 4 lq .. + 3 of the slab for BOTH operands (any k order the two sides share is a dot product), times the
                // operands' powers of two (pscale_a / pscale_b: exact), three f16 MFMAs per block (lo x hi, hi x lo, hi x hi)
                cpg_f16x8 fa[2][MI], fb[2][NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(cur + oa0 + mi * 512) * pscale_a;
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(cur + oa1 + mi * 512) * pscale_a;
                    uint32_t h[4], l[4];
                    split2h_pair(x0[0], x0[1], h[0], l[0]);
                    split2h_pair(x0[2], x0[3], h[1], l[1]);
                    split2h_pair(x1[0], x1[1], h[2], l[2]);
                    split2h_pair(x1[2], x1[3], h[3], l[3]);
                    fa[0][mi] = __builtin_bit_cast(cpg_f16x8, make_uint4(h[0], h[1], h[2], h[3]));
                    fa[1][mi] = __builtin_bit_cast(cpg_f16x8, make_uint4(l[0], l[1], l[2], l[3]));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(cur + ob0 + ni * 512) * pscale_b;
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(cur + ob1 + ni * 512) * pscale_b;
                    uint32_t h[4], l[4];
                    split2h_pair(x0[0], x0[1], h[0], l[0]);
                    split2h_pair(x0[2], x0[3], h[1], l[1]);
                    split2h_pair(x1[0], x1[1], h[2], l[2]);
                    split2h_pair(x1[2], x1[3], h[3], l[3]);
                    fb[0][ni] = __builtin_bit_cast(cpg_f16x8, make_uint4(h[0], h[1], h[2], h[3]));
                    fb[1][ni] = __builtin_bit_cast(cpg_f16x8, make_uint4(l[0], l[1], l[2], l[3]));
                }
                constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[HA[t]][mi], fb[HB[t]][ni], acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 av[MI], bv[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(cur + (h ? oa1 : oa0) + mi * 512);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(cur + (h ? ob1 : ob0) + ni * 512);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
            }
            }
        };
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < KT) issue(i, smem + i * SF);
        int kt = 0;
        for (; kt + NS <= KT; kt += NS) {   // NS slabs per trip: the stage of every access is a compile-time offset
#pragma unroll
            for (int i = 0; i < NS; ++i) slab(kt + i, smem + i * SF, smem + ((i + NS - 1) % NS) * SF);
        }
#pragma unroll
        for (int i = 0; i < NS - 1; ++i)
            if (kt + i < KT) slab(kt + i, smem + i * SF, smem + ((i + NS - 1) % NS) * SF);
        // the last slab's fragment reads are done when its MFMAs have their operands; callers that reuse the LDS ring must
        // place a barrier first
    }
};

