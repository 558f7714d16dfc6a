// f32 MFMA tile engine for gfx950 (v_mfma_f32_16x16x4_f32: exact f32, 64 FLOP/clk/SIMD).
//
// One 256-thread workgroup (4 wave64) computes a BM x BN tile of  C = A * B  with the contraction
// dimension K streamed through LDS in BK-deep slabs, double buffered (global -> registers while the
// MFMAs of the current slab run, registers -> the other LDS buffer, one barrier per slab).
//
// Each operand can be stored either with K contiguous ("KC": rows of h / rows of W, the natural torch
// layouts) or with its free dimension contiguous ("XC": transposed use, e.g. dW = dY^T X).  The LDS image
// is chosen per layout so that the per-lane fragment reads (ds_read_b32) are bank-conflict free:
//   KC -> [X][BK+2]   (lanes 0..15 walk X: stride 2 mod 32 banks, lanes 16..31 sit on the odd banks)
//   XC -> [BK][X+16]  (lanes 0..15 consecutive banks, lanes 16..31 shifted by 16)
// MFMA fragment mapping (cdna guide section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C/D: col = l&15, row = (l>>4)*4 + reg.
//
// The N side supports a "segmented" row map so one tile can hold the r,z,n gate rows of the same hidden
// units: n_local -> jblk = n_local/(16*NSEG), seg = (n_local/16)%NSEG, j = j0 + jblk*16 + n_local%16,
// global row/col = seg*seg_stride + j, valid iff j < seg_len.  NSEG=1 is the plain map.
#pragma once
#include "cpg_common.h"

struct OpA {
    const float* p;
    int ld;
    int m0;               // first global M index of this tile
    int M;                // bound on the M index
    const uint8_t* mask;  // optional keep-mask with the same indexing as p (value = p * (mask ? mscale : 0))
    float mscale;
};

struct OpB {
    const float* p;
    int ld;
    int j0;          // first j of this tile
    int seg_len;     // bound on j
    int seg_stride;  // row/col offset between segments
    const uint8_t* mask;
    float mscale;
};

template <int BM_, int BN_, int BK_, int WM_, int WN_, int NSEG_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NSEG = NSEG_;
    static constexpr int NT = 256;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int MI = WTM / 16, NI = WTN / 16;
    static constexpr int AV = BM * BK / 4 / NT;  // float4 staging registers per thread
    static constexpr int BV = BN * BK / 4 / NT;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tile must be a multiple of 16");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging must divide evenly");
    static_assert(WTN % (16 * NSEG) == 0, "a wave must own whole segment groups");
    template <bool KC>
    static constexpr int lda() { return KC ? BK + 2 : BM + 16; }
    template <bool KC>
    static constexpr int ldb() { return KC ? BK + 2 : BN + 16; }
    template <bool KC>
    static constexpr int a_elems() { return KC ? BM * (BK + 2) : BK * (BM + 16); }
    template <bool KC>
    static constexpr int b_elems() { return KC ? BN * (BK + 2) : BK * (BN + 16); }
    template <bool AKC, bool BKC>
    static constexpr int smem_floats() { return 2 * (a_elems<AKC>() + b_elems<BKC>()); }
};

template <int NSEG>
__device__ __forceinline__ void bmap(const OpB& b, int n_local, int& idx, int& j) {
    const int jblk = n_local / (16 * NSEG);
    const int seg = (n_local / 16) % NSEG;
    j = b.j0 + jblk * 16 + (n_local & 15);
    idx = seg * b.seg_stride + j;
}

__device__ __forceinline__ float4 apply_mask4(float4 v, const uint8_t* m, float s, bool vec) {
    if (vec) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(m);
        v.x = k.x ? v.x * s : 0.f;
        v.y = k.y ? v.y * s : 0.f;
        v.z = k.z ? v.z * s : 0.f;
        v.w = k.w ? v.w * s : 0.f;
    }
    return v;
}

// Guarded 4-wide load of p[0..3] where element c is in range iff c < nvalid.
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* p, int nvalid, const uint8_t* mask, float ms) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid <= 0) return v;
    if (VEC && nvalid >= 4) {
        v = *reinterpret_cast<const float4*>(p);
        if (mask) v = apply_mask4(v, mask, ms, true);
        return v;
    }
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < nvalid) {
            float x = p[c];
            if (mask) x = mask[c] ? x * ms : 0.f;
            t[c] = x;
        }
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <class TC, bool A_KC, bool B_KC, bool AVEC, bool BVEC>
struct MainLoop {
    static constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK;
    static constexpr int LDA = TC::template lda<A_KC>();
    static constexpr int LDB = TC::template ldb<B_KC>();
    static constexpr int ASZ = TC::template a_elems<A_KC>();
    static constexpr int BSZ = TC::template b_elems<B_KC>();

    __device__ static __forceinline__ void gload_a(const OpA& a, int k0, int K, float4 (&r)[TC::AV]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            if (A_KC) {
                const int row = v / (BK / 4), kq = v % (BK / 4);
                const int gm = a.m0 + row, k = k0 + 4 * kq;
                const int nvalid = (gm < a.M) ? (K - k) : 0;
                const size_t off = (size_t)gm * a.ld + k;
                r[i] = load4<AVEC>(a.p + off, nvalid, a.mask ? a.mask + off : nullptr, a.mscale);
            } else {
                const int kk = v / (BM / 4), mq = v % (BM / 4);
                const int gk = k0 + kk, gm = a.m0 + 4 * mq;
                const int nvalid = (gk < K) ? (a.M - gm) : 0;
                const size_t off = (size_t)gk * a.ld + gm;
                r[i] = load4<AVEC>(a.p + off, nvalid, a.mask ? a.mask + off : nullptr, a.mscale);
            }
        }
    }

    __device__ static __forceinline__ void gload_b(const OpB& b, int k0, int K, float4 (&r)[TC::BV]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            if (B_KC) {
                const int nl = v / (BK / 4), kq = v % (BK / 4);
                int idx, j;
                bmap<TC::NSEG>(b, nl, idx, j);
                const int k = k0 + 4 * kq;
                const int nvalid = (j < b.seg_len) ? (K - k) : 0;
                const size_t off = (size_t)idx * b.ld + k;
                r[i] = load4<BVEC>(b.p + off, nvalid, b.mask ? b.mask + off : nullptr, b.mscale);
            } else {
                const int kk = v / (BN / 4), nq = v % (BN / 4);
                int idx, j;
                bmap<TC::NSEG>(b, 4 * nq, idx, j);
                const int gk = k0 + kk;
                const int nvalid = (gk < K) ? (b.seg_len - j) : 0;
                const size_t off = (size_t)gk * b.ld + idx;
                r[i] = load4<BVEC>(b.p + off, nvalid, b.mask ? b.mask + off : nullptr, b.mscale);
            }
        }
    }

    __device__ static __forceinline__ void sstore_a(float* As, const float4 (&r)[TC::AV]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            if (A_KC) {
                const int row = v / (BK / 4), kq = v % (BK / 4);
                float2* d = reinterpret_cast<float2*>(As + row * LDA + 4 * kq);
                d[0] = make_float2(r[i].x, r[i].y);
                d[1] = make_float2(r[i].z, r[i].w);
            } else {
                const int kk = v / (BM / 4), mq = v % (BM / 4);
                *reinterpret_cast<float4*>(As + kk * LDA + 4 * mq) = r[i];
            }
        }
    }

    __device__ static __forceinline__ void sstore_b(float* Bs, const float4 (&r)[TC::BV]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            if (B_KC) {
                const int nl = v / (BK / 4), kq = v % (BK / 4);
                float2* d = reinterpret_cast<float2*>(Bs + nl * LDB + 4 * kq);
                d[0] = make_float2(r[i].x, r[i].y);
                d[1] = make_float2(r[i].z, r[i].w);
            } else {
                const int kk = v / (BN / 4), nq = v % (BN / 4);
                *reinterpret_cast<float4*>(Bs + kk * LDB + 4 * nq) = r[i];
            }
        }
    }

    // acc[mi][ni] += A_tile * B_tile over the whole K range.  smem: TC::smem_floats<A_KC,B_KC>() floats.
    __device__ static __forceinline__ void run(const OpA& a, const OpB& b, int K, float* smem,
                                               f32x4 (&acc)[TC::MI][TC::NI]) {
        float* As[2] = {smem, smem + ASZ};
        float* Bs[2] = {smem + 2 * ASZ, smem + 2 * ASZ + BSZ};
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / TC::WN, wn = wave % TC::WN;
        const int l15 = lane & 15, lq = lane >> 4;
        float4 ra[TC::AV], rb[TC::BV];
        const int KT = (K + BK - 1) / BK;
        gload_a(a, 0, K, ra);
        gload_b(b, 0, K, rb);
        sstore_a(As[0], ra);
        sstore_b(Bs[0], rb);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) {
                gload_a(a, (kt + 1) * BK, K, ra);
                gload_b(b, (kt + 1) * BK, K, rb);
            }
            const float* Ac = As[cur];
            const float* Bc = Bs[cur];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                float af[TC::MI], bf[TC::NI];
#pragma unroll
                for (int mi = 0; mi < TC::MI; ++mi) {
                    const int x = wm * TC::WTM + mi * 16 + l15;
                    af[mi] = A_KC ? Ac[x * LDA + kk + lq] : Ac[(kk + lq) * LDA + x];
                }
#pragma unroll
                for (int ni = 0; ni < TC::NI; ++ni) {
                    const int x = wn * TC::WTN + ni * 16 + l15;
                    bf[ni] = B_KC ? Bc[x * LDB + kk + lq] : Bc[(kk + lq) * LDB + x];
                }
#pragma unroll
                for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TC::NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
            }
            if (kt + 1 < KT) {
                sstore_a(As[cur ^ 1], ra);
                sstore_b(Bs[cur ^ 1], rb);
            }
            __syncthreads();
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
