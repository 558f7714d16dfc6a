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

// Diagnostic builds only (tools/ablate.sh): -DCPG_ABLATE=<mask> removes one phase of the slab loop after the first slab so its
// cost can be measured in isolation.  1: no global loads / LDS writes   2: no LDS fragment reads   4: no barrier.
// Results of such builds are wrong by construction; the shipped library is built without the macro.
#ifndef CPG_ABLATE
#define CPG_ABLATE 0
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

// XCD-aware tile order (MI355X: 8 XCDs with private 4 MiB L2s; workgroup `id` is observed to run on XCD id % 8 - used for
// speed only, any placement is correct).  The default x-fastest order puts every N-tile column on its own XCD, so each
// XCD streams the WHOLE M-side operand (measured on the dW_hh product: FETCH_SIZE 1.29 GB raw vs 0.52 GB algorithmic).
// Remapped order: the gx workgroups that share one (y,z) - the same M-side rows / K-chunk - run back to back on ONE XCD,
// and each XCD owns a contiguous 1/8 of the (y,z) combinations.
__device__ __forceinline__ void xcd_tile_order(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int combos = gy * gz;
    if (combos % 8 != 0) {
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

// 4-wide operand fetch of p[off .. off+3]; element c is in range iff c < nvalid.  Loads are UNCONDITIONAL (clamped to
// offset 0, which always exists); the validity bits are returned and applied when the registers are written to LDS,
// i.e. AFTER the MFMA phase the loads overlap with - a branch or a select right behind a load makes hipcc wait for the
// load where it is issued and serialises the staging phase (cdna guide section 5, trap (c)).
// VEC (16-byte loads) requires the caller to guarantee alignment AND that vectors never straddle a bound
// (nvalid is then either <= 0 or >= 4).
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* p, size_t off, int nvalid, unsigned& okbits) {
    if (VEC) {
        const bool ok = nvalid >= 4;
        okbits = ok ? 0xFu : 0u;
        return *reinterpret_cast<const float4*>(p + (ok ? off : 0));
    }
    float t[4];
    okbits = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool ok = c < nvalid;
        okbits |= ok ? (1u << c) : 0u;
        t[c] = p[ok ? off + c : 0];
    }
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <bool VEC>
__device__ __forceinline__ uchar4 loadmask4(const uint8_t* m, size_t off, unsigned okbits) {
    if (VEC) return *reinterpret_cast<const uchar4*>(m + (okbits ? off : 0));
    uchar4 k;
    k.x = m[(okbits & 1u) ? off : 0];
    k.y = m[(okbits & 2u) ? off + 1 : 0];
    k.z = m[(okbits & 4u) ? off + 2 : 0];
    k.w = m[(okbits & 8u) ? off + 3 : 0];
    return k;
}

__device__ __forceinline__ float4 finish4(float4 v, unsigned okbits, bool has_mask, uchar4 k, float ms) {
    if (has_mask) {
        v.x = k.x ? v.x * ms : 0.f;
        v.y = k.y ? v.y * ms : 0.f;
        v.z = k.z ? v.z * ms : 0.f;
        v.w = k.w ? v.w * ms : 0.f;
    }
    v.x = (okbits & 1u) ? v.x : 0.f;
    v.y = (okbits & 2u) ? v.y : 0.f;
    v.z = (okbits & 4u) ? v.z : 0.f;
    v.w = (okbits & 8u) ? v.w : 0.f;
    return v;
}

template <class TC, bool A_KC, bool B_KC, bool AVEC, bool BVEC, bool MASKS = false>
struct MainLoop {
    static constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK;
    static constexpr int LDA = TC::template lda<A_KC>();
    static constexpr int LDB = TC::template ldb<B_KC>();
    static constexpr int ASZ = TC::template a_elems<A_KC>();
    static constexpr int BSZ = TC::template b_elems<B_KC>();

    struct Stage {  // one K-slab of both operands in flight in registers
        float4 a[TC::AV], b[TC::BV];
        uchar4 am[TC::AV], bm[TC::BV];
        unsigned aok[TC::AV], bok[TC::BV];
    };

    __device__ static __forceinline__ void gload(const OpA& a, const OpB& b, int k0, int K, Stage& st) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            size_t off;
            int nvalid;
            if (A_KC) {
                const int row = v / (BK / 4), kq = v % (BK / 4);
                const int gm = a.m0 + row, k = k0 + 4 * kq;
                nvalid = (gm < a.M) ? (K - k) : 0;
                off = (size_t)gm * a.ld + k;
            } else {
                const int kk = v / (BM / 4), mq = v % (BM / 4);
                const int gk = k0 + kk, gm = a.m0 + 4 * mq;
                nvalid = (gk < K) ? (a.M - gm) : 0;
                off = (size_t)gk * a.ld + gm;
            }
            st.a[i] = load4<AVEC>(a.p, off, nvalid, st.aok[i]);
            if (MASKS && a.mask) st.am[i] = loadmask4<AVEC>(a.mask, off, st.aok[i]);
        }
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            size_t off;
            int nvalid, idx, j;
            if (B_KC) {
                const int nl = v / (BK / 4), kq = v % (BK / 4);
                bmap<TC::NSEG>(b, nl, idx, j);
                const int k = k0 + 4 * kq;
                nvalid = (j < b.seg_len) ? (K - k) : 0;
                off = (size_t)idx * b.ld + k;
            } else {
                const int kk = v / (BN / 4), nq = v % (BN / 4);
                bmap<TC::NSEG>(b, 4 * nq, idx, j);
                const int gk = k0 + kk;
                nvalid = (gk < K) ? (b.seg_len - j) : 0;
                off = (size_t)gk * b.ld + idx;
            }
            st.b[i] = load4<BVEC>(b.p, off, nvalid, st.bok[i]);
            if (MASKS && b.mask) st.bm[i] = loadmask4<BVEC>(b.mask, off, st.bok[i]);
        }
    }

    __device__ static __forceinline__ void sstore(const OpA& a, const OpB& b, float* As, float* Bs, const Stage& st) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < TC::AV; ++i) {
            const int v = tid + i * TC::NT;
            const float4 r = finish4(st.a[i], st.aok[i], MASKS && a.mask != nullptr, st.am[i], a.mscale);
            if (A_KC) {
                const int row = v / (BK / 4), kq = v % (BK / 4);
                float2* d = reinterpret_cast<float2*>(As + row * LDA + 4 * kq);
                d[0] = make_float2(r.x, r.y);
                d[1] = make_float2(r.z, r.w);
            } else {
                const int kk = v / (BM / 4), mq = v % (BM / 4);
                *reinterpret_cast<float4*>(As + kk * LDA + 4 * mq) = r;
            }
        }
#pragma unroll
        for (int i = 0; i < TC::BV; ++i) {
            const int v = tid + i * TC::NT;
            const float4 r = finish4(st.b[i], st.bok[i], MASKS && b.mask != nullptr, st.bm[i], b.mscale);
            if (B_KC) {
                const int nl = v / (BK / 4), kq = v % (BK / 4);
                float2* d = reinterpret_cast<float2*>(Bs + nl * LDB + 4 * kq);
                d[0] = make_float2(r.x, r.y);
                d[1] = make_float2(r.z, r.w);
            } else {
                const int kk = v / (BN / 4), nq = v % (BN / 4);
                *reinterpret_cast<float4*>(Bs + kk * LDB + 4 * nq) = r;
            }
        }
    }

    static constexpr int KH = (BK / 8) * 4;      // k-steps are split in two halves per slab
    static constexpr int NS0 = KH / 4, NS1 = (BK - KH) / 4;

    template <int NS>
    struct Frag {  // fragments of NS consecutive k-steps
        float a[NS][TC::MI], b[NS][TC::NI];
    };

    template <int NS>
    __device__ static __forceinline__ void read_frags(const float* Ab, const float* Bb, int k0, Frag<NS>& f) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kk = k0 + 4 * s;
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi) f.a[s][mi] = A_KC ? Ab[mi * 16 * LDA + kk] : Ab[kk * LDA + mi * 16];
#pragma unroll
            for (int ni = 0; ni < TC::NI; ++ni) f.b[s][ni] = B_KC ? Bb[ni * 16 * LDB + kk] : Bb[kk * LDB + ni * 16];
        }
    }

    template <int NS>
    __device__ static __forceinline__ void mfmas(const Frag<NS>& f, f32x4 (&acc)[TC::MI][TC::NI]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mi = 0; mi < TC::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s][mi], f.b[s][ni], acc[mi][ni], 0, 0, 0);
    }

    // One slab: [reads half0] [reads half1] | MFMAs half0 | LDS writes of the next slab | MFMAs half1.
    // The sched_barriers pin that order: left alone, hipcc sinks every ds_read next to its first use and reuses the
    // same two VGPRs (read -> lgkmcnt(0) -> 4 MFMAs -> read ...), exposing the LDS latency 8 times per slab
    // (measured: 40-45 % MFMA utilisation in steady state).  With both halves' fragments in flight the counted
    // lgkmcnt waits fall behind >= 16 queued MFMAs.
    template <bool STORE>
    __device__ static __forceinline__ void slab(const OpA& a, const OpB& b, const float* Ac, const float* Bc, float* An,
                                                float* Bn, const Stage& st, f32x4 (&acc)[TC::MI][TC::NI]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / TC::WN, wn = wave % TC::WN;
        const int l15 = lane & 15, lq = lane >> 4;
        const float* Ab = A_KC ? Ac + (wm * TC::WTM + l15) * LDA + lq : Ac + lq * LDA + wm * TC::WTM + l15;
        const float* Bb = B_KC ? Bc + (wn * TC::WTN + l15) * LDB + lq : Bc + lq * LDB + wn * TC::WTN + l15;
        Frag<NS0> f0;
        Frag<NS1> f1;
        if (CPG_ABLATE & 2) {  // keep the registers live and opaque, but do not touch the LDS
#pragma unroll
            for (int s = 0; s < NS0; ++s) {
#pragma unroll
                for (int mi = 0; mi < TC::MI; ++mi) { f0.a[s][mi] = acc[mi][0][0]; f1.a[s][mi] = acc[mi][0][1]; }
#pragma unroll
                for (int ni = 0; ni < TC::NI; ++ni) { f0.b[s][ni] = acc[0][ni][2]; f1.b[s][ni] = acc[0][ni][3]; }
            }
        } else {
            read_frags<NS0>(Ab, Bb, 0, f0);
            read_frags<NS1>(Ab, Bb, KH, f1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas<NS0>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (STORE && !(CPG_ABLATE & 1)) sstore(a, b, An, Bn, st);
        __builtin_amdgcn_sched_barrier(0);
        mfmas<NS1>(f1, acc);
    }

    // acc[mi][ni] += A_tile * B_tile over the whole K range.  Uses TC::smem_floats<A_KC,B_KC>() floats of dynamic LDS.
    // The two LDS buffers are addressed with compile-time offsets from the __shared__ symbol itself (2x unrolled slab
    // loop): runtime-selected buffer pointers degrade to flat_* accesses whose waits also drain the global prefetch.
    __device__ static __forceinline__ void run(const OpA& a, const OpB& b, int K, f32x4 (&acc)[TC::MI][TC::NI]) {
        extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
        float* const A0 = cpg_smem;
        float* const A1 = cpg_smem + ASZ;
        float* const B0 = cpg_smem + 2 * ASZ;
        float* const B1 = cpg_smem + 2 * ASZ + BSZ;
        Stage st;
        const int KT = (K + BK - 1) / BK;
        gload(a, b, 0, K, st);
        sstore(a, b, A0, B0, st);
        __syncthreads();
#if CPG_LOOP_UNROLL2
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 1 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, (kt + 1) * BK, K, st);
                slab<true>(a, b, A0, B0, A1, B1, st, acc);
            } else {
                slab<false>(a, b, A0, B0, A1, B1, st, acc);
            }
            if (!(CPG_ABLATE & 4)) __syncthreads();
            if (kt + 1 >= KT) break;
            if (kt + 2 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, (kt + 2) * BK, K, st);
                slab<true>(a, b, A1, B1, A0, B0, st, acc);
            } else {
                slab<false>(a, b, A1, B1, A0, B0, st, acc);
            }
            if (!(CPG_ABLATE & 4)) __syncthreads();
        }
#else
        // single loop body; the buffer toggle is integer arithmetic on offsets from the __shared__ symbol (keeps the
        // accesses ds_*), and avoids the accumulator copies hipcc inserts between the two halves of an unrolled pair
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            const float* Ac = cpg_smem + cur * ASZ;
            const float* Bc = cpg_smem + 2 * ASZ + cur * BSZ;
            float* An = cpg_smem + (cur ^ 1) * ASZ;
            float* Bn = cpg_smem + 2 * ASZ + (cur ^ 1) * BSZ;
            if (kt + 1 < KT) {
                if (!(CPG_ABLATE & 1)) gload(a, b, (kt + 1) * BK, K, st);
                slab<true>(a, b, Ac, Bc, An, Bn, st, acc);
            } else {
                slab<false>(a, b, Ac, Bc, An, Bn, st, acc);
            }
            if (!(CPG_ABLATE & 4)) __syncthreads();
        }
#endif
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
