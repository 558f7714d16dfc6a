// dW[M, N] = A^T B over R rows with BOTH operands handed over as f16-pair planes in memory (the dW_hh product behind the all-T plane
// form of the BPTT chain: A = the gate-gradient planes the backward steps leave for every t, B = the state planes).  Conversion-free:
// operands go global -> LDS by LDS-DMA (no staging registers, no split in the loop), fragments come out of LDS transposed
// (ds_read_b64_tr_b16), three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block (a_lo b_hi + a_hi b_lo + a_hi b_hi: f32-grade).
//
// Plane image of an operand X [R rows, C columns], C % 32 == 0:  row r = C/32 segments of 128 bytes, segment g =
// [32 high halves | 32 low halves] of columns 32 g .. 32 g + 31 (the layout pair_store4 writes: pair_engine.h).  A's segments
// carry a power-of-two scale per (32-row block, exponent group): values were multiplied by 2^e, e = ex[(r / 32) * ngroups + group(g)]
// (INT_MAX: the segment holds zeros); the kernel brings every segment of a column group to the group's common scale
// 2^emin[group] (emin <= every e of the group: the factor 2^(emin - e) <= 1 is applied to the f16 fragments - exact while the
// product stays normal, the 2^-24 absolute floor of the pair otherwise: what splitting at the common scale would have given) and
// takes 2^emin back out of the output rows.  B is unscaled.
//
// LDS slab (32 rows of R) of a segment: 32 rows x 128 bytes, 16-byte chunk slot s of row r holds the row's chunk s ^ f(r),
// f(r) = ((r >> 1) & 3) << 1 - a source-side swizzle (an LDS-DMA image cannot be padded): the 16 lanes of a transposing read
// touch 4 rows x 32 bytes, rows of equal parity land in distinct bank octets.
#pragma once
#include "gemm_core.h"
#include <limits.h>
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on the clobber list of the LDS-DMA statement below: nothing else in these kernels uses m0

struct PairTnArgs {
    const uint16_t* A; size_t lda;   // plane image of A: elements (uint16) per row = 2 * (columns of the image)
    const int* a_ex;                 // [R / 32][a_groups]: exponent of A's (32-row block, group); null = unscaled
    const int* a_emin;               // [a_groups]: common exponent per group (<= every a_ex of the group); null with a_ex null
    int a_groups, a_seg_per_group;   // segment g of A (counted from A's column 0) belongs to group g / a_seg_per_group; with G =
                                     // a_seg_per_group > 1 the image interleaves G blocks of M / G columns each (segment g = block g % G,
                                     // columns 32 (g / G) ..): output row of image column m is (g % G) (M / G) + 32 (g / G) + m % 32
    const uint16_t* B; size_t ldb;
    float* C; int ldc;               // [S][M][ldc] partial slabs (S = gridDim.z), or the result itself when S == 1
    size_t slab_stride;
    int M, N, R, r_chunk;            // r_chunk % 32 == 0: rows per z
    int accumulate;                  // S == 1 only
};

// NP = 2: f16-pair images (a 128-byte segment = [32 hi | 32 lo] of 32 columns, three f16 MFMAs per block);
// NP = 1: ONE bf16 plane (a segment = 64 columns; one bf16 MFMA per block; no exponents): the bf16 compute mode's dW_hh product on its
//         bf16 gate gradients [R, lda] and a bf16 copy of the states - the same conversion-free loop with half the LDS traffic per column
template <int WM, int WN, int NS, int NP = 2>
struct PairTn {
    static constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    static constexpr int SEGC = NP == 2 ? 32 : 64;              // columns per 128-byte segment
    static constexpr int SA = BM / SEGC, SB = BN / SEGC;        // segments per slab
    static constexpr int STAGE = (SA + SB) * 4096;              // bytes
    static constexpr int NW = WM * WN;
    static constexpr int DMA = (SA + SB) * 4;                   // wave-instructions per slab (8 rows x 128 bytes each)
    static_assert(DMA % NW == 0, "every wave issues the same number of LDS-DMA instructions per slab");
    static constexpr int LPS = DMA / NW;
    static constexpr size_t smem_bytes(int max_blocks) { return (size_t)NS * STAGE + (size_t)max_blocks * SA * sizeof(int); }
};

// ABL (diagnostic builds of tools/micro/wgrad_planes.hip, wrong results): 1 no MFMAs, 2 no fragment reads, 4 no LDS-DMA after the prologue
template <int WM, int WN, int NS, int ABL = 0, int PIPE = 0, int NP = 2>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_tn_kernel(PairTnArgs g) {
    using P = PairTn<WM, WN, NS, NP>;
    constexpr int BM = P::BM, BN = P::BN, SA = P::SA, STAGE = P::STAGE, LPS = P::LPS, NW = P::NW, SEGC = P::SEGC;
    constexpr int WS = 64 / SEGC;   // segments of a wave's 64 columns
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    char* const smem = reinterpret_cast<char*>(cpg_smem);
    const __attribute__((address_space(3))) char* const smem_lds = (const __attribute__((address_space(3))) char*)cpg_smem;
    int bx, by, bz;
    xcd_tile_order(bx, by, bz);
    const int m0 = by * BM, n0 = bx * BN;
    const int r0 = bz * g.r_chunk, rows = min(g.R - r0, g.r_chunk);
    const int KT = rows / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- exponent factors of this workgroup's slabs: fac[kt][seg] as packed f16 pairs (2^(emin - e), 0 for an all-zero segment)
    uint32_t* const fac = reinterpret_cast<uint32_t*>(smem + (size_t)NS * STAGE);
    if (NP == 2 && g.a_ex) {
        for (int i = tid; i < KT * SA; i += P::NT) {
            const int kt = i / SA, sg = i - kt * SA;
            const int grp = (m0 / 32 + sg) / g.a_seg_per_group;
            const int e = g.a_ex[(size_t)(r0 / 32 + kt) * g.a_groups + grp];
            const int em = g.a_emin[grp];
            uint32_t bits = 0;
            if (e != INT_MAX) {
                const int d = max(em - e, -30);
                const _Float16 h = (_Float16)__builtin_bit_cast(float, (unsigned)(127 + d) << 23);
                bits = (uint32_t)__builtin_bit_cast(uint16_t, h);
            }
            fac[i] = bits | (bits << 16);
        }
    }

    // ---- LDS-DMA: instruction i of this wave moves rows 8 j .. 8 j + 7 of one segment (128 bytes per row, chunk slots swizzled)
    const uint16_t* src[LPS];
    int dst[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int u = wave * LPS + i;            // 0 .. DMA-1: (segment, row octet)
        const int sg = u >> 2, j = u & 3;
        const int row = 8 * j + (lane >> 3);
        const int ch = (lane & 7) ^ (((row >> 1) & 3) << 1);
        const bool isA = sg < SA;
        const uint16_t* base = isA ? g.A + (size_t)(m0 / SEGC + sg) * 64 : g.B + (size_t)(n0 / SEGC + (sg - SA)) * 64;
        src[i] = base + (size_t)(r0 + row) * (isA ? g.lda : g.ldb) + 8 * ch;
        dst[i] = sg * 4096 + j * 1024;           // + 16 * lane by the instruction itself
    }
    const size_t a32 = 32 * g.lda, b32 = 32 * g.ldb;
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int sg = (wave * LPS + i) >> 2;
            const uint16_t* p = src[i] + (size_t)kt * (sg < SA ? a32 : b32);
            // LDS-DMA as inline asm: behind the builtin the compiler waits vmcnt(0) in front of EVERY later LDS read that carries no
            // alias scope (the transposing-read builtin never does) - each slab then waited for the slab being prefetched.  The counted
            // s_waitcnt in slab() orders these loads; nothing else in the loop uses vmcnt.
            const uint32_t lds = (uint32_t)(uintptr_t)(smem_lds + stage * STAGE + dst[i]);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(lds) : "memory", "m0");
        }
    };

    // ---- fragment addresses (bytes inside a stage): block b of 16 columns = segment b / 2, half u = b & 1; plane p; k half h
    const int s = lane & 15, q = lane >> 4;
    const int krow = 4 * q + (s >> 2);                       // + 16 h
    const int fsw = ((krow >> 1) & 3) << 1;                  // f(krow) = f(krow + 16)
    auto frag_off = [&](int seg0, int b, int p, int h) {   // block b (16 columns) of the wave's 64 columns, whose first segment is seg0
        const int seg = NP == 2 ? seg0 + (b >> 1) : seg0;
        const int c = (NP == 2 ? (4 * p + 2 * (b & 1) + ((s & 3) >> 1)) : (2 * b + ((s & 3) >> 1))) ^ fsw;
        return seg * 4096 + (krow + 16 * h) * 128 + c * 16 + (s & 1) * 8;
    };
    int offA[4][NP][2], offB[4][NP][2];                      // [16-column block of the wave's 64][plane][k half]
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                offA[b][p][h] = frag_off(wm * WS, b, p, h);
                offB[b][p][h] = frag_off(SA + wn * WS, b, p, h);
            }

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto rd = [&](const char* st, int off0, int off1) {
        if (ABL & 2) return __builtin_bit_cast(cpg_f16x8, acc[0][(off0 >> 4) & 3] + acc[1][(off1 >> 5) & 3]);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + off1));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        return __builtin_bit_cast(cpg_f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    constexpr int AHEAD = NS - 1;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < KT) issue(i, i);
    __syncthreads();   // fac[] visible
    // Software pipeline over slabs: iteration kt reads ALL fragments of slab kt into one register set and multiplies slab kt - 1 out
    // of the other - a wave's LDS reads run under its own MFMAs (the matrix pipe does not idle across the slab barrier, also at one
    // wave per SIMD).  A stage is free for the next LDS-DMA once every wave has READ it: reads of slab kt - 1 are waited for
    // (lgkmcnt(0)) in front of iteration kt's barrier.
    cpg_f16x8 fA[PIPE ? 2 : 1][4][NP], fB[PIPE ? 2 : 1][4][NP];
    uint32_t f2[PIPE ? 2 : 1][2] = {{0x3c003c00u, 0x3c003c00u}};   // [register set][segment of the wave]: 2^(emin - e) twice
    auto load_frags = [&](int buf, int kt, int cur) {
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int p = 0; p < NP; ++p) fB[buf][b][p] = rd(st, offB[b][p][0], offB[b][p][1]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int p = 0; p < NP; ++p) fA[buf][mi][p] = rd(st, offA[mi][p][0], offA[mi][p][1]);
        if (NP == 2 && g.a_ex) {
            f2[buf][0] = fac[kt * SA + wm * 2];
            f2[buf][1] = fac[kt * SA + wm * 2 + 1];
        } else {
            f2[buf][0] = f2[buf][1] = 0x3c003c00u;
        }
    };
    auto multiply = [&](int buf) {
        if (ABL & 1) return;
        if constexpr (NP == 1) {   // one bf16 plane: one MFMA per block
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cpg_bf16x8, fA[buf][mi][0]),
                                                                          __builtin_bit_cast(cpg_bf16x8, fB[buf][ni][0]), acc[mi][ni], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            cpg_f16x8 a0 = fA[buf][mi][0], a1 = fA[buf][mi][NP - 1];
            if (g.a_ex) {   // the segment's power of two, applied where the fragment is consumed (the reads stay wait-free)
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 f = __builtin_bit_cast(h2, f2[buf][mi >> 1]);
                auto sc = [&](cpg_f16x8 v) {
                    uint4 w = __builtin_bit_cast(uint4, v);
                    w.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w.x) * f);
                    w.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w.y) * f);
                    w.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w.z) * f);
                    w.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w.w) * f);
                    return __builtin_bit_cast(cpg_f16x8, w);
                };
                a0 = sc(a0);
                a1 = sc(a1);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, fB[buf][ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, fB[buf][ni][NP - 1], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, fB[buf][ni][0], acc[mi][ni], 0, 0, 0);
        }
    };
    auto slab = [&](int kt, int cur, int refill, int buf) {
        // slab kt has landed once only the loads of the slabs issued after it are outstanding: min(AHEAD - 1, KT - 1 - kt) slabs
        const int after = min(AHEAD - 1, KT - 1 - kt);
        if (AHEAD >= 4 && after == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPS) : "memory");
        else if (AHEAD >= 3 && after == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS) : "memory");
        else if (AHEAD >= 2 && after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + AHEAD < KT && !(ABL & 4)) issue(kt + AHEAD, refill);
        if constexpr (PIPE) {
            load_frags(buf, kt, cur);
            if (kt > 0) multiply(buf ^ 1);
        } else {
            load_frags(0, kt, cur);
            multiply(0);
        }
    };
    static_assert(NS % 2 == 0 || NS == 3, "the slab loop is unrolled over stages x register sets");
    constexpr int UN = NS % 2 == 0 ? NS : 2 * NS;   // stage and register set of every access are compile-time
    int kt = 0;
    for (; kt + UN <= KT; kt += UN) {
#pragma unroll
        for (int i = 0; i < UN; ++i) slab(kt + i, i % NS, (i + NS - 1) % NS, i & 1);
    }
#pragma unroll
    for (int i = 0; i < UN - 1; ++i)
        if (kt + i < KT) slab(kt + i, i % NS, (i + NS - 1) % NS, i & 1);
    if constexpr (PIPE) {
        if (KT > 0) {
            if ((KT - 1) & 1) multiply(1); else multiply(0);
        }
    }

    // ---- epilogue: rows of the output take 2^-emin of their group back out; partial slab z (or the result when not split)
    float* C = g.C + (size_t)bz * g.slab_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int mrow = m0 + wm * 64 + mi * 16 + 4 * q;
        float sc = 1.f;
        const int seg = mrow / 32, G = NP == 2 ? g.a_seg_per_group : 1;
        if (NP == 2 && g.a_ex) {
            const int em = g.a_emin[seg / G];
            sc = __builtin_bit_cast(float, (unsigned)(127 - (em == INT_MAX ? 0 : em)) << 23);
        }
        const int orow = NP == 2 ? (seg % G) * (g.M / G) + 32 * (seg / G) + (mrow & 31) : mrow;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = n0 + wn * 64 + ni * 16 + s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(orow + r) * g.ldc + col;
                float v = acc[mi][ni][r] * sc;
                if (g.accumulate && gridDim.z == 1) v += C[o];
                C[o] = v;
            }
        }
    }
}
