// f16-pair hand-off between the launches of a BPTT chain (direct-to-LDS backward steps of csrc/gru.hip and csrc/lstm.hip, PREC 3 of
// DlLoop in gemm_core.h).  A launch leaves the G recurrent blocks of its gate gradients dG [B, G H] once more, as the NEXT launch's
// A operand: planes[B][2 G H] f16, k-groups of 32 in the order (32-column group, block), each 128 bytes = [32 hi | 32 lo] of the
// values times 2^e, e = ex[(row / 32) * (H / 32) + group] chosen so that the largest magnitude of the 32 rows x 32 columns x G
// blocks lands in [2^13, 2^14) (INT_MAX: all zero).  W_hh^T is laid out the same way (cpg_pair_w) times 2^e_w, e_w = the matrix' own exponent
// (weight_exp_from_parts below).  The consumer
// rescales its accumulators (an exact power of two) where the exponent changes along k and skips groups that are all zero or more
// than 2^60 below the largest one.  ex_min[group] keeps the smallest exponent any launch of the sequence recorded: the column scale
// of the dW_hh product on f16 pairs (gemm.hip, PREC 8).
#pragma once
#include "gemm_core.h"
#include <limits.h>

// (pair_pow2, WX_PARTS, weight_exp_from_parts, cpg_weight_absmax: gemm_core.h - every engine that splits weights shares them)

// ---- consumer side: exponents of this wave's 32 rows (lane l holds group l's), the rescaling hook of DlLoop::run, the final factor
template <int MI, int NI, int G>
struct PairConsumer {
    int ev, e_ref, e_cur;
    bool live;
    __device__ __forceinline__ void init(const int* ex_row /* this wave's 32-row block */, int ngroups, int lane) {
        ev = lane < ngroups ? ex_row[lane] : INT_MAX;
        e_ref = ev;   // the largest-magnitude group's exponent (the smallest)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) e_ref = min(e_ref, __shfl_xor(e_ref, o));
        e_cur = e_ref == INT_MAX ? 0 : e_ref;
        live = false;
    }
    // ahead of slab kt (G slabs per 32-column group): true = multiply this slab
    __device__ __forceinline__ bool pre(int kt, f32x4 (&acc)[MI][NI]) {
        const int gi = kt / G;
        if (kt - G * gi == 0) {
            const int e = __builtin_amdgcn_readlane(ev, gi);
            live = e != INT_MAX && e - e_ref <= 60;
            if (live && e != e_cur) {
                const float f = pair_pow2(e - e_cur);   // |e - e_cur| <= 60
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] *= f;
                e_cur = e;
            }
        }
        return live;
    }
    // accumulators hold (sum) x 2^(e_cur + w_exp): back to the unit of the result, two exact factors (each a normal f32)
    __device__ __forceinline__ void finish(f32x4 (&acc)[MI][NI], int w_exp) const {
        const float f0 = pair_pow2(-e_cur), f1 = pair_pow2(-w_exp);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = acc[mi][ni] * f0 * f1;
    }
};

// ---- producer side.  A wave of the tile owns 32 rows x WCOLS columns; vmax = the largest |value| of its G
// blocks (per lane on entry).  Returns the exponent of the wave's 32 x 32 group (two waves share one when BN = 32: `red` = 4 floats
// of LDS that every wave is done with, the barriers are the workgroup's), records it in ex_out / ex_min.
// WCOLS: columns of a wave tile - 32 (the wave owns its group) or 16: waves w and w ^ 1 (neighbours along the columns) share a group
template <int WCOLS>
__device__ __forceinline__ int pair_group_exponent(float vmax, float* red, int wave, int lane, int* ex_out_entry, int* ex_min_entry) {
    static_assert(WCOLS == 32 || WCOLS == 16, "a wave tile is one 32-column exponent group or half of one");
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if constexpr (WCOLS == 16) {
        __syncthreads();
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        vmax = fmaxf(red[wave & ~1], red[wave | 1]);
    }
    int e = INT_MAX;
    if (vmax > 0.f) {
        int fe = 0;
        if (vmax < 3.0e38f) { (void)frexpf(vmax, &fe); e = max(-100, min(100, 14 - fe)); }
        else e = 0;   // an infinity among the values: unscaled, it (and any NaN) reaches the planes as it is
    }
    if (lane == 0 && (WCOLS == 32 || (wave & 1) == 0)) {
        *ex_out_entry = e;
        // (the table only decreases: a stale read can only cause a redundant atomic)
        if (e != INT_MAX && e < __hip_atomic_load(ex_min_entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(ex_min_entry, e);
    }
    return e;
}

// four consecutive columns (col .. col + 3) of block q of one row: 8 bytes of high halves, 8 bytes of low halves
template <int G>
__device__ __forceinline__ void pair_store4(uint16_t* planes, size_t row, int H, int col, int q, const f32x4 v) {
    uint16_t* const d = planes + row * 2 * G * H + (size_t)(G * (col / 32) + q) * 64 + (col & 31);
    uint32_t h0, l0, h1, l1;
    split2h_pair(v[0], v[1], h0, l0);
    split2h_pair(v[2], v[3], h1, l1);
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(l0, l1);
}

// scratch of one direction: two plane images (ping-pong over the steps), their two exponent tables, the column minima
// (+ WX_PARTS ints behind ex_min: the partial maxima of W_hh, see weight_exp_from_parts)
static inline size_t pair_scratch_bytes(int rows, int H, int G) {
    return 2 * ((size_t)rows * 2 * G * H * sizeof(uint16_t) + (size_t)(rows / 32) * (H / 32) * sizeof(int)) + (size_t)(H / 32 + WX_PARTS) * sizeof(int);
}
static inline void pair_split(void* scratch, int B, int H, int G, uint16_t* (&pp)[2], int* (&ex)[2], int*& ex_min) {
    const size_t plane = (size_t)B * 2 * G * H;
    pp[0] = (uint16_t*)scratch;
    pp[1] = pp[0] + plane;
    ex[0] = (int*)(pp[1] + plane);
    ex[1] = ex[0] + (size_t)(B / 32) * (H / 32);
    ex_min = ex[1] + (size_t)(B / 32) * (H / 32);
}
// ---- "all-T planes" form (round 5): the plane images of EVERY step of a sequence are kept - [T][B][2 G H] f16 with their exponent
// tables [T][B/32][H/32] - and become the only copy of the G recurrent gate-gradient blocks (the f32 dG of those blocks is not
// written): the next BPTT launch reads step t+1's image as before, the dW_hh product reads all of them as its A operand through
// LDS-DMA and transposing LDS reads with no conversion in its loop (pair_tn.h), the input-side reductions widen them exactly.
// Beside them the backward steps leave the state planes [T][B][2 H] f16 (h_prev of step t, unscaled: |h| <= 1) - the dW_hh
// product's B operand - which they have in registers anyway.
struct ApScratch {
    uint16_t* planes;    // [T][B][2 G H]
    uint16_t* hplanes;   // [T][B][2 H]
    int* ex;             // [T][B/32][H/32]
    int* ex_min;         // [H/32], then WX_PARTS ints: partial maxima of W_hh (wx = ex_min + H/32)
};
static inline size_t ap_scratch_bytes(int T, int B, int H, int G) {
    return (size_t)T * B * 2 * (G + 1) * H * sizeof(uint16_t) + ((size_t)T * (B / 32) * (H / 32) + (size_t)(H / 32) + WX_PARTS) * sizeof(int);
}
static inline ApScratch ap_split(void* scratch, int T, int B, int H, int G) {
    ApScratch a;
    a.planes = (uint16_t*)scratch;
    a.hplanes = a.planes + (size_t)T * B * 2 * G * H;
    a.ex = (int*)(a.hplanes + (size_t)T * B * 2 * H);
    a.ex_min = a.ex + (size_t)T * (B / 32) * (H / 32);
    return a;
}
// W_hh [G H, H] -> f16-pair image of W_hh^T times 2^e_w (csrc/gru.hip), resetting ex_min for a new sequence; wx = ex_min + H/32 receives
// the matrix' partial maxima first (one more launch) and is what the consuming step kernels take e_w from
int cpg_pair_w(const float* w_hh, int G, int H, uint16_t* out, int* ex_min, hipStream_t s);
