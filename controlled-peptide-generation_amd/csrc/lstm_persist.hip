// Persistent LSTM forward sequence kernel for gfx950: the WHOLE time loop of one direction of one LSTM layer in ONE launch.
//
// The LSTM is an extension (the reference has no LSTM; BASELINE.json's configs name one): torch.nn.LSTM semantics,
// gate row order i,f,g,o; c' = f*c + i*g; h' = o*tanh(c').  Same decomposition as csrc/gru_persist.hip, with the tile shaped for
// four gates:
//   * a workgroup owns CT = 8 hidden units - the i, f, g, o rows of W_hh for them: 32 x H, TWO full 16-column MFMA blocks
//     (block 0 = [i | f], block 1 = [g | o]) - for 512 batch rows and keeps that slice in LDS for the whole sequence as three
//     bf16 planes (32 x H x 6 B = 101 KB at H = 512).  At B = 2048, H = 512: 4 row groups x 64 column tiles = 256 workgroups;
//   * each of its 8 waves (two per SIMD) owns 64 rows; the state operand goes global -> registers -> MFMA A fragments, the time
//     loop has no workgroup barrier; the cell state of the wave's own elements stays in registers;
//   * lane (u = l & 7, half = (l >> 3) & 1) of a 16-lane group holds, after the product, gates (i, g) [half 0] or (f, o) [half 1]
//     of unit u for four rows.  The two halves swap what the other needs with one DPP rotate per value (row_ror:8) so that
//     half 0 runs the cell of rows 0-1 and half 1 that of rows 2-3 of each 4-row group: every lane does cell arithmetic;
//   * the column-tile workgroups of a row tile exchange h_t already split (three bf16 planes in per-step slots laid out
//     [k-block][row][32 k], write-through stores, one arrival counter per row tile in its own 256-byte line, relaxed polls,
//     bounded spins + sticky error word) exactly as the GRU kernel does.
// Arithmetic: the per-step kernel's (same split, same six-term order per block, same cell formulas), so
// tests/test_lstm.py compares the two within f32 summation-order noise.
#include "gemm_core.h"
#include "cpg_internal.h"
#include <stdlib.h>

namespace {

// NG (round 4): 8-unit GROUPS per workgroup.  The LDS holds NP x 32 NG gate rows of W_hh planes: with the three planes of the
// f32-grade mode that is one group (101 KB at H = 512); the bf16 compute mode keeps ONE plane, so FOUR groups fit (32 units, 135 KB).
// The workgroup count stays the chip's (rows per workgroup shrink with it: 512 / NG), and what a workgroup reads of the exchanged
// state per step - rows x H x 2 NP bytes - shrinks by NG: 524 -> 131 KB at H = 512 in the bf16 mode, with a quarter of the producers
// per row tile to wait for.  A wave owns 64 / NG rows and, per group, the two MFMA blocks [i|f], [g|o] described above.
constexpr int L_WAVES = 8;
// cell nonlinearities: 0 the library forms of the per-step kernels, 1 (default) the hardware exp2 / rcp forms of cpg_common.h
#ifndef CPG_PERSIST_FAST_CELL
#define CPG_PERSIST_FAST_CELL 1
#endif
__device__ __forceinline__ float l_sigmoid(float x) { return CPG_PERSIST_FAST_CELL ? cell_sigmoidf(x) : sigmoidf_(x); }
__device__ __forceinline__ float l_tanh(float x) { return CPG_PERSIST_FAST_CELL ? cell_tanhf(x) : tanhf(x); }
#ifndef CPG_LSTM_PERSIST_DEPTH
#define CPG_LSTM_PERSIST_DEPTH 2
#endif
constexpr int L_DEPTH = CPG_LSTM_PERSIST_DEPTH;  // register ring over k-blocks
constexpr int L_MIN_WROWS = 16;      // smallest row tile (NG = 4): the arrival counters are laid out for it
constexpr int L_CNT_STRIDE = 64;     // words between arrival counters (one 256-byte line each)
constexpr int L_TBW = 16;
constexpr unsigned L_SPIN_LIMIT = 400000u;

typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));

struct LFwdArgs {
    const float* w_hh;     // [4H,H]
    const float* b_hh;     // [4H]
    const int32_t* tok;    // [T,B] or null
    const float* tab;      // [V,4H] or null
    const float* rowc;     // [B,4H] or null
    const float* dense;    // [T,B,4H] or null
    float* hs;             // [(T+1),B,H]
    float* cs;             // [(T+1),B,H]
    float* gates;          // [T,4,B,H] or null
    unsigned* cnt;
    unsigned* err;
    unsigned* err_host;   // host-mapped copy of the sticky error word, or null
    uint16_t* xch;         // [(T+1) slots][3 planes][H/32 k-blocks][B][32] bf16
    int T, B, H, reverse, groups, S;
};

__device__ __forceinline__ bool l_wait_ge(unsigned* p, unsigned target, unsigned* err, unsigned* err_host, bool& dead) {
    if (dead) return false;
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > L_SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (err_host) __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            dead = true;
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

// value of lane (l + 8) % 16 of the same 16-lane row: DPP row_ror:8
__device__ __forceinline__ float ror8(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ float l_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

// Cell-layout values (lane (u, half, lq) holds rows 4 lq + 2 half + {0,1} of unit u of a 16 x 8 tile) -> row layout: lane
// (row l >> 2, columns 4 (l & 3) .. +3), valid for (l & 3) < 2, through the wave's own 1 KB LDS buffer.
__device__ __forceinline__ f32x4 cell_to_rows(float* tb, float v0, float v1, int lane) {
    const int u = lane & 7, half = (lane >> 3) & 1, lq = lane >> 4;
    tb[(4 * lq + 2 * half) * L_TBW + u] = v0;
    tb[(4 * lq + 2 * half + 1) * L_TBW + u] = v1;
    return *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * L_TBW + 4 * (lane & 3));
}

template <int NP, int NG>
__global__ __launch_bounds__(L_WAVES * 64, 2) void lstm_seq_fwd_persist_kernel(LFwdArgs a) {
    constexpr int L_CT = 8 * NG;            // hidden units per workgroup
    constexpr int L_NC = 32 * NG;           // gate columns per workgroup: two MFMA blocks per group
    constexpr int L_WROWS = 64 / NG;        // rows per wave
    constexpr int L_MI = L_WROWS / 16;
    constexpr int L_HM = L_MI >= 2 ? 2 : 1; // row blocks per product pass
    static_assert(NG == 1 || NG == 2 || NG == 4, "1, 2 or 4 groups of 8 hidden units");
    extern __shared__ __attribute__((aligned(16))) uint32_t lpsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x % a.groups, ct = blockIdx.x / a.groups;
    const int H = a.H, B = a.B, T = a.T, S = a.S;
    const int j0 = ct * L_CT;
    const int NCT = H / L_CT, KB = H / 32;
    const int PLW = L_NC * S;
    uint32_t* const planes = lpsm;
    float* const tb = reinterpret_cast<float*>(lpsm + NP * PLW) + wave * (16 * L_TBW);

    // ---- W_hh slice -> bf16 planes in LDS, once per sequence: plane[c = group*32 + gate*8 + u][k pair]
    // NP = 2 (f16 pair, gemm_core.h): the slice is scaled by a power of two so that its largest magnitude lands in [2^13, 2^14), the
    // accumulators by the inverse - as in csrc/gru_persist.hip
    float wscale = 1.f, descale = 1.f;
    if constexpr (NP == 2) {
        float m = 0.f;
        for (int idx = tid; idx < L_NC * (H / 2); idx += L_WAVES * 64) {
            const int c = idx / (H / 2), kp = idx - c * (H / 2);
            const float2 v = *reinterpret_cast<const float2*>(a.w_hh + ((size_t)(((c >> 3) & 3) * H + j0 + 8 * (c >> 5) + (c & 7))) * H + 2 * kp);
            m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* const red = reinterpret_cast<float*>(lpsm);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < L_WAVES; ++w) m = fmaxf(m, red[w]);
        __syncthreads();
        int e = 0;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);
        e = max(-100, min(100, 14 - e));
        wscale = ldexpf(1.f, e);
        descale = ldexpf(1.f, -e);
    }
    for (int idx = tid; idx < L_NC * (H / 2); idx += L_WAVES * 64) {
        const int c = idx / (H / 2), kp = idx - c * (H / 2);
        const float2 v = *reinterpret_cast<const float2*>(a.w_hh + ((size_t)(((c >> 3) & 3) * H + j0 + 8 * (c >> 5) + (c & 7))) * H + 2 * kp);
        uint32_t w0, w1 = 0, w2 = 0;
        if (NP == 3) split3_pair(v.x, v.y, w0, w1, w2);
        else if (NP == 2) split2h_pair(v.x * wscale, v.y * wscale, w0, w1);
        else w0 = cvt_pk_bf16(v.x, v.y);
        planes[c * S + kp] = w0;
        if (NP >= 2) planes[PLW + c * S + kp] = w1;
        if (NP == 3) planes[2 * PLW + c * S + kp] = w2;
    }
    __syncthreads();

    const int rt = g * L_WAVES + wave;   // row tile of this wave
    const int row0 = rt * L_WROWS;
    if (row0 >= B) return;
    const int l15 = lane & 15, lq = lane >> 4;
    const int u = lane & 7, half = (lane >> 3) & 1;
    const int col = j0 + u;                       // hidden unit of this lane's cell elements IN GROUP 0 (group gp: + 8 gp)
    const int srow = lane >> 2, scq = lane & 3;   // row-layout coordinates after cell_to_rows
    const size_t BH = (size_t)B * H;

    // exchange rows are padded to an even count: a k-block of a plane then starts on a 128-byte line.  (With odd B a line held the
    // last row of one k-block and row 0 of the next - rows of two row tiles with their own arrival counters: the reader of the last
    // tile cached row 0's chunk before its producers had written it, and row 0's tile then read the stale copy.)
    const unsigned Bx = ((unsigned)B + 1u) & ~1u;
    const unsigned plane_bytes = (unsigned)((size_t)Bx * H * 2), kb_bytes = Bx * 64u;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(a.xch, 0, (unsigned)(T + 1) * 3u * plane_bytes, 0x00020000);
    bool dead = false;

    // per-lane constants of the cell: rows 16 mi + 4 lq + 2 half + e (e = 0, 1), unit u, all four gates
    float rc[NG][L_MI][2][4], cst[NG][L_MI][2], bh[NG][4];
    const size_t slot0 = (size_t)(a.reverse ? T : 0) * BH;
#pragma unroll
    for (int gp = 0; gp < NG; ++gp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bh[gp][q] = a.b_hh[q * H + col + 8 * gp];
#pragma unroll
        for (int mi = 0; mi < L_MI; ++mi)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = min(row0 + 16 * mi + 4 * lq + 2 * half + e, B - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) rc[gp][mi][e][q] = a.rowc ? a.rowc[(size_t)row * 4 * H + q * H + col + 8 * gp] : 0.f;
                cst[gp][mi][e] = a.cs[slot0 + (size_t)row * H + col + 8 * gp];
            }
    }

    // row-layout publish of a group's 8 columns (16 bytes per plane): the lane with (l & 3) == 0 collects its neighbour's four columns
    auto publish = [&](const f32x4 v, int row, unsigned slot_off, int jg) {
        const float n0 = l_xor1(v[0]), n1 = l_xor1(v[1]), n2 = l_xor1(v[2]), n3 = l_xor1(v[3]);
        if (scq == 0 && row < B) {
            uint32_t w0[4], w1[4], w2[4];
            if (NP == 2) {
                split2h_pair(v[0], v[1], w0[0], w1[0]);
                split2h_pair(v[2], v[3], w0[1], w1[1]);
                split2h_pair(n0, n1, w0[2], w1[2]);
                split2h_pair(n2, n3, w0[3], w1[3]);
            } else if (NP == 3) {
                split3_pair(v[0], v[1], w0[0], w1[0], w2[0]);
                split3_pair(v[2], v[3], w0[1], w1[1], w2[1]);
                split3_pair(n0, n1, w0[2], w1[2], w2[2]);
                split3_pair(n2, n3, w0[3], w1[3], w2[3]);
            } else {
                w0[0] = cvt_pk_bf16(v[0], v[1]); w0[1] = cvt_pk_bf16(v[2], v[3]);
                w0[2] = cvt_pk_bf16(n0, n1); w0[3] = cvt_pk_bf16(n2, n3);
            }
            const int voff = row * 64 + (jg & 31) * 2;
            const unsigned off = slot_off + (unsigned)(jg >> 5) * kb_bytes;
            __builtin_amdgcn_raw_buffer_store_b128(lu32x4{w0[0], w0[1], w0[2], w0[3]}, rx, voff, off, 16);   // 16 = sc1: write-through
            if (NP >= 2) __builtin_amdgcn_raw_buffer_store_b128(lu32x4{w1[0], w1[1], w1[2], w1[3]}, rx, voff, off + plane_bytes, 16);
            if (NP == 3) __builtin_amdgcn_raw_buffer_store_b128(lu32x4{w2[0], w2[1], w2[2], w2[3]}, rx, voff, off + 2 * plane_bytes, 16);
        }
    };

    // h0 enters the exchange like any step's output: slot 0, arrival #1
#pragma unroll
    for (int gp = 0; gp < NG; ++gp)
#pragma unroll
        for (int mi = 0; mi < L_MI; ++mi) {
            const int row = row0 + 16 * mi + srow;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (scq < 2) v = *reinterpret_cast<const f32x4*>(a.hs + slot0 + (size_t)min(row, B - 1) * H + j0 + 8 * gp + 4 * scq);
            publish(v, row, 0u, j0 + 8 * gp);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(a.cnt + rt * L_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    int aoff[L_MI];
#pragma unroll
    for (int mi = 0; mi < L_MI; ++mi) aoff[mi] = min(row0 + 16 * mi + l15, B - 1) * 64 + 16 * lq;
    const uint32_t* const bbase = planes + l15 * S + 4 * lq;

    for (int p = 0; p < T; ++p) {
        const int tt = a.reverse ? T - 1 - p : p;
        const unsigned in_off = (unsigned)p * 3u * plane_bytes, out_off = (unsigned)(p + 1) * 3u * plane_bytes;

        // input-side pre-activations of this step (independent of the recurrence, fetched before the wait)
        float gi[NG][L_MI][2][4];
#pragma unroll
        for (int mi = 0; mi < L_MI; ++mi)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = min(row0 + 16 * mi + 4 * lq + 2 * half + e, B - 1);
                const float* tt_ = a.tok ? a.tab + (size_t)a.tok[(size_t)tt * B + row] * 4 * H + col : nullptr;
                const float* td_ = a.dense ? a.dense + ((size_t)tt * B + row) * 4 * H + col : nullptr;
#pragma unroll
                for (int gp = 0; gp < NG; ++gp) {
                    float x[4] = {rc[gp][mi][e][0], rc[gp][mi][e][1], rc[gp][mi][e][2], rc[gp][mi][e][3]};
                    if (tt_) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] += tt_[q * H + 8 * gp];
                    }
                    if (td_) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] += td_[q * H + 8 * gp];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) gi[gp][mi][e][q] = x[q];
                }
            }

        l_wait_ge(a.cnt + rt * L_CNT_STRIDE, (unsigned)(NCT * (p + 1)), a.err, a.err_host, dead);

        // ---- two passes over the wave's 64 rows (32 rows = 2 row blocks each): product, then the cell of those rows.  (One pass
        // over all four row blocks needs 96 registers of operand ring + 32 of accumulators next to the 72 of per-row constants:
        // it spilled.)  acc[m][blk] = h_prev[rows, H] . W_hh[gate rows of 8 units, H]^T  (blk 0 = [i|f], 1 = [g|o])
        float* const hout = a.hs + (size_t)(a.reverse ? tt : tt + 1) * BH;
        float* const cout = a.cs + (size_t)(a.reverse ? tt : tt + 1) * BH;
#pragma unroll
        for (int hp = 0; hp < L_MI / L_HM; ++hp) {
            f32x4 acc[L_HM][2 * NG];
#pragma unroll
            for (int m = 0; m < L_HM; ++m)
#pragma unroll
                for (int q = 0; q < 2 * NG; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            lu32x4 buf[L_DEPTH][L_HM][NP];
            auto load = [&](lu32x4 (&b)[L_HM][NP], int kb) {
#pragma unroll
                for (int m = 0; m < L_HM; ++m)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        b[m][pl] = __builtin_amdgcn_raw_buffer_load_b128(rx, aoff[hp * L_HM + m], in_off + pl * plane_bytes + kb * kb_bytes, 0);
            };
            auto compute = [&](const lu32x4 (&bf)[L_HM][NP], int kb) {
                cpg_bf16x8 fb[2 * NG][NP];
#pragma unroll
                for (int q = 0; q < 2 * NG; ++q)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        fb[q][pl] = *reinterpret_cast<const cpg_bf16x8*>(bbase + pl * PLW + q * 16 * S + kb * 16);
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
                if constexpr (NP == 2) {
                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int m = 0; m < L_HM; ++m)
#pragma unroll
                            for (int q = 0; q < 2 * NG; ++q)
                                acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cpg_f16x8, bf[m][HA[t]]),
                                                                                   __builtin_bit_cast(cpg_f16x8, fb[q][HB[t]]), acc[m][q], 0, 0, 0);
                    return;
                }
#pragma unroll
                for (int t = (NP == 3 ? 0 : 5); t < 6; ++t)
#pragma unroll
                    for (int m = 0; m < L_HM; ++m)
#pragma unroll
                        for (int q = 0; q < 2 * NG; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cpg_bf16x8, bf[m][NP == 3 ? TA[t] : 0]),
                                                                                fb[q][NP == 3 ? TB[t] : 0], acc[m][q], 0, 0, 0);
            };
#pragma unroll
            for (int d = 0; d < L_DEPTH - 1; ++d)
                if (d < KB) load(buf[d], d);
            for (int kb = 0; kb < KB; kb += L_DEPTH) {
#pragma unroll
                for (int d = 0; d < L_DEPTH; ++d) {
                    if (kb + d < KB) {
                        if (kb + d + L_DEPTH - 1 < KB) load(buf[(d + L_DEPTH - 1) % L_DEPTH], kb + d + L_DEPTH - 1);
                        compute(buf[d], kb + d);
                    }
                }
            }

            // ---- cell.  Lane (u, half) holds, per 4-row group: block 0 = i (half 0) | f (half 1), block 1 = g (half 0) | o (half 1)
            // for rows 0..3.  Half 0 runs rows 0,1 and half 1 rows 2,3: each sends the partner the two gates it holds of the
            // partner's rows and receives the two it lacks of its own (one row_ror:8 per value).
#pragma unroll
            for (int m = 0; m < L_HM; ++m)
#pragma unroll
            for (int gp = 0; gp < NG; ++gp) {
                if constexpr (NP == 2) {
                    acc[m][2 * gp] *= descale;
                    acc[m][2 * gp + 1] *= descale;
                }
                const int mi = hp * L_HM + m;
                float ig[2], fg[2], gg[2], og[2], hv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // what this lane sends: its block values of the PARTNER's rows (half 0 -> rows 2,3 ; half 1 -> rows 0,1)
                    const float s0 = half ? acc[m][2 * gp][e] : acc[m][2 * gp][2 + e];
                    const float s1 = half ? acc[m][2 * gp + 1][e] : acc[m][2 * gp + 1][2 + e];
                    const float r0 = ror8(s0), r1 = ror8(s1);   // partner's block 0 / block 1 value of MY row e
                    const float m0 = half ? acc[m][2 * gp][2 + e] : acc[m][2 * gp][e];
                    const float m1 = half ? acc[m][2 * gp + 1][2 + e] : acc[m][2 * gp + 1][e];
                    const float pi = half ? r0 : m0, pf = half ? m0 : r0, pg = half ? r1 : m1, po = half ? m1 : r1;
                    ig[e] = l_sigmoid(gi[gp][mi][e][0] + (pi + bh[gp][0]));
                    fg[e] = l_sigmoid(gi[gp][mi][e][1] + (pf + bh[gp][1]));
                    gg[e] = l_tanh(gi[gp][mi][e][2] + (pg + bh[gp][2]));
                    og[e] = l_sigmoid(gi[gp][mi][e][3] + (po + bh[gp][3]));
                    const float cn = fg[e] * cst[gp][mi][e] + ig[e] * gg[e];
                    cst[gp][mi][e] = cn;
                    hv[e] = og[e] * l_tanh(cn);
                }
                // ---- publish h_t (split planes, write-through), then the f32 slabs and the saved gates, all as 16-byte row accesses
                const int row = row0 + 16 * mi + srow;
                const f32x4 hrow = cell_to_rows(tb, hv[0], hv[1], lane);
                if (p + 1 < T) publish(hrow, row, out_off, j0 + 8 * gp);
                const f32x4 crow = cell_to_rows(tb, cst[gp][mi][0], cst[gp][mi][1], lane);
                const bool st = scq < 2 && row < B;
                const size_t o = (size_t)row * H + j0 + 8 * gp + 4 * scq;
                if (st) {
                    *reinterpret_cast<f32x4*>(hout + o) = hrow;
                    *reinterpret_cast<f32x4*>(cout + o) = crow;
                }
                if (a.gates) {
                    float* const gb = a.gates + (size_t)tt * 4 * BH;
                    const f32x4 v0 = cell_to_rows(tb, ig[0], ig[1], lane);
                    const f32x4 v1 = cell_to_rows(tb, fg[0], fg[1], lane);
                    const f32x4 v2 = cell_to_rows(tb, gg[0], gg[1], lane);
                    const f32x4 v3 = cell_to_rows(tb, og[0], og[1], lane);
                    if (st) {
                        __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(gb + o));
                        __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(gb + BH + o));
                        __builtin_nontemporal_store(v2, reinterpret_cast<f32x4*>(gb + 2 * BH + o));
                        __builtin_nontemporal_store(v3, reinterpret_cast<f32x4*>(gb + 3 * BH + o));
                    }
                }
            }
        }
        // the arrival must follow the exchange stores only; the wave's in-order memory pipe drains the slab stores with them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(a.cnt + rt * L_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the launch leaves its arrival counters at zero for the next one (as gru_persist.hip: the last wave of a row tile to sign off
    // zeroes the counter and the sign-off word behind it; no memset node in front of the launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        unsigned* const c = a.cnt + rt * L_CNT_STRIDE;
        if (__hip_atomic_fetch_add(c + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)NCT - 1u) {
            __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int l_plane_stride_words(int H) {
    int s = H / 2;
    while (s % 64 != 8) ++s;
    return s;
}
size_t l_lds_bytes(int H, int np, int ng) { return ((size_t)np * 32 * ng * l_plane_stride_words(H) + L_WAVES * 16 * L_TBW) * 4; }
size_t l_cnt_words(int B) { return (size_t)cdiv(B, L_MIN_WROWS) * L_CNT_STRIDE; }   // one line per row tile of the smallest tile height
size_t l_sync_words(int B) { return (l_cnt_words(B) + 16 + 63) / 64 * 64; }

template <int NP, int NG>
long l_resident(size_t lds) {   // workgroups the current device holds at once, by the occupancy API's count
    const void* k = reinterpret_cast<const void*>(lstm_seq_fwd_persist_kernel<NP, NG>);
    if (cpg_allow_big_lds(k, 160 * 1024) != 0) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_fwd_persist_kernel<NP, NG>, L_WAVES * 64, lds) != hipSuccess) return 0;
    return (long)per_cu * cpg_device_cus();
}

// Groups of 8 hidden units per workgroup for a [B, H] sequence in the current compute mode: the largest of 4 / 2 / 1 whose plane slice
// fits the LDS, divides H, and whose workgroups (row groups of 8 waves x 64 / NG rows, times H / (8 NG) column tiles) are all
// co-resident; 0 = not covered.  More groups = fewer, wider column tiles: less state re-read per workgroup, fewer producers per tile.
int l_pick_ng(int B, int H) {
    const int np = cpg_persist_planes();
    const CpgOptVal cap = cpg_opt(OPT_LSTM_PERSIST_NG);
    const int top = cap.set && (cap.i == 1 || cap.i == 2) ? (int)cap.i : 4;
    // three planes: one group (wider forms do not fit 256 registers without spills); the f16 pair: up to two
    for (int ng = np == 1 ? top : np == 2 ? min(top, 2) : 1; ng >= 1; ng >>= 1) {
        if (H % (8 * ng) != 0) continue;
        const size_t lds = l_lds_bytes(H, np, ng);
        if (lds > 160 * 1024) continue;
        const long wgs = (long)cdiv(cdiv(B, 64 / ng), L_WAVES) * (H / (8 * ng));
        long fit = 0;
        if (np == 1) fit = ng == 4 ? l_resident<1, 4>(lds) : ng == 2 ? l_resident<1, 2>(lds) : l_resident<1, 1>(lds);
        else if (np == 2) fit = ng == 2 ? l_resident<2, 2>(lds) : l_resident<2, 1>(lds);
        else fit = l_resident<3, 1>(lds);
        if (wgs <= fit) return ng;
    }
    return 0;
}

}  // namespace

// 1 when the persistent LSTM forward kernel covers [B rows, H hidden] on this device in the current compute mode (every workgroup
// co-resident).  Option lstm_persist = 0 disables the path (per-step launches).
CPG_EXPORT int cpg_lstm_persistent_fits(int B, int H) {
    const CpgOptVal& o = cpg_opt(OPT_LSTM_PERSIST);
    if (o.set && o.i == 0) return 0;
    if (B <= 0 || H < 32 || H % 32 != 0) return 0;
    if ((size_t)(B + 1) * H * 6 * 64 > (size_t)3 << 30) return 0;   // exchange slots are addressed through one 32-bit buffer range
    return l_pick_ng(B, H) > 0;
}

// Name of the kernel a persistent launch runs, as rocprofv3 prints it (bench.py's roofline object)
CPG_EXPORT int cpg_lstm_persistent_kernel_name(int B, int H, char* buf, int n) {
    return snprintf(buf, n, "lstm_seq_fwd_persist_kernel<%d, %d>", cpg_persist_planes(), l_pick_ng(B, H));
}

CPG_EXPORT size_t cpg_lstm_persistent_scratch_bytes(int T, int B, int H) {
    return l_sync_words(B) * sizeof(unsigned) + (size_t)(T + 1) * 3 * ((B + 1) & ~1) * H * sizeof(uint16_t);
}

// Whole forward sequence in one launch; arguments as cpg_lstm_seq_fwd.  sync_scratch: cpg_lstm_persistent_scratch_bytes(T,B,H)
// bytes of device memory, zeroed by the caller when allocated (every launch leaves the counters at zero again; the error
// word is sticky, cpg_lstm_persistent_status reads it).
CPG_EXPORT int cpg_lstm_seq_fwd_persistent(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                                           const int32_t* tok, const float* tab, const float* rowc, const float* dense,
                                           float* hs, float* cs, float* gates, void* sync_scratch, void* err_host, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && b_hh && hs && cs && sync_scratch);
    CPG_CHECK_ARG((tok == nullptr) == (tab == nullptr));
    const int ng = cpg_lstm_persistent_fits(B, H) ? l_pick_ng(B, H) : 0;
    if (ng == 0) {
        cpg_set_error("cpg_lstm_seq_fwd_persistent: B=%d H=%d does not fit the persistent kernel on this device", B, H);
        return -5;
    }
    hipStream_t s = (hipStream_t)stream;
    // (no memset: the counters are zero at allocation and every launch leaves them at zero)
    LFwdArgs a;
    a.w_hh = w_hh; a.b_hh = b_hh; a.tok = tok; a.tab = tab; a.rowc = rowc; a.dense = dense; a.hs = hs; a.cs = cs; a.gates = gates;
    a.cnt = (unsigned*)sync_scratch;
    a.err = a.cnt + l_cnt_words(B);
    a.err_host = (unsigned*)err_host;
    a.xch = (uint16_t*)(a.cnt + l_sync_words(B));
    a.T = T; a.B = B; a.H = H; a.reverse = reverse;
    a.groups = cdiv(cdiv(B, 64 / ng), L_WAVES);
    a.S = l_plane_stride_words(H);
    const int np = cpg_persist_planes();
    const size_t lds = l_lds_bytes(H, np, ng);
    const dim3 grid(a.groups * (H / (8 * ng))), block(L_WAVES * 64);
#define CPG_LP(NP_, NG_) hipLaunchKernelGGL((lstm_seq_fwd_persist_kernel<NP_, NG_>), grid, block, lds, s, a)
    if (np == 1) { if (ng == 4) CPG_LP(1, 4); else if (ng == 2) CPG_LP(1, 2); else CPG_LP(1, 1); }
    else if (np == 2) { if (ng == 2) CPG_LP(2, 2); else CPG_LP(2, 1); }
    else CPG_LP(3, 1);
#undef CPG_LP
    CPG_LAUNCH_CHECK();
    return 0;
}

CPG_EXPORT size_t cpg_lstm_persistent_err_offset(int B) { return l_cnt_words(B) * sizeof(unsigned); }

CPG_EXPORT int cpg_lstm_persistent_status(int B, const void* sync_scratch, void* stream) {
    unsigned v = 0;
    const unsigned* p = (const unsigned*)sync_scratch + l_cnt_words(B);
    if (hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}
