// Persistent GRU sequence kernels for gfx950: the WHOLE time loop of one direction of one layer in ONE launch.
//
// Replaces the per-step launches of csrc/gru.hip (torch.nn.GRU at models/encoder.py:25-30,42; models/decoder.py:40-41,77)
// when the problem fits the chip (cpg_gru_persistent_fits).  Why: a step of the recurrence at B=2048, H=512 is a
// [2048,512] x [512,1536] product - 8 us of matrix-pipe time as six bf16 MFMAs per block on 3-way split operands - but a
// launch per step spends 38 us on it: every launch re-stages and re-converts W_hh through LDS in every row tile, pays the
// launch ramp, two dependent gather latencies before its epilogue and a store tail after it.
//
// Decomposition (rows are independent recurrences; columns need the whole previous state):
//   * a workgroup owns CT hidden units (the r, z, n rows of W_hh for them: 3 CT x H) for a group of 256 batch rows and
//     keeps that W_hh slice in LDS for the whole sequence, ALREADY split into three bf16 planes.  CT = 16 (three 16-column
//     MFMA blocks, one per gate: 48 x H x 6 B = 147 KB at H = 512, the reason for one workgroup per CU) up to H = 512;
//     CT = 8 for 512 < H <= 1024 (24 x H x 6 B = 147 KB at H = 1024; BASELINE.json configs[4] width): TWO column blocks,
//     [r | z] and [n | n again], the half-rows of a 16-lane group swap r / z with one DPP rotate (row_ror:8) and the lower
//     half runs the cell - the matrix pipe then works at 3/4 efficiency (24 useful of 32 columns);
//   * each of its 8 waves (two per SIMD: one wave's cell arithmetic and waits run under the other's MFMAs) owns 32 of the
//     rows.  The state operand goes global -> registers -> MFMA A fragments directly: every element is used by exactly one
//     wave of the workgroup, so nothing is shared through LDS and the time loop has NO workgroup barrier; the per-row
//     constant input term and the previous state of the wave's own 32 x 16 outputs stay in registers across steps;
//   * the 32 column-tile workgroups of a row tile hand h_t to each other ALREADY SPLIT: the producer converts its own
//     32 x 16 outputs once and writes three bf16 planes into a two-slot exchange buffer (the first form of this kernel let
//     every consumer convert the full state tile itself: 2816 VALU instructions per wave and step, the kernel was VALU-bound
//     at 31.6 us per step - PMC in profiles/).  Hand-off = the placement-independent recipe of the CDNA guide (Guideline 16,
//     R1): write-through (sc1) 16-byte stores, vmcnt(0), one relaxed agent-scope atomic add on the row tile's arrival
//     counter; consumers poll that counter relaxed, then read with sc1 loads (L1 bypassed: no acquire fence needed).
//     Slot reuse is safe: a wave overwrites slot (p+1)&1 only after all 32 producers of its row tile have ARRIVED for step
//     p-1, i.e. finished reading it.  Counters are zero at allocation and every launch leaves them at zero; every spin is bounded.
//   * the f32 state slab and the saved gates are plain / non-temporal stores issued behind the arrival.
// Arithmetic is the per-step kernel's (same split, same MFMA order, same cell formulas): results are f32-grade and the
// golden / oracle parity tests run unchanged on this path.
#include "gemm_core.h"
#include "cpg_internal.h"
#include <stdlib.h>
#include <map>
#include <mutex>

#ifndef CPG_PERSIST_ACQUIRE
#define CPG_PERSIST_ACQUIRE 0
#endif
#ifndef CPG_PERSIST_PLAIN_STORES
#define CPG_PERSIST_PLAIN_STORES 0
#endif
#ifndef CPG_PERSIST_ROTATE
#define CPG_PERSIST_ROTATE 0
#endif
#ifndef CPG_PERSIST_PLAIN_LOADS
#define CPG_PERSIST_PLAIN_LOADS 0   // 1: exchange slots are never reused inside a launch and are read with plain (L1/L2-allocating) loads;
                                    // 0: two-slot ring read with sc1 loads (always served behind the L2).  With the arrival counters on
                                    // their own lines the sc1 form is the faster one: 19.7 vs 21.9 us per step (f32-grade, B=2048, H=512)
#endif
// Diagnostic builds only (results wrong by construction): 1 no waits, 2 A operand loaded once per step, 4 no MFMAs,
// 8 no cell transcendental math, 16 no gate stores, 32 no publish drain (vmcnt) before the arrival
#ifndef CPG_PERSIST_ABLATE
#define CPG_PERSIST_ABLATE 0
#endif
#ifndef CPG_PERSIST_XCD_FAST
#define CPG_PERSIST_XCD_FAST 1   // 0: always the placement-independent write-through hand-off
#endif

#ifndef CPG_PERSIST_FAST_CELL
#define CPG_PERSIST_FAST_CELL 1
#endif
// Diagnostic builds only (-DCPG_DIAG -DCPG_PERSIST_TRACE=1, tools/persist_trace.py): every wave writes 8 time stamps (10-ns
// ticks) per step into a trace area behind the exchange slots - where a step's time goes, wave by wave.
#ifndef CPG_PERSIST_TRACE
#define CPG_PERSIST_TRACE 0
#endif
// Row tiles per wave.  1: a wave owns ONE 32-row tile (one arrival counter, one chain of steps).  2: the wave's 32 rows are TWO
// 16-row tiles with their own arrival counters, processed one after the other inside every step: while the wave works on tile B,
// the 32 producers of tile A finish and arrive - the wait for the slowest of them (6 of 19.6 us per step in the one-tile form,
// tools/persist_trace.py) runs under useful work instead of idling the wave.  Same registers (the tiles share them), twice the
// B-fragment LDS reads and counter traffic.
#ifndef CPG_PERSIST_SUBTILES
#define CPG_PERSIST_SUBTILES 1
#endif

#if CPG_PERSIST_TRACE && !defined(CPG_DIAG)
#error "CPG_PERSIST_TRACE needs -DCPG_DIAG"
#endif
#if CPG_PERSIST_TRACE
#define P_STAMP(i)                                                                                                       \
    do {                                                                                                                 \
        const unsigned long long t__ = __builtin_amdgcn_s_memrealtime();                                                 \
        if (lane == 0) a.trace[(((size_t)blockIdx.x * P_WAVES + wave) * T + p) * 8 + (i)] = t__;                         \
    } while (0)
#else
#define P_STAMP(i) do {} while (0)
#endif

namespace {

// cell nonlinearities: 0 the library forms of the per-step kernels, 1 (default) the hardware exp2 / rcp forms of cpg_common.h
__device__ __forceinline__ float p_sigmoid(float x) {
#if CPG_PERSIST_FAST_CELL
    return cell_sigmoidf(x);
#else
    return sigmoidf_(x);
#endif
}
__device__ __forceinline__ float p_tanh(float x) {
#if CPG_PERSIST_FAST_CELL
    return cell_tanhf(x);
#else
    return tanhf(x);
#endif
}

#ifndef CPG_PERSIST_WAVES
#define CPG_PERSIST_WAVES 8       // two waves per SIMD: one wave's cell arithmetic, stores and waits run under the other's MFMAs
                                  // (21.8 us per step against 26.8 with 4 - once the arrival counters sit on separate lines)
#endif
#ifndef CPG_PERSIST_DEFER
#define CPG_PERSIST_DEFER 0       // 1: the f32 state / gate stores of step p are issued after the wait of step p+1
#endif
#ifndef CPG_PERSIST_DEPTH
#define CPG_PERSIST_DEPTH 3       // k-blocks of the state operand in flight per wave (8 waves: f32-grade 21.8 / 21.7 / 22.0 us per step at
#endif                            // depth 2 / 3 / 4; bf16 mode 10.5 / 10.0 / 10.1)
#ifndef CPG_PERSIST_WAVES_BF16
#define CPG_PERSIST_WAVES_BF16 16 // bf16 compute mode (one plane: half the exchange reads, 96 VGPRs): sixteen waves of 16 rows - four chains per
                                  // SIMD - 227 -> 215 us per sequence; the f32-grade form is bound by its exchange reads and gains nothing (EXPERIMENTS R6.8)
#endif
constexpr int P_DEPTH = CPG_PERSIST_DEPTH;
constexpr int waves_of(int np) { return np == 1 ? CPG_PERSIST_WAVES_BF16 : CPG_PERSIST_WAVES; }
constexpr int P_WAVES = CPG_PERSIST_WAVES;   // (the kernel shadows these four with the values of its own plane count)
constexpr int P_WROWS = 256 / P_WAVES;   // rows per wave
constexpr int P_MI = P_WROWS / 16;
constexpr int P_SUB = CPG_PERSIST_SUBTILES;          // row tiles per wave
constexpr int P_MIS = P_MI / P_SUB;                  // 16-row blocks per row tile
static_assert(P_MI % P_SUB == 0 && P_MIS >= 1, "row tiles per wave must divide the wave's row blocks");
static_assert(P_SUB == 1 || !CPG_PERSIST_DEFER, "deferred gate stores are a one-tile-per-wave form");
#ifndef CPG_PERSIST_TBW
#define CPG_PERSIST_TBW 16
#endif
constexpr int P_TBW = CPG_PERSIST_TBW;  // words per row of the per-wave 16x16 transposition buffer
// Phase offset between the two waves of a SIMD (waves w and w + P_WAVES/2 own different row tiles = independent chains): the
// second set starts this many 10-ns ticks late, so that one wave's product phase (exchange loads, MFMAs) runs against the
// other's epilogue phase (gate / state traffic, cell arithmetic) instead of both doing the same thing at the same time.
#ifndef CPG_PERSIST_PHASE_FWD
#define CPG_PERSIST_PHASE_FWD 0
#endif
#ifndef CPG_PERSIST_CNT_STRIDE
#define CPG_PERSIST_CNT_STRIDE 64  // words between arrival counters: one 256-byte line each, so the adds and polls of different
                                   // row tiles do not queue on one memory channel
#endif
constexpr int P_CNT_STRIDE = CPG_PERSIST_CNT_STRIDE;
constexpr size_t P_XCC_WORDS = 1024;   // XCD table: >= row groups x column tiles of one launch (8 x 128); its last word records the path taken
constexpr unsigned P_SPIN_LIMIT = 400000u;  // ~0.2 s of polling before a wave gives up (sets the error word)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PFwdArgs {
    const float* w_hh;     // [3H,H]
    const float* b_hh;     // [3H]
    const int32_t* tok;    // [T,B] or null
    const float* tab;      // [V,3H] or null
    const float* rowc;     // [B,3H] or null
    const float* dense;    // [T,B,3H] or null
    float* hs;             // [(T+1),B,H]
    float* gates;          // [T,4,B,H] or null; bf16 elements when gates_bf16 (NP = 1 only)
    int gates_bf16;
    unsigned* cnt;         // [row tiles] arrival counters (zero at launch; the launch leaves them at zero)
    unsigned* err;         // sticky error word
    unsigned* err_host;    // the same word in host-mapped (pinned) memory, or null: the host sees a timeout without any copy
    uint16_t* xch;         // [2 slots][3 planes][H/32 k-blocks][B][32] bf16: the state as the consumers want it - a wave's
                           // A-fragment load (16 rows x 32 k of one plane) is ONE contiguous KB.  (With rows H apart the 32
                           // CUs of an XCD that read the same tile at the same time camped on a few L2 channels: 62.8 us/step.)
    int T, B, H, reverse, groups, S;  // S: words per plane row (H/2 data + pad so that S % 64 == 8)
    int row0, row1;                   // rows [row0, row1) of the B-row problem are this launch's
    unsigned* xcc;                    // [groups][column tiles]: XCC id + 1 of every workgroup of this launch (same-XCD check)
#if CPG_PERSIST_TRACE
    unsigned long long* trace;        // [workgroups][waves][T][8]
#endif
};

__device__ __forceinline__ void phase_delay(int wave, int waves, unsigned ticks) {
    if (ticks == 0 || wave < waves / 2) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Wait until *p >= target (relaxed agent-scope polls: every lane reads the same word).  Bounded: on timeout the sticky error
// word is set and the wave is `dead`: it stops waiting, and everything it stores from then on is NaN (state slab, saved
// gates, exchange planes), so the failure reaches the loss / the decoded ids instead of passing as plausible numbers.
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* err, unsigned* err_host, bool& dead) {
    if (dead) return false;
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P_SPIN_LIMIT) {
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

// 16x16 tile held in the MFMA accumulator layout (lane (u = l&15, rq = l>>4), reg -> row 4rq+reg, col u) -> one f32x4 per
// lane in row layout (lane -> row l>>2, cols 4(l&3)..+3), through the wave's own LDS buffer (no other wave touches it).
__device__ __forceinline__ f32x4 acc_to_rows(float* tb, const float (&v)[4], int lane) {
    const int u = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[(4 * rq + r) * P_TBW + u] = v[r];
    return *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * P_TBW + 4 * (lane & 3));
}

// value of the neighbouring lane (l ^ 1)
__device__ __forceinline__ float lane_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

// value of lane (l + 8) % 16 of the same 16-lane row: DPP row_ror:8
__device__ __forceinline__ float row_ror8(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
}

// Row-layout tile (lane -> row l>>2, 4 columns) -> three bf16 planes in the exchange slot: even lanes collect their odd
// neighbour's four columns and store 8 columns = 16 bytes per plane, write-through.
template <int NP, bool PLAIN = false>
__device__ __forceinline__ void publish_rows(const f32x4 v, const __amdgpu_buffer_rsrc_t rx, int voff, unsigned plane_bytes,
                                             unsigned slot_off, bool ok, int lane) {
    const float n0 = lane_xor1(v[0]), n1 = lane_xor1(v[1]), n2 = lane_xor1(v[2]), n3 = lane_xor1(v[3]);
    if ((lane & 1) == 0 && ok) {   // ok also carries the column bound of narrow (CT = 8) tiles
        uint32_t w0[4], w1[4], w2[4];
        if (NP == 2) {   // f16 pair
            split2h_pair(v[0], v[1], w0[0], w1[0]);
            split2h_pair(v[2], v[3], w0[1], w1[1]);
            split2h_pair(n0, n1, w0[2], w1[2]);
            split2h_pair(n2, n3, w0[3], w1[3]);
        } else if (NP == 3) {
            split3_pair(v[0], v[1], w0[0], w1[0], w2[0]);
            split3_pair(v[2], v[3], w0[1], w1[1], w2[1]);
            split3_pair(n0, n1, w0[2], w1[2], w2[2]);
            split3_pair(n2, n3, w0[3], w1[3], w2[3]);
        } else {  // bf16 compute mode: one plane, operands rounded to nearest even
            w0[0] = cvt_pk_bf16(v[0], v[1]); w0[1] = cvt_pk_bf16(v[2], v[3]);
            w0[2] = cvt_pk_bf16(n0, n1); w0[3] = cvt_pk_bf16(n2, n3);
        }
        constexpr int AUX = (CPG_PERSIST_PLAIN_STORES || PLAIN) ? 0 : 16;   // 16 = sc1: write-through; 0: the line stays in this XCD's L2
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w0[1], w0[2], w0[3]}, rx, voff, slot_off, AUX);
        if (NP >= 2) __builtin_amdgcn_raw_buffer_store_b128(u32x4{w1[0], w1[1], w1[2], w1[3]}, rx, voff, slot_off + plane_bytes, AUX);
        if (NP == 3) __builtin_amdgcn_raw_buffer_store_b128(u32x4{w2[0], w2[1], w2[2], w2[3]}, rx, voff, slot_off + 2 * plane_bytes, AUX);
    }
}

// NP: planes of the split - 2 = f32-grade on an f16 pair (three MFMAs per block, gemm_core.h; the default), 3 = f32-grade on three
//     bf16 planes (six MFMAs per block; option f32_engine = bf16x3), 1 = bf16 compute mode (cpg_set_compute_mode(1))
// CT: hidden units per workgroup - 16 (one MFMA column block per gate) or 8 (blocks [r | z] and [n | n], see the header)
typedef uint32_t pk_u32x4 __attribute__((ext_vector_type(4)));

template <int NP, int CT>
__global__ __launch_bounds__(waves_of(NP) * 64, waves_of(NP) / 4) void gru_seq_fwd_persist_kernel(PFwdArgs a) {
    constexpr int P_WAVES = waves_of(NP), P_WROWS = 256 / P_WAVES, P_MI = P_WROWS / 16, P_MIS = P_MI / P_SUB;   // waves, rows per wave, ...
    static_assert(P_MI >= 1 && P_MI % P_SUB == 0, "a wave owns whole 16-row blocks");
    constexpr int NC = 3 * CT;              // gate columns (LDS plane rows) of the workgroup
    constexpr int NB = CT == 16 ? 3 : 2;    // MFMA column blocks
    static_assert(CT == 16 || CT == 8, "16 units: one block per gate; 8 units: [r|z] and [n|n]");
    extern __shared__ __attribute__((aligned(16))) uint32_t psm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x % a.groups, ct = blockIdx.x / a.groups;
    const int H = a.H, B = a.B, T = a.T, S = a.S;
    const int Bend = a.row1;                // this launch covers rows [row_begin, row_end) of the B-row problem
    const int j0 = ct * CT;
    const int NCT = H / CT, KB = H / 32;
    const int PLW = NC * S;
    uint32_t* const planes = psm;
    float* const tb = reinterpret_cast<float*>(psm + NP * PLW) + wave * (16 * P_TBW);

    // ---- W_hh slice -> NP planes in LDS, once per sequence: plane[c = gate*CT + u][k pair]
    // f16 pair: the slice is multiplied by a power of two first so that its largest magnitude lands in [2^13, 2^14) - the low
    // plane of a weight of typical size (0.04) would otherwise be an f16 subnormal (absolute 2^-25: 7e-7 of the weight); the
    // accumulators are multiplied by the inverse (both exact)
    float wscale = 1.f, descale = 1.f;
    if constexpr (NP == 2) {
        float m = 0.f;
        for (int idx = tid; idx < NC * (H / 2); idx += P_WAVES * 64) {
            const int c = idx / (H / 2), kp = idx - c * (H / 2);
            const float2 v = *reinterpret_cast<const float2*>(a.w_hh + ((size_t)((c / CT) * H + j0 + (c % CT))) * H + 2 * kp);
            m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* const red = reinterpret_cast<float*>(psm);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < P_WAVES; ++w) m = fmaxf(m, red[w]);
        __syncthreads();
        int e = 0;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);   // m = f 2^e, f in [0.5, 1)
        e = max(-100, min(100, 14 - e));
        wscale = ldexpf(1.f, e);
        descale = ldexpf(1.f, -e);
    }
    for (int idx = tid; idx < NC * (H / 2); idx += P_WAVES * 64) {
        const int c = idx / (H / 2), kp = idx - c * (H / 2);
        const float2 v = *reinterpret_cast<const float2*>(a.w_hh + ((size_t)((c / CT) * H + j0 + (c % CT))) * H + 2 * kp);
        uint32_t w0, w1 = 0, w2 = 0;
        if (NP == 3) split3_pair(v.x, v.y, w0, w1, w2);
        else if (NP == 2) split2h_pair(v.x * wscale, v.y * wscale, w0, w1);
        else w0 = cvt_pk_bf16(v.x, v.y);
        planes[c * S + kp] = w0;
        if (NP >= 2) planes[PLW + c * S + kp] = w1;
        if (NP == 3) planes[2 * PLW + c * S + kp] = w2;
    }
    __syncthreads();

    const int rt = g * P_WAVES + wave;   // row tile of this wave (counted from row_begin)
    const int row0 = a.row0 + rt * P_WROWS;
    if (row0 >= Bend) return;            // wave-uniform; nobody waits for a tile that does not exist
    phase_delay(wave, P_WAVES, CPG_PERSIST_PHASE_FWD);
    const int l15 = lane & 15, lq = lane >> 4;
    const int srow = lane >> 2, scq = lane & 3;  // row-layout coordinates after acc_to_rows
    const int hcol = j0 + (l15 & (CT - 1));      // hidden unit of this lane's accumulator elements
    const bool cvalid = 4 * scq < CT;            // row layout: this lane's four columns belong to the tile
    // column of the 3H gate axis that block b's accumulator of this lane belongs to
    int gcol[NB];
    if constexpr (CT == 16) {
#pragma unroll
        for (int b = 0; b < NB; ++b) gcol[b] = b * H + hcol;
    } else {
        gcol[0] = (l15 >> 3) * H + hcol;   // lower half-row: r of unit l15, upper half-row: z of unit l15 - 8
        gcol[1] = 2 * H + hcol;            // n (both half-rows multiply the same eight W_hn rows; the upper copy is unused)
    }

    // exchange rows are padded to an even count: a k-block of a plane then starts on a 128-byte line, so no line holds rows of two
    // row tiles (two arrival counters) - a reader of one tile would otherwise cache the other's not yet written chunk
    const unsigned Bx = ((unsigned)B + 1u) & ~1u;
    const unsigned plane_bytes = (unsigned)((size_t)Bx * H * 2), kb_bytes = Bx * 64u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xch, (unsigned)(CPG_PERSIST_PLAIN_LOADS ? T + 1 : 2) * 3u * plane_bytes);
    bool dead = false;

    // per-lane constants of the epilogue: row of accumulator element (mi, reg), clamped for the loads
    float rc[P_MI][4][NB], hprev[P_MI][4];
    const size_t slot0 = (size_t)(a.reverse ? T : 0) * B * H;
    float bh[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bh[b] = a.b_hh[gcol[b]];
#pragma unroll
    for (int mi = 0; mi < P_MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + 16 * mi + 4 * lq + r, Bend - 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) rc[mi][r][b] = a.rowc ? a.rowc[(size_t)row * 3 * H + gcol[b]] : 0.f;
            hprev[mi][r] = a.hs[slot0 + (size_t)row * H + hcol];
        }
        // h0 enters the exchange like any step's output: slot 0, arrival #1
        const int row = row0 + 16 * mi + srow;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.hs + slot0 + (size_t)min(row, Bend - 1) * H + j0 + (cvalid ? 4 * scq : 0));
        publish_rows<NP>(v, rx, row * 64 + ((j0 & 31) + 4 * (scq & ~1)) * 2, plane_bytes, (unsigned)(j0 >> 5) * kb_bytes,
                         row < Bend && 4 * (scq & ~1) < CT, lane);
    }
    // Same-XCD fast path (see the time loop): every workgroup posts the id of the XCD it runs on before its first arrival
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) + 1u;   // HW_REG_XCC_ID[3:0] + 1
    if (lane == 0) __hip_atomic_store(a.xcc + (size_t)g * NCT + ct, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // row tile s of this wave: rows [row0 + 16 s P_MIS, ...); a tile that starts past the end does not exist (nobody waits for it)
    unsigned* const cnt0 = a.cnt + (size_t)rt * P_SUB * P_CNT_STRIDE;
    if (lane == 0) {
#pragma unroll
        for (int sb = 0; sb < P_SUB; ++sb)
            if (row0 + 16 * sb * P_MIS < Bend) __hip_atomic_fetch_add(cnt0 + sb * P_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // A-operand addressing: lane (l15, lq) of row block mi reads 8 consecutive k of row row0 + 16 mi + l15 (16 bytes of a plane)
    int aoff[P_MI];
#pragma unroll
    for (int mi = 0; mi < P_MI; ++mi) aoff[mi] = min(row0 + 16 * mi + l15, Bend - 1) * 64 + 16 * lq;
    // B fragments: block b, lane l15 -> plane row; CT = 8: block 1 wraps (both half-rows read the eight n rows)
    const uint32_t* bbase[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bbase[b] = planes + (CT == 16 ? 16 * b + l15 : (b == 0 ? l15 : 16 + (l15 & 7))) * S + 4 * lq;

    // saved-for-backward gates and the f32 state of the last finished step (not part of the hand-off)
    float rg[P_MI][4], zg[P_MI][4], ng[P_MI][4], hn[P_MI][4];
    f32x4 hrow[P_MI];
    int pend_tt = -1;
    auto flush = [&](int mi0, int mi1) {
        if (pend_tt < 0) return;
        float* const hout = a.hs + (size_t)(a.reverse ? pend_tt : pend_tt + 1) * B * H;
#pragma unroll
        for (int mi = mi0; mi < mi1; ++mi) {
            const int row = row0 + 16 * mi + srow;
            if (row < Bend && cvalid) *reinterpret_cast<f32x4*>(hout + (size_t)row * H + j0 + 4 * scq) = hrow[mi];
        }
        if (a.gates && !(CPG_PERSIST_ABLATE & 16)) {
            const size_t BH = (size_t)B * H;
            float* const gbase = a.gates + (size_t)pend_tt * 4 * BH;
#pragma unroll
            for (int mi = mi0; mi < mi1; ++mi) {
                const int row = row0 + 16 * mi + srow;
                const f32x4 v0 = acc_to_rows(tb, rg[mi], lane);
                const f32x4 v1 = acc_to_rows(tb, zg[mi], lane);
                const f32x4 v2 = acc_to_rows(tb, ng[mi], lane);
                const f32x4 v3 = acc_to_rows(tb, hn[mi], lane);
                if constexpr (NP == 1) {
                    if (a.gates_bf16) {   // bf16 compute mode: [B,H][4] bf16, (r,z,n,hn) of an element adjacent - 32 contiguous bytes per lane
                        if (row < Bend && cvalid) {
                            pk_u32x4* d = reinterpret_cast<pk_u32x4*>(reinterpret_cast<uint16_t*>(a.gates) +
                                                                      4 * ((size_t)pend_tt * BH + (size_t)row * H + j0 + 4 * scq));
                            __builtin_nontemporal_store(pk_u32x4{cvt_pk_bf16(v0[0], v1[0]), cvt_pk_bf16(v2[0], v3[0]),
                                                                 cvt_pk_bf16(v0[1], v1[1]), cvt_pk_bf16(v2[1], v3[1])}, d);
                            __builtin_nontemporal_store(pk_u32x4{cvt_pk_bf16(v0[2], v1[2]), cvt_pk_bf16(v2[2], v3[2]),
                                                                 cvt_pk_bf16(v0[3], v1[3]), cvt_pk_bf16(v2[3], v3[3])}, d + 1);
                        }
                        continue;
                    }
                }
                if (row < Bend && cvalid) {
                    float* d = gbase + (size_t)row * H + j0 + 4 * scq;
                    __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(d));
                    __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(d + BH));
                    __builtin_nontemporal_store(v2, reinterpret_cast<f32x4*>(d + 2 * BH));
                    __builtin_nontemporal_store(v3, reinterpret_cast<f32x4*>(d + 3 * BH));
                }
            }
        }
        pend_tt = -1;
    };

    bool fast = false;   // wave-uniform: set after the first wait when the row tile's producers share this XCD
    for (int p = 0; p < T; ++p) {
        const int tt = a.reverse ? T - 1 - p : p;
        // Exchange slot of step p's input / output: a two-slot ring read with sc1 loads (CPG_PERSIST_PLAIN_LOADS = 1: one slot
        // PER STEP, written once, write-through, first read only after its row tile's arrival counter says every producer is
        // done - no cache can hold a stale copy - and read with plain loads that allocate in L2).
        const unsigned in_off = (unsigned)(CPG_PERSIST_PLAIN_LOADS ? p : (p & 1)) * 3u * plane_bytes;
        const unsigned out_off = (unsigned)(CPG_PERSIST_PLAIN_LOADS ? p + 1 : ((p + 1) & 1)) * 3u * plane_bytes;

#pragma unroll
        for (int sb = 0; sb < P_SUB; ++sb) {
        constexpr int dummy_ = 0; (void)dummy_;
        const int MI0 = sb * P_MIS, MI1 = MI0 + P_MIS;
        if (row0 + 16 * MI0 >= Bend) continue;      // wave-uniform: this row tile does not exist
        if (sb == 0) P_STAMP(0);
        // input-side pre-activations of this step: independent of the recurrence, fetched before the wait.  (Hoisting the
        // source tests out of the element loops - one clause of 8 token loads, then one of 8 x NB gathers - was built and measured:
        // faster in a synthetic micro-benchmark, 75-100 us per sequence SLOWER inside the training step; not kept.)
        float gi[P_MI][4][NB];   // (only rows MI0..MI1 of the arrays below are live in this phase)
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < NB; ++b) gi[mi][r][b] = rc[mi][r][b];
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(row0 + 16 * mi + 4 * lq + r, Bend - 1);
                if (a.tok) {
                    const float* t = a.tab + (size_t)a.tok[(size_t)tt * B + row] * 3 * H;
#pragma unroll
                    for (int b = 0; b < NB; ++b) gi[mi][r][b] += t[gcol[b]];
                }
                if (a.dense) {
                    const float* t = a.dense + ((size_t)tt * B + row) * 3 * H;
#pragma unroll
                    for (int b = 0; b < NB; ++b) gi[mi][r][b] += t[gcol[b]];
                }
            }

        if (!(CPG_PERSIST_ABLATE & 1)) wait_ge(cnt0 + sb * P_CNT_STRIDE, (unsigned)(NCT * (p + 1)), a.err, a.err_host, dead);
        if (CPG_PERSIST_XCD_FAST && NP >= 2 && p == 0 && sb == 0 && !dead) {
            // All NCT producers of this row tile have arrived once, so all of them have posted their XCD.  If every one of them
            // sits on THIS XCD they share one L2, and from here on the tile's planes are published with PLAIN stores: the lines stay
            // in that L2 (an sc1 store writes through and drops them, and the readers then fetch at the cross-XCD rate) and are
            // still read with sc1 loads, which bypass the CU's L1 only.  Measured per launch, NOT assumed from blockIdx: any other
            // placement keeps the write-through stores.  f32-grade (three planes) only: with the single plane of the bf16 mode the
            // plain stores are 9 % SLOWER (273 against 251 us per sequence).  L2-local (workgroup-scope) arrival adds were
            // measured too: no effect in either mode, not used.
            bool same = true;
            for (int c = lane; c < NCT; c += 64)
                same = same && __hip_atomic_load(a.xcc + (size_t)g * NCT + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_xcc;
            fast = __builtin_amdgcn_ballot_w64(!same) == 0;
            if (blockIdx.x == 0 && wave == 0 && lane == 0) a.xcc[P_XCC_WORDS - 1] = fast ? 1u : 2u;   // for tests / diagnostics
        }
        if (sb == 0) P_STAMP(1);
        if (CPG_PERSIST_DEFER) flush(MI0, MI1);
#if CPG_PERSIST_ACQUIRE
        // plain (L2-allocating) loads behind ONE agent-scope acquire
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif

        // ---- recurrent product: acc[mi][b] = h_prev[32 rows, H] . (block b's 16 rows of the W_hh slice)^T
        f32x4 acc[P_MI][NB];
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[mi][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 buf[P_DEPTH][P_MI][NP];  // register ring: the loads of k-block kb + P_DEPTH - 1 are in flight while kb is multiplied
        // CPG_PERSIST_ROTATE: each column tile starts at its own k-block and wraps around (spreads the 32 readers of a plane
        // over the L2 channels; sums over k-blocks in a rotated, per-column-tile deterministic order)
        const int rot = CPG_PERSIST_ROTATE ? (ct * KB) / NCT : 0;
        auto load = [&](u32x4 (&bf)[P_MI][NP], int kbi) {
            int kb = kbi + rot;
            kb = kb >= KB ? kb - KB : kb;
#pragma unroll
            for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    bf[mi][pl] = __builtin_amdgcn_raw_buffer_load_b128(rx, aoff[mi], in_off + pl * plane_bytes + kb * kb_bytes,
                                                                       (CPG_PERSIST_ACQUIRE || CPG_PERSIST_PLAIN_LOADS) ? 0 : 16);
        };
        auto compute = [&](const u32x4 (&bf)[P_MI][NP], int kbi) {
            int kb = kbi + rot;
            kb = kb >= KB ? kb - KB : kb;
            cpg_bf16x8 fb[NB][NP];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    fb[b][pl] = *reinterpret_cast<const cpg_bf16x8*>(bbase[b] + pl * PLW + kb * 16);
            if (CPG_PERSIST_ABLATE & 4) {
#pragma unroll
                for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[mi][b] += __builtin_bit_cast(f32x4, bf[mi][0]) * __builtin_bit_cast(f32x4, fb[b][0]);
                return;
            }
            // six products per block in the per-step kernel's order, walked TERM BY TERM over the NB P_MI independent
            // accumulators: a dependent MFMA waits for its predecessor to leave the pipe, independent ones issue back to back
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
            if constexpr (NP == 2) {   // f16 pair: low x high, high x low, high x high
                constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[mi][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cpg_f16x8, bf[mi][HA[t]]),
                                                                                __builtin_bit_cast(cpg_f16x8, fb[b][HB[t]]), acc[mi][b], 0, 0, 0);
                return;
            }
#pragma unroll
            for (int t = (NP == 3 ? 0 : 5); t < 6; ++t)
#pragma unroll
                for (int mi = MI0; mi < MI1; ++mi)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[mi][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cpg_bf16x8, bf[mi][NP == 3 ? TA[t] : 0]),
                                                                             fb[b][NP == 3 ? TB[t] : 0], acc[mi][b], 0, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < P_DEPTH - 1; ++d)
            if (d < KB) load(buf[d], d);
        for (int kb = 0; kb < KB; kb += P_DEPTH) {
#pragma unroll
            for (int d = 0; d < P_DEPTH; ++d) {
                if (kb + d < KB) {
                    if (kb + d + P_DEPTH - 1 < KB && !(CPG_PERSIST_ABLATE & 2)) load(buf[(d + P_DEPTH - 1) % P_DEPTH], kb + d + P_DEPTH - 1);
                    compute(buf[d], kb + d);
                    if (kb + d == 0) if (sb == 0) P_STAMP(2);
                }
            }
        }
        if (sb == 0) P_STAMP(3);

        // ---- cell (same formulas and association as gru_step_fwd_kernel)
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (NP == 2) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[mi][b][r] *= descale;
                }
                float pr, pz, pn;     // r / z pre-activations, and gi_n
                if constexpr (CT == 16) {
                    pr = gi[mi][r][0] + (acc[mi][0][r] + bh[0]);
                    pz = gi[mi][r][1] + (acc[mi][1][r] + bh[1]);
                    pn = gi[mi][r][2];
                    hn[mi][r] = acc[mi][2][r] + bh[2];
                } else {
                    pr = gi[mi][r][0] + (acc[mi][0][r] + bh[0]);   // lower half-row: r, upper half-row: z
                    pz = 0.f;
                    pn = gi[mi][r][1];
                    hn[mi][r] = acc[mi][1][r] + bh[1];
                }
                if (CPG_PERSIST_ABLATE & 8) {
                    rg[mi][r] = pr;
                    zg[mi][r] = CT == 16 ? pz : row_ror8(pr);
                    ng[mi][r] = pn + rg[mi][r] * hn[mi][r];
                } else {
                    rg[mi][r] = p_sigmoid(pr);
                    zg[mi][r] = CT == 16 ? p_sigmoid(pz) : row_ror8(rg[mi][r]);   // CT = 8: the upper half-row's sigmoid IS z
                    ng[mi][r] = p_tanh(pn + rg[mi][r] * hn[mi][r]);
                }
                hprev[mi][r] = (1.f - zg[mi][r]) * ng[mi][r] + zg[mi][r] * hprev[mi][r];
            }
            if (__builtin_amdgcn_readfirstlane((int)dead)) {   // a timed-out wait (wave-uniform): make the damage visible
#pragma unroll
                for (int r = 0; r < 4; ++r) hprev[mi][r] = rg[mi][r] = __builtin_nanf("");
            }
        }
        if (sb == 0) P_STAMP(4);
        // ---- publish h_t (split planes, write-through), drain, one arrival per wave; the f32 slab goes out behind it
#pragma unroll
        for (int mi = MI0; mi < MI1; ++mi) {
            hrow[mi] = acc_to_rows(tb, hprev[mi], lane);
            const int row = row0 + 16 * mi + srow;
            if (p + 1 < T) {
                if (fast)
                    publish_rows<NP, true>(hrow[mi], rx, row * 64 + ((j0 & 31) + 4 * (scq & ~1)) * 2, plane_bytes,
                                           out_off + (unsigned)(j0 >> 5) * kb_bytes, row < Bend && 4 * (scq & ~1) < CT, lane);
                else
                    publish_rows<NP, false>(hrow[mi], rx, row * 64 + ((j0 & 31) + 4 * (scq & ~1)) * 2, plane_bytes,
                                            out_off + (unsigned)(j0 >> 5) * kb_bytes, row < Bend && 4 * (scq & ~1) < CT, lane);
            }
        }
        if (!(CPG_PERSIST_ABLATE & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sb == 0) P_STAMP(5);
        if (lane == 0) __hip_atomic_fetch_add(cnt0 + sb * P_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_tt = tt;
        if (!CPG_PERSIST_DEFER) flush(MI0, MI1);
        if (sb == 0) P_STAMP(6);
        }   // row tiles of the wave
    }
    if (CPG_PERSIST_DEFER) flush(0, P_MI);
    // ---- the launch leaves its arrival counters at zero for the next one (round 6: a memset node in front of every launch cost
    // ~4 us + a ~6 us bubble).  A wave whose own adds are acknowledged (vmcnt) signs off on the word behind its tile's counter; the
    // wave that signs off last knows every consumer is past its final wait and every producer's adds are performed: it zeroes both.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
#pragma unroll
        for (int sb = 0; sb < P_SUB; ++sb) {
            if (row0 + 16 * sb * P_MIS >= Bend) continue;
            unsigned* const c = cnt0 + sb * P_CNT_STRIDE;
            if (__hip_atomic_fetch_add(c + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)NCT - 1u) {
                __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int plane_stride_words(int H) {
    int s = H / 2;
    while (s % 64 != 8) ++s;   // rows start 8 words apart modulo the 64 banks: conflict-free ds_read_b128 fragments
    return s;
}

size_t fwd_lds_bytes(int H, int ct, int np) { return ((size_t)np * 3 * ct * plane_stride_words(H) + waves_of(np) * 16 * P_TBW) * 4; }

// hidden units per workgroup for width H: 16 while the 48 x H plane slice fits the LDS, else 8 (24 x H), else 0 (not covered)
int pick_ct(int H, int np) {
    if (H < 32 || H % 32 != 0) return 0;
    if (fwd_lds_bytes(H, 16, np) <= 160 * 1024) return 16;
    if (fwd_lds_bytes(H, 8, np) <= 160 * 1024) return 8;
    return 0;
}

}  // namespace

// Workgroups of the persistent kernel the CURRENT device holds at once with `lds` bytes of dynamic LDS each, as the occupancy
// API reports it (registers, LDS, waves) - not an assumption about one workgroup per CU.
template <int NP, int CT>
static long resident_workgroups(size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, long> cache;   // (device, LDS bytes) -> workgroups
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, lds});
    if (it != cache.end()) return it->second;
    const void* k = reinterpret_cast<const void*>(gru_seq_fwd_persist_kernel<NP, CT>);
    if (cpg_allow_big_lds(k, 160 * 1024) != 0) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_seq_fwd_persist_kernel<NP, CT>, waves_of(NP) * 64, lds) != hipSuccess) return 0;
    return cache[{dev, lds}] = (long)per_cu * cpg_device_cus();
}

// Batch rows ONE persistent launch covers at width H on this device (0: width not covered): every workgroup (CT hidden
// units x 256 rows) co-resident by the occupancy API's count.  Wider batches run as consecutive launches over row ranges
// (rows are independent recurrences).  Option gru_persist = 0 disables the path (per-step launches).  The caller must own
// the device: a second process (ranks sharing a GPU) or a concurrent kernel holding CUs breaks co-residency - the host side
// switches the path off in that case (CPG_SHARED_DEVICE), and a wait that times out poisons the outputs with NaN.
CPG_EXPORT int cpg_gru_persistent_rows(int H) {
    const CpgOptVal& o = cpg_opt(OPT_GRU_PERSIST);
    if (o.set && o.i == 0) return 0;
    const int np = cpg_persist_planes();
    const int ct = pick_ct(H, np);
    if (ct == 0) return 0;
    const size_t lds = fwd_lds_bytes(H, ct, np);
    const long fit = np == 1   ? (ct == 16 ? resident_workgroups<1, 16>(lds) : resident_workgroups<1, 8>(lds))
                     : np == 2 ? (ct == 16 ? resident_workgroups<2, 16>(lds) : resident_workgroups<2, 8>(lds))
                               : (ct == 16 ? resident_workgroups<3, 16>(lds) : resident_workgroups<3, 8>(lds));
    const long groups = fit / (H / ct);          // row groups of 256 rows (8 waves x 32 rows, or 16 x 16)
    return (int)(groups * 256);
}

// 1 when ONE launch covers a [B rows, H hidden] sequence
CPG_EXPORT int cpg_gru_persistent_fits(int B, int H) {
    if (B <= 0) return 0;
    if ((size_t)(B + 1) * H * 6 * 64 > (size_t)3 << 30) return 0;   // exchange slots are addressed through one 32-bit buffer range
    return B <= cpg_gru_persistent_rows(H);
}

static size_t cnt_words(int B) { return (size_t)cdiv(B, 16) * P_SUB * P_CNT_STRIDE; }   // one line per row tile of the smallest tile height (16 rows)
static size_t sync_words(int B) { return (cnt_words(B) + 16 + P_XCC_WORDS + 63) / 64 * 64; }  // counters + error word + XCD table, 256-byte multiple

// Byte offset, inside the scratch, of a word the last launch set to 1 (its first row tile's producers shared one XCD: the hand-off
// stayed inside that L2) or 2 (write-through hand-off).  Diagnostics only.
CPG_EXPORT size_t cpg_gru_persistent_path_offset(int B) { return (cnt_words(B) + 16 + P_XCC_WORDS - 1) * sizeof(unsigned); }

CPG_EXPORT size_t cpg_gru_persistent_scratch_bytes(int T, int B, int H) {
    // counters + error word, then the exchange slots: three bf16 planes of the state, two slots (a slot per step + the initial
    // state with CPG_PERSIST_PLAIN_LOADS)
    size_t n = sync_words(B) * sizeof(unsigned) + (size_t)(CPG_PERSIST_PLAIN_LOADS ? T + 1 : 2) * 3 * ((B + 1) & ~1) * H * sizeof(uint16_t);
#if CPG_PERSIST_TRACE
    n = (n + 255) / 256 * 256 + (size_t)cdiv(B, 256) * (H / 8) * 16 * T * 8 * sizeof(unsigned long long);
#endif
    return n;
}

// Whole forward sequence of rows [row_begin, row_end) in one launch; other arguments as cpg_gru_seq_fwd.  sync_scratch:
// device memory of cpg_gru_persistent_scratch_bytes(T,B,H) bytes, ZEROED BY THE CALLER when allocated: arrival counters
// (every launch leaves them at zero again), a sticky error word (set by a wave whose wait timed out, never cleared
// here) and the exchange slots.
CPG_EXPORT int cpg_gru_seq_fwd_persistent(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                                          const int32_t* tok, const float* tab, const float* rowc, const float* dense,
                                          float* hs, float* gates, int row_begin, int row_end, void* sync_scratch, void* err_host, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && b_hh && hs && sync_scratch);
    CPG_CHECK_ARG((tok == nullptr) == (tab == nullptr));
    CPG_CHECK_ARG(0 <= row_begin && row_begin < row_end && row_end <= B);
    const int rows = row_end - row_begin;
    if (rows > cpg_gru_persistent_rows(H) || (size_t)(B + 1) * H * 6 * 64 > (size_t)3 << 30) {
        cpg_set_error("cpg_gru_seq_fwd_persistent: %d rows of B=%d at H=%d do not fit one persistent launch on this device", rows, B, H);
        return -5;
    }
    hipStream_t s = (hipStream_t)stream;
    // (no memset: the counters are zero when the scratch is allocated and every launch leaves them at zero - see the kernel's end)
    PFwdArgs a;
    a.w_hh = w_hh; a.b_hh = b_hh; a.tok = tok; a.tab = tab; a.rowc = rowc; a.dense = dense; a.hs = hs; a.gates = gates;
    a.gates_bf16 = gates && cpg_gru_store_bf16(B, H, true);
    a.cnt = (unsigned*)sync_scratch;
    a.err = a.cnt + cnt_words(B);
    a.xcc = a.err + 16;
    a.err_host = (unsigned*)err_host;
    a.xch = (uint16_t*)(a.cnt + sync_words(B));
    a.T = T; a.B = B; a.H = H; a.reverse = reverse;
    a.row0 = row_begin; a.row1 = row_end;
    a.S = plane_stride_words(H);
#if CPG_PERSIST_TRACE
    {
        const size_t base = sync_words(B) * sizeof(unsigned) + (size_t)(CPG_PERSIST_PLAIN_LOADS ? T + 1 : 2) * 3 * ((B + 1) & ~1) * H * sizeof(uint16_t);
        a.trace = (unsigned long long*)((char*)sync_scratch + (base + 255) / 256 * 256);
    }
#endif
    // (the > 64 KB dynamic-LDS opt-in happened in cpg_gru_persistent_rows -> resident_workgroups, per device)
    const int np = cpg_persist_planes();
    const int ct = pick_ct(H, np);
    a.groups = cdiv(rows, 256);          // row groups of 256 rows: waves_of(np) waves of 256 / waves_of(np) rows
    const dim3 grid(a.groups * (H / ct)), block(waves_of(np) * 64);
    const size_t lds = fwd_lds_bytes(H, ct, np);
    if (np == 1 && ct == 16) hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<1, 16>), grid, block, lds, s, a);
    else if (np == 1) hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<1, 8>), grid, block, lds, s, a);
    else if (np == 2 && ct == 16) hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<2, 16>), grid, block, lds, s, a);
    else if (np == 2) hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<2, 8>), grid, block, lds, s, a);
    else if (ct == 16) hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<3, 16>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gru_seq_fwd_persist_kernel<3, 8>), grid, block, lds, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// The cell nonlinearities of the persistent kernels, evaluated elementwise (tests: units in the last place against float64)
namespace {
__global__ void cell_probe_kernel(const float* x, float* sg, float* th, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        sg[i] = p_sigmoid(x[i]);
        th[i] = p_tanh(x[i]);
    }
}
}  // namespace
CPG_EXPORT int cpg_persistent_cell_probe(const float* x, float* sigmoid_out, float* tanh_out, int n, void* stream) {
    CPG_CHECK_ARG(x && sigmoid_out && tanh_out && n > 0);
    hipLaunchKernelGGL(cell_probe_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, sigmoid_out, tanh_out, n);
    CPG_LAUNCH_CHECK();
    return 0;
}

// Name of the kernel a persistent launch at width H runs, as rocprofv3 prints it (bench.py's roofline object)
CPG_EXPORT int cpg_gru_persistent_kernel_name(int H, char* buf, int n) {
    const int np = cpg_persist_planes();
    return snprintf(buf, n, "gru_seq_fwd_persist_kernel<%d, %d>", np, pick_ct(H, np));
}

// Byte offset of the sticky error word inside sync_scratch (tests plant a value there).  On the hot path the host does not
// read it: a wave that times out also sets err_host (pinned, host-mapped memory handed to the launch), which the host looks at
// without any stream operation.
CPG_EXPORT size_t cpg_gru_persistent_err_offset(int B) { return cnt_words(B) * sizeof(unsigned); }

// Error word of the last persistent launch that used this scratch (synchronises the stream): 0 = every wait completed.
CPG_EXPORT int cpg_gru_persistent_status(int B, const void* sync_scratch, void* stream) {
    unsigned v = 0;
    const unsigned* p = (const unsigned*)sync_scratch + cnt_words(B);
    if (hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}

