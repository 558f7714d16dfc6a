// Persistent GRU sequence kernels for gfx950: the WHOLE time loop of one direction of one layer in ONE launch.
//
// Replaces the per-step launches of csrc/gru.hip (torch.nn.GRU at models/encoder.py:25-30,42; models/decoder.py:40-41,77)
// when the problem fits the chip (cpg_gru_persistent_fits).  Why: a step of the recurrence at B=2048, H=512 is a
// [2048,512] x [512,1536] product - 8 us of matrix-pipe time as six bf16 MFMAs per block on 3-way split operands - but a
// launch per step spends 38 us on it: every launch re-stages and re-converts W_hh through LDS in every row tile, pays the
// launch ramp, two dependent gather latencies before its epilogue and a store tail after it.
//
// Decomposition (rows are independent recurrences; columns need the whole previous state):
//   * a workgroup owns CT = 16 hidden units (the r, z, n rows of W_hh for them: 48 x H) for a group of 256 batch rows and
//     keeps that W_hh slice in LDS for the whole sequence, ALREADY split into three bf16 planes (48 x H x 6 B = 147 KB at
//     H = 512: the reason for CT = 16 and for one workgroup per CU);
//   * each of its 4 waves owns 64 of the rows.  The state operand h_{t-1}[64 rows, H] goes global -> registers -> MFMA
//     A fragments directly (every element is used by exactly one wave, so there is nothing to share through LDS and the
//     time loop has NO workgroup barrier); the per-row constant input term and the previous state of the wave's own
//     64 x 16 outputs stay in registers across steps;
//   * the 32 column-tile workgroups of a row tile exchange h_t through the state slab itself: write-through (sc1) 16-byte
//     stores, vmcnt(0), one relaxed agent-scope atomic add on the row tile's arrival counter; consumers poll that counter
//     relaxed, then read the slab with sc1 loads (L1 bypassed: no acquire fence needed) - the placement-independent
//     hand-off of the CDNA guide (Guideline 16, R1).  Every step writes its own slab slot: no buffer is ever reused,
//     so there is no write-after-read hazard.  Counters are zeroed by the host before every launch; spins are bounded.
// Arithmetic is the per-step kernel's (same split, same MFMA order, same cell formulas): results are f32-grade and the
// golden / oracle parity tests run unchanged on this path.
#include "gemm_core.h"
#include "cpg_internal.h"
#include <stdlib.h>

namespace {

constexpr int P_CT = 16;          // hidden units per workgroup
constexpr int P_WROWS = 64;       // rows per wave
constexpr int P_WAVES = 4;
constexpr int P_NC = 3 * P_CT;    // gate columns per workgroup
constexpr int P_TBW = 20;         // words per row of the per-wave 16x16 transposition buffer
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
    float* gates;          // [T,4,B,H] or null
    unsigned* cnt;         // [row tiles] arrival counters (zeroed before the launch)
    unsigned* err;         // error word (zeroed before the launch)
    int T, B, H, reverse, groups, S;  // S: words per plane row (H/2 data + pad so that S % 64 == 8)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Wait until *p >= target (relaxed agent-scope polls: every lane reads the same word).  Bounded: on timeout the error
// word is set and the wave carries on with whatever is in memory - the run is wrong, but it ends.
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* err, bool& dead) {
    if (dead) return false;
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P_SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

// 16x16 tile held in the MFMA accumulator layout (lane (u = l&15, rq = l>>4), reg -> row 4rq+reg, col u) -> one float4 per
// lane in row layout (lane -> row l>>2, cols 4(l&3)..+3), through the wave's own LDS buffer (no other wave touches it).
__device__ __forceinline__ f32x4 acc_to_rows(float* tb, const float (&v)[4], int lane) {
    const int u = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[(4 * rq + r) * P_TBW + u] = v[r];
    return *reinterpret_cast<const f32x4*>(tb + (lane >> 2) * P_TBW + 4 * (lane & 3));
}

__global__ __launch_bounds__(256, 1) void gru_seq_fwd_persist_kernel(PFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t psm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x % a.groups, ct = blockIdx.x / a.groups;
    const int H = a.H, B = a.B, T = a.T, S = a.S;
    const int j0 = ct * P_CT;
    const int NCT = H / P_CT, KB = H / 32;
    const int PLW = P_NC * S;
    uint32_t* const planes = psm;
    float* const tb = reinterpret_cast<float*>(psm + 3 * PLW) + wave * (16 * P_TBW);

    // ---- W_hh slice -> three bf16 planes in LDS, once per sequence: plane[c = gate*16 + u][k pair]
    for (int idx = tid; idx < P_NC * (H / 2); idx += 256) {
        const int c = idx / (H / 2), kp = idx - c * (H / 2);
        const float2 v = *reinterpret_cast<const float2*>(a.w_hh + ((size_t)((c >> 4) * H + j0 + (c & 15))) * H + 2 * kp);
        uint32_t w0, w1, w2;
        split3_pair(v.x, v.y, w0, w1, w2);
        planes[c * S + kp] = w0;
        planes[PLW + c * S + kp] = w1;
        planes[2 * PLW + c * S + kp] = w2;
    }
    __syncthreads();

    const int rt = g * P_WAVES + wave;   // row tile of this wave
    const int row0 = rt * P_WROWS;
    if (row0 >= B) return;               // wave-uniform; nobody waits for a tile that does not exist
    const int l15 = lane & 15, lq = lane >> 4;
    const int col = j0 + l15;            // hidden unit of this lane's accumulator elements

    // per-lane constants of the epilogue: row of accumulator element (mi, reg), clamped for the loads
    float rc[4][4][3], hprev[4][4];
    const size_t slot0 = (size_t)(a.reverse ? T : 0) * B * H;
    float bh[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) bh[q] = a.b_hh[q * H + col];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + 16 * mi + 4 * lq + r, B - 1);
#pragma unroll
            for (int q = 0; q < 3; ++q) rc[mi][r][q] = a.rowc ? a.rowc[(size_t)row * 3 * H + q * H + col] : 0.f;
            hprev[mi][r] = a.hs[slot0 + (size_t)row * H + col];
        }
    // A-operand addressing: lane (l15, lq) of row block mi reads 8 consecutive k of row row0 + 16 mi + l15
    int aoff[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = (min(row0 + 16 * mi + l15, B - 1) * H + 8 * lq) * 4;
    const uint32_t* const bbase = planes + l15 * S + 4 * lq;
    const unsigned slab_bytes = (unsigned)((size_t)B * H * 4);
    bool dead = false;

    for (int p = 0; p < T; ++p) {
        const int tt = a.reverse ? T - 1 - p : p;
        const float* const hin = a.hs + (size_t)(a.reverse ? tt + 1 : tt) * B * H;
        float* const hout = a.hs + (size_t)(a.reverse ? tt : tt + 1) * B * H;

        // input-side pre-activations of this step: independent of the recurrence, fetched before the wait
        float gi[4][4][3];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(row0 + 16 * mi + 4 * lq + r, B - 1);
                float x0 = rc[mi][r][0], x1 = rc[mi][r][1], x2 = rc[mi][r][2];
                if (a.tok) {
                    const float* t = a.tab + (size_t)a.tok[(size_t)tt * B + row] * 3 * H + col;
                    x0 += t[0]; x1 += t[H]; x2 += t[2 * H];
                }
                if (a.dense) {
                    const float* t = a.dense + ((size_t)tt * B + row) * 3 * H + col;
                    x0 += t[0]; x1 += t[H]; x2 += t[2 * H];
                }
                gi[mi][r][0] = x0; gi[mi][r][1] = x1; gi[mi][r][2] = x2;
            }

        if (p > 0) wait_ge(a.cnt + rt, (unsigned)(NCT * p), a.err, dead);

        // ---- recurrent product: acc[mi][gate] = h_prev[64 rows, H] . W_hh[gate rows of 16 units, H]^T
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(hin, slab_bytes);
        f32x4 acc[4][3];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[mi][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 bufA[4][2], bufB[4][2];
        auto load = [&](u32x4 (&buf)[4][2], int kb) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                buf[mi][0] = __builtin_amdgcn_raw_buffer_load_b128(rin, aoff[mi] + kb * 128, 0, 16);
                buf[mi][1] = __builtin_amdgcn_raw_buffer_load_b128(rin, aoff[mi] + kb * 128 + 16, 0, 16);
            }
        };
        auto compute = [&](const u32x4 (&buf)[4][2], int kb) {
            cpg_bf16x8 fb[3][3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[q][pl] = *reinterpret_cast<const cpg_bf16x8*>(bbase + pl * PLW + q * 16 * S + kb * 16);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f32x4 lo = __builtin_bit_cast(f32x4, buf[mi][0]), hi = __builtin_bit_cast(f32x4, buf[mi][1]);
                uint32_t w0[4], w1[4], w2[4];
                split3_pair(lo[0], lo[1], w0[0], w1[0], w2[0]);
                split3_pair(lo[2], lo[3], w0[1], w1[1], w2[1]);
                split3_pair(hi[0], hi[1], w0[2], w1[2], w2[2]);
                split3_pair(hi[2], hi[3], w0[3], w1[3], w2[3]);
                const cpg_bf16x8 fa0 = __builtin_bit_cast(cpg_bf16x8, make_uint4(w0[0], w0[1], w0[2], w0[3]));
                const cpg_bf16x8 fa1 = __builtin_bit_cast(cpg_bf16x8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
                const cpg_bf16x8 fa2 = __builtin_bit_cast(cpg_bf16x8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    f32x4 c = acc[mi][q];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa2, fb[q][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb[q][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb[q][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb[q][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb[q][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb[q][0], c, 0, 0, 0);
                    acc[mi][q] = c;
                }
            }
        };
        load(bufA, 0);
        for (int kb = 0; kb < KB; kb += 2) {
            if (kb + 1 < KB) load(bufB, kb + 1);
            compute(bufA, kb);
            if (kb + 2 < KB) load(bufA, kb + 2);
            if (kb + 1 < KB) compute(bufB, kb + 1);
        }

        // ---- cell (same formulas and association as gru_step_fwd_kernel)
        float rg[4][4], zg[4][4], ng[4][4], hn[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hn[mi][r] = acc[mi][2][r] + bh[2];
                rg[mi][r] = sigmoidf_(gi[mi][r][0] + (acc[mi][0][r] + bh[0]));
                zg[mi][r] = sigmoidf_(gi[mi][r][1] + (acc[mi][1][r] + bh[1]));
                ng[mi][r] = tanhf(gi[mi][r][2] + rg[mi][r] * hn[mi][r]);
                hprev[mi][r] = (1.f - zg[mi][r]) * ng[mi][r] + zg[mi][r] * hprev[mi][r];
            }

        // ---- publish h_t: write-through 16-byte stores, drain, one arrival per wave
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(hout, slab_bytes);
        const int srow = lane >> 2, scq = lane & 3;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const f32x4 v = acc_to_rows(tb, hprev[mi], lane);
            const int row = row0 + 16 * mi + srow;
            if (row < B)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (row * H + j0 + 4 * scq) * 4, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(a.cnt + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ---- saved-for-backward gates (not part of the hand-off: streamed out behind the arrival)
        if (a.gates) {
            const size_t BH = (size_t)B * H;
            float* const gbase = a.gates + (size_t)tt * 4 * BH;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = row0 + 16 * mi + srow;
                const f32x4 v0 = acc_to_rows(tb, rg[mi], lane);
                const f32x4 v1 = acc_to_rows(tb, zg[mi], lane);
                const f32x4 v2 = acc_to_rows(tb, ng[mi], lane);
                const f32x4 v3 = acc_to_rows(tb, hn[mi], lane);
                if (row < B) {
                    float* d = gbase + (size_t)row * H + j0 + 4 * scq;
                    __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(d));
                    __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(d + BH));
                    __builtin_nontemporal_store(v2, reinterpret_cast<f32x4*>(d + 2 * BH));
                    __builtin_nontemporal_store(v3, reinterpret_cast<f32x4*>(d + 3 * BH));
                }
            }
        }
    }
}

int plane_stride_words(int H) {
    int s = H / 2;
    while (s % 64 != 8) ++s;   // rows start 8 words apart modulo the 64 banks: conflict-free ds_read_b128 fragments
    return s;
}

size_t fwd_lds_bytes(int H) { return ((size_t)3 * P_NC * plane_stride_words(H) + P_WAVES * 16 * P_TBW) * 4; }

int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        cus = pr.multiProcessorCount;
    }
    return cus;
}

}  // namespace

// 1 when the persistent kernels cover a [B rows, H hidden] sequence on this device: one workgroup (16 hidden units x 256
// rows) per CU, all of them co-resident.  CPG_GRU_PERSIST=0 disables the path (per-step launches), =1 is the default.
CPG_EXPORT int cpg_gru_persistent_fits(int B, int H) {
    const char* e = getenv("CPG_GRU_PERSIST");
    if (e && atoi(e) == 0) return 0;
    if (B <= 0 || H < 32 || H % 32 != 0) return 0;
    if (fwd_lds_bytes(H) > 160 * 1024) return 0;
    const int groups = cdiv(cdiv(B, P_WROWS), P_WAVES);
    const long wgs = (long)groups * (H / P_CT);
    const int cus = device_cus();
    return cus > 0 && wgs <= cus;
}

CPG_EXPORT size_t cpg_gru_persistent_scratch_bytes(int B) { return (size_t)(cdiv(B, P_WROWS) + 16) * sizeof(unsigned); }

// Whole forward sequence in one launch; arguments as cpg_gru_seq_fwd (all rows).  sync_scratch: device memory of
// cpg_gru_persistent_scratch_bytes(B) bytes, ZEROED BY THE CALLER when allocated: arrival counters (re-zeroed here on the
// stream before every launch) followed by a sticky error word (set by a wave whose wait timed out, never cleared here).
CPG_EXPORT int cpg_gru_seq_fwd_persistent(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                                          const int32_t* tok, const float* tab, const float* rowc, const float* dense,
                                          float* hs, float* gates, void* sync_scratch, void* stream) {
    CPG_CHECK_ARG(T > 0 && B > 0 && H > 0 && w_hh && b_hh && hs && sync_scratch);
    CPG_CHECK_ARG((tok == nullptr) == (tab == nullptr));
    if (!cpg_gru_persistent_fits(B, H)) {
        cpg_set_error("cpg_gru_seq_fwd_persistent: B=%d H=%d does not fit the persistent kernel on this device", B, H);
        return -5;
    }
    hipStream_t s = (hipStream_t)stream;
    const int nrt = cdiv(B, P_WROWS);
    CPG_HIP(hipMemsetAsync(sync_scratch, 0, (size_t)nrt * sizeof(unsigned), s));  // counters only: the error word is sticky
    PFwdArgs a;
    a.w_hh = w_hh; a.b_hh = b_hh; a.tok = tok; a.tab = tab; a.rowc = rowc; a.dense = dense; a.hs = hs; a.gates = gates;
    a.cnt = (unsigned*)sync_scratch;
    a.err = a.cnt + nrt;
    a.T = T; a.B = B; a.H = H; a.reverse = reverse;
    a.groups = cdiv(nrt, P_WAVES);
    a.S = plane_stride_words(H);
    const size_t smem = fwd_lds_bytes(H);
    static bool attr = false;
    if (!attr) {
        CPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_fwd_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(gru_seq_fwd_persist_kernel, dim3(a.groups * (H / P_CT)), dim3(256), smem, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// Error word of the last persistent launch that used this scratch (synchronises the stream): 0 = every wait completed.
CPG_EXPORT int cpg_gru_persistent_status(int B, const void* sync_scratch, void* stream) {
    unsigned v = 0;
    const unsigned* p = (const unsigned*)sync_scratch + cdiv(B, P_WROWS);
    if (hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}
