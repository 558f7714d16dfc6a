// Whole-loop greedy decoding for small decoders (hidden <= 128, vocab <= 32): RNN_VAE.sample_G(sample_mode='greedy'),
// models/model.py:225-385 with GRUDecoder.forward_sample, models/decoder.py:86-109, as ONE persistent launch.
//
// The reference's default decoder (h = z+c = 102, vocab 24) is far too small for one-launch-per-step to be efficient:
// W_hh is 125 KB, a step over N rows re-reads h [N,H] and the constant input term rowc [N,3H] from HBM and writes h back,
// 25 times.  Here a workgroup owns a tile of 64 latent samples for all T steps and nothing but the chosen token ids leaves
// the CU:
//   * W_hh lives in REGISTERS as f32 MFMA B-fragments: wave w owns hidden units [32w, 32w+32) of all three gates
//     (6 column tiles x K/4 k-steps = 156 VGPRs at H=102), so the r/z/n pre-activations of a unit meet in one lane;
//   * the hidden state tile [64][H], the per-sample constant term rowc [64][3H], the token table tab[V][3H] (= W_ih[:, :E]
//     . emb, models/decoder.py:67,92) and fc [V][H] are staged in LDS once per tile (149 KB of the CU's 160 KB);
//   * per step: h . W_hh^T on the matrix cores (A fragments read 16 B per lane from LDS with the contraction index
//     permuted as in gemm_core.h), the GRU cell in the accumulator layout, h' back to LDS, the vocabulary projection as
//     a second small MFMA product, first-max argmax + finished/EOS bookkeeping by 16 lanes per wave.
// One workgroup (4 waves, 1 per SIMD) per CU; the launch is persistent over tiles.
#include "cpg_internal.h"

namespace {

constexpr int RM = 64;  // latent samples per workgroup tile
constexpr int MT = RM / 16;
constexpr int LGS = 33;  // logits row stride in LDS (floats)

// K = 16*G + (up to 4*R) contraction steps: G full 16-deep groups (one ds_read_b128 per lane feeds four MFMA k-steps,
// lane group q supplies k = 16g + 4q + j at step j) + R plain k-steps for the tail (k = 16G + 4r + q).
template <int G, int R>
struct FusedCfg {
    static constexpr int KSTEPS = 4 * G + R;
    static constexpr int KP = 16 * G + 4 * R;
    // row stride = 4 * odd: the 8 rows one ds_read_b128 lane group touches land in disjoint bank quads
    static constexpr int LDH = ((KP / 4 + 1) % 2 == 1) ? KP + 4 : KP + 8;
};

struct GreedyArgs {
    const float* h0;    // [N,H]   = [z;c]
    const float* rowc;  // [N,3H]  W_ih[:, E:] . [z;c] + b_ih
    const float* tab;   // [Vt,3H] W_ih[:, :E] . emb[tok]
    const float* w_hh;  // [3H,H]
    const float* b_hh;  // [3H]
    const float* fc_w;  // [V,H]
    const float* fc_b;  // [V]
    int64_t* ids;       // [N,ld_ids]; this kernel writes columns 1..T
    int* unfinished;    // [T] += rows still running after each step
    int N, H, V, Vt, T, ld_ids, start, pad, eos, ntiles;
};

template <int G, int R>
__device__ __forceinline__ int k_of_step(int s, int lq) {
    return s < 4 * G ? 16 * (s >> 2) + 4 * lq + (s & 3) : 16 * G + 4 * (s - 4 * G) + lq;
}

template <int G, int R>
__global__ __launch_bounds__(256, 1) void decode_greedy_fused_kernel(GreedyArgs a) {
    using C = FusedCfg<G, R>;
    extern __shared__ float4 cpg_fused_smem[];
    const int H = a.H, H3 = 3 * a.H, V = a.V;
    float* h_l = reinterpret_cast<float*>(cpg_fused_smem);  // [RM][LDH]
    float* fc_l = h_l + RM * C::LDH;                         // [V][LDH]
    float* rowc_l = fc_l + V * C::LDH;                       // [RM][3H]
    float* tab_l = rowc_l + RM * H3;                         // [Vt][3H]
    float* logit_l = tab_l + a.Vt * H3;                      // [RM][LGS]
    int* tok_l = reinterpret_cast<int*>(logit_l + RM * LGS);  // [RM]
    int* fin_l = tok_l + RM;                                 // [RM]
    int* live_l = fin_l + RM;                                // [4] rows still running, per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;

    // ---- once per workgroup: W_hh fragments -> registers, b_hh -> registers, tab / fc -> LDS
    float Bf[C::KSTEPS][6];
    float bh[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int g = nt >> 1, unit = 32 * wave + 16 * (nt & 1) + l15;
        bh[nt] = unit < H ? a.b_hh[g * H + unit] : 0.f;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int k = k_of_step<G, R>(s, lq);
            Bf[s][nt] = (unit < H && k < H) ? a.w_hh[(size_t)(g * H + unit) * H + k] : 0.f;
        }
    }
    for (int i = tid; i < a.Vt * H3; i += 256) tab_l[i] = a.tab[i];
    for (int i = tid; i < V * C::LDH; i += 256) {
        const int v = i / C::LDH, k = i - v * C::LDH;
        fc_l[i] = k < H ? a.fc_w[(size_t)v * H + k] : 0.f;
    }
    const int vclamp0 = min(l15, V - 1), vclamp1 = min(16 + l15, V - 1);
    const float fcb0 = l15 < V ? a.fc_b[l15] : 0.f, fcb1 = 16 + l15 < V ? a.fc_b[16 + l15] : 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * RM, nrows = min(RM, a.N - row0);
        __syncthreads();  // previous tile fully retired before its LDS state is replaced
        for (int i = tid; i < RM * C::LDH; i += 256) {
            const int r = i / C::LDH, k = i - r * C::LDH;
            h_l[i] = (r < nrows && k < H) ? a.h0[(size_t)(row0 + r) * H + k] : 0.f;
        }
        {
            const float* src = a.rowc + (size_t)row0 * H3;
            const int n = nrows * H3;
            for (int i = tid; i < RM * H3; i += 256) rowc_l[i] = i < n ? src[i] : 0.f;
        }
        if (tid < RM) {
            tok_l[tid] = a.start;
            fin_l[tid] = 0;
        }
        if (tid < 4) live_l[tid] = 1;
        __syncthreads();

        for (int step = 0; step < a.T; ++step) {
            // ---- h . W_hh^T for this wave's 32 hidden units x 3 gates, all 64 rows
            f32x4 acc[MT][6];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    af[mt] = *reinterpret_cast<const f32x4*>(&h_l[(mt * 16 + l15) * C::LDH + 16 * g + 4 * lq]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 6; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], Bf[4 * g + j][nt], acc[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float at[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) at[mt] = h_l[(mt * 16 + l15) * C::LDH + 16 * G + 4 * r + lq];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[mt], Bf[4 * G + r][nt], acc[mt][nt], 0, 0, 0);
            }
            __syncthreads();  // every wave is done reading h (and last step's tokens are in tok_l)
            if (live_l[0] + live_l[1] + live_l[2] + live_l[3] == 0) break;  // whole tile finished: the rest stays <pad>

            // ---- GRU cell in the accumulator layout; h' goes straight back to this wave's own columns of h_l
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = mt * 16 + 4 * lq + jj;
                    const float* tr = tab_l + tok_l[row] * H3;
                    const float* rc = rowc_l + row * H3;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const int unit = 32 * wave + 16 * sub + l15;
                        if (unit < C::KP) {
                            float hv = 0.f;
                            if (unit < H) {
                                const float gi_r = tr[unit] + rc[unit];
                                const float gi_z = tr[H + unit] + rc[H + unit];
                                const float gi_n = tr[2 * H + unit] + rc[2 * H + unit];
                                const float hn = acc[mt][4 + sub][jj] + bh[4 + sub];
                                const float rg = sigmoidf_(gi_r + (acc[mt][sub][jj] + bh[sub]));
                                const float zg = sigmoidf_(gi_z + (acc[mt][2 + sub][jj] + bh[2 + sub]));
                                const float ng = tanhf(gi_n + rg * hn);
                                const float hold = h_l[row * C::LDH + unit];
                                hv = (1.f - zg) * ng + zg * hold;
                            }
                            h_l[row * C::LDH + unit] = hv;
                        }
                    }
                }
            __syncthreads();

            // ---- vocabulary projection of this wave's 16 rows: logits = h' fc_w^T + fc_b
            f32x4 lg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(&h_l[(wave * 16 + l15) * C::LDH + 16 * g + 4 * lq]);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&fc_l[vclamp0 * C::LDH + 16 * g + 4 * lq]);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(&fc_l[vclamp1 * C::LDH + 16 * g + 4 * lq]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], b0[j], lg[0], 0, 0, 0);
                    lg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], b1[j], lg[1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float at = h_l[(wave * 16 + l15) * C::LDH + 16 * G + 4 * r + lq];
                lg[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, fc_l[vclamp0 * C::LDH + 16 * G + 4 * r + lq], lg[0], 0, 0, 0);
                lg[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, fc_l[vclamp1 * C::LDH + 16 * G + 4 * r + lq], lg[1], 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int row = wave * 16 + 4 * lq + jj;
                logit_l[row * LGS + l15] = lg[0][jj] + fcb0;
                logit_l[row * LGS + 16 + l15] = lg[1][jj] + fcb1;
            }
            __syncthreads();

            // ---- token selection: lanes 0..15 of each wave take one row each (torch.argmax: first maximum)
            int live = 0;
            if (lane < 16) {
                const int row = wave * 16 + lane;
                if (row < nrows) {
                    const float* l = logit_l + row * LGS;
                    float best = -INFINITY;
                    int arg = 0;
                    for (int v = 0; v < V; ++v) {
                        const float x = l[v];
                        if (x > best) {
                            best = x;
                            arg = v;
                        }
                    }
                    const bool fin = fin_l[row] != 0;
                    const int t = fin ? a.pad : arg;
                    if (t == a.eos) fin_l[row] = 1;
                    live = (fin || t == a.eos) ? 0 : 1;
                    a.ids[(size_t)(row0 + row) * a.ld_ids + 1 + step] = t;
                    tok_l[row] = t;
                }
            }
            const unsigned long long lb = __ballot(live);
            if (lane == 0) {
                const int nl = __popcll(lb);
                live_l[wave] = nl;
                if (nl) atomicAdd(&a.unfinished[step], nl);
            }
        }
    }
}

template <int G, int R>
int launch_greedy(const GreedyArgs& a, int device_cus, hipStream_t s) {
    using C = FusedCfg<G, R>;
    const size_t floats = (size_t)RM * C::LDH + (size_t)a.V * C::LDH + (size_t)RM * 3 * a.H + (size_t)a.Vt * 3 * a.H + RM * LGS;
    const size_t bytes = floats * 4 + (2 * RM + 4) * sizeof(int);
    auto kern = decode_greedy_fused_kernel<G, R>;
    CPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int grid = a.ntiles < device_cus ? a.ntiles : device_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, s, a);
    CPG_LAUNCH_CHECK();
    return 0;
}

// contraction depth the dispatch below pads H to
int fused_kp(int H) {
    if (H / 16 == 6 && (H % 16 + 3) / 4 == 2) return 104;
    return H <= 32 ? 32 : H <= 64 ? 64 : H <= 96 ? 96 : 128;
}

}  // namespace

CPG_EXPORT size_t cpg_decode_greedy_fused_lds_bytes(int H, int V, int Vt) {
    if (H <= 0 || H > 128 || V <= 0 || V > 32 || Vt <= 0) return 0;
    const int kp = fused_kp(H), ldh = ((kp / 4 + 1) % 2 == 1) ? kp + 4 : kp + 8;
    return ((size_t)RM * ldh + (size_t)V * ldh + (size_t)RM * 3 * H + (size_t)Vt * 3 * H + RM * LGS) * 4 + (2 * RM + 4) * sizeof(int);
}

CPG_EXPORT int cpg_decode_greedy_fused(const float* h0, const float* rowc, const float* tab, int Vt, const float* w_hh,
                                       const float* b_hh, const float* fc_w, const float* fc_b, int N, int H, int V, int T,
                                       int start, int pad, int eos, int64_t* ids, int ld_ids, int* unfinished, void* stream) {
    CPG_CHECK_ARG(h0 && rowc && tab && w_hh && b_hh && fc_w && fc_b && ids && unfinished);
    CPG_CHECK_ARG(N > 0 && T > 0 && ld_ids >= T + 1 && H > 0 && H <= 128 && V > 0 && V <= 32 && Vt > 0);
    CPG_CHECK_ARG(start >= 0 && start < Vt && pad >= 0 && pad < Vt && eos >= 0 && V <= Vt);
    int dev = 0, cus = 0, lds = 0;
    CPG_HIP(hipGetDevice(&dev));
    CPG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CPG_HIP(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const size_t need = cpg_decode_greedy_fused_lds_bytes(H, V, Vt);
    if (need > (size_t)lds) {
        cpg_set_error("cpg_decode_greedy_fused: needs %zu bytes of LDS per workgroup, device offers %d", need, lds);
        return -3;
    }
    GreedyArgs a{h0, rowc, tab, w_hh, b_hh, fc_w, fc_b, ids, unfinished, N, H, V, Vt, T, ld_ids, start, pad, eos, cdiv(N, RM)};
    hipStream_t s = (hipStream_t)stream;
    const int g = H / 16, r = (H % 16 + 3) / 4;
    // instantiated shapes: the reference default (h_dim = 102 -> 6 groups + 2 tail steps) and the padded general case
    if (g == 6 && r == 2) return launch_greedy<6, 2>(a, cus, s);
    if (H <= 32) return launch_greedy<2, 0>(a, cus, s);
    if (H <= 64) return launch_greedy<4, 0>(a, cus, s);
    if (H <= 96) return launch_greedy<6, 0>(a, cus, s);
    return launch_greedy<8, 0>(a, cus, s);
}
