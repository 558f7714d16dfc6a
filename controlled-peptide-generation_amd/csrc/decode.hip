// Vocabulary projection and the per-step token selection of autoregressive decoding.
//   cpg_vocab_fc_fwd / _bwd   nn.Dropout(p_out) + nn.Linear(h_dim, n_vocab)        models/decoder.py:43-45,83
//   cpg_greedy_select         argmax + finished masking of RNN_VAE.sample_G         models/model.py:310-311,350-353,362-363
//   cpg_categorical_select    Categorical(logits/temp).sample() by inverse CDF      models/model.py:308-309,350-353
//   cpg_beam_select           log_softmax + Beam.advance + hidden-state reorder      models/model.py:314-328,387-404; models/Beam.py:56-105
#include "cpg_internal.h"

CPG_EXPORT int cpg_vocab_fc_fwd(const float* hs, const uint8_t* keep, float scale, const float* w, const float* b,
                                float* logits, int R, int H, int V, void* stream) {
    CPG_CHECK_ARG(hs && w && logits && R > 0 && H > 0 && V > 0);
    return cpg_gemm_nt(hs, H, keep, scale, w, H, b, logits, V, R, V, H, 0, (hipStream_t)stream);
}

// ---- backward for small vocabularies (V <= 32: the peptide alphabet has 24 symbols) -----------------------------------------
// dhs = (dlogits W) .* keep is a K = V product whose output (R x H f32: 105 MB at config B) is the whole cost, and dw = dlogits^T hs
// an M = V product that reads hs once; on the tile engine the two took 79 + 43 us (+ 41 us of reductions) at config B - staging
// 32-deep slabs for a 24-deep contraction.  Here a wave owns 256 columns (one float4 per lane): W's / the accumulators' V float4
// live in registers, a row's V gradient values are wave-uniform (scalar loads), the arithmetic is plain f32 FMA (exact products,
// f32 accumulation in row order), rows stream through with four in flight per wave.
__device__ __forceinline__ float lane_value(float x, int l) {   // x of lane l (compile-time l) as a wave-uniform value
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

template <int VV, bool MASK>
__global__ __launch_bounds__(256) void vocab_bwd_dh_kernel(const float* __restrict__ dl, const uint8_t* __restrict__ keep, float scale,
                                                           const float* __restrict__ w, float* __restrict__ dhs, int R, int H, int V,
                                                           const float* __restrict__ gnum, const float* __restrict__ gden) {
    constexpr int NR = 8;
    const float gs = gnum ? gnum[0] / (gden ? fmaxf(gden[0], 1.f) : 1.f) : 1.f;   // dl arrives unscaled (cpg_recon_ce_tm_fwd): times gout / count   // rows per trip; the next trip's loads are issued before this trip's arithmetic (no branch in the pipelined loop)
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nqb = H >> 8, nrs = (gridDim.x * 4) / nqb;   // 256-column blocks, row streams
    const int qb = gw % nqb, rs = gw / nqb;
    const int col = qb * 256 + 4 * lane;
    const bool lvok = lane < V;
    const int lv = lvok ? lane : 0;
    float4 wv[VV];
#pragma unroll
    for (int v = 0; v < VV; ++v) wv[v] = v < V ? *reinterpret_cast<const float4*>(w + (size_t)v * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint8_t* kbase = MASK ? keep + col : nullptr;
    struct Trip {
        uchar4 kp[NR];
        float dv[NR];   // lane v holds dlogits[row][v] (0 from lane V on)
    };
    auto load = [&](int rb, Trip& t) {   // rows rb + j nrs, all < R
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t r = (size_t)(rb + j * nrs);
            if constexpr (MASK) t.kp[j] = *reinterpret_cast<const uchar4*>(kbase + r * H);
            t.dv[j] = dl[r * V + lv];
        }
    };
    auto row = [&](int r, uchar4 kp, float dv) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        dv = lvok ? dv * gs : 0.f;
#pragma unroll
        for (int v = 0; v < VV; ++v) {
            const float g = lane_value(dv, v);
            a.x = fmaf(g, wv[v].x, a.x);
            a.y = fmaf(g, wv[v].y, a.y);
            a.z = fmaf(g, wv[v].z, a.z);
            a.w = fmaf(g, wv[v].w, a.w);
        }
        if constexpr (MASK) {
            a.x *= kp.x ? scale : 0.f;
            a.y *= kp.y ? scale : 0.f;
            a.z *= kp.z ? scale : 0.f;
            a.w *= kp.w ? scale : 0.f;
        }
        *reinterpret_cast<float4*>(dhs + (size_t)r * H + col) = a;
    };
    auto work = [&](int rb, const Trip& t) {
#pragma unroll
        for (int j = 0; j < NR; ++j) row(rb + j * nrs, t.kp[j], t.dv[j]);
    };
    const int cnt = rs < R ? (R - rs + nrs - 1) / nrs : 0;   // rows of this stream: rs + i nrs
    const int npair = cnt / (2 * NR);                          // pipelined part: pairs of trips
    const int step = NR * nrs;
    if (npair > 0) {
        Trip ta, tb;
        load(rs, ta);
        for (int p = 0; p < npair; ++p) {
            const int rb = rs + 2 * p * step;
            const int nx = rs + 2 * (p + 1 < npair ? p + 1 : p) * step;   // the last pair re-loads itself: no branch around the loads
            load(rb + step, tb);
            work(rb, ta);
            load(nx, ta);
            work(rb + step, tb);
        }
    }
    for (int i = npair * 2 * NR; i < cnt; ++i) {
        const size_t r = (size_t)(rs + i * nrs);
        uchar4 kp = make_uchar4(1, 1, 1, 1);
        if constexpr (MASK) kp = *reinterpret_cast<const uchar4*>(kbase + r * H);
        row((int)r, kp, dl[r * V + lv]);
    }
}

// part[g][v][:] = sum over the rows of workgroup g of dlogits[r][v] * (hs[r][:] .* keep*scale); part_db[g'][v] = sum of dlogits[r][v].
// A workgroup's four waves are min(H/256, 4) column blocks x row lanes; the row lanes are summed through LDS in lane order.
template <int VV, bool MASK>
__global__ __launch_bounds__(256) void vocab_bwd_dw_kernel(const float* __restrict__ dl, const float* __restrict__ hs,
                                                           const uint8_t* __restrict__ keep, float scale, float* __restrict__ part,
                                                           float* __restrict__ part_db, int R, int H, int V, int rows_per_wg,
                                                           const float* __restrict__ gnum, const float* __restrict__ gden) {
    constexpr int NR = 4;
    const float gs = gnum ? gnum[0] / (gden ? fmaxf(gden[0], 1.f) : 1.f) : 1.f;
    extern __shared__ __attribute__((aligned(16))) float cpg_smem[];
    float4* red = reinterpret_cast<float4*>(cpg_smem);   // [qpw][VV][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = H >> 8, qpw = nqb < 4 ? nqb : 4, nrl = 4 / qpw;
    const int ql = wave % qpw, rl = wave / qpw;
    const int qb = blockIdx.y * qpw + ql;
    const int col = qb * 256 + 4 * lane;
    const int r0 = blockIdx.x * rows_per_wg + rl, r1 = min(R, (int)(blockIdx.x + 1) * rows_per_wg);
    const bool lvok = lane < V;
    const int lv = lvok ? lane : 0;
    float4 acc[VV];
#pragma unroll
    for (int v = 0; v < VV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dbacc = 0.f;
    const uint8_t* kbase = MASK ? keep + col : nullptr;
    struct Trip {
        float4 hv[NR];
        uchar4 kp[NR];
        float dv[NR];
    };
    auto load = [&](int rb, Trip& t) {   // rows rb + j nrl, all < r1
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const size_t r = (size_t)(rb + j * nrl);
            t.hv[j] = *reinterpret_cast<const float4*>(hs + r * H + col);
            if constexpr (MASK) t.kp[j] = *reinterpret_cast<const uchar4*>(kbase + r * H);
            t.dv[j] = dl[r * V + lv];
        }
    };
    auto row = [&](float4 h, uchar4 kp, float dv) {
        dv = lvok ? dv * gs : 0.f;
        if constexpr (MASK) {
            h.x *= kp.x ? scale : 0.f;
            h.y *= kp.y ? scale : 0.f;
            h.z *= kp.z ? scale : 0.f;
            h.w *= kp.w ? scale : 0.f;
        }
#pragma unroll
        for (int v = 0; v < VV; ++v) {
            const float g = lane_value(dv, v);
            acc[v].x = fmaf(g, h.x, acc[v].x);
            acc[v].y = fmaf(g, h.y, acc[v].y);
            acc[v].z = fmaf(g, h.z, acc[v].z);
            acc[v].w = fmaf(g, h.w, acc[v].w);
        }
        dbacc += dv;
    };
    auto work = [&](const Trip& t) {
#pragma unroll
        for (int j = 0; j < NR; ++j) row(t.hv[j], t.kp[j], t.dv[j]);
    };
    const int cnt = r0 < r1 ? (r1 - r0 + nrl - 1) / nrl : 0;
    const int npair = cnt / (2 * NR);
    const int step = NR * nrl;
    if (npair > 0) {
        Trip ta, tb;
        load(r0, ta);
        for (int p = 0; p < npair; ++p) {
            const int rb = r0 + 2 * p * step;
            const int nx = r0 + 2 * (p + 1 < npair ? p + 1 : p) * step;
            load(rb + step, tb);
            work(ta);
            load(nx, ta);
            work(tb);
        }
    }
    for (int i = npair * 2 * NR; i < cnt; ++i) {
        const size_t r = (size_t)(r0 + i * nrl);
        uchar4 kp = make_uchar4(1, 1, 1, 1);
        if constexpr (MASK) kp = *reinterpret_cast<const uchar4*>(kbase + r * H);
        row(*reinterpret_cast<const float4*>(hs + r * H + col), kp, dl[r * V + lv]);
    }
    for (int l = 1; l < nrl; ++l) {
        if (rl == l) {
#pragma unroll
            for (int v = 0; v < VV; ++v) red[(ql * VV + v) * 64 + lane] = acc[v];
        }
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int v = 0; v < VV; ++v) {
                const float4 t = red[(ql * VV + v) * 64 + lane];
                acc[v].x += t.x;
                acc[v].y += t.y;
                acc[v].z += t.z;
                acc[v].w += t.w;
            }
        }
        __syncthreads();
    }
    if (rl == 0) {
#pragma unroll
        for (int v = 0; v < VV; ++v)
            if (v < V) *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * V + v) * H + col) = acc[v];
    }
    if (qb == 0 && lane < 32) part_db[((size_t)blockIdx.x * nrl + rl) * 32 + lane] = dbacc;
}

// dw[v][c] (+)= sum_g part[g][v][c], db[v] (+)= sum_g part_db[g][v]: 32 float4 columns x 32 slab lanes per block, fixed order.
__global__ __launch_bounds__(1024) void vocab_bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ part_db, int G, int GDB,
                                                               int V, int H, float* dw, float* db, int accumulate) {
    __shared__ float4 red[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int nq = V * H / 4;
    const int nblk = (nq + 31) / 32;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((int)blockIdx.x < nblk) {
        const int q = blockIdx.x * 32 + tx;
        if (q < nq)
            for (int g = ty; g < G; g += 32) {
                const float4 t = *reinterpret_cast<const float4*>(part + ((size_t)g * nq + q) * 4);
                s.x += t.x;
                s.y += t.y;
                s.z += t.z;
                s.w += t.w;
            }
    } else if (tx < V) {   // the bias block: one column per tx
        for (int g = ty; g < GDB; g += 32) s.x += part_db[(size_t)g * 32 + tx];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
        float4 t = red[0][tx];
#pragma unroll
        for (int j = 1; j < 32; ++j) {
            t.x += red[j][tx].x;
            t.y += red[j][tx].y;
            t.z += red[j][tx].z;
            t.w += red[j][tx].w;
        }
        if ((int)blockIdx.x < nblk) {
            const int q = blockIdx.x * 32 + tx;
            if (q < nq && dw) {
                float4* o = reinterpret_cast<float4*>(dw) + q;
                if (accumulate) {
                    const float4 c = *o;
                    t.x += c.x;
                    t.y += c.y;
                    t.z += c.z;
                    t.w += c.w;
                }
                *o = t;
            }
        } else if (tx < V && db) {
            db[tx] = accumulate ? db[tx] + t.x : t.x;
        }
    }
}

// the streaming form covers: V <= 32, H a multiple of 256, enough rows to fill the chip; CPG_VOCAB_BWD=gemm keeps the tile engine
static bool vocab_bwd_streams(int R, int H, int V) {
    static const bool off = [] { const char* e = getenv("CPG_VOCAB_BWD"); return e && !strcmp(e, "gemm"); }();
    // (H / 256 = 3 is NOT covered: four waves per workgroup would be 3 column blocks + a fourth wave re-walking rows of the first -
    // its bias partial landed in the next workgroup's slot; that width goes to the tile engine - round-5 advisor finding)
    const int nqb = H / 256;
    return !off && V <= 32 && H % 256 == 0 && (nqb == 1 || nqb == 2 || nqb % 4 == 0) && R >= 4096;
}
static int vocab_bwd_wgs(int R) {
    int g = 2 * cpg_device_cus();
    if (g > R / 32) g = R / 32;
    return g < 1 ? 1 : g;
}

CPG_EXPORT size_t cpg_vocab_fc_bwd_workspace(int R, int H, int V) {
    size_t a = cpg_gemm_tn_workspace(R, V, H), b = cpg_colsum_workspace(R, V);
    size_t c = vocab_bwd_streams(R, H, V) ? (size_t)vocab_bwd_wgs(R) * ((size_t)V * H + 4 * 32) * sizeof(float) : 0;
    a = a > b ? a : b;
    return (a > c ? a : c) + (size_t)R * V * sizeof(float) + 256;   // + a scaled copy of dlogits for the tile-engine fallback with g / count
}

template <int VV>
static int vocab_bwd_launch(const float* dl, const float* hs, const uint8_t* keep, float scale, const float* w, float* dhs, float* dw, float* db,
                            int R, int H, int V, int accumulate, float* ws, hipStream_t s, const float* gnum, const float* gden) {
    if (dhs) {
        int g = 2 * cpg_device_cus();
        while ((g * 4) % (H / 256)) ++g;
        if (keep)
            hipLaunchKernelGGL((vocab_bwd_dh_kernel<VV, true>), dim3(g), dim3(256), 0, s, dl, keep, scale, w, dhs, R, H, V, gnum, gden);
        else
            hipLaunchKernelGGL((vocab_bwd_dh_kernel<VV, false>), dim3(g), dim3(256), 0, s, dl, keep, scale, w, dhs, R, H, V, gnum, gden);
        CPG_LAUNCH_CHECK();
    }
    if (dw || db) {
        const int G = vocab_bwd_wgs(R), nqb = H / 256, qpw = nqb < 4 ? nqb : 4, nrl = 4 / qpw;
        const int rows = cdiv(R, G);
        float* part = ws;
        float* part_db = ws + (size_t)G * V * H;
        const size_t smem = nrl > 1 ? (size_t)qpw * VV * 64 * sizeof(float4) : 0;
        if (keep)
            hipLaunchKernelGGL((vocab_bwd_dw_kernel<VV, true>), dim3(G, nqb / qpw), dim3(256), smem, s, dl, hs, keep, scale, part, part_db, R, H, V, rows, gnum, gden);
        else
            hipLaunchKernelGGL((vocab_bwd_dw_kernel<VV, false>), dim3(G, nqb / qpw), dim3(256), smem, s, dl, hs, keep, scale, part, part_db, R, H, V, rows, gnum, gden);
        CPG_LAUNCH_CHECK();
        const int nblk = cdiv(V * H / 4, 32);
        hipLaunchKernelGGL(vocab_bwd_final_kernel, dim3(nblk + 1), dim3(32, 32), 0, s, part, part_db, G, G * nrl, V, H, dw, db, accumulate);
        CPG_LAUNCH_CHECK();
    }
    return 0;
}

// dhs[R,H] = (dlogits W) .* keep*scale ; dw[V,H] (+)= dlogits^T (hs .* keep*scale) ; db[V] (+)= colsum(dlogits)
// g / count (optional device scalars): dlogits arrives UNSCALED (cpg_recon_ce_tm_fwd) and is multiplied by g[0] / max(count[0], 1) on
// the way in (count null: by g[0]); the tile-engine fallback scales a copy first (workspace: + R V floats then).
__global__ void scale_rows_kernel(const float* __restrict__ x, size_t n, const float* gnum, const float* gden, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * (gnum[0] / (gden ? fmaxf(gden[0], 1.f) : 1.f));
}
CPG_EXPORT int cpg_vocab_fc_bwd(const float* dlogits, const float* hs, const uint8_t* keep, float scale, const float* w,
                                float* dhs, float* dw, float* db, int R, int H, int V, int accumulate, const float* g, const float* count,
                                void* workspace, size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(dlogits && hs && w && R > 0 && H > 0 && V > 0 && (g || !count));
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    if (vocab_bwd_streams(R, H, V) && (!(dw || db) || (dw && db && workspace && workspace_bytes >= cpg_vocab_fc_bwd_workspace(R, H, V) - 256)) &&
        aligned16(hs) && aligned16(w) && (!dhs || aligned16(dhs)) && (!dw || aligned16(dw))) {
        float* ws = (float*)workspace;
        if (V <= 8) return vocab_bwd_launch<8>(dlogits, hs, keep, scale, w, dhs, dw, db, R, H, V, accumulate, ws, s, g, count);
        if (V <= 16) return vocab_bwd_launch<16>(dlogits, hs, keep, scale, w, dhs, dw, db, R, H, V, accumulate, ws, s, g, count);
        if (V <= 24) return vocab_bwd_launch<24>(dlogits, hs, keep, scale, w, dhs, dw, db, R, H, V, accumulate, ws, s, g, count);
        return vocab_bwd_launch<32>(dlogits, hs, keep, scale, w, dhs, dw, db, R, H, V, accumulate, ws, s, g, count);
    }
    if (g) {   // the generic products take a scaled copy (tail of the workspace)
        const size_t need = cpg_vocab_fc_bwd_workspace(R, H, V);
        CPG_CHECK_ARG(workspace && workspace_bytes >= need);
        float* sc = (float*)((char*)workspace + need - 256 - (size_t)R * V * sizeof(float));
        const size_t n = (size_t)R * V;
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dlogits, n, g, count, sc);
        CPG_LAUNCH_CHECK();
        dlogits = sc;
        workspace_bytes = need - 256 - (size_t)R * V * sizeof(float);
    }
    if (dhs) rc = cpg_gemm_nn(dlogits, V, w, H, dhs, H, R, H, V, 0, keep, scale, s);
    if (rc) return rc;
    if (dw) {
        rc = cpg_gemm_tn(dlogits, V, hs, H, keep, scale, dw, H, R, V, H, accumulate, (float*)workspace, workspace_bytes, s);
        if (rc) return rc;
    }
    if (db) rc = cpg_colsum(dlogits, V, R, V, db, accumulate, (float*)workspace, workspace_bytes, s);
    return rc;
}

// ------------------------------------------------------------------------------------------ greedy
__global__ void min_partial_kernel(const float* x, size_t n, float* part) {
    __shared__ float red[4];
    float m = INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fminf(m, x[i]);
    m = -wave_max(-m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

__global__ void min_final_kernel(const float* part, int n, float* out) {
    float m = INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) m = fminf(m, part[i]);
    m = -wave_max(-m);
    if (threadIdx.x == 0) out[0] = m;
}

// One thread per row.  tok = first-max index (torch.argmax tie rule); finished rows emit PAD; rows that emit EOS become
// finished.  unfinished[step] counts rows still running AFTER this step so the host can cut the output where the
// reference's loop breaks (model.py:362-363) with a single sync at the end.
__global__ void greedy_select_kernel(const float* logits, int N, int V, uint8_t* finished, int64_t* ids, int ld_ids, int col,
                                     int32_t* tok_next, int pad, int eos, const float* neg_src, int m0, int m1, int m2,
                                     int* unfinished, int step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int live = 0;
    if (i < N) {
        const float* l = logits + (size_t)i * V;
        const float neg = neg_src ? -2.f * fabsf(neg_src[0]) : 0.f;
        float best = -INFINITY;
        int arg = 0;
        for (int v = 0; v < V; ++v) {
            float x = l[v];
            if (neg_src && (v == m0 || v == m1 || v == m2)) x = neg;
            if (x > best) {
                best = x;
                arg = v;
            }
        }
        const bool fin = finished[i] != 0;
        const int t = fin ? pad : arg;
        if (t == eos) finished[i] = 1;
        live = (fin || t == eos) ? 0 : 1;
        ids[(size_t)i * ld_ids + col] = t;
        tok_next[i] = t;
    }
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&unfinished[step], __popcll(b));
}

CPG_EXPORT int cpg_greedy_select(const float* logits, int N, int V, uint8_t* finished, int64_t* ids, int ld_ids, int col,
                                 int32_t* tok_next, int pad, int start, int eos, int prevent_empty, float* scratch,
                                 int* unfinished, int step, void* stream) {
    CPG_CHECK_ARG(logits && finished && ids && tok_next && unfinished && N > 0 && V > 0);
    hipStream_t s = (hipStream_t)stream;
    const float* neg = nullptr;
    if (prevent_empty) {
        // large_neg = -2*|min(logits)| over the whole batch (model.py:299-305)
        CPG_CHECK_ARG(scratch);
        const int nb = 256;
        hipLaunchKernelGGL(min_partial_kernel, dim3(nb), dim3(256), 0, s, logits, (size_t)N * V, scratch + 1);
        hipLaunchKernelGGL(min_final_kernel, dim3(1), dim3(64), 0, s, scratch + 1, nb, scratch);
        neg = scratch;
    }
    hipLaunchKernelGGL(greedy_select_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, logits, N, V, finished, ids, ld_ids, col,
                       tok_next, pad, eos, neg, pad, start, eos, unfinished, step);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ categorical
// torch.distributions.Categorical(logits = logits / temp).sample() of RNN_VAE.sample_G (models/model.py:308-309) with the
// uniform draw of every row passed in: token = first index whose cumulative softmax probability exceeds u (inverse CDF in
// index order - what ATen's multinomial does with its own uniform).  One thread per row; finished / <eos> / <pad> handling
// and the unfinished counter as in greedy_select_kernel.
__global__ void categorical_select_kernel(const float* logits, int N, int V, float inv_temp, const double* u, uint8_t* finished,
                                          int64_t* ids, int ld_ids, int col, int32_t* tok_next, int pad, int eos,
                                          const float* neg_src, int m0, int m1, int m2, int* unfinished, int step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int live = 0;
    if (i < N) {
        const float* l = logits + (size_t)i * V;
        const float neg = neg_src ? -2.f * fabsf(neg_src[0]) : 0.f;
        float mx = -INFINITY;
        for (int v = 0; v < V; ++v) {
            float x = l[v];
            if (neg_src && (v == m0 || v == m1 || v == m2)) x = neg;
            mx = fmaxf(mx, x * inv_temp);
        }
        double tot = 0.0;
        for (int v = 0; v < V; ++v) {
            float x = l[v];
            if (neg_src && (v == m0 || v == m1 || v == m2)) x = neg;
            tot += (double)expf(x * inv_temp - mx);
        }
        const double thr = u[i] * tot;
        double cum = 0.0;
        int arg = V - 1;
        for (int v = 0; v < V; ++v) {
            float x = l[v];
            if (neg_src && (v == m0 || v == m1 || v == m2)) x = neg;
            cum += (double)expf(x * inv_temp - mx);
            if (thr < cum) {
                arg = v;
                break;
            }
        }
        const bool fin = finished[i] != 0;
        const int t = fin ? pad : arg;
        if (t == eos) finished[i] = 1;
        live = (fin || t == eos) ? 0 : 1;
        ids[(size_t)i * ld_ids + col] = t;
        tok_next[i] = t;
    }
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&unfinished[step], __popcll(b));
}

CPG_EXPORT int cpg_categorical_select(const float* logits, int N, int V, float temp, const double* uniforms, uint8_t* finished,
                                      int64_t* ids, int ld_ids, int col, int32_t* tok_next, int pad, int start, int eos,
                                      int prevent_empty, float* scratch, int* unfinished, int step, void* stream) {
    CPG_CHECK_ARG(logits && uniforms && finished && ids && tok_next && unfinished && N > 0 && V > 0 && temp > 0.f);
    hipStream_t s = (hipStream_t)stream;
    const float* neg = nullptr;
    if (prevent_empty) {
        CPG_CHECK_ARG(scratch);
        const int nb = 256;
        hipLaunchKernelGGL(min_partial_kernel, dim3(nb), dim3(256), 0, s, logits, (size_t)N * V, scratch + 1);
        hipLaunchKernelGGL(min_final_kernel, dim3(1), dim3(64), 0, s, scratch + 1, nb, scratch);
        neg = scratch;
    }
    hipLaunchKernelGGL(categorical_select_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, logits, N, V, 1.f / temp, uniforms,
                       finished, ids, ld_ids, col, tok_next, pad, eos, neg, pad, start, eos, unfinished, step);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ beam
// One THREAD per sentence (the whole Beam.advance of a sentence is ~K*V = 120 candidates: a wave per sentence spends its
// time in cross-lane reductions with 40 of 64 lanes idle; a thread per sentence keeps the K-best list in registers and
// reads its K logits rows as contiguous 96-byte runs).  Rows are beam-major (row = k*N + i).
// Restates Beam.advance: BOS column := -1e20; first step uses beam 0 only; children of EOS-ended beams := -1e20;
// top-K over the flat K*V candidates (ties: lower flat index first); done when top beam is EOS and >= n_best finished.
#define CPG_MAX_BEAM 32   // widest beam any instantiation holds (KMAX = 8 for the usual widths, 32 above: static_eval.py's beam 15)
#define CPG_BEAM_VREG 32   // vocabularies up to this size are held in registers
template <bool VREG, int KMAX>
__global__ void beam_select_kernel(const float* __restrict__ logits, int N, int V, int K, int step, int n_best, int min_length,
                                   int bos, int eos, float* scores, int32_t* last_tok, int32_t* n_finished, uint8_t* done,
                                   int32_t* hist_tok, int32_t* hist_prev, float* hist_score, int32_t* origin,
                                   int32_t* tok_next, int* n_active) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = false;
    if (i < N) {
        if (done[i]) {  // not advanced: keeps its last tokens; reference re-applies the last origin (irrelevant once done)
            for (int k = 0; k < K; ++k) tok_next[(size_t)k * N + i] = last_tok[(size_t)i * K + k];
        } else {
            float bs[KMAX];
            int bc[KMAX];
#pragma unroll
            for (int p = 0; p < KMAX; ++p) {
                bs[p] = -INFINITY;
                bc[p] = 0;
            }
            float worst = -INFINITY;  // bs[K-1]
            const int kmax = step == 0 ? 1 : K;  // first step: only beam 0 is a candidate row
            for (int k = 0; k < kmax; ++k) {
                const float* l = logits + ((size_t)k * N + i) * V;
                float lv[CPG_BEAM_VREG];
                float m = -INFINITY;
                if (VREG) {
#pragma unroll
                    for (int v4 = 0; v4 < CPG_BEAM_VREG / 4; ++v4)
                        if (v4 * 4 < V) {
                            const float4 t = *reinterpret_cast<const float4*>(l + v4 * 4);
                            lv[v4 * 4 + 0] = t.x;
                            lv[v4 * 4 + 1] = t.y;
                            lv[v4 * 4 + 2] = t.z;
                            lv[v4 * 4 + 3] = t.w;
                        }
#pragma unroll
                    for (int v = 0; v < CPG_BEAM_VREG; ++v)
                        if (v < V) m = fmaxf(m, lv[v]);
                } else {
                    for (int v = 0; v < V; ++v) m = fmaxf(m, l[v]);
                }
                float se = 0.f;
                if (VREG) {
#pragma unroll
                    for (int v = 0; v < CPG_BEAM_VREG; ++v)
                        if (v < V) se += expf(lv[v] - m);
                } else {
                    for (int v = 0; v < V; ++v) se += expf(l[v] - m);
                }
                const float lse = m + logf(se);
                const bool parent_eos = (step > 0) && last_tok[(size_t)i * K + k] == eos;
                const float base = step > 0 ? scores[(size_t)i * K + k] : 0.f;
                auto consider = [&](int v, float x) {
                    float lp = x - lse;
                    if (step + 1 < min_length && v == eos) lp = -1e20f;
                    if (v == bos) lp = -1e20f;
                    float cs = step > 0 ? lp + base : lp;
                    if (parent_eos) cs = -1e20f;
                    if (!(cs > worst)) return;  // ties keep the earlier (lower flat index) candidate
                    int cc = k * V + v;
                    bool carried = false;
#pragma unroll
                    for (int p = 0; p < KMAX; ++p) {
                        if (p >= K) break;
                        if (carried ? (cs >= bs[p]) : (cs > bs[p])) {
                            const float fs = bs[p];
                            const int fc = bc[p];
                            bs[p] = cs;
                            bc[p] = cc;
                            cs = fs;
                            cc = fc;
                            carried = true;
                        }
                    }
#pragma unroll
                    for (int p = 0; p < KMAX; ++p)
                        if (p == K - 1) worst = bs[p];
                };
                if (VREG) {
#pragma unroll
                    for (int v = 0; v < CPG_BEAM_VREG; ++v)
                        if (v < V) consider(v, lv[v]);
                } else {
                    for (int v = 0; v < V; ++v) consider(v, l[v]);
                }
            }
            int nf = n_finished[i];
            int top = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k >= K) break;
                const int pk = bc[k] / V, tk = bc[k] - pk * V;
                if (k == 0) top = tk;
                scores[(size_t)i * K + k] = bs[k];
                last_tok[(size_t)i * K + k] = tk;
                origin[(size_t)i * K + k] = pk;
                const size_t h = ((size_t)step * N + i) * K + k;
                hist_tok[h] = tk;
                hist_prev[h] = pk;
                hist_score[h] = bs[k];
                tok_next[(size_t)k * N + i] = tk;
                if (tk == eos) ++nf;
            }
            n_finished[i] = nf;
            if (top == eos && nf >= n_best) done[i] = 1; else active = true;
        }
    }
    const unsigned long long mask = __ballot(active);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(n_active, __popcll(mask));
}

// h_out[k*N+i] = h_in[origin[i][k]*N + i]
__global__ void beam_reorder_kernel(const float* h_in, float* h_out, const int32_t* origin, int N, int K, int H) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)K * N * H) return;
    const int c = idx % H;
    const size_t r = idx / H;
    const int i = r % N, k = r / N;
    const int pk = origin[(size_t)i * K + k];
    h_out[idx] = h_in[((size_t)pk * N + i) * H + c];
}

CPG_EXPORT int cpg_beam_select(const float* logits, int N, int V, int K, int step, int n_best, int min_length, int bos, int eos,
                               float* scores, int32_t* last_tok, int32_t* n_finished, uint8_t* done, int32_t* hist_tok,
                               int32_t* hist_prev, float* hist_score, int32_t* origin, int32_t* tok_next, int* n_active,
                               const float* h_in, float* h_out, int H, void* stream) {
    CPG_CHECK_ARG(logits && scores && last_tok && n_finished && done && hist_tok && hist_prev && hist_score && origin);
    CPG_CHECK_ARG(N > 0 && V > 0 && K > 0 && K <= CPG_MAX_BEAM && tok_next && n_active && H >= 0);
    CPG_CHECK_ARG(H == 0 || (h_in && h_out && h_in != h_out));   // H = 0: no state is moved (the next plane step gathers through origin)
    hipStream_t s = (hipStream_t)stream;
    CPG_CHECK_ARG(K <= V);   // the first step ranks the V children of beam 0 only (topk(size) over V candidates, models/Beam.py:82-84)
#define CPG_BEAM_LAUNCH(VR, KM)                                                                                                      \
    hipLaunchKernelGGL((beam_select_kernel<VR, KM>), dim3(cdiv(N, 64)), dim3(64), 0, s, logits, N, V, K, step, n_best, min_length,   \
                       bos, eos, scores, last_tok, n_finished, done, hist_tok, hist_prev, hist_score, origin, tok_next,              \
                       n_active + step)
    const bool vreg = V <= CPG_BEAM_VREG && V % 4 == 0;
    if (K <= 8) {
        if (vreg) CPG_BEAM_LAUNCH(true, 8); else CPG_BEAM_LAUNCH(false, 8);
    } else {
        if (vreg) CPG_BEAM_LAUNCH(true, CPG_MAX_BEAM); else CPG_BEAM_LAUNCH(false, CPG_MAX_BEAM);
    }
#undef CPG_BEAM_LAUNCH
    CPG_LAUNCH_CHECK();
    if (H == 0) return 0;
    const size_t n = (size_t)K * N * H;
    hipLaunchKernelGGL(beam_reorder_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h_in, h_out, origin, N, K, H);
    CPG_LAUNCH_CHECK();
    return 0;
}

// h_out[k*N+i] = h_in[origin[i][k]*N + i] for one more [K*N,H] state array (the LSTM cell state: cpg_beam_select reorders h)
CPG_EXPORT int cpg_beam_reorder(const float* h_in, float* h_out, const int32_t* origin, int N, int K, int H, void* stream) {
    CPG_CHECK_ARG(h_in && h_out && origin && h_in != h_out && N > 0 && K > 0 && H > 0);
    const size_t n = (size_t)K * N * H;
    hipLaunchKernelGGL(beam_reorder_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h_in, h_out, origin,
                       N, K, H);
    CPG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ beam hypotheses
// Beam.sort_finished + Beam.get_hyp (models/Beam.py:110-132) for every sentence, one thread each, from the recorded
// (token, back-pointer, score) history [T,N,K].  Finished entries are ranked in insertion order (step asc, beam asc) by
// raw summed log-prob, ties keep the earlier entry (python's stable sort); when fewer than n_best finished the live beam's
// first (n_best - n_finished) entries of the last advanced step are appended.  hyps[i][b] = <start> + tokens, -1 padded.
template <int KMAX>
__global__ void beam_hyp_kernel(const int32_t* __restrict__ tok, const int32_t* __restrict__ prev,
                                const float* __restrict__ score, int T, int N, int K, int n_best, int eos, int start,
                                int32_t* __restrict__ hyps, int32_t* __restrict__ lens, float* __restrict__ out_sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float bs[KMAX];
    int bt[KMAX], bk[KMAX];
#pragma unroll
    for (int p = 0; p < KMAX; ++p) {
        bs[p] = -INFINITY;
        bt[p] = 0;
        bk[p] = 0;
    }
    auto insert = [&](float cs, int ct, int ck) {
        bool carried = false;  // a displaced entry is older than everything below it: it wins ties on the way down
#pragma unroll
        for (int p = 0; p < KMAX; ++p) {
            if (p >= n_best) break;
            if (carried ? (cs >= bs[p]) : (cs > bs[p])) {
                const float fs = bs[p];
                const int ft = bt[p], fk = bk[p];
                bs[p] = cs;
                bt[p] = ct;
                bk[p] = ck;
                cs = fs;
                ct = ft;
                ck = fk;
                carried = true;
            }
        }
    };
    int Ti = 0, nfin = 0;
    for (int t = 0; t < T; ++t) {
        const size_t b = ((size_t)t * N + i) * K;
        if (tok[b] >= 0) ++Ti;
        for (int k = 0; k < K; ++k)
            if (tok[b + k] == eos) {
                ++nfin;
                insert(score[b + k], t + 1, k);
            }
    }
    const int need = min(max(n_best - nfin, 0), n_best);
    const int last = min(max(Ti - 1, 0), T - 1);
    for (int j = 0; j < need; ++j) insert(score[((size_t)last * N + i) * K + j], Ti, j);
    const int L = T + 1;
#pragma unroll
    for (int p = 0; p < KMAX; ++p) {
        if (p >= n_best) break;
        int32_t* h = hyps + ((size_t)i * n_best + p) * L;
        const int tl = bt[p];
        int cur = bk[p];
        h[0] = start;
        for (int j = T - 1; j >= 0; --j) {
            if (j < tl) {
                const size_t b = ((size_t)j * N + i) * K + cur;
                h[j + 1] = tok[b];
                cur = prev[b];
            } else {
                h[j + 1] = -1;
            }
        }
        lens[(size_t)i * n_best + p] = tl + 1;
        out_sc[(size_t)i * n_best + p] = bs[p];
    }
}

CPG_EXPORT int cpg_beam_hypotheses(const int32_t* hist_tok, const int32_t* hist_prev, const float* hist_score, int T, int N,
                                   int K, int n_best, int eos, int start, int32_t* hyps, int32_t* lens, float* scores,
                                   void* stream) {
    CPG_CHECK_ARG(hist_tok && hist_prev && hist_score && hyps && lens && scores);
    CPG_CHECK_ARG(T > 0 && N > 0 && K > 0 && K <= CPG_MAX_BEAM && n_best > 0 && n_best <= K);
    if (n_best <= 8)
        hipLaunchKernelGGL(beam_hyp_kernel<8>, dim3(cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream, hist_tok, hist_prev, hist_score,
                           T, N, K, n_best, eos, start, hyps, lens, scores);
    else
        hipLaunchKernelGGL(beam_hyp_kernel<CPG_MAX_BEAM>, dim3(cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream, hist_tok, hist_prev,
                           hist_score, T, N, K, n_best, eos, start, hyps, lens, scores);
    CPG_LAUNCH_CHECK();
    return 0;
}
