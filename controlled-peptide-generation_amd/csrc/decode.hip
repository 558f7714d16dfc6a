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

CPG_EXPORT size_t cpg_vocab_fc_bwd_workspace(int R, int H, int V) {
    size_t a = cpg_gemm_tn_workspace(R, V, H), b = cpg_colsum_workspace(R, V);
    return (a > b ? a : b) + 256;
}

// dhs[R,H] = (dlogits W) .* keep*scale ; dw[V,H] (+)= dlogits^T (hs .* keep*scale) ; db[V] (+)= colsum(dlogits)
CPG_EXPORT int cpg_vocab_fc_bwd(const float* dlogits, const float* hs, const uint8_t* keep, float scale, const float* w,
                                float* dhs, float* dw, float* db, int R, int H, int V, int accumulate, void* workspace,
                                size_t workspace_bytes, void* stream) {
    CPG_CHECK_ARG(dlogits && hs && w && R > 0 && H > 0 && V > 0);
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
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
