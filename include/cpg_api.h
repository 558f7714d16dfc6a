/* libcpg_hip.so - C ABI of the MI355X (gfx950) peptide WAE-training + CLaSS-sampling hot path.
 *
 * The reference (IBM/controlled-peptide-generation) is pure Python on PyTorch and has NO native/FFI boundary of
 * its own; the operations below are the ones PyTorch dispatched for it (cuDNN GRU, cuBLAS, fused
 * cross-entropy, elementwise kernels) plus the numpy/sklearn arithmetic of the CLaSS sampler.  Each entry point cites
 * the reference call site it stands behind.  The Python host above this ABI (package
 * `controlled-peptide-generation_amd/`) mirrors the reference's own API (models.model.RNN_VAE, losses.*, train_vae,
 * density_modeling.RejSampleBase) and binds these symbols with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t (>0) or a negative cpg code otherwise;
 *     `cpg_last_error()` returns a thread-local message;
 *   - the caller owns every buffer (device pointers, e.g. torch tensor.data_ptr()); nothing is allocated inside;
 *     scratch comes in through `workspace` arguments sized by the matching `*_workspace()` query;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; 0 = default stream);
 *   - dense arrays are row-major f32 unless stated; `ld*` = leading dimension in elements;
 *   - all randomness is an input (or comes from the explicit cpg_rng_* counter-based streams);
 *   - process-wide state is exactly two things, both set through this ABI: the compute mode (cpg_set_compute_mode) and
 *     the option table of launch-policy knobs (cpg_set_option; initialised ONCE from CPG_<NAME> environment variables -
 *     no launch path reads the environment); everything else is re-entrant across streams; one process per GPU;
 *   - cpg_version() changes whenever a signature below changes: a binding must refuse a library whose version differs.
 *   - reductions use a fixed two-stage partition: results are run-to-run deterministic.
 */
#ifndef CPG_API_H
#define CPG_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPG_API

/* ---- library ---------------------------------------------------------------------------------------------- */
CPG_API const char* cpg_last_error(void);
CPG_API int cpg_version(void);
CPG_API int cpg_device_count(void);
/* Launch-policy options ("gru_persist", "gru_fwd_bm", "gru_bwd_dl", "gru_bwd_tile", "gru_bwd_dl2", "gru_bwd_stagger",
 * "lstm_persist", "lstm_bwd_dl", "tn_tile", "tn_split", "gemm_tile", "dgi_mode", "mmd_dl", "bf16_store", "bf16_dg"; KNOBS.md).  value: a decimal
 * number or a short token such as "64x32"; null or "" returns the option to the built-in policy.  Unknown name: -2. */
CPG_API int cpg_set_option(const char* name, const char* value);
/* copies the option's text ("" when unset) into buf; returns 1 when set, 0 when unset, -2 for an unknown name */
CPG_API int cpg_get_option(const char* name, char* buf, int n);

/* ---- layout ----------------------------------------------------------------------------------------------- */
/* ids int64 [B,T] (loader layout, data_processing/dataset.py:242-244) -> tok int32 [T,B] time-major; applies
 * WordDropout (models/decoder.py:117-133: masked positions := <unk>) when wd_mask (uint8 [B,T]) is non-null. */
CPG_API int cpg_tokens_prepare(const int64_t* ids, const uint8_t* wd_mask, int B, int T, int unk, int32_t* tok,
                               void* stream);
/* dst[d1][d0][inner] = src[d0][d1][inner]  (time-major <-> batch-major views of [T,B,*]) */
CPG_API int cpg_transpose01_f32(const float* src, int d0, int d1, int inner, float* dst, void* stream);
CPG_API int cpg_transpose01_u8(const uint8_t* src, int d0, int d1, int inner, uint8_t* dst, void* stream);

/* ---- dense products (nn.Linear / `@`): models/encoder.py:35-36,50-51; models/decoder.py:43-45; losses.py:85 ------ */
/* Y[M,N] (+)= X[M,K] W[N,K]^T + bias[N] */
CPG_API int cpg_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                           int M, int N, int K, int accumulate, void* stream);
/* Exponent record of a weight matrix W [rows, cols] (row stride ld): cpg_weight_exp_bytes() bytes of device memory holding partial maxima
 * of |W|, from which every kernel that splits W into f16 pairs derives the SAME power of two 2^e_w (max|W| 2^e_w in [2^13, 2^14)) - any
 * finite f32 weight has a pair image (rounds 4-5 scaled weights by a fixed 2^8: |w| >= 256 became an f16 infinity).  One launch.  The
 * entry points that build a weight image themselves (BPTT chains, plane products) do this internally; the ones that split W on the fly
 * take the record as `wx` and run their bf16x3 / exact-f32 engine without it. */
CPG_API size_t cpg_weight_exp_bytes(void);
CPG_API int cpg_weight_exp(const float* w, int rows, int cols, int ld, void* wx, void* stream);
/* one record for TWO matrices (their joint largest magnitude): operands chained into the same accumulators */
CPG_API int cpg_weight_exp2(const float* w1, int rows1, int cols1, int ld1, const float* w2, int rows2, int cols2, int ld2, void* wx,
                            void* stream);
/* cpg_linear_fwd for an input whose magnitudes are O(1) - recurrent states (|x| < 65504; absolute precision 2^-25 below 2^-14): large
 * products run on f16 pairs (three f16 MFMAs per block), same results within f32 rounding.  The caller vouches for the INPUT's range;
 * the weights' range is covered by wx = cpg_weight_exp(W) (null: exact-f32 engine). */
CPG_API int cpg_linear_fwd_pairs(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int M,
                                 int N, int K, int accumulate, const void* wx, void* stream);
/* ---- grouped small products (round 6): up to 6 independent nn.Linear-shaped problems of ONE form in one launch (a flat grid over all
 * their tiles) - the token tables, the two encoder heads (models/encoder.py:35-36,50-51) and their gradients each fill a fraction of the
 * chip when launched alone.  A problem may chain two (A, B, K) segments into the same accumulators (an input that is the concatenation
 * of two tensors needs no cat; dX = dY1 W1 + dY2 W2 is one product) and send result columns >= n_split to a second destination (the
 * gradient of a concatenation needs no slicing copies).  form: 0 = NT  C[M,N] (+)= sum_s A_s[M,K_s] B_s[N,K_s]^T (+ bias);
 * 1 = NN  C (+)= sum_s A_s[M,K_s] B_s[K_s,N] (+ bias);  2 = TN  C (+)= sum_s A_s[K_s,M]^T B_s[K_s,N] (no bias).  Same engines and
 * sums as cpg_linear_fwd / _bwd_input / _bwd_weight - except form 0 problems flagged `pairs` (all of a group), which run the direct-to-LDS
 * loop on f16 pairs (22-bit operands, f32 accumulation: f32-grade like the recurrent products).  probs: nprob records of
 * cpg_gemm_group_prob_bytes() bytes each. */
typedef struct CpgGemmProb {
    const float* A[2];
    const float* B[2];
    int lda[2], ldb[2], K[2];   /* K[1] = 0: one segment */
    int M, N;
    float* C;
    float* C2;                  /* optional: columns >= n_split are written to C2[:, col - n_split] */
    int ldc, ldc2, n_split;
    int accumulate;
    const float* bias;          /* [N], forms 0 / 1, or null */
    const void* wx_a;           /* form 0 with pairs = 1: exponent records (cpg_weight_exp over the operand) of A / B, or null (2^0: an */
    const void* wx_b;           /* operand of O(1) magnitudes) */
    int pairs;                  /* 1 (form 0, rows 16-byte aligned, K multiples of 32): f32-grade products on f16 pairs - three f16 MFMAs */
    int pad_;                   /* per block on operands split in registers, each times the power of two of its record */
} CpgGemmProb;
CPG_API int cpg_gemm_group_prob_bytes(void);
CPG_API int cpg_gemm_group(int form, int nprob, const void* probs, void* stream);
/* Backward of n <= 4 token tables tab_i = emb W_i^T + b_i (V <= 32 rows, E <= 256), two launches: dW_i (+)= dtab_i^T emb (the [G, E]
 * column block of W_ih's gradient, row stride lddw[i]), db_i [G] (+)= column sums of dtab_i, demb [V, lde] (+)= sum_i dtab_i W_i with row
 * skip_row untouched (accumulating) / zero: nn.Embedding(padding_idx), models/model.py:47.  dtab / W / dW / db: HOST arrays of n device
 * pointers (dW / db entries and demb may be null).  workspace: cpg_token_tables_bwd_workspace bytes. */
CPG_API size_t cpg_token_tables_bwd_workspace(int n, int V, int G, int E);
CPG_API int cpg_token_tables_bwd(int n, int V, int G, int E, const void* const* dtab, const void* const* W, const int* ldw,
                                 const float* emb, int lde_in, void* const* dW, const int* lddw, int accumulate_w, void* const* db,
                                 int accumulate_db, float* demb, int lde, int accumulate_emb, int skip_row, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* out_i[N] (+)= column sums of X_i [M, N] (row stride ld[i]) for nmat <= 4 matrices in one single-stage launch (X / ld / out: HOST arrays) */
CPG_API int cpg_colsum_multi(int nmat, const void* const* X, const int* ld, int M, int N, void* const* out, int accumulate, void* stream);
/* dst[c][r] = src[r][c] for r < R, c < C; zeros for R <= r < Rpad (dst [C, ldd >= Rpad]): the k-rows operand of an NN / TN product as
 * K-contiguous rows padded to whole 32-deep slabs, for cpg_gemm_group's direct-to-LDS NT form. */
CPG_API int cpg_transpose_pad(const float* src, int lds, int R, int C, float* dst, int ldd, int Rpad, void* stream);
/* dX[M,K] (+)= dY[M,N] W[N,K] */
CPG_API int cpg_linear_bwd_input(const float* dY, int lddy, const float* W, int ldw, float* dX, int lddx, int M, int N,
                                 int K, int accumulate, void* stream);
CPG_API size_t cpg_linear_bwd_weight_workspace(int M, int N, int K);
/* dW[N,K] (+)= dY[M,N]^T X[M,K] ; db[N] (+)= column sums of dY (db may be null) */
CPG_API int cpg_linear_bwd_weight(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw, float* db,
                                  int M, int N, int K, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream);
/* nn.Dropout in front of nn.Linear - the inter-layer dropout of nn.GRU(dropout=p_dropout) at models/encoder.py:25-30 (train
 * mode, every layer's output but the last) feeding the next layer's W_ih product.  keep: uint8 0/1, indexed like the f32 array
 * it multiplies (X forward / weight gradient, dX for the input gradient); the mask itself is an input (or a cpg_rng_bernoulli_u8 draw).
 *   Y[M,N] (+)= (X .* keep*scale)[M,K] W[N,K]^T + bias[N] ;  dX = (dY W) .* keep*scale ;  dW (+)= dY^T (X .* keep*scale) */
CPG_API int cpg_linear_masked_fwd(const float* X, int ldx, const uint8_t* keep, float scale, const float* W, int ldw,
                                  const float* bias, float* Y, int ldy, int M, int N, int K, int accumulate, void* stream);
CPG_API int cpg_linear_masked_bwd_input(const float* dY, int lddy, const float* W, int ldw, const uint8_t* keep, float scale,
                                        float* dX, int lddx, int M, int N, int K, void* stream);
CPG_API int cpg_linear_masked_bwd_weight(const float* dY, int lddy, const float* X, int ldx, const uint8_t* keep, float scale,
                                         float* dW, int lddw, float* db, int M, int N, int K, int accumulate, void* workspace,
                                         size_t workspace_bytes, void* stream);
CPG_API size_t cpg_colsum_workspace_bytes(int M, int N);
/* out[N] (+)= column sums of X[M,N] */
CPG_API int cpg_colsum_f32(const float* X, int ld, int M, int N, float* out, int accumulate, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Y[M,N] (+)= X[M,K] B[K,N] */
CPG_API int cpg_matmul_nn(const float* X, int ldx, const float* Bm, int ldb, float* Y, int ldy, int M, int N, int K,
                          int accumulate, void* stream);

/* Compute mode of the recurrent products (step products and dW_hh): 0 = f32-grade (default, the parity path), 1 = bf16
 * (BASELINE.json configs[1]/[4]: operands rounded to bf16 when staged, one bf16 MFMA per block, f32 accumulation, f32
 * storage and f32 master weights).  Process-wide; read by the launchers at call time. */
CPG_API int cpg_set_compute_mode(int mode);
CPG_API int cpg_get_compute_mode(void);

/* ---- GRU (torch.nn.GRU as used at models/encoder.py:25-30,42 and models/decoder.py:40-41,77,98) ---------------------
 * Gate row order r,z,n.  The input-side pre-activation of step t, row b is the SUM of the non-null sources
 *     tab[tok[t,b], :]   token table  emb @ W_ih[:, :E]^T + b_ih           ([V,3H]; tok int32 [T,B])
 *     rowc[b, :]         constant over time, e.g. [z;c] @ W_ih[:, E:]^T    ([B,3H])
 *     dense[t,b,:]       arbitrary per-step term (upper encoder layers)    ([T,B,3H])
 * State slab hs [(T+1),B,H]: forward direction hs[0]=h0 (caller fills), h_t -> hs[t+1];
 *                            reverse direction hs[T]=h0 (caller fills), h_t -> hs[t].
 * gates [T,4,B,H] receives r,z,n and (W_hn h + b_hn) per step for the backward pass (null for inference): f32 elements, or
 * bf16 elements laid out [T,B,H,4] (the four values of an element adjacent; a buffer of half the bytes behind the same pointer
 * type) when cpg_gru_gates_bf16(B, H, ragged) answers 1 - the bf16 compute mode on dense batches the direct-to-LDS backward step
 * covers.  Ask once per sequence, allocate accordingly,
 * and keep compute mode / options unchanged until its backward pass has been enqueued (which refuses a mismatch it can see).
 * Batch rows are independent recurrences: a call covers rows [row_begin,row_end) of the B-row problem (0,B for all). */
/* step_rows (optional, DEVICE int32 [T], may be null): only rows < step_rows[t] are live at time t.  For length-sorted
 * teacher-forced batches: once all remaining targets of a row are <pad> (losses.py:27 ignores them) its state is never
 * needed again, so the tail of the batch drops out step by step.  The counts are read by the kernels (no host sync);
 * state / gate slots of dead (t,row) pairs are left untouched - hand in zeroed slabs if they are read elsewhere. */
CPG_API int cpg_gru_gates_bf16(int B, int H, int ragged /* step_rows given */);
/* bf16 GRADIENT storage (bf16 compute mode only): 1 when the gate gradients dG [T,B,4H] of a sequence are bf16 elements as well - a
 * buffer of half the bytes behind the same pointer type - i.e. where cpg_gru_gates_bf16 answers 1, H % 128 == 0, B % 128 == 0 and
 * the sequence has a token table of 0 < V <= 31 rows (every consumer of dG then has a bf16 form: the next BPTT step's operand, the
 * dW_hh product, the one-pass input-side reduction).  Option bf16_dg = 0 keeps f32.  The caller asks once per sequence, sizes dG
 * accordingly and passes the answer as `dg_bf16` to cpg_gru_seq_bwd / cpg_gru_biseq_bwd / cpg_gru_wgrad_hh / cpg_gru_dgi_reduce
 * (which with bf16 gradients must be given the token table: db_hh comes out of its column sums, cpg_gru_wgrad_hh's db_hh = null). */
CPG_API int cpg_gru_dg_bf16(int B, int H, int ragged, int V);
CPG_API int cpg_gru_seq_fwd(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh, const int32_t* tok,
                            const float* tab, const float* rowc, const float* dense, float* hs, float* gates,
                            int row_begin, int row_end, const int32_t* step_rows,
                            const void* wx /* cpg_weight_exp record of w_hh [3H,H] (f32-grade mode: the step kernel's f16-pair engine takes the
                            weights' power of two from it), or null: the bf16x3 engine, which needs no range guard */, void* stream);
/* one decode step = GRUDecoder.forward_sample's recurrent part (models/decoder.py:86-99); wx as above */
CPG_API int cpg_gru_step_fwd(int B, int H, const float* w_hh, const float* b_hh, const int32_t* tok, const float* tab,
                             const float* rowc, const float* h_prev, float* h_out, const void* wx, void* stream);
/* BPTT.  dhs_ext [T,B,H]: gradient arriving at each step's output (time-aligned; may be null);
 * dh_last [B,H]: gradient on the final state (may be null); dG out [T,B,4H] = (dr_pre, dz_pre, d(W_hn h+b_hn), dn_pre):
 * columns 0..3H are the hidden-side gate gradients, columns {0..2H, 3H..4H} the input-side ones;
 * dH_scratch [2,B,H]; dh0 [B,H] gradient of the initial state (null to skip).
 * w_hhT_scratch [H,3H] (optional): receives W_hh^T once per call for the direct-to-LDS step kernel (both operands
 * K-contiguous; full 32 x 32 tiles of dense batches); null keeps the register-staged kernel for every shape.
 * pair_scratch (optional; with w_hhT_scratch): cpg_gru_bwd_pair_bytes(row_end - row_begin, H, 1) bytes for the f16-pair form of
 * that step (f32-grade mode, 64-row tiles): every launch then also writes the three recurrent blocks of its dG as f16 pairs
 * times a power of two per 32 x 32 group - the next launch's operand, three f16 MFMAs per block in place of eight f32 ones -
 * and w_hhT_scratch receives W_hh^T in the same form, times a power of two chosen from its largest magnitude (one more small launch per
 * sequence; any finite weight is covered).  null (or a query answer of 0): the exact-f32 product. */
CPG_API size_t cpg_gru_bwd_pair_bytes(int rows, int H, int ndir /* 1 | 2: directions per launch */);
CPG_API int cpg_gru_seq_bwd(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                            const float* dhs_ext, const float* dh_last, float* dG, float* dH_scratch, float* dh0,
                            int row_begin, int row_end, const int32_t* step_rows /* as in cpg_gru_seq_fwd; dG rows of dead
                            (t,row) pairs are not written: pass a zeroed dG */, float* w_hhT_scratch, void* pair_scratch, int dg_bf16,
                            void* stream);
/* Both directions of one biGRU layer in lock step, ONE launch per step for the pair (launch p: time p forward, time
 * T-1-p reverse).  Arguments as in cpg_gru_seq_fwd / _bwd per direction (_f forward, _r reverse); no initial-state
 * gradient (the encoder starts from h0 = 0). */
CPG_API int cpg_gru_biseq_fwd(int T, int B, int H, const float* w_hh_f, const float* b_hh_f, const float* w_hh_r,
                              const float* b_hh_r, const int32_t* tok, const float* tab_f, const float* tab_r,
                              const float* dense_f, const float* dense_r, float* hs_f, float* hs_r, float* gates_f,
                              float* gates_r, const void* wx_f, const void* wx_r /* as cpg_gru_seq_fwd's wx; both or the bf16x3 engine */,
                              void* stream);
CPG_API int cpg_gru_biseq_bwd(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                              const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                              const float* dhs_ext_r, const float* dh_last_f /* [B,H] gradient on the final state of the
                              direction, or null */, const float* dh_last_r, float* dG_f, float* dG_r, float* scratch_f,
                              float* scratch_r, float* w_hhT_scratch_f,
                              float* w_hhT_scratch_r /* as in cpg_gru_seq_bwd; both or neither */,
                              void* pair_scratch_f, void* pair_scratch_r /* cpg_gru_bwd_pair_bytes(B, H, 2) bytes each, or null */,
                              int dg_bf16, void* stream);
/* Persistent form: the WHOLE time loop of one direction in ONE launch (csrc/gru_persist.hip): each workgroup keeps the
 * W_hh rows of 16 hidden units in LDS (already split into bf16 planes) for 256 batch rows and the column-tile workgroups of
 * a row tile hand h_t to each other through the state slab (write-through stores + arrival counters), so nothing is
 * re-staged, re-launched or re-gathered per step.  Same arguments and results as cpg_gru_seq_fwd over all rows.
 * cpg_gru_persistent_fits: 1 when (B,H) is covered on this device (H % 32 == 0, the plane slice fits the LDS, every
 * workgroup co-resident by hipOccupancyMaxActiveBlocksPerMultiprocessor's count; option gru_persist = 0 disables).
 * sync_scratch: cpg_gru_persistent_scratch_bytes(T,B,H) bytes of device memory, zeroed by the caller once (arrival counters,
 * which every launch leaves at zero again, a sticky error word at byte cpg_gru_persistent_err_offset(B), the bf16-plane exchange slots).
 * A wait that times out (workgroups not co-resident: another process or kernel holds CUs) sets the error word - and
 * *err_host, so the host notices without a copy or a synchronisation - and NaN-poisons everything the wave stores afterwards.
 * cpg_gru_persistent_status synchronises the stream and returns the error word.
 * Do not run two persistent launches concurrently on different streams: each needs all of its workgroups resident.
 * Hand-off flavour (f32-grade mode): every workgroup posts the XCD it runs on (HW_REG_XCC_ID); a row tile whose producers all sit
 * on one XCD publishes its planes with plain stores from the first step on (the lines stay in that L2), any other placement keeps
 * the write-through stores.  Decided per launch from what the hardware reports, never from blockIdx; the word at byte
 * cpg_gru_persistent_path_offset(B) of the scratch records the last such launch's choice (1 same-XCD, 2 write-through). */
CPG_API int cpg_gru_persistent_fits(int B, int H);
/* batch rows one launch covers at width H on this device (0: width not covered; 16 hidden units per workgroup up to H = 512,
 * 8 up to H = 1024); wider batches run as consecutive launches over row ranges [row_begin, row_end) */
CPG_API int cpg_gru_persistent_rows(int H);
CPG_API size_t cpg_gru_persistent_scratch_bytes(int T, int B, int H);
CPG_API size_t cpg_gru_persistent_err_offset(int B);
CPG_API size_t cpg_gru_persistent_path_offset(int B);
CPG_API int cpg_gru_seq_fwd_persistent(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                                       const int32_t* tok, const float* tab, const float* rowc, const float* dense,
                                       float* hs, float* gates, int row_begin, int row_end, void* sync_scratch,
                                       void* err_host /* pinned, host-mapped uint32 that a timed-out wave also sets; may be null */,
                                       void* stream);
CPG_API int cpg_gru_persistent_kernel_name(int H, char* buf, int n);
/* sigmoid / tanh exactly as the persistent sequence kernels (GRU and LSTM) evaluate them - hardware exp2 / rcp forms, <= 2 / 3 ulp
 * (csrc/cpg_common.h) where torch.nn.GRU's cell (models/encoder.py:25-30, models/decoder.py:40-41) calls the library forms - for n
 * elements: tests measure the units in the last place against float64 */
CPG_API int cpg_persistent_cell_probe(const float* x, float* sigmoid_out, float* tanh_out, int n, void* stream);
CPG_API int cpg_gru_persistent_status(int B, const void* sync_scratch, void* stream);
/* Launcher introspection (bench.py labels its roofline object with these instead of literals): the kernel a step launch /
 * a dW = dY^T X product would run, named as rocprofv3 prints it (no "void ", no argument list); returns the length.
 * kind 0 forward step, 1 backward step; ndir 1 | 2 (paired biGRU launches); have_wt: W_hh^T handed to the backward. */
CPG_API int cpg_gru_step_kernel_name(int kind, int B, int H, int ndir, int have_wt, char* buf, int n);
CPG_API int cpg_gru_step_kernel_is_split(int kind, int B, int H, int ndir, int have_wt);
CPG_API int cpg_gemm_tn_kernel_name(int Mr, int N, int Kd, int dy_pairs /* the f16-pair form cpg_gru_wgrad_hh runs with a pair scratch */,
                                    char* buf, int n);
CPG_API int cpg_gemm_tn_split(int Mr, int N, int Kd, int dy_pairs);
CPG_API size_t cpg_gru_wgrad_workspace(int T, int B, int H, int V);
/* dw_hh[3H,H] (+)= sum_t dgh_t^T h_{prev(t)} ; db_hh[3H] (+)= sum dgh (db_hh may be null).
 * pair_scratch (optional): the scratch the sequence's cpg_gru_seq_bwd / cpg_gru_biseq_bwd call received (same B and H, enqueued
 * before this call) - the product then runs on f16 pairs with the column exponents those launches recorded. */
CPG_API int cpg_gru_wgrad_hh(int T, int B, int H, int reverse, const float* dG, const float* hs, float* dw_hh,
                             float* db_hh, int accumulate, void* workspace, size_t workspace_bytes, const void* pair_scratch,
                             int dg_bf16, void* stream);
/* dtab[V,3H] (+)= sum of input-side gate gradients grouped by token ; dsum[4H] (+)= column sums of dG (dsum[0:3H] is the
 * b_hh gradient) ; drowc[B,3H] (+)= sum over time (any may be null).  dtab and dsum come from ONE pass over dG:
 * dG^T . [onehot(tok) | 1] on the matrix cores. */
CPG_API int cpg_gru_dgi_reduce(int T, int B, int H, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                               float* drowc, int accumulate, void* workspace, size_t workspace_bytes, int dg_bf16, void* stream);

/* ---- All-T planes form of the f16-pair BPTT chain (round 5; csrc/pair_engine.h ApScratch, csrc/pair_tn.h).  Same results as the
 * calls above (nn.GRU's backward, models/encoder.py:25-30,42 / models/decoder.py:40-41,77 under train_vae.py:39), different storage:
 * the three recurrent gate-gradient blocks (dr_pre, dz_pre, d(W_hn h + b_hn)) of EVERY step are kept ONLY as the f16-pair plane
 * images the next backward launch reads anyway ([T][B][6H] f16 + one power-of-two exponent per 32 x 32 group, [T][B/32][H/32]) - no
 * f32 copy of them is written - and each step also leaves h_prev as unscaled f16-pair planes ([T][B][2H]).  `ap` holds all of it:
 * cpg_gru_ap_bytes(T, B, H, ndir) bytes per direction, 16-byte aligned (0 = not covered: f32-grade mode, whole dense batches,
 * H % 128 == 0, B % 128 == 0, the f16-pair step available; option gru_ap = 0 answers 0).  dN [T,B,H] receives the one block that
 * is not recurrent: dn_pre, the n-gate's input-side gradient.  Consumers:
 *   cpg_gru_wgrad_hh_ap    dW_hh (+)= planes^T x state planes: LDS-DMA operands, transposing LDS reads, three f16 MFMAs per block,
 *                          NO conversion in the loop; segments are brought to their column group's smallest exponent on the way;
 *   cpg_gru_dgi_reduce_ap  token-table gradient, column sums (dsum[4H] as cpg_gru_dgi_reduce: [:3H] = db_hh) and sums over time
 *                          from the planes (widened exactly: (hi + lo) 2^-e) + dN, one pass on the matrix cores (V <= 31 rows).
 * Every value that reaches a gradient is the f16 pair's 22-bit form of the f32 value (2^-22 relative to the largest magnitude of its
 * 32 x 32 group), as in the f16-pair backward step itself.
 * bf16 compute mode (cpg_set_compute_mode(1), where cpg_gru_dg_bf16 answers 1): the same entry points keep the mode's storage - `dN` of
 * the _bwd_ap calls IS the bf16 gate-gradient buffer [T,B,4H] of cpg_gru_seq_bwd(dg_bf16 = 1), cpg_gru_dgi_reduce(dg_bf16 = 1) reads it
 * as before - and `ap` (cpg_gru_ap_bytes = T B H x 2 bytes) receives h_prev of every step rounded to bf16, so that
 * cpg_gru_wgrad_hh_ap(ap, dG_bf16) runs the same conversion-free loop on ONE bf16 plane per operand (one MFMA per block). */
CPG_API size_t cpg_gru_ap_bytes(int T, int B, int H, int ndir /* 1 | 2: directions per launch */);
CPG_API int cpg_gru_seq_bwd_ap(int T, int B, int H, int reverse, const float* w_hh, const float* hs, const float* gates,
                               const float* dhs_ext, const float* dh_last, float* dN, float* dH_scratch, float* dh0,
                               float* w_hhT_scratch, void* ap, void* stream);
CPG_API int cpg_gru_biseq_bwd_ap(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* hs_f,
                                 const float* hs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                 const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dN_f,
                                 float* dN_r, float* scratch_f, float* scratch_r, float* w_hhT_scratch_f,
                                 float* w_hhT_scratch_r, void* ap_f, void* ap_r, void* stream);
CPG_API int cpg_gru_wgrad_hh_ap(int T, int B, int H, const void* ap, const void* dG_bf16 /* bf16 compute mode only, else null */,
                                float* dw_hh, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
CPG_API int cpg_gru_dgi_reduce_ap(int T, int B, int H, const void* ap, const float* dN, const int32_t* tok, int V, float* dtab,
                                  float* dsum, float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* ---- nn.Linear-shaped products over many rows on f16-pair plane images (round 5; csrc/planes.hip): the input projection of an upper
 * encoder layer (models/encoder.py:25-30: nn.GRU(num_layers > 1): layer l reads the concatenated outputs of layer l-1) and its two
 * gradients - at BASELINE.json configs[4] dimensions 27 of the step's 58 ms before.  Every operand is turned into an image ONCE by a
 * bandwidth-bound pass (a row = 128-byte segments of [32 high halves | 32 low halves] f16: x = hi + lo, 22 significand bits) and the
 * products run conversion-free: LDS-DMA operands, three f16 MFMAs per block, f32 accumulation - f32-grade like the f16-pair recurrence.
 *   cpg_planes_ok(R, K, N)     1 where the forms cover a product of R rows, contraction K, N outputs (f32-grade mode, multiples of 128)
 *   cpg_pair_rows              x = [x1 | x2] (x2 optional) f32 -> image [R][2 (C1 + C2)], unscaled (|x| <= 65504; states: |x| <= 1)
 *   cpg_linear_fwd_planes      Y [R, N] (+)= x W^T + bias from x's image; scratch = cpg_weight_image_bytes(N, K) bytes (W's image x 2^e_w + its exponent record)
 *   cpg_grad_planes            gate gradients dG [R, ldg] (G = 3 | 4 blocks of H columns at column offsets off[]) -> `gp`
 *                              (cpg_grad_planes_bytes): image [R][2 G H] in the order (32-unit group, block) times ONE power of two per
 *                              (32 rows x group), the exponent table, the smallest exponent per group
 *   cpg_linear_bwd_input_planes   dX [R, In] (+)= dGin W   (W [G H, In]); scratch = cpg_weight_image_bytes(In, G H) bytes (image of W^T + exponent record)
 *   cpg_linear_bwd_weight_planes  dW [G H, In] (+)= dGin^T x from gp and x's image (csrc/pair_tn.h); workspace per the _workspace query */
CPG_API int cpg_planes_ok(int R, int K, int N);
CPG_API size_t cpg_pair_rows_bytes(int R, int C);
CPG_API size_t cpg_weight_image_bytes(int R, int C);   /* image of a weight matrix [R, C] + the exponent record (cpg_weight_exp) behind it */
CPG_API int cpg_pair_rows(const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int R, void* img, void* stream);
CPG_API int cpg_linear_fwd_planes(const void* ximg, int R, int K, const float* W, int ldw, const float* bias, float* Y, int ldy, int N,
                                  int accumulate, void* scratch, size_t scratch_bytes, void* stream);
CPG_API size_t cpg_grad_planes_bytes(int R, int H, int G);
CPG_API int cpg_grad_planes(const float* dG, int ldg, int R, int H, int G, const int* off, void* gp, void* stream);
CPG_API int cpg_linear_bwd_input_planes(const void* gp, int R, int H, int G, const float* W, int ldw, float* dX, int lddx, int In,
                                        int accumulate, void* scratch, size_t scratch_bytes, void* stream);
CPG_API size_t cpg_linear_bwd_weight_planes_workspace(int R, int H, int G, int In);
CPG_API int cpg_linear_bwd_weight_planes(const void* gp, int R, int H, int G, const void* ximg, int In, float* dW, int lddw, int accumulate,
                                         void* workspace, size_t workspace_bytes, void* stream);
/* One GRU decode step on plane images: GRUDecoder.forward_sample's recurrent part (models/decoder.py:86-99) for decode chains over MANY
 * rows of decoders too wide for the whole-loop kernels (CLaSS at config-B / C width: models/model.py:295-363 over 10^5..10^6 rows).
 * The state travels as (h f32 [N,H], its f16-pair image [N][2H]: cpg_pair_rows makes the first one, every step writes the next);
 * W_hh's image (cpg_weight_image_bytes(3H, H) bytes: rows in tile order, times 2^e_w, + its exponent record) is built once per decode.  Same cell arithmetic and the same
 * 22-bit operands as cpg_gru_step_fwd, no conversion in the product loop.  cpg_gru_step_planes_ok: f32-grade mode, N % 128 == 0,
 * N >= 1024, H % 128 == 0.  cpg_beam_select with H = 0 advances the beams WITHOUT moving any state (h_in / h_out ignored): the next
 * plane step gathers through `origin`. */
CPG_API int cpg_gru_step_planes_ok(int N, int H);
CPG_API int cpg_gru_step_w_image(const float* w_hh, int H, void* wimg, void* stream);
CPG_API int cpg_gru_step_fwd_planes(int N, int H, const void* wimg, const float* b_hh, const int32_t* tok, const float* tab,
                                    const float* rowc, int rowc_rows /* row r reads rowc[r % rowc_rows]: beam-major rows share a sentence's term */,
                                    const float* h_prev, const void* hp_in,
                                    const int32_t* origin /* [nsent][K] beam back-pointers or null: row k nsent + i takes its previous state (h_prev
                                    AND image) from row origin[i][k] nsent + i - _update_hidden (models/model.py:378-385) folded into the loads */,
                                    int nsent, int K, float* h_out, void* hp_out, void* stream);
/* split factor over the rows that cpg_gru_wgrad_hh_ap's product dW[M,N] over R rows runs with (bench.py: workgroups per launch) */
CPG_API int cpg_pair_tn_split(int M, int N, int R);

/* ---- LSTM (NOT in the reference, which is GRU-only - SURVEY F2; semantics = torch.nn.LSTM, gate row order i,f,g,o) --------
 * Same conventions as the GRU entry points; cs is the cell-state slab [(T+1),B,H] (c0 in slot 0 / T), gates [T,4,B,H] =
 * i,f,g,o, dG [T,B,4H] = pre-activation gradients (identical for the input and the hidden side). */
/* Persistent form of cpg_lstm_seq_fwd (csrc/lstm_persist.hip): the whole time loop of one direction in ONE launch - a workgroup
 * keeps the i,f,g,o rows of W_hh of 8 hidden units in LDS (split bf16 planes) for 512 batch rows, the column-tile workgroups of
 * a row tile hand h_t to each other through per-step plane slots + arrival counters (as cpg_gru_seq_fwd_persistent).  Same
 * arguments and results as cpg_lstm_seq_fwd.  cpg_lstm_persistent_fits: 1 when (B,H) is covered on this device
 * (option lstm_persist = 0 disables); sync_scratch: cpg_lstm_persistent_scratch_bytes(T,B,H) bytes, zeroed by the caller when
 * allocated; cpg_lstm_persistent_status reads its sticky error word (0 = no wait has timed out). */
CPG_API int cpg_lstm_persistent_fits(int B, int H);
CPG_API size_t cpg_lstm_persistent_scratch_bytes(int T, int B, int H);
CPG_API int cpg_lstm_seq_fwd_persistent(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh,
                                        const int32_t* tok, const float* tab, const float* rowc, const float* dense,
                                        float* hs, float* cs, float* gates, void* sync_scratch, void* err_host, void* stream);
CPG_API size_t cpg_lstm_persistent_err_offset(int B);
/* name of the kernel a persistent LSTM launch runs at this shape in the current compute mode (profiling label) */
CPG_API int cpg_lstm_persistent_kernel_name(int B, int H, char* buf, int n);
CPG_API int cpg_lstm_persistent_status(int B, const void* sync_scratch, void* stream);
/* Both directions of one biLSTM layer in lock step, ONE launch per step for the pair (as cpg_gru_biseq_bwd): arguments as
 * cpg_lstm_seq_bwd per direction; dh_last_* [B,H] (optional, both or neither) = gradient on each direction's final hidden state;
 * no initial-state gradients.  Extension (the reference has no LSTM). */
CPG_API int cpg_lstm_biseq_bwd(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* cs_f,
                               const float* cs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                               const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f, float* dG_r,
                               float* scratch_f, float* scratch_r, float* w_hhT_scratch_f, float* w_hhT_scratch_r,
                               void* pair_scratch_f, void* pair_scratch_r /* cpg_lstm_bwd_pair_bytes(B, H) bytes each, or null */,
                               void* stream);
/* Launcher introspection (as cpg_gru_step_kernel_name): kind 0 forward step, 1 backward step. */
CPG_API int cpg_lstm_step_kernel_name(int kind, int B, int H, char* buf, int n);
CPG_API int cpg_lstm_step_kernel_is_split(int kind, int B, int H);
CPG_API int cpg_lstm_seq_fwd(int T, int B, int H, int reverse, const float* w_hh, const float* b_hh, const int32_t* tok,
                             const float* tab, const float* rowc, const float* dense, float* hs, float* cs, float* gates,
                             void* stream);
CPG_API int cpg_lstm_step_fwd(int B, int H, const float* w_hh, const float* b_hh, const int32_t* tok, const float* tab,
                              const float* rowc, const float* h_prev, const float* c_prev, float* h_out, float* c_out,
                              void* stream);
/* f16-pair form of the direct-to-LDS backward step, as cpg_gru_bwd_pair_bytes: bytes of scratch per direction, 0 = not covered */
CPG_API size_t cpg_lstm_bwd_pair_bytes(int B, int H);
CPG_API int cpg_lstm_seq_bwd(int T, int B, int H, int reverse, const float* w_hh, const float* cs, const float* gates,
                             const float* dhs_ext, float* dG, float* scratch, float* dh0, float* dc0,
                             float* w_hhT_scratch /* [H,4H] or null: receives W_hh^T for the direct-to-LDS step kernel */,
                             void* pair_scratch /* cpg_lstm_bwd_pair_bytes(B, H) bytes or null */, void* stream);
CPG_API int cpg_lstm_wgrad_hh(int T, int B, int H, int reverse, const float* dG, const float* hs, float* dw_hh,
                              float* db_hh, int accumulate, void* workspace, size_t workspace_bytes,
                              const void* pair_scratch /* the sequence's, or null */, void* stream);
CPG_API int cpg_lstm_dgi_reduce(int T, int B, int H, const float* dG, const int32_t* tok, int V, float* dtab, float* dsum,
                                float* drowc, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* All-T planes form of the LSTM extension's f16-pair BPTT chain (as cpg_gru_*_ap; csrc/pair_engine.h ApScratch with four blocks):
 * the plane images of dG that the backward steps hand to each other are KEPT for every step in `ap` (cpg_lstm_ap_bytes(T, B, H) bytes
 * per direction; 0 = not covered: f32-grade mode, B % 128 == 0, H % 128 == 0, the f16-pair step; option gru_ap = 0 switches it off)
 * and cpg_lstm_wgrad_hh_ap forms dw_hh [4H,H] (+)= images^T x image(h_prev) with no conversion in its loop (it images the state
 * slab hs [T+1,B,H] itself; workspace as cpg_gru_wgrad_workspace).  dG [T,B,4H] f32 may be null in the _ap calls: with an LSTM the
 * input-side gradient is dG and cpg_lstm_dgi_reduce_ap reads the images (token-table layers; their bias gradient is that reduction's
 * column sums); layers with a dense input term pass dG as well. */
CPG_API size_t cpg_lstm_ap_bytes(int T, int B, int H);
CPG_API int cpg_lstm_seq_bwd_ap(int T, int B, int H, int reverse, const float* w_hh, const float* cs, const float* gates,
                                const float* dhs_ext, float* dG, float* scratch, float* dh0, float* dc0, float* w_hhT_scratch,
                                void* ap, void* stream);
CPG_API int cpg_lstm_biseq_bwd_ap(int T, int B, int H, const float* w_hh_f, const float* w_hh_r, const float* cs_f,
                                  const float* cs_r, const float* gates_f, const float* gates_r, const float* dhs_ext_f,
                                  const float* dhs_ext_r, const float* dh_last_f, const float* dh_last_r, float* dG_f, float* dG_r,
                                  float* scratch_f, float* scratch_r, float* w_hhT_scratch_f, float* w_hhT_scratch_r,
                                  void* ap_f, void* ap_r, void* stream);
CPG_API int cpg_lstm_wgrad_hh_ap(int T, int B, int H, int reverse, void* ap, const float* hs, float* dw_hh, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* token-grouped sums / column sums / sums over time of the input-side gradient (= dG for an LSTM) read from the kept images;
 * results as cpg_lstm_dgi_reduce.  With it cpg_lstm_seq_bwd_ap / _biseq_bwd_ap may be given dG = null (token-table layers). */
CPG_API int cpg_lstm_dgi_reduce_ap(int T, int B, int H, const void* ap, const int32_t* tok, int V, float* dtab, float* dsum, float* drowc,
                                   int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- vocabulary projection: nn.Dropout(p_out)+nn.Linear(h_dim,n_vocab), models/decoder.py:43-45,83,107 ----------- */
/* logits[R,V] = (hs[R,H] .* keep*scale) W[V,H]^T + b   (keep uint8 [R,H] or null) */
CPG_API int cpg_vocab_fc_fwd(const float* hs, const uint8_t* keep, float scale, const float* w, const float* b,
                             float* logits, int R, int H, int V, void* stream);
CPG_API size_t cpg_vocab_fc_bwd_workspace(int R, int H, int V);
/* g / count (optional device scalars): dlogits is UNSCALED (cpg_recon_ce_tm_fwd's softmax - onehot) and enters times
 * g[0] / max(count[0], 1) (count null: times g[0]) - the trainer's form, in which neither the target count nor the upstream gradient is
 * known when the forward pass writes dlogits. */
CPG_API int cpg_vocab_fc_bwd(const float* dlogits, const float* hs, const uint8_t* keep, float scale, const float* w,
                             float* dhs, float* dw, float* db, int R, int H, int V, int accumulate, const float* g,
                             const float* count, void* workspace, size_t workspace_bytes, void* stream);
/* losses.recon_dec (losses.py:18-31) on TIME-MAJOR logits [T B, V] (as the vocabulary projection writes them): out[0] = sum of NLL over
 * the non-<pad> targets, out[1] = their number, out[2] = out[0] / max(out[1], 1); dl [T B, V] = softmax - onehot of every scored row
 * (zeros elsewhere), UNSCALED: the backward of the projection applies gout / count (cpg_vocab_fc_bwd).  workspace: 512 floats. */
CPG_API int cpg_recon_ce_tm_fwd(const int64_t* ids, const float* logits_tm, int B, int T, int V, int pad, float* out, float* dl,
                                float* workspace, void* stream);

/* ---- decoding: RNN_VAE.sample_G, models/model.py:225-385; models/Beam.py ----------------------------------------- */
/* greedy: tok = argmax (first max), finished rows emit <pad>, rows that emit <eos> become finished; writes column `col`
 * of ids [N,ld_ids] and tok_next; unfinished[step] += rows still running after this step.  prevent_empty masks
 * pad/start/eos with -2*|min(logits)| (model.py:299-305); scratch: 257 floats. */
CPG_API int cpg_greedy_select(const float* logits, int N, int V, uint8_t* finished, int64_t* ids, int ld_ids, int col,
                              int32_t* tok_next, int pad, int start, int eos, int prevent_empty, float* scratch,
                              int* unfinished, int step, void* stream);
/* categorical: tok ~ Categorical(logits = logits/temp) (models/model.py:308-309) by inverse CDF in index order with one
 * uniform per row PASSED IN (uniforms f64 [N], e.g. from cpg_rng_uniform_f64; parity tests replay the reference's draws);
 * finished / <eos> / prevent_empty / unfinished[] exactly as cpg_greedy_select. */
CPG_API int cpg_categorical_select(const float* logits, int N, int V, float temp, const double* uniforms, uint8_t* finished,
                                   int64_t* ids, int ld_ids, int col, int32_t* tok_next, int pad, int start, int eos,
                                   int prevent_empty, float* scratch, int* unfinished, int step, void* stream);
/* Whole greedy loop (model.py:225-385 with decoder.py:86-109) as ONE persistent launch for small decoders: W_hh in
 * registers, hidden state / rowc / token table / fc staged in LDS, only token ids leave the CU.  Requires H <= 128,
 * V <= 32 and cpg_decode_greedy_fused_lds_bytes(H,V,Vt) <= the device's LDS per workgroup (returns -3 otherwise; the
 * per-step entry points above cover every other shape).  ids [N,ld_ids] must be pre-filled with <pad> and column 0 with
 * <start>; columns 1..T are written while a row is running; unfinished[t] += rows still running after step t. */
CPG_API size_t cpg_decode_greedy_fused_lds_bytes(int H, int V, int Vt);
CPG_API int cpg_decode_greedy_fused(const float* h0, const float* rowc, const float* tab, int Vt, const float* w_hh,
                                    const float* b_hh, const float* fc_w, const float* fc_b, int N, int H, int V, int T,
                                    int start, int pad, int eos, int64_t* ids, int ld_ids, int* unfinished, void* stream);
/* Whole beam search (model.py:258-276,314-328,364-376,387-404; Beam.py:56-105) as ONE persistent launch for small
 * decoders: a workgroup tile holds whole sentences (K beams each, sentence-major rows), Beam.advance and the hidden-state
 * re-gather happen in LDS.  h0/rowc: one row per sentence.  hist_tok must be pre-filled with -1 (steps a finished
 * sentence is not advanced on keep it); hist_* [T,N,K] feed cpg_beam_hypotheses.  Same shape limits as the greedy one
 * plus K <= 8, K <= V; -3 if cpg_decode_beam_fused_lds_bytes exceeds the device's LDS per workgroup. */
/* Launcher introspection: the whole-loop decode kernel a launch runs (kind 0 greedy, 1 beam), as rocprofv3 prints it. */
CPG_API int cpg_decode_fused_kernel_name(int kind, int H, int K, char* buf, int n);
CPG_API size_t cpg_decode_beam_fused_lds_bytes(int H, int V, int Vt, int K);
CPG_API int cpg_decode_beam_fused(const float* h0, const float* rowc, const float* tab, int Vt, const float* w_hh,
                                  const float* b_hh, const float* fc_w, const float* fc_b, int N, int H, int V, int T, int K,
                                  int n_best, int min_length, int bos, int eos, int32_t* hist_tok, int32_t* hist_prev,
                                  float* hist_score, void* stream);
/* beam: one Beam.advance for every sentence (rows beam-major: row = k*N + i) + hidden-state reorder.  K <= 32 and K <= V
 * (the first step ranks the V children of beam 0 only, models/Beam.py:82-84); the reference's static_eval.py:130 uses K = 15.
 * scores/last_tok/origin [N,K]; n_finished, done [N]; hist_* [T,N,K]; n_active[step] += sentences not yet done. */
CPG_API int cpg_beam_select(const float* logits, int N, int V, int K, int step, int n_best, int min_length, int bos,
                            int eos, float* scores, int32_t* last_tok, int32_t* n_finished, uint8_t* done,
                            int32_t* hist_tok, int32_t* hist_prev, float* hist_score, int32_t* origin, int32_t* tok_next,
                            int* n_active, const float* h_in, float* h_out, int H, void* stream);
/* the same back-pointer reorder for one more beam-major state array [K*N,H] (the LSTM extension's cell state) */
CPG_API int cpg_beam_reorder(const float* h_in, float* h_out, const int32_t* origin, int N, int K, int H, void* stream);
/* Beam.sort_finished + get_hyp (models/Beam.py:110-132) for all sentences from the recorded history: finished entries
 * ranked by raw summed log-prob (stable, insertion order = step then beam), topped up from the live beam of the last
 * advanced step.  hyps int32 [N,n_best,T+1] (<start> first, -1 padded), lens/scores [N,n_best]. */
CPG_API int cpg_beam_hypotheses(const int32_t* hist_tok, const int32_t* hist_prev, const float* hist_score, int T, int N,
                                int K, int n_best, int eos, int start, int32_t* hyps, int32_t* lens, float* scores,
                                void* stream);

/* ---- losses (losses.py) ------------------------------------------------------------------------------------ */
/* recon_dec, losses.py:18-31: targets = cat(ids[:,1:], PAD). out[0] = sum NLL over non-PAD targets, out[1] = their count.
 * workspace: 512 floats. */
CPG_API int cpg_recon_ce_fwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, float* out,
                             float* workspace, void* stream);
/* the same with the loss itself appended: out[2] = out[0] / max(out[1], 1) (F.cross_entropy(..., reduction='mean',
 * ignore_index=PAD), losses.py:27-31).  out: 3 floats. */
CPG_API int cpg_recon_ce_loss_fwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, float* out,
                                  float* workspace, void* stream);
/* train_vae.py:35-37, `loss = recon + beta*regu + l1w*L1 + klw*KLpen`: out[0] = sum_i w_i * t_i[0] over the non-null device
 * scalars, products and sums rounded one by one, left to right; cpg_scale_fanout4: out[i] = g[0] * w_i (its gradient). */
/* wdev (optional, device float[4]) replaces w0..w3: a captured (hipGraph) training step reads the annealed beta from memory. */
CPG_API int cpg_weighted_sum4(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1,
                              float w2, float w3, const float* wdev, float* out, void* stream);
CPG_API int cpg_scale_fanout4(const float* g, float w0, float w1, float w2, float w3, const float* wdev, float* out,
                              void* stream);
/* dlogits = gout[0] * (softmax - onehot) / count[0] on valid rows (gout, count: device scalars) */
CPG_API int cpg_recon_ce_bwd(const int64_t* ids, const float* logits, int B, int T, int V, int pad, const float* gout,
                             const float* count, float* dlogits, void* stream);
/* RNN_VAE.sample_z, models/model.py:107-112: z = mu + exp(logvar/2)*eps */
CPG_API int cpg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n, void* stream);
CPG_API int cpg_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, size_t n,
                            void* stream);
/* out[0..4] = sums for kl_gaussianprior (losses.py:8-10), kl_gaussian_sharedmu (:13-15), |logvar| (train_vae.py:33),
 * |mu|, logvar (train_vae.py:44-45).  workspace: 1280 floats. */
CPG_API int cpg_latent_stats_fwd(const float* mu, const float* logvar, size_t n, float* out, float* workspace,
                                 void* stream);
CPG_API int cpg_latent_stats_bwd(const float* mu, const float* logvar, size_t n, int B, const float* g_kl,
                                 const float* g_klmu, const float* g_l1, float* dmu, float* dlogvar, int accumulate,
                                 void* stream);
/* The latent block of a training step, fused (round 6): RNN_VAE.sample_z + sample_c_prior + GRUDecoder.init_hidden + the three analytic
 * latent penalties (models/model.py:107-126, models/decoder.py:53-54, losses.py:8-15, train_vae.py:33) as one elementwise launch + its
 * 5-value final sum.  eps_in / c_in (optional): injected draws; null = drawn here from the counter streams (seed, off_eps / off_c, base)
 * - the numbers cpg_rng_normal / cpg_rng_onehot2 write for the same (seed, offset); C = 2 then.  Outputs: z [B,Z]; zc = [z ; c]
 * [B, Z + C] (the decoder's initial state and constant input: no concatenation launch); c_out [B,C]; eps_out [B,Z] (optional: what the
 * backward needs when eps was drawn here); out5 = (kl, kl_sharedmu, logvar_L1 - each / B -, sum |mu|, sum logvar).
 * workspace: cpg_latent_fused_workspace() bytes.  Backward: dmu, dlogvar from the gradients on z (dz), on zc (dzc, row stride ldzc;
 * its first Z columns) and on the three penalties (device scalars; any of these may be null). */
CPG_API size_t cpg_latent_fused_workspace(void);
CPG_API int cpg_latent_fused_fwd(const float* mu, const float* logvar, const float* eps_in, const float* c_in, int B, int Z, int C,
                                 uint64_t seed, uint64_t off_eps, uint64_t off_c, const uint64_t* base, float p_one, float* eps_out,
                                 float* z, float* zc, float* c_out, float* out5, float* workspace, void* stream);
CPG_API int cpg_latent_fused_bwd(const float* dz, const float* dzc, int ldzc, const float* mu, const float* logvar, const float* eps,
                                 int B, int Z, const float* g_kl, const float* g_klmu, const float* g_l1, float* dmu, float* dlogvar,
                                 int ldo /* row stride of dmu / dlogvar */, int Zp /* >= Z columns written, zeros from Z on */, void* stream);
/* mmd_rf, losses.py:59-93: raw = z @ rf_w by cpg_matmul_nn; sums[R] = sum_b cos(raw/sigma + rf_b)*sqrt(2/R) */
CPG_API int cpg_rf_feature_sums(const float* raw, const float* rf_b, int Bn, int R, float sigma, float* sums,
                                float* workspace, size_t workspace_bytes, void* stream);
CPG_API int cpg_rf_loss(const float* sums1, const float* sums2, int R, int B_global, float* loss, float* diff,
                        void* stream);
/* The same term in three launches instead of eight (round 6): cpg_rf_features = the feature sums of nx <= 2 inputs (z and z_prior:
 * x0, x1 [Bn, Z], row stride ldx) against one basis in ONE grouped launch whose epilogue applies cos(. / sigma + rf_b) sqrt(2 / R) and
 * sums each 64-row chunk - part [nx][chunks][R], cpg_rf_features_workspace bytes; the [Bn, R] feature matrix never exists, raw0
 * (optional) receives x0 rf_w for the backward pass.  cpg_rf_sums_loss: sums over the chunks (chunk order) -> sums1, sums2 [R], and,
 * where loss / diff are given, diff = (s1 - s2) / B_global and loss = sum diff^2 in the same launch (null: a data-parallel caller
 * all-reduces the sums, then calls cpg_rf_loss). */
CPG_API size_t cpg_rf_features_workspace(int nx, int Bn, int R);
CPG_API int cpg_rf_features(int nx, const float* x0, const float* x1, int ldx, int Bn, int Z, const float* rf_w, int R, const float* rf_b,
                            float sigma, float* raw0, float* part, size_t part_bytes, void* stream);
CPG_API int cpg_rf_sums_loss(const float* part, int chunks, int R, int B_global, float* sums1, float* sums2, float* loss, float* diff,
                             void* stream);
CPG_API int cpg_rf_bwd(const float* raw, const float* rf_b, const float* diff, const float* gout, int Bn, int R,
                       float sigma, int B_global, float* dpre, void* stream);
/* mmd_full_kernel, losses.py:47-56,96-108 (incl. the `H - diag(H)` broadcast).
 * kernel: 0 "gaussian", 1 "laplace", 2 "energy" - compute_mmd_kernel, losses.py:102-107. */
CPG_API size_t cpg_mmd_full_workspace(int N, int D);
CPG_API int cpg_mmd_full_fwd(const float* z1, const float* z2, int N, int D, float sigma, int kernel, float* out, float* P,
                             float* Q, void* workspace, size_t workspace_bytes, void* stream);
CPG_API int cpg_mmd_full_bwd(const float* z1, const float* z2, const float* P, const float* Q, const float* gout, int N,
                             int D, float sigma, int kernel, float* dz1, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ---- optimiser: clip_grad_norm_ + Adam, train_vae.py:15,39-42 ---------------------------------------------------- */
CPG_API size_t cpg_sumsq_workspace(void);
CPG_API int cpg_sumsq(const float* x, size_t n, float mult, int accumulate, float* out, float* workspace, void* stream);
/* iter (optional, device int32): the step number is step_mult * iter[0] + step, formed on the device (captured steps) */
CPG_API int cpg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, int step, const float* sumsq, float max_norm, int coef_pow, float gscale,
                          const int32_t* iter, int step_mult, void* stream);
/* One-launch forms (round 6): the whole flat buffer's clipped-norm partials (cpg_sumsq_segs: SUMSQ partial sums of w_i g_i^2 into
 * `workspace`, w_i = the multiplicity of element i's parameter) and the whole Adam iteration (cpg_adam_step_segs: every block reduces
 * the partials in the same order, then updates its elements; an element of a parameter listed m times takes m consecutive Adam steps
 * with its gradient scaled by coef^m - SURVEY F6).  dup_off / dup_len: the PADDED flat-buffer segments (multiples of 4 elements;
 * padding holds zero gradients) of the ndup <= 2 parameters listed dup_mult <= 4 (the same for all) times.  iter: device int32, the
 * completed iterations (step numbers are formed from it; the caller advances it afterwards: cpg_step_counters_add). */
CPG_API int cpg_sumsq_segs(const float* g, size_t n, int ndup, const unsigned long long* dup_off, const unsigned long long* dup_len,
                           const int* dup_mult, float* workspace, void* stream);
CPG_API int cpg_adam_step_segs(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                               const float* partials /* or null: no clipping */, float* sumsq_out /* optional: sum of squares */,
                               float max_norm, float gscale, const int32_t* iter, int ndup, const unsigned long long* dup_off,
                               const unsigned long long* dup_len, const int* dup_mult, void* stream);
/* *rng_base += rng_by ; *iter += iter_by (either pointer may be null): the device-side counters a training step advances, one launch */
CPG_API int cpg_step_counters_add(uint64_t* rng_base, uint64_t rng_by, int32_t* iter, int32_t iter_by, void* stream);

/* ---- CLaSS sampler: density_modeling.py:43-60,79-80 (sklearn GaussianMixture.sample / LogisticRegression) ---------- */
CPG_API int cpg_gmm_sample(const double* means, const double* covars, const int32_t* comp, const double* normals, int n,
                           int D, float* z, void* stream);
CPG_API int cpg_lr_score_accept(const float* z, int n, int D, const double* coef, const double* intercept,
                                const int32_t* target, int A, const double* uniforms, double* probs, double* accum,
                                uint8_t* accepted, void* stream);

/* Residue rows of decoded ids: what dataset.idx2sentences(..., print_special_tokens=False) keeps of each row
 * (data_processing/dataset.py:285-300, called from sample_pipeline.py:129-139), as arrays.  ids int16 [n, L] (ids below
 * first_residue - the 4 specials, or -1 padding - are dropped); letters uint8 [n, L] left-aligned, zero-filled; counts [n]. */
CPG_API int cpg_residue_rows(const int16_t* ids, size_t n, int L, int first_residue, uint8_t* letters, int32_t* counts,
                             void* stream);

/* ---- CNN classifier (ADJACENT row): models/classifier.py:39-60, reached through q_c='classifier' (models/model.py:186-188) ----
 * pooled[B, nconv*F] = max_p relu(bias + sum_dw tab[dw][ids[b,p+dw]]) for filters of widths min_width..min_width+nconv-1;
 * tabs = per layer [w][V][F] tables emb @ W[:,0,dw,:]^T (cpg_linear_fwd), layers back to back; bias [nconv,F].
 * argpos (optional, int16 [B, nconv*F]): position each maximum came from (-1: ReLU flat) - what the backward needs.
 * _bwd: dtabs (same layout as tabs) and dbias from dpooled; deterministic (batch walked in order per (layer, filter)). */
CPG_API int cpg_cnn_classifier_pool(const int64_t* ids, int B, int T, int V, int F, int min_width, int nconv,
                                    const float* tabs, const float* bias, float* pooled, int16_t* argpos, void* stream);
CPG_API int cpg_cnn_classifier_pool_bwd(const int64_t* ids, const int16_t* argpos, const float* dpooled, int B, int T, int V,
                                        int F, int min_width, int nconv, float* dtabs, float* dbias, void* stream);

/* ---- RCCL collectives of the path (the reference is single-device: new) -----------------------------------
 * In-process communicator, one process per GPU.  librccl is bound at run time (the copy a PyTorch process has already loaded
 * is reused); cpg_comm_available() == 0 when it cannot be.  Launcher: rank 0 -> cpg_comm_unique_id(id[128]) -> ship the
 * bytes to every rank -> all ranks cpg_comm_init(id, rank, world, &comm) (collective).
 * cpg_allreduce_f32: in-place SUM of the flat gradient buffer (train_vae.py's step under data parallelism), asynchronous on
 * `stream` - issue it on a side stream to overlap with the rest of the backward pass.
 * cpg_allgatherv: rank r contributes counts[r] BYTES (counts: host array, identical on every rank); recv gets them back to back
 * in rank order - the accepted / decoded rows of a CLaSS sampling round (sample_pipeline.py:299-322 on sharded rounds). */
CPG_API int cpg_comm_available(void);
CPG_API int cpg_comm_unique_id(void* id128);
CPG_API int cpg_comm_init(const void* id128, int rank, int world, void** comm);
CPG_API int cpg_comm_destroy(void* comm);
/* ranks the communicator reports (ncclCommCount), -1 if unavailable: bench.py prints it as rccl.ranks_seen */
CPG_API int cpg_comm_count(void* comm);
CPG_API int cpg_allreduce_f32(void* comm, float* buf, size_t n, void* stream);
CPG_API int cpg_allgatherv(void* comm, const void* send, const size_t* counts, int rank, int world, void* recv, void* stream);

/* ---- counter-based random streams (Philox4x32-10) for callers that do not inject the draws ----------------------- */
/* base (optional, device uint64): added to `offset` on the device.  A training step replayed from a hipGraph keeps host-side
 * offsets relative to the step and advances *base once per step with cpg_counter_add_u64. */
CPG_API int cpg_rng_normal(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream);
CPG_API int cpg_rng_uniform(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream);
CPG_API int cpg_rng_uniform_f64(double* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream);
CPG_API int cpg_rng_bernoulli_u8(uint8_t* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base,
                                 void* stream);
/* c ~ Cat([1 - p_one, p_one]) as one-hot float rows out[n][2] from the draws cpg_rng_bernoulli_u8 makes with the same
 * (seed, offset, base): RNN_VAE.sample_c_prior (models/model.py:122-126) in one launch. */
CPG_API int cpg_rng_onehot2(float* out, size_t n, float p_one, uint64_t seed, uint64_t offset, const uint64_t* base, void* stream);
/* device-side step counters of a captured training step: *p += by (one thread) */
CPG_API int cpg_counter_add_u64(uint64_t* p, uint64_t by, void* stream);
CPG_API int cpg_counter_add_i32(int32_t* p, int32_t by, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPG_API_H */
