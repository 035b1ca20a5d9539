/* shg_vqa.h - C ABI of libshgvqa.so, the MI355X (gfx950) implementation of the SHG-VQA hot path.
 *
 * The reference (aurooj/SHG-VQA) is 100 % PyTorch eager and has no FFI of its own; its seam is the
 * nn.Module API.  Each entry point below states which reference
 * arithmetic it replaces (file:line under AGQA/src/).  The host side (the shg_vqa_amd python package) mirrors the
 * reference's module interface and reaches these functions through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every pointer is a DEVICE pointer owned by the caller
 *     unless the comment says "host".  The library never allocates or frees device memory and holds
 *     no mutable global state besides the last error string, the tuning table, a launch counter for tests and the per-device "shared-memory limit
 *     raised" bits of its kernels; workspaces and the executor's event ring are caller-owned handles.
 *   - Every function only enqueues work on `stream` (a hipStream_t passed as void*); it is safe to
 *     call during hipGraph stream capture.
 *   - Return value: 0 = ok, SHG_ERR_INVALID (<0) = bad argument (see shg_last_error_string()),
 *     >0 = a hipError_t from the launch.
 *   - dtype: SHG_F32 or SHG_BF16 selects the storage type of activations/operands.  Statistics
 *     (mean/rstd/LSE), losses, optimiser state and master weights are always fp32.
 *   - Matrices are row-major and contiguous unless a stride argument exists.
 *   - Dropout masks are a pure function of (seed_state, stream_id, element index): `seed_state`
 *     points to two uint64 {seed, step} in device memory (may be NULL when p == 0) so that a captured
 *     graph gets fresh masks on every replay; `stream_id` separates call sites within a step.
 */
#ifndef SHG_VQA_H
#define SHG_VQA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHG_F32 0
#define SHG_BF16 1

#define SHG_ERR_INVALID (-1)

#define SHG_ACT_NONE 0
#define SHG_ACT_GELU 1 /* erf form, modeling_capsbert.py:127-133 */
#define SHG_ACT_RELU 2 /* transformer.py:195 (decoder FFN) */
/* shg_gemm_act: OR-ed to the activation code, `pre` receives act'(A . B + bias) instead of the pre-activation (bf16 outputs only);
 * shg_gemm_dact then takes act = SHG_ACT_SAVED_GRAD and multiplies by the stored derivative: the erf-GELU derivative (an exponential,
 * a reciprocal and a degree-5 polynomial per element) is computed once, in the forward epilogue next to the GELU itself with which it
 * shares all of that, instead of in the input-gradient GEMM's epilogue where nothing overlaps it (12 576 x 3 072: 105 -> 82 us). */
#define SHG_ACT_SAVE_GRAD 0x100
#define SHG_ACT_SAVED_GRAD 3

#define SHG_MASK_NONE 0
#define SHG_MASK_KEY 1  /* additive fp32 [B, Sk]   (BERT (1-m)*-1e4 masks, modeling_capsbert.py:1826-1842) */
#define SHG_MASK_FULL 2 /* additive fp32 [Sq, Sk]  (block-causal -inf/0 mask, entry.py:114-121) */

int shg_version(void);
const char* shg_last_error_string(void);
/* Tuning switches.  The library reads NO environment variables; every dispatch threshold / kernel variant choice is one entry of a
 * table with its measured-best default (csrc/common.h `enum Tune`, csrc/api.hip): "attn_nb", "attn_nb_dq", "attn_nb_dkv",
 * "tile_order", "gemm4_max_tiles", "streamk_sigma", "streamk", "gemm8", "gemm8_min_tiles", "splitk_target", "splitk_min_steps",
 * "large_min_k", "wgrad_group", "conv_wgrad_remainder", "bertadam_mode", "bertadam_blocks", "gemm8_tile_m", "attn_bwd_fused",
 * "epilogue_side", "repeat_family" (diagnostic: launches of a kernel family are issued twice, tools/family_cost.py).
 * A binding sets them once (the Python host maps SHG_<NAME> environment variables onto them at load time); host-side only, takes
 * effect at the next launch.  shg_set_tuning: 0 or SHG_ERR_INVALID (unknown name); shg_get_tuning: value or INT64_MIN;
 * shg_tuning_name(i): name of entry i, NULL past the end.  (The reference has no counterpart: its knobs are argparse flags,
 * param.py:20-160.) */
int shg_set_tuning(const char* name, int64_t value);
int64_t shg_get_tuning(const char* name);
const char* shg_tuning_name(int index);
/* Diagnostics: number of convolution launches so far that used the stream-K work split of the 256 x 256 kernel (gemm.hip:
 * every CU gets the same number of K-tiles).  Tests use it to prove the path was exercised. */
int64_t shg_gemm_streamk_launches(void);
/* Host evaluation of that work split for a launch of n_tiles (128..255) output tiles with nk K-tiles each: segment `seg` of
 * workgroup `block` (0..255) -> out[6] = {tile, first K-tile, number of K-tiles, owner (1: head, runs the epilogue),
 * partial-sum slot a tail publishes to, number of published parts the owner adds}.  Returns 1, or 0 when the workgroup has no
 * such segment (SHG_ERR_INVALID on bad arguments).  Pure host arithmetic, no GPU needed. */
int shg_streamk_plan(int n_tiles, int nk, int block, int seg, int* out);
/* the WEIGHTED plan (tiles of different length: the conv forward in position-major row order leaves out the taps that read the
 * zero border for every row of a tile): nk_tile[n_tiles] K-tiles per tile; same descriptor, slots 3 tile + part; the XCDs' runs
 * of tiles are cut by K-tiles, not by tiles; -1 = these lengths do not fit the plan */
int shg_streamk_plan_weighted(int n_tiles, const uint16_t* nk_tile, int block, int seg, int* out);
/* Caller-owned workspace of that split (partial-sum slots + flags): shg_streamk_workspace_bytes() bytes of device memory,
 * 16-byte aligned, initialised ONCE by shg_streamk_workspace_init (zeroes the flags on `stream`; launches leave them zero).
 * One workspace serves the launches of one stream at a time; pass it to shg_conv3d_k533_fwd (NULL: no split). */
int64_t shg_streamk_workspace_bytes(void);
int shg_streamk_workspace_init(void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hungarian matcher, per-frame branch.
 * Replaces HungarianMatcher.forward (lxrt/matcher.py:62-80) including the softmax, the
 * out_prob[:, tgt_ids] gather, the .cpu() copy and the python loop over
 * scipy.optimize.linear_sum_assignment; also builds the target-class grid of
 * AGQA.loss_labels / get_target_classes (tasks/agqaHGQA.py:178-220).
 *   logits     [n_frames, per_frame, n_classes]  (dtype)      per_frame <= 128.  per_frame <= 8 (the per-frame problems of
 *              --LossHGPerFrame): one lane per problem; 9..128 (per-clip matching, matcher.py:82-104, one problem per
 *              sample with per_frame = num_queries): one wave per problem, the column scan spread over the lanes
 *   tgt        [n_frames, per_frame] int64, class ids of the frame's targets, first tgt_len[f] valid
 *   tgt_len    [n_frames] int32 (0..per_frame)
 *   out_query  [n_frames, per_frame] int64: matched query index within the frame, ascending, -1 padded
 *   out_target [n_frames, per_frame] int64: matched target index within the frame, -1 padded
 *   out_grid   [n_frames, per_frame] int64: class id for every query slot (background_class unless matched)
 * Assignment indices are bit-identical to SciPy's (float64 JV solver with SciPy's tie rules).
 */
int shg_hungarian_per_frame(const void* logits, int dtype, int n_frames, int per_frame, int n_classes,
                            const int64_t* tgt, const int32_t* tgt_len, int64_t background_class,
                            int64_t* out_query, int64_t* out_target, int64_t* out_grid, void* stream);

/* Same solver on explicit cost matrices (test entry point; lxrt/matcher.py:79).
 *   cost [n, rows, cols_max] fp32, n_cols [n] int32 (<= cols_max <= 8, rows <= 8) */
int shg_lsap_batched(const float* cost, int n, int rows, int cols_max, const int32_t* n_cols,
                     int64_t* out_row, int64_t* out_col, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Set loss: class-weighted cross entropy over every query slot.
 * Replaces F.cross_entropy(logits.transpose(1,2), grid, weight) in AGQA.loss_labels
 * (tasks/agqaHGQA.py:223) and its backward.
 *   fwd : row_stats [4, rows] fp32 = {lse[r]}, {w*nll[r]}, {w[r]}, {match flag};  sums[0..1] = {sum w*nll, sum w}
 *         (deterministic two-stage reduction), sums[2] = #rows whose arg-max equals a non-background
 *         target, sums[3] = #rows with a non-background target (for class_error, agqaHGQA.py:228).
 *   loss = sums[0]/sums[1] is formed by the caller (after an all-reduce of `sums` under data parallel).
 *   bwd : dlogits[r,c] = gscale[0] * w[t_r] * (softmax(r)[c] - [c==t_r]) / sums[1]   (gscale NULL => 1);
 *         dlogits rows are ldd >= n_classes elements apart and columns n_classes..ldd-1 are zeroed, so the
 *         buffer can feed shg_gemm directly (16-byte rows).
 */
int shg_weighted_ce_fwd(const void* logits, int dtype, int64_t rows, int n_classes, const int64_t* target,
                        const float* class_weight, int64_t background_class, float* row_stats, float* sums,
                        void* stream);
int shg_weighted_ce_bwd(const void* logits, int dtype, int64_t rows, int n_classes, const int64_t* target,
                        const float* class_weight, const float* row_stats, const float* sums,
                        const float* gscale, void* dlogits, int64_t ldd, void* stream);

/* BCEWithLogitsLoss(mean) * n_classes  (tasks/agqaHGQA.py:344-345).
 *   loss[0] = n_classes * mean(bce);  dlogits = gscale[0] * (sigmoid(x) - y) / rows            */
int shg_bce_logits_fwd_bwd(const void* logits, int dtype, int64_t rows, int n_classes, const float* target,
                           const float* gscale, float* loss, void* dlogits, int64_t ldd, void* stream);

/* The step's scalar loss arithmetic in one launch each way (agqaHGQA.py:344-378):
 *   total[0] = bce[0] * bce_scale + rel_sums[0] / rel_sums[1] + act_sums[0] / act_sums[1]
 *   diag[5]  = { bce, rel CE, act CE, rel class error %, act class error % }   (class error = 100 - 100 * sums[2] / max(sums[3], 1))
 * rel_sums / act_sums are the [4] outputs of shg_weighted_ce_fwd (after the data-parallel all-reduce, if any).
 * bwd: d_total [1] (NULL = 1) -> d_rel_sums [4], d_act_sums [4] (numerator and denominator slots), d_bce [1]. */
int shg_loss_combine_fwd(const float* rel_sums, const float* act_sums, const float* bce, float bce_scale, float* total,
                         float* diag, void* stream);
int shg_loss_combine_bwd(const float* d_total, const float* rel_sums, const float* act_sums, float bce_scale,
                         float* d_rel_sums, float* d_act_sums, float* d_bce, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused epilogues.
 * y = act(x + bias), optional dropout.  Replaces BertIntermediate's bias+erf-GELU
 * (modeling_capsbert.py:472-475) and the decoder FFN's bias+ReLU+dropout (transformer.py:230).
 *   x, y [rows, cols] (dtype); bias [cols] fp32 or NULL.  bwd: dx = dy * act'(x+bias) * keep/(1-p),
 *   dbias_partial [n_partials, cols] fp32 (column sums of dx per row-chunk; reduce with shg_colsum_finish).
 */
int shg_bias_act_fwd(const void* x, const float* bias, void* y, int dtype, int64_t rows, int cols, int act,
                     float p_drop, const uint64_t* seed_state, uint64_t stream_id, void* stream);
int shg_bias_act_bwd(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial,
                     int n_partials, int dtype, int64_t rows, int cols, int act, float p_drop,
                     const uint64_t* seed_state, uint64_t stream_id, void* stream);
/* The same with two views (the conv stack's backward, modeling_capsbert.py:991-996 / :1037-1073 run in reverse):
 *   dy is read from groups of dy_group_stride rows, skipping the first dy_row_offset rows of every group and taking
 *   dy_rows_per_group (the token gradients [B, 1 + 392, C] without the cls rows; dy_rows_per_group == 0: plain [rows, cols]);
 *   dx2 (optional, same dtype) receives a second copy of the result, row r at row dx2_rows[r] (int32 table; NULL: r) - the
 *   zero-bordered layout the input-gradient convolution gathers from. */
int shg_bias_act_bwd_view(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial, int n_partials,
                          int dtype, int64_t rows, int cols, int act, float p_drop, const uint64_t* seed_state,
                          uint64_t stream_id, int64_t dy_rows_per_group, int64_t dy_group_stride, int64_t dy_row_offset,
                          void* dx2, const int32_t* dx2_rows, void* stream);
/* ... and with x_rows (int32 [rows] or NULL): result row r reads x at row x_rows[r] and writes dx at row x_rows[r] (dy, dx2 and the
 * bias-gradient sums are indexed by r as before) - the conv stack's backward with its saved pre-activation and its dense result in
 * position-major rows (shg_conv3d_k533_prepare_ex, row_order 1) while the incoming token gradients stay in sequence order.  No
 * dropout in this form. */
int shg_bias_act_bwd_rows(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial, int n_partials,
                          int dtype, int64_t rows, int cols, int act, float p_drop, const uint64_t* seed_state,
                          uint64_t stream_id, int64_t dy_rows_per_group, int64_t dy_group_stride, int64_t dy_row_offset,
                          void* dx2, const int32_t* dx2_rows, const int32_t* x_rows, void* stream);

/* z = dropout(act(x + bias)) + residual;  y = LayerNorm(z) * gamma + beta.
 * Replaces BertAttOutput / BertOutput (modeling_capsbert.py:431-435, :485-489), the decoder's
 * residual+norm{1,2,3} (transformer.py:220-232), the MLP heads' GELU+LayerNorm
 * (tasks/agqa_model.py:105-110) and the LayerNorm of the embedding blocks (modeling_capsbert.py:322, :353).
 *   x [rows, cols] (dtype); bias [cols] fp32 or NULL; residual [rows, cols] (dtype) or NULL
 *   y [rows, cols] (dtype); z_out [rows, cols] (dtype) or NULL (pre-norm sum, needed by bwd);
 *   mean, rstd [rows] fp32.  cols <= 4096, cols % 8 == 0.
 * bwd: given dy and z (as saved), produces dz-style gradients:
 *   dres (may be NULL) = dz;  dx = dz * keep/(1-p) * act'(x+bias)  (x needed only when act != NONE)
 *   dgamma_partial/dbeta_partial/dbias_partial [n_partials, cols] fp32.
 */
int shg_bias_act_drop_res_ln_fwd(const void* x, const float* bias, const void* residual, const float* gamma,
                                 const float* beta, void* y, void* z_out, float* mean, float* rstd, int dtype,
                                 int64_t rows, int cols, int act, float eps, float p_drop,
                                 const uint64_t* seed_state, uint64_t stream_id, void* stream);
int shg_bias_act_drop_res_ln_bwd(const void* dy, const void* z, const void* x, const float* bias,
                                 const float* gamma, const float* mean, const float* rstd, void* dx, void* dres,
                                 float* dgamma_partial, float* dbeta_partial, float* dbias_partial,
                                 int n_partials, int dtype, int64_t rows, int cols, int act, float p_drop,
                                 const uint64_t* seed_state, uint64_t stream_id, void* stream);
/* partial[p, c] = sum over the p-th row chunk of x[r, c]  (x rows ld elements apart; bias gradient of a GEMM
 * whose bias is added in its epilogue, e.g. the Q/K/V projections modeling_capsbert.py:373-375) */
int shg_colsum_partial(const void* x, int dtype, int64_t rows, int cols, int64_t ld, float* partial, int n_partials,
                       void* stream);
/* out[c] (+)= sum_p partial[p, c]   (deterministic second stage of the column reductions) */
int shg_colsum_finish(const float* partial, int n_partials, int cols, float* out, int accumulate, void* stream);
/* outs[i][c] += sum_p partials[i][p, c] for i < count <= 4 equally shaped partial buffers in ONE launch: the
 * gamma / beta / bias gradients of one LayerNorm backward (modeling_capsbert.py:431-435, :485-489).
 * `partials` / `outs` are HOST arrays of device pointers. */
int shg_colsum_finish_multi(const float* const* partials, float* const* outs, int count, int n_partials, int cols,
                            void* stream);
/* number of row-chunks (n_partials) the bwd kernels use for `rows` */
int shg_colsum_partials(int64_t rows);
/* out[c] += sum_r x[r, c] in ONE launch (the bias gradient of an nn.Linear, modeling_capsbert.py:373-375, next to its
 * weight-gradient GEMM): 16-byte loads, 64 rows per workgroup, one 64-lane fp32 atomic per wave and 64 columns.
 * bf16 / fp32, cols % 8 (bf16) or % 4 (fp32) == 0, ld likewise, x 16-byte aligned; cols <= 4096 (bf16) / 2048 (fp32).  The summation order
 * across workgroups is not fixed (fp32 atomics), like the bias sums of shg_gemm_dact. */
int shg_colsum_accumulate(const void* x, int dtype, int64_t rows, int cols, int64_t ld, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-head attention, head dim 64.
 * O = dropout(softmax(scale * Q K^T + mask)) V, never materialising the [B,H,Sq,Sk] scores.
 * Replaces BertAttention.forward's scores/softmax/dropout/context (modeling_capsbert.py:394-421)
 * and the core of nn.MultiheadAttention in the DETR decoder (transformer.py:219-229).
 *   q [B, Sq, H, 64] addressed as q + b*q_bstride + s*q_sstride + h*64 (strides in elements), same for k, v;
 *   o [B, Sq, H*64] contiguous; lse [B, H, Sq] fp32 (log-sum-exp of the scaled, masked scores).
 *   mask: SHG_MASK_KEY -> fp32 [B, Sk]; SHG_MASK_FULL -> fp32 [Sq, Sk]; -inf allowed.
 * bwd recomputes the probabilities from lse; delta [B,H,Sq] fp32 is workspace.
 * Dropout on the probabilities (modeling_capsbert.py:404-406, p = attention_probs_dropout_prob; transformer.py:219-229): the forward
 * call draws the keep decisions and writes them to keep_mask as lane masks (shg_attention_keep_mask_bytes(B, H, Sq, Sk) bytes,
 * 128-byte aligned, caller-owned; layout in csrc/attention.hip); the backward call of the same attention reads them instead of
 * re-drawing.  keep_mask is required when p_drop > 0 and ignored (may be NULL) otherwise.
 * dbias_q / dbias_k / dbias_v (backward; each fp32 [H * 64] or NULL): the column sums of dq / dk / dv over all rows are ADDED to
 * them (fp32 atomics) - the bias gradients of the query / key / value projections (BertSelfAttention's nn.Linear biases,
 * modeling_capsbert.py:375-380; nn.MultiheadAttention.in_proj_bias), without a separate pass over the gradient buffers.
 */
int64_t shg_attention_keep_mask_bytes(int B, int H, int Sq, int Sk);
int shg_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H,
                      int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride, int64_t k_sstride,
                      int64_t v_bstride, int64_t v_sstride, int mask_kind, const float* mask, float scale,
                      float p_drop, const uint64_t* seed_state, uint64_t stream_id, uint64_t* keep_mask, void* stream);
int shg_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, int dtype, int B, int H,
                      int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride, int64_t k_sstride,
                      int64_t v_bstride, int64_t v_sstride, int64_t dq_bstride, int64_t dq_sstride,
                      int64_t dk_bstride, int64_t dk_sstride, int64_t dv_bstride, int64_t dv_sstride,
                      int mask_kind, const float* mask, float scale, float p_drop, const uint64_t* seed_state,
                      uint64_t stream_id, const uint64_t* keep_mask, float* dbias_q, float* dbias_k, float* dbias_v,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM on the matrix cores:  C[M,N] = A . B (+ bias[N]), fp32 accumulation.
 * Replaces the nn.Linear GEMMs of the path (modeling_capsbert.py:373-375, :427, :466, :481;
 * transformer.py:192-196) and their backward.
 *   Rows are read in 16-byte chunks: lda/ldb must be multiples of the chunk and cover the contiguous extent
 *   rounded up to it; padding elements must be finite.
 *   a_kmajor = 1: A stored [M, K] (lda = row stride);  0: A stored [K, M]
 *   b_kmajor = 1: B stored [N, K] (ldb = row stride);  0: B stored [K, N]
 *   C (dtype_c: SHG_F32 or SHG_BF16) row stride ldc; accumulate != 0 adds into C (the sum is formed in fp32
 *   and rounded once: weight gradients into the fp32 arena, input gradients onto a residual gradient).
 *   Weight-gradient shapes (both operands contraction-strided, accumulate) with few output tiles are
 *   split along K over several workgroups whose partial sums are added with fp32 atomics.
 *   forward  y = x W^T : a_kmajor=1 (x), b_kmajor=1 (W [N,K])
 *   dgrad   dx = dy W  : a_kmajor=1 (dy [M,N]), b_kmajor=0 (W [N,K] read as [Kred=N][K])
 *   wgrad   dW = dy^T x: a_kmajor=0 (dy [M,N] as [Kred=M][N]), b_kmajor=0 (x [M,K] as [Kred=M][K])
 */
int shg_gemm(const void* a, const void* b, void* c, const float* bias, int dtype_ab, int dtype_c, int64_t M,
             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
             int accumulate, void* stream);
/* C [M, N] (+)= sum over n_seg segments of A_s [M, seg_k] . B_s [seg_k, N] in ONE launch: A_s = a + s * a_seg_stride (elements; rows
 * of lda, the contraction index contiguous), B_s = b + s * b_seg_stride (rows = contraction index, ldb).  The gradient of the
 * decoders w.r.t. their memory (transformer.py:212-233 in reverse: every layer's cross-attention adds dK/dV . W_kv) as one K = layers x
 * 2 H contraction instead of one accumulating GEMM per layer (one rounding of the sum to the output type instead of one per layer). */
int shg_gemm_kseg(const void* a, const void* b, void* c, int dtype, int64_t M, int64_t N, int64_t seg_k, int n_seg, int64_t lda,
                  int64_t ldb, int64_t ldc, int64_t a_seg_stride, int64_t b_seg_stride, int accumulate, void* stream);
/* Weight gradients of n nn.Linear layers in as few launches as possible:  gw[n_out, n_in] += dy[rows, n_out]^T . x[rows, n_in]
 * (fp32 gradient, row stride n_in; dy / x row strides ldy / ldx).  Runs of bf16 problems with rows % 64 == 0 go out as ONE
 * grid of 256 x 256 tiles over all of them (a decoder layer's eight weight gradients are 9-24 tiles each: alone none fills a
 * sixth of the chip; split along the rows with fp32 atomics only when even the whole group is small); anything else falls
 * back to shg_gemm per problem.  Replaces the dW part of the backward of modeling_capsbert.py:373-375, :427, :466, :481 and
 * transformer.py:192-196. */
typedef struct shg_wgrad_problem {
    const void* dy;
    const void* x;
    float* gw;
    int64_t rows, n_out, n_in, ldy, ldx;
} shg_wgrad_problem_t;
int shg_wgrad_group(const shg_wgrad_problem_t* problems, int n, int dtype, void* stream);

/* C = act(A . B + bias) with the activation in the epilogue and, when `pre` is given, the pre-activation
 * values A . B + bias written to pre [M, N] (row stride N, dtype_c) for the backward pass: BertIntermediate's
 * Linear + erf-GELU (modeling_capsbert.py:472-475, :127-133) and the heads' Linear + GELU (agqa_model.py:105-110)
 * as one kernel.  act: SHG_ACT_*.  p_drop > 0: dropout on the activation's output with the masks of shg_bias_act_fwd
 * (seed_state / stream_id as there): the decoder's linear1 + ReLU + dropout (transformer.py:230).  shg_gemm_dact takes the
 * same three arguments for the matching backward (mask applied to the incoming gradient, then act'). */
int shg_gemm_act(const void* a, const void* b, void* c, const float* bias, int dtype_ab, int dtype_c, int64_t M,
                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor, int act,
                 void* pre, float p_drop, const uint64_t* seed_state, uint64_t stream_id, void* stream);

/* Input gradient through Linear + activation in one kernel:  dx[M,N] = (dy[M,K] . W[K,N]) * act'(pre[M,N]) and
 * dbias[n] += sum_m dx[m,n] (fp32 atomics; dbias may be null).  dy row stride lda, W stored [K, N] with row stride
 * ldb (the forward weight of the FOLLOWING Linear, [out = K, in = N]), pre contiguous [M, N]: the backward of
 * BertIntermediate's GELU (modeling_capsbert.py:472-475) folded into BertOutput.dense's input-gradient GEMM. */
int shg_gemm_dact(const void* dy, const void* w, void* dx, const void* pre, float* dbias, int dtype, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, float p_drop, const uint64_t* seed_state,
                  uint64_t stream_id, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv3d with kernel (5,3,3), "valid" in T, zero padding 1 in H and W, as an implicit GEMM over a
 * channels-last input.  Replaces ZeroPad2d(1) -> nn.Conv3d(k=(5,3,3)) -> GeLU of
 * VisualFeatEncoder (modeling_capsbert.py:991-996).
 *   x  [B, T, H+2, W+2, Cin]   (dtype) channels-last, spatially pre-padded with zeros
 *   w  [Cout, 5, 3, 3, Cin]    (dtype) = the reference weight [Cout,Cin,5,3,3] permuted
 *   y  [B, T-4, H, W, Cout]    (dtype_y), y = act(conv + bias); if pad_out != 0 y is written into a
 *      spatially padded [B, T-4, H+2, W+2, Cout] buffer (border untouched, must be pre-zeroed);
 *      y_pre (optional) [B, T-4, H, W, Cout] receives conv + bias before the activation (needed by backward)
 *   wgrad: dw [Cout,5,3,3,Cin] fp32 (+= when accumulate), dy [B,T-4,H,W,Cout] (dtype)
 *   dgrad: dx [B,Tp-4,H,W,Cin] from dy_padded [B,Tp,H+2,W+2,Cout] = dy zero-padded by 4 in T and 1 in H/W
 *          (Tp = T_out + 8): the forward gather over dy_padded with the weight read flipped and
 *          "contraction strided" straight from w [Cout,5,3,3,Cin] - no transposed weight copy.
 *   workspace: shg_conv3d_k533_workspace_bytes(B,T,H,W) bytes, filled once per shape by
 *   shg_conv3d_k533_prepare (gather tables: position of every output row in the padded tensors).
 */
int64_t shg_conv3d_k533_workspace_bytes(int B, int T, int H, int W);
int shg_conv3d_k533_prepare(void* workspace, int B, int T, int H, int W, void* stream);
int shg_conv3d_k533_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int B, int T, int H,
                        int W, int Cin, int Cout, int act, int pad_out, void* y_pre, const void* workspace,
                        void* streamk_workspace, void* stream);
int shg_conv3d_k533_wgrad(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W,
                          int Cin, int Cout, int accumulate, const void* workspace, void* stream);
/* the same for the output channels [c0, c0 + cn) only (multiples of 8): rows c0.. of dw, columns c0.. of dy.  A data-parallel
 * step issues conv1's weight gradient (283 MB of fp32 gradients, the last kernel of backward) as two such launches, so that the
 * all-reduce of the first two thirds runs under the last third instead of behind everything. */
int shg_conv3d_k533_wgrad_slice(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin,
                                int Cout, int c0, int cn, int accumulate, const void* workspace, void* stream);
/* OVERWRITE form for a gradient that has exactly one writer per step: rows [c0, c0 + cn) of dw are SET (they need not be zero
 * beforehand - the optimiser then skips zeroing them, 4 bytes per parameter less in its sweep) and the sum of their squares is
 * ADDED to *sumsq (fp64, device) - the convolution's share of clip_grad_norm_'s global norm (agqaHGQA.py:391) without a second
 * pass over 283 MB: on the 8-phase kernel it comes out of the accumulators (whole 256 x 256 tiles), otherwise a pass over the
 * finished rows follows.  cn a multiple of 8; the fused form needs cn % 256 == 0. */
int shg_conv3d_k533_wgrad_sumsq(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin,
                                int Cout, int c0, int cn, double* sumsq, const void* workspace, void* stream);
int shg_conv3d_k533_dgrad(const void* dy_padded, const void* w, void* dx, int dtype, int B, int Tp, int H, int W,
                          int Cin, int Cout, const void* workspace, void* stream);
/* Row orders (round 3).  The gather tables make the order of the GEMM rows (output positions) a free choice; row_order 1 is
 * POSITION-MAJOR: row = ((h W + w) B + b) To + to instead of ((b To + to) H + h) W + w.  With B To a multiple of 64 every
 * spatial position then owns whole K-tiles of the weight gradient's contraction, and a tile (one kernel tap) skips the positions
 * where its tap reads the zero border (Conv3d padding (0, 1, 1), modeling_capsbert.py:560-566: 18 % of the products of a 3 x 3
 * window on a 7 x 7 grid are with zeros) - same sums, bit for bit, in fewer K-tiles.
 *   shg_conv3d_k533_workspace_bytes_ex / _prepare_ex: tables for a row order; order 1 appends two tables after the two of
 *     order 0 (each ((M * 4 + 255) / 256) * 256 bytes): std2row[m] = position-major row of standard row m, row2std = its inverse.
 *   shg_conv3d_k533_fwd takes either workspace (its dense outputs y_pre / y then have that row order; pad_out output is a layout,
 *     not an order).  shg_conv3d_k533_dgrad_rows: row m of dx is written at dx_rows[m] (the NEXT layer's std2row: its input
 *     gradient arrives in the order its weight gradient contracts over); row_order 2 = FRAME-MAJOR tables (row = ((to B + b) H + h)
 *     W + w, workspace_bytes_ex / prepare_ex with 2): dy is padded by four frames, so 20 of the 60 (output frame, kt) pairs read
 *     padding - a tile keeps the temporal taps that read data for any of its frames (needs streamk_workspace: the weighted plan
 *     balances the tiles); NULL tables / row_order 0 / NULL workspace = shg_conv3d_k533_dgrad.
 *   shg_conv3d_k533_wgrad_ex: the general weight gradient - slice [c0, c0 + cn), accumulate or overwrite, optional fused sum of
 *     squares (only with accumulate = 0), row order of x's table / dy's rows. */
/* shg_conv3d_k533_fwd with row tables: the pre-activation row m is written at row pre_rows[m], the dense output (pad_out = 0) row m
 * at row y_rows[m] (NULL: row m) - e.g. a forward in standard row order that leaves y_pre in the order the backward works in.
 * row_order = the order of the workspace's tables; with 1 (position-major) a tile leaves out the taps that read only the zero
 * border for all of its rows, and the stream-K launch balances the tiles' different lengths with its weighted plan. */
int shg_conv3d_k533_fwd_rows(const void* x, const void* w, const float* bias, void* y, int dtype, int B, int T, int H,
                             int W, int Cin, int Cout, int act, int pad_out, void* y_pre, const int32_t* pre_rows,
                             const int32_t* y_rows, int row_order, const void* workspace, void* streamk_workspace, void* stream);
int64_t shg_conv3d_k533_workspace_bytes_ex(int B, int T, int H, int W, int row_order);
int shg_conv3d_k533_prepare_ex(void* workspace, int B, int T, int H, int W, int row_order, void* stream);
int shg_conv3d_k533_dgrad_rows(const void* dy_padded, const void* w, void* dx, int dtype, int B, int Tp, int H, int W,
                               int Cin, int Cout, const int32_t* dx_rows, int row_order, const void* workspace,
                               void* streamk_workspace, void* stream);
int shg_conv3d_k533_wgrad_ex(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin,
                             int Cout, int c0, int cn, int accumulate, double* sumsq, int row_order, const void* workspace,
                             void* stream);
/* NCDHW fp32 features -> channels-last, spatially zero-padded (dtype) : [B,C,T,H,W] -> [B,T,H+2,W+2,C] */
int shg_ncdhw_to_padded_cl(const float* x, void* y, int dtype, int B, int C, int T, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser on a flat parameter arena.
 * Replaces nn.utils.clip_grad_norm_(params, max_norm) + BertAdam.step (tasks/agqaHGQA.py:391-392;
 * lxrt/optimization.py:101-180): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * u = m/(sqrt(v)+eps) + wd*p; p -= lr * sched(step/t_total) * u, no bias correction, where g is the
 * gradient scaled by clip = min(max_norm/(norm+1e-6), 1).
 *   shg_sumsq: partial[blocks] -> norm_sq[0] (fp32 accumulate in fp64 inside) of grad[0..n)
 *   state: step_state[0] = number of updates already applied (int64, device); incremented by the kernel.
 *   shadow (may be NULL): bf16 copy of the updated parameters, written in the same pass.
 *   bump_step: bit 0 = increment step_state afterwards; bit 1 = also zero the gradient in this pass (the
 *   optimizer.zero_grad() of the next step, agqaHGQA.py:387, without a separate sweep over the arena).
 */
int shg_sumsq(const float* x, int64_t n, double* partial, int n_partial, float* out_norm, void* stream);
/* the same in two stages, for a norm over several ranges of the arena plus sums other kernels have already accumulated:
 * shg_sumsq_partial fills partial[0 .. n_partial) for one range (any stream); shg_sumsq_final: out_norm[0] =
 * sqrt(sum of partial[0 .. n_partial) + (extra ? extra[0] : 0)) and resets extra[0] to 0 for the next step. */
int shg_sumsq_partial(const float* x, int64_t n, double* partial, int n_partial, void* stream);
int shg_sumsq_final(const double* partial, int n_partial, double* extra, float* out_norm, void* stream);
int shg_bertadam_arena(float* param, float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                       const float* grad_norm, float max_norm, float lr, float warmup, int64_t t_total,
                       float b1, float b2, float eps, float weight_decay, int64_t* step_state, int bump_step,
                       void* stream);
/* *p += delta (single-thread kernel; keeps step / dropout counters on the device so a captured graph advances them) */
int shg_add_i64(int64_t* p, int64_t delta, void* stream);
/* dst(dtype) = cast(src fp32), n elements */
int shg_cast_f32(const float* src, void* dst, int dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small elementwise helpers of the decoder executor (transformer.py:213-226: `tgt + query_pos`, and its backward).
 *   shg_add:             out[i] = a[i] + b[i]                                   (n elements of dtype, n % 8 == 0)
 *   shg_add2_accumulate: acc1[i] += c[i];  acc2[i] = init2 ? c[i] : acc2[i] + c[i]   (acc1 may be NULL)
 * Sums are formed in fp32 and rounded once. */
int shg_add(const void* a, const void* b, void* out, int dtype, int64_t n, void* stream);
/* Token assembly of VisualFeatEncoder (modeling_capsbert.py:1053-1072): out [B, n_tok, C] (dtype) with
 * out[b, 0] = cls + pos[0] and out[b, 1 + t] = tok[b, t] + pos[1 + t];  tok [B, n_tok - 1, C] (dtype), cls [C], pos [>= n_tok, C] fp32. */
int shg_tokens_assemble(const void* tok, const float* cls, const float* pos, void* out, int dtype, int B, int n_tok, int C,
                        void* stream);
int shg_add2_accumulate(void* acc1, void* acc2, const void* c, int init2, int dtype, int64_t n, void* stream);

/* shg_bias_act_drop_res_ln_fwd with a second output y_pos = y + pos (pos, y_pos [rows, cols] (dtype), both NULL or both set):
 * the decoder feeds `tgt + query_pos` into the next attention's projections (transformer.py:216, :222), so the LayerNorm that
 * produces tgt writes the sum in the same pass. */
int shg_bias_act_drop_res_ln_fwd_pos(const void* x, const float* bias, const void* residual, const float* gamma,
                                     const float* beta, void* y, void* z_out, float* mean, float* rstd, const void* pos,
                                     void* y_pos, int dtype, int64_t rows, int cols, int act, float eps, float p_drop,
                                     const uint64_t* seed_state, uint64_t stream_id, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sub-layer executor: ONE call enqueues every kernel of a transformer sub-layer (6-16 launches), so that the host
 * issues ~100 calls per training step instead of ~1 150 (the decoder / hyper-graph segments of the step run at the
 * host's pace otherwise).  The arithmetic is exactly the sequence of entry points above; nothing here computes.
 *
 *   shg_exec_t     caller-owned handle: a ring of hipEvents used to order the weight-gradient stream behind the main
 *                  stream (create once per device; destroy at exit).  The library keeps no global state for it.
 *   shg_run_t      per-call context: streams, dropout seed, dtype.
 *   shg_linear_t   one nn.Linear: operand copy of the weight in the compute dtype (the bf16 shadow or the fp32 master),
 *                  fp32 bias, and the fp32 gradient slices (NULL: parameter does not train).
 *   shg_norm_t     one nn.LayerNorm.
 * Activations are row-major [rows, features] in `dtype`, contiguous, 16-byte aligned.  `saved` (forward -> backward) and
 * `scratch` (backward temporaries, also read by the weight-gradient stream after the call returns: give every call its
 * own) are caller-owned device buffers of shg_*_saved_bytes / shg_*_scratch_bytes bytes.
 * Weight / bias / LayerNorm gradients are ACCUMULATED into the gradient slices (on run->wgrad_stream when set).
 * Dropout call sites: each sub-layer uses stream ids sid and sid + 1; a decoder layer uses sid .. sid + 5.
 */
typedef struct shg_exec shg_exec_t;
/* sizeof() of the structs below as this library was compiled, for bindings that mirror them:
 * which = 0 shg_run_t, 1 shg_linear_t, 2 shg_norm_t, 3 shg_attn_sublayer_t, 4 shg_ffn_sublayer_t, 5 shg_decoder_layer_t */
int shg_abi_sizeof(int which);
shg_exec_t* shg_exec_create(int n_events);
void shg_exec_destroy(shg_exec_t* ex);
/* Deferred weight gradients (shg_run_t.defer_wgrad): number of 256 x 256 output tiles queued so far, and the flush: orders
 * `wgrad_stream` behind every stream that produced a queued operand, then issues the queue with shg_wgrad_group and the
 * bias column sums behind it.  Returns 0 with an empty queue. */
int64_t shg_exec_pending_tiles(const shg_exec_t* ex);
int shg_exec_flush_wgrads(shg_exec_t* ex, int dtype, void* wgrad_stream);

typedef struct shg_run {
    int32_t dtype;            /* SHG_F32 / SHG_BF16: activations and operands */
    int32_t training;         /* 0: every dropout probability is taken as 0 */
    void* stream;             /* hipStream_t of the dependent chain */
    void* wgrad_stream;       /* hipStream_t for weight gradients, or NULL: inline on `stream` */
    shg_exec_t* exec;         /* needed when wgrad_stream is set */
    const uint64_t* seed_state;
    int32_t defer_wgrad;      /* != 0 (needs wgrad_stream): weight / bias gradients are queued on `exec` instead of launched; the
                                 caller keeps their operands alive and calls shg_exec_flush_wgrads (grouped launches) */
    int32_t kv_ahead;         /* != 0 (needs wgrad_stream): shg_decoder_fwd issues the key / value projections of `memory` for ALL
                                 layers on wgrad_stream (idle during a forward pass) behind one event, and the chain waits per layer */
} shg_run_t;

typedef struct shg_linear {
    const void* w;            /* [out, in] row-major, compute dtype */
    const float* bias;        /* [out] or NULL */
    float* gw;                /* [out, in] fp32 gradient or NULL */
    float* gb;                /* [out] fp32 gradient or NULL */
} shg_linear_t;

typedef struct shg_norm {
    const float* gamma;
    const float* beta;
    float* g_gamma;           /* NULL: does not train */
    float* g_beta;
    float eps;
    float pad_;
} shg_norm_t;

#define SHG_ATTN_SELF 0       /* q,k,v = W_a x (W_a [3H,H]);                    BertSelfattLayer, modeling_capsbert.py:450-460 */
#define SHG_ATTN_CROSS 1      /* q = W_a x (W_a [H,H]); k,v = W_b mem (W_b [2H,H]);  BertCrossattLayer, modeling_capsbert.py:438-447 */
#define SHG_ATTN_DEC_SELF 2   /* q,k = W_a (x+pos) (W_a [2H,H]); v = W_b x (W_b [H,H]);  transformer.py:216-221 */
#define SHG_ATTN_DEC_CROSS 3  /* q = W_a (x+pos); k,v = W_b mem;                  transformer.py:222-227 */

/* y = LayerNorm(x + dropout(W_o attention(...) + b_o))   (BertAttention + BertAttOutput, modeling_capsbert.py:384-435;
 * nn.MultiheadAttention + dropout + norm of the DETR decoder layer, transformer.py:216-227) */
typedef struct shg_attn_sublayer {
    int32_t mode;             /* SHG_ATTN_* */
    int32_t heads;            /* hidden = heads * 64 */
    int32_t mask_kind;        /* SHG_MASK_* */
    int32_t pad_;
    float scale, p_attn, p_out, pad2_;
    const float* mask;        /* per mask_kind */
    shg_linear_t a, b, o;
    shg_norm_t ln;
} shg_attn_sublayer_t;

/* y = LayerNorm(x + dropout(W_2 dropout(act(W_1 x + b_1)) + b_2))   (BertIntermediate + BertOutput,
 * modeling_capsbert.py:463-489; linear1 / ReLU / dropout / linear2 / dropout3 / norm3, transformer.py:230-232) */
typedef struct shg_ffn_sublayer {
    int32_t act;              /* SHG_ACT_* */
    int32_t pad_;
    float p_inner, p_out;
    shg_linear_t l1, l2;      /* l1 [F, H], l2 [H, F] */
    shg_norm_t ln;
} shg_ffn_sublayer_t;

typedef struct shg_decoder_layer {
    shg_attn_sublayer_t self_attn, cross_attn;     /* modes SHG_ATTN_DEC_SELF / SHG_ATTN_DEC_CROSS */
    shg_ffn_sublayer_t ffn;
} shg_decoder_layer_t;

/* x [B*Sq, H]; xpos = x + pos (decoder modes, else NULL); mem [B*Sk, H] (cross modes, else NULL; Sk = Sq then);
 * y [B*Sq, H]; pos / y_pos: optional second output y + pos (both NULL or both set). */
int64_t shg_attn_sublayer_saved_bytes(int mode, int dtype, int B, int Sq, int Sk, int heads);
int64_t shg_attn_sublayer_scratch_bytes(int mode, int dtype, int B, int Sq, int Sk, int heads);
int shg_attn_sublayer_fwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x,
                          const void* xpos, const void* mem, void* y, const void* pos, void* y_pos, void* saved,
                          uint64_t sid);
/* dy [B*Sq, H] -> dx (gradient w.r.t. x: residual path + projections; NULL: not needed), dxpos (gradient w.r.t. xpos,
 * decoder modes; NULL: not needed), dmem (cross modes; NULL: not needed; dmem_accumulate != 0 adds into it). */
int shg_attn_sublayer_bwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x,
                          const void* xpos, const void* mem, const void* saved, const void* dy, void* dx, void* dxpos,
                          void* dmem, int dmem_accumulate, void* scratch, uint64_t sid);

int64_t shg_ffn_sublayer_saved_bytes(int dtype, int64_t rows, int H, int F);
int64_t shg_ffn_sublayer_scratch_bytes(int dtype, int64_t rows, int H, int F);
int shg_ffn_sublayer_fwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x, void* y,
                         const void* pos, void* y_pos, void* saved, uint64_t sid);
int shg_ffn_sublayer_bwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x,
                         const void* saved, const void* dy, void* dx, void* scratch, uint64_t sid);

/* n_layers DETR decoder layers (post-norm, transformer.py:86-124, :212-233) in one call:
 *   tgt [B*Q, H] or NULL (= zeros, agqa_model.py:234), memory [B*S, H], query_pos [B*Q, H]; the block-causal target mask
 *   (entry.py:114-121) sits in every layer's self_attn.mask; out [B*Q, H] = the last layer's output.
 * bwd: d_out -> d_tgt (NULL when tgt was NULL or not needed), d_query_pos, d_memory (both written, not accumulated). */
int64_t shg_decoder_saved_bytes(int n_layers, int dtype, int B, int Q, int S, int heads, int F);
int64_t shg_decoder_scratch_bytes(int n_layers, int dtype, int B, int Q, int S, int heads, int F);
int shg_decoder_fwd(const shg_decoder_layer_t* layers, int n_layers, const shg_run_t* R, int B, int Q, int S, int F,
                    const void* tgt, const void* memory, const void* query_pos, void* out, void* saved, uint64_t sid);
int shg_decoder_bwd(const shg_decoder_layer_t* layers, int n_layers, const shg_run_t* R, int B, int Q, int S, int F,
                    const void* tgt, const void* memory, const void* query_pos, const void* saved, const void* d_out,
                    void* d_tgt, void* d_query_pos, void* d_memory, void* scratch, uint64_t sid);

#ifdef __cplusplus
}
#endif
#endif /* SHG_VQA_H */
