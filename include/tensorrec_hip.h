/*
 * include/tensorrec_hip.h -- C ABI of libtensorrec_hip.so (gfx950 / MI355X).
 *
 * The reference (jfkirk/tensorrec) has no FFI of its own: its hot path is a chain of TensorFlow-1.x op calls made
 * from Python.  Each entry point below replaces one such op chain; the reference call site it stands in for is
 * cited as tensorrec/<file>:<line>.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name says otherwise; the library never allocates, frees,
 *     synchronises or keeps a pointer after returning; `stream` is a hipStream_t (NULL = default stream) and
 *     all work is enqueued on it;
 *   - matrices are row-major fp32 unless stated; indices int32, row pointers int64;
 *   - return value: TREC_OK (0) or an error code; trec_last_error() gives the message (thread-local);
 *     nothing throws across this boundary.
 */
#ifndef TENSORREC_HIP_H
#define TENSORREC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TREC_OK 0
#define TREC_ERR_INVALID 1
#define TREC_ERR_LAUNCH 2
#define TREC_ERR_UNSUPPORTED 3

/* dtype of the score-GEMM operands */
#define TREC_DTYPE_F32 0   /* exact fp32 (v_mfma_f32_32x32x2_f32): k-ordered fmaf chain */
#define TREC_DTYPE_BF16 1  /* bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16)  */
/* prediction mode */
#define TREC_MODE_DOT 0        /* also cosine: operands are row-normalised first          */
#define TREC_MODE_EUCLIDEAN 1

int trec_abi_version(void);
const char* trec_last_error(void);
int trec_device_cu_count(void);
/* benchmark-only switches between kernel variants (e.g. "spmm_rows", "spmm_nt"); never needed for correctness */
int trec_set_tuning(const char* name, int32_t value);
int trec_get_tuning(const char* name, int dflt);

/* ---- K1: sparse features x dense weights ------------------------------------------------------------------
 * tf.sparse_tensor_dense_matmul: representation_graphs.py:40 (Linear), :119 (ReLU layer 1),
 * recommendation_graphs.py:15 (biases).  CSR operand (indptr[n_rows+1], indices[nnz], values[nnz]); if val_perm is
 * non-NULL the value of entry j is values[val_perm[j]] (transposed operand of the backward pass:
 * dW = X^T . dOut is this same call on the transposed CSR).  epilogue: 0 none | 1 row L2-normalise
 * (representation_graphs.py:57; writes 1/norm to out_inv_norm if non-NULL) | 2 add col_bias[d] then ReLU
 * (representation_graphs.py:119-120) | 3 none, and out_inv_norm[r] = sum of the row's values (the bias gradient that
 * accompanies a gradient gather).  accumulate != 0: out += result (and out_inv_norm += for epilogue 3); epilogue 0 / 3 only. */
int trec_spmm_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* val_perm,
                  int64_t n_rows, int64_t nnz, const float* W, int32_t d, const float* col_bias, int32_t epilogue,
                  int32_t accumulate, float* out, float* out_inv_norm, void* stream);
/* K1 for matrices with exactly ONE non-zero per row (identity / indicator features -- every user or item is its own
 * feature, the case of BASELINE configs[2]): out[r] = values[r] * W[indices[r]].  The row pointer is the identity and is
 * not read.  d % 4 == 0, d <= 1024; indices / values 16-byte aligned. */
int trec_spmm_one_per_row(const int32_t* indices, const float* values, int64_t n_rows, const float* W, int32_t d, float* out,
                          void* stream);
/* K1 with the operand of the filtered top-k (K2f) as its epilogue: out = X . W in fp32 AND its bf16 image [n_rows, d],
 * row_stats [n_rows][2] = {||row||, ||row - bf16(row)||} and (gstats non-NULL, zero-initialised) their running maxima --
 * what trec_score_prep_filter computes in a separate pass.  d = 32 / 64 / 128 / 256 (no padding), dot products (no
 * normalisation).  trec_absmax: running maximum of |x| into *out (the |bias| term of the bound, gstats[2]). */
int trec_spmm_csr_filter(const int64_t* indptr, const int32_t* indices, const float* values, int64_t n_rows, int64_t nnz,
                         const float* W, int32_t d, float* out, void* out_bf16, float* row_stats, float* gstats,
                         void* stream);
int trec_absmax(const float* x, int64_t n, float* out, void* stream);
/* The same gather over a CSR with packed entries int2 {column, value bits} (trec_group_pairs_by_item, packed mode);
 * epilogue 0 or 3 (out_rowsum[r] = sum of the row's values); accumulate != 0: out += (and out_rowsum +=).            */
int trec_spmm_csr_packed(const int64_t* indptr, const void* entries, int64_t n_rows, const float* W, int32_t d,
                         int32_t epilogue, int32_t accumulate, float* out, float* out_rowsum, void* stream);
/* project_biases, recommendation_graphs.py:4-19: out[r] = sum_j X[r,j] * beta[j] (fmaf chain in CSR order).
 * beta == NULL: out[r] = sum of the row's (permuted) values -- the bias gradients of the serial predictions, where
 * rows are a user's / an item's pairs; 16 lanes share a row (coalesced), indices are not read.                  */
int trec_spmv_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* val_perm,
                  int64_t n_rows, const float* beta, float* out, void* stream);
/* K1 for skewed row lengths (csrc/spmm_split.hip): the gradient gathers of the fit step run on transposed structures
 * whose rows are items / feature columns, and real interaction data is Zipf-shaped (one MovieLens-20M item holds
 * ~136,000 pairs).  Rows longer than the split threshold (2048 non-zeros) are cut into chunks that any subgroup of the
 * chip can take; a row's chunk partials are added in chunk order, so the result is deterministic.  Rows at or below the
 * threshold are the CSR-order fmaf chain of trec_spmm_csr, bit for bit.  Either (indices, values[, val_perm]) or
 * packed_entries (int2 {column, value bits}) describes the non-zeros.  own (nullable, [n_rows, d]): gather
 * (own[row,:] - W[col,:]) instead of W[col,:] -- the Euclidean pair gradient, prediction_graphs.py:105-117
 * differentiated, with the coefficients of trec_pair_euclid_coef as values.  out_rowsum (nullable): sums of the rows'
 * values.  accumulate != 0: out += (and out_rowsum +=).  workspace: trec_csr_split_workspace_bytes(nnz, d) bytes.
 * d: a multiple of 4 up to 1024 (float4 lanes), or any width up to 256 (single-column lanes).                     */
int64_t trec_csr_split_workspace_bytes(int64_t nnz, int32_t d);
int trec_spmm_csr_split(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* val_perm,
                        const void* packed_entries, int64_t n_rows, int64_t nnz, const float* W, int32_t d,
                        const float* own, int32_t accumulate, float* out, float* out_rowsum, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* trec_spmv_csr with the same treatment of long rows (workspace: trec_csr_split_workspace_bytes(nnz, 1)) */
int trec_spmv_csr_split(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* val_perm,
                        int64_t n_rows, int64_t nnz, const float* beta, float* out, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* tf.sparse_tensor_to_dense, representation_graphs.py:74 (FeaturePassThrough) */
int trec_csr_to_dense(const int64_t* indptr, const int32_t* indices, const float* values, int64_t n_rows,
                      int32_t n_cols, float* out, void* stream);
/* tf.nn.l2_normalize(x, 1): prediction_graphs.py:68-69, recommendation_graphs.py:119-120; and its gradient */
int trec_row_l2norm_fwd(const float* x, int64_t n_rows, int32_t d, float* y, float* inv_norm, void* stream);
int trec_row_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int64_t n_rows, int32_t d, float* dx,
                        void* stream);
/* gradient of tf.nn.relu (representation_graphs.py:119) and of the broadcast bias add (column sums) */
int trec_relu_bwd(const float* out, const float* dout, int64_t n, float* dpre, void* stream);
/* n_slices > 1: the rows are summed as n_slices slices in parallel (workspace: n_slices * d floats) and the slice sums
 * added in slice order -- deterministic; n_slices <= 1: one pass, d / 64 workgroups (short matrices) */
int trec_colsum(const float* x, int64_t n_rows, int32_t d, float* out, float* workspace, int32_t n_slices, void* stream);
/* tf.matmul(relu, linear_weights), representation_graphs.py:121, and its two gradients:
 * C[M,N] (+)= op(A) . op(B), fp32 on MFMA; trans flags select A^T / B^T (row-major storage, lda/ldb/ldc in elements).
 * splits > 1: K is cut into that many slices computed by separate workgroups (the weight gradient has a small output
 * and K = all users) and added in slice order; workspace: splits * M * N floats (NULL with splits <= 1). */
int trec_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, int32_t accumulate, float* workspace,
                  int32_t splits, void* stream);
/* trec_gemm_f32's contract with both operands split into two bf16 terms (x = hi + lo) and three bf16 MFMAs per product block, fp32
 * accumulation: ~1e-5 relative per product.  For products that ARE gradients (TF's autodiff of tensorrec.py:487-489 on the dense
 * coefficient matrix of the tiled WMRB step: 1e-4 bar), not for values compared with the oracle's fmaf chain. */
int trec_gemm_f32_split_bf16(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, int32_t accumulate, float* workspace,
                             int32_t splits, float* a_colsum_parts, void* stream);
/* (a_colsum_parts, nullable, trans_a only: float [splits][M] -- per K slice the column sums of the stored [K, M] matrix A, taken by
 * the threads that stage A: the separate pass over a 14.8 GB coefficient matrix for its column sums disappears) */

/* ---- K2: user x item score contraction --------------------------------------------------------------------
 * tf.matmul(user_repr, item_repr, transpose_b=True): prediction_graphs.py:50 (DotProduct), :94 (Euclidean),
 * recommendation_graphs.py:121 (relative_cosine); + bias_prediction_dense, recommendation_graphs.py:41.
 * Operands are prepared once by trec_score_prep: [n, kpad] with kpad = trec_score_kpad(d), zero padded, fp32 or bf16,
 * optionally row-normalised (cosine) and/or with the squared row norm emitted (euclidean, prediction_graphs.py:87-90). */
int trec_score_kpad(int32_t d);                                   /* 32 / 64 / 128 / 256, or -1 if d > 256       */
int trec_score_rows_per_workgroup(int32_t dtype, int32_t kpad);
int trec_score_tile_rows(int32_t dtype, int32_t kpad);
int trec_score_topk_capacity(int32_t k);                          /* entries per partial list: 8, 12 or 16; -1 if k > 16 */
int trec_score_prep(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize, int32_t dtype, void* out,
                    float* out_sqnorm, void* stream);
/* predict(): out[u, i] = score(u, i) (+ user_bias[u]) (+ item_bias[i]) -- tensorrec.py:662 */
int trec_score_gemm_store(const void* users, const void* items, int32_t dtype, int32_t kpad, int64_t n_users,
                          int64_t n_items, const float* user_bias, const float* item_bias, int32_t mode,
                          const float* user_sqnorm, const float* item_sqnorm, float* out, int64_t ld_out,
                          int32_t variant, void* stream);
/* fused top-k (first tf.nn.top_k of rank_predictions, recommendation_graphs.py:80, truncated to <= 16): writes
 * n_parts = trec_score_topk_parts(...) sorted partial lists per user, [n_users, n_parts, capacity] values + item
 * indices (index + item_index_base; empty slots = (-inf, -1)), capacity = trec_score_topk_capacity(k); finish with
 * trec_topk_merge.  variant: bit 0 = stage item tiles with global_load_lds (else through registers); bits 1.. =
 * experimental tilings of the bf16 / K=128 / capacity-12 configuration (0 = default).                        */
int trec_score_topk_parts(int32_t dtype, int32_t kpad, int64_t n_items, int32_t n_chunks);
int trec_score_gemm_topk(const void* users, const void* items, int32_t dtype, int32_t kpad, int64_t n_users,
                         int64_t n_items, int32_t item_index_base, const float* user_bias, const float* item_bias,
                         int32_t mode, const float* user_sqnorm, const float* item_sqnorm, int32_t n_chunks,
                         int32_t capacity, float* part_vals, int32_t* part_idx, int32_t variant, void* stream);
/* Two-stage exact top-k (data-independent cost; what predict_top_k uses for large item sets):
 *   1. trec_score_gemm_blockmax: blockmax[s * bm_stride + u] = max exact score of user u over items
 *      [s*sb_rows, (s+1)*sb_rows)  (sb_rows a multiple of 128) -- the score kernel with a branch-free epilogue;
 *   2. trec_topk_select_blocks: per user the k superblocks with the largest maxima, (max desc, index asc) -- they
 *      contain the exact top-k, ties included (proof in csrc/topk2.hip);
 *   3. trec_topk_group_keys + trec_group_pairs_by_item + trec_topk_pad_counts + trec_exclusive_scan_i32 +
 *      trec_topk_fill_groups: (user, slot) pairs grouped by superblock, padded to whole workgroups, operand rows gathered;
 *      trec_score_gemm_topk_grouped: every workgroup re-scores one superblock for its gathered rows (fused lists);
 *   4. trec_topk_merge over the k*2 lists of each user.
 * trec_score_gemm_blockmax, variant bit 5 (32): the caller is a FILTER (K2f / K2c) -- the bf16 maxima only have to obey the
 * error bound, not equal another kernel's scores bit for bit: the v_mfma_f32_16x16x32_bf16 form of the kernel is used
 * (bf16 dot / cosine, kpad 64 / 128; more work per joule at the chip's power cap). */
int trec_score_gemm_blockmax(const void* users, const void* items, int32_t dtype, int32_t kpad, int64_t n_users,
                             int64_t n_items, const float* user_bias, const float* item_bias, int32_t mode,
                             const float* user_sqnorm, const float* item_sqnorm, int32_t sb_rows, int32_t n_chunks,
                             float* blockmax, int64_t bm_stride, int32_t variant, void* stream);
/* tau (nullable) [n_users]: the k-th largest superblock maximum = a floor of the final k-th best score;
 * sel_max (nullable) [k, n_users]: the maxima of the selected superblocks, in the blockmax layout.
 * trec_topk_group_keys: keys[i] = sel[i], or -1 (skipped by trec_group_pairs_by_item) for empty slots and -- when floor [n_users] is given
 * -- for superblocks with sel_max < floor[i / k].  With item shards the floor is the k-th largest superblock maximum
 * over ALL shards: all-gather the ranks' sel_max ([world * k, n_users] IS a blockmax table) and run
 * trec_topk_select_blocks on it; re-scoring then totals ~k superblocks per user over all shards instead of k per shard. */
int trec_topk_select_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k, int32_t* sel,
                            float* sel_max, float* tau, void* stream);
int trec_topk_group_keys(const int32_t* sel, const float* sel_max, const float* floor, int64_t n, int32_t k, int32_t n_sb,
                         int32_t* keys, void* stream);
int trec_topk_pad_counts(const int64_t* indptr_t, int32_t n_sb, int32_t rows_wg, int32_t* cnt_pad, void* stream);
int trec_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* workspace_i64, int64_t* out, void* stream);
int trec_topk_fill_groups(const int64_t* pstart, const int64_t* indptr_t, const int32_t* users_t, const int32_t* perm_t,
                          int32_t n_sb, int32_t rows_wg, int64_t max_rows, const void* users_op, int32_t row_bytes,
                          const float* user_bias, const float* user_sq, const float* user_tau, void* G, float* g_bias,
                          float* g_sq, float* g_tau, int32_t* row_pair, int32_t* rblock_chunk, void* stream);
int trec_score_gemm_topk_grouped(const void* users_g, const void* items, int32_t dtype, int32_t kpad, int64_t n_rows_g,
                                 int64_t n_items, int32_t item_index_base, const float* user_bias_g,
                                 const float* item_bias, int32_t mode, const float* user_sqnorm_g,
                                 const float* item_sqnorm, int32_t sb_rows, const int32_t* rblock_chunk,
                                 const int32_t* row_pair, const float* row_floor, int32_t capacity, float* part_vals,
                                 int32_t* part_idx, int32_t variant, const int32_t* row_index, void* stream);
/* The index form of trec_topk_fill_groups: no operand copy -- row_user[v] = the operand row of grouped row v (-1 =
 * padding), to be passed to trec_score_gemm_topk_grouped as row_index together with the UNgathered users / user_bias /
 * user_sqnorm / row_floor arrays (part_vals may then be NULL: only the item ids are listed). */
int trec_topk_fill_groups_index(const int64_t* pstart, const int64_t* indptr_t, const int32_t* users_t,
                                const int32_t* perm_t, int32_t n_sb, int32_t rows_wg, int64_t max_rows, int32_t* row_user,
                                int32_t* row_pair, int32_t* rblock_chunk, void* stream);

/* trec_topk_select_blocks with a list of k <= 64 entries per user and tau = entry k_tau - 1 (k_tau <= k). */
int trec_topk_select_blocks_ex(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k,
                               int32_t k_tau, int32_t* sel, float* sel_max, float* tau, void* stream);

/* ---- K2f: the exact fp32 top-k through a bf16 filter (csrc/topk_filter.hip) ----------------------------------
 * The reference's contraction is float32 (prediction_graphs.py:49-50, input_utils.py:16) and its order is
 * tf.nn.top_k's (recommendation_graphs.py:80).  Stage 1 runs on bf16 MFMA; a proven bound eps_u >= |bf16 score - fp32
 * score| turns it into a filter (keep every superblock / item within 2 eps_u of the k-th largest superblock maximum),
 * and the survivors are re-scored by the reference's k-ordered fp32 fmaf chain: values and ids are bit-identical to the
 * fp32 MFMA path and to oracle/tr_oracle.c.  Dot and cosine (normalize != 0) scores.
 *   trec_score_prep_filter: repr [n, d] fp32 -> out_bf16 [n, kpad] (what trec_score_prep(dtype bf16) writes),
 *     out_f32 [n, kpad] (nullable; what trec_score_prep(dtype fp32) writes), row_stats [n][2] = {||x||, ||x - bf16(x)||}
 *     of the (normalised) row, and -- gstats non-NULL, zero-initialised by the caller -- running maxima
 *     gstats[0..2] = {max ||x||, max ||x - bf16(x)||, max |bias|} (the item side; item shards all-reduce them with MAX).
 *   trec_topk_filter_floor: floor[u] = tau[u] - 2 eps_u rounded down (tau = the k-th largest superblock maximum from
 *     trec_topk_select_blocks; with item shards the k-th largest over all shards); flag[u] = 1 and *n_flagged += 1 when
 *     the bound is not finite.
 *   trec_topk_collect_blocks: second pass over the superblock maxima -- keys[u * ksel + c] = the c-th superblock (in
 *     index order) whose maximum is >= floor[u], -1 in unused slots, count[u] = min(c, ksel); users with more than ksel
 *     such superblocks are flagged.  keys go straight to trec_group_pairs_by_item (pair = u * ksel + c), then the
 *     grouping entry points above with user_tau = floor, and trec_score_gemm_topk_grouped with variant bit 4 set (lists
 *     independent of each other).
 *   trec_topk_filter_finish: part_idx is the [n_users * ksel * 2, capacity] item-id lists of the grouped pass;
 *     users_f32 / items_f32 the fp32 operands (row strides ld_*, contraction length kdim); writes the exact top-k and
 *     flags users whose lists were full or who had more than 64 survivors.  Flagged users must be re-done by the caller
 *     (ops.score_topk_filtered: a wide second pass -- larger ksel, 16-entry lists, trec_topk_filter_finish_wide -- then the
 *     exact fp32 MFMA path for what is left).  trec_topk_collect_blocks empties the slots of a user it flags (count 0). */
int trec_score_prep_filter(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize, const float* bias,
                           float* out_f32, void* out_bf16, float* row_stats, float* gstats, void* stream);
/* ---- K2q / K2c: the int8 -> bf16 -> fp32 cascade of the exact top-k (csrc/score_blockmax_i8.hip, csrc/topk_cascade.hip;
 * same place in the reference as K2f: prediction_graphs.py:49-50 + recommendation_graphs.py:80) ------------------------
 * The bf16 stage-1 kernel runs at the chip's power limit; the int8 MFMA contracts twice the elements per cycle and is
 * exact on integers.  Operand rows are quantised (users: one scale; items: one scale per superblock, so that a
 * superblock's maximum stays an integer maximum); the quantisation error of every row is measured, which gives a proven
 * bound e(u, s) >= |int8 score - fp32 score| per user and superblock.  Only (superblock, user) pairs whose int8 maximum + e
 * reaches the k-th largest (int8 maximum - e) are re-scored in bf16; the bf16 maxima replace those entries of the table and
 * the bf16 filter above runs unchanged on it (proof: csrc/topk_cascade.hip).
 *   trec_score_prep_i8: repr [n, d] fp32 -> out_q int8 [n, kpad], row_stats [n][2] = {||x||, ||x - scale q||}.
 *     side 0 (users): scales[0] = min(clip_sigmas * rms, max |x|) / 127 (clip_sigmas <= 0: nothing clips); workspace 16 B.
 *     side 1 (items): sb_stats [n_sb][4], zero-initialised = {b_s = max |y| / 127 over superblock s (sb_rows rows), max
 *       ||y|| + ||dy||, max ||dy||, max |bias - a b_s bias_q| / a (the bias quantisation error per unit of user scale a)};
 *       with a bias also side 2 (users first).
 *     side 2: bias_q = rint(bias / (scales[0] b_s)), sb_stats[s][3] and gstats[2] = max |bias| (both zeroed by the caller)
 *       for the current user scale; the item rows are quantised once per catalogue.  kpad <= 128.
 *   User scale CLASSES (the default): one scale for all users follows the largest of them and leaves small / heavy-tailed /
 *     sparse rows a handful of levels.  trec_score_row_scale_i8: nat [n] = the scale each row wants (the best of max |x| / 127
 *     and a half / a quarter of it by the error norm each leaves), gmax [1] (zeroed) = their maximum.  The host rounds them UP to
 *     a geometric ladder of classes, sorts the users by class and gives every int8 workgroup
 *     (trec_score_blockmax_i8_rows_per_workgroup rows) the scale of its first row.  trec_score_prep_i8_users: row r quantised
 *     with wg_scale[r / wg_rows].  trec_score_bias_i8_classes: bias_q [n_classes][n_items] for the classes in use (the MFMA's
 *     C operand is in units of ITS users' scale product), sb_stats[s][3] = the maximum over them.
 *   trec_score_user_err_i8: r_err [n_users][4] = {||x||, ||x - a q||, ck (|b_u| + gstats[2]), a_u}, the users' part of e(u, s)
 *     (a_u = wg_scale[u / wg_rows], or scales[0] without classes).
 *   trec_score_gemm_blockmax_i8: blockmax[s * bm_stride + u] = a_u * b_s * max over the items of superblock s of
 *     (q_u . q_i + bias_q[class][i]) + user_bias[u]; exact integer arithmetic on v_mfma_i32_16x16x64_i8.  kpad 64 / 128.
 *     wg_scale / wg_class [workgroups of the launch] (nullable: scales[0], one bias table): the users' scale and class.
 *     With user_err / chunk_top / top_k (10 or 16): chunk_top [n_chunks_eff * top_k][bm_stride] = per chunk of superblocks
 *     and user the top_k largest lower bounds (sorted, -inf padded); trec_topk_select_blocks over it gives tau.
 *   trec_topk_rows_count / trec_topk_rows_fill: the pairs with table[s][u] + e(u, s) >= thr[u] (two floats below),
 *     grouped by superblock by a row-wise stream compaction (the table is superblock-major: no sort).  count: block_off
 *     [n_sb * trec_topk_rows_user_blocks(n_users)], row_total / row_pad [n_sb], pstart int64 [n_sb + 1] -- pstart[n_sb] is the
 *     number of resident rows (a multiple of 512); status int64[2] = {pstart[n_sb], 1 if it exceeds cap_rows}.  fill:
 *     row_user [cap_rows] (user ids ascending inside a superblock, -1 = padding), rblock_chunk [cap_rows / 512] (-1 for
 *     the idle workgroups beyond the kept pairs; all of them after an overflow, which the caller reads from status when
 *     the pipeline has drained and answers with the dense bf16 stage 1).  No host round trip between the stages.
 *   trec_topk_rows_collect: the same compaction in ONE pass -- row_user [n_sb][rcap] with a fixed capacity per superblock
 *     (rcap % 512 == 0), slots handed out by one atomicAdd per (workgroup, row) on row_count [n_sb] (zero-initialised;
 *     ends as the number of users kept, possibly above rcap: status[1]); the order of a superblock's users follows the
 *     atomics (no result depends on it).
 *   trec_score_gemm_blockmax_grouped: the hand-scheduled bf16 stage-1 kernel over those pairs only; workgroup w re-scores
 *     superblock rblock_chunk[w] for its 512 rows and writes blockmax[rblock_chunk[w] * bm_stride + row_user[r]];
 *     wgs_per_row > 0: the layout of trec_topk_rows_collect (rblock_chunk = row_count, superblock = w / wgs_per_row,
 *     n_rows_g = n_sb * wgs_per_row * 512). */
int trec_score_prep_i8(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t side, float clip_sigmas,
                       int32_t sb_rows, const float* bias, float* scales, double* workspace, void* out_q,
                       float* row_stats, int32_t* bias_q, float* sb_stats, float* gstats, void* stream);
int trec_score_row_scale_i8(const float* repr, int64_t n, int32_t d, float* nat, float* gmax, void* stream);
int trec_score_prep_i8_users(const float* repr, int64_t n, int32_t d, int32_t kpad, const float* wg_scale, int32_t wg_rows,
                             void* out_q, float* row_stats, void* stream);
int trec_score_bias_i8_classes(const float* bias, int64_t n, int32_t sb_rows, const float* ladder, const int32_t* class_used,
                               int32_t n_classes, float* sb_stats, int32_t* bias_q, float* gstats, void* stream);
int32_t trec_score_blockmax_i8_rows_per_workgroup(int32_t top_k);
int trec_score_user_err_i8(const float* user_stats, const float* user_bias, const float* gstats, int32_t kdim,
                           int64_t n_users, const float* scales, const float* wg_scale, int32_t wg_rows, float* r_err,
                           void* stream);
int trec_score_gemm_blockmax_i8(const void* users_q, const void* items_q, int32_t kpad, int64_t n_users, int64_t n_items,
                                const float* user_bias, const int32_t* item_bias_q, const float* scales,
                                const float* sb_stats, int32_t sb_rows, int32_t n_chunks, float* blockmax,
                                int64_t bm_stride, const float* user_err, float* chunk_top, int32_t top_k,
                                const float* wg_scale, const int32_t* wg_class, int32_t wg_rows, void* stream);
/* The user side of the cascade, prepared on the device without a host read (csrc/user_prep.hip; replaces the user
 * representation's way into tf.matmul of tensorrec/prediction_graphs.py:49-50, after the l2_normalize of :64-69 for cosine):
 * scale class per row -> stable counting sort by class -> bands of two classes padded to whole int8 workgroups -> ONE gather pass
 * that writes the fp32 / bf16 / int8 operands, both error norms and the user bias in layout order.  The layout has
 * trec_user_prep_alloc_rows(n, wg_rows) rows -- a bound the host knows; int8 workgroups beyond the padded row count have
 * wg_scale 0 and exit at once, layout rows without a source (src < 0) are zero and keep nothing (trec_topk_cascade_floor).
 *   src [n_alloc] int32: caller's row of a layout row or -1; pos [n]: layout row of a caller's row; wg_scale / wg_class
 *   [n_alloc / wg_rows]; ladder [64] = the classes' scales gmax 2^(-c/4); class_used [64]; gmax [1] zeroed by the caller;
 *   meta int32[2] = {padded rows, n}; workspace of trec_user_prep_workspace_bytes(n) bytes; out_f32 / bias_sorted nullable.
 * trec_fill_zero: nbytes of zeros on the stream (the counters / maxima a call starts from, as one block). */
int64_t trec_user_prep_alloc_rows(int64_t n, int32_t wg_rows);
int64_t trec_user_prep_workspace_bytes(int64_t n);
int trec_user_prep_sorted(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize, const float* user_bias,
                          int32_t wg_rows, int64_t n_alloc, void* workspace, int64_t workspace_bytes, int32_t* src,
                          int32_t* pos, float* wg_scale, int32_t* wg_class, float* ladder, int32_t* class_used, float* gmax,
                          int32_t* meta, float* out_f32, void* out_bf16, float* row_stats, void* out_q, float* row_stats8,
                          float* bias_sorted, void* stream);
int trec_fill_zero(void* p, int64_t nbytes, void* stream);
int32_t trec_topk_rows_user_blocks(int64_t n_users);
int trec_topk_rows_count(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                         const float* user_err, const float* sb_stats, int32_t kdim, int32_t* block_off,
                         int32_t* row_total, int32_t* row_pad, int64_t* pstart, int64_t cap_rows, int64_t* status,
                         void* stream);
int trec_topk_rows_fill(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                        const float* user_err, const float* sb_stats, int32_t kdim, const int32_t* block_off,
                        const int32_t* row_total, const int64_t* pstart, int64_t cap_rows, const int64_t* status,
                        int32_t* row_user, int32_t* rblock_chunk, void* stream);
int trec_topk_rows_collect(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                           const float* user_err, const float* sb_stats, int32_t kdim, int32_t rcap, int32_t* row_count,
                           int32_t* row_user, int64_t* status, void* stream);
int trec_score_gemm_blockmax_grouped(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                     int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                     const int32_t* rblock_chunk, const int32_t* row_user, float* blockmax,
                                     int64_t bm_stride, int32_t wgs_per_row, void* stream);
/* Skewed catalogues (fitted models, Zipf popularity: a few superblocks are wanted by most users although only 2-3% of all
 * pairs are kept).  trec_topk_rows_hot, after trec_topk_rows_collect: superblocks with row_count > rcap are listed in hot_list
 * [hot_cap] (ascending, -1 padded) and their row_count is zeroed -- the fixed-capacity grouped launch skips them;
 * status [3] = {resident rows of both launches, 1 when more than hot_cap superblocks are hot or more than max_pairs pairs would be refined: the
 * caller falls back to the dense bf16 stage 1, the number of hot superblocks (the grid of the hot launch: hot_cap = that)}.  trec_score_gemm_blockmax_hot: the dense bf16 filter kernel over the listed superblocks only, EVERY user,
 * maxima written over the table's entries (same arithmetic as tf.matmul of tensorrec/prediction_graphs.py:49-50 in bf16,
 * used as a bounded filter like the grouped form). */
int trec_topk_rows_hot(int32_t* row_count, int32_t n_sb, int32_t rcap, int64_t n_users, int32_t* hot_list, int32_t hot_cap,
                       int64_t max_pairs, int64_t* status, void* stream);
int trec_score_gemm_blockmax_hot(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_users,
                                 int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                 const int32_t* hot_list, int32_t hot_cap, float* blockmax, int64_t bm_stride, void* stream);
/* The refining launches that also LIST candidates (csrc/topk_cascade.hip, kernel form LIST of csrc/score_blockmax.hip; same
 * place in the reference: the fp32 tf.matmul + tf.nn.top_k of prediction_graphs.py:49-50 / recommendation_graphs.py:80, used
 * as a bounded filter).  Same work as trec_score_gemm_blockmax_grouped (fixed-capacity layout: wgs_per_row > 0, rblock_chunk =
 * row_count) / trec_score_gemm_blockmax_hot, and besides the maxima every ITEM of a refined pair whose bf16 score (+ user bias)
 * reaches cand_floor[user] -- tauLB - eps, trec_topk_filter_floor_ex(tau8, ..., mult = 1): +inf lists nothing -- is appended
 * to cand [n_users][cand_cap] of {int32 item id + item_index_base, float score bits}; cand_n [n_users] (zeroed by the caller)
 * counts the appends, beyond cand_cap they are dropped.  trec_topk_candidates_finish: tau = the k-th largest listed score,
 * survivors = listed items >= tau - 2 eps (user_stats / item_gstats as for trec_topk_filter_floor), exact fp32 re-scoring
 * and top-k as trec_topk_filter_finish; users with cand_n > cand_cap (64, 128, 192 or 256) or more than 64 survivors are
 * flagged, users whose cand_floor is +inf are skipped (their outputs are -inf / -1).  This replaces, behind the int8 stage,
 * the table scan, the grouping by superblock and the grouped list kernel of the bf16 filter.  trec_topk_dense_users, before
 * the refining launches: a user for whom `limit` or more of 32 sampled superblocks reach the threshold of
 * trec_topk_rows_collect (the int8 bound says nothing about its row) gets cand_floor = +inf and is flagged at once -- listing
 * most of the catalogue for it would only end in the same flag. */
int trec_topk_dense_users(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                          const float* user_err, const float* sb_stats, int32_t kdim, int32_t limit, float* cand_floor,
                          int32_t* flag, int32_t* n_flagged, void* stream);
int trec_score_gemm_refine_candidates(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                      int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                      const int32_t* row_count, const int32_t* row_user, float* blockmax, int64_t bm_stride,
                                      int32_t wgs_per_row, const float* cand_floor, int32_t* cand_n, void* cand,
                                      int32_t cand_cap, int32_t item_index_base, const int32_t* wg_map, int32_t n_wgs,
                                      void* stream);
/* Only the workgroup slots of the [n_sb][wgs_per_row] grid that hold rows (after trec_topk_rows_hot): wg_start [n_sb + 1] =
 * exclusive prefix of min(ceil(row_count / 512), wgs_per_row), wg_map[wg_start[s] + j] = s * wgs_per_row + j (map_cap entries
 * at most).  trec_score_gemm_refine_candidates(wg_map, n_wgs) then launches n_wgs workgroups instead of n_sb * wgs_per_row
 * (1.9M at 1M users, 98k of them with rows); wg_map NULL = the full grid. */
int trec_topk_rows_wg_map(const int32_t* row_count, int32_t n_sb, int32_t wgs_per_row, int32_t* wg_start, int32_t* wg_map,
                          int64_t map_cap, void* stream);
/* The same map with group_rows users per workgroup slot instead of 512 (entries beyond wg_start[n_sb] are left as the caller
 * preset them: an id >= n_sb * wgs_per_row is an idle workgroup of trec_score_gemm_refine_candidates_resident). */
int trec_topk_rows_wg_map_ex(const int32_t* row_count, int32_t n_sb, int32_t wgs_per_row, int32_t group_rows, int32_t* wg_start,
                             int32_t* wg_map, int64_t map_cap, void* stream);
/* trec_score_gemm_refine_candidates with the ITEMS resident (csrc/refine_resident.hip; same bf16-path maxima over the table entries,
 * same candidate lists -- tensorrec/prediction_graphs.py:50 + recommendation_graphs.py:41 reduced towards the first tf.nn.top_k of
 * recommendation_graphs.py:80): workgroup w keeps the 512 items of superblock wg_map[w] / segs_per_row in registers and streams
 * segment wg_map[w] % segs_per_row (seg_rows users, a multiple of 64) of its user list row_user [n_sb][rcap] through LDS.
 * sb_rows must be 512; row_count [n_sb] is clamped to rcap (0 = nothing to do: hot superblocks). */
int trec_score_gemm_refine_candidates_resident(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_items,
                                               const float* user_bias, const float* item_bias, int32_t sb_rows, int32_t n_sb,
                                               const int32_t* row_count, const int32_t* row_user, int32_t rcap, float* blockmax,
                                               int64_t bm_stride, const float* cand_floor, int32_t* cand_n, void* cand,
                                               int32_t cand_cap, int32_t item_index_base, const int32_t* wg_map, int32_t n_wgs,
                                               int32_t segs_per_row, int32_t seg_rows, void* stream);
int trec_score_gemm_refine_candidates_hot(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_users,
                                          int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                          const int32_t* hot_list, int32_t hot_cap, float* blockmax, int64_t bm_stride,
                                          const float* cand_floor, int32_t* cand_n, void* cand, int32_t cand_cap,
                                          int32_t item_index_base, void* stream);
int trec_topk_candidates_finish(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                const float* user_stats, const float* item_gstats, const float* users_f32,
                                const float* items_f32, int64_t ld_users, int64_t ld_items, int32_t kdim,
                                const float* user_bias, const float* item_bias, int32_t item_index_base, int64_t n_users,
                                int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag, int32_t* n_flagged,
                                const int32_t* out_index, int32_t lanes_per_user, void* stream);
/* The same finish in two launches for lists of mixed length (the single-GPU cascade: ~15 candidates per user, a few users with
 * hundreds): 16 lanes per user with cands_per_lane (1 / 2 / 4) candidates per lane -- four users per wave --, users with longer
 * lists appended to over_list [n_users] (over_count [1], zeroed by the caller) and finished by the wave-per-user form on a fixed
 * grid that reads the count on the device.  Same results (recommendation_graphs.py:73-82 restricted to the first k places).      */
int trec_topk_candidates_finish_mixed(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                      const float* user_stats, const float* item_gstats, const float* users_f32,
                                      const float* items_f32, int64_t ld_users, int64_t ld_items, int32_t kdim,
                                      const float* user_bias, const float* item_bias, int32_t item_index_base, int64_t n_users,
                                      int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag, int32_t* n_flagged,
                                      const int32_t* out_index, int32_t cands_per_lane, int32_t* over_list, int32_t* over_count,
                                      void* stream);
/* The finish of the wide route (17 <= k <= 64; lists of up to 1,024 candidates made with the provisional floor): one wave per user
 * re-scores EVERY listed item with the reference's fp32 chain and takes the k best (recommendation_graphs.py:73-82 restricted to the
 * first k places); users whose list is incomplete (more entries than cand_cap, fewer than k) are flagged, users flagged before or
 * with a +inf floor are skipped.  out_index as for trec_topk_candidates_finish.                                                  */
int trec_topk_candidates_finish_wide(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                     const float* users_f32, const float* items_f32, int64_t ld_users, int64_t ld_items,
                                     int32_t kdim, const float* user_bias, const float* item_bias, int32_t item_index_base,
                                     int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag, int32_t* n_flagged,
                                     const int32_t* out_index, void* stream);
/* The cascade's thresholds in one pass over the users, before trec_topk_rows_collect: tau [n_users] IN / OUT = the k-th largest
 * int8 lower bound (+inf on return for layout rows without a source: src [n_users] nullable, trec_user_prep_sorted);
 * floor0 = tau - eps rounded down twice (the provisional floor of the candidate lists; +inf and flag = 1 when the bound is
 * unusable); n_flagged [1] zeroed by the caller; cand_n nullable [n_users], zeroed here.  out_index of
 * trec_topk_candidates_finish (nullable): user u's lists go to row out_index[u], negative = no output.  lanes_per_user: 0 / 64 = a
 * wave per user (up to cand_cap candidates); 16 = four users per wave for SHORT lists (item shards of an N-GPU run: ~27 / N
 * candidates per user and shard) -- a user with more than 16 candidates is flagged and re-done by the caller. */
int trec_topk_cascade_floor(float* tau, const int32_t* src, const float* user_stats, const float* user_bias,
                            const float* item_gstats, int32_t kdim, int64_t n_users, float* floor0, int32_t* flag,
                            int32_t* n_flagged, int32_t* cand_n, void* stream);

/* The cascade's PRE-REFINEMENT (csrc/topk_filter.hip): the k superblocks holding a user's k largest int8 lower bounds
 * are refined first; tau = max(tau8, min of their bf16 maxima - eps) is a sharper lower bound of the k-th best score of
 * tf.nn.top_k (recommendation_graphs.py:80) for the compaction and the candidate floor.  trec_topk_prerefine_rows: selection over
 * the chunk lists written with top_k | 0x100 (tagged lower bounds) -> per-superblock user lists (layout of trec_topk_rows_collect)
 * + sel_sb [n_users][k] + ok [n_users]; the bf16 launch over them is trec_score_gemm_refine_candidates (it also lists their
 * candidates; trec_topk_prerefine_tau with listed = 1 then saves the maxima, takes the pairs out of the compaction (-inf) and raises
 * tau and the candidate floor) or trec_score_gemm_blockmax_grouped (listed = 0: the entries become +inf and the listing launch
 * refines them again). */
int32_t trec_topk_prerefine_max_superblocks(void);
int trec_topk_prerefine_rows(const int32_t* sel, const float* sel_val, int32_t k, int32_t top_k, int32_t sb_per_chunk, int32_t n_sb,
                             int64_t n_users, const int32_t* src, int32_t rcap, int32_t* sel_sb, int32_t* row_count,
                             int32_t* row_user, int32_t* ok, void* stream);
int trec_topk_prerefine_tau(const int32_t* sel_sb, const int32_t* ok, int32_t k, float* table, int64_t stride, int64_t n_users,
                            const int32_t* src, const float* user_stats, const float* user_bias, const float* item_gstats,
                            int32_t kdim, float* tau, int32_t listed, float* vals, float* cand_floor, void* stream);
/* The pre-refinement without a table pass behind its launch (round 6): trec_topk_prerefine_rows_pos also records sel_pos [n_users][k],
 * the position of every placed pair inside its superblock's list; trec_score_gemm_refine_candidates_marked (declared with the other
 * refining launches) leaves the bf16 maxima in pre_max [n_sb * rcap] by list position and marks the table entries -inf itself;
 * trec_topk_prerefine_tau_listed reads the maxima back from there (vals, tau, cand_floor as trec_topk_prerefine_tau with listed != 0).
 * These two take k <= 64 (the wide route, 17 <= k <= 64, pre-refines too); trec_topk_prerefine_rows / _tau keep k <= 16.
 * Same reference arithmetic as the calls they replace (recommendation_graphs.py:73-82 restricted to what the exact top-k needs). */
int trec_topk_prerefine_rows_pos(const int32_t* sel, const float* sel_val, int32_t k, int32_t top_k, int32_t sb_per_chunk,
                                 int32_t n_sb, int64_t n_users, const int32_t* src, int32_t rcap, int32_t* sel_sb,
                                 int32_t* row_count, int32_t* row_user, int32_t* ok, int32_t* sel_pos, void* stream);
int trec_topk_prerefine_tau_listed(const int32_t* sel_sb, const int32_t* sel_pos, const int32_t* ok, int32_t k, const float* pre_max,
                                   int32_t rcap, int64_t n_users, const int32_t* src, const float* user_stats,
                                   const float* user_bias, const float* item_gstats, int32_t kdim, float* tau, float* vals,
                                   float* cand_floor, void* stream);
int trec_score_gemm_refine_candidates_marked(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                             int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                             const int32_t* row_count, const int32_t* row_user, float* blockmax, int64_t bm_stride,
                                             int32_t wgs_per_row, const float* cand_floor, int32_t* cand_n, void* cand,
                                             int32_t cand_cap, int32_t item_index_base, const int32_t* wg_map, int32_t n_wgs,
                                             float* pre_max, void* stream);
/* The exact EUCLIDEAN top-k (tensorrec/prediction_graphs.py:84-100 + tf.nn.top_k of recommendation_graphs.py:73-82) through the
 * dot-product cascade: per user, nearest = largest g = u.i - r_i / 2, a dot product with item "bias" -r_i / 2.  After the cascade
 * gave the kc largest g per user (16 for k <= 12; 32 / 64 from the wide cascade's lists for k up to 48; kc <= 64) and
 * trec_pair_score_exact their reference-chain scores (biases included),
 * trec_topk_euclid_certify orders them by (score desc, id asc), writes the first k, and flags the users for whom an item OUTSIDE the
 * kc could still reach the first k places (score upper bound from the kc-th largest g and the largest item bias; csrc/euclid_topk.hip):
 * those are re-done on the exact fp32 MFMA path.  item_gstats: trec_score_prep_filter's maxima with bias = -r / 2. */
int trec_topk_euclid_certify(const int32_t* cand_idx, const float* cand_g, const float* exact, int32_t kc, int32_t k,
                             const float* user_sq, const float* user_bias, const float* item_gstats, const float* bias_max,
                             int32_t kdim, int64_t n_users, float* out_vals, int32_t* out_idx, int32_t* flag,
                             int32_t* n_flagged, const float* lambda_, const float* bias_min, void* stream);
int trec_topk_filter_floor(const float* tau, const float* user_stats, const float* user_bias, const float* item_gstats,
                           int32_t kdim, int64_t n_users, float* floor, int32_t* flag, int32_t* n_flagged, void* stream);
int trec_topk_collect_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, const float* floor,
                             int32_t ksel, int32_t* keys, int32_t* count, int32_t* flag, int32_t* n_flagged, void* stream);
/* select + collect in ONE pass over the table (behind the int8 stage): trec_topk_filter_floor_ex(tau8, ..., mult = 4) gives a
 * provisional floor that cannot exceed the final one (tau16 >= tau8 - eps); trec_topk_scan_blocks keeps the k largest entries of
 * every column (tau; sel_max [k][n_users] for item shards) AND lists the entries >= that floor (cand_s / cand_v
 * [cand_cap][n_users] -- slot-major -- in superblock order, cand_n = their number, possibly above cand_cap); trec_topk_prune_candidates applies
 * the final floor to the ~45 candidates of a user instead of its 1,954 table entries and writes keys / count exactly as
 * trec_topk_collect_blocks does; users with an incomplete candidate list are marked in redo [n_users] and
 * trec_topk_collect_blocks_masked collects exactly those from the table (workgroups without a marked user exit at once). */
int trec_topk_filter_floor_ex(const float* tau, const float* user_stats, const float* user_bias, const float* item_gstats,
                              int32_t kdim, int64_t n_users, float mult, float* floor, int32_t* flag, int32_t* n_flagged,
                              void* stream);
int trec_topk_scan_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k, const float* floor0,
                          int32_t cand_cap, float* sel_max, float* tau, int32_t* cand_s, float* cand_v, int32_t* cand_n,
                          void* stream);
int trec_topk_prune_candidates(const int32_t* cand_s, const float* cand_v, const int32_t* cand_n, int32_t cand_cap,
                               const float* floor, int32_t ksel, int64_t n_users, int32_t* keys, int32_t* count, int32_t* flag,
                               int32_t* n_flagged, int32_t* redo, void* stream);
int trec_topk_collect_blocks_masked(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, const float* floor,
                                    int32_t ksel, int32_t* keys, int32_t* count, int32_t* flag, int32_t* n_flagged,
                                    const int32_t* redo, void* stream);
int trec_topk_filter_finish(const int32_t* part_idx, int32_t capacity, int32_t ksel, const int32_t* count,
                            const float* users_f32, const float* items_f32, int64_t ld_users, int64_t ld_items,
                            int32_t kdim, const float* user_bias, const float* item_bias, int32_t item_index_base,
                            int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                            int32_t* n_flagged, void* stream);
/* The same finish without capacity limits (any ksel, any number of survivors, k <= 16): the WIDE second pass over the users
 * the first pass flagged (trec_topk_collect_blocks with a larger ksel, 16-entry stage-3 lists); flags only a full list. */
int trec_topk_filter_finish_wide(const int32_t* part_idx, int32_t capacity, int32_t ksel, const int32_t* count,
                                 const float* users_f32, const float* items_f32, int64_t ld_users, int64_t ld_items,
                                 int32_t kdim, const float* user_bias, const float* item_bias, int32_t item_index_base,
                                 int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                                 int32_t* n_flagged, void* stream);

/* k best of n_cand candidates per user, ordered (value desc, index asc) = tf.nn.top_k tie rule; also the merge
 * step after the all-gather of per-shard lists */
int trec_topk_merge(const float* part_vals, const int32_t* part_idx, int64_t n_users, int32_t n_cand, int32_t k,
                    float* out_vals, int32_t* out_idx, void* stream);

/* ---- K3: per-pair ("serial") scores -----------------------------------------------------------------------
 * prediction_graphs.py:52-55 (dot), :70-72 (cosine after l2norm), :105-117 (euclidean) fused with
 * bias_prediction_serial, recommendation_graphs.py:55-57.  xu == NULL: user of pair p is p / pairs_per_user
 * (the [U, S] sample layout of util.py:16-19).  out_sqdist (nullable): the pair's accumulator before the epilogue --
 * the squared distance in Euclidean mode, kept for the backward pass.  bwd accumulates (+=) into dU / dV / d_*_bias
 * (fp32 atomics). */
int trec_pair_score_fwd(const float* U, const float* V, const int32_t* xu, const int32_t* xi, int64_t n_pairs,
                        int32_t pairs_per_user, int32_t d, int32_t mode, const float* user_bias,
                        const float* item_bias, float* out, float* out_sqdist, void* stream);
int trec_pair_score_bwd(const float* U, const float* V, const int32_t* xu, const int32_t* xi, const float* grad,
                        int64_t n_pairs, int32_t pairs_per_user, int32_t d, int32_t mode, float* dU, float* dV,
                        float* d_user_bias, float* d_item_bias, void* stream);

/* Euclidean pairs, backward: coef[p] = -grad[p] / sqrt(D_p), 0 where D_p < 1e-16 was clamped (tf.maximum passes no
 * gradient there, prediction_graphs.py:113-115); dU[u] = sum_p coef[p] (U[u] - V[i_p]) and
 * dV[i] = sum_p coef[p] (V[i] - U[u_p]) are then trec_spmm_csr_split gathers with `own` set.  sqdist (nullable): the
 * squared distances the forward pass kept (out_sqdist) -- then an elementwise pass, U / V / xu / xi are not read. */
int trec_pair_euclid_coef(const float* U, const float* V, const int32_t* xu, const int32_t* xi, const float* grad,
                          const float* sqdist, int64_t n_pairs, int32_t pairs_per_user, int32_t d, float* coef,
                          void* stream);

/* Group a pair list by item (counting sort on the device; pairs with a negative item are skipped): writes the transposed structure (indptr_t[n_items+1],
 * users_t[n_pairs], perm_t[n_pairs]) so that the item-side gradient of sampled serial predictions is the trec_spmm_csr
 * segmented gather (values = grad, val_perm = perm_t, indices = users_t) instead of atomics.
 * workspace_i32: 2*n_items int32; workspace_i64: ceil(n_items/1024)+1 int64.  counts_given != 0: the first n_items
 * entries of workspace_i32 already hold the histogram of xi (trec_wmrb_fused_step counts it while it gathers);
 * ranks (nullable, with counts_given): what those histogram atomics returned, i.e. every pair's position inside its
 * bucket -- the fill pass then needs no atomics; values_in / values_out (nullable, with ranks): the pairs' values are
 * scattered along (values_out[slot] = values_in[pair]) so that the gather reads them in order; perm_t may then be NULL.
 * values_in with values_out == NULL: PACKED -- users_t is an int2[n_pairs] buffer receiving {user, value bits}, one
 * 8-byte store per pair; consume it with trec_spmm_csr_packed. */
int trec_group_pairs_by_item(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user,
                             int64_t n_items, int32_t* workspace_i32, int64_t* workspace_i64, int64_t* indptr_t,
                             int32_t* users_t, int32_t* perm_t, int32_t counts_given, const int32_t* ranks,
                             const float* values_in, float* values_out, void* stream);
/* The same grouping for <= 32,768 buckets and >= 4M pairs (MovieLens-shaped catalogues: 3.7e8 sampled pairs over 26,744 items)
 * without a single global atomic: per-run counters in the LDS of one workgroup, a column scan over the runs, placement through LDS
 * cursors (csrc/segment.hip).  trec_group_pairs_lds_runs: the runs the form uses (0: it does not apply); run_counts: int32
 * [runs][n_items] scratch; workspace_i32: n_items int32; workspace_i64: ceil(n_items / 1024) + 1 int64. */
int32_t trec_group_pairs_lds_runs(int64_t n_pairs, int64_t n_items);
int trec_group_pairs_by_item_lds(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user, int64_t n_items,
                                 int32_t* workspace_i32, int64_t* workspace_i64, int32_t* run_counts, int64_t* indptr_t,
                                 int32_t* users_t, int32_t* perm_t, void* stream);
/* trec_group_pairs_by_item's ranked PACKED form (counts_given = 1, ranks, values_in; entries int2 [n_pairs] = {user, value bits})
 * with the fill done in two levels: pairs are first appended to the staging region of their destination WINDOW of
 * 2^window_log2 entries (one global atomic per workgroup and window), then placed window by window while the window sits in
 * L2 (csrc/segment.hip) -- the 1e8 sampled pairs of the 1M x 1M fit.  trec_group_pairs_staged_bytes: the staging bytes, 0 when
 * the size is not covered (fewer than 2 or more than 2,048 windows).  workspace_i32 [2 * n_items]: first half = the histogram. */
int64_t trec_group_pairs_staged_bytes(int64_t n_pairs, int32_t window_log2);
int trec_group_pairs_by_item_staged(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user, int64_t n_items,
                                    int32_t* workspace_i32, int64_t* workspace_i64, int64_t* indptr_t, int32_t* entries,
                                    const int32_t* ranks, const float* values_in, void* staging, int64_t staging_bytes,
                                    int32_t window_log2, void* stream);

/* Grouping by item WITHOUT ranks (csrc/segment.hip, round 5): bins of 4,096 items; tiles of 8,192 pairs sorted by bin in LDS and
 * written out run by run; ONE workgroup per bin counts, scans (= indptr of its items) and places.  No global atomic per pair: the
 * fused WMRB kernel then needs no histogram (sample_hist = NULL).  For pair lists roughly uniform over the items -- the sampled pairs
 * of tensorrec.py:298-302, whose item-side gradient is the gather of trec_spmm_csr_packed over these entries.
 * trec_group_pairs_binned_bytes: workspace bytes, 0 = size not covered (more than 2M items, fewer than 2^22 pairs). */
int64_t trec_group_pairs_binned_bytes(int64_t n_pairs, int64_t n_items);
int trec_group_pairs_by_item_binned(const int32_t* xu, const int32_t* xi, const float* values, int64_t n_pairs, int32_t pairs_per_user,
                                    int64_t n_items, int32_t drop_zero_values, void* workspace, int64_t workspace_bytes,
                                    int64_t* indptr_t, int32_t* entries, void* stream);

/* ---- K4: ranks ----------------------------------------------------------------------------------------------
 * rank_predictions, recommendation_graphs.py:73-82 (double tf.nn.top_k) as an exact count; int32, 1 = best. */
int trec_rank_rows(const float* scores, int64_t n_users, int64_t n_items, int64_t ld_scores, int32_t* ranks,
                   int64_t ld_ranks, void* stream);
/* The same ranks for rows of more than 32768 items: the row is sorted in 32768-item chunks (one workgroup each, LDS) and
 * every item adds, for each OTHER chunk, the number of that chunk's items that beat it -- a binary search in the chunk's
 * sorted keys (ties go to the lower chunk, so no index data is needed across chunks).  workspace:
 * trec_rank_rows_workspace_bytes(n_users, n_items) bytes, 16-byte aligned; n_users <= 65535 per call. */
int64_t trec_rank_rows_workspace_bytes(int64_t n_users, int64_t n_items);
int trec_rank_rows_chunked(const float* scores, int64_t n_users, int64_t n_items, int64_t ld_scores, int32_t* ranks,
                           int64_t ld_ranks, void* workspace, int64_t workspace_bytes, void* stream);
/* K2r -- ranks of selected pairs WITHOUT a score slab (csrc/score_rank.hip): the count of rank_predictions
 * (recommendation_graphs.py:73-82) as the epilogue of the fp32 MFMA score kernel.  users_f32 / items_f32: fp32 operands
 * [n, kpad] (trec_score_prep, dtype fp32).  Resident row r scores operand row row_user[r] and counts, for each of its
 * row_tn[r] <= trec_score_rankcount_max_targets() targets (tgt_item / tgt_score [row_t0[r] ...]; item ids are global:
 * local row + item_index_base), the items of THIS call's item range that beat it: counts[t] += #{j : s_j > t or
 * (s_j == t and j < item_t)}.  counts must be zero-initialised; item shards add their counts (all-reduce SUM) and the
 * rank is count + 1.  tgt_score must be the exact score of the pair: trec_pair_score_exact (the same k-ordered fmaf
 * chain, (s + b_u) + b_i, Euclidean transform of prediction_graphs.py:84-100 in mode 1).  n_chunks <= 0: automatic. */
int trec_score_rankcount_max_targets(void);
int trec_pair_score_exact(const float* users_f32, const float* items_f32, int64_t ld, int32_t kdim, const int32_t* xu,
                          const int32_t* xi, int64_t n_pairs, const float* user_bias, const float* item_bias,
                          int32_t mode, const float* user_sqnorm, const float* item_sqnorm, int32_t item_index_base,
                          float* out, void* stream);
int trec_score_gemm_rankcount(const float* users_f32, const float* items_f32, int32_t kpad, int64_t n_rows,
                              int64_t n_items, int32_t item_index_base, const float* user_bias, const float* item_bias,
                              int32_t mode, const float* user_sqnorm, const float* item_sqnorm, const int32_t* row_user,
                              const int32_t* row_t0, const int32_t* row_tn, const int32_t* tgt_item,
                              const float* tgt_score, int32_t n_chunks, int32_t* counts, void* stream);
/* partial rank counts of selected (user, item) pairs over item columns [begin, end) of a score slab whose first
 * column is global item `col_offset`; item shards sum their counts (add_one on exactly one of them) */
int trec_rank_of_pairs(const float* scores, int64_t ld_scores, int64_t col_offset, int64_t begin, int64_t end,
                       const int32_t* xu, const int32_t* xi, const float* target_scores, int64_t n_pairs,
                       int32_t add_one, int32_t* out, void* stream);
/* The same counts for pairs GROUPED BY USER (pair_indptr[n_users+1]: user u of the score slab owns pairs
 * pair_indptr[u] .. pair_indptr[u+1]; xi / target_scores / out are indexed by pair): every slice of a user's row is read
 * once for all of the user's targets instead of once per pair.  out is overwritten. */
int trec_rank_of_pairs_by_user(const float* scores, int64_t ld_scores, int64_t col_offset, int64_t begin, int64_t end,
                               const int64_t* pair_indptr, const int32_t* xi, const float* target_scores,
                               int64_t n_users, int64_t n_pairs, int32_t add_one, int32_t* out, void* stream);

/* ---- K6: losses ---------------------------------------------------------------------------------------------
 * WMRB / BalancedWMRB, loss_graphs.py:153-180 / :189-227.  Interactions are CSR over users (indptr[n_users+1]);
 * pos_slot[p] = position of interaction p in the compacted positive vector or -1; pos_weight (NULL for plain
 * WMRB) = value_p / per-item positive sum.  loss, smr: [n_positive].                                          */
int trec_wmrb_fwd(const int64_t* indptr, const int32_t* pos_slot, const float* pos_weight, const float* pred_serial,
                  const float* sample_pred, int64_t n_users, int64_t n_items, int32_t n_sampled, float* loss,
                  float* smr, void* stream);
int trec_wmrb_bwd(const int64_t* indptr, const int32_t* pos_slot, const float* pos_weight, const float* pred_serial,
                  const float* sample_pred, const float* smr, const float* grad_loss, int64_t n_users,
                  int64_t n_items, int32_t n_sampled, float* d_pred_serial, float* d_sample_pred, void* stream);
/* One WMRB training step's user side in one pass (csrc/wmrb_fused.hip): serial + sample predictions
 * (prediction_graphs.py:52-55 + recommendation_graphs.py:55-57 over tensorrec.py:384-395), the WMRB / BalancedWMRB loss
 * (loss_graphs.py:153-227) and, for the SUM of the loss vector (what tensorrec.py:487-489 minimises), its gradient
 * w.r.t. the user representation and user bias, with every item row gathered from HBM once (register-resident).  Dot
 * scores on the given representations (cosine: pass the normalised rows).  samples: int32 [n_users, n_sampled];
 * x_item: int32 item of every interaction (CSR order).  Outputs: loss [n_positive], pred_serial [n_interactions],
 * dU [n_users, d], d_user_bias [n_users] (NULL iff user_bias is NULL), coef_samples [n_users, n_sampled] and
 * coef_pairs [n_interactions] = d(sum loss)/d(prediction) per pair -- the values of the item-side gathers
 * (trec_group_pairs_by_item + trec_spmm_csr) and item-bias segment sums.  sample_hist (nullable, int32 [n_items],
 * zeroed by the caller): += the number of times every item was sampled -- the histogram pass of the counting sort;
 * sample_rank (nullable, int32 [n_users, n_sampled]): the value each of those atomics returned (rank inside the bucket).
 * trec_wmrb_fused_lds_bytes: dynamic LDS the launch needs, or -1 if (n_sampled <= 256, d % 4 == 0, d <= 256,
 * n_sampled + max interactions per user <= 256 rows, 128 for d > 128) does not hold -- then run the unfused kernels.   */
int trec_wmrb_fused_lds_bytes(int32_t n_sampled, int32_t max_interactions_per_user, int32_t d);
int trec_wmrb_fused_step(const float* U, const float* V, const float* user_bias, const float* item_bias,
                         const int64_t* indptr, const int32_t* x_item, const int32_t* pos_slot, const float* pos_weight,
                         const int32_t* samples, int64_t n_users, int64_t n_items, int32_t n_sampled, int32_t d,
                         int32_t max_interactions_per_user, float* loss, float* pred_serial, float* dU,
                         float* d_user_bias, float* coef_samples, float* coef_pairs, int32_t* sample_hist,
                         int32_t* sample_rank, void* stream);
/* The same step for S in the thousands, long interaction rows and Euclidean scores (csrc/wmrb_tiled.hip): the user's S + n_u item
 * rows are streamed twice in tiles (scores into LDS, then the coefficient-weighted sum dU) instead of living in registers.
 * mode 0: dot scores (prediction_graphs.py:52-55; cosine = dot on normalised rows, :70-72); mode 1: euclidean,
 * -sqrt(max(sum (u - i)^2, 1e-16)) (:105-117).  val_samples [n_users, n_sampled] / val_pairs [nnz]: per pair the value the item
 * side sums -- mode 0: g = d(sum loss)/d score, dV[i] = sum g U[u]; mode 1: c = -g / sqrt(D) (0 where D was clamped),
 * dV[i] = sum c (V[i] - U[u]).  raw_samples / raw_pairs (both or neither; mode 1 with item biases): g itself, d b_i = sum g.
 * dense_g (or NULL): a ZEROED [n_users, ldg] matrix, ldg >= n_items -- every pair's value is added at (user, item); the item side
 * is then G^T . U (trec_gemm_f32) instead of a sort + gather, the right form when n_sampled is a sizeable share of n_items.
 * dU may be NULL with dense_g: the second sweep over the rows is skipped, val_rowsum [n_users] receives the sum of each user's
 * values and the caller forms dU = G . V (mode 0) or val_rowsum[u] U[u] - (G . V)[u] (mode 1).
 * trec_wmrb_tiled_lds_bytes: dynamic LDS of the launch, or -1 when not covered (d % 4 == 0, d <= 512, and
 * 2 (n_sampled + longest row) + 2 (longest row) + 8 d (16 d for d <= 64) floats within 128 KB) -- then run the unfused kernels.                     */
int trec_wmrb_tiled_lds_bytes(int32_t n_sampled, int32_t max_interactions_per_user, int32_t d);
int trec_wmrb_tiled_step(const float* U, const float* V, const float* user_bias, const float* item_bias,
                         const int64_t* indptr, const int32_t* x_item, const int32_t* pos_slot, const float* pos_weight,
                         const int32_t* samples, int64_t n_users, int64_t n_items, int32_t n_sampled, int32_t d, int32_t mode,
                         int32_t max_interactions_per_user, float* loss, float* pred_serial, float* dU, float* d_user_bias,
                         float* val_samples, float* val_pairs, float* raw_samples, float* raw_pairs, float* dense_g,
                         int64_t ldg, float* val_rowsum, void* stream);
/* ONE kernel per training step for models that fit on chip (csrc/step_coop.hip; BASELINE.json configs[1]): the whole of
 * session.run(tf_optimizer) (tensorrec.py:617-622) for LinearRepresentation (identity user features; any item features) + DotProduct +
 * WMRB / BalancedWMRB -- item tower forward, sampling, the tiled WMRB step per user, d V = G^T . U, item tower backward, TF-form Adam on
 * every variable -- as one cooperative launch with grid-wide barriers between its four phases.  Weights W* and Adam slots *_m / *_v are
 * updated in place (bias pointers: all six or none); loss [P+] and pred_serial [nnz] are written for the caller's log.  samples
 * [n_users, n_sampled] or NULL: drawn in the kernel, the same bits as trec_sample_items(n_users, user_base, n_items, n_sampled, 0,
 * seed, step).  f_* : CSR of the item features, ft_*: CSR of their transpose (values through ft_perm).  lr_t / l2 as
 * trec_adam_tf_step (l2 on all four variables: the reference regularises its bias variables too, tensorrec.py:313).  trec_fit_step_coop_workspace_floats: floats of workspace, or -1 when the model
 * is not covered (d % 4 == 0, d <= 128, phases' LDS within 64 KB, G = n_users x n_items within 256 MB); TREC_ERR_UNSUPPORTED (3) when the
 * device refuses the cooperative launch -- the caller then runs the multi-launch step.                                            */
int64_t trec_fit_step_coop_workspace_floats(int64_t n_users, int64_t n_items, int32_t d, int32_t n_sampled,
                                            int32_t max_interactions_per_user);
int trec_fit_step_coop(float* Wu, float* Wu_m, float* Wu_v, float* Wi, float* Wi_m, float* Wi_v, float* bu, float* bu_m, float* bu_v,
                       float* bi, float* bi_m, float* bi_v, const int64_t* f_indptr, const int32_t* f_indices, const float* f_values,
                       const int64_t* ft_indptr, const int32_t* ft_rows, const int32_t* ft_perm, const int64_t* indptr,
                       const int32_t* x_item, const int32_t* pos_slot, const float* pos_weight, const int32_t* samples, int64_t n_users,
                       int64_t n_items, int64_t n_item_features, int32_t d, int32_t n_sampled, int32_t max_interactions_per_user,
                       int64_t user_base, uint64_t seed, uint32_t step, float lr_t, float beta1, float beta2, float eps, float l2,
                       float* workspace, int64_t workspace_floats, float* loss, float* pred_serial, void* stream);
/* out[i] += sum of val over the pairs of two lists (either may be empty) whose id is i -- the item-bias gradient d b_i = sum g of
 * bias_prediction_serial (recommendation_graphs.py:44-57) under scores whose row gradient carries another coefficient.  out is
 * NOT cleared; ids outside [0, n_items) are skipped; the sums of an item are added in arrival order.                           */
int trec_item_weighted_hist(const int32_t* ids_a, const float* val_a, int64_t n_a, const int32_t* ids_b, const float* val_b,
                            int64_t n_b, int32_t n_items, float* out, void* stream);
/* RMSE, loss_graphs.py:58-59 */
/* Dense and separation losses (csrc/loss_dense.hip) -- tensorrec/loss_graphs.py:62-72 (RMSEDense), :75-97 (Separation), :100-134
 * (SeparationDense) as streaming reductions: kind 0 = Separation over the serial predictions (pred [n_pairs], values [n_pairs],
 * rows = n_pairs, cols = 1, no indices), 1 = SeparationDense, 2 = RMSEDense (pred [rows, cols] contiguous; xu / xi / values = the
 * interactions, one entry per cell).  st: double[16] statistics kept for the backward pass; loss: float[1].
 * trec_dense_loss_bwd: d loss / d pred (same shape as pred) times the upstream gradient gl[0]. */
int trec_dense_loss_fwd(int32_t kind, const float* pred, int64_t rows, int64_t cols, const int32_t* xu, const int32_t* xi,
                        const float* values, int64_t n_pairs, double* st, float* loss, void* stream);
/* the forward pass in the three phases a user-sharded fit adds the other ranks' sums between (phase 0: pass 1; 1: means + pass 2;
 * 2: loss + backward coefficients; the caller all-reduces st[0..9] after phases 0 and 1; n_all_total = dense predictions over
 * all ranks): the scalar losses of loss_graphs.py:58-134 on the UNION of the user shards (the batching axis of tensorrec.py:199-217) */
int trec_dense_loss_fwd_phase(int32_t kind, int32_t phase, const float* pred, int64_t rows, int64_t cols, const int32_t* xu,
                              const int32_t* xi, const float* values, int64_t n_pairs, int64_t n_all_total, double* st, float* loss,
                              void* stream);
int trec_dense_loss_bwd(int32_t kind, const float* pred, int64_t rows, int64_t cols, const int32_t* xu, const int32_t* xi,
                        const float* values, int64_t n_pairs, const double* st, const float* gl, float* d_pred, void* stream);
/* The dense losses WITHOUT the [n_users, n_items] prediction (csrc/loss_dense.hip, "factored"): dot-product scores are bilinear,
 * p_ui = x_u . y_i with x_u = [u | b_u | 1], y_i = [v_i | 1 | b_i], so RMSEDense / SeparationDense (loss_graphs.py:62-72, :100-134)
 * need sum p = (sum x) . (sum y) and sum p^2 = <X^T X, Y^T Y>_F only, and their gradient is A X (Y^T Y) + B 1 (sum y)^T.
 * trec_gram_f64: G [D, D] double = X^T X (X float [n, D], row stride ld; D <= 1024; G cleared inside).
 * trec_dense_loss_factored_phase: trec_dense_loss_fwd_phase with the dense sums given as m = {sum p, sum p^2} over this rank's
 * n_all_local predictions and the interactions given by their serial predictions (kind 1 / 2).
 * trec_dense_loss_factored_bwd: coef double[2] = {A, B}; d_serial [n_pairs] = the corrections at the interaction cells.      */
int trec_gram_f64(const float* X, int64_t n, int32_t D, int64_t ld, double* G, void* stream);
int trec_dense_loss_factored_phase(int32_t kind, int32_t phase, const double* m, const float* pred_serial, const float* values,
                                   int64_t n_pairs, int64_t n_all_local, int64_t n_all_total, double* st, float* loss,
                                   void* stream);
int trec_dense_loss_factored_bwd(int32_t kind, const float* pred_serial, const float* values, int64_t n_pairs, const double* st,
                                 const float* gl, float* d_serial, double* coef, void* stream);
int trec_rmse_fwd(const float* y, const float* pred, int64_t n, float* partial_ws, int32_t n_partial, float* loss,
                  void* stream);
int trec_rmse_bwd(const float* y, const float* pred, const float* loss, const float* grad_loss, int64_t n,
                  float* d_pred, void* stream);

/* ---- K7: negative sampling ----------------------------------------------------------------------------------
 * sample_items, util.py:12-21 (host np.random.choice per user behind tf.py_func, tensorrec.py:298-302).
 * out: int32 [n_users, n_sampled], user-major.  replace == 0: distinct per user (keyed permutation).  Row r is the
 * stream of GLOBAL user user_base + r, so user shards reproduce the whole-population draw.                      */
int trec_sample_items(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled, int32_t replace,
                      uint64_t seed, uint32_t step, int32_t* out, void* stream);

/* ---- K8: optimiser --------------------------------------------------------------------------------------------
 * tf.train.AdamOptimizer(lr).minimize (tensorrec.py:489) + gradient of alpha * sum(tf.nn.l2_loss(w)) (:487-488):
 * g' = grad + w * l2_coef, then the TF-1.x ApplyAdam element update.  In place on w, m, v.                    */
int trec_adam_tf_step(float* w, float* m, float* v, const float* grad, int64_t n, float lr_t, float beta1,
                      float beta2, float epsilon, float l2_coef, void* stream);

/* ---- input files ------------------------------------------------------------------------------------------------
 * CRC-32C (Castagnoli) of host bytes for the TFRecord framing that input_utils.py:103-105 / :141 delegate to
 * TensorFlow.  crc = 0 starts a checksum, or pass the previous return value to continue one.  Returns the checksum's 32
 * bits (as int).  HOST pointer, no stream: the only entry point that touches host memory.                          */
int trec_crc32c(const void* data, uint64_t n, uint32_t crc);

/* ---- K9: mixture of tastes -----------------------------------------------------------------------------------
 * collapse_mixture_of_tastes, recommendation_graphs.py:85-109 (tf.stack + reduce_max, or + softmax(axis=0) * predictions +
 * reduce_sum), fused with the bias add that follows it (bias_prediction_dense / _serial, :33-57; tensorrec.py:432-449).
 * preds / attn: fp32 [n_tastes, n] (attn NULL = max over tastes), out fp32 [n].
 * bias_mode 0: none; 1: serial pairs -- element e is (x_user[e] or e / span when x_user is NULL, x_item[e]);
 * 2: dense -- element e is (e / span, e % span) with span = n_items.  out = (collapsed + user_bias[u]) + item_bias[i].
 * _bwd: d_preds (and d_attn) [n_tastes, n] from grad_out [n]; reduce_max shares the gradient evenly between tied
 * tastes as TF does.  Bias gradients are grad_out summed per user / item (trec_spmv_csr on the pair structure).      */
int trec_collapse_tastes_fwd(const float* preds, const float* attn, int32_t n_tastes, int64_t n, int32_t bias_mode,
                             const float* user_bias, const float* item_bias, const int32_t* x_user,
                             const int32_t* x_item, int64_t span, float* out, void* stream);
int trec_collapse_tastes_bwd(const float* preds, const float* attn, const float* grad_out, int32_t n_tastes, int64_t n,
                             float* d_preds, float* d_attn, void* stream);

/* ---- whole steps as HIP graphs ----------------------------------------------------------------------------------
 * A captured graph is replayed every step, so what changes from step to step must live in device memory:
 * state = float[4] { beta1_power, beta2_power, lr_t, sample step (uint32 bits) }.  trec_adam_schedule_advance (first
 * node of the graph) multiplies the powers by beta (TF's float32 running powers), forms lr_t = lr * sqrt(1 - b2p) /
 * (1 - b1p) exactly as the host does, and bumps the sample step; trec_sample_items_dev / trec_adam_tf_step_dev are
 * trec_sample_items / trec_adam_tf_step reading step / lr_t from that state.                                       */
int trec_adam_schedule_advance(float* state, float learning_rate, float beta1, float beta2, int32_t bump_sample_step,
                               void* stream);
int trec_adam_tf_step_dev(float* w, float* m, float* v, const float* grad, int64_t n, const float* state, float beta1,
                          float beta2, float epsilon, float l2_coef, void* stream);
int trec_sample_items_dev(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled, int32_t replace,
                          uint64_t seed, const uint32_t* step_dev, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TENSORREC_HIP_H */
