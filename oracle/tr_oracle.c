/*
 * oracle/tr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the arithmetic on TensorRec's scoring/training hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; tensorrec_amd/ never does.
 *
 * The reference (jfkirk/tensorrec v0.26.2) is pure Python over TensorFlow 1.x ops;
 * TensorFlow is neither vendored under /root/reference nor installable here, so the
 * TF kernels are restated from their documented semantics and pinned against the
 * reference's own known-answer tests (tests/golden/, see oracle/README.md).
 *
 * Floating-point contract of THIS file (what "bit-exact" means in the parity tests):
 *   - every multiply-accumulate is one fmaf(), accumulated in the order written here
 *     (k = 0..d-1 for contractions, CSR order for SpMM);
 *   - nothing is re-associated: build with -ffp-contract=off and no -ffast-math.
 * The HIP fp32 kernels are written to the same order, so fp32 results can be
 * compared for equality, not just within tolerance.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- *
 * tf.sparse_tensor_dense_matmul(X, W)
 *   reference call sites: tensorrec/representation_graphs.py:40 (Linear),
 *   :119 (ReLU first layer), tensorrec/recommendation_graphs.py:15 (biases).
 * X is CSR (indptr int64, indices int32, values fp32), W is [n_features, d].
 * out[r,:] = sum_j X[r,j] * W[j,:], accumulated in CSR (row-major COO) order.
 * val_perm (nullable) indirects the value array: value of entry j = values[val_perm[j]]
 * (used for the transposed operand in the backward pass).
 * ------------------------------------------------------------------------- */
void orc_spmm_csr(const int64_t *indptr, const int32_t *indices, const float *values,
                  const int32_t *val_perm, int64_t n_rows, const float *W, int32_t d, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        float *o = out + r * (int64_t)d;
        for (int32_t c = 0; c < d; ++c) o[c] = 0.0f;
        for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
            const float v = values[val_perm ? val_perm[j] : j];
            const float *w = W + (int64_t)indices[j] * d;
            for (int32_t c = 0; c < d; ++c) o[c] = fmaf(v, w[c], o[c]);
        }
    }
}

/* ------------------------------------------------------------------------- *
 * tf.matmul(U, V, transpose_b=True)  -- DotProductPredictionGraph dense,
 *   tensorrec/prediction_graphs.py:49-50; also relative_cosine
 *   (recommendation_graphs.py:121) after row normalisation.
 * out[u,i] = fmaf chain over k = 0..d-1 starting from 0.
 * Optional biases follow bias_prediction_dense (recommendation_graphs.py:41):
 *   (pred + user_bias[u]) + item_bias[i], in that order.
 * ------------------------------------------------------------------------- */
void orc_score_dense(const float *U, const float *V, int64_t n_users, int64_t n_items, int32_t d,
                     const float *user_bias, const float *item_bias, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t u = 0; u < n_users; ++u) {
        const float *a = U + u * (int64_t)d;
        for (int64_t i = 0; i < n_items; ++i) {
            const float *b = V + i * (int64_t)d;
            float acc = 0.0f;
            for (int32_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
            if (user_bias) acc = acc + user_bias[u];
            if (item_bias) acc = acc + item_bias[i];
            out[u * n_items + i] = acc;
        }
    }
}

/* ------------------------------------------------------------------------- *
 * EuclideanSimilarityPredictionGraph dense, tensorrec/prediction_graphs.py:84-100:
 *   distance = (r_user - 2.0 * U.V^T) + r_item ; max(distance, 1e-16) ; -sqrt
 * r_user / r_item are passed in (sum of squares per row, computed by the caller).
 * ------------------------------------------------------------------------- */
void orc_score_dense_euclid(const float *U, const float *V, int64_t n_users, int64_t n_items, int32_t d,
                            const float *r_user, const float *r_item,
                            const float *user_bias, const float *item_bias, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t u = 0; u < n_users; ++u) {
        const float *a = U + u * (int64_t)d;
        for (int64_t i = 0; i < n_items; ++i) {
            const float *b = V + i * (int64_t)d;
            float acc = 0.0f;
            for (int32_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
            float dist = (r_user[u] - 2.0f * acc) + r_item[i];
            dist = dist > 1e-16f ? dist : 1e-16f;
            float s = -1.0f * sqrtf(dist);
            if (user_bias) s = s + user_bias[u];
            if (item_bias) s = s + item_bias[i];
            out[u * n_items + i] = s;
        }
    }
}

/* ------------------------------------------------------------------------- *
 * rank_predictions, tensorrec/recommendation_graphs.py:73-82:
 *   idx   = tf.nn.top_k(pred, k=I)[1]      (descending; equal values -> lower index first)
 *   ranks = tf.nn.top_k(-idx, k=I)[1] + 1  (position of every item in idx, +1)
 * Restated literally: a stable descending sort, then the inverse permutation.
 * Output int32 (TF top_k index dtype).
 * ------------------------------------------------------------------------- */
typedef struct { float s; int32_t i; } orc_si_t;

static int orc_cmp_desc_stable(const void *pa, const void *pb)
{
    const orc_si_t *a = (const orc_si_t *)pa, *b = (const orc_si_t *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

void orc_rank_rows(const float *scores, int64_t n_users, int64_t n_items, int32_t *ranks)
{
#pragma omp parallel
    {
        orc_si_t *buf = (orc_si_t *)malloc(sizeof(orc_si_t) * (size_t)(n_items > 0 ? n_items : 1));
#pragma omp for schedule(static)
        for (int64_t u = 0; u < n_users; ++u) {
            const float *s = scores + u * n_items;
            for (int64_t i = 0; i < n_items; ++i) { buf[i].s = s[i]; buf[i].i = (int32_t)i; }
            qsort(buf, (size_t)n_items, sizeof(orc_si_t), orc_cmp_desc_stable);  /* first top_k  */
            for (int64_t p = 0; p < n_items; ++p)                               /* second top_k */
                ranks[u * n_items + buf[p].i] = (int32_t)(p + 1);
        }
        free(buf);
    }
}

/* Top-k of every row: (value desc, index asc), the first k entries of the first
 * top_k above.  Rows shorter than k are padded with (-inf, -1). */
void orc_topk_rows(const float *scores, int64_t n_users, int64_t n_items, int32_t k,
                   float *out_vals, int32_t *out_idx)
{
#pragma omp parallel
    {
        orc_si_t *buf = (orc_si_t *)malloc(sizeof(orc_si_t) * (size_t)(n_items > 0 ? n_items : 1));
#pragma omp for schedule(static)
        for (int64_t u = 0; u < n_users; ++u) {
            const float *s = scores + u * n_items;
            for (int64_t i = 0; i < n_items; ++i) { buf[i].s = s[i]; buf[i].i = (int32_t)i; }
            qsort(buf, (size_t)n_items, sizeof(orc_si_t), orc_cmp_desc_stable);
            for (int32_t j = 0; j < k; ++j) {
                if (j < n_items) { out_vals[u * k + j] = buf[j].s; out_idx[u * k + j] = buf[j].i; }
                else { out_vals[u * k + j] = -INFINITY; out_idx[u * k + j] = -1; }
            }
        }
        free(buf);
    }
}

/* ------------------------------------------------------------------------- *
 * Serial (per-pair) dot product, DotProductPredictionGraph serial,
 *   tensorrec/prediction_graphs.py:52-55, + bias_prediction_serial
 *   (recommendation_graphs.py:55-57): (dot + ub[xu]) + ib[xi].
 * ------------------------------------------------------------------------- */
void orc_pair_dot(const float *U, const float *V, const int32_t *xu, const int32_t *xi, int64_t n_pairs,
                  int32_t d, const float *user_bias, const float *item_bias, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n_pairs; ++p) {
        const float *a = U + (int64_t)xu[p] * d, *b = V + (int64_t)xi[p] * d;
        float acc = 0.0f;
        for (int32_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
        if (user_bias) acc = acc + user_bias[xu[p]];
        if (item_bias) acc = acc + item_bias[xi[p]];
        out[p] = acc;
    }
}

int orc_abi_version(void) { return 1; }
