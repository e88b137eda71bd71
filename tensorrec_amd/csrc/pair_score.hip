// tensorrec_amd/csrc/pair_score.hip -- K3: per-pair ("serial") prediction scores and their backward.
//
// Replaces the tf.gather + multiply + reduce_sum chains at tensorrec/prediction_graphs.py:52-55 (dot),
// :70-72 (cosine, after the row normalisation done by trec_row_l2norm_fwd) and :105-117 (euclidean), fused
// with bias_prediction_serial (tensorrec/recommendation_graphs.py:55-57).  It is used twice per training
// step: over the P interactions and over the U*S sampled (user, item) pairs (tensorrec.py:384-395).
//
// HBM/L2-bound gather: a subgroup of LPR lanes owns one pair, reads both d-wide rows with float4 loads and
// reduces with __shfl_xor.  If `xu` is NULL the user of pair p is p / pairs_per_user (the user-major
// [U, S] sample layout of util.sample_items, util.py:16-19).
//
// Backward (gradients TF would scatter-add): dU[xu] += g * dS/dU, dV[xi] += g * dS/dV with fp32 atomics
// (global_atomic_add_f32).  Summation order is therefore not fixed: results agree with the oracle within
// fp32 round-off, not bit-for-bit (DESIGN.md section "determinism").
#include "common.hpp"

#define MODE_DOT 0
#define MODE_EUCLID 1
#define EUCLID_EPS 1e-16f

template <int VEC>
__global__ __launch_bounds__(256) void pair_score_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const int32_t* __restrict__ xu,
    const int32_t* __restrict__ xi, int64_t n_pairs, int32_t pairs_per_user, int d, int lpr_log2, int mode,
    const float* __restrict__ ub, const float* __restrict__ ib, float* __restrict__ out)
{
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2;
    if (p >= n_pairs) return;
    const int64_t u = xu ? (int64_t)xu[p] : p / pairs_per_user;
    const int64_t i = xi[p];
    const float* a = U + u * d;
    const float* b = V + i * d;
    float acc = 0.f;
    if (VEC == 4) {
        for (int c = sub_lane * 4; c < d; c += lpr * 4) {
            const f32x4 x = *(const f32x4*)(a + c), y = *(const f32x4*)(b + c);
            if (mode == MODE_DOT) {
                acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
            } else {
                const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                acc = fmaf(d0, d0, acc); acc = fmaf(d1, d1, acc); acc = fmaf(d2, d2, acc); acc = fmaf(d3, d3, acc);
            }
        }
    } else {
        for (int c = sub_lane; c < d; c += lpr) {
            if (mode == MODE_DOT) acc = fmaf(a[c], b[c], acc);
            else { const float df = a[c] - b[c]; acc = fmaf(df, df, acc); }
        }
    }
    for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub_lane == 0) {
        float s = acc;
        if (mode == MODE_EUCLID) s = -1.0f * sqrtf(fmaxf(acc, EUCLID_EPS));
        if (ub) s = s + ub[u];
        if (ib) s = s + ib[i];
        out[p] = s;
    }
}

// g = dL/ds.  dot: dU += g*V, dV += g*U.  euclid: s = -sqrt(D), dS/dU = -(u - v)/sqrt(D) = (u - v)/s_raw  (0 when
// D was clamped: tf.maximum passes the gradient to the constant side only when D < eps).  Bias gradients: += g.
template <int VEC>
__global__ __launch_bounds__(256) void pair_score_bwd_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const int32_t* __restrict__ xu,
    const int32_t* __restrict__ xi, const float* __restrict__ g, int64_t n_pairs, int32_t pairs_per_user, int d,
    int lpr_log2, int mode, float* __restrict__ dU, float* __restrict__ dV, float* __restrict__ dub,
    float* __restrict__ dib)
{
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2;
    if (p >= n_pairs) return;
    const int64_t u = xu ? (int64_t)xu[p] : p / pairs_per_user;
    const int64_t i = xi[p];
    const float gp = g[p];
    const float* a = U + u * d;
    const float* b = V + i * d;
    float coef = gp;
    if (mode == MODE_EUCLID) {
        float acc = 0.f;
        for (int c = sub_lane; c < d; c += lpr) { const float df = a[c] - b[c]; acc = fmaf(df, df, acc); }
        for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        coef = (acc >= EUCLID_EPS) ? -gp / sqrtf(acc) : 0.f;      // d(-sqrt(D))/dD * dD/du = -(u-v)/sqrt(D)
    }
    if (sub_lane == 0) {
        if (dub) atomicAdd(dub + u, gp);
        if (dib) atomicAdd(dib + i, gp);
    }
    if (coef == 0.f) return;
    for (int c = sub_lane * VEC; c < d; c += lpr * VEC) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float x = a[c + e], y = b[c + e];
            if (mode == MODE_DOT) {
                if (dU) atomicAdd(dU + u * d + c + e, coef * y);
                if (dV) atomicAdd(dV + i * d + c + e, coef * x);
            } else {
                const float t = coef * (x - y);
                if (dU) atomicAdd(dU + u * d + c + e, t);
                if (dV) atomicAdd(dV + i * d + c + e, -t);
            }
        }
    }
}

static void pair_geometry(int d, int& vec, int& lpr_log2)
{
    vec = (d % 4 == 0) ? 4 : 1;
    const int per = (d + vec - 1) / vec;
    lpr_log2 = 0;
    while ((1 << lpr_log2) < per && lpr_log2 < 6) ++lpr_log2;
}

extern "C" int trec_pair_score_fwd(const float* U, const float* V, const int32_t* xu, const int32_t* xi,
                                   int64_t n_pairs, int32_t pairs_per_user, int32_t d, int32_t mode,
                                   const float* user_bias, const float* item_bias, float* out, void* stream)
{
    TREC_REQUIRE(U && V && xi && out, "trec_pair_score_fwd: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_pair_score_fwd: need xu or pairs_per_user");
    TREC_REQUIRE(mode == MODE_DOT || mode == MODE_EUCLID, "trec_pair_score_fwd: mode must be 0 (dot) or 1 (euclidean)");
    TREC_REQUIRE(d >= 1, "trec_pair_score_fwd: d must be >= 1");
    if (n_pairs == 0) return TREC_OK;
    int vec, l2; pair_geometry(d, vec, l2);
    const unsigned blocks = (unsigned)ceil_div64(n_pairs << l2, 256);
    if (vec == 4)
        hipLaunchKernelGGL((pair_score_fwd_kernel<4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, xu, xi,
                           n_pairs, pairs_per_user, d, l2, mode, user_bias, item_bias, out);
    else
        hipLaunchKernelGGL((pair_score_fwd_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, xu, xi,
                           n_pairs, pairs_per_user, d, l2, mode, user_bias, item_bias, out);
    return trec_check_launch("trec_pair_score_fwd");
}

extern "C" int trec_pair_score_bwd(const float* U, const float* V, const int32_t* xu, const int32_t* xi,
                                   const float* grad, int64_t n_pairs, int32_t pairs_per_user, int32_t d,
                                   int32_t mode, float* dU, float* dV, float* d_user_bias, float* d_item_bias,
                                   void* stream)
{
    TREC_REQUIRE(U && V && xi && grad, "trec_pair_score_bwd: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_pair_score_bwd: need xu or pairs_per_user");
    TREC_REQUIRE(mode == MODE_DOT || mode == MODE_EUCLID, "trec_pair_score_bwd: mode must be 0 or 1");
    if (n_pairs == 0) return TREC_OK;
    int vec, l2; pair_geometry(d, vec, l2);
    const unsigned blocks = (unsigned)ceil_div64(n_pairs << l2, 256);
    if (vec == 4)
        hipLaunchKernelGGL((pair_score_bwd_kernel<4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, xu, xi,
                           grad, n_pairs, pairs_per_user, d, l2, mode, dU, dV, d_user_bias, d_item_bias);
    else
        hipLaunchKernelGGL((pair_score_bwd_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, xu, xi,
                           grad, n_pairs, pairs_per_user, d, l2, mode, dU, dV, d_user_bias, d_item_bias);
    return trec_check_launch("trec_pair_score_bwd");
}
