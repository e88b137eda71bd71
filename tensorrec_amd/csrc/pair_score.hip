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

// PP pairs per subgroup: the PP index loads, then the 2*PP row gathers are issued back to back, so a subgroup keeps PP
// random 512-byte rows in flight instead of one (the sampled pairs are HBM-latency-bound otherwise: 100M random item
// rows per step at the BASELINE fit shape).  Per pair the arithmetic is unchanged: one fmaf chain over the lane's
// columns in increasing order, then the xor-butterfly.
// UG > 0: implicit users with pairs_per_user % UG == 0 -- every aligned group of UG pairs belongs to ONE user, whose
// row is loaded once per group (a quarter of the cache traffic of the sampled pairs at UG = 4).
template <int VEC, int PP, int UG = 0>
__global__ __launch_bounds__(256) void pair_score_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const int32_t* __restrict__ xu,
    const int32_t* __restrict__ xi, int64_t n_pairs, int32_t pairs_per_user, int d, int lpr_log2, int mode,
    const float* __restrict__ ub, const float* __restrict__ ib, float* __restrict__ out, float* __restrict__ out_acc)
{
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t p0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2) * PP;
    if (p0 >= n_pairs) return;
    int64_t u[PP], i[PP];
    bool ok[PP];
#pragma unroll
    for (int r = 0; r < PP; ++r) {
        const int64_t p = p0 + r;
        ok[r] = p < n_pairs;
        const int64_t pc = ok[r] ? p : n_pairs - 1;       // (unconditional loads from a clamped pair: `ok ? xi[p] : 0` made every
        i[r] = (int64_t)xi[pc];                            // index load its own block behind s_waitcnt vmcnt(0))
        u[r] = xu ? (int64_t)xu[pc] : pc / pairs_per_user;
    }
    float acc[PP];
#pragma unroll
    for (int r = 0; r < PP; ++r) acc[r] = 0.f;
    if (VEC == 4) {
        for (int c = sub_lane * 4; c < d; c += lpr * 4) {
            f32x4 x[PP], y[PP];
#pragma unroll
            for (int r = 0; r < PP; ++r) {
                if (UG == 0 || r % (UG ? UG : 1) == 0) x[r] = *(const f32x4*)(U + u[r] * d + c);
                else x[r] = x[r - r % (UG ? UG : 1)];
                y[r] = *(const f32x4*)(V + i[r] * d + c);
            }
#pragma unroll
            for (int r = 0; r < PP; ++r) {
                if (mode == MODE_DOT) {
                    acc[r] = fmaf(x[r].x, y[r].x, acc[r]); acc[r] = fmaf(x[r].y, y[r].y, acc[r]);
                    acc[r] = fmaf(x[r].z, y[r].z, acc[r]); acc[r] = fmaf(x[r].w, y[r].w, acc[r]);
                } else {
                    const float d0 = x[r].x - y[r].x, d1 = x[r].y - y[r].y, d2 = x[r].z - y[r].z, d3 = x[r].w - y[r].w;
                    acc[r] = fmaf(d0, d0, acc[r]); acc[r] = fmaf(d1, d1, acc[r]);
                    acc[r] = fmaf(d2, d2, acc[r]); acc[r] = fmaf(d3, d3, acc[r]);
                }
            }
        }
    } else {
        for (int c = sub_lane; c < d; c += lpr) {
#pragma unroll
            for (int r = 0; r < PP; ++r) {
                const float a = U[u[r] * d + c], b = V[i[r] * d + c];
                if (mode == MODE_DOT) acc[r] = fmaf(a, b, acc[r]);
                else { const float df = a - b; acc[r] = fmaf(df, df, acc[r]); }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PP; ++r)
        for (int off = lpr >> 1; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
    if (sub_lane == 0) {
#pragma unroll
        for (int r = 0; r < PP; ++r) {
            if (!ok[r]) continue;
            float s = acc[r];
            if (out_acc) out_acc[p0 + r] = acc[r];           // squared distance: the backward pass needs it unclamped
            if (mode == MODE_EUCLID) s = -1.0f * sqrtf(fmaxf(acc[r], EUCLID_EPS));
            if (ub) s = s + ub[u[r]];
            if (ib) s = s + ib[i[r]];
            out[p0 + r] = s;
        }
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

// c_p = dS_p/dD-chain coefficient of a Euclidean pair: s = -sqrt(max(D, eps)), dS/du = -(u - v)/sqrt(D) -> with the
// upstream gradient g_p:  dU[u] += c_p (u - v), dV[i] -= c_p (u - v), c_p = -g_p / sqrt(D_p) (0 where D_p was clamped).
// Same arithmetic as pair_score_bwd_kernel; the accumulation itself then runs as a segmented gather (spmm_split.hip).
__global__ __launch_bounds__(256) void pair_euclid_coef_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const int32_t* __restrict__ xu,
    const int32_t* __restrict__ xi, const float* __restrict__ g, int64_t n_pairs, int32_t pairs_per_user, int d,
    int lpr_log2, float* __restrict__ coef)
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
    for (int c = sub_lane; c < d; c += lpr) { const float df = a[c] - b[c]; acc = fmaf(df, df, acc); }
    for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub_lane == 0) coef[p] = (acc >= EUCLID_EPS) ? -g[p] / sqrtf(acc) : 0.f;
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
                                   const float* user_bias, const float* item_bias, float* out, float* out_sqdist,
                                   void* stream)
{
    TREC_REQUIRE(U && V && xi && out, "trec_pair_score_fwd: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_pair_score_fwd: need xu or pairs_per_user");
    TREC_REQUIRE(mode == MODE_DOT || mode == MODE_EUCLID, "trec_pair_score_fwd: mode must be 0 (dot) or 1 (euclidean)");
    TREC_REQUIRE(d >= 1, "trec_pair_score_fwd: d must be >= 1");
    if (n_pairs == 0) return TREC_OK;
    int vec, l2; pair_geometry(d, vec, l2);
    // measured at the BASELINE fit shape (rocprofv3, 100M sampled pairs / 20M interactions, d = 128): sampled pairs
    // 17.3 ms at 1 pair per subgroup, 13.9 at 2, 15.4 at 4, 12.4 at 4 with the shared user row; interactions 3.9 / 3.1 / 3.6
    int pp = (n_pairs >= 65536) ? trec_get_tuning("pair_fwd_pp", (!xu && pairs_per_user % 4 == 0) ? 4 : 2) : 1;
    pp = (pp >= 4 && vec == 4) ? 4 : (pp >= 2 ? 2 : 1);
    const unsigned blocks = (unsigned)ceil_div64(ceil_div64(n_pairs, pp) << l2, 256);
#define TREC_PAIR_FWD(VECV, PPV)                                                                                    \
    hipLaunchKernelGGL((pair_score_fwd_kernel<VECV, PPV>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, xu, \
                       xi, n_pairs, pairs_per_user, d, l2, mode, user_bias, item_bias, out, out_sqdist)
    const int ug = trec_get_tuning("pair_fwd_user_group", 1);       // 1: share the user row inside a subgroup's pairs
    if (!xu && vec == 4 && ug && pp >= 2 && pairs_per_user % pp == 0) {
        if (pp == 4)
            hipLaunchKernelGGL((pair_score_fwd_kernel<4, 4, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V,
                               xu, xi, n_pairs, pairs_per_user, d, l2, mode, user_bias, item_bias, out, out_sqdist);
        else
            hipLaunchKernelGGL((pair_score_fwd_kernel<4, 2, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V,
                               xu, xi, n_pairs, pairs_per_user, d, l2, mode, user_bias, item_bias, out, out_sqdist);
    } else if (vec == 4 && pp == 4) TREC_PAIR_FWD(4, 4);
    else if (vec == 4 && pp == 2) TREC_PAIR_FWD(4, 2);
    else if (vec == 4) TREC_PAIR_FWD(4, 1);
    else if (pp >= 2) TREC_PAIR_FWD(1, 2);
    else TREC_PAIR_FWD(1, 1);
#undef TREC_PAIR_FWD
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

__global__ __launch_bounds__(256) void euclid_coef_from_sqdist_kernel(const float* __restrict__ sqdist,
                                                                     const float* __restrict__ g, int64_t n,
                                                                     float* __restrict__ coef)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float D = sqdist[p];
    coef[p] = (D >= EUCLID_EPS) ? -g[p] / sqrtf(D) : 0.f;
}

extern "C" int trec_pair_euclid_coef(const float* U, const float* V, const int32_t* xu, const int32_t* xi,
                                     const float* grad, const float* sqdist, int64_t n_pairs, int32_t pairs_per_user,
                                     int32_t d, float* coef, void* stream)
{
    TREC_REQUIRE(grad && coef, "trec_pair_euclid_coef: null pointer");
    if (sqdist) {            // the forward pass kept the squared distances (trec_pair_score_fwd out_sqdist)
        if (n_pairs == 0) return TREC_OK;
        hipLaunchKernelGGL(euclid_coef_from_sqdist_kernel, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0,
                           (hipStream_t)stream, sqdist, grad, n_pairs, coef);
        return trec_check_launch("trec_pair_euclid_coef");
    }
    TREC_REQUIRE(U && V && xi, "trec_pair_euclid_coef: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_pair_euclid_coef: need xu or pairs_per_user");
    TREC_REQUIRE(d >= 1, "trec_pair_euclid_coef: d must be >= 1");
    if (n_pairs == 0) return TREC_OK;
    int vec, l2; pair_geometry(d, vec, l2);
    hipLaunchKernelGGL(pair_euclid_coef_kernel, dim3((unsigned)ceil_div64(n_pairs << l2, 256)), dim3(256), 0,
                       (hipStream_t)stream, U, V, xu, xi, grad, n_pairs, pairs_per_user, d, l2, coef);
    return trec_check_launch("trec_pair_euclid_coef");
}
