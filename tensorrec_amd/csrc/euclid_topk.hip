// tensorrec_amd/csrc/euclid_topk.hip -- the exact Euclidean top-k through the DOT-product cascade, with a certificate per user.
//
// EuclideanSimilarityPredictionGraph (tensorrec/prediction_graphs.py:84-100): s(u, i) = -sqrt(max(r_u - 2 u.i + r_i, 1e-16)), then
// the biases (s + b_u) + b_i (tensorrec/recommendation_graphs.py:33-41), ranked by tf.nn.top_k (:73-82).  The superblock-maximum
// filters of the exact top-k work on dot products -- a maximum of integers (int8 stage) or of raw accumulators (bf16 stage) --
// and the square root with a per-item bias behind it does not commute with those maxima.  But for one user, ordering items by
// distance is ordering by  g(u, i) = u.i - r_i / 2  (r_u is a constant of the user): a dot product with the "item bias"
// -r_i / 2.  So:
//
//   1. the cascade (ops.score_topk_filtered) gives the exact top-K' of g, K' = 16 > k: the K' NEAREST items of every user;
//   2. trec_pair_score_exact re-scores those K' pairs with the reference's own chain (the oracle's bits, biases included);
//   3. euclid_certify_kernel (here) sorts them by (score desc, id asc), writes the first k, and CERTIFIES the user: every item
//      outside the K' has g <= G = the K'-th largest g, hence distance D >= r_u - 2 G - slack and score
//      <= -sqrt(max(r_u - 2 G - slack, 0)) + b_u + max_i b_i =: UB.  If the k-th best candidate score is STRICTLY above UB, no
//      outside item can enter the first k places (ties included), and the lists are the reference's.  Otherwise -- near-ties --
//      flag[u] = 1 and the caller re-does the user on the exact fp32 MFMA path.
//
// Item biases in the ordering (round 6).  With g alone the certificate fails as soon as item biases outweigh the distance gap
// between the k-th and the K'-th nearest item (sigma_b = 0.02 at distances ~16: 456 of 700 users re-done): the K' nearest are
// simply not the K' best.  The cascade therefore orders by  h(u, i) = u.i - r_i / 2 + lambda b_i  with ONE lambda >= 0 for
// everybody (the typical distance: d(-sqrt(D)) / d(u.i) = 1 / sqrt(D), so h ~ sqrt(D) x score) -- still a dot product with an
// item bias.  Any lambda is valid: an item outside the K' has h_i <= H, so D_i = r_u - 2 (h_i - lambda b_i) >= c + 2 lambda b_i
// with c = r_u - 2 H - slack, and its score is at most  b_u + phi(b_i),  phi(b) = b - sqrt(max(c + 2 lambda b, 0)).  phi is b
// itself below the kink b* = -c / (2 lambda) and convex above it, so over [min b, max b] its maximum sits at an end or at the
// kink: UB = b_u + max(phi(b_min), phi(b_max), b* if inside).  lambda = 0 gives the old bound back.
//
// slack covers every rounding between the real-valued g and what the kernels computed: the fp32 chain of u.i ((K + 2) 2^-24
// ||u|| ||v||), the addition of -r_i / 2 and the fp32 evaluation of r_i itself, the reference's three roundings of the distance
// and its correctly rounded sqrt; all of it is charged generously (x8) in double precision below.
#include "topk_common.hpp"

namespace {

constexpr int EC_MAX_ALL = 64;

// EC_MAX = candidate slots held in registers: 16 (k <= 12, the cascade's fused lists), 32 / 64 (k up to 48: the wide cascade's lists)
template <int EC_MAX>
__global__ __launch_bounds__(256) void euclid_certify_kernel(const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_g,
                                                            const float* __restrict__ exact, int kc, int k,
                                                            const float* __restrict__ user_sq, const float* __restrict__ user_bias,
                                                            const float* __restrict__ item_gstats, const float* __restrict__ bias_max,
                                                            int kdim, int64_t n_users, float* __restrict__ ov,
                                                            int32_t* __restrict__ oi, int32_t* __restrict__ flag,
                                                            int32_t* __restrict__ n_flagged, const float* __restrict__ lambda_,
                                                            const float* __restrict__ bias_min)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    unsigned long long key[EC_MAX];
    bool hole = false;
#pragma unroll
    for (int j = 0; j < EC_MAX; ++j) {
        key[j] = 0ull;
        if (j < kc) {
            const int32_t id = cand_idx[u * kc + j];
            if (id < 0) hole = true;
            else key[j] = merge_key(exact[u * kc + j], id);
        }
    }
    // insertion sort, descending keys = (score desc, id asc): kc <= EC_MAX entries in registers
#pragma unroll
    for (int a = 1; a < EC_MAX; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
            if (key[b] > key[b - 1]) { const unsigned long long t = key[b]; key[b] = key[b - 1]; key[b - 1] = t; }
        }
    }
    float tk = -INFINITY;
#pragma unroll
    for (int j = 0; j < EC_MAX; ++j) {
        if (j < k) {
            const unsigned int hi = (unsigned int)(key[j] >> 32);
            const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
            const bool empty = key[j] == 0ull;
            const float v = empty ? -INFINITY : __uint_as_float(bits);
            ov[u * k + j] = v;
            oi[u * k + j] = empty ? -1 : (int32_t)(~(unsigned int)key[j]);
            if (j == k - 1) tk = v;
        }
    }
    // ---- the certificate, in double
    const double ru = (double)user_sq[u];
    const double G = (double)cand_g[u * kc + kc - 1];                       // the K'-th largest g: every outside item has g <= G
    const double vmax = (double)item_gstats[0];                             // max ||v||
    const double rhalf = (double)item_gstats[2];                            // max |-r_i / 2| = max r_i / 2
    const double bmax = bias_max ? (double)bias_max[0] : 0.0;
    const double bu = user_bias ? (double)user_bias[u] : 0.0;
    const double ulp = 5.9604644775390625e-08;                              // 2^-24
    const double lam = (lambda_ && bias_max) ? (double)lambda_[0] : 0.0;    // weight of the item bias in the ordering (>= 0)
    const double bmin = (bias_min && bias_max) ? (double)bias_min[0] : bmax;
    const double babs = fmax(fabs(bmax), fabs(bmin));
    const double mag = sqrt(ru) * vmax + rhalf + fabs(G) + ru + lam * babs;
    const double slack = 8.0 * (double)(kdim + 8) * ulp * mag;
    const double c = ru - 2.0 * G - 2.0 * slack;                            // D_i >= c + 2 lambda b_i for every item outside the K'
    auto phi = [&](double b) {
        double dlo = c + 2.0 * lam * b;
        if (dlo < 0.0) dlo = 0.0;
        return b - sqrt(dlo) * (1.0 - 4.0 * ulp);
    };
    double best = fmax(phi(bmax), phi(bmin));
    if (lam > 0.0) {
        const double bk = -c / (2.0 * lam);                                 // the kink: below it phi(b) = b
        if (bk > bmin && bk < bmax) best = fmax(best, bk);
    }
    double dref = c + 2.0 * lam * bmax;
    if (dref < 0.0) dref = 0.0;
    double ub = best + bu;
    ub += 8.0 * ulp * (sqrt(dref) + sqrt(c > 0.0 ? c : 0.0) + fabs(bu) + babs) + 1e-30;
    if (!(lam >= 0.0) || lam != lam) ub = INFINITY;                         // (a negative or NaN weight certifies nothing)
    const bool ok = !hole && (double)tk > ub && tk == tk && G == G && ub == ub;
    flag[u] = ok ? 0 : 1;
    if (!ok) atomicAdd(n_flagged, 1);
}

}  // namespace

// cand_idx / cand_g [n_users, kc]: the kc = K' largest g(u, i) = u.i - r_i / 2 per user (ids, values; descending) from the
// dot-product cascade; exact [n_users, kc]: the reference-chain Euclidean scores (+ biases) of those pairs
// (trec_pair_score_exact); user_sq [n_users] = r_u as the score kernels use it; item_gstats [3] = the item operand's maxima
// {||v||, -, max r_i / 2} (trec_score_prep_filter with bias = -r / 2); bias_max [1] = max item bias (NULL: none).
// Writes the first k by (score desc, id asc); flag[u] = 1 (n_flagged [1], zeroed by the caller, counts them) when the certificate
// does not hold.  kc <= 64, k <= kc.  lambda_ [1] / bias_min [1] (nullable: 0 / = bias_max): the cascade ordered by
// u.i - r_i / 2 + lambda b_i -- cand_g holds those values, item_gstats[2] their largest |item term|.
extern "C" int trec_topk_euclid_certify(const int32_t* cand_idx, const float* cand_g, const float* exact, int32_t kc, int32_t k,
                                        const float* user_sq, const float* user_bias, const float* item_gstats,
                                        const float* bias_max, int32_t kdim, int64_t n_users, float* out_vals, int32_t* out_idx,
                                        int32_t* flag, int32_t* n_flagged, const float* lambda_, const float* bias_min, void* stream)
{
    TREC_REQUIRE(cand_idx && cand_g && exact && user_sq && item_gstats && out_vals && out_idx && flag && n_flagged,
                 "trec_topk_euclid_certify: null pointer");
    TREC_REQUIRE(kc >= 1 && kc <= EC_MAX_ALL && k >= 1 && k <= kc && kdim >= 1, "trec_topk_euclid_certify: need 1 <= k <= kc <= 64");
    if (n_users == 0) return TREC_OK;
#define TREC_EC(M)                                                                                                                  \
    hipLaunchKernelGGL(euclid_certify_kernel<M>, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream, cand_idx, \
                       cand_g, exact, kc, k, user_sq, user_bias, item_gstats, bias_max, kdim, n_users, out_vals, out_idx, flag,       \
                       n_flagged, lambda_, bias_min)
    if (kc <= 16) TREC_EC(16);
    else if (kc <= 32) TREC_EC(32);
    else TREC_EC(64);
#undef TREC_EC
    return trec_check_launch("trec_topk_euclid_certify");
}
