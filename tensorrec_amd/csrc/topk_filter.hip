// tensorrec_amd/csrc/topk_filter.hip -- K2f: the EXACT fp32 top-k at bf16-MFMA speed.
//
// The reference contracts user x item scores in float32 (tf.matmul, tensorrec/prediction_graphs.py:49-50; operands are
// float32, tensorrec/input_utils.py:16) and ranks them with tf.nn.top_k (tensorrec/recommendation_graphs.py:80).  The
// bf16 MFMA stage-1 kernel is 16x faster than the fp32 MFMA one but its scores differ from the fp32 ones by ~1e-3
// relative, so on its own it misses ~0.6% of the top-10 slots.  Here it is used as a FILTER with a proven error bound,
// and the survivors are re-scored exactly:
//
//   prep     trec_score_prep_filter: per operand row x (fp32, optionally l2-normalised) the bf16 image xh = bf16(x),
//            |x| = ||x||_2 and |dx| = ||x - xh||_2 (the ACTUAL rounding error of this row, not the worst case), and
//            device-side maxima of |x|, |dx|, |bias| over the item rows;
//   stage 1  bf16 superblock maxima  Mh[s][u]  (score_blockmax.hip, unchanged);
//   stage 2  trec_topk_select_blocks with a list of KSEL >= k entries;  tau_u = k-th largest Mh[.][u];
//            trec_topk_filter_floor:   eps_u >= |sh(u,i) - s(u,i)| for EVERY item i (bound below);
//                                      floor_u = tau_u - 2 eps_u;  superblocks with Mh >= floor_u are kept;
//   stage 3  grouped bf16 re-scoring (score_gemm.hip, lists made independent): every item with sh >= floor_u;
//   stage 4  trec_topk_filter_finish: the surviving items are re-scored in fp32 by the k-ordered fmaf chain of
//            oracle/tr_oracle.c:orc_score_dense (+ biases in the reference's order (s + b_u) + b_i) and the k best by
//            (value desc, index asc) are written -- bit-identical to the fp32 MFMA path and to the oracle.
//
// Why nothing of the true top-k is lost.  Let s = the fp32 score the reference order gives, sh = the bf16-path score,
// |sh - s| <= eps for all items of user u.  k superblocks have Mh >= tau, each holds an item with sh >= tau, hence
// s >= tau - eps: k distinct items, so the true k-th best t_k >= tau - eps.  An item of the true top-k (ties included:
// the order is total) has s >= t_k, hence sh >= s - eps >= tau - 2 eps = floor: it sits in a kept superblock and passes
// the stage-3 threshold.  Its exact fp32 score is then computed in stage 4 together with every other survivor, and the
// exact order among the survivors is the exact order among all items for the first k places.
// Capacity limits (KSEL superblocks per user, `capacity` entries per (user, superblock, half-wave) list, 64 survivors
// per user) cannot lose an item silently: a saturated selection list whose last entry still passes the floor, a full
// stage-3 list, more than 64 survivors, or a non-finite bound set flag[u], and the host re-does flagged users on the
// exact fp32 MFMA path (ops.score_topk_filtered).
//
// The bound.  x = user operand row, y = item operand row (fp32), xh / yh their bf16 images, X = sum_k x_k y_k in real
// arithmetic.  |sum xh yh - X| = |<xh - x, yh> + <x, yh - y>| <= |dx| |yh| + |x| |dy|  (Cauchy-Schwarz).  The fp32
// reference chain errs by <= (K+2) 2^-24 (|x||y| + |b_u| + |b_i|) from X + b_u + b_i; the MFMA chain (bf16 products are
// exact in fp32; K/16 chained 16-term blocks + the bias accumulator + one add) is charged 2^-22 per addition -- four
// times round-to-nearest, covering truncating adders.  Every norm is computed in fp32 (relative error < K 2^-23) and
// the sum is inflated by 2^-9; 1e-30 absorbs flushed denormals.  tests/test_gpu_filter.py measures max |sh - s| / eps
// on random, adversarially aligned and near-tied inputs.
#include "topk_common.hpp"
#include "score_common.hpp"

#define FILTER_CMAX 64          // survivors per user the finish kernel can re-score (one per lane)

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v)
{
    // v >= 0 (or NaN, which must win: it poisons the bound and flags every user): uint order == float order
    // read first: after the first few waves almost nobody raises the maximum, and 500k same-address atomics would cost
    // milliseconds (a stale L1 line only means one unnecessary atomic)
    if (__float_as_uint(v) > *(volatile unsigned int*)addr) atomicMax((unsigned int*)addr, __float_as_uint(v));
}

// sum over an aligned 32-lane group with DPP adds (no LDS crossbar): four rotations inside the 16-lane rows, then
// row_bcast15 carries the first row's total into the second -- complete in lanes 16..31 of the group (cf. wmrb_fused.hip)
__device__ __forceinline__ float dpp_sum32_upper(float x)
{
    int v, r;
    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); x += __int_as_float(r);
    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); x += __int_as_float(r);
    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); x += __int_as_float(r);
    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); x += __int_as_float(r);
    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); x += __int_as_float(r);
    return x;
}

// G lanes own one row; a lane handles float4 chunks (kt <= 256 = 2 * 32 lanes * 4).  Same normalisation arithmetic
// as score_prep_vec4_kernel (score_gemm.hip), so out_f32 / out_bf16 are bit-identical to trec_score_prep's outputs.
template <int G>
__global__ __launch_bounds__(256) void prep_filter_kernel(const float* __restrict__ x, int64_t n, int d, int kt,
                                                         int normalize, const float* __restrict__ bias,
                                                         float* __restrict__ out_f32, unsigned short* __restrict__ out_bf16,
                                                         float2* __restrict__ row_stats, float* __restrict__ gstats)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool ok = row < n;
    const int sub = threadIdx.x % G;
    const float* xr = x + (ok ? row : 0) * (int64_t)d;
    f32x4 v[2];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok && c < d) {
            if ((d & 3) == 0) v[it] = *(const f32x4*)(xr + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) v[it][e] = xr[c + e];
            }
        }
        ss = fmaf(v[it][0], v[it][0], ss); ss = fmaf(v[it][1], v[it][1], ss);
        ss = fmaf(v[it][2], v[it][2], ss); ss = fmaf(v[it][3], v[it][3], ss);
    }
    float scale = 1.0f;
    if (normalize) {
        for (int off = G / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        scale = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    }
    float sw = 0.f, se = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        f32x4 w = v[it];
        if (normalize) { w[0] *= scale; w[1] *= scale; w[2] *= scale; w[3] *= scale; }
        uint2 pk;                                                   // v_cvt_pk_bf16_f32: round-to-nearest-even
        pk.x = f32x2_to_bf16x2_bits(w[0], w[1]);
        pk.y = f32x2_to_bf16x2_bits(w[2], w[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int word = (e < 2) ? pk.x : pk.y;
            const float back = __uint_as_float((e & 1) ? (word & 0xffff0000u) : (word << 16));
            const float err = w[e] - back;                          // exact: the discarded low bits of w
            sw = fmaf(w[e], w[e], sw);
            se = fmaf(err, err, se);
        }
        if (ok && c < kt) {
            if (out_f32) *(f32x4*)(out_f32 + row * (int64_t)kt + c) = w;
            *(uint2*)(out_bf16 + row * (int64_t)kt + c) = pk;
        }
    }
    int writer = 0;                                        // the lane of the group that ends up with the sums
    if (G == 32) { sw = dpp_sum32_upper(sw); se = dpp_sum32_upper(se); writer = 31; }
    else for (int off = G / 2; off > 0; off >>= 1) { sw += __shfl_xor(sw, off, 64); se += __shfl_xor(se, off, 64); }
    if (sub == writer && ok) row_stats[row] = make_float2(sqrtf(sw), sqrtf(se));
    if (gstats) {
        float nw = (sub == writer && ok) ? sqrtf(sw) : 0.f, ne = (sub == writer && ok) ? sqrtf(se) : 0.f;
        float ab = (sub == writer && ok && bias) ? fabsf(bias[row]) : 0.f;
        // (one atomic per wave, not per row)  NaN must poison the bound, and fmaxf would drop it: turn it into +inf
        if (nw != nw) nw = INFINITY;
        if (ne != ne) ne = INFINITY;
        if (ab != ab) ab = INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            nw = fmaxf(nw, __shfl_xor(nw, off, 64));
            ne = fmaxf(ne, __shfl_xor(ne, off, 64));
            ab = fmaxf(ab, __shfl_xor(ab, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomic_max_nonneg(gstats + 0, nw);
            atomic_max_nonneg(gstats + 1, ne);
            atomic_max_nonneg(gstats + 2, ab);
        }
    }
}

// eps_u >= |bf16-path score - fp32 score| for every item (the bound of the header): st = {||x||, ||x - bf16(x)||} of the user,
// gstats = the item side's maxima {||y||, ||y - bf16(y)||, |bias|}
__device__ __forceinline__ float filter_eps(float2 st, float bu_abs, const float* __restrict__ gstats, int kdim)
{
    const float ni = gstats[0], ai = gstats[1], bi = gstats[2];
    const float ck = (float)(kdim + 2) * 2.98023224e-07f;                       // (K + 2) (2^-24 + 2^-22)
    const float eps = st.y * (ni * 1.00390625f) + st.x * ai + ck * (st.x * ni * 1.0078125f + bu_abs + bi);
    return eps * 1.001953125f + 1e-30f;
}

// floor_u = tau_u - mult eps_u (rounded DOWN twice; mult = 2 for the filter's floor); flag_u = 1 when the bound is unusable
// (then floor_u = -inf).  flag / n_flagged may be NULL (a provisional floor: trec_topk_filter_floor_ex).
__global__ __launch_bounds__(256) void filter_floor_kernel(const float* __restrict__ tau, const float2* __restrict__ ustats,
                                                          const float* __restrict__ user_bias,
                                                          const float* __restrict__ gstats, int kdim, int64_t n_users,
                                                          float mult, float* __restrict__ floor_, int32_t* __restrict__ flag,
                                                          int32_t* __restrict__ n_flagged)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const float eps = filter_eps(ustats[u], user_bias ? fabsf(user_bias[u]) : 0.f, gstats, kdim);
    const float t = tau[u];
    float f = t - mult * eps;
    bool bad = !(eps < INFINITY);                                               // inf or NaN
    if (t == -INFINITY) f = -INFINITY;                                          // fewer than k superblocks: keep all
    else if (!(f == f)) bad = true;
    else f = float_pred(float_pred(f));
    if (bad) f = -INFINITY;
    floor_[u] = f;
    if (flag) {
        flag[u] = bad ? 1 : 0;
        if (bad) atomicAdd(n_flagged, 1);
    }
}

// The thresholds of the cascade's candidate lists in ONE pass over the users (trec_topk_cascade_floor): tau = the k-th largest int8
// lower bound (trec_topk_select_blocks over the chunk lists).  Layout rows without a source (src[u] < 0: band padding, the tail of
// the allocation) keep nothing: tau = floor0 = +inf, never flagged.  Everybody else: floor0 = tau - eps rounded down twice (the
// PROVISIONAL floor of trec_score_gemm_refine_candidates), +inf and flagged when the bound is unusable; cand_n = 0.
__global__ __launch_bounds__(256) void cascade_floor_kernel(float* __restrict__ tau, const int32_t* __restrict__ src,
                                                           const float2* __restrict__ ustats, const float* __restrict__ user_bias,
                                                           const float* __restrict__ gstats, int kdim, int64_t n_users,
                                                           float* __restrict__ floor0, int32_t* __restrict__ flag,
                                                           int32_t* __restrict__ n_flagged, int32_t* __restrict__ cand_n)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    if (cand_n) cand_n[u] = 0;
    if (src && src[u] < 0) {
        tau[u] = INFINITY;
        floor0[u] = INFINITY;
        flag[u] = 0;
        return;
    }
    const float eps = filter_eps(ustats[u], user_bias ? fabsf(user_bias[u]) : 0.f, gstats, kdim);
    const float t = tau[u];
    float f = t - eps;
    bool bad = !(eps < INFINITY);                                               // inf or NaN
    if (t == -INFINITY) f = -INFINITY;                                          // fewer than k superblocks: list everything
    else if (!(f == f)) bad = true;
    else f = float_pred(float_pred(f));
    floor0[u] = bad ? INFINITY : f;                                             // a user without a usable bound lists nothing
    flag[u] = bad ? 1 : 0;
    if (bad) atomicAdd(n_flagged, 1);
}

// ---- the cascade's PRE-REFINEMENT (round 5): a sharper threshold before the compaction -----------------------------------------
// tau8 = the k-th largest int8 LOWER bound M8 - e sits a whole e (~0.017 at 1M x 1M normalised rows) below the k-th best score, and
// the compaction keeps every superblock whose UPPER bound M8 + e reaches it: a window of 2 e.  The k superblocks that hold a user's k
// largest lower bounds are refined first (the grouped bf16 kernel, 0.5 % of all pairs): each then holds an item with fp32 score >=
// M16 - eps (eps: the bf16 filter's bound, ~0.003), so tauA = min M16 - eps is ALSO a lower bound of the k-th best score, and it is
// what the compaction, the candidate floor and the lists then work with: tau = max(tau8, tauA).  Kept pairs 2.6 % -> ~1.3 %,
// candidates per user 27 -> ~14 (simulated on 256 x 1M; measured: DESIGN 5h).
//
// prerefine_rows_kernel: sel [n_users][k] = rows of the chunk lists that hold the user's k largest (tagged) lower bounds, val
// [k][n_users] their values; row / top_k = the chunk, lb_tag_index(value) = the superblock inside it.  Writes sel_sb [n_users][k]
// (superblock ids, -1 = no certificate in that slot) and appends the user to its superblocks' lists row_user [n_sb][rcap] (counts
// row_count [n_sb], zeroed by the caller; LDS counters per workgroup of 1024 users, ONE global atomic per (workgroup, superblock)).
// ok[u] = 1 when all k slots were placed (the user's tauA is then valid).
// KS = slots per user held in registers (16: k <= 16, four users per thread; 64: the wide route's k <= 64, one user per thread).
template <int KS>
__global__ __launch_bounds__(256) void prerefine_rows_kernel(const int32_t* __restrict__ sel, const float* __restrict__ val, int k,
                                                            int top_k, int sb_per_chunk, int32_t n_sb, int64_t n_users,
                                                            const int32_t* __restrict__ src, int32_t rcap,
                                                            int32_t* __restrict__ sel_sb, int32_t* __restrict__ row_count,
                                                            int32_t* __restrict__ row_user, int32_t* __restrict__ ok,
                                                            int32_t* __restrict__ sel_pos)
{
    extern __shared__ int cnt[];                                               // [n_sb]: count, then the global base of this workgroup's run
    for (int s = threadIdx.x; s < n_sb; s += 256) cnt[s] = 0;
    __syncthreads();
    constexpr int UPT = 64 / KS;                                               // users per thread
    int sb[UPT][KS];
    short lr[UPT][KS];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int64_t u = (int64_t)blockIdx.x * (256 * UPT) + i * 256 + threadIdx.x;
        const bool live = u < n_users && (!src || src[u] >= 0);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            sb[i][j] = -1; lr[i][j] = 0;
            if (live && j < k) {
                const int32_t row = sel[u * k + j];
                const float v = val[(int64_t)j * n_users + u];
                if (row >= 0 && v > -INFINITY) {
                    const int s = (row / top_k) * sb_per_chunk + lb_tag_index(v);
                    if (s >= 0 && s < n_sb) { sb[i][j] = s; lr[i][j] = (short)atomicAdd(&cnt[s], 1); }
                }
            }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_sb; s += 256) {
        const int c = cnt[s];
        cnt[s] = c ? atomicAdd(&row_count[s], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int64_t u = (int64_t)blockIdx.x * (256 * UPT) + i * 256 + threadIdx.x;
        if (u >= n_users) continue;
        bool all = true;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (j >= k) break;
            const int s = sb[i][j];
            if (s < 0) { all = false; sel_sb[u * k + j] = -1; continue; }
            const int pos = cnt[s] + lr[i][j];
            sel_sb[u * k + j] = pos < rcap ? s : -1;                           // (a full list: the pair is left to the compaction)
            if (sel_pos) sel_pos[u * k + j] = pos;                             // (where in its superblock's list: the maximum is read back from there)
            if (pos < rcap) row_user[(int64_t)s * rcap + pos] = (int32_t)u;
            else all = false;
        }
        ok[u] = all ? 1 : 0;
    }
}

// After the pre-refining launch: tau[u] = max(tau[u], min_j table[sel_sb[u][j]][u] - eps_u) for the users with ok[u] (all k slots
// placed).  Then the listed entries are taken out of the compaction's way:
//   listed == 0 (the pre-refining launch only wrote maxima): every listed entry becomes +inf -- the compaction keeps those pairs
//     whatever its own bound says and the listing launch refines them again; a refined entry is never compared through the int8
//     bound e, which need not cover eps;
//   listed != 0 (it also LISTED their candidates, trec_score_gemm_refine_candidates with the provisional floor tau8 - eps): the
//     bf16 maxima are saved to vals [n_users][k], the entries become -inf -- the compaction drops the pairs, nothing is refined
//     twice.  The caller puts vals back into the columns of the users it has to re-do from the table.
// Either way cand_floor[u] (nullable) rises to tau - eps for the launches still to come (a user that lists nothing keeps +inf).
__global__ __launch_bounds__(256) void prerefine_tau_kernel(const int32_t* __restrict__ sel_sb, const int32_t* __restrict__ ok, int k,
                                                           float* __restrict__ table, int64_t stride, int64_t n_users,
                                                           const int32_t* __restrict__ src, const float2* __restrict__ ustats,
                                                           const float* __restrict__ user_bias, const float* __restrict__ gstats,
                                                           int kdim, float* __restrict__ tau, int listed, float* __restrict__ vals,
                                                           float* __restrict__ cand_floor)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const bool live = !(src && src[u] < 0);
    float m = INFINITY;
    int32_t s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = (live && j < k) ? sel_sb[u * k + j] : -1;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = s[j] >= 0 ? table[(int64_t)s[j] * stride + u] : INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fminf(m, (v[j] == v[j]) ? v[j] : -INFINITY);          // a NaN certifies nothing
    if (listed) {
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < k) vals[u * k + j] = v[j];
    }
    if (!live) return;
    if (ok[u]) {
        const float eps = filter_eps(ustats[u], user_bias ? fabsf(user_bias[u]) : 0.f, gstats, kdim);
        float t = m - eps;
        if (eps < INFINITY && t == t && m < INFINITY) {
            t = float_pred(float_pred(t));
            if (t > tau[u]) {
                tau[u] = t;
                if (cand_floor && cand_floor[u] < INFINITY) {
                    const float f = float_pred(float_pred(t - eps));           // the provisional floor of the launches to come
                    if (f > cand_floor[u]) cand_floor[u] = f;
                }
            }
        }
    }
    const float mark = listed ? -INFINITY : INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (s[j] >= 0) table[(int64_t)s[j] * stride + u] = mark;
}

// The same threshold step behind the MARKING pre-refining launch (trec_score_gemm_refine_candidates_marked): the maxima are read back from
// the lists -- pre_max[sel_sb * rcap + sel_pos], a few tens of MB that the launch has just written -- instead of from 10 random
// entries per user of the 7.8 GB table, and the -inf marks are already in place: no table access at all (0.54 -> ~0.2 ms at 1M users).
__global__ __launch_bounds__(256) void prerefine_tau_listed_kernel(const int32_t* __restrict__ sel_sb, const int32_t* __restrict__ sel_pos,
                                                                  const int32_t* __restrict__ ok, int k, const float* __restrict__ pre_max,
                                                                  int64_t rcap, int64_t n_users, const int32_t* __restrict__ src,
                                                                  const float2* __restrict__ ustats, const float* __restrict__ user_bias,
                                                                  const float* __restrict__ gstats, int kdim, float* __restrict__ tau,
                                                                  float* __restrict__ vals, float* __restrict__ cand_floor)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const bool live = !(src && src[u] < 0);
    float m = INFINITY;
    for (int j0 = 0; j0 < k; j0 += 16) {                                       // (k <= 16: one round; the wide route: up to four)
        int32_t s[16], ps[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s[j] = (live && j0 + j < k) ? sel_sb[u * k + j0 + j] : -1;
            ps[j] = (live && j0 + j < k) ? sel_pos[u * k + j0 + j] : 0;
        }
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = s[j] >= 0 ? pre_max[(int64_t)s[j] * rcap + ps[j]] : INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) m = fminf(m, (v[j] == v[j]) ? v[j] : -INFINITY);      // a NaN certifies nothing
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j0 + j < k) vals[u * k + j0 + j] = v[j];
    }
    if (!live || !ok[u]) return;
    const float eps = filter_eps(ustats[u], user_bias ? fabsf(user_bias[u]) : 0.f, gstats, kdim);
    float t = m - eps;
    if (eps < INFINITY && t == t && m < INFINITY) {
        t = float_pred(float_pred(t));
        if (t > tau[u]) {
            tau[u] = t;
            if (cand_floor && cand_floor[u] < INFINITY) {
                const float f = float_pred(float_pred(t - eps));               // the provisional floor of the launches to come
                if (f > cand_floor[u]) cand_floor[u] = f;
            }
        }
    }
}

#ifndef FILTER_RB
#define FILTER_RB 8          // survivors re-scored per round (their fp32 rows staged in LDS)
#endif
// The second half of the finish kernels: ``total`` (<= FILTER_CMAX) survivors' item ids sit in cand[] (LDS of this wave), the
// user's row in registers uw.  Exact fp32 scores by the reference's k-ordered fmaf chain, then (s + b_u) + b_i, then the k
// best by (value desc, index asc) to ov / oi.
__device__ __forceinline__ void finish_rescore_topk(int total, const int32_t* cand, float* urow, float* rows, const f32x4 (&uw)[4],
                                                    int lane, int kdim, int chunks, int rstride, bool vec,
                                                    const float* __restrict__ V, int64_t ld_v, int32_t item_index_base,
                                                    const float* __restrict__ item_bias, bool has_user_bias, float bu, int k,
                                                    float* __restrict__ ov, int32_t* __restrict__ oi, int64_t u)
{
    __builtin_amdgcn_wave_barrier();              // cand[] / rows[] are private to this wave; a wave's DS operations execute in order
    // ---- exact fp32 scores of the survivors: the reference's k-ordered fmaf chain, then (s + b_u) + b_i
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    unsigned long long key = EMPTY;                // lane r ends up holding survivor (round * FILTER_RB + r)'s key ...
    unsigned long long mine = EMPTY;               // ... moved to lane (round * FILTER_RB + r) here
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = q * 64 + lane;
        if (ch < chunks) *(f32x4*)(urow + ch * 4) = uw[q];        // broadcast from LDS by every chain step
    }
    const float* a = urow;
    for (int r0 = 0; r0 < total; r0 += FILTER_RB) {
        const int nr = (total - r0 < FILTER_RB) ? total - r0 : FILTER_RB;
        for (int idx = lane; idx < nr * chunks; idx += 64) {
            const int r = idx / chunks, ch = idx - r * chunks;
            const float* src = V + (int64_t)(cand[r0 + r] - item_index_base) * ld_v + ch * 4;
            f32x4 w;
            if (vec) w = *(const f32x4*)src;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
            }
            *(f32x4*)(rows + r * rstride + ch * 4) = w;
        }
        const int32_t item = (lane < nr) ? cand[r0 + lane] : item_index_base;
        const float ibv = (item_bias && lane < nr) ? item_bias[item - item_index_base] : 0.f;   // rides with the row loads
        __builtin_amdgcn_wave_barrier();
        if (lane < nr) {
            const float* b = rows + lane * rstride;
            float acc = 0.0f;
            int kk = 0;
            {
                for (; kk + 4 <= kdim; kk += 4) {
                    const f32x4 a4 = *(const f32x4*)(a + kk);
                    const f32x4 b4 = *(const f32x4*)(b + kk);
                    acc = __fmaf_rn(a4[0], b4[0], acc); acc = __fmaf_rn(a4[1], b4[1], acc);
                    acc = __fmaf_rn(a4[2], b4[2], acc); acc = __fmaf_rn(a4[3], b4[3], acc);
                }
            }
            for (; kk < kdim; ++kk) acc = __fmaf_rn(a[kk], b[kk], acc);
            if (has_user_bias) acc = acc + bu;
            if (item_bias) acc = acc + ibv;
            key = merge_key(acc, item);
        } else key = EMPTY;
        __builtin_amdgcn_wave_barrier();
        // lane r0 + r takes over lane r's key (r0 is a multiple of 16: a fixed rotation per round)
        {
            const int srcl = (lane - r0) & 63;
            const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)key, srcl, 64);
            const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(key >> 32), srcl, 64);
            if (lane >= r0 && lane < r0 + nr) mine = ((unsigned long long)hi << 32) | lo;
        }
    }
    // ---- the k best by (value desc, index asc)
    for (int t = 0; t < k; ++t) {
        const unsigned long long best = wave_max_u64(mine);
        if (mine == best && best != EMPTY) mine = EMPTY;
        if (lane == 0) {
            const unsigned int hi = (unsigned int)(best >> 32);
            const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
            ov[u * k + t] = (best == EMPTY) ? -INFINITY : __uint_as_float(bits);
            oi[u * k + t] = (best == EMPTY) ? -1 : (int32_t)(~(unsigned int)best);
        }
    }
}

// One wave per user.  Only the first count[u] slots of a user have stage-3 lists.  Survivors are re-scored FILTER_RB at a time: their fp32
// rows are fetched with coalesced 16-byte loads (a row = KT/4 consecutive lanes) into LDS, then lane r walks row r with
// the reference's k-ordered fmaf chain (a row per lane straight from global memory is a 16-byte access per 512-byte
// row per load: 4.6 ms at 1M users, ~15 survivors each; staged: see DESIGN.md).
#define FILTER_SPEC 24       // slots whose id lists are fetched before the user's slot count is known
template <int CPL>
__global__ __launch_bounds__(256) void filter_finish_kernel(
    const int32_t* __restrict__ pi, int cap, int ksel, const int32_t* __restrict__ count, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged)
{
    extern __shared__ __attribute__((aligned(16))) char fsmem[];
    const int wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= n_users) return;
    const int lane = lane_id();
    const int kd4 = (kdim + 3) & ~3;                         // floats staged per row (operand rows are padded to kpad >= kd4)
    const int rstride = kd4 + 4;                             // +4 floats: lanes r = 0..15 start on distinct 4-bank groups
    int32_t* cand = (int32_t*)fsmem + wave * FILTER_CMAX;
    float* urow = (float*)(fsmem + 4 * FILTER_CMAX * 4) + (size_t)wave * kd4;
    float* rows = (float*)(fsmem + 4 * FILTER_CMAX * 4) + (size_t)4 * kd4 + (size_t)wave * FILTER_RB * rstride;
    // ---- everything that does not depend on anything else leaves at once: the user's count, its row and bias, and --
    // speculatively, before the count is known -- the id lists of its first FILTER_SPEC slots (their entries are masked
    // by the count afterwards; a user with more kept slots pays one more round trip).  The wave is latency-bound: every
    // dependent load round costs ~1-2 us and a CU holds 16 of these waves.
    const int c_u = count[u];
    const int64_t base = u * (int64_t)ksel * 2 * cap;
    const int n_ent = ksel * 2 * cap;
    const int n_spec = (n_ent < FILTER_SPEC * 2 * cap) ? n_ent : FILTER_SPEC * 2 * cap;
    const int chunks = kd4 >> 2;
    const bool vec = ((ld_v & 3) == 0) && ((ld_u & 3) == 0);
    int32_t id[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = c * 64 + lane;
        id[c] = (j < n_spec) ? pi[base + j] : -1;
    }
    f32x4 uw[4];                                            // the user's row: kd4 <= 1024 floats = 4 float4 per lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = q * 64 + lane;
        uw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ch < chunks) {
            const float* src = U + u * ld_u + ch * 4;
            if (vec) uw[q] = *(const f32x4*)src;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) uw[q][e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
            }
        }
    }
    const float bu = user_bias ? user_bias[u] : 0.f;
    // ---- kept slots: the first count[u] of the user's ksel slots (trec_topk_collect_blocks)
    const unsigned long long keptmask = c_u >= 64 ? ~0ull : ((1ull << c_u) - 1ull);
    // ---- survivors: every valid entry of a kept slot's two lists; a full list may have dropped items above the floor
    bool lossy = false;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = c * 64 + lane;
        const bool keep = j < n_ent && ((keptmask >> (j / (2 * cap))) & 1ull);
        if (keep && j >= n_spec) id[c] = pi[base + j];            // beyond the speculated slots (rare)
        if (!keep) id[c] = -1;
        if (id[c] >= 0 && (j % cap) == cap - 1) lossy = true;
    }
    int total = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(id[c] >= 0);
        const int pos = total + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (id[c] >= 0 && pos < FILTER_CMAX) cand[pos] = id[c];
        total += __builtin_popcountll(m);
    }
    const bool over = __builtin_amdgcn_ballot_w64(lossy) != 0ull || total > FILTER_CMAX;
    if (over && lane == 0 && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
    if (total > FILTER_CMAX) total = FILTER_CMAX;
    finish_rescore_topk(total, cand, urow, rows, uw, lane, kdim, chunks, rstride, vec, V, ld_v, item_index_base, item_bias,
                        user_bias != nullptr, bu, k, ov, oi, u);
}

// The same finish without capacity limits, for the WIDE second pass over the users the first pass flagged (hundreds of kept
// superblocks, hundreds of survivors: rows near a few dominant items early in training, near-duplicate catalogues).  One wave
// per user streams through the id lists of its count[u] kept slots 64 entries at a time; survivors queue up in LDS and are
// re-scored WIDE_NEW at a time (the exact k-ordered fmaf chain, 8 rows staged per round like filter_finish_kernel); after every
// batch the running top-k (lanes 0..k-1) and the batch's keys (lanes 16..63) go through k rounds of a DPP wave maximum.
// Only a FULL stage-3 list (it may have dropped an item above the floor) still flags the user.
#define WIDE_NEW 48
__global__ __launch_bounds__(256) void filter_finish_wide_kernel(
    const int32_t* __restrict__ pi, int cap, int ksel, const int32_t* __restrict__ count, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged)
{
    extern __shared__ __attribute__((aligned(16))) char fsmem[];
    const int wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= n_users) return;
    const int lane = lane_id();
    const int kd4 = (kdim + 3) & ~3;
    const int rstride = kd4 + 4;
    int32_t* cand = (int32_t*)fsmem + wave * 128;                       // queue: at most 47 left over + 64 new ids
    float* urow = (float*)(fsmem + 4 * 128 * 4) + (size_t)wave * kd4;
    float* rows = (float*)(fsmem + 4 * 128 * 4) + (size_t)4 * kd4 + (size_t)wave * FILTER_RB * rstride;
    const int chunks = kd4 >> 2;
    const bool vec = ((ld_v & 3) == 0) && ((ld_u & 3) == 0);
    int c_u = count[u];
    if (c_u > ksel) c_u = ksel;
    const int64_t base = u * (int64_t)ksel * 2 * cap;
    const int n_ent = c_u * 2 * cap;                                    // the kept slots are the first count[u] ones
    for (int ch = lane; ch < chunks; ch += 64) {
        const float* src = U + u * ld_u + ch * 4;
        f32x4 w;
        if (vec) w = *(const f32x4*)src;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
        }
        *(f32x4*)(urow + ch * 4) = w;
    }
    const float bu = user_bias ? user_bias[u] : 0.f;
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    unsigned long long best = EMPTY;                                    // lane t < k: the t-th best key so far
    bool lossy = false;
    int queued = 0;
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < n_ent || queued > 0; e0 += 64) {
        if (e0 < n_ent) {
            const int j = e0 + lane;
            const int32_t id = (j < n_ent) ? pi[base + j] : -1;
            if (id >= 0 && (j % cap) == cap - 1) lossy = true;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(id >= 0);
            if (id >= 0) cand[queued + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = id;
            queued += __builtin_popcountll(m);
            __builtin_amdgcn_wave_barrier();
        }
        const bool last = e0 + 64 >= n_ent;
        while (queued >= WIDE_NEW || (last && queued > 0)) {
            const int nb = queued < WIDE_NEW ? queued : WIDE_NEW;
            unsigned long long mine = (lane < 16) ? best : EMPTY;      // lanes 16 .. 16 + nb - 1 take the batch's keys
            for (int r0 = 0; r0 < nb; r0 += FILTER_RB) {
                const int nr = (nb - r0 < FILTER_RB) ? nb - r0 : FILTER_RB;
                for (int idx = lane; idx < nr * chunks; idx += 64) {
                    const int r = idx / chunks, ch = idx - r * chunks;
                    const float* src = V + (int64_t)(cand[r0 + r] - item_index_base) * ld_v + ch * 4;
                    f32x4 w;
                    if (vec) w = *(const f32x4*)src;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
                    }
                    *(f32x4*)(rows + r * rstride + ch * 4) = w;
                }
                const int32_t item = (lane < nr) ? cand[r0 + lane] : item_index_base;
                const float ibv = (item_bias && lane < nr) ? item_bias[item - item_index_base] : 0.f;
                __builtin_amdgcn_wave_barrier();
                unsigned long long key = EMPTY;
                if (lane < nr) {
                    const float* b = rows + lane * rstride;
                    float acc = 0.0f;
                    int kk = 0;
                    for (; kk + 4 <= kdim; kk += 4) {
                        const f32x4 a4 = *(const f32x4*)(urow + kk);
                        const f32x4 b4 = *(const f32x4*)(b + kk);
                        acc = __fmaf_rn(a4[0], b4[0], acc); acc = __fmaf_rn(a4[1], b4[1], acc);
                        acc = __fmaf_rn(a4[2], b4[2], acc); acc = __fmaf_rn(a4[3], b4[3], acc);
                    }
                    for (; kk < kdim; ++kk) acc = __fmaf_rn(urow[kk], b[kk], acc);
                    if (user_bias) acc = acc + bu;
                    if (item_bias) acc = acc + ibv;
                    key = merge_key(acc, item);
                }
                __builtin_amdgcn_wave_barrier();
                {   // lane 16 + r0 + r takes over lane r's key
                    const int srcl = (lane - 16 - r0) & 63;
                    const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)key, srcl, 64);
                    const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(key >> 32), srcl, 64);
                    if (lane >= 16 + r0 && lane < 16 + r0 + nr) mine = ((unsigned long long)hi << 32) | lo;
                }
            }
            // the k best of (running top-k, this batch): keys are unique (an item sits in exactly one list)
            unsigned long long nxt = EMPTY;
            for (int t = 0; t < k; ++t) {
                const unsigned long long b = wave_max_u64(mine);
                if (mine == b && b != EMPTY) mine = EMPTY;
                if (lane == t) nxt = b;
            }
            best = nxt;
            // drop the batch from the queue
            const int rest = queued - nb;
            const int32_t tmp = (lane < rest) ? cand[nb + lane] : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) cand[lane] = tmp;
            __builtin_amdgcn_wave_barrier();
            queued = rest;
        }
    }
    if (__builtin_amdgcn_ballot_w64(lossy) != 0ull && lane == 0 && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
    if (lane < k) {
        const unsigned int hi = (unsigned int)(best >> 32);
        const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
        ov[u * k + lane] = (best == EMPTY) ? -INFINITY : __uint_as_float(bits);
        oi[u * k + lane] = (best == EMPTY) ? -1 : (int32_t)(~(unsigned int)best);
    }
}


// ---- the finish behind the cascade's candidate lists (csrc/topk_candidates.hip) ---------------------------------------------
// One wave per user.  cand[u][0 .. n) = {item id, bf16-path score sh} of EVERY item of the user's refined superblocks with
// sh >= cand_floor[u] (blockmax_bf16x16_kernel<.., LIST>).  tau = the k-th largest sh among them, floor = tau - 2 eps (the
// filter's floor, from item scores instead of superblock maxima), survivors = candidates with sh >= floor, re-scored exactly.
// A user with more than ``cap`` candidates (its list is incomplete) or more than FILTER_CMAX survivors is flagged; a user whose
// provisional floor is +inf (padding rows; users the floor kernel flagged for an unusable bound) is skipped.
template <int CPL>
__device__ __forceinline__ void candidates_finish_user(
    const int64_t u, const int32_t* __restrict__ cand_n, const int2* __restrict__ cand_list, int cap, const float* __restrict__ cand_floor,
    const float2* __restrict__ ustats, const float* __restrict__ gstats, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
    const int32_t* __restrict__ out_index)
{
    extern __shared__ __attribute__((aligned(16))) char fsmem[];
    const int wave = threadIdx.x >> 6;
    if (u >= n_users) return;
    // users sorted by scale class: the result row is the CALLER's row of this layout row; rows without one write nothing
    const int64_t uo = out_index ? (int64_t)out_index[u] : u;
    if (uo < 0) return;                                     // (wave-uniform)
    const int lane = lane_id();
    const int kd4 = (kdim + 3) & ~3;
    const int rstride = kd4 + 4;
    int32_t* cand = (int32_t*)fsmem + wave * FILTER_CMAX;
    float* urow = (float*)(fsmem + 4 * FILTER_CMAX * 4) + (size_t)wave * kd4;
    float* rows = (float*)(fsmem + 4 * FILTER_CMAX * 4) + (size_t)4 * kd4 + (size_t)wave * FILTER_RB * rstride;
    // everything that depends on nothing leaves at once: the count, the floor, the list (masked by the count afterwards), the row
    const int n = cand_n[u];
    const float f0 = cand_floor[u];
    const int chunks = kd4 >> 2;
    const bool vec = ((ld_v & 3) == 0) && ((ld_u & 3) == 0);
    int2 ent[CPL];                                          // the first 64 entries before the count is known (a user has ~32), the rest after
    ent[0] = cand_list[u * (int64_t)cap + lane];
    f32x4 uw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = q * 64 + lane;
        uw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ch < chunks) {
            const float* src = U + u * ld_u + ch * 4;
            if (vec) uw[q] = *(const f32x4*)src;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) uw[q][e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
            }
        }
    }
    const float bu = user_bias ? user_bias[u] : 0.f;
    const float2 st = ustats[u];
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    const bool skip = !(f0 < INFINITY);
    const bool over = n > cap;
#pragma unroll
    for (int c = 1; c < CPL; ++c) {
        ent[c] = make_int2(0, 0);
        if (n > c * 64 && !over) ent[c] = cand_list[u * (int64_t)cap + c * 64 + lane];
    }
    if (skip || over) {                                     // (wave-uniform)
        if (over && !skip && lane == 0 && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
        if (lane < k) { ov[uo * k + lane] = -INFINITY; oi[uo * k + lane] = -1; }
        return;
    }
    // ---- tau = the k-th largest candidate score (keys: value desc, id asc -- an item is listed once, keys are unique)
    unsigned long long key[CPL], w[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        key[c] = (c * 64 + lane < n) ? merge_key(__int_as_float(ent[c].y), ent[c].x) : EMPTY;
        w[c] = key[c];
    }
    unsigned long long kth = EMPTY;
    for (int t = 0; t < k; ++t) {
        unsigned long long m = w[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) m = w[c] > m ? w[c] : m;
        kth = wave_max_u64(m);
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (w[c] == kth && kth != EMPTY) w[c] = EMPTY;
    }
    float tau = -INFINITY;                                  // fewer than k candidates: every one survives
    if (kth != EMPTY) {
        const unsigned int hi = (unsigned int)(kth >> 32);
        tau = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);
    }
    const float eps = filter_eps(st, fabsf(bu), gstats, kdim);
    float fl = tau - 2.0f * eps;
    if (tau == -INFINITY) fl = -INFINITY;
    else fl = float_pred(float_pred(fl));                   // (eps is finite here: cand_floor was)
    // ---- survivors
    int total = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const bool keep = (c * 64 + lane < n) && __int_as_float(ent[c].y) >= fl;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const int pos = total + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (keep && pos < FILTER_CMAX) cand[pos] = ent[c].x;
        total += __builtin_popcountll(m);
    }
    if (total > FILTER_CMAX) {
        if (lane == 0 && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
        total = FILTER_CMAX;
    }
    finish_rescore_topk(total, cand, urow, rows, uw, lane, kdim, chunks, rstride, vec, V, ld_v, item_index_base, item_bias,
                        user_bias != nullptr, bu, k, ov, oi, uo);
}

template <int CPL>
__global__ __launch_bounds__(256) void candidates_finish_kernel(
    const int32_t* __restrict__ cand_n, const int2* __restrict__ cand_list, int cap, const float* __restrict__ cand_floor,
    const float2* __restrict__ ustats, const float* __restrict__ gstats, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
    const int32_t* __restrict__ out_index)
{
    candidates_finish_user<CPL>((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), cand_n, cand_list, cap, cand_floor, ustats, gstats, U, V,
                                ld_u, ld_v, kdim, user_bias, item_bias, item_index_base, n_users, k, ov, oi, flag, n_flagged, out_index);
}

// the same wave-per-user finish over a LIST of users (the ones whose lists are too long for the 16-lane form): a fixed grid whose
// waves stride over the list; its length is read from device memory, so no host read sits between the two finish launches
template <int CPL>
__global__ __launch_bounds__(256) void candidates_finish_listed_kernel(
    const int32_t* __restrict__ user_list, const int32_t* __restrict__ user_count,
    const int32_t* __restrict__ cand_n, const int2* __restrict__ cand_list, int cap, const float* __restrict__ cand_floor,
    const float2* __restrict__ ustats, const float* __restrict__ gstats, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
    const int32_t* __restrict__ out_index)
{
    const int count = *user_count;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += gridDim.x * 4)
        candidates_finish_user<CPL>((int64_t)user_list[i], cand_n, cand_list, cap, cand_floor, ustats, gstats, U, V, ld_u, ld_v, kdim,
                                    user_bias, item_bias, item_index_base, n_users, k, ov, oi, flag, n_flagged, out_index);
}

// ---- the same finish for SHORT lists: four users per wave ---------------------------------------------------------------------
// One wave per user is a latency chain (count -> list -> floor -> item rows -> 128-step fmaf chain -> k rounds of maxima): ~17 us
// however few candidates the user has.  On an item shard of an N-GPU run a user lists ~27 / N candidates (3-4 at N = 8) and the
// chain is all there is: 2 ms per rank for 1M users, replicated on every rank.  Here 16 lanes own a user and lane j owns
// candidate j: the k-th largest listed score by DPP row maxima (a DPP row IS the 16 lanes), the floor, and every surviving
// lane walks ITS candidate's fp32 item row straight from memory with the reference's k-ordered fmaf chain against the user
// row broadcast from LDS -- no compaction, no staging; an item shard of a few hundred MB sits in the Infinity Cache.  A user
// with more than 16 candidates is flagged (the caller re-does it on its table column): at 3-4 expected candidates that is rare.
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long x)
{
    x = dpp_max_u64<0x128, 0xf>(x);      // row_ror:8
    x = dpp_max_u64<0x124, 0xf>(x);      // row_ror:4
    x = dpp_max_u64<0x122, 0xf>(x);      // row_ror:2
    x = dpp_max_u64<0x121, 0xf>(x);      // row_ror:1   -> every lane: the maximum of its 16-lane row
    return x;
}

// CPL candidates per lane (lane j owns candidates j, 16 + j, ...: lists of up to 16 * CPL entries; the single-GPU cascade lists ~15
// per user, a wave then finishes FOUR users where the wave-per-user form finishes one).  over_list / over_count (nullable): users with
// a longer list are appended there for candidates_finish_listed_kernel instead of being flagged.
template <int CPL>
__global__ __launch_bounds__(256) void candidates_finish16_kernel(
    const int32_t* __restrict__ cand_n, const int2* __restrict__ cand_list, int cap, const float* __restrict__ cand_floor,
    const float2* __restrict__ ustats, const float* __restrict__ gstats, const float* __restrict__ U,
    const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim, const float* __restrict__ user_bias,
    const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k, float* __restrict__ ov,
    int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
    const int32_t* __restrict__ out_index, int32_t* __restrict__ over_list, int32_t* __restrict__ over_count)
{
    extern __shared__ __attribute__((aligned(16))) char fsmem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, gl = lane & 15;
    const int kd4 = (kdim + 3) & ~3;
    float* urow = (float*)fsmem + (size_t)(wave * 4 + grp) * kd4;           // this user's fp32 row, read by its 16 lanes
    const int64_t u = ((int64_t)blockIdx.x * 4 + wave) * 4 + grp;
    const bool live = u < n_users;
    const int64_t uc = live ? u : n_users - 1;
    const int64_t uo = out_index ? (int64_t)out_index[uc] : uc;
    const int n = cand_n[uc];
    const float f0 = cand_floor[uc];
    int2 ent[CPL];                                           // (unconditional loads inside the list's capacity, masked by the count)
#pragma unroll
    for (int c = 0; c < CPL; ++c) ent[c] = cand_list[uc * (int64_t)cap + (gl + 16 * c < cap ? gl + 16 * c : cap - 1)];
    const bool vec = ((ld_v & 3) == 0) && ((ld_u & 3) == 0);
    for (int ch = gl; ch < (kd4 >> 2); ch += 16) {
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        const float* src = U + uc * ld_u + ch * 4;
        if (vec) w = *(const f32x4*)src;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
        }
        *(f32x4*)(urow + ch * 4) = w;
    }
    const float bu = user_bias ? user_bias[uc] : 0.f;
    const float2 st = ustats[uc];
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    const bool skip = !live || uo < 0 || !(f0 < INFINITY);                 // (uniform over the 16 lanes of the user)
    const bool over = n > 16 * CPL || n > cap;
    // ---- tau = the k-th largest listed score of this user (row maxima: nothing crosses the 16-lane rows)
    unsigned long long wkey[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
        wkey[c] = (!skip && !over && gl + 16 * c < n) ? merge_key(__int_as_float(ent[c].y), ent[c].x) : EMPTY;
    unsigned long long kth = EMPTY;
    for (int t = 0; t < k; ++t) {
        unsigned long long m = wkey[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) m = wkey[c] > m ? wkey[c] : m;
        kth = row16_max_u64(m);
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (wkey[c] == kth && kth != EMPTY) wkey[c] = EMPTY;
    }
    float tau = -INFINITY;
    if (kth != EMPTY) {
        const unsigned int hi = (unsigned int)(kth >> 32);
        tau = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);
    }
    const float eps = filter_eps(st, fabsf(bu), gstats, kdim);
    float fl = tau - 2.0f * eps;
    if (tau == -INFINITY) fl = -INFINITY;
    else fl = float_pred(float_pred(fl));
    __builtin_amdgcn_wave_barrier();                                       // urow is private to this wave: DS operations of a wave execute in order
    // ---- exact fp32 score of this lane's candidates: the reference's k-ordered fmaf chain, then (s + b_u) + b_i
    unsigned long long mine[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const bool keep = !skip && !over && gl + 16 * c < n && __int_as_float(ent[c].y) >= fl;
        mine[c] = EMPTY;
        if (c > 0 && __builtin_amdgcn_ballot_w64(keep) == 0ull) continue;   // (wave-uniform: nobody holds a survivor in this slot)
        const int64_t it = keep ? (int64_t)ent[c].x - item_index_base : 0;
        const float* b = V + it * ld_v;
        float acc = 0.0f;
        int kk = 0;
        if (vec) {
            for (; kk + 4 <= kdim; kk += 4) {
                const f32x4 a4 = *(const f32x4*)(urow + kk);
                const f32x4 b4 = *(const f32x4*)(b + kk);
                acc = __fmaf_rn(a4[0], b4[0], acc); acc = __fmaf_rn(a4[1], b4[1], acc);
                acc = __fmaf_rn(a4[2], b4[2], acc); acc = __fmaf_rn(a4[3], b4[3], acc);
            }
        }
        for (; kk < kdim; ++kk) acc = __fmaf_rn(urow[kk], b[kk], acc);
        if (user_bias) acc = acc + bu;
        if (item_bias) acc = acc + item_bias[it];
        if (keep) mine[c] = merge_key(acc, ent[c].x);
    }
    // ---- the k best by (value desc, index asc): lane t of the user's row writes place t
    unsigned long long place = EMPTY;
    for (int t = 0; t < k; ++t) {
        unsigned long long m = mine[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) m = mine[c] > m ? mine[c] : m;
        const unsigned long long best = row16_max_u64(m);
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (mine[c] == best && best != EMPTY) mine[c] = EMPTY;
        if (gl == t) place = best;
    }
    if (live && uo >= 0) {
        const bool handed_on = over && !skip && over_list != nullptr;      // the listed kernel writes this user's rows
        if (gl < k && !handed_on) {
            const unsigned int hi = (unsigned int)(place >> 32);
            const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
            ov[uo * k + gl] = (place == EMPTY) ? -INFINITY : __uint_as_float(bits);
            oi[uo * k + gl] = (place == EMPTY) ? -1 : (int32_t)(~(unsigned int)place);
        }
        if (over && !skip && gl == 0) {
            if (over_list) over_list[atomicAdd(over_count, 1)] = (int32_t)u;
            else if (flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
        }
    }
}

extern "C" int trec_score_prep_filter(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize,
                                      const float* bias, float* out_f32, void* out_bf16, float* row_stats, float* gstats,
                                      void* stream)
{
    TREC_REQUIRE(repr && out_bf16 && row_stats, "trec_score_prep_filter: null pointer");
    TREC_REQUIRE(d >= 1 && kpad >= d && kpad % 4 == 0 && kpad <= 256, "trec_score_prep_filter: need d <= kpad <= 256, kpad % 4 == 0");
    TREC_REQUIRE(((uintptr_t)repr % 16) == 0 || (d & 3) != 0, "trec_score_prep_filter: repr must be 16-byte aligned");
    if (n == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int g = kpad >= 128 ? 32 : kpad / 4;              // 8, 16 or 32 lanes per row
    const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
#define TREC_PF(GV) hipLaunchKernelGGL(prep_filter_kernel<GV>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, normalize, bias, out_f32, (unsigned short*)out_bf16, (float2*)row_stats, gstats)
    if (g == 32) TREC_PF(32);
    else if (g == 16) TREC_PF(16);
    else TREC_PF(8);
#undef TREC_PF
    return trec_check_launch("trec_score_prep_filter");
}

extern "C" int trec_topk_filter_floor_ex(const float* tau, const float* user_stats, const float* user_bias,
                                         const float* item_gstats, int32_t kdim, int64_t n_users, float mult, float* floor_,
                                         int32_t* flag, int32_t* n_flagged, void* stream);

extern "C" int trec_topk_filter_floor(const float* tau, const float* user_stats, const float* user_bias,
                                      const float* item_gstats, int32_t kdim, int64_t n_users, float* floor_, int32_t* flag,
                                      int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(tau && user_stats && item_gstats && floor_ && flag && n_flagged, "trec_topk_filter_floor: null pointer");
    return trec_topk_filter_floor_ex(tau, user_stats, user_bias, item_gstats, kdim, n_users, 2.0f, floor_, flag, n_flagged, stream);
}

// floor = tau - mult * eps (rounded down twice) with the same eps; flag / n_flagged may be NULL.  mult = 4 on tau8 gives the
// provisional floor of trec_topk_scan_blocks (the final floor tau16 - 2 eps is >= tau8 - 3 eps).
extern "C" int trec_topk_filter_floor_ex(const float* tau, const float* user_stats, const float* user_bias,
                                         const float* item_gstats, int32_t kdim, int64_t n_users, float mult, float* floor_,
                                         int32_t* flag, int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(tau && user_stats && item_gstats && floor_ && (!flag == !n_flagged) && mult >= 0.f, "trec_topk_filter_floor_ex: bad arguments");
    TREC_REQUIRE(kdim >= 1, "trec_topk_filter_floor: bad sizes");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(filter_floor_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream,
                       tau, (const float2*)user_stats, user_bias, item_gstats, kdim, n_users, mult, floor_, flag, n_flagged);
    return trec_check_launch("trec_topk_filter_floor");
}

extern "C" int trec_topk_filter_finish(const int32_t* part_idx, int32_t capacity, int32_t ksel, const int32_t* count,
                                       const float* users_f32,
                                       const float* items_f32, int64_t ld_users, int64_t ld_items, int32_t kdim,
                                       const float* user_bias, const float* item_bias, int32_t item_index_base,
                                       int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                                       int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(part_idx && count && users_f32 && items_f32 && out_vals && out_idx && flag && n_flagged,
                 "trec_topk_filter_finish: null pointer");
    TREC_REQUIRE(ksel >= 1 && ksel <= 64 && k >= 1 && k <= FILTER_CMAX, "trec_topk_filter_finish: need ksel <= 64, k <= 64");
    TREC_REQUIRE(capacity >= 1 && ksel * 2 * capacity <= 64 * 32, "trec_topk_filter_finish: ksel * 2 * capacity <= 2048");
    TREC_REQUIRE(kdim >= 1 && kdim <= 1024 && ld_users >= kdim && ld_items >= ((kdim + 3) & ~3),
                 "trec_topk_filter_finish: need kdim <= 1024 and item rows padded to a multiple of 4");
    if (n_users == 0) return TREC_OK;
    const unsigned blocks = (unsigned)ceil_div64(n_users, 4);
    hipStream_t st = (hipStream_t)stream;
    const int cpl = (ksel * 2 * capacity + 63) / 64;
    const int kd4 = (kdim + 3) & ~3;
    const size_t lds = 4 * FILTER_CMAX * 4 + (size_t)4 * kd4 * 4 + (size_t)4 * FILTER_RB * (kd4 + 4) * 4;
#define TREC_FF(CPLV)                                                                                                  \
    (void)hipFuncSetAttribute((const void*)filter_finish_kernel<CPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((filter_finish_kernel<CPLV>), dim3(blocks), dim3(256), lds, st, part_idx, capacity, ksel, count, \
                       users_f32, items_f32, ld_users, ld_items, kdim, user_bias, item_bias,          \
                       item_index_base, n_users, k, out_vals, out_idx, flag, n_flagged)
    if (cpl <= 4) { TREC_FF(4); }
    else if (cpl <= 8) { TREC_FF(8); }
    else if (cpl <= 12) { TREC_FF(12); }
    else if (cpl <= 16) { TREC_FF(16); }
    else { TREC_FF(32); }
#undef TREC_FF
    return trec_check_launch("trec_topk_filter_finish");
}

// trec_topk_filter_finish without its capacity limits (any ksel, any number of survivors; k <= 16): the wide second pass over
// the users the first pass flagged.  Same arguments; flags only users with a full stage-3 list.
extern "C" int trec_topk_filter_finish_wide(const int32_t* part_idx, int32_t capacity, int32_t ksel, const int32_t* count,
                                            const float* users_f32, const float* items_f32, int64_t ld_users, int64_t ld_items,
                                            int32_t kdim, const float* user_bias, const float* item_bias,
                                            int32_t item_index_base, int64_t n_users, int32_t k, float* out_vals,
                                            int32_t* out_idx, int32_t* flag, int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(part_idx && count && users_f32 && items_f32 && out_vals && out_idx && flag && n_flagged,
                 "trec_topk_filter_finish_wide: null pointer");
    TREC_REQUIRE(ksel >= 1 && k >= 1 && k <= 16 && capacity >= 1, "trec_topk_filter_finish_wide: need k <= 16");
    TREC_REQUIRE(kdim >= 1 && kdim <= 1024 && ld_users >= kdim && ld_items >= ((kdim + 3) & ~3),
                 "trec_topk_filter_finish_wide: need kdim <= 1024 and item rows padded to a multiple of 4");
    if (n_users == 0) return TREC_OK;
    const int kd4 = (kdim + 3) & ~3;
    const size_t lds = 4 * 128 * 4 + (size_t)4 * kd4 * 4 + (size_t)4 * FILTER_RB * (kd4 + 4) * 4;
    (void)hipFuncSetAttribute((const void*)filter_finish_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(filter_finish_wide_kernel, dim3((unsigned)ceil_div64(n_users, 4)), dim3(256), lds, (hipStream_t)stream,
                       part_idx, capacity, ksel, count, users_f32, items_f32, ld_users, ld_items, kdim, user_bias, item_bias,
                       item_index_base, n_users, k, out_vals, out_idx, flag, n_flagged);
    return trec_check_launch("trec_topk_filter_finish_wide");
}

// The finish behind the candidate lists of trec_score_gemm_refine_candidates(_hot): cand_n [n_users], cand [n_users][cand_cap]
// {item id, score bits}, cand_floor [n_users] the provisional floor the lists were made with (+inf: the user is skipped),
// user_stats [n_users][2] / item_gstats [3] as for trec_topk_filter_floor.  Writes the exact top-k; flags users with more than
// cand_cap candidates or more than 64 survivors (the caller re-does them: ops.score_topk_filtered).  out_index (nullable,
// [n_users]): user u's lists go to row out_index[u] of out_vals / out_idx, a negative entry writes nothing -- the users of
// trec_user_prep_sorted leave in the caller's order without a permutation pass.
extern "C" int trec_topk_candidates_finish(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                           const float* user_stats, const float* item_gstats, const float* users_f32,
                                           const float* items_f32, int64_t ld_users, int64_t ld_items, int32_t kdim,
                                           const float* user_bias, const float* item_bias, int32_t item_index_base,
                                           int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                                           int32_t* n_flagged, const int32_t* out_index, int32_t lanes_per_user, void* stream)
{
    TREC_REQUIRE(lanes_per_user == 0 || lanes_per_user == 16 || lanes_per_user == 64, "trec_topk_candidates_finish: lanes_per_user must be 0 / 64 (a wave per user) or 16");
    TREC_REQUIRE(cand_n && cand && cand_floor && user_stats && item_gstats && users_f32 && items_f32 && out_vals && out_idx &&
                 flag && n_flagged, "trec_topk_candidates_finish: null pointer");
    TREC_REQUIRE(cand_cap >= 64 && cand_cap % 64 == 0 && cand_cap <= 256, "trec_topk_candidates_finish: cand_cap must be 64, 128, 192 or 256");
    TREC_REQUIRE(k >= 1 && k <= 16, "trec_topk_candidates_finish: need k <= 16");
    TREC_REQUIRE(kdim >= 1 && kdim <= 1024 && ld_users >= kdim && ld_items >= ((kdim + 3) & ~3),
                 "trec_topk_candidates_finish: need kdim <= 1024 and item rows padded to a multiple of 4");
    if (n_users == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int kd4 = (kdim + 3) & ~3;
    if (lanes_per_user == 16) {
        // short lists (item shards): 16 lanes per user, 16 users per workgroup; users with more than 16 candidates are flagged
        TREC_REQUIRE(k <= 16, "trec_topk_candidates_finish: the 16-lane form needs k <= 16");
        const size_t lds16 = (size_t)16 * kd4 * 4;
        hipLaunchKernelGGL(candidates_finish16_kernel<1>, dim3((unsigned)ceil_div64(n_users, 16)), dim3(256), lds16, st, cand_n,
                           (const int2*)cand, cand_cap, cand_floor, (const float2*)user_stats, item_gstats, users_f32, items_f32,
                           ld_users, ld_items, kdim, user_bias, item_bias, item_index_base, n_users, k, out_vals, out_idx, flag,
                           n_flagged, out_index, (int32_t*)nullptr, (int32_t*)nullptr);
        return trec_check_launch("trec_topk_candidates_finish (16 lanes per user)");
    }
    const unsigned blocks = (unsigned)ceil_div64(n_users, 4);
    const size_t lds = 4 * FILTER_CMAX * 4 + (size_t)4 * kd4 * 4 + (size_t)4 * FILTER_RB * (kd4 + 4) * 4;
#define TREC_CF(CPLV)                                                                                                  \
    (void)hipFuncSetAttribute((const void*)candidates_finish_kernel<CPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((candidates_finish_kernel<CPLV>), dim3(blocks), dim3(256), lds, st, cand_n, (const int2*)cand, cand_cap, \
                       cand_floor, (const float2*)user_stats, item_gstats, users_f32, items_f32, ld_users, ld_items, kdim,   \
                       user_bias, item_bias, item_index_base, n_users, k, out_vals, out_idx, flag, n_flagged, out_index)
    if (cand_cap == 64) { TREC_CF(1); }
    else if (cand_cap == 128) { TREC_CF(2); }
    else if (cand_cap == 192) { TREC_CF(3); }
    else { TREC_CF(4); }
#undef TREC_CF
    return trec_check_launch("trec_topk_candidates_finish");
}

// ---- the finish of the WIDE route (17 <= k <= 64, lists of up to 1,024 candidates): one wave per user, lane j owns candidates j,
// 64 + j, ... (CPL per lane), re-scores every one of them with the reference's k-ordered fmaf chain straight from its fp32 item row
// (the user row broadcast from LDS) and the wave takes the k best by k rounds of a 64-bit wave maximum -- no floor, no survivor limit:
// a list holds every item whose fp32 score reaches the k-th best, so its k best ARE the answer.  Replaces a finish made of
// library calls and dense [n_users, 1024] torch masks.  A user whose list is incomplete (more entries than slots, fewer than k) is
// flagged; a user flagged before, or whose provisional floor is +inf, is skipped (-inf / -1 rows).
template <int CPL>
__global__ __launch_bounds__(256) void candidates_finish_wide_kernel(
    const int32_t* __restrict__ cand_n, const int2* __restrict__ cand_list, int cap, const float* __restrict__ cand_floor,
    const float* __restrict__ U, const float* __restrict__ V, int64_t ld_u, int64_t ld_v, int kdim,
    const float* __restrict__ user_bias, const float* __restrict__ item_bias, int32_t item_index_base, int64_t n_users, int k,
    float* __restrict__ ov, int32_t* __restrict__ oi, int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
    const int32_t* __restrict__ out_index)
{
    extern __shared__ __attribute__((aligned(16))) char fsmem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= n_users) return;
    const int64_t uo = out_index ? (int64_t)out_index[u] : u;
    if (uo < 0) return;
    const int kd4 = (kdim + 3) & ~3;
    float* urow = (float*)fsmem + (size_t)wave * kd4;
    const int n = cand_n[u];
    const float f0 = cand_floor[u];
    const bool vec = ((ld_v & 3) == 0) && ((ld_u & 3) == 0);
    for (int ch = lane; ch < (kd4 >> 2); ch += 64) {
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        const float* src = U + u * ld_u + ch * 4;
        if (vec) w = *(const f32x4*)src;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (ch * 4 + e < kdim) ? src[e] : 0.f;
        }
        *(f32x4*)(urow + ch * 4) = w;
    }
    const float bu = user_bias ? user_bias[u] : 0.f;
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    const bool skip = !(f0 < INFINITY) || flag[u] != 0;      // (wave-uniform)
    const bool bad = n > cap || n < k;
    if (skip || bad) {
        if (bad && !skip && lane == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
        if (lane < k) { ov[uo * k + lane] = -INFINITY; oi[uo * k + lane] = -1; }
        return;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long key[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        key[c] = EMPTY;
        if (c * 64 >= n) continue;                            // (wave-uniform)
        const bool have = c * 64 + lane < n;
        const int2 ent = cand_list[u * (int64_t)cap + (have ? c * 64 + lane : 0)];
        const int64_t it = (int64_t)ent.x - item_index_base;
        const float* b = V + it * ld_v;
        float acc = 0.0f;
        int kk = 0;
        if (vec) {
            for (; kk + 4 <= kdim; kk += 4) {
                const f32x4 a4 = *(const f32x4*)(urow + kk);
                const f32x4 b4 = *(const f32x4*)(b + kk);
                acc = __fmaf_rn(a4[0], b4[0], acc); acc = __fmaf_rn(a4[1], b4[1], acc);
                acc = __fmaf_rn(a4[2], b4[2], acc); acc = __fmaf_rn(a4[3], b4[3], acc);
            }
        }
        for (; kk < kdim; ++kk) acc = __fmaf_rn(urow[kk], b[kk], acc);
        if (user_bias) acc = acc + bu;
        if (item_bias) acc = acc + item_bias[it];
        if (have) key[c] = merge_key(acc, ent.x);
    }
    // ---- the k best by (value desc, index asc): lane t keeps place t (an item is listed once: keys are unique, and equal keys -- the
    // harmless double listing of a pre-refined pair inside a hot superblock -- leave together)
    unsigned long long place = EMPTY;
    for (int t = 0; t < k; ++t) {
        unsigned long long m = key[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) m = key[c] > m ? key[c] : m;
        const unsigned long long best = wave_max_u64(m);
#pragma unroll
        for (int c = 0; c < CPL; ++c) if (key[c] == best && best != EMPTY) key[c] = EMPTY;
        if (lane == t) place = best;
    }
    if (lane < k) {
        const unsigned int hi = (unsigned int)(place >> 32);
        const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
        ov[uo * k + lane] = (place == EMPTY) ? -INFINITY : __uint_as_float(bits);
        oi[uo * k + lane] = (place == EMPTY) ? -1 : (int32_t)(~(unsigned int)place);
    }
}

extern "C" int trec_topk_candidates_finish_wide(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                                const float* users_f32, const float* items_f32, int64_t ld_users, int64_t ld_items,
                                                int32_t kdim, const float* user_bias, const float* item_bias, int32_t item_index_base,
                                                int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                                                int32_t* n_flagged, const int32_t* out_index, void* stream)
{
    TREC_REQUIRE(cand_n && cand && cand_floor && users_f32 && items_f32 && out_vals && out_idx && flag && n_flagged,
                 "trec_topk_candidates_finish_wide: null pointer");
    TREC_REQUIRE(cand_cap >= 64 && cand_cap % 64 == 0 && cand_cap <= 1024, "trec_topk_candidates_finish_wide: cand_cap must be a multiple of 64 up to 1024");
    TREC_REQUIRE(k >= 1 && k <= 64, "trec_topk_candidates_finish_wide: need k <= 64");
    TREC_REQUIRE(kdim >= 1 && kdim <= 1024 && ld_users >= kdim && ld_items >= ((kdim + 3) & ~3),
                 "trec_topk_candidates_finish_wide: need kdim <= 1024 and item rows padded to a multiple of 4");
    if (n_users == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int kd4 = (kdim + 3) & ~3;
    const size_t lds = (size_t)4 * kd4 * 4;
    const unsigned blocks = (unsigned)ceil_div64(n_users, 4);
#define TREC_CFW(CPLV)                                                                                                          \
    hipLaunchKernelGGL((candidates_finish_wide_kernel<CPLV>), dim3(blocks), dim3(256), lds, st, cand_n, (const int2*)cand, cand_cap, \
                       cand_floor, users_f32, items_f32, ld_users, ld_items, kdim, user_bias, item_bias, item_index_base, n_users, k, \
                       out_vals, out_idx, flag, n_flagged, out_index)
    if (cand_cap <= 256) { TREC_CFW(4); }
    else if (cand_cap <= 512) { TREC_CFW(8); }
    else { TREC_CFW(16); }
#undef TREC_CFW
    return trec_check_launch("trec_topk_candidates_finish_wide");
}

// The finish in two launches for lists of MIXED length (the single-GPU cascade: ~15 candidates per user, a few users with hundreds):
// (1) 16 lanes per user, cands_per_lane (1 / 2 / 4) candidates per lane -- four users per wave instead of one; users whose list is
// longer than 16 * cands_per_lane are appended to over_list [n_users] (over_count [1] zeroed by the caller); (2) the wave-per-user
// finish over that list on a fixed grid, its length read on the device.  Same results as trec_topk_candidates_finish.
extern "C" int trec_topk_candidates_finish_mixed(const int32_t* cand_n, const void* cand, int32_t cand_cap, const float* cand_floor,
                                                 const float* user_stats, const float* item_gstats, const float* users_f32,
                                                 const float* items_f32, int64_t ld_users, int64_t ld_items, int32_t kdim,
                                                 const float* user_bias, const float* item_bias, int32_t item_index_base,
                                                 int64_t n_users, int32_t k, float* out_vals, int32_t* out_idx, int32_t* flag,
                                                 int32_t* n_flagged, const int32_t* out_index, int32_t cands_per_lane,
                                                 int32_t* over_list, int32_t* over_count, void* stream)
{
    TREC_REQUIRE(cands_per_lane == 1 || cands_per_lane == 2 || cands_per_lane == 4, "trec_topk_candidates_finish_mixed: cands_per_lane must be 1, 2 or 4");
    TREC_REQUIRE(cand_n && cand && cand_floor && user_stats && item_gstats && users_f32 && items_f32 && out_vals && out_idx &&
                 flag && n_flagged && over_list && over_count, "trec_topk_candidates_finish_mixed: null pointer");
    TREC_REQUIRE(cand_cap >= 64 && cand_cap % 64 == 0 && cand_cap <= 256, "trec_topk_candidates_finish_mixed: cand_cap must be 64, 128, 192 or 256");
    TREC_REQUIRE(k >= 1 && k <= 16, "trec_topk_candidates_finish_mixed: need k <= 16");
    TREC_REQUIRE(kdim >= 1 && kdim <= 1024 && ld_users >= kdim && ld_items >= ((kdim + 3) & ~3) && n_users < ((int64_t)1 << 31),
                 "trec_topk_candidates_finish_mixed: need kdim <= 1024, item rows padded to a multiple of 4, n_users < 2^31");
    if (n_users == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int kd4 = (kdim + 3) & ~3;
    const size_t lds16 = (size_t)16 * kd4 * 4;
#define TREC_CF16(CPLV)                                                                                                         \
    hipLaunchKernelGGL(candidates_finish16_kernel<CPLV>, dim3((unsigned)ceil_div64(n_users, 16)), dim3(256), lds16, st, cand_n,  \
                       (const int2*)cand, cand_cap, cand_floor, (const float2*)user_stats, item_gstats, users_f32, items_f32,    \
                       ld_users, ld_items, kdim, user_bias, item_bias, item_index_base, n_users, k, out_vals, out_idx, flag,     \
                       n_flagged, out_index, over_list, over_count)
    if (cands_per_lane == 1) TREC_CF16(1); else if (cands_per_lane == 2) TREC_CF16(2); else TREC_CF16(4);
#undef TREC_CF16
    const int64_t want = ceil_div64(n_users, 4 * 64);                    // (a wave of the listed kernel per ~64 users: lists that long are rare)
    const unsigned blocks = (unsigned)(want < 64 ? 64 : (want > 4096 ? 4096 : want));
    const size_t lds = 4 * FILTER_CMAX * 4 + (size_t)4 * kd4 * 4 + (size_t)4 * FILTER_RB * (kd4 + 4) * 4;
#define TREC_CFL(CPLV)                                                                                                          \
    (void)hipFuncSetAttribute((const void*)candidates_finish_listed_kernel<CPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((candidates_finish_listed_kernel<CPLV>), dim3(blocks), dim3(256), lds, st, (const int32_t*)over_list,      \
                       (const int32_t*)over_count, cand_n, (const int2*)cand, cand_cap, cand_floor, (const float2*)user_stats,   \
                       item_gstats, users_f32, items_f32, ld_users, ld_items, kdim, user_bias, item_bias, item_index_base, n_users, \
                       k, out_vals, out_idx, flag, n_flagged, out_index)
    if (cand_cap == 64) { TREC_CFL(1); }
    else if (cand_cap == 128) { TREC_CFL(2); }
    else if (cand_cap == 192) { TREC_CFL(3); }
    else { TREC_CFL(4); }
#undef TREC_CFL
    return trec_check_launch("trec_topk_candidates_finish_mixed");
}

// The cascade's thresholds in one pass (cascade_floor_kernel): tau [n_users] IN / OUT (+inf for rows without a source), src
// nullable [n_users] (trec_user_prep_sorted), floor0 / flag [n_users] out, n_flagged [1] zeroed by the caller, cand_n nullable
// [n_users] zeroed here.  Must run before trec_topk_rows_collect (which reads tau).
extern "C" int trec_topk_cascade_floor(float* tau, const int32_t* src, const float* user_stats, const float* user_bias,
                                       const float* item_gstats, int32_t kdim, int64_t n_users, float* floor0, int32_t* flag,
                                       int32_t* n_flagged, int32_t* cand_n, void* stream)
{
    TREC_REQUIRE(tau && user_stats && item_gstats && floor0 && flag && n_flagged, "trec_topk_cascade_floor: null pointer");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(cascade_floor_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream, tau, src,
                       (const float2*)user_stats, user_bias, item_gstats, kdim, n_users, floor0, flag, n_flagged, cand_n);
    return trec_check_launch("trec_topk_cascade_floor");
}

// The cascade's pre-refinement, step 1 (prerefine_rows_kernel): from the selection over the TAGGED chunk lists (sel [n_users][k]
// rows, sel_val [k][n_users] values; lists written with top_k | 0x100) to per-superblock user lists in the fixed-capacity layout of
// trec_topk_rows_collect (row_count [n_sb] zeroed by the caller, row_user [n_sb][rcap], rcap % 512 == 0) for
// trec_score_gemm_blockmax_grouped, plus sel_sb [n_users][k] (superblock ids, -1 = empty slot) and ok [n_users].
// n_sb * 4 bytes of LDS: n_sb <= trec_topk_prerefine_max_superblocks().
extern "C" int32_t trec_topk_prerefine_max_superblocks(void) { return 16000; }

extern "C" int trec_topk_prerefine_rows(const int32_t* sel, const float* sel_val, int32_t k, int32_t top_k, int32_t sb_per_chunk,
                                        int32_t n_sb, int64_t n_users, const int32_t* src, int32_t rcap, int32_t* sel_sb,
                                        int32_t* row_count, int32_t* row_user, int32_t* ok, void* stream)
{
    TREC_REQUIRE(sel && sel_val && sel_sb && row_count && row_user && ok, "trec_topk_prerefine_rows: null pointer");
    TREC_REQUIRE(k >= 1 && k <= 16 && top_k >= 1 && sb_per_chunk >= 1 && sb_per_chunk <= (1 << TREC_LB_TAG_BITS) && n_sb >= 1 &&
                 n_sb <= trec_topk_prerefine_max_superblocks() && rcap >= 512 && rcap % 512 == 0, "trec_topk_prerefine_rows: bad sizes");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(prerefine_rows_kernel<16>, dim3((unsigned)ceil_div64(n_users, 1024)), dim3(256), (size_t)n_sb * 4, (hipStream_t)stream,
                       sel, sel_val, k, top_k, sb_per_chunk, n_sb, n_users, src, rcap, sel_sb, row_count, row_user, ok, (int32_t*)nullptr);
    return trec_check_launch("trec_topk_prerefine_rows");
}

// ... and with sel_pos [n_users][k]: the position of every placed (user, slot) pair inside its superblock's list (undefined where
// sel_sb is -1) -- what trec_topk_prerefine_tau_listed reads the maxima back by.
extern "C" int trec_topk_prerefine_rows_pos(const int32_t* sel, const float* sel_val, int32_t k, int32_t top_k, int32_t sb_per_chunk,
                                            int32_t n_sb, int64_t n_users, const int32_t* src, int32_t rcap, int32_t* sel_sb,
                                            int32_t* row_count, int32_t* row_user, int32_t* ok, int32_t* sel_pos, void* stream)
{
    TREC_REQUIRE(sel && sel_val && sel_sb && row_count && row_user && ok && sel_pos, "trec_topk_prerefine_rows_pos: null pointer");
    TREC_REQUIRE(k >= 1 && k <= 64 && top_k >= 1 && sb_per_chunk >= 1 && sb_per_chunk <= (1 << TREC_LB_TAG_BITS) && n_sb >= 1 &&
                 n_sb <= trec_topk_prerefine_max_superblocks() && rcap >= 512 && rcap % 512 == 0, "trec_topk_prerefine_rows_pos: bad sizes");
    if (n_users == 0) return TREC_OK;
    if (k <= 16)
        hipLaunchKernelGGL(prerefine_rows_kernel<16>, dim3((unsigned)ceil_div64(n_users, 1024)), dim3(256), (size_t)n_sb * 4,
                           (hipStream_t)stream, sel, sel_val, k, top_k, sb_per_chunk, n_sb, n_users, src, rcap, sel_sb, row_count, row_user,
                           ok, sel_pos);
    else                                                    // (the wide route, 17 <= k <= 64: one user per thread)
        hipLaunchKernelGGL(prerefine_rows_kernel<64>, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), (size_t)n_sb * 4,
                           (hipStream_t)stream, sel, sel_val, k, top_k, sb_per_chunk, n_sb, n_users, src, rcap, sel_sb, row_count, row_user,
                           ok, sel_pos);
    return trec_check_launch("trec_topk_prerefine_rows_pos");
}

// Step 2 behind trec_score_gemm_refine_candidates_marked: tau [n_users] IN / OUT raised to min_j pre_max[sel_sb[u][j] * rcap +
// sel_pos[u][j]] - eps_u where that is larger (users with ok[u]), vals [n_users][k] = those maxima (+inf where no pair was placed),
// cand_floor [n_users] raised to the new tau - eps where it was finite.  The table is not touched: the launch marked its entries.
extern "C" int trec_topk_prerefine_tau_listed(const int32_t* sel_sb, const int32_t* sel_pos, const int32_t* ok, int32_t k,
                                              const float* pre_max, int32_t rcap, int64_t n_users, const int32_t* src,
                                              const float* user_stats, const float* user_bias, const float* item_gstats, int32_t kdim,
                                              float* tau, float* vals, float* cand_floor, void* stream)
{
    TREC_REQUIRE(sel_sb && sel_pos && ok && pre_max && user_stats && item_gstats && tau && vals && k >= 1 && k <= 64 && rcap >= 1,
                 "trec_topk_prerefine_tau_listed: bad arguments");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(prerefine_tau_listed_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream, sel_sb,
                       sel_pos, ok, k, pre_max, (int64_t)rcap, n_users, src, (const float2*)user_stats, user_bias, item_gstats, kdim, tau,
                       vals, cand_floor);
    return trec_check_launch("trec_topk_prerefine_tau_listed");
}

// Step 2, after the bf16 launch over those lists (prerefine_tau_kernel): tau [n_users] IN / OUT is raised to
// min_j table[sel_sb[u][j]][u] - eps_u where that is larger (users with ok[u]).  listed == 0 (the launch was
// trec_score_gemm_blockmax_grouped): every listed table entry becomes +inf.  listed != 0 (trec_score_gemm_refine_candidates, which
// also listed the candidates): vals [n_users][k] receives the bf16 maxima, the entries become -inf (the compaction drops the pairs)
// and cand_floor [n_users] rises to the new tau - eps where it was finite.
extern "C" int trec_topk_prerefine_tau(const int32_t* sel_sb, const int32_t* ok, int32_t k, float* table, int64_t stride,
                                       int64_t n_users, const int32_t* src, const float* user_stats, const float* user_bias,
                                       const float* item_gstats, int32_t kdim, float* tau, int32_t listed, float* vals,
                                       float* cand_floor, void* stream)
{
    TREC_REQUIRE(sel_sb && ok && table && user_stats && item_gstats && tau && k >= 1 && k <= 16 && stride >= n_users,
                 "trec_topk_prerefine_tau: bad arguments");
    TREC_REQUIRE(!listed || vals, "trec_topk_prerefine_tau: the listed form needs vals");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(prerefine_tau_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream, sel_sb, ok, k,
                       table, stride, n_users, src, (const float2*)user_stats, user_bias, item_gstats, kdim, tau, listed, vals,
                       cand_floor);
    return trec_check_launch("trec_topk_prerefine_tau");
}
