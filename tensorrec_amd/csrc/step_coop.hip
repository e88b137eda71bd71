// tensorrec_amd/csrc/step_coop.hip -- ONE kernel per training step for models that fit on chip (BASELINE.json configs[1]: 943 x 1,682,
// d = 64, WMRB, S = 168).
//
// The reference runs `session.run(tf_optimizer)` once per batch (tensorrec/tensorrec.py:617-622): sampler callback, both representation
// graphs, serial + sampled predictions, WMRB, autodiff, Adam on every variable.  As separate launches that is ~25 kernels of a few
// microseconds of work each, replayed from a HIP graph at ~27 us per node: 0.68 ms per epoch, 1 % of any roofline.  Here the whole step
// is one COOPERATIVE launch (hipLaunchCooperativeKernel: every workgroup resident, grid-wide barriers between the phases), one
// persistent workgroup per CU:
//   phase 1  item tower forward, K1's arithmetic (fmaf in CSR order): V = X_i . W_i, b_i = X_i . beta_i; G (the dense coefficient
//            matrix [n_users, ldg]) cleared
//   phase 2  per user (identity user features: the user's representation IS its row of W_u): the samples drawn in the kernel (the same
//            Philox / Feistel bits as trec_sample_items) or read from a table, then the tiled WMRB body of csrc/wmrb_tiled_body.hpp --
//            scores in LDS, loss, coefficients scatter-added into G, dU from a second sweep over the rows
//   phase 3  d V = G^T . U (the item side of the backward pass; G is ~15 % dense at S = 168 of 1,682 items): tiles of 16 items x a
//            segment of users staged in LDS, d b_i = column sums of G
//   phase 4  item tower backward on the transposed CSR (d W_i = X_i^T . dV, d beta_i = X_i^T . d b_i, fmaf in transposed-CSR order)
//            with the TF-form Adam update of every row right behind its gradient; Adam on the users' rows
// Results: within summation order of the multi-launch path (same bar against the oracle: tests/test_gpu_shapes.py, the configs[1]
// record of bench.py).  The Adam arithmetic is adam.hip's (individually rounded operations: compiled with -ffp-contract=off).
#include "wmrb_tiled_body.hpp"
#include "sampler_common.hpp"
#include <hip/hip_cooperative_groups.h>

namespace cg = cooperative_groups;

namespace {

__device__ __forceinline__ void adam_elem(float& w, float& m, float& v, float g, float lr_t, float omb1, float omb2, float eps,
                                          float l2)
{
    const float gg = (l2 != 0.f) ? __fadd_rn(g, __fmul_rn(w, l2)) : g;
    m = __fadd_rn(m, __fmul_rn(__fsub_rn(gg, m), omb1));
    v = __fadd_rn(v, __fmul_rn(__fsub_rn(__fmul_rn(gg, gg), v), omb2));
    w = __fsub_rn(w, __fdiv_rn(__fmul_rn(m, lr_t), __fadd_rn(sqrtf(v), eps)));
}

constexpr int COOP_IB = 16;           // items per workgroup tile of phase 3
constexpr int COOP_P3_LDS = 32 * 1024;   // LDS of a phase-3 tile: a segment of user rows + their coefficients for COOP_IB items

// users per phase-3 segment: at least 8 segments (tiles = item blocks x segments must cover every workgroup), and a segment's
// [seg_len][COOP_IB + d] floats within COOP_P3_LDS
__host__ __device__ inline int64_t coop_seg_len(int64_t n_users, int d)
{
    int64_t by_lds = COOP_P3_LDS / (4 * (COOP_IB + d));
    int64_t by_count = (n_users + 7) / 8;
    int64_t s = by_count < by_lds ? by_count : by_lds;
    if (s > 128) s = 128;                    // (phase 3 stages a segment with 8 loads per thread: 128 x 16 coefficients, <= 2,048 float4 of rows)
    return s < 16 ? 16 : s;
}

struct CoopArgs {
    // weights and their Adam slots (updated in place)
    float* Wu; float* Wu_m; float* Wu_v;            // [n_users, d]   (identity user features)
    float* Wi; float* Wi_m; float* Wi_v;            // [n_item_features, d]
    float* bu; float* bu_m; float* bu_v;            // [n_users] or null (unbiased)
    float* bi; float* bi_m; float* bi_v;            // [n_item_features] or null
    // item features: CSR and the CSR of the transpose (values through perm_t)
    const int64_t* f_indptr; const int32_t* f_indices; const float* f_values;
    const int64_t* ft_indptr; const int32_t* ft_rows; const int32_t* ft_perm;
    // interactions
    const int64_t* indptr; const int32_t* xi; const int32_t* pos_slot; const float* pos_weight;
    const int32_t* samples;                         // [n_users, S] or null: drawn here
    // workspace
    float* V; float* ib; float* G; float* dU; float* dub; float* dV; float* dib;
    // outputs
    float* loss; float* pred_serial;
    int64_t n_users, n_items, n_item_features, ldg, user_base;
    int32_t S, d, max_rows, max_pos, sample_bits;
    uint32_t seed_lo, seed_hi, step;
    float ratio, lr_t, beta1, beta2, eps, l2;
    long long* clk;                                 // diagnostics (tuning coop_clocks): wall_clock64 of workgroup 0 at the phase boundaries, or null
};

template <int ITERS, int RB, int LPR>
__global__ __launch_bounds__(256) void fit_step_coop_kernel(CoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    cg::grid_group grid = cg::this_grid();
    const int tid = threadIdx.x;
    const int d = a.d, d4 = a.d >> 2;
    const int lpr = d4 <= 16 ? 16 : 32;                  // lanes that share a row (float4 each); rows per workgroup pass = 256 / lpr
    const int sub = tid % lpr, grp = tid / lpr, n_grp = 256 / lpr;
    const bool col_ok = sub < d4;
#define COOP_STAMP(i) do { if (a.clk && blockIdx.x == 0 && tid == 0) a.clk[i] = wall_clock64(); } while (0)
    COOP_STAMP(0);

    // ---- phase 1: item tower forward + clear G -------------------------------------------------------------------------------
    for (int64_t r = (int64_t)blockIdx.x * n_grp + grp; r < a.n_items; r += (int64_t)gridDim.x * n_grp) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float accb = 0.f;
        for (int64_t j = a.f_indptr[r]; j < a.f_indptr[r + 1]; ++j) {
            const float x = a.f_values[j];
            const int64_t c = a.f_indices[j];
            if (col_ok) {
                const f32x4 w = *(const f32x4*)(a.Wi + c * d + sub * 4);
                acc.x = fmaf(x, w.x, acc.x); acc.y = fmaf(x, w.y, acc.y); acc.z = fmaf(x, w.z, acc.z); acc.w = fmaf(x, w.w, acc.w);
            }
            if (a.bi) accb = fmaf(x, a.bi[c], accb);
        }
        if (col_ok) *(f32x4*)(a.V + r * d + sub * 4) = acc;
        if (a.bi && sub == 0) a.ib[r] = accb;
    }
    {
        const int64_t n4 = (a.n_users * a.ldg) >> 2;     // (ldg % 4 == 0)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n4; i += (int64_t)gridDim.x * 256) ((f32x4*)a.G)[i] = z;
        // (d V [n_items, d] and d b_i [n_items, padded to 4] follow each other in the workspace: phase 3 adds its segments into them)
        const int64_t m4 = (a.n_items * d + ((a.n_items + 3) & ~(int64_t)3)) >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < m4; i += (int64_t)gridDim.x * 256) ((f32x4*)a.dV)[i] = z;
    }
    COOP_STAMP(1);
    grid.sync();
    COOP_STAMP(2);

    // ---- phase 2: the users ---------------------------------------------------------------------------------------------------
    {
        int32_t* l_samp = (int32_t*)(lds + tiled_lds_floats(a.max_rows, a.max_pos, d));      // behind the body's own arrays
        TiledOut o = {a.loss, a.pred_serial, a.dU, a.bu ? a.dub : nullptr, nullptr, nullptr, nullptr, nullptr, a.G, a.ldg, nullptr};
        for (int64_t u = blockIdx.x; u < a.n_users; u += gridDim.x) {
            const int32_t* samp;
            if (a.samples) samp = a.samples + u * a.S;
            else {
                const SampleKeys keys = sample_keys(u + a.user_base, a.step, a.seed_lo, a.seed_hi);
                for (int s = tid; s < a.S; s += 256) l_samp[s] = sample_distinct((uint32_t)s, a.sample_bits, keys, (int32_t)a.n_items);
                __syncthreads();
                samp = l_samp;
            }
            wmrb_tiled_user<ITERS, RB, 0, LPR>(lds, u, a.Wu, a.V, a.bu, a.bi ? a.ib : nullptr, a.indptr, a.xi, a.pos_slot, a.pos_weight, samp,
                                          a.S, d, a.ratio, a.max_rows, a.max_pos, o);
        }
    }
    COOP_STAMP(3);
    grid.sync();
    COOP_STAMP(4);

    // ---- phase 3: d V += G^T . U, d b_i += column sums of G.  A tile = COOP_IB items x one segment of users; the segment's user
    // rows and its coefficients for these items are staged in LDS (coalesced), every thread = (item, float4 column group) then runs
    // over the segment from LDS; the segments' partial sums meet in d V by float atomics (order-free up to rounding) ----
    {
        const int64_t n_tiles_i = (a.n_items + COOP_IB - 1) / COOP_IB;
        const int64_t seg_len = coop_seg_len(a.n_users, d);
        const int64_t n_seg = (a.n_users + seg_len - 1) / seg_len;
        float* l_g = lds;                                  // [seg_len][COOP_IB]
        float* l_u = lds + seg_len * COOP_IB;              // [seg_len][d]
        const int k = tid / 16, c4 = tid % 16;
        for (int64_t t = blockIdx.x; t < n_tiles_i * n_seg; t += gridDim.x) {
            const int64_t ti = t / n_seg, seg = t % n_seg;
            const int64_t i0 = ti * COOP_IB, u0 = seg * seg_len;
            const int64_t u1 = u0 + seg_len < a.n_users ? u0 + seg_len : a.n_users;
            const int nu = (int)(u1 - u0);
            __syncthreads();
            {
                // every load of the tile leaves before the first LDS store (a copy loop of unknown length is one round trip per
                // iteration: measured 20 us per tile): at most 8 + 8 per thread by coop_seg_len's caps
                float gq[8];
                f32x4 uq[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = tid + 256 * q;
                    const int ec = e < nu * COOP_IB ? e : 0;
                    const int uu = ec / COOP_IB, kk = ec % COOP_IB;
                    const int64_t col = i0 + kk < a.n_items ? i0 + kk : a.n_items - 1;
                    gq[q] = a.G[(u0 + uu) * a.ldg + col];
                    if (i0 + kk >= a.n_items) gq[q] = 0.f;
                    const int e4 = e < nu * d4 ? e : 0;
                    uq[q] = ((const f32x4*)(a.Wu + u0 * d))[e4];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = tid + 256 * q;
                    if (e < nu * COOP_IB) l_g[e] = gq[q];
                    if (e < nu * d4) ((f32x4*)l_u)[e] = uq[q];
                }
            }
            __syncthreads();
            if (i0 + k < a.n_items) {
                for (int cc = c4; cc < d4; cc += 16) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                    for (int uu = 0; uu < nu; ++uu) {
                        const float g = l_g[uu * COOP_IB + k];
                        const f32x4 w = *(const f32x4*)(l_u + uu * d + cc * 4);
                        acc.x = fmaf(g, w.x, acc.x); acc.y = fmaf(g, w.y, acc.y); acc.z = fmaf(g, w.z, acc.z); acc.w = fmaf(g, w.w, acc.w);
                    }
                    float* out = a.dV + (i0 + k) * d + cc * 4;
                    unsafeAtomicAdd(out + 0, acc.x); unsafeAtomicAdd(out + 1, acc.y); unsafeAtomicAdd(out + 2, acc.z); unsafeAtomicAdd(out + 3, acc.w);
                }
                if (a.bi && c4 == 0) {
                    float bsum = 0.f;
                    for (int uu = 0; uu < nu; ++uu) bsum += l_g[uu * COOP_IB + k];
                    unsafeAtomicAdd(a.dib + i0 + k, bsum);
                }
            }
        }
    }
    COOP_STAMP(5);
    grid.sync();
    COOP_STAMP(6);

    // ---- phase 4: item tower backward + Adam; Adam on the users' rows ------------------------------------------------------------
    {
        const float omb1 = __fsub_rn(1.0f, a.beta1), omb2 = __fsub_rn(1.0f, a.beta2);
        float* l_red = lds;                                 // [n_grp][d + 1] partial sums of one feature row
        for (int64_t f = blockIdx.x; f < a.n_item_features; f += gridDim.x) {
            const int64_t j0 = a.ft_indptr[f], j1 = a.ft_indptr[f + 1];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float accb = 0.f;
            // the row's entries are dealt to the groups in turn; every group keeps CSR order inside its share, the shares are added
            // in group order (deterministic)
            for (int64_t jb = j0 + grp; jb < j1; jb += 4 * n_grp) {          // four of the group's entries in flight
                int32_t pj[4], rj[4];
                float xj[4];
                f32x4 gj[4];
                float bj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t j = jb + q * n_grp < j1 ? jb + q * n_grp : j1 - 1;
                    pj[q] = a.ft_perm[j]; rj[q] = a.ft_rows[j];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xj[q] = a.f_values[pj[q]];
                    gj[q] = *(const f32x4*)(a.dV + (int64_t)rj[q] * d + (col_ok ? sub * 4 : 0));
                    bj[q] = a.bi ? a.dib[rj[q]] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (jb + q * n_grp < j1) {
                        acc.x = fmaf(xj[q], gj[q].x, acc.x); acc.y = fmaf(xj[q], gj[q].y, acc.y);
                        acc.z = fmaf(xj[q], gj[q].z, acc.z); acc.w = fmaf(xj[q], gj[q].w, acc.w);
                        accb = fmaf(xj[q], bj[q], accb);
                    }
                }
            }
            __syncthreads();
            if (col_ok) *(f32x4*)(l_red + grp * (d + 4) + sub * 4) = acc;
            if (sub == 0) l_red[grp * (d + 4) + d] = accb;
            __syncthreads();
            if (grp == 0) {
                if (col_ok) {
                    f32x4 g = *(const f32x4*)(l_red + sub * 4);
                    for (int q = 1; q < n_grp; ++q) {
                        const f32x4 p = *(const f32x4*)(l_red + q * (d + 4) + sub * 4);
                        g.x += p.x; g.y += p.y; g.z += p.z; g.w += p.w;
                    }
                    f32x4 w = *(f32x4*)(a.Wi + f * d + sub * 4), m = *(f32x4*)(a.Wi_m + f * d + sub * 4), v = *(f32x4*)(a.Wi_v + f * d + sub * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float we = w[e], me = m[e], ve = v[e]; adam_elem(we, me, ve, g[e], a.lr_t, omb1, omb2, a.eps, a.l2); w[e] = we; m[e] = me; v[e] = ve; }
                    *(f32x4*)(a.Wi + f * d + sub * 4) = w; *(f32x4*)(a.Wi_m + f * d + sub * 4) = m; *(f32x4*)(a.Wi_v + f * d + sub * 4) = v;
                }
                if (a.bi && sub == 0) {
                    float g = l_red[d];
                    for (int q = 1; q < n_grp; ++q) g += l_red[q * (d + 4) + d];
                    float w = a.bi[f], m = a.bi_m[f], v = a.bi_v[f];
                    adam_elem(w, m, v, g, a.lr_t, omb1, omb2, a.eps, a.l2);      // (the bias variables are regularised too: tensorrec.py:313, :487)
                    a.bi[f] = w; a.bi_m[f] = m; a.bi_v[f] = v;
                }
            }
        }
        for (int64_t u = (int64_t)blockIdx.x * n_grp + grp; u < a.n_users; u += (int64_t)gridDim.x * n_grp) {
            if (col_ok) {
                const f32x4 g = *(const f32x4*)(a.dU + u * d + sub * 4);
                f32x4 w = *(f32x4*)(a.Wu + u * d + sub * 4), m = *(f32x4*)(a.Wu_m + u * d + sub * 4), v = *(f32x4*)(a.Wu_v + u * d + sub * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { float we = w[e], me = m[e], ve = v[e]; adam_elem(we, me, ve, g[e], a.lr_t, omb1, omb2, a.eps, a.l2); w[e] = we; m[e] = me; v[e] = ve; }
                *(f32x4*)(a.Wu + u * d + sub * 4) = w; *(f32x4*)(a.Wu_m + u * d + sub * 4) = m; *(f32x4*)(a.Wu_v + u * d + sub * 4) = v;
            }
            if (a.bu && sub == 0) {
                float w = a.bu[u], m = a.bu_m[u], v = a.bu_v[u];
                adam_elem(w, m, v, a.dub[u], a.lr_t, omb1, omb2, a.eps, a.l2);
                a.bu[u] = w; a.bu_m[u] = m; a.bu_v[u] = v;
            }
        }
    }
    COOP_STAMP(7);
#undef COOP_STAMP
}

int64_t coop_lds_bytes(int32_t n_sampled, int32_t max_pos, int32_t d, int64_t n_users)
{
    const int64_t p2 = (tiled_lds_floats((int64_t)n_sampled + max_pos, max_pos, d) + n_sampled + 4) * 4;
    const int64_t p3 = coop_seg_len(n_users, d) * (COOP_IB + d) * 4;
    const int64_t p4 = 16 * ((int64_t)d + 4) * 4;
    int64_t m = p2 > p3 ? p2 : p3;
    return m > p4 ? m : p4;
}

}  // namespace

// Workspace floats of trec_fit_step_coop, or -1 when the model is not covered: d % 4 == 0, d <= 128, the LDS of its phases within
// 64 KB (n_sampled + longest interaction row in the low thousands; n_users <= ~4,000), ldg = n_items rounded up to 4.
// Layout: V [n_items, d] | ib [n_items] | G [n_users, ldg] | dU [n_users, d] | dub [n_users] | dV [n_items, d] | dib [n_items]
extern "C" int64_t trec_fit_step_coop_workspace_floats(int64_t n_users, int64_t n_items, int32_t d, int32_t n_sampled,
                                                       int32_t max_interactions_per_user)
{
    if (n_users < 1 || n_items < 1 || d < 4 || d % 4 != 0 || d > 128 || n_sampled < 1 || n_sampled > n_items ||
        max_interactions_per_user < 0) return -1;
    if (coop_lds_bytes(n_sampled, max_interactions_per_user, d, n_users) > 64 * 1024) return -1;
    const int64_t ldg = (n_items + 3) / 4 * 4;
    if (n_users * ldg > ((int64_t)1 << 26)) return -1;                 // G up to 256 MB: beyond that the multi-launch path is not launch-bound
    return n_items * d + n_items + n_users * ldg + n_users * d + n_users + n_items * d + n_items + 64;
}

// One optimiser step of Linear (identity user features) + Linear (any item features) + DotProduct + WMRB / BalancedWMRB in ONE
// cooperative launch (tensorrec.py:617-622 for this model family).  Weights and Adam slots are updated in place; loss [P+] and
// pred_serial [nnz] are written for the caller's log.  samples: [n_users, n_sampled] int32, or NULL -- the kernel then draws what
// trec_sample_items(n_users, user_base, n_items, n_sampled, 0, seed, step) would.  lr_t / l2: as trec_adam_tf_step (l2 applies to the
// two weight tables, not to the biases).  bias pointers: all six or none.  Returns TREC_ERR_UNSUPPORTED when the device cannot hold
// one workgroup per compute unit cooperatively (the caller then runs the multi-launch step).
extern "C" int trec_fit_step_coop(float* Wu, float* Wu_m, float* Wu_v, float* Wi, float* Wi_m, float* Wi_v, float* bu, float* bu_m,
                                  float* bu_v, float* bi, float* bi_m, float* bi_v, const int64_t* f_indptr, const int32_t* f_indices,
                                  const float* f_values, const int64_t* ft_indptr, const int32_t* ft_rows, const int32_t* ft_perm,
                                  const int64_t* indptr, const int32_t* x_item, const int32_t* pos_slot, const float* pos_weight,
                                  const int32_t* samples, int64_t n_users, int64_t n_items, int64_t n_item_features, int32_t d,
                                  int32_t n_sampled, int32_t max_interactions_per_user, int64_t user_base, uint64_t seed, uint32_t step,
                                  float lr_t, float beta1, float beta2, float eps, float l2, float* workspace,
                                  int64_t workspace_floats, float* loss, float* pred_serial, void* stream)
{
    TREC_REQUIRE(Wu && Wu_m && Wu_v && Wi && Wi_m && Wi_v && f_indptr && f_indices && f_values && ft_indptr && ft_rows && ft_perm &&
                 indptr && workspace && loss && pred_serial, "trec_fit_step_coop: null pointer");
    TREC_REQUIRE((!bu) == (!bu_m) && (!bu) == (!bu_v) && (!bu) == (!bi) && (!bu) == (!bi_m) && (!bu) == (!bi_v),
                 "trec_fit_step_coop: the six bias pointers go together");
    const int64_t need = trec_fit_step_coop_workspace_floats(n_users, n_items, d, n_sampled, max_interactions_per_user);
    if (need < 0) {
        trec_set_last_error("trec_fit_step_coop: model not covered (see trec_fit_step_coop_workspace_floats)");
        return TREC_ERR_UNSUPPORTED;
    }
    TREC_REQUIRE(workspace_floats >= need, "trec_fit_step_coop: workspace too small");
    TREC_REQUIRE(max_interactions_per_user == 0 || (x_item && pos_slot), "trec_fit_step_coop: null interaction arrays");
    const int64_t ldg = (n_items + 3) / 4 * 4;
    CoopArgs a;
    a.Wu = Wu; a.Wu_m = Wu_m; a.Wu_v = Wu_v; a.Wi = Wi; a.Wi_m = Wi_m; a.Wi_v = Wi_v;
    a.bu = bu; a.bu_m = bu_m; a.bu_v = bu_v; a.bi = bi; a.bi_m = bi_m; a.bi_v = bi_v;
    a.f_indptr = f_indptr; a.f_indices = f_indices; a.f_values = f_values; a.ft_indptr = ft_indptr; a.ft_rows = ft_rows; a.ft_perm = ft_perm;
    a.indptr = indptr; a.xi = x_item; a.pos_slot = pos_slot; a.pos_weight = pos_weight; a.samples = samples;
    float* w = workspace;
    a.V = w; w += n_items * d;
    a.ib = w; w += (n_items + 3) / 4 * 4;
    a.G = w; w += n_users * ldg;
    a.dU = w; w += n_users * d;
    a.dub = w; w += (n_users + 3) / 4 * 4;
    a.dV = w; w += n_items * d;
    a.dib = w;
    a.loss = loss; a.pred_serial = pred_serial;
    a.n_users = n_users; a.n_items = n_items; a.n_item_features = n_item_features; a.ldg = ldg; a.user_base = user_base;
    a.S = n_sampled; a.d = d; a.max_rows = n_sampled + max_interactions_per_user; a.max_pos = max_interactions_per_user;
    a.sample_bits = sample_bits((int32_t)n_items);
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.step = step;
    // (the last 64 floats of the workspace are slack: eight 8-byte clock stamps fit there)
    a.clk = trec_get_tuning("coop_clocks", 0) ? (long long*)(workspace + ((need - 64 + 1) & ~(int64_t)1)) : nullptr;
    a.ratio = (float)n_items / (float)n_sampled; a.lr_t = lr_t; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.l2 = l2;
    const size_t lds = (size_t)coop_lds_bytes(n_sampled, max_interactions_per_user, d, n_users);
    int dev = 0, cus = 0, coop = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop || cus < 1) {
        trec_set_last_error("trec_fit_step_coop: the device does not support cooperative launches");
        return TREC_ERR_UNSUPPORTED;
    }
    void* args[] = {(void*)&a};
    const void* fn = (d <= 64) ? (const void*)fit_step_coop_kernel<1, 12, 16> : (const void*)fit_step_coop_kernel<1, 12, 32>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu < 1) {
        trec_set_last_error("trec_fit_step_coop: no workgroup of the step fits a compute unit");
        return TREC_ERR_UNSUPPORTED;
    }
    const int cap = trec_get_tuning("coop_wg_per_cu", 1);   // (measured at configs[1]: 2 and 4 per CU are slower -- the workgroups that wait in the barrier poll)
    if (per_cu > cap) per_cu = cap < 1 ? 1 : cap;                         // (more resident workgroups shorten every phase's rounds; the barrier grows with them)
    const hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)(cus * per_cu)), dim3(256), args, (unsigned)lds, (hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trec_set_last_error("trec_fit_step_coop: cooperative launch refused");
        return TREC_ERR_UNSUPPORTED;
    }
    return trec_check_launch("trec_fit_step_coop");
}
