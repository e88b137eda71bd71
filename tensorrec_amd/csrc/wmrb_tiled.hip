// tensorrec_amd/csrc/wmrb_tiled.hip -- the one-pass WMRB step for users whose rows do NOT fit in registers, and for
// Euclidean scores: S in the thousands (BASELINE.json configs[4]: S = 10 % of 26,744 items), any row length LDS holds.
//
// Same mathematics as csrc/wmrb_fused.hip (tensorrec.py:384-395, :437-449; loss_graphs.py:153-227; the trainer minimises the
// SUM of the loss vector, tensorrec.py:487-489), scores per pair as the reference's SERIAL prediction graphs compute them
// (prediction_graphs.py:52-55 dot, :105-117 euclidean: -sqrt(max(sum (u - i)^2, 1e-16))):
//   pass 1  the user's row stays in the registers of all 8 subgroups; its S + n_u item rows are streamed in tiles of 8 * RB rows
//           (RB rows in flight per subgroup), each row reduced to its score, which stays in LDS (S + n_u floats);
//   loss    hinge sums / active counts per interaction and the per-sample coefficients, from LDS, as in the register kernel;
//   pass 2  the rows are streamed once more (the 27 MB table of configs[4] lives in the L2s / the Infinity Cache) and summed
//           with their coefficients into dU.
// Unfused, this step gathered every 1 KB row four times and sent scores, squared distances and coefficients through HBM in
// between.  What leaves the kernel: loss, serial predictions, dU, d b_u, and per pair the value the item side sums
// (dot: the coefficient g; euclidean: c = -g / sqrt(D), dV[i] = sum c (V[i] - U[u]) -- the convention of pair_score.hip) plus,
// for euclidean scores with biases, the raw g (d b_i = sum g).  With `dense_g` the values are also added into a zeroed dense
// [n_users, ldg] matrix: at configs[4]'s density (10 % of all cells) the item side is a GEMM, G^T . U on fp32 MFMA, not a
// sort + gather of 3.7e8 pairs.
#include "common.hpp"
#include <math.h>

namespace {

#define EUCLID_EPS 1e-16f

// sum over the 32 lanes of a subgroup (DPP adds, as in wmrb_fused.hip): total valid in lanes 16..31
__device__ __forceinline__ float dpp_add(float x, const int ctrl_tag)
{
    int v = __float_as_int(x), r;
    switch (ctrl_tag) {
        case 8: r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); break;    // row_ror:8
        case 4: r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); break;    // row_ror:4
        case 2: r = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); break;    // row_ror:2
        case 1: r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); break;    // row_ror:1
        default: r = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); break;   // row_bcast15 into rows 1, 3
    }
    return x + __int_as_float(r);
}

struct TiledOut {
    float* loss; float* pred_serial; float* dU; float* dub;
    float* val_samples; float* val_pairs;      // what the item side sums per pair (g, or c = -g / sqrt(D))
    float* raw_samples; float* raw_pairs;      // g itself (euclidean + item biases), or null
    float* dense_g; int64_t ldg;               // zeroed [n_users, ldg]: += value at (user, item), or null
    float* val_rowsum;                         // [n_users] sum of the user's values, written when dU is null (no pass 2: dU from G . V)
};

// ITERS: float4 chunks per lane of a 32-lane subgroup (d <= 128 * ITERS); RB: rows in flight per subgroup; MODE 0 dot, 1 euclid
template <int ITERS, int RB, int MODE>
__global__ __launch_bounds__(256, ITERS <= 2 ? 4 : 2) void wmrb_user_tiled_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ ub, const float* __restrict__ ib,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ xi, const int32_t* __restrict__ pos_slot,
    const float* __restrict__ pos_weight, const int32_t* __restrict__ samples, int64_t n_users, int32_t S, int d, float ratio,
    int32_t max_rows, int32_t max_pos, TiledOut o)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // y [mr4] scores | cf [mr4] squared distance, then the pair's value | bc [mp4] {1 - y_q, c_q} | partial dU [8][d] | red [8]
    const int mr4 = (max_rows + 3) & ~3, mp4 = (max_pos + 3) & ~3;
    float* l_y = lds;
    float* l_cf = l_y + mr4;
    float2* l_bc = (float2*)(l_cf + mr4);
    float* l_part = (float*)(l_bc + mp4);
    float* l_red = l_part + 8 * d;

    const int64_t u = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = tid & 31, sg = tid >> 5;
    const int64_t b = indptr[u], e = indptr[u + 1];
    const int n_pos = (int)(e - b);
    const int R = S + n_pos;

    if (n_pos == 0) {
        // no interactions: no loss terms, every coefficient is 0
        for (int s = tid; s < S; s += 256) {
            o.val_samples[u * S + s] = 0.f;
            if (o.raw_samples) o.raw_samples[u * S + s] = 0.f;
        }
        if (o.dU) for (int c = tid; c < d; c += 256) o.dU[u * d + c] = 0.f;
        if (o.val_rowsum && tid == 0) o.val_rowsum[u] = 0.f;
        if (o.dub && tid == 0) o.dub[u] = 0.f;
        return;
    }

    f32x4 x[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 32 + sub) * 4;
        const f32x4 v = *(const f32x4*)(U + u * d + (c < d ? c : 0));
        x[it] = (c < d) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float bu = ub ? ub[u] : 0.f;

    // ---- pass 1: scores of the rows j = j0 + sg + 8 r.  Every load is unconditional, from a clamped (valid) address, selected
    // afterwards (`cond ? load : 0` becomes an exec-masked block behind s_waitcnt vmcnt(0)) ----
    for (int j0 = 0; j0 < R; j0 += 8 * RB) {
        int32_t item[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int j = j0 + sg + 8 * r;
            const int q = j - S;
            const int32_t is = samples[u * S + (j < S ? j : S - 1)];
            const int32_t iq = xi[b + (q < 0 ? 0 : (q < n_pos ? q : n_pos - 1))];
            item[r] = (j < S) ? is : iq;
        }
        f32x4 y[RB][ITERS];
        float bi[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (it * 32 + sub) * 4;
                y[r][it] = *(const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0));
            }
            bi[r] = ib ? ib[item[r]] : 0.f;                   // (uniform branch; one address per subgroup)
        }
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float a = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const bool in = (it * 32 + sub) * 4 < d;      // (columns past d: both sides count as zero)
                if (MODE == 0) {
                    const f32x4 w = in ? y[r][it] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    a = fmaf(x[it].x, w.x, a); a = fmaf(x[it].y, w.y, a); a = fmaf(x[it].z, w.z, a); a = fmaf(x[it].w, w.w, a);
                } else {
                    const f32x4 w = in ? y[r][it] : x[it];
                    const float d0 = x[it].x - w.x, d1 = x[it].y - w.y, d2 = x[it].z - w.z, d3 = x[it].w - w.w;
                    a = fmaf(d0, d0, a); a = fmaf(d1, d1, a); a = fmaf(d2, d2, a); a = fmaf(d3, d3, a);
                }
            }
            acc[r] = a;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float t = acc[r];
            t = dpp_add(t, 8); t = dpp_add(t, 4); t = dpp_add(t, 2); t = dpp_add(t, 1);
            acc[r] = dpp_add(t, 0);
        }
        if (sub == 31) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int j = j0 + sg + 8 * r;
                if (j < R) {
                    float s = acc[r];
                    if (MODE == 1) { l_cf[j] = s; s = -1.0f * sqrtf(fmaxf(s, EUCLID_EPS)); }
                    if (ub) s = s + bu;
                    if (ib) s = s + bi[r];
                    l_y[j] = s;
                    if (j >= S) o.pred_serial[b + (j - S)] = s;
                }
            }
        }
    }
    __syncthreads();

    // ---- loss terms (loss_graphs.py:153-180): 8 threads per interaction walk the S sample scores; lane k == 0 of the eight
    // then holds hinge sum and active count (fixed combination order), and finishes the interaction ----
    float raw_sum = 0.f;                                     // this thread's share of d b_u = sum of every pair's g
    float val_sum = 0.f;                                     // ... and of the sum of the values (euclidean: dU = val_sum U - G . V)
    for (int q0 = 0; q0 < n_pos; q0 += 32) {
        const int q = q0 + (tid >> 3), k = tid & 7;
        const bool live = q < n_pos;
        const float yq = l_y[S + (live ? q : 0)];
        const float base = 1.0f - yq;
        float acc = 0.f;
        int cnt = 0;
        if (live) {
            for (int s4 = k * 4; s4 < S; s4 += 32) {
                const f32x4 v = *(const f32x4*)(l_y + s4);            // (l_y is padded to a multiple of 4: entries >= S masked)
                const float t0 = base + v.x, t1 = base + v.y, t2 = base + v.z, t3 = base + v.w;
                const bool m1 = s4 + 1 < S, m2 = s4 + 2 < S, m3 = s4 + 3 < S;
                acc += fmaxf(t0, 0.f); cnt += (t0 >= 0.f) ? 1 : 0;
                if (m1) { acc += fmaxf(t1, 0.f); cnt += (t1 >= 0.f) ? 1 : 0; }
                if (m2) { acc += fmaxf(t2, 0.f); cnt += (t2 >= 0.f) ? 1 : 0; }
                if (m3) { acc += fmaxf(t3, 0.f); cnt += (t3 >= 0.f) ? 1 : 0; }
            }
        }
        float fc = (float)cnt;
        acc += __shfl_xor(acc, 1, 64); fc += __shfl_xor(fc, 1, 64);
        acc += __shfl_xor(acc, 2, 64); fc += __shfl_xor(fc, 2, 64);
        acc += __shfl_xor(acc, 4, 64); fc += __shfl_xor(fc, 4, 64);
        if (live && k == 0) {
            const int32_t slot = pos_slot[b + q];
            float c = 0.f, dp = 0.f;
            if (slot >= 0) {
                const float w = pos_weight ? pos_weight[b + q] : 1.f;
                float smr = ratio * acc;
                if (pos_weight) smr = smr * w;
                c = ratio / (1.0f + smr);                               // d loss_p / d (hinge sum), upstream gradient 1
                if (pos_weight) c = c * w;
                o.loss[slot] = logf(smr + 1.0f);
                dp = -c * fc;
            }
            l_bc[q] = make_float2(base, c);
            float val = dp;
            if (MODE == 1) { const float D = l_cf[S + q]; val = (D >= EUCLID_EPS) ? -dp / sqrtf(D) : 0.f; }
            l_cf[S + q] = val;
            o.val_pairs[b + q] = val;
            if (o.raw_pairs) o.raw_pairs[b + q] = dp;
            if (o.dense_g) unsafeAtomicAdd(o.dense_g + u * o.ldg + xi[b + q], val);
            raw_sum += dp;
            val_sum += val;
        }
    }
    __syncthreads();
    // per sample: its coefficient over the user's interactions (c_q = 0 for non-positive ones); four samples per thread share
    // every {1 - y_q, c_q} read
    for (int s0 = tid; s0 < S; s0 += 1024) {
        float ys[4], g[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int s = s0 + 256 * m;
            ys[m] = l_y[s < S ? s : S - 1];
            g[m] = 0.f;
        }
        for (int q = 0; q < n_pos; ++q) {
            const float2 bc = l_bc[q];
#pragma unroll
            for (int m = 0; m < 4; ++m) g[m] += (bc.x + ys[m] >= 0.f) ? bc.y : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int s = s0 + 256 * m;
            if (s < S) {
                float val = g[m];
                if (MODE == 1) { const float D = l_cf[s]; val = (D >= EUCLID_EPS) ? -g[m] / sqrtf(D) : 0.f; }
                l_cf[s] = val;
                o.val_samples[u * S + s] = val;
                if (o.raw_samples) o.raw_samples[u * S + s] = g[m];
                if (o.dense_g) unsafeAtomicAdd(o.dense_g + u * o.ldg + samples[u * S + s], val);
                raw_sum += g[m];
                val_sum += val;
            }
        }
    }
    __syncthreads();

    // ---- pass 2: dU_u = sum_j val_j * row_j (dot) / sum_j val_j * (U_u - row_j) (euclidean); skipped when the caller takes dU
    // from the dense matrix (G . V on fp32 MFMA: the second sweep over the rows is 45 ms of configs[4]'s step, the GEMM 19) ----
    if (o.dU) {
        f32x4 part[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) part[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < R; j0 += 8 * RB) {
            int32_t item[RB];
            float cf[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int j = j0 + sg + 8 * r;
                const int q = j - S;
                const int32_t is = samples[u * S + (j < S ? j : S - 1)];
                const int32_t iq = xi[b + (q < 0 ? 0 : (q < n_pos ? q : n_pos - 1))];
                item[r] = (j < S) ? is : iq;
                const float v = l_cf[j < R ? j : 0];
                cf[r] = (j < R) ? v : 0.f;
            }
            f32x4 y[RB][ITERS];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int c = (it * 32 + sub) * 4;
                    y[r][it] = *(const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0));
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    f32x4 w = y[r][it];
                    if (MODE == 1) { w.x = x[it].x - w.x; w.y = x[it].y - w.y; w.z = x[it].z - w.z; w.w = x[it].w - w.w; }
                    part[it].x = fmaf(cf[r], w.x, part[it].x); part[it].y = fmaf(cf[r], w.y, part[it].y);
                    part[it].z = fmaf(cf[r], w.z, part[it].z); part[it].w = fmaf(cf[r], w.w, part[it].w);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 32 + sub) * 4;
            if (c < d) *(f32x4*)(l_part + sg * d + c) = part[it];
        }
    }
    if (o.dub) {
        for (int off = 32; off > 0; off >>= 1) raw_sum += __shfl_xor(raw_sum, off, 64);
        if (lane == 0) l_red[wave] = raw_sum;
    }
    if (o.val_rowsum) {
        for (int off = 32; off > 0; off >>= 1) val_sum += __shfl_xor(val_sum, off, 64);
        if (lane == 0) l_red[4 + wave] = val_sum;
    }
    __syncthreads();
    if (o.dU) {
        for (int c = tid; c < d; c += 256) {
            float acc = l_part[c];
#pragma unroll
            for (int g8 = 1; g8 < 8; ++g8) acc += l_part[g8 * d + c];
            o.dU[u * d + c] = acc;
        }
    }
    if (o.dub && tid == 0) o.dub[u] = (l_red[0] + l_red[1]) + (l_red[2] + l_red[3]);
    if (o.val_rowsum && tid == 0) o.val_rowsum[u] = (l_red[4] + l_red[5]) + (l_red[6] + l_red[7]);
}

// d b_i = sum of g over the pairs of item i, for two pair lists (sampled pairs, interactions): a weighted histogram.  The item
// range fits LDS (n_items <= 32,768): every workgroup bins its slice of the pairs in LDS and adds its bins to `out` once.
__global__ __launch_bounds__(1024) void item_hist_lds_kernel(const int32_t* __restrict__ ids_a, const float* __restrict__ val_a,
                                                            int64_t n_a, const int32_t* __restrict__ ids_b,
                                                            const float* __restrict__ val_b, int64_t n_b, int32_t n_items,
                                                            float* __restrict__ out)
{
    extern __shared__ float bins[];
    for (int i = threadIdx.x; i < n_items; i += blockDim.x) bins[i] = 0.f;
    __syncthreads();
    const int64_t n = n_a + n_b;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < n ? p0 + per : n;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const bool a = p < n_a;
        const int32_t id = a ? ids_a[p] : ids_b[p - n_a];
        const float v = a ? val_a[p] : val_b[p - n_a];
        if (id >= 0 && id < n_items && v != 0.f) atomicAdd(bins + id, v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
        const float v = bins[i];
        if (v != 0.f) unsafeAtomicAdd(out + i, v);
    }
}

__global__ __launch_bounds__(256) void item_hist_global_kernel(const int32_t* __restrict__ ids_a, const float* __restrict__ val_a,
                                                              int64_t n_a, const int32_t* __restrict__ ids_b,
                                                              const float* __restrict__ val_b, int64_t n_b, int32_t n_items,
                                                              float* __restrict__ out)
{
    const int64_t n = n_a + n_b;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const bool a = p < n_a;
        const int32_t id = a ? ids_a[p] : ids_b[p - n_a];
        const float v = a ? val_a[p] : val_b[p - n_a];
        if (id >= 0 && id < n_items && v != 0.f) unsafeAtomicAdd(out + id, v);
    }
}

constexpr int64_t TILED_MAX_LDS = 128 * 1024;

}  // namespace

// Dynamic LDS of trec_wmrb_tiled_step, or -1 when the configuration is not covered: d % 4 == 0, d <= 512, n_sampled >= 1, and
// 2 * (n_sampled + longest interaction row) + 2 * (longest row) + 8 * d floats within 128 KB.
extern "C" int trec_wmrb_tiled_lds_bytes(int32_t n_sampled, int32_t max_interactions_per_user, int32_t d)
{
    if (n_sampled < 1 || d < 4 || d % 4 != 0 || d > 512 || max_interactions_per_user < 0) return -1;
    const int64_t mr4 = ((int64_t)n_sampled + max_interactions_per_user + 3) & ~(int64_t)3;
    const int64_t mp4 = ((int64_t)max_interactions_per_user + 3) & ~(int64_t)3;
    const int64_t bytes = (2 * mr4 + 2 * mp4 + 8 * (int64_t)d + 8) * 4;
    return bytes <= TILED_MAX_LDS ? (int)bytes : -1;
}

extern "C" int trec_wmrb_tiled_step(const float* U, const float* V, const float* user_bias, const float* item_bias,
                                    const int64_t* indptr, const int32_t* x_item, const int32_t* pos_slot,
                                    const float* pos_weight, const int32_t* samples, int64_t n_users, int64_t n_items,
                                    int32_t n_sampled, int32_t d, int32_t mode, int32_t max_interactions_per_user, float* loss,
                                    float* pred_serial, float* dU, float* d_user_bias, float* val_samples, float* val_pairs,
                                    float* raw_samples, float* raw_pairs, float* dense_g, int64_t ldg, float* val_rowsum,
                                    void* stream)
{
    TREC_REQUIRE(U && V && indptr && samples && loss && pred_serial && val_samples && val_pairs,
                 "trec_wmrb_tiled_step: null pointer");
    TREC_REQUIRE(dU || (dense_g && val_rowsum), "trec_wmrb_tiled_step: without dU the caller needs dense_g and val_rowsum");
    TREC_REQUIRE(mode == 0 || mode == 1, "trec_wmrb_tiled_step: mode must be 0 (dot) or 1 (euclidean)");
    TREC_REQUIRE(!user_bias == !d_user_bias, "trec_wmrb_tiled_step: user_bias and d_user_bias go together");
    TREC_REQUIRE(!raw_samples == !raw_pairs, "trec_wmrb_tiled_step: raw_samples and raw_pairs go together");
    TREC_REQUIRE(!dense_g || ldg >= n_items, "trec_wmrb_tiled_step: ldg must cover the items");
    const int lds = trec_wmrb_tiled_lds_bytes(n_sampled, max_interactions_per_user, d);
    if (lds < 0) {
        trec_set_last_error("trec_wmrb_tiled_step: configuration not covered (see trec_wmrb_tiled_lds_bytes)");
        return TREC_ERR_UNSUPPORTED;
    }
    if (n_users == 0) return TREC_OK;
    TREC_REQUIRE(max_interactions_per_user == 0 || (x_item && pos_slot), "trec_wmrb_tiled_step: null interaction arrays");
    const float ratio = (float)n_items / (float)n_sampled;
    const int32_t max_rows = n_sampled + max_interactions_per_user;
    hipStream_t st = (hipStream_t)stream;
    TiledOut o = {loss, pred_serial, dU, d_user_bias, val_samples, val_pairs, raw_samples, raw_pairs, dense_g, ldg, val_rowsum};
#define TREC_TILED(IT, RB, MD)                                                                                                  \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute((const void*)wmrb_user_tiled_kernel<IT, RB, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)TILED_MAX_LDS);                                                                      \
        hipLaunchKernelGGL((wmrb_user_tiled_kernel<IT, RB, MD>), dim3((unsigned)n_users), dim3(256), lds, st, U, V, user_bias,   \
                           item_bias, indptr, x_item, pos_slot, pos_weight, samples, n_users, n_sampled, d, ratio, max_rows,    \
                           max_interactions_per_user, o);                                                                       \
    } while (0)
    if (mode == 0) {
        if (d <= 128) TREC_TILED(1, 12, 0);
        else if (d <= 256) TREC_TILED(2, 8, 0);
        else TREC_TILED(4, 4, 0);
    } else {
        if (d <= 128) TREC_TILED(1, 12, 1);
        else if (d <= 256) TREC_TILED(2, 8, 1);
        else TREC_TILED(4, 4, 1);
    }
#undef TREC_TILED
    return trec_check_launch("trec_wmrb_tiled_step");
}

// out[i] += sum of val over the pairs (two lists, either may be empty) whose id is i; ids outside [0, n_items) are skipped.
// `out` is NOT cleared here.  The sums of an item are added in arrival order (not bit-reproducible run to run).
extern "C" int trec_item_weighted_hist(const int32_t* ids_a, const float* val_a, int64_t n_a, const int32_t* ids_b,
                                       const float* val_b, int64_t n_b, int32_t n_items, float* out, void* stream)
{
    TREC_REQUIRE(out && n_items >= 1 && n_a >= 0 && n_b >= 0, "trec_item_weighted_hist: bad arguments");
    TREC_REQUIRE((n_a == 0 || (ids_a && val_a)) && (n_b == 0 || (ids_b && val_b)), "trec_item_weighted_hist: null list");
    if (n_a + n_b == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (n_items <= 32768) {
        const int lds = n_items * 4;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)item_hist_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        const int64_t want = (n_a + n_b + 65535) / 65536;
        const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
        hipLaunchKernelGGL(item_hist_lds_kernel, dim3(grid), dim3(1024), lds, st, ids_a, val_a, n_a, ids_b, val_b, n_b, n_items, out);
    } else {
        const int64_t want = (n_a + n_b + 1023) / 1024;
        const unsigned grid = (unsigned)(want > 4096 ? 4096 : want);
        hipLaunchKernelGGL(item_hist_global_kernel, dim3(grid), dim3(256), 0, st, ids_a, val_a, n_a, ids_b, val_b, n_b, n_items, out);
    }
    return trec_check_launch("trec_item_weighted_hist");
}
