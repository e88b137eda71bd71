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
#include "wmrb_tiled_body.hpp"

namespace {

template <int ITERS, int RB, int MODE, int LPR>
__global__ __launch_bounds__(256, ITERS <= 2 ? 4 : 2) void wmrb_user_tiled_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ ub, const float* __restrict__ ib,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ xi, const int32_t* __restrict__ pos_slot,
    const float* __restrict__ pos_weight, const int32_t* __restrict__ samples, int64_t n_users, int32_t S, int d, float ratio,
    int32_t max_rows, int32_t max_pos, TiledOut o)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t u = blockIdx.x;
    wmrb_tiled_user<ITERS, RB, MODE, LPR>(lds, u, U, V, ub, ib, indptr, xi, pos_slot, pos_weight, samples + u * S, S, d, ratio, max_rows,
                                     max_pos, o);
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
// 2 * (n_sampled + longest interaction row) + 2 * (longest row) + 8 * d (16 * d for d <= 64) floats within 128 KB.
extern "C" int trec_wmrb_tiled_lds_bytes(int32_t n_sampled, int32_t max_interactions_per_user, int32_t d)
{
    if (n_sampled < 1 || d < 4 || d % 4 != 0 || d > 512 || max_interactions_per_user < 0) return -1;
    const int64_t bytes = tiled_lds_floats((int64_t)n_sampled + max_interactions_per_user, max_interactions_per_user, d) * 4;
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
#define TREC_TILED(IT, RB, MD, LP)                                                                                                \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute((const void*)wmrb_user_tiled_kernel<IT, RB, MD, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)TILED_MAX_LDS);                                                                      \
        hipLaunchKernelGGL((wmrb_user_tiled_kernel<IT, RB, MD, LP>), dim3((unsigned)n_users), dim3(256), lds, st, U, V, user_bias,   \
                           item_bias, indptr, x_item, pos_slot, pos_weight, samples, n_users, n_sampled, d, ratio, max_rows,    \
                           max_interactions_per_user, o);                                                                       \
    } while (0)
    if (mode == 0) {
        if (d <= 64) TREC_TILED(1, 12, 0, 16);
        else if (d <= 128) TREC_TILED(1, 12, 0, 32);
        else if (d <= 256) TREC_TILED(2, 8, 0, 32);
        else TREC_TILED(4, 4, 0, 32);
    } else {
        if (d <= 64) TREC_TILED(1, 12, 1, 16);
        else if (d <= 128) TREC_TILED(1, 12, 1, 32);
        else if (d <= 256) TREC_TILED(2, 8, 1, 32);
        else TREC_TILED(4, 4, 1, 32);
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
