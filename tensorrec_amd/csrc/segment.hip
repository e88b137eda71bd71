// tensorrec_amd/csrc/segment.hip -- group (user, item) pairs by item on the device: a counting sort that yields the
// transposed (CSC-like) structure of a pair list, so that the item-side gradient of the sampled serial predictions
//     dV[i] = sum over pairs p with item(p) = i of  g[p] * U[user(p)]
// (the scatter-add TF's autodiff emits for tf.gather, tensorrec/prediction_graphs.py:53-54 applied to the U*S sampled
// pairs of tensorrec/tensorrec.py:390-395) becomes the K1 segmented gather instead of n_pairs * d fp32 atomics.
//
// Pairs with a NEGATIVE item key belong to no bucket and are skipped (the two-stage top-k drops pruned (user, superblock)
// pairs this way: routing millions of them into one dummy bucket would serialise on a single atomic counter).
//
// Three passes: histogram of items (int32 atomics, one per pair), exclusive scan (two-level, 1024 items per block),
// fill (one int32 atomic per pair for the slot inside its bucket).  The order of pairs INSIDE a bucket depends on
// atomic arrival order, so the fp32 sums downstream are reproducible only up to summation order (as with atomics).
#include "common.hpp"

__global__ __launch_bounds__(256) void seg_hist_kernel(const int32_t* __restrict__ xi, int64_t n_pairs,
                                                      int32_t* __restrict__ counts)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * 256) {
        const int32_t i = xi[p];
        if (i >= 0) atomicAdd(counts + i, 1);            // negative key = "not in any bucket" (dropped pair)
    }
}

// block b scans counts[b*1024 .. +1024) -> local exclusive prefix (int64) into indptr, block total into block_sum[b]
__global__ __launch_bounds__(256) void seg_scan_local_kernel(const int32_t* __restrict__ counts, int64_t n,
                                                            int64_t* __restrict__ indptr, int64_t* __restrict__ block_sum)
{
    __shared__ int64_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int64_t v[4];
    int64_t run = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = run; run += (base + e < n) ? counts[base + e] : 0; }
    // inclusive scan of `run` across the wave, then across the 4 waves
    int64_t inc = run;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t excl = woff + inc - run;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (base + e < n) indptr[base + e] = excl + v[e];
    if (threadIdx.x == 255) block_sum[blockIdx.x] = woff + inc;
}

// single block: exclusive scan of the block sums in place (n_blocks is small: n / 1024)
__global__ __launch_bounds__(256) void seg_scan_blocks_kernel(int64_t* __restrict__ block_sum, int n_blocks,
                                                             int64_t* __restrict__ total_out)
{
    __shared__ int64_t carry_s;
    __shared__ int64_t wsum[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < n_blocks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int64_t x = (i < n_blocks) ? block_sum[i] : 0;
        int64_t inc = x;
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t woff = carry_s;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < n_blocks) block_sum[i] = woff + inc - x;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(256) void seg_add_offsets_kernel(int64_t* __restrict__ indptr, int64_t n,
                                                             const int64_t* __restrict__ block_sum,
                                                             const int64_t* __restrict__ total)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) indptr[i] += block_sum[i >> 10];
    if (i == n) indptr[n] = *total;
}

__global__ __launch_bounds__(256) void seg_fill_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                      int64_t n_pairs, int32_t pairs_per_user,
                                                      const int64_t* __restrict__ indptr, int32_t* __restrict__ cursor,
                                                      int32_t* __restrict__ users_t, int32_t* __restrict__ perm_t)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * 256) {
        const int32_t i = xi[p];
        if (i < 0) continue;
        const int64_t slot = indptr[i] + atomicAdd(cursor + i, 1);
        users_t[slot] = xu ? xu[p] : (int32_t)(p / pairs_per_user);
        perm_t[slot] = (int32_t)p;
    }
}

// ---- few buckets (<= SEG_SMALL_MAX): privatised counters ------------------------------------------------------------
// The two-stage top-k groups ~10 M (user, superblock) pairs into a few hundred to a few thousand superblocks: global
// atomics would queue thousands deep on each counter.  Every workgroup takes a contiguous run of SEG_SMALL_RUN pairs,
// counts it in LDS, and touches each global counter once (histogram: add its count; fill: reserve its range), so the
// global traffic is (runs x buckets) instead of one contended atomic per pair.
#define SEG_SMALL_MAX 4096
#define SEG_SMALL_RUN 8192

__global__ __launch_bounds__(256) void seg_hist_small_kernel(const int32_t* __restrict__ xi, int64_t n_pairs,
                                                            int32_t n_items, int32_t* __restrict__ counts)
{
    __shared__ int32_t l_cnt[SEG_SMALL_MAX];
    for (int i = threadIdx.x; i < n_items; i += 256) l_cnt[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * SEG_SMALL_RUN;
    const int64_t p1 = (p0 + SEG_SMALL_RUN < n_pairs) ? p0 + SEG_SMALL_RUN : n_pairs;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int32_t i = xi[p];
        if (i >= 0) atomicAdd(l_cnt + i, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += 256)
        if (l_cnt[i]) atomicAdd(counts + i, l_cnt[i]);
}

__global__ __launch_bounds__(256) void seg_fill_small_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                            int64_t n_pairs, int32_t pairs_per_user, int32_t n_items,
                                                            const int64_t* __restrict__ indptr, int32_t* __restrict__ cursor,
                                                            int32_t* __restrict__ users_t, int32_t* __restrict__ perm_t)
{
    __shared__ int32_t l_cnt[SEG_SMALL_MAX];        // pass 1: count of the run per bucket; pass 2: next local rank
    __shared__ int32_t l_base[SEG_SMALL_MAX];       // start of this run's range inside the bucket
    for (int i = threadIdx.x; i < n_items; i += 256) l_cnt[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * SEG_SMALL_RUN;
    const int64_t p1 = (p0 + SEG_SMALL_RUN < n_pairs) ? p0 + SEG_SMALL_RUN : n_pairs;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int32_t i = xi[p];
        if (i >= 0) atomicAdd(l_cnt + i, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_items; i += 256) {
        const int32_t c = l_cnt[i];
        l_base[i] = c ? atomicAdd(cursor + i, c) : 0;
        l_cnt[i] = 0;
    }
    __syncthreads();
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int32_t i = xi[p];
        if (i < 0) continue;
        const int64_t slot = indptr[i] + l_base[i] + atomicAdd(l_cnt + i, 1);
        users_t[slot] = xu ? xu[p] : (int32_t)(p / pairs_per_user);
        perm_t[slot] = (int32_t)p;
    }
}

// atomic-free fill: the rank of every pair inside its bucket is already known (returned by the histogram atomics)
// values_in != NULL: the pairs' values travel with them (values_out[slot] = values_in[p]), so the gather that follows reads
// them in order instead of through perm_t (one random 4-byte read per pair less); with values_out == NULL they are packed
// next to the user id -- users_t is then an int2[n_pairs] buffer and every pair costs ONE 8-byte scattered store
__global__ __launch_bounds__(256) void seg_fill_ranked_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                             const int32_t* __restrict__ ranks, int64_t n_pairs,
                                                             int32_t pairs_per_user, const int64_t* __restrict__ indptr,
                                                             int32_t* __restrict__ users_t, int32_t* __restrict__ perm_t,
                                                             const float* __restrict__ values_in,
                                                             float* __restrict__ values_out)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * 256) {
        if (xi[p] < 0) continue;
        const int64_t slot = indptr[xi[p]] + ranks[p];
        const int32_t usr = xu ? xu[p] : (int32_t)(p / pairs_per_user);
        if (values_in && !values_out) {          // packed: ONE 8-byte scattered store {user, value bits} into users_t
            ((int2*)users_t)[slot] = make_int2(usr, __float_as_int(values_in[p]));
            continue;
        }
        users_t[slot] = usr;
        if (perm_t) perm_t[slot] = (int32_t)p;
        if (values_in) values_out[slot] = values_in[p];
    }
}

// workspace_i32: 2 * n_items int32 (counts, cursors); workspace_i64: ceil(n_items/1024) + 1 int64
// counts_given != 0: workspace_i32[0 .. n_items) already holds the histogram of xi (e.g. counted by the kernel that
// consumed the pairs, trec_wmrb_fused_step) -- the histogram pass is skipped; ranks (nullable, needs counts_given):
// the value each of those histogram atomics returned = the pair's position inside its bucket -- the fill is atomic-free
extern "C" int trec_group_pairs_by_item(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user,
                                        int64_t n_items, int32_t* workspace_i32, int64_t* workspace_i64,
                                        int64_t* indptr_t, int32_t* users_t, int32_t* perm_t, int32_t counts_given,
                                        const int32_t* ranks, const float* values_in, float* values_out, void* stream)
{
    TREC_REQUIRE(xi && workspace_i32 && workspace_i64 && indptr_t && users_t, "trec_group_pairs_by_item: null pointer");
    TREC_REQUIRE(perm_t || (ranks && values_in), "trec_group_pairs_by_item: perm_t may be NULL only when values travel (ranked fill)");
    TREC_REQUIRE(!values_in || ranks, "trec_group_pairs_by_item: values travel only with ranks");
    TREC_REQUIRE(values_in || !values_out, "trec_group_pairs_by_item: values_out without values_in");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_group_pairs_by_item: need xu or pairs_per_user");
    TREC_REQUIRE(n_pairs < ((int64_t)1 << 31) && n_items >= 1, "trec_group_pairs_by_item: n_pairs must fit int32");
    TREC_REQUIRE(!ranks || counts_given, "trec_group_pairs_by_item: ranks come with counts_given");
    hipStream_t st = (hipStream_t)stream;
    int32_t* counts = workspace_i32;
    int32_t* cursor = workspace_i32 + n_items;
    const int n_blocks = (int)ceil_div64(n_items, 1024);
    int64_t* block_sum = workspace_i64;
    int64_t* total = workspace_i64 + n_blocks;
    int32_t* zero_from = counts_given ? cursor : workspace_i32;
    if (hipMemsetAsync(zero_from, 0, sizeof(int32_t) * (counts_given ? 1 : 2) * (size_t)n_items, st) != hipSuccess) {
        trec_set_last_error("trec_group_pairs_by_item: memset failed");
        return TREC_ERR_LAUNCH;
    }
    int64_t gb = ceil_div64(n_pairs, 256);
    if (gb > 8192) gb = 8192;
    if (gb < 1) gb = 1;
    const bool small = n_items <= SEG_SMALL_MAX && n_pairs >= 4 * SEG_SMALL_RUN;
    const unsigned runs = (unsigned)ceil_div64(n_pairs, SEG_SMALL_RUN);
    if (!counts_given) {
        if (small) hipLaunchKernelGGL(seg_hist_small_kernel, dim3(runs), dim3(256), 0, st, xi, n_pairs, (int32_t)n_items, counts);
        else hipLaunchKernelGGL(seg_hist_kernel, dim3((unsigned)gb), dim3(256), 0, st, xi, n_pairs, counts);
    }
    hipLaunchKernelGGL(seg_scan_local_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, counts, n_items, indptr_t, block_sum);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(256), 0, st, block_sum, n_blocks, total);
    hipLaunchKernelGGL(seg_add_offsets_kernel, dim3((unsigned)ceil_div64(n_items + 1, 256)), dim3(256), 0, st, indptr_t, n_items, block_sum, total);
    if (small && !ranks)
        hipLaunchKernelGGL(seg_fill_small_kernel, dim3(runs), dim3(256), 0, st, xu, xi, n_pairs, pairs_per_user, (int32_t)n_items, indptr_t, cursor, users_t, perm_t);
    else if (ranks)
        hipLaunchKernelGGL(seg_fill_ranked_kernel, dim3((unsigned)gb), dim3(256), 0, st, xu, xi, ranks, n_pairs, pairs_per_user, indptr_t, users_t, perm_t, values_in, values_out);
    else
        hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)gb), dim3(256), 0, st, xu, xi, n_pairs, pairs_per_user, indptr_t, cursor, users_t, perm_t);
    return trec_check_launch("trec_group_pairs_by_item");
}

// ---- the ranked, packed fill in two levels (the data-parallel fit's 1e8 sampled pairs) ----------------------------------------
// seg_fill_ranked_kernel stores 8 bytes per pair at indptr[item] + rank: 1e8 scattered stores, each the only one its 64-byte
// sector sees before it leaves L2 -- ~6.4 GB written for 0.8 GB of entries, 3.7 ms of a 26 ms epoch.  Here the destination is cut
// into WINDOWS of 2^L entries.  Pass A (seg_stage_kernel): a workgroup takes 8,192 consecutive pairs, computes every pair's slot,
// counts its pairs per window in LDS, reserves room in the windows' staging regions with ONE global atomic per (workgroup,
// window) and appends {slot} and {user, value} there -- a window's region is as large as the window (slots are unique), so no
// histogram pass is needed.  Pass B (seg_place_kernel): the workgroups of a window read its region in order and store the
// entries at their slots; the workgroups of a window share an XCD (workgroup id mod 8).
// Measured at 1e8 pairs (fill + scan, HIP events): L = 17 (763 windows of 1 MB, the L2-sized design) 6.0 ms -- pass A 4.5 ms:
// its runs are ~10 pairs per window, i.e. the partial-sector stores it was meant to remove, twice; L = 19 4.4, 21 3.5, **22
// 3.1** (24 windows of 32 MB: pass A's runs are 340 pairs, pass B's windows live in the Infinity Cache), 23 3.5, 24 3.7 = the
// one-level fill.  A pass A that sorts its tile in LDS and writes whole sectors is what the 1 MB windows would need; not built.
constexpr int STG_PT = 32;                 // pairs per thread of pass A (8,192 per workgroup)
constexpr int STG_MAX_WINDOWS = 2048;      // LDS counters of pass A
constexpr int STG_CURSOR_STRIDE = 32;      // one global cursor per 128-byte line: every workgroup of pass A adds to every window's cursor,
                                           // and atomics on one line serialise (measured: no difference at 763 windows -- the stores dominate -- kept)

__global__ __launch_bounds__(256, 4) void seg_stage_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                       const int32_t* __restrict__ ranks, const float* __restrict__ values_in,
                                                       int64_t n_pairs, int32_t pairs_per_user, const int64_t* __restrict__ indptr,
                                                       int32_t window_log2, int32_t n_windows, int32_t* __restrict__ cursor,
                                                       int32_t* __restrict__ keys, int2* __restrict__ payload)
{
    __shared__ int hist[STG_MAX_WINDOWS];
    for (int i = threadIdx.x; i < n_windows; i += 256) hist[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * (256 * STG_PT) + threadIdx.x;
    int32_t slot[STG_PT];
#pragma unroll
    for (int k0 = 0; k0 < STG_PT; k0 += 4) {               // four pairs at a time: ids and ranks, then the bucket starts (all unconditional)
        int32_t it[4], rk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t p = p0 + (int64_t)(k0 + q) * 256;
            const int64_t pc = p < n_pairs ? p : n_pairs - 1;
            it[q] = xi[pc];
            rk[q] = ranks[pc];
        }
        int64_t base[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[q] = indptr[it[q] < 0 ? 0 : it[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t p = p0 + (int64_t)(k0 + q) * 256;
            const bool ok = p < n_pairs && it[q] >= 0;
            const int32_t sl = ok ? (int32_t)(base[q] + rk[q]) : -1;
            slot[k0 + q] = sl;
            if (ok) atomicAdd(&hist[sl >> window_log2], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_windows; i += 256) {
        const int c = hist[i];
        hist[i] = c ? atomicAdd(cursor + (int64_t)i * STG_CURSOR_STRIDE, c) : 0;     // this workgroup's first position in window i's region: an LDS cursor from here on
    }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < STG_PT; k0 += 4) {               // (values and users of four pairs together, unconditionally)
        float vl[4];
        int32_t us[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t p = p0 + (int64_t)(k0 + q) * 256;
            const int64_t pc = p < n_pairs ? p : n_pairs - 1;
            vl[q] = values_in[pc];
            us[q] = xu ? xu[pc] : (int32_t)(pc / pairs_per_user);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int32_t sl = slot[k0 + q];
            if (sl < 0) continue;
            const int w = sl >> window_log2;
            const int64_t pos = ((int64_t)w << window_log2) + atomicAdd(&hist[w], 1);
            keys[pos] = sl;
            payload[pos] = make_int2(us[q], __float_as_int(vl[q]));
        }
    }
}

__global__ __launch_bounds__(256) void seg_place_kernel(const int32_t* __restrict__ keys, const int2* __restrict__ payload,
                                                       const int32_t* __restrict__ cursor, int32_t window_log2, int32_t n_windows,
                                                       int32_t wgs_per_window, int2* __restrict__ entries)
{
    const int x = blockIdx.x & 7, t = blockIdx.x >> 3;      // XCD x takes the windows w = x (mod 8), in order
    const int w = (t / wgs_per_window) * 8 + x, j = t % wgs_per_window;
    if (w >= n_windows) return;
    const int n = cursor[(int64_t)w * STG_CURSOR_STRIDE];
    const int64_t base = (int64_t)w << window_log2;
    for (int i0 = j * 1024 + (int)threadIdx.x; i0 < n; i0 += wgs_per_window * 1024) {
        int32_t key[4];
        int2 pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 256 * q;
            const int ic = i < n ? i : n - 1;
            key[q] = keys[base + ic];
            pl[q] = payload[base + ic];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i0 + 256 * q < n) entries[key[q]] = pl[q];
    }
}

// staging bytes of trec_group_pairs_by_item_staged for n_pairs pairs and windows of 2^window_log2 entries, or 0 when that form
// does not cover the size (more than 2,048 windows, fewer than two)
extern "C" int64_t trec_group_pairs_staged_bytes(int64_t n_pairs, int32_t window_log2)
{
    if (n_pairs < 2 || window_log2 < 8 || window_log2 > 24) return 0;
    const int64_t n_windows = ceil_div64(n_pairs, (int64_t)1 << window_log2);
    if (n_windows < 2 || n_windows > STG_MAX_WINDOWS) return 0;
    return (n_windows << window_log2) * 12 + n_windows * STG_CURSOR_STRIDE * 4;
}

// trec_group_pairs_by_item with counts_given = 1, ranks and values (entries int2 [n_pairs] = {user, value bits}), the fill done in
// two levels through ``staging`` (trec_group_pairs_staged_bytes).  workspace_i32 [2 * n_items]: its first half holds the histogram.
extern "C" int trec_group_pairs_by_item_staged(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user,
                                               int64_t n_items, int32_t* workspace_i32, int64_t* workspace_i64, int64_t* indptr_t,
                                               int32_t* entries, const int32_t* ranks, const float* values_in, void* staging,
                                               int64_t staging_bytes, int32_t window_log2, void* stream)
{
    TREC_REQUIRE(xi && workspace_i32 && workspace_i64 && indptr_t && entries && ranks && values_in && staging,
                 "trec_group_pairs_by_item_staged: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_group_pairs_by_item_staged: need xu or pairs_per_user");
    TREC_REQUIRE(n_pairs < ((int64_t)1 << 31) && n_items >= 1, "trec_group_pairs_by_item_staged: n_pairs must fit int32");
    const int64_t need = trec_group_pairs_staged_bytes(n_pairs, window_log2);
    TREC_REQUIRE(need > 0 && staging_bytes >= need, "trec_group_pairs_by_item_staged: size not covered / staging too small");
    hipStream_t st = (hipStream_t)stream;
    const int32_t n_windows = (int32_t)ceil_div64(n_pairs, (int64_t)1 << window_log2);
    int32_t* keys = (int32_t*)staging;
    int2* payload = (int2*)(keys + ((int64_t)n_windows << window_log2));
    int32_t* cursor = (int32_t*)(payload + ((int64_t)n_windows << window_log2));
    const int n_blocks = (int)ceil_div64(n_items, 1024);
    int64_t* block_sum = workspace_i64;
    int64_t* total = workspace_i64 + n_blocks;
    if (hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n_windows * STG_CURSOR_STRIDE, st) != hipSuccess) {
        trec_set_last_error("trec_group_pairs_by_item_staged: memset failed");
        return TREC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(seg_scan_local_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, workspace_i32, n_items, indptr_t, block_sum);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(256), 0, st, block_sum, n_blocks, total);
    hipLaunchKernelGGL(seg_add_offsets_kernel, dim3((unsigned)ceil_div64(n_items + 1, 256)), dim3(256), 0, st, indptr_t, n_items, block_sum, total);
    hipLaunchKernelGGL(seg_stage_kernel, dim3((unsigned)ceil_div64(n_pairs, 256 * STG_PT)), dim3(256), 0, st, xu, xi, ranks, values_in,
                       n_pairs, pairs_per_user, indptr_t, window_log2, n_windows, cursor, keys, payload);
    int32_t wgs = (int32_t)(((int64_t)1 << window_log2) / 2048);
    if (wgs < 1) wgs = 1;
    const unsigned blocks = (unsigned)(((n_windows + 7) / 8) * 8 * wgs);
    hipLaunchKernelGGL(seg_place_kernel, dim3(blocks), dim3(256), 0, st, keys, payload, cursor, window_log2, n_windows, wgs, (int2*)entries);
    return trec_check_launch("trec_group_pairs_by_item_staged");
}

// ---- grouping WITHOUT ranks: a two-level partition through LDS (round 5) ----------------------------------------------------------
// The ranked forms above need, per pair, the rank its histogram atomic returned: 1e8 global atomics inside the fused WMRB kernel
// (1.3 ms of its 11.4) before a fill of 3.0 ms.  The sampled pairs are uniform over the items BY CONSTRUCTION (the sampler draws
// uniformly), so a fixed partition is balanced:
//   bins of 4,096 consecutive items (245 at 1M items).
//   (1) seg_bin_count_kernel: pairs per bin (LDS counters per tile, one global atomic per tile and bin); a one-workgroup scan gives
//       the bins' bases.
//   (2) seg_bin_partition_kernel: a workgroup sorts a TILE of 8,192 pairs by bin IN LDS (counting sort: LDS atomics give the local
//       ranks, a scan the bin starts), reserves its run in every bin's region with one global atomic per bin and writes the records
//       {item, user, value} bin after bin, dword by dword: consecutive lanes write consecutive addresses, whole sectors (a run is
//       ~33 records of 12 B at 1M items; scattered per-record stores at that run length were what made 1 MB windows slow above).
//   (3) per bin: its records are counted per item in LDS (4,096 counters), the scan of the counts IS indptr for its items -- no
//       global histogram, no global scan --, and {user, value} go to base + (LDS cursor of the item)++: 8-byte stores scattered
//       inside the bin's own ~3 MB of output (three launches, BIN_SLICES workgroups per bin: see below).
// No global atomic per pair anywhere.  The order of pairs inside an item's bucket follows atomic arrival, as in every other form.
// Measured at 1e8 pairs over 1M items (rocprofv3, profiles/r05_fit_kernel_stats.csv): count 0.14, partition 0.65, item counts
// 0.25, placement 1.68 ms -- the placement's scattered stores are what is left of the old fill; 16 slices per bin kept on one XCD
// were slower (2.9-3.1 ms in all against 2.55).
constexpr int BIN_LOG2 = 12;
constexpr int BIN_ITEMS = 1 << BIN_LOG2;
constexpr int BIN_MAX_BINS = 512;          // LDS tables of the partition pass; 2M items
constexpr int BIN_TILE = 8192;             // pairs per workgroup of the partition pass (96 KB of records in LDS)
constexpr int BIN_SLICES_MAX = 32;         // workgroups per bin of the placement: tuning group_pairs_bin_slices (default 4), at most this
                                           // (measured at 1e8 pairs: 4 -> 2.63 ms for the grouping, 8 / 16 / 32 -> 2.97-3.03: profiles/r06_fit_bin_slices_ab.txt)

struct __attribute__((packed, aligned(4))) BinRecord { int32_t item, user; float value; };

__global__ __launch_bounds__(1024) void seg_bin_count_kernel(const int32_t* __restrict__ xi, const float* __restrict__ drop_zero_of,
                                                            int64_t n_pairs, int32_t n_bins, int32_t* __restrict__ bin_count)
{
    __shared__ int cnt[BIN_MAX_BINS];
    for (int i = threadIdx.x; i < n_bins; i += 1024) cnt[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * (4 * BIN_TILE);
#pragma unroll 4
    for (int q = 0; q < 32; ++q) {
        const int64_t p = p0 + q * 1024 + threadIdx.x;
        int32_t it = p < n_pairs ? xi[p] : -1;
        if (drop_zero_of && p < n_pairs && drop_zero_of[p] == 0.f) it = -1;
        if ((it >> BIN_LOG2) >= n_bins) it = -1;              // (an id beyond the catalogue would index LDS out of bounds: skipped)
        if (it >= 0) atomicAdd(&cnt[it >> BIN_LOG2], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += 1024)
        if (cnt[i]) atomicAdd(bin_count + i, cnt[i]);
}

// bin_base [n_bins + 1] = exclusive scan of bin_count; bin_cursor [n_bins] = 0.  One workgroup.
__global__ __launch_bounds__(256) void seg_bin_scan_kernel(const int32_t* __restrict__ bin_count, int32_t n_bins,
                                                          int64_t* __restrict__ bin_base, int32_t* __restrict__ bin_cursor)
{
    __shared__ long long part[256];
    const int per = (n_bins + 255) / 256;
    long long s = 0;
    for (int j = 0; j < per; ++j) { const int i = threadIdx.x * per + j; if (i < n_bins) s += bin_count[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long acc = 0;
        for (int t = 0; t < 256; ++t) { const long long c = part[t]; part[t] = acc; acc += c; }
        bin_base[n_bins] = acc;
    }
    __syncthreads();
    long long acc = part[threadIdx.x];
    for (int j = 0; j < per; ++j) {
        const int i = threadIdx.x * per + j;
        if (i < n_bins) { bin_base[i] = acc; acc += bin_count[i]; bin_cursor[i] = 0; }
    }
}

__global__ __launch_bounds__(1024) void seg_bin_partition_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                                const float* __restrict__ values, int64_t n_pairs,
                                                                int32_t pairs_per_user, int32_t drop_zero, int32_t n_bins,
                                                                const int64_t* __restrict__ bin_base, int32_t* __restrict__ bin_cursor,
                                                                int32_t* __restrict__ records)
{
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    int32_t* flat = (int32_t*)bsm;                                      // [3 * BIN_TILE] the tile's records {item, user, value}, sorted by bin
    int* start = (int*)(bsm + (size_t)BIN_TILE * 12);                   // [n_bins + 1] local start of every bin in the sorted tile
    int* cnt = start + BIN_MAX_BINS + 1;                                // [n_bins] counts
    long long* off = (long long*)(cnt + BIN_MAX_BINS + 1);              // [n_bins] record index in the bin's region of local record 0 of the bin
    for (int i = threadIdx.x; i < n_bins; i += 1024) cnt[i] = 0;
    __syncthreads();
    constexpr int PT = BIN_TILE / 1024;
    const int64_t p0 = (int64_t)blockIdx.x * BIN_TILE;
    // everything the tile reads leaves together: items, values, users
    int32_t it[PT], us[PT], lr[PT];
    float vl[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int64_t p = p0 + q * 1024 + threadIdx.x;
        const int64_t pc = p < n_pairs ? p : n_pairs - 1;
        const int32_t t = xi[pc];
        vl[q] = values[pc];
        it[q] = (p < n_pairs && !(drop_zero && vl[q] == 0.f) && (t >> BIN_LOG2) < n_bins) ? t : -1;     // (ids beyond the catalogue: skipped as in the count pass)
        us[q] = xu ? xu[pc] : (int32_t)((uint32_t)pc / (uint32_t)pairs_per_user);          // (n_pairs < 2^31)
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) lr[q] = it[q] >= 0 ? atomicAdd(&cnt[it[q] >> BIN_LOG2], 1) : 0;
    __syncthreads();
    // exclusive scan of the counts (n_bins <= 512: the first 512 threads matter, two wave levels)
    {
        __shared__ int wsum[16];
        const int i = threadIdx.x;
        const int c = i < n_bins ? cnt[i] : 0;
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if ((i & 63) >= o) inc += t;
        }
        if ((i & 63) == 63) wsum[i >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (i >> 6); ++w) base += wsum[w];
        if (i < n_bins) {
            start[i] = base + inc - c;
            // the tile's run in the bin's region: one global atomic per bin that has records
            const int g = c ? atomicAdd(bin_cursor + i, c) : 0;
            off[i] = bin_base[i] + g - (base + inc - c);
        }
        if (i == n_bins - 1) start[n_bins] = base + inc;
        __syncthreads();
    }
    // the records, sorted by bin, in LDS (three dwords each: stride 3 is conflict-free)
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        if (it[q] >= 0) {
            const int w = (start[it[q] >> BIN_LOG2] + lr[q]) * 3;
            flat[w] = it[q];
            flat[w + 1] = us[q];
            flat[w + 2] = __float_as_int(vl[q]);
        }
    }
    __syncthreads();
    // ... and out, DWORD by dword: consecutive lanes write consecutive addresses of the bin's run (12-byte stores per lane would be
    // three instructions of stride-12 partial writes each)
    const int total3 = start[n_bins] * 3;
    for (int w = threadIdx.x; w < total3; w += 1024) {
        const int j = w / 3;
        const int b = flat[j * 3] >> BIN_LOG2;
        records[(off[b] + j) * 3 + (w - j * 3)] = flat[w];
    }
}

// (3) in three launches, BIN_SLICES workgroups per bin (one workgroup per bin had 245 of them walk 4.9 MB twice: latency): slice g
// of a bin's records is counted per item in LDS (seg_bin_count_items_kernel -> run_counts [bin][slice][4096]); one workgroup per bin
// turns the counts into every slice's first slot per item and writes the bin's indptr (seg_bin_scan_items_kernel); every slice
// places its records from LDS cursors that start there (seg_bin_place_kernel).  Slices keep their order inside a bucket.
__global__ __launch_bounds__(1024) void seg_bin_count_items_kernel(const BinRecord* __restrict__ records, const int64_t* __restrict__ bin_base,
                                                                  int32_t* __restrict__ run_counts, int BIN_SLICES)
{
    __shared__ int cnt[BIN_ITEMS];
    const int b = blockIdx.x / BIN_SLICES, g = blockIdx.x % BIN_SLICES;
    const int64_t r0 = bin_base[b], n = bin_base[b + 1] - r0;
    const int64_t s0 = r0 + n * g / BIN_SLICES, s1 = r0 + n * (g + 1) / BIN_SLICES;
    for (int i = threadIdx.x; i < BIN_ITEMS; i += 1024) cnt[i] = 0;
    __syncthreads();
    const int32_t item0 = b << BIN_LOG2;
    for (int64_t j = s0 + threadIdx.x; j < s1; j += 1024) atomicAdd(&cnt[records[j].item - item0], 1);
    __syncthreads();
    int32_t* out = run_counts + (int64_t)blockIdx.x * BIN_ITEMS;
    for (int i = threadIdx.x; i < BIN_ITEMS; i += 1024) out[i] = cnt[i];
}

__global__ __launch_bounds__(1024) void seg_bin_scan_items_kernel(int32_t* __restrict__ run_counts, const int64_t* __restrict__ bin_base,
                                                                 int64_t n_items, int64_t* __restrict__ indptr, int BIN_SLICES)
{
    __shared__ int wsum[16];
    const int b = blockIdx.x;
    const int64_t r0 = bin_base[b], r1 = bin_base[b + 1];
    int32_t* rc = run_counts + (int64_t)b * BIN_SLICES * BIN_ITEMS;
    const int i4 = threadIdx.x * 4;
    int tot[4] = {0, 0, 0, 0};
    // per item: the slices' counts become their exclusive prefix over the slices; tot = the item's count
    for (int g = 0; g < BIN_SLICES; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = rc[g * BIN_ITEMS + i4 + e];
            rc[g * BIN_ITEMS + i4 + e] = tot[e];
            tot[e] += c;
        }
    }
    const int c = tot[0] + tot[1] + tot[2] + tot[3];
    int inc = c;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if ((threadIdx.x & 63) >= off) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    int st[4];
    st[0] = base + inc - c; st[1] = st[0] + tot[0]; st[2] = st[1] + tot[1]; st[3] = st[2] + tot[2];
    const int32_t item0 = b << BIN_LOG2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (item0 + i4 + e < n_items) indptr[item0 + i4 + e] = r0 + st[e];
        for (int g = 0; g < BIN_SLICES; ++g) rc[g * BIN_ITEMS + i4 + e] += st[e];      // the slice's first slot of the item, relative to r0
    }
    if (b == (int)gridDim.x - 1 && threadIdx.x == 0) indptr[n_items] = r1;
}

__global__ __launch_bounds__(1024) void seg_bin_place_kernel(const BinRecord* __restrict__ records, const int64_t* __restrict__ bin_base,
                                                            const int32_t* __restrict__ run_base, int2* __restrict__ entries, int BIN_SLICES)
{
    __shared__ int cur[BIN_ITEMS];
    const int b = blockIdx.x / BIN_SLICES, g = blockIdx.x % BIN_SLICES;
    const int64_t r0 = bin_base[b], n = bin_base[b + 1] - r0;
    const int64_t s0 = r0 + n * g / BIN_SLICES, s1 = r0 + n * (g + 1) / BIN_SLICES;
    const int32_t* rb = run_base + (int64_t)blockIdx.x * BIN_ITEMS;
    for (int i = threadIdx.x; i < BIN_ITEMS; i += 1024) cur[i] = rb[i];
    __syncthreads();
    const int32_t item0 = b << BIN_LOG2;
    for (int64_t j = s0 + threadIdx.x; j < s1; j += 1024) {
        const BinRecord r = records[j];
        const int slot = atomicAdd(&cur[r.item - item0], 1);
        entries[r0 + slot] = make_int2(r.user, __float_as_int(r.value));
    }
}

// ---- (3'), round 6: a SECOND partition level instead of the scattered placement --------------------------------------------------
// The placement above writes one 8-byte entry per record to base + (LDS cursor of its item)++ : 1e8 scattered stores into 3 MB
// regions, 1.5 ms of the grouping's 2.6.  Two levels make every store part of a whole run:
//   fine bins of 2^sub_log2 consecutive items, sized so that a fine bin's records (~13,000 expected) fit in LDS;
//   (1') seg_fine_count_kernel counts the pairs per FINE bin (LDS counters, a few hundred persistent workgroups, one global atomic
//        per workgroup and non-empty counter); seg_fine_scan_kernel turns them into the fine bins' bases, the bins' bases (a bin =
//        2^(12 - sub_log2) fine bins) and the tile map of level 2;
//   (2)  seg_bin_partition_kernel as above (bins of 4,096 items);
//   (2') seg_bin_subpartition_kernel: a workgroup takes a tile of 8,192 records of ONE bin, sorts it by fine bin in LDS (ranks from
//        wave ballots: <= 64 fine bins per bin, plain LDS atomics on so few counters would serialise) and writes the runs (~128-256
//        records each) into the fine bins' regions, dword by dword;
//   (3') seg_fine_place_kernel: one workgroup per fine bin loads its records, counts per item in LDS, scans (= indptr of its items),
//        places {user, value} in LDS and copies the fine bin's entries out as ONE contiguous run.  A fine bin with more records
//        than FINE_CAP (skewed lists, drop-zero on a learned model) places straight into global memory from LDS cursors instead.
constexpr int FINE_CAP = 16384;            // entries of a fine bin staged in LDS (128 KB)
constexpr int FINE_EXPECT = 13000;         // expected records per fine bin the size is chosen for (3,000 above the mean = 26 sigma)
constexpr int FINE_MAX = 16384;            // fine bins (LDS counters of the count pass)
constexpr int FINE_Q = FINE_CAP / 1024;    // records per thread of the placement's fast path

// fine-bin size for a pair list, or BIN_LOG2 (= no second level) when a bin itself is small enough / the list is too dense
static int fine_log2_for(int64_t n_pairs, int64_t n_items)
{
    int s = BIN_LOG2;
    while (s > 6 && (double)n_pairs / (double)n_items * (double)((int64_t)1 << s) > (double)FINE_EXPECT) --s;
    // (lists so dense that even 64 items overflow the LDS stage -- thousands of pairs per item -- keep the one-level form: every
    // fine bin would take the cursor path)
    if ((double)n_pairs / (double)n_items * (double)((int64_t)1 << s) > (double)FINE_CAP) return BIN_LOG2;
    const int64_t n_bins = ceil_div64(n_items, BIN_ITEMS);
    while (s < BIN_LOG2 && (n_bins << (BIN_LOG2 - s)) > FINE_MAX) ++s;
    return s;
}

__global__ __launch_bounds__(1024) void seg_fine_count_kernel(const int32_t* __restrict__ xi, const float* __restrict__ drop_zero_of,
                                                             int64_t n_pairs, int32_t n_bins, int32_t sub_log2, int32_t n_fine,
                                                             int32_t* __restrict__ fine_count)
{
    extern __shared__ __attribute__((aligned(16))) char csm[];
    int* cnt = (int*)csm;                                               // [n_fine]
    for (int i = threadIdx.x; i < n_fine; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int64_t p0 = (int64_t)blockIdx.x * BIN_TILE; p0 < n_pairs; p0 += (int64_t)gridDim.x * BIN_TILE) {
        int32_t it[BIN_TILE / 1024];
        float vl[BIN_TILE / 1024];
#pragma unroll
        for (int q = 0; q < BIN_TILE / 1024; ++q) {
            const int64_t p = p0 + q * 1024 + threadIdx.x;
            const int64_t pc = p < n_pairs ? p : n_pairs - 1;
            it[q] = xi[pc];
            vl[q] = drop_zero_of ? drop_zero_of[pc] : 1.f;
            if (p >= n_pairs) it[q] = -1;
        }
#pragma unroll
        for (int q = 0; q < BIN_TILE / 1024; ++q) {
            if (vl[q] == 0.f || (it[q] >> BIN_LOG2) >= n_bins) it[q] = -1;
            if (it[q] >= 0) atomicAdd(&cnt[it[q] >> sub_log2], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_fine; i += 1024)
        if (cnt[i]) atomicAdd(fine_count + i, cnt[i]);
}

// One workgroup: fine_base [n_fine + 1] = exclusive scan of fine_count, fine_cursor = 0; bin_base [n_bins + 1] = the fine bases at
// bin boundaries, bin_cursor = 0; tile_start [n_bins + 1] = exclusive scan of ceil(records of the bin / BIN_TILE).
__global__ __launch_bounds__(1024) void seg_fine_scan_kernel(const int32_t* __restrict__ fine_count, int32_t n_fine, int32_t n_bins,
                                                            int32_t sub_log2, int64_t* __restrict__ fine_base,
                                                            int32_t* __restrict__ fine_cursor, int64_t* __restrict__ bin_base,
                                                            int32_t* __restrict__ bin_cursor, int32_t* __restrict__ tile_start)
{
    __shared__ long long part[1024];
    __shared__ int tiles[BIN_MAX_BINS + 1];
    const int per = (n_fine + 1023) / 1024;
    long long s = 0;
    for (int j = 0; j < per; ++j) { const int i = threadIdx.x * per + j; if (i < n_fine) s += fine_count[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long acc = 0;
        for (int t = 0; t < 1024; ++t) { const long long c = part[t]; part[t] = acc; acc += c; }
        fine_base[n_fine] = acc;
    }
    __syncthreads();
    long long acc = part[threadIdx.x];
    for (int j = 0; j < per; ++j) {
        const int i = threadIdx.x * per + j;
        if (i < n_fine) { fine_base[i] = acc; acc += fine_count[i]; fine_cursor[i] = 0; }
    }
    __threadfence_block();
    __syncthreads();
    const int sh = BIN_LOG2 - sub_log2;
    if ((int)threadIdx.x <= n_bins) {
        const int b = threadIdx.x;
        const long long lo = fine_base[(int64_t)b << sh < n_fine ? (int64_t)b << sh : n_fine];
        bin_base[b] = lo;
        if (b < n_bins) {
            const long long hi = fine_base[(int64_t)(b + 1) << sh < n_fine ? (int64_t)(b + 1) << sh : n_fine];
            tiles[b] = (int)((hi - lo + BIN_TILE - 1) / BIN_TILE);
            bin_cursor[b] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0;
        for (int b = 0; b < n_bins; ++b) { tile_start[b] = a; a += tiles[b]; }
        tile_start[n_bins] = a;
    }
}

__global__ __launch_bounds__(1024) void seg_bin_subpartition_kernel(const BinRecord* __restrict__ rec_in, const int64_t* __restrict__ bin_base,
                                                                   const int32_t* __restrict__ tile_start, int32_t n_bins,
                                                                   int32_t sub_log2, const int64_t* __restrict__ fine_base,
                                                                   int32_t* __restrict__ fine_cursor, int32_t* __restrict__ rec_out)
{
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    int32_t* flat = (int32_t*)bsm;                                      // [3 * BIN_TILE] the tile's records, sorted by fine bin
    long long* off = (long long*)(bsm + (size_t)BIN_TILE * 12);         // [64] record index in the fine bin's region of local record 0 of it
    int* start = (int*)(off + 64);                                      // [65]
    int* cnt = start + 65;                                              // [64]
    if ((int)blockIdx.x >= tile_start[n_bins]) return;
    // the bin of this tile: the last b with tile_start[b] <= blockIdx.x (every thread the same search: uniform, cached loads)
    int lo = 0, hi = n_bins - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int b = lo;
    const int64_t r0 = bin_base[b] + (int64_t)((int)blockIdx.x - tile_start[b]) * BIN_TILE;
    const int64_t rest = bin_base[b + 1] - r0;
    const int n = rest < BIN_TILE ? (int)rest : BIN_TILE;
    const int sh = BIN_LOG2 - sub_log2, nsub = 1 << sh;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    constexpr int PT = BIN_TILE / 1024;
    const int lane = threadIdx.x & 63;
    BinRecord rc[PT];
    int lr[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int j = q * 1024 + threadIdx.x;
        rc[q] = rec_in[r0 + (j < n ? j : 0)];
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int j = q * 1024 + threadIdx.x;
        const bool valid = j < n;
        const int sb = (rc[q].item >> sub_log2) & (nsub - 1);
        // the lanes of this wave with the same fine bin (six ballots), rank = the ones below me; ONE LDS atomic per distinct fine bin
        unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < 6; ++bit) {
            const unsigned long long bb = __builtin_amdgcn_ballot_w64((sb >> bit) & 1);
            m &= ((sb >> bit) & 1) ? bb : ~bb;
        }
        lr[q] = 0;
        if (valid) {
            const int leader = __builtin_ctzll(m);
            const int below = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            int base = 0;
            if (lane == leader) base = atomicAdd(&cnt[sb], __builtin_popcountll(m));
            base = __shfl(base, leader, 64);
            lr[q] = base + below;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        const int c = i < nsub ? cnt[i] : 0;
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (i >= o) inc += t;
        }
        start[i] = inc - c;
        if (i == 63) start[64] = inc;
        if (c) {
            const int64_t f = ((int64_t)b << sh) + i;
            const int g = atomicAdd(fine_cursor + f, c);
            off[i] = fine_base[f] + g - (inc - c);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        if (q * 1024 + (int)threadIdx.x < n) {
            const int sb = (rc[q].item >> sub_log2) & (nsub - 1);
            const int w = (start[sb] + lr[q]) * 3;
            flat[w] = rc[q].item;
            flat[w + 1] = rc[q].user;
            flat[w + 2] = __float_as_int(rc[q].value);
        }
    }
    __syncthreads();
    const int total3 = n * 3;
    for (int w = threadIdx.x; w < total3; w += 1024) {
        const int j = w / 3;
        const int sb = (flat[j * 3] >> sub_log2) & (nsub - 1);
        rec_out[(off[sb] + j) * 3 + (w - j * 3)] = flat[w];
    }
}

__global__ __launch_bounds__(1024) void seg_fine_place_kernel(const BinRecord* __restrict__ records, const int64_t* __restrict__ fine_base,
                                                             int32_t sub_log2, int32_t n_fine, int64_t n_items,
                                                             int64_t* __restrict__ indptr, int2* __restrict__ entries)
{
    extern __shared__ __attribute__((aligned(16))) char psm[];
    int2* out = (int2*)psm;                                             // [FINE_CAP]
    int* cnt = (int*)(psm + (size_t)FINE_CAP * 8);                      // [BIN_ITEMS]
    __shared__ int wsum[16];
    const int f = blockIdx.x;
    const int64_t r0 = fine_base[f];
    const int64_t n = fine_base[f + 1] - r0;
    const int64_t item0 = (int64_t)f << sub_log2;
    for (int i = threadIdx.x; i < BIN_ITEMS; i += 1024) cnt[i] = 0;
    __syncthreads();
    const bool fast = n <= FINE_CAP;
    const int64_t safe = r0 > 0 ? r0 - 1 : 0;                            // (an empty fine bin at the very end: no read past the records)
    int it[FINE_Q], us[FINE_Q], vl[FINE_Q], rk[FINE_Q];
    if (fast) {
#pragma unroll
        for (int q = 0; q < FINE_Q; ++q) {
            const int j = q * 1024 + threadIdx.x;
            const BinRecord r = records[j < n ? r0 + j : safe];
            it[q] = j < n ? (int)(r.item - item0) : -1;
            us[q] = r.user;
            vl[q] = __float_as_int(r.value);
        }
#pragma unroll
        for (int q = 0; q < FINE_Q; ++q) { rk[q] = 0; if (it[q] >= 0) rk[q] = atomicAdd(&cnt[it[q]], 1); }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += 1024) atomicAdd(&cnt[records[r0 + j].item - item0], 1);
    }
    __syncthreads();
    // exclusive scan of the (up to 4,096) counts in place: thread t owns entries 4 t .. 4 t + 3
    {
        const int i4 = threadIdx.x * 4;
        int c4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c4[e] = cnt[i4 + e];
        const int c = c4[0] + c4[1] + c4[2] + c4[3];
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if ((int)(threadIdx.x & 63) >= o) inc += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        int st = base + inc - c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cnt[i4 + e] = st;
            if (i4 + e < (1 << sub_log2) && item0 + i4 + e < n_items) indptr[item0 + i4 + e] = r0 + st;
            st += c4[e];
        }
    }
    if (f == n_fine - 1 && threadIdx.x == 0) indptr[n_items] = fine_base[n_fine];
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int q = 0; q < FINE_Q; ++q)
            if (it[q] >= 0) out[cnt[it[q]] + rk[q]] = make_int2(us[q], vl[q]);
        __syncthreads();
        for (int j = threadIdx.x; j < (int)n; j += 1024) entries[r0 + j] = out[j];
    } else {
        for (int64_t j = threadIdx.x; j < n; j += 1024) {
            const BinRecord r = records[r0 + j];
            const int slot = atomicAdd(&cnt[r.item - item0], 1);
            entries[r0 + slot] = make_int2(r.user, __float_as_int(r.value));
        }
    }
}

// workspace bytes of trec_group_pairs_by_item_binned, or 0 when the form does not cover the size (more than 512 bins of 4,096
// items, fewer than 2^22 pairs)
extern "C" int64_t trec_group_pairs_binned_bytes(int64_t n_pairs, int64_t n_items)
{
    const int64_t n_bins = ceil_div64(n_items, BIN_ITEMS);
    if (n_items < 1 || n_bins > BIN_MAX_BINS || n_pairs < ((int64_t)1 << 22) || n_pairs >= ((int64_t)1 << 31)) return 0;
    {
        // the partition pass needs ~106.5 KB of dynamic LDS per workgroup: a device that cannot give it takes the ranked grouping
        int dev = 0, max_lds = 0;
        const int need_lds = BIN_TILE * 12 + (2 * BIN_MAX_BINS + 2) * 4 + BIN_MAX_BINS * 8 + 16;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
            max_lds > 0 && max_lds < need_lds) return 0;
    }
    // (the two-level form's second record array, the fine bins' counts / bases / cursors and the tile map come on top)
    return n_pairs * (int64_t)sizeof(BinRecord) + (n_bins + 1) * 8 + 2 * n_bins * 4 + n_bins * BIN_SLICES_MAX * (int64_t)BIN_ITEMS * 4 + 64 +
           n_pairs * (int64_t)sizeof(BinRecord) + 8 + (FINE_MAX + 1) * 8 + 2 * FINE_MAX * 4 + (BIN_MAX_BINS + 1) * 4 + 64;
}

// Group (user, item, value) pairs by item without ranks (see above): xi [n_pairs] items (negative: skipped), users from xu or
// p / pairs_per_user, values [n_pairs].  Writes indptr_t int64 [n_items + 1] and entries int2 [n_pairs] = {user, value bits} (the
// operand of trec_spmm_csr_packed).  drop_zero_values != 0: pairs whose value is exactly 0 are left out as well (a WMRB sample that
// violates no margin has coefficient 0 and adds nothing to its item's gradient: the sort and the item-side gather shrink to the
// active pairs -- most of them are inactive once a model has learned anything); indptr_t[n_items] = pairs kept.  Meant for pair lists that are roughly uniform over the items (sampled pairs): a bin is
// placed by BIN_SLICES workgroups.  workspace: trec_group_pairs_binned_bytes(n_pairs, n_items) bytes.
extern "C" int trec_group_pairs_by_item_binned(const int32_t* xu, const int32_t* xi, const float* values, int64_t n_pairs,
                                               int32_t pairs_per_user, int64_t n_items, int32_t drop_zero_values, void* workspace,
                                               int64_t workspace_bytes, int64_t* indptr_t, int32_t* entries, void* stream)
{
    TREC_REQUIRE(xi && values && workspace && indptr_t && entries, "trec_group_pairs_by_item_binned: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_group_pairs_by_item_binned: need xu or pairs_per_user");
    const int64_t need = trec_group_pairs_binned_bytes(n_pairs, n_items);
    TREC_REQUIRE(need > 0 && workspace_bytes >= need, "trec_group_pairs_by_item_binned: size not covered / workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int32_t n_bins = (int32_t)ceil_div64(n_items, BIN_ITEMS);
    BinRecord* records = (BinRecord*)workspace;
    int64_t* bin_base = (int64_t*)((char*)workspace + (n_pairs * (int64_t)sizeof(BinRecord) + 7) / 8 * 8);
    int32_t* bin_count = (int32_t*)(bin_base + n_bins + 1);
    int32_t* bin_cursor = bin_count + n_bins;
    int32_t* run_counts = bin_cursor + n_bins;
    const int sub_log2 = fine_log2_for(n_pairs, n_items);
    int max_lds_dev = 0;
    {
        int d0 = 0;
        if (hipGetDevice(&d0) != hipSuccess || hipDeviceGetAttribute(&max_lds_dev, hipDeviceAttributeMaxSharedMemoryPerBlock, d0) != hipSuccess)
            max_lds_dev = 0;
    }
    // (the fine bins' placement stages 128 KB of entries in LDS: a device without it keeps the one-level form)
    if (sub_log2 < BIN_LOG2 && trec_get_tuning("group_pairs_two_level", 1) != 0 && max_lds_dev >= FINE_CAP * 8 + BIN_ITEMS * 4 + 64) {
        // ---- two partition levels, every store part of a run (see above)
        char* extra = (char*)(run_counts + (int64_t)n_bins * BIN_SLICES_MAX * BIN_ITEMS);
        extra = (char*)(((uintptr_t)extra + 7) / 8 * 8);
        BinRecord* records2 = (BinRecord*)extra;
        int64_t* fine_base = (int64_t*)(extra + (n_pairs * (int64_t)sizeof(BinRecord) + 7) / 8 * 8);
        int32_t* fine_count = (int32_t*)(fine_base + FINE_MAX + 1);
        int32_t* fine_cursor = fine_count + FINE_MAX;
        int32_t* tile_start = fine_cursor + FINE_MAX;
        const int32_t n_fine = n_bins << (BIN_LOG2 - sub_log2);
        if (hipMemsetAsync(fine_count, 0, sizeof(int32_t) * (size_t)n_fine, st) != hipSuccess) {
            trec_set_last_error("trec_group_pairs_by_item_binned: memset failed");
            return TREC_ERR_LAUNCH;
        }
        static bool attr2_set[64] = {};
        int dev2 = 0;
        TREC_REQUIRE(hipGetDevice(&dev2) == hipSuccess && dev2 >= 0 && dev2 < 64, "trec_group_pairs_by_item_binned: no current device");
        const int lds1 = BIN_TILE * 12 + (2 * BIN_MAX_BINS + 2) * 4 + BIN_MAX_BINS * 8 + 16;
        const int lds2 = BIN_TILE * 12 + 64 * 8 + (65 + 64) * 4 + 16;
        const int lds3 = FINE_CAP * 8 + BIN_ITEMS * 4;
        const int ldsc = FINE_MAX * 4;
        if (!attr2_set[dev2]) {
            TREC_REQUIRE(hipFuncSetAttribute((const void*)seg_bin_partition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds1) == hipSuccess &&
                         hipFuncSetAttribute((const void*)seg_bin_subpartition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds2) == hipSuccess &&
                         hipFuncSetAttribute((const void*)seg_fine_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds3) == hipSuccess &&
                         hipFuncSetAttribute((const void*)seg_fine_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsc) == hipSuccess,
                         "trec_group_pairs_by_item_binned: the device refused the dynamic LDS size");
            attr2_set[dev2] = true;
        }
        int64_t count_wgs = ceil_div64(n_pairs, 24 * (int64_t)BIN_TILE);
        if (count_wgs > 1024) count_wgs = 1024;
        if (count_wgs < 1) count_wgs = 1;
        hipLaunchKernelGGL(seg_fine_count_kernel, dim3((unsigned)count_wgs), dim3(1024), (size_t)n_fine * 4, st, xi,
                           drop_zero_values ? values : (const float*)nullptr, n_pairs, n_bins, sub_log2, n_fine, fine_count);
        hipLaunchKernelGGL(seg_fine_scan_kernel, dim3(1), dim3(1024), 0, st, fine_count, n_fine, n_bins, sub_log2, fine_base, fine_cursor,
                           bin_base, bin_cursor, tile_start);
        hipLaunchKernelGGL(seg_bin_partition_kernel, dim3((unsigned)ceil_div64(n_pairs, BIN_TILE)), dim3(1024), lds1, st, xu, xi, values,
                           n_pairs, pairs_per_user, drop_zero_values, n_bins, bin_base, bin_cursor, (int32_t*)records);
        hipLaunchKernelGGL(seg_bin_subpartition_kernel, dim3((unsigned)(ceil_div64(n_pairs, BIN_TILE) + n_bins)), dim3(1024), lds2, st, records,
                           bin_base, tile_start, n_bins, sub_log2, fine_base, fine_cursor, (int32_t*)records2);
        hipLaunchKernelGGL(seg_fine_place_kernel, dim3((unsigned)n_fine), dim3(1024), lds3, st, records2, fine_base, sub_log2, n_fine, n_items,
                           indptr_t, (int2*)entries);
        return trec_check_launch("trec_group_pairs_by_item_binned (two levels)");
    }
    if (hipMemsetAsync(bin_count, 0, sizeof(int32_t) * (size_t)n_bins, st) != hipSuccess) {
        trec_set_last_error("trec_group_pairs_by_item_binned: memset failed");
        return TREC_ERR_LAUNCH;
    }
    static bool attr_set[64] = {};
    int dev = 0;
    TREC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "trec_group_pairs_by_item_binned: no current device");
    const int lds = BIN_TILE * 12 + (2 * BIN_MAX_BINS + 2) * 4 + BIN_MAX_BINS * 8 + 16;
    if (!attr_set[dev]) {
        TREC_REQUIRE(hipFuncSetAttribute((const void*)seg_bin_partition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess,
                     "trec_group_pairs_by_item_binned: the device refused the dynamic LDS size");
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(seg_bin_count_kernel, dim3((unsigned)ceil_div64(n_pairs, 4 * BIN_TILE)), dim3(1024), 0, st, xi,
                       drop_zero_values ? values : (const float*)nullptr, n_pairs, n_bins, bin_count);
    hipLaunchKernelGGL(seg_bin_scan_kernel, dim3(1), dim3(256), 0, st, bin_count, n_bins, bin_base, bin_cursor);
    hipLaunchKernelGGL(seg_bin_partition_kernel, dim3((unsigned)ceil_div64(n_pairs, BIN_TILE)), dim3(1024), lds, st, xu, xi, values, n_pairs,
                       pairs_per_user, drop_zero_values, n_bins, bin_base, bin_cursor, (int32_t*)records);
    int slices = trec_get_tuning("group_pairs_bin_slices", 4);
    if (slices < 1) slices = 1;
    if (slices > BIN_SLICES_MAX) slices = BIN_SLICES_MAX;
    hipLaunchKernelGGL(seg_bin_count_items_kernel, dim3((unsigned)(n_bins * slices)), dim3(1024), 0, st, records, bin_base, run_counts, slices);
    hipLaunchKernelGGL(seg_bin_scan_items_kernel, dim3((unsigned)n_bins), dim3(1024), 0, st, run_counts, bin_base, n_items, indptr_t, slices);
    hipLaunchKernelGGL(seg_bin_place_kernel, dim3((unsigned)(n_bins * slices)), dim3(1024), 0, st, records, bin_base, run_counts, (int2*)entries, slices);
    return trec_check_launch("trec_group_pairs_by_item_binned");
}

// exclusive prefix sum of int32 counts into int64: out[i] = sum_{j<i} counts[j], out[n] = total.
// workspace_i64: ceil(n/1024) + 1 int64
extern "C" int trec_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* workspace_i64, int64_t* out, void* stream)
{
    TREC_REQUIRE(counts && workspace_i64 && out && n >= 1, "trec_exclusive_scan_i32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int n_blocks = (int)ceil_div64(n, 1024);
    int64_t* block_sum = workspace_i64;
    int64_t* total = workspace_i64 + n_blocks;
    hipLaunchKernelGGL(seg_scan_local_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, counts, n, out, block_sum);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(256), 0, st, block_sum, n_blocks, total);
    hipLaunchKernelGGL(seg_add_offsets_kernel, dim3((unsigned)ceil_div64(n + 1, 256)), dim3(256), 0, st, out, n, block_sum, total);
    return trec_check_launch("trec_exclusive_scan_i32");
}

// ---- a few tens of thousands of buckets, hundreds of millions of pairs: counters in LDS, no global atomics ------------------
// BASELINE.json configs[4] (MovieLens-20M-shaped): 3.7e8 sampled pairs per step over 26,744 items -- 13,800 pairs per bucket.
// The plain path issues one global atomic per pair in the histogram AND in the fill, thousands deep on every counter: 16 + 30 ms
// of a 200 ms epoch (profiles/r04_cfg4_fit_kernel_stats.csv).  The counters of <= 32,768 buckets fit the LDS of ONE workgroup
// (128 KB), so: (1) every workgroup counts a contiguous RUN of pairs in LDS and writes its counters out, run_counts [n_runs]
// [n_items]; (2) one thread per bucket scans its column over the runs (exclusive, in place) and leaves the bucket's total; the
// usual scan of the totals gives indptr; (3) every workgroup loads ITS row of run bases into LDS and places its run --
// slot = indptr[item] + (LDS cursor)++ -- so a run's pairs of one item land on consecutive slots.  No global atomic anywhere;
// the order of pairs inside a (run, item) group follows LDS atomic arrival (runs themselves are in order).
#define SEG_LDS_MAX 32768

__global__ __launch_bounds__(1024) void seg_lds_count_kernel(const int32_t* __restrict__ xi, int64_t n_pairs, int32_t n_items,
                                                            int64_t run_len, int32_t* __restrict__ run_counts)
{
    extern __shared__ int32_t l_cnt[];
    for (int i = threadIdx.x; i < n_items; i += 1024) l_cnt[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * run_len;
    const int64_t p1 = (p0 + run_len < n_pairs) ? p0 + run_len : n_pairs;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 1024) {
        const int32_t i = xi[p];
        if (i >= 0) atomicAdd(l_cnt + i, 1);
    }
    __syncthreads();
    int32_t* out = run_counts + (int64_t)blockIdx.x * n_items;
    for (int i = threadIdx.x; i < n_items; i += 1024) out[i] = l_cnt[i];
}

__global__ __launch_bounds__(256) void seg_lds_scan_runs_kernel(int32_t* __restrict__ run_counts, int32_t n_runs, int32_t n_items,
                                                               int32_t* __restrict__ counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    int32_t acc = 0;
    for (int r = 0; r < n_runs; ++r) {
        const int32_t c = run_counts[(int64_t)r * n_items + i];
        run_counts[(int64_t)r * n_items + i] = acc;
        acc += c;
    }
    counts[i] = acc;
}

__global__ __launch_bounds__(1024) void seg_lds_fill_kernel(const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                           int64_t n_pairs, int32_t pairs_per_user, int32_t n_items,
                                                           int64_t run_len, const int32_t* __restrict__ run_base,
                                                           const int64_t* __restrict__ indptr, int32_t* __restrict__ users_t,
                                                           int32_t* __restrict__ perm_t)
{
    extern __shared__ int32_t l_cur[];
    const int32_t* base = run_base + (int64_t)blockIdx.x * n_items;
    for (int i = threadIdx.x; i < n_items; i += 1024) l_cur[i] = base[i];
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * run_len;
    const int64_t p1 = (p0 + run_len < n_pairs) ? p0 + run_len : n_pairs;
    // four pairs per thread and step: their key loads, LDS cursors, row-pointer gathers and stores are independent chains
    int64_t p = p0 + threadIdx.x;
    for (; p + 3 * 1024 < p1; p += 4 * 1024) {
        int32_t it[4];
        int64_t slot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) it[q] = xi[p + q * 1024];
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[q] = it[q] >= 0 ? indptr[it[q]] + atomicAdd(l_cur + it[q], 1) : -1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (slot[q] >= 0) {
                const int64_t pp = p + q * 1024;
                users_t[slot[q]] = xu ? xu[pp] : (int32_t)(pp / pairs_per_user);
                perm_t[slot[q]] = (int32_t)pp;
            }
    }
    for (; p < p1; p += 1024) {
        const int32_t i = xi[p];
        if (i < 0) continue;
        const int64_t slot = indptr[i] + atomicAdd(l_cur + i, 1);
        users_t[slot] = xu ? xu[p] : (int32_t)(p / pairs_per_user);
        perm_t[slot] = (int32_t)p;
    }
}

// runs the LDS form uses for n_pairs (0: the form does not apply) and its workspace: run_counts int32 [n_runs][n_items]
extern "C" int32_t trec_group_pairs_lds_runs(int64_t n_pairs, int64_t n_items)
{
    if (n_items < 1 || n_items > SEG_LDS_MAX || n_pairs < ((int64_t)1 << 22)) return 0;
    // the counters of every bucket live in ONE workgroup's LDS: a device with less of it than the buckets need (a build for an
    // architecture with 64 KB) takes the global-atomic path instead of failing at launch
    int dev = 0, lds_max = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || n_items * 4 > lds_max) return 0;
    // one run per workgroup and CU.  (Measured on configs[4], 3.7e8 pairs over 26,744 buckets: 1,024 runs 21.4 ms, 256 runs
    // 22.6 ms -- the placement is bound by its 7.4e8 scattered 4-byte stores (PMC: 23 GB written for 3 GB of payload), which a
    // run's ~54 consecutive slots per bucket do not change: the slots of a cache line are written by different threads far apart
    // in time.  Coalescing them takes a two-level partition through LDS tiles; not built.)
    int64_t runs = ceil_div64(n_pairs, 262144);
    if (runs > 256) runs = 256;
    if (runs < 1) runs = 1;
    return (int32_t)runs;
}

// trec_group_pairs_by_item for <= 32,768 buckets and >= 4M pairs without a single global atomic (see above).
// workspace_i32: n_items int32 (the buckets' totals); workspace_i64: ceil(n_items / 1024) + 1 int64; run_counts: int32
// [trec_group_pairs_lds_runs(n_pairs, n_items)][n_items].  Outputs as trec_group_pairs_by_item (users_t, perm_t both required).
extern "C" int trec_group_pairs_by_item_lds(const int32_t* xu, const int32_t* xi, int64_t n_pairs, int32_t pairs_per_user,
                                            int64_t n_items, int32_t* workspace_i32, int64_t* workspace_i64,
                                            int32_t* run_counts, int64_t* indptr_t, int32_t* users_t, int32_t* perm_t,
                                            void* stream)
{
    TREC_REQUIRE(xi && workspace_i32 && workspace_i64 && run_counts && indptr_t && users_t && perm_t, "trec_group_pairs_by_item_lds: null pointer");
    TREC_REQUIRE(xu || pairs_per_user >= 1, "trec_group_pairs_by_item_lds: need xu or pairs_per_user");
    TREC_REQUIRE(n_pairs < ((int64_t)1 << 31), "trec_group_pairs_by_item_lds: n_pairs must fit int32");
    const int32_t n_runs = trec_group_pairs_lds_runs(n_pairs, n_items);
    TREC_REQUIRE(n_runs >= 1, "trec_group_pairs_by_item_lds: needs <= 32768 buckets and >= 4M pairs (trec_group_pairs_lds_runs)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t run_len = ceil_div64(n_pairs, n_runs);
    const int lds = (int)n_items * 4;
    // the dynamic-LDS limit is a per-DEVICE attribute of the function: set once per device id, and checked
    static bool attr_set[64] = {};
    int dev = 0;
    TREC_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "trec_group_pairs_by_item_lds: no current device");
    if (!attr_set[dev]) {
        const hipError_t e1 = hipFuncSetAttribute((const void*)seg_lds_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const hipError_t e2 = hipFuncSetAttribute((const void*)seg_lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        TREC_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "trec_group_pairs_by_item_lds: the device refused the dynamic LDS size");
        attr_set[dev] = lds >= SEG_LDS_MAX * 4;          // (a smaller request is repeated until the largest one has been granted)
    }
    int32_t* counts = workspace_i32;
    const int n_blocks = (int)ceil_div64(n_items, 1024);
    int64_t* block_sum = workspace_i64;
    int64_t* total = workspace_i64 + n_blocks;
    hipLaunchKernelGGL(seg_lds_count_kernel, dim3((unsigned)n_runs), dim3(1024), lds, st, xi, n_pairs, (int32_t)n_items, run_len, run_counts);
    hipLaunchKernelGGL(seg_lds_scan_runs_kernel, dim3((unsigned)ceil_div64(n_items, 256)), dim3(256), 0, st, run_counts, n_runs, (int32_t)n_items, counts);
    hipLaunchKernelGGL(seg_scan_local_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, counts, n_items, indptr_t, block_sum);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(256), 0, st, block_sum, n_blocks, total);
    hipLaunchKernelGGL(seg_add_offsets_kernel, dim3((unsigned)ceil_div64(n_items + 1, 256)), dim3(256), 0, st, indptr_t, n_items, block_sum, total);
    hipLaunchKernelGGL(seg_lds_fill_kernel, dim3((unsigned)n_runs), dim3(1024), lds, st, xu, xi, n_pairs, pairs_per_user, (int32_t)n_items, run_len, run_counts, indptr_t, users_t, perm_t);
    return trec_check_launch("trec_group_pairs_by_item_lds");
}
