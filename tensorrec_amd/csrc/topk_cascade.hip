// tensorrec_amd/csrc/topk_cascade.hip -- K2c: the glue of the int8 -> bf16 -> fp32 cascade of the exact top-k.
//
//   stage 0  trec_score_gemm_blockmax_i8 (score_blockmax_i8.hip): table[s][u] = int8 maximum of superblock s for user u
//   select   trec_topk_select_blocks: tau8_u = the k-th largest entry of column u
//   floor    trec_topk_filter_floor_i8 (here): floor8_u = tau8_u - 2 eps8_u, eps8_u >= |int8 score - fp32 score| proven
//            from the MEASURED quantisation error norms of trec_score_prep_i8
//   compact  trec_topk_rows_count / trec_topk_rows_fill (here): the (superblock, user) pairs with table >= floor8, already
//            grouped by superblock -- the table is superblock-major, so a row-wise stream compaction IS the grouping
//            (no sort): per superblock a run of user ids padded with -1 to whole 512-row workgroups
//   stage 1  trec_score_gemm_blockmax_grouped (here + score_blockmax.hip): the hand-scheduled bf16 kernel on the kept pairs
//            only (~3% of them at 1M x 1M); each bf16 maximum REPLACES the int8 entry of the table
//   then the bf16 filter of topk_filter.hip runs unchanged on the mixed table (select, floor16, collect, bf16 lists, fp32).
//
// Why the mixed table is sound (eps16 <= eps8 is NOT needed): every entry of column u -- int8 or bf16 -- certifies an item
// of its superblock with fp32 score >= entry - eps_kind >= entry - max(eps8, eps16).  (1) A superblock holding a true
// top-k item has int8 maximum >= tau8 - 2 eps8, so it was refined and its entry is a bf16 maximum M16 >= that item's
// fp32 score - eps16.  (2) With eps16 <= eps8 the k largest entries of the mixed column are all refined ones (k refined
// entries have M16 >= tau8 - eps8 - eps16 >= floor8 > every unrefined entry), so tau16 and floor16 = tau16 - 2 eps16 are
// exactly what the bf16 filter computes from a pure bf16 table restricted to the refined superblocks; with eps16 > eps8
// an unrefined entry among the k largest still certifies an item with fp32 score >= entry - eps8 >= entry - eps16, and
// the argument of topk_filter.hip goes through verbatim.  Unrefined entries that pass floor16 are false positives: their
// superblocks are re-scored like any other.
//
// Replaces (as a filter) tf.matmul of tensorrec/prediction_graphs.py:49-50 + tf.nn.top_k of
// tensorrec/recommendation_graphs.py:80; the results come from the fp32 finish of topk_filter.hip, bit-identical to the oracle.
#include "score_common.hpp"
#include "topk_common.hpp"

namespace {

constexpr int CROWS = 8;          // table rows (superblocks) per workgroup
constexpr int CUSERS = 1024;      // users per workgroup: one float4 per thread
constexpr int GROUP_ROWS = 512;   // resident rows per workgroup of the grouped bf16 kernel

__global__ __launch_bounds__(256) void filter_floor_i8_kernel(const float* __restrict__ tau, const float2* __restrict__ ustats,
                                                             const float* __restrict__ user_bias,
                                                             const float* __restrict__ gstats, int kdim, int64_t n_users,
                                                             float* __restrict__ floor_, int32_t* __restrict__ flag,
                                                             int32_t* __restrict__ n_flagged)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    // gstats: max over items of {||y|| + ||dy||  (>= ||b q_i||),  ||dy||,  |bias|,  |bias - scale_prod bias_q|}
    const float ni = gstats[0], ai = gstats[1], bi = gstats[2], db = gstats[3];
    const float2 st = ustats[u];                                                // {||x||, ||x - a q_u||}
    const float bu = user_bias ? fabsf(user_bias[u]) : 0.f;
    // int8 score (real arithmetic) = a b (q_u . q_i + bq_i) + b_u;  fp32 score = fl-chain(x . y) + b_u + b_i:
    //   |x . y - a b q_u . q_i| = |<x - a q_u, b q_i> + <x, y - b q_i>| <= ||dx|| ||b q_i|| + ||x|| ||dy||
    //   the reference's chain and its two bias adds, the table's conversion and add: (K + 4) roundings of terms bounded by
    //   ||x|| ||y|| + |b_u| + |b_i|, covered by ck below with room to spare; the bias quantisation adds db.
    const float ck = (float)(kdim + 4) * 2.98023224e-07f;                       // (K + 4) (2^-24 + 2^-22)
    float eps = st.y * ni + st.x * ai + ck * (st.x * ni + bu + bi) + db;
    eps = eps * 1.001953125f + 1e-30f;
    const float t = tau[u];
    float f = t - 2.0f * eps;
    bool bad = !(eps < INFINITY);
    if (t == -INFINITY) f = -INFINITY;                                          // fewer than k superblocks: keep all
    else if (!(f == f)) bad = true;
    else f = float_pred(float_pred(f));
    if (bad) f = -INFINITY;                                                     // every superblock of this user is refined
    floor_[u] = f;
    if (flag) flag[u] = bad ? 1 : 0;
    if (bad && n_flagged) atomicAdd(n_flagged, 1);
}

// table rows s0 .. s0 + CROWS - 1, users u .. u + 3 of this thread: bit (4 r + e) of the result = table[s0 + r][u + e] >= floor
__device__ __forceinline__ unsigned int tile_bits(const float* __restrict__ table, int32_t n_sb, int64_t n_users, int64_t stride,
                                                  const float* __restrict__ floor_, int32_t s0, int64_t u)
{
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = (u + e < n_users) ? floor_[u + e] : INFINITY;
    const bool vec = (stride % 4 == 0) && (((uintptr_t)table % 16) == 0) && (u + 3 < stride);
    unsigned int bits = 0;
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        const int32_t s = s0 + r;
        f32x4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (s < n_sb) {
            const float* src = table + (int64_t)s * stride + u;
            if (vec) v = __builtin_nontemporal_load((const f32x4*)src);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (u + e < n_users) v[e] = src[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (s < n_sb && u + e < n_users && !(v[e] < f[e])) bits |= 1u << (4 * r + e);
    }
    return bits;
}

__global__ __launch_bounds__(256) void rows_count_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                        int64_t stride, const float* __restrict__ floor_, int32_t n_ublk,
                                                        int32_t* __restrict__ blockcnt)
{
    __shared__ int cnt[CROWS];
    if (threadIdx.x < CROWS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t s0 = blockIdx.y * CROWS;
    const int64_t u = (int64_t)blockIdx.x * CUSERS + threadIdx.x * 4;
    const unsigned int bits = tile_bits(table, n_sb, n_users, stride, floor_, s0, u);
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        int c = __builtin_popcount((bits >> (4 * r)) & 15u);
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[r], c);
    }
    __syncthreads();
    if (threadIdx.x < CROWS && s0 + threadIdx.x < n_sb)
        blockcnt[(int64_t)(s0 + threadIdx.x) * n_ublk + blockIdx.x] = cnt[threadIdx.x];
}

// one workgroup per table row: exclusive scan of the row's block counts in place, the row's total and its padded size
__global__ __launch_bounds__(256) void rows_scan_kernel(int32_t* __restrict__ blockcnt, int32_t n_ublk,
                                                       int32_t* __restrict__ row_total, int32_t* __restrict__ row_pad)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int s = blockIdx.x;
    int32_t* row = blockcnt + (int64_t)s * n_ublk;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_ublk; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int c = b < n_ublk ? row[b] : 0;
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if ((threadIdx.x & 63) >= off) inc += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        if (b < n_ublk) row[b] = base + inc - c;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        row_total[s] = carry_s;
        row_pad[s] = (carry_s + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
    }
}

// single workgroup: pstart[s] = sum of the padded sizes of rows < s, pstart[n_sb] = the grouped launch's resident rows
// status[0] = pstart[n_sb], status[1] = 1 when that exceeds cap_rows (the caller's row_user capacity): pass 2 and the grouped
// launch then do nothing and the caller falls back to the dense bf16 stage 1
__global__ __launch_bounds__(256) void rows_pstart_kernel(const int32_t* __restrict__ row_pad, int32_t n_sb,
                                                         int64_t* __restrict__ pstart, int64_t cap_rows,
                                                         int64_t* __restrict__ status)
{
    __shared__ long long wsum[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int s0 = 0; s0 < n_sb; s0 += 256) {
        const int s = s0 + threadIdx.x;
        const long long c = s < n_sb ? row_pad[s] : 0;
        long long inc = c;
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(inc, off, 64);
            if ((threadIdx.x & 63) >= off) inc += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long base = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        if (s < n_sb) pstart[s] = base + inc - c;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pstart[n_sb] = carry_s;
        status[0] = carry_s;
        status[1] = carry_s > cap_rows ? 1 : 0;
    }
}

// workgroups of the grouped launch beyond the kept pairs (all of them after an overflow) are idle: superblock id -1
__global__ __launch_bounds__(256) void rows_tail_kernel(const int64_t* __restrict__ status, int64_t cap_wgs,
                                                       int32_t* __restrict__ rblock_chunk)
{
    const int64_t first = status[1] ? 0 : status[0] / GROUP_ROWS;
    for (int64_t w = first + (int64_t)blockIdx.x * 256 + threadIdx.x; w < cap_wgs; w += (int64_t)gridDim.x * 256)
        rblock_chunk[w] = -1;
}

__global__ __launch_bounds__(256) void rows_fill_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                       int64_t stride, const float* __restrict__ floor_, int32_t n_ublk,
                                                       const int32_t* __restrict__ blockoff,
                                                       const int32_t* __restrict__ row_total,
                                                       const int64_t* __restrict__ pstart, int32_t* __restrict__ row_user,
                                                       int32_t* __restrict__ rblock_chunk, const int64_t* __restrict__ status)
{
    __shared__ int wsum[CROWS][4];
    if (status[1]) return;                                       // more pairs than row_user holds: nothing is refined
    const int32_t s0 = blockIdx.y * CROWS;
    const int64_t u = (int64_t)blockIdx.x * CUSERS + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int bits = tile_bits(table, n_sb, n_users, stride, floor_, s0, u);
    int pre[CROWS];
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        const int c = __builtin_popcount((bits >> (4 * r)) & 15u);
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        pre[r] = inc - c;
        if (lane == 63) wsum[r][wave] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        const int32_t s = s0 + r;
        if (s >= n_sb) break;
        const unsigned int m = (bits >> (4 * r)) & 15u;
        if (m) {
            int base = pre[r];
            for (int w = 0; w < wave; ++w) base += wsum[r][w];
            int64_t dst = pstart[s] + blockoff[(int64_t)s * n_ublk + blockIdx.x] + base;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((m >> e) & 1u) row_user[dst++] = (int32_t)(u + e);
        }
        if (blockIdx.x == 0) {                                   // the row's padding entries and its workgroups' superblock ids
            const int64_t p0 = pstart[s], p1 = pstart[s + 1];
            for (int64_t j = p0 + row_total[s] + threadIdx.x; j < p1; j += 256) row_user[j] = -1;
            for (int64_t w = p0 / GROUP_ROWS + threadIdx.x; w < p1 / GROUP_ROWS; w += 256) rblock_chunk[w] = s;
        }
    }
}

}  // namespace

extern "C" int trec_topk_filter_floor_i8(const float* tau, const float* user_stats, const float* user_bias,
                                         const float* item_gstats, int32_t kdim, int64_t n_users, float* floor_,
                                         int32_t* flag, int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(tau && user_stats && item_gstats && floor_, "trec_topk_filter_floor_i8: null pointer");
    TREC_REQUIRE(kdim >= 1, "trec_topk_filter_floor_i8: bad sizes");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(filter_floor_i8_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream,
                       tau, (const float2*)user_stats, user_bias, item_gstats, kdim, n_users, floor_, flag, n_flagged);
    return trec_check_launch("trec_topk_filter_floor_i8");
}

extern "C" int32_t trec_topk_rows_user_blocks(int64_t n_users) { return (int32_t)ceil_div64(n_users, CUSERS); }

// pass 1 of the row-wise compaction: block_off [n_sb][trec_topk_rows_user_blocks(n_users)] (scratch for pass 2),
// row_total [n_sb], pstart [n_sb + 1] (int64; pstart[n_sb] = resident rows of the grouped launch, a multiple of 512),
// status int64[2] = {pstart[n_sb], overflow: it exceeds cap_rows} -- nothing here needs the host
extern "C" int trec_topk_rows_count(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* floor_,
                                    int32_t* block_off, int32_t* row_total, int32_t* row_pad, int64_t* pstart,
                                    int64_t cap_rows, int64_t* status, void* stream)
{
    TREC_REQUIRE(table && floor_ && block_off && row_total && row_pad && pstart && status, "trec_topk_rows_count: null pointer");
    TREC_REQUIRE(cap_rows >= 0 && cap_rows % GROUP_ROWS == 0, "trec_topk_rows_count: cap_rows must be a multiple of 512");
    TREC_REQUIRE(n_sb >= 1 && n_users >= 1 && stride >= n_users, "trec_topk_rows_count: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int n_ublk = (int)ceil_div64(n_users, CUSERS);
    hipLaunchKernelGGL(rows_count_kernel, dim3((unsigned)n_ublk, (unsigned)((n_sb + CROWS - 1) / CROWS)), dim3(256), 0, st,
                       table, n_sb, n_users, stride, floor_, n_ublk, block_off);
    hipLaunchKernelGGL(rows_scan_kernel, dim3((unsigned)n_sb), dim3(256), 0, st, block_off, n_ublk, row_total, row_pad);
    hipLaunchKernelGGL(rows_pstart_kernel, dim3(1), dim3(256), 0, st, row_pad, n_sb, pstart, cap_rows, status);
    return trec_check_launch("trec_topk_rows_count");
}

// pass 2: row_user [cap_rows], first pstart[n_sb] entries = the kept users of superblock 0, padding (-1), those of
// superblock 1, ... (ascending user ids inside a superblock); rblock_chunk [cap_rows / 512] = the superblock of each
// 512-row workgroup, -1 for the workgroups beyond pstart[n_sb] / 512 (all of them when status[1] is set)
extern "C" int trec_topk_rows_fill(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* floor_,
                                   const int32_t* block_off, const int32_t* row_total, const int64_t* pstart,
                                   int64_t cap_rows, const int64_t* status, int32_t* row_user, int32_t* rblock_chunk,
                                   void* stream)
{
    TREC_REQUIRE(table && floor_ && block_off && row_total && pstart && status && row_user && rblock_chunk,
                 "trec_topk_rows_fill: null pointer");
    TREC_REQUIRE(cap_rows >= GROUP_ROWS && cap_rows % GROUP_ROWS == 0, "trec_topk_rows_fill: cap_rows must be a multiple of 512");
    TREC_REQUIRE(n_sb >= 1 && n_users >= 1 && stride >= n_users, "trec_topk_rows_fill: bad sizes");
    const int n_ublk = (int)ceil_div64(n_users, CUSERS);
    hipLaunchKernelGGL(rows_fill_kernel, dim3((unsigned)n_ublk, (unsigned)((n_sb + CROWS - 1) / CROWS)), dim3(256), 0,
                       (hipStream_t)stream, table, n_sb, n_users, stride, floor_, n_ublk, block_off, row_total, pstart,
                       row_user, rblock_chunk, status);
    const int64_t cap_wgs = cap_rows / GROUP_ROWS;
    unsigned tb = (unsigned)ceil_div64(cap_wgs, 256);
    if (tb > 1024) tb = 1024;
    hipLaunchKernelGGL(rows_tail_kernel, dim3(tb), dim3(256), 0, (hipStream_t)stream, status, cap_wgs, rblock_chunk);
    return trec_check_launch("trec_topk_rows_fill");
}

// bf16 superblock maxima of the kept (superblock, user) pairs, written over the table's entries:
// blockmax[rblock_chunk[w] * bm_stride + row_user[r]] for every resident row r of workgroup w = r / 512 with row_user[r] >= 0
extern "C" int trec_score_gemm_blockmax_grouped(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                                int64_t n_items, const float* user_bias, const float* item_bias,
                                                int32_t sb_rows, const int32_t* rblock_chunk, const int32_t* row_user,
                                                float* blockmax, int64_t bm_stride, void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && rblock_chunk && row_user && blockmax, "trec_score_gemm_blockmax_grouped: null pointer");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_blockmax_grouped: kpad must be 64 or 128");
    TREC_REQUIRE(n_rows_g % GROUP_ROWS == 0 && n_rows_g < ((int64_t)1 << 40), "trec_score_gemm_blockmax_grouped: n_rows_g % 512 != 0");
    TREC_REQUIRE(sb_rows >= 64 && sb_rows % 64 == 0, "trec_score_gemm_blockmax_grouped: sb_rows must be a multiple of 64");
    if (n_rows_g == 0) return TREC_OK;
    TREC_REQUIRE(n_rows_g / GROUP_ROWS < ((int64_t)1 << 31), "trec_score_gemm_blockmax_grouped: too many workgroups");
    ScoreParams p = {};
    p.R = users_bf16; p.T = items_bf16; p.n_r = n_rows_g; p.n_t = n_items;
    p.chunk_len = sb_rows; p.n_chunks = 1;
    p.r_bias = user_bias; p.t_bias = item_bias;
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / 64;
    p.rblock_chunk = rblock_chunk; p.row_index = row_user;
    return launch_blockmax_pipelined_grouped(p, kpad, (hipStream_t)stream);
}
