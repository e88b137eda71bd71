// tensorrec_amd/csrc/topk_cascade.hip -- K2c: the glue of the int8 -> bf16 -> fp32 cascade of the exact top-k.
//
//   stage 0  trec_score_gemm_blockmax_i8 (score_blockmax_i8.hip): table[s][u] = M8 = int8 maximum of superblock s for user u,
//            and -- kept in registers, written once per chunk of superblocks -- the k largest LOWER bounds M8 - e(u, s)
//   select   trec_topk_select_blocks over the chunks' lists (k rows per chunk, not the table): tauLB_u = the k-th largest
//            lower bound.  e(u, s) >= |int8 score - fp32 score| for every item of superblock s: proven from the MEASURED
//            quantisation error norms of trec_score_prep_i8, per user and per superblock (i8_pair_err, score_common.hpp)
//   compact  trec_topk_rows_count / trec_topk_rows_fill (here): the (superblock, user) pairs whose UPPER bound M8 + e(u, s)
//            reaches tauLB_u, already grouped by superblock -- the table is superblock-major, so a row-wise stream
//            compaction IS the grouping (no sort): per superblock a run of user ids padded with -1 to whole 512-row workgroups
//   stage 1  trec_score_gemm_blockmax_grouped (here + score_blockmax.hip): the hand-scheduled bf16 kernel on the kept pairs
//            only (~4% of them at 1M x 1M); each bf16 maximum REPLACES the int8 entry of the table
//   then the bf16 filter of topk_filter.hip runs unchanged on the mixed table (select, floor16, collect, bf16 lists, fp32).
//
// Nothing is lost: k superblocks have M8 - e >= tauLB, each holds an item with fp32 score >= tauLB, so the true k-th best
// t_k >= tauLB.  A top-k item has fp32 score >= t_k >= tauLB and int8 score >= fp32 - e, so its superblock has
// M8 + e >= tauLB: it is refined, and its table entry becomes a bf16 maximum M16 >= that item's fp32 score - eps16.
// The mixed table is sound for the bf16 filter: an UNrefined entry has M8 < tauLB - e(u, s).  The k certifying
// superblocks above are refined with M16 >= tauLB - eps16, so an unrefined entry with e(u, s) >= eps16 lies strictly below
// k refined entries and is never among the k largest of its column; one with e(u, s) < eps16 certifies an item with fp32
// score >= M8 - e > M8 - eps16 like a bf16 entry would.  Either way every one of the k largest entries certifies an item
// with fp32 score >= entry - eps16, which is all the argument of topk_filter.hip uses for tau16 and floor16 = tau16 - 2 eps16.
// Unrefined entries that pass floor16 are false positives: their superblocks are re-scored like any other.
//
// Replaces (as a filter) tf.matmul of tensorrec/prediction_graphs.py:49-50 + tf.nn.top_k of
// tensorrec/recommendation_graphs.py:80; the results come from the fp32 finish of topk_filter.hip, bit-identical to the oracle.
#include "score_common.hpp"
#include "topk_common.hpp"

namespace {

constexpr int CROWS = 8;          // table rows (superblocks) per group: one float4 load per row and thread in flight
constexpr int CGROUPS = 4;        // groups of rows per workgroup
constexpr int CUSERS = 1024;      // users per workgroup: one float4 per thread
constexpr int GROUP_ROWS = 512;   // resident rows per workgroup of the grouped bf16 kernel

// table rows s0 .. s0 + CROWS - 1, users u .. u + 3 of this thread: bit (4 r + e) of the result = the superblock's UPPER
// bound table[s0 + r][u + e] + e8 reaches thr[u + e] (the k-th largest LOWER bound, two floats down).  e8 here is
// nx A_s + ex B_s + a_u C_s + cu (C_s: the bias quantisation error per unit of user scale, sb_stats[s][3]) with the inflation of i8_pair_err folded into the per-row / per-user constants (rounded up):
// it need not equal the int8 kernel's evaluation bit for bit, both only have to dominate the true error.
struct UserConsts { float f[4], nx[4], ex[4], au[4]; unsigned int valid; };

__device__ __forceinline__ UserConsts load_user_consts(const float* __restrict__ thr, const float* __restrict__ user_err,
                                                       int64_t n_users, int64_t u)
{
    const float infl = 1.0029296875f;                      // 1 + 3 * 2^-10 > (1 + 2^-9) (1 + 2^-12): covers the re-association
    UserConsts c;
    c.valid = 0u;
    // the eight loads leave together, unconditionally, from clamped users (the empty asm keeps the compiler from sinking each
    // of them back into the branch that uses it -- four dependent thr -> user_err round trips per workgroup otherwise)
    float thv[4];
    f32x4 uvv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t ue_i = u + e < n_users ? u + e : n_users - 1;
        thv[e] = thr[ue_i];
        uvv[e] = *(const f32x4*)(user_err + ue_i * 4);                     // {||x||, ||x - a q||, cu, a}
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(thv[e]), "+v"(uvv[e]));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // a threshold of +inf keeps nothing WHATEVER the table holds: rows without a source (band padding, the idle int8
        // workgroups of trec_user_prep_sorted, whose table entries were never written) and users flagged before the compaction
        const bool in = u + e < n_users && thv[e] < INFINITY;
        const float th = thv[e];
        const f32x4 uv = uvv[e];
        if (in) c.valid |= 1u << e;
        const f32x4 ue = in ? uv : (f32x4){0.f, 0.f, 0.f, 0.f};
        c.nx[e] = ue[0];
        c.ex[e] = ue[1];
        c.au[e] = ue[3];
        const float cu = in ? ue[2] * infl + 2e-30f : 0.f;
        c.f[e] = in ? float_pred(float_pred(th)) - cu : INFINITY;                 // v + nx A + ex B + C >= thr - cu
        if (in) c.f[e] = float_pred(c.f[e]);                                      // the subtraction may have rounded up
    }
    return c;
}

__device__ __forceinline__ unsigned int tile_bits(const float* __restrict__ table, int32_t n_sb, int64_t n_users, int64_t stride,
                                                  const UserConsts& c, const float* __restrict__ sb_stats, int kdim,
                                                  int32_t s0, int64_t u)
{
    const float infl = 1.0029296875f;
    const float ck = (float)(kdim + 6) * 2.98023224e-07f;
    const bool vec = (stride % 4 == 0) && (((uintptr_t)table % 16) == 0) && (u + 3 < stride);
    unsigned int bits = 0;
    // all CROWS row loads leave together, from clamped (always valid) rows, before the first comparison: with the loads inside
    // `if (s < n_sb)` every row was its own block ending in s_waitcnt vmcnt(0) -- eight serial round trips per group
    f32x4 v[CROWS], ss[CROWS];
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        const int32_t s = s0 + r < n_sb ? s0 + r : n_sb - 1;
        const float* src = table + (int64_t)s * stride + u;
        if (vec) v[r] = __builtin_nontemporal_load((const f32x4*)src);
        else {
            v[r] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int e = 0; e < 4; ++e) if (u + e < n_users) v[r][e] = src[e];
        }
        ss[r] = *(const f32x4*)(sb_stats + (int64_t)s * 4);
    }
#pragma unroll
    for (int r = 0; r < CROWS; ++r) {
        const int32_t s = s0 + r;
        const float A = (ss[r][2] + ck * ss[r][1]) * infl, B = ss[r][1] * infl, C = ss[r][3] * infl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (s < n_sb && ((c.valid >> e) & 1u) && !(__fmaf_rn(c.nx[e], A, __fmaf_rn(c.ex[e], B, __fmaf_rn(c.au[e], C, v[r][e]))) < c.f[e]))
                bits |= 1u << (4 * r + e);
    }
    return bits;
}

// A workgroup owns 1024 users x CGROUPS groups of CROWS table rows: the users' constants are loaded once (per 8-row group
// they were half as many bytes again as the table itself).
__global__ __launch_bounds__(256) void rows_count_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                        int64_t stride, const float* __restrict__ thr,
                                                        const float* __restrict__ user_err,
                                                        const float* __restrict__ sb_stats, int kdim, int32_t n_ublk,
                                                        int32_t* __restrict__ blockcnt)
{
    __shared__ int cnt[CGROUPS][CROWS];
    if (threadIdx.x < CGROUPS * CROWS) (&cnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * CUSERS + threadIdx.x * 4;
    const UserConsts c = load_user_consts(thr, user_err, n_users, u);
    for (int g = 0; g < CGROUPS; ++g) {
        const int32_t s0 = (blockIdx.y * CGROUPS + g) * CROWS;
        if (s0 >= n_sb) break;
        const unsigned int bits = tile_bits(table, n_sb, n_users, stride, c, sb_stats, kdim, s0, u);
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            int k = __builtin_popcount((bits >> (4 * r)) & 15u);
            for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off, 64);
            if ((threadIdx.x & 63) == 0 && k) atomicAdd(&cnt[g][r], k);
        }
    }
    __syncthreads();
    if (threadIdx.x < CGROUPS * CROWS) {
        const int32_t s = blockIdx.y * CGROUPS * CROWS + threadIdx.x;
        if (s < n_sb) blockcnt[(int64_t)s * n_ublk + blockIdx.x] = (&cnt[0][0])[threadIdx.x];
    }
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
                                                       int64_t stride, const float* __restrict__ thr,
                                                       const float* __restrict__ user_err,
                                                       const float* __restrict__ sb_stats, int kdim, int32_t n_ublk,
                                                       const int32_t* __restrict__ blockoff,
                                                       const int32_t* __restrict__ row_total,
                                                       const int64_t* __restrict__ pstart, int32_t* __restrict__ row_user,
                                                       int32_t* __restrict__ rblock_chunk, const int64_t* __restrict__ status)
{
    __shared__ int wsum[2][CROWS][4];
    if (status[1]) return;                                       // more pairs than row_user holds: nothing is refined
    const int64_t u = (int64_t)blockIdx.x * CUSERS + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const UserConsts c = load_user_consts(thr, user_err, n_users, u);
    for (int g = 0; g < CGROUPS; ++g) {
        const int32_t s0 = (blockIdx.y * CGROUPS + g) * CROWS;
        if (s0 >= n_sb) break;
        const unsigned int bits = tile_bits(table, n_sb, n_users, stride, c, sb_stats, kdim, s0, u);
        int pre[CROWS];
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int k = __builtin_popcount((bits >> (4 * r)) & 15u);
            int inc = k;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            pre[r] = inc - k;
            if (lane == 63) wsum[g & 1][r][wave] = inc;
        }
        __syncthreads();                                         // (two alternating buffers: one barrier per group is enough)
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int32_t s = s0 + r;
            if (s >= n_sb) break;
            const unsigned int m = (bits >> (4 * r)) & 15u;
            if (m) {
                int base = pre[r];
                for (int w = 0; w < wave; ++w) base += wsum[g & 1][r][w];
                int64_t dst = pstart[s] + blockoff[(int64_t)s * n_ublk + blockIdx.x] + base;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((m >> e) & 1u) row_user[dst++] = (int32_t)(u + e);
            }
            if (blockIdx.x == 0) {                               // the row's padding entries and its workgroups' superblock ids
                const int64_t p0 = pstart[s], p1 = pstart[s + 1];
                for (int64_t j = p0 + row_total[s] + threadIdx.x; j < p1; j += 256) row_user[j] = -1;
                for (int64_t w = p0 / GROUP_ROWS + threadIdx.x; w < p1 / GROUP_ROWS; w += 256) rblock_chunk[w] = s;
            }
        }
    }
}

// ---- the same compaction in ONE pass over the table: fixed capacity per superblock --------------------------------------
// row_user is [n_sb][rcap]; a workgroup (1024 users x 32 rows) takes, per row, a run of slots with ONE atomicAdd on the row's
// counter and writes its kept users there (ascending inside the run; the order of the runs of different workgroups follows
// the atomics -- which users share a workgroup of the bf16 stage changes from run to run, no result does).  Slots beyond
// rcap are dropped and row_count[s] > rcap tells (rows_status_kernel) that the call must fall back.
__global__ __launch_bounds__(256) void rows_collect_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                          int64_t stride, const float* __restrict__ thr,
                                                          const float* __restrict__ user_err,
                                                          const float* __restrict__ sb_stats, int kdim, int32_t rcap,
                                                          int32_t* __restrict__ row_count, int32_t* __restrict__ row_user, int rows_fast)
{
    __shared__ int wsum[2][CROWS][4];
    __shared__ int base_s[2][CROWS];
    // rows_fast: consecutive workgroups take DIFFERENT row groups of one user block (grid x = row groups): the ~2,000 workgroups in
    // flight then spread their slot atomics over every row counter instead of queueing on the 32 counters of one row group
    const unsigned ublk = rows_fast ? blockIdx.y : blockIdx.x, rgrp = rows_fast ? blockIdx.x : blockIdx.y;
    const int64_t u = (int64_t)ublk * CUSERS + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const UserConsts c = load_user_consts(thr, user_err, n_users, u);
    for (int g = 0; g < CGROUPS; ++g) {
        const int32_t s0 = (rgrp * CGROUPS + g) * CROWS;
        if (s0 >= n_sb) break;
        const unsigned int bits = tile_bits(table, n_sb, n_users, stride, c, sb_stats, kdim, s0, u);
        int pre[CROWS];
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int k = __builtin_popcount((bits >> (4 * r)) & 15u);
            int inc = k;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            pre[r] = inc - k;
            if (lane == 63) wsum[g & 1][r][wave] = inc;
        }
        __syncthreads();
        if (threadIdx.x < CROWS && s0 + threadIdx.x < n_sb) {
            const int total = wsum[g & 1][threadIdx.x][0] + wsum[g & 1][threadIdx.x][1] + wsum[g & 1][threadIdx.x][2] +
                              wsum[g & 1][threadIdx.x][3];
            base_s[g & 1][threadIdx.x] = total ? atomicAdd(&row_count[s0 + threadIdx.x], total) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int32_t s = s0 + r;
            if (s >= n_sb) break;
            const unsigned int m = (bits >> (4 * r)) & 15u;
            if (m) {
                int idx = base_s[g & 1][r] + pre[r];
                for (int w = 0; w < wave; ++w) idx += wsum[g & 1][r][w];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((m >> e) & 1u) {
                        if (idx < rcap) row_user[(int64_t)s * rcap + idx] = (int32_t)(u + e);
                        ++idx;
                    }
            }
        }
    }
}

// The same pass without any workgroup barrier: every WAVE (256 users) takes its own run of slots per row -- the eight rows' wave
// totals sit in lanes 0..7, which issue the eight atomics together -- so the waves of a workgroup drift apart and one wave's loads
// overlap another's atomics and stores (the barrier form holds a workgroup's four waves in step: load, scan, atomic, store).
// Runs are ~4 users long instead of ~15; the list order inside a superblock is as undetermined as before.
__global__ __launch_bounds__(256) void rows_collect_wave_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                               int64_t stride, const float* __restrict__ thr,
                                                               const float* __restrict__ user_err,
                                                               const float* __restrict__ sb_stats, int kdim, int32_t rcap,
                                                               int32_t* __restrict__ row_count, int32_t* __restrict__ row_user)
{
    const int64_t u = (int64_t)blockIdx.y * CUSERS + threadIdx.x * 4;
    const int lane = threadIdx.x & 63;
    const UserConsts c = load_user_consts(thr, user_err, n_users, u);
    for (int g = 0; g < CGROUPS; ++g) {
        const int32_t s0 = (blockIdx.x * CGROUPS + g) * CROWS;
        if (s0 >= n_sb) break;
        const unsigned int bits = tile_bits(table, n_sb, n_users, stride, c, sb_stats, kdim, s0, u);
        if (__builtin_amdgcn_ballot_w64(bits != 0u) == 0ull) continue;      // (wave-uniform: nothing kept in these 8 rows x 256 users)
        int pre[CROWS];
        int mine = 0;
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int k = __builtin_popcount((bits >> (4 * r)) & 15u);
            int inc = k;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            pre[r] = inc - k;
            const int tot = __shfl(inc, 63, 64);
            if (lane == r) mine = tot;
        }
        int base = 0;
        if (lane < CROWS && mine > 0 && s0 + lane < n_sb) base = atomicAdd(&row_count[s0 + lane], mine);
#pragma unroll
        for (int r = 0; r < CROWS; ++r) {
            const int32_t s = s0 + r;
            const int b = __shfl(base, r, 64);
            if (s >= n_sb) break;
            const unsigned int m = (bits >> (4 * r)) & 15u;
            if (m) {
                int idx = b + pre[r];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((m >> e) & 1u) {
                        if (idx < rcap) row_user[(int64_t)s * rcap + idx] = (int32_t)(u + e);
                        ++idx;
                    }
            }
        }
    }
}

// Users the int8 bound says nothing about (a row far smaller than its scale class, a row of outliers): nearly every superblock
// reaches their threshold, so every superblock would be refined for them and -- with the candidate lists -- most items listed,
// only for the finish kernel to flag them anyway.  32 sampled table rows (4 groups of CROWS, evenly spaced) decide: ``limit`` or
// more kept -> cand_floor = +inf (the refining launches list nothing for the user) and the user is flagged now.  A heuristic
// for speed only: flagged users are re-done exactly by the caller.
__global__ __launch_bounds__(256) void dense_users_kernel(const float* __restrict__ table, int32_t n_sb, int64_t n_users,
                                                         int64_t stride, const float* __restrict__ thr,
                                                         const float* __restrict__ user_err,
                                                         const float* __restrict__ sb_stats, int kdim, int32_t limit,
                                                         float* __restrict__ cand_floor, int32_t* __restrict__ flag,
                                                         int32_t* __restrict__ n_flagged)
{
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (u >= n_users) return;
    const UserConsts c = load_user_consts(thr, user_err, n_users, u);
    int cnt[4] = {0, 0, 0, 0};
    const int32_t groups = n_sb / CROWS;
    for (int j = 0; j < 4; ++j) {
        const int32_t s0 = (int32_t)(((int64_t)groups * j) / 4) * CROWS;
        const unsigned int bits = tile_bits(table, n_sb, n_users, stride, c, sb_stats, kdim, s0, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) cnt[e] += __builtin_popcount(bits & (0x11111111u << e));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (u + e < n_users && cnt[e] >= limit && cand_floor[u + e] < INFINITY) {
            cand_floor[u + e] = INFINITY;
            if (flag[u + e] == 0) { flag[u + e] = 1; atomicAdd(n_flagged, 1); }
        }
}

// status[0] = resident rows the grouped launch works on (every row's count, capped, rounded up to whole workgroups),
// status[1] = 1 when some superblock kept more users than rcap
__global__ __launch_bounds__(256) void rows_status_kernel(const int32_t* __restrict__ row_count, int32_t n_sb, int32_t rcap,
                                                         int64_t* __restrict__ status)
{
    __shared__ long long tot[4];
    __shared__ int over[4];
    long long t = 0;
    int o = 0;
    for (int s = threadIdx.x; s < n_sb; s += 256) {
        const int c = row_count[s];
        o |= c > rcap;
        t += ((c < rcap ? c : rcap) + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
    }
    for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); o |= __shfl_xor(o, off, 64); }
    if ((threadIdx.x & 63) == 0) { tot[threadIdx.x >> 6] = t; over[threadIdx.x >> 6] = o; }
    __syncthreads();
    if (threadIdx.x == 0) {
        status[0] = tot[0] + tot[1] + tot[2] + tot[3];
        status[1] = (over[0] | over[1] | over[2] | over[3]) ? 1 : 0;
    }
}

// "Hot" superblocks: rows kept by more users than the fixed capacity holds (a few very popular items are wanted by most
// users although only 2-3% of ALL pairs are kept: fitted models, Zipf catalogues).  They are listed -- ascending, -1 padded --
// for a dense pass over every user (trec_score_gemm_blockmax_hot) and their counts are zeroed so that the fixed-capacity
// grouped launch skips them; the call only fails (status[1]) when there are more than hot_cap of them or when both launches
// together would refine more than max_pairs (superblock, user) pairs (the int8 bound is too loose to pay).
// status[0] = resident rows both launches work on.  Single workgroup.
__global__ __launch_bounds__(256) void rows_hot_kernel(int32_t* __restrict__ row_count, int32_t n_sb, int32_t rcap,
                                                      int64_t n_users, int32_t* __restrict__ hot_list, int32_t hot_cap,
                                                      int64_t max_pairs, int64_t* __restrict__ status)
{
    __shared__ int wcnt[4];
    __shared__ int base_s;
    __shared__ long long tot[4], totp[4];
    if (threadIdx.x == 0) base_s = 0;
    long long t = 0, tp = 0;                                    // padded resident rows / kept pairs of the rows that are not hot
    __syncthreads();
    for (int s0 = 0; s0 < n_sb; s0 += 256) {
        const int s = s0 + threadIdx.x;
        const int c = s < n_sb ? row_count[s] : 0;
        const bool hot = c > rcap;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hot);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wcnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int idx = base_s + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) idx += wcnt[w];
        if (hot) {
            if (idx < hot_cap) hot_list[idx] = s;
            row_count[s] = 0;
        } else {
            t += (c + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
            tp += c;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); tp += __shfl_xor(tp, off, 64); }
    if ((threadIdx.x & 63) == 0) { tot[threadIdx.x >> 6] = t; totp[threadIdx.x >> 6] = tp; }
    __syncthreads();
    const int n_hot = base_s;
    for (int j = n_hot + threadIdx.x; j < hot_cap; j += 256) hot_list[j] = -1;
    if (threadIdx.x == 0) {
        const long long per_hot = (n_users + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        const long long nh = n_hot < hot_cap ? n_hot : hot_cap;
        status[0] = tot[0] + tot[1] + tot[2] + tot[3] + nh * per_hot;
        const long long pairs = totp[0] + totp[1] + totp[2] + totp[3] + (long long)n_hot * n_users;
        status[1] = (n_hot > hot_cap || pairs > max_pairs) ? 1 : 0;
        status[2] = n_hot;
    }
}

}  // namespace

extern "C" int32_t trec_topk_rows_user_blocks(int64_t n_users) { return (int32_t)ceil_div64(n_users, CUSERS); }

// pass 1 of the row-wise compaction: block_off [n_sb][trec_topk_rows_user_blocks(n_users)] (scratch for pass 2),
// row_total [n_sb], pstart [n_sb + 1] (int64; pstart[n_sb] = resident rows of the grouped launch, a multiple of 512),
// status int64[2] = {pstart[n_sb], overflow: it exceeds cap_rows} -- nothing here needs the host
extern "C" int trec_topk_rows_count(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                                    const float* user_err, const float* sb_stats, int32_t kdim, int32_t* block_off,
                                    int32_t* row_total, int32_t* row_pad, int64_t* pstart, int64_t cap_rows,
                                    int64_t* status, void* stream)
{
    TREC_REQUIRE(table && thr && user_err && sb_stats && block_off && row_total && row_pad && pstart && status,
                 "trec_topk_rows_count: null pointer");
    TREC_REQUIRE(cap_rows >= 0 && cap_rows % GROUP_ROWS == 0, "trec_topk_rows_count: cap_rows must be a multiple of 512");
    TREC_REQUIRE(n_sb >= 1 && n_users >= 1 && stride >= n_users, "trec_topk_rows_count: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int n_ublk = (int)ceil_div64(n_users, CUSERS);
    hipLaunchKernelGGL(rows_count_kernel, dim3((unsigned)n_ublk, (unsigned)((n_sb + CROWS * CGROUPS - 1) / (CROWS * CGROUPS))), dim3(256), 0, st,
                       table, n_sb, n_users, stride, thr, user_err, sb_stats, kdim, n_ublk, block_off);
    hipLaunchKernelGGL(rows_scan_kernel, dim3((unsigned)n_sb), dim3(256), 0, st, block_off, n_ublk, row_total, row_pad);
    hipLaunchKernelGGL(rows_pstart_kernel, dim3(1), dim3(256), 0, st, row_pad, n_sb, pstart, cap_rows, status);
    return trec_check_launch("trec_topk_rows_count");
}

// pass 2: row_user [cap_rows], first pstart[n_sb] entries = the kept users of superblock 0, padding (-1), those of
// superblock 1, ... (ascending user ids inside a superblock); rblock_chunk [cap_rows / 512] = the superblock of each
// 512-row workgroup, -1 for the workgroups beyond pstart[n_sb] / 512 (all of them when status[1] is set)
extern "C" int trec_topk_rows_fill(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                                   const float* user_err, const float* sb_stats, int32_t kdim, const int32_t* block_off,
                                   const int32_t* row_total, const int64_t* pstart, int64_t cap_rows, const int64_t* status,
                                   int32_t* row_user, int32_t* rblock_chunk, void* stream)
{
    TREC_REQUIRE(table && thr && user_err && sb_stats && block_off && row_total && pstart && status && row_user && rblock_chunk,
                 "trec_topk_rows_fill: null pointer");
    TREC_REQUIRE(cap_rows >= GROUP_ROWS && cap_rows % GROUP_ROWS == 0, "trec_topk_rows_fill: cap_rows must be a multiple of 512");
    TREC_REQUIRE(n_sb >= 1 && n_users >= 1 && stride >= n_users, "trec_topk_rows_fill: bad sizes");
    const int n_ublk = (int)ceil_div64(n_users, CUSERS);
    hipLaunchKernelGGL(rows_fill_kernel, dim3((unsigned)n_ublk, (unsigned)((n_sb + CROWS * CGROUPS - 1) / (CROWS * CGROUPS))), dim3(256), 0,
                       (hipStream_t)stream, table, n_sb, n_users, stride, thr, user_err, sb_stats, kdim, n_ublk, block_off,
                       row_total, pstart, row_user, rblock_chunk, status);
    const int64_t cap_wgs = cap_rows / GROUP_ROWS;
    unsigned tb = (unsigned)ceil_div64(cap_wgs, 256);
    if (tb > 1024) tb = 1024;
    hipLaunchKernelGGL(rows_tail_kernel, dim3(tb), dim3(256), 0, (hipStream_t)stream, status, cap_wgs, rblock_chunk);
    return trec_check_launch("trec_topk_rows_fill");
}

// The one-pass form of the compaction: row_user [n_sb][rcap] (rcap a multiple of 512), row_count [n_sb] zero-initialised by
// the caller (ends as the number of users kept per superblock, possibly above rcap), status int64[2] = {resident rows,
// overflow}.  The order of a superblock's users is not deterministic (runs of ascending ids in atomic order).
extern "C" int trec_topk_rows_collect(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                                      const float* user_err, const float* sb_stats, int32_t kdim, int32_t rcap,
                                      int32_t* row_count, int32_t* row_user, int64_t* status, void* stream)
{
    TREC_REQUIRE(table && thr && user_err && sb_stats && row_count && row_user && status, "trec_topk_rows_collect: null pointer");
    TREC_REQUIRE(n_sb >= 1 && n_users >= 1 && stride >= n_users, "trec_topk_rows_collect: bad sizes");
    TREC_REQUIRE(rcap >= GROUP_ROWS && rcap % GROUP_ROWS == 0, "trec_topk_rows_collect: rcap must be a multiple of 512");
    hipStream_t st = (hipStream_t)stream;
    const int n_ublk = (int)ceil_div64(n_users, CUSERS);
    const unsigned n_rgrp = (unsigned)((n_sb + CROWS * CGROUPS - 1) / (CROWS * CGROUPS));
    const int rows_fast = trec_get_tuning("rows_collect_rows_fast", 1) != 0 && n_ublk <= 65535;
    if (rows_fast && trec_get_tuning("rows_collect_per_wave", 1) != 0)
        hipLaunchKernelGGL(rows_collect_wave_kernel, dim3(n_rgrp, (unsigned)n_ublk), dim3(256), 0, st, table, n_sb, n_users, stride, thr,
                           user_err, sb_stats, kdim, rcap, row_count, row_user);
    else
    hipLaunchKernelGGL(rows_collect_kernel, rows_fast ? dim3(n_rgrp, (unsigned)n_ublk) : dim3((unsigned)n_ublk, n_rgrp),
                       dim3(256), 0, st, table, n_sb, n_users, stride, thr, user_err, sb_stats, kdim, rcap, row_count, row_user, rows_fast);
    hipLaunchKernelGGL(rows_status_kernel, dim3(1), dim3(256), 0, st, row_count, n_sb, rcap, status);
    return trec_check_launch("trec_topk_rows_collect");
}

// After trec_topk_rows_collect: superblocks whose count exceeds rcap go to hot_list [hot_cap] (ascending, -1 padded) and
// their row_count becomes 0 (the fixed-capacity grouped launch skips them; trec_score_gemm_blockmax_hot refines them for every
// user).  status = {resident rows of both launches, 1 when more than hot_cap superblocks are hot or more than max_pairs pairs
// would be refined (the caller falls back to the dense bf16 stage 1)}.
extern "C" int trec_topk_rows_hot(int32_t* row_count, int32_t n_sb, int32_t rcap, int64_t n_users, int32_t* hot_list,
                                  int32_t hot_cap, int64_t max_pairs, int64_t* status, void* stream)
{
    TREC_REQUIRE(row_count && hot_list && status, "trec_topk_rows_hot: null pointer");
    TREC_REQUIRE(n_sb >= 1 && rcap >= 1 && hot_cap >= 1 && n_users >= 1, "trec_topk_rows_hot: bad sizes");
    hipLaunchKernelGGL(rows_hot_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_count, n_sb, rcap, n_users, hot_list,
                       hot_cap, max_pairs, status);
    return trec_check_launch("trec_topk_rows_hot");
}

// bf16 maxima of the listed superblocks for EVERY user, written over the table's entries: the dense bf16 filter kernel
// (v_mfma_f32_16x16x32_bf16) with chunk c = superblock hot_list[c]; workgroups of -1 entries exit at once.
extern "C" int trec_score_gemm_blockmax_hot(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_users,
                                            int64_t n_items, const float* user_bias, const float* item_bias, int32_t sb_rows,
                                            const int32_t* hot_list, int32_t hot_cap, float* blockmax, int64_t bm_stride,
                                            void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && hot_list && blockmax, "trec_score_gemm_blockmax_hot: null pointer");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_blockmax_hot: kpad must be 64 or 128");
    TREC_REQUIRE(sb_rows >= 64 && sb_rows % 64 == 0 && hot_cap >= 1 && bm_stride >= n_users, "trec_score_gemm_blockmax_hot: bad sizes");
    if (n_users == 0 || n_items == 0) return TREC_OK;
    ScoreParams p = {};
    p.R = users_bf16; p.T = items_bf16; p.n_r = n_users; p.n_t = n_items;
    p.chunk_len = sb_rows; p.n_chunks = hot_cap;
    p.r_bias = user_bias; p.t_bias = item_bias;
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / 64;
    p.rblock_chunk = hot_list;
    return launch_blockmax_filter16(p, kpad, (hipStream_t)stream);
}

// bf16 superblock maxima of the kept (superblock, user) pairs, written over the table's entries:
// blockmax[rblock_chunk[w] * bm_stride + row_user[r]] for every resident row r of workgroup w = r / 512 with row_user[r] >= 0
extern "C" int trec_score_gemm_blockmax_grouped(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                                int64_t n_items, const float* user_bias, const float* item_bias,
                                                int32_t sb_rows, const int32_t* rblock_chunk, const int32_t* row_user,
                                                float* blockmax, int64_t bm_stride, int32_t wgs_per_row, void* stream)
{
    // wgs_per_row > 0: the fixed-capacity layout of trec_topk_rows_collect -- row_user is [n_sb][wgs_per_row * 512],
    // rblock_chunk is row_count [n_sb]; workgroup w works on superblock w / wgs_per_row and exits when its 512 rows lie beyond
    // the superblock's count
    TREC_REQUIRE(wgs_per_row >= 0, "trec_score_gemm_blockmax_grouped: wgs_per_row < 0");
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
    p.rblock_chunk = rblock_chunk; p.row_index = row_user; p.capacity = wgs_per_row;
    p.grp_band_major = trec_get_tuning("cascade_band_major", 0);      // measured slower (8.25 vs 6.83 ms at 1M x 1M): see DESIGN 5d
    return launch_blockmax_pipelined_grouped(p, kpad, (hipStream_t)stream);
}

// ---- the refining launches that also LIST -------------------------------------------------------------------
// Same launches as trec_score_gemm_blockmax_grouped (fixed-capacity layout) / trec_score_gemm_blockmax_hot, with the kernel's
// LIST form: besides writing the bf16 maxima over the table's entries, every item of a refined (superblock, user) pair whose
// bf16 score (+ user bias) reaches cand_floor[user] is appended to cand[user][slot] = {item id + item_index_base, score bits},
// slot = atomicAdd(cand_n[user], 1) (entries beyond cand_cap are dropped, the count keeps growing: trec_topk_candidates_finish
// flags such a user).  cand_n must be zero before the first of the two launches.
//
// Why the lists are enough (tauLB = the k-th largest int8 lower bound, eps = the bf16 filter's bound of topk_filter.hip,
// cand_floor = tauLB - eps rounded down, sh = the bf16-path score, s = the fp32 score): a top-k item has s >= t_k >= tauLB, so
// its superblock is refined (header above) and sh >= s - eps >= cand_floor: it is listed.  The list S therefore holds k items;
// tau = the k-th largest sh in S; the k items at or above it have s >= tau - eps, hence t_k >= tau - eps, and a top-k item has
// sh >= t_k - eps >= tau - 2 eps: it survives the finish kernel's floor and is re-scored exactly.  Item shards: tauLB is the
// exchanged one, tau stays local (k local candidates certify k items of the whole catalogue just as well).
extern "C" int trec_score_gemm_refine_candidates(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                                 int64_t n_items, const float* user_bias, const float* item_bias,
                                                 int32_t sb_rows, const int32_t* row_count, const int32_t* row_user,
                                                 float* blockmax, int64_t bm_stride, int32_t wgs_per_row,
                                                 const float* cand_floor, int32_t* cand_n, void* cand, int32_t cand_cap,
                                                 int32_t item_index_base, const int32_t* wg_map, int32_t n_wgs, void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && row_count && row_user && blockmax && cand_floor && cand_n && cand,
                 "trec_score_gemm_refine_candidates: null pointer");
    TREC_REQUIRE(n_wgs >= 0 && (wg_map || n_wgs == 0), "trec_score_gemm_refine_candidates: n_wgs without wg_map");
    if (wg_map && n_wgs == 0) return TREC_OK;
    TREC_REQUIRE(wgs_per_row > 0 && cand_cap >= 1, "trec_score_gemm_refine_candidates: needs the fixed-capacity layout (wgs_per_row > 0)");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_refine_candidates: kpad must be 64 or 128");
    TREC_REQUIRE(n_rows_g % GROUP_ROWS == 0 && n_rows_g < ((int64_t)1 << 40), "trec_score_gemm_refine_candidates: n_rows_g % 512 != 0");
    TREC_REQUIRE(sb_rows >= 64 && sb_rows % 64 == 0 && sb_rows <= 65536, "trec_score_gemm_refine_candidates: sb_rows must be a multiple of 64, <= 65536");
    TREC_REQUIRE(trec_get_tuning("blockmax_bf16_mfma16", 1) != 0, "trec_score_gemm_refine_candidates: needs the 16x16x32 kernel (tuning blockmax_bf16_mfma16)");
    if (n_rows_g == 0) return TREC_OK;
    TREC_REQUIRE(n_rows_g / GROUP_ROWS < ((int64_t)1 << 31), "trec_score_gemm_refine_candidates: too many workgroups");
    ScoreParams p = {};
    p.R = users_bf16; p.T = items_bf16; p.n_r = n_rows_g; p.n_t = n_items;
    p.chunk_len = sb_rows; p.n_chunks = 1;
    p.r_bias = user_bias; p.t_bias = item_bias;
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / 64;
    p.rblock_chunk = row_count; p.row_index = row_user; p.capacity = wgs_per_row;
    p.cand_floor = cand_floor; p.cand_n = cand_n; p.cand = (int2*)cand; p.cand_cap = cand_cap; p.t_index_base = item_index_base;
    p.cand_diag = trec_get_tuning("cascade_cand_diag", 0);
    p.wg_map = wg_map; p.n_wgs = n_wgs;
    return launch_blockmax_pipelined_grouped(p, kpad, (hipStream_t)stream);
}

// The PRE-refining launch in its marking form: as trec_score_gemm_refine_candidates, but the bf16 maximum of list entry r (row_user[r])
// goes to pre_max[r] (float [n_rows_g], the list's layout) and the table entry becomes -inf at once; trec_topk_prerefine_tau_listed
// reads the maxima back by list position.
extern "C" int trec_score_gemm_refine_candidates_marked(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_rows_g,
                                                        int64_t n_items, const float* user_bias, const float* item_bias,
                                                        int32_t sb_rows, const int32_t* row_count, const int32_t* row_user,
                                                        float* blockmax, int64_t bm_stride, int32_t wgs_per_row,
                                                        const float* cand_floor, int32_t* cand_n, void* cand, int32_t cand_cap,
                                                        int32_t item_index_base, const int32_t* wg_map, int32_t n_wgs, float* pre_max,
                                                        void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && row_count && row_user && blockmax && cand_floor && cand_n && cand && pre_max,
                 "trec_score_gemm_refine_candidates_marked: null pointer");
    TREC_REQUIRE(n_wgs >= 0 && (wg_map || n_wgs == 0), "trec_score_gemm_refine_candidates_marked: n_wgs without wg_map");
    if (wg_map && n_wgs == 0) return TREC_OK;
    TREC_REQUIRE(wgs_per_row > 0 && cand_cap >= 1, "trec_score_gemm_refine_candidates_marked: needs the fixed-capacity layout (wgs_per_row > 0)");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_refine_candidates_marked: kpad must be 64 or 128");
    TREC_REQUIRE(n_rows_g % GROUP_ROWS == 0 && n_rows_g < ((int64_t)1 << 40), "trec_score_gemm_refine_candidates_marked: n_rows_g % 512 != 0");
    TREC_REQUIRE(sb_rows >= 64 && sb_rows % 64 == 0 && sb_rows <= 65536, "trec_score_gemm_refine_candidates_marked: sb_rows must be a multiple of 64, <= 65536");
    TREC_REQUIRE(trec_get_tuning("blockmax_bf16_mfma16", 1) != 0, "trec_score_gemm_refine_candidates_marked: needs the 16x16x32 kernel (tuning blockmax_bf16_mfma16)");
    if (n_rows_g == 0) return TREC_OK;
    TREC_REQUIRE(n_rows_g / GROUP_ROWS < ((int64_t)1 << 31), "trec_score_gemm_refine_candidates_marked: too many workgroups");
    ScoreParams p = {};
    p.R = users_bf16; p.T = items_bf16; p.n_r = n_rows_g; p.n_t = n_items;
    p.chunk_len = sb_rows; p.n_chunks = 1;
    p.r_bias = user_bias; p.t_bias = item_bias;
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / 64;
    p.rblock_chunk = row_count; p.row_index = row_user; p.capacity = wgs_per_row;
    p.cand_floor = cand_floor; p.cand_n = cand_n; p.cand = (int2*)cand; p.cand_cap = cand_cap; p.t_index_base = item_index_base;
    p.cand_diag = trec_get_tuning("cascade_cand_diag", 0);
    p.wg_map = wg_map; p.n_wgs = n_wgs;
    p.pre_max = pre_max;
    return launch_blockmax_pipelined_grouped(p, kpad, (hipStream_t)stream);
}

extern "C" int trec_score_gemm_refine_candidates_hot(const void* users_bf16, const void* items_bf16, int32_t kpad, int64_t n_users,
                                                     int64_t n_items, const float* user_bias, const float* item_bias,
                                                     int32_t sb_rows, const int32_t* hot_list, int32_t hot_cap, float* blockmax,
                                                     int64_t bm_stride, const float* cand_floor, int32_t* cand_n, void* cand,
                                                     int32_t cand_cap, int32_t item_index_base, void* stream)
{
    TREC_REQUIRE(users_bf16 && items_bf16 && hot_list && blockmax && cand_floor && cand_n && cand,
                 "trec_score_gemm_refine_candidates_hot: null pointer");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_refine_candidates_hot: kpad must be 64 or 128");
    TREC_REQUIRE(sb_rows >= 64 && sb_rows % 64 == 0 && sb_rows <= 65536 && hot_cap >= 1 && bm_stride >= n_users && cand_cap >= 1,
                 "trec_score_gemm_refine_candidates_hot: bad sizes");
    if (n_users == 0 || n_items == 0) return TREC_OK;
    ScoreParams p = {};
    p.R = users_bf16; p.T = items_bf16; p.n_r = n_users; p.n_t = n_items;
    p.chunk_len = sb_rows; p.n_chunks = hot_cap;
    p.r_bias = user_bias; p.t_bias = item_bias;
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / 64;
    p.rblock_chunk = hot_list;
    p.cand_floor = cand_floor; p.cand_n = cand_n; p.cand = (int2*)cand; p.cand_cap = cand_cap; p.t_index_base = item_index_base;
    return launch_blockmax_filter16(p, kpad, (hipStream_t)stream);
}

// Before the refining launches that list candidates: users for whom ``limit`` or more of 32 sampled superblocks reach their
// threshold (the criterion of trec_topk_rows_collect on the int8 table) get cand_floor = +inf -- nothing is listed for them --
// and are flagged (flag / n_flagged as trec_topk_filter_floor_ex left them).  n_sb >= 32.
extern "C" int trec_topk_dense_users(const float* table, int32_t n_sb, int64_t n_users, int64_t stride, const float* thr,
                                     const float* user_err, const float* sb_stats, int32_t kdim, int32_t limit,
                                     float* cand_floor, int32_t* flag, int32_t* n_flagged, void* stream)
{
    TREC_REQUIRE(table && thr && user_err && sb_stats && cand_floor && flag && n_flagged, "trec_topk_dense_users: null pointer");
    TREC_REQUIRE(n_sb >= 32 && stride >= n_users && limit >= 1 && limit <= 32, "trec_topk_dense_users: need n_sb >= 32, 1 <= limit <= 32");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(dense_users_kernel, dim3((unsigned)ceil_div64(n_users, 1024)), dim3(256), 0, (hipStream_t)stream, table, n_sb,
                       n_users, stride, thr, user_err, sb_stats, kdim, limit, cand_floor, flag, n_flagged);
    return trec_check_launch("trec_topk_dense_users");
}

// ---- only the workgroups that hold rows -----------------------------------------------------------------------------------
// The fixed-capacity layout gives every superblock wgs_per_row workgroup slots (50 % of the users: 981 at 1M) of which ~50
// hold rows; launched as a full [n_sb][wgs_per_row] grid, 1.8M of 1.9M workgroups start -- with their 79 KB of LDS and 256
// VGPRs allotted -- only to exit.  trec_topk_rows_wg_map lists the occupied slots: wg_start [n_sb + 1] = exclusive prefix of
// min(ceil(row_count / 512), wgs_per_row), wg_map[wg_start[s] + j] = s * wgs_per_row + j.  The host knows the total from the
// status of trec_topk_rows_hot ((rows - hot rows * padded users) / 512) and launches exactly that many.
namespace {
__global__ __launch_bounds__(256) void wg_start_kernel(const int32_t* __restrict__ row_count, int32_t n_sb, int32_t wgs_per_row,
                                                      int32_t* __restrict__ wg_start, int32_t group_rows)
{
    __shared__ int wsum[4];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s0 = 0; s0 < n_sb; s0 += 256) {
        const int s = s0 + threadIdx.x;
        int c = s < n_sb ? (row_count[s] + group_rows - 1) / group_rows : 0;
        if (c > wgs_per_row) c = wgs_per_row;
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int excl = base_s + inc - c;
        for (int w = 0; w < wave; ++w) excl += wsum[w];
        if (s < n_sb) wg_start[s] = excl;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) wg_start[n_sb] = base_s;
}

__global__ __launch_bounds__(256) void wg_map_kernel(const int32_t* __restrict__ wg_start, int32_t wgs_per_row, int64_t map_cap,
                                                    int32_t* __restrict__ wg_map)
{
    const int s = blockIdx.x;
    const int b = wg_start[s], e = wg_start[s + 1];
    for (int j = threadIdx.x; j < e - b; j += 256)
        if ((int64_t)b + j < map_cap) wg_map[b + j] = s * wgs_per_row + j;
}
}  // namespace

extern "C" int trec_topk_rows_wg_map(const int32_t* row_count, int32_t n_sb, int32_t wgs_per_row, int32_t* wg_start,
                                     int32_t* wg_map, int64_t map_cap, void* stream)
{
    TREC_REQUIRE(row_count && wg_start && wg_map, "trec_topk_rows_wg_map: null pointer");
    TREC_REQUIRE(n_sb >= 1 && wgs_per_row >= 1 && map_cap >= 0 && (int64_t)n_sb * wgs_per_row < ((int64_t)1 << 31),
                 "trec_topk_rows_wg_map: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wg_start_kernel, dim3(1), dim3(256), 0, st, row_count, n_sb, wgs_per_row, wg_start, GROUP_ROWS);
    hipLaunchKernelGGL(wg_map_kernel, dim3((unsigned)n_sb), dim3(256), 0, st, wg_start, wgs_per_row, map_cap, wg_map);
    return trec_check_launch("trec_topk_rows_wg_map");
}

// The same map with group_rows users per workgroup slot instead of 512 (the item-resident refining launch,
// trec_score_gemm_refine_candidates_resident: segments of 2,048 users): wg_map[wg_start[s] + j] = s * wgs_per_row + j for
// j < min(ceil(row_count[s] / group_rows), wgs_per_row); entries beyond wg_start[n_sb] are not written (the caller presets them idle).
extern "C" int trec_topk_rows_wg_map_ex(const int32_t* row_count, int32_t n_sb, int32_t wgs_per_row, int32_t group_rows,
                                        int32_t* wg_start, int32_t* wg_map, int64_t map_cap, void* stream)
{
    TREC_REQUIRE(row_count && wg_start && wg_map, "trec_topk_rows_wg_map_ex: null pointer");
    TREC_REQUIRE(n_sb >= 1 && wgs_per_row >= 1 && group_rows >= 1 && map_cap >= 0 && (int64_t)n_sb * wgs_per_row < ((int64_t)1 << 31),
                 "trec_topk_rows_wg_map_ex: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wg_start_kernel, dim3(1), dim3(256), 0, st, row_count, n_sb, wgs_per_row, wg_start, group_rows);
    hipLaunchKernelGGL(wg_map_kernel, dim3((unsigned)n_sb), dim3(256), 0, st, wg_start, wgs_per_row, map_cap, wg_map);
    return trec_check_launch("trec_topk_rows_wg_map_ex");
}
