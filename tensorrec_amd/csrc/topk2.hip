// tensorrec_amd/csrc/topk2.hip -- stages 2 and 3a of the two-stage exact top-k (stage 1 and 3b live in score_gemm.hip).
//
//   stage 1  score kernel, BLOCKMAX epilogue: M[s][u] = max exact score of user u over item superblock s
//   stage 2  select_blocks_kernel: per user the K superblocks with the largest maxima, order (max desc, s asc)
//   stage 3a group (user, slot) pairs by superblock (segment.hip counting sort), pad every group to whole workgroups,
//            gather the users' operand rows into that order
//   stage 3b score kernel, grouped TOPK epilogue: each workgroup re-scores ONE superblock for 256 gathered users
//   stage 4  trec_topk_merge over the K * 2 lists of each user
//
// Exactness (ties included).  Let x be one of the true top-K items (order: value desc, index asc) and X its superblock.
// If K superblocks B_1..B_K ranked before X under (max desc, s asc), each holds an item y_j with value max(B_j) >= max(X)
// >= value(x); where the values are equal, B_j < X as superblocks are index ranges, so index(y_j) < index(x).  Then K
// distinct items beat x -- contradiction.  Hence the K selected superblocks contain the whole top-K.
#include "common.hpp"

template <int KSEL>
__device__ __forceinline__ void sel_insert(float (&tv)[KSEL], int32_t (&ti)[KSEL], float s, int32_t id)
{
    bool ge_j = tv[KSEL - 1] >= s;
#pragma unroll
    for (int j = KSEL - 1; j >= 0; --j) {
        const bool ge_jm1 = (j == 0) ? true : (tv[j - 1] >= s);
        const float pv = (j == 0) ? 0.f : tv[j - 1];
        const int32_t pi = (j == 0) ? 0 : ti[j - 1];
        tv[j] = ge_j ? tv[j] : (ge_jm1 ? s : pv);
        ti[j] = ge_j ? ti[j] : (ge_jm1 ? id : pi);
        ge_j = ge_jm1;
    }
}

// one lane per user; blockmax is [n_sb][stride] so a wave reads 64 consecutive users per superblock (coalesced)
template <int KSEL>
__global__ __launch_bounds__(256) void select_blocks_kernel(const float* __restrict__ blockmax, int32_t n_sb,
                                                           int64_t n_users, int64_t stride, int32_t k, int32_t k_tau,
                                                           int32_t* __restrict__ sel, float* __restrict__ sel_max,
                                                           float* __restrict__ tau)
{
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = u < n_users;
    float tv[KSEL];
    int32_t ti[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    // the list holds the KSEL best; tau = entry k_tau - 1 (the k-th largest maximum: a floor of the final k-th best score)
    for (int32_t s = 0; s < n_sb; ++s) {
        const float v = ok ? blockmax[(int64_t)s * stride + u] : -INFINITY;
        const bool hit = (v > tv[KSEL - 1]) || (ti[KSEL - 1] < 0 && ok);      // strict: earlier superblocks win ties
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) sel_insert<KSEL>(tv, ti, hit ? v : -INFINITY, hit ? s : -1);
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < k) {
                sel[u * k + j] = ti[j];
                if (sel_max) sel_max[(int64_t)j * n_users + u] = tv[j];     // [k][n_users]: the blockmax layout
            }
        // k superblocks have a maximum >= tv[k-1], i.e. k items score >= it: a valid floor for the final k-th best score
        // (-inf while fewer than k superblocks exist).  The re-scoring pass starts its lists from this threshold.
        if (tau) {
            float t = -INFINITY;
#pragma unroll
            for (int j = 0; j < KSEL; ++j)
                if (j == k_tau - 1 && ti[j] >= 0) t = tv[j];
            tau[u] = t;
        }
    }
}

// The same selection with 16-byte loads: a workgroup of 256 users stages tiles of SEL_TS superblock rows through LDS
// (thread t loads float4 = 4 consecutive users of a row, two rows per tile; register prefetch of the next tile), then
// every thread walks ITS column of the tile.  The table is 7.8 GB at 1M x 1M: with 4-byte-per-lane loads the one-pass
// kernel above reads it at 2.8 TB/s.  Needs stride % 4 == 0 and a 16-byte aligned table; same insertion order (s
// ascending) as above, hence the same result.
#define SEL_TS 8
template <int KSEL>
__global__ __launch_bounds__(256) void select_blocks_tiled_kernel(const float* __restrict__ blockmax, int32_t n_sb,
                                                                 int64_t n_users, int64_t stride, int32_t k, int32_t k_tau,
                                                                 int32_t* __restrict__ sel, float* __restrict__ sel_max,
                                                                 float* __restrict__ tau)
{
    __shared__ __attribute__((aligned(16))) float tile[2][SEL_TS][256];
    const int t = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * 256;
    const int64_t u = u0 + t;
    const bool ok = u < n_users;
    const int lrow = t >> 6, lcol = (t & 63) * 4;               // this thread loads rows lrow, lrow + 4 of a tile
    const bool colok = u0 + lcol + 3 < stride;
    float tv[KSEL];
    int32_t ti[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    auto fetch = [&](int32_t s0, f32x4 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int32_t srow = s0 + lrow + 4 * i;
            r[i] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (srow < n_sb && colok) r[i] = __builtin_nontemporal_load((const f32x4*)(blockmax + (int64_t)srow * stride + u0 + lcol));
            else if (srow < n_sb) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (u0 + lcol + e < stride) r[i][e] = blockmax[(int64_t)srow * stride + u0 + lcol + e];
            }
        }
    };
    f32x4 cur[2], nxt[2];
    fetch(0, cur);
    int buf = 0;
    for (int32_t s0 = 0; s0 < n_sb; s0 += SEL_TS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(f32x4*)(&tile[buf][lrow + 4 * i][lcol]) = cur[i];
        if (s0 + SEL_TS < n_sb) fetch(s0 + SEL_TS, nxt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SEL_TS; ++r) {
            const int32_t s = s0 + r;
            const float v = (ok && s < n_sb) ? tile[buf][r][t] : -INFINITY;
            const bool hit = s < n_sb && ((v > tv[KSEL - 1]) || (ti[KSEL - 1] < 0 && ok));
            if (__builtin_amdgcn_ballot_w64(hit) != 0ull) sel_insert<KSEL>(tv, ti, hit ? v : -INFINITY, hit ? s : -1);
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
        buf ^= 1;                                               // the other buffer is free: its readers passed the barrier
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j < k) {
                sel[u * k + j] = ti[j];
                if (sel_max) sel_max[(int64_t)j * n_users + u] = tv[j];
            }
        if (tau) {
            float t = -INFINITY;
#pragma unroll
            for (int j = 0; j < KSEL; ++j)
                if (j == k_tau - 1 && ti[j] >= 0) t = tv[j];
            tau[u] = t;
        }
    }
}

// Pass 2 of the filtered selection (topk_filter.hip): every superblock whose maximum reaches the user's floor, in
// superblock order -- keys[u * ksel + c] = s for the c-th such superblock (c < ksel), -1 in the unused slots;
// count[u] = min(c, ksel).  More than ksel qualifying superblocks: flag[u] = 1 (the caller re-does the user exactly).
// Same tiling as select_blocks_tiled_kernel (16-byte loads through LDS, register prefetch of the next tile).
__global__ __launch_bounds__(256) void collect_blocks_tiled_kernel(const float* __restrict__ blockmax, int32_t n_sb,
                                                                  int64_t n_users, int64_t stride,
                                                                  const float* __restrict__ floor_, int32_t ksel,
                                                                  int32_t* __restrict__ keys, int32_t* __restrict__ count,
                                                                  int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
                                                                  const int32_t* __restrict__ redo)
{
    __shared__ __attribute__((aligned(16))) float tile[2][SEL_TS][256];
    const int t = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * 256;
    const int64_t u = u0 + t;
    // redo (nullable): only the users marked there are collected (the ones whose candidate list of the one-pass scan was
    // incomplete); a workgroup none of whose 256 users is marked leaves before touching the table
    const bool ok = u < n_users && (!redo || redo[u] != 0);
    if (redo && !__syncthreads_or(ok ? 1 : 0)) return;
    const int lrow = t >> 6, lcol = (t & 63) * 4;
    const bool vec = (stride % 4 == 0) && (((uintptr_t)blockmax % 16) == 0) && (u0 + lcol + 3 < stride);
    const float fl = ok ? floor_[u] : INFINITY;
    int32_t c = 0;
    auto fetch = [&](int32_t s0, f32x4 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int32_t srow = s0 + lrow + 4 * i;
            r[i] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (srow < n_sb && vec) r[i] = __builtin_nontemporal_load((const f32x4*)(blockmax + (int64_t)srow * stride + u0 + lcol));
            else if (srow < n_sb) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (u0 + lcol + e < stride) r[i][e] = blockmax[(int64_t)srow * stride + u0 + lcol + e];
            }
        }
    };
    f32x4 cur[2], nxt[2];
    fetch(0, cur);
    int buf = 0;
    for (int32_t s0 = 0; s0 < n_sb; s0 += SEL_TS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(f32x4*)(&tile[buf][lrow + 4 * i][lcol]) = cur[i];
        if (s0 + SEL_TS < n_sb) fetch(s0 + SEL_TS, nxt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SEL_TS; ++r) {
            const int32_t s = s0 + r;
            const float v = tile[buf][r][t];
            if (ok && s < n_sb && !(v < fl)) {
                if (c < ksel) keys[u * ksel + c] = s;
                ++c;
            }
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
        buf ^= 1;
    }
    if (ok) {
        // a user with more qualifying superblocks than slots is flagged and re-done by the caller (a wider pass, then the exact
        // path): nothing of its partial selection is used, so its slots are emptied and the later stages skip it
        const bool over = c > ksel;
        for (int32_t j = over ? 0 : c; j < ksel; ++j) keys[u * ksel + j] = -1;
        count[u] = over ? 0 : c;
        if (over && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
    }
}

// ---- select + collect in ONE pass over the table (the cascade's bf16 filter, round 3) ------------------------------------
// trec_topk_select_blocks (tau = the k-th largest entry of a user's column) and trec_topk_collect_blocks (every entry >= floor =
// tau - 2 eps) each read the 7.8 GB table.  The second pass only exists because the floor is not known during the first.  But a
// PROVISIONAL floor is: with the int8 stage in front, floor >= tau8 - 3 eps (k superblocks certify items with fp32 score >=
// tau8; they are refined, so their entries are >= tau8 - eps: tau >= tau8 - eps).  One pass keeps the k largest entries (sorted
// registers, as select does) AND appends every entry >= floor0 to the user's candidate list (superblock, value; at most
// cand_cap, in superblock order, stored slot-major [cand_cap][n_users]); a second kernel that touches ~45 candidates per user instead of 1,954 table entries prunes
// them with the final floor.  cand_n[u] counts ALL entries >= floor0: above cand_cap the list is incomplete and the prune
// kernel flags the user.
template <int KSEL>
__global__ __launch_bounds__(256) void scan_blocks_tiled_kernel(const float* __restrict__ blockmax, int32_t n_sb,
                                                               int64_t n_users, int64_t stride, int32_t k,
                                                               const float* __restrict__ floor0, int32_t cand_cap,
                                                               float* __restrict__ sel_max, float* __restrict__ tau,
                                                               int32_t* __restrict__ cand_s, float* __restrict__ cand_v,
                                                               int32_t* __restrict__ cand_n)
{
    __shared__ __attribute__((aligned(16))) float tile[2][SEL_TS][256];
    const int t = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * 256;
    const int64_t u = u0 + t;
    const bool ok = u < n_users;
    const int lrow = t >> 6, lcol = (t & 63) * 4;
    const bool vec = (stride % 4 == 0) && (((uintptr_t)blockmax % 16) == 0) && (u0 + lcol + 3 < stride);
    float tv[KSEL];
    int32_t ti[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; ++j) { tv[j] = -INFINITY; ti[j] = -1; }
    const float f0 = ok ? floor0[u] : INFINITY;
    int32_t c = 0;
    // candidate j of user u lives at [j][u] (slot-major): the prune kernel's lanes then read consecutive addresses
    int32_t* cs = cand_s + u;
    float* cv = cand_v + u;
    auto fetch = [&](int32_t s0, f32x4 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int32_t srow = s0 + lrow + 4 * i;
            r[i] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (srow < n_sb && vec) r[i] = __builtin_nontemporal_load((const f32x4*)(blockmax + (int64_t)srow * stride + u0 + lcol));
            else if (srow < n_sb) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (u0 + lcol + e < stride) r[i][e] = blockmax[(int64_t)srow * stride + u0 + lcol + e];
            }
        }
    };
    f32x4 cur[2], nxt[2];
    fetch(0, cur);
    int buf = 0;
    for (int32_t s0 = 0; s0 < n_sb; s0 += SEL_TS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(f32x4*)(&tile[buf][lrow + 4 * i][lcol]) = cur[i];
        if (s0 + SEL_TS < n_sb) fetch(s0 + SEL_TS, nxt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SEL_TS; ++r) {
            const int32_t s = s0 + r;
            const float v = (ok && s < n_sb) ? tile[buf][r][t] : -INFINITY;
            const bool hit = s < n_sb && ((v > tv[KSEL - 1]) || (ti[KSEL - 1] < 0 && ok));
            if (__builtin_amdgcn_ballot_w64(hit) != 0ull) sel_insert<KSEL>(tv, ti, hit ? v : -INFINITY, hit ? s : -1);
            if (ok && s < n_sb && !(v < f0)) {                   // (NaN entries stay candidates, as in collect)
                if (c < cand_cap) { cs[(int64_t)c * n_users] = s; cv[(int64_t)c * n_users] = v; }
                ++c;
            }
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
        buf ^= 1;
    }
    if (ok) {
        cand_n[u] = c;
        if (sel_max) {
#pragma unroll
            for (int j = 0; j < KSEL; ++j)
                if (j < k) sel_max[(int64_t)j * n_users + u] = tv[j];
        }
        float tk = -INFINITY;
#pragma unroll
        for (int j = 0; j < KSEL; ++j)
            if (j == k - 1 && ti[j] >= 0) tk = tv[j];
        tau[u] = tk;
    }
}

// keys / count of trec_topk_collect_blocks from the candidate lists: entries with value >= floor, in superblock order.
// Flags (and empties) a user with more than ksel of them or with an incomplete candidate list.  One thread per user.
__global__ __launch_bounds__(256) void prune_candidates_kernel(const int32_t* __restrict__ cand_s, const float* __restrict__ cand_v,
                                                              const int32_t* __restrict__ cand_n, int32_t cand_cap,
                                                              const float* __restrict__ floor_, int32_t ksel, int64_t n_users,
                                                              int32_t* __restrict__ keys, int32_t* __restrict__ count,
                                                              int32_t* __restrict__ flag, int32_t* __restrict__ n_flagged,
                                                              int32_t* __restrict__ redo)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const int32_t n_all = cand_n[u];
    const int32_t n = n_all < cand_cap ? n_all : cand_cap;
    const float fl = floor_[u];
    const int32_t* cs = cand_s + u;                                     // [j][u]: coalesced over the lanes
    const float* cv = cand_v + u;
    int32_t* ku = keys + u * (int64_t)ksel;
    int32_t c = 0;
    for (int32_t j = 0; j < n; ++j) {
        if (!(cv[(int64_t)j * n_users] < fl)) {
            if (c < ksel) ku[c] = cs[(int64_t)j * n_users];
            ++c;
        }
    }
    // an incomplete list may have lost entries >= floor (unless the floor turned out +inf: padding rows keep nothing): such a
    // user is marked for the masked collect pass that follows (a loose int8 bound leaves hundreds of entries above the
    // provisional floor); the keys written here are then overwritten
    const bool incomplete = n_all > cand_cap && fl < INFINITY;
    redo[u] = incomplete ? 1 : 0;
    const bool over = c > ksel && !incomplete;
    for (int32_t j = (over || incomplete) ? 0 : c; j < ksel; ++j) ku[j] = -1;
    count[u] = (over || incomplete) ? 0 : c;
    if (over && flag[u] == 0) { flag[u] = 1; atomicAdd(n_flagged, 1); }
}

// keys for the counting sort: superblock id, or -1 (skipped by the sort) for empty slots and for superblocks whose
// maximum lies below the user's floor.  The floor is any lower bound of the user's final k-th best score -- with item
// shards, the MAX over ranks of the per-shard tau (every shard's k-th best is a floor of the global k-th best): a
// superblock with max < floor holds no item of the global top-k (strict: a score equal to the floor may still tie in).
// sel_max is [k][n_users] (the blockmax layout), so the all-gathered maxima of all ranks can go straight back through
// select_blocks_kernel, whose tau output is then the k-th largest superblock maximum over ALL shards.
__global__ __launch_bounds__(256) void group_keys_kernel(const int32_t* __restrict__ sel,
                                                        const float* __restrict__ sel_max,
                                                        const float* __restrict__ floor_, int64_t n, int32_t k,
                                                        int32_t n_sb, int32_t* __restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool keep = sel[i] >= 0;
    if (keep && floor_) keep = !(sel_max[(i % k) * (n / k) + i / k] < floor_[i / k]);
    keys[i] = keep ? sel[i] : -1;            // negative keys are skipped by the counting sort (no shared dummy counter)
}

// padded group sizes: every superblock's list of users is rounded up to whole workgroups of `rows_wg` rows
__global__ __launch_bounds__(256) void pad_counts_kernel(const int64_t* __restrict__ indptr_t, int32_t n_sb,
                                                        int32_t rows_wg, int32_t* __restrict__ cnt_pad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n_sb) return;
    if (i == n_sb) { cnt_pad[i] = 0; return; }                      // the dummy bucket is never re-scored
    const int64_t c = indptr_t[i + 1] - indptr_t[i];
    cnt_pad[i] = (int32_t)((c + rows_wg - 1) / rows_wg * rows_wg);
}

// One thread per 16-byte chunk of the gathered operand matrix G[max_rows][row_bytes].  A workgroup owns FILL_ROWS
// consecutive output rows and first copies the group starts (pstart, <= 16 KB for up to 2048 superblocks) into LDS:
// the per-row binary search then runs on LDS instead of being 11 dependent global loads per wave (that latency chain
// was the whole cost of this kernel: 2.7 ms for 10 M rows before, with every 16 threads of a row repeating it).
#define FILL_ROWS 1024
#define FILL_LDS_SB 2048
__global__ __launch_bounds__(256) void fill_groups_kernel(
    const int64_t* __restrict__ pstart, const int64_t* __restrict__ indptr_t, const int32_t* __restrict__ users_t,
    const int32_t* __restrict__ perm_t, int32_t n_sb, int32_t rows_wg, int64_t max_rows, const char* __restrict__ users_op,
    int32_t row_bytes, const float* __restrict__ user_bias, const float* __restrict__ user_sq,
    const float* __restrict__ user_tau, char* __restrict__ G, float* __restrict__ g_bias, float* __restrict__ g_sq,
    float* __restrict__ g_tau, int32_t* __restrict__ row_pair, int32_t* __restrict__ rblock_chunk)
{
    __shared__ int64_t l_pstart[FILL_LDS_SB + 1];
    const bool in_lds = n_sb <= FILL_LDS_SB;
    if (in_lds) {
        for (int i = threadIdx.x; i <= n_sb; i += 256) l_pstart[i] = pstart[i];
        __syncthreads();
    }
    const int64_t* ps = in_lds ? l_pstart : pstart;
    const int ch = row_bytes / 16;
    const int64_t total = ps[n_sb];
    const int64_t row0 = (int64_t)blockIdx.x * FILL_ROWS;
    const int64_t row1 = (row0 + FILL_ROWS < max_rows) ? row0 + FILL_ROWS : max_rows;
    for (int64_t idx = row0 * ch + threadIdx.x; idx < row1 * ch; idx += 256) {
        const int64_t v = idx / ch;
        const int c = (int)(idx - v * ch);
        int32_t sb = -1;
        int64_t src_user = 0;
        int32_t pair = -1;
        if (v < total) {
            int lo = 0, hi = n_sb - 1;                              // last sb with pstart[sb] <= v and a non-empty group
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (ps[mid] <= v) lo = mid; else hi = mid - 1;
            }
            sb = lo;
            const int64_t o = v - ps[sb];
            const int64_t cnt = indptr_t[sb + 1] - indptr_t[sb];
            if (o < cnt) {
                const int64_t e = indptr_t[sb] + o;
                src_user = users_t[e];
                pair = perm_t[e];
            }
        }
        *(u32x4*)(G + v * row_bytes + c * 16) = *(const u32x4*)(users_op + src_user * row_bytes + c * 16);
        if (c == 0) {
            row_pair[v] = pair;
            if (g_bias) g_bias[v] = user_bias[src_user];
            if (g_sq) g_sq[v] = user_sq[src_user];
            if (g_tau) g_tau[v] = (pair >= 0) ? user_tau[src_user] : INFINITY;       // padding rows never insert
            if (v % rows_wg == 0) rblock_chunk[v / rows_wg] = sb;      // -1 beyond the padded total: workgroup exits at once
        }
    }
}

// k = entries kept per user (list length, <= 64), k_tau <= k = which entry is reported as tau
// fill_groups without the operand copy: one thread per grouped row writes row_user[v] (the operand row the grouped score
// kernel loads through its row_index; -1 = padding), row_pair[v] and, for the first row of a workgroup, rblock_chunk.
__global__ __launch_bounds__(256) void fill_groups_index_kernel(
    const int64_t* __restrict__ pstart, const int64_t* __restrict__ indptr_t, const int32_t* __restrict__ users_t,
    const int32_t* __restrict__ perm_t, int32_t n_sb, int32_t rows_wg, int64_t max_rows, int32_t* __restrict__ row_user,
    int32_t* __restrict__ row_pair, int32_t* __restrict__ rblock_chunk)
{
    __shared__ int64_t l_pstart[FILL_LDS_SB + 1];
    const bool in_lds = n_sb <= FILL_LDS_SB;
    if (in_lds) {
        for (int i = threadIdx.x; i <= n_sb; i += 256) l_pstart[i] = pstart[i];
        __syncthreads();
    }
    const int64_t* ps = in_lds ? l_pstart : pstart;
    const int64_t total = ps[n_sb];
    const int64_t row0 = (int64_t)blockIdx.x * FILL_ROWS;
    const int64_t row1 = (row0 + FILL_ROWS < max_rows) ? row0 + FILL_ROWS : max_rows;
    for (int64_t v = row0 + threadIdx.x; v < row1; v += 256) {
        int32_t sb = -1, user = -1, pair = -1;
        if (v < total) {
            int lo = 0, hi = n_sb - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (ps[mid] <= v) lo = mid; else hi = mid - 1;
            }
            sb = lo;
            const int64_t o = v - ps[sb];
            if (o < indptr_t[sb + 1] - indptr_t[sb]) {
                const int64_t e = indptr_t[sb] + o;
                user = users_t[e];
                pair = perm_t[e];
            }
        }
        row_user[v] = user;
        row_pair[v] = pair;
        if (v % rows_wg == 0) rblock_chunk[v / rows_wg] = sb;
    }
}

extern "C" int trec_topk_fill_groups_index(const int64_t* pstart, const int64_t* indptr_t, const int32_t* users_t,
                                           const int32_t* perm_t, int32_t n_sb, int32_t rows_wg, int64_t max_rows,
                                           int32_t* row_user, int32_t* row_pair, int32_t* rblock_chunk, void* stream)
{
    TREC_REQUIRE(pstart && indptr_t && users_t && perm_t && row_user && row_pair && rblock_chunk,
                 "trec_topk_fill_groups_index: null pointer");
    TREC_REQUIRE(rows_wg >= 1 && max_rows % rows_wg == 0, "trec_topk_fill_groups_index: max_rows % rows_wg");
    if (max_rows == 0) return TREC_OK;
    hipLaunchKernelGGL(fill_groups_index_kernel, dim3((unsigned)ceil_div64(max_rows, FILL_ROWS)), dim3(256), 0,
                       (hipStream_t)stream, pstart, indptr_t, users_t, perm_t, n_sb, rows_wg, max_rows, row_user, row_pair,
                       rblock_chunk);
    return trec_check_launch("trec_topk_fill_groups_index");
}

extern "C" int trec_topk_select_blocks_ex(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k,
                                          int32_t k_tau, int32_t* sel, float* sel_max, float* tau, void* stream)
{
    TREC_REQUIRE(blockmax && sel, "trec_topk_select_blocks: null pointer");
    TREC_REQUIRE(k >= 1 && k <= 64 && n_sb >= 1 && k_tau >= 1 && k_tau <= k, "trec_topk_select_blocks: need 1 <= k_tau <= k <= 64, n_sb >= 1");
    if (n_users == 0) return TREC_OK;
    const unsigned blocks = (unsigned)ceil_div64(n_users, 256);
    hipStream_t st = (hipStream_t)stream;
    const bool tiled = stride % 4 == 0 && ((uintptr_t)blockmax % 16) == 0 && n_sb >= 4 * SEL_TS &&
                       trec_get_tuning("select_tiled", 1);
#define TREC_SEL(KS)                                                                                                   \
    do {                                                                                                               \
        if (tiled) hipLaunchKernelGGL((select_blocks_tiled_kernel<KS>), dim3(blocks), dim3(256), 0, st, blockmax, n_sb, \
                                      n_users, stride, k, k_tau, sel, sel_max, tau);                                   \
        else hipLaunchKernelGGL((select_blocks_kernel<KS>), dim3(blocks), dim3(256), 0, st, blockmax, n_sb, n_users,    \
                                stride, k, k_tau, sel, sel_max, tau);                                                  \
    } while (0)
    // the list length is a template parameter (registers): exact for k <= 16, rounded up above (entries beyond k unused)
    switch (k) {
        case 1: TREC_SEL(1); break;   case 2: TREC_SEL(2); break;   case 3: TREC_SEL(3); break;   case 4: TREC_SEL(4); break;
        case 5: TREC_SEL(5); break;   case 6: TREC_SEL(6); break;   case 7: TREC_SEL(7); break;   case 8: TREC_SEL(8); break;
        case 9: TREC_SEL(9); break;   case 10: TREC_SEL(10); break; case 11: TREC_SEL(11); break; case 12: TREC_SEL(12); break;
        case 13: TREC_SEL(13); break; case 14: TREC_SEL(14); break; case 15: TREC_SEL(15); break; case 16: TREC_SEL(16); break;
        default:
            if (k <= 20) TREC_SEL(20);
            else if (k <= 24) TREC_SEL(24);
            else if (k <= 32) TREC_SEL(32);
            else if (k <= 40) TREC_SEL(40);
            else if (k <= 48) TREC_SEL(48);
            else TREC_SEL(64);
            break;
    }
#undef TREC_SEL
    return trec_check_launch("trec_topk_select_blocks");
}

extern "C" int trec_topk_select_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k,
                                       int32_t* sel, float* sel_max, float* tau, void* stream)
{
    TREC_REQUIRE(k >= 1 && k <= 16, "trec_topk_select_blocks: need 1 <= k <= 16");
    return trec_topk_select_blocks_ex(blockmax, n_sb, n_users, stride, k, k, sel, sel_max, tau, stream);
}

extern "C" int trec_topk_collect_blocks_masked(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride,
                                               const float* floor_, int32_t ksel, int32_t* keys, int32_t* count, int32_t* flag,
                                               int32_t* n_flagged, const int32_t* redo, void* stream);

extern "C" int trec_topk_collect_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride,
                                        const float* floor_, int32_t ksel, int32_t* keys, int32_t* count, int32_t* flag,
                                        int32_t* n_flagged, void* stream)
{
    return trec_topk_collect_blocks_masked(blockmax, n_sb, n_users, stride, floor_, ksel, keys, count, flag, n_flagged, nullptr, stream);
}

// trec_topk_collect_blocks for the users with redo[u] != 0 only (redo == NULL: everybody): the others' keys / count are left
// alone, and a workgroup of 256 users without a marked one exits before reading the table.
extern "C" int trec_topk_collect_blocks_masked(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride,
                                               const float* floor_, int32_t ksel, int32_t* keys, int32_t* count, int32_t* flag,
                                               int32_t* n_flagged, const int32_t* redo, void* stream)
{
    TREC_REQUIRE(blockmax && floor_ && keys && count && flag && n_flagged, "trec_topk_collect_blocks: null pointer");
    TREC_REQUIRE(ksel >= 1 && ksel <= 4096 && n_sb >= 1 && stride >= n_users, "trec_topk_collect_blocks: need 1 <= ksel <= 4096");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(collect_blocks_tiled_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0,
                       (hipStream_t)stream, blockmax, n_sb, n_users, stride, floor_, ksel, keys, count, flag, n_flagged, redo);
    return trec_check_launch("trec_topk_collect_blocks");
}

extern "C" int trec_topk_group_keys(const int32_t* sel, const float* sel_max, const float* floor_, int64_t n, int32_t k,
                                    int32_t n_sb, int32_t* keys, void* stream)
{
    TREC_REQUIRE(sel && keys && k >= 1, "trec_topk_group_keys: null pointer");
    TREC_REQUIRE(!floor_ || sel_max, "trec_topk_group_keys: a floor needs the selected maxima");
    if (n == 0) return TREC_OK;
    hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, sel,
                       sel_max, floor_, n, k, n_sb, keys);
    return trec_check_launch("trec_topk_group_keys");
}

extern "C" int trec_topk_pad_counts(const int64_t* indptr_t, int32_t n_sb, int32_t rows_wg, int32_t* cnt_pad, void* stream)
{
    TREC_REQUIRE(indptr_t && cnt_pad && rows_wg >= 1, "trec_topk_pad_counts: bad arguments");
    hipLaunchKernelGGL(pad_counts_kernel, dim3((unsigned)((n_sb + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       indptr_t, n_sb, rows_wg, cnt_pad);
    return trec_check_launch("trec_topk_pad_counts");
}

extern "C" int trec_topk_fill_groups(const int64_t* pstart, const int64_t* indptr_t, const int32_t* users_t,
                                     const int32_t* perm_t, int32_t n_sb, int32_t rows_wg, int64_t max_rows,
                                     const void* users_op, int32_t row_bytes, const float* user_bias,
                                     const float* user_sq, const float* user_tau, void* G, float* g_bias, float* g_sq,
                                     float* g_tau, int32_t* row_pair, int32_t* rblock_chunk, void* stream)
{
    TREC_REQUIRE(pstart && indptr_t && users_t && perm_t && users_op && G && row_pair && rblock_chunk,
                 "trec_topk_fill_groups: null pointer");
    TREC_REQUIRE(row_bytes % 16 == 0 && max_rows % rows_wg == 0, "trec_topk_fill_groups: row_bytes % 16, max_rows % rows_wg");
    TREC_REQUIRE((g_bias == nullptr) == (user_bias == nullptr) && (g_sq == nullptr) == (user_sq == nullptr) &&
                     (g_tau == nullptr) == (user_tau == nullptr),
                 "trec_topk_fill_groups: bias / sqnorm buffers must come in pairs");
    if (max_rows == 0) return TREC_OK;
    hipLaunchKernelGGL(fill_groups_kernel, dim3((unsigned)ceil_div64(max_rows, FILL_ROWS)), dim3(256), 0, (hipStream_t)stream,
                       pstart, indptr_t, users_t, perm_t, n_sb, rows_wg, max_rows, (const char*)users_op, row_bytes,
                       user_bias, user_sq, user_tau, (char*)G, g_bias, g_sq, g_tau, row_pair, rblock_chunk);
    return trec_check_launch("trec_topk_fill_groups");
}

// The k-th largest entry of every user's column (tau, and sel_max [k][n_users] for item shards) AND the candidates for the
// collect step in ONE pass over the table: cand_s / cand_v [cand_cap][n_users] = the entries >= floor0[u] in superblock order,
// cand_n [n_users] = how many there were (possibly more than cand_cap).  floor0 must not exceed the final floor (see the
// kernel's header).  k <= 16.
extern "C" int trec_topk_scan_blocks(const float* blockmax, int32_t n_sb, int64_t n_users, int64_t stride, int32_t k,
                                     const float* floor0, int32_t cand_cap, float* sel_max, float* tau, int32_t* cand_s,
                                     float* cand_v, int32_t* cand_n, void* stream)
{
    TREC_REQUIRE(blockmax && floor0 && tau && cand_s && cand_v && cand_n, "trec_topk_scan_blocks: null pointer");
    TREC_REQUIRE(k >= 1 && k <= 16 && n_sb >= 1 && cand_cap >= 1 && stride >= n_users, "trec_topk_scan_blocks: need 1 <= k <= 16");
    if (n_users == 0) return TREC_OK;
    const unsigned blocks = (unsigned)ceil_div64(n_users, 256);
    hipStream_t st = (hipStream_t)stream;
#define TREC_SCAN(KS) hipLaunchKernelGGL((scan_blocks_tiled_kernel<KS>), dim3(blocks), dim3(256), 0, st, blockmax, n_sb, n_users, \
                                         stride, k, floor0, cand_cap, sel_max, tau, cand_s, cand_v, cand_n)
    if (k <= 4) TREC_SCAN(4);
    else if (k <= 8) TREC_SCAN(8);
    else if (k <= 10) TREC_SCAN(10);
    else if (k <= 12) TREC_SCAN(12);
    else TREC_SCAN(16);
#undef TREC_SCAN
    return trec_check_launch("trec_topk_scan_blocks");
}

// the collect step from the candidate lists: keys [n_users][ksel] / count as trec_topk_collect_blocks writes them; redo
// [n_users] = 1 for the users whose candidate list was incomplete (trec_topk_collect_blocks_masked re-does exactly those)
extern "C" int trec_topk_prune_candidates(const int32_t* cand_s, const float* cand_v, const int32_t* cand_n, int32_t cand_cap,
                                          const float* floor_, int32_t ksel, int64_t n_users, int32_t* keys, int32_t* count,
                                          int32_t* flag, int32_t* n_flagged, int32_t* redo, void* stream)
{
    TREC_REQUIRE(cand_s && cand_v && cand_n && floor_ && keys && count && flag && n_flagged && redo, "trec_topk_prune_candidates: null pointer");
    TREC_REQUIRE(ksel >= 1 && cand_cap >= 1, "trec_topk_prune_candidates: bad sizes");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(prune_candidates_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream,
                       cand_s, cand_v, cand_n, cand_cap, floor_, ksel, n_users, keys, count, flag, n_flagged, redo);
    return trec_check_launch("trec_topk_prune_candidates");
}
