// tensorrec_amd/csrc/rank.hip -- K4: item ranks per user, exact and sort-free.
//
// Replaces rank_predictions (tensorrec/recommendation_graphs.py:73-82): two full tf.nn.top_k sorts per user.
// The double sort is exactly a count (tests/test_oracle_golden.py::test_rank_is_counting...):
//     rank[u, i] = 1 + #{ j : s[u,j] > s[u,i]  or  (s[u,j] == s[u,i] and j < i) }
// so ranks are integer work on the score matrix: bit-exact by construction, and a count over an item range is
// additive -- item shards just sum their partial counts (sharding.py), no cross-shard sort.
//
// rank_rows_kernel: grid (item blocks of 1024, users).  A block owns 1024 target items of one user (4 per
// thread, in registers) and streams the user's whole score row through LDS in 2048-float tiles; every LDS read
// is a wave-wide broadcast of a float4, compared against the 4 register targets (16 compares per ds_read_b128).
// Tiles entirely below / above the block's target range need a single predicate (>= resp. >); only the diagonal
// tiles evaluate the index tie-break.
//
// rank_rows_sorted_kernel (rows of at most 32768 items): the count above is a position in the sorted row.  One workgroup
// per user maps the row to order-preserving 32-bit keys in LDS (128 KB at 32768), sorts them with an in-place bitonic
// network, and every item finds  #{keys > its key} = NP - upper_bound  by binary search: O(I log^2 I) LDS work instead
// of I^2 compares (26,744 items: 0.13 ms per row against 14 ms).  Ties need the index order the keys do not carry:
// the (few) tied items go to a second list of (class, index) entries that is sorted as well -- an entry's offset from
// the start of its class is #{j < i : s_j == s_i}.  A row with more than 2048 tied items (identical scores everywhere:
// integer-valued predictions) is marked with rank -1 and recounted by rank_rows_kernel, which otherwise exits.
//
// rank_of_pairs_kernel: ranks for selected (user, item) pairs only, counting over a [begin, end) item range --
// the item-shardable form used for evaluation at sizes where the [U, I] matrix cannot exist.
#include "common.hpp"

#define RANK_TGT 1024
#define RANK_TILE 2048

// FIXUP: only the rows the sorted kernel marked (rank -1 on every item) are counted; each block tests an element that
// only it writes
template <bool FIXUP>
__global__ __launch_bounds__(256) void rank_rows_kernel(const float* __restrict__ scores, int64_t n_items,
                                                       int64_t ld, int32_t* __restrict__ ranks, int64_t ld_out)
{
    __shared__ __attribute__((aligned(16))) float tile[RANK_TILE];
    const int64_t u = blockIdx.y;
    const float* row = scores + u * ld;
    const int64_t tgt0 = (int64_t)blockIdx.x * RANK_TGT;
    if (FIXUP && ranks[u * ld_out + tgt0] != -1) return;
    // thread t owns targets tgt0 + t + 256*e  (coalesced loads/stores)
    float tv[4];
    int64_t tix[4];
    int cnt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        tix[e] = tgt0 + threadIdx.x + 256 * e;
        tv[e] = (tix[e] < n_items) ? row[tix[e]] : 0.f;
        cnt[e] = 0;
    }
    const int64_t tgt_end = tgt0 + RANK_TGT;
    for (int64_t j0 = 0; j0 < n_items; j0 += RANK_TILE) {
        __syncthreads();
        for (int q = threadIdx.x; q < RANK_TILE; q += 256) {
            const int64_t j = j0 + q;
            tile[q] = (j < n_items) ? row[j] : -INFINITY;     // padding never beats anything (and NaN compares false)
        }
        __syncthreads();
        const int64_t jend = j0 + RANK_TILE;
        const bool padded = jend > n_items;
        if (jend <= tgt0 && !padded) {
            // every j in the tile is below every target index: beats when s_j >= s_i
            for (int q = 0; q < RANK_TILE; q += 4) {
                const f32x4 s = *(const f32x4*)(tile + q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cnt[e] += (s.x >= tv[e]) + (s.y >= tv[e]) + (s.z >= tv[e]) + (s.w >= tv[e]);
            }
        } else if (j0 >= tgt_end) {
            // every j is above every target index: beats only when strictly greater (padding is -inf)
            for (int q = 0; q < RANK_TILE; q += 4) {
                const f32x4 s = *(const f32x4*)(tile + q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cnt[e] += (s.x > tv[e]) + (s.y > tv[e]) + (s.z > tv[e]) + (s.w > tv[e]);
            }
        } else {
            for (int q = 0; q < RANK_TILE; ++q) {
                const float s = tile[q];
                const int64_t j = j0 + q;
                if (j >= n_items) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) cnt[e] += (s > tv[e]) || (s == tv[e] && j < tix[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (tix[e] < n_items) ranks[u * ld_out + tix[e]] = cnt[e] + 1;
}

#define RANK_SORT_MAX 32768      // keys of one row in LDS: 128 KB
#define RANK_TIE_CAP 2048        // tied items handled by the (class, index) list: 16 KB

// order-preserving map float -> uint32 (larger score = larger key); -0.0 ranks as +0.0 (they compare equal)
__device__ __forceinline__ uint32_t rank_key(float f)
{
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename K>
__device__ __forceinline__ void bitonic_sort_lds(K* a, int n, int tid, int nthreads)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < (n >> 1); idx += nthreads) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const K x = a[i], y = a[p];
                if ((x > y) == ((i & k) == 0)) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(1024) void rank_rows_sorted_kernel(const float* __restrict__ scores, int n_items, int64_t ld,
                                                               int32_t* __restrict__ ranks, int64_t ld_out, int NP)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_sh[];
    uint32_t* keys = rank_sh;                                    // [NP] ascending after the sort; padding = 0 in front
    unsigned long long* ties = (unsigned long long*)(rank_sh + NP);      // [RANK_TIE_CAP] (class << 32) | item
    int* n_ties = (int*)(ties + RANK_TIE_CAP);
    const int tid = threadIdx.x, nt = blockDim.x;
    const float* row = scores + (int64_t)blockIdx.x * ld;
    int32_t* out = ranks + (int64_t)blockIdx.x * ld_out;
    for (int q = tid; q < NP; q += nt) keys[q] = q < n_items ? rank_key(row[q]) : 0u;
    if (tid == 0) *n_ties = 0;
    __syncthreads();
    bitonic_sort_lds(keys, NP, tid, nt);
    for (int i = tid; i < n_items; i += nt) {
        const uint32_t k = rank_key(row[i]);
        int lo = 0, hi = NP;                                     // upper bound: first position with keys[pos] > k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] <= k) lo = mid + 1; else hi = mid;
        }
        if (lo >= 2 && keys[lo - 2] == k) {                      // another item has the same score
            const int slot = atomicAdd(n_ties, 1);
            if (slot < RANK_TIE_CAP) ties[slot] = ((unsigned long long)(uint32_t)lo << 32) | (uint32_t)i;
        } else {
            out[i] = 1 + NP - lo;
        }
    }
    __syncthreads();
    const int T = *n_ties;
    if (T == 0) return;
    if (T > RANK_TIE_CAP) {                                      // ties everywhere: recount the whole row (rank_rows_kernel<true>)
        for (int i = tid; i < n_items; i += nt) out[i] = -1;
        return;
    }
    int TP = 2;
    while (TP < T) TP <<= 1;
    for (int q = T + tid; q < TP; q += nt) ties[q] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(ties, TP, tid, nt);
    for (int p = tid; p < T; p += nt) {
        const unsigned long long e = ties[p];
        const unsigned long long cls = e & 0xffffffff00000000ull;
        int lo = 0, hi = p;                                      // first entry of this class
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ties[mid] < cls) lo = mid + 1; else hi = mid;
        }
        out[(uint32_t)e] = 1 + NP - (int)(e >> 32) + (p - lo);
    }
}

// ---- rows longer than RANK_SORT_MAX: sorted CHUNKS ---------------------------------------------------------------
// rank[i] - 1 = sum over chunks c of #{j in c : j beats i}.  Inside i's own chunk that is the sorted-chunk position
// (ties through the (class, index) list, as above).  For another chunk the index order is decided by the chunk order
// alone -- every item of a LOWER chunk wins a tie, none of a HIGHER chunk does -- so the count is NP_c - lower_bound(k)
// resp. NP_c - upper_bound(k) in that chunk's sorted keys: no index data, no 64-bit keys.
//   pass A  rank_chunk_sort_kernel   grid (chunks, users): sort the chunk's keys in LDS, write the in-chunk ranks and the
//           sorted keys (workspace [users][chunks][RANK_SORT_MAX]); a chunk with > RANK_TIE_CAP tied items flags its row;
//   pass B  rank_chunk_merge_kernel  grid (chunks, users): the chunk's items (32 per thread, keys in registers) binary-
//           search every OTHER chunk of the row, staged through LDS one chunk at a time; flagged rows are set to -1
//           and recounted by rank_rows_kernel<true>.
__device__ __forceinline__ int rank_chunk_np(int64_t n_items, int c)
{
    const int64_t left = n_items - (int64_t)c * RANK_SORT_MAX;
    const int len = left < RANK_SORT_MAX ? (int)left : RANK_SORT_MAX;
    int NP = 64;
    while (NP < len) NP <<= 1;
    return NP;
}

__global__ __launch_bounds__(1024) void rank_chunk_sort_kernel(const float* __restrict__ scores, int64_t n_items, int64_t ld,
                                                              int32_t* __restrict__ ranks, int64_t ld_out,
                                                              uint32_t* __restrict__ ws_keys, int n_chunks,
                                                              int32_t* __restrict__ heavy)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_sh[];
    const int c = blockIdx.x;
    const int64_t u = blockIdx.y;
    const int NP = rank_chunk_np(n_items, c);
    const int64_t i0 = (int64_t)c * RANK_SORT_MAX;
    const int len = (n_items - i0 < RANK_SORT_MAX) ? (int)(n_items - i0) : RANK_SORT_MAX;
    uint32_t* keys = rank_sh;
    unsigned long long* ties = (unsigned long long*)(rank_sh + RANK_SORT_MAX);
    int* n_ties = (int*)(ties + RANK_TIE_CAP);
    const int tid = threadIdx.x, nt = blockDim.x;
    const float* row = scores + u * ld + i0;
    int32_t* out = ranks + u * ld_out + i0;
    for (int q = tid; q < NP; q += nt) keys[q] = q < len ? rank_key(row[q]) : 0u;
    if (tid == 0) *n_ties = 0;
    __syncthreads();
    bitonic_sort_lds(keys, NP, tid, nt);
    uint32_t* wsk = ws_keys + ((int64_t)u * n_chunks + c) * RANK_SORT_MAX;
    for (int q = tid; q < NP; q += nt) wsk[q] = keys[q];
    for (int i = tid; i < len; i += nt) {
        const uint32_t k = rank_key(row[i]);
        int lo = 0, hi = NP;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] <= k) lo = mid + 1; else hi = mid;
        }
        if (lo >= 2 && keys[lo - 2] == k) {
            const int slot = atomicAdd(n_ties, 1);
            if (slot < RANK_TIE_CAP) ties[slot] = ((unsigned long long)(uint32_t)lo << 32) | (uint32_t)i;
        } else {
            out[i] = 1 + NP - lo;
        }
    }
    __syncthreads();
    const int T = *n_ties;
    if (T == 0) return;
    if (T > RANK_TIE_CAP) {
        if (tid == 0) heavy[u] = 1;
        return;
    }
    int TP = 2;
    while (TP < T) TP <<= 1;
    for (int q = T + tid; q < TP; q += nt) ties[q] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(ties, TP, tid, nt);
    for (int p = tid; p < T; p += nt) {
        const unsigned long long e = ties[p];
        const unsigned long long cls = e & 0xffffffff00000000ull;
        int lo = 0, hi = p;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ties[mid] < cls) lo = mid + 1; else hi = mid;
        }
        out[(uint32_t)e] = 1 + NP - (int)(e >> 32) + (p - lo);
    }
}

#define RANK_MERGE_Q (RANK_SORT_MAX / 1024)      // query items per thread
__global__ __launch_bounds__(1024) void rank_chunk_merge_kernel(const float* __restrict__ scores, int64_t n_items,
                                                               int64_t ld, int32_t* __restrict__ ranks, int64_t ld_out,
                                                               const uint32_t* __restrict__ ws_keys, int n_chunks,
                                                               const int32_t* __restrict__ heavy)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_sh[];
    const int own = blockIdx.x;
    const int64_t u = blockIdx.y;
    const int64_t i0 = (int64_t)own * RANK_SORT_MAX;
    const int len = (n_items - i0 < RANK_SORT_MAX) ? (int)(n_items - i0) : RANK_SORT_MAX;
    const int tid = threadIdx.x;
    int32_t* out = ranks + u * ld_out + i0;
    if (heavy[u]) {                                           // recounted by rank_rows_kernel<true>
        for (int i = tid; i < len; i += 1024) out[i] = -1;
        return;
    }
    const float* row = scores + u * ld + i0;
    uint32_t qk[RANK_MERGE_Q];
    int add[RANK_MERGE_Q];
#pragma unroll
    for (int e = 0; e < RANK_MERGE_Q; ++e) {
        const int i = tid + 1024 * e;
        qk[e] = i < len ? rank_key(row[i]) : 0xffffffffu;
        add[e] = 0;
    }
    for (int c = 0; c < n_chunks; ++c) {
        if (c == own) continue;
        const int NP = rank_chunk_np(n_items, c);
        const uint32_t* wsk = ws_keys + ((int64_t)u * n_chunks + c) * RANK_SORT_MAX;
        __syncthreads();
        for (int q = tid * 4; q < NP; q += 4096) *(u32x4*)(rank_sh + q) = *(const u32x4*)(wsk + q);
        __syncthreads();
        const bool lower = c < own;                           // items of a lower chunk win ties
#pragma unroll
        for (int e = 0; e < RANK_MERGE_Q; ++e) {
            const uint32_t k = qk[e];
            int lo = 0, hi = NP;                              // lower: first pos with key >= k;  else: first pos with key > k
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint32_t v = rank_sh[mid];
                const bool right = lower ? (v < k) : (v <= k);
                if (right) lo = mid + 1; else hi = mid;
            }
            add[e] += NP - lo;
        }
    }
#pragma unroll
    for (int e = 0; e < RANK_MERGE_Q; ++e) {
        const int i = tid + 1024 * e;
        if (i < len) out[i] += add[e];
    }
}

// one wave per (user, item) pair; counts over items [begin, end) of the user's score row.
// out[p] (+)= count  (+1 added by the caller once all shards are summed, or here when add_one != 0)
__global__ __launch_bounds__(256) void rank_of_pairs_kernel(const float* __restrict__ scores, int64_t ld,
                                                           int64_t col_offset, int64_t begin, int64_t end,
                                                           const int32_t* __restrict__ xu,
                                                           const int32_t* __restrict__ xi,
                                                           const float* __restrict__ target_scores, int64_t n_pairs,
                                                           int add_one, int32_t* __restrict__ out)
{
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (p >= n_pairs) return;
    const int lane = lane_id();
    const int64_t u = xu[p], i = xi[p];
    const float si = target_scores[p];
    const float* row = scores + u * ld - col_offset;     // row[j] valid for j in [begin, end)
    int cnt = 0;
    for (int64_t j = begin + lane; j < end; j += 64) {
        const float s = row[j];
        cnt += (s > si) || (s == si && j < i);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) out[p] = cnt + (add_one ? 1 : 0);
}

// Pairs grouped by user (pair_indptr[u] .. pair_indptr[u+1] are user u's pairs): a workgroup owns (user, item slice) and
// reads the slice ONCE for all of the user's targets, eight at a time in registers -- the wave-per-pair kernel above
// streams the user's row once per pair (20 positives per user over 1M items: 80 MB per user instead of 4).  Counts are
// integers, so the slices' partial counts are added with atomics without losing exactness; out must be zeroed (the +1
// is added by slice 0).
#define RANK_BU_TGT 8
__global__ __launch_bounds__(256) void rank_of_pairs_by_user_kernel(
    const float* __restrict__ scores, int64_t ld, int64_t col_offset, int64_t begin, int64_t end, int64_t slice_len,
    const int64_t* __restrict__ pair_indptr, const int32_t* __restrict__ xi, const float* __restrict__ target_scores,
    int add_one, int32_t* __restrict__ out)
{
    __shared__ int part[4][RANK_BU_TGT];
    const int64_t u = blockIdx.y;
    const int64_t p0 = pair_indptr[u], p1 = pair_indptr[u + 1];
    if (p0 == p1) return;
    const int64_t b = begin + (int64_t)blockIdx.x * slice_len;
    const int64_t e = b + slice_len < end ? b + slice_len : end;
    if (b >= e && !(add_one && blockIdx.x == 0)) return;          // (an empty range still owes the +1)
    const float* row = scores + u * ld - col_offset;              // row[j] valid for j in [begin, end)
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int64_t q0 = p0; q0 < p1; q0 += RANK_BU_TGT) {
        float tv[RANK_BU_TGT];
        int64_t ti[RANK_BU_TGT];
        int cnt[RANK_BU_TGT];
#pragma unroll
        for (int t = 0; t < RANK_BU_TGT; ++t) {
            const bool v = q0 + t < p1;
            tv[t] = v ? target_scores[q0 + t] : INFINITY;         // nothing beats +inf with index -1: count stays 0
            ti[t] = v ? (int64_t)xi[q0 + t] : -1;
            cnt[t] = 0;
        }
        for (int64_t j = b + threadIdx.x; j < e; j += 256) {
            const float s = row[j];
#pragma unroll
            for (int t = 0; t < RANK_BU_TGT; ++t) cnt[t] += (s > tv[t]) || (s == tv[t] && j < ti[t]);
        }
#pragma unroll
        for (int t = 0; t < RANK_BU_TGT; ++t) {
            int c = cnt[t];
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            if (lane == 0) part[w][t] = c;
        }
        __syncthreads();
        if (threadIdx.x < RANK_BU_TGT && q0 + threadIdx.x < p1) {
            const int t = threadIdx.x;
            const int total = part[0][t] + part[1][t] + part[2][t] + part[3][t] + ((add_one && blockIdx.x == 0) ? 1 : 0);
            if (total) atomicAdd(out + q0 + t, total);
        }
        __syncthreads();
    }
}

extern "C" int trec_rank_rows(const float* scores, int64_t n_users, int64_t n_items, int64_t ld_scores,
                              int32_t* ranks, int64_t ld_ranks, void* stream)
{
    TREC_REQUIRE(scores && ranks, "trec_rank_rows: null pointer");
    TREC_REQUIRE(ld_scores >= n_items && ld_ranks >= n_items, "trec_rank_rows: leading dimension < n_items");
    TREC_REQUIRE(n_users <= 65535 * (int64_t)65535, "trec_rank_rows: too many users for one launch");
    if (n_users == 0 || n_items == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned bx = (unsigned)ceil_div64(n_items, RANK_TGT);
    bool sorted = n_items >= 64 && n_items <= RANK_SORT_MAX && n_users < (1ll << 31) && trec_get_tuning("rank_sorted", 1);
    if (sorted) {
        int NP = 64;
        while (NP < n_items) NP <<= 1;
        const size_t lds = (size_t)NP * 4 + RANK_TIE_CAP * 8 + 16;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute((const void*)rank_rows_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            sorted = false;                                      // cannot have the LDS: count instead
        } else {
            const int threads = NP / 2 < 1024 ? NP / 2 : 1024;
            hipLaunchKernelGGL(rank_rows_sorted_kernel, dim3((unsigned)n_users), dim3(threads), lds, st, scores,
                               (int)n_items, ld_scores, ranks, ld_ranks, NP);
        }
    }
    // gridDim.y is limited to 65535: launch in user slabs
    for (int64_t u0 = 0; u0 < n_users; u0 += 65535) {
        const unsigned by = (unsigned)((n_users - u0 < 65535) ? (n_users - u0) : 65535);
        if (sorted)
            hipLaunchKernelGGL(rank_rows_kernel<true>, dim3(bx, by), dim3(256), 0, st, scores + u0 * ld_scores, n_items,
                               ld_scores, ranks + u0 * ld_ranks, ld_ranks);
        else
            hipLaunchKernelGGL(rank_rows_kernel<false>, dim3(bx, by), dim3(256), 0, st, scores + u0 * ld_scores, n_items,
                               ld_scores, ranks + u0 * ld_ranks, ld_ranks);
    }
    return trec_check_launch("trec_rank_rows");
}

// workspace of trec_rank_rows_chunked for a slab of n_users rows: sorted keys of every chunk + one flag per row
extern "C" int64_t trec_rank_rows_workspace_bytes(int64_t n_users, int64_t n_items)
{
    if (n_items <= RANK_SORT_MAX) return 0;
    const int64_t n_chunks = ceil_div64(n_items, RANK_SORT_MAX);
    return n_users * n_chunks * RANK_SORT_MAX * 4 + n_users * 4;
}

// rows of more than RANK_SORT_MAX items: sorted chunks + cross-chunk binary searches (see above)
extern "C" int trec_rank_rows_chunked(const float* scores, int64_t n_users, int64_t n_items, int64_t ld_scores,
                                      int32_t* ranks, int64_t ld_ranks, void* workspace, int64_t workspace_bytes,
                                      void* stream)
{
    TREC_REQUIRE(scores && ranks && workspace, "trec_rank_rows_chunked: null pointer");
    TREC_REQUIRE(ld_scores >= n_items && ld_ranks >= n_items, "trec_rank_rows_chunked: leading dimension < n_items");
    TREC_REQUIRE(n_items > RANK_SORT_MAX && n_users <= 65535, "trec_rank_rows_chunked: need n_items > 32768, n_users <= 65535");
    TREC_REQUIRE(workspace_bytes >= trec_rank_rows_workspace_bytes(n_users, n_items), "trec_rank_rows_chunked: workspace too small");
    TREC_REQUIRE(((uintptr_t)workspace % 16) == 0, "trec_rank_rows_chunked: workspace must be 16-byte aligned");
    if (n_users == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int n_chunks = (int)ceil_div64(n_items, RANK_SORT_MAX);
    uint32_t* ws_keys = (uint32_t*)workspace;
    int32_t* heavy = (int32_t*)(ws_keys + n_users * n_chunks * (int64_t)RANK_SORT_MAX);
    if (hipMemsetAsync(heavy, 0, sizeof(int32_t) * (size_t)n_users, st) != hipSuccess) {
        trec_set_last_error("trec_rank_rows_chunked: memset failed");
        return TREC_ERR_LAUNCH;
    }
    const size_t lds_a = (size_t)RANK_SORT_MAX * 4 + RANK_TIE_CAP * 8 + 16, lds_b = (size_t)RANK_SORT_MAX * 4;
    if (hipFuncSetAttribute((const void*)rank_chunk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess ||
        hipFuncSetAttribute((const void*)rank_chunk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) != hipSuccess) {
        (void)hipGetLastError();
        trec_set_last_error("trec_rank_rows_chunked: cannot reserve the LDS");
        return TREC_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(rank_chunk_sort_kernel, dim3((unsigned)n_chunks, (unsigned)n_users), dim3(1024), lds_a, st, scores,
                       n_items, ld_scores, ranks, ld_ranks, ws_keys, n_chunks, heavy);
    hipLaunchKernelGGL(rank_chunk_merge_kernel, dim3((unsigned)n_chunks, (unsigned)n_users), dim3(1024), lds_b, st, scores,
                       n_items, ld_scores, ranks, ld_ranks, ws_keys, n_chunks, heavy);
    const unsigned bx = (unsigned)ceil_div64(n_items, RANK_TGT);
    hipLaunchKernelGGL(rank_rows_kernel<true>, dim3(bx, (unsigned)n_users), dim3(256), 0, st, scores, n_items, ld_scores,
                       ranks, ld_ranks);
    return trec_check_launch("trec_rank_rows_chunked");
}

extern "C" int trec_rank_of_pairs(const float* scores, int64_t ld_scores, int64_t col_offset, int64_t begin,
                                  int64_t end, const int32_t* xu, const int32_t* xi, const float* target_scores,
                                  int64_t n_pairs, int32_t add_one, int32_t* out, void* stream)
{
    TREC_REQUIRE(scores && xu && xi && target_scores && out, "trec_rank_of_pairs: null pointer");
    TREC_REQUIRE(begin <= end, "trec_rank_of_pairs: begin > end");
    if (n_pairs == 0) return TREC_OK;
    hipLaunchKernelGGL(rank_of_pairs_kernel, dim3((unsigned)ceil_div64(n_pairs * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, scores, ld_scores, col_offset, begin, end, xu, xi, target_scores, n_pairs,
                       add_one, out);
    return trec_check_launch("trec_rank_of_pairs");
}

extern "C" int trec_rank_of_pairs_by_user(const float* scores, int64_t ld_scores, int64_t col_offset, int64_t begin,
                                          int64_t end, const int64_t* pair_indptr, const int32_t* xi,
                                          const float* target_scores, int64_t n_users, int64_t n_pairs, int32_t add_one,
                                          int32_t* out, void* stream)
{
    TREC_REQUIRE(scores && pair_indptr && out, "trec_rank_of_pairs_by_user: null pointer");
    TREC_REQUIRE(n_pairs == 0 || (xi && target_scores), "trec_rank_of_pairs_by_user: null pair arrays");
    TREC_REQUIRE(begin <= end, "trec_rank_of_pairs_by_user: begin > end");
    if (n_pairs == 0 || n_users == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(int32_t) * (size_t)n_pairs, st) != hipSuccess) {
        trec_set_last_error("trec_rank_of_pairs_by_user: memset failed");
        return TREC_ERR_LAUNCH;
    }
    if (begin == end && !add_one) return TREC_OK;                // an empty item range contributes only the +1
    // enough (user, slice) workgroups to fill the chip, slices of at least 4096 items
    const int64_t len = end - begin > 0 ? end - begin : 1;
    int64_t n_slices = ceil_div64(4096, n_users);
    const int64_t max_slices = ceil_div64(len, 4096);
    if (n_slices > max_slices) n_slices = max_slices;
    if (n_slices < 1) n_slices = 1;
    const int64_t slice_len = ceil_div64(len, n_slices);
    n_slices = ceil_div64(len, slice_len);
    for (int64_t u0 = 0; u0 < n_users; u0 += 65535) {            // gridDim.y is limited to 65535
        const unsigned by = (unsigned)((n_users - u0 < 65535) ? (n_users - u0) : 65535);
        hipLaunchKernelGGL(rank_of_pairs_by_user_kernel, dim3((unsigned)n_slices, by), dim3(256), 0, st,
                           scores + u0 * ld_scores, ld_scores, col_offset, begin, end, slice_len, pair_indptr + u0, xi,
                           target_scores, add_one, out);
    }
    return trec_check_launch("trec_rank_of_pairs_by_user");
}
