// tensorrec_amd/csrc/score_rank.hip -- K2r: exact ranks of chosen (user, item) pairs WITHOUT a score slab.
//
// rank_predictions (tensorrec/recommendation_graphs.py:73-82) is a count:
//     rank[u, i] = 1 + #{ j : s[u,j] > s[u,i]  or  (s[u,j] == s[u,i] and j < i) }
// and evaluation (tensorrec/eval.py) only needs it at the positive test pairs.  trec_rank_of_pairs counts over a
// [users, items] fp32 slab in HBM -- 4 TB of traffic per million users at 1M items.  Here the count is the EPILOGUE of
// the fp32 MFMA score kernel: the 32 x 32 score tile is compared with the targets of its users while it is in
// registers, so nothing of size U x I exists.  Scores are the exact k-ordered fp32 fmaf chain
// (v_mfma_f32_32x32x2_f32, bit-identical to oracle/tr_oracle.c:orc_score_dense) with the biases added in the
// reference's order (s + b_u) + b_i, or the Euclidean transform of prediction_graphs.py:84-100; the target scores come
// from trec_pair_score_exact, the same chain one pair per thread, so "==" between a tile score and a target is exact.
// Counts over disjoint item ranges add (item chunks here, item shards across ranks: sharding.reduce_rank_counts).
//
// Layout: lane & 31 = resident row (a user, or one group of <= RANKC_QMAX targets of a user: users with more targets
// occupy several rows through row_user), the lane's 16 accumulator registers = 16 items of that row; the two half-waves
// hold different items of the same row and add their counts at the end.  A target costs 16 compares + 16 adds per
// 32-item block against 64 fp32 MFMAs (4096 cycles) -- up to ~30 targets per row ride under the MFMA time.
#include "topk_common.hpp"

#define RANKC_QMAX 32

struct RankCountParams {
    const float* U;            // [n_users, KT] fp32 operand (zero padded, cosine: normalised)
    const float* T;            // [n_items, KT]
    int64_t n_rows, n_items;
    int64_t chunk_len;         // items per chunk (multiple of 32)
    int n_rblocks;
    const int32_t* row_user;   // [n_rows] operand row of resident row r
    const int32_t* row_t0;     // [n_rows] first target (index into tgt_*) of row r
    const int32_t* row_tn;     // [n_rows] number of targets of row r (<= RANKC_QMAX)
    const int32_t* tgt_item;   // [n_pairs] GLOBAL item id of a target
    const float* tgt_score;    // [n_pairs] its exact score
    const float* u_bias;       // nullable [n_users]
    const float* t_bias;       // nullable [n_items]
    const float* u_sq;         // euclid [n_users]
    const float* t_sq;         // euclid [n_items]
    int32_t item_index_base;   // global id of item row 0
    int32_t* counts;           // [n_pairs], += number of items of this launch's range that beat the target
};

__device__ __forceinline__ int rc_cd_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int KT, bool EUCLID, bool BIAS>
__global__ __launch_bounds__(256, (KT >= 256 ? 1 : 2)) void score_rankcount_kernel(RankCountParams p)
{
    constexpr int BN = 32;                  // items per tile = one 32 x 32 MFMA block
    constexpr int RB = KT * 4;              // bytes per operand row
    constexpr int CH = RB / 16;             // 16-byte chunks per row
    constexpr int KS = KT / 2;              // MFMA k-steps (v_mfma_f32_32x32x2_f32)
    constexpr int TILE_BYTES = BN * RB;
    constexpr int NSLOT = BN * CH / 256;    // 16-byte staging slots per thread per tile
    static_assert(NSLOT >= 1, "tile too small for 256 threads");
    extern __shared__ __attribute__((aligned(16))) char rsmem[];
    float* side = (float*)(rsmem + 2 * TILE_BYTES);        // [2][2 * BN]: item bias, item squared norm

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int rblock = blockIdx.x % p.n_rblocks;
    const int chunk = blockIdx.x / p.n_rblocks;
    const int64_t row = ((int64_t)rblock * 4 + wave) * 32 + l31;
    const bool row_ok = row < p.n_rows;
    const int64_t src = row_ok ? p.row_user[row] : 0;
    const int t0 = row_ok ? p.row_t0[row] : 0;
    const int tn = row_ok ? p.row_tn[row] : 0;
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_items) ? t_begin + p.chunk_len : p.n_items;
    const int n_tiles = (int)((t_end - t_begin + BN - 1) / BN);
    if (n_tiles <= 0) return;

    float rff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rff[ks] = p.U[src * KT + 2 * ks + half];
    const float ub = (BIAS && p.u_bias) ? p.u_bias[src] : 0.f;
    const float usq = EUCLID ? p.u_sq[src] : 0.f;

    // targets of my row: scores and item ids in LDS ([q][32 rows] per wave: both half-waves read the same word), counts
    // in registers
    float* tq_lds = side + 2 * 2 * BN + wave * (2 * RANKC_QMAX * 32);
    int32_t* iq_lds = (int32_t*)(tq_lds + RANKC_QMAX * 32);
    int32_t cnt[RANKC_QMAX];
#pragma unroll
    for (int q = 0; q < RANKC_QMAX; ++q) {
        const bool v = q < tn;
        if (half == 0) {
            tq_lds[q * 32 + l31] = v ? p.tgt_score[t0 + q] : INFINITY;     // nothing beats (+inf, index -1): the count stays 0
            iq_lds[q * 32 + l31] = v ? p.tgt_item[t0 + q] : -1;
        }
        cnt[q] = 0;
    }
    int qmax = tn;                                           // wave-uniform number of target slots in use
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(qmax, off, 64); qmax = o > qmax ? o : qmax; }
    qmax = __builtin_amdgcn_readfirstlane(qmax);

    // ---- staging: slot q = i * 256 + tid -> (row, physical 16-byte chunk), XOR-swizzled as in score_gemm.hip
    // LDS position (row, pc) = slot q * 16 bytes holds the row's LOGICAL chunk pc ^ swz(row); readers undo it
    int slot_row[NSLOT], slot_src[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * 256 + tid;
        const int r = q / CH, pc = q % CH;
        const int sw = (CH >= 16) ? (r & 15) : ((CH == 8) ? ((r >> 1) & 7) : 0);
        slot_row[i] = r;
        slot_src[i] = (pc ^ sw) * 16;
    }
    u32x4 stage[NSLOT];
    float side_b = 0.f, side_q = 0.f;
    auto stage_issue = [&](int tile) {
        const int64_t row0 = t_begin + (int64_t)tile * BN;
        if (tid < BN) {
            int64_t g = row0 + tid;
            if (g >= p.n_items) g = p.n_items - 1;
            side_b = (BIAS && p.t_bias) ? p.t_bias[g] : 0.f;
            side_q = EUCLID ? p.t_sq[g] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int64_t g = row0 + slot_row[i];
            if (g >= p.n_items) g = p.n_items - 1;          // clamped rows are masked in the epilogue
            stage[i] = *(const u32x4*)((const char*)p.T + g * (int64_t)RB + slot_src[i]);
        }
    };
    auto stage_commit = [&](int buf) {
        if (tid < BN) { side[buf * 2 * BN + tid] = side_b; side[buf * 2 * BN + BN + tid] = side_q; }
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) *(u32x4*)(rsmem + buf * TILE_BYTES + (i * 256 + tid) * 16) = stage[i];
    };

    stage_issue(0);
    stage_commit(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1);
        const char* rowp = rsmem + buf * TILE_BYTES + l31 * RB;
        const int sw = (CH >= 16) ? (l31 & 15) : ((CH == 8) ? ((l31 >> 1) & 7) : 0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 2 * ks + half;
            const float tf = *(const float*)(rowp + (((k >> 2) ^ sw) * 16) + (k & 3) * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tf, rff[ks], acc, 0, 0, 0);
        }
        // ---- scores of my row against items blk0 + rc_cd_row(r, half), in the reference's arithmetic
        const float* sd = side + buf * 2 * BN;
        const int64_t loc0 = t_begin + (int64_t)t * BN;                       // local index of the block's first item
        const int rows_left = (int)((p.n_items - loc0 < BN) ? (p.n_items - loc0) : BN);
        float s[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 tb4 = {0.f, 0.f, 0.f, 0.f}, ts4 = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) tb4 = *(const f32x4*)(sd + 8 * q4 + 4 * half);
            if (EUCLID) ts4 = *(const f32x4*)(sd + BN + 8 * q4 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * q4 + e];
                if (EUCLID) {
                    float dist = (usq - 2.0f * v) + ts4[e];
                    dist = fmaxf(dist, 1e-16f);
                    v = -1.0f * sqrtf(dist);
                }
                if (BIAS) v = (v + ub) + tb4[e];
                if (8 * q4 + 4 * half + e >= rows_left) v = -INFINITY;      // rows past the end never beat a finite target
                s[4 * q4 + e] = v;
            }
        }
        const int32_t blk0 = (int32_t)loc0 + p.item_index_base, blk1 = blk0 + BN;
#pragma unroll
        for (int q = 0; q < RANKC_QMAX; ++q) {
            if (q < qmax) {                                                     // wave-uniform
                const float tv = tq_lds[q * 32 + l31];
                const int32_t ti = iq_lds[q * 32 + l31];
                const bool inside = ti >= blk0 && ti < blk1;
                // every item of a block below the target's index wins a tie: s >= t  <=>  s > float_pred(t)
                const float thr = (ti >= blk1) ? float_pred(tv) : tv;
                int c = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) c += (s[r] > thr) ? 1 : 0;
                if (tv == -INFINITY && ti >= blk1) {
                    // a target scoring -inf ties with every -inf item, and float_pred(-inf) = -inf cannot express ">=": in a
                    // block wholly below the target's index EVERY valid row beats or ties-and-precedes it (rows
                    // past the end, set to -inf above, do not count).  NaN scores stay outside the guarantee.
                    c = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c += (rc_cd_row(r, half) < rows_left) ? 1 : 0;
                }
                if (__builtin_amdgcn_ballot_w64(inside) != 0ull) {              // the block holding the target itself
                    if (inside) {
                        c = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int32_t item = blk0 + rc_cd_row(r, half);
                            const bool valid = rc_cd_row(r, half) < rows_left;
                            c += (valid && ((s[r] > tv) || (s[r] == tv && item < ti))) ? 1 : 0;
                        }
                    }
                }
                cnt[q] += c;
            }
        }
        if (t + 1 < n_tiles) stage_commit(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < RANKC_QMAX; ++q) {
        if (q < qmax) {
            const int c = cnt[q] + __shfl_xor(cnt[q], 32, 64);
            if (half == 0 && q < tn && c) atomicAdd(p.counts + t0 + q, c);
        }
    }
}

// one thread per pair: the k-ordered fp32 fmaf chain of the MFMA kernel (and of the oracle), biases / Euclidean
// transform in the reference's order -- the scores the rank counts compare against
__global__ __launch_bounds__(256) void pair_score_exact_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                              int64_t ld, int kdim, const int32_t* __restrict__ xu,
                                                              const int32_t* __restrict__ xi, int64_t n_pairs,
                                                              const float* __restrict__ u_bias,
                                                              const float* __restrict__ t_bias, int euclid,
                                                              const float* __restrict__ u_sq, const float* __restrict__ t_sq,
                                                              int32_t item_index_base, float* __restrict__ out)
{
    const int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pidx >= n_pairs) return;
    const int64_t u = xu[pidx], i = (int64_t)xi[pidx] - item_index_base;
    const float* a = U + u * ld;
    const float* b = V + i * ld;
    float acc = 0.0f;
    int k = 0;
    if ((ld & 3) == 0) {
        for (; k + 4 <= kdim; k += 4) {
            const f32x4 a4 = *(const f32x4*)(a + k);
            const f32x4 b4 = *(const f32x4*)(b + k);
            acc = __fmaf_rn(a4[0], b4[0], acc); acc = __fmaf_rn(a4[1], b4[1], acc);
            acc = __fmaf_rn(a4[2], b4[2], acc); acc = __fmaf_rn(a4[3], b4[3], acc);
        }
    }
    for (; k < kdim; ++k) acc = __fmaf_rn(a[k], b[k], acc);
    if (euclid) {
        float dist = (u_sq[u] - 2.0f * acc) + t_sq[i];
        dist = fmaxf(dist, 1e-16f);
        acc = -1.0f * sqrtf(dist);
    }
    if (u_bias) acc = acc + u_bias[u];
    if (t_bias) acc = acc + t_bias[i];
    out[pidx] = acc;
}

extern "C" int trec_pair_score_exact(const float* users_f32, const float* items_f32, int64_t ld, int32_t kdim,
                                     const int32_t* xu, const int32_t* xi, int64_t n_pairs, const float* user_bias,
                                     const float* item_bias, int32_t mode, const float* user_sqnorm,
                                     const float* item_sqnorm, int32_t item_index_base, float* out, void* stream)
{
    TREC_REQUIRE(users_f32 && items_f32 && out, "trec_pair_score_exact: null pointer");
    TREC_REQUIRE(kdim >= 1 && ld >= kdim, "trec_pair_score_exact: need 1 <= kdim <= ld");
    TREC_REQUIRE(mode == 0 || (user_sqnorm && item_sqnorm), "trec_pair_score_exact: euclidean mode needs squared norms");
    if (n_pairs == 0) return TREC_OK;
    TREC_REQUIRE(xu && xi, "trec_pair_score_exact: null pair arrays");
    hipLaunchKernelGGL(pair_score_exact_kernel, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0, (hipStream_t)stream,
                       users_f32, items_f32, ld, kdim, xu, xi, n_pairs, user_bias, item_bias, mode, user_sqnorm,
                       item_sqnorm, item_index_base, out);
    return trec_check_launch("trec_pair_score_exact");
}

template <int KT, bool EUCLID>
static int launch_rankcount(const RankCountParams& p, int n_chunks, hipStream_t st)
{
    constexpr int LDS = 2 * 32 * KT * 4 + 2 * 2 * 32 * 4 + 4 * 2 * RANKC_QMAX * 32 * 4;
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)n_chunks;
    if (p.u_bias || p.t_bias) {
        auto kern = score_rankcount_kernel<KT, EUCLID, true>;
        if (LDS > 32 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    } else {
        auto kern = score_rankcount_kernel<KT, EUCLID, false>;
        if (LDS > 32 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    }
    return trec_check_launch("trec_score_gemm_rankcount");
}

extern "C" int trec_score_rankcount_max_targets(void) { return RANKC_QMAX; }

extern "C" int trec_score_gemm_rankcount(const float* users_f32, const float* items_f32, int32_t kpad, int64_t n_rows,
                                         int64_t n_items, int32_t item_index_base, const float* user_bias,
                                         const float* item_bias, int32_t mode, const float* user_sqnorm,
                                         const float* item_sqnorm, const int32_t* row_user, const int32_t* row_t0,
                                         const int32_t* row_tn, const int32_t* tgt_item, const float* tgt_score,
                                         int32_t n_chunks, int32_t* counts, void* stream)
{
    TREC_REQUIRE(users_f32 && items_f32 && row_user && row_t0 && row_tn && tgt_item && tgt_score && counts,
                 "trec_score_gemm_rankcount: null pointer");
    TREC_REQUIRE(kpad == 32 || kpad == 64 || kpad == 128 || kpad == 256, "trec_score_gemm_rankcount: kpad must be 32/64/128/256");
    TREC_REQUIRE(mode == 0 || mode == 1, "trec_score_gemm_rankcount: mode must be 0 (dot) or 1 (euclidean)");
    TREC_REQUIRE(mode == 0 || (user_sqnorm && item_sqnorm), "trec_score_gemm_rankcount: euclidean mode needs squared norms");
    TREC_REQUIRE(n_items < ((int64_t)1 << 31) && n_rows < ((int64_t)1 << 31), "trec_score_gemm_rankcount: sizes must fit int32");
    if (n_rows == 0 || n_items == 0) return TREC_OK;
    RankCountParams p = {};
    p.U = users_f32; p.T = items_f32; p.n_rows = n_rows; p.n_items = n_items;
    p.n_rblocks = (int)ceil_div64(n_rows, 128);
    if (n_chunks < 1) {                                  // enough workgroups for ~4 rounds of 2 per CU
        n_chunks = 1;
        while ((int64_t)p.n_rblocks * n_chunks < 2048 && ceil_div64(n_items, n_chunks * 2) >= 1024) n_chunks *= 2;
    }
    p.chunk_len = ceil_div64(ceil_div64(n_items, n_chunks), 32) * 32;
    n_chunks = (int)ceil_div64(n_items, p.chunk_len);
    p.row_user = row_user; p.row_t0 = row_t0; p.row_tn = row_tn; p.tgt_item = tgt_item; p.tgt_score = tgt_score;
    p.u_bias = user_bias; p.t_bias = item_bias; p.u_sq = user_sqnorm; p.t_sq = item_sqnorm;
    p.item_index_base = item_index_base; p.counts = counts;
    hipStream_t st = (hipStream_t)stream;
#define TREC_RC(KTV)                                                              \
    if (kpad == KTV) return mode ? launch_rankcount<KTV, true>(p, n_chunks, st)   \
                                 : launch_rankcount<KTV, false>(p, n_chunks, st);
    TREC_RC(128) TREC_RC(64) TREC_RC(32) TREC_RC(256)
#undef TREC_RC
    return TREC_ERR_UNSUPPORTED;
}
