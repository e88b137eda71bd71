// tensorrec_amd/csrc/score_common.hpp -- launch parameters shared by the score kernels (score_gemm.hip,
// score_blockmax.hip).
#pragma once
#include "common.hpp"

struct ScoreParams {
    const void* R;            // resident operand [n_r, KT] (fp32 or bf16), users
    const void* T;            // streamed operand [n_t, KT], items
    int64_t n_r, n_t;
    int64_t chunk_len;        // streamed rows per chunk (multiple of BN)
    int n_chunks;
    int n_rblocks;
    const float* r_bias;      // nullable [n_r]   user bias   (added first, recommendation_graphs.py:41)
    const float* t_bias;      // nullable [n_t]   item bias   (added second)
    const float* r_sqnorm;    // euclid only [n_r]
    const float* t_sqnorm;    // euclid only [n_t]
    int euclid;
    float* out;               // STORE: [n_r, ld_out]
    int64_t ld_out;
    float* part_vals;         // TOPK: [n_r, n_parts, KTOP]  (KTOP = list capacity: 8, 12 or 16)
    int32_t* part_idx;
    int capacity;             // slots per partial list in part_vals / part_idx (>= the kernel's KTOP)
    int n_parts;              // 2 * n_chunks
    int32_t t_index_base;     // added to item indices written by TOPK (item shards)
    // BLOCKMAX: blockmax[(chunk superblock base + s) * bm_stride + user], superblock = sb_tiles tiles of BN rows
    float* blockmax;
    int64_t bm_stride;
    int sb_tiles;
    // grouped TOPK (stage 3): the item range of a workgroup comes from a table, results go where row_pair says
    const int32_t* rblock_chunk;   // nullable [n_rblocks]: chunk (= superblock) index of this resident block, -1 = idle
    const int32_t* row_pair;       // nullable [n_r]: output list id of a resident row, -1 = padding row
    const float* row_floor;        // nullable [n_r]: a known lower bound of the row's final k-th best score (lists start there)
    const int32_t* row_index;      // grouped TOPK: nullable [n_r], resident row r is R[row_index[r]] (bias / sqnorm / floor too)
    int independent_lists;         // grouped TOPK: a list's threshold never rises from the partner half-wave's list (variant bit 4)
    const float* scales;           // int8 BLOCKMAX (score_blockmax_i8.hip): device float[3] = {user scale (one class), unused, unused}
    const float* sb_stats;         // int8 BLOCKMAX: [n_sb][4] = {item scale b_s, max ||y|| + ||dy||, max ||dy||, max |bias - a b_s bq| / a}
    const float* r_err;            // int8 BLOCKMAX: nullable [n_r][4] = {||x||, ||x - a q||, ck (|b_u| + max |b_i|), a} -> per-chunk top lists
    // int8 BLOCKMAX with user scale CLASSES (users sorted by class; a workgroup's users share one scale):
    const float* wg_scale;         // nullable [n_rblocks]: the user scale of workgroup rblock (else scales[0])
    const int32_t* wg_class;       // nullable [n_rblocks]: its class -> item biases t_bias + wg_class * bias_stride
    int64_t bias_stride;
    int grp_band_major;            // grouped bf16 BLOCKMAX, fixed-capacity layout: workgroup order (list chunk j, superblock s) instead of
                                   // (s, j): the workgroups running together share a band of users (their rows come from L2)
    // bf16 BLOCKMAX that also lists candidates (the cascade's refining launches, topk_candidates.hip): every item whose bf16
    // score reaches cand_floor[user] is appended to cand[user][0 .. cand_cap) = {item id + t_index_base, score bits}
    const float* cand_floor;       // nullable [n users]
    int32_t* cand_n;               // [n users] entries appended so far (may exceed cand_cap: the list is then incomplete)
    int2* cand;                    // [n users][cand_cap]
    int32_t cand_cap;
    const int32_t* wg_map;         // nullable [n_wgs]: grouped launch, fixed-capacity layout: workgroup w of the launch is slot wg_map[w] of the
    int32_t n_wgs;                 //   [n_sb][capacity] grid (trec_topk_rows_wg_map: only the slots that hold rows are launched)
    float* pre_max;                // grouped LIST launch of the PRE-refinement (nullable): the bf16 maximum of resident row r goes to pre_max[r]
                                   // (the list's own layout, next to row_index[r]) and the table entry is marked -inf right here -- the
                                   // threshold kernel behind the launch then never touches the table (trec_topk_prerefine_tau_listed)
    int cand_diag;                 // builds with -DTREC_CAND_DIAG only (tuning cascade_cand_diag): low bits 1 = the queues are emptied without
                                   // looking, 2 = atomics but no stores; +8 no maxima stores, +32 no counter gather, +64 / +128 throttled row gathers
    float* chunk_top;              // int8 BLOCKMAX: [n_chunks * top_k][bm_stride]: the top_k largest LOWER BOUNDS of a chunk per user
    int top_k;
    int top_tag;                   // int8 BLOCKMAX: the lower bounds carry the superblock's index inside its chunk in their low
                                   // TREC_LB_TAG_BITS bits (lb_tag below): the pre-refinement of the cascade needs to know WHICH
                                   // superblocks hold a user's k largest lower bounds (trec_topk_prerefine_rows)
};

// A lower bound with an index in its low bits.  tagged <= lb always (a smaller lower bound is still a lower bound: at most two
// units of 2^-11 relative, ~1 % of the int8 bound itself), tagged values of one chunk are distinct, -inf (nothing certified) stays
// -inf.  Floats are mapped to a monotone signed integer key (negative floats -> negative keys), the key is floored to the multiple
// of 4096 strictly below it and the index added; lb_tag_index recovers it (two's complement: key mod 4096).
#define TREC_LB_TAG_BITS 12
__device__ __forceinline__ int lb_key(float x)
{
    const int b = __float_as_int(x);
    return b >= 0 ? b : (int)(0x80000000u - (unsigned int)b);
}
__device__ __forceinline__ float lb_tag(float lb, int idx)
{
    if (!(fabsf(lb) < 1e37f)) return -INFINITY;                               // -inf, NaN, or too large to move: certifies nothing
    const int k2 = ((lb_key(lb) >> TREC_LB_TAG_BITS) - 1) * (1 << TREC_LB_TAG_BITS) + idx;
    const int b2 = k2 >= 0 ? k2 : (int)(0x80000000u - (unsigned int)k2);
    return __int_as_float(b2);
}
__device__ __forceinline__ int lb_tag_index(float tagged) { return lb_key(tagged) & ((1 << TREC_LB_TAG_BITS) - 1); }

// |int8 score - fp32 score| <= i8_pair_err for every item of a superblock with statistics (yh, dy, db) and a user with
// (nx, ex, cu) -- the bound of csrc/topk_cascade.hip; ONE definition, evaluated identically (no contraction) by the int8
// kernel (lower bounds M - e) and by the compaction (upper bounds M + e)
__device__ __forceinline__ float i8_pair_err(float nx, float ex, float cu, float yh, float dy, float db, int kdim)
{
    // (K + 6) (2^-24 + 2^-22): the reference's own fp32 chain (K + 2), the two roundings of a b_s m + b_u, the int32 -> float
    // conversion of the maximum (exact below 2^24; the integer item bias may reach 2^30) and the bias error's own evaluation
    const float ck = (float)(kdim + 6) * 2.98023224e-07f;
    const float e = nx * (dy + ck * yh) + ex * yh + db + cu;
    return e * 1.001953125f + 1e-30f;
}

// software-pipelined BLOCKMAX kernel (score_blockmax.hip): bf16 dot / cosine, kpad 64 or 128.  Returns
// TREC_ERR_UNSUPPORTED when the configuration is not covered (the caller then uses the generic kernel).
int launch_blockmax_pipelined(const ScoreParams& p, int kt, hipStream_t stream);
// the 16x16x32 bf16 form for the filters (any summation order): TREC_ERR_UNSUPPORTED when switched off / not covered
int launch_blockmax_filter16(const ScoreParams& p, int kt, hipStream_t stream);
// grouped bf16 form (stage 2 of the int8 cascade): see score_blockmax.hip
int launch_blockmax_pipelined_grouped(const ScoreParams& p, int kt, hipStream_t stream);
// the exact fp32 form (kpad 64 or 128, dot / cosine); sb_rows = superblock height in item rows
int launch_blockmax_pipelined_f32(const ScoreParams& p, int kt, int sb_rows, hipStream_t stream);
