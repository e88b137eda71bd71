// tensorrec_amd/csrc/score_gemm.hip -- K2: the user x item score contraction on MFMA with fused epilogues.
//
// Replaces tf.matmul(user_repr, item_repr, transpose_b=True) at tensorrec/prediction_graphs.py:50
// (DotProduct), :94 (Euclidean) and tensorrec/recommendation_graphs.py:121 (relative_cosine), the bias
// broadcast at recommendation_graphs.py:41 and -- in TOPK mode -- the first tf.nn.top_k of
// rank_predictions (recommendation_graphs.py:80) truncated to k, WITHOUT writing the [U, I] matrix.
//
// Shape of the problem: K = n_components is short (<= 256), so there is no K loop over tiles: the
// "resident" operand R (users) lives in registers as MFMA fragments for the whole kernel and the
// "streamed" operand T (items) flows through a double-buffered, XOR-swizzled LDS tile, exactly once
// per workgroup.  Every workgroup is 4 waves (one per SIMD); a wave owns NCB column blocks of 32
// resident rows.  Two workgroups share a CU so one's MFMAs overlap the other's epilogue VALU.
//
//   STORE epilogue  acc = mfma(R, T): rows = users, cols = items  -> lanes run along items, so the
//                   fp32 tile is written with 128-byte row segments (predict()).
//   TOPK  epilogue  acc = mfma(T, R): rows = items, cols = users -> lane & 31 IS the user, so each
//                   lane keeps a sorted top-KTOP list for its user in registers and the common path
//                   is "max of my 16 scores <= my threshold -> skip".  Lists are merged by
//                   trec_topk_merge (wave-shuffle selection, value desc / index asc).
//
// Arithmetic types (runtime `dtype`): 0 = fp32 on v_mfma_f32_32x32x2_f32 -- an exact fp32 fmaf chain
// in k order, bit-identical to oracle/tr_oracle.c:orc_score_dense; 1 = bf16 operands on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulate (the throughput mode named by BASELINE.json).
#include "score_common.hpp"
#include "topk_common.hpp"
#include <math.h>

#define EPI_STORE 0
#define EPI_TOPK 1
#define EPI_BLOCKMAX 2     // per (user, item superblock) maximum only: stage 1 of the two-stage exact top-k
#define KTOP_MAX 16

template <int DT> struct ElemOf;
template <> struct ElemOf<0> { static constexpr int BYTES = 4; };
template <> struct ElemOf<1> { static constexpr int BYTES = 2; };

template <int CH> __device__ __forceinline__ int swz(int row) {
    // physical 16-byte chunk = logical chunk ^ swz(row); makes a 16-lane ds_read_b128 group hit 16 distinct slots
    if (CH >= 16) return row & 15;
    if (CH == 8) return (row >> 1) & 7;
    if (CH == 4) return (row >> 2) & 3;
    return 0;
}

// ablation helpers: keep a value live without using it (device pass only: "v"/"s" are AMDGPU register constraints)
__device__ __forceinline__ void keep_alive(const f32x16& x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(x));
#else
    (void)x;
#endif
}
__device__ __forceinline__ void keep_alive_mask(unsigned long long x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(x));
#else
    (void)x;
#endif
}

// returns x, but the compiler cannot see through it: keeps rare-path index math inside its (wave-uniform) branch
// instead of being hoisted/if-converted into the common path
__device__ __forceinline__ int opaque_uniform(int x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}

__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

__device__ __forceinline__ int cd_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }  // 32x32 C/D row of reg r

// sorted-descending insertion; `s` is -inf for lanes that do not qualify (then nothing moves)
template <int KTOP>
__device__ __forceinline__ void topk_insert(float (&tv)[KTOP], int32_t (&ti)[KTOP], float s, int32_t id)
{
    bool ge_j = tv[KTOP - 1] >= s;
#pragma unroll
    for (int j = KTOP - 1; j >= 0; --j) {
        const bool ge_jm1 = (j == 0) ? true : (tv[j - 1] >= s);
        const float pv = (j == 0) ? 0.f : tv[j - 1];
        const int32_t pi = (j == 0) ? 0 : ti[j - 1];
        tv[j] = ge_j ? tv[j] : (ge_jm1 ? s : pv);
        ti[j] = ge_j ? ti[j] : (ge_jm1 ? id : pi);
        ge_j = ge_jm1;
    }
}

// ABL (ablation, tuning builds only): 0 = full kernel, 1 = no epilogue at all (MFMA + staging ceiling),
// 2 = common path only (block bound evaluated, rare path never taken)
template <int DT, int KT, int BN, int NCB, int EPI, bool GLDS, bool EUCLID, int WPS, int KTOP, bool BIAS, int ABL = 0>
__global__ __launch_bounds__(256, WPS) void score_gemm_kernel(ScoreParams p)
{
    constexpr int ES = ElemOf<DT>::BYTES;
    constexpr int RB = KT * ES;              // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row
    constexpr int KS = (DT == 1) ? KT / 16 : KT / 2;   // MFMA k-steps
    constexpr int TILE_BYTES = BN * RB;
    constexpr int NSLOT = BN * CH / 256;     // 16-byte staging slots per thread per tile
    constexpr bool PRIO = (ABL & 4) != 0;    // tuning: raise wave priority while issuing the MFMA block
    constexpr int RB_UNROLL = (EPI == 2 /* EPI_BLOCKMAX */) ? BN / 32 : 1;
    static_assert(BN * CH % 256 == 0 && NSLOT >= 1, "tile too small for 256 threads");
    static_assert(BN % 32 == 0, "BN must be a multiple of the 32-row MFMA block");

    constexpr int RW = 4 * NCB * 32;         // resident rows per workgroup
    // LDS: 2 * TILE_BYTES | STORE: r_bias[RW], r_sq[RW] | TOPK: per buffer t_bias[BN], t_sq[BN], t_bias block max[BN/32]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TSIDE = 2 * BN + 32;       // floats of per-tile side data per buffer (block maxima padded to 32)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    constexpr bool LANEUSER = (EPI != EPI_STORE);     // acc = mfma(items, users): lane & 31 is the user
    // bf16 dot/cosine with biases: the item bias is the initial accumulator (see the rb loop); euclidean needs the raw dot
    constexpr bool BIAS_IN_ACC = BIAS && DT == 1 && !EUCLID;
    const int rblock = blockIdx.x % p.n_rblocks;
    int chunk = blockIdx.x / p.n_rblocks;
    if (EPI == EPI_TOPK && p.rblock_chunk) {
        chunk = p.rblock_chunk[rblock];
        if (chunk < 0) return;                          // whole workgroup idle (padding of the grouped launch)
    }
    const int64_t r_base = ((int64_t)rblock * 4 + wave) * (NCB * 32);       // first resident row of this wave
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BN - 1) / BN);

    // ---- resident fragments: straight from global into registers, once ----
    bf16x8 rfb[(DT == 1) ? NCB : 1][(DT == 1) ? KS : 1];
    float rff[(DT == 0) ? NCB : 1][(DT == 0) ? KS : 1];
    // grouped TOPK through an index (stage 3 of the filtered top-k): resident row r is operand row row_index[r] -- the
    // user rows are gathered by these loads instead of by a separate copy pass; -1 = padding row (reads row 0, never lists)
    int64_t src_row[NCB];
    bool pad_row[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        int64_t row = r_base + cb * 32 + l31;
        if (row >= p.n_r) row = p.n_r - 1;                                   // clamped rows are never written
        pad_row[cb] = false;
        if (EPI == EPI_TOPK && p.row_index) {
            const int32_t s = p.row_index[row];
            pad_row[cb] = s < 0;
            row = s < 0 ? 0 : s;
        }
        src_row[cb] = row;
        const char* src = (const char*)p.R + row * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (DT == 1) rfb[cb][ks] = *(const bf16x8*)(src + (ks * 2 + half) * 16);
            else rff[cb][ks] = *(const float*)(src + (2 * ks + half) * 4);
        }
    }

    // ---- epilogue constants ----
    float r_bias_col[NCB], r_sq_col[NCB];          // TOPK: my user's bias / squared norm (lane & 31 is the user)
    float* r_bias_lds = (float*)(smem + 2 * TILE_BYTES);   // STORE: per resident row, read back as float4 per 4-row group
    float* r_sq_lds = r_bias_lds + RW;
    if (LANEUSER) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int64_t u = src_row[cb];
            r_bias_col[cb] = (BIAS && p.r_bias) ? p.r_bias[u] : 0.f;
            r_sq_col[cb] = EUCLID ? p.r_sqnorm[u] : 0.f;
        }
    } else {
        for (int i = tid; i < RW; i += 256) {
            int64_t u = (int64_t)rblock * RW + i;
            if (u >= p.n_r) u = p.n_r - 1;
            r_bias_lds[i] = (BIAS && p.r_bias) ? p.r_bias[u] : 0.f;
            r_sq_lds[i] = EUCLID ? p.r_sqnorm[u] : 0.f;
        }
    }

    float tv[(EPI == EPI_TOPK) ? NCB : 1][KTOP];
    int32_t ti[(EPI == EPI_TOPK) ? NCB : 1][KTOP];
    // One threshold per list: a score can only matter if it is  > my own KTOP-th best (strict: earlier equal values in
    // this lane have lower indices)  AND  >= the KTOP-th best of the OTHER half-wave's list for the same user (a
    // valid lower bound of the final KTOP-th best; non-strict because an equal value there may carry a higher index).
    // Both tests fold into  v > thr  with thr = max(own, float_pred(partner)).
    float thr[NCB], tau_pp[NCB];
    if (EPI == EPI_TOPK) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            thr[cb] = -INFINITY; tau_pp[cb] = -INFINITY;
            if (p.row_floor) {          // grouped re-scoring: scores below the floor cannot be in the final top-k
                tau_pp[cb] = pad_row[cb] ? INFINITY : float_pred(p.row_floor[src_row[cb]]);
                thr[cb] = tau_pp[cb];
            }
#pragma unroll
            for (int j = 0; j < KTOP; ++j) { tv[cb][j] = -INFINITY; ti[cb][j] = -1; }
        }
    }
    float* tside = (float*)(smem + 2 * TILE_BYTES);           // TOPK / BLOCKMAX side data, [2][TSIDE]
    float bm[NCB];                                            // BLOCKMAX: running maximum over the current superblock
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bm[cb] = -INFINITY;

    // ---- staging of the streamed tile: slot q = i*256 + tid  ->  (row, physical chunk) ----
    u32x4 stage[GLDS ? 1 : NSLOT];
    float side_b = 0.f, side_q = 0.f;
    // per-thread source offsets of the NSLOT staging slots inside a tile (bytes, relative to the tile's first row):
    // computed once; a tile advances every slot by BN rows.  Only the last tile of the last chunk needs clamping.
    int slot_row[NSLOT], slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CH, pc = q % CH;
        slot_row[i] = row;
        slot_off[i] = row * RB + ((pc ^ swz<CH>(row)) * 16);
    }
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BN;
        const bool clamp = row0 + BN > p.n_t;                 // wave-uniform
        if (LANEUSER && tid < BN) {
            int64_t g = row0 + tid;
            const bool ok = g < p.n_t;
            if (!ok) g = p.n_t - 1;
            side_b = (BIAS && p.t_bias) ? (ok ? p.t_bias[g] : -INFINITY) : 0.f;   // -inf: padded rows never raise the block max
            side_q = EUCLID ? p.t_sqnorm[g] : 0.f;
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BN * RB);     // scalar
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off[i];
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);                   // last valid row of this tile
                if (slot_row[i] > last) off -= (slot_row[i] - last) * RB;    // re-read the last valid row
            }
            const char* src = tile_base + off;
            if (GLDS) {
                char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;   // wave-uniform; lane*16 is implicit
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                stage[i] = *(const u32x4*)src;
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if (LANEUSER && tid < BN) {
            float* sd = tside + buf * TSIDE;
            sd[tid] = side_b;
            sd[BN + tid] = side_q;
            float m = side_b;                                   // max over each 32-item block (lanes 0-31 / 32-63)
            for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if ((tid & 31) == 0) sd[2 * BN + (tid >> 5)] = m;
        }
        if (!GLDS) {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) *(u32x4*)(smem + buf * TILE_BYTES + (i * 256 + tid) * 16) = stage[i];
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    stage_issue(0, 0);
    stage_commit(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        const char* tile = smem + buf * TILE_BYTES;
        const int64_t tile_row0 = t_begin + (int64_t)t * BN;
        const bool partial = tile_row0 + BN > p.n_t;

        // BLOCKMAX has a branch-free epilogue: unrolling lets the scheduler overlap block rb's max chain with block rb+1's
        // LDS reads and MFMAs.  The list epilogues keep the loop rolled (their rare path is large).
#pragma unroll(RB_UNROLL)
        for (int rb = 0; rb < BN / 32; ++rb) {
            // bf16 + biases: the item bias IS the initial accumulator (C operand of the first MFMA) -- zero VALU cost.
            // bf16 scores are therefore defined as fl-chain(b_i + sum_k u_k i_k) + b_u in every bf16 epilogue (there is
            // no bit-exactness claim for bf16); padded rows carry b_i = -inf and mask themselves.  fp32 keeps the
            // reference's order (s + b_u) + b_i with explicit adds.
            f32x16 acc[NCB];
            if (BIAS_IN_ACC && LANEUSER) {
                const float* sdi = tside + buf * TSIDE + rb * 32 + 4 * half;
                f32x16 c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb4 = *(const f32x4*)(sdi + 8 * q);
                    c0[4 * q] = tb4[0]; c0[4 * q + 1] = tb4[1]; c0[4 * q + 2] = tb4[2]; c0[4 * q + 3] = tb4[3];
                }
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = c0;
            } else if (BIAS_IN_ACC) {
                int64_t it = t_begin + (int64_t)t * BN + rb * 32 + l31;          // STORE: my column's item
                if (it >= p.n_t) it = p.n_t - 1;
                const float tbv0 = p.t_bias ? p.t_bias[it] : 0.f;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][r] = tbv0;
            } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            }

            const int lrow = rb * 32 + l31;
            const char* rowp = tile + lrow * RB;
            const int sw = swz<CH>(lrow);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (DT == 1) {
                    const bf16x8 tf = *(const bf16x8*)(rowp + (((ks * 2 + half) ^ sw) * 16));
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (LANEUSER) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, rfb[cb][ks], acc[cb], 0, 0, 0);
                        else acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfb[cb][ks], tf, acc[cb], 0, 0, 0);
                    }
                } else {
                    const int k = 2 * ks + half;
                    const float tf = *(const float*)(rowp + (((k >> 2) ^ sw) * 16) + (k & 3) * 4);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (LANEUSER) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(tf, rff[cb][ks], acc[cb], 0, 0, 0);
                        else acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(rff[cb][ks], tf, acc[cb], 0, 0, 0);
                    }
                }
            }

            if (PRIO) __builtin_amdgcn_s_setprio(0);
            const int64_t blk_row0 = tile_row0 + rb * 32;     // first streamed row (item) of this 32-block
            if (EPI == EPI_BLOCKMAX) {
                // stage 1 of the two-stage top-k: only the maximum EXACT score of this 32-row block per user, folded
                // into the superblock maximum.  No lists, no data-dependent branch: every wave does the same work.
                // Bias order: fp32 follows the reference, (s + b_u) + b_i per element; bf16 (no bit-exactness claim
                // against fp32 anyway) uses (s + b_i) + b_u in ALL its epilogues, which lets b_u be added once after
                // the max (fp32 addition is monotone, so max_r fl(x_r + b_u) == fl(max_r x_r + b_u)).
                const float* sd = tside + buf * TSIDE;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    float m;
                    if (partial || EUCLID) {                        // wave-uniform: last tile of the last chunk, or euclid
                        const int rows_left = opaque_uniform((int)(p.n_t - t_begin) - (t * BN + rb * 32));
                        m = -INFINITY;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 tb4 = {0.f, 0.f, 0.f, 0.f}, tq4 = {0.f, 0.f, 0.f, 0.f};
                            if (BIAS) tb4 = *(const f32x4*)(sd + rb * 32 + 8 * q + 4 * half);
                            if (EUCLID) tq4 = *(const f32x4*)(sd + BN + rb * 32 + 8 * q + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[cb][4 * q + e];
                                if (EUCLID) {
                                    float dist = (r_sq_col[cb] - 2.0f * v) + tq4[e];
                                    dist = fmaxf(dist, 1e-16f);
                                    v = -1.0f * sqrtf(dist);
                                }
                                if (BIAS) v = BIAS_IN_ACC ? v + r_bias_col[cb] : (v + r_bias_col[cb]) + tb4[e];
                                if (8 * q + 4 * half + e >= rows_left) v = -INFINITY;
                                m = fmaxf(m, v);
                            }
                        }
                    } else if (BIAS && !BIAS_IN_ACC) {
                        f32x16 s = acc[cb];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 tb4 = *(const f32x4*)(sd + rb * 32 + 8 * q + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                s[4 * q + e] = (s[4 * q + e] + r_bias_col[cb]) + tb4[e];
                        }
                        m = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
                        for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, s[r]), s[r + 1]);
                        m = fmaxf(m, s[15]);
                    } else {
                        m = fmaxf(fmaxf(acc[cb][0], acc[cb][1]), acc[cb][2]);
#pragma unroll
                        for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[cb][r]), acc[cb][r + 1]);
                        m = fmaxf(m, acc[cb][15]);
                        if (BIAS) m = m + r_bias_col[cb];          // bf16: b_i already in the accumulator, b_u after the max
                    }
                    bm[cb] = fmaxf(bm[cb], m);
                }
            } else if (EPI == EPI_TOPK && (ABL & 3) == 1) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) keep_alive(acc[cb]);
            } else if (EPI == EPI_TOPK) {
                // rows of acc = items blk_row0 + cd_row(r, half); col = my user.  Two-level filter:
                //  (1) block bound: (max_r acc + user_bias) + max item bias of the block.  fp32 rounding is monotone,
                //      so no score of the block can exceed it; if it cannot enter the list the block costs ~13 VALU.
                //  (2) exact scores in the reference's order (acc + ub) + ib only for blocks that pass (1).
                // All index math is 32-bit and relative to the chunk; the partial-tile mask exists only inside (2).
                const float* sd = tside + buf * TSIDE;
                const int blk_in_chunk = t * BN + rb * 32;                   // first item of the block, chunk-relative
                const int rows_left = (int)(p.n_t - t_begin) - blk_in_chunk; // valid rows from the block start (>= 1)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    // ---- common path: one v_max3 chain per 4-row group, one compare ----
                    float gm[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gm[q] = fmaxf(fmaxf(fmaxf(acc[cb][4 * q], acc[cb][4 * q + 1]), acc[cb][4 * q + 2]), acc[cb][4 * q + 3]);
                    unsigned long long need = ~0ull;
                    if (!EUCLID) {
                        float bound = fmaxf(fmaxf(fmaxf(gm[0], gm[1]), gm[2]), gm[3]);
                        if (BIAS) bound = BIAS_IN_ACC ? bound + r_bias_col[cb]
                                                      : (bound + r_bias_col[cb]) + sd[2 * BN + rb];
                        need = __builtin_amdgcn_ballot_w64(bound > thr[cb]);
                    }
                    if ((ABL & 3) == 2) { keep_alive_mask(need); need = 0ull; }
                    if (need != 0ull) {
                        // ---- rare path: exact scores, then only the 4-row groups (and elements) that can matter ----
                        f32x16 s = acc[cb];
                        if (BIAS || EUCLID) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x4 tb4, tq4;
                                if (BIAS) tb4 = *(const f32x4*)(sd + rb * 32 + 8 * q + 4 * half);
                                if (EUCLID) tq4 = *(const f32x4*)(sd + BN + rb * 32 + 8 * q + 4 * half);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float v = s[4 * q + e];
                                    if (EUCLID) {
                                        float dist = (r_sq_col[cb] - 2.0f * v) + tq4[e];
                                        dist = fmaxf(dist, 1e-16f);
                                        v = -1.0f * sqrtf(dist);
                                    }
                                    if (BIAS) v = BIAS_IN_ACC ? v + r_bias_col[cb] : (v + r_bias_col[cb]) + tb4[e];
                                    s[4 * q + e] = v;
                                }
                            }
                        }
                        if (partial) {      // clamped rows duplicate the last item: mask them (last tile of the last chunk only)
                            const int rl = opaque_uniform(rows_left);
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (cd_row(r, half) >= rl) s[r] = -INFINITY;
                        }
                        const int32_t id0 = (int32_t)t_begin + blk_in_chunk + p.t_index_base + 4 * half;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float g = fmaxf(fmaxf(fmaxf(s[4 * q], s[4 * q + 1]), s[4 * q + 2]), s[4 * q + 3]);
                            if (__builtin_amdgcn_ballot_w64(g > thr[cb]) != 0ull) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float v = s[4 * q + e];
                                    const bool hit = v > thr[cb];
                                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                                        topk_insert<KTOP>(tv[cb], ti[cb], hit ? v : -INFINITY, id0 + 8 * q + e);
                                        thr[cb] = fmaxf(tv[cb][KTOP - 1], tau_pp[cb]);
                                    }
                                }
                            }
                        }
                        // (independent lists -- the bf16 FILTER of the exact top-k, topk_filter.hip -- keep every score
                        // >= the row's floor that fits: what a list drops is then below ITS OWN last entry only)
                        if (!p.independent_lists)
                            tau_pp[cb] = fmaxf(tau_pp[cb], float_pred(__shfl_xor(tv[cb][KTOP - 1], 32, 64)));
                        thr[cb] = fmaxf(tv[cb][KTOP - 1], tau_pp[cb]);
                    }
                }
            } else {
                const int64_t item = blk_row0 + l31;
                const bool item_ok = item < p.n_t;
                const int64_t ic = item_ok ? item : p.n_t - 1;
                const float tbv = (BIAS && p.t_bias) ? p.t_bias[ic] : 0.f;
                const float tqv = EUCLID ? p.t_sqnorm[ic] : 0.f;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int lrow4 = (wave * NCB + cb) * 32 + 8 * q + 4 * half;     // 4 consecutive resident rows
                        const f32x4 rb4 = *(const f32x4*)(r_bias_lds + lrow4);
                        const f32x4 rq4 = *(const f32x4*)(r_sq_lds + lrow4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = q * 4 + e;
                            const int64_t u = r_base + cb * 32 + cd_row(r, half);
                            float v = acc[cb][r];
                            if (EUCLID) {
                                float dist = (rq4[e] - 2.0f * v) + tqv;
                                dist = fmaxf(dist, 1e-16f);
                                v = -1.0f * sqrtf(dist);
                            }
                            if (BIAS) v = BIAS_IN_ACC ? v + rb4[e] : (v + rb4[e]) + tbv;
                            if (item_ok && u < p.n_r) p.out[u * p.ld_out + item] = v;
                        }
                    }
            }
        }

        if (EPI == EPI_BLOCKMAX && (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles)) {
            // end of a superblock: combine the two half-wave maxima of each user, 32 consecutive floats per store
            const int64_t sb = t_begin / ((int64_t)p.sb_tiles * BN) + t / p.sb_tiles;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float v = fmaxf(bm[cb], __shfl_xor(bm[cb], 32, 64));
                const int64_t u = r_base + cb * 32 + l31;
                if (half == 0 && u < p.n_r) p.blockmax[sb * p.bm_stride + u] = v;
                bm[cb] = -INFINITY;
            }
        }
        if (t + 1 < n_tiles) stage_commit(buf ^ 1);
        __syncthreads();
    }

    if (EPI == EPI_TOPK) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int64_t u = r_base + cb * 32 + l31;
            int64_t list = -1;
            if (u < p.n_r) list = p.row_pair ? (int64_t)p.row_pair[u] * 2 + half : u * p.n_parts + (chunk * 2 + half);
            if (list >= 0) {
                const int64_t o = list * p.capacity;
                if (KTOP % 4 == 0 && p.capacity == KTOP) {
#pragma unroll
                    for (int j = 0; j + 3 < KTOP; j += 4) {
                        if (p.part_vals)      // (the filtered top-k re-scores the listed ids exactly: bf16 values unused)
                            *(f32x4*)(p.part_vals + o + j) = (f32x4){tv[cb][j], tv[cb][j + 1], tv[cb][j + 2], tv[cb][j + 3]};
                        *(int4*)(p.part_idx + o + j) = make_int4(ti[cb][j], ti[cb][j + 1], ti[cb][j + 2], ti[cb][j + 3]);
                    }
                } else if (!p.part_vals) {
#pragma unroll
                    for (int j = 0; j < KTOP; ++j) p.part_idx[o + j] = ti[cb][j];
                    for (int j = KTOP; j < p.capacity; ++j) p.part_idx[o + j] = -1;
                } else {
#pragma unroll
                    for (int j = 0; j < KTOP; ++j) { p.part_vals[o + j] = tv[cb][j]; p.part_idx[o + j] = ti[cb][j]; }
                    for (int j = KTOP; j < p.capacity; ++j) { p.part_vals[o + j] = -INFINITY; p.part_idx[o + j] = -1; }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// topk_merge: one wave per user; candidates [n_cand] = n_parts * kcap, CPL per lane in registers; k rounds of
// wave-wide arg-best on (value desc, index asc).  A candidate is packed into ONE 64-bit key -- high word: the float
// mapped monotonically onto unsigned, low word: ~index -- so "best" is an unsigned 64-bit maximum, taken across the
// wave with DPP row rotations + row broadcasts (VALU-only; the ds_bpermute butterflies this replaces cost ~1200 cycles
// per round, 10 rounds per user).
template <int CPL>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi,
                                                        int64_t n_users, int n_cand, int k, float* __restrict__ ov,
                                                        int32_t* __restrict__ oi)
{
    const int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= n_users) return;
    const int lane = lane_id();
    const unsigned long long EMPTY = merge_key(-INFINITY, 0x7fffffff);
    unsigned long long key[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = c * 64 + lane;
        key[c] = EMPTY;
        if (j < n_cand) {
            const int32_t raw = pi[u * n_cand + j];
            if (raw >= 0) key[c] = merge_key(pv[u * n_cand + j], raw);
        }
    }
    for (int t = 0; t < k; ++t) {
        unsigned long long best = key[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) best = key[c] > best ? key[c] : best;
        best = wave_max_u64(best);
        // every lane now holds the winner; retire it where it lives (indices are unique among real candidates)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (key[c] == best && best != EMPTY) key[c] = EMPTY;
        if (lane == 0) {
            const unsigned int hi = (unsigned int)(best >> 32);
            const unsigned int bits = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
            ov[u * k + t] = (best == EMPTY) ? -INFINITY : __uint_as_float(bits);
            oi[u * k + t] = (best == EMPTY) ? -1 : (int32_t)(~(unsigned int)best);
        }
    }
}

// ------------------------------------------------------------------------------------------
// score_prep: fp32 representation [n, d] -> MFMA operand [n, KT] (fp32 or bf16, zero padded to KT),
// optionally row-normalised (cosine: prediction_graphs.py:68-69 / recommendation_graphs.py:119-120) and/or
// with the row squared norm emitted (euclidean: prediction_graphs.py:87-90).  One wave per row.
__global__ __launch_bounds__(256) void score_prep_kernel(const float* __restrict__ x, int64_t n, int d, int kt,
                                                        int normalize, int dtype, void* __restrict__ out,
                                                        float* __restrict__ sqnorm)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n) return;
    const int lane = lane_id();
    const float* xr = x + row * (int64_t)d;
    float scale = 1.0f;
    if (normalize || sqnorm) {
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) ss = fmaf(xr[c], xr[c], ss);
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (normalize) scale = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        if (sqnorm && lane == 0) sqnorm[row] = ss;    // squared norm of the UN-normalised row
    }
    for (int c0 = lane * 2; c0 < kt; c0 += 128) {
        float a = (c0 < d) ? xr[c0] : 0.f, b = (c0 + 1 < d) ? xr[c0 + 1] : 0.f;
        if (normalize) { a *= scale; b *= scale; }
        if (dtype == 0) {
            float* o = (float*)out + row * (int64_t)kt + c0;
            o[0] = a; o[1] = b;
        } else {
            unsigned int packed = (unsigned int)f32_to_bf16_rne(a) | ((unsigned int)f32_to_bf16_rne(b) << 16);
            *((unsigned int*)((unsigned short*)out + row * (int64_t)kt + c0)) = packed;
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int DT, int KT, int BN, int NCB, int EPI, bool GLDS, bool EUCLID, int KTOP, bool BIAS, int WPS_OVERRIDE = 0, int ABL = 0>
static int launch_score_b(const ScoreParams& p, hipStream_t st)
{
    constexpr int LDS = 2 * BN * KT * ElemOf<DT>::BYTES +
                        (EPI == EPI_STORE ? 2 * 4 * NCB * 32 * 4 : 2 * (2 * BN + 32) * 4);     // TOPK and BLOCKMAX: side data
    // 2 workgroups per CU (256 registers per lane) unless the resident fragments + top-k lists need more
    constexpr int WPS = WPS_OVERRIDE ? WPS_OVERRIDE
                        : (EPI == EPI_BLOCKMAX ? ((KT == 256 || (DT == 0 && KT >= 128)) ? 2 : 3)   // no lists: small
                                               : (((DT == 0 && KT >= 128) || KT == 256 || (EUCLID && KT >= 128)) ? 1 : 2));
    auto kern = score_gemm_kernel<DT, KT, BN, NCB, EPI, GLDS, EUCLID, WPS, KTOP, BIAS, ABL>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm");
}

// a missing bias pointer in a biased launch is treated as zeros (x + 0.0f == x), so one flag covers both sides
template <int DT, int KT, int BN, int NCB, int EPI, bool GLDS, bool EUCLID, int KTOP, int WPS_OVERRIDE = 0>
static int launch_score(const ScoreParams& p, hipStream_t st)
{
    if (p.r_bias || p.t_bias) return launch_score_b<DT, KT, BN, NCB, EPI, GLDS, EUCLID, KTOP, true, WPS_OVERRIDE>(p, st);
    return launch_score_b<DT, KT, BN, NCB, EPI, GLDS, EUCLID, KTOP, false, WPS_OVERRIDE>(p, st);
}

struct ScoreCfg { int bn, ncb; };
static ScoreCfg score_cfg(int dtype, int kt)
{
    if (dtype == 1) return ScoreCfg{64, kt <= 128 ? 2 : 1};
    return ScoreCfg{32, 1};
}

extern "C" int trec_score_kpad(int32_t d)
{
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    if (d <= 256) return 256;
    return -1;
}

// resident rows handled by one workgroup for (dtype, kpad): 4 waves * NCB * 32
extern "C" int trec_score_rows_per_workgroup(int32_t dtype, int32_t kpad) { return 4 * score_cfg(dtype, kpad).ncb * 32; }
extern "C" int trec_score_tile_rows(int32_t dtype, int32_t kpad) { return score_cfg(dtype, kpad).bn; }
// capacity of the per-lane lists for a requested k: 8, 12 or 16 (k > 16 is not supported by the fused epilogue)
extern "C" int trec_score_topk_capacity(int32_t k) { return k <= 8 ? 8 : (k <= 12 ? 12 : (k <= 16 ? 16 : -1)); }

template <int EPI, int KTOP>
static int dispatch_score(int dtype, int kt, int variant, const ScoreParams& p, hipStream_t st)
{
    const bool glds = (variant & 1) != 0;
#define TREC_SCORE_CASE(DT, KTV, BNV, NCBV)                                                  \
    if (dtype == DT && kt == KTV) {                                                          \
        if (p.euclid) return launch_score<DT, KTV, BNV, NCBV, EPI, false, true, KTOP>(p, st);  \
        if (glds && DT == 1) return launch_score<DT, KTV, BNV, NCBV, EPI, true, false, KTOP>(p, st);  \
        return launch_score<DT, KTV, BNV, NCBV, EPI, false, false, KTOP>(p, st);             \
    }
    TREC_SCORE_CASE(1, 128, 64, 2)
    TREC_SCORE_CASE(0, 128, 32, 1)
#ifndef TREC_SCORE_MINIMAL      // -DTREC_SCORE_MINIMAL: only K = 128 (fast rebuilds while tuning)
    TREC_SCORE_CASE(1, 32, 64, 2)
    TREC_SCORE_CASE(1, 64, 64, 2)
    TREC_SCORE_CASE(1, 256, 64, 1)
    TREC_SCORE_CASE(0, 32, 32, 1)
    TREC_SCORE_CASE(0, 64, 32, 1)
    TREC_SCORE_CASE(0, 256, 32, 1)
#endif
#undef TREC_SCORE_CASE
    trec_set_last_error("trec_score_gemm: unsupported (dtype, kpad)");
    return TREC_ERR_UNSUPPORTED;
}

// experimental tilings of the hot configuration (bf16, K = 128, dot, capacity 12), selected by variant >> 1
static ScoreCfg score_cfg_variant(int dtype, int kt, int variant)
{
    if (dtype == 1 && kt == 128) {
        switch (variant >> 1) {
            case 1: return ScoreCfg{128, 2};     // default tiling with 128-row item tiles (half the barriers)
            case 2: return ScoreCfg{64, 1};      // 32 users per wave, 4 workgroups / CU
            case 3: return ScoreCfg{64, 1};      // 3 workgroups / CU, 32 users per wave
            default: break;
        }
    }
    return score_cfg(dtype, kt);
}

static int fill_common(ScoreParams& p, const void* users, const void* items, int dtype, int kpad, int64_t n_users,
                       int64_t n_items, const float* user_bias, const float* item_bias, int mode,
                       const float* user_sqnorm, const float* item_sqnorm, int n_chunks, int variant = 0)
{
    TREC_REQUIRE(users && items, "trec_score_gemm: null operand");
    TREC_REQUIRE(dtype == 0 || dtype == 1, "trec_score_gemm: dtype must be 0 (fp32) or 1 (bf16)");
    TREC_REQUIRE(kpad == 32 || kpad == 64 || kpad == 128 || kpad == 256, "trec_score_gemm: kpad must be 32/64/128/256");
    TREC_REQUIRE(n_users >= 1 && n_items >= 1, "trec_score_gemm: empty operand");
    TREC_REQUIRE(mode == 0 || mode == 1, "trec_score_gemm: mode must be 0 (dot) or 1 (euclidean)");
    TREC_REQUIRE(mode == 0 || (user_sqnorm && item_sqnorm), "trec_score_gemm: euclidean mode needs squared norms");
    TREC_REQUIRE(n_chunks >= 1, "trec_score_gemm: n_chunks must be >= 1");
    TREC_REQUIRE(n_items < (int64_t)1 << 31 && n_users < (int64_t)1 << 31, "trec_score_gemm: sizes must fit int32");
    const ScoreCfg c = score_cfg_variant(dtype, kpad, variant);
    p.R = users; p.T = items; p.n_r = n_users; p.n_t = n_items;
    p.n_rblocks = (int)ceil_div64(n_users, 4 * c.ncb * 32);
    int64_t cl = ceil_div64(ceil_div64(n_items, n_chunks), 128) * 128;     // multiple of every BN in use
    p.chunk_len = cl;
    p.n_chunks = (int)ceil_div64(n_items, cl);
    p.r_bias = user_bias; p.t_bias = item_bias; p.r_sqnorm = user_sqnorm; p.t_sqnorm = item_sqnorm;
    p.euclid = mode;
    return TREC_OK;
}

extern "C" int trec_score_gemm_store(const void* users, const void* items, int32_t dtype, int32_t kpad,
                                     int64_t n_users, int64_t n_items, const float* user_bias,
                                     const float* item_bias, int32_t mode, const float* user_sqnorm,
                                     const float* item_sqnorm, float* out, int64_t ld_out, int32_t variant,
                                     void* stream)
{
    ScoreParams p = {};
    TREC_REQUIRE(out && ld_out >= n_items, "trec_score_gemm_store: bad output");
    // enough chunks to put >= 2 workgroups on every CU when the user count alone cannot
    const int rows_wg = trec_score_rows_per_workgroup(dtype, kpad);
    int n_chunks = 1;
    if (rows_wg > 0) {
        const int64_t rblocks = ceil_div64(n_users, rows_wg);
        while (rblocks * n_chunks < 1024 && n_chunks < 64 && ceil_div64(n_items, n_chunks * 2) >= 256) n_chunks *= 2;
    }
    int rc = fill_common(p, users, items, dtype, kpad, n_users, n_items, user_bias, item_bias, mode, user_sqnorm,
                         item_sqnorm, n_chunks);
    if (rc) return rc;
    p.out = out; p.ld_out = ld_out;
    return dispatch_score<EPI_STORE, 8>(dtype, kpad, variant, p, (hipStream_t)stream);
}

// number of partial lists per user that trec_score_gemm_topk writes for a requested chunk count
extern "C" int trec_score_topk_parts(int32_t dtype, int32_t kpad, int64_t n_items, int32_t n_chunks)
{
    const ScoreCfg c = ScoreCfg{128, 0};     // chunk lengths are rounded to 128 rows: valid for every tiling (BN | 128)
    if (n_chunks < 1 || n_items < 1) return -1;
    const int64_t cl = ceil_div64(ceil_div64(n_items, n_chunks), c.bn) * c.bn;
    return 2 * (int)ceil_div64(n_items, cl);
}

extern "C" int trec_score_gemm_topk(const void* users, const void* items, int32_t dtype, int32_t kpad,
                                    int64_t n_users, int64_t n_items, int32_t item_index_base,
                                    const float* user_bias, const float* item_bias, int32_t mode,
                                    const float* user_sqnorm, const float* item_sqnorm, int32_t n_chunks,
                                    int32_t capacity, float* part_vals, int32_t* part_idx, int32_t variant,
                                    void* stream)
{
    ScoreParams p = {};
    TREC_REQUIRE(part_vals && part_idx, "trec_score_gemm_topk: null workspace");
    TREC_REQUIRE(capacity == 8 || capacity == 12 || capacity == 16, "trec_score_gemm_topk: capacity must be 8, 12 or 16");
    const bool experimental = dtype == 1 && kpad == 128 && capacity == 12 && mode == 0 && (variant >> 1) != 0 &&
                              (variant >> 1) <= 6;
    int rc = fill_common(p, users, items, dtype, kpad, n_users, n_items, user_bias, item_bias, mode, user_sqnorm,
                         item_sqnorm, n_chunks, experimental ? variant : 0);
    if (rc) return rc;
    p.part_vals = part_vals; p.part_idx = part_idx; p.n_parts = 2 * p.n_chunks; p.t_index_base = item_index_base;
    p.capacity = capacity;
    if (experimental) {
        hipStream_t st = (hipStream_t)stream;
        const bool glds = variant & 1;
        switch (variant >> 1) {
            case 1: return glds ? launch_score<1, 128, 128, 2, EPI_TOPK, true, false, 12, 2>(p, st)
                                : launch_score<1, 128, 128, 2, EPI_TOPK, false, false, 12, 2>(p, st);
            case 2: return launch_score<1, 128, 64, 1, EPI_TOPK, true, false, 12, 4>(p, st);   // 32 users / wave, 4 WG / CU
            case 3: return glds ? launch_score<1, 128, 64, 1, EPI_TOPK, true, false, 12, 3>(p, st)
                                : launch_score<1, 128, 64, 1, EPI_TOPK, false, false, 12, 3>(p, st);
            case 4: return launch_score_b<1, 128, 64, 2, EPI_TOPK, true, false, 12, true, 2, 1>(p, st);   // ablation
            case 5: return launch_score_b<1, 128, 64, 2, EPI_TOPK, true, false, 12, true, 2, 2>(p, st);   // ablation
            default: return launch_score_b<1, 128, 64, 2, EPI_TOPK, true, false, 12, true, 2, 4>(p, st);  // full + setprio
        }
    }
    if (capacity == 8) return dispatch_score<EPI_TOPK, 8>(dtype, kpad, variant, p, (hipStream_t)stream);
    if (capacity == 12) return dispatch_score<EPI_TOPK, 12>(dtype, kpad, variant, p, (hipStream_t)stream);
    return dispatch_score<EPI_TOPK, 16>(dtype, kpad, variant, p, (hipStream_t)stream);
}

// ---- two-stage exact top-k: stage 1 (per-superblock maxima) and stage 3b (grouped re-scoring) ----------------------
// superblock rows must be a multiple of 128 (every tile height in use divides it)
extern "C" int trec_score_gemm_blockmax(const void* users, const void* items, int32_t dtype, int32_t kpad,
                                        int64_t n_users, int64_t n_items, const float* user_bias,
                                        const float* item_bias, int32_t mode, const float* user_sqnorm,
                                        const float* item_sqnorm, int32_t sb_rows, int32_t n_chunks, float* blockmax,
                                        int64_t bm_stride, int32_t variant, void* stream)
{
    ScoreParams p = {};
    TREC_REQUIRE(blockmax && bm_stride >= n_users, "trec_score_gemm_blockmax: bad output");
    TREC_REQUIRE(sb_rows >= 128 && sb_rows % 128 == 0, "trec_score_gemm_blockmax: sb_rows must be a multiple of 128");
    int rc = fill_common(p, users, items, dtype, kpad, n_users, n_items, user_bias, item_bias, mode, user_sqnorm,
                         item_sqnorm, n_chunks);
    if (rc) return rc;
    const ScoreCfg c = score_cfg(dtype, kpad);
    p.chunk_len = ceil_div64(ceil_div64(n_items, n_chunks), sb_rows) * sb_rows;        // chunks are whole superblocks
    p.n_chunks = (int)ceil_div64(n_items, p.chunk_len);
    p.blockmax = blockmax; p.bm_stride = bm_stride; p.sb_tiles = sb_rows / c.bn;
    // the hot configuration (bf16 dot / cosine, K = 64 / 128) has a hand-scheduled kernel; "blockmax_pipelined" = 0
    // selects the generic one (A/B runs and tests)
    // variant bit 5: the caller is a FILTER (K2f): the maxima need not equal any other kernel's scores bit for bit, only obey
    // the bound -> the 16x16x32 form (17% more work per joule at the power cap)
    if (dtype == 1 && !mode && (kpad == 64 || kpad == 128) && c.bn == 64 && (variant & 32)) {
        rc = launch_blockmax_filter16(p, kpad, (hipStream_t)stream);
        if (rc != TREC_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == 1 && !mode && (kpad == 64 || kpad == 128) && c.bn == 64 && trec_get_tuning("blockmax_pipelined", 1)) {
        rc = launch_blockmax_pipelined(p, kpad, (hipStream_t)stream);
        if (rc != TREC_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == 0 && !mode && (kpad == 64 || kpad == 128) && trec_get_tuning("blockmax_pipelined_f32", 1)) {
        rc = launch_blockmax_pipelined_f32(p, kpad, sb_rows, (hipStream_t)stream);
        if (rc != TREC_ERR_UNSUPPORTED) return rc;
    }
    return dispatch_score<EPI_BLOCKMAX, 8>(dtype, kpad, variant, p, (hipStream_t)stream);
}

// users_g: operand rows gathered by trec_topk_fill_groups ([n_rows_g, kpad], n_rows_g a multiple of the rows per
// workgroup) -- or, with row_index (int32 [n_rows_g], -1 = padding row), the UNgathered operand: grouped row r is
// users_g[row_index[r]] and user_bias_g / user_sqnorm_g / row_floor are indexed the same way; workgroup w re-scores superblock rblock_chunk[w] (rows [s*sb_rows, (s+1)*sb_rows) of `items`) for its
// rows and writes the two half-wave lists of row r to list ids 2*row_pair[r], 2*row_pair[r]+1 of part_vals / part_idx.
extern "C" int trec_score_gemm_topk_grouped(const void* users_g, const void* items, int32_t dtype, int32_t kpad,
                                            int64_t n_rows_g, int64_t n_items, int32_t item_index_base,
                                            const float* user_bias_g, const float* item_bias, int32_t mode,
                                            const float* user_sqnorm_g, const float* item_sqnorm, int32_t sb_rows,
                                            const int32_t* rblock_chunk, const int32_t* row_pair,
                                            const float* row_floor, int32_t capacity, float* part_vals,
                                            int32_t* part_idx, int32_t variant, const int32_t* row_index, void* stream)
{
    ScoreParams p = {};
    TREC_REQUIRE(part_idx && rblock_chunk && row_pair, "trec_score_gemm_topk_grouped: null pointer");
    TREC_REQUIRE(capacity == 8 || capacity == 12 || capacity == 16, "trec_score_gemm_topk_grouped: capacity must be 8, 12 or 16");
    TREC_REQUIRE(sb_rows >= 128 && sb_rows % 128 == 0, "trec_score_gemm_topk_grouped: sb_rows must be a multiple of 128");
    if (n_rows_g == 0) return TREC_OK;
    int rc = fill_common(p, users_g, items, dtype, kpad, n_rows_g, n_items, user_bias_g, item_bias, mode,
                         user_sqnorm_g, item_sqnorm, 1);
    if (rc) return rc;
    p.chunk_len = sb_rows; p.n_chunks = 1;
    p.rblock_chunk = rblock_chunk; p.row_pair = row_pair; p.row_floor = row_floor;
    p.part_vals = part_vals; p.part_idx = part_idx; p.n_parts = 2; p.t_index_base = item_index_base;
    p.capacity = capacity;
    p.independent_lists = (variant >> 4) & 1;
    p.row_index = row_index;
    if (capacity == 8) return dispatch_score<EPI_TOPK, 8>(dtype, kpad, variant & 1, p, (hipStream_t)stream);
    if (capacity == 12) return dispatch_score<EPI_TOPK, 12>(dtype, kpad, variant & 1, p, (hipStream_t)stream);
    return dispatch_score<EPI_TOPK, 16>(dtype, kpad, variant & 1, p, (hipStream_t)stream);
}

extern "C" int trec_topk_merge(const float* part_vals, const int32_t* part_idx, int64_t n_users, int32_t n_cand,
                               int32_t k, float* out_vals, int32_t* out_idx, void* stream)
{
    TREC_REQUIRE(part_vals && part_idx && out_vals && out_idx, "trec_topk_merge: null pointer");
    TREC_REQUIRE(k >= 1 && n_cand >= 1 && n_cand <= 1024, "trec_topk_merge: need 1 <= n_cand <= 1024, k >= 1");
    if (n_users == 0) return TREC_OK;
    const unsigned blocks = (unsigned)ceil_div64(n_users * 64, 256);
    hipStream_t st = (hipStream_t)stream;
    const int cpl = (n_cand + 63) / 64;
#define TREC_MERGE(CPLV) hipLaunchKernelGGL((topk_merge_kernel<CPLV>), dim3(blocks), dim3(256), 0, st, part_vals, part_idx, n_users, n_cand, k, out_vals, out_idx)
    if (cpl <= 1) TREC_MERGE(1);
    else if (cpl <= 2) TREC_MERGE(2);
    else if (cpl <= 4) TREC_MERGE(4);
    else if (cpl <= 8) TREC_MERGE(8);
    else TREC_MERGE(16);
#undef TREC_MERGE
    return trec_check_launch("trec_topk_merge");
}

// vectorised form for d % 4 == 0 (rows 16-byte aligned): G = min(32, KT / 4) lanes own one row, a lane converts float4
// chunks (16 B loads, 8 B bf16 / 16 B fp32 stores) -- 768 B of traffic per 128-wide row at HBM speed instead of the
// 8-byte accesses of the generic kernel above.
template <int G>
__global__ __launch_bounds__(256) void score_prep_vec4_kernel(const float* __restrict__ x, int64_t n, int d, int kt,
                                                             int normalize, int dtype, void* __restrict__ out,
                                                             float* __restrict__ sqnorm)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (row >= n) return;
    const int sub = threadIdx.x % G;
    const float* xr = x + row * (int64_t)d;
    f32x4 v[2];                                   // kt <= 256 = 2 * 32 lanes * 4
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < d) v[it] = *(const f32x4*)(xr + c);
        ss = fmaf(v[it][0], v[it][0], ss); ss = fmaf(v[it][1], v[it][1], ss);
        ss = fmaf(v[it][2], v[it][2], ss); ss = fmaf(v[it][3], v[it][3], ss);
    }
    float scale = 1.0f;
    if (normalize || sqnorm) {
        for (int off = G / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (normalize) scale = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        if (sqnorm && sub == 0) sqnorm[row] = ss;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        if (c >= kt) continue;
        f32x4 w = v[it];
        if (normalize) { w[0] *= scale; w[1] *= scale; w[2] *= scale; w[3] *= scale; }
        if (dtype == 0) {
            *(f32x4*)((float*)out + row * (int64_t)kt + c) = w;
        } else {
            uint2 pk;
            pk.x = (unsigned int)f32_to_bf16_rne(w[0]) | ((unsigned int)f32_to_bf16_rne(w[1]) << 16);
            pk.y = (unsigned int)f32_to_bf16_rne(w[2]) | ((unsigned int)f32_to_bf16_rne(w[3]) << 16);
            *(uint2*)((unsigned short*)out + row * (int64_t)kt + c) = pk;
        }
    }
}

extern "C" int trec_score_prep(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize,
                               int32_t dtype, void* out, float* out_sqnorm, void* stream)
{
    TREC_REQUIRE(repr && out, "trec_score_prep: null pointer");
    TREC_REQUIRE(d >= 1 && kpad >= d && kpad % 2 == 0, "trec_score_prep: need kpad >= d, kpad even");
    TREC_REQUIRE(dtype == 0 || dtype == 1, "trec_score_prep: dtype must be 0 or 1");
    if (n == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d % 4 == 0 && kpad % 4 == 0 && kpad <= 256 && ((uintptr_t)repr % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        const int g = kpad >= 128 ? 32 : kpad / 4;              // 8, 16 or 32 lanes per row
        const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
        if (g == 32) hipLaunchKernelGGL(score_prep_vec4_kernel<32>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, normalize, dtype, out, out_sqnorm);
        else if (g == 16) hipLaunchKernelGGL(score_prep_vec4_kernel<16>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, normalize, dtype, out, out_sqnorm);
        else hipLaunchKernelGGL(score_prep_vec4_kernel<8>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, normalize, dtype, out, out_sqnorm);
        return trec_check_launch("trec_score_prep");
    }
    hipLaunchKernelGGL(score_prep_kernel, dim3((unsigned)ceil_div64(n * 64, 256)), dim3(256), 0, st,
                       repr, n, d, kpad, normalize, dtype, out, out_sqnorm);
    return trec_check_launch("trec_score_prep");
}
