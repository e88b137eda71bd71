// tensorrec_amd/csrc/score_blockmax_i8.hip -- K2q: stage 0 of the cascaded exact top-k, the int8 pre-filter.
//
// The bf16 stage-1 kernel (score_blockmax.hip) sits at the chip's power limit, so the way to make the exact top-k faster is
// to do FEWER bf16 flops.  The int8 MFMA contracts twice as many elements per cycle as the bf16 one and its arithmetic is
// EXACT (int8 x int8 products, int32 accumulation: no rounding at all).  Operand rows are quantised to int8
// (trec_score_prep_i8: q = clamp(rint(x / scale), +-127); users: ONE scale a, items: one scale b_s per superblock), so that
// for a user u   score(u, i) ~ a b_s (sum_k q_u[k] q_i[k] + bq_i) + b_u   and the maximum over the items of a superblock is an
// INTEGER maximum of the raw accumulators: the same one-v_max3-per-accumulator-pair epilogue as the bf16 kernel, with the
// item bias (in integer units of a b_s) as the initial accumulator.  The quantisation error of every row is measured, not
// assumed (|x - scale q| per row, clipping included), which gives a proven bound e(u, s) >= |int8 score - fp32 score| per
// user and superblock (i8_pair_err, score_common.hpp); superblocks whose upper bound M8 + e lies below the user's k-th
// largest lower bound M8 - e cannot hold a top-k item and never reach the bf16 stage (csrc/topk_cascade.hip).  At
// 1M x 1M, d = 128, normalised rows: e ~ 0.021, 4.3% of the (superblock, user) pairs survive.
//
// Replaces (as a filter in front of them) tf.matmul of tensorrec/prediction_graphs.py:49-50 + the first tf.nn.top_k of
// tensorrec/recommendation_graphs.py:80; nothing it computes is returned to the caller -- survivors are re-scored in bf16
// (bounded again) and finally in fp32, bit-identical to the oracle.
//
// Two forms of the kernel: blockmax_i8_kernel on v_mfma_i32_32x32x32_i8 (the first one; tuning blockmax_i8_mfma = 0) and
// blockmax_i8x16_kernel on v_mfma_i32_16x16x64_i8 (the default: 17% more work per joule at the power cap, see its header).
// Plan of the first form (K = 128 bytes per row): a wave owns NCB x 32 users whose int8 fragments stay in registers (NCB x 4
// k-steps x 4 VGPRs); item tiles of 128 rows x 128 B = 16 KB are double-buffered in LDS through global_load_lds with the
// 16-byte-chunk XOR swizzle of score_gemm.hip; a 32-item block is 4 k-steps of NCB MFMAs fed by ONE ds_read_b128 each; the
// block's epilogue (8 v_max3_i32 per accumulator) runs right after its last k-step, the next block's bias row is read straight
// into accumulator 0.  Lane & 31 is the user (acc = mfma(items, users)), exactly the orientation of the bf16 kernel.
#include "score_common.hpp"
#include <math.h>
#include <limits.h>
#include <type_traits>

namespace {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int BNQ = 128;        // item rows per tile (four 32-row MFMA blocks)

// TK > 0: the kernel also keeps, per user, the TK largest LOWER bounds M - e(u, s) of the superblocks of its chunk (sorted
// registers, one insertion per superblock end) and writes them to chunk_top: the k-th largest over the chunks' lists is
// tau8 -- no pass over the 7.8 GB table for it.
template <int KT, bool BIAS, int NCB, int WPS, int TK>
__global__ __launch_bounds__(256, WPS) void blockmax_i8_kernel(ScoreParams p)
{
    constexpr int RB = KT;                   // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row (8 at K = 128)
    constexpr int KS = KT / 32;              // MFMA k-steps per block
    constexpr int TILE_BYTES = BNQ * RB;
    constexpr int NSLOT = BNQ * CH / 256;    // 16-byte staging slots per thread per tile
    constexpr int NBLK = BNQ / 32;
    constexpr int NSTEP = NBLK * KS;
    static_assert(KT == 64 || KT == 128 || KT == 256, "int8 BLOCKMAX covers K = 64 / 128 / 256");

    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][TILE_BYTES] item tiles | [2][BNQ] integer item biases
    int* side = (int*)(smem + 2 * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int rblock = blockIdx.x % p.n_rblocks;
    const int chunk = blockIdx.x / p.n_rblocks;
    const int64_t r_base = ((int64_t)rblock * 4 + wave) * (NCB * 32);
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BNQ - 1) / BNQ);

    // ---- resident user fragments: lane holds k = 32 ks + 16 half + 0..15 of its user ----
    v4i32 rfq[NCB][KS];
    float r_bias[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        int64_t row = r_base + cb * 32 + l31;
        if (row >= p.n_r) row = p.n_r - 1;                       // clamped rows are never written
        const char* src = (const char*)p.R + row * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rfq[cb][ks] = *(const v4i32*)(src + (ks * 2 + half) * 16);
        r_bias[cb] = (BIAS && p.r_bias) ? p.r_bias[row] : 0.f;
    }

    // ---- staging: slot q = i*256 + tid -> (row, physical chunk); source offsets fixed per thread ----
    int slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CH, pc = q % CH;
        const int sw = CH >= 16 ? (row & 15) : (CH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3));
        slot_off[i] = row * RB + ((pc ^ sw) * 16);
    }
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    // integer item biases (units of the scale product): one table per user scale class
    const int* t_bias_q = p.t_bias ? (const int*)p.t_bias + (p.wg_class ? (int64_t)p.wg_class[rblock] * p.bias_stride : 0) : nullptr;
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BNQ;
        const bool clamp = row0 + BNQ > p.n_t;                   // wave-uniform: only the very last tile
        if (BIAS && wave < 2) {                                  // 128 bias words: waves 0 and 1, one 4-byte-per-lane DMA each
            int64_t g = row0 + wave * 64 + lane;
            if (g >= p.n_t) g = p.n_t - 1;                       // duplicate of the last valid item: max unchanged
            if (t_bias_q) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t_bias_q + g),
                                                 (__attribute__((address_space(3))) void*)(side + buf * BNQ + wave * 64), 4, 0, 0);
            } else {
                side[buf * BNQ + wave * 64 + lane] = 0;
            }
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BNQ * RB);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off[i];
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);
                const int row = (i * 256 + tid) / CH;
                if (row > last) off -= (row - last) * RB;
            }
            char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;       // wave-uniform; lane*16 is implicit
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // per-lane LDS offsets of the KS operand chunks of "my" item row inside a 32-row block
    int koff[KS];
    {
        const int sw = CH >= 16 ? (l31 & 15) : (CH == 8 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3));
        // rows l31 + 32 j have the same swizzle: 32 j leaves (row & 15), ((row >> 1) & 7) and ((row >> 2) & 3) alone
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) koff[ks] = l31 * RB + (((ks * 2 + half) ^ sw) * 16);
    }

    v16i32 acc[NCB];
    int bm[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bm[cb] = INT_MIN;
    float top[TK ? NCB : 1][TK ? TK : 1];
    float e_nx[TK ? NCB : 1], e_ex[TK ? NCB : 1], e_cu[TK ? NCB : 1];
    if (TK) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            int64_t row = r_base + cb * 32 + l31;
            if (row >= p.n_r) row = p.n_r - 1;
            e_nx[cb] = p.r_err[row * 4]; e_ex[cb] = p.r_err[row * 4 + 1]; e_cu[cb] = p.r_err[row * 4 + 2];
#pragma unroll
            for (int j = 0; j < TK; ++j) top[cb][j] = -INFINITY;
        }
    }

    auto read_c0 = [&](v16i32& c, const int* sdi) {           // integer item biases of the block's 16 rows of this half-wave
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4i32 tb4 = *(const v4i32*)(sdi + 8 * q);
            c[4 * q] = tb4[0]; c[4 * q + 1] = tb4[1]; c[4 * q + 2] = tb4[2]; c[4 * q + 3] = tb4[3];
        }
    };

    auto tile_body = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char* tb = smem + buf * TILE_BYTES;
        const int* sd = side + buf * BNQ + 4 * half;
        v4i32 tf[3];
        if (BIAS) read_c0(acc[0], sd);
        tf[0] = *(const v4i32*)(tb + koff[0]);
        tf[1] = *(const v4i32*)(tb + (KS > 1 ? koff[1 % KS] : 32 * RB + koff[0]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int blk = s / KS, ks = s % KS;
            if (s + 2 < NSTEP)
                tf[(s + 2) % 3] = *(const v4i32*)(tb + ((s + 2) / KS) * 32 * RB + koff[(s + 2) % KS]);
            if (ks == 0) {
                if (BIAS) {
#pragma unroll
                    for (int cb = NCB - 1; cb >= 0; --cb)
                        acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(tf[s % 3], rfq[cb][0], acc[0], 0, 0, 0);
                } else {
                    const v16i32 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(tf[s % 3], rfq[cb][0], z, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(tf[s % 3], rfq[cb][ks], acc[cb], 0, 0, 0);
            }
            if (ks == KS - 1) {
                // block epilogue: accumulator 0 first, then it takes the next block's bias row while the others finish
#pragma unroll
                for (int j = 0; j < 8; ++j) bm[0] = max(max(bm[0], acc[0][2 * j]), acc[0][2 * j + 1]);
                if (BIAS && blk + 1 < NBLK) read_c0(acc[0], sd + 32 * (blk + 1));
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int cb = 1; cb < NCB; ++cb) bm[cb] = max(max(bm[cb], acc[cb][2 * j]), acc[cb][2 * j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // integer score units -> float: a_user * (item scale of the superblock); a_user = the scale of this workgroup's users
    const float a_user = p.wg_scale ? p.wg_scale[rblock] : p.scales[0];
    // the NEXT superblock's statistics are fetched while the first tile of the current one is computed: issued at a
    // superblock end they would sit, fresh, in front of the s_waitcnt vmcnt(0) that closes every tile
    const int64_t sb0 = t_begin / ((int64_t)p.sb_tiles * BNQ);
    f32x4 ss_cur = *(const f32x4*)(p.sb_stats + sb0 * 4), ss_next = ss_cur;
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        if ((t % p.sb_tiles) == 0 && t + p.sb_tiles < n_tiles)
            ss_next = *(const f32x4*)(p.sb_stats + (sb0 + t / p.sb_tiles + 1) * 4);
        if (buf == 0) tile_body(std::integral_constant<int, 0>{});
        else tile_body(std::integral_constant<int, 1>{});

        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
            // end of a superblock: combine the two half-wave maxima of each user, convert, add the user bias, store, reset
            const int64_t sb = sb0 + t / p.sb_tiles;
            const float scale = a_user * ss_cur[0];
            const float yh = TK ? ss_cur[1] : 0.f, dy = TK ? ss_cur[2] : 0.f, db = TK ? a_user * ss_cur[3] : 0.f;
            ss_cur = ss_next;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int o = __shfl_xor(bm[cb], 32, 64);
                const int m = bm[cb] > o ? bm[cb] : o;
                float v = (float)m * scale;                   // (exact below 2^24; beyond it one rounding, charged in i8_pair_err)
                if (BIAS) v = v + r_bias[cb];
                const int64_t u = r_base + cb * 32 + l31;
                if (half == 0 && u < p.n_r) p.blockmax[sb * p.bm_stride + u] = v;
                bm[cb] = INT_MIN;
                if (TK) {
                    float lb = v - i8_pair_err(e_nx[cb], e_ex[cb], e_cu[cb], yh, dy, db, KT);
                    lb = (lb == lb) ? lb : -INFINITY;          // a NaN certifies nothing
                    if (p.top_tag) lb = lb_tag(lb, t / p.sb_tiles);      // (uniform) the superblock's index inside this chunk
                    // sorted insertion into a descending list, one v_med3 per slot: new t_j = median(t_j, t_{j-1}, lb)
                    // (lb <= t_j: t_j stays | t_j < lb <= t_{j-1}: lb lands here | lb > t_{j-1}: t_{j-1} moves down)
#pragma unroll
                    for (int j = TK - 1; j >= 1; --j) top[cb][j] = __builtin_amdgcn_fmed3f(top[cb][j], top[cb][j - 1], lb);
                    top[cb][0] = fmaxf(top[cb][0], lb);
                }
            }
        }
        if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (TK) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int64_t u = r_base + cb * 32 + l31;
            if (half == 0 && u < p.n_r) {
#pragma unroll
                for (int j = 0; j < TK; ++j) p.chunk_top[((int64_t)chunk * TK + j) * p.bm_stride + u] = top[cb][j];
            }
        }
    }
}

// ---- the same stage on v_mfma_i32_16x16x64_i8 ------------------------------------------------------------------------
// Bare MFMA streams on random operands, ~1 s each at the 1.3 kW cap (scripts/probe/mfma_stream.hip): i8 32x32x32 3.45 Pop/s,
// i8 16x16x64 4.05 Pop/s (bf16 32x32x16: 1.80 PF).  The 16x16x64 form does the same work per operand byte (a 16-item x 64-k
// fragment, one ds_read_b128, feeds 8 MFMAs of 16 users each = the 128 users of a wave) and needs half the accumulator
// registers (8 x 4).  Lane = (k-group g = lane >> 4 of the operands | row-group g of the result, user / item row lane & 15):
// the maxima of the four row-groups are combined by two shuffles at a superblock end, after which row-group g owns NUB / 4
// of the user blocks (their table stores, bound constants and top-k register lists: NUB / 4 x TK registers, not NUB x TK).
// NUB: 16-user blocks per wave (8: 128 users, 12: 192 users); NW: waves per workgroup sharing one item tile stream (4: two
// workgroups per CU, 8: one)
// BT: item rows per tile (one barrier per tile)
// RDL: where a step's LDS operand prefetch (for step s + 2) sits.  0: before the step's MFMAs (the first form).  The
// compiler closes every block start with s_waitcnt lgkmcnt(0) -- a full drain, although the operands the MFMAs need were
// read two steps earlier -- so the prefetch issued right in front of it is waited for at its full LDS latency, once per 24
// MFMAs.  1 (default): the prefetch is issued AFTER the step's MFMAs, so the next drain finds it a whole group of max3 old:
// 78.1 -> 75.7 ms at 1M x 1M on one box (0.656 -> 0.677 of 5 Pop/s).  2: in the middle of the MFMAs (77.2).  3: after them,
// with the block's max3 interleaved between the MFMAs by sched_group_barrier (75.5: the same -- two waves per SIMD already
// cover each other's max3 phase).
template <int KT, bool BIAS, int TK, int NUB, int NW, int BT, int RDL = 1>
__global__ __launch_bounds__(NW * 64, 8 / NW) void blockmax_i8x16_kernel(ScoreParams p)
{
    constexpr int NT = NW * 64;              // threads per workgroup
    constexpr int OW = NUB / 4;              // user blocks a row-group owns at superblock ends
    static_assert(NUB % 4 == 0, "user blocks per wave must split over the four row-groups");
    constexpr int RB = KT;                   // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row (8 at K = 128, 4 at K = 64)
    constexpr int KS = KT / 64;              // MFMA k-steps per block
    constexpr int TILE_BYTES = BT * RB;
    constexpr int NSLOT = BT * CH / NT;
    constexpr int NBLK = BT / 16;           // 16-item blocks per tile
    constexpr int NSTEP = NBLK * KS;
    static_assert(KT == 64 || KT == 128, "int8 16x16x64 BLOCKMAX covers K = 64 / 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][TILE_BYTES] item tiles | [2][BT] integer item biases
    int* side = (int*)(smem + 2 * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lu = lane & 15;
    const int rblock = blockIdx.x % p.n_rblocks;
    const int chunk = blockIdx.x / p.n_rblocks;
    // an idle workgroup (trec_user_prep_sorted sizes the user layout by a bound the host knows; the workgroups beyond the padded
    // row count carry scale 0): nothing to compute, nothing written -- its users' thresholds are +inf
    if (p.wg_scale && p.wg_scale[rblock] == 0.f) return;
    const int64_t r_base = ((int64_t)rblock * NW + wave) * (NUB * 16);
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BT - 1) / BT);
    // physical 16-byte chunk = logical chunk ^ swz(row): 8 consecutive rows of one logical chunk (what 8 consecutive lanes
    // read) land on 8 distinct chunk positions = all 32 banks
    auto swz = [](int row) { return CH == 8 ? (row & 7) : ((row >> 1) & 3); };

    // ---- resident user fragments: lane holds k = 64 ks + 16 g + 0..15 of user lu of each block ----
    v4i32 rfq[NUB][KS];
#pragma unroll
    for (int ub = 0; ub < NUB; ++ub) {
        int64_t row = r_base + ub * 16 + lu;
        if (row >= p.n_r) row = p.n_r - 1;                       // clamped rows are never written
        const char* src = (const char*)p.R + row * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rfq[ub][ks] = *(const v4i32*)(src + (ks * 4 + g) * 16);
    }
    // the OW users this lane owns at superblock ends
    int64_t own_u[OW];
    float own_bias[OW], e_nx[OW], e_ex[OW], e_cu[OW];
    float top[TK ? OW : 1][TK ? TK : 1];
#pragma unroll
    for (int o = 0; o < OW; ++o) {
        own_u[o] = r_base + (OW * g + o) * 16 + lu;
        const int64_t row = own_u[o] < p.n_r ? own_u[o] : p.n_r - 1;
        own_bias[o] = (BIAS && p.r_bias) ? p.r_bias[row] : 0.f;
        if (TK) {
            e_nx[o] = p.r_err[row * 4]; e_ex[o] = p.r_err[row * 4 + 1]; e_cu[o] = p.r_err[row * 4 + 2];
#pragma unroll
            for (int j = 0; j < TK; ++j) top[o][j] = -INFINITY;
        }
    }

    int slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * NT + tid;
        const int row = q / CH, pc = q % CH;
        slot_off[i] = row * RB + ((pc ^ swz(row)) * 16);
    }
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    const int* t_bias_q = p.t_bias ? (const int*)p.t_bias + (p.wg_class ? (int64_t)p.wg_class[rblock] * p.bias_stride : 0) : nullptr;
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BT;
        const bool clamp = row0 + BT > p.n_t;                   // wave-uniform: only the very last tile
        if (BIAS && wave < BT / 64) {
            int64_t gi = row0 + wave * 64 + lane;
            if (gi >= p.n_t) gi = p.n_t - 1;                     // duplicate of the last valid item: max unchanged
            if (t_bias_q) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t_bias_q + gi),
                                                 (__attribute__((address_space(3))) void*)(side + buf * BT + wave * 64), 4, 0, 0);
            } else {
                side[buf * BT + wave * 64 + lane] = 0;
            }
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BT * RB);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off[i];
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);
                const int row = (i * NT + tid) / CH;
                if (row > last) off -= (row - last) * RB;
            }
            char* dst = smem + buf * TILE_BYTES + (i * NT + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // per-lane LDS offsets of the KS operand chunks of item row lu of a 16-row block (rows lu + 16 b share the swizzle)
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = lu * RB + (((ks * 4 + g) ^ swz(lu)) * 16);

    v4i32 acc[NUB];
    int bm[NUB];
#pragma unroll
    for (int ub = 0; ub < NUB; ++ub) bm[ub] = INT_MIN;

    auto tile_body = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char* tb = smem + buf * TILE_BYTES;
        const int* sd = side + buf * BT + 4 * g;                // the block's integer biases of result rows 4 g .. 4 g + 3
        v4i32 tf[3];
        v4i32 c0 = {0, 0, 0, 0};
        if (BIAS) c0 = *(const v4i32*)sd;
        tf[0] = *(const v4i32*)(tb + koff[0]);
        tf[1] = *(const v4i32*)(tb + (KS > 1 ? koff[1 % KS] : 16 * RB + koff[0]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int blk = s / KS, ks = s % KS;
            auto prefetch = [&]() {
                if (s + 2 < NSTEP) tf[(s + 2) % 3] = *(const v4i32*)(tb + ((s + 2) / KS) * 16 * RB + koff[(s + 2) % KS]);
            };
            if (RDL == 0) prefetch();
            if (ks == 0) {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub) {
                    acc[ub] = __builtin_amdgcn_mfma_i32_16x16x64_i8(tf[s % 3], rfq[ub][0], c0, 0, 0, 0);
                    if (RDL == 2 && ub == NUB / 2 - 1) { __builtin_amdgcn_sched_barrier(0); prefetch(); __builtin_amdgcn_sched_barrier(0); }
                }
                if (RDL == 1) __builtin_amdgcn_sched_barrier(0);
                if (BIAS && blk + 1 < NBLK) c0 = *(const v4i32*)(sd + 16 * (blk + 1));     // lands under this block's MFMAs
            } else {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub) {
                    acc[ub] = __builtin_amdgcn_mfma_i32_16x16x64_i8(tf[s % 3], rfq[ub][ks], acc[ub], 0, 0, 0);
                    if (RDL == 2 && ub == NUB / 2 - 1) { __builtin_amdgcn_sched_barrier(0); prefetch(); __builtin_amdgcn_sched_barrier(0); }
                }
                if (RDL == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (RDL == 1 || RDL == 3) prefetch();
            if (ks == KS - 1) {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub) {
                    bm[ub] = max(max(bm[ub], acc[ub][0]), acc[ub][1]);
                    bm[ub] = max(max(bm[ub], acc[ub][2]), acc[ub][3]);
                }
            }
            if (RDL == 3) {
                // pin the order inside the step: 3 MFMAs ahead, then (2 max3, 1 MFMA) pairs, the prefetch read in the middle
                if (ks == KS - 1) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
                    for (int i = 0; i < NUB - 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i == (NUB - 3) / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NUB / 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NUB - NUB / 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float a_user = p.wg_scale ? p.wg_scale[rblock] : p.scales[0];      // the scale of this workgroup's users
    const int64_t sb0 = t_begin / ((int64_t)p.sb_tiles * BT);
    f32x4 ss_cur = *(const f32x4*)(p.sb_stats + sb0 * 4), ss_next = ss_cur;
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        if ((t % p.sb_tiles) == 0 && t + p.sb_tiles < n_tiles)
            ss_next = *(const f32x4*)(p.sb_stats + (sb0 + t / p.sb_tiles + 1) * 4);
        if (buf == 0) tile_body(std::integral_constant<int, 0>{});
        else tile_body(std::integral_constant<int, 1>{});

        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
            // end of a superblock: combine the four row-groups' maxima of every user; row-group g then finishes its OW blocks
            const int64_t sb = sb0 + t / p.sb_tiles;
            const float scale = a_user * ss_cur[0];
            const float yh = TK ? ss_cur[1] : 0.f, dy = TK ? ss_cur[2] : 0.f, db = TK ? a_user * ss_cur[3] : 0.f;
            ss_cur = ss_next;
            int m[NUB];
#pragma unroll
            for (int ub = 0; ub < NUB; ++ub) {
                int x = bm[ub];
                const int o1 = __shfl_xor(x, 16, 64);
                x = x > o1 ? x : o1;
                const int o2 = __shfl_xor(x, 32, 64);
                m[ub] = x > o2 ? x : o2;
                bm[ub] = INT_MIN;
            }
#pragma unroll
            for (int o = 0; o < OW; ++o) {
                const int mo = g == 0 ? m[o] : (g == 1 ? m[OW + o] : (g == 2 ? m[2 * OW + o] : m[3 * OW + o]));
                float v = (float)mo * scale;                  // (exact below 2^24; beyond it one rounding, charged in i8_pair_err)
                if (BIAS) v = v + own_bias[o];
                if (own_u[o] < p.n_r) p.blockmax[sb * p.bm_stride + own_u[o]] = v;
                if (TK) {
                    float lb = v - i8_pair_err(e_nx[o], e_ex[o], e_cu[o], yh, dy, db, KT);
                    lb = (lb == lb) ? lb : -INFINITY;          // a NaN certifies nothing
                    if (p.top_tag) lb = lb_tag(lb, t / p.sb_tiles);      // (uniform) the superblock's index inside this chunk
#pragma unroll
                    for (int j = TK - 1; j >= 1; --j) top[o][j] = __builtin_amdgcn_fmed3f(top[o][j], top[o][j - 1], lb);
                    top[o][0] = fmaxf(top[o][0], lb);
                }
            }
        }
        if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (TK) {
#pragma unroll
        for (int o = 0; o < OW; ++o)
            if (own_u[o] < p.n_r) {
#pragma unroll
                for (int j = 0; j < TK; ++j) p.chunk_top[((int64_t)chunk * TK + j) * p.bm_stride + own_u[o]] = top[o][j];
            }
    }
}

template <int KT, bool BIAS, int TK, int NUB, int NW, int BT, int RDL = 1>
int launch_i8x16(ScoreParams p, int sb_rows, hipStream_t st)
{
    constexpr int LDS = 2 * BT * KT + 2 * BT * 4;
    auto kern = blockmax_i8x16_kernel<KT, BIAS, TK, NUB, NW, BT, RDL>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.n_rblocks = (int)ceil_div64(p.n_r, NW * NUB * 16);
    p.sb_tiles = sb_rows / BT;
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), LDS, st, p);
    return trec_check_launch("trec_score_gemm_blockmax_i8 (16x16x64)");
}

template <int KT, bool BIAS, int NCB, int WPS, int TK>
int launch_i8(ScoreParams p, int sb_rows, hipStream_t st)
{
    constexpr int LDS = 2 * BNQ * KT + 2 * BNQ * 4;
    auto kern = blockmax_i8_kernel<KT, BIAS, NCB, WPS, TK>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.n_rblocks = (int)ceil_div64(p.n_r, 4 * NCB * 32);
    p.sb_tiles = sb_rows / BNQ;
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm_blockmax_i8");
}

// fp32 rows -> int8 rows [n, kpad] (zero padded), q = clamp(rint(x / scale), +-127), + what the bound needs per row:
// {||x||, ||x - scale q||} (the actual quantisation error of this row, clipping included).  Users: ONE scale (*scale_ptr).
// Items: one scale per superblock of sb_rows rows, sb_stats[s][0] (= the superblock's max |y| / 127: no item clips), and the
// running maxima sb_stats[s][1] = max ||y|| + ||dy|| (>= ||scale q||), sb_stats[s][2] = max ||dy|| over the superblock's rows.
// wg_rows > 0 (users with scale classes): row r is quantised with scale_ptr[r / wg_rows] -- the users are sorted by class
// and the wg_rows rows of one int8 workgroup share a scale.
template <int G>
__global__ __launch_bounds__(256) void prep_i8_kernel(const float* __restrict__ x, int64_t n, int d, int kt,
                                                     const float* __restrict__ scale_ptr, int sb_rows,
                                                     float* __restrict__ sb_stats, signed char* __restrict__ out_q,
                                                     float2* __restrict__ row_stats, int wg_rows)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool ok = row < n;
    const int sub = threadIdx.x % G;
    const int64_t sb = sb_rows ? (ok ? row : n - 1) / sb_rows : 0;
    const float scale = sb_rows ? sb_stats[sb * 4] : (wg_rows > 0 ? scale_ptr[(ok ? row : n - 1) / wg_rows] : *scale_ptr);
    const float inv = 1.0f / scale;
    const float* xr = x + (ok ? row : 0) * (int64_t)d;
    float sw = 0.f, se = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && c < d) {
            if ((d & 3) == 0) v = *(const f32x4*)(xr + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) v[e] = xr[c + e];
            }
        }
        unsigned int pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float q = rintf(v[e] * inv);
            q = fminf(fmaxf(q, -127.f), 127.f);
            if (!(q == q)) q = 0.f;                                     // NaN input: the error norm below turns NaN -> the bound turns inf
            const float err = v[e] - q * scale;
            sw = fmaf(v[e], v[e], sw);
            se = fmaf(err, err, se);
            pk |= ((unsigned int)(int)q & 0xffu) << (8 * e);
        }
        if (ok && c < kt) *(unsigned int*)(out_q + row * (int64_t)kt + c) = pk;
    }
    for (int off = G / 2; off > 0; off >>= 1) { sw += __shfl_xor(sw, off, 64); se += __shfl_xor(se, off, 64); }
    float nw = sqrtf(sw), ne = sqrtf(se);
    if (sub == 0 && ok) row_stats[row] = make_float2(nw, ne);
}

// one workgroup per superblock: sb_stats[s][1] = max ||y|| + ||dy||, sb_stats[s][2] = max ||dy|| over its rows (NaN -> inf)
__global__ __launch_bounds__(256) void sb_reduce_kernel(const float2* __restrict__ row_stats, int64_t n, int sb_rows,
                                                       float* __restrict__ sb_stats)
{
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * sb_rows, r1 = (r0 + sb_rows < n) ? r0 + sb_rows : n;
    float g0 = 0.f, g1 = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const float2 st = row_stats[r];
        float a = st.x + st.y, b = st.y;
        if (a != a) a = INFINITY;
        if (b != b) b = INFINITY;
        g0 = fmaxf(g0, a); g1 = fmaxf(g1, b);
    }
    for (int off = 32; off > 0; off >>= 1) { g0 = fmaxf(g0, __shfl_xor(g0, off, 64)); g1 = fmaxf(g1, __shfl_xor(g1, off, 64)); }
    __shared__ float w0[4], w1[4];
    if ((threadIdx.x & 63) == 0) { w0[threadIdx.x >> 6] = g0; w1[threadIdx.x >> 6] = g1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sb_stats[s * 4 + 1] = fmaxf(fmaxf(w0[0], w0[1]), fmaxf(w0[2], w0[3]));
        sb_stats[s * 4 + 2] = fmaxf(fmaxf(w1[0], w1[1]), fmaxf(w1[2], w1[3]));
    }
}

// sb_stats[s][0] = (max |y| over the rows of superblock s) / 127 -- 1.0 for an all-zero or non-finite superblock
__global__ __launch_bounds__(1024) void sb_scale_kernel(const float* __restrict__ x, int64_t n, int d, int sb_rows,
                                                       float* __restrict__ sb_stats)
{
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * sb_rows;
    const int64_t r1 = (r0 + sb_rows < n) ? r0 + sb_rows : n;
    const int64_t n_elem = (r1 - r0) * d;
    const float* xs = x + r0 * (int64_t)d;
    float am = 0.f;
    if ((d & 3) == 0) {
        for (int64_t i = (int64_t)threadIdx.x * 4; i < n_elem; i += 4096) {
            const f32x4 v = *(const f32x4*)(xs + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float a = fabsf(v[e]); am = (a > am || a != a) ? a : am; }     // NaN sticks
        }
    } else {
        for (int64_t i = threadIdx.x; i < n_elem; i += 1024) { const float a = fabsf(xs[i]); am = (a > am || a != a) ? a : am; }
    }
    if (am != am) am = INFINITY;
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    __shared__ float wmax[16];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        am = 0.f;
        for (int w = 0; w < 16; ++w) am = fmaxf(am, wmax[w]);
        float sc = am / 127.0f;
        if (!(sc > 0.f) || !(sc < INFINITY)) sc = 1.0f;
        sb_stats[s * 4] = sc;
    }
}

// item biases in integer units of the scale product a_c * b_s, one table per user scale class c (blockIdx.y; classes nobody
// uses are skipped): bias_q[c][i] = rint(bias / product), gstats[2] = max |bias| (all items),
// sb_stats[s][3] = max over the classes in use of max |bias - product * bias_q| / a_c  (superblock s; zeroed by the caller) --
// the bias part of e(u, s) is a_u times it.  One workgroup per (superblock, class).
__global__ __launch_bounds__(256) void bias_i8_kernel(const float* __restrict__ bias, int64_t n, const float* __restrict__ ladder,
                                                     const int32_t* __restrict__ class_used, int sb_rows,
                                                     float* __restrict__ sb_stats, int* __restrict__ bias_q,
                                                     float* __restrict__ gstats)
{
    const int64_t s = blockIdx.x;
    const int c = blockIdx.y;
    if (class_used && class_used[c] == 0) return;
    const int64_t r0 = s * sb_rows, r1 = (r0 + sb_rows < n) ? r0 + sb_rows : n;
    const float a_c = ladder[c];
    const float sp = a_c * sb_stats[s * 4];                             // the same product the score kernel forms
    int* bq_c = bias_q + (int64_t)c * n;
    float g2 = 0.f, g3 = 0.f;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
        const float b = bias[i];
        float bq = rintf(b / sp);
        // |bq| <= 2^30: the int32 accumulator (|q.q| <= 2^21) cannot overflow.  (Round 2 clamped at 2^22 to keep the int -> float
        // conversion of the maximum exact; with a scale per user class the product a_c b_s of a small user and a superblock
        // of small items is ~1e-8 and a bias of 1 is 1e8 units: clamped, the bias error made every bound useless -- fitted
        // models fell back to bf16.  The conversion now rounds (2^-24 relative): one more unit of c_K in i8_pair_err.)
        bq = fminf(fmaxf(bq, -1073741824.f), 1073741824.f);
        if (!(bq == bq)) bq = 0.f;
        bq_c[i] = (int)bq;
        float a2 = fabsf(b), a3 = fabsf(b - bq * sp);
        if (a2 != a2) a2 = INFINITY;
        if (a3 != a3) a3 = INFINITY;
        g2 = fmaxf(g2, a2); g3 = fmaxf(g3, a3);
    }
    for (int off = 32; off > 0; off >>= 1) { g2 = fmaxf(g2, __shfl_xor(g2, off, 64)); g3 = fmaxf(g3, __shfl_xor(g3, off, 64)); }
    __shared__ float w2[4], w3[4];
    if ((threadIdx.x & 63) == 0) { w2[threadIdx.x >> 6] = g2; w3[threadIdx.x >> 6] = g3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g2 = fmaxf(fmaxf(w2[0], w2[1]), fmaxf(w2[2], w2[3]));
        g3 = fmaxf(fmaxf(w3[0], w3[1]), fmaxf(w3[2], w3[3]));
        float rel = (g3 / a_c) * 1.0000005f;                            // rounded up: a_u * rel must dominate the class's error
        if (!(rel == rel)) rel = INFINITY;
        if (__float_as_uint(rel) > *(volatile unsigned int*)(sb_stats + s * 4 + 3))
            atomicMax((unsigned int*)(sb_stats + s * 4 + 3), __float_as_uint(rel));
        if (__float_as_uint(g2) > *(volatile unsigned int*)(gstats + 2)) atomicMax((unsigned int*)(gstats + 2), __float_as_uint(g2));
    }
}

// The scale a user row WANTS: the best of max |x| / 127 (nothing clips), half and a quarter of it (the largest elements clip)
// by the quantisation error norm each would leave -- rows with a few huge elements are better off clipping them, rows
// without outliers are not.  nat[row], and gmax[0] = the largest of them (zero-initialised by the caller).
template <int G>
__global__ __launch_bounds__(256) void row_natscale_kernel(const float* __restrict__ x, int64_t n, int d,
                                                          float* __restrict__ nat, float* __restrict__ gmax)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool ok = row < n;
    const int sub = threadIdx.x % G;
    const float* xr = x + (ok ? row : 0) * (int64_t)d;
    f32x4 v[2];
    float am = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok && c < d) {
            if ((d & 3) == 0) v[it] = *(const f32x4*)(xr + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) v[it][e] = xr[c + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = fabsf(v[it][e]); am = (a > am || a != a) ? a : am; }     // NaN sticks
    }
    if (am != am) am = INFINITY;
    for (int off = G / 2; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    float best = am / 127.0f;
    if (am > 0.f && am < INFINITY) {
        float err[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int cnd = 0; cnd < 3; ++cnd) {
            const float sc = (am / 127.0f) * (cnd == 0 ? 1.0f : (cnd == 1 ? 0.5f : 0.25f));
            const float inv = 1.0f / sc;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float q = rintf(v[it][e] * inv);
                    q = fminf(fmaxf(q, -127.f), 127.f);
                    const float r = v[it][e] - q * sc;
                    err[cnd] = fmaf(r, r, err[cnd]);
                }
            for (int off = G / 2; off > 0; off >>= 1) err[cnd] += __shfl_xor(err[cnd], off, 64);
        }
        if (err[1] < err[0] && err[1] <= err[2]) best *= 0.5f;
        else if (err[2] < err[0] && err[2] < err[1]) best *= 0.25f;
    }
    if (!(best > 0.f)) best = 0.f;                                       // an all-zero row wants nothing (smallest class)
    if (sub == 0 && ok) nat[row] = best;
    float gm = (sub == 0 && ok) ? best : 0.f;
    for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(gm) > *(volatile unsigned int*)gmax) atomicMax((unsigned int*)gmax, __float_as_uint(gm));
}

// r_err[u] = {||x_u||, ||x_u - a q_u||, ck (|b_u| + max |b_i|), a_u}: the user's part of i8_pair_err (a_u: the scale its row was
// quantised with -- wg_scale[u / wg_rows] with scale classes, scales[0] otherwise)
__global__ __launch_bounds__(256) void user_err_i8_kernel(const float2* __restrict__ ustats, const float* __restrict__ user_bias,
                                                         const float* __restrict__ gstats, int kdim, int64_t n_users,
                                                         const float* __restrict__ scales, const float* __restrict__ wg_scale,
                                                         int wg_rows, float* __restrict__ r_err)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const float2 st = ustats[u];
    const float bu = user_bias ? fabsf(user_bias[u]) : 0.f;
    const float ck = (float)(kdim + 6) * 2.98023224e-07f;
    *(f32x4*)(r_err + u * 4) = (f32x4){st.x, st.y, ck * (bu + gstats[2]), wg_scale ? wg_scale[u / wg_rows] : scales[0]};
}

// sum of squares (double) and maximum magnitude (float bits) of a [n, d] matrix into ws[0] / the low word of ws[1]
// (zero-initialised by the caller): the users' scale is derived from them
__global__ __launch_bounds__(256) void sumsq_absmax_kernel(const float* __restrict__ x, int64_t n_elem, double* __restrict__ ws)
{
    double acc = 0.0;
    float am = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n_elem; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *(const f32x4*)(x + i);
        acc += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = fabsf(v[e]); am = (a > am || a != a) ? a : am; }     // NaN sticks
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n_elem & ~(int64_t)3; i < n_elem; ++i) {
            acc += (double)x[i] * x[i];
            const float a = fabsf(x[i]);
            am = (a > am || a != a) ? a : am;
        }
    if (am != am) am = INFINITY;
    for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off, 64); am = fmaxf(am, __shfl_xor(am, off, 64)); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(ws, acc);
        atomicMax((unsigned int*)(ws + 1), __float_as_uint(am));
    }
}

// scales[0] = min(clip_sigmas * rms, max |x|) / 127 (clip_sigmas <= 0: max |x| / 127, nothing clips)
__global__ void scale_from_stats_kernel(const double* __restrict__ ws, double n_elem, float clip_sigmas, float* __restrict__ scales)
{
    const double rms = sqrt(ws[0] / n_elem);
    const float am = __uint_as_float(*(const unsigned int*)(ws + 1));
    float top = am;
    if (clip_sigmas > 0.f && (float)(clip_sigmas * rms) < top) top = (float)(clip_sigmas * rms);
    float s = top / 127.0f;
    if (!(s > 0.f) || !(s < INFINITY)) s = 1.0f;                        // all-zero or non-finite input: any scale is as good
    scales[0] = s;
}

}  // namespace

// int8 operands for the pre-filter.
// side 0, users: ONE scale, scales[0] = min(clip_sigmas * rms, max |x|) / 127; out_q [n, kpad], row_stats [n][2]; workspace 16 B.
// side 1, items: one scale per superblock of sb_rows rows -- sb_stats [n_sb][4] (zero-initialised by the caller) =
//   {b_s = max |y| / 127 over the superblock, max ||y|| + ||dy||, max ||dy||, max |bias - a b_s bias_q|}; with a bias also what
//   side 2 does (the users must have been prepared: it needs scales[0]).
// side 2, item biases for the CURRENT user scale (repr / out_q / row_stats unused): bias_q = rint(bias / (scales[0] b_s)),
//   sb_stats[s][3], gstats[2] = max |bias| -- both zeroed by the caller; the item rows are quantised once, a new batch of users
//   with its own scale only needs side 2 again.
extern "C" int trec_score_prep_i8(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t side, float clip_sigmas,
                                  int32_t sb_rows, const float* bias, float* scales, double* workspace, void* out_q,
                                  float* row_stats, int32_t* bias_q, float* sb_stats, float* gstats, void* stream)
{
    TREC_REQUIRE(scales && (side == 0 || side == 1 || side == 2), "trec_score_prep_i8: side must be 0 (users), 1 (items) or 2 (item biases)");
    TREC_REQUIRE(!bias || (bias_q && gstats && side >= 1), "trec_score_prep_i8: a bias needs bias_q, gstats and side 1 / 2");
    TREC_REQUIRE(side == 0 || (sb_stats && sb_rows >= 1), "trec_score_prep_i8: the item side needs sb_stats and sb_rows");
    if (n == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (side != 2) {
        TREC_REQUIRE(repr && out_q && row_stats, "trec_score_prep_i8: null pointer");
        TREC_REQUIRE(d >= 1 && kpad >= d && kpad % 4 == 0 && kpad <= 128, "trec_score_prep_i8: need d <= kpad <= 128, kpad % 4 == 0");
        TREC_REQUIRE(((uintptr_t)repr % 16) == 0, "trec_score_prep_i8: repr must be 16-byte aligned");
        if (side == 0) {
            TREC_REQUIRE(workspace, "trec_score_prep_i8: the user side needs the 16-byte workspace");
            if (hipMemsetAsync(workspace, 0, 2 * sizeof(double), st) != hipSuccess) {
                trec_set_last_error("trec_score_prep_i8: memset failed");
                return TREC_ERR_LAUNCH;
            }
            const int64_t n_elem = n * (int64_t)d;
            unsigned sb = (unsigned)ceil_div64(n_elem, 1024 * 8);
            if (sb > 4096) sb = 4096;
            if (sb < 1) sb = 1;
            hipLaunchKernelGGL(sumsq_absmax_kernel, dim3(sb), dim3(256), 0, st, repr, n_elem, workspace);
            hipLaunchKernelGGL(scale_from_stats_kernel, dim3(1), dim3(1), 0, st, workspace, (double)n_elem, clip_sigmas, scales);
        } else {
            hipLaunchKernelGGL(sb_scale_kernel, dim3((unsigned)ceil_div64(n, sb_rows)), dim3(1024), 0, st, repr, n, d, sb_rows, sb_stats);
        }
        const int g = kpad >= 128 ? 32 : kpad / 4;
        const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
#define TREC_PQ(GV) hipLaunchKernelGGL(prep_i8_kernel<GV>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, scales, side == 1 ? sb_rows : 0, sb_stats, (signed char*)out_q, (float2*)row_stats, 0)
        if (g == 32) TREC_PQ(32);
        else if (g == 16) TREC_PQ(16);
        else TREC_PQ(8);
#undef TREC_PQ
        if (side == 1)
            hipLaunchKernelGGL(sb_reduce_kernel, dim3((unsigned)ceil_div64(n, sb_rows)), dim3(256), 0, st, (const float2*)row_stats, n, sb_rows, sb_stats);
    }
    if (side >= 1 && bias)          // one class: ladder = scales[0]
        hipLaunchKernelGGL(bias_i8_kernel, dim3((unsigned)ceil_div64(n, sb_rows), 1), dim3(256), 0, st, bias, n, scales, (const int32_t*)nullptr, sb_rows, sb_stats, bias_q, gstats);
    return trec_check_launch("trec_score_prep_i8");
}

// ---- user scale CLASSES ------------------------------------------------------------------------------------------------
// One int8 scale for all users follows the largest of them: users with small rows (and every user of a heavy-tailed or sparse
// population) are then quantised with a handful of levels and their bounds e(u, s) are useless.  Instead every user row gets
// the scale it wants (trec_score_row_scale_i8), rounded UP to a geometric ladder of classes; the users are sorted by class
// (host side: ops.score_prep_filter(sort_users=True)) so that the rows of one int8 workgroup share a class, and the integer item
// biases exist once per class in use (the kernel's C operand must be in units of ITS users' scale product).

// nat [n] = the scale each row wants, gmax [1] (zero-initialised by the caller) = their maximum
extern "C" int trec_score_row_scale_i8(const float* repr, int64_t n, int32_t d, float* nat, float* gmax, void* stream)
{
    TREC_REQUIRE(repr && nat && gmax, "trec_score_row_scale_i8: null pointer");
    TREC_REQUIRE(d >= 1 && d <= 256 && ((uintptr_t)repr % 16) == 0, "trec_score_row_scale_i8: need d <= 256 and 16-byte aligned rows");
    if (n == 0) return TREC_OK;
    const int kp = (d + 3) / 4 * 4;
    const int g = kp > 64 ? 32 : (kp > 32 ? 16 : 8);
    const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
    hipStream_t st = (hipStream_t)stream;
    if (g == 32) hipLaunchKernelGGL(row_natscale_kernel<32>, dim3(blocks), dim3(256), 0, st, repr, n, d, nat, gmax);
    else if (g == 16) hipLaunchKernelGGL(row_natscale_kernel<16>, dim3(blocks), dim3(256), 0, st, repr, n, d, nat, gmax);
    else hipLaunchKernelGGL(row_natscale_kernel<8>, dim3(blocks), dim3(256), 0, st, repr, n, d, nat, gmax);
    return trec_check_launch("trec_score_row_scale_i8");
}

// users (sorted by class): row r is quantised with wg_scale[r / wg_rows]; out_q [n, kpad], row_stats [n][2] = {||x||, ||x - a q||}
extern "C" int trec_score_prep_i8_users(const float* repr, int64_t n, int32_t d, int32_t kpad, const float* wg_scale,
                                        int32_t wg_rows, void* out_q, float* row_stats, void* stream)
{
    TREC_REQUIRE(repr && wg_scale && out_q && row_stats && wg_rows >= 1, "trec_score_prep_i8_users: bad arguments");
    TREC_REQUIRE(d >= 1 && kpad >= d && kpad % 4 == 0 && kpad <= 128, "trec_score_prep_i8_users: need d <= kpad <= 128, kpad % 4 == 0");
    TREC_REQUIRE(((uintptr_t)repr % 16) == 0, "trec_score_prep_i8_users: repr must be 16-byte aligned");
    if (n == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int g = kpad >= 128 ? 32 : kpad / 4;
    const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
#define TREC_PQU(GV) hipLaunchKernelGGL(prep_i8_kernel<GV>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, wg_scale, 0, (float*)nullptr, (signed char*)out_q, (float2*)row_stats, wg_rows)
    if (g == 32) TREC_PQU(32);
    else if (g == 16) TREC_PQU(16);
    else TREC_PQU(8);
#undef TREC_PQU
    return trec_check_launch("trec_score_prep_i8_users");
}

// integer item biases per class: bias_q [n_classes][n] (only the classes with class_used[c] != 0 are written),
// sb_stats[s][3] (zeroed by the caller) = max over those classes of the bias quantisation error per unit of user scale,
// gstats[2] (zeroed by the caller) = max |bias|; ladder [n_classes] = the classes' user scales; sb_stats[s][0] = item scales
extern "C" int trec_score_bias_i8_classes(const float* bias, int64_t n, int32_t sb_rows, const float* ladder,
                                          const int32_t* class_used, int32_t n_classes, float* sb_stats, int32_t* bias_q,
                                          float* gstats, void* stream)
{
    TREC_REQUIRE(bias && ladder && sb_stats && bias_q && gstats, "trec_score_bias_i8_classes: null pointer");
    TREC_REQUIRE(sb_rows >= 1 && n_classes >= 1 && n_classes <= 65535, "trec_score_bias_i8_classes: bad sizes");
    if (n == 0) return TREC_OK;
    hipLaunchKernelGGL(bias_i8_kernel, dim3((unsigned)ceil_div64(n, sb_rows), (unsigned)n_classes), dim3(256), 0, (hipStream_t)stream,
                       bias, n, ladder, class_used, sb_rows, sb_stats, bias_q, gstats);
    return trec_check_launch("trec_score_bias_i8_classes");
}

// rows of users one int8 workgroup covers under the current tuning (the granularity of a scale class boundary)
extern "C" int32_t trec_score_blockmax_i8_rows_per_workgroup(int32_t top_k)
{
    if (trec_get_tuning("blockmax_i8_mfma", 1) == 0) return 4 * 4 * 32;
    const int waves = trec_get_tuning("blockmax_i8_waves", 4) == 8 ? 8 : 4;
    const int users = trec_get_tuning("blockmax_i8_users", top_k > 10 ? 128 : 192) == 192 ? 192 : 128;
    return waves * users;
}

// r_err [n_users][4] = the users' part of the int8 bound ({||x||, ||x - a q||, ck (|b_u| + gstats[2]), a_u}); gstats[2] = max |item
// bias| over ALL items (item shards all-reduce it with MAX first); a_u = wg_scale[u / wg_rows] (scale classes) or scales[0]
extern "C" int trec_score_user_err_i8(const float* user_stats, const float* user_bias, const float* gstats, int32_t kdim,
                                      int64_t n_users, const float* scales, const float* wg_scale, int32_t wg_rows,
                                      float* r_err, void* stream)
{
    TREC_REQUIRE(user_stats && gstats && r_err && kdim >= 1, "trec_score_user_err_i8: bad arguments");
    TREC_REQUIRE((wg_scale && wg_rows >= 1) || scales, "trec_score_user_err_i8: need wg_scale + wg_rows or scales");
    TREC_REQUIRE(((uintptr_t)r_err % 16) == 0, "trec_score_user_err_i8: r_err must be 16-byte aligned");
    if (n_users == 0) return TREC_OK;
    hipLaunchKernelGGL(user_err_i8_kernel, dim3((unsigned)ceil_div64(n_users, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)user_stats, user_bias, gstats, kdim, n_users, scales, wg_scale, wg_rows, r_err);
    return trec_check_launch("trec_score_user_err_i8");
}

// blockmax[s * bm_stride + u] = scales[0] * b_s * max over the items of superblock s of (q_u . q_i + bq_i) + user_bias[u]
// (users_q / items_q: int8 [n, kpad]; item_bias_q: int32 [n_items] or NULL; scales / sb_stats: trec_score_prep_i8).
// user_err + chunk_top + top_k (all or none): chunk_top [n_chunks_eff * top_k][bm_stride] receives, per chunk of superblocks
// and user, the top_k largest lower bounds M - e(u, s), sorted descending, -inf padded; top_k = 10 or 16;
// n_chunks_eff = ceil(n_items / chunk_len), chunk_len = ceil(ceil(n_items / n_chunks) / sb_rows) * sb_rows.
// top_k | 0x100 (TREC_TOPK_TAGGED): every lower bound carries the index of its superblock INSIDE its chunk in its low 12 bits
// (lb_tag, score_common.hpp: the tagged value is a slightly smaller, still valid lower bound; chunks of at most 4096 superblocks).
extern "C" int trec_score_gemm_blockmax_i8(const void* users_q, const void* items_q, int32_t kpad, int64_t n_users,
                                           int64_t n_items, const float* user_bias, const int32_t* item_bias_q,
                                           const float* scales, const float* sb_stats, int32_t sb_rows, int32_t n_chunks,
                                           float* blockmax, int64_t bm_stride, const float* user_err, float* chunk_top,
                                           int32_t top_k, const float* wg_scale, const int32_t* wg_class, int32_t wg_rows,
                                           void* stream)
{
    const int top_tag = (top_k & 0x100) ? 1 : 0;
    top_k &= 0xff;
    // wg_rows (with wg_scale): the rows per workgroup the caller laid the users out for -- the scales and bias tables are
    // indexed by workgroup, so a tuning changed between preparation and launch must fail loudly, not shift them
    TREC_REQUIRE(!wg_scale || wg_rows == trec_score_blockmax_i8_rows_per_workgroup(top_k ? top_k : 10),
                 "trec_score_gemm_blockmax_i8: the user operand was laid out for another workgroup height (tuning changed since the preparation?)");
    TREC_REQUIRE(users_q && items_q && (scales || wg_scale) && sb_stats && blockmax && bm_stride >= n_users, "trec_score_gemm_blockmax_i8: bad arguments");
    TREC_REQUIRE(!wg_class || wg_scale, "trec_score_gemm_blockmax_i8: wg_class comes with wg_scale");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_blockmax_i8: kpad must be 64 or 128");
    TREC_REQUIRE(sb_rows >= BNQ && sb_rows % BNQ == 0, "trec_score_gemm_blockmax_i8: sb_rows must be a multiple of 128");
    TREC_REQUIRE(n_users >= 1 && n_items >= 1 && n_chunks >= 1, "trec_score_gemm_blockmax_i8: empty operand");
    TREC_REQUIRE(n_items < (int64_t)1 << 31 && n_users < (int64_t)1 << 31, "trec_score_gemm_blockmax_i8: sizes must fit int32");
    TREC_REQUIRE((chunk_top != nullptr) == (user_err != nullptr) && (chunk_top ? (top_k == 10 || top_k == 16) : top_k == 0),
                 "trec_score_gemm_blockmax_i8: user_err, chunk_top and top_k (10 or 16) come together");
    ScoreParams p = {};
    p.R = users_q; p.T = items_q; p.n_r = n_users; p.n_t = n_items;
    p.chunk_len = ceil_div64(ceil_div64(n_items, n_chunks), sb_rows) * sb_rows;        // chunks are whole superblocks
    p.n_chunks = (int)ceil_div64(n_items, p.chunk_len);
    p.r_bias = user_bias; p.t_bias = (const float*)item_bias_q;
    p.blockmax = blockmax; p.bm_stride = bm_stride;
    p.scales = scales; p.sb_stats = sb_stats; p.r_err = user_err; p.chunk_top = chunk_top; p.top_k = top_k;
    p.top_tag = top_tag;
    TREC_REQUIRE(!top_tag || (top_k && p.chunk_len / sb_rows <= (1 << TREC_LB_TAG_BITS)),
                 "trec_score_gemm_blockmax_i8: tagged lower bounds need the lists and chunks of at most 4096 superblocks");
    p.wg_scale = wg_scale; p.wg_class = item_bias_q ? wg_class : nullptr; p.bias_stride = n_items;
    hipStream_t st = (hipStream_t)stream;
    const bool bias = user_bias || item_bias_q;
    // "blockmax_i8_mfma": 1 (default) = v_mfma_i32_16x16x64_i8, 0 = v_mfma_i32_32x32x32_i8 (A/B runs)
    if (trec_get_tuning("blockmax_i8_mfma", 1) != 0) {
#define TREC_I8X3(KTV, TKV, NUBV, NWV, BTV) (bias ? launch_i8x16<KTV, true, TKV, NUBV, NWV, BTV>(p, sb_rows, st) : launch_i8x16<KTV, false, TKV, NUBV, NWV, BTV>(p, sb_rows, st))
#define TREC_I8X2(KTV, TKV, NUBV, NWV) (tile == 256 && sb_rows % 256 == 0 ? TREC_I8X3(KTV, TKV, NUBV, NWV, 256) : TREC_I8X3(KTV, TKV, NUBV, NWV, 128))
#define TREC_I8X(KTV, TKV) (users == 192 ? (waves == 8 ? TREC_I8X2(KTV, TKV, 12, 8) : TREC_I8X2(KTV, TKV, 12, 4)) \
                                       : (waves == 8 ? TREC_I8X2(KTV, TKV, 8, 8) : TREC_I8X2(KTV, TKV, 8, 4)))
        const int tile = trec_get_tuning("blockmax_i8_tile", 128);             // item rows per tile: 128 or 256
        const int waves = trec_get_tuning("blockmax_i8_waves", 4);             // waves per workgroup: 4 or 8
        // users per wave: 192 (12 blocks of 16: a third fewer LDS reads and tile streams per flop; 78.8 vs 81.9 ms at 1M x 1M
        // with the 10-slot lists, profiles/r02_power_trace_i8.txt) unless the 16-slot lists need the registers
        const int users = trec_get_tuning("blockmax_i8_users", top_k > 10 ? 128 : 192);
        // the late-prefetch form exists for the default shape only (K = 128, 10-slot lists, 192 users per wave, 4 waves, 128-row tiles)
        // where a step's operand prefetch sits (see the kernel header): 1 (default, every shape) = after the step's MFMAs;
        // 0 / 2 / 3 = the first form / mid-step / pinned interleave, for the default shape only (A/B: 78.1 / 77.2 / 75.5 ms against
        // 75.7 for form 1 on one box, 1M x 1M)
        const int rdl = trec_get_tuning("blockmax_i8_rdlate", 1);
        if (kpad == 128 && top_k == 10 && users == 192 && waves == 4 && tile == 128 && rdl != 1) {
            if (rdl == 2) return bias ? launch_i8x16<128, true, 10, 12, 4, 128, 2>(p, sb_rows, st) : launch_i8x16<128, false, 10, 12, 4, 128, 2>(p, sb_rows, st);
            if (rdl == 3) return bias ? launch_i8x16<128, true, 10, 12, 4, 128, 3>(p, sb_rows, st) : launch_i8x16<128, false, 10, 12, 4, 128, 3>(p, sb_rows, st);
            return bias ? launch_i8x16<128, true, 10, 12, 4, 128, 0>(p, sb_rows, st) : launch_i8x16<128, false, 10, 12, 4, 128, 0>(p, sb_rows, st);
        }
        if (kpad == 128) return top_k == 0 ? TREC_I8X(128, 0) : (top_k == 10 ? TREC_I8X(128, 10) : TREC_I8X(128, 16));
        return top_k == 0 ? TREC_I8X(64, 0) : (top_k == 10 ? TREC_I8X(64, 10) : TREC_I8X(64, 16));
#undef TREC_I8X
#undef TREC_I8X2
#undef TREC_I8X3
    }
#define TREC_I8(KTV, TKV) (bias ? launch_i8<KTV, true, 4, 2, TKV>(p, sb_rows, st) : launch_i8<KTV, false, 4, 2, TKV>(p, sb_rows, st))
    if (kpad == 128) return top_k == 0 ? TREC_I8(128, 0) : (top_k == 10 ? TREC_I8(128, 10) : TREC_I8(128, 16));
    return top_k == 0 ? TREC_I8(64, 0) : (top_k == 10 ? TREC_I8(64, 10) : TREC_I8(64, 16));
#undef TREC_I8
}
