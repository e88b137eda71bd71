// tensorrec_amd/csrc/score_blockmax.hip -- K2, stage 1 of the two-stage exact top-k, hand-scheduled.
//
// Same mathematics and data layout as score_gemm_kernel<.., EPI_BLOCKMAX, ..> (score_gemm.hip) for the hot
// configuration -- bf16 operands, dot / cosine, K = 64 or 128: M[s][u] = max over the 512-item superblock s of
// fl-chain(b_i + sum_k u_k i_k) + b_u, the user x item contraction of tensorrec/prediction_graphs.py:50 reduced by the
// first tf.nn.top_k of rank_predictions (recommendation_graphs.py:80) to what the exact top-k needs.
//
// What differs is the instruction schedule inside a wave.  The generic kernel leaves it to the compiler, which (at
// 3 waves/SIMD = 168 VGPRs) keeps ONE LDS read in flight (`ds_read; s_waitcnt lgkmcnt(0); 2 MFMA` per k-step) and
// runs the max-chain epilogue of a 32-item block between that block's last MFMA and the next block's first one: a
// single wave keeps the matrix pipe ~50% busy and relies on its two SIMD neighbours for the rest (measured: 63%).
// Here one 64-item tile is ONE straight-line pipeline of 2*KS steps, pinned with sched_barrier:
//     step s:  ds_read for step s+2   |   2 MFMAs of step s   |   a slice of the max chain of the PREVIOUS 32-item block
// so LDS latency hides behind two steps (128 MFMA cycles) and the epilogue of block b runs under the MFMAs of block
// b+1 (two accumulator sets).  The epilogue of a tile's second block runs under the next tile's first block, across
// the barrier; it is flushed early only at superblock ends (every 8th tile).
//
// Partial tiles need no masking: staging re-reads the last valid item row (and its bias) for rows past the end, and a
// maximum is unchanged by duplicates.
#include "score_common.hpp"
#include <math.h>
#include <type_traits>

namespace {

constexpr int BN = 64;          // item rows per tile (two 32-row MFMA blocks)

__device__ __forceinline__ int swz16(int row, int ch) {
    // physical 16-byte chunk = logical chunk ^ swz(row) (same swizzle as score_gemm.hip)
    return ch >= 16 ? (row & 15) : ((row >> 1) & 7);
}

// NCB: 32-user column blocks per wave (2: 256 users per workgroup, 2-3 workgroups per CU; 4: 512 users per workgroup,
// one workgroup per CU -- half the LDS reads, tile loads and barriers per flop, the wave hides its own latencies)
// OVL: two accumulator sets, the epilogue of a block under the next block's MFMAs (false: one set, the epilogue right
// after the block -- the register plan that lets a wave own 96 users)
// GRP (grouped form, stage 2 of the int8 cascade): workgroup w owns superblock rblock_chunk[w] only; its resident rows are
// users row_index[w * rows + r] (-1 = padding row) and each maximum goes to blockmax[superblock][that user] -- the refined
// entries of the table the int8 stage wrote.
template <int KT, bool BIAS, int NCB, int WPS, bool OVL = true, int NBUF = 2, bool GRP = false>
__global__ __launch_bounds__(256, WPS) void blockmax_pipe_kernel(ScoreParams p)
{
    constexpr int RB = KT * 2;               // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row
    constexpr int KS = KT / 16;              // MFMA k-steps per block
    constexpr int TILE_BYTES = BN * RB;
    constexpr int NSLOT = BN * CH / 256;     // 16-byte staging slots per thread per tile
    constexpr int NSTEP = 2 * KS;            // pipeline steps per tile
    constexpr int NOPS = NCB * 8;            // epilogue ops of one block: 8 v_max3 per accumulator, folded into bm directly
    constexpr int OPS = (NOPS + KS - 4) / (KS - 3);     // ops per step: the epilogue runs in local steps 1 .. KS-3
    static_assert(KT == 64 || KT == 128, "pipelined BLOCKMAX covers K = 64 / 128");

    // NBUF LDS buffers: tile t is computed while tiles t+1 .. t+NBUF-1 are in flight (global_load_lds).  Two is the
    // shipped form: a third buffer (two tiles of MFMAs for a tile's loads to land) measured 1.2% SLOWER.
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [NBUF][TILE_BYTES] item tiles | [NBUF][BN] item biases
    float* side = (float*)(smem + NBUF * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int rblock = GRP ? (int)blockIdx.x : (int)(blockIdx.x % p.n_rblocks);
    const int chunk = GRP ? p.rblock_chunk[rblock] : (int)(blockIdx.x / p.n_rblocks);
    if (GRP && chunk < 0) return;                                // idle workgroup of the grouped launch
    const int64_t r_base = ((int64_t)rblock * 4 + wave) * (NCB * 32);
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BN - 1) / BN);

    // ---- resident user fragments, straight from global, once ----
    bf16x8 rfb[NCB][KS];
    float r_bias[NCB];
    int32_t dst_user[GRP ? NCB : 1];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        int64_t row = r_base + cb * 32 + l31;
        if (row >= p.n_r) row = p.n_r - 1;                       // clamped rows are never written
        if (GRP) {
            dst_user[cb] = p.row_index[row];
            row = dst_user[cb] < 0 ? 0 : dst_user[cb];           // padding rows compute on user 0, never written
        }
        const char* src = (const char*)p.R + row * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rfb[cb][ks] = *(const bf16x8*)(src + (ks * 2 + half) * 16);
        r_bias[cb] = (BIAS && p.r_bias) ? p.r_bias[row] : 0.f;
    }

    // ---- staging: slot q = i*256 + tid -> (row, physical chunk); source offsets fixed per thread ----
    int slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CH, pc = q % CH;
        slot_off[i] = row * RB + ((pc ^ swz16(row, CH)) * 16);
    }
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    // the item-bias row of a tile travels like the tile itself: wave 0 issues ONE 4-byte-per-lane global_load_lds (no
    // staging register, no commit store); a NULL item bias (user biases only) is a row of zeros
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BN;
        const bool clamp = row0 + BN > p.n_t;                    // wave-uniform: only the very last tile
        if (BIAS && wave == 0) {
            int64_t g = row0 + lane;
            if (g >= p.n_t) g = p.n_t - 1;                       // duplicate of the last valid item: max unchanged
            if (p.t_bias) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.t_bias + g),
                                                 (__attribute__((address_space(3))) void*)(side + buf * BN), 4, 0, 0);
            } else {
                side[buf * BN + lane] = 0.f;
            }
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BN * RB);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off[i];
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);
                const int row = (i * 256 + tid) / CH;                // recomputed here: the rare path owns no registers
                if (row > last) off -= (row - last) * RB;
            }
            char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;       // wave-uniform; lane*16 is implicit
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // wait until the OLDEST tile in flight has landed, leaving `younger` newer tiles' loads outstanding (loads retire in
    // issue order; wave 0 carries one more load per tile: the bias row)
    auto stage_wait = [&](int younger) {
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) {
            if (BIAS && p.t_bias && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
        } else {
            if (BIAS && p.t_bias && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSLOT + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSLOT) : "memory");
        }
    };

    // ---- per-lane LDS read offsets of the KS operand chunks of "my" item row (row l31 of a 32-row block) ----
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = l31 * RB + (((ks * 2 + half) ^ swz16(l31, CH)) * 16);
    // rows 32..63 of a tile: (l31 + 32) has the same swizzle for CH = 16 (row & 15); for CH = 8 the swizzle is
    // ((row >> 1) & 7) and 32 >> 1 = 16 leaves the low three bits alone as well -> one offset table serves both blocks.

    f32x16 accA[NCB], accB[NCB];
    float bm[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        bm[cb] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[cb][r] = -INFINITY; accB[cb][r] = -INFINITY; }
    }

    // one op of a block's epilogue: I = NCB*j + cb, j = 0..7:  bm = max3(bm, x[2j], x[2j+1]).  The user bias is added once
    // per superblock, after the maximum (fp32 addition is monotone: max_r fl(x_r + b) == fl(max_r x_r + b)).
    auto epi_op = [&](int I, f32x16 (&x)[NCB]) {
        const int cb = I % NCB, j = I / NCB;
        bm[cb] = fmaxf(fmaxf(bm[cb], x[cb][2 * j]), x[cb][2 * j + 1]);
    };
    auto read_c0 = [&](f32x16& c, const float* sdi) {      // item biases of the block's 16 rows of this half-wave
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 tb4 = *(const f32x4*)(sdi + 8 * q);
            c[4 * q] = tb4[0]; c[4 * q + 1] = tb4[1]; c[4 * q + 2] = tb4[2]; c[4 * q + 3] = tb4[3];
        }
    };
    // the NCB MFMAs of one step (acc = mfma(items, users): lane & 31 is the user).  With biases the item bias row, read
    // into the LAST accumulator, is the C operand of every first MFMA (the last one accumulates in place).
    auto mfma_step = [&](f32x16 (&acc)[NCB], const bf16x8& tfv, int ks) __attribute__((always_inline)) {
        if (ks == 0) {
            if (BIAS) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfv, rfb[cb][0], acc[NCB - 1], 0, 0, 0);
            } else {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfv, rfb[cb][0], z, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfv, rfb[cb][ks], acc[cb], 0, 0, 0);
        }
    };

    // one tile: 2*KS pipeline steps.  `buf` is a compile-time constant at the call site (LDS immediates).
    auto tile_body = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char* tb = smem + buf * TILE_BYTES;
        const float* sd = side + buf * BN + 4 * half;
        bf16x8 tf[3];
        if (BIAS) read_c0(accA[NCB - 1], sd);
        tf[0] = *(const bf16x8*)(tb + koff[0]);
        tf[1] = *(const bf16x8*)(tb + koff[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int blk = s / KS, ks = s % KS;
            // (1) LDS reads two steps ahead; block B's bias row once block B's accumulators are free
            if (s + 2 < NSTEP)
                tf[(s + 2) % 3] = *(const bf16x8*)(tb + ((s + 2) / KS) * 32 * RB + koff[(s + 2) % KS]);
            if (BIAS && s == KS - 2) read_c0(accB[NCB - 1], sd + 32);
            // (2) the MFMAs of this step
            if (blk == 0) mfma_step(accA, tf[s % 3], ks);
            else mfma_step(accB, tf[s % 3], ks);
            // (3) a slice of the epilogue of the block BEFORE this one (B of the previous tile under A, A under B);
            //     it starts one step late so that the block's last MFMA has landed
#pragma unroll
            for (int i = 0; i < OPS; ++i) {
                const int I = (ks - 1) * OPS + i;
                if (ks >= 1 && I < NOPS) {
                    if (blk == 0) epi_op(I, accB);
                    else epi_op(I, accA);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // single accumulator set: same LDS pipeline, every block's epilogue right after its last MFMA step.  The item-bias
    // row of a block is read straight into accumulator 0 (no staging registers): for block A at tile start, for block B
    // as soon as block A's epilogue is done with accumulator 0 -- the other accumulators' max chains hide the LDS
    // latency.  The first MFMA step then feeds every accumulator from it, accumulator 0 itself last (in place).
    auto tile_body_single = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char* tb = smem + buf * TILE_BYTES;
        const float* sd = side + buf * BN + 4 * half;
        bf16x8 tf[3];
        if (BIAS) read_c0(accA[0], sd);
        tf[0] = *(const bf16x8*)(tb + koff[0]);
        tf[1] = *(const bf16x8*)(tb + koff[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int blk = s / KS, ks = s % KS;
            if (s + 2 < NSTEP)
                tf[(s + 2) % 3] = *(const bf16x8*)(tb + ((s + 2) / KS) * 32 * RB + koff[(s + 2) % KS]);
            if (ks == 0) {
                if (BIAS) {
#pragma unroll
                    for (int cb = NCB - 1; cb >= 0; --cb)
                        accA[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[s % 3], rfb[cb][0], accA[0], 0, 0, 0);
                } else {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        accA[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[s % 3], rfb[cb][0], z, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    accA[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[s % 3], rfb[cb][ks], accA[cb], 0, 0, 0);
            }
            if (ks == KS - 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) epi_op(j * NCB, accA);                     // accumulator 0 first ...
                if (BIAS && blk == 0) read_c0(accA[0], sd + 32);                       // ... then it takes block B's biases
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int cb = 1; cb < NCB; ++cb) epi_op(j * NCB + cb, accA);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: NBUF - 1 tiles in flight, tile 0 landed
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < n_tiles) stage_issue(i, i);
    {
        const int inflight = (n_tiles < NBUF - 1 ? n_tiles : NBUF - 1) - 1;        // tiles issued after tile 0
        stage_wait(inflight);
    }
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (t + NBUF - 1 < n_tiles) stage_issue(t + NBUF - 1, (buf + NBUF - 1) % NBUF);
        if (OVL) {
            if (buf == 0) tile_body(std::integral_constant<int, 0>{});
            else if (buf == 1 || NBUF == 2) tile_body(std::integral_constant<int, 1>{});
            else tile_body(std::integral_constant<int, NBUF - 1>{});
        } else {
            if (buf == 0) tile_body_single(std::integral_constant<int, 0>{});
            else if (buf == 1 || NBUF == 2) tile_body_single(std::integral_constant<int, 1>{});
            else tile_body_single(std::integral_constant<int, NBUF - 1>{});
        }

        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
            // end of a superblock: finish block B now, combine the two half-wave maxima of each user, store, reset
#pragma unroll
            for (int I = 0; I < NOPS; ++I) if (OVL) epi_op(I, accB);
            const int64_t sb = t_begin / ((int64_t)p.sb_tiles * BN) + t / p.sb_tiles;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float v = fmaxf(bm[cb], __shfl_xor(bm[cb], 32, 64));
                if (BIAS) v = v + r_bias[cb];
                const int64_t u = r_base + cb * 32 + l31;
                if (GRP) {
                    if (half == 0 && dst_user[cb] >= 0) p.blockmax[(int64_t)chunk * p.bm_stride + dst_user[cb]] = v;
                } else if (half == 0 && u < p.n_r) p.blockmax[sb * p.bm_stride + u] = v;
                bm[cb] = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) accB[cb][r] = -INFINITY;        // the deferred epilogue becomes a no-op
            }
        }
        if (t + 1 < n_tiles) {
            // tile t + 1 must have landed; tiles t + 2 .. min(t + NBUF - 1, n_tiles - 1) stay in flight
            int younger = n_tiles - 2 - t;
            if (younger > NBUF - 2) younger = NBUF - 2;
            stage_wait(younger);
        }
        __syncthreads();
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
    }
}

// ---- the filter form on v_mfma_f32_16x16x32_bf16 -----------------------------------------------------------------------
// Bare MFMA streams at the 1.3 kW cap (scripts/probe/mfma_stream.hip): bf16 32x32x16 1.80 PF, bf16 16x16x32 2.10 PF -- the
// 16x16 shapes do 17% more work per joule.  Same plan as the int8 16x16x64 kernel (score_blockmax_i8.hip): lane = (k-group /
// result row-group g = lane >> 4, user / item row lane & 15); a wave owns 8 blocks of 16 users (128 VGPRs of fragments at
// K = 128); a 16-item x 32-k fragment (one ds_read_b128) feeds 8 MFMAs; 2 v_max3_f32 per 4-register accumulator; the item
// bias is the initial accumulator; the four row-groups' maxima meet by two shuffles at a superblock end and row-group g
// stores user blocks 2g, 2g + 1.  The partial sums of a score are added in a different order than in the 32x32x16 kernels:
// this form serves the FILTERS only (K2f / K2c: their bound charges every addition of the chain, in any order); the plain
// bf16 two-stage top-k, whose stage 3 must reproduce stage 1's maxima bit for bit, keeps the 32x32x16 kernel.
// GRP: the grouped launch of the cascade (one superblock per workgroup, rows through row_index), as in blockmax_pipe_kernel.
// NUB: 16-user blocks per wave (8: 128 users, 512 per workgroup, 2 workgroups per CU; 4: 64 users, 256 per workgroup, 3 per CU --
// the grouped form's alternative: its workgroups start with a dependent gather of their user rows, and more of them in flight
// hide more of that latency, at twice the item-tile reads per flop)
// RDL: the step's LDS operand prefetch is issued AFTER its MFMAs (true) instead of before (false, the default): the change
// that gave the int8 kernel 3% (score_blockmax_i8.hip) measured slightly SLOWER here (dense 151.5 vs 150.5 ms, grouped 6.79 vs
// 6.74: 8 user blocks and 4 k-steps per block leave the drain less exposed); kept as tuning blockmax_bf16_rdlate = 1.
// LIST (the cascade's refining launches, csrc/topk_candidates.hip): besides the maxima, every ITEM whose bf16 score reaches the
// user's provisional floor p.cand_floor[u] is appended to the user's candidate list (p.cand[u][slot], slot from an atomic on
// p.cand_n[u]).  In the MFMA loop that costs one v_max3 + v_max + v_cmp per 4-item accumulator; a lane whose four scores hold a
// hit copies them (ds_write_b128) with a 4-byte code to a queue in LDS that only its wave uses (slots from ballot + mbcnt: no
// atomics, no waiting); the queue is emptied -- scores + user bias compared exactly, atomics, 8-byte stores -- at the end of
// the superblock, and inside it only if it runs more than LQ_FLUSH entries full: then the 256 entries half a block step can add
// always fit (popular items of fitted catalogues hit for every user of the wave at once), so no hit is ever dropped.
constexpr int LQ_CAP = 448;               // queue entries per wave (20 bytes each: 35,840 bytes per workgroup)
constexpr int LQ_FLUSH = 192;             // a queue fuller than this is emptied before the next half block step (192 + 4 * 64 <= 448)

#ifdef TREC_CAND_DIAG
// diagnostics build only (make diag; scripts/gpu_refine_diag.sh): where a workgroup of the refining launch (GRP && LIST) spends
// its life, in units of the 100 MHz wall clock, one row per workgroup of the LAST launch -- {entry -> operands resident (row ids
// -> user rows, floors, counters, biases, first item tile), the tile loop, the superblock end (queue flush + maxima stores: issue
// time, the stores drain behind it), whole life}.  Four stamps per workgroup and plain stores: stamps around every tile body and
// atomics on shared counters (the first version) slowed the kernel itself 2x -- its numbers described the tool.
constexpr int REFINE_DIAG_WGS = 1 << 17;
__device__ unsigned long long g_refine_clk[(size_t)REFINE_DIAG_WGS * 4];      // per workgroup of the LAST launch: prologue, tile loop, superblock end, whole life
#endif

template <int KT, bool BIAS, bool GRP, int NUB = 8, bool RDL = false, bool LIST = false>
__global__ __launch_bounds__(256, NUB == 8 ? 2 : 3) void blockmax_bf16x16_kernel(ScoreParams p)
{
#ifdef TREC_CAND_DIAG
    const unsigned long long dg_t0 = wall_clock64();
    unsigned long long dg_flush = 0, dg_t1 = 0;
#endif
    constexpr int RW = 4 * NUB * 16;         // resident rows per workgroup
    constexpr int OW = NUB / 4;
    constexpr int RB = KT * 2;               // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row (16 at K = 128, 8 at K = 64)
    constexpr int KS = KT / 32;              // MFMA k-steps per block
    constexpr int TILE_BYTES = BN * RB;
    constexpr int NSLOT = BN * CH / 256;
    constexpr int NBLK = BN / 16;            // 16-item blocks per tile
    constexpr int NSTEP = NBLK * KS;
    static_assert(KT == 64 || KT == 128, "bf16 16x16x32 BLOCKMAX covers K = 64 / 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][TILE_BYTES] item tiles | [2][BN] item biases
    float* side = (float*)(smem + 2 * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lu = lane & 15;
    // LIST: per wave a queue of (four scores, code) and the floor / bias / id of its NUB * 16 users, behind the tiles
    constexpr int LQ_OFF = 2 * TILE_BYTES + 2 * BN * 4;
    f32x4* qv = (f32x4*)(smem + LQ_OFF) + wave * LQ_CAP;
    int32_t* qc = (int32_t*)(smem + LQ_OFF + 4 * LQ_CAP * 16) + wave * LQ_CAP;
    float* q_fl = (float*)(smem + LQ_OFF + 4 * LQ_CAP * 20) + wave * (NUB * 16);
    float* q_bu = q_fl + 4 * NUB * 16;
    int32_t* q_id = (int32_t*)(q_bu + 4 * NUB * 16);
    float* q_ta = (float*)(q_id + 4 * NUB * 16);                // ... and the accumulator threshold of each (read once, in the prologue)
    int qn = 0;                                                  // wave-uniform: entries in the queue
    int rblock = GRP ? (p.wg_map ? p.wg_map[blockIdx.x] : (int)blockIdx.x) : (int)(blockIdx.x % p.n_rblocks);
    if (GRP && p.capacity > 0 && p.grp_band_major && !p.wg_map) {
        // fixed-capacity layout walked band-major: consecutive workgroups take list chunk j of superblocks 0, 1, 2, ... -- the
        // users of chunk j of every superblock's (roughly ascending) list lie in one band of ~512 / (kept fraction) users, so
        // the 128 KB of gathered user rows of the workgroups running together come from L2 instead of 256-byte random reads
        // of the 256 MB table, and the operand that misses is the item superblock: a sequential, double-buffered stream
        const int n_sb_g = p.n_rblocks / p.capacity;
        rblock = (int)(blockIdx.x % n_sb_g) * p.capacity + (int)(blockIdx.x / n_sb_g);
    }
    // grouped launch, two layouts: a list (rblock_chunk[w] = the superblock of workgroup w, -1 = idle, padding rows -1 in
    // row_index) or fixed capacity (p.capacity workgroups per superblock, rblock_chunk = the superblocks' row counts)
    // dense launch over a LIST of superblocks (the cascade's "hot" superblocks -- kept by more users than the fixed capacity
    // holds -- are refined for every user: trec_score_gemm_blockmax_hot): p.rblock_chunk[blockIdx.x / n_rblocks], -1 = idle
    const bool fixed = GRP && p.capacity > 0;
    const int chunk = !GRP ? (p.rblock_chunk ? p.rblock_chunk[blockIdx.x / p.n_rblocks] : (int)(blockIdx.x / p.n_rblocks))
                           : (fixed ? rblock / p.capacity : p.rblock_chunk[rblock]);
    int rows_here = RW;                                          // fixed layout: valid rows of this workgroup
    if (fixed) {
        int cnt = p.rblock_chunk[chunk];
        if (cnt > p.capacity * RW) cnt = p.capacity * RW;
        rows_here = cnt - (rblock % p.capacity) * RW;
    }
    if (chunk < 0 || (GRP && rows_here <= 0)) return;            // idle workgroup of a grouped / listed launch
    const int64_t r_base = ((int64_t)rblock * 4 + wave) * (NUB * 16);
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BN - 1) / BN);
    // physical chunk = logical chunk ^ (row & (CH - 1)): 8 consecutive rows of one logical chunk (8 consecutive lanes) hit 8
    // distinct 16-byte positions modulo 128 bytes = all 32 banks
    auto swz = [](int row) { return row & (CH - 1); };

    // staging slot q = i * 256 + tid -> (row, physical chunk); 256 / CH rows per slot round is a multiple of CH, so the swizzle
    // term does not depend on i: ONE offset register, slot i is 256 * 16 bytes further (four separate offsets were spilled by
    // the LIST form and re-loaded from scratch in every tile step, behind s_waitcnt vmcnt(0) -- in the middle of the next tile's DMA)
    static_assert((256 / CH) % CH == 0, "swizzle term independent of the slot round");
    const int slot_off0 = (tid / CH) * RB + (((tid % CH) ^ swz(tid / CH)) * 16);
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BN;
        const bool clamp = row0 + BN > p.n_t;                    // wave-uniform: only the very last tile
        if (BIAS && wave == 0) {
            int64_t gi = row0 + lane;
            if (gi >= p.n_t) gi = p.n_t - 1;                     // duplicate of the last valid item: max unchanged
            if (p.t_bias) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.t_bias + gi),
                                                 (__attribute__((address_space(3))) void*)(side + buf * BN), 4, 0, 0);
            } else {
                side[buf * BN + lane] = 0.f;
            }
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BN * RB);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off0 + i * 4096;
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);
                const int row = (i * 256 + tid) / CH;
                if (row > last) off -= (row - last) * RB;
            }
            char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // grouped launches: the first item tile is requested BEFORE the dependent user gathers (row_index -> rows), not behind
    // them (refining launch 10.25 -> 9.80 ms at 1M x 1M); the dense launches keep their order (contiguous rows, nothing to hide)
    if (GRP) stage_issue(0, 0);
    // ---- resident user fragments: lane holds k = 32 ks + 8 g + 0..7 of user lu of each block ----
    // The prologue of a grouped launch is a chain of dependent gathers (row ids -> user rows, floors, counters, biases).  Written
    // with the loads under `real ? ... : ...` the compiler made an exec-masked block per load, each ending in s_waitcnt
    // vmcnt(0): ~30 serial round trips per workgroup, 13 of the 19 us a refining workgroup spent before its first MFMA
    // (profiles/r04_refine_clocks.json; the row gathers themselves: 4 us).  Now: lane l of a wave owns local rows l, 64 + l, ...
    // -- ONE coalesced, unconditional read of their row ids, ONE batch of unconditional gathers of their list parameters (each
    // user once per wave, not once per k-group lane), the results through LDS / shuffles to the lanes that need them.
    constexpr int NRI = NUB * 16 / 64;
    int32_t ri[NRI];                                             // source row (of R and the per-user arrays), -1: padding / no user
#pragma unroll
    for (int h = 0; h < NRI; ++h) {
        const int local = h * 64 + lane;
        const int64_t row = r_base + local;
        const bool inb = row < p.n_r;
        int32_t idx = (int32_t)row;
        if (GRP) idx = p.row_index[inb ? row : p.n_r - 1];
        const bool pad = !inb || (fixed && wave * (NUB * 16) + local >= rows_here);
        ri[h] = (pad || idx < 0) ? -1 : idx;
    }
    if (LIST && !GRP) {
        // the dense listing launch over the HOT superblocks (trec_score_gemm_refine_candidates_hot): a user whose entry of this
        // superblock is -inf was listed for it by the pre-refining launch already (trec_topk_prerefine_tau marked it) -- it sits
        // this superblock out (no second copy of its candidates in its 128 slots, the saved maximum stays where it is)
        float mk[NRI];
#pragma unroll
        for (int h = 0; h < NRI; ++h) mk[h] = p.blockmax[(int64_t)chunk * p.bm_stride + (ri[h] >= 0 ? ri[h] : 0)];
#pragma unroll
        for (int h = 0; h < NRI; ++h) if (mk[h] == -INFINITY) ri[h] = -1;
    }
    bf16x8 rfb[NUB][KS];
    float thr_a[NUB];                                            // LIST: an accumulator below this cannot reach the user's floor
    int64_t frow[NUB];
    bool real[NUB];
#pragma unroll
    for (int ub = 0; ub < NUB; ++ub) {
        const int32_t s = __shfl(ri[ub >> 2], (ub & 3) * 16 + lu, 64);
        real[ub] = s >= 0;
        frow[ub] = s >= 0 ? s : 0;                               // padding rows compute on user 0 (never written)
        const char* src = (const char*)p.R + frow[ub] * (int64_t)RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rfb[ub][ks] = *(const bf16x8*)(src + (ks * 4 + g) * 16);
        thr_a[ub] = INFINITY;
#ifdef TREC_CAND_DIAG
        // (A/B: the row gathers throttled -- a wait after every block / every second block)
        if ((p.cand_diag & 64) || ((p.cand_diag & 128) && (ub & 1))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    if (LIST) {
        float flv[NRI], buv[NRI];
        int32_t cnv[NRI];
        const float* rbias = (BIAS && p.r_bias) ? p.r_bias : p.cand_floor;
#pragma unroll
        for (int h = 0; h < NRI; ++h) {
            const int64_t r = ri[h] >= 0 ? ri[h] : 0;
            flv[h] = p.cand_floor[r];
            cnv[h] = p.cand_n[r];
            buv[h] = rbias[r];
        }
#pragma unroll
        for (int h = 0; h < NRI; ++h) {
            const bool rl = ri[h] >= 0;
            float fl = rl ? flv[h] : INFINITY;
#ifdef TREC_CAND_DIAG
            if (!(p.cand_diag & 32))                             // (the cost of this gather)
#endif
            if (rl && cnv[h] > p.cand_cap) fl = INFINITY;        // the list is already incomplete: the user will be re-done
            const float bu = (BIAS && p.r_bias && rl) ? buv[h] : 0.f;
            // acc + bu >= fl (evaluated exactly when the queue is emptied) implies acc >= thr_a: fl - bu less three roundings
            const float t = fl - bu;
            float ta = t - (fabsf(fl) + fabsf(bu) + fabsf(t)) * 2.4e-7f;
            if (bu == 0.f) ta = fl;
            if (!(fl < INFINITY)) ta = INFINITY;                 // padding rows, users without a usable bound: nothing is listed
            else if (!(ta == ta)) ta = -INFINITY;                // (fl = -inf: everything is)
            q_fl[h * 64 + lane] = fl;
            q_bu[h * 64 + lane] = bu;
            q_id[h * 64 + lane] = ri[h];
            q_ta[h * 64 + lane] = ta;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ub = 0; ub < NUB; ++ub) thr_a[ub] = q_ta[ub * 16 + lu];
    }
    // the users this lane stores at superblock ends: blocks OW g .. OW g + OW - 1 (rows it already holds for block OW g + o).
    // The LIST form reads them (and their biases) from its LDS arrays when a superblock ends -- six registers it does not have.
    int64_t own_u[OW];
    float own_bias[OW];
#pragma unroll
    for (int o = 0; o < OW; ++o) { own_u[o] = -1; own_bias[o] = 0.f; }
    if (!LIST) {
#pragma unroll
        for (int o = 0; o < OW; ++o) {
            const int64_t r0 = real[o] ? frow[o] : -1, r1 = real[OW + o] ? frow[OW + o] : -1;
            const int64_t r2 = real[2 * OW + o] ? frow[2 * OW + o] : -1, r3 = real[3 * OW + o] ? frow[3 * OW + o] : -1;
            own_u[o] = g == 0 ? r0 : (g == 1 ? r1 : (g == 2 ? r2 : r3));
        }
        if (BIAS && p.r_bias) {
            float ob[OW];
#pragma unroll
            for (int o = 0; o < OW; ++o) {
                ob[o] = p.r_bias[own_u[o] >= 0 ? own_u[o] : 0];
                asm volatile("" : "+v"(ob[o]));                  // (keeps the load where it is: not sunk into the select below)
            }
#pragma unroll
            for (int o = 0; o < OW; ++o) own_bias[o] = own_u[o] >= 0 ? ob[o] : 0.f;
        }
    }

    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = lu * RB + (((ks * 4 + g) ^ swz(lu)) * 16);      // rows lu + 16 b: same swizzle

    f32x4 acc[NUB];
    float bm[NUB];
#pragma unroll
    for (int ub = 0; ub < NUB; ++ub) bm[ub] = -INFINITY;

    // LIST: empty this wave's queue.  Entry = the four scores of items item0 .. item0 + 3 of one user (code = user slot << 16 |
    // 4-item group of the superblock); a score that reaches the floor with the user bias added takes the next slot of the
    // user's list.  One lane, one entry: its (up to four) hits take their slots with independent atomics, then the 8-byte stores.
    int tile_now = 0;                                            // first 16-item block of the tile the step loop is in, counted inside its superblock
    int64_t sb_first = t_begin;                                  // first item of the superblock whose entries the queue holds
    auto queue_entry = [&](int i, int n) __attribute__((always_inline)) {
        const bool in = i < n;
        const int32_t code = in ? qc[i] : 0;
        const f32x4 a = in ? qv[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ul = code >> 16;
        const int32_t uid = in ? q_id[ul] : -1;
        const float fl = q_fl[ul], bu = q_bu[ul];
        const int64_t item0 = sb_first + (int64_t)(code & 0xffff) * 4;
        float v[4];
        int32_t slot[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = BIAS ? a[e] + bu : a[e];
            slot[e] = 0x7fffffff;
            if (uid >= 0 && v[e] >= fl && item0 + e < p.n_t) slot[e] = atomicAdd(p.cand_n + uid, 1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#ifdef TREC_CAND_DIAG
            if ((p.cand_diag & 3) == 2) continue;
#endif
            if (slot[e] < p.cand_cap)
                p.cand[(int64_t)uid * p.cand_cap + slot[e]] = make_int2((int32_t)(item0 + e) + p.t_index_base, __float_as_int(v[e]));
        }
    };
    auto queue_flush = [&](auto unrc) __attribute__((always_inline)) {
        constexpr int UNR = decltype(unrc)::value;              // entries per lane and round (their atomics are in flight together)
#ifdef TREC_CAND_DIAG                                          // diagnostics build only (the cost of the parts: DESIGN 5e)
        const int n = (p.cand_diag & 3) == 1 ? 0 : qn;
#else
        const int n = qn;
#endif
#pragma unroll 1
        for (int i0 = 0; i0 < n; i0 += 64 * UNR) {
#pragma unroll
            for (int j = 0; j < UNR; ++j) queue_entry(i0 + j * 64 + lane, n);
        }
        qn = 0;
    };

    auto tile_body = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const char* tb = smem + buf * TILE_BYTES;
        const float* sd = side + buf * BN + 4 * g;               // the block's item biases of result rows 4 g .. 4 g + 3
        bf16x8 tf[3];
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) c0 = *(const f32x4*)sd;
        tf[0] = *(const bf16x8*)(tb + koff[0]);
        tf[1] = *(const bf16x8*)(tb + koff[1 % KS]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int blk = s / KS, ks = s % KS;
            if (!RDL && s + 2 < NSTEP)
                tf[(s + 2) % 3] = *(const bf16x8*)(tb + ((s + 2) / KS) * 16 * RB + koff[(s + 2) % KS]);
            if (ks == 0) {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub)
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[s % 3], rfb[ub][0], c0, 0, 0, 0);
                if (RDL) __builtin_amdgcn_sched_barrier(0);
                if (BIAS && blk + 1 < NBLK) c0 = *(const f32x4*)(sd + 16 * (blk + 1));      // lands under this block's MFMAs
            } else {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub)
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[s % 3], rfb[ub][ks], acc[ub], 0, 0, 0);
                if (RDL) __builtin_amdgcn_sched_barrier(0);
            }
            if (RDL && s + 2 < NSTEP)
                tf[(s + 2) % 3] = *(const bf16x8*)(tb + ((s + 2) / KS) * 16 * RB + koff[(s + 2) % KS]);
            if (ks == KS - 1 && !LIST) {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub) {
                    bm[ub] = fmaxf(fmaxf(bm[ub], acc[ub][0]), acc[ub][1]);
                    bm[ub] = fmaxf(fmaxf(bm[ub], acc[ub][2]), acc[ub][3]);
                }
            }
            if (ks == KS - 1 && LIST) {
#pragma unroll
                for (int ub = 0; ub < NUB; ++ub) {
                    // (a wave of 128 users meets ~4 hits per 16-item block on Gaussian rows: the branch per accumulator set is taken
                    // for four of ten; ONE test per block step for all eight sets measured slower, 13.2 against 10.2 ms)
                    if ((ub % (NUB / 2)) == 0 && __builtin_expect(qn > LQ_FLUSH, 0)) queue_flush(std::integral_constant<int, 1>{});
                    const float m4 = fmaxf(fmaxf(fmaxf(acc[ub][0], acc[ub][1]), acc[ub][2]), acc[ub][3]);
                    bm[ub] = fmaxf(bm[ub], m4);
                    const bool hit = m4 >= thr_a[ub];
                    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                    if (hm != 0ull) {                            // wave-uniform
                        const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(hm >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned int)hm, 0u));
                        if (hit) {
                            // 4-item group of the superblock, from an opaque copy of the lane id: written as a loop invariant
                            // the compiler hoists (blk << 2) + g for every blk, spills the copies and re-loads them from
                            // scratch in every block step -- behind s_waitcnt vmcnt(0), i.e. behind the next tile's DMA
                            int lo = lane;
                            asm volatile("" : "+v"(lo));
                            qv[pos] = acc[ub];
                            qc[pos] = (((ub * 16) + (lo & 15)) << 16) | (((tile_now + blk) << 2) + (lo >> 4));
                        }
                        qn += __builtin_popcountll(hm);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (!GRP) stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef TREC_CAND_DIAG
    dg_t1 = wall_clock64();
#endif

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        if (LIST) tile_now = (t % p.sb_tiles) * NBLK;
        if (buf == 0) tile_body(std::integral_constant<int, 0>{});
        else tile_body(std::integral_constant<int, 1>{});

        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
#ifdef TREC_CAND_DIAG
            const unsigned long long dg_b = wall_clock64();      // (stamps only at superblock ends: per-tile stamps slowed the loop itself)
#endif
            if (LIST) { queue_flush(std::integral_constant<int, 3>{}); sb_first = t_begin + (int64_t)(t + 1) * BN; }
            // end of a superblock: the four row-groups' maxima of every user meet; row-group g stores blocks OW g .. OW g + OW - 1
            const int64_t sb = GRP ? (int64_t)chunk : t_begin / ((int64_t)p.sb_tiles * BN) + t / p.sb_tiles;
            float m[NUB];
#pragma unroll
            for (int ub = 0; ub < NUB; ++ub) {
                float x = bm[ub];
                x = fmaxf(x, __shfl_xor(x, 16, 64));
                m[ub] = fmaxf(x, __shfl_xor(x, 32, 64));
                bm[ub] = -INFINITY;
            }
            if (LIST) {
                // (the select over g below compiles to a scratch array indexed by g -- eight scratch stores, two loads and an
                // s_waitcnt vmcnt(0) that also waits for the queue's stores: the LIST form passes the maxima through the idle
                // threshold array in LDS instead)
                if (g == 0) {
#pragma unroll
                    for (int ub = 0; ub < NUB; ++ub) q_ta[ub * 16 + lu] = m[ub];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int o = 0; o < OW; ++o) {
                    const int slot = (OW * g + o) * 16 + lu;
                    float v = q_ta[slot];
                    const int32_t ou = q_id[slot];
                    if (BIAS) v = v + q_bu[slot];
#ifdef TREC_CAND_DIAG
                    if (p.cand_diag & 8) continue;               // (the cost of the scattered maxima stores)
#endif
                    if (GRP && p.pre_max) {
                        // the pre-refining launch: the maximum stays with the list (coalesced, read back by list position), the
                        // table entry is marked -inf -- the compaction no longer sees the pair -- in the same breath
                        if (ou >= 0) {
                            // (the wave's first list row from scalars: r_base itself, kept alive to here, cost the LIST form a spill)
                            const int64_t rb = ((int64_t)__builtin_amdgcn_readfirstlane(rblock) * 4 + wave) * (NUB * 16);
                            p.pre_max[rb + slot] = v;
                            p.blockmax[sb * p.bm_stride + ou] = -INFINITY;
                        }
                    } else if (ou >= 0) p.blockmax[sb * p.bm_stride + ou] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (dense LIST launches: the next superblock's maxima follow)
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int o = 0; o < OW; ++o) {
                    float v = g == 0 ? m[o] : (g == 1 ? m[OW + o] : (g == 2 ? m[2 * OW + o] : m[3 * OW + o]));
                    if (BIAS) v = v + own_bias[o];
                    if (own_u[o] >= 0) p.blockmax[sb * p.bm_stride + own_u[o]] = v;
                }
            }
#ifdef TREC_CAND_DIAG
            dg_flush += wall_clock64() - dg_b;
#endif
        }
        if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#ifdef TREC_CAND_DIAG
    if (GRP && LIST && tid == 0 && blockIdx.x < REFINE_DIAG_WGS) {
        // (plain stores into the workgroup's own row: atomics on eight shared counters -- a first version -- slowed the kernel 2x)
        const unsigned long long dg_e = wall_clock64();
        unsigned long long* row = g_refine_clk + (size_t)blockIdx.x * 4;
        row[0] = dg_t1 - dg_t0;
        row[1] = dg_e - dg_t1 - dg_flush;
        row[2] = dg_flush;
        row[3] = dg_e - dg_t0;
    }
#endif
}

template <int KT, bool BIAS, bool GRP, int NUB = 8, bool RDL = false, bool LIST = false>
int launch_bf16x16(ScoreParams p, hipStream_t st)
{
    constexpr int LDS = 2 * BN * KT * 2 + 2 * BN * 4 + (LIST ? 4 * LQ_CAP * 20 + 4 * 4 * NUB * 16 * 4 : 0);
    constexpr int RW = 4 * NUB * 16;
    auto kern = blockmax_bf16x16_kernel<KT, BIAS, GRP, NUB, RDL, LIST>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if (GRP && p.capacity > 0) p.capacity = p.capacity * 512 / RW;           // the caller counts a superblock's list in 512-row units
    p.n_rblocks = GRP ? (int)(p.n_r / RW) : (int)ceil_div64(p.n_r, RW);
    const unsigned blocks = GRP ? (p.wg_map ? (unsigned)p.n_wgs : (unsigned)p.n_rblocks) : (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    if (blocks == 0) return TREC_OK;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch(GRP ? "trec_score_gemm_blockmax_grouped (16x16x32)" : "trec_score_gemm_blockmax (16x16x32)");
}

// ---- the exact (fp32) form --------------------------------------------------------------------------------------------
// Same stage 1 on v_mfma_f32_32x32x2_f32 -- an exact k-ordered fmaf chain, bit-identical to oracle/tr_oracle.c -- with
// the reference's bias order (s + b_u) + b_i per element before the maximum.  An fp32 MFMA occupies the pipe for 64
// cycles, so the only thing that matters is that the wave never waits on LDS between them: the generic kernel issues
// `ds_read_b32; s_waitcnt lgkmcnt(0); MFMA` per k-step (42% of the fp32 MFMA peak); here the item values of the next
// group of 8 k-steps are read while the current group's 16 MFMAs run.  64 users per wave (128 VGPRs of resident
// fragments), 64-item tiles of 32 KB, 2 workgroups per CU.
template <int KT, bool BIAS>
__global__ __launch_bounds__(256, 2) void blockmax_pipe_f32_kernel(ScoreParams p)
{
    constexpr int NCB = 2;
    constexpr int RB = KT * 4;               // bytes per operand row
    constexpr int CH = RB / 16;              // 16-byte chunks per row (16 or 32)
    constexpr int KS = KT / 2;               // MFMA k-steps per block
    constexpr int G = 8;                     // k-steps per prefetch group
    constexpr int TILE_BYTES = BN * RB;
    constexpr int NSLOT = BN * CH / 256;
    static_assert(KT == 64 || KT == 128, "pipelined fp32 BLOCKMAX covers K = 64 / 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2][TILE_BYTES] item tiles | [2][BN] item biases
    float* side = (float*)(smem + 2 * TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int rblock = blockIdx.x % p.n_rblocks;
    const int chunk = blockIdx.x / p.n_rblocks;
    const int64_t r_base = ((int64_t)rblock * 4 + wave) * (NCB * 32);
    const int64_t t_begin = (int64_t)chunk * p.chunk_len;
    const int64_t t_end = (t_begin + p.chunk_len < p.n_t) ? t_begin + p.chunk_len : p.n_t;
    const int n_tiles = (int)((t_end - t_begin + BN - 1) / BN);

    float rff[NCB][KS];
    float r_bias[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        int64_t row = r_base + cb * 32 + l31;
        if (row >= p.n_r) row = p.n_r - 1;
        const float* src = (const float*)p.R + row * (int64_t)KT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rff[cb][ks] = src[2 * ks + half];
        r_bias[cb] = (BIAS && p.r_bias) ? p.r_bias[row] : 0.f;
    }

    int slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = i * 256 + tid;
        const int row = q / CH, pc = q % CH;
        slot_off[i] = row * RB + ((pc ^ (row & 15)) * 16);
    }
    const char* t_chunk = (const char*)p.T + t_begin * (int64_t)RB;
    float side_b = 0.f;
    auto stage_issue = [&](int tile, int buf) {
        const int64_t row0 = t_begin + (int64_t)tile * BN;
        const bool clamp = row0 + BN > p.n_t;
        if (BIAS && tid < BN) {
            int64_t g = row0 + tid;
            if (g >= p.n_t) g = p.n_t - 1;
            side_b = p.t_bias ? p.t_bias[g] : 0.f;
        }
        const char* tile_base = t_chunk + (int64_t)tile * (BN * RB);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            int off = slot_off[i];
            if (clamp) {
                const int last = (int)(p.n_t - 1 - row0);
                const int row = (i * 256 + tid) / CH;
                if (row > last) off -= (row - last) * RB;
            }
            char* dst = smem + buf * TILE_BYTES + (i * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto stage_commit = [&](int buf) {
        if (BIAS && tid < BN) side[buf * BN + tid] = side_b;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // k-step ks of "my" item row reads element k = 2 ks + half: chunk ks >> 1 (swizzled by row & 15), word 2 (ks & 1) + half
    const int sw = l31 & 15;
    const int row_off = l31 * RB + half * 4;
    auto tf_addr = [&](int ks) { return row_off + (((ks >> 1) ^ sw) * 16) + (ks & 1) * 8; };

    float bm[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bm[cb] = -INFINITY;

    stage_issue(0, 0);
    stage_commit(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        const char* tb = smem + buf * TILE_BYTES;
        const float* sd = side + buf * BN + 4 * half;
#pragma unroll 1
        for (int rb = 0; rb < BN / 32; ++rb) {
            const char* blk = tb + rb * 32 * RB;
            f32x16 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            float cur[G], nxt[G];
#pragma unroll
            for (int g = 0; g < G; ++g) cur[g] = *(const float*)(blk + tf_addr(g));
#pragma unroll                       // fully unrolled: the resident fragments must be addressed with constant indices
            for (int ks0 = 0; ks0 < KS; ks0 += G) {
                if (ks0 + G < KS) {
#pragma unroll
                    for (int g = 0; g < G; ++g) nxt[g] = *(const float*)(blk + tf_addr(ks0 + G + g));
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[g], rff[cb][ks0 + g], acc[cb], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) cur[g] = nxt[g];
            }
            // epilogue: (s + b_u) + b_i per element (tensorrec/recommendation_graphs.py:41), then the block maximum
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float m = bm[cb];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 tb4 = {0.f, 0.f, 0.f, 0.f};
                    if (BIAS) tb4 = *(const f32x4*)(sd + rb * 32 + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[cb][4 * q + e];
                        if (BIAS) v = (v + r_bias[cb]) + tb4[e];
                        m = fmaxf(m, v);
                    }
                }
                bm[cb] = m;
            }
        }
        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
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
}

template <int KT, bool BIAS>
int launch_f32(ScoreParams p, int sb_rows, hipStream_t st)
{
    constexpr int LDS = 2 * BN * KT * 4 + 2 * BN * 4;
    auto kern = blockmax_pipe_f32_kernel<KT, BIAS>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.n_rblocks = (int)ceil_div64(p.n_r, 4 * 2 * 32);
    p.sb_tiles = sb_rows / BN;
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm_blockmax (pipelined fp32)");
}

template <int KT, bool BIAS, int NCB, int WPS, bool OVL, int NBUF>
int launch_one_nbuf(ScoreParams p, hipStream_t st)
{
    constexpr int LDS = NBUF * BN * KT * 2 + NBUF * BN * 4;
    auto kern = blockmax_pipe_kernel<KT, BIAS, NCB, WPS, OVL, NBUF>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.n_rblocks = (int)ceil_div64(p.n_r, 4 * NCB * 32);
    const unsigned blocks = (unsigned)p.n_rblocks * (unsigned)p.n_chunks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm_blockmax (pipelined)");
}

// Two LDS buffers by default; "blockmax_nbuf" = 3 selects the triple-buffered form.  Measured on one box, 1M x 1M x 128,
// biased (profiles/r02_stage1_ab.txt): round-1 kernel 165.7 ms | 2 buffers with the bias row through global_load_lds
// (no staging register, no commit store) 164.6 ms | 3 buffers 166.7 ms -- a tile's loads already land within one tile of
// MFMAs; the third buffer only costs LDS.
template <int KT, bool BIAS, int NCB, int WPS, bool OVL = true>
int launch_one(ScoreParams p, hipStream_t st)
{
    if (trec_get_tuning("blockmax_nbuf", 2) == 3) return launch_one_nbuf<KT, BIAS, NCB, WPS, OVL, 3>(p, st);
    return launch_one_nbuf<KT, BIAS, NCB, WPS, OVL, 2>(p, st);
}

}  // namespace

// grouped form: p.n_r padded resident rows (a multiple of 512), p.rblock_chunk [n_r / 512], p.row_index [n_r],
// p.chunk_len = the superblock height
template <int KT, bool BIAS>
int launch_grouped(ScoreParams p, hipStream_t st)
{
    constexpr int LDS = 2 * BN * KT * 2 + 2 * BN * 4;
    auto kern = blockmax_pipe_kernel<KT, BIAS, 4, 2, false, 2, true>;
    static bool attr_set = false;
    if (!attr_set && LDS > 32 * 1024) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.n_rblocks = (int)(p.n_r / 512);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.n_rblocks), dim3(256), LDS, st, p);
    return trec_check_launch("trec_score_gemm_blockmax_grouped");
}

int launch_blockmax_pipelined_grouped(const ScoreParams& p, int kt, hipStream_t st)
{
    const bool bias = p.r_bias || p.t_bias;
    if (trec_get_tuning("blockmax_bf16_mfma16", 1) != 0) {       // the 16x16x32 form (filters only: see blockmax_bf16x16_kernel)
        // (the 64-users-per-wave A/B form has no LIST instance and decodes workgroup slots in its own units: it only serves the
        // plain fixed-capacity launch -- with candidate lists or a workgroup map the knob is ignored)
        if (kt == 128 && p.capacity > 0 && !p.cand && !p.wg_map && trec_get_tuning("cascade_grouped_nub", 8) == 4)
            return bias ? launch_bf16x16<128, true, true, 4>(p, st) : launch_bf16x16<128, false, true, 4>(p, st);
        if (p.cand) {                                             // the refining launch that also lists candidates (LIST)
            if (kt == 128) return bias ? launch_bf16x16<128, true, true, 8, false, true>(p, st) : launch_bf16x16<128, false, true, 8, false, true>(p, st);
            if (kt == 64) return bias ? launch_bf16x16<64, true, true, 8, false, true>(p, st) : launch_bf16x16<64, false, true, 8, false, true>(p, st);
        }
        if (kt == 128 && bias && trec_get_tuning("blockmax_bf16_rdlate", 0) != 0) return launch_bf16x16<128, true, true, 8, true>(p, st);
        if (kt == 128) return bias ? launch_bf16x16<128, true, true>(p, st) : launch_bf16x16<128, false, true>(p, st);
        if (kt == 64) return bias ? launch_bf16x16<64, true, true>(p, st) : launch_bf16x16<64, false, true>(p, st);
    }
    if (p.capacity > 0) {
        trec_set_last_error("trec_score_gemm_blockmax_grouped: the fixed-capacity layout needs the 16x16x32 form (tuning blockmax_bf16_mfma16 = 1)");
        return TREC_ERR_UNSUPPORTED;
    }
    if (kt == 128) return bias ? launch_grouped<128, true>(p, st) : launch_grouped<128, false>(p, st);
    if (kt == 64) return bias ? launch_grouped<64, true>(p, st) : launch_grouped<64, false>(p, st);
    return TREC_ERR_UNSUPPORTED;
}

// filter use (variant bit 5 of trec_score_gemm_blockmax): the 16x16x32 form, any summation order.  With p.rblock_chunk set
// (trec_score_gemm_blockmax_hot) chunk c of the launch is superblock p.rblock_chunk[c] (-1: idle), p.chunk_len = sb_rows.
int launch_blockmax_filter16(const ScoreParams& p, int kt, hipStream_t st)
{
    if (p.euclid || (trec_get_tuning("blockmax_bf16_mfma16", 1) == 0 && !p.rblock_chunk)) return TREC_ERR_UNSUPPORTED;
    const bool bias = p.r_bias || p.t_bias;
    if (p.cand) {
        if (kt == 128) return bias ? launch_bf16x16<128, true, false, 8, false, true>(p, st) : launch_bf16x16<128, false, false, 8, false, true>(p, st);
        if (kt == 64) return bias ? launch_bf16x16<64, true, false, 8, false, true>(p, st) : launch_bf16x16<64, false, false, 8, false, true>(p, st);
    }
    if (kt == 128 && bias && trec_get_tuning("blockmax_bf16_rdlate", 0) != 0) return launch_bf16x16<128, true, false, 8, true>(p, st);
    if (kt == 128) return bias ? launch_bf16x16<128, true, false>(p, st) : launch_bf16x16<128, false, false>(p, st);
    if (kt == 64) return bias ? launch_bf16x16<64, true, false>(p, st) : launch_bf16x16<64, false, false>(p, st);
    return TREC_ERR_UNSUPPORTED;
}

int launch_blockmax_pipelined(const ScoreParams& p, int kt, hipStream_t st)
{
    if (p.euclid) return TREC_ERR_UNSUPPORTED;
    const bool bias = p.r_bias || p.t_bias;
    // users per wave / accumulator sets / workgroups per CU, measured at 1M x 1M x 128, biased (profiles/r01_k2_ablation.txt):
    //   5 (default): 128 users, one set,  2/CU  1528 TF      3:  96 users, one set,  2/CU  1468 TF
    //   2:            64 users, two sets, 2/CU  1290-1340 TF  4: 128 users, two sets, 1/CU  1257 TF
    const int shape = trec_get_tuning("blockmax_shape", 5);
    if (kt == 128 && shape == 3) return bias ? launch_one<128, true, 3, 2, false>(p, st) : launch_one<128, false, 3, 2, false>(p, st);
    if (kt == 128 && shape == 5) return bias ? launch_one<128, true, 4, 2, false>(p, st) : launch_one<128, false, 4, 2, false>(p, st);
    if (kt == 128 && shape == 4) return bias ? launch_one<128, true, 4, 1>(p, st) : launch_one<128, false, 4, 1>(p, st);
    if (kt == 128) return bias ? launch_one<128, true, 2, 2>(p, st) : launch_one<128, false, 2, 2>(p, st);
    if (kt == 64 && shape == 2) return bias ? launch_one<64, true, 2, 3>(p, st) : launch_one<64, false, 2, 3>(p, st);
    if (kt == 64) return bias ? launch_one<64, true, 4, 2, false>(p, st) : launch_one<64, false, 4, 2, false>(p, st);
    return TREC_ERR_UNSUPPORTED;
}

// fp32 (exact) operands; sb_rows: the superblock height in item rows (a multiple of 128)
int launch_blockmax_pipelined_f32(const ScoreParams& p, int kt, int sb_rows, hipStream_t st)
{
    if (p.euclid || sb_rows % BN != 0) return TREC_ERR_UNSUPPORTED;
    const bool bias = p.r_bias || p.t_bias;
    if (kt == 128) return bias ? launch_f32<128, true>(p, sb_rows, st) : launch_f32<128, false>(p, sb_rows, st);
    if (kt == 64) return bias ? launch_f32<64, true>(p, sb_rows, st) : launch_f32<64, false>(p, sb_rows, st);
    return TREC_ERR_UNSUPPORTED;
}

#ifdef TREC_CAND_DIAG
// diagnostics build only (not in include/tensorrec_hip.h): read and optionally reset the refining launch's clock sums
extern "C" int trec_refine_diag_read(unsigned long long* out, long long n_words, int reset)
{
    if (n_words > (long long)REFINE_DIAG_WGS * 4) n_words = (long long)REFINE_DIAG_WGS * 4;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_refine_clk), sizeof(unsigned long long) * (size_t)n_words) != hipSuccess) return TREC_ERR_LAUNCH;
    (void)reset;
    return TREC_OK;
}
#endif
