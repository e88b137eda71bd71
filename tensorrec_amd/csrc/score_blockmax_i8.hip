// tensorrec_amd/csrc/score_blockmax_i8.hip -- K2q: stage 0 of the cascaded exact top-k, the int8 pre-filter.
//
// The bf16 stage-1 kernel (score_blockmax.hip) sits at the chip's power limit (0.61 of the bf16 MFMA peak), so the way to
// make the exact top-k faster is to do FEWER bf16 flops.  v_mfma_i32_32x32x32_i8 contracts twice as many elements per
// cycle as the bf16 MFMA and its arithmetic is EXACT (int8 x int8 products, int32 accumulation: no rounding at all).
// Operand rows are quantised to int8 with ONE scale per side (trec_score_prep_i8: q = clamp(rint(x / scale), +-127)),
// so that for a user u   score(u, i) ~ a b (sum_k q_u[k] q_i[k] + bq_i) + b_u   and the maximum over the items of a
// superblock is an INTEGER maximum of the raw accumulators: the same one-v_max3-per-MFMA epilogue as the bf16 kernel, with
// the item bias (in integer units of a b) as the initial accumulator.  The quantisation error of every row is measured,
// not assumed (|x - a q| per row, clipping included), which gives a proven bound eps8_u >= |int8 score - fp32 score|
// exactly like the bf16 filter's (csrc/topk_filter.hip); superblocks whose int8 maximum is below
// (k-th largest int8 maximum) - 2 eps8_u cannot hold a top-k item and never reach the bf16 stage.  At 1M x 1M, d = 128,
// normalised rows: eps8 ~ 0.018, ~3% of the superblocks survive.
//
// Replaces (as a filter in front of them) tf.matmul of tensorrec/prediction_graphs.py:49-50 + the first tf.nn.top_k of
// tensorrec/recommendation_graphs.py:80; nothing it computes is returned to the caller -- survivors are re-scored in bf16
// (bounded again) and finally in fp32, bit-identical to the oracle.
//
// Kernel plan (K = 128 bytes per row): a wave owns NCB x 32 users whose int8 fragments stay in registers (NCB x 4 k-steps x
// 4 VGPRs); item tiles of 128 rows x 128 B = 16 KB are double-buffered in LDS through global_load_lds with the 16-byte-chunk
// XOR swizzle of score_gemm.hip; a 32-item block is 4 k-steps of NCB MFMAs fed by ONE ds_read_b128 each; the block's
// epilogue (8 v_max3_i32 per accumulator) runs right after its last k-step, the next block's bias row is read straight
// into accumulator 0.  Lane & 31 is the user (acc = mfma(items, users)), exactly the orientation of the bf16 kernel.
#include "score_common.hpp"
#include <math.h>
#include <limits.h>
#include <type_traits>

namespace {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int BNQ = 128;        // item rows per tile (four 32-row MFMA blocks)

template <int KT, bool BIAS, int NCB, int WPS>
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
    const int* t_bias_q = (const int*)p.t_bias;                   // integer item biases (units of the scale product)
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

    const float scale = p.scales[2];             // a b: integer score units -> float
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage_issue(t + 1, buf ^ 1);
        if (buf == 0) tile_body(std::integral_constant<int, 0>{});
        else tile_body(std::integral_constant<int, 1>{});

        if (((t + 1) % p.sb_tiles) == 0 || t + 1 == n_tiles) {
            // end of a superblock: combine the two half-wave maxima of each user, convert, add the user bias, store, reset
            const int64_t sb = t_begin / ((int64_t)p.sb_tiles * BNQ) + t / p.sb_tiles;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int o = __shfl_xor(bm[cb], 32, 64);
                const int m = bm[cb] > o ? bm[cb] : o;
                float v = (float)m * scale;                   // |m| < 2^24: the conversion is exact
                if (BIAS) v = v + r_bias[cb];
                const int64_t u = r_base + cb * 32 + l31;
                if (half == 0 && u < p.n_r) p.blockmax[sb * p.bm_stride + u] = v;
                bm[cb] = INT_MIN;
            }
        }
        if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <int KT, bool BIAS, int NCB, int WPS>
int launch_i8(ScoreParams p, int sb_rows, hipStream_t st)
{
    constexpr int LDS = 2 * BNQ * KT + 2 * BNQ * 4;
    auto kern = blockmax_i8_kernel<KT, BIAS, NCB, WPS>;
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

// fp32 rows -> int8 rows [n, kpad] (zero padded) with ONE scale for the whole side, + what the bound needs per row:
// {||x||, ||x - scale q||} (the actual quantisation error of this row, clipping included); items also get their bias in
// integer units of scale_prod and the maxima over rows of ||x|| + ||dx||, ||dx||, |bias|, |bias - scale_prod bq|.
template <int G>
__global__ __launch_bounds__(256) void prep_i8_kernel(const float* __restrict__ x, int64_t n, int d, int kt,
                                                     const float* __restrict__ scale_ptr, signed char* __restrict__ out_q,
                                                     float2* __restrict__ row_stats, float* __restrict__ gstats)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool ok = row < n;
    const int sub = threadIdx.x % G;
    const float scale = *scale_ptr;
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
            if (!(q == q)) q = 0.f;                                     // NaN input: the error norm below turns NaN and flags the user
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
    if (gstats) {
        float g0 = (sub == 0 && ok) ? nw + ne : 0.f, g1 = (sub == 0 && ok) ? ne : 0.f;
        if (g0 != g0) g0 = INFINITY;
        if (g1 != g1) g1 = INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            g0 = fmaxf(g0, __shfl_xor(g0, off, 64)); g1 = fmaxf(g1, __shfl_xor(g1, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const float g[2] = {g0, g1};
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (__float_as_uint(g[e]) > *(volatile unsigned int*)(gstats + e)) atomicMax((unsigned int*)(gstats + e), __float_as_uint(g[e]));
        }
    }
}

__global__ void scale_prod_kernel(float* __restrict__ scales) { scales[2] = scales[0] * scales[1]; }

// item biases in integer units of the scale product (scales[0] * scales[1], also written to scales[2]):
// bias_q = rint(bias / product), gstats[2] = max |bias|, gstats[3] = max |bias - product * bias_q|
__global__ __launch_bounds__(256) void bias_i8_kernel(const float* __restrict__ bias, int64_t n, float* __restrict__ scales,
                                                     int* __restrict__ bias_q, float* __restrict__ gstats)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float sp = scales[0] * scales[1];
    if (i == 0) scales[2] = sp;
    float g2 = 0.f, g3 = 0.f;
    if (i < n) {
        const float b = bias[i];
        float bq = rintf(b / sp);
        bq = fminf(fmaxf(bq, -4194304.f), 4194304.f);                   // |bq| <= 2^22: accumulators stay below 2^24
        if (!(bq == bq)) bq = 0.f;
        bias_q[i] = (int)bq;
        g2 = fabsf(b);
        g3 = fabsf(b - bq * sp);
        if (g2 != g2) g2 = INFINITY;
        if (g3 != g3) g3 = INFINITY;
    }
    for (int off = 32; off > 0; off >>= 1) { g2 = fmaxf(g2, __shfl_xor(g2, off, 64)); g3 = fmaxf(g3, __shfl_xor(g3, off, 64)); }
    if ((threadIdx.x & 63) == 0) {
        if (__float_as_uint(g2) > *(volatile unsigned int*)(gstats + 2)) atomicMax((unsigned int*)(gstats + 2), __float_as_uint(g2));
        if (__float_as_uint(g3) > *(volatile unsigned int*)(gstats + 3)) atomicMax((unsigned int*)(gstats + 3), __float_as_uint(g3));
    }
}

// sum of squares (double) and maximum magnitude (float bits) of a [n, d] matrix into ws[0] / the low word of ws[1]
// (zero-initialised by the caller): the scale of a side is derived from them
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

// scales[side] = min(clip_sigmas * rms, max |x|) / 127 (clip_sigmas <= 0: max |x| / 127, nothing clips); side 1 also
// writes scales[2] = scales[0] * scales[1]
__global__ void scale_from_stats_kernel(const double* __restrict__ ws, double n_elem, float clip_sigmas, int side,
                                        float* __restrict__ scales)
{
    const double rms = sqrt(ws[0] / n_elem);
    const float am = __uint_as_float(*(const unsigned int*)(ws + 1));
    float top = am;
    if (clip_sigmas > 0.f && (float)(clip_sigmas * rms) < top) top = (float)(clip_sigmas * rms);
    float s = top / 127.0f;
    if (!(s > 0.f) || !(s < INFINITY)) s = 1.0f;                        // all-zero or non-finite input: any scale is as good
    scales[side] = s;
    if (side == 1) scales[2] = scales[0] * s;
}

}  // namespace

// int8 operands for the pre-filter.  scales: float[3] on the device = {user-side scale, item-side scale, their product}.
// side 0: users (scales[0]); side 1: items (scales[1], gstats[0..1]) and -- with a bias -- what side 2 does; side 2: no
// quantisation (repr / out_q / row_stats unused): scales[2] and the item biases in units of the CURRENT product, gstats[2..3]
// (the item rows are quantised once; a new batch of users with its own scale only needs side 2 again).
// workspace: 16 bytes (zeroed here).  row_stats [n][2]; gstats (items) float[4], zero-initialised by the caller.
extern "C" int trec_score_prep_i8(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t side, float clip_sigmas,
                                  const float* bias, float* scales, double* workspace, void* out_q, float* row_stats,
                                  int32_t* bias_q, float* gstats, void* stream)
{
    TREC_REQUIRE(scales && (side == 0 || side == 1 || side == 2), "trec_score_prep_i8: side must be 0 (users), 1 (items) or 2 (item biases)");
    TREC_REQUIRE(!bias || (bias_q && gstats && side >= 1), "trec_score_prep_i8: a bias needs bias_q, gstats and side 1 / 2");
    if (n == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (side != 2) {
        TREC_REQUIRE(repr && workspace && out_q && row_stats, "trec_score_prep_i8: null pointer");
        TREC_REQUIRE(d >= 1 && kpad >= d && kpad % 4 == 0 && kpad <= 128, "trec_score_prep_i8: need d <= kpad <= 128, kpad % 4 == 0");
        TREC_REQUIRE(side == 0 || gstats, "trec_score_prep_i8: the item side needs gstats");
        TREC_REQUIRE(((uintptr_t)repr % 16) == 0, "trec_score_prep_i8: repr must be 16-byte aligned");
        if (hipMemsetAsync(workspace, 0, 2 * sizeof(double), st) != hipSuccess) {
            trec_set_last_error("trec_score_prep_i8: memset failed");
            return TREC_ERR_LAUNCH;
        }
        const int64_t n_elem = n * (int64_t)d;
        unsigned sb = (unsigned)ceil_div64(n_elem, 1024 * 8);
        if (sb > 4096) sb = 4096;
        if (sb < 1) sb = 1;
        hipLaunchKernelGGL(sumsq_absmax_kernel, dim3(sb), dim3(256), 0, st, repr, n_elem, workspace);
        hipLaunchKernelGGL(scale_from_stats_kernel, dim3(1), dim3(1), 0, st, workspace, (double)n_elem, clip_sigmas, side, scales);
        const int g = kpad >= 128 ? 32 : kpad / 4;
        const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
#define TREC_PQ(GV) hipLaunchKernelGGL(prep_i8_kernel<GV>, dim3(blocks), dim3(256), 0, st, repr, n, d, kpad, scales + side, (signed char*)out_q, (float2*)row_stats, side == 1 ? gstats : (float*)nullptr)
        if (g == 32) TREC_PQ(32);
        else if (g == 16) TREC_PQ(16);
        else TREC_PQ(8);
#undef TREC_PQ
    }
    if (side >= 1 && bias)
        hipLaunchKernelGGL(bias_i8_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, bias, n, scales, bias_q, gstats);
    else if (side == 2)
        hipLaunchKernelGGL(scale_prod_kernel, dim3(1), dim3(1), 0, st, scales);
    return trec_check_launch("trec_score_prep_i8");
}

// blockmax[s * bm_stride + u] = scale_prod * max over the items of superblock s of (sum_k q_u q_i + bq_i) + user_bias[u]
// (users_q / items_q: int8 [n, kpad]; item_bias_q: int32 [n_items] or NULL; scales: the device array of trec_score_prep_i8)
extern "C" int trec_score_gemm_blockmax_i8(const void* users_q, const void* items_q, int32_t kpad, int64_t n_users,
                                           int64_t n_items, const float* user_bias, const int32_t* item_bias_q,
                                           const float* scales, int32_t sb_rows, int32_t n_chunks, float* blockmax,
                                           int64_t bm_stride, void* stream)
{
    TREC_REQUIRE(users_q && items_q && scales && blockmax && bm_stride >= n_users, "trec_score_gemm_blockmax_i8: bad arguments");
    TREC_REQUIRE(kpad == 64 || kpad == 128, "trec_score_gemm_blockmax_i8: kpad must be 64 or 128");
    TREC_REQUIRE(sb_rows >= BNQ && sb_rows % BNQ == 0, "trec_score_gemm_blockmax_i8: sb_rows must be a multiple of 128");
    TREC_REQUIRE(n_users >= 1 && n_items >= 1 && n_chunks >= 1, "trec_score_gemm_blockmax_i8: empty operand");
    TREC_REQUIRE(n_items < (int64_t)1 << 31 && n_users < (int64_t)1 << 31, "trec_score_gemm_blockmax_i8: sizes must fit int32");
    ScoreParams p = {};
    p.R = users_q; p.T = items_q; p.n_r = n_users; p.n_t = n_items;
    p.chunk_len = ceil_div64(ceil_div64(n_items, n_chunks), sb_rows) * sb_rows;        // chunks are whole superblocks
    p.n_chunks = (int)ceil_div64(n_items, p.chunk_len);
    p.r_bias = user_bias; p.t_bias = (const float*)item_bias_q;
    p.blockmax = blockmax; p.bm_stride = bm_stride;
    p.scales = scales;
    hipStream_t st = (hipStream_t)stream;
    const bool bias = user_bias || item_bias_q;
    const int shape = trec_get_tuning("blockmax_i8_shape", 0);
    if (kpad == 128) {
        if (shape == 1) return bias ? launch_i8<128, true, 4, 3>(p, sb_rows, st) : launch_i8<128, false, 4, 3>(p, sb_rows, st);
        if (shape == 2) return bias ? launch_i8<128, true, 2, 3>(p, sb_rows, st) : launch_i8<128, false, 2, 3>(p, sb_rows, st);
        return bias ? launch_i8<128, true, 4, 2>(p, sb_rows, st) : launch_i8<128, false, 4, 2>(p, sb_rows, st);
    }
    return bias ? launch_i8<64, true, 4, 2>(p, sb_rows, st) : launch_i8<64, false, 4, 2>(p, sb_rows, st);
}
