// tensorrec_amd/csrc/wmrb_fused.hip -- K3 + K6 + the user side of their backward in ONE pass per user.
//
// A WMRB training step of the reference (tensorrec.py:384-395, :437-449, loss_graphs.py:153-227, then
// AdamOptimizer.minimize of the SUMMED loss, tensorrec.py:487-489) evaluates, per user u,
//     yhat_p  = <U_u, V_i(p)> + b_u + b_i(p)      for the user's interactions p           (serial predictions)
//     yhat_us = <U_u, V_i(u,s)> + b_u + b_i(u,s)  for the user's S sampled items          (sample predictions)
//     loss_p  = log(1 + smr_p),  smr_p = (I/S) * w_p * sum_s max(0, 1 - yhat_p + yhat_us)   (positive p only)
// and differentiates sum_p loss_p.  Unfused (pair_score.hip + loss.hip + K1 gathers) every item row V_i is gathered
// from HBM twice per step -- once for the prediction, once for dU_u = sum_j coef_j V_i(j) -- with the predictions,
// the loss terms and the coefficients round-tripping through HBM in between.  Here a workgroup owns a user: its
// S + n_u item rows are gathered ONCE, all at the same time, into the REGISTERS of the 8 subgroups (up to RMAX rows
// each -- one round of HBM latency per user instead of one per row batch, and only ~5 KB of LDS per workgroup, so 4-5
// workgroups per CU overlap their phases); predictions, loss and coefficients never leave the chip, and the user's
// gradient row is the coefficient-weighted sum of the rows still sitting in the registers.  What goes back to HBM is
// what the item side needs (one coefficient per pair, grouped by item later) plus the outputs: loss, serial
// predictions, dU, d b_u.
//
// The upstream gradient is the constant 1 (the trainer differentiates the SUM of the loss vector); that is what makes
// the coefficients computable in the same pass.  Per pair: one fmaf chain per lane over its columns as in
// pair_score_fwd_kernel, then DPP rotations instead of the xor-butterfly; hinge sums run sequentially over the samples
// (no cross-lane reduction at all) -- predictions, losses and gradients equal the unfused kernels' up to summation order.
#include "common.hpp"
#include <math.h>

namespace {

// sum over the 32 lanes of a subgroup with DPP adds (one VALU instruction each, no LDS crossbar): four rotations
// inside the 16-lane rows, then row_bcast15 carries row 0's total into row 1 -- the full sum is valid in lanes 16..31
// of the subgroup (read it from lane 31).  ds_bpermute-based __shfl_xor costs ~5x more issue slots and latency here.
__device__ __forceinline__ float dpp_add(float x, const int ctrl_tag)
{
    int v = __float_as_int(x), r;
    switch (ctrl_tag) {
        case 8: r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); break;    // row_ror:8
        case 4: r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); break;    // row_ror:4
        case 2: r = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); break;    // row_ror:2
        case 1: r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); break;    // row_ror:1
        default: r = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); break;   // row_bcast15 into rows 1, 3
    }
    return x + __int_as_float(r);
}

constexpr int SR = 4;                 // sample registers per lane in the loss phase: S <= 256

// A/B switches of the gather (variant libraries: -DTREC_WMRB_NT / -DTREC_WMRB_NO_EARLY).  Non-temporal row loads were MEASURED
// slower (13.5 ms against 11.4 for the whole kernel at 1M x 1M, profiles/r05_wmrb_fused_ab.txt): the rows are read once per pair,
// but `nt` loads of 16 B per lane did not stream past the L2 any cheaper -- plain loads stay the default.
#ifdef TREC_WMRB_NT
#define TREC_ROW_LOAD(p) __builtin_nontemporal_load(p)
#else
#define TREC_ROW_LOAD(p) (*(p))
#endif

// The ablation switches of scripts/bench_fused_ablate.py (histogram atomics / loss phases / dU off: WRONG results, only the
// time is read) exist only in a build with -DTREC_WMRB_ABLATE; the shipped kernel has none of the branches and no tuning
// lookup per launch (a knob left set would train wrong gradients silently).
#ifdef TREC_WMRB_ABLATE
#define TREC_ABLATED(bit) (ablate & (bit))
#else
#define TREC_ABLATED(bit) false
#endif

// ITERS: float4 chunks per lane of a 32-lane subgroup (d <= 128 * ITERS); RMAX: item rows a subgroup holds
// SURE: the first SURE register rows of every subgroup are sampled items for certain (8 * SURE <= S): their gathers leave as soon
// as the sample ids are there, without waiting for the indptr -> interaction-id chain the rows past S hang on
template <int ITERS, int RMAX, int SURE = 0>
__global__ __launch_bounds__(256, (ITERS == 1 && RMAX == 16) ? 4 : 1) void wmrb_user_fused_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ ub, const float* __restrict__ ib,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ xi, const int32_t* __restrict__ pos_slot,
    const float* __restrict__ pos_weight, const int32_t* __restrict__ samples, int64_t n_users, int32_t S, int d,
    float ratio, int32_t max_rows, float* __restrict__ loss, float* __restrict__ pred_serial, float* __restrict__ dU,
    float* __restrict__ dub, float* __restrict__ coef_samples, float* __restrict__ coef_pairs,
    int32_t* __restrict__ sample_hist, int32_t* __restrict__ sample_rank, int ablate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // y [max_rows4] | coef [max_rows4] | c [max_rows4] | base [max_rows4] | tmp [16 * max_rows4] | partial dU [8][d]
    // (max_rows4 = max_rows rounded up to 4 so that float4 reads of y stay inside the array and aligned)
    const int mr4 = (max_rows + 3) & ~3;
    float* l_y = lds;
    float* l_coef = l_y + mr4;
    float* l_c = l_coef + mr4;
    float* l_base = l_c + mr4;
    float* l_tmp = l_base + mr4;
    float* l_part = l_tmp + 16 * mr4;

    const int64_t u = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = tid & 31, sg = tid >> 5;             // 8 subgroups of 32 lanes
    // Every load of the gather phase is UNCONDITIONAL, from a clamped (always valid) address, and selected afterwards: written
    // as `cond ? load : 0` the compiler wraps each load in an exec-masked block that ends in s_waitcnt vmcnt(0) -- the 16 loads
    // of the interaction ids, the biases, the user row and the histogram atomic became ~10 SERIAL round trips per user where
    // three dependent ones are needed (ids -> rows / biases -> atomics; the refining launch of the top-k cascade had the same
    // disease).  Rows past R read item 0 and are never used unguarded (predictions not written, dU sum guarded).
    // The sampled item ids do not depend on the user's interaction range: their loads leave together with indptr's.
    int32_t item[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int j = sg + 8 * r;
        item[r] = samples[u * S + (j < S ? j : S - 1)];
    }
    const int64_t b = indptr[u], e = indptr[u + 1];
    const int n_pos = (int)(e - b);
    const int R = S + n_pos;                             // rows of this user: samples first, then interactions

    if (n_pos == 0) {
        // no interactions: no loss terms, every coefficient is 0 (the samples of this user are never used)
        for (int s = tid; s < S; s += 256) {
            coef_samples[u * S + s] = 0.f;
            if (sample_hist) {                                                   // the sort still sees these pairs
                const int32_t rk = atomicAdd(sample_hist + samples[u * S + s], 1);
                if (sample_rank) sample_rank[u * S + s] = rk;
            }
        }
        for (int c = tid; c < d; c += 256) dU[u * d + c] = 0.f;
        if (dub && tid == 0) dub[u] = 0.f;
        return;
    }

    // interaction `tid` of this user (n_pos <= 256): its slot and weight are fetched now, used in phase (c2)
    const int qi = tid < n_pos ? tid : n_pos - 1;
    const int32_t slot_ld = pos_slot[b + qi];
    const float w_ld = pos_weight ? pos_weight[b + qi] : 1.f;     // (uniform branch)
    const int32_t my_slot = (tid < n_pos) ? slot_ld : -1;
    const float my_w = (tid < n_pos) ? w_ld : 1.f;

    // ---- (a) the user's row, in the registers of every subgroup ----
    f32x4 x[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 32 + sub) * 4;
        const f32x4 v = TREC_ROW_LOAD((const f32x4*)(U + u * d + (c < d ? c : 0)));
        x[it] = (c < d) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float bu = ub ? ub[u] : 0.f;

    // ---- (b) every subgroup gathers its rows j = sg + 8 r (r < RMAX) in ONE batch and keeps them ----
    f32x4 y[RMAX][ITERS];
    // the rows that are samples for certain first: their addresses need nothing but the sample ids
#pragma unroll
    for (int r = 0; r < SURE; ++r) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 32 + sub) * 4;
            y[r][it] = TREC_ROW_LOAD((const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0)));
        }
    }
    {
        int32_t xid[RMAX];                               // the interaction item of row j >= S (n_pos >= 1 here)
#pragma unroll
        for (int r = SURE; r < RMAX; ++r) {
            const int q = sg + 8 * r - S;
            xid[r] = xi[b + (q < 0 ? 0 : (q < n_pos ? q : n_pos - 1))];
        }
#pragma unroll
        for (int r = SURE; r < RMAX; ++r) {
            const int j = sg + 8 * r;
            item[r] = (j < S) ? item[r] : ((j < R) ? xid[r] : 0);
        }
    }
    // Lane 16 + (r & 15) of a subgroup OWNS the subgroup's row r (lanes 16..31 are where the DPP reduction leaves the
    // 32-lane dot product): it fetches the row's item bias -- in the same batch of loads as the rows --, issues the row's
    // histogram atomic and writes its prediction; none of that costs RMAX registers in every lane.
    constexpr int H = RMAX / 16;
    const int own = sub - 16;
    float my_bi[H];
    int32_t my_item[H];                                  // the item of the owned row: picked from the ids every lane holds
    // (16 v_cndmask per owned row instead of two more loads: written as loads -- samples / xi again at the owner's index -- the
    // compiler sank the xi one into an exec-masked block ending in s_waitcnt vmcnt(0), one more dependent round trip in front of
    // the bias load and, with the early rows above, a wait for ALL of them before the last rows could leave)
#pragma unroll
    for (int h = 0; h < H; ++h) {
        int32_t mine = 0;
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) mine = (own == r16) ? item[r16 + 16 * h] : mine;
        my_item[h] = mine;                               // (rows past R hold item 0: a valid address, never used unguarded)
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int j = sg + 8 * (own + 16 * h);
        const float v = ib ? ib[my_item[h]] : 0.f;           // (uniform branch)
        my_bi[h] = (own >= 0 && j < R) ? v : 0.f;
    }
#pragma unroll
    for (int r = SURE; r < RMAX; ++r) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 32 + sub) * 4;
            y[r][it] = TREC_ROW_LOAD((const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0)));
        }
    }
    if (d < ITERS * 128) {                               // (uniform: columns past d contribute zeros)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if ((it * 32 + sub) * 4 >= d) y[r][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // predictions: no per-row branch (a wave holds two subgroups with different rows), so the RMAX reduction chains
    // interleave; rows past R are zeros and are simply not written
    float dot[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            acc = fmaf(x[it].x, y[r][it].x, acc); acc = fmaf(x[it].y, y[r][it].y, acc);
            acc = fmaf(x[it].z, y[r][it].z, acc); acc = fmaf(x[it].w, y[r][it].w, acc);
        }
        dot[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        float t = dot[r];
        t = dpp_add(t, 8); t = dpp_add(t, 4); t = dpp_add(t, 2); t = dpp_add(t, 1);
        dot[r] = dpp_add(t, 0);                            // lanes 16..31 of the subgroup: the 32-lane total
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float mine = 0.f;
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) mine = (own == r16) ? dot[r16 + 16 * h] : mine;
        const int j = sg + 8 * (own + 16 * h);
        if (own >= 0 && j < R) {
            float s = mine;
            if (ub) s = s + bu;
            if (ib) s = s + my_bi[h];
            l_y[j] = s;
            if (j >= S) pred_serial[b + (j - S)] = s;
        }
    }
    // histogram of the counting sort to come, by the owner lanes: ONE atomic instruction per subgroup and 16 rows instead
    // of RMAX from lane 0.  The value an atomic returns is the pair's rank inside its item's bucket, which makes the sort's
    // fill pass atomic-free; it is stored at the very end of the kernel.  Issued AFTER the rows have been consumed: returning
    // atomics and loads come back in order, so an atomic issued between the row loads and their first use is waited for first
    // (rk stays UNDEFINED for lanes that issue none: merging it with a constant would put the wait right here).
    int32_t rk[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int j = sg + 8 * (own + 16 * h);
        if (sample_hist && own >= 0 && j < S && !TREC_ABLATED(1))
            rk[h] = __hip_atomic_fetch_add(sample_hist + my_item[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    // ---- (c) WMRB terms and coefficients (upstream gradient = 1), without a single cross-lane operation:
    //      first threads = interactions, each walking the S sample predictions (LDS broadcast reads) for its hinge sum
    //      and active count; then threads = samples, each walking the interactions for its coefficient ----
    // (c1) hinge sum and active count of interaction q, 8 threads per interaction, each over every 8th float4 of samples
    for (int q0 = 0; q0 < (TREC_ABLATED(2) ? 0 : n_pos); q0 += 32) {
        const int q = q0 + (tid >> 3), k = tid & 7;
        float acc = 0.f;
        int cnt = 0;
        if (q < n_pos) {
            const float base = 1.0f - l_y[S + q];
            for (int s4 = k * 4; s4 < S; s4 += 32) {
                const f32x4 v = *(const f32x4*)(l_y + s4);            // l_y is padded: entries >= S are masked below
                const float t0 = base + v.x, t1 = base + v.y, t2 = base + v.z, t3 = base + v.w;
                const bool m1 = s4 + 1 < S, m2 = s4 + 2 < S, m3 = s4 + 3 < S;
                acc += fmaxf(t0, 0.f); cnt += (t0 >= 0.f) ? 1 : 0;
                if (m1) { acc += fmaxf(t1, 0.f); cnt += (t1 >= 0.f) ? 1 : 0; }
                if (m2) { acc += fmaxf(t2, 0.f); cnt += (t2 >= 0.f) ? 1 : 0; }
                if (m3) { acc += fmaxf(t3, 0.f); cnt += (t3 >= 0.f) ? 1 : 0; }
            }
            l_tmp[q * 16 + k] = acc;
            l_tmp[q * 16 + 8 + k] = (float)cnt;
        }
    }
    __syncthreads();
    // (c2) one thread per interaction: combine the 8 partials in fixed order, loss term and coefficients
    if (tid < n_pos) {
        const int q = tid;
        float c = 0.f, dp = 0.f;
        const float base = 1.0f - l_y[S + q];
        if (my_slot >= 0) {
            float acc = 0.f, cnt = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc += l_tmp[q * 16 + k]; cnt += l_tmp[q * 16 + 8 + k]; }
            float smr = ratio * acc;
            if (pos_weight) smr = smr * my_w;
            c = ratio / (1.0f + smr);                               // d loss_p / d (hinge sum), go = 1
            if (pos_weight) c = c * my_w;
            loss[my_slot] = logf(smr + 1.0f);
            dp = -c * cnt;
        }
        l_c[q] = c;
        l_base[q] = base;
        l_coef[S + q] = dp;
        coef_pairs[b + q] = dp;
    }
    __syncthreads();
    // (c3) one thread per sample: its coefficient over the user's interactions
    for (int s = tid; s < S; s += 256) {
        const float ys = l_y[s];
        float g = 0.f;
        for (int q = 0; q < (TREC_ABLATED(2) ? 0 : n_pos); ++q) g += (l_base[q] + ys >= 0.f) ? l_c[q] : 0.f;     // l_c = 0 for non-positives
        l_coef[s] = g;
        coef_samples[u * S + s] = g;
    }
    __syncthreads();

    // ---- (d) dU_u = sum_j coef_j * V_row_j: every subgroup sums ITS rows from registers, 8 partials meet in LDS ----
    {
        f32x4 part[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) part[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int j = sg + 8 * r;
            if (j < R && !TREC_ABLATED(4)) {
                const float cf = l_coef[j];
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    part[it].x = fmaf(cf, y[r][it].x, part[it].x); part[it].y = fmaf(cf, y[r][it].y, part[it].y);
                    part[it].z = fmaf(cf, y[r][it].z, part[it].z); part[it].w = fmaf(cf, y[r][it].w, part[it].w);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 32 + sub) * 4;
            if (c < d) *(f32x4*)(l_part + sg * d + c) = part[it];
        }
    }
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float acc = l_part[c];
#pragma unroll
        for (int g8 = 1; g8 < 8; ++g8) acc += l_part[g8 * d + c];
        dU[u * d + c] = acc;
    }
    if (dub && wave == 0) {
        float acc = 0.f;
        for (int j = lane; j < R; j += 64) acc += l_coef[j];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) dub[u] = acc;
    }
    if (sample_hist && sample_rank && own >= 0 && !TREC_ABLATED(1)) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int j = sg + 8 * (own + 16 * h);
            if (j < S) sample_rank[u * S + j] = rk[h];
        }
    }
}

static int rows_capacity(int d) { return d <= 128 ? 256 : 128; }      // 8 subgroups x RMAX rows

}  // namespace

// Dynamic LDS the fused step needs, or -1 if the configuration is not covered (the host then runs the unfused
// kernels): n_sampled <= 256, d % 4 == 0, d <= 256, and n_sampled + the longest interaction row <= 256 (128 for d > 128)
// -- the rows of a user live in the registers of its workgroup.
extern "C" int trec_wmrb_fused_lds_bytes(int32_t n_sampled, int32_t max_interactions_per_user, int32_t d)
{
    if (n_sampled < 1 || n_sampled > 64 * SR || d < 4 || d % 4 != 0 || d > 256 || max_interactions_per_user < 0) return -1;
    const int64_t rows = (int64_t)n_sampled + max_interactions_per_user;
    if (rows > rows_capacity(d)) return -1;
    const int64_t r4 = (rows + 3) & ~(int64_t)3;
    return (int)((20 * r4 + 8 * (int64_t)d) * 4);
}

extern "C" int trec_wmrb_fused_step(const float* U, const float* V, const float* user_bias, const float* item_bias,
                                    const int64_t* indptr, const int32_t* x_item, const int32_t* pos_slot,
                                    const float* pos_weight, const int32_t* samples, int64_t n_users, int64_t n_items,
                                    int32_t n_sampled, int32_t d, int32_t max_interactions_per_user, float* loss,
                                    float* pred_serial, float* dU, float* d_user_bias, float* coef_samples,
                                    float* coef_pairs, int32_t* sample_hist, int32_t* sample_rank, void* stream)
{
    TREC_REQUIRE(U && V && indptr && samples && loss && pred_serial && dU && coef_samples && coef_pairs,
                 "trec_wmrb_fused_step: null pointer");
    TREC_REQUIRE(!user_bias == !d_user_bias, "trec_wmrb_fused_step: user_bias and d_user_bias go together");
    TREC_REQUIRE(sample_hist || !sample_rank, "trec_wmrb_fused_step: sample_rank needs sample_hist");
    const int lds = trec_wmrb_fused_lds_bytes(n_sampled, max_interactions_per_user, d);
    if (lds < 0) {
        trec_set_last_error("trec_wmrb_fused_step: configuration not covered (see trec_wmrb_fused_lds_bytes)");
        return TREC_ERR_UNSUPPORTED;
    }
    if (n_users == 0) return TREC_OK;
    TREC_REQUIRE(max_interactions_per_user == 0 || (x_item && pos_slot), "trec_wmrb_fused_step: null interaction arrays");
    const float ratio = (float)n_items / (float)n_sampled;
    const int32_t max_rows = n_sampled + max_interactions_per_user;
    hipStream_t st = (hipStream_t)stream;
#ifdef TREC_WMRB_ABLATE
    const int ablate = trec_get_tuning("wmrb_ablate", 0);
#else
    const int ablate = 0;
#endif
#define TREC_FUSED(IT, RM, SU)                                                                                             \
    hipLaunchKernelGGL((wmrb_user_fused_kernel<IT, RM, SU>), dim3((unsigned)n_users), dim3(256), lds, st, U, V, user_bias,  \
                       item_bias, indptr, x_item, pos_slot, pos_weight, samples, n_users, n_sampled, d, ratio, max_rows, \
                       loss, pred_serial, dU, d_user_bias, coef_samples, coef_pairs, sample_hist, sample_rank, ablate)
#ifdef TREC_WMRB_NO_EARLY
    const int sure = 0;
#else
    const int sure = n_sampled / 8;                      // register rows r < sure hold samples in every subgroup (8 r + 7 < S)
#endif
    if (d <= 128 && max_rows <= 128) {
        if (sure >= 12) TREC_FUSED(1, 16, 12);
        else if (sure >= 8) TREC_FUSED(1, 16, 8);
        else if (sure >= 4) TREC_FUSED(1, 16, 4);
        else TREC_FUSED(1, 16, 0);
    } else if (d <= 128) {
        if (sure >= 16) TREC_FUSED(1, 32, 16);
        else TREC_FUSED(1, 32, 0);
    } else {
        if (sure >= 8) TREC_FUSED(2, 16, 8);
        else TREC_FUSED(2, 16, 0);
    }
#undef TREC_FUSED
    return trec_check_launch("trec_wmrb_fused_step");
}
