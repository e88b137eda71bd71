// tensorrec_amd/csrc/wmrb_tiled_body.hpp -- the per-user body of the tiled one-pass WMRB step, shared by the stand-alone kernel
// (csrc/wmrb_tiled.hip: one workgroup per user) and by the single-kernel training step of small models (csrc/step_coop.hip: a
// persistent workgroup walks several users and generates their samples itself).  See wmrb_tiled.hip for the design.
#pragma once
#include "common.hpp"
#include <math.h>

namespace {


#define EUCLID_EPS 1e-16f

// sum over the 32 lanes of a subgroup (DPP adds, as in wmrb_fused.hip): total valid in lanes 16..31
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

struct TiledOut {
    float* loss; float* pred_serial; float* dU; float* dub;
    float* val_samples; float* val_pairs;      // what the item side sums per pair (g, or c = -g / sqrt(D))
    float* raw_samples; float* raw_pairs;      // g itself (euclidean + item biases), or null
    float* dense_g; int64_t ldg;               // zeroed [n_users, ldg]: += value at (user, item), or null
    float* val_rowsum;                         // [n_users] sum of the user's values, written when dU is null (no pass 2: dU from G . V)
};

// One user's step.  lds: the workgroup's dynamic LDS (trec_wmrb_tiled_lds_bytes); samp: the user's S sampled item ids (global or
// LDS memory).  All 256 threads of the workgroup call it together; it ends with a barrier, so the caller may reuse the LDS at once.
// ITERS: float4 chunks per lane of an LPR-lane subgroup (d <= 4 * LPR * ITERS); RB: rows in flight per subgroup; MODE 0 dot, 1 euclid;
// LPR: lanes that share a row -- 32, or 16 for d <= 64 (a 64-wide row fills 16 float4 lanes: sixteen subgroups keep twice the rows in
// flight, which is what a latency-bound sweep over a few hundred rows is short of)
constexpr int tiled_subgroups(int d) { return d <= 64 ? 16 : 8; }
// floats of dynamic LDS the body needs: y [mr4] | cf [mr4] | bc [mp4] float2 | partial dU [subgroups][d] | red [8]
__host__ __device__ inline int64_t tiled_lds_floats(int64_t max_rows, int64_t max_pos, int d)
{
    const int64_t mr4 = (max_rows + 3) & ~(int64_t)3, mp4 = (max_pos + 3) & ~(int64_t)3;
    return 2 * mr4 + 2 * mp4 + (int64_t)tiled_subgroups(d) * d + 8;
}

template <int ITERS, int RB, int MODE, int LPR = 32>
__device__ __forceinline__ void wmrb_tiled_user(
    float* __restrict__ lds, const int64_t u, const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ ub,
    const float* __restrict__ ib, const int64_t* __restrict__ indptr, const int32_t* __restrict__ xi,
    const int32_t* __restrict__ pos_slot, const float* __restrict__ pos_weight, const int32_t* samp, int32_t S, int d, float ratio,
    int32_t max_rows, int32_t max_pos, const TiledOut& o)
{
    // y [mr4] scores | cf [mr4] squared distance, then the pair's value | bc [mp4] {1 - y_q, c_q} | partial dU [8][d] | red [8]
    const int mr4 = (max_rows + 3) & ~3, mp4 = (max_pos + 3) & ~3;
    float* l_y = lds;
    float* l_cf = l_y + mr4;
    float2* l_bc = (float2*)(l_cf + mr4);
    float* l_part = (float*)(l_bc + mp4);
    constexpr int NSG = 256 / LPR;                       // subgroups of the workgroup
    float* l_red = l_part + NSG * d;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = tid & (LPR - 1), sg = tid / LPR;
    const int64_t b = indptr[u], e = indptr[u + 1];
    const int n_pos = (int)(e - b);
    const int R = S + n_pos;

    if (n_pos == 0) {
        // no interactions: no loss terms, every coefficient is 0
        for (int s = tid; s < S; s += 256) {
            if (o.val_samples) o.val_samples[u * S + s] = 0.f;
            if (o.raw_samples) o.raw_samples[u * S + s] = 0.f;
        }
        if (o.dU) for (int c = tid; c < d; c += 256) o.dU[u * d + c] = 0.f;
        if (o.val_rowsum && tid == 0) o.val_rowsum[u] = 0.f;
        if (o.dub && tid == 0) o.dub[u] = 0.f;
        return;                                              // (uniform over the workgroup; no LDS was touched)
    }

    f32x4 x[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + sub) * 4;
        const f32x4 v = *(const f32x4*)(U + u * d + (c < d ? c : 0));
        x[it] = (c < d) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float bu = ub ? ub[u] : 0.f;

    // ---- pass 1: scores of the rows j = j0 + sg + 8 r.  Every load is unconditional, from a clamped (valid) address, selected
    // afterwards (`cond ? load : 0` becomes an exec-masked block behind s_waitcnt vmcnt(0)) ----
    for (int j0 = 0; j0 < R; j0 += NSG * RB) {
        int32_t item[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int j = j0 + sg + NSG * r;
            const int q = j - S;
            const int32_t is = samp[j < S ? j : S - 1];
            const int32_t iq = xi[b + (q < 0 ? 0 : (q < n_pos ? q : n_pos - 1))];
            item[r] = (j < S) ? is : iq;
        }
        f32x4 y[RB][ITERS];
        float bi[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (it * LPR + sub) * 4;
                y[r][it] = *(const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0));
            }
            bi[r] = ib ? ib[item[r]] : 0.f;                   // (uniform branch; one address per subgroup)
        }
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float a = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const bool in = (it * LPR + sub) * 4 < d;      // (columns past d: both sides count as zero)
                if (MODE == 0) {
                    const f32x4 w = in ? y[r][it] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    a = fmaf(x[it].x, w.x, a); a = fmaf(x[it].y, w.y, a); a = fmaf(x[it].z, w.z, a); a = fmaf(x[it].w, w.w, a);
                } else {
                    const f32x4 w = in ? y[r][it] : x[it];
                    const float d0 = x[it].x - w.x, d1 = x[it].y - w.y, d2 = x[it].z - w.z, d3 = x[it].w - w.w;
                    a = fmaf(d0, d0, a); a = fmaf(d1, d1, a); a = fmaf(d2, d2, a); a = fmaf(d3, d3, a);
                }
            }
            acc[r] = a;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float t = acc[r];
            t = dpp_add(t, 8); t = dpp_add(t, 4); t = dpp_add(t, 2); t = dpp_add(t, 1);     // every lane of a 16-lane row: the row's sum
            acc[r] = (LPR == 32) ? dpp_add(t, 0) : t;                                        // 32 lanes: row 0's sum into row 1
        }
        if (sub == LPR - 1) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int j = j0 + sg + NSG * r;
                if (j < R) {
                    float s = acc[r];
                    if (MODE == 1) { l_cf[j] = s; s = -1.0f * sqrtf(fmaxf(s, EUCLID_EPS)); }
                    if (ub) s = s + bu;
                    if (ib) s = s + bi[r];
                    l_y[j] = s;
                    if (j >= S) o.pred_serial[b + (j - S)] = s;
                }
            }
        }
    }
    __syncthreads();

    // ---- loss terms (loss_graphs.py:153-180): 8 threads per interaction walk the S sample scores; lane k == 0 of the eight
    // then holds hinge sum and active count (fixed combination order), and finishes the interaction ----
    float raw_sum = 0.f;                                     // this thread's share of d b_u = sum of every pair's g
    float val_sum = 0.f;                                     // ... and of the sum of the values (euclidean: dU = val_sum U - G . V)
    for (int q0 = 0; q0 < n_pos; q0 += 32) {
        const int q = q0 + (tid >> 3), k = tid & 7;
        const bool live = q < n_pos;
        const float yq = l_y[S + (live ? q : 0)];
        const float base = 1.0f - yq;
        // what lane k == 0 needs from memory once the sums are there leaves now (clamped, unconditional), under the walk over S
        const int64_t pq = b + (live ? q : 0);
        const int32_t slot_ld = pos_slot[pq];
        const float w_ld = pos_weight ? pos_weight[pq] : 1.f;
        const int32_t xi_ld = xi[pq];
        float acc = 0.f;
        int cnt = 0;
        if (live) {
            for (int s4 = k * 4; s4 < S; s4 += 32) {
                const f32x4 v = *(const f32x4*)(l_y + s4);            // (l_y is padded to a multiple of 4: entries >= S masked)
                const float t0 = base + v.x, t1 = base + v.y, t2 = base + v.z, t3 = base + v.w;
                const bool m1 = s4 + 1 < S, m2 = s4 + 2 < S, m3 = s4 + 3 < S;
                acc += fmaxf(t0, 0.f); cnt += (t0 >= 0.f) ? 1 : 0;
                if (m1) { acc += fmaxf(t1, 0.f); cnt += (t1 >= 0.f) ? 1 : 0; }
                if (m2) { acc += fmaxf(t2, 0.f); cnt += (t2 >= 0.f) ? 1 : 0; }
                if (m3) { acc += fmaxf(t3, 0.f); cnt += (t3 >= 0.f) ? 1 : 0; }
            }
        }
        float fc = (float)cnt;
        acc += __shfl_xor(acc, 1, 64); fc += __shfl_xor(fc, 1, 64);
        acc += __shfl_xor(acc, 2, 64); fc += __shfl_xor(fc, 2, 64);
        acc += __shfl_xor(acc, 4, 64); fc += __shfl_xor(fc, 4, 64);
        if (live && k == 0) {
            const int32_t slot = slot_ld;
            float c = 0.f, dp = 0.f;
            if (slot >= 0) {
                const float w = w_ld;
                float smr = ratio * acc;
                if (pos_weight) smr = smr * w;
                c = ratio / (1.0f + smr);                               // d loss_p / d (hinge sum), upstream gradient 1
                if (pos_weight) c = c * w;
                o.loss[slot] = logf(smr + 1.0f);
                dp = -c * fc;
            }
            l_bc[q] = make_float2(base, c);
            float val = dp;
            if (MODE == 1) { const float D = l_cf[S + q]; val = (D >= EUCLID_EPS) ? -dp / sqrtf(D) : 0.f; }
            l_cf[S + q] = val;
            if (o.val_pairs) o.val_pairs[b + q] = val;
            if (o.raw_pairs) o.raw_pairs[b + q] = dp;
            if (o.dense_g) unsafeAtomicAdd(o.dense_g + u * o.ldg + xi_ld, val);
            raw_sum += dp;
            val_sum += val;
        }
    }
    __syncthreads();
    // per sample: its coefficient over the user's interactions (c_q = 0 for non-positive ones); four samples per thread share
    // every {1 - y_q, c_q} read
    for (int s0 = tid; s0 < S; s0 += 1024) {
        float ys[4], g[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int s = s0 + 256 * m;
            ys[m] = l_y[s < S ? s : S - 1];
            g[m] = 0.f;
        }
#pragma unroll 4
        for (int q = 0; q < n_pos; ++q) {
            const float2 bc = l_bc[q];
#pragma unroll
            for (int m = 0; m < 4; ++m) g[m] += (bc.x + ys[m] >= 0.f) ? bc.y : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int s = s0 + 256 * m;
            if (s < S) {
                float val = g[m];
                if (MODE == 1) { const float D = l_cf[s]; val = (D >= EUCLID_EPS) ? -g[m] / sqrtf(D) : 0.f; }
                l_cf[s] = val;
                if (o.val_samples) o.val_samples[u * S + s] = val;
                if (o.raw_samples) o.raw_samples[u * S + s] = g[m];
                if (o.dense_g) unsafeAtomicAdd(o.dense_g + u * o.ldg + samp[s], val);
                raw_sum += g[m];
                val_sum += val;
            }
        }
    }
    __syncthreads();

    // ---- pass 2: dU_u = sum_j val_j * row_j (dot) / sum_j val_j * (U_u - row_j) (euclidean); skipped when the caller takes dU
    // from the dense matrix (G . V on fp32 MFMA: the second sweep over the rows is 45 ms of configs[4]'s step, the GEMM 19) ----
    if (o.dU) {
        f32x4 part[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) part[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < R; j0 += NSG * RB) {
            int32_t item[RB];
            float cf[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int j = j0 + sg + NSG * r;
                const int q = j - S;
                const int32_t is = samp[j < S ? j : S - 1];
                const int32_t iq = xi[b + (q < 0 ? 0 : (q < n_pos ? q : n_pos - 1))];
                item[r] = (j < S) ? is : iq;
                const float v = l_cf[j < R ? j : 0];
                cf[r] = (j < R) ? v : 0.f;
            }
            f32x4 y[RB][ITERS];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int c = (it * LPR + sub) * 4;
                    y[r][it] = *(const f32x4*)(V + (int64_t)item[r] * d + (c < d ? c : 0));
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    f32x4 w = y[r][it];
                    if (MODE == 1) { w.x = x[it].x - w.x; w.y = x[it].y - w.y; w.z = x[it].z - w.z; w.w = x[it].w - w.w; }
                    part[it].x = fmaf(cf[r], w.x, part[it].x); part[it].y = fmaf(cf[r], w.y, part[it].y);
                    part[it].z = fmaf(cf[r], w.z, part[it].z); part[it].w = fmaf(cf[r], w.w, part[it].w);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * LPR + sub) * 4;
            if (c < d) *(f32x4*)(l_part + sg * d + c) = part[it];
        }
    }
    if (o.dub) {
        for (int off = 32; off > 0; off >>= 1) raw_sum += __shfl_xor(raw_sum, off, 64);
        if (lane == 0) l_red[wave] = raw_sum;
    }
    if (o.val_rowsum) {
        for (int off = 32; off > 0; off >>= 1) val_sum += __shfl_xor(val_sum, off, 64);
        if (lane == 0) l_red[4 + wave] = val_sum;
    }
    __syncthreads();
    if (o.dU) {
        for (int c = tid; c < d; c += 256) {
            float acc = l_part[c];
#pragma unroll
            for (int g8 = 1; g8 < NSG; ++g8) acc += l_part[g8 * d + c];
            o.dU[u * d + c] = acc;
        }
    }
    if (o.dub && tid == 0) o.dub[u] = (l_red[0] + l_red[1]) + (l_red[2] + l_red[3]);
    if (o.val_rowsum && tid == 0) o.val_rowsum[u] = (l_red[4] + l_red[5]) + (l_red[6] + l_red[7]);
    __syncthreads();
}

}  // namespace
