// tensorrec_amd/csrc/loss.hip -- K6: WMRB / BalancedWMRB forward + backward, and the RMSE loss.
//
// WMRB (tensorrec/loss_graphs.py:153-180): for every POSITIVE interaction p = (u, i)
//     loss_p = log(1 + (n_items / S) * sum_s max(0, 1 - yhat_p + yhat[u, s]))          (a [P+] vector)
// BalancedWMRB (:189-227) multiplies the sum by value_p / (sum of positive values of item i).
// TF materialises a [P+, S] tensor (boolean_mask + gather, :167-174); here one workgroup owns one user: the
// user's S sampled predictions sit in LDS and every wave walks a share of the user's positives, so nothing of
// size P+ x S ever exists.  Backward (what TF autodiff yields, with tf.maximum passing the gradient when the
// hinge argument is >= 0): c_p = go_p * (n_items/S) * w_p / (1 + smr_p),
//     d yhat_p     = -c_p * #{s active}            d yhat[u, s] = sum_{p in user u} c_p * [active(p, s)]
// The second sum is done with threads owning s and looping over p (no atomics, fixed order -> deterministic).
//
// RMSE (loss_graphs.py:58-59): sqrt(mean((y - yhat)^2)); backward d yhat_p = go * -(y_p - yhat_p) / (P * loss).
#include "common.hpp"

#define WMRB_MAX_POS_LDS 1024      // positives of one user staged per pass

// interactions are CSR over users: pos entries of user u are [indptr[u], indptr[u+1]); pos_slot[p] is the index of
// interaction p inside the compacted [P+] vector (-1 for non-positive interactions, which WMRB ignores).
__global__ __launch_bounds__(256) void wmrb_fwd_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ pos_slot, const float* __restrict__ pos_weight,
    const float* __restrict__ pred, const float* __restrict__ samp, int32_t S, float ratio,
    float* __restrict__ loss, float* __restrict__ smr_out)
{
    extern __shared__ float lds[];           // [S] sampled predictions of this user
    const int64_t u = blockIdx.x;
    const int64_t b = indptr[u], e = indptr[u + 1];
    if (b == e) return;
    // (four loads per thread in flight, then the four LDS stores: a load -> store loop waits for every load on its own)
    for (int s0 = threadIdx.x; s0 < S; s0 += 1024) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int s = s0 + 256 * k; t[k] = samp[u * S + (s < S ? s : S - 1)]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int s = s0 + 256 * k; if (s < S) lds[s] = t[k]; }
    }
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int64_t p = b + wave; p < e; p += 4) {
        const int32_t slot = pos_slot[p];
        if (slot < 0) continue;
        const float yp = pred[p];
        float acc = 0.f;
        for (int s = lane; s < S; s += 64) acc += fmaxf((1.0f - yp) + lds[s], 0.f);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) {
            float smr = ratio * acc;
            if (pos_weight) smr = smr * pos_weight[p];        // value_p / gathered item sum (balanced)
            smr_out[slot] = smr;
            loss[slot] = logf(smr + 1.0f);
        }
    }
}

__global__ __launch_bounds__(256) void wmrb_bwd_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ pos_slot, const float* __restrict__ pos_weight,
    const float* __restrict__ pred, const float* __restrict__ samp, const float* __restrict__ smr,
    const float* __restrict__ grad_out, int32_t S, float ratio, float* __restrict__ d_pred,
    float* __restrict__ d_samp)
{
    extern __shared__ float lds[];           // [S] samples | [WMRB_MAX_POS_LDS] yp | [WMRB_MAX_POS_LDS] c
    float* l_s = lds;
    float* l_yp = lds + S;
    float* l_c = l_yp + WMRB_MAX_POS_LDS;
    const int64_t u = blockIdx.x;
    const int64_t b = indptr[u], e = indptr[u + 1];
    for (int s0 = threadIdx.x; s0 < S; s0 += 1024) {         // (batched like the forward kernel's fill)
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int s = s0 + 256 * k; t[k] = samp[u * S + (s < S ? s : S - 1)]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int s = s0 + 256 * k; if (s < S) l_s[s] = t[k]; }
    }
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // per-thread accumulators for the samples this thread owns (s = tid, tid+256, ...): kept in LDS-free registers
    // by walking sample blocks of 256 at a time in the outer loop below.
    for (int64_t p0 = b; p0 < e || p0 == b; p0 += WMRB_MAX_POS_LDS) {
        const int np = (int)((e - p0 < WMRB_MAX_POS_LDS) ? (e - p0) : WMRB_MAX_POS_LDS);
        __syncthreads();
        for (int q = threadIdx.x; q < np; q += 256) {
            const int64_t p = p0 + q;
            const int32_t slot = pos_slot[p];
            float c = 0.f;
            if (slot >= 0) {
                c = grad_out[slot] * ratio / (1.0f + smr[slot]);
                if (pos_weight) c = c * pos_weight[p];
            }
            l_yp[q] = pred[p];
            l_c[q] = c;
        }
        __syncthreads();
        // (1) gradient w.r.t. the positive's own prediction: waves over positives, lanes over samples
        for (int q = wave; q < np; q += 4) {
            const float c = l_c[q];
            const int64_t p = p0 + q;
            if (c == 0.f) { if (lane == 0 && p0 + q < e) d_pred[p] = 0.f; continue; }
            const float base = 1.0f - l_yp[q];
            int cnt = 0;
            for (int s = lane; s < S; s += 64) cnt += (base + l_s[s] >= 0.f);
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (lane == 0) d_pred[p] = -c * (float)cnt;
        }
        // (2) gradient w.r.t. the shared samples: threads own s, loop over this pass's positives in order
        for (int s = threadIdx.x; s < S; s += 256) {
            const float ys = l_s[s];
            float acc = (p0 == b) ? 0.f : d_samp[u * S + s];
            for (int q = 0; q < np; ++q) {
                const float c = l_c[q];
                acc += ((1.0f - l_yp[q]) + ys >= 0.f) ? c : 0.f;
            }
            d_samp[u * S + s] = acc;
        }
        if (e == b) break;
    }
}

// ---- wave-per-user forms (S <= 256) -------------------------------------------------------------------------
// At the BASELINE fit shape a user has ~20 positives and 100 samples: a 256-thread workgroup per user is mostly idle
// and the launch is bound by workgroup dispatch (1M workgroups).  Here a wave owns a user: the samples sit in <= 4
// registers per lane (sample s = r*64 + lane, the same lane/order split as the workgroup kernels, so results are
// bit-identical), the user's positives are loaded by the lanes in parallel (64 per pass) and broadcast with readlane.
#define WMRB_WAVE_SR 4

__global__ __launch_bounds__(256) void wmrb_fwd_wave_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ pos_slot, const float* __restrict__ pos_weight,
    const float* __restrict__ pred, const float* __restrict__ samp, int64_t n_users, int32_t S, float ratio,
    float* __restrict__ loss, float* __restrict__ smr_out)
{
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (u >= n_users) return;
    const int lane = lane_id();
    const int64_t b = indptr[u], e = indptr[u + 1];
    if (b == e) return;
    float ys[WMRB_WAVE_SR];
#pragma unroll
    for (int r = 0; r < WMRB_WAVE_SR; ++r) ys[r] = (r * 64 + lane < S) ? samp[u * S + r * 64 + lane] : -INFINITY;
    for (int64_t p0 = b; p0 < e; p0 += 64) {
        const int np = (int)((e - p0 < 64) ? (e - p0) : 64);
        int32_t slot = -1;
        float yp = 0.f, w = 1.f;
        if (lane < np) {
            slot = pos_slot[p0 + lane];
            yp = pred[p0 + lane];
            if (pos_weight) w = pos_weight[p0 + lane];
        }
        float my_smr = 0.f;
        for (int q = 0; q < np; ++q) {
            if (__shfl(slot, q, 64) < 0) continue;                 // wave-uniform
            const float base = 1.0f - __shfl(yp, q, 64);
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < WMRB_WAVE_SR; ++r) acc += fmaxf(base + ys[r], 0.f);     // -inf padding adds 0
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == q) my_smr = ratio * acc;
        }
        if (lane < np && slot >= 0) {
            if (pos_weight) my_smr = my_smr * w;
            smr_out[slot] = my_smr;
            loss[slot] = logf(my_smr + 1.0f);
        }
    }
}

__global__ __launch_bounds__(256) void wmrb_bwd_wave_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ pos_slot, const float* __restrict__ pos_weight,
    const float* __restrict__ pred, const float* __restrict__ samp, const float* __restrict__ smr,
    const float* __restrict__ grad_out, int64_t n_users, int32_t S, float ratio, float* __restrict__ d_pred,
    float* __restrict__ d_samp)
{
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (u >= n_users) return;
    const int lane = lane_id();
    const int64_t b = indptr[u], e = indptr[u + 1];
    float ys[WMRB_WAVE_SR], acc[WMRB_WAVE_SR];
#pragma unroll
    for (int r = 0; r < WMRB_WAVE_SR; ++r) {
        ys[r] = (r * 64 + lane < S) ? samp[u * S + r * 64 + lane] : -INFINITY;
        acc[r] = 0.f;
    }
    for (int64_t p0 = b; p0 < e; p0 += 64) {
        const int np = (int)((e - p0 < 64) ? (e - p0) : 64);
        float c = 0.f, yp = 0.f;
        if (lane < np) {
            const int32_t slot = pos_slot[p0 + lane];
            yp = pred[p0 + lane];
            if (slot >= 0) {
                c = grad_out[slot] * ratio / (1.0f + smr[slot]);
                if (pos_weight) c = c * pos_weight[p0 + lane];
            }
        }
        float my_dp = 0.f;
        for (int q = 0; q < np; ++q) {
            const float cq = __shfl(c, q, 64);
            if (cq == 0.f) continue;                               // wave-uniform; d_pred stays 0
            const float base = 1.0f - __shfl(yp, q, 64);
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < WMRB_WAVE_SR; ++r) {
                const bool active = base + ys[r] >= 0.f;           // -inf padding is never active
                cnt += __popcll(__builtin_amdgcn_ballot_w64(active));
                acc[r] += active ? cq : 0.f;
            }
            if (lane == q) my_dp = -cq * (float)cnt;
        }
        if (lane < np) d_pred[p0 + lane] = my_dp;
    }
#pragma unroll
    for (int r = 0; r < WMRB_WAVE_SR; ++r)
        if (r * 64 + lane < S) d_samp[u * S + r * 64 + lane] = acc[r];
}

// ---- RMSE -------------------------------------------------------------------------------------------------
// pass 1: per-block partial sums of squared error (fixed tree order), pass 2 (one block): combine, sqrt, and
// optionally the backward scale.  Two launches keep the reduction order independent of scheduling.
__global__ __launch_bounds__(256) void sqerr_partial_kernel(const float* __restrict__ y, const float* __restrict__ pred,
                                                           int64_t n, float* __restrict__ partial)
{
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float e = y[i] - pred[i];
        acc = fmaf(e, e, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void rmse_finish_kernel(const float* __restrict__ partial, int n_partial, int64_t n,
                                                         float* __restrict__ loss)
{
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sqrtf(red[0] / (float)n);
}

__global__ __launch_bounds__(256) void rmse_bwd_kernel(const float* __restrict__ y, const float* __restrict__ pred,
                                                      const float* __restrict__ loss, const float* __restrict__ grad_out,
                                                      int64_t n, float* __restrict__ d_pred)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // d sqrt(m)/dm = 0.5/sqrt(m); dm/de = 2e/n; de/dpred = -1
    const float scale = grad_out[0] / ((float)n * loss[0]);
    d_pred[i] = -(y[i] - pred[i]) * scale;
}

extern "C" int trec_wmrb_fwd(const int64_t* indptr, const int32_t* pos_slot, const float* pos_weight,
                             const float* pred_serial, const float* sample_pred, int64_t n_users, int64_t n_items,
                             int32_t n_sampled, float* loss, float* smr, void* stream)
{
    TREC_REQUIRE(indptr && pos_slot && pred_serial && sample_pred && loss && smr, "trec_wmrb_fwd: null pointer");
    TREC_REQUIRE(n_sampled >= 1 && n_sampled <= 16384, "trec_wmrb_fwd: n_sampled must be in [1, 16384]");
    if (n_users == 0) return TREC_OK;
    const float ratio = (float)n_items / (float)n_sampled;
    if (n_sampled <= 64 * WMRB_WAVE_SR && trec_get_tuning("wmrb_wave", 1)) {
        hipLaunchKernelGGL(wmrb_fwd_wave_kernel, dim3((unsigned)ceil_div64(n_users * 64, 256)), dim3(256), 0,
                           (hipStream_t)stream, indptr, pos_slot, pos_weight, pred_serial, sample_pred, n_users, n_sampled,
                           ratio, loss, smr);
        return trec_check_launch("trec_wmrb_fwd");
    }
    hipLaunchKernelGGL(wmrb_fwd_kernel, dim3((unsigned)n_users), dim3(256), sizeof(float) * n_sampled,
                       (hipStream_t)stream, indptr, pos_slot, pos_weight, pred_serial, sample_pred, n_sampled, ratio,
                       loss, smr);
    return trec_check_launch("trec_wmrb_fwd");
}

extern "C" int trec_wmrb_bwd(const int64_t* indptr, const int32_t* pos_slot, const float* pos_weight,
                             const float* pred_serial, const float* sample_pred, const float* smr,
                             const float* grad_loss, int64_t n_users, int64_t n_items, int32_t n_sampled,
                             float* d_pred_serial, float* d_sample_pred, void* stream)
{
    TREC_REQUIRE(indptr && pos_slot && pred_serial && sample_pred && smr && grad_loss && d_pred_serial && d_sample_pred,
                 "trec_wmrb_bwd: null pointer");
    TREC_REQUIRE(n_sampled >= 1 && n_sampled <= 16384, "trec_wmrb_bwd: n_sampled must be in [1, 16384]");
    if (n_users == 0) return TREC_OK;
    const float ratio = (float)n_items / (float)n_sampled;
    if (n_sampled <= 64 * WMRB_WAVE_SR && trec_get_tuning("wmrb_wave", 1)) {
        hipLaunchKernelGGL(wmrb_bwd_wave_kernel, dim3((unsigned)ceil_div64(n_users * 64, 256)), dim3(256), 0,
                           (hipStream_t)stream, indptr, pos_slot, pos_weight, pred_serial, sample_pred, smr, grad_loss,
                           n_users, n_sampled, ratio, d_pred_serial, d_sample_pred);
        return trec_check_launch("trec_wmrb_bwd");
    }
    const size_t lds = sizeof(float) * ((size_t)n_sampled + 2 * WMRB_MAX_POS_LDS);
    hipLaunchKernelGGL(wmrb_bwd_kernel, dim3((unsigned)n_users), dim3(256), lds, (hipStream_t)stream, indptr, pos_slot,
                       pos_weight, pred_serial, sample_pred, smr, grad_loss, n_sampled, ratio, d_pred_serial,
                       d_sample_pred);
    return trec_check_launch("trec_wmrb_bwd");
}

extern "C" int trec_rmse_fwd(const float* y, const float* pred, int64_t n, float* partial_ws, int32_t n_partial,
                             float* loss, void* stream)
{
    TREC_REQUIRE(y && pred && partial_ws && loss, "trec_rmse_fwd: null pointer");
    TREC_REQUIRE(n >= 1 && n_partial >= 1 && n_partial <= 65535, "trec_rmse_fwd: bad sizes");
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, y, pred, n,
                       partial_ws);
    hipLaunchKernelGGL(rmse_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial_ws, n_partial, n, loss);
    return trec_check_launch("trec_rmse_fwd");
}

extern "C" int trec_rmse_bwd(const float* y, const float* pred, const float* loss, const float* grad_loss, int64_t n,
                             float* d_pred, void* stream)
{
    TREC_REQUIRE(y && pred && loss && grad_loss && d_pred, "trec_rmse_bwd: null pointer");
    if (n == 0) return TREC_OK;
    hipLaunchKernelGGL(rmse_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, y, pred,
                       loss, grad_loss, n, d_pred);
    return trec_check_launch("trec_rmse_bwd");
}
