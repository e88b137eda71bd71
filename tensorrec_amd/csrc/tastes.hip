// tensorrec_amd/csrc/tastes.hip -- K9: collapse of the mixture of tastes (+ the bias add that follows it).
//
// tensorrec/recommendation_graphs.py:85-109: T prediction tensors (one per taste) of the same shape are stacked and
// collapsed element by element,
//     no attention:  out = max_t p_t                                          (tf.reduce_max, :107)
//     attention:     out = sum_t p_t * softmax_t(a)  with  softmax_t(a) = exp(a_t - max a) / sum_t' exp(a_t' - max a)
//                                                                             (tf.nn.softmax(axis=0), :98-103)
// after which the reference adds the projected biases (tensorrec.py:432-449).  TF runs this as stack / softmax /
// multiply / reduce_sum over [T, n] copies (>= 6 passes over HBM); here it is ONE streaming pass: T (or 2T) coalesced
// loads and one store per element -- algorithmic bytes 4(T+1) (max) or 4(2T+1) (attention) per element, HBM-bound.
// Sums run over t = 0..T-1 in order, every product and sum rounded separately (the file is built with
// -ffp-contract=off), which is the order of the NumPy oracle.
//
// Backward (what TF autodiff yields):
//     max:        d p_t = g * [p_t == out] / #{t': p_t' == out}               (reduce_max splits ties evenly)
//     attention:  d p_t = g * w_t ,   d a_t = g * w_t * (p_t - out)
// The bias gradients are g itself summed per user / per item; the host does that with the segmented K1 kernels.
#include "common.hpp"

#define TASTES_MAX 16

// index of the (user, item) of element e: mode 0 = no biases, 1 = serial pairs, 2 = dense [n / n_cols, n_cols]
__device__ __forceinline__ float add_biases(float c, int64_t e, int mode, const float* __restrict__ ub,
                                            const float* __restrict__ ib, const int32_t* __restrict__ xu,
                                            const int32_t* __restrict__ xi, int64_t span)
{
    if (mode == 0) return c;
    int64_t u, i;
    if (mode == 1) {
        u = xu ? (int64_t)xu[e] : e / span;
        i = xi[e];
    } else {
        u = e / span;
        i = e - u * span;
    }
    if (ub) c = c + ub[u];
    if (ib) c = c + ib[i];
    return c;
}

template <bool ATTN>
__global__ __launch_bounds__(256) void collapse_fwd_kernel(
    const float* __restrict__ preds, const float* __restrict__ attn, int T, int64_t n, int mode,
    const float* __restrict__ ub, const float* __restrict__ ib, const int32_t* __restrict__ xu,
    const int32_t* __restrict__ xi, int64_t span, float* __restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        float c;
        if (ATTN) {
            float a[TASTES_MAX];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    a[t] = attn[(int64_t)t * n + e];
                    m = fmaxf(m, a[t]);
                }
            float z = 0.f;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    a[t] = expf(a[t] - m);
                    z = z + a[t];
                }
            c = 0.f;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) c = c + preds[(int64_t)t * n + e] * (a[t] / z);
        } else {
            c = preds[e];
            for (int t = 1; t < T; ++t) c = fmaxf(c, preds[(int64_t)t * n + e]);
        }
        out[e] = add_biases(c, e, mode, ub, ib, xu, xi, span);
    }
}

template <bool ATTN>
__global__ __launch_bounds__(256) void collapse_bwd_kernel(
    const float* __restrict__ preds, const float* __restrict__ attn, const float* __restrict__ g, int T, int64_t n,
    float* __restrict__ d_preds, float* __restrict__ d_attn)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        const float go = g[e];
        if (ATTN) {
            float a[TASTES_MAX], p[TASTES_MAX];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    a[t] = attn[(int64_t)t * n + e];
                    p[t] = preds[(int64_t)t * n + e];
                    m = fmaxf(m, a[t]);
                }
            float z = 0.f;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    a[t] = expf(a[t] - m);
                    z = z + a[t];
                }
            float c = 0.f;
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    a[t] = a[t] / z;
                    c = c + p[t] * a[t];
                }
#pragma unroll
            for (int t = 0; t < TASTES_MAX; ++t)
                if (t < T) {
                    const float gw = go * a[t];
                    d_preds[(int64_t)t * n + e] = gw;
                    d_attn[(int64_t)t * n + e] = gw * (p[t] - c);
                }
        } else {
            float c = preds[e];
            for (int t = 1; t < T; ++t) c = fmaxf(c, preds[(int64_t)t * n + e]);
            int cnt = 0;
            for (int t = 0; t < T; ++t) cnt += (preds[(int64_t)t * n + e] == c) ? 1 : 0;
            const float share = go / (float)cnt;
            for (int t = 0; t < T; ++t) d_preds[(int64_t)t * n + e] = (preds[(int64_t)t * n + e] == c) ? share : 0.f;
        }
    }
}

static inline int collapse_grid(int64_t n) {
    int64_t blocks = ceil_div64(n, 256);
    const int64_t cap = 256 * 16;                 // 16 workgroups per CU, grid-stride beyond that
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

extern "C" int trec_collapse_tastes_fwd(const float* preds, const float* attn, int32_t n_tastes, int64_t n,
                                        int32_t bias_mode, const float* user_bias, const float* item_bias,
                                        const int32_t* x_user, const int32_t* x_item, int64_t span, float* out,
                                        void* stream)
{
    TREC_REQUIRE(n >= 0 && n_tastes >= 1 && n_tastes <= TASTES_MAX, "trec_collapse_tastes_fwd: 1 <= n_tastes <= 16");
    if (n == 0) return TREC_OK;
    TREC_REQUIRE(preds && out, "trec_collapse_tastes_fwd: null pointer");
    TREC_REQUIRE(bias_mode >= 0 && bias_mode <= 2, "trec_collapse_tastes_fwd: bias_mode must be 0, 1 or 2");
    if (bias_mode == 1) TREC_REQUIRE(x_item && (x_user || span > 0), "trec_collapse_tastes_fwd: serial mode needs indices");
    if (bias_mode == 2) TREC_REQUIRE(span > 0 && n % span == 0, "trec_collapse_tastes_fwd: dense mode needs n_cols | n");
    if (!user_bias && !item_bias) bias_mode = 0;
    hipStream_t s = (hipStream_t)stream;
    if (attn)
        hipLaunchKernelGGL(collapse_fwd_kernel<true>, dim3(collapse_grid(n)), dim3(256), 0, s, preds, attn, n_tastes, n,
                           bias_mode, user_bias, item_bias, x_user, x_item, span, out);
    else
        hipLaunchKernelGGL(collapse_fwd_kernel<false>, dim3(collapse_grid(n)), dim3(256), 0, s, preds, attn, n_tastes,
                           n, bias_mode, user_bias, item_bias, x_user, x_item, span, out);
    return trec_check_launch("trec_collapse_tastes_fwd");
}

extern "C" int trec_collapse_tastes_bwd(const float* preds, const float* attn, const float* grad_out, int32_t n_tastes,
                                        int64_t n, float* d_preds, float* d_attn, void* stream)
{
    TREC_REQUIRE(n >= 0 && n_tastes >= 1 && n_tastes <= TASTES_MAX, "trec_collapse_tastes_bwd: 1 <= n_tastes <= 16");
    if (n == 0) return TREC_OK;
    TREC_REQUIRE(preds && grad_out && d_preds, "trec_collapse_tastes_bwd: null pointer");
    TREC_REQUIRE(!attn || d_attn, "trec_collapse_tastes_bwd: d_attn missing");
    hipStream_t s = (hipStream_t)stream;
    if (attn)
        hipLaunchKernelGGL(collapse_bwd_kernel<true>, dim3(collapse_grid(n)), dim3(256), 0, s, preds, attn, grad_out,
                           n_tastes, n, d_preds, d_attn);
    else
        hipLaunchKernelGGL(collapse_bwd_kernel<false>, dim3(collapse_grid(n)), dim3(256), 0, s, preds, attn, grad_out,
                           n_tastes, n, d_preds, d_attn);
    return trec_check_launch("trec_collapse_tastes_bwd");
}
