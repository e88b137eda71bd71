// tensorrec_amd/csrc/loss_dense.hip -- the dense and separation losses as streaming reductions.
//
//   RMSEDenseLossGraph        tensorrec/loss_graphs.py:62-72    sqrt(mean((sparse_to_dense(interactions) - prediction)^2))
//   SeparationLossGraph       tensorrec/loss_graphs.py:75-97    1 - Normal(mu_n - mu_p, sqrt(var_n + var_p)).cdf(0) over the interactions,
//                                                              positives = {y > 0}, negatives = {y <= 0}, tf.nn.moments (population)
//   SeparationDenseLossGraph  tensorrec/loss_graphs.py:100-134  the same over ALL n_users x n_items predictions, every pair that is
//                                                              not a positive interaction counting as a negative
//
// The reference builds masks, boolean_mask copies and dense interaction matrices ([U, I] twice over) and reduces them with five
// TF ops per moment.  Here the [U, I] prediction is read ONCE per pass by a grid-stride reduction (HBM-bound: 4 B per
// prediction), the interactions enter as their sparse list -- "all minus the positives" gives the negatives' moments -- and the
// backward pass is one streaming write of the dense gradient plus a scatter over the interactions.  Sums are accumulated in
// double (per thread, per wave by shuffles, one atomicAdd per workgroup), so the result does not depend on the reduction order
// beyond 1e-12; the tests hold it to 1e-5 of the reference's own arithmetic (fixtures made by running the reference's source).
#include "common.hpp"
#include <math.h>

namespace {

__device__ __forceinline__ double wave_sum(double x)
{
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// out[0] += sum x, out[1] += sum (x - mu)^2 over a dense [rows, cols] matrix with row stride ld
__global__ __launch_bounds__(256) void dense_moments_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                           const double* __restrict__ mu_ptr, double* __restrict__ out)
{
    const double mu = mu_ptr ? mu_ptr[0] : 0.0;
    double s = 0.0, q = 0.0;
    const int64_t n = rows * cols;
    if (ld == cols && (cols & 3) == 0 && ((uintptr_t)x % 16) == 0) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
            const f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + i));
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double t = (double)v[e]; s += t; q += (t - mu) * (t - mu); }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const double t = (double)x[(i / cols) * ld + (i % cols)];
            s += t; q += (t - mu) * (t - mu);
        }
    }
    s = wave_sum(s); q = wave_sum(q);
    __shared__ double ws[4], wq[4];
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, ws[0] + ws[1] + ws[2] + ws[3]);
        atomicAdd(out + 1, wq[0] + wq[1] + wq[2] + wq[3]);
    }
}

// The interactions' share, gathered from the dense prediction (pred != NULL: x = pred[xu * ld + xi]) or taken from the serial
// predictions (pred == NULL: x = serial[p]).  y = values[p].  out (double[8]):
//   [0] n_pos  [1] sum_pos x  [2] sum_pos (x - mu[0])^2  [3] sum_pos (x - mu[1])^2
//   [4] n_neg  [5] sum_neg x  [6] sum_neg (x - mu[1])^2  [7] sum (y^2 - 2 y x)
// (positive: y > 0; "neg" here = EXPLICIT entries with y <= 0; mu[0] = positives' mean, mu[1] = negatives' mean, both 0 in pass 1)
__global__ __launch_bounds__(256) void pair_moments_kernel(const float* __restrict__ pred, int64_t ld, const float* __restrict__ serial,
                                                          const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                          const float* __restrict__ y, int64_t n_pairs,
                                                          const double* __restrict__ mu, double* __restrict__ out)
{
    const double mp = mu ? mu[0] : 0.0, mn = mu ? mu[1] : 0.0;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * 256) {
        const double x = pred ? (double)pred[(int64_t)xu[p] * ld + xi[p]] : (double)serial[p];
        const double yy = (double)y[p];
        if (yy > 0.0) { a[0] += 1.0; a[1] += x; a[2] += (x - mp) * (x - mp); a[3] += (x - mn) * (x - mn); }
        else { a[4] += 1.0; a[5] += x; a[6] += (x - mn) * (x - mn); }
        a[7] += yy * yy - 2.0 * yy * x;
    }
    __shared__ double wsum[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double t = wave_sum(a[j]);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][j] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(out + threadIdx.x, wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
}

// st (double[16]) after both passes -> loss[0] and the backward coefficients, in place:
//   in : [0..1] dense {sum x, sum (x - mu_n)^2} (dense form only)   [2..9] pair_moments pass-2 output   [10] n_all (dense) or -1
//        [11] mu_p  [12] mu_n
//   out: [13] scale = sqrt(var_n + var_p)   [14] loc = mu_n - mu_p   [15] -phi(z) / scale with z = -loc / scale
__global__ void separation_finish_kernel(double* __restrict__ st, float* __restrict__ loss)
{
    const double* pm = st + 2;
    const double np = pm[0];
    double nn, qn;
    if (st[10] >= 0.0) { nn = st[10] - np; qn = st[1] - pm[3]; }           // dense: negatives = everything but the positives
    else { nn = pm[4]; qn = pm[6]; }
    const double var_p = pm[2] / np, var_n = qn / nn;
    const double loc = st[12] - st[11];
    const double scale = sqrt(var_n + var_p);
    const double z = (0.0 - loc) / scale;
    const double cdf0 = 0.5 * (1.0 + erf(z * 0.70710678118654752440));
    loss[0] = (float)(1.0 - cdf0);
    st[13] = scale; st[14] = loc;
    st[15] = -0.39894228040143267794 * exp(-0.5 * z * z) / scale;
}

// means between the passes: st[11] = mu_p, st[12] = mu_n (dense: negatives = all - positives)
// (n_all: the number of dense predictions, -1 for the serial form; kept in st[10] for the finish and the backward pass)
__global__ void separation_means_kernel(double* __restrict__ st, double n_all)
{
    const double* pm = st + 2;
    const double np = pm[0];
    st[10] = n_all;
    st[11] = pm[1] / np;
    if (n_all >= 0.0) st[12] = (st[0] - pm[1]) / (n_all - np);
    else st[12] = pm[5] / pm[4];
    for (int j = 0; j < 10; ++j) st[j] = 0.0;                            // pass 2 accumulates into the same slots (counts again too)
}

// d loss / d x for one prediction of the given class (gl = upstream gradient):
//   positive: c / n_p * (1 + loc (x - mu_p) / scale^2)      negative: c / n_n * (-1 + loc (x - mu_n) / scale^2),   c = st[15]
__device__ __forceinline__ float sep_grad(double x, bool positive, const double* st, double np, double nn, double gl)
{
    const double s2 = st[13] * st[13];
    const double g = positive ? st[15] / np * (1.0 + st[14] * (x - st[11]) / s2) : st[15] / nn * (-1.0 + st[14] * (x - st[12]) / s2);
    return (float)(gl * g);
}

__global__ __launch_bounds__(256) void separation_bwd_serial_kernel(const float* __restrict__ serial, const float* __restrict__ y,
                                                                   int64_t n, const double* __restrict__ st,
                                                                   const float* __restrict__ gl, float* __restrict__ dx)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    dx[p] = sep_grad((double)serial[p], y[p] > 0.f, st, st[2], st[6], (double)gl[0]);
}

// dense backward, pass 1: every entry as a negative; pass 2 (pairs): the positives get their own formula
__global__ __launch_bounds__(256) void separation_bwd_dense_kernel(const float* __restrict__ x, int64_t n, const double* __restrict__ st,
                                                                  const float* __restrict__ gl, float* __restrict__ dx)
{
    const double np = st[2], nn = st[10] - st[2], g = (double)gl[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = sep_grad((double)x[i], false, st, np, nn, g);
}

__global__ __launch_bounds__(256) void separation_bwd_pairs_kernel(const float* __restrict__ x, int64_t ld, const int32_t* __restrict__ xu,
                                                                  const int32_t* __restrict__ xi, const float* __restrict__ y,
                                                                  int64_t n_pairs, const double* __restrict__ st,
                                                                  const float* __restrict__ gl, float* __restrict__ dx)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs || !(y[p] > 0.f)) return;
    const int64_t at = (int64_t)xu[p] * ld + xi[p];
    dx[at] = sep_grad((double)x[at], true, st, st[2], st[10] - st[2], (double)gl[0]);
}

// RMSE dense: st[0] = sum p^2 (dense pass, mu = 0 -> slot 1), st[9] = sum (y^2 - 2 y p); loss = sqrt((st[1] + st[9]) / n_all)
__global__ void rmse_dense_finish_kernel(double* __restrict__ st, float* __restrict__ loss, double n_all)
{
    st[10] = n_all;
    const double mse = (st[1] + st[9]) / st[10];
    const double l = sqrt(mse > 0.0 ? mse : 0.0);
    loss[0] = (float)l;
    st[15] = l > 0.0 ? 1.0 / (l * st[10]) : 0.0;                           // d loss / d p = (p - y) / (loss * n_all)
}

__global__ __launch_bounds__(256) void rmse_dense_bwd_kernel(const float* __restrict__ x, int64_t n, const double* __restrict__ st,
                                                            const float* __restrict__ gl, float* __restrict__ dx)
{
    const float c = (float)((double)gl[0] * st[15]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = c * x[i];
}

__global__ __launch_bounds__(256) void rmse_dense_bwd_pairs_kernel(int64_t ld, const int32_t* __restrict__ xu, const int32_t* __restrict__ xi,
                                                                  const float* __restrict__ y, int64_t n_pairs,
                                                                  const double* __restrict__ st, const float* __restrict__ gl,
                                                                  float* __restrict__ dx)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const float c = (float)((double)gl[0] * st[15]);
    dx[(int64_t)xu[p] * ld + xi[p]] -= c * y[p];                           // (duplicates were summed at upload: one entry per cell)
}

// ---- the dense losses WITHOUT the dense prediction: dot-product scores are bilinear, p_ui = x_u . y_i with x_u = [u | b_u | 1] and
// y_i = [v_i | 1 | b_i], so the two sums the dense passes take over all n_users x n_items predictions are
//     sum p = (sum_u x_u) . (sum_i y_i)          sum p^2 = sum_ui (x_u^T y_i)^2 = <X^T X, Y^T Y>_F
// -- two D x D Gram matrices (D = d + 2) in double instead of U * I predictions: 1M x 1M in milliseconds, where the [U, I]
// tensor (4 TB) cannot exist.  Likewise backward: every cell's gradient is affine in its prediction (A p + B), hence
//     dX = A X (Y^T Y) + B 1 (sum y)^T,   dY = A Y (X^T X) + B 1 (sum x)^T,
// plus the corrections at the interaction cells, which flow through the serial predictions.

// G[D, D] += X^T X over a slice of rows (double accumulation; X float [n, ld]); one workgroup = one 32 x 32 tile of G x one slice
__global__ __launch_bounds__(256) void gram_f64_kernel(const float* __restrict__ X, int64_t n, int D, int64_t ld,
                                                      int64_t rows_per_slice, double* __restrict__ G)
{
    __shared__ float As[32][33], Bs[32][33];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj < ti) return;                                  // G is symmetric: the upper triangle of tiles is computed, mirrored below
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_slice;
    const int64_t r1 = r0 + rows_per_slice < n ? r0 + rows_per_slice : n;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int64_t rb = r0; rb < r1; rb += 32) {
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int rr = e >> 5, cc = e & 31;
            const int64_t row = rb + rr;
            const int ca = ti * 32 + cc, cb = tj * 32 + cc;
            As[rr][cc] = (row < r1 && ca < D) ? X[row * ld + ca] : 0.f;
            Bs[rr][cc] = (row < r1 && cb < D) ? X[row * ld + cb] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) {
            const double a0 = (double)As[rr][2 * ty], a1 = (double)As[rr][2 * ty + 1];
            const double b0 = (double)Bs[rr][2 * tx], b1 = (double)Bs[rr][2 * tx + 1];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gi = ti * 32 + 2 * ty + i, gj = tj * 32 + 2 * tx + j;
            if (gi < D && gj < D) {
                atomicAdd(G + (int64_t)gi * D + gj, acc[i][j]);
                if (tj != ti) atomicAdd(G + (int64_t)gj * D + gi, acc[i][j]);
            }
        }
}

// st[0] += sum p, st[1] += sum (p - mu)^2 over n predictions whose plain sums are m = {sum p, sum p^2}
__global__ void factored_moments_add_kernel(const double* __restrict__ m, double n, const double* __restrict__ mu_ptr,
                                            double* __restrict__ st)
{
    const double mu = mu_ptr ? mu_ptr[0] : 0.0;
    st[0] += m[0];
    st[1] += m[1] - 2.0 * mu * m[0] + n * mu * mu;
}

// backward of the factored forms: d loss / d p = A p + B on EVERY cell (coef[0] = A, coef[1] = B), and the correction of the
// interaction cells as the gradient of their serial predictions
//   RMSE dense:        A = gl / (loss n), B = 0;                cell correction  -A y
//   separation dense:  every cell as a negative;                 cell correction  g_pos(p) - g_neg(p) for y > 0
__global__ __launch_bounds__(256) void factored_bwd_kernel(int kind, const float* __restrict__ serial, const float* __restrict__ y,
                                                          int64_t n_pairs, const double* __restrict__ st, const float* __restrict__ gl,
                                                          float* __restrict__ d_serial, double* __restrict__ coef)
{
    const double g = (double)gl[0];
    const double np = st[2], nn = st[10] - st[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (kind == 2) { coef[0] = g * st[15]; coef[1] = 0.0; }
        else {
            const double s2 = st[13] * st[13];
            coef[0] = g * st[15] / nn * st[14] / s2;
            coef[1] = g * st[15] / nn * (-1.0 - st[14] * st[12] / s2);
        }
    }
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    if (kind == 2) d_serial[p] = (float)(-(g * st[15]) * (double)y[p]);
    else d_serial[p] = (y[p] > 0.f) ? sep_grad((double)serial[p], true, st, np, nn, g) - sep_grad((double)serial[p], false, st, np, nn, g) : 0.f;
}

unsigned grid_for(int64_t n, int per_thread)
{
    int64_t b = ceil_div64(n, 256 * (int64_t)per_thread);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// kind 0: SeparationLossGraph (serial: pred_serial [n_pairs], values [n_pairs]);
// kind 1: SeparationDenseLossGraph (pred [rows, cols] row-major, contiguous; xu / xi / values: the interactions, one entry per cell);
// kind 2: RMSEDenseLossGraph (same inputs as kind 1).
// st: double[16] workspace (kept for the backward pass), loss: float[1].
//
// trec_dense_loss_fwd_phase: the same forward pass cut where a user-sharded (data-parallel) fit has to add the other ranks' sums
// (tensorrec/tensorrec.py:199-217 is the axis the users are split on; the loss is ONE scalar over the union batch):
//   phase 0: zero st, pass 1 (counts, sums; RMSE dense: everything)              -> caller all-reduces st[0..9] (SUM)
//   phase 1: means from st with n_all_total predictions, pass 2 (centred moments) -> caller all-reduces st[0..9] (SUM)   [separation only]
//   phase 2: the loss and the backward coefficients from st (n_all_total for RMSE dense)
// n_all_total: the number of dense predictions over ALL ranks (ignored by kind 0).  The backward pass needs no change: every
// coefficient it reads is in st.
extern "C" int trec_dense_loss_fwd_phase(int32_t kind, int32_t phase, const float* pred, int64_t rows, int64_t cols, const int32_t* xu,
                                         const int32_t* xi, const float* values, int64_t n_pairs, int64_t n_all_total, double* st,
                                         float* loss, void* stream)
{
    TREC_REQUIRE(kind >= 0 && kind <= 2 && phase >= 0 && phase <= 2 && pred && st && loss && (values || n_pairs == 0),
                 "trec_dense_loss_fwd_phase: bad arguments");
    TREC_REQUIRE(kind == 0 || ((xu && xi) || n_pairs == 0), "trec_dense_loss_fwd_phase: the dense forms need the interaction indices");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_all = rows * cols;
    const float* dense = kind == 0 ? (const float*)nullptr : pred;
    const float* serial = kind == 0 ? pred : (const float*)nullptr;
    if (phase == 0) {
        if (hipMemsetAsync(st, 0, 16 * sizeof(double), s) != hipSuccess) { trec_set_last_error("trec_dense_loss_fwd: memset failed"); return TREC_ERR_LAUNCH; }
        if (kind != 0 && n_all) hipLaunchKernelGGL(dense_moments_kernel, dim3(grid_for(n_all, 16)), dim3(256), 0, s, pred, rows, cols, cols, (const double*)nullptr, st);
        if (n_pairs) hipLaunchKernelGGL(pair_moments_kernel, dim3(grid_for(n_pairs, 4)), dim3(256), 0, s, dense, cols, serial, xu, xi, values, n_pairs, (const double*)nullptr, st + 2);
        return trec_check_launch("trec_dense_loss_fwd (pass 1)");
    }
    if (phase == 1) {
        if (kind == 2) return TREC_OK;
        // ---- means, then pass 2: centred second moments (tf.nn.moments: mean of squared differences from the mean)
        hipLaunchKernelGGL(separation_means_kernel, dim3(1), dim3(1), 0, s, st, kind == 1 ? (double)n_all_total : -1.0);
        if (kind == 1 && n_all) hipLaunchKernelGGL(dense_moments_kernel, dim3(grid_for(n_all, 16)), dim3(256), 0, s, pred, rows, cols, cols, (const double*)(st + 12), st);
        if (n_pairs) hipLaunchKernelGGL(pair_moments_kernel, dim3(grid_for(n_pairs, 4)), dim3(256), 0, s, dense, cols, serial, xu, xi, values, n_pairs, (const double*)(st + 11), st + 2);
        return trec_check_launch("trec_dense_loss_fwd (pass 2)");
    }
    if (kind == 2) hipLaunchKernelGGL(rmse_dense_finish_kernel, dim3(1), dim3(1), 0, s, st, loss, (double)n_all_total);
    else hipLaunchKernelGGL(separation_finish_kernel, dim3(1), dim3(1), 0, s, st, loss);
    return trec_check_launch("trec_dense_loss_fwd (finish)");
}

extern "C" int trec_dense_loss_fwd(int32_t kind, const float* pred, int64_t rows, int64_t cols, const int32_t* xu, const int32_t* xi,
                                   const float* values, int64_t n_pairs, double* st, float* loss, void* stream)
{
    for (int phase = 0; phase < 3; ++phase) {
        const int rc = trec_dense_loss_fwd_phase(kind, phase, pred, rows, cols, xu, xi, values, n_pairs, rows * cols, st, loss, stream);
        if (rc != TREC_OK) return rc;
    }
    return TREC_OK;
}

// d loss / d pred, same shapes as trec_dense_loss_fwd's pred; st from the forward pass, gl: float[1] upstream gradient
extern "C" int trec_dense_loss_bwd(int32_t kind, const float* pred, int64_t rows, int64_t cols, const int32_t* xu, const int32_t* xi,
                                   const float* values, int64_t n_pairs, const double* st, const float* gl, float* d_pred,
                                   void* stream)
{
    TREC_REQUIRE(kind >= 0 && kind <= 2 && pred && st && gl && d_pred, "trec_dense_loss_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_all = rows * cols;
    if (kind == 0) {
        if (n_pairs) hipLaunchKernelGGL(separation_bwd_serial_kernel, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0, s, pred, values, n_pairs, st, gl, d_pred);
    } else if (kind == 1) {
        hipLaunchKernelGGL(separation_bwd_dense_kernel, dim3(grid_for(n_all, 8)), dim3(256), 0, s, pred, n_all, st, gl, d_pred);
        if (n_pairs) hipLaunchKernelGGL(separation_bwd_pairs_kernel, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0, s, pred, cols, xu, xi, values, n_pairs, st, gl, d_pred);
    } else {
        hipLaunchKernelGGL(rmse_dense_bwd_kernel, dim3(grid_for(n_all, 8)), dim3(256), 0, s, pred, n_all, st, gl, d_pred);
        if (n_pairs) hipLaunchKernelGGL(rmse_dense_bwd_pairs_kernel, dim3((unsigned)ceil_div64(n_pairs, 256)), dim3(256), 0, s, cols, xu, xi, values, n_pairs, st, gl, d_pred);
    }
    return trec_check_launch("trec_dense_loss_bwd");
}

// G[D, D] (double, row-major) = X^T X for X float [n, D] with row stride ld; G is cleared here.  With a column of ones in X the
// matching row of G holds the column sums.  D <= 1024.
extern "C" int trec_gram_f64(const float* X, int64_t n, int32_t D, int64_t ld, double* G, void* stream)
{
    TREC_REQUIRE(X && G && n >= 0 && D >= 1 && D <= 1024 && ld >= D, "trec_gram_f64: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(G, 0, (size_t)D * D * sizeof(double), s) != hipSuccess) { trec_set_last_error("trec_gram_f64: memset failed"); return TREC_ERR_LAUNCH; }
    if (n == 0) return TREC_OK;
    const int tiles = (D + 31) / 32;
    // ~2,048 workgroups over the upper triangle of tiles; slices of whole 32-row blocks
    int64_t slices = 2048 / ((int64_t)tiles * (tiles + 1) / 2);
    if (slices < 1) slices = 1;
    int64_t per = ceil_div64(ceil_div64(n, slices), 32) * 32;
    if (per < 32) per = 32;
    slices = ceil_div64(n, per);
    TREC_REQUIRE(slices <= 65535, "trec_gram_f64: too many row slices");
    hipLaunchKernelGGL(gram_f64_kernel, dim3((unsigned)tiles, (unsigned)tiles, (unsigned)slices), dim3(256), 0, s, X, n, (int)D, ld, per, G);
    return trec_check_launch("trec_gram_f64");
}

// trec_dense_loss_fwd_phase for the factored dense forms (kind 1 / 2 only): the dense sums come from m = {sum p, sum p^2} over this
// rank's n_all_local predictions (Gram matrices, see above) instead of a pass over a [rows, cols] tensor; the interactions enter
// through their SERIAL predictions.  Phases and all-reduce points exactly as trec_dense_loss_fwd_phase.
extern "C" int trec_dense_loss_factored_phase(int32_t kind, int32_t phase, const double* m, const float* pred_serial,
                                              const float* values, int64_t n_pairs, int64_t n_all_local, int64_t n_all_total,
                                              double* st, float* loss, void* stream)
{
    TREC_REQUIRE((kind == 1 || kind == 2) && phase >= 0 && phase <= 2 && m && st && loss && ((pred_serial && values) || n_pairs == 0),
                 "trec_dense_loss_factored_phase: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (phase == 0) {
        if (hipMemsetAsync(st, 0, 16 * sizeof(double), s) != hipSuccess) { trec_set_last_error("trec_dense_loss_factored_phase: memset failed"); return TREC_ERR_LAUNCH; }
        hipLaunchKernelGGL(factored_moments_add_kernel, dim3(1), dim3(1), 0, s, m, (double)n_all_local, (const double*)nullptr, st);
        if (n_pairs) hipLaunchKernelGGL(pair_moments_kernel, dim3(grid_for(n_pairs, 4)), dim3(256), 0, s, (const float*)nullptr, (int64_t)0, pred_serial,
                                        (const int32_t*)nullptr, (const int32_t*)nullptr, values, n_pairs, (const double*)nullptr, st + 2);
        return trec_check_launch("trec_dense_loss_factored_phase (pass 1)");
    }
    if (phase == 1) {
        if (kind == 2) return TREC_OK;
        hipLaunchKernelGGL(separation_means_kernel, dim3(1), dim3(1), 0, s, st, (double)n_all_total);
        hipLaunchKernelGGL(factored_moments_add_kernel, dim3(1), dim3(1), 0, s, m, (double)n_all_local, (const double*)(st + 12), st);
        if (n_pairs) hipLaunchKernelGGL(pair_moments_kernel, dim3(grid_for(n_pairs, 4)), dim3(256), 0, s, (const float*)nullptr, (int64_t)0, pred_serial,
                                        (const int32_t*)nullptr, (const int32_t*)nullptr, values, n_pairs, (const double*)(st + 11), st + 2);
        return trec_check_launch("trec_dense_loss_factored_phase (pass 2)");
    }
    if (kind == 2) hipLaunchKernelGGL(rmse_dense_finish_kernel, dim3(1), dim3(1), 0, s, st, loss, (double)n_all_total);
    else hipLaunchKernelGGL(separation_finish_kernel, dim3(1), dim3(1), 0, s, st, loss);
    return trec_check_launch("trec_dense_loss_factored_phase (finish)");
}

// backward of the factored forms: coef (double[2]) = {A, B} of d loss / d p = A p + B on every cell, d_serial [n_pairs] = the
// corrections at the interaction cells (gradient of their serial predictions); gl: float[1] upstream gradient
extern "C" int trec_dense_loss_factored_bwd(int32_t kind, const float* pred_serial, const float* values, int64_t n_pairs,
                                            const double* st, const float* gl, float* d_serial, double* coef, void* stream)
{
    TREC_REQUIRE((kind == 1 || kind == 2) && st && gl && coef && ((pred_serial && values && d_serial) || n_pairs == 0),
                 "trec_dense_loss_factored_bwd: bad arguments");
    const int64_t blocks = n_pairs ? ceil_div64(n_pairs, 256) : 1;
    hipLaunchKernelGGL(factored_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (int)kind, pred_serial, values,
                       n_pairs, st, gl, d_serial, coef);
    return trec_check_launch("trec_dense_loss_factored_bwd");
}
