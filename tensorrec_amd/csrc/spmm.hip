// tensorrec_amd/csrc/spmm.hip -- K1: CSR gather-SpMM and the row-wise companions.
//
// Replaces tf.sparse_tensor_dense_matmul at tensorrec/representation_graphs.py:40 (Linear),
// :119 (ReLU layer 1) and tensorrec/recommendation_graphs.py:15 (bias projection), the row
// tf.nn.l2_normalize at representation_graphs.py:57 / prediction_graphs.py:68-69 /
// recommendation_graphs.py:119-120, and tf.nn.relu(tf.add(..)) at representation_graphs.py:119-120.
// The backward of the SpMM w.r.t. the weights (a dense [F,d] gradient in TF) is the same
// kernel run on the transposed CSR, so it is deterministic and atomics-free.
//
// HBM-bound: every output row is a gather of nnz(row) weight rows (d*4 B each) plus one row
// write.  A "subgroup" of LPR = pow2(d/4) lanes owns R consecutive rows; each lane keeps
// float4 accumulators and the first non-zero of all R rows is fetched as one batch so that a
// wave has R*64/LPR independent weight-row gathers in flight (identity features = 1 nnz/row).
// Accumulation order is CSR order with one fmaf per element == oracle/tr_oracle.c:orc_spmm_csr.
#include "common.hpp"

#define L2NORM_EPS 1e-12f

template <int LOG2W>
__device__ __forceinline__ float subgroup_sum(float v, int lpr) {
    // butterfly over the lpr lanes of this subgroup (lpr is a power of two <= 64)
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// EPI: 0 none | 1 row L2-normalise (writes inv norm) | 2 + col_bias, ReLU | 3 none, and the row sums of the VALUES go
// to out_inv (the bias gradient that accompanies a gradient gather: g summed per row, for free)
// NT: non-temporal output stores (rows are written once); NTL: non-temporal weight-row loads as well (embedding-style
// gathers where every weight row is referenced about once, so caching it only evicts useful lines)
// PACKED: `indices` points at int2 {column, value bits} entries (one 8-byte load per non-zero; values / val_perm unused)
// -- the layout the counting sort of the fit step writes with ONE scattered store per pair
// EPI 4 (the operand of the filtered top-k straight from the gather, csrc/topk_filter.hip): besides the fp32 row, its
// bf16 image [n_rows, d], {||row||, ||row - bf16(row)||} and the running maxima of both (what trec_score_prep_filter
// computes in a separate pass over the representation)
struct SpmmFilterOut {
    unsigned short* bf16;
    float2* row_stats;
    float* gstats;
};

template <int ITERS, int R, int EPI, bool NT = false, bool NTL = false, bool PACKED = false>
__global__ __launch_bounds__(256) void spmm_csr_vec4_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int32_t* __restrict__ val_perm, int64_t n_rows, const float* __restrict__ W, int d, int lpr_log2,
    const float* __restrict__ col_bias, int accumulate, float* __restrict__ out, float* __restrict__ out_inv,
    SpmmFilterOut fo = SpmmFilterOut{nullptr, nullptr, nullptr})
{
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t sg = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2;
    const int64_t row0 = sg * R;
    if (row0 >= n_rows) return;

    int col[ITERS];
    bool cvalid[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        col[it] = (it * lpr + sub_lane) * 4;
        cvalid[it] = col[it] < d;
    }

    f32x4 acc[R][ITERS];
    float vsum[R];
    int64_t s[R], e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        const bool v = row < n_rows;
        s[r] = v ? indptr[row] : 0;
        e[r] = v ? indptr[row + 1] : 0;
        vsum[r] = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            acc[r][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (accumulate && v && cvalid[it]) acc[r][it] = *(const f32x4*)(out + row * (int64_t)d + col[it]);
        }
    }
    // batch the first non-zero of the R rows: R independent index loads, then R independent gathers
    int32_t c0[R];
    float v0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        c0[r] = 0; v0[r] = 0.f;
        if (s[r] < e[r]) {
            if (PACKED) {
                const int2 en = ((const int2*)indices)[s[r]];
                c0[r] = en.x; v0[r] = __int_as_float(en.y);
            } else {
                c0[r] = indices[s[r]];
                v0[r] = values[val_perm ? (int64_t)val_perm[s[r]] : s[r]];
            }
        }
    }
    f32x4 x0[R][ITERS];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            x0[r][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s[r] < e[r] && cvalid[it]) {
                const f32x4* src = (const f32x4*)(W + (int64_t)c0[r] * d + col[it]);
                x0[r][it] = NTL ? __builtin_nontemporal_load(src) : *src;
            }
        }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (s[r] < e[r]) {
            if (EPI == 3) vsum[r] += v0[r];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                acc[r][it].x = fmaf(v0[r], x0[r][it].x, acc[r][it].x);
                acc[r][it].y = fmaf(v0[r], x0[r][it].y, acc[r][it].y);
                acc[r][it].z = fmaf(v0[r], x0[r][it].z, acc[r][it].z);
                acc[r][it].w = fmaf(v0[r], x0[r][it].w, acc[r][it].w);
            }
        }
    // remaining non-zeros, two gathers in flight per row
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t j = s[r] + 1;
        for (; j + 1 < e[r]; j += 2) {
            int32_t ca, cb;
            float va, vb;
            if (PACKED) {
                const int2 ea = ((const int2*)indices)[j], eb = ((const int2*)indices)[j + 1];
                ca = ea.x; va = __int_as_float(ea.y); cb = eb.x; vb = __int_as_float(eb.y);
            } else {
                ca = indices[j]; cb = indices[j + 1];
                va = values[val_perm ? (int64_t)val_perm[j] : j];
                vb = values[val_perm ? (int64_t)val_perm[j + 1] : j + 1];
            }
            if (EPI == 3) { vsum[r] += va; vsum[r] += vb; }
            f32x4 xa[ITERS], xb[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                xa[it] = xb[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (cvalid[it]) {
                    xa[it] = *(const f32x4*)(W + (int64_t)ca * d + col[it]);
                    xb[it] = *(const f32x4*)(W + (int64_t)cb * d + col[it]);
                }
            }
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                acc[r][it].x = fmaf(vb, xb[it].x, fmaf(va, xa[it].x, acc[r][it].x));
                acc[r][it].y = fmaf(vb, xb[it].y, fmaf(va, xa[it].y, acc[r][it].y));
                acc[r][it].z = fmaf(vb, xb[it].z, fmaf(va, xa[it].z, acc[r][it].z));
                acc[r][it].w = fmaf(vb, xb[it].w, fmaf(va, xa[it].w, acc[r][it].w));
            }
        }
        if (j < e[r]) {
            int32_t ca;
            float va;
            if (PACKED) {
                const int2 ea = ((const int2*)indices)[j];
                ca = ea.x; va = __int_as_float(ea.y);
            } else {
                ca = indices[j];
                va = values[val_perm ? (int64_t)val_perm[j] : j];
            }
            if (EPI == 3) vsum[r] += va;
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if (cvalid[it]) {
                    const f32x4 xa = *(const f32x4*)(W + (int64_t)ca * d + col[it]);
                    acc[r][it].x = fmaf(va, xa.x, acc[r][it].x);
                    acc[r][it].y = fmaf(va, xa.y, acc[r][it].y);
                    acc[r][it].z = fmaf(va, xa.z, acc[r][it].z);
                    acc[r][it].w = fmaf(va, xa.w, acc[r][it].w);
                }
        }
    }
    // epilogue + store
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        if (row >= n_rows) continue;
        if (EPI == 1) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if (cvalid[it])
                    ss += acc[r][it].x * acc[r][it].x + acc[r][it].y * acc[r][it].y +
                          acc[r][it].z * acc[r][it].z + acc[r][it].w * acc[r][it].w;
            ss = subgroup_sum<0>(ss, lpr);
            const float inv = 1.0f / sqrtf(fmaxf(ss, L2NORM_EPS));
#pragma unroll
            for (int it = 0; it < ITERS; ++it) acc[r][it] *= inv;
            if (out_inv && sub_lane == 0) out_inv[row] = inv;
        } else if (EPI == 3) {
            if (out_inv && sub_lane == 0) out_inv[row] = accumulate ? out_inv[row] + vsum[r] : vsum[r];
        } else if (EPI == 4) {
            float sw = 0.f, se = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if (cvalid[it]) {
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2_bits(acc[r][it][0], acc[r][it][1]);
                    pk.y = f32x2_to_bf16x2_bits(acc[r][it][2], acc[r][it][3]);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float w = acc[r][it][e4];
                        const unsigned int word = (e4 < 2) ? pk.x : pk.y;
                        const float back = __uint_as_float((e4 & 1) ? (word & 0xffff0000u) : (word << 16));
                        const float err = w - back;                         // exact: the discarded low bits
                        sw = fmaf(w, w, sw);
                        se = fmaf(err, err, se);
                    }
                    if (NT) __builtin_nontemporal_store(pk.x | ((unsigned long long)pk.y << 32),
                                                        (unsigned long long*)(fo.bf16 + row * (int64_t)d + col[it]));
                    else *(uint2*)(fo.bf16 + row * (int64_t)d + col[it]) = pk;
                }
            int writer = 0;
            if (lpr == 32) {
                // 32-lane rows (d = 128): DPP adds -- four rotations inside the 16-lane rows, then row_bcast15 carries the
                // first row's total into the second; the sum is complete in lane 31 (no LDS crossbar, cf. wmrb_fused.hip)
                auto dpp_sum32 = [](float x) {
                    int v, r;
                    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); x += __int_as_float(r);
                    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); x += __int_as_float(r);
                    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); x += __int_as_float(r);
                    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); x += __int_as_float(r);
                    v = __float_as_int(x); r = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); x += __int_as_float(r);
                    return x;
                };
                sw = dpp_sum32(sw);
                se = dpp_sum32(se);
                writer = 31;
            } else {
                sw = subgroup_sum<0>(sw, lpr);
                se = subgroup_sum<0>(se, lpr);
            }
            if (sub_lane == writer) {
                const float nw = sqrtf(sw), ne = sqrtf(se);
                fo.row_stats[row] = make_float2(nw, ne);
                if (fo.gstats) {      // guarded atomic max (non-negative floats order like their bit patterns; NaN -> +inf)
                    const unsigned int bw = __float_as_uint(nw == nw ? nw : INFINITY), be = __float_as_uint(ne == ne ? ne : INFINITY);
                    if (bw > *(volatile unsigned int*)(fo.gstats + 0)) atomicMax((unsigned int*)(fo.gstats + 0), bw);
                    if (be > *(volatile unsigned int*)(fo.gstats + 1)) atomicMax((unsigned int*)(fo.gstats + 1), be);
                }
            }
        } else if (EPI == 2) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if (cvalid[it]) {
                    const f32x4 b = *(const f32x4*)(col_bias + col[it]);
                    acc[r][it].x = fmaxf(acc[r][it].x + b.x, 0.f);
                    acc[r][it].y = fmaxf(acc[r][it].y + b.y, 0.f);
                    acc[r][it].z = fmaxf(acc[r][it].z + b.z, 0.f);
                    acc[r][it].w = fmaxf(acc[r][it].w + b.w, 0.f);
                }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            if (cvalid[it]) {
                f32x4* dst = (f32x4*)(out + row * (int64_t)d + col[it]);
                if (NT) __builtin_nontemporal_store(acc[r][it], dst);      // written once, not re-read by this kernel
                else *dst = acc[r][it];
            }
    }
}

// generic fallback: any d (d % 4 != 0 or d > 1024); one thread per output element
__global__ __launch_bounds__(256) void spmm_csr_scalar_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int32_t* __restrict__ val_perm, int64_t n_rows, const float* __restrict__ W, int d, int accumulate,
    float* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * d) return;
    const int64_t row = idx / d;
    const int c = (int)(idx - row * d);
    float acc = accumulate ? out[idx] : 0.f;
    for (int64_t j = indptr[row]; j < indptr[row + 1]; ++j) {
        const float v = values[val_perm ? (int64_t)val_perm[j] : j];
        acc = fmaf(v, W[(int64_t)indices[j] * d + c], acc);
    }
    out[idx] = acc;
}

// one wave per row: y = x * 1/sqrt(max(sum x^2, eps))  (tf.nn.l2_normalize(x, 1) [external])
__global__ __launch_bounds__(256) void row_l2norm_fwd_kernel(const float* __restrict__ x, int64_t n_rows, int d,
                                                            float* __restrict__ y, float* __restrict__ inv_out)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    const int lane = lane_id();
    const float* xr = x + row * (int64_t)d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss = fmaf(xr[c], xr[c], ss);
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.0f / sqrtf(fmaxf(ss, L2NORM_EPS));
    for (int c = lane; c < d; c += 64) y[row * (int64_t)d + c] = xr[c] * inv;
    if (inv_out && lane == 0) inv_out[row] = inv;
}

// dx = (dy - y * sum(y*dy)) * inv ; where the norm was clamped (inv == 1/sqrt(eps)) dx = dy * inv
__global__ __launch_bounds__(256) void row_l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv,
                                                            const float* __restrict__ dy, int64_t n_rows, int d,
                                                            float* __restrict__ dx)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    const int lane = lane_id();
    const float* yr = y + row * (int64_t)d;
    const float* gr = dy + row * (int64_t)d;
    float t = 0.f;
    for (int c = lane; c < d; c += 64) t = fmaf(yr[c], gr[c], t);
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    const float iv = inv[row];
    const float inv_eps = 1.0f / sqrtf(L2NORM_EPS);
    if (iv >= inv_eps) t = 0.f;
    for (int c = lane; c < d; c += 64) dx[row * (int64_t)d + c] = (gr[c] - yr[c] * t) * iv;
}

__global__ __launch_bounds__(256) void bias_relu_kernel(float* __restrict__ x, const float* __restrict__ col_bias,
                                                       int64_t n, int d)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    x[idx] = fmaxf(x[idx] + col_bias[idx % d], 0.f);
}

// d_pre = d_out * (out > 0)   (gradient of tf.nn.relu)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                      int64_t n, float* __restrict__ dpre)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    dpre[idx] = out[idx] > 0.f ? dout[idx] : 0.f;
}

// column sums of a row-major [n_rows, d] matrix (gradient of the broadcast ReLU bias).
// grid = (column tiles of 64, row slices): each block walks its slice of the rows with 4 waves and combines them through
// LDS; with more than one slice the per-slice sums go to a workspace and colsum_finish_kernel adds them in slice order
// (deterministic).  One slice (no workspace) is the whole job in one launch -- d / 64 workgroups, fine for short
// matrices only: 16 workgroups took 8.7 ms over a 138k x 1024 matrix that the chip reads in 0.1 ms.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t n_rows, int d,
                                                    int64_t rows_per_slice, float* __restrict__ out)
{
    __shared__ float part[4][64];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t r1 = r0 + rows_per_slice < n_rows ? r0 + rows_per_slice : n_rows;
    float acc = 0.f;
    if (c < d)
        for (int64_t r = r0 + w; r < r1; r += 4) acc += x[r * (int64_t)d + c];
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < d) out[(int64_t)blockIdx.y * d + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int n_slices, int d,
                                                           float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    float acc = 0.f;
    for (int s = 0; s < n_slices; ++s) acc += partial[(int64_t)s * d + c];
    out[c] = acc;
}

// b[r] = sum_j X[r,j] * beta[j]   (project_biases, recommendation_graphs.py:4-19), fmaf in CSR order
__global__ __launch_bounds__(256) void spmv_csr_kernel(const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices,
                                                      const float* __restrict__ values,
                                                      const int32_t* __restrict__ val_perm, int64_t n_rows,
                                                      const float* __restrict__ beta, float* __restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    float acc = 0.f;
    for (int64_t j = indptr[row]; j < indptr[row + 1]; ++j)
        acc = fmaf(values[val_perm ? (int64_t)val_perm[j] : j], beta[indices[j]], acc);
    out[row] = acc;
}

// beta == NULL: plain row sums of the (permuted) values -- the bias gradients of the serial predictions (g summed per
// user / per item).  Rows there are long (S samples per user, ~S per item), so 16 lanes share a row with coalesced
// strided reads instead of one thread walking it.
__global__ __launch_bounds__(256) void segsum_csr_kernel(const int64_t* __restrict__ indptr,
                                                        const float* __restrict__ values,
                                                        const int32_t* __restrict__ val_perm, int64_t n_rows,
                                                        float* __restrict__ out, int accumulate = 0)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= n_rows) return;
    const int sub = threadIdx.x & 15;
    const int64_t b = indptr[row], e = indptr[row + 1];
    float acc = 0.f;
    for (int64_t j = b + sub; j < e; j += 16) acc += values[val_perm ? (int64_t)val_perm[j] : j];
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) out[row] = accumulate ? out[row] + acc : acc;
}

// dense[r, c] = X[r, c]  (tf.sparse_tensor_to_dense, representation_graphs.py:74); out pre-zeroed by caller
__global__ __launch_bounds__(256) void csr_to_dense_kernel(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const float* __restrict__ values, int64_t n_rows, int n_cols,
                                                          float* __restrict__ out)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    for (int64_t j = indptr[row] + lane_id(); j < indptr[row + 1]; j += 64)
        out[row * (int64_t)n_cols + indices[j]] = values[j];
}

// ------------------------------------------------------------------------------------------
static int pow2ceil_log2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

template <int EPI>
static int launch_vec4(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* val_perm,
                       int64_t n_rows, const float* W, int d, const float* col_bias, int accumulate, float* out,
                       float* out_inv, int64_t nnz_hint, hipStream_t st,
                       SpmmFilterOut fo = SpmmFilterOut{nullptr, nullptr, nullptr})
{
    const int n4 = d / 4;
    int lpr_log2 = pow2ceil_log2(n4);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int lpr = 1 << lpr_log2;
    const int iters = (n4 + lpr - 1) / lpr;
    // R = 4 rows per subgroup when rows are short (embedding-style gathers), else 1
    const bool short_rows = nnz_hint >= 0 && nnz_hint <= 4 * n_rows && n_rows >= 4096;
    int R = short_rows ? 4 : 1;
    // defaults measured on MI355X (scripts/bench_k1.py): non-temporal stores +7% on 1M x 128 gathers
    // non-temporal weight-row loads only when rows are ~1 non-zero (indicator features: every weight row is read about
    // once); -1 = that heuristic, 0/1 force.  +5% on top (71% of 8 TB/s on 1M x 128)
    int tune_ntl = trec_get_tuning("spmm_ntload", -1);
    if (tune_ntl < 0) tune_ntl = (nnz_hint >= 0 && 4 * nnz_hint <= 5 * n_rows) ? 1 : 0;
    const int tune_nt = trec_get_tuning("spmm_nt", 1);
    const int64_t subgroups = ceil_div64(n_rows, R);
    const int64_t threads = subgroups * lpr;
    const unsigned blocks = (unsigned)ceil_div64(threads, 256);
#define TREC_SPMM_LAUNCH(IT, RR)                                                                                   \
    hipLaunchKernelGGL((spmm_csr_vec4_kernel<IT, RR, EPI>), dim3(blocks), dim3(256), 0, st, indptr, indices, values, \
                       val_perm, n_rows, W, d, lpr_log2, col_bias, accumulate, out, out_inv, fo)
    if (R == 4 && iters == 1 && tune_nt && !accumulate) {
        if (tune_ntl) hipLaunchKernelGGL((spmm_csr_vec4_kernel<1, 4, EPI, true, true>), dim3(blocks), dim3(256), 0, st, indptr, indices, values, val_perm, n_rows, W, d, lpr_log2, col_bias, accumulate, out, out_inv, fo);
        else hipLaunchKernelGGL((spmm_csr_vec4_kernel<1, 4, EPI, true, false>), dim3(blocks), dim3(256), 0, st, indptr, indices, values, val_perm, n_rows, W, d, lpr_log2, col_bias, accumulate, out, out_inv, fo);
    } else if (R >= 4) {
        if (iters == 1) TREC_SPMM_LAUNCH(1, 4);
        else if (iters == 2) TREC_SPMM_LAUNCH(2, 4);
        else TREC_SPMM_LAUNCH(4, 4);
    } else {
        if (iters == 1) TREC_SPMM_LAUNCH(1, 1);
        else if (iters == 2) TREC_SPMM_LAUNCH(2, 1);
        else TREC_SPMM_LAUNCH(4, 1);
    }
#undef TREC_SPMM_LAUNCH
    return trec_check_launch("trec_spmm_csr");
}

// out (+)= X . W for a CSR whose entries are packed int2 {column, value bits} (what trec_group_pairs_by_item writes in
// packed mode); long rows (pairs grouped by item): one row per subgroup.  epilogue 0 or 3 (row sums of the values).
extern "C" int trec_spmm_csr_packed(const int64_t* indptr, const void* entries, int64_t n_rows, const float* W, int32_t d,
                                    int32_t epilogue, int32_t accumulate, float* out, float* out_rowsum, void* stream)
{
    TREC_REQUIRE(indptr && entries && W && out, "trec_spmm_csr_packed: null pointer");
    TREC_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, "trec_spmm_csr_packed: d must be a multiple of 4, <= 1024");
    TREC_REQUIRE(epilogue == 0 || (epilogue == 3 && out_rowsum), "trec_spmm_csr_packed: epilogue 0 or 3 (+ row-sum output)");
    if (n_rows == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int n4 = d / 4;
    int lpr_log2 = pow2ceil_log2(n4);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int lpr = 1 << lpr_log2;
    const int iters = (n4 + lpr - 1) / lpr;
    const unsigned blocks = (unsigned)ceil_div64(n_rows * lpr, 256);
    const int32_t* idx = (const int32_t*)entries;
#define TREC_SPMM_PK(IT, EP)                                                                                          \
    hipLaunchKernelGGL((spmm_csr_vec4_kernel<IT, 1, EP, false, false, true>), dim3(blocks), dim3(256), 0, st, indptr, idx, \
                       nullptr, nullptr, n_rows, W, d, lpr_log2, nullptr, accumulate, out, out_rowsum)
    if (epilogue == 3) { if (iters == 1) TREC_SPMM_PK(1, 3); else if (iters == 2) TREC_SPMM_PK(2, 3); else TREC_SPMM_PK(4, 3); }
    else { if (iters == 1) TREC_SPMM_PK(1, 0); else if (iters == 2) TREC_SPMM_PK(2, 0); else TREC_SPMM_PK(4, 0); }
#undef TREC_SPMM_PK
    return trec_check_launch("trec_spmm_csr_packed");
}

extern "C" int trec_spmm_csr(const int64_t* indptr, const int32_t* indices, const float* values,
                             const int32_t* val_perm, int64_t n_rows, int64_t nnz, const float* W, int32_t d,
                             const float* col_bias, int32_t epilogue, int32_t accumulate, float* out,
                             float* out_inv_norm, void* stream)
{
    TREC_REQUIRE(indptr && W && out, "trec_spmm_csr: null pointer");
    TREC_REQUIRE(nnz == 0 || (indices && values), "trec_spmm_csr: null indices/values with nnz != 0");
    TREC_REQUIRE(d >= 1 && n_rows >= 0, "trec_spmm_csr: bad sizes");
    TREC_REQUIRE(epilogue >= 0 && epilogue <= 3, "trec_spmm_csr: epilogue must be 0, 1, 2 or 3");
    TREC_REQUIRE(epilogue != 2 || col_bias, "trec_spmm_csr: epilogue 2 needs col_bias");
    TREC_REQUIRE(epilogue != 3 || out_inv_norm, "trec_spmm_csr: epilogue 3 needs the row-sum output");
    TREC_REQUIRE(!(accumulate && (epilogue == 1 || epilogue == 2)), "trec_spmm_csr: accumulate cannot be combined with epilogue 1 / 2");
    if (n_rows == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d % 4 == 0 && d <= 1024) {
        if (epilogue == 0) return launch_vec4<0>(indptr, indices, values, val_perm, n_rows, W, d, col_bias, accumulate, out, out_inv_norm, nnz, st);
        if (epilogue == 1) return launch_vec4<1>(indptr, indices, values, val_perm, n_rows, W, d, col_bias, accumulate, out, out_inv_norm, nnz, st);
        if (epilogue == 3) return launch_vec4<3>(indptr, indices, values, val_perm, n_rows, W, d, col_bias, accumulate, out, out_inv_norm, nnz, st);
        return launch_vec4<2>(indptr, indices, values, val_perm, n_rows, W, d, col_bias, accumulate, out, out_inv_norm, nnz, st);
    }
    const int64_t total = n_rows * (int64_t)d;
    hipLaunchKernelGGL(spmm_csr_scalar_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, indptr,
                       indices, values, val_perm, n_rows, W, d, accumulate, out);
    int rc = trec_check_launch("trec_spmm_csr(scalar)");
    if (rc) return rc;
    if (epilogue == 1) {
        hipLaunchKernelGGL(row_l2norm_fwd_kernel, dim3((unsigned)ceil_div64(n_rows * 64, 256)), dim3(256), 0, st, out,
                           n_rows, d, out, out_inv_norm);
        return trec_check_launch("trec_spmm_csr(l2norm)");
    }
    if (epilogue == 3) {
        hipLaunchKernelGGL(segsum_csr_kernel, dim3((unsigned)ceil_div64(n_rows * 16, 256)), dim3(256), 0, st, indptr, values,
                           val_perm, n_rows, out_inv_norm, accumulate);
        return trec_check_launch("trec_spmm_csr(row sums)");
    }
    if (epilogue == 2) {
        hipLaunchKernelGGL(bias_relu_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, out, col_bias,
                           total, d);
        return trec_check_launch("trec_spmm_csr(bias_relu)");
    }
    return TREC_OK;
}

// ---- exactly ONE non-zero per row (identity / indicator features: every user or item is its own feature) -------------
// out[r] = values[r] * W[indices[r]]: the CSR row pointer is the identity, so it is neither read (8 bytes per row and one
// hop of the indptr -> index -> row load chain less) nor needed.  A 32-lane group (d = 128; d / 4 lanes in general, ITERS
// float4 per lane) owns 4 consecutive rows: ONE 16-byte load fetches their four column indices, one their four values.
template <int ITERS>
__global__ __launch_bounds__(256) void spmm_one_per_row_kernel(const int32_t* __restrict__ indices,
                                                              const float* __restrict__ values, int64_t n_rows,
                                                              const float* __restrict__ W, int d, int lpr_log2,
                                                              float* __restrict__ out)
{
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t sg = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2;
    const int64_t row0 = sg * 4;
    if (row0 >= n_rows) return;
    int32_t c[4];
    float v[4];
    if (row0 + 3 < n_rows) {
        const int4 c4 = *(const int4*)(indices + row0);
        const f32x4 v4 = *(const f32x4*)(values + row0);
        c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
        v[0] = v4[0]; v[1] = v4[1]; v[2] = v4[2]; v[3] = v4[3];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = row0 + r < n_rows;
            c[r] = ok ? indices[row0 + r] : 0;
            v[r] = ok ? values[row0 + r] : 0.f;
        }
    }
    f32x4 x[4][ITERS];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int col = (it * lpr + sub_lane) * 4;
            x[r][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (col < d && row0 + r < n_rows)
                x[r][it] = __builtin_nontemporal_load((const f32x4*)(W + (int64_t)c[r] * d + col));
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int col = (it * lpr + sub_lane) * 4;
            if (col < d && row0 + r < n_rows) {
                f32x4 o;                                   // fmaf(v, w, 0) of the oracle's chain == v * w
                o[0] = v[r] * x[r][it][0]; o[1] = v[r] * x[r][it][1]; o[2] = v[r] * x[r][it][2]; o[3] = v[r] * x[r][it][3];
                __builtin_nontemporal_store(o, (f32x4*)(out + (row0 + r) * (int64_t)d + col));
            }
        }
}

extern "C" int trec_spmm_one_per_row(const int32_t* indices, const float* values, int64_t n_rows, const float* W, int32_t d,
                                     float* out, void* stream)
{
    TREC_REQUIRE(indices && values && W && out, "trec_spmm_one_per_row: null pointer");
    TREC_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, "trec_spmm_one_per_row: d must be a multiple of 4 up to 1024");
    TREC_REQUIRE(((uintptr_t)indices % 16) == 0 && ((uintptr_t)values % 16) == 0, "trec_spmm_one_per_row: 16-byte aligned index / value arrays");
    if (n_rows == 0) return TREC_OK;
    const int n4 = d / 4;
    int lpr_log2 = pow2ceil_log2(n4);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int lpr = 1 << lpr_log2;
    const int iters = (n4 + lpr - 1) / lpr;
    const int64_t threads = ceil_div64(n_rows, 4) * lpr;
    const unsigned blocks = (unsigned)ceil_div64(threads, 256);
    hipStream_t st = (hipStream_t)stream;
    if (iters == 1) hipLaunchKernelGGL(spmm_one_per_row_kernel<1>, dim3(blocks), dim3(256), 0, st, indices, values, n_rows, W, d, lpr_log2, out);
    else if (iters == 2) hipLaunchKernelGGL(spmm_one_per_row_kernel<2>, dim3(blocks), dim3(256), 0, st, indices, values, n_rows, W, d, lpr_log2, out);
    else hipLaunchKernelGGL(spmm_one_per_row_kernel<4>, dim3(blocks), dim3(256), 0, st, indices, values, n_rows, W, d, lpr_log2, out);
    return trec_check_launch("trec_spmm_one_per_row");
}

// K1 with the filtered top-k's operand as its epilogue (EPI 4): out = X . W (fp32, the exact operand) AND its bf16 image,
// the per-row {||row||, ||row - bf16(row)||} and -- gstats non-NULL, zero-initialised -- their running maxima.  For
// representations that go into the score kernels as they are (dot products, d = 32 / 64 / 128 / 256: no padding, no
// normalisation); everything else takes trec_score_prep_filter.
extern "C" int trec_spmm_csr_filter(const int64_t* indptr, const int32_t* indices, const float* values, int64_t n_rows,
                                    int64_t nnz, const float* W, int32_t d, float* out, void* out_bf16, float* row_stats,
                                    float* gstats, void* stream)
{
    TREC_REQUIRE(indptr && W && out && out_bf16 && row_stats, "trec_spmm_csr_filter: null pointer");
    TREC_REQUIRE(nnz == 0 || (indices && values), "trec_spmm_csr_filter: null indices/values with nnz != 0");
    TREC_REQUIRE(d == 32 || d == 64 || d == 128 || d == 256, "trec_spmm_csr_filter: d must be 32, 64, 128 or 256");
    if (n_rows == 0) return TREC_OK;
    return launch_vec4<4>(indptr, indices, values, nullptr, n_rows, W, d, nullptr, 0, out, nullptr, nnz, (hipStream_t)stream,
                          SpmmFilterOut{(unsigned short*)out_bf16, (float2*)row_stats, gstats});
}

// running maximum of |x| into *out (non-negative float, zero-initialised by the caller): the |bias| term of the filter's bound
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out)
{
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = fabsf(x[i]);
        m = fmaxf(m, a == a ? a : INFINITY);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *(volatile unsigned int*)out)
        atomicMax((unsigned int*)out, __float_as_uint(m));
}

extern "C" int trec_absmax(const float* x, int64_t n, float* out, void* stream)
{
    TREC_REQUIRE(x && out, "trec_absmax: null pointer");
    if (n == 0) return TREC_OK;
    unsigned blocks = (unsigned)ceil_div64(n, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return trec_check_launch("trec_absmax");
}

extern "C" int trec_row_l2norm_fwd(const float* x, int64_t n_rows, int32_t d, float* y, float* inv_norm, void* stream)
{
    TREC_REQUIRE(x && y && d >= 1, "trec_row_l2norm_fwd: bad arguments");
    if (n_rows == 0) return TREC_OK;
    hipLaunchKernelGGL(row_l2norm_fwd_kernel, dim3((unsigned)ceil_div64(n_rows * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, n_rows, d, y, inv_norm);
    return trec_check_launch("trec_row_l2norm_fwd");
}

extern "C" int trec_row_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int64_t n_rows, int32_t d,
                                   float* dx, void* stream)
{
    TREC_REQUIRE(y && inv_norm && dy && dx && d >= 1, "trec_row_l2norm_bwd: bad arguments");
    if (n_rows == 0) return TREC_OK;
    hipLaunchKernelGGL(row_l2norm_bwd_kernel, dim3((unsigned)ceil_div64(n_rows * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, y, inv_norm, dy, n_rows, d, dx);
    return trec_check_launch("trec_row_l2norm_bwd");
}

extern "C" int trec_relu_bwd(const float* out, const float* dout, int64_t n, float* dpre, void* stream)
{
    TREC_REQUIRE(out && dout && dpre, "trec_relu_bwd: null pointer");
    if (n == 0) return TREC_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, out,
                       dout, n, dpre);
    return trec_check_launch("trec_relu_bwd");
}

extern "C" int trec_colsum(const float* x, int64_t n_rows, int32_t d, float* out, float* workspace, int32_t n_slices,
                           void* stream)
{
    TREC_REQUIRE(x && out && d >= 1, "trec_colsum: bad arguments");
    TREC_REQUIRE(n_slices <= 1 || workspace, "trec_colsum: n_slices > 1 needs a workspace of n_slices * d floats");
    hipStream_t st = (hipStream_t)stream;
    const unsigned tiles = (unsigned)((d + 63) / 64);
    if (n_slices <= 1) {
        hipLaunchKernelGGL(colsum_kernel, dim3(tiles), dim3(256), 0, st, x, n_rows, d, n_rows, out);
        return trec_check_launch("trec_colsum");
    }
    const int64_t rows_per_slice = ceil_div64(n_rows, n_slices);
    hipLaunchKernelGGL(colsum_kernel, dim3(tiles, (unsigned)n_slices), dim3(256), 0, st, x, n_rows, d, rows_per_slice,
                       workspace);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, workspace, n_slices, d,
                       out);
    return trec_check_launch("trec_colsum");
}

extern "C" int trec_spmv_csr(const int64_t* indptr, const int32_t* indices, const float* values,
                             const int32_t* val_perm, int64_t n_rows, const float* beta, float* out, void* stream)
{
    TREC_REQUIRE(indptr && out, "trec_spmv_csr: null pointer");   // indices/values may be NULL when nnz == 0
    if (n_rows == 0) return TREC_OK;
    if (!beta) {              // beta = ones: segmented sum of the values
        hipLaunchKernelGGL(segsum_csr_kernel, dim3((unsigned)ceil_div64(n_rows * 16, 256)), dim3(256), 0,
                           (hipStream_t)stream, indptr, values, val_perm, n_rows, out);
        return trec_check_launch("trec_spmv_csr(segment sum)");
    }
    hipLaunchKernelGGL(spmv_csr_kernel, dim3((unsigned)ceil_div64(n_rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       indptr, indices, values, val_perm, n_rows, beta, out);
    return trec_check_launch("trec_spmv_csr");
}

extern "C" int trec_csr_to_dense(const int64_t* indptr, const int32_t* indices, const float* values, int64_t n_rows,
                                 int32_t n_cols, float* out, void* stream)
{
    TREC_REQUIRE(indptr && out, "trec_csr_to_dense: null pointer");
    if (n_rows == 0) return TREC_OK;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)n_rows * (size_t)n_cols, (hipStream_t)stream);
    if (e != hipSuccess) { trec_set_last_error("trec_csr_to_dense: memset failed"); return TREC_ERR_LAUNCH; }
    hipLaunchKernelGGL(csr_to_dense_kernel, dim3((unsigned)ceil_div64(n_rows * 64, 256)), dim3(256), 0,
                       (hipStream_t)stream, indptr, indices, values, n_rows, n_cols, out);
    return trec_check_launch("trec_csr_to_dense");
}
