// tensorrec_amd/csrc/spmm_split.hip -- K1 for SKEWED row lengths: the segmented gathers of the backward pass.
//
// The gradient gathers of the fit step are K1 on a transposed structure: rows = items (pairs grouped by item,
// tensorrec/prediction_graphs.py:52-55 / :105-117 differentiated) or feature columns (tf.sparse_tensor_dense_matmul
// differentiated, representation_graphs.py:40, :119).  Real interaction data is Zipf-shaped: on a MovieLens-20M-shaped
// problem the most popular item holds ~136,000 of 20M pairs and a demographic indicator column ~14,000 rows, and the
// one-subgroup-per-row kernel of spmm.hip walks such a row serially (measured: 27 ms for a 540k-non-zero matrix, 117 ms
// for the atomic fallback of the Euclidean pairs).  Here rows longer than SPLIT_T non-zeros are cut into chunks of
// SPLIT_T / 2, every chunk is a CSR-order fmaf chain of its own (any subgroup of the chip takes any chunk), and a row's
// chunk partials are added in chunk order -- deterministic, atomics only to hand out chunk slots.
//
//   plan    one thread per row: long rows reserve `nch` consecutive chunk slots (atomicAdd on a counter) and list them
//   rows    rows <= SPLIT_T: the kernel of spmm.hip in its one-row-per-subgroup form (same chain, same bits)
//   chunks  a fixed grid strides over the listed chunks: partial[c, :] and the sum of the chunk's values
//   reduce  per long row: out = (((out?) + p0) + p1) + ...
//
// `own` (nullable): gather (own[row, :] - W[col, :]) instead of W[col, :] -- the Euclidean pair gradient
// dU[u] = sum_p c_p (U[u] - V[i_p]), dV[i] = sum_p c_p (V[i] - U[u_p]) with c_p from trec_pair_euclid_coef.
#include "common.hpp"

namespace {

constexpr int SPLIT_T_DEFAULT = 2048;

struct SplitPlan {
    int32_t* hdr;          // [0] chunks handed out, [1] long rows listed
    int64_t* long_row;     // [max_long]
    int32_t* long_base;    // [max_long]
    int32_t* long_nch;     // [max_long]
    int64_t* chunk_row;    // [max_chunks]
    int32_t* chunk_k;      // [max_chunks]
    float* psum;           // [max_chunks]
    float* partial;        // [max_chunks, d]
    int64_t max_long, max_chunks;
};

static inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

static int64_t plan_layout(int64_t nnz, int d, int split_t, char* base, SplitPlan* p)
{
    // a row of n > T non-zeros takes ceil(n / (T/2)) <= 2n/T + 1 < 3n/T chunks  ->  at most 3 nnz / T chunks in total
    const int64_t max_long = nnz / split_t + 1;
    const int64_t max_chunks = 3 * nnz / split_t + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up16(bytes); return o; };
    const size_t o_hdr = take(16), o_lr = take(8 * max_long), o_lb = take(4 * max_long), o_ln = take(4 * max_long),
                 o_cr = take(8 * max_chunks), o_ck = take(4 * max_chunks), o_ps = take(4 * max_chunks),
                 o_pa = take(4 * (size_t)max_chunks * (size_t)d);
    if (p) {
        p->hdr = (int32_t*)(base + o_hdr); p->long_row = (int64_t*)(base + o_lr); p->long_base = (int32_t*)(base + o_lb);
        p->long_nch = (int32_t*)(base + o_ln); p->chunk_row = (int64_t*)(base + o_cr); p->chunk_k = (int32_t*)(base + o_ck);
        p->psum = (float*)(base + o_ps); p->partial = (float*)(base + o_pa);
        p->max_long = max_long; p->max_chunks = max_chunks;
    }
    return (int64_t)off;
}

__global__ __launch_bounds__(256) void split_plan_kernel(const int64_t* __restrict__ indptr, int64_t n_rows, int split_t,
                                                        SplitPlan p)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t n = indptr[row + 1] - indptr[row];
    if (n <= split_t) return;
    const int chunk = split_t >> 1;
    const int nch = (int)((n + chunk - 1) / chunk);
    const int base = atomicAdd(p.hdr, nch);
    const int li = atomicAdd(p.hdr + 1, 1);
    p.long_row[li] = row; p.long_base[li] = base; p.long_nch[li] = nch;
    for (int k = 0; k < nch; ++k) { p.chunk_row[base + k] = row; p.chunk_k[base + k] = k; }
}

struct Entry { int32_t col; float val; };

// VEC = 4: a lane owns float4 column groups (d % 4 == 0, rows 16-byte aligned); VEC = 1: a lane owns single columns (any d)
template <int VEC> struct VecOf { typedef f32x4 T; };
template <> struct VecOf<1> { typedef float T; };
__device__ __forceinline__ void vfma(f32x4& acc, float v, const f32x4& x)
{
    acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
}
__device__ __forceinline__ void vfma(float& acc, float v, const float& x) { acc = fmaf(v, x, acc); }
template <typename T> __device__ __forceinline__ T vzero();
template <> __device__ __forceinline__ f32x4 vzero<f32x4>() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }

template <bool PACKED>
__device__ __forceinline__ Entry load_entry(const int32_t* __restrict__ indices, const float* __restrict__ values,
                                            const int32_t* __restrict__ val_perm, int64_t j)
{
    Entry en;
    if (PACKED) {
        const int2 e2 = ((const int2*)indices)[j];
        en.col = e2.x; en.val = __int_as_float(e2.y);
    } else {
        en.col = indices[j];
        en.val = values[val_perm ? (int64_t)val_perm[j] : j];
    }
    return en;
}

// acc (+)= sum over non-zeros [j0, j1) of val * x(col), x = W row or (own - W row); CSR order, one fmaf per element.
// Four row gathers in flight, and the NEXT four entries {column, value} are fetched before the current four rows are: an
// iteration used to be two dependent round trips (entries -> rows), i.e. ~4 us per 4 rows of a subgroup whatever the
// bandwidth; with the entries one step ahead it is one (round 4: configs[4]'s item-side gather of 3.7e8 sampled pairs,
// 1 KB rows from a cache-resident table, ran at 3.6 TB/s -- latency-, not bandwidth-bound).  Same chain, same bits.
template <int ITERS, bool PACKED, int VEC, bool PERM>
__device__ __forceinline__ void gather_range_impl(const int32_t* __restrict__ indices, const float* __restrict__ values,
                                             const int32_t* __restrict__ val_perm, int64_t j0, int64_t j1,
                                             const float* __restrict__ W, int d, const int (&col)[ITERS],
                                             const bool (&cvalid)[ITERS], bool diff,
                                             const typename VecOf<VEC>::T (&ownv)[ITERS],
                                             typename VecOf<VEC>::T (&acc)[ITERS], float& vsum)
{
    typedef typename VecOf<VEC>::T V;
    int64_t j = j0;
    // A three-stage pipeline over blocks of four entries, every load UNCONDITIONAL (indices clamped into [j0, j1), columns past
    // d read column 0 and are zeroed): per iteration the value indices (val_perm) and columns of block b + 2, the values of
    // block b + 1 (through the indices fetched an iteration ago) and the four row gathers of block b leave together -- ONE
    // round trip per four rows.  Written entry by entry (`more ? load_entry(...) : ...`, values[val_perm[j]]) each entry was
    // a dependent perm -> value chain behind its own s_waitcnt vmcnt(0): five to six serial round trips per four rows, and
    // the MovieLens-20M-shaped epoch spent 100 of its 193 ms in this loop (profiles/r04b_cfg4_fit_kernel_stats.csv).
    if (j + 3 < j1) {
        const int64_t last = j1 - 1;
        Entry en[4];                                                      // block b: {column, value}
        int32_t nc[4];                                                    // block b + 1: columns ...
        int32_t nv[4];                                                    // ... and where its values are (pair ids fit int32)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j2 = j + 4 + q < last ? j + 4 + q : last;
            if (PACKED) {
                const int2 e2 = ((const int2*)indices)[j + q];
                en[q].col = e2.x; en[q].val = __int_as_float(e2.y);
                nv[q] = (int32_t)j2; nc[q] = 0;
            } else {
                en[q].col = indices[j + q];
                nc[q] = indices[j2];
                nv[q] = PERM ? val_perm[j2] : (int32_t)j2;
            }
        }
        if (!PACKED) {
            int32_t cv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[q] = PERM ? val_perm[j + q] : (int32_t)(j + q);
#pragma unroll
            for (int q = 0; q < 4; ++q) en[q].val = values[cv[q]];
        }
        for (; j + 3 < j1; j += 4) {
            Entry nx[4];
            int32_t fc[4];
            int32_t fv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                 // values of block b + 1, indices of block b + 2
                const int64_t j3 = j + 8 + q < last ? j + 8 + q : last;
                if (PACKED) {
                    const int2 e2 = ((const int2*)indices)[nv[q]];
                    nx[q].col = e2.x; nx[q].val = __int_as_float(e2.y);
                    fv[q] = (int32_t)j3; fc[q] = 0;
                } else {
                    nx[q].col = nc[q];
                    nx[q].val = values[nv[q]];
                    fc[q] = indices[j3];
                    fv[q] = PERM ? val_perm[j3] : (int32_t)j3;
                }
            }
            V x[4][ITERS];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int it = 0; it < ITERS; ++it)
                    x[q][it] = *(const V*)(W + (int64_t)en[q].col * d + (cvalid[it] ? col[it] : 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vsum += en[q].val;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const V xq = cvalid[it] ? x[q][it] : vzero<V>();
                    const V xv = diff ? ownv[it] - xq : xq;
                    vfma(acc[it], en[q].val, xv);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { en[q] = nx[q]; nc[q] = fc[q]; nv[q] = fv[q]; }
        }
    }
    for (; j < j1; ++j) {
        const Entry e1 = load_entry<PACKED>(indices, values, PERM ? val_perm : nullptr, j);
        vsum += e1.val;
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            if (cvalid[it]) {
                const V xr = *(const V*)(W + (int64_t)e1.col * d + col[it]);
                const V xv = diff ? ownv[it] - xr : xr;
                vfma(acc[it], e1.val, xv);
            }
    }
}

// (the loop exists twice -- values through val_perm or in place -- so that no load result meets a constant in a phi)
template <int ITERS, bool PACKED, int VEC>
__device__ __forceinline__ void gather_range(const int32_t* __restrict__ indices, const float* __restrict__ values,
                                             const int32_t* __restrict__ val_perm, int64_t j0, int64_t j1,
                                             const float* __restrict__ W, int d, const int (&col)[ITERS],
                                             const bool (&cvalid)[ITERS], bool diff,
                                             const typename VecOf<VEC>::T (&ownv)[ITERS],
                                             typename VecOf<VEC>::T (&acc)[ITERS], float& vsum)
{
    if (!PACKED && val_perm) gather_range_impl<ITERS, PACKED, VEC, true>(indices, values, val_perm, j0, j1, W, d, col, cvalid, diff, ownv, acc, vsum);
    else gather_range_impl<ITERS, PACKED, VEC, false>(indices, values, val_perm, j0, j1, W, d, col, cvalid, diff, ownv, acc, vsum);
}

// rows of at most split_t non-zeros: one row per subgroup of lpr lanes (longer rows are left to the chunk kernels)
template <int ITERS, bool PACKED, int VEC>
__global__ __launch_bounds__(256) void split_rows_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int32_t* __restrict__ val_perm, int64_t n_rows, const float* __restrict__ W, int d, int lpr_log2,
    const float* __restrict__ own, int accumulate, int split_t, float* __restrict__ out, float* __restrict__ out_rowsum)
{
    typedef typename VecOf<VEC>::T V;
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2;
    if (row >= n_rows) return;
    const int64_t s = indptr[row], e = indptr[row + 1];
    if (e - s > split_t) return;
    int col[ITERS];
    bool cvalid[ITERS];
    V acc[ITERS], ownv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        col[it] = (it * lpr + sub_lane) * VEC;
        cvalid[it] = col[it] < d;
        acc[it] = ownv[it] = vzero<V>();
        if (cvalid[it]) {
            if (accumulate) acc[it] = *(const V*)(out + row * (int64_t)d + col[it]);
            if (own) ownv[it] = *(const V*)(own + row * (int64_t)d + col[it]);
        }
    }
    float vsum = 0.f;
    gather_range<ITERS, PACKED, VEC>(indices, values, val_perm, s, e, W, d, col, cvalid, own != nullptr, ownv, acc, vsum);
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (cvalid[it]) *(V*)(out + row * (int64_t)d + col[it]) = acc[it];
    if (out_rowsum && sub_lane == 0) out_rowsum[row] = accumulate ? out_rowsum[row] + vsum : vsum;
}

template <int ITERS, bool PACKED, int VEC>
__global__ __launch_bounds__(256) void split_chunks_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int32_t* __restrict__ val_perm, const float* __restrict__ W, int d, int lpr_log2,
    const float* __restrict__ own, int split_t, SplitPlan p)
{
    typedef typename VecOf<VEC>::T V;
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t n_sg = ((int64_t)gridDim.x * blockDim.x) >> lpr_log2;
    const int n_chunks = p.hdr[0];
    const int chunk = split_t >> 1;
    int col[ITERS];
    bool cvalid[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) { col[it] = (it * lpr + sub_lane) * VEC; cvalid[it] = col[it] < d; }
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2; c < n_chunks; c += n_sg) {
        const int64_t row = p.chunk_row[c];
        const int64_t j0 = indptr[row] + (int64_t)p.chunk_k[c] * chunk;
        const int64_t re = indptr[row + 1];
        const int64_t j1 = j0 + chunk < re ? j0 + chunk : re;
        V acc[ITERS], ownv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            acc[it] = ownv[it] = vzero<V>();
            if (own && cvalid[it]) ownv[it] = *(const V*)(own + row * (int64_t)d + col[it]);
        }
        float vsum = 0.f;
        gather_range<ITERS, PACKED, VEC>(indices, values, val_perm, j0, j1, W, d, col, cvalid, own != nullptr, ownv, acc, vsum);
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            if (cvalid[it]) *(V*)(p.partial + c * (int64_t)d + col[it]) = acc[it];
        if (sub_lane == 0) p.psum[c] = vsum;
    }
}

template <int ITERS, int VEC>
__global__ __launch_bounds__(256) void split_reduce_kernel(int d, int lpr_log2, int accumulate, SplitPlan p,
                                                          float* __restrict__ out, float* __restrict__ out_rowsum)
{
    typedef typename VecOf<VEC>::T V;
    const int lpr = 1 << lpr_log2;
    const int sub_lane = threadIdx.x & (lpr - 1);
    const int64_t n_sg = ((int64_t)gridDim.x * blockDim.x) >> lpr_log2;
    const int n_long = p.hdr[1];
    for (int64_t li = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log2; li < n_long; li += n_sg) {
        const int64_t row = p.long_row[li];
        const int base = p.long_base[li], nch = p.long_nch[li];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int col = (it * lpr + sub_lane) * VEC;
            if (col >= d) continue;
            V acc = vzero<V>();
            if (accumulate) acc = *(const V*)(out + row * (int64_t)d + col);
            for (int k = 0; k < nch; ++k) acc += *(const V*)(p.partial + (int64_t)(base + k) * d + col);
            *(V*)(out + row * (int64_t)d + col) = acc;
        }
        if (out_rowsum && sub_lane == 0) {
            float s = accumulate ? out_rowsum[row] : 0.f;
            for (int k = 0; k < nch; ++k) s += p.psum[base + k];
            out_rowsum[row] = s;
        }
    }
}

// ---- SpMV / segment sums with the same plan
// short rows, beta != NULL: one thread per row, fmaf chain in CSR order (== spmv_csr_kernel)
__global__ __launch_bounds__(256) void split_spmv_rows_kernel(const int64_t* __restrict__ indptr,
                                                             const int32_t* __restrict__ indices,
                                                             const float* __restrict__ values,
                                                             const int32_t* __restrict__ val_perm, int64_t n_rows,
                                                             const float* __restrict__ beta, int split_t,
                                                             float* __restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    if (e - b > split_t) return;
    float acc = 0.f;
    for (int64_t j = b; j < e; ++j) acc = fmaf(values[val_perm ? (int64_t)val_perm[j] : j], beta[indices[j]], acc);
    out[row] = acc;
}

// short rows, beta == NULL: 16 lanes per row (== segsum_csr_kernel)
__global__ __launch_bounds__(256) void split_segsum_rows_kernel(const int64_t* __restrict__ indptr,
                                                               const float* __restrict__ values,
                                                               const int32_t* __restrict__ val_perm, int64_t n_rows,
                                                               int split_t, float* __restrict__ out)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= n_rows) return;
    const int sub = threadIdx.x & 15;
    const int64_t b = indptr[row], e = indptr[row + 1];
    if (e - b > split_t) return;
    float acc = 0.f;
    for (int64_t j = b + sub; j < e; j += 16) acc += values[val_perm ? (int64_t)val_perm[j] : j];
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) out[row] = acc;
}

// one wave per chunk: lanes stride over the chunk, butterfly at the end
__global__ __launch_bounds__(256) void split_spmv_chunks_kernel(const int64_t* __restrict__ indptr,
                                                               const int32_t* __restrict__ indices,
                                                               const float* __restrict__ values,
                                                               const int32_t* __restrict__ val_perm,
                                                               const float* __restrict__ beta, int split_t, SplitPlan p)
{
    const int lane = lane_id();
    const int64_t n_w = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int n_chunks = p.hdr[0];
    const int chunk = split_t >> 1;
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_chunks; c += n_w) {
        const int64_t row = p.chunk_row[c];
        const int64_t j0 = indptr[row] + (int64_t)p.chunk_k[c] * chunk;
        const int64_t re = indptr[row + 1];
        const int64_t j1 = j0 + chunk < re ? j0 + chunk : re;
        float acc = 0.f;
        for (int64_t j = j0 + lane; j < j1; j += 64) {
            const float v = values[val_perm ? (int64_t)val_perm[j] : j];
            acc = beta ? fmaf(v, beta[indices[j]], acc) : acc + v;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) p.psum[c] = acc;
    }
}

__global__ __launch_bounds__(256) void split_spmv_reduce_kernel(SplitPlan p, float* __restrict__ out)
{
    const int n_long = p.hdr[1];
    for (int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; li < n_long; li += (int64_t)gridDim.x * blockDim.x) {
        const int base = p.long_base[li], nch = p.long_nch[li];
        float s = 0.f;
        for (int k = 0; k < nch; ++k) s += p.psum[base + k];
        out[p.long_row[li]] = s;
    }
}

static int pow2ceil_log2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

static int make_plan(const int64_t* indptr, int64_t n_rows, int64_t nnz, int d, void* workspace, int64_t workspace_bytes,
                     int split_t, SplitPlan* plan, hipStream_t st, const char* who)
{
    const int64_t need = plan_layout(nnz, d, split_t, nullptr, nullptr);
    if (!workspace || workspace_bytes < need) {
        trec_set_last_error("split gather: workspace smaller than trec_csr_split_workspace_bytes()");
        return TREC_ERR_INVALID;
    }
    plan_layout(nnz, d, split_t, (char*)workspace, plan);
    if (hipMemsetAsync(plan->hdr, 0, 16, st) != hipSuccess) { trec_set_last_error("split gather: memset failed"); return TREC_ERR_LAUNCH; }
    hipLaunchKernelGGL(split_plan_kernel, dim3((unsigned)ceil_div64(n_rows, 256)), dim3(256), 0, st, indptr, n_rows, split_t,
                       *plan);
    return trec_check_launch(who);
}

}  // namespace

extern "C" int64_t trec_csr_split_workspace_bytes(int64_t nnz, int32_t d)
{
    if (nnz < 0 || d < 1) return -1;
    return plan_layout(nnz, d, trec_get_tuning("split_t", SPLIT_T_DEFAULT), nullptr, nullptr);
}

extern "C" int trec_spmm_csr_split(const int64_t* indptr, const int32_t* indices, const float* values,
                                   const int32_t* val_perm, const void* packed_entries, int64_t n_rows, int64_t nnz,
                                   const float* W, int32_t d, const float* own, int32_t accumulate, float* out,
                                   float* out_rowsum, void* workspace, int64_t workspace_bytes, void* stream)
{
    TREC_REQUIRE(indptr && W && out, "trec_spmm_csr_split: null pointer");
    TREC_REQUIRE(nnz == 0 || packed_entries || (indices && values), "trec_spmm_csr_split: null indices/values with nnz != 0");
    TREC_REQUIRE(d >= 1 && ((d % 4 == 0 && d <= 1024) || d <= 256),
                 "trec_spmm_csr_split: d must be a multiple of 4 up to 1024, or any value up to 256");
    TREC_REQUIRE(n_rows >= 0 && nnz >= 0, "trec_spmm_csr_split: bad sizes");
    if (n_rows == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int split_t = trec_get_tuning("split_t", SPLIT_T_DEFAULT);
    SplitPlan plan;
    int rc = make_plan(indptr, n_rows, nnz, d, workspace, workspace_bytes, split_t, &plan, st, "trec_spmm_csr_split(plan)");
    if (rc) return rc;
    const bool vec4 = d % 4 == 0;
    const int per = vec4 ? d / 4 : d;                            // column groups a subgroup has to cover
    int lpr_log2 = pow2ceil_log2(per);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int lpr = 1 << lpr_log2;
    const int iters = (per + lpr - 1) / lpr;
    const unsigned row_blocks = (unsigned)ceil_div64(n_rows * lpr, 256);
    const int64_t chunk_want = ceil_div64(plan.max_chunks * lpr, 256);
    const unsigned chunk_blocks = (unsigned)(chunk_want < 4096 ? chunk_want : 4096);
    const int64_t long_want = ceil_div64(plan.max_long * lpr, 256);
    const unsigned reduce_blocks = (unsigned)(long_want < 2048 ? long_want : 2048);
    const int32_t* idx = packed_entries ? (const int32_t*)packed_entries : indices;
#define TREC_SPLIT_LAUNCH(IT, PK, VC)                                                                                   \
    do {                                                                                                                \
        hipLaunchKernelGGL((split_rows_kernel<IT, PK, VC>), dim3(row_blocks), dim3(256), 0, st, indptr, idx, values,     \
                           val_perm, n_rows, W, d, lpr_log2, own, accumulate, split_t, out, out_rowsum);                \
        hipLaunchKernelGGL((split_chunks_kernel<IT, PK, VC>), dim3(chunk_blocks), dim3(256), 0, st, indptr, idx, values, \
                           val_perm, W, d, lpr_log2, own, split_t, plan);                                               \
        hipLaunchKernelGGL((split_reduce_kernel<IT, VC>), dim3(reduce_blocks), dim3(256), 0, st, d, lpr_log2,            \
                           accumulate, plan, out, out_rowsum);                                                          \
    } while (0)
#define TREC_SPLIT_ITERS(PK, VC)                                                                                        \
    do {                                                                                                                \
        if (iters == 1) TREC_SPLIT_LAUNCH(1, PK, VC); else if (iters == 2) TREC_SPLIT_LAUNCH(2, PK, VC);                \
        else TREC_SPLIT_LAUNCH(4, PK, VC);                                                                              \
    } while (0)
    if (vec4) { if (packed_entries) TREC_SPLIT_ITERS(true, 4); else TREC_SPLIT_ITERS(false, 4); }
    else { if (packed_entries) TREC_SPLIT_ITERS(true, 1); else TREC_SPLIT_ITERS(false, 1); }
#undef TREC_SPLIT_ITERS
#undef TREC_SPLIT_LAUNCH
    return trec_check_launch("trec_spmm_csr_split");
}

extern "C" int trec_spmv_csr_split(const int64_t* indptr, const int32_t* indices, const float* values,
                                   const int32_t* val_perm, int64_t n_rows, int64_t nnz, const float* beta, float* out,
                                   void* workspace, int64_t workspace_bytes, void* stream)
{
    TREC_REQUIRE(indptr && out, "trec_spmv_csr_split: null pointer");
    TREC_REQUIRE(nnz == 0 || values, "trec_spmv_csr_split: null values with nnz != 0");
    TREC_REQUIRE(!beta || nnz == 0 || indices, "trec_spmv_csr_split: beta needs indices");
    if (n_rows == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int split_t = trec_get_tuning("split_t", SPLIT_T_DEFAULT);
    SplitPlan plan;
    int rc = make_plan(indptr, n_rows, nnz, 1, workspace, workspace_bytes, split_t, &plan, st, "trec_spmv_csr_split(plan)");
    if (rc) return rc;
    if (beta)
        hipLaunchKernelGGL(split_spmv_rows_kernel, dim3((unsigned)ceil_div64(n_rows, 256)), dim3(256), 0, st, indptr,
                           indices, values, val_perm, n_rows, beta, split_t, out);
    else
        hipLaunchKernelGGL(split_segsum_rows_kernel, dim3((unsigned)ceil_div64(n_rows * 16, 256)), dim3(256), 0, st, indptr,
                           values, val_perm, n_rows, split_t, out);
    const int64_t chunk_want = ceil_div64(plan.max_chunks * 64, 256);
    hipLaunchKernelGGL(split_spmv_chunks_kernel, dim3((unsigned)(chunk_want < 4096 ? chunk_want : 4096)), dim3(256), 0, st,
                       indptr, indices, values, val_perm, beta, split_t, plan);
    const int64_t long_want = ceil_div64(plan.max_long, 256);
    hipLaunchKernelGGL(split_spmv_reduce_kernel, dim3((unsigned)(long_want < 1024 ? long_want : 1024)), dim3(256), 0, st,
                       plan, out);
    return trec_check_launch("trec_spmv_csr_split");
}
