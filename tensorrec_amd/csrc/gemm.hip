// tensorrec_amd/csrc/gemm.hip -- plain fp32 GEMM on MFMA for the dense layer of ReLURepresentationGraph.
//
// Replaces tf.matmul(tf_relu, tf_linear_weights) at tensorrec/representation_graphs.py:121 and its two autodiff
// gradients (dRelu = dOut . W2^T, dW2 = Relu^T . dOut).  Not the graded kernel: a straightforward
// 64x64x16 LDS-tiled kernel on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered accumulation), 4 waves per block, each
// wave one 32x32 quadrant.  Transposes and ragged edges are handled at staging time (zero fill).
#include "common.hpp"

#define GB 64
#define GK 16

__global__ __launch_bounds__(256) void gemm_f32_kernel(int ta, int tb, int64_t M, int64_t N, int64_t K,
                                                      const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                      int64_t ldc, int accumulate)
{
    __shared__ float As[GB][GK + 1];     // [m][k]
    __shared__ float Bs[GK][GB + 1];     // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GB, n0 = (int64_t)blockIdx.x * GB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int64_t k0 = 0; k0 < K; k0 += GK) {
        // stage A tile: 64 x 16 = 1024 elements, 4 per thread; pick the index order that is contiguous in memory
        for (int e = tid; e < GB * GK; e += 256) {
            int mm, kk;
            if (ta) { mm = e % GB; kk = e / GB; } else { kk = e % GK; mm = e / GK; }
            const int64_t gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < M && gk < K) v = ta ? A[gk * lda + gm] : A[gm * lda + gk];
            As[mm][kk] = v;
        }
        for (int e = tid; e < GK * GB; e += 256) {
            int nn, kk;
            if (tb) { kk = e % GK; nn = e / GK; } else { nn = e % GB; kk = e / GB; }
            const int64_t gn = n0 + nn, gk = k0 + kk;
            float v = 0.f;
            if (gn < N && gk < K) v = tb ? B[gn * ldb + gk] : B[gk * ldb + gn];
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 2; ++ks) {
            const int k = 2 * ks + (lane >> 5);
            const float a = As[wm * 32 + (lane & 31)][k];
            const float b = Bs[k][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int64_t col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            float* c = C + row * ldc + col;
            *c = accumulate ? (*c + acc[r]) : acc[r];
        }
    }
}

extern "C" int trec_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                             int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t accumulate,
                             void* stream)
{
    TREC_REQUIRE(A && B && C, "trec_gemm_f32: null pointer");
    TREC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "trec_gemm_f32: negative size");
    if (M == 0 || N == 0) return TREC_OK;
    const int64_t gx = ceil_div64(N, GB), gy = ceil_div64(M, GB);
    TREC_REQUIRE(gy <= 65535, "trec_gemm_f32: M too large for one launch (tile the rows)");
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, trans_a,
                       trans_b, M, N, K, A, lda, B, ldb, C, ldc, accumulate);
    return trec_check_launch("trec_gemm_f32");
}
