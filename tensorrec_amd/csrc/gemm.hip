// tensorrec_amd/csrc/gemm.hip -- fp32 GEMM on MFMA for the dense layer of ReLURepresentationGraph.
//
// Replaces tf.matmul(tf_relu, tf_linear_weights) at tensorrec/representation_graphs.py:121 and its two autodiff
// gradients (dRelu = dOut . W2^T, dW2 = Relu^T . dOut).  Exact fp32 products on v_mfma_f32_32x32x2_f32.
//
// 128 x 128 x 16 tiles, 4 waves in a 2 x 2 grid, each wave a 64 x 64 quadrant = 2 x 2 MFMA blocks (64 accumulator
// registers): 4 LDS reads feed 4 MFMAs (256 matrix-pipe cycles), so the kernel is matrix-pipe bound once the tile loads
// hide -- the next tile's global loads are issued into registers before the current tile is multiplied.  Operands are
// staged k-major ([k][m] / [k][n], row stride 132 floats): an operand whose tile dimension is contiguous in memory is
// copied with 16-byte loads and stores, one whose k dimension is contiguous is transposed on the way into LDS (four
// 4-byte stores whose banks differ across the lanes that share a row).  Ragged edges are zero-filled at staging time.
//
// dW2 = Relu^T . dOut has a tiny output ([relu_size, n_components]) and a huge K (all users): `splits` > 1 cuts K into
// slices that run as separate workgroups (grid.z), each writing its partial product to the workspace; gemm_splitk_reduce
// adds the slices in order, so the result does not depend on scheduling.
#include "common.hpp"

namespace {

constexpr int TM = 128, TN = 128, TK = 16, LDT = 132;

// one float4 of a [rows, cols] row-major matrix at (row, col..col+3), zero beyond the edges
__device__ __forceinline__ f32x4 load4(const float* __restrict__ base, int64_t ld, int64_t row, int64_t col, int64_t rows,
                                       int64_t cols, bool vec_ok)
{
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (row >= rows) return v;
    const float* p = base + row * ld + col;
    if (vec_ok && col + 3 < cols) return *(const f32x4*)p;
    if (col < cols) v.x = p[0];
    if (col + 1 < cols) v.y = p[1];
    if (col + 2 < cols) v.z = p[2];
    if (col + 3 < cols) v.w = p[3];
    return v;
}

// Staging of one operand tile [TK][128] (k-major in LDS) from global memory.
// KCONT = false: element (k, x) at src[k * ld + x]  (tile dimension contiguous): slot -> (k = idx / 32, x4 = idx % 32)
// KCONT = true : element (k, x) at src[x * ld + k]  (k contiguous)             : slot -> (x = idx / 4,  k4 = idx % 4)
template <bool KCONT>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int64_t ld, int64_t x0, int64_t k0, int64_t X,
                                          int64_t Kend, bool vec_ok, int tid, f32x4 (&r)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if (KCONT) r[i] = load4(src, ld, x0 + (idx >> 2), k0 + 4 * (idx & 3), X, Kend, vec_ok);
        else r[i] = load4(src, ld, k0 + (idx >> 5), x0 + 4 * (idx & 31), Kend, X, vec_ok);
    }
}

template <bool KCONT>
__device__ __forceinline__ void tile_store(float (*T)[LDT], int tid, const f32x4 (&r)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if (KCONT) {
            const int x = idx >> 2, k = 4 * (idx & 3);
            T[k][x] = r[i].x; T[k + 1][x] = r[i].y; T[k + 2][x] = r[i].z; T[k + 3][x] = r[i].w;
        } else {
            *(f32x4*)&T[idx >> 5][4 * (idx & 31)] = r[i];
        }
    }
}

// AK: A is stored [M, K] (k contiguous, trans_a == 0); otherwise [K, M].  BK: B is stored [N, K] (trans_b != 0).
template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                         const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                         int64_t ldc, int accumulate, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];     // [k][m]
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];     // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
    const int64_t kb = (int64_t)blockIdx.z * k_per_split;
    const int64_t ke = kb + k_per_split < K ? kb + k_per_split : K;
    const bool a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const bool b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[2], rb[2];
    if (kb < ke) {
        tile_load<AK>(A, lda, m0, kb, M, ke, a_vec, tid, ra);
        tile_load<BK>(B, ldb, n0, kb, N, ke, b_vec, tid, rb);
    }
    for (int64_t k0 = kb; k0 < ke; k0 += TK) {
        tile_store<AK>(As, tid, ra);
        tile_store<BK>(Bs, tid, rb);
        __syncthreads();
        if (k0 + TK < ke) {                                      // next tile in flight while this one is multiplied
            tile_load<AK>(A, lda, m0, k0 + TK, M, ke, a_vec, tid, ra);
            tile_load<BK>(B, ldb, n0, k0 + TK, N, ke, b_vec, tid, rb);
        }
#pragma unroll
        for (int ks = 0; ks < TK / 2; ++ks) {
            const int k = 2 * ks + half;
            const float a0 = As[k][wm * 64 + l31], a1 = As[k][wm * 64 + 32 + l31];
            const float b0 = Bs[k][wn * 64 + l31], b1 = Bs[k][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    float* dst = partial ? partial + (int64_t)blockIdx.z * M * N : C;
    const int64_t ldd = partial ? N : ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M && col < N) {
                    float* c = dst + row * ldd + col;
                    *c = (accumulate && !partial) ? (*c + acc[i][j][r]) : acc[i][j][r];
                }
            }
        }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits, int64_t M,
                                                                int64_t N, float* __restrict__ C, int64_t ldc,
                                                                int accumulate)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int64_t row = idx / N, col = idx - row * N;
    float acc = accumulate ? C[row * ldc + col] : 0.f;
    for (int s = 0; s < splits; ++s) acc += partial[(int64_t)s * M * N + idx];
    C[row * ldc + col] = acc;
}

}  // namespace

extern "C" int trec_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                             int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t accumulate,
                             float* workspace, int32_t splits, void* stream)
{
    TREC_REQUIRE(A && B && C, "trec_gemm_f32: null pointer");
    TREC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "trec_gemm_f32: negative size");
    TREC_REQUIRE(splits <= 1 || workspace, "trec_gemm_f32: splits > 1 needs a workspace of splits * M * N floats");
    if (M == 0 || N == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (splits < 1) splits = 1;
    int64_t k_per = ceil_div64(ceil_div64(K, splits), TK) * TK;
    if (k_per < TK) k_per = TK;
    splits = (int32_t)ceil_div64(K > 0 ? K : 1, k_per);
    const int64_t gx = ceil_div64(N, TN), gy = ceil_div64(M, TM);
    TREC_REQUIRE(gy <= 65535 && splits <= 65535, "trec_gemm_f32: M too large for one launch (tile the rows)");
    float* partial = splits > 1 ? workspace : nullptr;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)splits);
#define TREC_GEMM(AKV, BKV)                                                                                            \
    hipLaunchKernelGGL((gemm_f32_kernel<AKV, BKV>), grid, dim3(256), 0, st, M, N, K, k_per, A, lda, B, ldb, C, ldc,     \
                       accumulate, partial)
    if (!trans_a && !trans_b) TREC_GEMM(true, false);
    else if (!trans_a && trans_b) TREC_GEMM(true, true);
    else if (trans_a && !trans_b) TREC_GEMM(false, false);
    else TREC_GEMM(false, true);
#undef TREC_GEMM
    if (splits > 1)
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, st, partial,
                           splits, M, N, C, ldc, accumulate);
    return trec_check_launch("trec_gemm_f32");
}
