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


// ---- the same product on bf16 MFMA with SPLIT operands (round 6): x = hi + lo, hi = bf16(x), lo = bf16(x - hi) --------------------
// C = Ah.Bh + Ah.Bl + Al.Bh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: every operand carries 16 mantissa bits (relative
// error <= 2^-17), the dropped Al.Bl term is <= 2^-16 of a product -- ~1.5e-5 relative per product before the sum averages it out,
// against the 1e-4 bar of the gradients these products are (tensorrec.py:487-489).  Three MFMAs of 32 cycles do what eight fp32 MFMAs
// of 64 cycles do: the GEMMs on the dense coefficient matrix of BASELINE configs[4] (G^T.U, G.V: 1.9 TFLOP each, 0.6-0.7 of the
// 157 TF fp32 peak = 17-20 ms) become bound by the 14.8 GB read of G instead.  NOT used where a product is compared with the
// oracle's fmaf chain (the ReLU layer keeps trec_gemm_f32).
// 128 x 128 x 64 tiles (64 KB of loads in flight per workgroup: with 32-wide slabs the kernel waited for memory, 8-10 ms per GEMM
// where reading G takes 2.5); waves 0-1 stage A, waves 2-3 stage B: a thread owns eight (row, 8 consecutive k) octets, splits them and
// writes hi / lo as one 16-byte LDS store each ([row][k] bf16, 128-byte rows, octets XOR-swizzled); an operand whose ROW dimension is contiguous
// in memory is read as 8 k-rows of float4 and transposed in registers.
constexpr int SK = 64;                      // k per slab; an LDS row = SK bf16 = 128 bytes = eight 16-byte octets, no padding:
// octet o of row r sits at physical octet o ^ ((r >> 2) & 7).  The staging writes of a row-contiguous operand put lanes four rows
// apart (a float4 = four rows): with any 16-byte-aligned row stride those rows fall on two bank groups (16-way conflicts: the first
// version ran its all-transposed instance 35 % slower than the other); with the swizzle 8 consecutive lanes cover all 32 banks, and
// the MFMA fragment reads (32 consecutive rows of one octet) meet 8 bank groups x 4 lanes -- the 4 cycles 512 bytes take anyway.
__device__ __forceinline__ int soff(int row, int oct) { return row * SK + ((oct ^ ((row >> 2) & (SK / 8 - 1))) << 3); }
constexpr int NOCT = SK / 8;                // 8-element k-octets per row and slab
constexpr int NR = 2 * (SK / 8);            // float4 registers of a thread's share of a slab (128 rows x SK / 128 threads / 4)

// a thread's share of a [128 rows][SK] slab as float4s.  KCONT (element (row, k) at src[row * ld + k]): octet t % NOCT of rows
// t / NOCT + (128 / NOCT) o -- NOCT lanes cover the contiguous SK * 4 bytes of a row's slab; r[2 o], r[2 o + 1] = its eight values.
// Otherwise (element at src[k * ld + row]): rows 4 (t % 32) .. + 3 at k = 8 (t / 32 + 4 q) + j -- 32 lanes cover 512 contiguous
// bytes of a k-row; r[8 q + j] = the four rows' values at that k.  t = thread index inside its staging half (0..127).
template <bool KCONT>
__device__ __forceinline__ void split_load(const float* __restrict__ src, int64_t ld, int64_t x0, int64_t k0, int64_t X, int64_t Kend,
                                           bool vec_ok, int t, f32x4 (&r)[NR])
{
    // interior slabs (all but the ragged edges): unconditional 16-byte loads, issued back to back -- load4's edge tests made every
    // load its own block of branches (and its own wait)
    const bool interior = vec_ok && x0 + 128 <= X && k0 + SK <= Kend;   // (workgroup-uniform)
    if (KCONT) {
        if (interior) {
            const float* p = src + (x0 + t / NOCT) * ld + k0 + 8 * (t % NOCT);
#pragma unroll
            for (int o = 0; o < NR / 2; ++o) {
                r[2 * o] = *(const f32x4*)(p + (int64_t)(128 / NOCT) * o * ld);
                r[2 * o + 1] = *(const f32x4*)(p + (int64_t)(128 / NOCT) * o * ld + 4);
            }
            return;
        }
#pragma unroll
        for (int o = 0; o < NR / 2; ++o) {
            const int64_t row = x0 + t / NOCT + (128 / NOCT) * o, k = k0 + 8 * (t % NOCT);
            r[2 * o] = load4(src, ld, row, k, X, Kend, vec_ok);
            r[2 * o + 1] = load4(src, ld, row, k + 4, X, Kend, vec_ok);
        }
    } else {
        if (interior) {
            const float* p = src + (k0 + 8 * (t >> 5)) * ld + x0 + 4 * (t & 31);
#pragma unroll
            for (int q = 0; q < NR / 8; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) r[8 * q + j] = *(const f32x4*)(p + (int64_t)(32 * q + j) * ld);
            return;
        }
#pragma unroll
        for (int q = 0; q < NR / 8; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[8 * q + j] = load4(src, ld, k0 + 8 * ((t >> 5) + 4 * q) + j, x0 + 4 * (t & 31), Kend, X, vec_ok);
    }
}

__device__ __forceinline__ void split_pack(const float (&x)[8], u32x4& hi, u32x4& lo)
{
    unsigned int h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        h[p] = f32x2_to_bf16x2_bits(x[2 * p], x[2 * p + 1]);
        const float b0 = __uint_as_float(h[p] << 16), b1 = __uint_as_float(h[p] & 0xffff0000u);
        l[p] = f32x2_to_bf16x2_bits(x[2 * p] - b0, x[2 * p + 1] - b1);
    }
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}

template <bool KCONT>
__device__ __forceinline__ void split_store(unsigned short* H, unsigned short* L, int t, const f32x4 (&r)[NR])
{
    if (KCONT) {
#pragma unroll
        for (int o = 0; o < NR / 2; ++o) {
            const float x[8] = {r[2 * o].x, r[2 * o].y, r[2 * o].z, r[2 * o].w, r[2 * o + 1].x, r[2 * o + 1].y, r[2 * o + 1].z, r[2 * o + 1].w};
            u32x4 hi, lo;
            split_pack(x, hi, lo);
            const int at = soff(t / NOCT + (128 / NOCT) * o, t % NOCT);
            *(u32x4*)(H + at) = hi;
            *(u32x4*)(L + at) = lo;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NR / 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x[8] = {r[8 * q][e], r[8 * q + 1][e], r[8 * q + 2][e], r[8 * q + 3][e],
                                    r[8 * q + 4][e], r[8 * q + 5][e], r[8 * q + 6][e], r[8 * q + 7][e]};
                u32x4 hi, lo;
                split_pack(x, hi, lo);
                const int at = soff(4 * (t & 31) + e, (t >> 5) + 4 * q);
                *(u32x4*)(H + at) = hi;
                *(u32x4*)(L + at) = lo;
            }
    }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                            const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                            int64_t ldc, int accumulate, float* __restrict__ partial,
                                                            float* __restrict__ a_sums)
{
    extern __shared__ __attribute__((aligned(16))) char gsm[];           // four [128][SK] bf16 arrays: 65,536 bytes
    unsigned short* Ah = (unsigned short*)gsm;
    unsigned short* Al = Ah + TM * SK;
    unsigned short* Bh = Al + TM * SK;
    unsigned short* Bl = Bh + TN * SK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
    const int64_t kb = (int64_t)blockIdx.z * k_per_split;
    const int64_t ke = kb + k_per_split < K ? kb + k_per_split : K;
    const bool a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const bool b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    const bool stage_a = tid < 128;                              // (wave-uniform)
    const int t = tid & 127;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // (one slab of global loads in flight per workgroup, in registers.  Measured and dropped: two slabs of 32 with the multiply loop
    // written twice -- spills at two workgroups per CU, 2.8x slower; two slabs of 64 at ONE workgroup per CU (launch bounds (256, 1),
    // 140 AGPRs of overflow, no scratch) -- 14.1 / 9.6 ms against 9.6 / 7.7: the second workgroup hides more than the second slab)
    f32x4 rr[NR];
    auto load_slab = [&](int64_t k0) __attribute__((always_inline)) {
        if (stage_a) split_load<AK>(A, lda, m0, k0, M, ke, a_vec, t, rr);
        else split_load<BK>(B, ldb, n0, k0, N, ke, b_vec, t, rr);
    };
    // a_sums (A stored [K, M] only): sum over k of every row of op(A) -- the column sums of the stored matrix -- ride along in the
    // threads that stage A (they hold its values anyway): four running sums per thread for its four rows, met at the end
    const bool want_sums = !AK && a_sums != nullptr && blockIdx.x == 0;  // (one column of workgroups reports them)
    f32x4 asum = {0.f, 0.f, 0.f, 0.f};
    if (kb < ke) load_slab(kb);
    for (int64_t k0 = kb; k0 < ke; k0 += SK) {
        if (!AK && want_sums && stage_a) {
#pragma unroll
            for (int q = 0; q < NR; ++q) { asum[0] += rr[q][0]; asum[1] += rr[q][1]; asum[2] += rr[q][2]; asum[3] += rr[q][3]; }
        }
        if (stage_a) split_store<AK>(Ah, Al, t, rr);
        else split_store<BK>(Bh, Bl, t, rr);
        __syncthreads();
        if (k0 + SK < ke) load_slab(k0 + SK);                    // next slab in flight while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < SK / 16; ++kk) {
            const int oct = kk * 2 + half;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = soff(wm * 64 + i * 32 + l31, oct), rb = soff(wn * 64 + i * 32 + l31, oct);
                ah[i] = *(const bf16x8*)(Ah + ra);
                al[i] = *(const bf16x8*)(Al + ra);
                bh[i] = *(const bf16x8*)(Bh + rb);
                bl[i] = *(const bf16x8*)(Bl + rb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // (small terms first: they meet the accumulator before the large one does)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);                   // (the next k-step's eight fragments are not read ahead: registers)
        }
        __syncthreads();
    }
    if (!AK && want_sums) {
        // the four k-groups' partial sums of each row meet through LDS (the operand buffers are free now), in k-group order
        float* red = (float*)gsm;                                // [4][128]
        if (stage_a) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(t >> 5) * TM + 4 * (t & 31) + e] = asum[e];
        }
        __syncthreads();
        if (tid < TM && m0 + tid < M)
            a_sums[(int64_t)blockIdx.z * M + m0 + tid] = ((red[tid] + red[TM + tid]) + red[2 * TM + tid]) + red[3 * TM + tid];
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


// The same contract as trec_gemm_f32 with both operands split into two bf16 terms (three bf16 MFMAs per product block, fp32
// accumulation; ~1e-5 relative per product): for products that are gradients held to 1e-4 (the dense-coefficient GEMMs of the tiled
// WMRB step), not for values compared with the oracle's fmaf chain.
extern "C" int trec_gemm_f32_split_bf16(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t accumulate,
                                        float* workspace, int32_t splits, float* a_colsum_parts, void* stream)
{
    // a_colsum_parts (nullable; trans_a only): float [splits][M] -- per K slice the sums over k of op(A)'s rows, i.e. the column
    // sums of the stored [K, M] matrix; the caller adds the slices (in order)
    TREC_REQUIRE(!a_colsum_parts || trans_a, "trec_gemm_f32_split_bf16: column sums come with trans_a (A stored [K, M])");
    TREC_REQUIRE(A && B && C, "trec_gemm_f32_split_bf16: null pointer");
    TREC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "trec_gemm_f32_split_bf16: negative size");
    TREC_REQUIRE(splits <= 1 || workspace, "trec_gemm_f32_split_bf16: splits > 1 needs a workspace of splits * M * N floats");
    if (M == 0 || N == 0) return TREC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (splits < 1) splits = 1;
    int64_t k_per = ceil_div64(ceil_div64(K, splits), SK) * SK;
    if (k_per < SK) k_per = SK;
    splits = (int32_t)ceil_div64(K > 0 ? K : 1, k_per);
    const int64_t gx = ceil_div64(N, TN), gy = ceil_div64(M, TM);
    TREC_REQUIRE(gy <= 65535 && splits <= 65535, "trec_gemm_f32_split_bf16: M too large for one launch (tile the rows)");
    float* partial = splits > 1 ? workspace : nullptr;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)splits);
    constexpr int LDS3 = 4 * TM * SK * 2;
#define TREC_GEMM3(AKV, BKV)                                                                                           \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<AKV, BKV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3); \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_bf16x3_kernel<AKV, BKV>), grid, dim3(256), LDS3, st, M, N, K, k_per, A, lda, B, ldb, C, ldc, \
                           accumulate, partial, a_colsum_parts);                                                       \
    } while (0)
    if (!trans_a && !trans_b) { TREC_GEMM3(true, false); }
    else if (!trans_a && trans_b) { TREC_GEMM3(true, true); }
    else if (trans_a && !trans_b) { TREC_GEMM3(false, false); }
    else { TREC_GEMM3(false, true); }
#undef TREC_GEMM3
    if (splits > 1)
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, st, partial,
                           splits, M, N, C, ldc, accumulate);
    return trec_check_launch("trec_gemm_f32_split_bf16");
}
