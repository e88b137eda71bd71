// tensorrec_amd/csrc/user_prep.hip -- the user side of the int8 -> bf16 -> fp32 cascade, prepared on the device in five launches.
//
// What it replaces (round 3: ops.score_prep_filter(sort_users=True) + ops.score_prep_i8_pair): the class of every user row, a
// stable torch.sort by class (20 rocprim merge launches), bincount / cumsum / index arithmetic for the band layout, ONE HOST
// READ of the padded row count in front of an idle GPU, an index_select of the whole [U, d] representation (512 MB at 1M x 128),
// then two operand passes that each read the rows again -- ~45 launches, most of them ATen glue.  Here:
//
//   1  natscale_kernel     nat[u] = the int8 scale row u wants (the best of max |x| / 127 and a half / a quarter of it by the
//                          quantisation error each leaves), gmax = their maximum            -- one read of the rows
//   2  class_hist_kernel   class c(u) = clamp(floor(4 log2(gmax / nat[u])), 0, 63): a geometric ladder of scales, 4 per octave;
//                          per-block histograms of the 64 classes; the layout's source map is reset to -1
//   3  class_scan_kernel   (one workgroup) class totals -> bands of two adjacent classes, each padded to whole int8 workgroups
//                          of ``wg_rows`` rows -> per-(block, class) output offsets; per int8 workgroup its class and scale
//                          (the class of its first row = its largest scale), the ladder, the classes in use, the padded row count
//   4  class_place_kernel  a STABLE counting sort: pos[u] = offset of (block, class) + rank of u among the block's users of that
//                          class (ballot / popcount inside a wave, wave counts through LDS); src[pos[u]] = u
//   5  prep_sorted_kernel  layout row r <- caller's row src[r] (one 512-byte gather per row, the only read of the rows after 1):
//                          the exact fp32 operand, its bf16 image with {||x||, ||x - bf16(x)||}, its int8 image under the
//                          workgroup's scale with {||x||, ||x - a q||}, and the user bias in layout order.  Rows without a
//                          source (band padding, the tail of the allocation) are zero.
//
// The layout is sized by a BOUND known to the host -- n + 32 bands x (wg_rows - 1) padding rows, rounded up to whole workgroups
// -- so nothing here or downstream waits for the device: int8 workgroups beyond the padded row count have scale 0 and exit at
// once (blockmax_i8x16_kernel), rows without a source keep nothing (their thresholds are +inf: trec_topk_cascade_floor) and
// write no result.  Deterministic: the same rows give the same layout on every rank of an item-sharded run.
//
// Reference: the user representation entering tf.matmul of tensorrec/prediction_graphs.py:49-50 (cosine: after the
// l2_normalize of :64-69); the sort is an internal layout -- results leave in the caller's order (trec_topk_candidates_finish
// writes through src).
#include "common.hpp"
#include <math.h>

namespace {

constexpr int UP_CLASSES = 64;       // scale classes: gmax * 2^(-c / 4), c = 0 .. 63 (16 octaves; smaller rows share the last)
constexpr int UP_BAND = 2;           // adjacent classes per band: the rows of one int8 workgroup come from ONE band
constexpr int UP_BANDS = UP_CLASSES / UP_BAND;
constexpr int UP_TB = 2048;          // users per block of the histogram / placement kernels
constexpr int UP_SEGS = 16;          // segments of blocks the scan's 1024 threads walk (64 classes x 16)

template <int G>
__global__ __launch_bounds__(256) void natscale_kernel(const float* __restrict__ x, int64_t n, int d, int normalize,
                                                      float* __restrict__ nat, float* __restrict__ gmax)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool ok = row < n;
    const int sub = threadIdx.x % G;
    const float* xr = x + (ok ? row : 0) * (int64_t)d;
    f32x4 v[2];
    float am = 0.f, ss = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (it * G + sub) * 4;
        v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok && c < d) {
            if ((d & 3) == 0) v[it] = *(const f32x4*)(xr + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) v[it][e] = xr[c + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = fabsf(v[it][e]);
            am = (a > am || a != a) ? a : am;                              // NaN sticks
            ss = fmaf(v[it][e], v[it][e], ss);
        }
    }
    if (am != am) am = INFINITY;
    for (int off = G / 2; off > 0; off >>= 1) { am = fmaxf(am, __shfl_xor(am, off, 64)); ss += __shfl_xor(ss, off, 64); }
    float best = am / 127.0f;
    if (am > 0.f && am < INFINITY) {
        float err[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int cnd = 0; cnd < 3; ++cnd) {
            const float sc = (am / 127.0f) * (cnd == 0 ? 1.0f : (cnd == 1 ? 0.5f : 0.25f));
            const float inv = 1.0f / sc;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float q = rintf(v[it][e] * inv);
                    q = fminf(fmaxf(q, -127.f), 127.f);
                    const float r = v[it][e] - q * sc;
                    err[cnd] = fmaf(r, r, err[cnd]);
                }
            for (int off = G / 2; off > 0; off >>= 1) err[cnd] += __shfl_xor(err[cnd], off, 64);
        }
        if (err[1] < err[0] && err[1] <= err[2]) best *= 0.5f;
        else if (err[2] < err[0] && err[2] < err[1]) best *= 0.25f;
    }
    // cosine: the operand is the normalised row x / max(||x||, 1e-6) (prep_sorted_kernel), whose wanted scale is the raw one over
    // the norm (which of the three candidates wins does not depend on the row's scale)
    if (normalize) best = best / fmaxf(sqrtf(ss), 1e-6f);
    if (!(best > 0.f)) best = (best != best) ? best : 0.f;               // an all-zero row wants nothing (smallest class); NaN stays
    if (sub == 0 && ok) nat[row] = best;
    float gm = (sub == 0 && ok) ? best : 0.f;
    if (gm != gm) gm = INFINITY;                                         // a non-finite row poisons gmax: everybody is flagged later
    for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(gm) > *(volatile unsigned int*)gmax) atomicMax((unsigned int*)gmax, __float_as_uint(gm));
}

__device__ __forceinline__ int scale_class(float nat, float g)
{
    float c = floorf((float)(UP_CLASSES / 16) * log2f(g / nat));        // nat = 0 -> +inf -> the last class; NaN -> class 0
    c = fminf(fmaxf(c, 0.f), (float)(UP_CLASSES - 1));
    return (int)c;
}

// cls[u] and the per-block class histograms; src[0 .. n_alloc) = -1
__global__ __launch_bounds__(256) void class_hist_kernel(const float* __restrict__ nat, const float* __restrict__ gmax, int64_t n,
                                                        unsigned char* __restrict__ cls, int32_t* __restrict__ blockhist,
                                                        int32_t* __restrict__ src, int64_t n_alloc)
{
    __shared__ int hist[UP_CLASSES];
    if (threadIdx.x < UP_CLASSES) hist[threadIdx.x] = 0;
    __syncthreads();
    float g = gmax[0];
    g = g > 0.f ? g : 1.0f;
    const int64_t u0 = (int64_t)blockIdx.x * UP_TB;
    for (int j = 0; j < UP_TB / 256; ++j) {
        const int64_t u = u0 + j * 256 + threadIdx.x;
        if (u < n) {
            const int c = scale_class(nat[u], g);
            cls[u] = (unsigned char)c;
            atomicAdd(&hist[c], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < UP_CLASSES) blockhist[(int64_t)blockIdx.x * UP_CLASSES + threadIdx.x] = hist[threadIdx.x];
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_alloc; r += (int64_t)gridDim.x * 256) src[r] = -1;
}

// One workgroup of 1024 threads = 64 classes x 16 segments of blocks.  blockhist becomes, in place, the output offset of every
// (block, class); meta[0] = padded row count, meta[1] = real rows
__global__ __launch_bounds__(1024) void class_scan_kernel(int32_t* __restrict__ blockhist, int32_t n_blk, const float* __restrict__ gmax,
                                                         int64_t n, int32_t wg_rows, int32_t n_wg_alloc,
                                                         float* __restrict__ wg_scale, int32_t* __restrict__ wg_class,
                                                         float* __restrict__ ladder, int32_t* __restrict__ class_used,
                                                         int32_t* __restrict__ meta)
{
    __shared__ int part[UP_SEGS][UP_CLASSES];
    __shared__ int total[UP_CLASSES], cbase[UP_CLASSES], used[UP_CLASSES];
    __shared__ int bstart[UP_BANDS + 1], bpad[UP_BANDS];
    __shared__ float lad[UP_CLASSES];
    const int c = threadIdx.x & (UP_CLASSES - 1), seg = threadIdx.x / UP_CLASSES;
    const int per = (n_blk + UP_SEGS - 1) / UP_SEGS;
    const int b0 = seg * per, b1 = (b0 + per < n_blk) ? b0 + per : n_blk;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += blockhist[(int64_t)b * UP_CLASSES + c];
    part[seg][c] = s;
    if (threadIdx.x < UP_CLASSES) used[threadIdx.x] = 0;
    __syncthreads();
    if (seg == 0) {
        int t = 0;
        for (int j = 0; j < UP_SEGS; ++j) { const int v = part[j][c]; part[j][c] = t; t += v; }       // exclusive over the segments
        total[c] = t;
        float g = gmax[0];
        g = g > 0.f ? g : 1.0f;
        lad[c] = g * exp2f(-(float)c / (float)(UP_CLASSES / 16));
        ladder[c] = lad[c];
    }
    __syncthreads();
    if (threadIdx.x < UP_BANDS) {
        const int cnt = total[UP_BAND * threadIdx.x] + total[UP_BAND * threadIdx.x + 1];
        bpad[threadIdx.x] = (cnt + wg_rows - 1) / wg_rows * wg_rows;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int b = 0; b < UP_BANDS; ++b) { bstart[b] = t; t += bpad[b]; }
        bstart[UP_BANDS] = t;
        meta[0] = t;
        meta[1] = (int32_t)n;
    }
    __syncthreads();
    if (seg == 0) cbase[c] = bstart[c / UP_BAND] + ((c & 1) ? total[c - 1] : 0);
    __syncthreads();
    int run = cbase[c] + part[seg][c];
    for (int b = b0; b < b1; ++b) {
        const int v = blockhist[(int64_t)b * UP_CLASSES + c];
        blockhist[(int64_t)b * UP_CLASSES + c] = run;
        run += v;
    }
    const int n_pad = bstart[UP_BANDS];
    for (int w = threadIdx.x; w < n_wg_alloc; w += 1024) {
        const int64_t r0 = (int64_t)w * wg_rows;
        float sc = 0.f;                                                   // beyond the padded rows: an idle int8 workgroup
        int wc = -1;
        if (r0 < n_pad) {
            int b = 0;
            while (b + 1 < UP_BANDS && r0 >= bstart[b + 1]) ++b;          // (empty bands share their start with the next one)
            while (bpad[b] == 0 && b + 1 < UP_BANDS) ++b;
            const int off = (int)(r0 - bstart[b]);
            wc = off < total[UP_BAND * b] ? UP_BAND * b : UP_BAND * b + 1;
            sc = lad[wc];
            used[wc] = 1;
        }
        wg_scale[w] = sc;
        wg_class[w] = wc;
    }
    __syncthreads();
    if (threadIdx.x < UP_CLASSES) class_used[threadIdx.x] = used[threadIdx.x];
}

// the stable placement: users of one class keep their order (block by block, 256-user tile by tile, wave by wave, lane by lane)
__global__ __launch_bounds__(256) void class_place_kernel(const unsigned char* __restrict__ cls, const int32_t* __restrict__ blockoff,
                                                         int64_t n, int32_t* __restrict__ pos, int32_t* __restrict__ src)
{
    __shared__ int run[UP_CLASSES];
    __shared__ int wcount[4][UP_CLASSES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < UP_CLASSES) run[threadIdx.x] = blockoff[(int64_t)blockIdx.x * UP_CLASSES + threadIdx.x];
    const int64_t u0 = (int64_t)blockIdx.x * UP_TB;
    for (int j = 0; j < UP_TB / 256; ++j) {
        (&wcount[0][0])[threadIdx.x] = 0;
        __syncthreads();
        const int64_t u = u0 + j * 256 + threadIdx.x;
        const int c = u < n ? (int)cls[u] : -1;
        int rank = 0;
        bool todo = c >= 0;
        while (true) {                                                    // one round per distinct class of the wave
            const unsigned long long act = __builtin_amdgcn_ballot_w64(todo);
            if (act == 0ull) break;
            const int first = __builtin_ctzll(act);
            const int cc = __builtin_amdgcn_readlane(c, first);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(todo && c == cc);
            if (todo && c == cc) {
                rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (lane == first) wcount[wave][cc] = __builtin_popcountll(m);
                todo = false;
            }
        }
        __syncthreads();
        if (c >= 0) {
            int p = run[c] + rank;
            for (int w = 0; w < wave; ++w) p += wcount[w][c];
            pos[u] = p;
            src[p] = (int32_t)u;
        }
        __syncthreads();
        if (threadIdx.x < UP_CLASSES)
            run[threadIdx.x] += wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] + wcount[3][threadIdx.x];
        __syncthreads();
    }
}

// G lanes own one LAYOUT row (kt = 4 G columns: 32, 64 or 128).  The normalisation and the bf16 image follow prep_filter_kernel
// (topk_filter.hip), the int8 image prep_i8_kernel (score_blockmax_i8.hip): the same arithmetic per element.
template <int G>
__global__ __launch_bounds__(256) void prep_sorted_kernel(const float* __restrict__ x, int64_t n_alloc, int d, int kt, int normalize,
                                                         const int32_t* __restrict__ src, const float* __restrict__ wg_scale,
                                                         int wg_rows, const float* __restrict__ bias,
                                                         float* __restrict__ out_f32, unsigned short* __restrict__ out_bf16,
                                                         float2* __restrict__ row_stats, signed char* __restrict__ out_q,
                                                         float2* __restrict__ row_stats8, float* __restrict__ bias_sorted)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (row >= n_alloc) return;                                           // (whole groups: 256 % G == 0)
    const int sub = threadIdx.x % G;
    const int c = sub * 4;
    const int32_t s = src[row];
    const bool ok = s >= 0;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok && c < d) {
        const float* xr = x + (int64_t)s * d;
        if ((d & 3) == 0) v = *(const f32x4*)(xr + c);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < d) v[e] = xr[c + e];
        }
    }
    f32x4 w = v;
    if (normalize) {
        float ss = 0.f;
        ss = fmaf(v[0], v[0], ss); ss = fmaf(v[1], v[1], ss); ss = fmaf(v[2], v[2], ss); ss = fmaf(v[3], v[3], ss);
        for (int off = G / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float scale = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        w[0] *= scale; w[1] *= scale; w[2] *= scale; w[3] *= scale;
    }
    // ---- bf16 image (v_cvt_pk_bf16_f32: round-to-nearest-even) and its rounding-error norm
    uint2 pk;
    pk.x = f32x2_to_bf16x2_bits(w[0], w[1]);
    pk.y = f32x2_to_bf16x2_bits(w[2], w[3]);
    float sw = 0.f, se = 0.f, se8 = 0.f;
    // ---- int8 image under the workgroup's scale and its quantisation-error norm (clipping included)
    const float a = ok ? wg_scale[row / wg_rows] : 1.0f;
    const float inv = 1.0f / a;
    unsigned int pq = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned int word = (e < 2) ? pk.x : pk.y;
        const float back = __uint_as_float((e & 1) ? (word & 0xffff0000u) : (word << 16));
        const float err = w[e] - back;                                    // exact: the discarded low bits of w
        sw = fmaf(w[e], w[e], sw);
        se = fmaf(err, err, se);
        float q = rintf(w[e] * inv);
        q = fminf(fmaxf(q, -127.f), 127.f);
        if (!(q == q)) q = 0.f;                                           // NaN input: the error norm turns NaN -> the bound turns inf
        const float err8 = w[e] - q * a;
        se8 = fmaf(err8, err8, se8);
        pq |= ((unsigned int)(int)q & 0xffu) << (8 * e);
    }
    if (out_f32) *(f32x4*)(out_f32 + row * (int64_t)kt + c) = w;
    *(uint2*)(out_bf16 + row * (int64_t)kt + c) = pk;
    *(unsigned int*)(out_q + row * (int64_t)kt + c) = pq;
    for (int off = G / 2; off > 0; off >>= 1) {
        sw += __shfl_xor(sw, off, 64); se += __shfl_xor(se, off, 64); se8 += __shfl_xor(se8, off, 64);
    }
    if (sub == 0) {
        const float nw = sqrtf(sw);
        row_stats[row] = make_float2(nw, sqrtf(se));
        row_stats8[row] = make_float2(nw, sqrtf(se8));
        if (bias_sorted) bias_sorted[row] = (ok && bias) ? bias[s] : 0.f;
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
// rows of the layout for n users and int8 workgroups of wg_rows rows: every band may end in a partly filled workgroup
extern "C" int64_t trec_user_prep_alloc_rows(int64_t n, int32_t wg_rows)
{
    if (n <= 0 || wg_rows <= 0) return 0;
    const int64_t bands = n < UP_BANDS ? n : UP_BANDS;                    // (at most one partly filled workgroup per non-empty band)
    return ceil_div64(n + bands * (wg_rows - 1), wg_rows) * wg_rows;
}

// bytes of scratch trec_user_prep_sorted needs: nat [n] float | cls [n] bytes | per-block class offsets
extern "C" int64_t trec_user_prep_workspace_bytes(int64_t n)
{
    const int64_t n_blk = ceil_div64(n > 0 ? n : 1, UP_TB);
    return ceil_div64(n * 4, 256) * 256 + ceil_div64(n, 256) * 256 + n_blk * UP_CLASSES * 4 + 256;
}

// The whole user side of the cascade (see the header).  repr [n, d] fp32 in the caller's order; kpad in {32, 64, 128};
// n_alloc = trec_user_prep_alloc_rows(n, wg_rows).  Outputs, all in LAYOUT order unless said otherwise:
//   src [n_alloc] int32 (the caller's row of a layout row, -1: none), pos [n] int32 (the layout row of caller's row u),
//   wg_scale / wg_class [n_alloc / wg_rows] (0 / -1: an idle workgroup), ladder [64], class_used [64] int32,
//   gmax [1] (ZEROED BY THE CALLER), meta int32[2] = {padded rows, n},
//   out_f32 [n_alloc, kpad] (NULL: not wanted), out_bf16, row_stats [n_alloc][2], out_q int8, row_stats8 [n_alloc][2],
//   bias_sorted [n_alloc] (NULL without a user bias).
extern "C" int trec_user_prep_sorted(const float* repr, int64_t n, int32_t d, int32_t kpad, int32_t normalize,
                                     const float* user_bias, int32_t wg_rows, int64_t n_alloc, void* workspace,
                                     int64_t workspace_bytes, int32_t* src, int32_t* pos, float* wg_scale, int32_t* wg_class,
                                     float* ladder, int32_t* class_used, float* gmax, int32_t* meta, float* out_f32,
                                     void* out_bf16, float* row_stats, void* out_q, float* row_stats8, float* bias_sorted,
                                     void* stream)
{
    TREC_REQUIRE(repr && workspace && src && pos && wg_scale && wg_class && ladder && class_used && gmax && meta && out_bf16 &&
                 row_stats && out_q && row_stats8, "trec_user_prep_sorted: null pointer");
    TREC_REQUIRE(n >= 1 && n < ((int64_t)1 << 31) - 65536, "trec_user_prep_sorted: need 1 <= n < 2^31");
    TREC_REQUIRE(d >= 1 && kpad >= d && (kpad == 32 || kpad == 64 || kpad == 128), "trec_user_prep_sorted: kpad must be 32, 64 or 128 (>= d)");
    TREC_REQUIRE(((uintptr_t)repr % 16) == 0 || (d & 3) != 0, "trec_user_prep_sorted: repr must be 16-byte aligned");
    TREC_REQUIRE(wg_rows >= 1 && n_alloc == trec_user_prep_alloc_rows(n, wg_rows), "trec_user_prep_sorted: n_alloc must be trec_user_prep_alloc_rows(n, wg_rows)");
    TREC_REQUIRE(workspace_bytes >= trec_user_prep_workspace_bytes(n), "trec_user_prep_sorted: workspace too small");
    TREC_REQUIRE(!bias_sorted || user_bias, "trec_user_prep_sorted: bias_sorted without a user bias");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* nat = (float*)ws;                      ws += ceil_div64(n * 4, 256) * 256;
    unsigned char* cls = (unsigned char*)ws;      ws += ceil_div64(n, 256) * 256;
    int32_t* blockhist = (int32_t*)ws;
    const int n_blk = (int)ceil_div64(n, UP_TB);
    const int g = kpad / 4;                                               // 8, 16 or 32 lanes per row
    {
        const unsigned blocks = (unsigned)ceil_div64(n * g, 256);
        if (g == 32) hipLaunchKernelGGL(natscale_kernel<32>, dim3(blocks), dim3(256), 0, st, repr, n, d, normalize, nat, gmax);
        else if (g == 16) hipLaunchKernelGGL(natscale_kernel<16>, dim3(blocks), dim3(256), 0, st, repr, n, d, normalize, nat, gmax);
        else hipLaunchKernelGGL(natscale_kernel<8>, dim3(blocks), dim3(256), 0, st, repr, n, d, normalize, nat, gmax);
    }
    hipLaunchKernelGGL(class_hist_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, nat, gmax, n, cls, blockhist, src, n_alloc);
    hipLaunchKernelGGL(class_scan_kernel, dim3(1), dim3(1024), 0, st, blockhist, n_blk, gmax, n, wg_rows, (int32_t)(n_alloc / wg_rows),
                       wg_scale, wg_class, ladder, class_used, meta);
    hipLaunchKernelGGL(class_place_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, cls, blockhist, n, pos, src);
    {
        const unsigned blocks = (unsigned)ceil_div64(n_alloc * g, 256);
#define TREC_PS(GV) hipLaunchKernelGGL(prep_sorted_kernel<GV>, dim3(blocks), dim3(256), 0, st, repr, n_alloc, d, kpad, normalize, src, wg_scale, \
                                       wg_rows, user_bias, out_f32, (unsigned short*)out_bf16, (float2*)row_stats, (signed char*)out_q,        \
                                       (float2*)row_stats8, bias_sorted)
        if (g == 32) TREC_PS(32);
        else if (g == 16) TREC_PS(16);
        else TREC_PS(8);
#undef TREC_PS
    }
    return trec_check_launch("trec_user_prep_sorted");
}

// n bytes of zeros on the stream (the counters and maxima a call starts from, as ONE block: hipMemsetAsync)
extern "C" int trec_fill_zero(void* p, int64_t nbytes, void* stream)
{
    TREC_REQUIRE(p || nbytes == 0, "trec_fill_zero: null pointer");
    if (nbytes <= 0) return TREC_OK;
    if (hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream) != hipSuccess) {
        trec_set_last_error("trec_fill_zero: hipMemsetAsync failed");
        return TREC_ERR_LAUNCH;
    }
    return TREC_OK;
}
