// tensorrec_amd/csrc/topk_common.hpp -- helpers shared by the top-k kernels (score_gemm.hip, topk_filter.hip):
// the (value desc, index asc) order of tf.nn.top_k (recommendation_graphs.py:80) as ONE unsigned 64-bit key, and its
// wave-wide maximum by DPP.
#pragma once
#include "common.hpp"
#include <math.h>

// largest float strictly below x (x finite or -inf, never NaN): v >= x  <=>  v > float_pred(x)
__device__ __forceinline__ float float_pred(float x)
{
    const unsigned int u = __float_as_uint(x);
    if (x == -INFINITY) return x;
    if ((u << 1) == 0u) return __uint_as_float(0x80000001u);
    return __uint_as_float(x > 0.f ? u - 1u : u + 1u);
}


__device__ __forceinline__ unsigned long long merge_key(float v, int32_t id)
{
    const unsigned int u = (v == 0.f) ? 0u : __float_as_uint(v);            // -0.0 and +0.0 compare equal: one key
    const unsigned int hi = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)hi << 32) | (unsigned int)(~id);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long x)
{
    const int lo = (int)(unsigned int)x, hi = (int)(unsigned int)(x >> 32);
    const unsigned int tlo = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned int thi = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long t = ((unsigned long long)thi << 32) | tlo;
    return t > x ? t : x;
}

// maximum over the 64 lanes, returned in every lane
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long x)
{
    x = dpp_max_u64<0x128, 0xf>(x);      // row_ror:8
    x = dpp_max_u64<0x124, 0xf>(x);      // row_ror:4
    x = dpp_max_u64<0x122, 0xf>(x);      // row_ror:2
    x = dpp_max_u64<0x121, 0xf>(x);      // row_ror:1   -> every lane: maximum of its 16-lane row
    x = dpp_max_u64<0x142, 0xa>(x);      // row_bcast15 -> rows 1, 3 also cover rows 0, 2
    x = dpp_max_u64<0x143, 0xc>(x);      // row_bcast31 -> row 3 covers all four rows
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)x, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(x >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

