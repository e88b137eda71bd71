// tensorrec_amd/csrc/common.hpp -- shared helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is portable to other targets on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TREC_WAVE 64

// status codes returned by every extern "C" entry point (include/tensorrec_hip.h)
#define TREC_OK 0
#define TREC_ERR_INVALID 1   // bad argument (null pointer, unsupported size, ...)
#define TREC_ERR_LAUNCH 2    // hipGetLastError() != hipSuccess after a launch
#define TREC_ERR_UNSUPPORTED 3

extern "C" void trec_set_last_error(const char* msg);
// tuning knobs (benchmark-only switches between kernel variants); unknown names read as `dflt`
extern "C" int trec_get_tuning(const char* name, int dflt);

static inline int trec_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        trec_set_last_error(buf);
        return TREC_ERR_LAUNCH;
    }
    return TREC_OK;
}

#define TREC_REQUIRE(cond, msg)            \
    do {                                   \
        if (!(cond)) {                     \
            trec_set_last_error(msg);      \
            return TREC_ERR_INVALID;       \
        }                                  \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
// two fp32 -> packed bf16 pair (low half = a) in ONE instruction: v_cvt_pk_bf16_f32 (round-to-nearest-even, the same
// bits as f32_to_bf16_rne for every non-NaN input)
typedef __bf16 trec_bf16x2 __attribute__((ext_vector_type(2)));
typedef float trec_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int f32x2_to_bf16x2_bits(float a, float b) {
    const trec_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, trec_bf16x2));
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
