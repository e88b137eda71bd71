// Bare MFMA streams at the power cap: what rate does the chip sustain on random operands with nothing else going on?
//   i8 32x32x32, i8 16x16x64, bf16 32x32x16 -- 8 waves per CU, 4 (or 8) independent accumulator chains per wave, ~1 s each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void stream(const int* __restrict__ src, int* __restrict__ out, int iters, int mask)
{
    const int l = threadIdx.x + blockIdx.x * 256;
    v4i a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *(const v4i*)(src + ((l * 8 + i) * 4) % 65536);
        b[i] = *(const v4i*)(src + ((l * 8 + 4 + i) * 4) % 65536);
        for (int e = 0; e < 4; ++e) { a[i][e] &= mask; b[i][e] &= mask; }
    }
    if (MODE == 0) {
        v16i c[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], b[i], c[i], 0, 0, 0);
        int s = 0;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
        out[l] = s;
    } else if (MODE == 1) {
        v4i c[8] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j], b[i & 3], c[i], 0, 0, 0);
        int s = 0;
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += c[i][e];
        out[l] = s;
    } else if (MODE == 3) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f c[8] = {};
        bf16x8 fa[4], fb[4];
        for (int i = 0; i < 4; ++i) { fa[i] = __builtin_bit_cast(bf16x8, a[i]); fb[i] = __builtin_bit_cast(bf16x8, b[i]); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j], fb[i & 3], c[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += c[i][e];
        out[l] = (int)s;
    } else {
        v16f c[4] = {};
        bf16x8 fa[4], fb[4];
        for (int i = 0; i < 4; ++i) { fa[i] = __builtin_bit_cast(bf16x8, a[i]); fb[i] = __builtin_bit_cast(bf16x8, b[i]); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[i], c[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
        out[l] = (int)s;
    }
}

int main(int argc, char** argv)
{
    int* src; int* out;
    const int blocks = 256 * 2;
    hipMalloc(&src, 65536 * 4 + 64); hipMalloc(&out, blocks * 256 * 4);
    int* h = (int*)malloc(65536 * 4);
    unsigned s = 99;
    for (int i = 0; i < 65536; ++i) { s = s * 1664525u + 1013904223u; unsigned v = s; s = s * 1664525u + 1013904223u; h[i] = (int)((v & 0xffff0000u) | (s >> 16)); }
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int mode; const char* name; double ops; int mask; } cases[] = {
        {0, "i8 32x32x32, full-range operands", 65536.0 * 16, -1},
        {0, "i8 32x32x32, operands masked to +-31 (0x1f1f1f1f)", 65536.0 * 16, 0x1f1f1f1f},
        {1, "i8 16x16x64, full-range operands", 32768.0 * 32, -1},
        {2, "bf16 32x32x16, random bit patterns masked to finite (0x3fff3fff)", 32768.0 * 16, 0x3fff3fff},
        {3, "bf16 16x16x32, random bit patterns masked to finite (0x3fff3fff)", 16384.0 * 32, 0x3fff3fff},
        {2, "bf16 32x32x16, again", 32768.0 * 16, 0x3fff3fff},
        {1, "i8 16x16x64, again", 32768.0 * 32, -1},
    };
    for (auto& c : cases) {
        const int iters = argc > 1 ? atoi(argv[1]) : 60000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (c.mode == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters, c.mask);
            else if (c.mode == 1) hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters, c.mask);
            else if (c.mode == 3) hipLaunchKernelGGL(stream<3>, dim3(blocks), dim3(256), 0, 0, src, out, iters, c.mask);
            else hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(256), 0, 0, src, out, iters, c.mask);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double total = c.ops * iters * (double)blocks * 4;       // per wave per iteration: c.ops
            if (rep == 1) printf("%-70s %8.1f ms  %7.1f Tops/s\n", c.name, ms, total / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
