// Probe of v_mfma_i32_32x32x32_i8's operand layout (run once on the GPU box): D = A . B^T for A [32 rows][32 k] int8 and
// B [32 cols][32 k] int8 with lane l holding 16 consecutive k of row / column l & 31: k = 16 * (l >> 5) + 0..15.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t* A, const int8_t* B, int* D)
{
    const int l = threadIdx.x, half = l >> 5, l31 = l & 31;
    v4i a = *(const v4i*)(A + l31 * 32 + half * 16);
    v4i b = *(const v4i*)(B + l31 * 32 + half * 16);
    v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = c[r];
}
int main()
{
    int8_t hA[32 * 32], hB[32 * 32];
    int hD[32 * 32], ref[32 * 32];
    unsigned s = 12345;
    for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (int8_t)((s >> 16) % 255 - 127); s = s * 1664525u + 1013904223u; hB[i] = (int8_t)((s >> 16) % 255 - 127); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int acc = 0; for (int kk = 0; kk < 32; ++kk) acc += (int)hA[i * 32 + kk] * (int)hB[j * 32 + kk]; ref[i * 32 + j] = acc; }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 layout probe: %d mismatches of 1024 (D[row = item][col = user], k = 16*half + 0..15)\n", bad);
    if (bad) { for (int i = 0; i < 4; ++i) printf("  got %d ref %d\n", hD[i], ref[i]); }
    return bad != 0;
}
