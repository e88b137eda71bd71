// What does a random ROW gather deliver when the table sits in the L2 / the Infinity Cache / HBM?  (VERDICT r4 "next" 5 and 8: the
// ceiling configs[4]'s gathers and a cache-blocked WMRB step would be priced against.)
//
// One wave per gathered row-group: a wave fetches ROWS_PER_WAVE consecutive random rows per iteration, 16 bytes per lane
// (512-byte rows: 32 lanes per row, two rows per wave-load; 1024-byte rows: one row per wave-load), 8 loads in flight per lane,
// accumulates into registers and writes one float per wave at the end -- the access pattern of spmm_csr_vec4 / wmrb_user_fused
// without their arithmetic.  Row ids come from a pre-generated int32 list (streamed, 4 bytes per 512 / 1024 gathered).
//
// usage: gather_ceiling [row_bytes=512] [pairs=60000000]     prints one JSON line per table size
//        gather_ceiling 0 [pairs]                         bandwidth against the bytes in flight per CU (512-byte rows, 512 MB table)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROW_BYTES, int INFLIGHT = 8>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, int64_t n_ids,
                                                     float* __restrict__ out)
{
    constexpr int LANES_PER_ROW = ROW_BYTES / 16;            // 32 (512 B) or 64 (1 KB)
    constexpr int ROWS_PER_LOAD = 64 / LANES_PER_ROW;        // rows one wave-load covers
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
    const int sub = lane / LANES_PER_ROW, col = lane % LANES_PER_ROW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t per_iter = (int64_t)INFLIGHT * ROWS_PER_LOAD;
    for (int64_t base = wave * per_iter; base + per_iter <= n_ids; base += n_waves * per_iter) {
        int32_t id[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) id[j] = ids[base + j * ROWS_PER_LOAD + sub];
        f32x4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) v[j] = *(const f32x4*)(table + (int64_t)id[j] * (ROW_BYTES / 4) + col * 4);
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += v[j];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[wave] = s;
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t xorshift() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

// bandwidth against the bytes a CU keeps in flight: 512-byte rows from a 512 MB table, wgs workgroups of 4 waves per CU, each lane
// INFLIGHT 16-byte loads outstanding (one wave-load = 1 KB)
template <int INFLIGHT>
void sweep_one(const float* d_table, const int32_t* d_ids, int64_t n_pairs, float* d_out, int wgs)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wgs;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_kernel<512, INFLIGHT>), dim3(blocks), dim3(256), 0, 0, d_table, d_ids, n_pairs, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((gather_kernel<512, INFLIGHT>), dim3(blocks), dim3(256), 0, 0, d_table, d_ids, n_pairs, d_out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("{\"sweep\": \"inflight\", \"workgroups_per_cu\": %d, \"loads_per_lane\": %d, \"kb_in_flight_per_cu\": %d, \"ms\": %.3f, \"gathered_gb_per_s\": %.1f}\n",
           wgs, INFLIGHT, wgs * 4 * INFLIGHT, ms, (double)n_pairs * 512 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

void sweep(int64_t n_pairs)
{
    int32_t* d_ids; float* d_out; float* d_table;
    hipMalloc(&d_table, 512ll << 20); hipMemset(d_table, 0, 512ll << 20);
    hipMalloc(&d_ids, n_pairs * 4);
    hipMalloc(&d_out, (size_t)256 * 16 * 4 * 4);
    std::vector<int32_t> ids(n_pairs);
    for (int64_t i = 0; i < n_pairs; ++i) ids[i] = (int32_t)(xorshift() % (uint64_t)(1 << 20));
    hipMemcpy(d_ids, ids.data(), n_pairs * 4, hipMemcpyHostToDevice);
    for (int wgs : {1, 2, 4, 8, 16}) {
        sweep_one<2>(d_table, d_ids, n_pairs, d_out, wgs);
        sweep_one<4>(d_table, d_ids, n_pairs, d_out, wgs);
        sweep_one<8>(d_table, d_ids, n_pairs, d_out, wgs);
        sweep_one<16>(d_table, d_ids, n_pairs, d_out, wgs);
    }
}

template <int ROW_BYTES>
void run(int64_t n_pairs)
{
    const double mbs[] = {4, 16, 27, 64, 110, 128, 192, 256, 384, 512, 1024};
    int32_t* d_ids; float* d_out; float* d_table;
    const int64_t max_bytes = 1024ll << 20;
    hipMalloc(&d_table, max_bytes);
    hipMemset(d_table, 0, max_bytes);
    hipMalloc(&d_ids, n_pairs * 4);
    const int blocks = 256 * 8;                              // 8 workgroups of 4 waves per CU
    hipMalloc(&d_out, (size_t)blocks * 4 * 4);
    std::vector<int32_t> ids(n_pairs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double mb : mbs) {
        const int64_t rows = (int64_t)(mb * 1048576.0) / ROW_BYTES;
        for (int64_t i = 0; i < n_pairs; ++i) ids[i] = (int32_t)(xorshift() % (uint64_t)rows);
        hipMemcpy(d_ids, ids.data(), n_pairs * 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(gather_kernel<ROW_BYTES>, dim3(blocks), dim3(256), 0, 0, d_table, d_ids, n_pairs, d_out);
        hipDeviceSynchronize();
        const int reps = 5;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gather_kernel<ROW_BYTES>, dim3(blocks), dim3(256), 0, 0, d_table, d_ids, n_pairs, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("{\"row_bytes\": %d, \"table_mb\": %.0f, \"rows\": %lld, \"gathers\": %lld, \"ms\": %.3f, \"gathered_gb_per_s\": %.1f}\n", ROW_BYTES, mb,
               (long long)rows, (long long)n_pairs, ms, (double)n_pairs * ROW_BYTES / (ms * 1e-3) / 1e9);
        fflush(stdout);
    }
    hipFree(d_table); hipFree(d_ids); hipFree(d_out);
}

int main(int argc, char** argv)
{
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 512;
    const int64_t n_pairs = argc > 2 ? atoll(argv[2]) : 60000000ll;
    if (row_bytes == 0) sweep(n_pairs);
    else if (row_bytes == 512) run<512>(n_pairs);
    else if (row_bytes == 1024) run<1024>(n_pairs);
    else { fprintf(stderr, "row_bytes must be 512 or 1024\n"); return 2; }
    return 0;
}
