// tensorrec_amd/csrc/sampler.hip -- K7: negative-item sampling on the device.
//
// Replaces the host callback sample_items (tensorrec/util.py:12-21, hooked in by tf.py_func at
// tensorrec/tensorrec.py:298-302): n_users separate np.random.choice calls, O(n_users * n_items) host work under
// the GIL every step.  Same contract: an int [n_users, n_sampled] table (user-major, util.py:16-19), uniform over ALL
// items (positives are not excluded, util.py:13), distinct within a user when replace == 0.
//
// NumPy's MT19937 stream cannot be reproduced by a parallel sampler, so this is a different (counter-based)
// generator with the same distribution; parity of a training step is checked with host-supplied samples
// ("replay"), and this kernel is checked bit-for-bit against its own integer restatement
// (oracle/device_sampler.py) plus distribution tests.
//
//   replace == 0 : sample s of user u is pi_u(s), where pi_u is a keyed pseudo-random PERMUTATION of
//                  [0, n_items): a 6-round unbalanced Feistel network on ceil(log2 n_items) bits with
//                  cycle-walking.  Distinctness is structural -- no rejection table, no memory, O(1) per sample.
//   replace == 1 : Philox4x32-10 word -> (word * n_items) >> 32.
// Keys come from Philox4x32-10 with counter (user, 0, step, stream) and key (seed_lo, seed_hi).
#include "common.hpp"

#include "sampler_common.hpp"

__global__ __launch_bounds__(256) void sample_items_kernel(int64_t n_users, int64_t user_base, int32_t n_items,
                                                          int32_t n_sampled, int replace, uint32_t seed_lo,
                                                          uint32_t seed_hi, uint32_t step, int bits,
                                                          int32_t* __restrict__ out,
                                                          const uint32_t* __restrict__ step_dev = nullptr)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_users * n_sampled) return;
    if (step_dev) step = *step_dev;             // the step counter lives on the device (HIP-graph replays)
    const int64_t ul = idx / n_sampled;
    const uint32_t s = (uint32_t)(idx - ul * n_sampled);
    const int64_t u = ul + user_base;          // streams are keyed by the GLOBAL user id: a user shard draws what the
                                               // whole-population run would draw for the same users
    if (replace) {
        const u4 r = philox4x32_10(u4{(uint32_t)u, s >> 2, step, 2u + (uint32_t)(u >> 32)}, seed_lo, seed_hi);
        const uint32_t w = (s & 3) == 0 ? r.x : (s & 3) == 1 ? r.y : (s & 3) == 2 ? r.z : r.w;
        out[idx] = (int32_t)(((uint64_t)w * (uint64_t)(uint32_t)n_items) >> 32);
        return;
    }
    const SampleKeys keys = sample_keys(u, step, seed_lo, seed_hi);
    out[idx] = sample_distinct(s, bits, keys, n_items);                         // cycle-walk back into [0, n_items)
}

// The same draw without replacement with the keys of a user made ONCE: the kernel above runs two Philox blocks (20 rounds) per SAMPLE
// for keys that depend on the user only -- 0.72 ms for 1M users x 100 samples, ten times what writing the 400 MB costs.  Here a
// workgroup owns SPB consecutive samples (a few users): one thread per user of the range makes its keys into LDS, then every thread
// walks its samples' Feistel rounds with them.  Bit-identical output.
constexpr int SAMPLER_SPB = 2048;       // samples per workgroup (8 per thread)
constexpr int SAMPLER_MAX_USERS = 256;  // users a workgroup's range may touch (n_sampled >= 8 keeps it below)

__global__ __launch_bounds__(256) void sample_items_keyed_kernel(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled,
                                                                uint32_t seed_lo, uint32_t seed_hi, uint32_t step, int bits,
                                                                int32_t* __restrict__ out, const uint32_t* __restrict__ step_dev)
{
    __shared__ uint32_t l_keys[SAMPLER_MAX_USERS + 1][6];
    if (step_dev) step = *step_dev;
    const int64_t total = n_users * (int64_t)n_sampled;
    const int64_t i0 = (int64_t)blockIdx.x * SAMPLER_SPB;
    const int64_t i1 = i0 + SAMPLER_SPB < total ? i0 + SAMPLER_SPB : total;
    const int64_t u_first = i0 / n_sampled, u_last = (i1 - 1) / n_sampled;
    for (int64_t t = threadIdx.x; t <= u_last - u_first; t += 256) {
        const SampleKeys keys = sample_keys(u_first + t + user_base, step, seed_lo, seed_hi);
#pragma unroll
        for (int q = 0; q < 6; ++q) l_keys[t][q] = keys.k[q];
    }
    __syncthreads();
    for (int64_t idx = i0 + threadIdx.x; idx < i1; idx += 256) {
        const int64_t ul = idx / n_sampled;
        const uint32_t s = (uint32_t)(idx - ul * n_sampled);
        SampleKeys keys;
#pragma unroll
        for (int q = 0; q < 6; ++q) keys.k[q] = l_keys[ul - u_first][q];
        out[idx] = sample_distinct(s, bits, keys, n_items);
    }
}

static int sample_items_impl(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled, int32_t replace,
                             uint64_t seed, uint32_t step, const uint32_t* step_dev, int32_t* out, void* stream);

extern "C" int trec_sample_items(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled,
                                 int32_t replace, uint64_t seed, uint32_t step, int32_t* out, void* stream)
{
    return sample_items_impl(n_users, user_base, n_items, n_sampled, replace, seed, step, nullptr, out, stream);
}

// the same draw with the step counter read from device memory at execution time, so that the launch can sit inside a
// HIP graph that is replayed every step (the counter is bumped by trec_adam_schedule_advance inside the same graph)
extern "C" int trec_sample_items_dev(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled,
                                     int32_t replace, uint64_t seed, const uint32_t* step_dev, int32_t* out,
                                     void* stream)
{
    TREC_REQUIRE(step_dev, "trec_sample_items_dev: null step counter");
    return sample_items_impl(n_users, user_base, n_items, n_sampled, replace, seed, 0u, step_dev, out, stream);
}

static int sample_items_impl(int64_t n_users, int64_t user_base, int32_t n_items, int32_t n_sampled, int32_t replace,
                             uint64_t seed, uint32_t step, const uint32_t* step_dev, int32_t* out, void* stream)
{
    TREC_REQUIRE(out, "trec_sample_items: null pointer");
    TREC_REQUIRE(n_items >= 1 && n_sampled >= 1, "trec_sample_items: n_items and n_sampled must be >= 1");
    // np.random.choice(replace=False) raises when size > population (util.py:13); same contract here
    TREC_REQUIRE(replace || n_sampled <= n_items, "trec_sample_items: cannot take a larger sample than population when replace is false");
    if (n_users == 0) return TREC_OK;
    const int bits = sample_bits(n_items);
    const int64_t total = n_users * (int64_t)n_sampled;
    if (!replace && n_sampled >= 8 && trec_get_tuning("sampler_keyed", 1) != 0) {
        hipLaunchKernelGGL(sample_items_keyed_kernel, dim3((unsigned)ceil_div64(total, SAMPLER_SPB)), dim3(256), 0, (hipStream_t)stream,
                           n_users, user_base, n_items, n_sampled, (uint32_t)seed, (uint32_t)(seed >> 32), step, bits, out, step_dev);
        return trec_check_launch("trec_sample_items");
    }
    hipLaunchKernelGGL(sample_items_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       n_users, user_base, n_items, n_sampled, replace, (uint32_t)seed, (uint32_t)(seed >> 32), step, bits,
                       out, step_dev);
    return trec_check_launch("trec_sample_items");
}
