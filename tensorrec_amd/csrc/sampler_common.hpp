// tensorrec_amd/csrc/sampler_common.hpp -- the generator behind K7 (csrc/sampler.hip): Philox4x32-10 keys and the keyed Feistel
// permutation of [0, n_items) whose images are a user's distinct samples.  Shared with the single-kernel training step
// (csrc/step_coop.hip), which draws a user's samples inside the kernel -- the same bits as trec_sample_items.
#pragma once
#include <stdint.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

struct u4 { uint32_t x, y, z, w; };

__host__ __device__ inline u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c.x, p1 = (uint64_t)PHILOX_M1 * c.z;
        u4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    return c;
}

__host__ __device__ inline uint32_t feistel_f(uint32_t r, uint32_t key)
{
    uint32_t h = r * 0x9E3779B1u + key;
    h ^= h >> 15; h *= 0x85EBCA77u;
    h ^= h >> 13; h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}

// one application of the keyed permutation on [0, 2^bits)
__host__ __device__ inline uint32_t feistel_permute(uint32_t x, int bits, const uint32_t* keys)
{
    int wl = bits >> 1, wr = bits - wl;                 // widths of (L, R)
    uint32_t L = x >> wr, R = x & ((1u << wr) - 1u);
    for (int r = 0; r < 6; ++r) {
        const uint32_t nl = R;
        const uint32_t nr = (L ^ feistel_f(R, keys[r])) & ((1u << wl) - 1u);
        L = nl; R = nr;
        const int t = wl; wl = wr; wr = t;
    }
    return (L << wr) | R;
}


// sample s of the stream (global user u, step): pi_u(s), cycle-walked back into [0, n_items)   (replace == 0)
struct SampleKeys { uint32_t k[6]; };
__host__ __device__ inline SampleKeys sample_keys(int64_t u, uint32_t step, uint32_t seed_lo, uint32_t seed_hi)
{
    const u4 ka = philox4x32_10(u4{(uint32_t)u, (uint32_t)(u >> 32), step, 0u}, seed_lo, seed_hi);
    const u4 kb = philox4x32_10(u4{(uint32_t)u, (uint32_t)(u >> 32), step, 1u}, seed_lo, seed_hi);
    return SampleKeys{{ka.x, ka.y, ka.z, ka.w, kb.x, kb.y}};
}
__host__ __device__ inline int32_t sample_distinct(uint32_t s, int bits, const SampleKeys& keys, int32_t n_items)
{
    uint32_t x = s;
    do { x = feistel_permute(x, bits, keys.k); } while (x >= (uint32_t)n_items);
    return (int32_t)x;
}
__host__ __device__ inline int sample_bits(int32_t n_items)
{
    int bits = 2;
    while (bits < 31 && (1u << bits) < (uint32_t)n_items) ++bits;
    return bits;
}
