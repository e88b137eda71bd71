"""
oracle/device_sampler.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Integer restatement (NumPy, vectorised) of tensorrec_amd/csrc/sampler.hip so that the device sampler -- which has
no counterpart in the reference beyond its contract (util.py:12-21: [n_users, S] items, uniform over all items,
distinct per user when replace is False) -- can be checked bit-for-bit.  Philox4x32-10 follows the published
algorithm (Salmon et al., SC'11: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, np.uint32).copy() for x in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & U32).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & U32).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _feistel_f(r, key):
    with np.errstate(over="ignore"):
        h = (r * np.uint32(0x9E3779B1) + key).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h * np.uint32(0x85EBCA77)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xC2B2AE3D)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def _feistel_permute(x, bits, keys):
    wl, wr = bits >> 1, bits - (bits >> 1)
    L = x >> np.uint32(wr)
    R = x & np.uint32((1 << wr) - 1)
    for r in range(6):
        nl = R
        nr = (L ^ _feistel_f(R, keys[r])) & np.uint32((1 << wl) - 1)
        L, R = nl, nr
        wl, wr = wr, wl
    return (L << np.uint32(wr)) | R


def sample_items(n_users, n_items, n_sampled, replace, seed, step, user_base=0):
    seed = int(seed) & (2 ** 64 - 1)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    u = np.repeat(np.arange(n_users, dtype=np.uint64) + np.uint64(user_base), n_sampled)
    s = np.tile(np.arange(n_sampled, dtype=np.uint32), n_users)
    u_lo, u_hi = (u & U32).astype(np.uint32), (u >> np.uint64(32)).astype(np.uint32)
    stepv = np.full(u.shape, step, np.uint32)
    if replace:
        r = philox4x32_10(u_lo, s >> np.uint32(2), stepv, np.uint32(2) + u_hi, k0, k1)
        w = np.choose(s & np.uint32(3), r)
        out = ((w.astype(np.uint64) * np.uint64(n_items)) >> np.uint64(32)).astype(np.int32)
        return out.reshape(n_users, n_sampled)
    bits = 2
    while bits < 31 and (1 << bits) < n_items:
        bits += 1
    ka = philox4x32_10(u_lo, u_hi, stepv, np.zeros_like(u_lo), k0, k1)
    kb = philox4x32_10(u_lo, u_hi, stepv, np.ones_like(u_lo), k0, k1)
    keys = [ka[0], ka[1], ka[2], ka[3], kb[0], kb[1]]
    x = s.copy()
    todo = np.ones(x.shape, bool)
    while todo.any():                                   # cycle-walk until every value lands in [0, n_items)
        x[todo] = _feistel_permute(x[todo], bits, [k[todo] for k in keys])
        todo = x >= np.uint32(n_items)
    return x.astype(np.int32).reshape(n_users, n_sampled)
