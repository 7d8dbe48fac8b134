// evg_mt.h -- stock-entropy compatibility mode (rng_mode = EVG_RNG_STOCK_MT19937), device side.
//
// The unmodified reference draws from numpy's legacy global generator: MT19937 seeded by np.random.seed(int)
// (init_genrand), np.random.randint(n) = masked rejection over 32-bit outputs, nothing consumed when n == 1
// (server.py:205, :338, :562).  Here every env owns one such generator: 624 key words + a position, stored
// struct-of-arrays (`mt_key[i][N]`, env fastest, so the lanes of a wavefront touch consecutive addresses) -- 2.5 KB
// per env, allocated only in this mode.  The stream is inherently sequential per env: this mode exists to replay the
// reference bit for bit under np.random.seed(s), not for speed (the keyed Philox mode of evg_rng.h is the fast path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evg {

constexpr int MT_N = 624, MT_M = 397;

struct MtGen {
    uint32_t* key;       // this env's column: word i at key[i * stride]
    size_t    stride;    // = number of envs
    uint32_t  pos;       // next word to temper; 624 = regenerate first
};

__device__ __forceinline__ uint32_t mt_mix(uint32_t hi, uint32_t lo, uint32_t far_) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7FFFFFFFu);
    return far_ ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}

// init_genrand: np.random.seed(seed) for a 32-bit integer seed
__device__ inline void mt_seed(MtGen& g, uint32_t seed) {
    for (int i = 0; i < MT_N; ++i) {
        g.key[(size_t)i * g.stride] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    g.pos = MT_N;
}

__device__ inline void mt_regenerate(MtGen& g) {
    uint32_t* k = g.key;
    const size_t s = g.stride;
    const uint32_t first = k[0];
    uint32_t cur = first;
    int i = 0;
    for (; i < MT_N - MT_M; ++i) { const uint32_t nxt = k[(size_t)(i + 1) * s]; k[(size_t)i * s] = mt_mix(cur, nxt, k[(size_t)(i + MT_M) * s]); cur = nxt; }
    for (; i < MT_N - 1; ++i) { const uint32_t nxt = k[(size_t)(i + 1) * s]; k[(size_t)i * s] = mt_mix(cur, nxt, k[(size_t)(i + MT_M - MT_N) * s]); cur = nxt; }
    k[(size_t)(MT_N - 1) * s] = mt_mix(cur, k[0], k[(size_t)(MT_M - 1) * s]);
    g.pos = 0;
}

__device__ inline uint32_t mt_next(MtGen& g) {
    if (g.pos >= (uint32_t)MT_N) mt_regenerate(g);
    uint32_t y = g.key[(size_t)g.pos * g.stride];
    g.pos += 1u;
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}

// RandomState.randint(n), 1 <= n <= 2^32: smallest all-ones mask >= n - 1, redraw while above
__device__ inline uint32_t mt_randint(MtGen& g, uint32_t n) {
    const uint32_t rng = n - 1u;
    if (rng == 0u) return 0u;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_next(g) & mask; } while (v > rng);
    return v;
}

}  // namespace evg
