// evg_rng.h -- counter-based RNG of the build (device side): Philox4x32-10, keyed exactly as
// oracle/rng_spec.py states it (seed, env id, episode, turn, node, player, group, block).
// Replaces the reference's unseeded global numpy stream (server.py:205,338,562).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evg {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;
constexpr int RNG_COMBAT = 0, RNG_ACTION = 1, RNG_SWARM = 2, RNG_DELAY = 3, RNG_EXPLORE = 4;

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a v_mul_hi_u32 / v_mul_lo_u32 pair
        const uint64_t p0 = (uint64_t)PHILOX_M0 * (uint64_t)c.x, p1 = (uint64_t)PHILOX_M1 * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return c;
}

// The ten round keys of a Philox block, (seed_lo + r W0, seed_hi + r W1), are constants of a handle: tabulated by evg_create (DevState::keys), so that a block
// in the step kernel costs a few scalar loads from the argument segment instead of eighteen scalar adds (round 5: 70 of the 486 scalar instructions a wavefront
// issued per turn were this key schedule).  k[2 r], k[2 r + 1] = the key of round r.
struct PhiloxKeys { uint32_t k[20]; };

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, const uint32_t (&rk)[20]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * (uint64_t)c.x, p1 = (uint64_t)PHILOX_M1 * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ rk[2 * r], lo1, hi0 ^ c.w ^ rk[2 * r + 1], lo0);
    }
    return c;
}

__device__ __forceinline__ uint4 rng_block(const uint32_t (&rk)[20], uint32_t env_id, uint32_t episode, int domain, uint32_t block, int turn, int node,
                                           int player, int group) {
    const uint4 ctr = make_uint4((block & 0x0FFFFFFFu) | ((uint32_t)domain << 28),
                                 ((uint32_t)turn & 0xFFu) | (((uint32_t)node & 0xFu) << 8) | (((uint32_t)player & 1u) << 12) |
                                     (((uint32_t)group & 0xFu) << 16),
                                 episode, env_id);
    return philox4x32_10(ctr, rk);
}

__device__ __forceinline__ uint4 rng_block(uint32_t seed_lo, uint32_t seed_hi, uint32_t env_id, uint32_t episode,
                                           int domain, uint32_t block, int turn, int node, int player, int group) {
    const uint4 ctr = make_uint4((block & 0x0FFFFFFFu) | ((uint32_t)domain << 28),
                                 ((uint32_t)turn & 0xFFu) | (((uint32_t)node & 0xFu) << 8) | (((uint32_t)player & 1u) << 12) |
                                     (((uint32_t)group & 0xFu) << 16),
                                 episode, env_id);
    return philox4x32_10(ctr, seed_lo, seed_hi);
}

// the eight 16-bit draws of a block (oracle/rng_spec.py `halves`): half h = bits 16*(h & 1) .. +15 of word h >> 1; h is a
// compile-time constant at every call site
__device__ __forceinline__ uint32_t rng_half(const uint32_t (&w)[4], int h) { return (h & 1) ? (w[h >> 1] >> 16) : (w[h >> 1] & 0xFFFFu); }

}  // namespace evg
