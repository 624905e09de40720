/*
 * pom_rng.h — the counter-based synthetic move stream shared by the device
 * stepper (pom_batch_step_random), the CPU baseline and the tests, so that a
 * CPU run and a GPU run on the same (seed, env, tick) see identical Move[4].
 *
 * The reference draws moves from std::mt19937_64 seeded by random_device
 * (/root/reference/src/agents/basic_agents.cpp:12-38), which is neither
 * reproducible nor parallel; only its *distributions* are kept:
 *   POM_DIST_HARMLESS  uniform {IDLE,UP,DOWN,LEFT,RIGHT}       (HarmlessAgent, :28-38)
 *   POM_DIST_RANDOM    uniform {IDLE,UP,DOWN,LEFT,RIGHT,BOMB}  (RandomAgent,   :12-22)
 *   POM_DIST_STRESS    BOMB 30 %, each direction 15 %, IDLE 10 %
 */
#ifndef POM_RNG_H_
#define POM_RNG_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define POM_HD __host__ __device__ __attribute__((always_inline)) inline
#elif defined(__cplusplus)
#define POM_HD inline
#else
#define POM_HD static inline
#endif

enum { POM_DIST_HARMLESS = 0, POM_DIST_RANDOM = 1, POM_DIST_STRESS = 2 };

POM_HD uint64_t pom_splitmix64(uint64_t z) /* host-side seeding helper, not on the per-tick path */
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* murmur3's 32-bit finaliser: 2 multiplies, 3 xor-shifts — 32-bit ops only, which is what a gfx950 VALU lane has */
POM_HD uint32_t pom_fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* one 64-bit draw per (seed, env, tick), 16 bits per agent: two independent 32-bit halves (agents 0, 1 / agents 2, 3), so that a
 * device lane that speaks for one agent computes only the half it needs */
POM_HD uint32_t pom_rng_draw_half(uint64_t seed, uint32_t env, uint32_t tick, int upper)
{
    const uint32_t k = (uint32_t)seed ^ (env * 0x9E3779B1u) ^ (tick * 0x7FEB352Du + (uint32_t)(seed >> 32));
    return pom_fmix32(upper ? k ^ 0x68E31DA4u : k);
}
POM_HD uint64_t pom_rng_draw(uint64_t seed, uint32_t env, uint32_t tick)
{
    return (uint64_t)pom_rng_draw_half(seed, env, tick, 0) | ((uint64_t)pom_rng_draw_half(seed, env, tick, 1) << 32);
}

POM_HD int32_t pom_rng_pick(uint32_t r16, int dist)
{
    if (dist == POM_DIST_STRESS) {
        /* cumulative /65536: IDLE .10, UP .25, DOWN .40, LEFT .55, RIGHT .70, BOMB 1 */
        if (r16 < 6554u) return 0;
        if (r16 < 16384u) return 1;
        if (r16 < 26214u) return 2;
        if (r16 < 36045u) return 3;
        if (r16 < 45875u) return 4;
        return 5;
    }
    uint32_t n = dist == POM_DIST_HARMLESS ? 5u : 6u;
    return (int32_t)((r16 * n) >> 16);
}

POM_HD void pom_rng_moves(uint64_t seed, uint32_t env, uint32_t tick, int dist, int32_t mv[4])
{
    uint64_t r = pom_rng_draw(seed, env, tick);
    mv[0] = pom_rng_pick((uint32_t)(r & 0xFFFF), dist);
    mv[1] = pom_rng_pick((uint32_t)((r >> 16) & 0xFFFF), dist);
    mv[2] = pom_rng_pick((uint32_t)((r >> 32) & 0xFFFF), dist);
    mv[3] = pom_rng_pick((uint32_t)((r >> 48) & 0xFFFF), dist);
}

#endif /* POM_RNG_H_ */
