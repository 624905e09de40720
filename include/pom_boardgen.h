/*
 * pom_boardgen.h — the start board of (seed, env, episode): the specification the device generator
 * (pom_batch_generate, fresh boards on auto-reset), the CPU checker (oracle/pom_boardgen_oracle.c) and the
 * tests share, like pom_rng.h does for moves.
 *
 * The reference's generator (State::Init / InitBoardItems, /root/reference/src/bboard/bboard.cpp:59-87,339-382) draws from
 * std::mt19937_64 through libstdc++'s uniform_int_distribution and reads an uninitialised queue slot: its
 * stream cannot be reproduced, its DISTRIBUTION is:
 *   1. every cell independently: passage 5/7, rigid 1/7, wood 1/7                              (bboard.cpp:59-74,349,357-358)
 *   2. ceil(woods / 2) of the wood cells, chosen uniformly, get a flag uniform in {1,2,3,4}
 *      (1 extra-bomb, 2 incr-range, 3 kick, 4 = nothing: FlagItem, bboard.cpp:182-189)          (bboard.cpp:367-381)
 *   3. agents 0..3 are written over the corners (0,0) (10,0) (10,10) (0,10), nothing is cleared around them (:322-333)
 *   4. everything else is a fresh State: aliveAgents 4, maxBombCount 1, bombStrength 1, every flame slot timeLeft 4
 *      (bboard.hpp:234-239,345,370)
 *
 * Here every random choice is a pure function of (seed, env, episode, draw index), 32-bit arithmetic only:
 *   key        = pom_board_key(seed, env, episode)        env = global env index, episode = 0 for the first game of an env
 *   draw(i)    = pom_board_draw(key, i)                   uniform 32 bits
 *   below(d,n) = pom_mulhi32(d, n)                        uniform in [0, n)
 *   step 1:  cell c = y*11+x, c = 0..120:  t = below(draw(c), 7);  t == 1 rigid, t == 2 wood, else passage
 *   step 2:  selection sampling over the wood cells in ascending c: with `left` woods not yet visited (this one included)
 *            and `need` flags not yet given (ceil(woods/2) at the start): the cell is chosen iff
 *            below(draw(128 + c), left) < need; a chosen cell gets flag 1 + (draw(256 + c) >> 30).
 *            Exactly ceil(woods/2) cells are chosen, every subset of that size equally likely.
 *   step 3, 4 as above.
 */
#ifndef POM_BOARDGEN_H_
#define POM_BOARDGEN_H_

#include <stdint.h>

#include "pom_rng.h"

POM_HD uint32_t pom_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

POM_HD uint32_t pom_board_key(uint64_t seed, uint32_t env, uint32_t episode)
{
    const uint32_t k = pom_fmix32(episode * 0x7FEB352Du + (uint32_t)(seed >> 32) + 0x5BD1E995u);
    return pom_fmix32((uint32_t)seed ^ (env * 0x9E3779B1u) ^ k);
}

POM_HD uint32_t pom_board_draw(uint32_t key, uint32_t i) { return pom_fmix32(key + (i + 1u) * 0x9E3779B9u); }

enum { POM_BOARD_DRAW_SELECT = 128, POM_BOARD_DRAW_FLAG = 256 };

#endif /* POM_BOARDGEN_H_ */
