/*
 * pom_boardgen_body.h — the pieces of the start-board specification (include/pom_boardgen.h) the device generator is made
 * of, each a pure function so that the identical source also runs on the host in tests/emul.  SURVEY.md §8 row f3.
 *
 * On gfx950 (pom_boardgen_wave in pom_kernels.h) a whole wavefront draws one env's board: lane l takes cells l and l+64
 * (pom_board_cell_kind), two ballots of "is wood" ARE the 121-bit wood set; every lane also computes its cells' selection
 * thresholds and flags (pom_board_threshold / pom_board_flag_code, two hashes each), so that the sequential part of the flag
 * pass — selection sampling over ~17 woods — is a countdown over ready-made numbers; lanes 0..48 write the other 49 dwords of
 * the fresh record (pom_fresh_row).  The host emulation does the same countdown with pom_board_flags.
 */
#ifndef POM_BOARDGEN_BODY_H_
#define POM_BOARDGEN_BODY_H_

#include "pom_boardgen.h"
#include "pom_packed.h"
#include "pom_step_body.h"

/* step 1 of the specification: 1 = rigid, 2 = wood, anything else = passage (ChooseItemOuter, bboard.cpp:59-74) */
POM_HD uint32_t pom_board_cell_kind(uint32_t key, int c) { return pom_mulhi32(pom_board_draw(key, (uint32_t)c), 7u); }
POM_HD int pom_board_cell_code(uint32_t kind) { return kind == 1u ? POM_C_RIGID : kind == 2u ? POM_C_WOOD : POM_C_PASSAGE; } /* the cell's code (pom_packed.h) */

/* step 2: which woods carry a flag.  Selection sampling in ascending cell order (exactly ceil(woods / 2) woods are chosen,
 * bboard.cpp:367-381): wood cell c, with `left` woods not yet visited (this one included), is chosen iff
 * pom_board_threshold(key, c, left) < need, where `need` counts down from ceil(woods / 2) with every choice.  The threshold
 * and the flag are pure functions of the cell — a wavefront computes them for all cells at once, only the countdown is
 * sequential. */
POM_HD uint32_t pom_board_threshold(uint32_t key, int c, int left)
{
    return pom_mulhi32(pom_board_draw(key, (uint32_t)(POM_BOARD_DRAW_SELECT + c)), (uint32_t)left);
}
POM_HD int pom_board_flag_code(uint32_t key, int c) { return POM_C_WOOD + 1 + (int)(pom_board_draw(key, (uint32_t)(POM_BOARD_DRAW_FLAG + c)) >> 30); }
/* the countdown done by one thread alone: w0 = wood cells 0..63, w1 = wood cells 64..120 (bit c - 64); put(c, code) rewrites a
 * chosen cell */
template <class Put>
POM_HD void pom_board_flags(uint32_t key, uint64_t w0, uint64_t w1, Put put)
{
    int left = __builtin_popcountll(w0) + __builtin_popcountll(w1);
    int need = (left + 1) >> 1;
    POM_NOUNROLL
    for (int half = 0; half < 2; half++) {
        uint64_t w = half ? w1 : w0;
        POM_NOUNROLL
        while (w != 0 && need > 0) { /* need <= left always: once they are equal every remaining wood is chosen */
            const int c = 64 * half + __builtin_ctzll(w);
            w &= w - 1;
            if ((int)pom_board_threshold(key, c, left) < need) {
                put(c, pom_board_flag_code(key, c));
                need--;
            }
            left--;
        }
    }
}

/* steps 3 and 4: dword r >= POM_REC_TIMESTEP of a fresh State with the agents in the corners (bboard.hpp:234-239,345,370;
 * PutAgentsInCorners, bboard.cpp:322-333), and the corner cells themselves */
POM_HD uint32_t pom_fresh_row(int r)
{
    if (r >= POM_REC_AGENTS && r < POM_REC_BOMBS) {
        const int i = (r - POM_REC_AGENTS) >> 1;
        if ((r - POM_REC_AGENTS) & 1) return 1u | (1u << 16); /* maxBombCount 1, bombStrength 1; top byte (flames.count / status / flags): 0 */
        const uint32_t x = (i == 1 || i == 2) ? POM_N - 1 : 0, y = (i == 2 || i == 3) ? POM_N - 1 : 0;
        return x | (y << 4) | (i == 0 ? 4u << 24 : 0u); /* agent 0's top byte: aliveAgents 4; the others' (queue indices and counts): 0 */
    }
    if (r >= POM_REC_FLAMES) return 4u << 16; /* Flame::timeLeft = 4 in every slot, live or not */
    return 0u;                                /* timeStep, bomb slots */
}
POM_HD int pom_corner_cell(int agent) { return agent == 0 ? 0 : agent == 1 ? POM_N - 1 : agent == 2 ? POM_CELLS - 1 : POM_CELLS - POM_N; }

#endif /* POM_BOARDGEN_BODY_H_ */
