/*
 * pom_boardgen_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of the start-board specification include/pom_boardgen.h, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to check the device generator.  The product
 * (pomcpp_amd/, libpom_batch.so) never includes, links or calls this file.
 *
 * What it follows in the reference (distribution only — the reference's own random stream is libstdc++-specific and its
 * generator reads an uninitialised slot, SURVEY.md §2, so there are no reference OUTPUTS to pin against; the distribution is
 * pinned by tests/test_boardgen.py):
 *   ChooseItemOuter / InitBoardItems   /root/reference/src/bboard/bboard.cpp:59-74, 346-382
 *   State::PutAgentsInCorners          /root/reference/src/bboard/bboard.cpp:322-333
 *   fresh State defaults               /root/reference/include/bboard.hpp:234-239, 345, 370
 */
#include "pom_boardgen_oracle.h"

#include <string.h>

#include "pom_boardgen.h"
#include "pom_oracle.h"
#include "pom_policy_oracle.h"
#include "pom_rng.h"

void pom_oracle_boardgen(uint64_t seed, uint32_t env, uint32_t episode, void* state_out)
{
    PomState* s = (PomState*)state_out;
    const uint32_t key = pom_board_key(seed, env, episode);
    /* a fresh State (what std::make_unique<State>() gives): zero except the default member initialisers */
    memset(s, 0, sizeof *s);
    s->aliveAgents = 4;                                  /* bboard.hpp:370 */
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        s->agents[i].maxBombCount = 1;                   /* bboard.hpp:235 */
        s->agents[i].bombStrength = 1;                   /* bboard.hpp:236 */
    }
    for (int k = 0; k < POM_MAX_BOMBS; k++) s->flames.queue[k].timeLeft = 4; /* bboard.hpp:345 */

    /* InitBoardItems, first loop (bboard.cpp:353-366): one draw in 0..6 per cell, row by row */
    int wood_cells[POM_CELLS], woods = 0;
    for (int c = 0; c < POM_CELLS; c++) {
        const uint32_t t = pom_mulhi32(pom_board_draw(key, (uint32_t)c), 7u);
        int32_t item = POM_PASSAGE;                      /* ChooseItemOuter, bboard.cpp:59-74: 1 -> rigid, 2 -> wood, else passage */
        if (t == 1u) item = POM_RIGID;
        if (t == 2u) {
            item = POM_WOOD;
            wood_cells[woods++] = c;
        }
        s->board[c / POM_BOARD_SIZE][c % POM_BOARD_SIZE] = item;
    }
    /* second loop (bboard.cpp:368-381): flags on ceil(woods / 2) distinct wood cells; here by selection sampling in cell order */
    int need = (woods + 1) / 2;
    for (int j = 0; j < woods; j++) {
        const int c = wood_cells[j];
        const uint32_t left = (uint32_t)(woods - j);
        if ((int)pom_mulhi32(pom_board_draw(key, (uint32_t)(POM_BOARD_DRAW_SELECT + c)), left) < need) {
            s->board[c / POM_BOARD_SIZE][c % POM_BOARD_SIZE] += 1 + (int32_t)(pom_board_draw(key, (uint32_t)(POM_BOARD_DRAW_FLAG + c)) >> 30);
            need--;
        }
    }
    /* InitState: PutAgentsInCorners(0, 1, 2, 3), bboard.cpp:322-333 */
    const int last = POM_BOARD_SIZE - 1;
    s->board[0][0] = POM_AGENT0 + 0;
    s->board[0][last] = POM_AGENT0 + 1;
    s->board[last][last] = POM_AGENT0 + 2;
    s->board[last][0] = POM_AGENT0 + 3;
    s->agents[1].x = s->agents[2].x = last;
    s->agents[2].y = s->agents[3].y = last;
}

int64_t pom_oracle_run_random_fresh(void* states, int32_t* episodes, int n, int ticks, uint64_t seed, uint64_t board_seed,
                                    int first_env, int tick0, int dist, int max_steps)
{
    PomState* s = (PomState*)states;
    int64_t steps = 0;
    for (int t = 0; t < ticks; t++) {
        for (int e = 0; e < n; e++) {
            PomState* st = &s[e];
            const int done = st->aliveAgents <= 1 || (max_steps > 0 && st->timeStep >= max_steps);
            if (done) pom_oracle_boardgen(board_seed, (uint32_t)(first_env + e), (uint32_t)++episodes[e], st);
            int32_t mv[4];
            pom_rng_moves(seed, (uint32_t)(first_env + e), (uint32_t)(tick0 + t), dist, mv);
            pom_oracle_step(st, mv);
            st->timeStep++;
            steps++;
        }
    }
    return steps;
}

int64_t pom_oracle_run_simple_fresh(void* states, int32_t* episodes, PomSimpleMem* mems, int n, int ticks, uint64_t seed,
                                    uint64_t board_seed, int first_env, int tick0, int max_steps)
{
    PomState* s = (PomState*)states;
    int64_t steps = 0;
    for (int t = 0; t < ticks; t++) {
        for (int e = 0; e < n; e++) {
            PomState* st = &s[e];
            if (st->aliveAgents <= 1 || (max_steps > 0 && st->timeStep >= max_steps)) {
                pom_oracle_boardgen(board_seed, (uint32_t)(first_env + e), (uint32_t)++episodes[e], st);
                memset(&mems[4 * e], 0, 4 * sizeof(PomSimpleMem)); /* a new game gets fresh agents */
            }
            const uint64_t r = pom_rng_draw(seed, (uint32_t)(first_env + e), (uint32_t)(tick0 + t));
            int32_t mv[4];
            for (int i = 0; i < 4; i++) {
                const int draw = (int)((((uint32_t)(r >> (16 * i)) & 0xFFFFu) * 5u) >> 16);
                mv[i] = st->agents[i].dead ? POM_MOVE_IDLE : pom_oracle_simple_act(st, i, &mems[4 * e + i], draw);
            }
            pom_oracle_step(st, mv);
            st->timeStep++;
            steps++;
        }
    }
    return steps;
}
