/*
 * pom_testgen.h — TEST INFRASTRUCTURE: synthetic boards for the fuzzers
 * (oracle/fuzz_diff.c, tests/emul/emul_fuzz.cpp).  Cell distribution of the
 * reference's InitBoardItems (/root/reference/src/bboard/bboard.cpp:59-74,346-382:
 * passage 5/7, rigid 1/7, wood 1/7, half the woods carry a flag 1..4), agents in the
 * corners (bboard.cpp:322-333), plus the stress / powerup-rich variants of SURVEY §8d.
 */
#ifndef POM_TESTGEN_H_
#define POM_TESTGEN_H_

#include <stdint.h>

#include "pom_oracle.h"

typedef struct { uint64_t s; } PomTestRng;
static inline uint32_t ptg_rnd(PomTestRng *r)
{
    r->s ^= r->s << 13;
    r->s ^= r->s >> 7;
    r->s ^= r->s << 17;
    return (uint32_t)(r->s >> 24);
}
static inline int ptg_rndn(PomTestRng *r, int n) { return (int)(ptg_rnd(r) % (uint32_t)n); }

/* scenario 0/1: reference distribution; 2: kick/chain stress (config 5); 3: powerup-rich, mixed agents */
static inline void pom_testgen_board(PomState *s, int scenario, PomTestRng *r)
{
    pom_oracle_init_state(s);
    int woods[121], nw = 0;
    int32_t *cells = &s->board[0][0];
    for (int c = 0; c < 121; c++) {
        int k = ptg_rndn(r, scenario == 2 ? 14 : 7);
        int v = POM_PASSAGE;
        if (k == 1) v = POM_RIGID;
        else if (k == 2) { v = POM_WOOD; woods[nw++] = c; }
        if (scenario == 3 && k >= 5) { v = POM_WOOD; woods[nw++] = c; }
        cells[c] = v;
    }
    int want = scenario == 3 ? nw : (nw + 1) / 2;
    for (int k = 0; k < want && nw > 0; k++) {
        int j = ptg_rndn(r, nw);
        int c = woods[j];
        woods[j] = woods[--nw];
        cells[c] += 1 + ptg_rndn(r, 4);
    }
    pom_oracle_put_agents_in_corners(s, 0, 1, 2, 3);
    if (scenario == 2) {
        for (int i = 0; i < 4; i++) {
            s->agents[i].canKick = 1;
            s->agents[i].maxBombCount = 5;
            s->agents[i].bombStrength = 4;
        }
        int life = 2;
        for (int k = 0; k < 8; k++) {
            int x = ptg_rndn(r, 11), y = ptg_rndn(r, 11);
            if (s->board[y][x] != POM_PASSAGE) continue;
            life += ptg_rndn(r, 2);
            if (life > 10) life = 10;
            int before = s->bombs.count;
            pom_oracle_plant_bomb(s, x, y, k & 3, life, 1);
            if (s->bombs.count > before && k < 2) {
                int *b = &s->bombs.queue[(s->bombs.index + before) % 20];
                *b = (*b & ~0xF00000) + ((1 + ptg_rndn(r, 4)) << 20);
            }
        }
    }
    if (scenario == 3) {
        for (int i = 0; i < 4; i++) {
            s->agents[i].canKick = (uint8_t)ptg_rndn(r, 2);
            s->agents[i].maxBombCount = 1 + ptg_rndn(r, 3);
            s->agents[i].bombStrength = 1 + ptg_rndn(r, 5);
        }
    }
}

#endif
