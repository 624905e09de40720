/*
 * fuzz_diff.c — TEST INFRASTRUCTURE.  Differential fuzzer that pins the
 * restatement (pom_oracle.c) against the compiled, unmodified reference
 * (oracle/_ref/libpomref.so): both step identical State + Move[4] inputs and
 * all 1000 meaningful bytes of the State are compared after every tick.
 *
 * Guards: the restatement runs first on a copy; a tick on which
 * it predicts one of the reference's crashing UBs (null GetBomb deref, bomb
 * queue overflow, endless bounce-back recursion) is never given to the
 * reference — the episode is restarted instead.  Lost-agent ticks (Q-UB1) are
 * run through the padded-moves shim and counted separately.
 *
 * usage: fuzz_diff <scenario 0..3> <steps> <seed>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pom_oracle.h"
#include "pom_rng.h"

void ref_step(void *state, const int *moves);
void ref_init_state(void *p);
int ref_state_size(void);

static uint64_t rng_state;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 24);
}
static int rndn(int n) { return (int)(rnd() % (uint32_t)n); }

/* board with the reference's cell distribution (bboard.cpp:59-74,346-382):
 * passage 5/7, rigid 1/7, wood 1/7, half the woods carry a flag 1..4 */
static void gen_board(PomState *s, int scenario)
{
    ref_init_state(s);
    int woods[121], nw = 0;
    int32_t *cells = &s->board[0][0];
    for (int c = 0; c < 121; c++) {
        int r = rndn(scenario == 2 ? 14 : 7);
        int v = POM_PASSAGE;
        if (r == 1) v = POM_RIGID;
        else if (r == 2) { v = POM_WOOD; woods[nw++] = c; }
        if (scenario == 3 && r >= 5) { v = POM_WOOD; woods[nw++] = c; } /* powerup-rich */
        cells[c] = v;
    }
    int want = (nw + 1) / 2;
    if (scenario == 3) want = nw;
    for (int k = 0; k < want && nw > 0; k++) {
        int j = rndn(nw);
        int c = woods[j];
        woods[j] = woods[--nw];
        cells[c] += 1 + rndn(4);
    }
    pom_oracle_put_agents_in_corners(s, 0, 1, 2, 3);
    if (scenario == 2) { /* kick / chain stress, SURVEY §8d config 5 */
        for (int i = 0; i < 4; i++) {
            s->agents[i].canKick = 1;
            s->agents[i].maxBombCount = 5;
            s->agents[i].bombStrength = 4;
        }
        int life = 2;
        for (int k = 0; k < 8; k++) {
            int x = rndn(11), y = rndn(11);
            if (s->board[y][x] != POM_PASSAGE) continue;
            life += rndn(2);
            if (life > 10) life = 10;
            int before = s->bombs.count;
            pom_oracle_plant_bomb(s, x, y, k & 3, life, 1);
            if (s->bombs.count > before && k < 2) {
                int *b = &s->bombs.queue[(s->bombs.index + before) % 20];
                *b = (*b & ~0xF00000) + ((1 + rndn(4)) << 20);
            }
        }
    }
    if (scenario == 3) {
        for (int i = 0; i < 4; i++) {
            s->agents[i].canKick = (uint8_t)rndn(2);
            s->agents[i].maxBombCount = 1 + rndn(3);
            s->agents[i].bombStrength = 1 + rndn(5);
        }
    }
}

static int states_equal(const PomState *a, const PomState *b)
{
    if (memcmp(a->board, b->board, sizeof a->board)) return 0;
    if (a->timeStep != b->timeStep || a->aliveAgents != b->aliveAgents) return 0;
    for (int i = 0; i < 4; i++)
        if (memcmp(&a->agents[i], &b->agents[i], 22)) return 0; /* skip 2 pad bytes */
    if (memcmp(&a->bombs, &b->bombs, sizeof a->bombs)) return 0;
    if (memcmp(&a->flames, &b->flames, sizeof a->flames)) return 0;
    return 1;
}

static void dump(const char *tag, const PomState *s)
{
    printf("== %s  t=%d alive=%d\n", tag, s->timeStep, s->aliveAgents);
    for (int y = 0; y < 11; y++) {
        for (int x = 0; x < 11; x++) printf("%9d ", s->board[y][x]);
        printf("\n");
    }
    for (int i = 0; i < 4; i++)
        printf("agent%d (%d,%d) bc=%d max=%d str=%d kick=%d dead=%d\n", i, s->agents[i].x, s->agents[i].y,
               s->agents[i].bombCount, s->agents[i].maxBombCount, s->agents[i].bombStrength, s->agents[i].canKick,
               s->agents[i].dead);
    printf("bombs idx=%d cnt=%d:", s->bombs.index, s->bombs.count);
    for (int i = 0; i < 20; i++) printf(" %08x", (unsigned)s->bombs.queue[i]);
    printf("\nflames idx=%d cnt=%d:", s->flames.index, s->flames.count);
    for (int i = 0; i < 20; i++)
        printf(" (%d,%d,t%d,s%d)", s->flames.queue[i].x, s->flames.queue[i].y, s->flames.queue[i].timeLeft,
               s->flames.queue[i].strength);
    printf("\n");
}

int main(int argc, char **argv)
{
    int scenario = argc > 1 ? atoi(argv[1]) : 1;
    long long steps = argc > 2 ? atoll(argv[2]) : 1000000;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
    rng_state = pom_splitmix64(seed) | 1;
    if (ref_state_size() != (int)sizeof(PomState)) {
        printf("layout mismatch: reference State is %d bytes\n", ref_state_size());
        return 2;
    }
    int dist = scenario == 0 ? POM_DIST_HARMLESS : scenario == 2 ? POM_DIST_STRESS : POM_DIST_RANDOM;

    PomState cur, a, b;
    long long done = 0, compared = 0, lost = 0, lost_mismatch = 0, skipped = 0, episodes = 0;
    long long ubcount[5] = { 0 };
    while (done < steps) {
        gen_board(&cur, scenario);
        episodes++;
        for (int t = 0; t < 800 && done < steps; t++) {
            int32_t mv[4];
            pom_rng_moves(seed, (uint32_t)episodes, (uint32_t)t, dist, mv);
            a = cur;
            uint32_t ub = pom_oracle_step(&a, mv);
            a.timeStep++;
            done++;
            for (int k = 0; k < 5; k++) if (ub & (1u << k)) ubcount[k]++;
            if (ub & ~(uint32_t)POM_UB_LOST_AGENT) { skipped++; break; } /* reference would crash */
            b = cur;
            ref_step(&b, mv);
            b.timeStep++;
            int eq = states_equal(&a, &b);
            if (ub & POM_UB_LOST_AGENT) {
                lost++;
                if (!eq) { lost_mismatch++; break; }
            } else {
                compared++;
                if (!eq) {
                    printf("MISMATCH scenario %d episode %lld tick %d moves %d %d %d %d\n", scenario, episodes, t, mv[0],
                           mv[1], mv[2], mv[3]);
                    dump("before", &cur);
                    dump("oracle", &a);
                    dump("reference", &b);
                    return 1;
                }
            }
            cur = a;
            if (cur.aliveAgents <= 1) break;
        }
    }
    printf("scenario %d seed %llu: steps %lld episodes %lld compared %lld mismatches 0 | lost-agent ticks %lld (mismatch %lld) | "
           "skipped-UB ticks %lld | flags lost %lld null_bomb %lld overflow %lld revert %lld badidx %lld\n",
           scenario, (unsigned long long)seed, done, episodes, compared, lost, lost_mismatch, skipped, ubcount[0],
           ubcount[1], ubcount[2], ubcount[3], ubcount[4]);
    return 0;
}
