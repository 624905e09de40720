/*
 * fuzz_env.c — TEST INFRASTRUCTURE.  Pins the restatement of Environment::Step's bookkeeping (pom_oracle_env_step,
 * SURVEY §8 row a12) against the compiled, unmodified reference Environment (oracle/_ref/libpomref.so, which now holds
 * /root/reference/src/bboard/environment.cpp too): identical start State, identical moves from four play-back agents,
 * compared after every Environment::Step(false): all 1000 meaningful State bytes (timeStep included), finished / winner /
 * draw, WHICH agents were asked for a move (act() only for live agents, environment.cpp:139-146), and that a finished game is
 * not stepped any more (environment.cpp:125-128).
 *
 * Guards.  Environment::Step hands bboard::Step a 4-entry local array whose entries for dead agents are never written
 * (environment.cpp:130) and whose element [-1] is read on lost-agent ticks (SURVEY Q-UB1): the reference is only stepped on
 * ticks where neither can matter — the restatement predicts no UB flag at all, and the tick's outcome is the same for every
 * combination of IDLE / UP / DOWN / LEFT / RIGHT in the dead agents' entries (their entries only enter FillDestPos /
 * FixSwitchMove).  Otherwise the game ends there and is counted as cut short.
 *
 * usage: fuzz_env <scenario 0..3> <steps> <seed>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pom_oracle.h"
#include "pom_rng.h"
#include "pom_testgen.h"

void *ref_env_new(const void *start_state);
void ref_env_delete(void *g);
int ref_env_step(void *g, const int *moves, void *state_out, int *done, int *winner, int *draw);

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

/* does any choice of the dead agents' Move entries change the tick? */
static int dead_moves_matter(const PomState *cur, const int32_t *mv, const PomState *idle_result)
{
    int dead[4], nd = 0;
    for (int i = 0; i < 4; i++)
        if (cur->agents[i].dead) dead[nd++] = i;
    int combos = 1;
    for (int k = 0; k < nd; k++) combos *= 5;
    for (int c = 1; c < combos; c++) {
        int32_t m2[4] = { mv[0], mv[1], mv[2], mv[3] };
        int r = c;
        for (int k = 0; k < nd; k++) {
            m2[dead[k]] = r % 5;
            r /= 5;
        }
        PomState b = *cur;
        pom_oracle_step(&b, m2);
        b.timeStep++;
        if (!states_equal(&b, idle_result)) return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    int scenario = argc > 1 ? atoi(argv[1]) : 1;
    long long steps = argc > 2 ? atoll(argv[2]) : 200000;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
    PomTestRng rng = { pom_splitmix64(seed) | 1 };
    int dist = scenario == 0 ? POM_DIST_HARMLESS : scenario == 2 ? POM_DIST_STRESS : POM_DIST_RANDOM;

    long long done_steps = 0, games = 0, finished = 0, draws = 0, cut_ub = 0, cut_dead = 0, frozen_checks = 0, dead_asked_checks = 0;
    while (done_steps < steps) {
        PomState cur;
        pom_testgen_board(&cur, scenario, &rng);
        games++;
        void *g = ref_env_new(&cur);
        PomEnvStatus st = { 0, -1, 0 };
        for (int t = 0; t < 800 && done_steps < steps; t++) {
            int32_t mv[4];
            pom_rng_moves(seed, (uint32_t)games, (uint32_t)t, dist, mv);
            for (int i = 0; i < 4; i++)
                if (cur.agents[i].dead) mv[i] = POM_MOVE_IDLE; /* what the restatement and the device define for them */
            PomState a = cur;
            PomEnvStatus sa = st;
            uint32_t ub = pom_oracle_env_step(&a, mv, &sa);
            if (ub) { cut_ub++; break; }
            if (cur.aliveAgents < 4 && dead_moves_matter(&cur, mv, &a)) { cut_dead++; break; }
            int alive_mask = 0;
            for (int i = 0; i < 4; i++) alive_mask |= (!cur.agents[i].dead) << i;
            if (alive_mask != 15) dead_asked_checks++;
            PomState b;
            int d, w, dr;
            int asked = ref_env_step(g, mv, &b, &d, &w, &dr);
            done_steps++;
            if (!states_equal(&a, &b) || d != sa.done || w != sa.winner || dr != sa.draw || asked != alive_mask) {
                printf("MISMATCH scenario %d game %lld tick %d: done %d/%d winner %d/%d draw %d/%d asked %x/%x states %s\n", scenario,
                       games, t, d, sa.done, w, sa.winner, dr, sa.draw, asked, alive_mask, states_equal(&a, &b) ? "equal" : "DIFFER");
                return 1;
            }
            cur = a;
            st = sa;
            if (st.done) {
                finished++;
                draws += st.draw;
                /* a finished game is not stepped: state, status untouched, nobody asked */
                asked = ref_env_step(g, mv, &b, &d, &w, &dr);
                PomEnvStatus s2 = st;
                PomState a2 = cur;
                pom_oracle_env_step(&a2, mv, &s2);
                frozen_checks++;
                if (asked != 0 || !states_equal(&b, &cur) || !states_equal(&a2, &cur) || d != st.done || w != st.winner || dr != st.draw) {
                    printf("MISMATCH: finished game %lld was stepped (asked %x)\n", games, asked);
                    return 1;
                }
                break;
            }
        }
        ref_env_delete(g);
    }
    printf("scenario %d seed %llu: Environment::Step calls %lld in %lld games, mismatches 0 | finished %lld (draws %lld), frozen-game checks "
           "%lld | steps with a dead agent (act() not asked) %lld | games cut short: UB tick %lld, dead agent's entry matters %lld\n",
           scenario, (unsigned long long)seed, done_steps, games, finished, draws, frozen_checks, dead_asked_checks, cut_ub, cut_dead);
    return 0;
}
