/*
 * fuzz_policy.c — TEST INFRASTRUCTURE.  Pins the SimpleAgent restatement (pom_policy_oracle.c) against the compiled,
 * unmodified reference agent (oracle/_ref/libpomref.so): four reference SimpleAgents and four restated ones play the same
 * games (ticks by the already pinned pom_oracle_step); every act() must return the same Move and leave the same agent
 * memory.  usage: fuzz_policy <scenario 0..3> <agent-steps> <seed>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pom_policy_oracle.h"
#include "pom_oracle.h"
#include "pom_testgen.h"
#include "pom_rng.h"

void* ref_simple_new(int id, unsigned long long seed);
void ref_simple_delete(void* p);
int ref_simple_peek_draw(void* p);
int ref_simple_act(void* p, const void* state);
void ref_simple_memory(void* p, int* out16);
void ref_simple_set_memory(void* p, const int* in16);

int main(int argc, char** argv)
{
    int scenario = argc > 1 ? atoi(argv[1]) : 1;
    long long want = argc > 2 ? atoll(argv[2]) : 200000;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
    PomTestRng rng = { pom_splitmix64(seed) | 1 };
    long long acts = 0, episodes = 0, moves_hist[6] = { 0 };
    PomState st;
    while (acts < want) {
        pom_testgen_board(&st, scenario, &rng);
        if (scenario == 1 && (episodes & 1)) { /* let them meet: drop two agents next to each other */
            st.board[0][0] = 0; st.board[0][10] = 0;
            pom_oracle_put_agent(&st, 4, 5, 0);
            pom_oracle_put_agent(&st, 6, 5, 1);
        }
        episodes++;
        void* ref[4];
        PomSimpleMem mem[4];
        memset(mem, 0, sizeof mem);
        for (int i = 0; i < 4; i++) {
            ref[i] = ref_simple_new(i, seed * 1000003ull + (uint64_t)episodes * 4 + i);
            /* `new SimpleAgent()` leaves the raw slots of its two FixedQueues uninitialised (heap garbage) and later reads
             * stale ones (simple_agent.cpp:47,113): parity is defined from zero-initialised agent memory */
            int zero[16] = { 0 };
            ref_simple_set_memory(ref[i], zero);
        }
        for (int t = 0; t < 300 && st.aliveAgents > 1; t++) {
            int32_t mv[4] = { 0, 0, 0, 0 };
            for (int i = 0; i < 4; i++) {
                if (st.agents[i].dead) continue;
                int draw = ref_simple_peek_draw(ref[i]);
                int m_ref = ref_simple_act(ref[i], &st);
                int m_ora = pom_oracle_simple_act(&st, i, &mem[i], draw);
                int rm[16];
                ref_simple_memory(ref[i], rm);
                acts++;
                if (m_ref != m_ora || memcmp(rm, &mem[i], sizeof rm)) {
                    printf("MISMATCH scenario %d episode %lld tick %d agent %d draw %d: reference move %d, restatement %d\n", scenario,
                           episodes, t, i, draw, m_ref, m_ora);
                    const int32_t* om = (const int32_t*)&mem[i];
                    for (int k = 0; k < 16; k++) printf("  mem[%d] ref %d ora %d\n", k, rm[k], om[k]);
                    return 1;
                }
                mv[i] = m_ref;
                moves_hist[m_ref < 6 ? m_ref : 5]++;
            }
            if (pom_oracle_step(&st, mv) & ~1u) break; /* leave the states the reference's Step cannot handle */
            st.timeStep++;
        }
        for (int i = 0; i < 4; i++) ref_simple_delete(ref[i]);
    }
    printf("fuzz_policy scenario %d seed %llu: %lld act() calls over %lld games, 0 mismatches; moves idle/up/down/left/right/bomb = "
           "%lld %lld %lld %lld %lld %lld\n", scenario, (unsigned long long)seed, acts, episodes, moves_hist[0], moves_hist[1],
           moves_hist[2], moves_hist[3], moves_hist[4], moves_hist[5]);
    return 0;
}
