/*
 * emul_fuzz.cpp — TEST-ONLY: random play, device tick body (host build, pom_emul.cpp) vs the
 * oracle (oracle/pom_oracle.c) on identical State + Move[4]; all 1004 bytes and the UB flags
 * must agree after every tick.  usage: emul_fuzz <scenario> <steps> <seed> [quad]
 * quad: the four-lanes-per-env build of the body (pom_emul_quad.cpp: the shipped kernel's shape) instead of the one-lane build
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pom_packed.h" /* status bits */

extern "C" {
#include "pom_oracle.h"
#include "pom_testgen.h"
#include "pom_rng.h"
uint32_t pom_emul_step(void* state_1004, const int32_t* moves, int env_mode, int max_steps, uint32_t* status_io);
uint32_t pom_emul_quad_step(void* state_1004, const int32_t* moves, int env_mode, int max_steps, uint32_t* status_io);
}

int main(int argc, char** argv)
{
    int scenario = argc > 1 ? atoi(argv[1]) : 1;
    long long steps = argc > 2 ? atoll(argv[2]) : 200000;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
    const int quad = argc > 4 && !strcmp(argv[4], "quad");
    PomTestRng rng = {pom_splitmix64(seed) | 1};
    int dist = scenario == 0 ? POM_DIST_HARMLESS : scenario == 2 ? POM_DIST_STRESS : POM_DIST_RANDOM;
    PomState cur, a, b;
    long long done = 0, episodes = 0, flagged = 0;
    while (done < steps) {
        pom_testgen_board(&cur, scenario, &rng);
        episodes++;
        PomEnvStatus est = {0, -1, 0};
        uint32_t status = 0;
        for (int t = 0; t < 800 && done < steps; t++) {
            int32_t mv[4];
            pom_rng_moves(seed, (uint32_t)episodes, (uint32_t)t, dist, mv);
            if (t % 7 == 3) mv[episodes & 3] = 9; /* out-of-range move values are inputs too */
            a = cur;
            b = cur;
            uint32_t ub_o = pom_oracle_env_step(&a, mv, &est);
            uint32_t ub_e = quad ? pom_emul_quad_step(&b, mv, 1, 0, &status) : pom_emul_step(&b, mv, 1, 0, &status);
            done++;
            for (int i = 0; i < 4; i++) a.agents[i].pad_[0] = a.agents[i].pad_[1] = 0;
            int eq = memcmp(&a, &b, sizeof a) == 0 && ub_o == ub_e;
            int st_eq = ((status & POM_ST_DONE) != 0) == (est.done != 0) && ((status & POM_ST_DRAW) != 0) == (est.draw != 0) &&
                        (int)((status >> POM_ST_WINNER_SHIFT) & 7) - 1 == est.winner;
            if (!eq || !st_eq) {
                printf("MISMATCH scenario %d episode %lld tick %d moves %d %d %d %d ub oracle %x emul %x status %x\n", scenario, episodes,
                       t, mv[0], mv[1], mv[2], mv[3], ub_o, ub_e, status);
                const int32_t *pa = (const int32_t*)&a, *pb = (const int32_t*)&b, *pc = (const int32_t*)&cur;
                for (int k = 0; k < 251; k++)
                    if (pa[k] != pb[k]) printf("  dword %d: before %d oracle %d emul %d\n", k, pc[k], pa[k], pb[k]);
                if (const char* f = getenv("POM_FUZZ_DUMP")) { /* the failing input, for a replay under a debugger */
                    FILE* o = fopen(f, "wb");
                    if (o) {
                        fwrite(&cur, sizeof cur, 1, o);
                        fwrite(mv, sizeof mv, 1, o);
                        fclose(o);
                    }
                }
                return 1;
            }
            if (ub_o) flagged++;
            cur = a;
            if (est.done) break;
        }
    }
    printf("emul_fuzz%s scenario %d seed %llu: %lld steps, %lld episodes, %lld flagged ticks, 0 mismatches\n", quad ? " (four lanes per env)" : "", scenario,
           (unsigned long long)seed, done, episodes, flagged);
    return 0;
}
