/*
 * ref_baseline.c — TEST INFRASTRUCTURE (bench.py's cpu_baseline leg, kind "reference").
 *
 * Times the UNMODIFIED reference bboard::Step (oracle/_ref/libpomref.so, compiled from /root/reference where it lies; only
 * the built library travels to the GPU box) on the bench workload: same boards, same pom_rng.h move stream, same auto-reset
 * rule as pom_oracle_run_random.  The reference has undefined behaviour on reachable states (null GetBomb,
 * queue overflow — it would take the process down), so every tick is first played by the restatement on a copy, UNTIMED; an
 * env whose tick it flags is advanced with the restatement's result and not counted, all others are stepped by the
 * reference inside the timed region.  Returns the reference's env-steps; *seconds accumulates the time of its calls.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pom_oracle.h"
#include "pom_rng.h"
#include "pom_state.h"

void ref_step(void* state, const int* moves); /* oracle/ref_shim.cpp: Step with the padded move array */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int64_t ref_run_random_timed(void* states, const void* initial, int n, int ticks, uint64_t seed, int first_env, int tick0, int dist,
                             int max_steps, double* seconds, int64_t* skipped)
{
    PomState* s = (PomState*)states;
    const PomState* init = (const PomState*)initial;
    PomState* trial = (PomState*)malloc((size_t)n * sizeof(PomState));
    int32_t* mv = (int32_t*)malloc((size_t)n * 4 * sizeof(int32_t));
    uint8_t* skip = (uint8_t*)malloc((size_t)n);
    int64_t steps = 0;
    for (int t = 0; t < ticks; t++) {
        for (int e = 0; e < n; e++) { /* untimed: restart rule, moves, and the restatement's verdict on this tick */
            if (s[e].aliveAgents <= 1 || (max_steps > 0 && s[e].timeStep >= max_steps)) s[e] = init[e];
            pom_rng_moves(seed, (uint32_t)(first_env + e), (uint32_t)(tick0 + t), dist, &mv[4 * e]);
            trial[e] = s[e];
            const uint32_t ub = pom_oracle_step(&trial[e], &mv[4 * e]);
            skip[e] = (ub & ~(uint32_t)POM_UB_LOST_AGENT) != 0; /* a lost agent is defined under the padded move array */
        }
        const double t0 = now_s();
        for (int e = 0; e < n; e++)
            if (!skip[e]) ref_step(&s[e], &mv[4 * e]);
        *seconds += now_s() - t0;
        for (int e = 0; e < n; e++) {
            if (skip[e]) {
                s[e] = trial[e];
                (*skipped)++;
            } else {
                steps++;
            }
            s[e].timeStep++;
        }
    }
    free(trial);
    free(mv);
    free(skip);
    return steps;
}

/*
 * BASELINE config 1: ONE env, i.i.d. uniform {IDLE, UP, DOWN, LEFT, RIGHT} moves per agent per tick (HarmlessAgent's
 * distribution, src/agents/basic_agents.cpp:28-38), one thread, `reps` x `ticks` ticks from the same start board — the
 * shape of unit_test/bboard/performance_test.cpp:55-59.  The move script is laid out first and played once by the
 * restatement (untimed) so that the reference is never timed into a tick with one of its crashing UBs: the script is cut
 * there (harmless play plants no bombs, so in practice only lost-agent ticks occur, which the padded move array defines).
 * Returns the reference's env-steps; *seconds accumulates the time of the whole replay loops (no per-tick timer calls).
 */
int64_t ref_run_single_timed(const void* start, int ticks, uint64_t seed, int dist, int reps, double* seconds)
{
    int32_t* mv = (int32_t*)malloc((size_t)ticks * 4 * sizeof(int32_t));
    PomState probe = *(const PomState*)start;
    int usable = 0;
    for (int t = 0; t < ticks; t++) {
        pom_rng_moves(seed, 0u, (uint32_t)t, dist, &mv[4 * t]);
        if (pom_oracle_step(&probe, &mv[4 * t]) & ~(uint32_t)POM_UB_LOST_AGENT) break;
        probe.timeStep++;
        usable++;
        if (probe.aliveAgents <= 1) break;
    }
    int64_t steps = 0;
    for (int r = 0; r < reps; r++) {
        PomState s = *(const PomState*)start;
        const double t0 = now_s();
        for (int t = 0; t < usable; t++) {
            ref_step(&s, &mv[4 * t]);
            s.timeStep++;
        }
        *seconds += now_s() - t0;
        steps += usable;
        if (memcmp(s.board, probe.board, sizeof s.board) != 0) { /* the replay must end where the restatement ended */
            steps = -1;
            break;
        }
    }
    free(mv);
    return steps;
}
