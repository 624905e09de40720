/*
 * pom_boardgen_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pom_boardgen_oracle.c).
 */
#ifndef POM_BOARDGEN_ORACLE_H_
#define POM_BOARDGEN_ORACLE_H_

#include <stdint.h>

#include "pom_policy_oracle.h"
#include "pom_state.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the start State of (seed, env, episode) as include/pom_boardgen.h specifies it */
void pom_oracle_boardgen(uint64_t seed, uint32_t env, uint32_t episode, void* state_out);

/* pom_oracle_run_random with fresh boards: a finished env starts episode[e] + 1 on a newly generated board
 * (episodes: int32[n], in/out) instead of replaying its first one */
int64_t pom_oracle_run_random_fresh(void* states, int32_t* episodes, int n, int ticks, uint64_t seed, uint64_t board_seed,
                                    int first_env, int tick0, int dist, int max_steps);

/* pom_oracle_run_simple (four SimpleAgents) with fresh boards */
int64_t pom_oracle_run_simple_fresh(void* states, int32_t* episodes, PomSimpleMem* mems, int n, int ticks, uint64_t seed,
                                    uint64_t board_seed, int first_env, int tick0, int max_steps);

#ifdef __cplusplus
}
#endif
#endif
