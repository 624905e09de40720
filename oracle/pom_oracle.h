/*
 * pom_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pom_oracle.c).
 * C interface of the CPU restatement of bboard::Step used as the parity checker.
 */
#ifndef POM_ORACLE_H_
#define POM_ORACLE_H_

#include <stdint.h>
#include "pom_state.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PomEnvStatus { /* Environment's finished / agentWon / isDraw, bboard.hpp:551-555 */
    int32_t done;
    int32_t winner; /* -1 until someone has won */
    int32_t draw;
} PomEnvStatus;

/* one tick of bboard::Step (step.cpp:9-284) on a 1004-byte State; returns POM_UB_* flags */
uint32_t pom_oracle_step(void *state, const int32_t *moves);
/* Environment::Step's Step + timeStep++ + done/winner/draw (environment.cpp:123-169) */
uint32_t pom_oracle_env_step(void *state, const int32_t *moves, PomEnvStatus *st);

/* State methods the reference's tests use to build boards */
void pom_oracle_init_state(void *state);
void pom_oracle_put_agent(void *state, int x, int y, int id);
void pom_oracle_put_agents_in_corners(void *state, int a0, int a1, int a2, int a3);
void pom_oracle_kill(void *state, int id);
void pom_oracle_plant_bomb(void *state, int x, int y, int id, int life_time, int set_item);
void pom_oracle_spawn_flame(void *state, int x, int y, int strength);

/* step utilities pinned by the reference's [step utilities] tests; xy = {x0,y0,...,x3,y3} */
void pom_oracle_dest_pos(const void *state, const int32_t *moves, int32_t *out_xy);
void pom_oracle_fix_switch_move(const void *state, int32_t *xy);
int pom_oracle_resolve_dependencies(const void *state, const int32_t *xy, int32_t *dependency, int32_t *chain);

/* CPU-baseline driver: n envs x ticks with the pom_rng.h move stream and auto-reset */
int64_t pom_oracle_run_random(void *states, const void *initial, int n, int ticks, uint64_t seed, int first_env,
                              int tick0, int dist, int max_steps);

#ifdef __cplusplus
}
#endif
#endif
