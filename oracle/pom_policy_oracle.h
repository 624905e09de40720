/*
 * pom_policy_oracle.h — TEST INFRASTRUCTURE (see pom_policy_oracle.c): CPU restatement of the reference's SimpleAgent policy.
 */
#ifndef POM_POLICY_ORACLE_H_
#define POM_POLICY_ORACLE_H_

#include <stdint.h>
#include "pom_state.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what a SimpleAgent remembers between act() calls (include/agents.hpp:55-76): recentPositions and moveQueue,
 * both FixedQueue<_,4> with their raw slots — stale slots are read (simple_agent.cpp:47,113) */
typedef struct PomSimpleMem {
    int32_t rp[4][2];   /* recentPositions.queue[i] = {x, y} */
    int32_t rp_index, rp_count;
    int32_t mq[4];      /* moveQueue.queue[i] */
    int32_t mq_index, mq_count;
} PomSimpleMem;

/* SimpleAgent::act (simple_agent.cpp:123-137) for agent `id`; `draw` is the value its one possible
 * intDist(rng) call returns (uniform 0..4).  Returns the Move and updates the memory. */
int32_t pom_oracle_simple_act(const void* state, int id, PomSimpleMem* mem, int draw);

/* strategy::IsAdjacentEnemy (strategy.cpp:297-313); strategy::FillRMap's raw map (distance | predecessor << 16) and, per reachable
 * cell other than the source, MoveTowardsPosition's answer (-1 elsewhere) (strategy.cpp:59-121); the stream's draw for an act() */
int32_t pom_oracle_is_adjacent_enemy(const void* state, int id, int distance);
void pom_oracle_fill_rmap(const void* state, int id, int32_t* map121, int32_t* move_to121);
int32_t pom_oracle_policy_draw(uint64_t seed, uint32_t env, uint32_t tick, int agent);

/* one round of act() for n envs (what pom_batch_policy_simple computes); done[e] != 0 marks a finished env (all IDLE) */
void pom_oracle_simple_policy(const void* states, PomSimpleMem* mems, int n, uint64_t seed, int first_env, int tick,
                              const int32_t* done, int32_t* moves_out);

/* Environment::Step with four SimpleAgents for n envs (environment.cpp:139-169): act for the alive agents (a dead agent's
 * Move entry is IDLE), then the tick; draws from the pom_rng.h stream.  mems: n x 4.  Returns env-steps executed. */
int64_t pom_oracle_run_simple(void* states, const void* initial, PomSimpleMem* mems, int n, int ticks, uint64_t seed,
                              int first_env, int tick0, int max_steps);

#ifdef __cplusplus
}
#endif
#endif
