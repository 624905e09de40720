/*
 * pom_batch.h — C-ABI of the MI355X batched Pommerman stepper (libpom_batch.so).
 *
 * Drop-in boundary for ONE path of dist1ll/pomcpp: the simulation tick
 *     void bboard::Step(State* state, Move* moves);          include/bboard.hpp:668, src/bboard/step.cpp:9
 * and the bookkeeping Environment::Step wraps around it      src/bboard/environment.cpp:123-169
 * re-laid out over n concurrent boards on one GPU.  States cross the boundary
 * as the reference's own 1004-byte `bboard::State` (pom_state.h), moves as the
 * reference's `Move` ints (0 IDLE, 1 UP, 2 DOWN, 3 LEFT, 4 RIGHT, 5 BOMB),
 * four per env INCLUDING dead agents (their entries are read by FillDestPos /
 * FixSwitchMove, step_utility.cpp:138-170 — SURVEY.md Q9).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every call returns a
 * PomError (the reference returns void and has UB on misuse; see pom_state.h
 * POM_UB_* for what the stepper does instead).  A handle is bound to one
 * device and one HIP stream; calls on one handle are not thread-safe, distinct
 * handles are independent.  Stepping is asynchronous on the handle's stream;
 * upload / download / status / counters synchronise it.
 *
 * There is deliberately no CPU fallback: without a HIP device every entry
 * point fails with POM_E_HIP.
 */
#ifndef POM_BATCH_H_
#define POM_BATCH_H_

#include <stdint.h>

#include "pom_rng.h"
#include "pom_state.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PomError {
    POM_OK = 0,
    POM_E_ARG = 1,             /* null handle, range outside [0, n_envs), bad option */
    POM_E_HIP = 2,             /* HIP runtime error or no device; pom_last_error() has the text */
    POM_E_UNREPRESENTABLE = 3, /* an uploaded State holds a value no reachable game state has (pom_packed.h) */
    POM_E_NOMEM = 4
} PomError;

/* stepping semantics */
enum {
    POM_MODE_RAW = 0, /* bare bboard::Step: no timeStep++, finished envs keep stepping (what the reference's tests call) */
    POM_MODE_ENV = 1  /* Environment::Step: skip finished envs, Step, timeStep++, done / winner / draw */
};

typedef struct PomBatchOptions {
    int32_t struct_size;  /* = sizeof(PomBatchOptions), for ABI growth */
    int32_t device;       /* HIP device ordinal */
    void*   stream;       /* hipStream_t to run on, or NULL: the library creates one */
    int32_t mode;         /* POM_MODE_RAW / POM_MODE_ENV */
    int32_t auto_reset;   /* ENV mode, what happens to a finished env (POM_RESET_*):
                             0 nothing: it stays finished and is not stepped (Environment::Step, environment.cpp:125-128);
                             1 POM_RESET_AT_START: the NEXT tick first puts it back on its start state (snapshot, or the next
                               generated board) and steps that in the same tick — the caller never sees the start state, and a
                               move supplied for that tick was chosen looking at the finished game;
                             2 POM_RESET_AT_END: the tick that finishes the episode also puts the env on its next start state.
                               State, observation and status then show the new episode's first state (status bit "restarted"),
                               the next move is a move for it, and the finished episode's outcome and final State are kept
                               (pom_batch_last_results, pom_batch_download_terminal) until the env finishes again.  The
                               sequence of states stepped is the same in both modes. */
    int32_t max_steps;    /* ENV mode: env is done once timeStep reaches this (0 = no limit); StartGame's bound, environment.cpp:71 */
    int64_t env_offset;   /* global index of env 0, keys the synthetic move stream when a job is sharded over GPUs */
    int32_t envs_per_wave; /* 0 = default (16); else 16, 32 or 64 envs per wavefront (results are identical) */
    int32_t streams;       /* 0 = choose (chained launches: 2 in a short call, 3 from 50 launches up; sub-batches by batch size); else 1..8: the streams chained
                              launches rotate over, and the sub-batches per step — each on an internal stream — where launches
                              are not chained (results are identical; 1 = plain launches in a row on the handle's stream) */
    int32_t lanes_per_env; /* 0 = default (4: a quad of adjacent lanes runs each env's tick and splits its order-free parts,
                              needs envs_per_wave 16); 1 = one lane per env */
    int32_t fresh_boards;  /* with auto_reset: a finished env starts its next game on a newly generated board (pom_boardgen.h:
                              board_seed, env_offset + env, games played) instead of replaying its snapshot; drawn on the
                              device inside the tick, the host is not involved */
    uint64_t board_seed;   /* seed of those boards; pom_batch_generate replaces it */
    int32_t issue_mode;    /* how the launches of a several-tick call (pom_batch_step_random / _step_simple) are issued — POM_ISSUE_*;
                              results do not depend on it.  0 = POM_ISSUE_CHAIN up to 196,608 envs (default kernel shape),
                              POM_ISSUE_THREADS otherwise */
    int32_t reserved_;
} PomBatchOptions;

/* PomBatchOptions.issue_mode (environment POM_ISSUE = direct | threads | graph | chain overrides it).  Measured per step at 65,536
 * envs, 20-tick call from an idle device / 500-tick call: CHAIN 11.4 - 12.3 / 9.2 us (profiles/r03y_*); with the batch as
 * three sub-batches on parallel streams: THREADS 16.2 - 17.0 / 14.5 us, DIRECT and GRAPH behind that (measured earlier in the
 * round, 18.0 / 15.5 for THREADS then: DIRECT 17.2 .. 26.4 (host-dependent) / 15.6 - 16.0 us; GRAPH 21.4 .. 23.5 / 16.1 us;
 * pomcpp_amd/csrc/pom_runtime.h, profiles/r03_issue_modes.txt) */
enum {
    POM_ISSUE_AUTO = 0,
    POM_ISSUE_DIRECT = 1,  /* the calling thread issues every launch; the library owns no thread */
    POM_ISSUE_THREADS = 2, /* one helper thread per internal sub-stream issues that stream's launches (created on first use, joined by
                              pom_batch_destroy; a helper that cannot be started falls back to DIRECT for its part) */
    POM_ISSUE_GRAPH = 3,   /* chunks of 20 ticks replayed as HIP graphs, one per sub-stream; no library-owned thread */
    POM_ISSUE_CHAIN = 4    /* chained launches (pomcpp_amd/csrc/pom_chain.h): every launch covers the WHOLE batch and plays one tick,
                              consecutive launches go to different internal HIP streams (two in a short call, three from 50 ticks
                              up), and a ticket word per 16-env tile orders that tile's ticks — a tile's next tick waits for the
                              same tile's previous tick only, not for the slowest wavefront of the launch before.  Asynchronous
                              like the other modes: the call returns when its launches are queued — by the calling thread and, in a
                              call of six launches or more, by the helper threads of the internal sub-streams (created on first
                              use, joined by pom_batch_destroy; environment POM_CHAIN_HELPERS=0: the calling thread alone); every
                              other call of this API joins the streams.  pom_batch_create probes the device once (workgroup -> XCD round-robin over
                              eight XCDs); where that does not hold, and for shapes without a chained twin (one lane per env,
                              several ticks per launch, one stream), launches are issued as with POM_ISSUE_THREADS.  A wavefront
                              that cannot play its tile (its predecessor did not show up within 2 s of wall-clock time) leaves
                              the tile alone; the next call that reads or changes the batch finds it and replays the tile's
                              missing ticks with ordinary launches (pom_batch_chain_stats counts such events: expected 0).
                              That check is ONE host synchronisation of the handle's stream: the first call after chained launches
                              that is not itself a chained call of the same kind — a step of another kind (pom_batch_step_device,
                              _step_device_observe, _step_device_range, _policy_simple, several ticks per launch), pom_batch_observe,
                              _snapshot, _generate, _reset_counters, _device_view, _set_streams, and every call that reads results
                              back — blocks the calling thread until the chained launches have finished.  pom_batch_flush,
                              _moves_device and _counters_device do NOT run it (they only join the streams): what they hand out may
                              lag by the ticks of a tile left behind until the next call that does */
};

enum { POM_RESET_OFF = 0, POM_RESET_AT_START = 1, POM_RESET_AT_END = 2 };

typedef struct PomBatch PomBatch;

/* counters accumulated since creation / pom_batch_reset_counters */
enum { POM_CNT_STEPS = 0, POM_CNT_EPISODES = 1, POM_CNT_RESETS = 2, POM_CNT_UB_TICKS = 3, POM_CNT_N = 4 };

const char* pom_last_error(void);
int pom_device_count(void);

int pom_batch_create(PomBatch** out, int64_t n_envs, const PomBatchOptions* opts);
int pom_batch_destroy(PomBatch* h);
int64_t pom_batch_size(const PomBatch* h);

/* host AoS (count x 1004 B) -> device; also becomes the envs' reset snapshot and clears their status */
int pom_batch_upload(PomBatch* h, const void* states, int64_t first, int64_t count);
/* device -> host AoS (count x 1004 B); agent padding bytes come back 0 */
int pom_batch_download(PomBatch* h, void* states, int64_t first, int64_t count);
/* make the current device state the reset snapshot */
int pom_batch_snapshot(PomBatch* h);
/* start boards without the host: every env gets the board (board_seed, env_offset + env, episode 0) of pom_boardgen.h — the
 * reference's InitState distribution (bboard.cpp:339-382: cells passage 5/7, rigid 1/7, wood 1/7, half the woods flagged,
 * agents in the corners) — as its state and its reset snapshot, generated on the device; the agents' memory starts afresh */
int pom_batch_generate(PomBatch* h, uint64_t board_seed);
/* games started so far by envs [first, first+count) (0 = still the first one), uint32 each */
int pom_batch_episodes(PomBatch* h, int64_t first, int64_t count, uint32_t* out);

/* one tick with explicit moves: int32[n_envs][4], host or device memory */
int pom_batch_step(PomBatch* h, const int32_t* moves_host);
int pom_batch_step_device(PomBatch* h, const int32_t* moves_dev);
/* `ticks` ticks of bboard::Step(State*, Move[4]) with explicit moves from a TAPE in device memory, int32[ticks][n_envs][4] (tick t
 * of env e reads moves_dev[(t * n_envs + e) * 4 ..]; dead agents' entries included): what K calls of pom_batch_step_device with
 * moves_dev + t * n_envs * 4 do — same states, same counters — but issued as chained launches where the handle chains
 * (POM_ISSUE_CHAIN; ticks >= 2), i.e. at the speed of pom_batch_step_random instead of one joined launch per tick.  For replays,
 * open-loop rollouts and planners that fix K moves ahead; a policy that needs the state of tick t to choose the move of tick
 * t + 1 calls pom_batch_step_device per tick.  The tape is read asynchronously: it was written on the handle's stream (or before
 * the call) and must stay unchanged until the handle has been synchronised (pom_batch_sync, or any call that reads results
 * back).  Does not advance the tick of the synthetic move stream.  With POM_RESET_AT_END the caller sees only the last tick's
 * "finished" marks; winner / length of each env's most recent episode are kept as always. */
int pom_batch_step_device_many(PomBatch* h, const int32_t* moves_dev, int32_t ticks);
/* `ticks` ticks with the pom_rng.h move stream (seed, env_offset+env, tick); ticks_per_launch >= 1 keeps
 * the env tile resident in LDS for that many ticks per kernel launch (1 = state round-trips HBM each tick) */
int pom_batch_step_random(PomBatch* h, uint64_t seed, int32_t dist, int32_t ticks, int32_t ticks_per_launch);
/* SimpleAgent policy on the device (agents::SimpleAgent, include/agents.hpp:55-76, src/agents/simple_agent.cpp):
 * pom_batch_policy_simple asks all four agents of every env for a Move as Environment::Step does (environment.cpp:139-146:
 * live agents only, a dead agent's entry is IDLE) into the handle's move buffer and updates the agents' memory
 * (recentPositions, moveQueue); pom_batch_step_policy ticks with those moves; pom_batch_step_simple does both `ticks` times.
 * The one random draw an act() may make comes from the pom_rng.h stream keyed (seed, env_offset+env, tick), uniform 0..4.
 * An env that the next step will restart is read from its snapshot and gets fresh agents; pom_batch_upload resets the agents
 * of the uploaded envs.  moves_out_host (nullable): int32[n_envs][4], synchronises. */
int pom_batch_policy_simple(PomBatch* h, uint64_t seed, int32_t* moves_out_host);
int pom_batch_step_policy(PomBatch* h);
int pom_batch_step_simple(PomBatch* h, uint64_t seed, int32_t ticks);
/* the handle's move buffer in device memory, int32[n_envs][4]: what pom_batch_policy_simple fills and pom_batch_step_policy
 * consumes.  A device-side consumer may overwrite entries in between (on the handle's stream, see pom_batch_stream) — e.g. a
 * learned policy for agent 0 playing against three SimpleAgents — without a host round trip. */
int pom_batch_moves_device(PomBatch* h, int32_t** moves_dev);
/* agent memory of envs [first, first+count): 16 int32 per agent, 4 agents per env: recentPositions {x,y}x4, index, count,
 * moveQueue x4, index, count (the members of SimpleAgent that survive between act() calls) */
int pom_batch_policy_memory(PomBatch* h, int64_t first, int64_t count, int32_t* out16);
/* the tick counter that keys the synthetic stream (advanced by step_random / step_policy / step_simple) */
int pom_batch_set_tick(PomBatch* h, int64_t tick);

/* per-env results for [first, first+count): any output pointer may be NULL.
 * done/draw are 0/1, winner is -1 or the agent id (Environment::IsDone/IsDraw/GetWinner, environment.cpp:195-208) */
int pom_batch_status(PomBatch* h, int64_t first, int64_t count, int32_t* done, int32_t* winner, int32_t* draw,
                     int32_t* alive, int32_t* time_step, uint32_t* ubflags);

/* POM_RESET_AT_END: per env of [first, first+count), any output may be NULL: `finished` = 1 if the env's latest tick ended an
 * episode (the env now stands on its next start state); winner / draw / length (timeStep reached) / alive of the env's most
 * recently finished episode (-1 / 0 / 0 / 0 while it has not finished any); pom_batch_download_terminal gives that
 * episode's final State (all-zero while there is none).  POM_E_ARG in the other reset modes. */
int pom_batch_last_results(PomBatch* h, int64_t first, int64_t count, int32_t* finished, int32_t* winner, int32_t* draw,
                           int32_t* length, int32_t* alive);
int pom_batch_download_terminal(PomBatch* h, void* states, int64_t first, int64_t count);

int pom_batch_counters(PomBatch* h, int64_t out[POM_CNT_N]);
/* same totals left in device memory (int64[POM_CNT_N]) on the handle's stream, e.g. for an RCCL all-reduce; does not block
 * (and therefore does not run the check behind chained launches: a tile caught up later adds its steps later) */
int pom_batch_counters_device(PomBatch* h, void* dev_int64x4);
int pom_batch_reset_counters(PomBatch* h);
int pom_batch_sync(PomBatch* h);
/* order everything stepped so far before whatever is queued next on the handle's stream, without blocking the host
 * (steps run on internal sub-streams; every other call of this API does this implicitly) */
int pom_batch_flush(PomBatch* h);
/* the opposite, ahead of time: order the internal sub-streams behind what is queued on the handle's stream NOW, so that the
 * next step's launches need no cross-stream event first (a latency-sensitive caller does this before it starts its clock;
 * stepping does it by itself otherwise) */
int pom_batch_fork(PomBatch* h);
/* change the number of streams launches go to (1..8, see PomBatchOptions.streams); results do not depend on it,
 * the best value depends on how many hardware queues the process has free — a caller may try a few and keep the fastest */
int pom_batch_set_streams(PomBatch* h, int32_t streams);
/* per-launch timing with HIP events on the launch streams: enable, step (at most 256 launches are kept), read the mean */
int pom_batch_profile(PomBatch* h, int enable);
int pom_batch_profile_read(PomBatch* h, double* mean_ms, int64_t* launches);
/* how a step is issued: envs per wavefront, lanes per env and kernel launches (sub-batches) per step */
int pom_batch_launch_shape(PomBatch* h, int32_t* envs_per_wave, int32_t* lanes_per_env, int32_t* launches_per_step);
/* how the launches of a several-tick call are issued: the POM_ISSUE_* in force (AUTO resolved; POM_ISSUE_CHAIN only while chained
 * launches are available to the handle) and the number of streams they go to */
int pom_batch_issue_info(PomBatch* h, int32_t* issue_mode, int32_t* streams);

/* chained launches since creation: out[0] launches issued, out[1] checks run (one per join that followed chained launches),
 * out[2] tiles that a check found left behind by a wavefront that could not play them, out[3] ticks replayed for those tiles */
int pom_batch_chain_stats(PomBatch* h, int64_t out[4]);

/* Self-test of the hand-off chained launches rest on, without the game: `launches` launches over `tiles` 7-KB records on `streams`
 * streams; every visit checks that its record is exactly what the visit before it left (all 1,792 dwords) and rewrites it.
 * out[0] records a visit found stale or torn (must be 0), out[1] dwords that differed, out[2] visits played, out[3] visits
 * expected, out[4] tiles whose final record / ticket word is not what `launches` clean visits leave, out[5] POM_CHAIN_E_* flags
 * raised.  POM_E_HIP where the device does not offer chained launches. */
int pom_chain_litmus(int32_t device, int64_t tiles, int32_t launches, int32_t streams, int64_t out[6]);

/* the hipStream_t the handle's work is ordered on (the one given at creation, or the library's own), so that a caller can
 * order its own device work against steps and observations with events instead of pom_batch_sync */
int pom_batch_stream(PomBatch* h, void** stream);

/* zero-copy view for device-side consumers (policies, observation kernels): packed records in tiles of 16 envs,
 * dword d of env e at base[(e / 16) * (16 * rec_dwords) + d * 16 + e % 16]; n_pad = envs the buffer holds (a multiple of 64);
 * layout in pomcpp_amd/csrc/pom_packed.h */
int pom_batch_device_view(PomBatch* h, void** base, int64_t* n_pad, int32_t* rec_dwords);

/*
 * Observation export: the current state of every env as dense planes for a training loop, written by one
 * kernel straight into caller-owned DEVICE memory (e.g. a torch tensor's data_ptr) on the handle's stream.  The reference has
 * no such function; the planes restate what its agents read off a State (Item codes and IS_* helpers bboard.hpp:54-109, Bomb
 * accessors :261-335, FLAME_ID :98-101, AgentInfo :228-245), one value per cell, row-major [y][x]:
 *    0 passage   1 rigid   2 wood (any flag)   3 Item::BOMB   4 flames   5 extra-bomb   6 incr-range   7 kick      (0 / 1)
 *    8..11       the cell shows agent 0..3 (per_agent: 8 = the viewer, 9..11 = agents id+1, id+2, id+3 mod 4)      (0 / 1)
 *    12 13 14    BMB_STRENGTH, BMB_TIME, BMB_DIR of the first live bomb in queue order on the cell (State::GetBomb order,
 *                bboard.cpp:277-287); also set under an agent, where the board shows no bomb
 *    15          timeLeft (clamped to 0..255) of the first live flame whose centre is the cell's FLAME_ID, on flame cells
 * planes:  [n][16][11][11] (per_agent = 0) or [n][4][16][11][11] (per_agent = 1: dead agents' views are written too), of
 *          uint8, IEEE half or float (all values are small integers, exact in each).
 * agent_attrs (nullable): int32 [n][4][8] = x, y, alive, ammo (maxBombCount - bombCount), bombCount, maxBombCount,
 *          bombStrength, canKick.   env_attrs (nullable): int32 [n][4] = timeStep, aliveAgents, status (1 done | 2 draw |
 *          4 timed out | 8 restarted: POM_RESET_AT_END put the env on this start state at the end of the last tick), winner
 *          (-1 = none) — the values pom_batch_status reports.
 *
 * POM_OBS_CODES — the compact form, uint8 [n][5][11][11], 605 bytes per env instead of 1,936 (the export is bound by the bytes it
 * writes): for training loops that expand the board themselves (an embedding look-up on the cell code).  The five arrays are the
 * ones a Pommerman observation carries:
 *    0  board: the small numbers of the reference's Item enum (bboard.hpp:54-71) — 0 passage, 1 rigid, 2 wood (any flag), 3 bomb,
 *       4 flames, 5 fog, 6 extra-bomb, 7 incr-range, 8 kick, 9 agent dummy, 10..13 agents 0..3; 255 for anything else
 *    1 2 3  bomb strength, life, direction (planes 12, 13, 14 above)        4  flame life (plane 15 above)
 * per_agent must be 0 (agents are named by id; agent_attrs says who is where).
 */
enum { POM_OBS_U8 = 0, POM_OBS_F16 = 1, POM_OBS_F32 = 2, POM_OBS_CODES = 3 };
enum { POM_OBS_PLANES = 16, POM_OBS_CODE_PLANES = 5, POM_OBS_AGENT_ATTRS = 8, POM_OBS_ENV_ATTRS = 4 };
int pom_batch_observe(PomBatch* h, void* planes_dev, int32_t dtype, int32_t per_agent, int32_t* agent_attrs_dev,
                      int32_t* env_attrs_dev);
/* pom_batch_step_device followed by pom_batch_observe in ONE launch: the kernel that plays the tick writes the observation of the
 * state it leaves behind while the tile is still in LDS (one read of the records and one launch per RL tick instead of two of
 * each; with POM_RESET_AT_END an env that has just finished shows its next start state, as pom_batch_observe would).  Same
 * outputs, bit for bit, as the two calls. */
int pom_batch_step_device_observe(PomBatch* h, const int32_t* moves_dev, void* planes_dev, int32_t dtype, int32_t per_agent,
                                  int32_t* agent_attrs_dev, int32_t* env_attrs_dev);

/* CLOSED-LOOP stepping: Step(State*, Move[4]) with THIS tick's moves from the caller every tick (the reference's real call shape:
 * Environment::Step collects act() of every agent and then steps, src/bboard/environment.cpp:139-149), without the whole chip
 * waiting for the caller's policy between two ticks.  One tick for the envs [first, first + count) only (whole tiles: first a
 * multiple of 16, count a multiple of 16 or reaching the batch's end), ONE launch on `stream` (NULL: the handle's stream), moves
 * from moves_dev = int32 [n][4] indexed by the env's number in the batch.  planes_dev != NULL: the observation of those envs
 * after the tick is written by the same launch (layout and arguments as pom_batch_observe, the arrays sized for the WHOLE batch;
 * only the range's part is written).  Nothing is forked or joined: the call is ordered by `stream` alone, so a caller that cuts
 * the batch into two or four ranges, each with a stream of its own carrying  policy(range) -> step(range) -> policy(range) ...,
 * has range A's policy running while range B steps — and the chains can be captured into a HIP graph (the call makes no
 * synchronising runtime call once the handle is settled: pom_batch_sync first).  The caller orders the streams behind whatever
 * put the batch into its present state.  The handle's tick (which keys pom_batch_step_random's move stream) does not advance.
 * Quad shape only (POM_E_ARG otherwise).  bench.py: other_configs.closed_loop_65536_envs. */
int pom_batch_step_device_range(PomBatch* h, int64_t first, int64_t count, const int32_t* moves_dev, void* stream, void* planes_dev,
                                int32_t dtype, int32_t per_agent, int32_t* agent_attrs_dev, int32_t* env_attrs_dev);
/* A stand-in for a learned policy in measurements and tests of the closed loop (NOT part of the stepper): one launch on `stream`
 * that writes Move[4] of the envs [first, first + count) into moves_dev (int32 [n][4]).  codes_dev != NULL: the POM_OBS_CODES
 * observation of the batch (uint8 [n][5][11][11]) — every byte of the range's observations is read and the moves depend on them;
 * NULL: the moves depend on (env, agent, tick) only.  The same function of its inputs on every call: tests recompute it. */
int pom_bench_policy(const uint8_t* codes_dev, int32_t* moves_dev, int64_t first, int64_t count, uint32_t tick, void* stream);

/* bboard::Step (include/bboard.hpp:668, src/bboard/step.cpp:9-284) for a single host State on the GPU (device 0): the literal
 * drop-in.  One launch per call: the kernel reads the State and Move[4] from a pinned page, plays the tick and writes the State
 * back; the call returns when that is done (about ten microseconds — the launch and the trip over PCIe, not the tick).
 * Re-entrant over distinct States like the reference's Step (performance_test.cpp:71-94 steps one env per std::thread): every
 * calling thread gets a pinned page and a stream of its own (up to 64 threads; more share), so threads stepping their own
 * States do not wait for each other.  Code that steps many States should still hand them to one PomBatch
 * (pom_batch_upload / pom_batch_step): a launch per State is latency, not throughput.
 * POM_E_UNREPRESENTABLE: the State holds a value the device record cannot hold; it is left as it is. */
int pom_step(void* state_1004, const int32_t moves[4]);

/* The same with Environment::Step's bookkeeping after the tick (src/bboard/environment.cpp:148-168): timeStep++, then
 * done / winner (-1 = none) / draw as pom_batch_status reports them, judged on this tick alone; max_steps > 0 also ends the
 * game at that timeStep.  The caller decides whether a game is stepped at all (the reference returns early from a finished
 * one, environment.cpp:125).  Outputs may be NULL.  ubflags: the POM_UB_* of this tick. */
int pom_env_step(void* state_1004, const int32_t moves[4], int32_t max_steps, int32_t* done, int32_t* winner, int32_t* draw,
                 uint32_t* ubflags);

#ifdef __cplusplus
}
#endif
#endif /* POM_BATCH_H_ */
