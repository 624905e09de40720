/*
 * pom_state.h — the byte layout of one Pommerman board state at the drop-in
 * boundary, plus the item / move / bomb-bitfield vocabulary of the game.
 *
 * This is the layout of the reference's `bboard::State`
 * (/root/reference/include/bboard.hpp:356-506; AgentInfo :228-245,
 * Flame :342-347, FixedQueue :115-188): 1004 bytes, all int32 except the two
 * bools of each agent.  The batch C-ABI (pom_batch.h) moves states across the
 * boundary in exactly this form so that an existing `bboard::State` can be
 * handed over with a reinterpret_cast.
 *
 * Plain C so that the oracle (gcc), the HIP library (hipcc) and ctypes /
 * cgo-style bindings can all include it.
 */
#ifndef POM_STATE_H_
#define POM_STATE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    POM_BOARD_SIZE     = 11,  /* bboard.hpp:17 */
    POM_CELLS          = 121,
    POM_AGENT_COUNT    = 4,   /* bboard.hpp:16 */
    POM_BOMB_LIFETIME  = 10,  /* bboard.hpp:21 */
    POM_FLAME_LIFETIME = 4,   /* bboard.hpp:24 */
    POM_MAX_BOMBS      = 20,  /* bboard.hpp:27 */
    POM_STATE_BYTES    = 1004
};

/* bboard.hpp:35-43 (Move) — Direction (:45-52) shares values 0..4 */
enum {
    POM_MOVE_IDLE = 0,
    POM_MOVE_UP = 1,
    POM_MOVE_DOWN = 2,
    POM_MOVE_LEFT = 3,
    POM_MOVE_RIGHT = 4,
    POM_MOVE_BOMB = 5
};

/* bboard.hpp:54-71 (Item) */
enum {
    POM_PASSAGE   = 0,
    POM_RIGID     = 1,
    POM_WOOD      = 2 << 8,   /* + powerup flag in the low bits */
    POM_BOMB      = 3,
    POM_FLAMES    = 4 << 16,  /* + (origin cell id << 3) + powerup flag */
    POM_FOG       = 5,
    POM_EXTRABOMB = 6,
    POM_INCRRANGE = 7,
    POM_KICK      = 8,
    POM_AGENT0    = 1 << 24   /* + agent id */
};

typedef struct PomAgentInfo {   /* bboard.hpp:228-245, 24 bytes */
    int32_t x;
    int32_t y;
    int32_t bombCount;
    int32_t maxBombCount;
    int32_t bombStrength;
    uint8_t canKick;
    uint8_t dead;
    uint8_t pad_[2];            /* compiler padding in the reference struct */
} PomAgentInfo;

typedef struct PomFlame {       /* bboard.hpp:342-347, 16 bytes */
    int32_t x;
    int32_t y;
    int32_t timeLeft;
    int32_t strength;
} PomFlame;

typedef struct PomBombQueue {   /* FixedQueue<Bomb,20>, bboard.hpp:115-121 */
    int32_t queue[POM_MAX_BOMBS];
    int32_t index;
    int32_t count;
} PomBombQueue;

typedef struct PomFlameQueue {  /* FixedQueue<Flame,20> */
    PomFlame queue[POM_MAX_BOMBS];
    int32_t index;
    int32_t count;
} PomFlameQueue;

typedef struct PomState {       /* bboard.hpp:356-506 */
    int32_t board[POM_BOARD_SIZE][POM_BOARD_SIZE]; /* [y][x]          @0   */
    int32_t timeStep;                              /*                 @484 */
    int32_t aliveAgents;                           /*                 @488 */
    PomAgentInfo agents[POM_AGENT_COUNT];          /*                 @492 */
    PomBombQueue bombs;                            /*                 @588 */
    PomFlameQueue flames;                          /*                 @676 */
} PomState;

#ifdef __cplusplus
static_assert(sizeof(PomState) == POM_STATE_BYTES, "State layout");
static_assert(offsetof(PomState, timeStep) == 484, "State layout");
static_assert(offsetof(PomState, agents) == 492, "State layout");
static_assert(offsetof(PomState, bombs) == 588, "State layout");
static_assert(offsetof(PomState, flames) == 676, "State layout");
#else
_Static_assert(sizeof(PomState) == POM_STATE_BYTES, "State layout");
_Static_assert(offsetof(PomState, timeStep) == 484, "State layout");
_Static_assert(offsetof(PomState, agents) == 492, "State layout");
_Static_assert(offsetof(PomState, bombs) == 588, "State layout");
_Static_assert(offsetof(PomState, flames) == 676, "State layout");
#endif

/*
 * Per-env flags raised when a tick runs into a situation in which the
 * reference itself has undefined behaviour.  The stepper never
 * crashes; it takes the documented fallback and sets the bit.
 */
enum {
    POM_UB_LOST_AGENT     = 1u << 0, /* step.cpp:36-46  roots exhausted: remaining agents do not move */
    POM_UB_NULL_BOMB      = 1u << 1, /* step.cpp:167    kicker on a BOMB cell without queue entry: direction set skipped */
    POM_UB_QUEUE_OVERFLOW = 1u << 2, /* step.cpp:191    21st bomb refused instead of overrunning bombDestinations[20] */
    POM_UB_REVERT_LOOP    = 1u << 3, /* step_utility.cpp:62-128 bounce-back chain cut after 8 hops */
    POM_UB_BAD_INDEX      = 1u << 4  /* an agent / cell index left its array; access skipped */
};

#ifdef __cplusplus
}
#endif
#endif /* POM_STATE_H_ */
