/*
 * pom_packed.h — the device-resident record of one board ("env") and its
 * lossless conversion from / to the 1004-byte boundary State (pom_state.h,
 * i.e. bboard::State, /root/reference/include/bboard.hpp:356-506).
 *
 * HBM layout: an array of 16-env TILES, struct-of-arrays inside a tile — dword d of env e lives at
 * buf[(e / 16) * 1792 + d * 16 + (e % 16)] (pom_rec_col, row stride POM_TILE_ENVS).  A tile is 7,168
 * contiguous bytes: the wavefront that owns it moves it with seven 1-KB instructions (16 bytes per lane),
 * every cache line full in both directions, and touches ONE region of memory instead of 112 rows that
 * lie n_pad * 4 bytes apart (rounds 1-2: 4.35 G env-steps/s at 524,288 envs against 6.5 G at 262,144 —
 * beyond the memory-side cache the strided rows cost DRAM and TLB locality).
 *
 * POM_REC_DWORDS = 112 dwords (448 B) per env instead of 251:
 *   [0..60]    board, 121 cells of 16 bits (cell c in dword c>>1, half c&1)
 *   [61]       timeStep
 *   [62]       aliveAgents:8 | bombs.index:8 | bombs.count:8 | flames.index:8
 *   [63]       flames.count:8 | status:8 | ubflags:16
 *   [64..71]   agents: A0[i] = x:8 | y:8 | bombCount:8 (signed) | canKick@24 | dead@25
 *                      A1[i] = maxBombCount:16 | bombStrength:16
 *   [72..91]   bombs.queue raw (all 20 slots: stale slots are state)
 *   [92..111]  flames.queue: x:8 | y:8 | timeLeft:8 (signed) | strength:8
 *
 * Cell code (16 bit) for board value v (Item, bboard.hpp:54-71):
 *   v < 0x4000                       -> v            (passage, rigid, bomb, fog, powerups, wood+flag)
 *   FLAMES <= v < FLAMES + 0x4000    -> 0x4000 | (v - FLAMES)
 *   AGENT0 <= v < AGENT0 + 4         -> 0x8000 | (v - AGENT0)
 * Every value the reference can produce from a valid start is representable;
 * anything else is rejected at upload (POM_E_UNREPRESENTABLE) instead of being
 * silently altered.
 */
#ifndef POM_PACKED_H_
#define POM_PACKED_H_

#include <stdint.h>

#include "pom_rng.h" /* POM_HD */
#include "pom_state.h"

enum {
    POM_REC_BOARD = 0,
    POM_REC_TIMESTEP = 61,
    POM_REC_META = 62,
    POM_REC_META2 = 63,
    POM_REC_AGENTS = 64,
    POM_REC_BOMBS = 72,
    POM_REC_FLAMES = 92,
    POM_REC_DWORDS = 112,
    POM_TILE_ENVS = 16,                              /* envs per tile of the device buffers = row stride of a column, in dwords */
    POM_TILE_DWORDS = POM_REC_DWORDS * POM_TILE_ENVS /* 1792 */
};

/* status byte of META2 */
enum {
    POM_ST_DONE = 1,     /* Environment::finished, environment.cpp:152-168 */
    POM_ST_DRAW = 2,     /* Environment::isDraw */
    POM_ST_WINNER_SHIFT = 2, /* 3 bits: agentWon + 1 (0 = nobody) */
    POM_ST_TIMEOUT = 32, /* timeStep reached max_steps (StartGame's loop bound, environment.cpp:71) */
    POM_ST_RESTARTED = 64 /* auto_reset at the end of the tick (POM_RESET_AT_END): the previous tick finished this env's episode and
                             put it on its next start state; the finished episode's record is in the terminal buffer */
};

enum { POM_C_PASSAGE = 0, POM_C_RIGID = 1, POM_C_BOMB = 3, POM_C_FLAME = 0x4000, POM_C_AGENT = 0x8000 };

/* where env e's column starts in a device buffer (dword units); its dword d is at pom_rec_col(e) + d * POM_TILE_ENVS */
POM_HD int64_t pom_rec_col(int64_t e) { return (e >> 4) * POM_TILE_DWORDS + (e & 15); }

POM_HD int pom_cell_encode(int32_t v) /* -1 if not representable */
{
    if (v >= 0 && v < 0x4000) return v;
    if (v >= POM_FLAMES && v < POM_FLAMES + 0x4000) return POM_C_FLAME | (v - POM_FLAMES);
    if (v >= POM_AGENT0 && v < POM_AGENT0 + POM_AGENT_COUNT) return POM_C_AGENT | (v - POM_AGENT0);
    return -1;
}

POM_HD int32_t pom_cell_decode(int e)
{
    if (e < 0x4000) return e;
    if (e < 0x8000) return POM_FLAMES + (e & 0x3FFF);
    return POM_AGENT0 + (e & 0x3FFF);
}

/*
 * Pack one boundary State into a record.  `rec` is addressed with a stride so
 * the same code fills a column of a device tile (stride = POM_TILE_ENVS) and a dense
 * record in host-side tests (stride = 1).  Returns 0, or 1 if a field does not
 * fit the record (nothing is written in that case... the caller zero-fills).
 */
POM_HD int pom_pack_state(const int32_t* st, uint32_t* rec, int64_t stride)
{
    const int32_t* board = st;                 /* @0    */
    const int32_t timeStep = st[121];          /* @484  */
    const int32_t alive = st[122];             /* @488  */
    const int32_t* agents = st + 123;          /* @492, 6 dwords each */
    const int32_t* bombs = st + 147;           /* @588  */
    const int32_t* flames = st + 169;          /* @676  */
    int bad = 0;

    for (int k = 0; k < 61; k++) {
        int lo = pom_cell_encode(board[2 * k]);
        int hi = (2 * k + 1 < POM_CELLS) ? pom_cell_encode(board[2 * k + 1]) : 0;
        bad |= (lo < 0) | (hi < 0);
        rec[(POM_REC_BOARD + k) * stride] = (uint32_t)(lo & 0xFFFF) | ((uint32_t)(hi & 0xFFFF) << 16);
    }
    rec[POM_REC_TIMESTEP * stride] = (uint32_t)timeStep;

    const int32_t bIdx = bombs[20], bCnt = bombs[21], fIdx = flames[80], fCnt = flames[81];
    bad |= (alive < -128) | (alive > 127);
    bad |= (bIdx < 0) | (bIdx >= POM_MAX_BOMBS) | (bCnt < 0) | (bCnt > POM_MAX_BOMBS);
    bad |= (fIdx < 0) | (fIdx >= POM_MAX_BOMBS) | (fCnt < 0) | (fCnt > 255);
    rec[POM_REC_META * stride] = ((uint32_t)alive & 0xFF) | ((uint32_t)bIdx << 8) | ((uint32_t)bCnt << 16) | ((uint32_t)fIdx << 24);
    rec[POM_REC_META2 * stride] = (uint32_t)fCnt & 0xFF; /* status, ubflags start clear */

    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        const int32_t* a = agents + 6 * i;
        const uint32_t flags = (uint32_t)a[5];
        const int kick = (flags & 0xFF) != 0, dead = ((flags >> 8) & 0xFF) != 0;
        bad |= (a[0] < 0) | (a[0] >= POM_BOARD_SIZE) | (a[1] < 0) | (a[1] >= POM_BOARD_SIZE);
        bad |= (a[2] < -128) | (a[2] > 127);
        bad |= (a[3] < -32768) | (a[3] > 32767) | (a[4] < 0) | (a[4] > 255);
        rec[(POM_REC_AGENTS + 2 * i) * stride] =
            (uint32_t)a[0] | ((uint32_t)a[1] << 8) | (((uint32_t)a[2] & 0xFF) << 16) | ((uint32_t)kick << 24) | ((uint32_t)dead << 25);
        rec[(POM_REC_AGENTS + 2 * i + 1) * stride] = ((uint32_t)a[3] & 0xFFFF) | ((uint32_t)a[4] << 16);
    }
    for (int k = 0; k < POM_MAX_BOMBS; k++)
        rec[(POM_REC_BOMBS + k) * stride] = (uint32_t)bombs[k];
    for (int k = 0; k < POM_MAX_BOMBS; k++) {
        const int32_t* f = flames + 4 * k;
        bad |= (f[0] < 0) | (f[0] >= POM_BOARD_SIZE) | (f[1] < 0) | (f[1] >= POM_BOARD_SIZE); /* also the stale slots */
        bad |= (f[2] < -128) | (f[2] > 127) | (f[3] < 0) | (f[3] > 255);
        rec[(POM_REC_FLAMES + k) * stride] =
            (uint32_t)f[0] | ((uint32_t)f[1] << 8) | (((uint32_t)f[2] & 0xFF) << 16) | ((uint32_t)f[3] << 24);
    }
    return bad;
}

POM_HD int32_t pom_sext8(uint32_t v) { return (int32_t)(int8_t)(v & 0xFF); }
POM_HD int32_t pom_sext16(uint32_t v) { return (int32_t)(int16_t)(v & 0xFFFF); }

/* inverse of pom_pack_state; the two padding bytes of each agent come out 0 */
POM_HD void pom_unpack_state(const uint32_t* rec, int64_t stride, int32_t* st)
{
    for (int k = 0; k < 61; k++) {
        uint32_t w = rec[(POM_REC_BOARD + k) * stride];
        st[2 * k] = pom_cell_decode((int)(w & 0xFFFF));
        if (2 * k + 1 < POM_CELLS) st[2 * k + 1] = pom_cell_decode((int)(w >> 16));
    }
    st[121] = (int32_t)rec[POM_REC_TIMESTEP * stride];
    const uint32_t m = rec[POM_REC_META * stride], m2 = rec[POM_REC_META2 * stride];
    st[122] = pom_sext8(m);
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        uint32_t a0 = rec[(POM_REC_AGENTS + 2 * i) * stride], a1 = rec[(POM_REC_AGENTS + 2 * i + 1) * stride];
        int32_t* a = st + 123 + 6 * i;
        a[0] = (int32_t)(a0 & 0xFF);
        a[1] = (int32_t)((a0 >> 8) & 0xFF);
        a[2] = pom_sext8(a0 >> 16);
        a[3] = pom_sext16(a1);
        a[4] = (int32_t)(a1 >> 16);
        a[5] = (int32_t)(((a0 >> 24) & 1) | (((a0 >> 25) & 1) << 8));
    }
    for (int k = 0; k < POM_MAX_BOMBS; k++)
        st[147 + k] = (int32_t)rec[(POM_REC_BOMBS + k) * stride];
    st[167] = (int32_t)((m >> 8) & 0xFF);
    st[168] = (int32_t)((m >> 16) & 0xFF);
    for (int k = 0; k < POM_MAX_BOMBS; k++) {
        uint32_t f = rec[(POM_REC_FLAMES + k) * stride];
        int32_t* o = st + 169 + 4 * k;
        o[0] = (int32_t)(f & 0xFF);
        o[1] = (int32_t)((f >> 8) & 0xFF);
        o[2] = pom_sext8(f >> 16);
        o[3] = (int32_t)(f >> 24);
    }
    st[249] = (int32_t)(m >> 24);
    st[250] = (int32_t)(m2 & 0xFF);
}

/* Chained launches (pom_chain.h): the tile words count visits in 28-bit fields (tickets in bits 63..36, stored visits in 27..0) and
 * are zeroed before either reaches 2^27.  A visit's distance from the first visit of the call whose launch it rides in is SIGNED:
 * launches of two calls can be in flight together, and a wavefront of the later call may draw a ticket of the earlier one.
 * Bit 28 (POM_CHAIN_POISON): a visitor could not play its tick (it waited out its time limit for the visit before it, or found
 * the tile stored through another XCD's L2).  Nobody steps a poisoned tile: its stored count stays at the number of ticks it
 * really played, and the host replays the rest after the next join (pom_runtime.h chain_settle). */
enum { POM_CHAIN_TICKET_SHIFT = 36, POM_CHAIN_COUNT_MASK = 0x0FFFFFFF, POM_CHAIN_POISON = 0x10000000 };
/* failure flags of chained launches (a device word, read back by chain_settle) */
enum { POM_CHAIN_E_TIMEOUT = 1, POM_CHAIN_E_XCD = 2, POM_CHAIN_E_UNEVEN = 4, POM_CHAIN_E_TAPE = 8 };
POM_HD uint32_t pom_chain_visit_distance(uint32_t visit, uint32_t first_visit_of_call)
{
    return (uint32_t)((int32_t)((visit - first_visit_of_call) << 4) >> 4);
}

#endif /* POM_PACKED_H_ */
