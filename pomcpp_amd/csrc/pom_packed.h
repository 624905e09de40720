/*
 * pom_packed.h — the device-resident record of one board ("env") and its
 * lossless conversion from / to the 1004-byte boundary State (pom_state.h,
 * i.e. bboard::State, /root/reference/include/bboard.hpp:356-506).
 *
 * HBM layout: an array of 16-env TILES, struct-of-arrays inside a tile — dword d >= 31 of env e lives at
 * buf[(e / 16) * 1280 + d * 16 + (e % 16)] (pom_rec_col, row stride POM_TILE_ENVS), and the board — the tile's first 31 rows,
 * 1,984 bytes — is laid out BY CELL: cell c of env e is byte c * 16 + (e % 16) of the tile (cells 121..123: zero), so that the
 * address of a cell in the wavefront's LDS copy of the tile is one shift-and-add of the cell number (round 5; with the cells of
 * an env packed four to a dword it was four instructions, and the tick does little else with its vector ALU than look at cells).  A tile is 5,120
 * contiguous bytes (40 lines of 128 B): the wavefront that owns it moves it with five 1-KB instructions (16 bytes per lane),
 * every cache line full in both directions, and touches ONE region of memory instead of 80 rows that
 * lie n_pad * 4 bytes apart (rounds 1-2: 4.35 G env-steps/s at 524,288 envs against 6.5 G at 262,144 —
 * beyond the memory-side cache the strided rows cost DRAM and TLB locality).
 *
 * POM_REC_DWORDS = 80 dwords (320 B) per env instead of 251 (rounds 1-4: 112, with 16-bit cells):
 *   [0..30]    board, 121 cells of 8 bits (a dense record — snapshot, terminal, host tests: cell c in byte c&3 of dword c>>2, the
 *              last three bytes 0; a tile: see above)
 *   [31]       timeStep
 *   [32..39]   agents: A0[i] = x:4 | y:4 | bombCount:8 (signed) @8 | canKick@16 | dead@17 | M0[i]:8 @24
 *                      A1[i] = maxBombCount:16 | bombStrength:8 @16 | M1[i]:8 @24
 *              The top bytes of the eight agent words carry what does not belong to an agent (two dwords saved: exactly five 1-KB
 *              moves per tile and direction):  M0 = aliveAgents (signed), bombs.index, bombs.count, flames.index
 *                                              M1 = flames.count, status, ubflags low byte, ubflags high byte
 *   [40..59]   bombs.queue raw (all 20 slots: stale slots are state)
 *   [60..79]   flames.queue: x:8 | y:8 | timeLeft:8 (signed) | strength:8
 *
 * Cell code (8 bit) for board value v (Item, bboard.hpp:54-71) — every value a game can reach from a valid start has one, and
 * the 256 codes are exactly used up:
 *   0 passage  1 rigid  2 Item::BOMB  3 / 4 / 5 extra-bomb / incr-range / kick  6..10 wood with flag 0..4  11..14 agent 0..3
 *   15 + id             a flame with FLAME_ID id = origin cell (0..120) and no power-up under it
 *   136 + 40 (f - 1) + 10 r + (d - 1)
 *                       a flame with power-up flag f = 1..3: the origin is NOT stored as a number but as where it lies from the
 *                       cell — d = 1..10 cells back along ray r (0: the cell is at +x of the origin, 1: -x, 2: +y, 3: -y).  A flagged
 *                       flame cell is a burnt wood, wood ends the ray that burns it, and SpawnFlame's rays run along the origin's
 *                       row and column (bboard.cpp:24-57, 198-263): in a reachable state the origin of such a cell always lies on
 *                       its row or column, at most 10 cells away.
 * So a cell is self-contained (no reference into the flame queue, whose slots are overwritten when it overflows), PopFlame's
 * "is this my flame" (bboard.cpp:148-180) is a compare on the code, and the record's board is 124 bytes instead of 244.
 * Values outside this set — fog, hand-written flame cells whose origin is not on their row / column, wood flags above 4, ... —
 * are rejected at upload (POM_E_UNREPRESENTABLE), never silently altered.
 */
#ifndef POM_PACKED_H_
#define POM_PACKED_H_

#include <stdint.h>

#include "pom_rng.h" /* POM_HD */
#include "pom_state.h"

enum {
    POM_REC_BOARD = 0,
    POM_REC_BOARD_DWORDS = 31,
    POM_REC_TIMESTEP = 31,
    POM_REC_AGENTS = 32,
    POM_REC_BOMBS = 40,
    POM_REC_FLAMES = 60,
    POM_REC_DWORDS = 80,
    POM_TILE_ENVS = 16,                              /* envs per tile of the device buffers = row stride of a column, in dwords */
    POM_TILE_DWORDS = POM_REC_DWORDS * POM_TILE_ENVS /* 1280 */
};
/* agent words */
enum { POM_AG_KICK = 1 << 16, POM_AG_DEAD = 1 << 17 };

/* The two "meta" words a record used to have, put together from / spread over the top bytes of its agent words:
 *   meta  = aliveAgents:8 | bombs.index:8 | bombs.count:8 | flames.index:8     (M0 of agents 0..3)
 *   meta2 = flames.count:8 | status:8 | ubflags:16                             (M1 of agents 0..3) */
POM_HD uint32_t pom_rec_meta(const uint32_t* rec, int64_t stride)
{
    return (rec[(POM_REC_AGENTS + 0) * stride] >> 24) | ((rec[(POM_REC_AGENTS + 2) * stride] >> 24) << 8) |
           ((rec[(POM_REC_AGENTS + 4) * stride] >> 24) << 16) | ((rec[(POM_REC_AGENTS + 6) * stride] >> 24) << 24);
}
POM_HD uint32_t pom_rec_meta2(const uint32_t* rec, int64_t stride)
{
    return (rec[(POM_REC_AGENTS + 1) * stride] >> 24) | ((rec[(POM_REC_AGENTS + 3) * stride] >> 24) << 8) |
           ((rec[(POM_REC_AGENTS + 5) * stride] >> 24) << 16) | ((rec[(POM_REC_AGENTS + 7) * stride] >> 24) << 24);
}
POM_HD void pom_rec_set_meta(uint32_t* rec, int64_t stride, uint32_t meta, uint32_t meta2)
{
    for (int i = 0; i < 4; i++) {
        uint32_t& a0 = rec[(POM_REC_AGENTS + 2 * i) * stride];
        uint32_t& a1 = rec[(POM_REC_AGENTS + 2 * i + 1) * stride];
        a0 = (a0 & 0x00FFFFFFu) | (((meta >> (8 * i)) & 0xFFu) << 24);
        a1 = (a1 & 0x00FFFFFFu) | (((meta2 >> (8 * i)) & 0xFFu) << 24);
    }
}

/* status byte of META2 */
enum {
    POM_ST_DONE = 1,     /* Environment::finished, environment.cpp:152-168 */
    POM_ST_DRAW = 2,     /* Environment::isDraw */
    POM_ST_WINNER_SHIFT = 2, /* 3 bits: agentWon + 1 (0 = nobody) */
    POM_ST_TIMEOUT = 32, /* timeStep reached max_steps (StartGame's loop bound, environment.cpp:71) */
    POM_ST_RESTARTED = 64 /* auto_reset at the end of the tick (POM_RESET_AT_END): the previous tick finished this env's episode and
                             put it on its next start state; the finished episode's record is in the terminal buffer */
};

enum {
    POM_C_PASSAGE = 0, POM_C_RIGID = 1, POM_C_BOMB = 2, POM_C_EXTRABOMB = 3, POM_C_INCRRANGE = 4, POM_C_KICK = 5,
    POM_C_WOOD = 6,    /* + flag 0..4 */
    POM_C_AGENT = 11,  /* + id */
    POM_C_FLAME = 15,  /* + origin cell */
    POM_C_FLAGGED = 136 /* + 40 (flag - 1) + 10 ray + (distance - 1) */
};

/* where env e's column starts in a device buffer (dword units); its dword d is at pom_rec_col(e) + d * POM_TILE_ENVS */
POM_HD int64_t pom_rec_col(int64_t e) { return (e >> 4) * POM_TILE_DWORDS + (e & 15); }

/* the code of a flame cell with power-up flag f (1..3) that lies d (1..10) cells along ray r (0 +x, 1 -x, 2 +y, 3 -y) from its origin */
POM_HD int pom_flagged_code(int f, int r, int d) { return POM_C_FLAGGED + 40 * (f - 1) + 10 * r + (d - 1); }

POM_HD int pom_cell_encode(int32_t v, int c /* the cell: y * 11 + x */) /* -1 if not representable */
{
    if (v == POM_PASSAGE) return POM_C_PASSAGE;
    if (v == POM_RIGID) return POM_C_RIGID;
    if (v == POM_BOMB) return POM_C_BOMB;
    if (v >= POM_EXTRABOMB && v <= POM_KICK) return POM_C_EXTRABOMB + (v - POM_EXTRABOMB);
    if (v >= POM_WOOD && v <= POM_WOOD + 4) return POM_C_WOOD + (v - POM_WOOD);
    if (v >= POM_AGENT0 && v < POM_AGENT0 + POM_AGENT_COUNT) return POM_C_AGENT + (v - POM_AGENT0);
    if (v >= POM_FLAMES && v < POM_FLAMES + (POM_CELLS << 3)) {
        const int id = (v - POM_FLAMES) >> 3, f = (v - POM_FLAMES) & 7;
        if (f == 0) return POM_C_FLAME + id;
        if (f > 3) return -1;
        const int cy = c / POM_BOARD_SIZE, cx = c - cy * POM_BOARD_SIZE, oy = id / POM_BOARD_SIZE, ox = id - oy * POM_BOARD_SIZE;
        if (oy == cy && ox != cx) return pom_flagged_code(f, cx > ox ? 0 : 1, cx > ox ? cx - ox : ox - cx);
        if (ox == cx && oy != cy) return pom_flagged_code(f, cy > oy ? 2 : 3, cy > oy ? cy - oy : oy - cy);
    }
    return -1;
}

POM_HD int32_t pom_cell_decode(int e, int c)
{
    if (e < POM_C_WOOD) return e <= POM_C_RIGID ? e : e == POM_C_BOMB ? POM_BOMB : POM_EXTRABOMB + (e - POM_C_EXTRABOMB);
    if (e < POM_C_AGENT) return POM_WOOD + (e - POM_C_WOOD);
    if (e < POM_C_FLAME) return POM_AGENT0 + (e - POM_C_AGENT);
    if (e < POM_C_FLAGGED) return POM_FLAMES + ((e - POM_C_FLAME) << 3);
    const int k = e - POM_C_FLAGGED, f = k / 40 + 1, r = (k % 40) / 10, d = k % 10 + 1;
    const int origin = c - d * (r == 0 ? 1 : r == 1 ? -1 : r == 2 ? POM_BOARD_SIZE : -POM_BOARD_SIZE);
    return POM_FLAMES + (origin << 3) + f;
}
/* FLAME_ID (bboard.hpp:98-101) of a flame code at cell c */
POM_HD int pom_flame_origin(int e, int c)
{
    if (e < POM_C_FLAGGED) return e - POM_C_FLAME;
    const int k = (e - POM_C_FLAGGED) % 40, r = k / 10, d = k % 10 + 1;
    return c - d * (r == 0 ? 1 : r == 1 ? -1 : r == 2 ? POM_BOARD_SIZE : -POM_BOARD_SIZE);
}

/* Where cell c of a record lives.  `rec` with stride 1: a dense record.  `rec` with stride POM_TILE_ENVS: the column of env
 * number `lane` (0..15) of a tile, i.e. rec = tile + lane — its board is the tile's byte c * 16 + lane. */
POM_HD int64_t pom_rec_cell_byte(int64_t stride, int lane, int c) { return stride == 1 ? (int64_t)c : (int64_t)c * POM_TILE_ENVS + lane - 4 * (int64_t)lane; }
POM_HD int pom_rec_cell(const uint32_t* rec, int64_t stride, int c, int lane = 0)
{
    return reinterpret_cast<const uint8_t*>(rec)[pom_rec_cell_byte(stride, lane, c)];
}
POM_HD void pom_rec_set_cell(uint32_t* rec, int64_t stride, int c, int code, int lane = 0)
{
    reinterpret_cast<uint8_t*>(rec)[pom_rec_cell_byte(stride, lane, c)] = (uint8_t)code;
}

/*
 * Pack one boundary State into a record.  `rec` is addressed with a stride so
 * the same code fills a column of a device tile (stride = POM_TILE_ENVS) and a dense
 * record in host-side tests (stride = 1).  Returns 0, or 1 if a field does not
 * fit the record (nothing is written in that case... the caller zero-fills).
 */
POM_HD int pom_pack_state(const int32_t* st, uint32_t* rec, int64_t stride, int lane = 0)
{
    const int32_t* board = st;                 /* @0    */
    const int32_t timeStep = st[121];          /* @484  */
    const int32_t alive = st[122];             /* @488  */
    const int32_t* agents = st + 123;          /* @492, 6 dwords each */
    const int32_t* bombs = st + 147;           /* @588  */
    const int32_t* flames = st + 169;          /* @676  */
    int bad = 0;

    for (int c = 0; c < 4 * POM_REC_BOARD_DWORDS; c++) {
        const int e = c < POM_CELLS ? pom_cell_encode(board[c], c) : 0; /* (the three bytes behind the board: 0) */
        bad |= e < 0;
        pom_rec_set_cell(rec, stride, c, e & 0xFF, lane);
    }
    rec[POM_REC_TIMESTEP * stride] = (uint32_t)timeStep;

    const int32_t bIdx = bombs[20], bCnt = bombs[21], fIdx = flames[80], fCnt = flames[81];
    bad |= (alive < -128) | (alive > 127);
    bad |= (bIdx < 0) | (bIdx >= POM_MAX_BOMBS) | (bCnt < 0) | (bCnt > POM_MAX_BOMBS);
    bad |= (fIdx < 0) | (fIdx >= POM_MAX_BOMBS) | (fCnt < 0) | (fCnt > 255);

    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        const int32_t* a = agents + 6 * i;
        const uint32_t flags = (uint32_t)a[5];
        const int kick = (flags & 0xFF) != 0, dead = ((flags >> 8) & 0xFF) != 0;
        bad |= (a[0] < 0) | (a[0] >= POM_BOARD_SIZE) | (a[1] < 0) | (a[1] >= POM_BOARD_SIZE);
        bad |= (a[2] < -128) | (a[2] > 127);
        bad |= (a[3] < -32768) | (a[3] > 32767) | (a[4] < 0) | (a[4] > 255);
        rec[(POM_REC_AGENTS + 2 * i) * stride] =
            ((uint32_t)a[0] & 0xF) | (((uint32_t)a[1] & 0xF) << 4) | (((uint32_t)a[2] & 0xFF) << 8) | (kick ? (uint32_t)POM_AG_KICK : 0u) | (dead ? (uint32_t)POM_AG_DEAD : 0u);
        rec[(POM_REC_AGENTS + 2 * i + 1) * stride] = ((uint32_t)a[3] & 0xFFFF) | (((uint32_t)a[4] & 0xFF) << 16);
    }
    pom_rec_set_meta(rec, stride, ((uint32_t)alive & 0xFF) | (((uint32_t)bIdx & 0xFF) << 8) | (((uint32_t)bCnt & 0xFF) << 16) | (((uint32_t)fIdx & 0xFF) << 24),
                     (uint32_t)fCnt & 0xFF); /* status, ubflags start clear */
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
POM_HD void pom_unpack_state(const uint32_t* rec, int64_t stride, int32_t* st, int lane = 0)
{
    for (int c = 0; c < POM_CELLS; c++) st[c] = pom_cell_decode(pom_rec_cell(rec, stride, c, lane), c);
    st[121] = (int32_t)rec[POM_REC_TIMESTEP * stride];
    const uint32_t m = pom_rec_meta(rec, stride), m2 = pom_rec_meta2(rec, stride);
    st[122] = pom_sext8(m);
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        uint32_t a0 = rec[(POM_REC_AGENTS + 2 * i) * stride], a1 = rec[(POM_REC_AGENTS + 2 * i + 1) * stride];
        int32_t* a = st + 123 + 6 * i;
        a[0] = (int32_t)(a0 & 0xF);
        a[1] = (int32_t)((a0 >> 4) & 0xF);
        a[2] = pom_sext8(a0 >> 8);
        a[3] = pom_sext16(a1);
        a[4] = (int32_t)((a1 >> 16) & 0xFF);
        a[5] = (int32_t)(((a0 & POM_AG_KICK) ? 1u : 0u) | ((a0 & POM_AG_DEAD) ? 0x100u : 0u));
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
