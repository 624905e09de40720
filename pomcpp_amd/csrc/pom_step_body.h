/*
 * pom_step_body.h — one simulation tick for ONE env, written against an
 * abstract per-lane store `A` so the identical source runs
 *   - on gfx950 with A = a column of the wavefront's LDS tile (pom_kernels.h), and
 *   - on the host with A = a plain array, ONLY inside tests/ (tests/emul), to fuzz
 *     the kernel's logic against the oracle without a GPU.  It is never a product
 *     CPU path: the shipped library has no host stepper.
 *
 * Semantics = the reference's `bboard::Step` (/root/reference/src/bboard/step.cpp:9-284)
 * with its helpers (step_utility.cpp, bboard.cpp State methods); every function
 * cites what it reproduces.  The design is MI355X-shaped rather than a translation:
 *   - lane = env; the wavefront executes the sequential tick 64 envs wide, loops run to
 *     the wave-wide maximum trip count under EXEC masking and whole phases are skipped
 *     when no lane needs them (hipcc emits s_cbranch_execz for the `if`s below);
 *   - the 4 agents live in VGPRs (A0/A1 words, accessed by 4-way selects, never by a
 *     runtime-indexed array, which would go to scratch);
 *   - destinations, moves and the dependency chain are nibble-packed into single
 *     registers so that "dynamic indexing" is a shift;
 *   - the recursive chain explosion (bboard.cpp:24-57,111-118,198-263) is an explicit
 *     frame stack in LDS (<= 21 frames), the tail-recursive bounce-back chain
 *     (step_utility.cpp:62-128) a bounded loop;
 *   - board cells are the 8-bit codes of pom_packed.h, bombs the reference's raw
 *     bit-packed ints, flames one dword each.
 *
 * G lanes per env.  The store says how many lanes of the wavefront work on one env (A::G): 1 in the
 * 64/32/16-envs-per-wavefront kernels and on the host, 4 ADJACENT lanes (a quad) in pom_step_kernel_quad.
 * All G lanes execute the tick in lock-step on identical register state ("replicated"): every decision is
 * taken G times, writes to the store go through the owner lane only (set_*).  Where the reference's order
 * does not matter the work is split instead — lane `sub` takes queue slots / flame arms / explosion rays
 * sub, sub+G, ... and writes with put_* — and the partial results are combined with the group reductions
 * gor / gmin / gadd (DPP quad permutes on the device, identities for G = 1).  Every split section is
 * read-only until a group vote has excluded the order-dependent cases, which then run replicated.
 *
 * Store interface A (all indices per lane):
 *   static constexpr int G;  int sub();  int gor(int) / gmin(int) / gadd(int) / gbcast<J>(int)
 *   void sync()   where split code goes on to READ what other lanes of the group have just written (free on the device: a
 *                 wavefront runs in lock-step; the four-lane host model of tests/emul needs to be told)
 *   set_x(...)  write by the owner lane only (replicated code)   put_x(...)  write by the calling lane (split code)
 *   int  cell(int c) / void set_cell(int c, int code)      c = y*11+x, 8-bit codes
 *   int  bomb(int slot) / void set_bomb(int slot, int v)    physical queue slot 0..19
 *   int  flame(int slot) / void set_flame(int slot, int v)  packed x|y<<8|time<<16|strength<<24
 *   int  bdest(int i) / void set_bdest(int i, int v)        byte: snapshot of bomb destinations
 *   int  frame(int d) / void set_frame(int d, int v)        explosion stack
 *   int  ag1(int i) / void set_ag1(int i, int v) / void put_ag1(int i, int v)   agent i's second word (maxBombCount | bombStrength << 16; the
 *                                                                 record's top byte is not the tick's business: kept as it is)
 *   void claims_clear() / void claim(int c) / int claims(int c)   a counter per cell (loop_b_todo): cleared by the env's lanes together
 *                                                                 (split), counted up by any of them (an atomic add on the device), read
 *                                                                 after a sync()
 */
#ifndef POM_STEP_BODY_H_
#define POM_STEP_BODY_H_

#include "pom_packed.h"

#if defined(__clang__)
#define POM_NOUNROLL _Pragma("nounroll")
#else
#define POM_NOUNROLL
#endif
/* Dynamic-trip loops are pinned: unrolled 4-8x by hipcc they blow the VGPR budget of the quad kernel (168 for three
 * wavefronts per SIMD) into scratch without saving a single LDS round trip. */
#define POM_N 11
#define POM_Q 20
#define POM_STACK_DEPTH 21 /* one frame per queued bomb + the initiating flame */

/* POM_DIAG builds (scripts/phase_stamps.py, never shipped) accumulate s_memtime deltas per phase */
#if defined(POM_DIAG) && defined(__HIP_DEVICE_COMPILE__)
#define POM_STAMP(L, k)                                   \
    do {                                                  \
        const long long now_ = (long long)clock64();      \
        (L).t_acc[k] += now_ - (L).t_last;                \
        (L).t_last = now_;                                \
    } while (0)
#else
#define POM_STAMP(L, k) ((void)0)
#endif
/* POM_TRUNC (diagnostic build, never shipped, results are wrong by design): the tick stops after phase L.trunc, so that the
 * instruction counters of ONE truncated launch minus those of the next shorter one give a phase's dynamic instruction count
 * (scripts/phase_insts.py) */
#if defined(POM_TRUNC)
#define POM_CUT(L, k)                 \
    do {                              \
        if ((L).trunc <= (k)) return; \
    } while (0)
#else
#define POM_CUT(L, k) ((void)0)
#endif
enum { POM_PH_LOAD = 0, POM_PH_FLAMES, POM_PH_AGENT_PREP, POM_PH_AGENT_LOOP, POM_PH_BOMB_PASS, POM_PH_BOMB_A, POM_PH_BOMB_B,
       POM_PH_TICK_BOMBS, POM_PH_EPILOGUE, POM_PH_STORE,
       POM_PH_X_SCAN, POM_PH_X_COMMIT, POM_PH_X_EPILOGUE, POM_PH_X_NEST, POM_PH_X_SHORT, POM_PH_RESTART, POM_PH_FLAMES_DEC, POM_PH_N }; /* X_*: inside explode_long / explode (their time is NOT in the phase that called them) */

struct PomLane { /* the register-resident part of one env */
    int a0[4];   /* x:4 | y:4 | bombCount:8 @8 | canKick@16 | dead@17 (the record's top byte is left as it came: pom_packed.h) — only ever indexed statically */
                 /* (the agents' second words — maxBombCount:16 | bombStrength:8 @16 — stay in the store, a.ag1(i): the tick looks at them
                  * when a bomb is planted, when one goes off by another's flame, when a power-up is picked up — four registers too many for that) */
    int alive, bIdx, bCnt, fIdx, fCnt;
    uint32_t ub;
#if defined(POM_DIAG)
    long long t_last, t_acc[POM_PH_N];
#endif
#if defined(POM_TRUNC)
    int trunc;
#endif
};

/* ---- cell-code predicates (Item helpers, bboard.hpp:73-109, on the 8-bit codes of pom_packed.h: 0 passage, 1 rigid, 2 bomb,
 * 3..5 power-ups, 6..10 wood, 11..14 agents, 15.. flames — every class a range) */
/* (bitwise | and one unsigned range compare, not || and &&: on lane-varying values hipcc turns the short-circuit forms into
 * nested exec-mask branches) */
POM_HD int pc_is_wood(int e) { return (unsigned)(e - POM_C_WOOD) < 5u; }
POM_HD int pc_is_powerup(int e) { return (unsigned)(e - POM_C_EXTRABOMB) < 3u; }
POM_HD int pc_is_walkable(int e) { return (int)pc_is_powerup(e) | (int)(e == 0); }
POM_HD int pc_is_flame(int e) { return e >= POM_C_FLAME; }
POM_HD int pc_is_agent(int e) { return (unsigned)(e - POM_C_AGENT) < 4u; }
POM_HD int pc_agent_id(int e) { return e - POM_C_AGENT; } /* of an agent cell */
POM_HD int pc_wood_flag(int e) { return (e - POM_C_WOOD) & 3; } /* WOOD_POWFLAG of a wood cell, bboard.hpp:106-109 (flag 4 reads as 0) */
POM_HD int pc_is_static_block(int e) { return (int)((unsigned)(e - POM_C_EXTRABOMB) < 8u) | (int)(e == POM_C_RIGID); } /* wood, power-up or rigid */
POM_HD int pc_blocks_bomb(int e) { return (int)((unsigned)(e - POM_C_EXTRABOMB) < 12u) | (int)(e == POM_C_RIGID); } /* ... or an agent */
POM_HD int pc_flag_item(int f) { return f == 0 ? 0 : f + (POM_C_EXTRABOMB - 1); } /* FlagItem, bboard.cpp:182-189: 1,2,3 -> extra-bomb, incr-range, kick */
/* the flame item SpawnFlameItem leaves (bboard.cpp:42-50) on the cell i steps along ray r from the origin cell c0; f: the flag of the
 * wood it burnt there (0: no wood, or wood without a power-up) */
POM_HD int pc_flame_code(int c0, int r, int i, int f) { return f == 0 ? POM_C_FLAME + c0 : POM_C_FLAGGED - 41 + 40 * f + 10 * r + i; }
/* PopFlame's test (bboard.cpp:160-176) for the cell i >= 1 steps along ray r from the popping flame's origin c0: -1 if the cell is
 * not a flame of that origin, else the item it gives way to */
POM_HD int pc_flame_pops_to(int e, int c0, int r, int i)
{
    const int k = e - (POM_C_FLAGGED - 1) - 10 * r - i; /* a flagged flame of this origin: 0, 40 or 80 */
    const int flagged = (int)(k == 0) | (int)(k == 40) | (int)(k == 80);
    const int item = POM_C_EXTRABOMB + (int)(k >= 40) + (int)(k >= 80);
    return e == POM_C_FLAME + c0 ? POM_C_PASSAGE : flagged ? item : -1;
}

/* ---- bomb word (bboard.hpp:261-335) */
POM_HD int pb_x(int b) { return b & 0xF; }
POM_HD int pb_y(int b) { return (b >> 4) & 0xF; }
POM_HD int pb_pos(int b) { return b & 0xFF; }
POM_HD int pb_id(int b) { return (b >> 8) & 0xF; }
POM_HD int pb_strength(int b) { return (b >> 12) & 0xF; }
POM_HD int pb_time(int b) { return (b >> 16) & 0xF; }
POM_HD int pb_dir(int b) { return (b >> 20) & 0xF; }
POM_HD int pb_set(int b, uint32_t mask, uint32_t v) { return (int)(((uint32_t)b & ~mask) + v); }

/* ---- 4-way register selects for the agents.
 * hipcc folds select(cond, load v[0], load v[1]) into a dynamically indexed load, which pins the whole
 * array in scratch (4 scratch_store + 1 scratch_load per use, hundreds of cycles each on gfx950).  Passing
 * each value through an empty asm makes it an opaque VGPR value, so the chain stays 3 v_cndmask. */
#if defined(__HIP_DEVICE_COMPILE__)
#define POM_IN_VGPR(x) asm("" : "+v"(x))
#else
#define POM_IN_VGPR(x) ((void)0)
#endif
POM_HD int sel4(int i, const int v[4])
{
    int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    POM_IN_VGPR(v0);
    POM_IN_VGPR(v1);
    POM_IN_VGPR(v2);
    POM_IN_VGPR(v3);
    const int lo = (i & 1) ? v1 : v0, hi = (i & 1) ? v3 : v2; /* two levels of selects: three v_cndmask, no branches */
    return (i & 2) ? hi : lo;
}
POM_HD uint32_t pick4(int i, const uint32_t v[4]) /* v[i], i in 0..3, as two levels of selects (a ternary chain on a lane-varying i becomes branches) */
{
    const uint32_t lo = (i & 1) ? v[1] : v[0], hi = (i & 1) ? v[3] : v[2];
    return (i & 2) ? hi : lo;
}
POM_HD void put4(int i, int v[4], int x)
{
    v[0] = i == 0 ? x : v[0];
    v[1] = i == 1 ? x : v[1];
    v[2] = i == 2 ? x : v[2];
    v[3] = i == 3 ? x : v[3];
}
POM_HD int ag_x(int a0) { return a0 & 0xF; }
POM_HD int ag_y(int a0) { return (a0 >> 4) & 0xF; }
POM_HD int ag_pos(int a0) { return a0 & 0xFF; } /* x | y << 4 */
POM_HD int ag_dead(int a0) { return (a0 >> 17) & 1; }
POM_HD int ag_kick(int a0) { return (a0 >> 16) & 1; }
POM_HD int ag_setpos(int a0, int x, int y) { return (a0 & ~0xFF) | x | (y << 4); }
POM_HD int ag_bombcount(int a0) { return pom_sext8((uint32_t)a0 >> 8); }
POM_HD int ag_bombcount_add(int a0, int d) { return (a0 & ~0xFF00) | (int)(((uint32_t)a0 + ((uint32_t)d << 8)) & 0xFF00u); } /* d may be -1: unsigned arithmetic */
POM_HD int ag_max_bombs(int a1) { return pom_sext16((uint32_t)a1); }
POM_HD int ag_strength(int a1) { return (a1 >> 16) & 0xFF; }
/* ConsumePowerup's bombStrength++ (step_utility.cpp:255): the record holds 8 bits of it — 255 stays 255 (a game would need 254 of
 * the range power-ups; the reference's int goes on counting) */
POM_HD int ag_strength_inc(int a1) { return ag_strength(a1) == 0xFF ? a1 : a1 + (1 << 16); }

POM_HD int wrap20(int p) /* 0 <= p < 40 */
{
    /* (one v_min_u32 instead of compare + select: on gfx950 a v_cmp's lane mask is not ready for the very next instruction, so every
     * select that cannot be scheduled away costs a compare, a wait and the select) */
    const unsigned u = (unsigned)p, w = u - (unsigned)POM_Q; /* wraps far above 40 when p < 20 */
    return (int)(u < w ? u : w);
}
POM_HD uint32_t pom_zero_bytes(uint32_t x) /* 0x80 in exactly the bytes of x that are 0 */
{
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
POM_HD int div11(int c) /* c / 11 for a cell index (exact for 0 <= c < 586): one full-rate multiply instead of v_mul_hi_u32's four passes */
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(c, 373) >> 12;
#else
    return c / POM_N;
#endif
}
POM_HD int oob(int x, int y) /* either coordinate outside 0..10 (as unsigned numbers a negative one is huge: one v_max_u32, one compare) */
{
    const unsigned ux = (unsigned)x, uy = (unsigned)y;
    return (ux > uy ? ux : uy) >= (unsigned)POM_N;
}

/* displacement of a Move / Direction (step_utility.cpp:9-31): 1 up(-y) 2 down(+y) 3 left(-x) 4 right(+x) */
/* (two bits per move in a constant — 1 + displacement, 1 for everything that is no direction, m = 0..15 — instead of two compares
 * and two selects each) */
POM_HD int mv_dx(int m) { return (int)((0x55555615u >> (2 * m)) & 3u) - 1; }
POM_HD int mv_dy(int m) { return (int)((0x55555561u >> (2 * m)) & 3u) - 1; }

template <class A>
struct PomStepper {
    A& a;
    PomLane& L;
    uint32_t oldp_ = 0; /* the agents' positions before the tick, a byte each (x | y << 4) */
    int irregular_ = 0; /* a bounce put an agent somewhere else than where he stood before the tick */
    int folded_ = 0;    /* between the bomb pass and TickBombs: the queued words already carry this tick's timer decrement */
    uint32_t occ_[4] = {0u, 0u, 0u, 0u}; /* explode_long's set of bomb cells */
    int occ_valid_ = 0, occ_keep_ = 0;   /* ... is current; ... may be kept for the next blast (TickBombs: nothing moves or is planted in between) */
    POM_HD PomStepper(A& a_, PomLane& l_) : a(a_), L(l_) {}

    POM_HD int bomb_at(int i) const { return a.bomb(wrap20(L.bIdx + i)); }
    POM_HD void set_bomb_at(int i, int v) { a.set_bomb(wrap20(L.bIdx + i), v); }

    /* first queue offset whose bomb sits on pos (x | y<<4), or -1:
     * HasBomb / GetBomb / GetBombIndex, bboard.cpp:265-311 */
    POM_HD int bomb_index(int pos) const
    {
        int r = 99; /* split: lane `sub` looks at offsets sub, sub+G, ...; the lowest hit of the group wins */
        POM_NOUNROLL
        for (int i = a.sub(); i < L.bCnt; i += A::G) {
            if (pb_pos(bomb_at(i)) == pos) {
                r = i;
                break;
            }
        }
        r = a.gmin(r);
        return r == 99 ? -1 : r;
    }
    POM_HD void put_bomb_at(int i, int v) { a.put_bomb(wrap20(L.bIdx + i), v); }
    /* the same scan done by ONE lane alone, for split code in which the lanes of a group look at different cells */
    POM_HD int bomb_index_alone(int pos) const
    {
        /* four slots per round, their reads in flight together (one LDS round trip per four bombs instead of one per bomb);
         * offsets past the count read stale slots of the ring, which the count test discards */
        POM_NOUNROLL
        for (int i = 0; i < L.bCnt; i += 4) {
            const int b0 = bomb_at(i), b1 = bomb_at(i + 1), b2 = bomb_at(i + 2), b3 = bomb_at(i + 3);
            int r = -1;
            r = ((int)(i + 3 < L.bCnt) & (int)(pb_pos(b3) == pos)) ? i + 3 : r;
            r = ((int)(i + 2 < L.bCnt) & (int)(pb_pos(b2) == pos)) ? i + 2 : r;
            r = ((int)(i + 1 < L.bCnt) & (int)(pb_pos(b1) == pos)) ? i + 1 : r;
            r = pb_pos(b0) == pos ? i : r;
            if (r >= 0) return r;
        }
        return -1;
    }

    POM_HD int get_agent(int x, int y) const /* bboard.cpp:289-299 */
    {
        const int want = x | (y << 4);
        int r = -1;
#pragma unroll
        for (int i = 3; i >= 0; i--)
            r = ((!ag_dead(L.a0[i])) & (int)(ag_pos(L.a0[i]) == want)) ? i : r;
        return r;
    }

    POM_HD void kill(int id) /* State::Kill, bboard.hpp:474-481 */
    {
        if (id >= POM_AGENT_COUNT) {
            L.ub |= POM_UB_BAD_INDEX;
            return;
        }
        int v = sel4(id, L.a0);
        if (!ag_dead(v)) {
            put4(id, L.a0, v | POM_AG_DEAD);
            L.alive--;
        }
    }

    POM_HD void owner_bombcount_dec(int b)
    {
        int id = pb_id(b);
        if (id >= POM_AGENT_COUNT) {
            L.ub |= POM_UB_BAD_INDEX;
            return;
        }
        put4(id, L.a0, ag_bombcount_add(sel4(id, L.a0), -1));
    }

    POM_HD void remove_at(int at) /* FixedQueue::RemoveAt, bboard.hpp:151-160 */
    {
        /* split: slot i-1 <- slot i for i = at+1+sub, +G, ...  Within one pass all reads are issued before all
         * writes (one ds_read, then one ds_write per wavefront), and slot i-1 was read one lane / one pass earlier */
        POM_NOUNROLL
        for (int i = at + 1 + a.sub(); i < L.bCnt; i += A::G) {
            const int v = bomb_at(i);
            put_bomb_at(i - 1, v);
        }
        a.sync(); /* whoever looks at the queue next sees all the moved slots */
        if (folded_) {
            /* The slot that falls out of the live range keeps its word (stale slots are state) — in the reference
             * the word as it was BEFORE TickBombs, which has not run yet there: take the folded decrement back out of it. */
            const int w = bomb_at(L.bCnt - 1);
            set_bomb_at(L.bCnt - 1, (w & (1 << 24)) ? (w & ~(1 << 24)) : (int)((uint32_t)w + (1u << 16)));
        }
        L.bCnt--;
    }

    /* ------------------------------------------------------------------ *
     * Explosions (bboard.cpp:24-57, 111-118, 191-263) without recursion.
     * One SpawnFlame activation = one frame: origin x,y, ray length s (clamped to 11: cells further out
     * are off the board), ray dir 0..3 (+x,-x,+y,-y), step i, and `rem`: the queue offset whose
     * ExplodeBombAt bookkeeping runs when the frame finishes (REM_TOP: ExplodeTopBomb + PopBomb,
     * bboard.cpp:93-97,191-196).  The ACTIVE frame lives in registers; only the suspended parents of a
     * chain sit in the store's frame rows, packed x:4|y:4|s:4|dir:3|i:4|rem:6.  One loop iteration = one
     * ray cell: one cell read, at most one cell write.
     * ------------------------------------------------------------------ */
    enum { REM_TOP = 62, REM_NONE = 63 };

    POM_HD int owner_strength(int b) /* agents[BMB_ID(b)].bombStrength: the owner's CURRENT strength, SURVEY Q3 */
    {
        const int owner = pb_id(b);
        if (owner < POM_AGENT_COUNT) return ag_strength(a.ag1(owner));
        L.ub |= POM_UB_BAD_INDEX;
        return 0;
    }

    /* SpawnFlame prologue, bboard.cpp:200-218 */
    POM_HD void flame_prologue(int x, int y, int strength, int e /* the origin cell, as it shows now */)
    {
        int slot = L.fIdx + L.fCnt; /* NextPos(): (index + count) % 20, count may exceed 20 (see tick_flames) */
        if (__builtin_expect(slot >= 2 * POM_Q, 0)) { /* only with more than 20 flames queued; a loop so that no division is speculated */
            POM_NOUNROLL
            while (slot >= POM_Q) slot -= POM_Q;
        } else {
            slot = wrap20(slot);
        }
        a.set_flame(slot, x | (y << 8) | (POM_FLAME_LIFETIME << 16) | ((strength & 0xFF) << 24));
        L.fCnt++;
        const int c = y * POM_N + x;
        if (pc_is_agent(e)) kill(pc_agent_id(e));
        a.set_cell(c, POM_C_FLAME + c);
    }
    POM_HD void flame_prologue(int x, int y, int strength) { flame_prologue(x, y, strength, a.cell(y * POM_N + x)); }

    /* bookkeeping after a frame's four rays: ExplodeTopBomb's PopBomb / ExplodeBombAt's RemoveAt */
    POM_HD void explode_epilogue(int rem)
    {
        if (rem == REM_TOP) {
            owner_bombcount_dec(bomb_at(0));
            L.bIdx = wrap20(L.bIdx + 1);
            L.bCnt--;
        } else if (rem != REM_NONE) {
            /* the slot is re-read after the nested chain: stale index, SURVEY Q2 */
            owner_bombcount_dec(bomb_at(rem));
            remove_at(rem);
        }
    }

    /* rays 0..3 = +x, -x, +y, -y.  Selects on the two bits of the ray number, not a ternary chain: with the ray number a lane's
     * own (lane r of a quad takes ray r) hipcc turns the chain into nested exec-mask branches — 20 scalar instructions and
     * three jumps for one multiply. */
    POM_HD static int ray_step(int dir) /* 1, -1, 11, -11: a byte each of one constant */
    {
        return (int)(int8_t)(0xF50BFF01u >> (8 * dir));
    }
    POM_HD static int ray_cell(int c0, int dir, int i)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return c0 + __mul24(i, ray_step(dir)); /* v_mad_i32_i24: full rate (v_mul_lo_u32 takes four passes) */
#else
        return c0 + i * ray_step(dir);
#endif
    }
    POM_HD static int ray_room(int x, int y, int s, int dir)
    {
        /* the cells between (x, y) and the edge in the four directions, a nibble each: 10 - x, x, 10 - y, y */
        const unsigned rooms = (unsigned)(POM_N - 1 - x) | ((unsigned)x << 4) | ((unsigned)(POM_N - 1 - y) << 8) | ((unsigned)y << 12);
        const int room = (int)((rooms >> (4 * dir)) & 0xFu);
        return room < s ? room : s;
    }

    /* Kill (bboard.hpp:474-481) for every agent in the bit set `victims`; all lanes of the group hold the same set */
    POM_HD void kill_set(int victims)
    {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int hit = (victims >> j) & 1 & !ag_dead(L.a0[j]);
            L.a0[j] |= hit << 17;
            L.alive -= hit;
        }
    }

    /* `top_word`: for rem == REM_TOP the caller may pass the head of the queue it has just looked at (-1: not known) */
    POM_HD void explode(int x, int y, int strength, int rem, int top_word = -1)
    {
        int s = strength < 0 ? 0 : strength > POM_N ? POM_N : strength;
        POM_STAMP(L, POM_PH_TICK_BOMBS); /* diagnostic builds: what ran before the blast is booked on the phase of the top explosions */
        /* Split fast path.  Read-only scan of the four rays (lane `sub` takes rays sub, sub+G, ...): how far does
         * each reach, and does any of them touch a BOMB or an agent cell?  If none does, no chain and no kill can
         * happen along the rays, every cell belongs to exactly one ray, and the order +x,-x,+y,-y is immaterial:
         * the rays are then written in parallel.  Otherwise nothing has been written yet and the literal
         * sequential engine below runs (replicated). */
        if (s > 2) { /* a long blast: no look-ahead, the segment engine takes ray after ray (explode_long) */
            explode_long(x, y, strength, rem);
            return;
        }
        {
            const int c0 = y * POM_N + x;
            const int e0 = a.cell(c0); /* the origin, read together with the rays' first cells: nothing is written before the vote */
            int chains = 0, victims = 0;
            int chain_key = 0; /* this lane's ray stopped at a cell with a queued bomb: distance << 12 | agent there | his id << 1 */
            uint32_t lens = 0; /* reach of ray r in nibble r */
            uint32_t ends = 0; /* power-up flag of the wood a ray ends on, 2 bits per ray (only a ray's last cell can be wood) */
            /* A ray of such a blast is one or two cells: both are read up front and judged without a loop (SpawnFlameItem's order, bboard.cpp:
             * 24-57: a queued bomb under a BOMB / agent item stops the look AT the cell, rigid before it, wood on it).  (Round 5; until then
             * two nested loops with a branch per cell — twice the instructions.  The four-at-a-time classification of scan_ray does not
             * pay for so few cells: measured, 1,011 -> 1,084 VALU per wavefront-tick on the headline.) */
            POM_NOUNROLL
            for (int r = a.sub(); r < 4; r += A::G) {
                const int lim = ray_room(x, y, s, r); /* 0, 1 or 2 cells */
                const int on1 = lim >= 1, on2 = lim >= 2;
                const int c1 = ray_cell(c0, r, on1); /* (no cell: the origin again, ignored) */
                const int e1 = a.cell(c1);
                const int ag1 = pc_is_agent(e1);
                int q1 = 0; /* a queued bomb sits under the item of cell 1 */
                if (on1 & ((int)(e1 == POM_C_BOMB) | ag1)) {
                    const int cy = div11(c1);
                    q1 = bomb_index_alone((c1 - cy * POM_N) | (cy << 4)) >= 0;
                }
                const int took1 = on1 & !q1 & (int)(e1 != POM_C_RIGID);
                const int wood1 = took1 & pc_is_wood(e1);
                int len = took1, vict = (took1 & ag1) << (pc_agent_id(e1) & 3), flag = wood1 ? pc_wood_flag(e1) : 0;
                if (q1) {
                    chains = 1;
                    chain_key = (1 << 12) | ag1 | ((pc_agent_id(e1) & 3) << 1);
                }
                if (took1 & !wood1 & on2) { /* the ray reaches its second cell (a blast of strength 2: skipped by a wavefront without one) */
                    const int c2 = ray_cell(c0, r, 2);
                    const int e2 = a.cell(c2);
                    const int ag2 = pc_is_agent(e2);
                    int q2 = 0;
                    if ((int)(e2 == POM_C_BOMB) | ag2) {
                        const int cy = div11(c2);
                        q2 = bomb_index_alone((c2 - cy * POM_N) | (cy << 4)) >= 0;
                    }
                    const int took2 = (int)(q2 == 0) & (int)(e2 != POM_C_RIGID);
                    len += took2;
                    vict |= (took2 & ag2) << (pc_agent_id(e2) & 3);
                    flag = (took2 & pc_is_wood(e2)) ? pc_wood_flag(e2) : flag;
                    if (q2) {
                        chains = 1;
                        chain_key = (2 << 12) | ag2 | ((pc_agent_id(e2) & 3) << 1);
                    }
                }
                victims |= vict; /* killed, the ray goes on (bboard.cpp:26-29) */
                ends |= (uint32_t)flag << (2 * r);
                lens |= (uint32_t)len << (4 * r);
            }
            if (__builtin_expect(!a.gor(chains), 1)) {
                flame_prologue(x, y, strength, e0);
                const int killed = a.gor(victims);
                if (killed) kill_set(killed); /* Kill, bboard.hpp:474-481, for every agent a ray met */
                POM_NOUNROLL
                for (int r = a.sub(); r < 4; r += A::G) { /* no reads: the look has seen every cell it writes (at most two per ray) */
                    const int len = (lens >> (4 * r)) & 0xF, f = (int)((ends >> (2 * r)) & 3u);
                    if (len >= 1) a.put_cell(ray_cell(c0, r, 1), pc_flame_code(c0, r, 1, len == 1 ? f : 0));
                    if (len >= 2) a.put_cell(ray_cell(c0, r, 2), pc_flame_code(c0, r, 2, f));
                }
                a.sync(); /* the next blast's scan looks at cells other lanes' rays have just written */
                if (rem == REM_TOP && top_word != -1) { /* nothing touched the queue: the head is still the word the caller saw */
                    owner_bombcount_dec(top_word);
                    L.bIdx = wrap20(L.bIdx + 1);
                    L.bCnt--;
                } else {
                    explode_epilogue(rem);
                }
                return;
            }
            /* a short blast that meets a queued bomb (nothing has been written yet): the look-and-commit engine; with a lane per
             * ray the scan above IS the engine's first look — each lane hands over what it saw on its ray */
            if (A::G == 4) {
                const int r = a.sub();
                explode_long(x, y, strength, rem, 1, (int)((lens >> (4 * r)) & 0xF), (int)((ends >> (2 * r)) & 3u), victims, chain_key, e0);
                return;
            }
        }
        explode_long(x, y, strength, rem);
    }

    /* The cells on which a queued bomb sits, as a 121-bit set in four words (bit c = y * 11 + x): one look at every queue slot,
     * the lanes of the group each taking a quarter of them.  Replicated result. */
    POM_HD void bomb_cells(uint32_t occ[4]) const
    {
        occ[0] = occ[1] = occ[2] = occ[3] = 0u;
        if (A::G == 1) { /* one lane per env: a plain loop over the live bombs (nothing to keep in flight for other lanes) */
            POM_NOUNROLL
            for (int k = 0; k < L.bCnt; k++) {
                const int b = bomb_at(k);
                const int idx = pb_y(b) * POM_N + pb_x(b);
                const uint32_t m = idx < POM_CELLS ? 1u << (idx & 31) : 0u;
                const int wd = idx >> 5;
                occ[0] |= wd == 0 ? m : 0u;
                occ[1] |= wd == 1 ? m : 0u;
                occ[2] |= wd == 2 ? m : 0u;
                occ[3] |= wd == 3 ? m : 0u;
            }
            return;
        }
        constexpr int NS = POM_Q / A::G; /* slots per lane */
        int w[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) w[q] = bomb_at(a.sub() + q * A::G); /* offsets past the count read stale slots: filtered below */
#pragma unroll
        for (int q = 0; q < NS; q++) {
            const int k = a.sub() + q * A::G;
            const int idx = pb_y(w[q]) * POM_N + pb_x(w[q]);
            const uint32_t m = (k < L.bCnt && idx < POM_CELLS) ? 1u << (idx & 31) : 0u;
            const int wd = idx >> 5;
            occ[0] |= wd == 0 ? m : 0u;
            occ[1] |= wd == 1 ? m : 0u;
            occ[2] |= wd == 2 ? m : 0u;
            occ[3] |= wd == 3 ? m : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) occ[k] = (uint32_t)a.gor((int)occ[k]);
    }
    POM_HD static int cell_in(const uint32_t occ[4], int c)
    {
        const int w = c >> 5;
        return (int)((pick4(w, occ) >> (c & 31)) & 1u);
    }
    /* first queue offset whose bomb sits on pos, or -1 — bomb_index() with every lane's slots fetched at once */
    POM_HD int bomb_index_wide(int pos) const
    {
        if (A::G == 1) return bomb_index(pos);
        constexpr int NS = POM_Q / A::G;
        int w[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) w[q] = bomb_at(a.sub() + q * A::G);
        int r = 99;
#pragma unroll
        for (int q = NS - 1; q >= 0; q--) {
            const int k = a.sub() + q * A::G;
            r = (k < L.bCnt && pb_pos(w[q]) == pos) ? k : r;
        }
        r = a.gmin(r);
        return r == 99 ? -1 : r;
    }

    /* One ray of a blast, read-only: from distance `start` to `lim`, four cells per LDS round trip.  Reports how far the
     * flame gets (`len`: the last cell that takes it, start - 1 if none), the flag of the wood it ends on (`ends`, with
     * `wood` set), the agents it kills on the way (`vict`), and — instead of going on — the first BOMB / agent cell that is in
     * `occ`, the cells with a queued bomb (`chain`: its distance, 0 = none; `info`: agent there | his id << 1):
     * SpawnFlameItem, bboard.cpp:24-57, without its writes.
     * The four cell codes of a round are classified together, a byte each of one dword (round 5; until then cell by cell, ~25
     * instructions each): with the top bit of every byte cleared, adding 128 - k sets it again iff the byte is >= k, so a class of
     * codes — rigid 1, Item::BOMB 2, wood 6..10, agents 11..14 — is two adds and an and-not for all four cells; the ray ends at the
     * LOWEST flagged byte, everything below it takes the flame. */
    POM_HD void scan_ray(int c0, int r, int start, int lim, const uint32_t occ[4], int& len, int& ends, int& wood, int& vict, int& chain,
                         int& info)
    {
        constexpr int W = 4; /* cells per round */
        len = start - 1;
        ends = wood = vict = chain = info = 0;
        const uint32_t H = 0x80808080u;
        POM_NOUNROLL
        for (int i0 = start; i0 <= lim; i0 += W) {
            uint32_t d = 0;
#pragma unroll
            for (int q = 0; q < W; q++) /* (clamped: stays on the ray) */
                d |= (uint32_t)a.cell(ray_cell(c0, r, i0 + q <= lim ? i0 + q : lim)) << (8 * q);
            const int n = lim - i0 + 1; /* how many of the round's cells are cells of the ray (>= 1) */
            const uint32_t onray = n >= W ? H : H >> (8 * (W - n));
            const uint32_t l = d & 0x7F7F7F7Fu, lo = ~d; /* lo: top bit set where the code is below 128 */
            const uint32_t ge1 = l + 0x7F7F7F7Fu, ge2 = l + 0x7E7E7E7Eu, ge3 = l + 0x7D7D7D7Du, ge6 = l + 0x7A7A7A7Au, ge11 = l + 0x75757575u,
                           ge15 = l + 0x71717171u;
            const uint32_t rigid = ge1 & ~ge2 & lo, bombc = ge2 & ~ge3 & lo, woodc = ge6 & ~ge11 & lo & onray, agentc = ge11 & ~ge15 & lo & H;
            /* where the ray ends whatever the bombs do: rigid (before it), wood (on it), the end of the ray */
            const uint32_t fixed = ((rigid | woodc) & onray & H) | (~onray & H);
            const uint32_t fixed_low = fixed & (0u - fixed); /* its lowest flag: 0 if none, then fixed_low - 1 is every byte */
            /* BOMB / agent cells before that: SpawnFlameItem sets off the first queued bomb on such a cell (bboard.cpp:30-40) */
            uint32_t cand = (bombc | agentc) & H & (fixed_low - 1u);
            uint32_t chain_at = 0u;
            POM_NOUNROLL
            while (cand) {
                const int q = __builtin_ctz(cand) >> 3;
                if (cell_in(occ, ray_cell(c0, r, i0 + q))) { /* (a candidate lies in front of the ray's end: a cell of the ray) */
                    chain_at = cand & (0u - cand);
                    break;
                }
                cand &= cand - 1u;
            }
            const uint32_t stop = fixed | chain_at;
            const uint32_t at = stop & (0u - stop), below = at - 1u;
            const int before = __builtin_popcount(below & H); /* cells of this round in front of the one the ray ends at (W: it goes on) */
            const int on_wood = (at & woodc) != 0u;
            len = i0 - 1 + before + on_wood;
            uint32_t dead = agentc & below; /* agents on cells the flame takes: killed, the ray goes on (bboard.cpp:26-29) */
            POM_NOUNROLL
            while (dead) {
                const int q = __builtin_ctz(dead) >> 3;
                dead &= dead - 1u;
                vict |= 1 << pc_agent_id((int)((d >> (8 * q)) & 0xFFu));
            }
            if (at) {
                const int e = (int)((d >> (8 * before)) & 0xFFu);
                if (on_wood) {
                    ends = pc_wood_flag(e);
                    wood = 1;
                }
                if (at & chain_at) {
                    chain = i0 + before;
                    info = pc_is_agent(e) | ((pc_agent_id(e) & 3) << 1);
                }
                break;
            }
        }
    }

    /* Long blasts (strength > 2: power-ups collected, the stress boards).  SpawnFlame's recursion (bboard.cpp:24-57,111-118,
     * 198-263) as a loop of LOOK and COMMIT over all four rays at once, lane r of the group taking ray r (with one lane per
     * env the same code takes them in turn).  Look (scan_ray, read-only): every ray from where it stands to its end, or to the
     * first cell with a queued bomb under a BOMB / agent item.  The rays are independent of each other except through such a
     * bomb — its nested explosion may change anything — so the FIRST ray (in the reference's order +x, -x, +y, -y) that meets
     * one decides: the rays before it are written in full, that ray up to the bomb's cell, the rays after it are forgotten
     * and looked at again after the nested explosion has returned (frame suspended on the stack, bboard.cpp:30-40; picked up
     * at SpawnFlameItem's tail, bboard.cpp:42-56).  Without such a bomb a whole blast is one look and one commit.
     * "A queued bomb under the item" is answered from a set of bomb cells built once per call (bomb_cells); explosions only
     * remove bombs, so the set can only be too large, and the one cell a look settles on is checked against the queue itself
     * (which also yields the bomb's index): a cell found empty there is struck from the set and the look repeated. */
    POM_HD void explode_long(int x, int y, int strength, int rem, int seen = 0, int seen_len = 0, int seen_ends = 0, int seen_vict = 0,
                             int seen_chain = 0, int seen_origin = 0)
    {
        int s = strength < 0 ? 0 : strength > POM_N ? POM_N : strength;
        POM_STAMP(L, POM_PH_TICK_BOMBS);
        if (seen) flame_prologue(x, y, strength, seen_origin); /* the caller has looked at the origin cell already */
        else flame_prologue(x, y, strength);
        int dir = 0, i = 1, sp = 0; /* the rays before `dir` are done, ray `dir` goes on at distance i, the others start at 1 */
        constexpr int NR = 4 / A::G;
        /* the set is kept from one blast of TickBombs to the next (top_explosions): between them bombs only disappear, and a set that
         * is too large is what the look-ups below are written for */
        uint32_t (&occ)[4] = occ_;
        if (!occ_valid_) bomb_cells(occ);
        occ_valid_ = occ_keep_;
        POM_NOUNROLL
        for (;;) {
            const int c0 = y * POM_N + x;
            int first = 0x7FFFFFFF; /* ray << 16 | distance << 12 | info of the first bomb cell met, smallest ray wins */
            int rstar, jq = -1;
            if (A::G == 1) {
                /* one lane per env: the rays in turn, each looked at and committed before the next (their cells are disjoint,
                 * so this is the same as looking at all of them first), stopping at the first ray that meets a bomb cell */
                POM_NOUNROLL
                for (int r = dir; r < 4 && first == 0x7FFFFFFF; r++) {
                    const int rs1 = r == dir ? i : 1;
                    int len1, ends1, wood1, vict1, chain1, info1;
                    scan_ray(c0, r, rs1, ray_room(x, y, s, r), occ, len1, ends1, wood1, vict1, chain1, info1);
                    if (chain1) {
                        const int c = ray_cell(c0, r, chain1);
                        const int cy = div11(c);
                        jq = bomb_index((c - cy * POM_N) | (cy << 4));
                        if (jq < 0) { /* the set was too large here: strike the cell and look at this ray again */
                            const uint32_t m = ~(1u << (c & 31));
                            const int w = c >> 5;
                            occ[0] &= w == 0 ? m : ~0u; occ[1] &= w == 1 ? m : ~0u; occ[2] &= w == 2 ? m : ~0u; occ[3] &= w == 3 ? m : ~0u;
                            r--;
                            continue;
                        }
                        first = (r << 16) | (chain1 << 12) | info1;
                    }
                    POM_NOUNROLL
                    for (int d = rs1; d <= len1; d++)
                        a.put_cell(ray_cell(c0, r, d), pc_flame_code(c0, r, d, (wood1 && d == len1) ? ends1 : 0));
                    kill_set(vict1);
                }
                rstar = first == 0x7FFFFFFF ? 4 : first >> 16;
            } else {
            int rs[NR], rlen[NR], rends[NR], rwood[NR], rvict[NR];
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const int r = a.sub() + q * A::G;
                rs[q] = r == dir ? i : 1;
                int chain = 0, info = 0;
                rlen[q] = rs[q] - 1;
                rends[q] = rwood[q] = rvict[q] = 0;
                if (seen) { /* the caller's scan of this very blast (explode): ray r of lane r from distance 1, nothing written since */
                    rlen[q] = seen_len;
                    rends[q] = seen_ends; /* 0 unless the ray ends on a flagged wood */
                    rwood[q] = 1;
                    rvict[q] = seen_vict;
                    chain = seen_chain >> 12;
                    info = seen_chain & 0xFFF;
                } else if (r >= dir) scan_ray(c0, r, rs[q], ray_room(x, y, s, r), occ, rlen[q], rends[q], rwood[q], rvict[q], chain, info);
                const int key = (r << 16) | (chain << 12) | info;
                first = (chain != 0 && key < first) ? key : first;
            }
            first = a.gmin(first);
            seen = 0;
            rstar = first == 0x7FFFFFFF ? 4 : first >> 16;
            if (rstar < 4) { /* the cell the look settled on: which bomb is it? */
                const int c = ray_cell(c0, rstar, (first >> 12) & 0xF);
                const int cy = div11(c);
                jq = bomb_index_wide((c - cy * POM_N) | (cy << 4));
                if (jq < 0) { /* none any more (the set is only ever too large): strike the cell and look again */
                    const uint32_t m = ~(1u << (c & 31));
                    const int w = c >> 5;
                    occ[0] &= w == 0 ? m : ~0u; occ[1] &= w == 1 ? m : ~0u; occ[2] &= w == 2 ? m : ~0u; occ[3] &= w == 3 ? m : ~0u;
                    continue;
                }
            }
            POM_STAMP(L, POM_PH_X_SCAN);
            int victims = 0;
#pragma unroll
            for (int q = 0; q < NR; q++) { /* commit: rays dir .. rstar */
                const int r = a.sub() + q * A::G;
                if (r >= dir && r <= rstar) {
                    /* ray rstar stopped its look AT the bomb's cell, so its len / vict cover exactly the cells before it */
                    victims |= rvict[q];
                    POM_NOUNROLL
                    for (int d = rs[q]; d <= rlen[q]; d++)
                        a.put_cell(ray_cell(c0, r, d), pc_flame_code(c0, r, d, (rwood[q] && d == rlen[q]) ? rends[q] : 0));
                }
            }
            {
                const int killed = a.gor(victims);
                if (killed) kill_set(killed); /* (mostly nobody stands in the blast: a wavefront without a victim skips the sixteen instructions) */
            }
            }
            POM_STAMP(L, POM_PH_X_COMMIT);
            if (rstar == 4) { /* the blast is complete: the caller's bookkeeping, then back into the parent */
                explode_epilogue(rem);
                POM_STAMP(L, POM_PH_X_EPILOGUE);
                if (sp == 0) return;
                sp--;
                const int fr = a.frame(sp);
                x = fr & 0xF; y = (fr >> 4) & 0xF; s = (fr >> 8) & 0xF;
                dir = (fr >> 12) & 7; i = (fr >> 15) & 0xF; rem = (fr >> 19) & 63;
                /* SpawnFlameItem tail, bboard.cpp:42-56, for the cell whose bomb has just gone off: read again */
                const int pc0 = y * POM_N + x;
                const int c = ray_cell(pc0, dir, i);
                const int e = a.cell(c);
                int go_on = 0;
                if (e != POM_C_RIGID) {
                    const int was_wood = pc_is_wood(e);
                    a.set_cell(c, pc_flame_code(pc0, dir, i, was_wood ? pc_wood_flag(e) : 0));
                    go_on = !was_wood;
                }
                if (go_on) i++;
                else { dir++; i = 1; }
                continue;
            }
            /* SpawnFlameItem head, bboard.cpp:26-40: the agent on the cell dies, the bomb under it goes off */
            const int d = (first >> 12) & 0xF;
            if (first & 1) kill((first >> 1) & 3);
            if (sp >= POM_STACK_DEPTH) { /* cannot happen with <= 20 queued bombs */
                L.ub |= POM_UB_BAD_INDEX;
                return;
            }
            a.set_frame(sp, x | (y << 4) | (s << 8) | (rstar << 12) | (d << 15) | (rem << 19));
            sp++;
            const int st2 = owner_strength(bomb_at(jq));
            const int c = ray_cell(c0, rstar, d);
            y = div11(c);
            x = c - y * POM_N;
            rem = jq;
            flame_prologue(x, y, st2);
            s = st2 < 0 ? 0 : st2 > POM_N ? POM_N : st2;
            dir = 0;
            i = 1;
            POM_STAMP(L, POM_PH_X_NEST);
        }
    }

    /* ------------------------------------------------------------------ */
    /* TickFlames, step_utility.cpp:208-222, first half: timeLeft-- of every queued flame.  Returns the head of the queue as
     * it stands afterwards (0 if the queue is empty) and the number of rounds the pop loop may take. */
    POM_HD void flames_dec(int& top, int& n)
    {
        top = 0;
        n = L.fCnt;
        if (L.fCnt <= 0) return;
        /* split over the lanes (offsets i and i+20 fall to the same lane) */
        int p = L.fIdx + a.sub();
        p = wrap20(p);
        POM_NOUNROLL
        for (int i = a.sub(); i < L.fCnt; i += A::G) {
            const int f = a.flame(p);
            const int nf = (f & ~0xFF0000) | ((f - 0x10000) & 0xFF0000);
            a.put_flame(p, nf);
            if (i == 0) top = nf;
            p = wrap20(p + A::G);
        }
        top = a.template gbcast<0>(top);
    }
    /* second half: PopFlame (bboard.cpp:148-180) while the head has run out, by the env's own lanes */
    POM_HD void flame_pops(int top, int n)
    {
        POM_NOUNROLL
        for (int k = 0; k < n; k++) {
            const int f = k == 0 ? top : a.flame(L.fIdx); /* the first look needs no trip to the queue */
            if (((f >> 16) & 0xFF) != 0) break; /* nothing pops, so flames[0] stays what it is for the remaining rounds */
            /* PopFlame: every flame cell on the +-strength cross that carries this origin's id gives way to the item
             * under it.  The cells are independent: the four arms are split over the lanes, the centre is the owner's */
            const int x = f & 0xFF, y = (f >> 8) & 0xFF;
            int s = (f >> 24) & 0xFF;
            s = s > POM_N ? POM_N : s; /* cells further out are out of bounds anyway */
            if (!oob(x, y)) {
                const int c0 = y * POM_N + x;
                /* the centre: a flame cell at its own origin never carries a flag (it is not on a ray of its flame) */
                if (a.cell(c0) == POM_C_FLAME + c0) a.set_cell(c0, POM_C_PASSAGE);
                POM_NOUNROLL
                for (int r = a.sub(); r < 4; r += A::G) {
                    const int lim = ray_room(x, y, s, r);
                    { /* the arm's first cell without a loop: most flames are of strength 1 */
                        const int c = ray_cell(c0, r, lim >= 1); /* (no cell: the centre again, ignored) */
                        const int to = pc_flame_pops_to(a.cell(c), c0, r, 1);
                        if ((int)(lim >= 1) & (int)(to >= 0)) a.put_cell(c, to);
                    }
                    POM_NOUNROLL
                    for (int i = 2; i <= lim; i++) {
                        const int c = ray_cell(c0, r, i);
                        const int to = pc_flame_pops_to(a.cell(c), c0, r, i);
                        if (to >= 0) a.put_cell(c, to);
                    }
                }
            }
            L.fIdx = wrap20(L.fIdx + 1);
            L.fCnt--;
        }
    }

    /* AgentBombChainReversion, step_utility.cpp:62-128; mvp = moves, one nibble per agent */
    /* `bombs_move` = 0: the caller knows that no queued bomb has a direction.  A resting bomb on the origin cell changes
     * nothing about the bounce (the agent item is put there either way and the chain ends), so the queue is not searched. */
    POM_HD void chain_reversion(uint32_t mvp, int id, int bombs_move = 1)
    {
        POM_NOUNROLL
        for (int hop = 0;; hop++) {
            if (hop >= 8) {
                L.ub |= POM_UB_REVERT_LOOP;
                return;
            }
            const int av = sel4(id, L.a0);
            const int m = (mvp >> (4 * id)) & 0xF;
            /* OriginPosition, step_utility.cpp:33-55: one step against the move */
            const int ox = ag_x(av) - mv_dx(m), oy = ag_y(av) - mv_dy(m);
            if (oob(ox, oy)) return;
            const int origin_agent = get_agent(ox, oy);
            const int okey = (ox + 1) | ((oy + 1) << 4);
            int bd = -1; /* the first bomb heading for the origin cell; split: lane `sub` looks at offsets sub, sub+G, ... */
            if (bombs_move) {
                bd = 99;
                POM_NOUNROLL
                for (int i = a.sub(); i < L.bCnt; i += A::G) {
                    if (a.bdest(i) == okey) {
                        bd = i;
                        break;
                    }
                }
                bd = a.gmin(bd);
                if (bd == 99) bd = -1;
            }
            put4(id, L.a0, ag_setpos(av, ox, oy));
            irregular_ |= (ox | (oy << 4)) != (int)((oldp_ >> (8 * id)) & 0xFF);
            a.set_cell(oy * POM_N + ox, POM_C_AGENT + id);
            if (origin_agent != -1) {
                id = origin_agent;
                continue;
            }
            if (bd != -1) {
                const int b = bomb_at(bd);
                const int dir = pb_dir(b);
                if (mv_dx(dir) == 0 && mv_dy(dir) == 0) { /* bounced back onto a resting bomb */
                    a.set_cell(oy * POM_N + ox, POM_C_AGENT + id);
                    return;
                }
                const int bx = ox - mv_dx(dir), by = oy - mv_dy(dir);
                if (oob(bx, by)) { /* reference would write outside the board (step_utility.cpp:111) */
                    L.ub |= POM_UB_BAD_INDEX;
                    return;
                }
                const int has_agent = get_agent(bx, by);
                set_bomb_at(bd, pb_set(pb_set(b, 0xF00000u, 0), 0xFFu, (uint32_t)bx + ((uint32_t)by << 4)));
                a.set_cell(by * POM_N + bx, POM_C_BOMB);
                if (has_agent != -1) {
                    id = has_agent;
                    continue;
                }
            }
            return;
        }
    }

    POM_HD int bomb_target_key(int b) const /* DesiredPosition(Bomb), step_utility.cpp:57-60; (x+1)|(y+1)<<4 */
    {
        const int d = pb_dir(b);
        return (pb_x(b) + mv_dx(d) + 1) | ((pb_y(b) + mv_dy(d) + 1) << 4);
    }

    /* HasBombCollision + ResolveBombCollision fused (step_utility.cpp:279-329): returns whether
     * bomb k collided; if so the colliders (and k) are already idled and the kicker bounced */
    POM_HD int bomb_collision(uint32_t mvp, int k)
    {
        const int b = bomb_at(k);
        const int key = bomb_target_key(b);
        int collided = 0;
        POM_NOUNROLL
        for (int i = k + a.sub(); i < L.bCnt; i += A::G) { /* split over the lanes; offset k itself can never hit */
            const int o = bomb_at(i);
            if (o != b && bomb_target_key(o) == key) {
                put_bomb_at(i, pb_set(o, 0xF00000u, 0));
                collided = 1;
            }
        }
        collided = a.gor(collided);
        if (collided && pb_dir(b) != 0) {
            const int nb = pb_set(b, 0xF00000u, 0);
            set_bomb_at(k, nb);
            const int ag = get_agent(pb_x(nb), pb_y(nb));
            if (ag > -1) {
                const int m = (mvp >> (4 * ag)) & 0xF;
                if (m != POM_MOVE_IDLE && m != POM_MOVE_BOMB) {
                    unforeseen_ = 1; /* (loop B's selection of bombs, loop_b_todo, does not cover what a bounce moves) */
                    chain_reversion(mvp, ag);
                    /* the bomb word is re-read: the chain may have moved this very bomb */
                    const int cur = bomb_at(k);
                    a.set_cell(pb_y(cur) * POM_N + pb_x(cur), POM_C_BOMB);
                }
            }
        }
        return collided;
    }

    /* ------------------------------------------------------------------ */
    /* one bomb of loop A, step.cpp:197-226 */
    POM_HD void loop_a_bomb(int mvp, int oldp, int k, int bombs_move)
    {
        const int b = bomb_at(k);
        const int bx = pb_x(b), by = pb_y(b), d = pb_dir(b);
        const int tx = bx + mv_dx(d), ty = by + mv_dy(d);
        int blocked = oob(tx, ty);
        if (!blocked) {
            const int e = a.cell(ty * POM_N + tx);
            blocked = pc_blocks_bomb(e);
        }
        if (blocked) {
            set_bomb_at(k, pb_set(b, 0xF00000u, 0));
            const int ag = get_agent(bx, by);
            if (ag > -1) {
                const int m = (mvp >> (4 * ag)) & 0xF;
                const int av = sel4(ag, L.a0);
                const int was = (oldp >> (8 * ag)) & 0xFF;
                if ((int)(m != POM_MOVE_IDLE) & (int)(m != POM_MOVE_BOMB) & (int)((ag_x(av) | (ag_y(av) << 4)) != was)) {
                    chain_reversion(mvp, ag, bombs_move);
                    if (get_agent(bx, by) == -1) a.set_cell(by * POM_N + bx, POM_C_BOMB);
                }
            }
        }
    }

    /* bboard::Step, step.cpp:9-284, in four pieces so that a kernel can put its own version of the two event loops
     * (flame pops, top-bomb explosions) between them; step() is the plain sequence. */
    POM_HD void step(const int mv_in[4]) { step_packed(pack_moves(mv_in)); }
    /* moves as a nibble per agent; anything outside 0..5 acts as "no displacement, not IDLE, not BOMB" -> code 6 */
    POM_HD static int clamp_move(int m) { return ((unsigned)m <= 5u) ? m : 6; }
    POM_HD static uint32_t pack_moves(const int mv_in[4])
    {
        uint32_t mvp = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) mvp |= (uint32_t)clamp_move(mv_in[i]) << (4 * i);
        return mvp;
    }
    /* a quad per env: lane m hands in agent m's move */
    POM_HD uint32_t pack_moves_quad(int mine) const
    {
        const int c = clamp_move(mine);
        return (uint32_t)a.template gbcast<0>(c) | ((uint32_t)a.template gbcast<1>(c) << 4) | ((uint32_t)a.template gbcast<2>(c) << 8) |
               ((uint32_t)a.template gbcast<3>(c) << 12);
    }
    POM_HD void step_packed(uint32_t mvp)
    {
        int ftop, fn, btop, bn;
        flames_dec(ftop, fn); /* step.cpp:15 */
        POM_STAMP(L, POM_PH_FLAMES_DEC);
        POM_CUT(L, 10);
        flame_pops(ftop, fn);
        POM_STAMP(L, POM_PH_FLAMES);
        POM_CUT(L, 20);
        step_middle(mvp, btop, bn);
        top_explosions(btop, bn);
        POM_STAMP(L, POM_PH_TICK_BOMBS);
    }
    /* TickBombs' second half, step_utility.cpp:233-244: explode the head of the queue while its timer has run out; `n` = 0
     * if the tick had no bombs */
    POM_HD void top_explosions(int top, int n)
    {
        occ_valid_ = 0;
        occ_keep_ = 1;
        POM_NOUNROLL
        for (int k = 0; k < n && L.bCnt > 0; k++) {
            const int c = k == 0 ? top : bomb_at(0); /* the first look needs no trip to the queue */
            if (pb_time(c) != 0) break;
            explode(pb_x(c), pb_y(c), pb_strength(c), REM_TOP, c);
        }
        occ_valid_ = occ_keep_ = 0;
    }
    /* FillPositions / FillDestPos (step_utility.cpp:130-152), dead agents included, and the two questions that decide what
     * the tick has to do about them.  Positions travel as a byte per agent (x | y << 4), destinations as (x+1) | (y+1) << 4.
     * With a quad per env lane m works out agent m's and the four are exchanged with quad broadcasts; the pairwise logic —
     * FixSwitchMove, ResolveDependencies: prep_dependencies — only runs for the envs in which some agent's destination
     * touches another agent's cell (`contact`; dead agents included: SURVEY Q9).  Otherwise, the usual case (the agents are
     * far apart), FixSwitchMove changes nothing and everyone is a root in index order.  `clash`: two live agents on one cell. */
    POM_HD void prep_positions(const uint32_t mvp, uint32_t& oldp, uint32_t& dstp, int& deadmask, int& contact, int& clash)
    {
        oldp = dstp = 0;
        deadmask = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) deadmask |= ag_dead(L.a0[i]) << i;
        contact = 0;
        clash = 0; /* two live agents on one cell */
        if (A::G == 4) {
            const int m = a.sub();
            const int av = sel4(m, L.a0);
            const int mvm = (int)((mvp >> (4 * m)) & 0xF);
            const int pxm = ag_x(av), pym = ag_y(av);
            const int dxm = pxm + mv_dx(mvm), dym = pym + mv_dy(mvm);
            const int pos8 = ag_pos(av);
            oldp = (uint32_t)a.template gbcast<0>(pos8) | ((uint32_t)a.template gbcast<1>(pos8) << 8) |
                   ((uint32_t)a.template gbcast<2>(pos8) << 16) | ((uint32_t)a.template gbcast<3>(pos8) << 24);
            const int d8 = ((dxm + 1) & 0xF) | (((dym + 1) & 0xF) << 4);
            dstp = (uint32_t)a.template gbcast<0>(d8) | ((uint32_t)a.template gbcast<1>(d8) << 8) |
                   ((uint32_t)a.template gbcast<2>(d8) << 16) | ((uint32_t)a.template gbcast<3>(d8) << 24);
            /* my destination as a position byte: off the board it carries a nibble 15 or 11, which no position has.  Both
             * questions — is my destination somebody's cell; do I, alive, share my cell with a live agent — are asked of all
             * four position bytes at once: an exact zero-byte test of oldp ^ (byte in every byte), my own byte masked out. */
            const uint32_t want = (uint32_t)((dxm & 0xF) | ((dym & 0xF) << 4));
            const uint32_t others = ~(0x80u << (8 * m));
            const uint32_t dead80 = (((uint32_t)deadmask * 0x00204081u) & 0x01010101u) << 7; /* 0x80 in the bytes of dead agents */
            const uint32_t hit_d = pom_zero_bytes(oldp ^ (want * 0x01010101u)) & others;
            const uint32_t hit_p = pom_zero_bytes(oldp ^ ((uint32_t)pos8 * 0x01010101u)) & others & ~dead80;
            int mine = (int)(hit_d != 0u) | ((int)((hit_p != 0u) & !ag_dead(av)) << 1);
            mine = a.gor(mine);
            contact = mine & 1;
            clash = mine >> 1;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int px = ag_x(L.a0[i]), py = ag_y(L.a0[i]);
                const int mvi = (int)((mvp >> (4 * i)) & 0xF);
                const int dx = px + mv_dx(mvi), dy = py + mv_dy(mvi);
                oldp |= (uint32_t)(px | (py << 4)) << (8 * i);
                dstp |= (uint32_t)(((dx + 1) & 0xF) | (((dy + 1) & 0xF) << 4)) << (8 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (i != j) contact |= ((((dstp >> (8 * i)) & 0xF) - 1) & 0xF) == ((oldp >> (8 * j)) & 0xF) &&
                                           ((((dstp >> (8 * i + 4)) & 0xF) - 1) & 0xF) == ((oldp >> (8 * j + 4)) & 0xF);
        }
    }
    /* FixSwitchMove (step_utility.cpp:154-170) and ResolveDependencies (step_utility.cpp:172-205): destinations corrected in
     * place; dependency / roots as nibbles, 0xF = -1 */
    POM_HD void prep_dependencies(const uint32_t mvp, const uint32_t oldp, const int deadmask, uint32_t& dstp, uint32_t& dep, uint32_t& roots,
                                  int& nroots)
    {
        int px[4], py[4], dx[4], dy[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            px[i] = (int)((oldp >> (8 * i)) & 0xF);
            py[i] = (int)((oldp >> (8 * i + 4)) & 0xF);
            const int mvi = (int)((mvp >> (4 * i)) & 0xF);
            dx[i] = px[i] + mv_dx(mvi);
            dy[i] = py[i] + mv_dy(mvi);
        }
        roots = 0xFFFF;
        nroots = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = i; j < 4; j++) {
                if ((int)(dx[i] == px[j]) & (int)(dy[i] == py[j]) & (int)(dx[j] == px[i]) & (int)(dy[j] == py[i])) {
                    dx[i] = px[i]; dy[i] = py[i];
                    dx[j] = px[j]; dy[j] = py[j];
                }
            }
        }
        /* ResolveDependencies (step_utility.cpp:172-205); dependency / roots as nibbles, 0xF = -1 */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int is_root = 1;
            if (!((deadmask >> i) & 1)) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j == i) continue;
                    if (is_root & !((deadmask >> j) & 1) & (int)(dx[i] == px[j]) & (int)(dy[i] == py[j])) {
                        dep = (dep & ~(0xFu << (4 * j))) | ((uint32_t)i << (4 * j));
                        is_root = 0;
                    }
                }
            }
            if (is_root) {
                roots = (roots & ~(0xFu << (4 * nroots))) | ((uint32_t)i << (4 * nroots));
                nroots++;
            }
        }
        dstp = 0; /* FixSwitchMove may have changed them */
#pragma unroll
        for (int i = 0; i < 4; i++) dstp |= (uint32_t)(((dx[i] + 1) & 0xF) | (((dy[i] + 1) & 0xF) << 4)) << (8 * i);
    }

    /* Which bombs does loop B (step.cpp:230-278) have to visit?  A RESTING bomb's turn is: HasBombCollision over the bombs from itself
     * on — some bomb with another word whose DesiredPosition is this bomb's cell — and, without one, the test of its own cell (PASSAGE
     * becomes BOMB, a flame sets it off).  Nothing happens on that turn, and nothing any other bomb's turn does can make something
     * happen on it, if the bomb's cell shows neither PASSAGE nor a flame and NO other bomb stands on or heads for that cell: a bomb
     * that is never moved, never idled, and whose cell nobody writes.  Those are skipped; the rest — every bomb with a direction, every
     * bomb whose cell is "ripe", every bomb whose cell is claimed twice (positions and targets of all bombs counted in a byte map in
     * LDS) — are visited in queue order as before.  What the selection cannot foresee — an explosion inside the loop (cells and queue
     * change), a bounced agent (AgentBombChainReversion moves agents and bombs) — sets `unforeseen_`, and the loop then visits every
     * bomb after the current one.  Stress boards: 12 -> 7 trips of the loop per wavefront-tick (round 5). */
    int unforeseen_ = 0;
    POM_HD uint32_t loop_b_todo()
    {
        a.claims_clear();
        a.sync();
        POM_NOUNROLL
        for (int k = a.sub(); k < L.bCnt; k += A::G) {
            const int b = bomb_at(k);
            const int d = pb_dir(b);
            const int bx = pb_x(b), by = pb_y(b), tx = bx + mv_dx(d), ty = by + mv_dy(d);
            a.claim(by * POM_N + bx);
            if ((int)(d != 0) & !oob(tx, ty) & (int)((mv_dx(d) | mv_dy(d)) != 0)) a.claim(ty * POM_N + tx);
        }
        a.sync();
        uint32_t f = 0;
        POM_NOUNROLL
        for (int k = a.sub(); k < L.bCnt; k += A::G) {
            const int b = bomb_at(k);
            const int c = pb_y(b) * POM_N + pb_x(b);
            const int e = a.cell(c);
            f |= (uint32_t)((int)(pb_dir(b) != 0) | (int)(e == POM_C_PASSAGE) | pc_is_flame(e) | (int)(a.claims(c) >= 2)) << k;
        }
        return (uint32_t)a.gor((int)f);
    }
    /* one bomb of loop B, step.cpp:232-277 */
    POM_HD void loop_b_bomb(uint32_t mvp, int k)
    {
        int b = bomb_at(k);
        const int bx = pb_x(b), by = pb_y(b), d = pb_dir(b);
        int tx = bx, ty = by; /* where a flame may set a bomb off at the end of this iteration */
        if (d == 0) {
            if (bomb_collision(mvp, k)) return;
            /* A resting bomb "moves" onto its own cell (step.cpp:243-272 with target == position): its word does not
             * change, its old cell still holds a bomb (itself), so all that is left is the cell test — PASSAGE becomes
             * BOMB, a flame sets off the first bomb queued on the cell (GetBombIndex: possibly an earlier one, SURVEY
             * Q8); a static item there makes the reference set the already resting bomb to rest. */
            const int c = by * POM_N + bx;
            const int e = a.cell(c);
            if (e == POM_C_PASSAGE) a.set_cell(c, POM_C_BOMB);
            if (!pc_is_flame(e)) return;
        } else {
            tx = bx + mv_dx(d);
            ty = by + mv_dy(d);
            int free_way = !oob(tx, ty);
            int tc = 0, te = 0;
            if (free_way) {
                tc = ty * POM_N + tx;
                te = a.cell(tc);
                free_way = !pc_is_static_block(te);
            }
            if (!free_way) {
                set_bomb_at(k, pb_set(b, 0xF00000u, 0));
                return;
            }
            if (bomb_collision(mvp, k)) return;
            b = bomb_at(k);
            set_bomb_at(k, pb_set(b, 0xFFu, (uint32_t)tx + ((uint32_t)ty << 4)));
            if (bomb_index(bx | (by << 4)) < 0 && a.cell(by * POM_N + bx) == POM_C_BOMB)
                a.set_cell(by * POM_N + bx, POM_C_PASSAGE);
            te = a.cell(tc);
            if (pc_is_walkable(te)) a.set_cell(tc, POM_C_BOMB);
            if (!pc_is_flame(te)) return;
        }
        /* ExplodeBombAt(GetBombIndex(target)), bboard.cpp:111-118 */
        const int j = bomb_index(tx | (ty << 4));
        const int jb = bomb_at(j);
        unforeseen_ = 1;
        explode(pb_x(jb), pb_y(jb), owner_strength(jb), j);
    }

    /* everything between the flame pops and the top-bomb explosions; returns the head of the bomb queue after the timer
     * decrement and the number of rounds the explosion loop may take */
    POM_HD void step_middle(const uint32_t mvp, int& top, int& n)
    {
        top = 0;
        n = 0;

        /* FillPositions / FillDestPos / FixSwitchMove / ResolveDependencies (step_utility.cpp:130-205): prep_positions and
         * prep_dependencies above */
        uint32_t oldp = 0, dstp = 0;
        uint32_t dep = 0xFFFF, roots = 0x3210;
        int nroots = 4;
        int deadmask = 0, contact = 0, clash = 0;
        prep_positions(mvp, oldp, dstp, deadmask, contact, clash);
        oldp_ = oldp;
        POM_CUT(L, 24);
        if (contact) prep_dependencies(mvp, oldp, deadmask, dstp, dep, roots, nroots);
        POM_CUT(L, 27);
        const int ouroboros = nroots == 0;
        int posb[4]; /* the agents' position bytes (x | y << 4) */
#pragma unroll
        for (int i = 0; i < 4; i++) posb[i] = (int)((oldp >> (8 * i)) & 0xFF);

        /* HasBomb(x, y) is only ever asked about the moving agent's own cell (step.cpp:89,127,152,172) and bombs
         * do not move during the agent loop: one pass over the queue answers it for all four agents */
        int on_bomb = 0; /* bit 4: some queued bomb has a direction (before this tick's kicks) */
        POM_NOUNROLL
        for (int k = a.sub(); k < L.bCnt; k += A::G) { /* split over the lanes, OR-combined */
            const int bw = bomb_at(k);
            const int bp = pb_pos(bw);
#pragma unroll
            for (int j = 0; j < 4; j++) on_bomb |= (bp == posb[j]) << j;
            on_bomb |= (pb_dir(bw) != 0) << 4;
        }
        on_bomb = a.gor(on_bomb);
        POM_STAMP(L, POM_PH_AGENT_PREP);
        POM_CUT(L, 30);
        /* agent loop, step.cpp:35-185 */
        int agents_done = 0;
        if (A::G == 4) {
            /* Quad path: lane m handles agent m.  The reference's loop visits the agents chain by chain (root, the agent that
             * waits for the root's cell, the one that waits for his, ...; step.cpp:36-61).  An agent reads and writes only his
             * own cell, his destination (shared destinations block both, step_utility.cpp:264-277, whichever comes first; a
             * shared flame kills both), his own registers and his own queue slot, and the only cell two agents of one tick both
             * care about is a dependant's destination = his predecessor's cell.  So the loop is run in ROUNDS: round d takes, in
             * parallel, every agent at depth d of his chain — all of round d-1 is written before round d reads.  Usually nobody
             * waits for anybody (nroots == 4): one round.  Queue order of planters = the reference's visiting order (`rank`).
             * Left to the literal loop below: two live agents on one cell (their writes would collide), a dependency cycle
             * (ouroboros) and lost agents (somebody the chains do not reach). */
            int par = !ouroboros && !clash;
            uint32_t rankp = 0x3210u, depthp = 0u; /* nibble per agent: position in the reference's visiting order, depth in his chain */
            int rounds = 1;
            if (par && nroots != 4) {
                /* lane ri walks the chain of root ri (dep[i] = who waits for agent i's cell); the chains' lengths, exchanged in
                 * the quad, give every chain its first position in the visiting order */
                const int ri = a.sub();
                int i = ri < nroots ? (int)((roots >> (4 * ri)) & 0xF) : 0xF;
                uint32_t dp = 0u, memb = 0u; /* my chain: depth per member, 0xF in the nibble of every member */
                int len = 0;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int on = i != 0xF;
                    const int sh = 4 * (i & 3);
                    dp |= on ? (uint32_t)d << sh : 0u;
                    memb |= on ? 0xFu << sh : 0u;
                    len += on;
                    i = on ? (int)((dep >> sh) & 0xF) : 0xF;
                }
                const int l0 = a.template gbcast<0>(len), l1 = a.template gbcast<1>(len), l2 = a.template gbcast<2>(len),
                          l3 = a.template gbcast<3>(len);
                const int b_lo = (ri & 1) ? l0 : 0, b_hi = (ri & 1) ? l0 + l1 + l2 : l0 + l1;
                const int base = (ri & 2) ? b_hi : b_lo; /* the lengths of the chains before mine */
                rankp = (uint32_t)a.gor((int)((dp + (uint32_t)base * 0x1111u) & memb)); /* depth <= 3, base <= 3: no carry between nibbles */
                depthp = (uint32_t)a.gor((int)dp);
                par = l0 + l1 + l2 + l3 == 4; /* everybody is reached: nobody lost */
                const int m01 = l0 > l1 ? l0 : l1, m23 = l2 > l3 ? l2 : l3;
                rounds = m01 > m23 ? m01 : m23; /* the longest chain */
            }
            if (par) {
                agents_done = 1;
                const int m = a.sub();
                int av = sel4(m, L.a0), a1v = a.ag1(m);
                const int a1_before = a1v;
                const int mvm = (mvp >> (4 * m)) & 0xF;
                const int live = !ag_dead(av);
                const int myrank = (int)((rankp >> (4 * m)) & 0xF), mydepth = (int)((depthp >> (4 * m)) & 0xF);
                /* plants: PlantBombModifiedLife(x, y, m, 11), bboard.cpp:125-146 — independent of everybody's movement */
                /* (bitwise & on 0 / 1 values here and below, not &&: on lane-varying operands hipcc turns the short-circuit forms into
                 * nested exec-mask branches — a dozen scalar instructions and two jumps for three compares) */
                const int wants = live & (int)(mvm == POM_MOVE_BOMB) & (int)(ag_bombcount(av) < ag_max_bombs(a1v));
                const int w_all = a.gor(wants << m);
                int slot_off = 0; /* planters visited before me */
#pragma unroll
                for (int j = 0; j < 4; j++) slot_off += ((w_all >> j) & 1) & ((int)((rankp >> (4 * j)) & 0xF) < myrank);
                const int fits = wants & (int)(L.bCnt + slot_off < POM_Q);
                int ubm = (wants & !fits) ? POM_UB_QUEUE_OVERFLOW : 0;
                int planted_moving = 0; /* PlantBomb leaves the slot's old direction nibble in place: the new bomb may move */
                if (fits) {
                    const int slot = wrap20(L.bIdx + L.bCnt + slot_off);
                    int b = a.bomb(slot); /* stale bits of the slot survive: SURVEY Q1 */
                    planted_moving = pb_dir(b) != 0;
                    b = pb_set(b, 0xF00u, (uint32_t)m << 8);
                    b = pb_set(b, 0xFFu, (uint32_t)ag_x(av) + ((uint32_t)ag_y(av) << 4));
                    b = pb_set(b, 0xF000u, (uint32_t)ag_strength(a1v) << 12);
                    b = pb_set(b, 0xF0000u, (uint32_t)(POM_BOMB_LIFETIME + 1) << 16);
                    a.put_bomb(slot, b);
                    av = ag_bombcount_add(av, 1);
                }
                /* where the agent wants to go */
                const int walks = live & (int)(mvm != POM_MOVE_IDLE) & (int)(mvm != POM_MOVE_BOMB);
                const int x = ag_x(av), y = ag_y(av);
                const int dkey = (dstp >> (8 * m)) & 0xFF;
                const int ddx = (dkey & 0xF) - 1, ddy = (dkey >> 4) - 1;
                const int goes = walks & !oob(ddx, ddy);
                const int dc = ddy * POM_N + ddx, oc = y * POM_N + x;
                const int vacated = ((on_bomb >> m) & 1) ? POM_C_BOMB : POM_C_PASSAGE;
                int died_sum = 0;
                POM_NOUNROLL
                for (int r = 0; r < rounds; r++) {
                    const int act = goes & (int)(mydepth == r);
                    int item = 0, collide = 0, shows_me = 0;
                    if (act) { /* what is there now: everything the earlier rounds did has been written */
                        item = a.cell(dc);
                        /* my own cell, asked together with the destination (one round trip): nobody writes it in this round —
                         * whoever wants it waits for me and moves in a later one */
                        shows_me = a.cell(oc) == POM_C_AGENT + m;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            collide |= (j != m) & !((deadmask >> j) & 1) & (((dstp >> (8 * j)) & 0xFF) == (uint32_t)dkey);
                    }
                    /* Will any bomb move this tick?  One has a direction already, somebody is about to kick one, or a bomb
                     * planted just now inherited one.  Only asked when nobody waits for anybody (the shortcut below). */
                    int bombs_move = 1;
                    if (nroots == 4) {
                        const int kicks = act & !pc_is_flame(item) & !collide & (int)(item == POM_C_BOMB) & ag_kick(av);
                        bombs_move = ((on_bomb >> 4) & 1) | a.gor(kicks | planted_moving);
                    }
                    int died = 0;
                    if (act) {
                        if (pc_is_flame(item)) { /* step.cpp:84-99 */
                            died = 1;
                            av |= POM_AG_DEAD;
                            if (shows_me) a.put_cell(oc, vacated);
                        } else if (!collide) {
                            if (pc_is_powerup(item)) { /* ConsumePowerup, step_utility.cpp:247-262 */
                                if (item == POM_C_EXTRABOMB) a1v = (a1v & ~0xFFFF) | ((a1v + 1) & 0xFFFF);
                                else if (item == POM_C_INCRRANGE) a1v = ag_strength_inc(a1v);
                                else av |= POM_AG_KICK;
                                item = POM_C_PASSAGE;
                            }
                            if (item == POM_C_PASSAGE) { /* step.cpp:120-140 */
                                if (shows_me) a.put_cell(oc, vacated);
                                a.put_cell(dc, POM_C_AGENT + m);
                                av = ag_setpos(av, ddx, ddy);
                            } else if (item == POM_C_BOMB) { /* step.cpp:147-184 */
                                /* Stepping onto a resting bomb without kicking it while no bomb moves at all: the agent loop
                                 * would put him there, bomb loop A (step.cpp:195-227) would find the bomb blocked by him and
                                 * bounce him straight back (nobody can have entered the cell he left: no dependency edge; no
                                 * bomb can be heading for it: none moves), restoring both cells and his position.  The pair
                                 * is skipped — provided a queued bomb really sits there (a BOMB item without one bounces nobody)
                                 * and his own cell shows him, so that "restoring" it changes nothing. */
                                if (!bombs_move && shows_me && bomb_index_alone(ddx | (ddy << 4)) >= 0) {
                                } else {
                                    a.put_cell(oc, vacated);
                                    a.put_cell(dc, POM_C_AGENT + m);
                                    av = ag_setpos(av, ddx, ddy);
                                    if (ag_kick(av)) {
                                        const int bi = bomb_index_alone(ddx | (ddy << 4)); /* GetBomb; the lanes are on different cells */
                                        if (bi < 0) ubm |= POM_UB_NULL_BOMB; /* step.cpp:167 dereferences nullptr */
                                        else put_bomb_at(bi, pb_set(bomb_at(bi), 0xF00000u, (uint32_t)mvm << 20));
                                    }
                                }
                            }
                        }
                    }
                    died_sum += died;
                    if (r + 1 < rounds) deadmask |= a.gor(died << m); /* HasDPCollision skips the dead, step_utility.cpp:268 */
                }
                /* back to identical registers in all four lanes */
                L.a0[0] = a.template gbcast<0>(av); L.a0[1] = a.template gbcast<1>(av);
                L.a0[2] = a.template gbcast<2>(av); L.a0[3] = a.template gbcast<3>(av);
                if (a1v != a1_before) a.put_ag1(m, a1v); /* (a power-up picked up: rare) */
                L.alive -= a.gadd(died_sum);
                L.bCnt += a.gadd(fits);
                L.ub |= (uint32_t)a.gor(ubm);
            }
        }
        if (!agents_done) {
            int root_idx = 0;
            int i = ouroboros ? 0 : (int)(roots & 0xF);
            POM_NOUNROLL
            for (int n = 0; n < 4; n++) {
                if (i == 0xF) {
                    root_idx++;
                    const int nx = root_idx < 4 ? (int)((roots >> (4 * root_idx)) & 0xF) : 0xF;
                    if (nx == 0xF) { /* reference reads moves[-1] here: SURVEY Q-UB1 */
                        L.ub |= POM_UB_LOST_AGENT;
                        break;
                    }
                    i = nx;
                }
                const int m = (mvp >> (4 * i)) & 0xF;
                const int av = sel4(i, L.a0);
                const int next = (int)((dep >> (4 * i)) & 0xF);
                if (ag_dead(av) || m == POM_MOVE_IDLE) {
                    i = next;
                    continue;
                }
                const int x = ag_x(av), y = ag_y(av);
                if (m == POM_MOVE_BOMB) { /* PlantBombModifiedLife(x, y, i, 11), bboard.cpp:125-146 */
                    const int a1v = a.ag1(i);
                    const int bomb_count = ag_bombcount(av), max_bombs = ag_max_bombs(a1v);
                    if (bomb_count < max_bombs) {
                        if (L.bCnt >= POM_Q) {
                            L.ub |= POM_UB_QUEUE_OVERFLOW; /* step.cpp:191 would overrun bombDestinations[20] */
                        } else {
                            const int slot = wrap20(L.bIdx + L.bCnt);
                            int b = a.bomb(slot); /* stale bits of the slot survive: SURVEY Q1 */
                            b = pb_set(b, 0xF00u, (uint32_t)i << 8);
                            b = pb_set(b, 0xFFu, (uint32_t)x + ((uint32_t)y << 4));
                            b = pb_set(b, 0xF000u, (uint32_t)ag_strength(a1v) << 12);
                            b = pb_set(b, 0xF0000u, (uint32_t)(POM_BOMB_LIFETIME + 1) << 16);
                            a.set_bomb(slot, b);
                            put4(i, L.a0, ag_bombcount_add(av, 1));
                            L.bCnt++;
#pragma unroll
                            for (int j = 0; j < 4; j++) on_bomb |= (ag_pos(L.a0[j]) == ag_pos(av)) << j;
                        }
                    }
                    i = next;
                    continue;
                }
                const int dkey = (dstp >> (8 * i)) & 0xFF;
                const int ddx = (dkey & 0xF) - 1, ddy = (dkey >> 4) - 1;
                if (oob(ddx, ddy)) {
                    i = next;
                    continue;
                }
                const int dc = ddy * POM_N + ddx;
                int item = a.cell(dc);
                if (ouroboros && bomb_index(ddx | (ddy << 4)) >= 0) item = POM_C_BOMB; /* step.cpp:71-82 */

                if (pc_is_flame(item)) { /* step.cpp:84-99 */
                    kill(i);
                    deadmask |= 1 << i;
                    const int oc = y * POM_N + x;
                    if (a.cell(oc) == POM_C_AGENT + i)
                        a.set_cell(oc, ((on_bomb >> i) & 1) ? POM_C_BOMB : POM_C_PASSAGE);
                    i = next;
                    continue;
                }
                /* HasDPCollision, step_utility.cpp:264-277 */
                int collide = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    collide |= (j != i) & !((deadmask >> j) & 1) & (((dstp >> (8 * j)) & 0xFF) == (uint32_t)dkey);
                if (collide) {
                    i = next;
                    continue;
                }
                if (pc_is_powerup(item)) { /* ConsumePowerup, step_utility.cpp:247-262 */
                    if (item == POM_C_EXTRABOMB) {
                        const int a1v = a.ag1(i);
                        a.set_ag1(i, (a1v & ~0xFFFF) | ((a1v + 1) & 0xFFFF));
                    } else if (item == POM_C_INCRRANGE) {
                        a.set_ag1(i, ag_strength_inc(a.ag1(i)));
                    } else {
                        put4(i, L.a0, sel4(i, L.a0) | POM_AG_KICK);
                    }
                    item = POM_C_PASSAGE;
                }
                const int oc = y * POM_N + x;
                if (item == POM_C_PASSAGE || (ouroboros && pc_is_agent(item))) { /* step.cpp:120-140 */
                    if (a.cell(oc) == POM_C_AGENT + i)
                        a.set_cell(oc, ((on_bomb >> i) & 1) ? POM_C_BOMB : POM_C_PASSAGE);
                    a.set_cell(dc, POM_C_AGENT + i);
                    put4(i, L.a0, ag_setpos(sel4(i, L.a0), ddx, ddy));
                } else if (item == POM_C_BOMB) { /* step.cpp:147-184: kicker and non-kicker both step on */
                    a.set_cell(oc, ((on_bomb >> i) & 1) ? POM_C_BOMB : POM_C_PASSAGE);
                    a.set_cell(dc, POM_C_AGENT + i);
                    const int cur = sel4(i, L.a0);
                    put4(i, L.a0, ag_setpos(cur, ddx, ddy));
                    if (ag_kick(cur)) {
                        const int bi = bomb_index(ddx | (ddy << 4));
                        if (bi < 0) L.ub |= POM_UB_NULL_BOMB; /* step.cpp:167 dereferences nullptr */
                        else set_bomb_at(bi, pb_set(bomb_at(bi), 0xF00000u, (uint32_t)m << 20));
                    }
                }
                i = next;
            }
        }

        POM_STAMP(L, POM_PH_AGENT_LOOP);
        POM_CUT(L, 40);
        if (L.bCnt > 0) {
            /* ResetBombFlags + FillBombDestPos, step_utility.cpp:331-337,146-152.  The same pass notes whether any
             * bomb of this env is moving: if none is, loop B below collapses to one cell test per bomb.  (Two resting bombs
             * on one cell — SURVEY Q8 — do "collide" in HasBombCollision, but ResolveBombCollision only sets directions that are
             * IDLE already to IDLE, and the bombs of such a group that skip their cell test share their cell with one that
             * makes it: without a moving bomb shared cells change nothing.  Rounds 1 - 3 tracked them in a 121-bit occupancy
             * set: ~50 VALU per wavefront-tick for a distinction without a difference.) */
            int moving = 0;
            int ripe = 0; /* some bomb's own cell shows PASSAGE or a flame: the only cells loop B's resting case acts on */
            uint32_t cand = 0; /* queue offsets of the resting bombs under an agent that walked onto them this tick */
            /* TickBombs' timer decrement (step_utility.cpp:226-231) is folded into this pass: nothing between here and TickBombs
             * reads a timer, and loops A / B only replace the position and direction fields of a word, which commutes with
             * `- (1 << 16)` as long as that does not borrow.  A bomb whose timer is already 0 (out-of-order timers)
             * WOULD borrow into the fields above: it is left as it is, marked in the (just cleared, otherwise unused) moved
             * nibble, and decremented where the reference does it, after loop B (the cold pass below). */
            int late = 0;
            uint32_t movers = 0; /* queue offsets of the bombs with a direction */
            POM_NOUNROLL
            for (int k = a.sub(); k < L.bCnt; k += A::G) { /* split: each lane its own slots, combined below */
                int b = pb_set(bomb_at(k), 0xF000000u, 0);
                const int key = bomb_target_key(b);
                if (pb_time(b) == 0) {
                    late = 1;
                    b |= 1 << 24;
                } else {
                    b = (int)((uint32_t)b - (1u << 16));
                }
                put_bomb_at(k, b);
                if (k == 0) top = b;
                a.put_bdest(k, key);
                moving |= pb_dir(b) != 0;
                movers |= (uint32_t)(pb_dir(b) != 0) << k;
                const int idx = pb_y(b) * POM_N + pb_x(b);
                /* loop A, looked at in the same pass (only meaningful if nothing moves, see below): a resting bomb under an
                 * agent that walked onto it this tick means a bounce-back */
                if (idx < POM_CELLS) {
                    const int e = a.cell(idx);
                    ripe |= (e == POM_C_PASSAGE) | pc_is_flame(e);
                    if (pc_blocks_bomb(e)) {
                        const int ag = get_agent(pb_x(b), pb_y(b));
                        const int ag3 = ag & 3; /* (ag == -1: agent 3's fields are read and the answer masked) */
                        const int m2 = (mvp >> (4 * ag3)) & 0xF;
                        const int was = (oldp >> (8 * ag3)) & 0xFF;
                        cand |= (uint32_t)((int)(ag > -1) & (int)(m2 != POM_MOVE_IDLE) & (int)(m2 != POM_MOVE_BOMB) & (int)(pb_pos(b) != was)) << k;
                    }
                }
            }
            if (A::G > 1) {
                /* one quad reduction for all the flags: cand (offsets 0..19) | ripe | late | moving */
                const uint32_t fl = (uint32_t)a.gor((int)(cand | ((uint32_t)ripe << 20) | ((uint32_t)late << 21) | ((uint32_t)moving << 23)));
                cand = fl & 0xFFFFFu;
                ripe = (int)((fl >> 20) & 1u);
                late = (int)((fl >> 21) & 1u);
                moving = (int)((fl >> 23) & 1u);
                movers = (uint32_t)a.gor((int)movers);
            }
            folded_ = 1;
            int touched = cand != 0; /* did anything after the pass get to write the queue?  (then its head is read again) */
            /* bomb loop A, step.cpp:195-227.  A resting bomb's "target" is its own cell: it is blocked iff an agent item (or,
             * never in practice, a static item) shows there; setting an idle bomb idle changes nothing, so while no bomb moves
             * the loop only matters for the bombs noted above, whose agent has to be bounced back — and only those are
             * visited, in queue order.  (A bounce only ever puts agents back where they stood before the tick: an agent it
             * brings onto another resting bomb has position == old position there and is not bounced, so no bomb outside the
             * noted set can come to matter during the loop.)  With a moving bomb in the queue the whole loop runs. */
            POM_STAMP(L, POM_PH_BOMB_PASS);
            POM_CUT(L, 50);
            int next = 0; /* loop A is done for the offsets below `next` */
            {
                /* ... and with moving bombs in the queue the same holds for the RESTING ones: besides the noted set only the bombs
                 * that have a direction can be blocked to any effect (round 4; until then a moving bomb anywhere in the queue
                 * sent the loop over all of it) */
                uint32_t todo = cand | movers;
                irregular_ = 0;
                POM_NOUNROLL
                while (todo && !irregular_) {
                    const int k = __builtin_ctz(todo);
                    todo &= todo - 1u;
                    loop_a_bomb(mvp, oldp, k, moving);
                    next = k + 1;
                }
                /* the argument above holds as long as every bounce was a plain step back; after any other (two agents on one
                 * cell: states past a lost-agent tick) the rest of the queue is walked bomb by bomb */
                if (!irregular_) next = L.bCnt;
            }
            POM_NOUNROLL
            for (int k = next; k < L.bCnt; k++) loop_a_bomb(mvp, oldp, k, moving);
            POM_STAMP(L, POM_PH_BOMB_A);
            POM_CUT(L, 60);
            /* bomb loop B, step.cpp:230-278 */
            touched |= moving | ripe;
            /* While no bomb moves and no two share a cell, HasBombCollision is false for every bomb and each one "moves" onto its
             * own cell (step.cpp:243-272): PASSAGE there becomes BOMB, a flame detonates the bomb, anything else stays (a static
             * item makes the reference set the already idle bomb idle).  If the pass above saw neither passage nor flame under a
             * bomb, loop B has nothing to do (loop A in between only writes agent and BOMB items).  Otherwise look first: without
             * a detonation the writes are independent and done in parallel (two bombs on one cell write the same BOMB item);
             * with one — as with moving bombs — the queue is walked in order by the literal loop. */
            int general = moving;
            if (!general && ripe) {
                int in_flame = 0;
                POM_NOUNROLL
                for (int k = a.sub(); k < L.bCnt; k += A::G) {
                    const int b = bomb_at(k);
                    in_flame |= pc_is_flame(a.cell(pb_y(b) * POM_N + pb_x(b)));
                }
                if (!a.gor(in_flame)) {
                    POM_NOUNROLL
                    for (int k = a.sub(); k < L.bCnt; k += A::G) {
                        const int b = bomb_at(k);
                        const int c = pb_y(b) * POM_N + pb_x(b);
                        if (a.cell(c) == POM_C_PASSAGE) a.put_cell(c, POM_C_BOMB);
                    }
                } else {
                    general = 1;
                }
            }
            if (general) {
                /* the queue in order (step.cpp:230-278) — but only the bombs loop B can do anything for or with (loop_b_todo); once
                 * something happens that the selection did not foresee, every bomb from there on */
                uint32_t todo = loop_b_todo();
                POM_NOUNROLL
                while (todo) {
                    const int k = __builtin_ctz(todo);
                    todo &= todo - 1u;
                    if (k >= L.bCnt) break;
                    unforeseen_ = 0;
                    loop_b_bomb(mvp, k);
                    if (unforeseen_) todo = k + 1 < POM_Q ? ((1u << POM_Q) - 1u) & ~((2u << k) - 1u) : 0u; /* offsets k+1 .. 19 (the count ends the loop) */
                }
            }
            POM_STAMP(L, POM_PH_BOMB_B);
            POM_CUT(L, 70);
            /* TickBombs, step_utility.cpp:224-245, first half: the timers were decremented in the pass above, except the marked
             * ones (cold: only states with out-of-order timers have them) */
            if (late) {
                POM_NOUNROLL
                for (int k = a.sub(); k < L.bCnt; k += A::G) { /* split */
                    const int b = bomb_at(k);
                    if (b & (1 << 24)) put_bomb_at(k, (int)(((uint32_t)b & ~(1u << 24)) - (1u << 16)));
                }
                touched = 1;
            }
            folded_ = 0;
            /* the head of the queue: what lane 0 wrote in the pass, unless something has touched the queue since */
            top = a.template gbcast<0>(top);
            if (touched && L.bCnt > 0) top = bomb_at(0);
            n = L.bCnt;
        }
    }
};

/* ---- record <-> lane registers ----------------------------------------- */
POM_HD void pom_lane_load(PomLane& L, const uint32_t* ag /* the record's 8 agent dwords */, uint32_t& status)
{
    /* what is not an agent's travels in the top bytes of the agent words (pom_packed.h) */
    L.alive = (int)ag[0] >> 24; /* (signed) */
    L.bIdx = (int)(ag[2] >> 24);
    L.bCnt = (int)(ag[4] >> 24);
    L.fIdx = (int)(ag[6] >> 24);
    L.fCnt = (int)(ag[1] >> 24);
    status = ag[3] >> 24;
    L.ub = (ag[5] >> 24) | ((ag[7] >> 24) << 8);
    for (int i = 0; i < 4; i++) L.a0[i] = (int)ag[2 * i];
}
/* the record's agent dword k (0..7) as it is stored: the agent word with this tick's counts, status and flags in its top byte.  Even k:
 * the lane's a0 words; odd k: `stored` is the second word as it stands in the store (the tick keeps those there) */
POM_HD uint32_t pom_lane_agent_word(const PomLane& L, uint32_t status, int k, uint32_t stored = 0)
{
    const uint32_t body = ((k & 1) ? stored : (uint32_t)L.a0[k >> 1]) & 0x00FFFFFFu;
    const uint32_t top = k == 0 ? (uint32_t)L.alive : k == 2 ? (uint32_t)L.bIdx : k == 4 ? (uint32_t)L.bCnt : k == 6 ? (uint32_t)L.fIdx :
                         k == 1 ? (uint32_t)L.fCnt : k == 3 ? status : k == 5 ? L.ub : L.ub >> 8;
    return body | (top << 24);
}

/* Environment::Step's bookkeeping after bboard::Step (environment.cpp:150-168); returns the new status byte */
POM_HD uint32_t pom_env_epilogue(const PomLane& L, int time_step_after, int max_steps, uint32_t status)
{
    if (L.alive == 1) {
        int w = 0;
        for (int i = 0; i < 4; i++)
            if (!ag_dead(L.a0[i])) w = i; /* last alive index wins, as the reference's loop leaves it */
        status |= POM_ST_DONE | ((uint32_t)(w + 1) << POM_ST_WINNER_SHIFT);
    }
    if (L.alive == 0) status |= POM_ST_DONE | POM_ST_DRAW;
    if (max_steps > 0 && time_step_after >= max_steps) status |= POM_ST_DONE | POM_ST_TIMEOUT;
    return status;
}

#endif /* POM_STEP_BODY_H_ */
