/*
 * pom_policy_body.h — the reference's heuristic policy `agents::SimpleAgent` for ONE agent, written against an abstract
 * per-lane store so the identical source runs on gfx950 (pom_policy_kernel in pom_kernels.hip, lane = agent) and, in
 * tests/emul only, on the host for fuzzing against the oracle.  SURVEY.md §8 row f1; config 3 of BASELINE.json.
 *
 * Semantics = SimpleAgent::act (/root/reference/src/agents/simple_agent.cpp:51-137) with the strategy helpers of
 * /root/reference/src/bboard/strategy.cpp and include/strategy.hpp, quirks included (file:line per function).  Two things are
 * made explicit that the reference leaves to chance: the one random draw of an act() is an input (uniform 0..4; the reference
 * draws from a random_device-seeded mt19937_64), and the agent's memory starts zeroed (the reference reads uninitialised
 * queue slots).
 *
 * The reachability map (FillRMap, a FIFO breadth-first search from the agent's cell) is not searched cell by cell.  All the
 * policy ever asks of it is (a) "is this cell reachable" and (b) MoveTowardsPosition: follow the predecessors from a target
 * back to the source and report on which side of the source the path starts.  A FIFO search with a fixed neighbour order
 * (DOWN, UP, RIGHT, LEFT: strategy.cpp:83-90) gives every cell the lexicographically smallest of its shortest paths, so the
 * side the path starts on is the smallest label among the cell's neighbours of the previous level.  That is a flood fill: four
 * 121-bit cell sets in registers (one per first step), each level dilated by shifts and masks and claimed in priority order.
 * No queue, no per-cell map, no LDS traffic; ~300 VALU per level instead of ~100 per CELL.  It is only run for agents whose
 * decision reads it, at one program point.
 *
 * Agent memory, 2 dwords: m0 = recentPositions.queue[0..3], a byte each: x:4 | y:4 two's-complement nibbles (-1 .. 11);
 *                         m1 = recentPositions.index:2 | count:3 @2 | moveQueue.queue[0..3] 3 bits each @5 | moveQueue.count:3 @17
 * (moveQueue.index is always 0: the queue is never popped).  All-zero = a fresh agent.
 *
 * Per env and tick two things are prepared ONCE, by the four lanes of the env together (pom_policy_prepare), instead of being
 * recomputed per query: the danger map — IsInDanger(x, y) for every cell: each bomb's cross rasterised with an LDS atomic min
 * of its timer — and two 121-bit sets, "walkable" and "agent", which the BFS tests in registers instead of reading cells.
 *
 * Store interface P:  int cell(int c)            16-bit board code (pom_packed.h), c = y*11+x
 *                     int bomb(int slot)         raw bomb word of physical slot
 *                     int danger(int c) / void danger_init(int c) / void danger_min(int c, int t)     per-env danger map
 *                     uint32_t setw(int k) / void set_or(int k, uint32_t bits) / void set_zero(int k)   per-env bit sets, k = 0..7
 *                     int member()               which of the env's 4 lanes this is
 */
#ifndef POM_POLICY_BODY_H_
#define POM_POLICY_BODY_H_

#include "pom_step_body.h"

struct PomPolicyEnv { /* what the policy reads of the env besides board and bombs */
    int a0[4], a1[4]; /* agent words of the packed record */
    int bIdx, bCnt;
};

enum { POM_DANGER_NONE = 99 };

/* Shared preparation, executed by all four lanes of an env (member m takes cells / bombs m, m+4, ...).  Two phases: every
 * lane must have finished clearing before any lane accumulates (on the device the wavefront runs them back to back in
 * lock-step; a sequential host emulation runs phase 0 for all four members, then phase 1). */
template <class P>
POM_HD void pom_policy_prepare_clear(P& p)
{
    const int m = p.member();
    if (m == 0)
        for (int k = 0; k < 8; k++) p.set_zero(k);
    POM_NOUNROLL
    for (int c = m; c < POM_CELLS; c += 4) p.danger_init(c);
}
template <class P>
POM_HD void pom_policy_prepare_fill(P& p, const PomPolicyEnv& E)
{
    const int m = p.member();
    /* walkable (IS_WALKABLE, bboard.hpp:81-84) -> words 0..3, agent cells (item >= AGENT0) -> words 4..7 */
    uint32_t w[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    POM_NOUNROLL
    for (int c = m; c < POM_CELLS; c += 4) {
        const int e = p.cell(c);
        const uint32_t bit = 1u << (c & 31);
        const int k = c >> 5;
        const uint32_t wb = pc_is_walkable(e) ? bit : 0u, gb = pc_is_agent(e) ? bit : 0u;
        w[0] |= k == 0 ? wb : 0u; w[1] |= k == 1 ? wb : 0u; w[2] |= k == 2 ? wb : 0u; w[3] |= k == 3 ? wb : 0u;
        g[0] |= k == 0 ? gb : 0u; g[1] |= k == 1 ? gb : 0u; g[2] |= k == 2 ? gb : 0u; g[3] |= k == 3 ? gb : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p.set_or(k, w[k]);
        p.set_or(4 + k, g[k]);
    }
    /* IsInDanger for every cell at once: min BMB_TIME over the bombs whose cross (IsInBombRange, strategy.hpp:163-169: the
     * +-strength row and column segments through the bomb, walls ignored) covers the cell */
    POM_NOUNROLL
    for (int i = m; i < E.bCnt; i += 4) {
        const int b = p.bomb(wrap20(E.bIdx + i));
        const int bx = pb_x(b), by = pb_y(b), s = pb_strength(b), t = pb_time(b);
        const int x0 = bx - s < 0 ? 0 : bx - s, x1 = bx + s > POM_N - 1 ? POM_N - 1 : bx + s;
        const int y0 = by - s < 0 ? 0 : by - s, y1 = by + s > POM_N - 1 ? POM_N - 1 : by + s;
        if (by < POM_N) {
            POM_NOUNROLL
            for (int x = x0; x <= x1; x++) p.danger_min(by * POM_N + x, t);
        }
        if (bx < POM_N) {
            POM_NOUNROLL
            for (int y = y0; y <= y1; y++) p.danger_min(y * POM_N + bx, t);
        }
    }
}

/* a set of board cells: bit c = y*11+x of a 121-bit number in four words */
struct PomCells {
    uint32_t w[4];
    POM_HD static PomCells zero() { return PomCells{{0u, 0u, 0u, 0u}}; }
    POM_HD int any() const { return (w[0] | w[1] | w[2] | w[3]) != 0; }
    POM_HD int has(int c) const
    {
        const int k = c >> 5;
        const uint32_t v = k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3];
        return (int)((v >> (c & 31)) & 1u);
    }
    POM_HD void add(int c)
    {
        const int k = c >> 5;
        const uint32_t b = 1u << (c & 31);
        w[0] |= k == 0 ? b : 0u; w[1] |= k == 1 ? b : 0u; w[2] |= k == 2 ? b : 0u; w[3] |= k == 3 ? b : 0u;
    }
    POM_HD void remove(int c)
    {
        const int k = c >> 5;
        const uint32_t b = ~(1u << (c & 31));
        w[0] &= k == 0 ? b : ~0u; w[1] &= k == 1 ? b : ~0u; w[2] &= k == 2 ? b : ~0u; w[3] &= k == 3 ? b : ~0u;
    }
    template <int S> POM_HD PomCells up_by() const /* cell c -> c + S; cells pushed past 120 vanish */
    {
        PomCells r;
        r.w[3] = ((w[3] << S) | (w[2] >> (32 - S))) & 0x01FFFFFFu; /* 121 = 96 + 25 */
        r.w[2] = (w[2] << S) | (w[1] >> (32 - S));
        r.w[1] = (w[1] << S) | (w[0] >> (32 - S));
        r.w[0] = w[0] << S;
        return r;
    }
    template <int S> POM_HD PomCells down_by() const /* cell c -> c - S */
    {
        PomCells r;
        r.w[0] = (w[0] >> S) | (w[1] << (32 - S));
        r.w[1] = (w[1] >> S) | (w[2] << (32 - S));
        r.w[2] = (w[2] >> S) | (w[3] << (32 - S));
        r.w[3] = w[3] >> S;
        return r;
    }
    POM_HD PomCells operator|(const PomCells& o) const { return PomCells{{w[0] | o.w[0], w[1] | o.w[1], w[2] | o.w[2], w[3] | o.w[3]}}; }
    POM_HD PomCells operator&(const PomCells& o) const { return PomCells{{w[0] & o.w[0], w[1] & o.w[1], w[2] & o.w[2], w[3] & o.w[3]}}; }
    POM_HD PomCells minus(const PomCells& o) const { return PomCells{{w[0] & ~o.w[0], w[1] & ~o.w[1], w[2] & ~o.w[2], w[3] & ~o.w[3]}}; }
    /* the four neighbours of every member: y+1, y-1, x+1 (not across the right edge), x-1 (not across the left edge) */
    POM_HD PomCells neighbours() const
    {
        /* column x = 0: cells 0, 11, 22, ...; column x = 10: cells 10, 21, ... */
        const PomCells col0{{0x00400801u, 0x00801002u, 0x01002004u, 0x00004008u}};
        const PomCells col10{{0x00200400u, 0x00400801u, 0x00801002u, 0x01002004u}};
        return up_by<POM_N>() | down_by<POM_N>() | up_by<1>().minus(col0) | down_by<1>().minus(col10);
    }
};

template <class P>
struct PomSimplePolicy {
    P& p;
    const PomPolicyEnv& E;
    int id, sx, sy; /* me, and where I stand (the search's source) */
    uint32_t m0, m1;
    PomCells first[4]; /* cells whose path from me starts DOWN / UP / RIGHT / LEFT; their union = the reachable cells */
    POM_HD PomSimplePolicy(P& p_, const PomPolicyEnv& e_, int id_, uint32_t m0_, uint32_t m1_)
        : p(p_), E(e_), id(id_), sx(0), sy(0), m0(m0_), m1(m1_)
    {
        first[0] = first[1] = first[2] = first[3] = PomCells::zero();
        const int av = sel4(id, E.a0);
        sx = ag_x(av);
        sy = ag_y(av);
    }

    /* ---- memory fields ---- */
    POM_HD static int nib2i(int v) { return v == 15 ? -1 : v; }
    POM_HD int rp_index() const { return (int)(m1 & 3); }
    POM_HD int rp_count() const { return (int)((m1 >> 2) & 7); }
    POM_HD int rp_key(int off) const { return (int)((m0 >> (8 * ((rp_index() + off) & 3))) & 0xFF); } /* x | y<<4 nibbles */
    POM_HD int mq_count() const { return (int)((m1 >> 17) & 7); }
    POM_HD int mq_at(int off) const { return (int)((m1 >> (5 + 3 * (off & 3))) & 7); }
    POM_HD void mq_set(int slot, int mv) { m1 = (m1 & ~(7u << (5 + 3 * slot))) | ((uint32_t)mv << (5 + 3 * slot)); }
    POM_HD void mq_set_count(int c) { m1 = (m1 & ~(7u << 17)) | ((uint32_t)c << 17); }
    POM_HD void mq_add(int mv) /* AddElem */
    {
        mq_set(mq_count() & 3, mv);
        mq_set_count(mq_count() + 1);
    }
    POM_HD static int pos_key(int x, int y) { return (x & 0xF) | ((y & 0xF) << 4); }

    /* ---- strategy helpers ---- */
    POM_HD int in_danger(int x, int y) const /* IsInDanger, strategy.cpp:229-249: one read of the prepared map; (x, y) on the board */
    {
        const int v = p.danger(y * POM_N + x);
        return v == POM_DANGER_NONE ? 0 : v;
    }
    POM_HD static int safe(int danger, int min) { return danger == 0 || danger >= min; } /* _safe_condition, strategy.cpp:199-202 */
    POM_HD int walkable_at(int x, int y) const { return !oob(x, y) && pc_is_walkable(p.cell(y * POM_N + x)); } /* _CheckPos */

    /* FillRMap, strategy.cpp:59-93, as a flood fill (see the file comment).  A cell can be entered if it is walkable or holds an
     * agent (strategy.cpp:43-44); agents are reached but not passed (:50-53); the source is never re-entered (:83-90). */
    POM_HD void build_map()
    {
        PomCells agents, open;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            agents.w[k] = p.setw(4 + k);
            open.w[k] = p.setw(k) | agents.w[k];
        }
        const int src = sy * POM_N + sx;
        open.remove(src);
        PomCells front[4];
        /* level 1: the source's own neighbours, claimed in the order DOWN, UP, RIGHT, LEFT */
        const int nx[4] = {sx, sx, sx + 1, sx - 1}, ny[4] = {sy + 1, sy - 1, sy, sy};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            first[k] = PomCells::zero();
            const int n = ny[k] * POM_N + nx[k];
            if (!oob(nx[k], ny[k]) && open.has(n)) {
                first[k].add(n);
                open.remove(n);
            }
            front[k] = first[k].minus(agents);
        }
        POM_NOUNROLL
        for (int level = 0; level < POM_CELLS; level++) {
            int grew = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { /* priority = label order: the smallest first step wins a cell */
                const PomCells got = front[k].neighbours() & open;
                open = open.minus(got);
                first[k] = first[k] | got;
                front[k] = got.minus(agents);
                grew |= got.any();
            }
            if (!grew) break;
        }
    }
    POM_HD int reached(int c) const { return first[0].has(c) | first[1].has(c) | first[2].has(c) | first[3].has(c); }
    POM_HD int move_towards(int tx, int ty) const /* MoveTowardsPosition, strategy.cpp:99-121 */
    {
        const int c = ty * POM_N + tx;
        if (first[0].has(c)) return POM_MOVE_DOWN;
        if (first[1].has(c)) return POM_MOVE_UP;
        if (first[2].has(c)) return POM_MOVE_RIGHT;
        if (first[3].has(c)) return POM_MOVE_LEFT;
        /* unreached target: its map entry is 0, i.e. "distance 0, predecessor cell 0".  The reference takes cell 0 for the
         * source if the agent stands there and answers by comparing coordinates (:107-113); otherwise IDLE (:115-118) */
        if (sx == 0 && sy == 0) {
            if (tx > 0) return POM_MOVE_RIGHT;
            if (ty > 0) return POM_MOVE_DOWN;
        }
        return POM_MOVE_IDLE;
    }
    POM_HD int move_towards_safe_place(int radius) /* strategy.cpp:123-140: the window's upper bounds are `radius` (sic) */
    {
        const int y0 = sy - radius < 0 ? 0 : sy - radius, y1 = radius < POM_N ? radius : POM_N;
        const int x0 = sx - radius < 0 ? 0 : sx - radius, x1 = radius < POM_N ? radius : POM_N;
        POM_NOUNROLL
        for (int y = y0; y < y1; y++) {
            POM_NOUNROLL
            for (int x = x0; x < x1; x++) {
                const int dx_ = x - sx, dy_ = y - sy;
                if ((dx_ < 0 ? -dx_ : dx_) + (dy_ < 0 ? -dy_ : dy_) > radius) continue;
                if (reached(y * POM_N + x) && safe(in_danger(x, y), 2)) return move_towards(x, y);
            }
        }
        return POM_MOVE_IDLE;
    }
    POM_HD int manhattan_to(int j) const
    {
        const int dx_ = ag_x(E.a0[j]) - sx, dy_ = ag_y(E.a0[j]) - sy;
        return (dx_ < 0 ? -dx_ : dx_) + (dy_ < 0 ? -dy_ : dy_);
    }
    POM_HD int move_towards_enemy(int radius) /* strategy.cpp:165-192 */
    {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int av = E.a0[j];
            if ((ag_x(av) == sx && ag_y(av) == sy) || ag_dead(av)) continue;
            if (manhattan_to(j) > radius) continue;
            return move_towards(ag_x(av), ag_y(av));
        }
        return POM_MOVE_IDLE;
    }
    POM_HD int adjacent_enemy(int distance) const /* IsAdjacentEnemy, strategy.cpp:297-313 */
    {
        int r = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) r |= (j != id) & !ag_dead(E.a0[j]) & (manhattan_to(j) <= distance);
        return r;
    }
    POM_HD int adjacent_wood() const /* IsAdjacentItem(state, id, 1, Item::WOOD), strategy.cpp:315-338 */
    {
        int r = 0;
        if (sy > 0) r |= pc_is_wood(p.cell((sy - 1) * POM_N + sx));
        if (sx > 0) r |= pc_is_wood(p.cell(sy * POM_N + sx - 1));
        r |= pc_is_wood(p.cell(sy * POM_N + sx));
        if (sx < POM_N - 1) r |= pc_is_wood(p.cell(sy * POM_N + sx + 1));
        if (sy < POM_N - 1) r |= pc_is_wood(p.cell((sy + 1) * POM_N + sx));
        return r;
    }
    POM_HD void safe_directions() /* strategy.cpp:203-226 */
    {
        if (walkable_at(sx + 1, sy) && safe(in_danger(sx + 1, sy), 2)) mq_add(POM_MOVE_RIGHT);
        if (walkable_at(sx - 1, sy) && safe(in_danger(sx - 1, sy), 2)) mq_add(POM_MOVE_LEFT);
        if (walkable_at(sx, sy + 1) && safe(in_danger(sx, sy + 1), 2)) mq_add(POM_MOVE_DOWN);
        if (walkable_at(sx, sy - 1) && safe(in_danger(sx, sy - 1), 2)) mq_add(POM_MOVE_UP);
    }
    POM_HD void sort_directions() /* SortDirections, strategy.hpp:130-152, with FixedQueue::RemoveAt / AddElem on raw slots */
    {
        const int moves = mq_count();
        int removes = 0;
        POM_NOUNROLL
        for (int i = 0; i < moves && removes < 4; i++) {
            const int mv = mq_at(i);
            const int key = pos_key(sx + mv_dx(mv), sy + mv_dy(mv));
            int hit = 0;
            for (int j = 0; j < 4; j++) hit |= (j < rp_count()) & (rp_key(j) == key);
            if (hit) {
                POM_NOUNROLL
                for (int k = i + 1; k < mq_count(); k++) mq_set((k - 1) & 3, mq_at(k)); /* RemoveAt(i) */
                mq_set_count(mq_count() - 1);
                mq_add(mq_at(i)); /* sic: re-adds what now sits at i, not what was removed */
                i--;
                removes++;
            }
        }
    }
    POM_HD int has_rp_loop() const /* _HasRPLoop, simple_agent.cpp:24-35 */
    {
        int ok = 1;
        for (int i = 0; i < 2; i++)
            if (i < rp_count() / 2) ok &= rp_key(i) == rp_key(i + 2);
        return ok;
    }
    POM_HD int one_safe_step(int draw) /* the common tail of _Decide and _MoveSafeOneSpace, simple_agent.cpp:37-48,105-121 */
    {
        mq_set_count(0);
        safe_directions();
        sort_directions();
        if (mq_count() == 0) return POM_MOVE_IDLE;
        return mq_at(draw % 2);
    }
    POM_HD int decide(int draw) /* _Decide, simple_agent.cpp:51-122 */
    {
        const int av = sel4(id, E.a0), a1v = sel4(id, E.a1);
        const int danger = in_danger(sx, sy);
        const int can_bomb = pom_sext8((uint32_t)av >> 16) < pom_sext16((uint32_t)a1v);
        const int adj1 = adjacent_enemy(1), near = adjacent_enemy(7), looping = has_rp_loop();
        /* The reachability map is read by MoveTowardsSafePlace (danger) and MoveTowardsEnemy (an enemy within 7, nothing more
         * urgent); it has no other effect, so it is built only for those agents — and at ONE program point: built lazily
         * inside the two branches, a wavefront would run the whole search twice under different lane masks. */
        if (danger > 0 || (can_bomb && !adj1 && near && !looping)) build_map();
        if (danger > 0) {
            const int mv = move_towards_safe_place(danger);
            const int px = sx + mv_dx(mv), py = sy + mv_dy(mv);
            if (walkable_at(px, py) && safe(in_danger(px, py), 2)) return mv;
            return one_safe_step(draw);
        }
        if (can_bomb) {
            if (adj1) return POM_MOVE_BOMB;
            if (near && looping) return draw % 4;
            if (near) {
                const int mv = move_towards_enemy(7);
                const int px = sx + mv_dx(mv), py = sy + mv_dy(mv);
                if (walkable_at(px, py) && safe(in_danger(px, py), 5)) return mv;
            }
            if (adjacent_wood()) return POM_MOVE_BOMB;
        }
        return one_safe_step(draw);
    }
    POM_HD int act(int draw) /* SimpleAgent::act, simple_agent.cpp:123-137 */
    {
        const int mv = decide(draw);
        const int key = pos_key(sx + mv_dx(mv), sy + mv_dy(mv));
        int idx = rp_index(), cnt = rp_count();
        if (cnt == 4) { /* RemainingCapacity() == 0: PopElem */
            idx = (idx + 1) & 3;
            cnt--;
        }
        const int slot = (idx + cnt) & 3;
        m0 = (m0 & ~(0xFFu << (8 * slot))) | ((uint32_t)key << (8 * slot));
        cnt++;
        m1 = (m1 & ~31u) | (uint32_t)idx | ((uint32_t)cnt << 2);
        return mv;
    }
};

/* agent memory <-> the 16-int form of oracle/pom_policy_oracle.h (tests, pom_batch_policy_memory) */
POM_HD void pom_policy_mem_unpack(uint32_t m0, uint32_t m1, int32_t out[16])
{
    for (int i = 0; i < 4; i++) {
        const int k = (m0 >> (8 * i)) & 0xFF;
        out[2 * i] = (k & 0xF) == 15 ? -1 : (k & 0xF);
        out[2 * i + 1] = (k >> 4) == 15 ? -1 : (k >> 4);
    }
    out[8] = (int32_t)(m1 & 3);
    out[9] = (int32_t)((m1 >> 2) & 7);
    for (int i = 0; i < 4; i++) out[10 + i] = (int32_t)((m1 >> (5 + 3 * i)) & 7);
    out[14] = 0;
    out[15] = (int32_t)((m1 >> 17) & 7);
}

#endif /* POM_POLICY_BODY_H_ */
