/*
 * pom_policy_body.h — the reference's heuristic policy `agents::SimpleAgent` for ONE agent, written against an abstract
 * per-lane store so the identical source runs on gfx950 (pom_policy_kernel in pom_kernels.hip, lane = agent) and, in
 * tests/emul only, on the host for fuzzing against the oracle.  SURVEY.md §8 row f1; config 3 of BASELINE.json.
 *
 * Semantics = SimpleAgent::act (/root/reference/src/agents/simple_agent.cpp:51-137) with the strategy helpers of
 * /root/reference/src/bboard/strategy.cpp and include/strategy.hpp, quirks included (file:line per function).  Two things are
 * made explicit that the reference leaves to chance: the one random draw of an act() is an input (uniform 0..4; the reference
 * draws from a random_device-seeded mt19937_64), and the agent's memory starts zeroed (the reference reads uninitialised
 * queue slots).  The reachability map is only built when a branch actually reads it (it has no other effect), which is what
 * makes the policy affordable 64 agents wide: most agents are neither in danger nor near an enemy.
 *
 * Agent memory, 2 dwords: m0 = recentPositions.queue[0..3], a byte each: x:4 | y:4 two's-complement nibbles (-1 .. 11);
 *                         m1 = recentPositions.index:2 | count:3 @2 | moveQueue.queue[0..3] 3 bits each @5 | moveQueue.count:3 @17
 * (moveQueue.index is always 0: the queue is never popped).  All-zero = a fresh agent.
 *
 * Store interface P:  int cell(int c)            16-bit board code (pom_packed.h), c = y*11+x
 *                     int bomb(int slot)         raw bomb word of physical slot
 *                     int rm(int c) / void set_rm(int c, int v)     reachability map entry: distance:8 | predecessor cell:8
 *                     void clear_rm()
 *                     int qe(int i) / void set_qe(int i, int c)     BFS queue of cell ids
 */
#ifndef POM_POLICY_BODY_H_
#define POM_POLICY_BODY_H_

#include "pom_step_body.h"

struct PomPolicyEnv { /* what the policy reads of the env besides board and bombs */
    int a0[4], a1[4]; /* agent words of the packed record */
    int bIdx, bCnt;
};

template <class P>
struct PomSimplePolicy {
    P& p;
    const PomPolicyEnv& E;
    int id, sx, sy; /* me, and where I stand (the BFS source) */
    uint32_t m0, m1;
    int have_map;
    POM_HD PomSimplePolicy(P& p_, const PomPolicyEnv& e_, int id_, uint32_t m0_, uint32_t m1_)
        : p(p_), E(e_), id(id_), sx(0), sy(0), m0(m0_), m1(m1_), have_map(0)
    {
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
    POM_HD int in_danger(int x, int y) const /* IsInDanger, strategy.cpp:229-249 with IsInBombRange, strategy.hpp:163-169 */
    {
        int min_time = 99;
        POM_NOUNROLL
        for (int i = 0; i < E.bCnt; i++) {
            const int b = p.bomb(wrap20(E.bIdx + i));
            const int bx = pb_x(b), by = pb_y(b), s = pb_strength(b);
            const int hit = (y == by && bx - s <= x && x <= bx + s) || (x == bx && by - s <= y && y <= by + s);
            const int t = pb_time(b);
            min_time = (hit && t < min_time) ? t : min_time;
        }
        return min_time == 99 ? 0 : min_time;
    }
    POM_HD static int safe(int danger, int min) { return danger == 0 || danger >= min; } /* _safe_condition, strategy.cpp:199-202 */
    POM_HD int walkable_at(int x, int y) const { return !oob(x, y) && pc_is_walkable(p.cell(y * POM_N + x)); } /* _CheckPos */

    /* TryAdd, strategy.cpp:37-57 */
    POM_HD void try_add(int c, int dist, int nx, int ny, int& tail)
    {
        if (oob(nx, ny)) return;
        const int n = ny * POM_N + nx;
        const int item = p.cell(n);
        if ((p.rm(n) & 0xFF) == 0 && (pc_is_walkable(item) || pc_is_agent(item))) {
            p.set_rm(n, (dist + 1) | (c << 8));
            if (!pc_is_agent(item)) { /* paths to agents are recorded, the search does not continue through them */
                p.set_qe(tail, n);
                tail++;
            }
        }
    }
    POM_HD void need_map() /* FillRMap, strategy.cpp:59-93 — built on first use */
    {
        if (have_map) return;
        have_map = 1;
        p.clear_rm();
        int head = 0, tail = 0;
        p.set_qe(tail++, sy * POM_N + sx);
        POM_NOUNROLL
        while (head != tail) {
            const int c = p.qe(head++);
            const int cy = c / POM_N, cx = c - cy * POM_N;
            const int dist = p.rm(c) & 0xFF;
            if (cx != sx || cy + 1 != sy) try_add(c, dist, cx, cy + 1, tail);
            if (cx != sx || cy - 1 != sy) try_add(c, dist, cx, cy - 1, tail);
            if (cx + 1 != sx || cy != sy) try_add(c, dist, cx + 1, cy, tail);
            if (cx - 1 != sx || cy != sy) try_add(c, dist, cx - 1, cy, tail);
        }
    }
    POM_HD int move_towards(int tx, int ty) /* MoveTowardsPosition, strategy.cpp:99-121 */
    {
        const int src = sy * POM_N + sx;
        int cur = ty * POM_N + tx;
        POM_NOUNROLL
        for (int guard = 0; guard < 4 * POM_CELLS; guard++) {
            const int e = p.rm(cur);
            const int pred = e >> 8;
            if (pred == src) {
                const int cy = cur / POM_N, cx = cur - cy * POM_N;
                if (cx > sx) return POM_MOVE_RIGHT;
                if (cx < sx) return POM_MOVE_LEFT;
                if (cy > sy) return POM_MOVE_DOWN;
                if (cy < sy) return POM_MOVE_UP;
            } else if ((e & 0xFF) == 0) {
                return POM_MOVE_IDLE;
            }
            cur = pred;
        }
        return POM_MOVE_IDLE; /* the reference would spin here; its callers never ask for the source itself */
    }
    POM_HD int move_towards_safe_place(int radius) /* strategy.cpp:123-140: the window's upper bounds are `radius` (sic) */
    {
        need_map();
        const int y0 = sy - radius < 0 ? 0 : sy - radius, y1 = radius < POM_N ? radius : POM_N;
        const int x0 = sx - radius < 0 ? 0 : sx - radius, x1 = radius < POM_N ? radius : POM_N;
        POM_NOUNROLL
        for (int y = y0; y < y1; y++) {
            POM_NOUNROLL
            for (int x = x0; x < x1; x++) {
                const int dx_ = x - sx, dy_ = y - sy;
                if ((dx_ < 0 ? -dx_ : dx_) + (dy_ < 0 ? -dy_ : dy_) > radius) continue;
                if ((p.rm(y * POM_N + x) & 0xFF) != 0 && safe(in_danger(x, y), 2)) return move_towards(x, y);
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
            need_map();
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
        if (danger > 0) {
            const int mv = move_towards_safe_place(danger);
            const int px = sx + mv_dx(mv), py = sy + mv_dy(mv);
            if (walkable_at(px, py) && safe(in_danger(px, py), 2)) return mv;
            return one_safe_step(draw);
        }
        if (pom_sext8((uint32_t)av >> 16) < pom_sext16((uint32_t)a1v)) {
            if (adjacent_enemy(1)) return POM_MOVE_BOMB;
            const int near = adjacent_enemy(7);
            if (near && has_rp_loop()) return draw % 4;
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
