/*
 * pom_policy_body.h — the reference's heuristic policy `agents::SimpleAgent` for ONE agent, written against an abstract
 * per-lane store so the identical source runs on gfx950 (pom_policy_kernel in pom_kernels.h, lane = agent) and, in
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
 * answer to (b) is: of the source's neighbours that lie on SOME shortest path to the target, the first in that order.  Both are
 * flood fills on 121-bit cell sets in registers, one dilation (4 multi-word shifts + masks) per level: (a) forwards from the
 * source until nothing grows — only the danger branch needs it; (b) backwards from the ONE target the decision is about, until
 * the flood first touches neighbours of the source — as many levels as the target is far.  No queue, no per-cell map, no LDS
 * traffic; (b) sits at one program point for both branches that need it.  On the device the floods of all agents of a wavefront are
 * run TOGETHER (round 3): the wavefront's flood jobs are dealt to its 16 quads, a quad holds one job's sets one 32-cell word per
 * lane (pom_quad_*_level below, ~25 VALU per level instead of ~70), so a wavefront pays for its longest flood at a third of the
 * price (pom_kernels.h: pom_coop_forward / pom_coop_backward).
 *
 * Agent memory, 2 dwords: m0 = recentPositions.queue[0..3], a byte each: x:4 | y:4 two's-complement nibbles (-1 .. 11);
 *                         m1 = recentPositions.index:2 | count:3 @2 | moveQueue.queue[0..3] 3 bits each @5 | moveQueue.count:3 @17
 * (moveQueue.index is always 0: the queue is never popped).  All-zero = a fresh agent.
 *
 * Per env and tick some things are prepared ONCE, by the four lanes of the env together (pom_policy_prepare_*), instead of being
 * recomputed per query: the danger map — IsInDanger(x, y) for every cell: the minimum timer over the bombs whose cross covers
 * it, a byte per cell — and three 121-bit cell sets, "walkable", "agent" and "safe", one 32-cell word per lane, which the floods and
 * the safe-place scan work on in registers instead of reading cells.
 *
 * Store interface P:  int cell(int c)            8-bit board code (pom_packed.h), c = y*11+x
 *                     int bomb(int slot)         raw bomb word of physical slot
 *                     uint32_t cells4(int k)     the codes of cells 4k .. 4k+3, a byte each; k up to 31 must be readable
 *                     int danger(int c) / void danger_init(int c) / void danger_put(int c, int t)     per-env danger map, 128
 *                                                entries of at least 8 bits (c up to 127 must be readable)
 *                     uint32_t setw(int k) / void set_put(int k, uint32_t bits)   per-env cell sets, words
 *                                                k = 0..3 walkable, 4..7 agents, 8..11 "safe" (_safe_condition(IsInDanger))
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

/* POM_DIAG builds (scripts/policy_stamps.py, never shipped): s_memtime deltas per phase of the policy kernel */
enum { POM_PP_LOAD = 0, POM_PP_PREPARE, POM_PP_PREDICATES, POM_PP_TARGET, POM_PP_PATH, POM_PP_TAIL, POM_PP_STORE, POM_PP_N };
#if defined(POM_DIAG) && defined(__HIP_DEVICE_COMPILE__)
#define POM_PSTAMP(k)                                 \
    do {                                              \
        const long long now_ = (long long)clock64();  \
        t_acc[k] += now_ - t_last;                    \
        t_last = now_;                                \
    } while (0)
#else
#define POM_PSTAMP(k) ((void)0)
#endif
/* POM_LEVEL_STATS (host analysis builds only, tests/emul/flood_levels.cpp): how many levels the two floods of an act() ran */
#if defined(POM_LEVEL_STATS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" int pom_stat_fwd, pom_stat_bwd;
#define POM_COUNT_LEVEL(v) ((v)++)
#else
#define POM_COUNT_LEVEL(v) ((void)0)
#endif

/* Four cell codes in a dword (pom_packed.h) -> 0x01 in the bytes of the walkable cells (passage 0, power-ups 3..5) and of the agent
 * cells (11..14).  Range tests on all four bytes at once: with the top bit of every byte cleared, adding 128 - k sets it again iff
 * the byte's low seven bits are >= k (no carry leaves a byte); codes >= 128 are flames. */
POM_HD void pom_cells_walk_agent(uint32_t d, uint32_t& walk, uint32_t& agent)
{
    const uint32_t l = d & 0x7F7F7F7Fu;
    const uint32_t ge1 = l + 0x7F7F7F7Fu, ge3 = l + 0x7D7D7D7Du, ge6 = l + 0x7A7A7A7Au, ge11 = l + 0x75757575u, ge15 = l + 0x71717171u;
    walk = ((((ge3 & ~ge6) | ~ge1) & ~d) >> 7) & 0x01010101u;
    agent = ((ge11 & ~ge15 & ~d) >> 7) & 0x01010101u;
}
/* the four 0 / 1 bytes of f as four bits — bits 0..3 (hi = 0) or 4..7 (hi = 1) — added to acc: one v_dot4_u32_u8 on the device */
POM_HD uint32_t pom_gather_flags(uint32_t f, uint32_t acc, int hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(f, hi ? 0x80402010u : 0x08040201u, acc, false);
#else
    const uint32_t n = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
    return acc + (hi ? n << 4 : n);
#endif
}

/* Shared preparation, executed by all four lanes of an env.  The 121-bit sets are four words and the env has four lanes:
 * member m builds word m of every set (cells 32m .. 32m+31) with plain stores, four cells at a time.  The
 * danger map is cleared and rasterised by bombs m, m+4, ...  Three phases — every lane must have finished one before any lane
 * starts the next (on the device the wavefront runs them back to back in lock-step; a sequential host emulation runs each
 * phase for all four members in turn). */
template <class P>
POM_HD void pom_policy_prepare_clear(P& p)
{
    const int m = p.member();
#pragma unroll
    for (int i = 0; i < 31; i++) {
        const int c = m + 4 * i;
        if (i < 30 || c < POM_CELLS) p.danger_init(c);
    }
}
template <class P>
POM_HD void pom_policy_prepare_fill(P& p, const PomPolicyEnv& E)
{
    const int m = p.member();
    /* walkable (IS_WALKABLE, bboard.hpp:81-84: passage or a power-up) -> word m, agent cells (item >= AGENT0) -> word 4+m: eight
     * board dwords of four cell codes each, classified four at a time (pom_cells_walk_agent) */
    uint32_t w = 0, g = 0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        uint32_t w0, g0, w1, g1;
        pom_cells_walk_agent(p.cells4(8 * m + i), w0, g0); /* cells 32m + 4i .. + 3; m = 3 runs past the board ... */
        pom_cells_walk_agent(p.cells4(8 * m + i + 1), w1, g1);
        w |= pom_gather_flags(w1, pom_gather_flags(w0, 0u, 0), 1) << (4 * i);
        g |= pom_gather_flags(g1, pom_gather_flags(g0, 0u, 0), 1) << (4 * i);
    }
    const uint32_t keep = m == 3 ? 0x01FFFFFFu : ~0u; /* ... into bytes that are not cells: 121 = 96 + 25 */
    p.set_put(m, w & keep);
    p.set_put(4 + m, g & keep);
    /* IsInDanger for every cell at once: min BMB_TIME over the bombs whose cross (IsInBombRange, strategy.hpp:163-169: the
     * +-strength row and column segments through the bomb, walls ignored) covers the cell.  Every lane walks every bomb and
     * takes one ARM of its cross (member 0: centre and +x, 1: -x, 2: +y, 3: -y): within a bomb the four lanes touch
     * different cells and the bombs come one after the other, so the minimum needs no atomic and the map can be bytes. */
    const int dx = (m == 0) - (m == 1), dy = (m == 2) - (m == 3);
    POM_NOUNROLL
    for (int i = 0; i < E.bCnt; i++) {
        const int b = p.bomb(wrap20(E.bIdx + i));
        const int bx = pb_x(b), by = pb_y(b), s = pb_strength(b), t = pb_time(b);
        if (bx >= POM_N || by >= POM_N) continue; /* upload validates live bombs; a stale word cannot index the map */
        POM_NOUNROLL
        for (int k = m == 0 ? 0 : 1; k <= s; k++) {
            const int x = bx + k * dx, y = by + k * dy;
            if (x < 0 || x >= POM_N || y < 0 || y >= POM_N) break;
            const int c = y * POM_N + x;
            if (t < p.danger(c)) p.danger_put(c, t);
        }
    }
}
/* third phase, after every lane's bombs are in the danger map: the cells that pass _safe_condition(IsInDanger(x, y), 2),
 * i.e. whose entry is not 1 (IsInDanger reports a minimum of 0 as "no danger"); word 8+m */
template <class P>
POM_HD void pom_policy_prepare_safe(P& p)
{
    const int m = p.member();
    uint32_t sb = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) sb |= (uint32_t)(p.danger(32 * m + i) != 1) << i; /* the map has 128 rows: no cell past 120 matters */
    p.set_put(8 + m, m == 3 ? sb & 0x01FFFFFFu : sb);
}

/* a set of board cells: bit c = y*11+x of a 121-bit number in four words */
struct PomCells {
    uint32_t w[4];
    POM_HD static PomCells zero() { return PomCells{{0u, 0u, 0u, 0u}}; }
    POM_HD int any() const { return (w[0] | w[1] | w[2] | w[3]) != 0; }
    POM_HD int has(int c) const
    {
        const int k = c >> 5;
        const uint32_t lo = (k & 1) ? w[1] : w[0], hi = (k & 1) ? w[3] : w[2]; /* two levels of selects, not a chain of branches */
        return (int)((((k & 2) ? hi : lo) >> (c & 31)) & 1u);
    }
    POM_HD void add(int c)
    {
        const int k = c >> 5;
        const uint32_t b = 1u << (c & 31);
        w[0] |= k == 0 ? b : 0u; w[1] |= k == 1 ? b : 0u; w[2] |= k == 2 ? b : 0u; w[3] |= k == 3 ? b : 0u;
    }
    POM_HD void add_range(int start, int len) /* cells start .. start+len-1, 1 <= len <= 11 */
    {
        const uint32_t run = (1u << len) - 1u;
        const int k = start >> 5, sh = start & 31;
        const uint32_t lo = run << sh, hi = sh + len > 32 ? run >> (32 - sh) : 0u;
        w[0] |= k == 0 ? lo : 0u;
        w[1] |= k == 1 ? lo : k == 0 ? hi : 0u;
        w[2] |= k == 2 ? lo : k == 1 ? hi : 0u;
        w[3] |= k == 3 ? lo : k == 2 ? hi : 0u;
    }
    POM_HD int lowest() const /* smallest member, -1 if empty */
    {
        if (w[0]) return __builtin_ctz(w[0]);
        if (w[1]) return 32 + __builtin_ctz(w[1]);
        if (w[2]) return 64 + __builtin_ctz(w[2]);
        if (w[3]) return 96 + __builtin_ctz(w[3]);
        return -1;
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

/* ---- a 121-bit cell set spread over the four lanes of a quad ------------------------------------------------------------------
 * One 32-cell word per lane: lane k of the quad holds cells 32k .. 32k+31.  A dilation then costs each lane two word exchanges
 * with its neighbours in the quad and a dozen logic ops instead of the ~50 that four words in one lane's registers take — what
 * the wave-cooperative floods of pom_kernels.h are made of.  Q says what "one word per lane" is: on the device a uint32_t per
 * lane with DPP quad permutes for prev / next (PomQuadLanes, pom_kernels.h); in tests/emul four words at once (PomQuadHost), so
 * that the very same level functions are fuzzed against the four-register floods below without a GPU.
 *   Q::W                     the word type;  & | ~ << >> work lane-wise
 *   q.prev(w) / q.next(w)    the word of lane k-1 / k+1 (0 at the ends)
 *   q.any(w)                 is any bit set in any of the four words (the same answer in all four lanes)
 *   q.bit(c)                 the set {c}
 *   q.col0() / q.col10()     the cells of column x = 0 / x = 10;  q.valid(): the 121 cells
 *   q.lowest(w)              the smallest member or 999;  q.gates_hit(w, g8): which of the four cells packed in g8 (a byte each,
 *                            0xFF = none) are members, as a 4-bit mask
 */
template <class Q>
POM_HD typename Q::W pom_quad_neighbours(const Q& q, typename Q::W w)
{
    const typename Q::W p = q.prev(w), n = q.next(w);
    return (((w << 11) | (p >> 21)) | ((w >> 11) | (n << 21)) | (((w << 1) | (p >> 31)) & ~q.col0()) | (((w >> 1) | (n << 31)) & ~q.col10())) & q.valid();
}
/* one level of FillRMap's reach (PomSimplePolicy::forward_reach below): returns whether anything grew */
template <class Q>
POM_HD bool pom_quad_forward_level(const Q& q, typename Q::W walk, typename Q::W agents, typename Q::W& front, typename Q::W& all)
{
    const typename Q::W nb = pom_quad_neighbours(q, front) & ~all;
    const typename Q::W grown = nb & walk;
    all = all | grown | (nb & agents);
    front = grown;
    return q.any(grown);
}
/* one level of MoveTowardsPosition's backward flood (PomSimplePolicy::path_local below): -1 = go on, else the flood is over and
 * the value is the 4-bit mask of gates this level reached (0: the flood died out) */
template <class Q>
POM_HD int pom_quad_backward_level(const Q& q, typename Q::W walk, typename Q::W gates, uint32_t g8, typename Q::W& front, typename Q::W& seen)
{
    front = pom_quad_neighbours(q, front) & ~seen & walk;
    if (!q.any(front)) return 0;
    seen = seen | front;
    const typename Q::W hit = front & gates;
    if (!q.any(hit)) return -1;
    return q.gates_hit(hit, g8);
}

/* word k (cells 32k .. 32k+31) of MoveTowardsSafePlace's scan window around (sx, sy) for `radius` (PomSimplePolicy::window below
 * is the whole set): only the three or four board rows that overlap the word are looked at */
POM_HD uint32_t pom_window_word(int k, int sx, int sy, int radius)
{
    const int lim = radius < POM_N ? radius : POM_N; /* exclusive upper bound of x and of y (sic, strategy.cpp:128-129) */
    const int y0 = sy - radius < 0 ? 0 : sy - radius;
    const int first = div11(32 * k);
    uint32_t w = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = first + i;
        const int dy_ = y - sy;
        const int rem = radius - (dy_ < 0 ? -dy_ : dy_);
        const int x0 = sx - rem < 0 ? 0 : sx - rem;
        const int x1 = sx + rem < lim - 1 ? sx + rem : lim - 1;
        const int ok = (int)(y >= y0) & (int)(y < lim) & (int)(rem >= 0) & (int)(x0 <= x1);
        const int s = y * POM_N + x0 - 32 * k; /* where the run starts relative to the word: -10 .. 43 */
        const uint32_t run = (1u << (x1 - x0 + 1 > 0 ? x1 - x0 + 1 : 0)) - 1u; /* at most 11 bits */
        const uint32_t bits = s >= 0 ? (s < 32 ? run << s : 0u) : run >> (-s);
        w |= ok ? bits : 0u;
    }
    return k == 3 ? w & 0x01FFFFFFu : w;
}

/* safe_directions() + sort_directions() (strategy.cpp:203-226, strategy.hpp:130-152) as ONE table look-up.  What the two do to the
 * move queue is a function of which of the four steps RIGHT, LEFT, DOWN, UP are safe (they are queued in that order) and which of
 * them lead onto a recent position ("hit"): SortDirections' loop — RemoveAt(i), then AddElem of what NOW sits at i (sic), at most
 * four removals, the bound fixed at the original count — only ever reads and writes the slots the safe steps were queued into.
 * 256 entries, built at compile time by running that very loop: bits 0..11 the queue's slots 0..3 (3 bits each; slots past the
 * number of safe steps are not touched and read 0 here), bits 12..14 the count.  (As a loop over lane-varying queues the sort
 * cost a wavefront its longest queue's 6 - 8 iterations: ~250 VALU of the fused SimpleAgent kernel's 4.3 k.) */
struct PomSortTable {
    uint16_t e[256];
};
constexpr PomSortTable pom_make_sort_table()
{
    PomSortTable t{};
    for (int safe = 0; safe < 16; safe++) {
        for (int hit = 0; hit < 16; hit++) {
            int q[4] = {0, 0, 0, 0}, cnt = 0;
            const int dirs[4] = {POM_MOVE_RIGHT, POM_MOVE_LEFT, POM_MOVE_DOWN, POM_MOVE_UP};
            for (int k = 0; k < 4; k++)
                if ((safe >> k) & 1) {
                    q[cnt & 3] = dirs[k];
                    cnt++;
                }
            const int moves = cnt;
            int removes = 0;
            for (int i = 0; i < moves && removes < 4; i++) {
                const int mv = q[i & 3];
                const int k = mv == POM_MOVE_RIGHT ? 0 : mv == POM_MOVE_LEFT ? 1 : mv == POM_MOVE_DOWN ? 2 : 3;
                if ((hit >> k) & 1) {
                    for (int j = i + 1; j < cnt; j++) q[(j - 1) & 3] = q[j & 3]; /* RemoveAt(i) */
                    cnt--;
                    q[cnt & 3] = q[i & 3]; /* AddElem(queue[i]): what now sits at i, not what was removed */
                    cnt++;
                    i--;
                    removes++;
                }
            }
            int bits = cnt << 12;
            for (int k = 0; k < moves; k++) bits |= q[k] << (3 * k);
            t.e[safe | (hit << 4)] = (uint16_t)bits;
        }
    }
    return t;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const PomSortTable pom_sort_table = pom_make_sort_table();
#else
static const PomSortTable pom_sort_table = pom_make_sort_table();
#endif

template <class P>
struct PomSimplePolicy {
    P& p;
    const PomPolicyEnv& E;
    int id, sx, sy; /* me, and where I stand (the search's source) */
    uint32_t m0, m1;
    int danger_ = 0, can_bomb_ = 0, adj1_ = 0, near_ = 0, looping_ = 0; /* begin()'s answers */
    PomCells all; /* the cells FillRMap reaches (GetDistance != 0); built by forward_reach() — the one-lane floods only */
#if defined(POM_DIAG)
    long long t_last, t_acc[POM_PP_N];
#endif
    POM_HD PomSimplePolicy(P& p_, const PomPolicyEnv& e_, int id_, uint32_t m0_, uint32_t m1_)
        : p(p_), E(e_), id(id_), sx(0), sy(0), m0(m0_), m1(m1_)
    {
        all = PomCells::zero();
        const int av = sel4(id, E.a0);
        sx = ag_x(av);
        sy = ag_y(av);
    }
    POM_HD int src_cell() const { return sy * POM_N + sx; }
    /* the env's walkable / agent cells as the searches see them: FillRMap never re-enters its source (strategy.cpp:83-90) */
    POM_HD int walk_has(int c) const { return (int)((p.setw(c >> 5) >> (c & 31)) & 1u) & (int)(c != src_cell()); }
    POM_HD int agent_has(int c) const { return (int)((p.setw(4 + (c >> 5)) >> (c & 31)) & 1u) & (int)(c != src_cell()); }

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
    POM_HD static int safe(int danger, int min) { return (int)(danger == 0) | (int)(danger >= min); } /* _safe_condition, strategy.cpp:199-202 */
    POM_HD int walkable_at(int x, int y) const /* _CheckPos */
    {
        if (oob(x, y)) return 0;
        int w = pc_is_walkable(p.cell(y * POM_N + x));
        POM_IN_VGPR(w); /* an opaque value: hipcc otherwise splits the range test into a chain of exec-mask branches at every use */
        return w;
    }

    /* (a) FillRMap, strategy.cpp:59-93: which cells get a distance.  A cell can be entered if it is walkable or holds an agent
     * (:43-44); the search continues only through walkable cells, agents are reached but not passed (:50-53).  The one-lane form
     * (host builds; the kernels run pom_quad_forward_level over the same sets, four lanes per flood). */
    POM_HD void forward_reach()
    {
        PomCells walk, agents;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            walk.w[k] = p.setw(k);
            agents.w[k] = p.setw(4 + k);
        }
        agents.remove(src_cell());
        walk.remove(src_cell());
        PomCells front = PomCells::zero();
        front.add(src_cell());
        all = PomCells::zero();
        POM_NOUNROLL
        for (int level = 0; level < POM_CELLS; level++) {
            POM_COUNT_LEVEL(pom_stat_fwd);
            const PomCells nb = front.neighbours().minus(all);
            const PomCells grown = nb & walk;
            all = all | grown | (nb & agents);
            front = grown;
            if (!grown.any()) break;
        }
    }
    /* MoveTowardsSafePlace, strategy.cpp:123-140.  The reference scans y from sy-radius while y < radius, x from sx-radius while
     * x < radius (sic: the upper bounds are `radius`, not origin + radius), skips cells off the board or further than `radius`
     * in Manhattan distance, and takes the first reachable cell that passes _safe_condition.  Scan order = ascending cell index,
     * so: window set (one run of cells per row) & reachable & safe, lowest member.  window(): the window set. */
    POM_HD PomCells window(int radius) const
    {
        const int lim = radius < POM_N ? radius : POM_N; /* exclusive upper bound of x and of y */
        PomCells win = PomCells::zero();
        const int y0 = sy - radius < 0 ? 0 : sy - radius;
        POM_NOUNROLL
        for (int y = y0; y < lim; y++) {
            const int dy_ = y - sy;
            const int rem = radius - (dy_ < 0 ? -dy_ : dy_);
            if (rem < 0) continue;
            const int x0 = sx - rem < 0 ? 0 : sx - rem;
            const int x1 = sx + rem < lim - 1 ? sx + rem : lim - 1;
            if (x0 <= x1) win.add_range(y * POM_N + x0, x1 - x0 + 1);
        }
        return win;
    }
    POM_HD int safe_place(int radius) /* the cell MoveTowardsSafePlace heads for, -1 if none; one-lane form, after forward_reach() */
    {
        PomCells safe_cells;
#pragma unroll
        for (int k = 0; k < 4; k++) safe_cells.w[k] = p.setw(8 + k);
        return (window(radius) & all & safe_cells).lowest();
    }
    POM_HD int manhattan_to(int j) const
    {
        const int dx_ = ag_x(E.a0[j]) - sx, dy_ = ag_y(E.a0[j]) - sy;
        return (dx_ < 0 ? -dx_ : dx_) + (dy_ < 0 ? -dy_ : dy_);
    }
    POM_HD int enemy_cell(int radius) const /* the cell MoveTowardsEnemy heads for (strategy.cpp:165-192), -1 if none */
    {
        int c = -1;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int av = E.a0[j];
            if ((ag_x(av) == sx && ag_y(av) == sy) || ag_dead(av)) continue;
            if (manhattan_to(j) > radius) continue;
            c = ag_y(av) * POM_N + ag_x(av);
        }
        return c;
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
    /* _CheckPos(x, y) && _safe_condition(IsInDanger(x, y), 2) without a branch: off the board the cell index is clamped to
     * the agent's own (both reads stay in the tile) and the answer masked */
    POM_HD int step_is_safe(int x, int y) const
    {
        const int on = !oob(x, y);
        const int c = on ? y * POM_N + x : sy * POM_N + sx;
        const int dv = p.danger(c);
        int ok = on & pc_is_walkable(p.cell(c)) & (int)(dv != 1); /* safe(danger, 2): no danger (map: 99, or a minimum of 0) or >= 2 */
        POM_IN_VGPR(ok);
        return ok;
    }
    POM_HD void safe_directions() /* strategy.cpp:203-226 */
    {
        const int r = step_is_safe(sx + 1, sy), l = step_is_safe(sx - 1, sy), d = step_is_safe(sx, sy + 1), u = step_is_safe(sx, sy - 1);
        if (r) mq_add(POM_MOVE_RIGHT);
        if (l) mq_add(POM_MOVE_LEFT);
        if (d) mq_add(POM_MOVE_DOWN);
        if (u) mq_add(POM_MOVE_UP);
    }
    POM_HD void sort_directions() /* SortDirections, strategy.hpp:130-152, with FixedQueue::RemoveAt / AddElem on raw slots */
    {
        const int moves = mq_count();
        int removes = 0;
        /* "is this cell one of the recent positions" on all four bytes of m0 at once: 0x80 in every byte that holds a live entry
         * (slots index .. index + count - 1, cyclically), and an exact zero-byte test of m0 ^ key-in-every-byte */
        const int rc = rp_count(), rsh = 8 * rp_index();
        const uint32_t live0 = rc >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * rc)) - 1u));
        const uint32_t live = (live0 << rsh) | (rsh ? live0 >> (32 - rsh) : 0u);
        POM_NOUNROLL
        for (int i = 0; i < moves && removes < 4; i++) {
            const int mv = mq_at(i);
            const int key = pos_key(sx + mv_dx(mv), sy + mv_dy(mv));
            const uint32_t x = m0 ^ ((uint32_t)key * 0x01010101u);
            const uint32_t zero = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); /* 0x80 exactly in the bytes of x that are 0 */
            const int hit = (zero & live) != 0u;
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
    /* is (x, y) one of the recent positions — on all four bytes of m0 at once: 0x80 in every byte that holds a live entry (slots
     * index .. index + count - 1, cyclically) and an exact zero-byte test of m0 ^ key-in-every-byte */
    POM_HD int is_recent(int x, int y, uint32_t live) const
    {
        const uint32_t v = m0 ^ ((uint32_t)pos_key(x, y) * 0x01010101u);
        const uint32_t zero = ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
        return (zero & live) != 0u;
    }
    POM_HD int one_safe_step(int draw) /* the common tail of _Decide and _MoveSafeOneSpace, simple_agent.cpp:37-48,105-121 */
    {
#if defined(POM_SORT_LOOP) /* the literal form: queue the safe steps, then SortDirections' loop */
        mq_set_count(0);
        safe_directions();
        sort_directions();
#else
        const int r = step_is_safe(sx + 1, sy), l = step_is_safe(sx - 1, sy), d = step_is_safe(sx, sy + 1), u = step_is_safe(sx, sy - 1);
        const int rc = rp_count(), rsh = 8 * rp_index();
        const uint32_t live0 = rc >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * rc)) - 1u));
        const uint32_t live = (live0 << rsh) | (rsh ? live0 >> (32 - rsh) : 0u);
        const int idx = r | (l << 1) | (d << 2) | (u << 3) | (is_recent(sx + 1, sy, live) << 4) | (is_recent(sx - 1, sy, live) << 5) |
                        (is_recent(sx, sy + 1, live) << 6) | (is_recent(sx, sy - 1, live) << 7);
        const uint32_t e = pom_sort_table.e[idx];
        const int queued = r + l + d + u;
        const uint32_t keep = (0xFFFu << (3 * queued)) & 0xFFFu; /* the slots behind the queued steps keep what they held */
        const uint32_t q = (((m1 >> 5) & 0xFFFu) & keep) | (e & 0xFFFu);
        m1 = (m1 & ~((0xFFFu << 5) | (7u << 17))) | (q << 5) | ((e >> 12) << 17);
#endif
        if (mq_count() == 0) return POM_MOVE_IDLE;
        return mq_at(draw % 2);
    }

    /* ---- _Decide (simple_agent.cpp:51-122) in the pieces between which its two searches sit.  The kernels run the searches for
     * all agents of a wavefront together (pom_kernels.h: pom_coop_forward / pom_coop_backward); act() below strings the pieces
     * together with the one-lane searches. ---- */
    /* the predicates; returns whether the agent is in danger, i.e. needs FillRMap's reach and a safe place in it */
    POM_HD int begin()
    {
        const int av = sel4(id, E.a0), a1v = sel4(id, E.a1);
        danger_ = in_danger(sx, sy);
        can_bomb_ = ag_bombcount(av) < ag_max_bombs(a1v);
        adj1_ = adjacent_enemy(1);
        near_ = adjacent_enemy(7);
        looping_ = has_rp_loop();
        POM_PSTAMP(POM_PP_PREDICATES);
        return danger_ > 0;
    }
    /* Which cell, if any, does this decision want a path to?  In danger: the first safe reachable cell of the scan window
     * (`safe_cell`: what the forward search found, -1 if nothing); else, allowed to bomb, an enemy within 7 and nothing more
     * urgent: that enemy. */
    POM_HD int pick_target(int safe_cell) const
    {
        if (danger_ > 0) return safe_cell;
        const int chasing = can_bomb_ && !adj1_ && near_ && !looping_;
        return chasing ? enemy_cell(7) : -1;
    }
    /* (b) MoveTowardsPosition, strategy.cpp:99-121, for a target != source: the Move of the first step, IDLE if the search
     * never reached the target.  Backwards from the target through walkable cells; the first level that touches walkable
     * neighbours of the source ("gates") decides, DOWN before UP before RIGHT before LEFT (the order FillRMap tries them in).
     * path_begin: what needs no search — `found` if the target is next to me; the gates, a cell index per byte of g8 in that
     * order (0xFF = none); returns whether the flood has to run. */
    POM_HD int path_begin(int target, int& found, uint32_t& g8) const
    {
        found = POM_MOVE_IDLE;
        g8 = 0xFFFFFFFFu;
        if (target < 0) return 0;
        const int nx[4] = {sx, sx, sx + 1, sx - 1}, ny[4] = {sy + 1, sy - 1, sy, sy};
        const int mvk[4] = {POM_MOVE_DOWN, POM_MOVE_UP, POM_MOVE_RIGHT, POM_MOVE_LEFT};
        const int enterable = walk_has(target) | agent_has(target);
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (oob(nx[k], ny[k])) continue;
            const int n = ny[k] * POM_N + nx[k];
            if (n == target && enterable) found = mvk[k]; /* the target is next to me */
            if (walk_has(n)) g8 = (g8 & ~(0xFFu << (8 * k))) | ((uint32_t)n << (8 * k));
        }
        /* The flood's first level is the target's walkable neighbours: it ends there if one of them is a gate, i.e. if the target
         * is two steps away through a gate.  A third of all floods are that short (tests/emul/flood_levels.sh): answered here,
         * they do not take a quad of the wavefront's cooperative floods. */
        if (found == POM_MOVE_IDLE && enterable) {
            const int ty = div11(target), tx = target - ty * POM_N;
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                const int ddx = nx[k] - tx, ddy = ny[k] - ty;
                if (((g8 >> (8 * k)) & 0xFF) != 0xFF && (ddx < 0 ? -ddx : ddx) + (ddy < 0 ? -ddy : ddy) == 1) found = mvk[k];
            }
        }
        return found == POM_MOVE_IDLE && enterable && g8 != 0xFFFFFFFFu;
    }
    /* the flood, one-lane form: which gates the first level that reaches any of them reaches (4-bit mask, 0 = none) */
    POM_HD int path_local(int target, uint32_t g8) const
    {
        PomCells walk, gates = PomCells::zero();
#pragma unroll
        for (int k = 0; k < 4; k++) walk.w[k] = p.setw(k);
        walk.remove(src_cell());
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (((g8 >> (8 * k)) & 0xFF) != 0xFF) gates.add((int)((g8 >> (8 * k)) & 0xFF));
        PomCells seen = PomCells::zero(), front = PomCells::zero();
        seen.add(target);
        front.add(target);
        POM_NOUNROLL
        for (int level = 0; level < POM_CELLS; level++) {
            POM_COUNT_LEVEL(pom_stat_bwd);
            front = front.neighbours().minus(seen) & walk;
            if (!front.any()) break;
            seen = seen | front;
            const PomCells hit = front & gates;
            if (hit.any()) {
                int m = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (((g8 >> (8 * k)) & 0xFF) != 0xFF && hit.has((int)((g8 >> (8 * k)) & 0xFF))) m |= 1 << k;
                return m;
            }
        }
        return 0;
    }
    POM_HD int path_end(int target, int found, int hit_mask) const
    {
        if (target < 0) return POM_MOVE_IDLE;
        const int mvk[4] = {POM_MOVE_DOWN, POM_MOVE_UP, POM_MOVE_RIGHT, POM_MOVE_LEFT};
        if (found == POM_MOVE_IDLE) {
#pragma unroll
            for (int k = 3; k >= 0; k--)
                if ((hit_mask >> k) & 1) found = mvk[k];
        }
        if (found != POM_MOVE_IDLE) return found;
        /* unreached target: its map entry is 0, i.e. "distance 0, predecessor cell 0".  The reference takes cell 0 for the
         * source if the agent stands there and answers by comparing coordinates (:107-113); otherwise IDLE (:115-118) */
        if (sx == 0 && sy == 0) {
            const int ty = div11(target), tx = target - ty * POM_N;
            if (tx > 0) return POM_MOVE_RIGHT;
            if (ty > 0) return POM_MOVE_DOWN;
        }
        return POM_MOVE_IDLE;
    }
    /* the rest of _Decide with the path's first step `mv` in hand, then SimpleAgent::act's bookkeeping (simple_agent.cpp:123-137) */
    POM_HD int finish(int mv, int draw)
    {
        POM_PSTAMP(POM_PP_PATH);
        /* the three ways out of _Decide that do not end in _MoveSafeOneSpace's tail; everything else falls through to it, at
         * one program point (three inlined copies under three lane masks would cost the wavefront three times) */
        const int px = sx + mv_dx(mv), py = sy + mv_dy(mv);
        const int step_ok = walkable_at(px, py);
        const int d_next = step_ok ? in_danger(px, py) : 0;
        int out = -1;
        if (danger_ > 0) {
            if (step_ok && safe(d_next, 2)) out = mv;
        } else if (can_bomb_) {
            if (adj1_) out = POM_MOVE_BOMB;
            else if (near_ && looping_) out = draw % 4;
            else if (near_ && step_ok && safe(d_next, 5)) out = mv;
            else if (adjacent_wood()) out = POM_MOVE_BOMB;
        }
        if (out < 0) out = one_safe_step(draw);
        const int key = pos_key(sx + mv_dx(out), sy + mv_dy(out));
        int idx = rp_index(), cnt = rp_count();
        if (cnt == 4) { /* RemainingCapacity() == 0: PopElem */
            idx = (idx + 1) & 3;
            cnt--;
        }
        const int slot = (idx + cnt) & 3;
        m0 = (m0 & ~(0xFFu << (8 * slot))) | ((uint32_t)key << (8 * slot));
        cnt++;
        m1 = (m1 & ~31u) | (uint32_t)idx | ((uint32_t)cnt << 2);
        POM_PSTAMP(POM_PP_TAIL);
        return out;
    }
    POM_HD int act(int draw) /* SimpleAgent::act with the one-lane searches (host builds) */
    {
        int safe_cell = -1;
        if (begin()) {
            forward_reach();
            safe_cell = safe_place(danger_);
        }
        const int target = pick_target(safe_cell);
        POM_PSTAMP(POM_PP_TARGET);
        int found;
        uint32_t g8;
        int hit = 0;
        if (path_begin(target, found, g8)) hit = path_local(target, g8);
        return finish(path_end(target, found, hit), draw);
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
