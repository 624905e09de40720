/*
 * pom_policy_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Literal CPU restatement of the reference's heuristic policy `agents::SimpleAgent`
 * (/root/reference/src/agents/simple_agent.cpp:1-148) and the strategy helpers it calls
 * (/root/reference/src/bboard/strategy.cpp, include/strategy.hpp), file:line cited per function.  It is the checker
 * for the HIP policy kernel.  Quirks are kept, not fixed: the scan window of
 * MoveTowardsSafePlace, SortDirections' "move to the back" that re-adds the wrong element, the stale moveQueue
 * slot read when only one direction is safe.  The agent's random draw is an INPUT (the reference draws from a
 * std::mt19937_64 seeded by random_device); oracle/fuzz_policy.c feeds both sides the same draw.
 *
 * Pinned against the compiled reference agent (oracle/_ref/libpomref.so) by oracle/fuzz_policy.c.
 */
#include "pom_policy_oracle.h"

#include <limits.h>
#include <string.h>

#include "pom_oracle.h"
#include "pom_rng.h"

#define N POM_BOARD_SIZE

typedef struct { int x, y; } Pos;
typedef struct { int map[N][N]; Pos source; } RMap; /* strategy.hpp:21-40: distance in the low half, predecessor index above */

static int oob(int x, int y) { return x < 0 || y < 0 || x >= N || y >= N; }
static int is_wood(int v) { return (v >> 8) == 2; }
static int is_walkable(int v) { return v == 0 || (v > 5 && v < 9); }
static int iabs(int v) { return v < 0 ? -v : v; }

static Pos desired(int x, int y, int move) /* util::DesiredPosition, step_utility.cpp:9-31 */
{
    Pos p = { x, y };
    if (move == POM_MOVE_UP) p.y -= 1;
    else if (move == POM_MOVE_DOWN) p.y += 1;
    else if (move == POM_MOVE_LEFT) p.x -= 1;
    else if (move == POM_MOVE_RIGHT) p.x += 1;
    return p;
}

static int rm_dist(const RMap* r, int x, int y) { return r->map[y][x] & 0xFFFF; }
static int rm_pred(const RMap* r, int x, int y) { return r->map[y][x] >> 16; }

static int in_bomb_range(int x, int y, int s, int px, int py) /* strategy.hpp:163-169 */
{
    return (py == y && (x - s <= px && px <= x + s)) || (px == x && (y - s <= py && py <= y + s));
}

static int is_in_danger(const PomState* st, int x, int y) /* strategy.cpp:229-249 */
{
    int min_time = INT_MAX;
    for (int i = 0; i < st->bombs.count; i++) {
        int b = st->bombs.queue[(st->bombs.index + i) % 20];
        if (in_bomb_range(b & 0xF, (b >> 4) & 0xF, (b >> 12) & 0xF, x, y)) {
            int t = (b >> 16) & 0xF;
            if (t < min_time) min_time = t;
        }
    }
    return min_time == INT_MAX ? 0 : min_time;
}

static int safe_condition(int danger, int min) { return danger == 0 || danger >= min; } /* strategy.cpp:199-202 */

static int check_pos(const PomState* st, int x, int y) /* strategy.cpp:194-197 */
{
    return !oob(x, y) && is_walkable(st->board[y][x]);
}

/* TryAdd, strategy.cpp:37-57 (the reference reads board[cy][cx] before its bounds test; the value is only used after it) */
static void try_add(const PomState* st, Pos* queue, int* qtail, RMap* r, Pos c, int cx, int cy)
{
    if (oob(cx, cy)) return;
    int item = st->board[cy][cx];
    if (rm_dist(r, cx, cy) == 0 && (is_walkable(item) || item >= POM_AGENT0)) {
        r->map[cy][cx] = (r->map[cy][cx] & 0xFFFF) + ((c.x + N * c.y) << 16);           /* SetPredecessor */
        r->map[cy][cx] = (r->map[cy][cx] & ~0xFFFF) + rm_dist(r, c.x, c.y) + 1;         /* SetDistance    */
        if (item < POM_AGENT0) queue[(*qtail)++] = (Pos){ cx, cy };
    }
}

static void fill_rmap(const PomState* st, RMap* r, int id) /* strategy.cpp:59-93 */
{
    memset(r->map, 0, sizeof r->map);
    int x = st->agents[id].x, y = st->agents[id].y;
    r->source = (Pos){ x, y };
    Pos queue[N * N + 4];
    int head = 0, tail = 0;
    queue[tail++] = (Pos){ x, y };
    while (head != tail) {
        Pos c = queue[head++];
        if (c.x != x || c.y + 1 != y) try_add(st, queue, &tail, r, c, c.x, c.y + 1);
        if (c.x != x || c.y - 1 != y) try_add(st, queue, &tail, r, c, c.x, c.y - 1);
        if (c.x + 1 != x || c.y != y) try_add(st, queue, &tail, r, c, c.x + 1, c.y);
        if (c.x - 1 != x || c.y != y) try_add(st, queue, &tail, r, c, c.x - 1, c.y);
    }
}

static int move_towards_position(const RMap* r, Pos position) /* strategy.cpp:99-121 */
{
    Pos curr = position;
    for (int guard = 0; guard < 4 * N * N; guard++) {
        int idx = rm_pred(r, curr.x, curr.y);
        int y = idx / N, x = idx % N;
        if (x == r->source.x && y == r->source.y) {
            if (curr.x > r->source.x) return POM_MOVE_RIGHT;
            if (curr.x < r->source.x) return POM_MOVE_LEFT;
            if (curr.y > r->source.y) return POM_MOVE_DOWN;
            if (curr.y < r->source.y) return POM_MOVE_UP;
        } else if (rm_dist(r, curr.x, curr.y) == 0) {
            return POM_MOVE_IDLE;
        }
        curr = (Pos){ x, y };
    }
    return POM_MOVE_IDLE; /* the reference would spin forever here; its callers never ask for the source itself */
}

static int move_towards_safe_place(const PomState* st, const RMap* r, int radius) /* strategy.cpp:123-140 */
{
    int ox = r->source.x, oy = r->source.y;
    for (int y = oy - radius; y < radius; y++) {      /* sic: the upper bounds are `radius`, not origin + radius */
        for (int x = ox - radius; x < radius; x++) {
            if (oob(x, y) || iabs(x - ox) + iabs(y - oy) > radius) continue;
            if (rm_dist(r, x, y) != 0 && safe_condition(is_in_danger(st, x, y), 2))
                return move_towards_position(r, (Pos){ x, y });
        }
    }
    return POM_MOVE_IDLE;
}

static int move_towards_enemy(const PomState* st, const RMap* r, int radius) /* strategy.cpp:165-192 */
{
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        const PomAgentInfo* inf = &st->agents[i];
        if ((inf->x == r->source.x && inf->y == r->source.y) || inf->dead) continue;
        if (iabs(inf->x - r->source.x) + iabs(inf->y - r->source.y) > radius) continue;
        return move_towards_position(r, (Pos){ inf->x, inf->y });
    }
    return POM_MOVE_IDLE;
}

static int is_adjacent_enemy(const PomState* st, int id, int distance) /* strategy.cpp:297-313 */
{
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        if (i == id || st->agents[i].dead) continue;
        if (iabs(st->agents[i].x - st->agents[id].x) + iabs(st->agents[i].y - st->agents[id].y) <= distance) return 1;
    }
    return 0;
}

static int is_adjacent_wood(const PomState* st, int id, int distance) /* IsAdjacentItem(.., Item::WOOD), strategy.cpp:315-338 */
{
    int ox = st->agents[id].x, oy = st->agents[id].y;
    for (int y = oy - distance; y <= oy + distance; y++)
        for (int x = ox - distance; x <= ox + distance; x++) {
            if (oob(x, y) || iabs(x - ox) + iabs(y - oy) > distance) continue;
            if (is_wood(st->board[y][x])) return 1; /* item == WOOD: IS_WOOD(item) && IS_WOOD(cell), or cell == WOOD (a wood too) */
        }
    return 0;
}

/* ---- the two FixedQueue<_,4> of the agent, raw slots and all (bboard.hpp:115-188) ---- */
static void mq_add(PomSimpleMem* m, int move)
{
    m->mq[(m->mq_index + m->mq_count) % 4] = move;
    m->mq_count++;
}
static int mq_at(const PomSimpleMem* m, int off) { return m->mq[(m->mq_index + off) % 4]; }
static void mq_remove_at(PomSimpleMem* m, int at)
{
    for (int i = at + 1; i < m->mq_count; i++) {
        int t = (m->mq_index + i) % 4;
        m->mq[(t - 1 + 4) % 4] = m->mq[t];
    }
    m->mq_count--;
}

static void safe_directions(const PomState* st, PomSimpleMem* m, int x, int y) /* strategy.cpp:203-226 */
{
    if (check_pos(st, x + 1, y) && safe_condition(is_in_danger(st, x + 1, y), 2)) mq_add(m, POM_MOVE_RIGHT);
    if (check_pos(st, x - 1, y) && safe_condition(is_in_danger(st, x - 1, y), 2)) mq_add(m, POM_MOVE_LEFT);
    if (check_pos(st, x, y + 1) && safe_condition(is_in_danger(st, x, y + 1), 2)) mq_add(m, POM_MOVE_DOWN);
    if (check_pos(st, x, y - 1) && safe_condition(is_in_danger(st, x, y - 1), 2)) mq_add(m, POM_MOVE_UP);
}

static void sort_directions(PomSimpleMem* m, int x, int y) /* strategy.hpp:130-152 */
{
    int moves = m->mq_count, total_removes = 0;
    for (int i = 0; i < moves && total_removes < 4; i++) {
        Pos pos = desired(x, y, mq_at(m, i));
        for (int j = 0; j < m->rp_count; j++) {
            const int32_t* p = m->rp[(m->rp_index + j) % 4];
            if (pos.x == p[0] && pos.y == p[1]) {
                mq_remove_at(m, i);
                mq_add(m, mq_at(m, i)); /* sic: re-adds what now sits at i, not what was removed */
                i--;
                total_removes++;
                break;
            }
        }
    }
}

static int has_rp_loop(const PomSimpleMem* m) /* simple_agent.cpp:24-35 */
{
    for (int i = 0; i < m->rp_count / 2; i++) {
        const int32_t* a = m->rp[(m->rp_index + i) % 4];
        const int32_t* b = m->rp[(m->rp_index + i + 2) % 4];
        if (!(a[0] == b[0] && a[1] == b[1])) return 0;
    }
    return 1;
}

static int move_safe_one_space(const PomState* st, int id, PomSimpleMem* m, int draw) /* simple_agent.cpp:37-48 */
{
    m->mq_count = 0;
    safe_directions(st, m, st->agents[id].x, st->agents[id].y);
    sort_directions(m, st->agents[id].x, st->agents[id].y);
    if (m->mq_count == 0) return POM_MOVE_IDLE;
    return mq_at(m, draw % 2);
}

static int decide(const PomState* st, int id, PomSimpleMem* m, int draw) /* simple_agent.cpp:51-122 */
{
    const PomAgentInfo* a = &st->agents[id];
    RMap r;
    fill_rmap(st, &r, id);
    int danger = is_in_danger(st, a->x, a->y);
    if (danger > 0) {
        int mv = move_towards_safe_place(st, &r, danger);
        Pos p = desired(a->x, a->y, mv);
        if (!oob(p.x, p.y) && is_walkable(st->board[p.y][p.x]) && safe_condition(is_in_danger(st, p.x, p.y), 2)) return mv;
        return move_safe_one_space(st, id, m, draw);
    }
    if (a->bombCount < a->maxBombCount) {
        if (is_adjacent_enemy(st, id, 1)) return POM_MOVE_BOMB;
        if (is_adjacent_enemy(st, id, 7) && has_rp_loop(m)) return draw % 4;
        if (is_adjacent_enemy(st, id, 7)) {
            int mv = move_towards_enemy(st, &r, 7);
            Pos p = desired(a->x, a->y, mv);
            if (!oob(p.x, p.y) && is_walkable(st->board[p.y][p.x]) && safe_condition(is_in_danger(st, p.x, p.y), 5)) return mv;
        }
        if (is_adjacent_wood(st, id, 1)) return POM_MOVE_BOMB;
    }
    m->mq_count = 0;
    safe_directions(st, m, a->x, a->y);
    sort_directions(m, a->x, a->y);
    if (m->mq_count == 0) return POM_MOVE_IDLE;
    return mq_at(m, draw % 2);
}

int32_t pom_oracle_simple_act(const void* state, int id, PomSimpleMem* mem, int draw) /* simple_agent.cpp:123-137 */
{
    const PomState* st = (const PomState*)state;
    int mv = decide(st, id, mem, draw);
    Pos p = desired(st->agents[id].x, st->agents[id].y, mv);
    if (4 - mem->rp_count == 0) { /* PopElem */
        mem->rp_index = (mem->rp_index + 1) % 4;
        mem->rp_count--;
    }
    int32_t* slot = mem->rp[(mem->rp_index + mem->rp_count) % 4];
    slot[0] = p.x;
    slot[1] = p.y;
    mem->rp_count++;
    return mv;
}

/* the strategy helpers by themselves, for the vectors recorded from the compiled reference (tests/golden/policy_traces.npz;
 * the reference's own [strategy] tests call them: unit_test/bboard/strategy_test.cpp) */
int32_t pom_oracle_is_adjacent_enemy(const void* state, int id, int distance) { return is_adjacent_enemy((const PomState*)state, id, distance); }
void pom_oracle_fill_rmap(const void* state, int id, int32_t* map121, int32_t* move_to121)
{
    RMap r;
    fill_rmap((const PomState*)state, &r, id);
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            map121[y * N + x] = r.map[y][x];
            const int ask = rm_dist(&r, x, y) != 0 && !(x == r.source.x && y == r.source.y);
            move_to121[y * N + x] = ask ? move_towards_position(&r, (Pos){ x, y }) : -1;
        }
}
/* the draw (0..4) that agent `agent` of env `env` is handed on tick `tick` of the synthetic stream (include/pom_rng.h), as
 * pom_oracle_simple_policy and the device policy take it */
int32_t pom_oracle_policy_draw(uint64_t seed, uint32_t env, uint32_t tick, int agent)
{
    const uint64_t r = pom_rng_draw(seed, env, tick);
    return (int32_t)((((uint32_t)(r >> (16 * agent)) & 0xFFFFu) * 5u) >> 16);
}

/* one round of act() for n envs: what pom_batch_policy_simple computes.  done[e] != 0 marks a finished env (all IDLE). */
void pom_oracle_simple_policy(const void* states, PomSimpleMem* mems, int n, uint64_t seed, int first_env, int tick,
                              const int32_t* done, int32_t* moves_out)
{
    const PomState* s = (const PomState*)states;
    for (int e = 0; e < n; e++) {
        const uint64_t r = pom_rng_draw(seed, (uint32_t)(first_env + e), (uint32_t)tick);
        for (int i = 0; i < 4; i++) {
            const int draw = (int)((((uint32_t)(r >> (16 * i)) & 0xFFFFu) * 5u) >> 16);
            const int skip = (done && done[e]) || s[e].agents[i].dead;
            moves_out[4 * e + i] = skip ? POM_MOVE_IDLE : pom_oracle_simple_act(&s[e], i, &mems[4 * e + i], draw);
        }
    }
}

int64_t pom_oracle_run_simple(void* states, const void* initial, PomSimpleMem* mems, int n, int ticks, uint64_t seed,
                              int first_env, int tick0, int max_steps)
{
    PomState* s = (PomState*)states;
    const PomState* init = (const PomState*)initial;
    int64_t steps = 0;
    for (int t = 0; t < ticks; t++) {
        for (int e = 0; e < n; e++) {
            PomState* st = &s[e];
            if (st->aliveAgents <= 1 || (max_steps > 0 && st->timeStep >= max_steps)) {
                *st = init[e];
                memset(&mems[4 * e], 0, 4 * sizeof(PomSimpleMem)); /* a new game gets fresh agents */
            }
            const uint64_t r = pom_rng_draw(seed, (uint32_t)(first_env + e), (uint32_t)(tick0 + t));
            int32_t mv[4];
            for (int i = 0; i < 4; i++) {
                const int draw = (int)((((uint32_t)(r >> (16 * i)) & 0xFFFFu) * 5u) >> 16);
                mv[i] = st->agents[i].dead ? POM_MOVE_IDLE : pom_oracle_simple_act(st, i, &mems[4 * e + i], draw);
            }
            pom_oracle_step(st, mv);
            st->timeStep++;
            steps++;
        }
    }
    return steps;
}
