/*
 * pom_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, deliberately literal CPU restatement of the reference's
 * simulation tick `bboard::Step(State*, Move*)` and of the post-step
 * bookkeeping of `Environment::Step`, written from the reference's behaviour
 * (file:line cited per function, paths relative to /root/reference).  It is the
 * checker the HIP path is compared against.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (pomcpp_amd/csrc) never links or calls it.
 *
 * Pinning: `make -C oracle ref` builds the unmodified reference sources into
 * oracle/_ref/libpomref.so; oracle/fuzz_diff.c steps both on identical inputs
 * (see DESIGN.md for the step counts), and tests/golden/ holds trajectories
 * produced by the compiled reference.
 *
 * Where the reference has undefined behaviour on reachable inputs the oracle
 * takes the fallback documented in SURVEY.md §9 and raises a POM_UB_* flag.
 */
#include "pom_oracle.h"

#include <string.h>

#define N POM_BOARD_SIZE
#define QN POM_MAX_BOMBS

typedef struct { int x, y; } Pos;

/* ---- item predicates: include/bboard.hpp:73-109 ------------------------- */
static int is_wood(int v) { return (v >> 8) == 2; }
static int is_powerup(int v) { return v > 5 && v < 9; }
static int is_walkable(int v) { return is_powerup(v) || v == 0; }
static int is_flame(int v) { return (v >> 16) == 4; }
static int is_agent(int v) { return v >= (1 << 24); }
static int is_static_mov_block(int v) { return is_wood(v) || is_powerup(v) || v == 1; }
static int flame_id(int v) { return (v & 0xFFFF) >> 3; }

/* ---- bomb bit fields: include/bboard.hpp:261-335 ------------------------ */
static int bmb_pos(int b) { return b & 0xFF; }
static int bmb_x(int b) { return b & 0xF; }
static int bmb_y(int b) { return (b & 0xF0) >> 4; }
static int bmb_id(int b) { return (b & 0xF00) >> 8; }
static int bmb_strength(int b) { return (b & 0xF000) >> 12; }
static int bmb_time(int b) { return (b & 0xF0000) >> 16; }
static int bmb_dir(int b) { return (b & 0xF00000) >> 20; }

/* all arithmetic on the bomb word is done unsigned: same bits as the
 * reference's int arithmetic, without signed-overflow UB in the checker */
static int set_field(int b, unsigned mask, unsigned v)
{
    return (int)(((unsigned)b & ~mask) + v);
}

/* ---- FixedQueue<T,20>: include/bboard.hpp:115-188 ----------------------- */
static int bq(const PomState *s, int off) { return (s->bombs.index + off) % QN; }
static int fq(const PomState *s, int off) { return (s->flames.index + off) % QN; }

static void bombs_remove_at(PomState *s, int at) /* bboard.hpp:151-160 */
{
    for (int i = at + 1; i < s->bombs.count; i++) {
        int t = (s->bombs.index + i) % QN;
        s->bombs.queue[(t - 1 + QN) % QN] = s->bombs.queue[t];
    }
    s->bombs.count--;
}

static int oob(int x, int y) /* step_utility.hpp:155-166 */
{
    return x < 0 || y < 0 || x >= N || y >= N;
}

static void kill_agent(PomState *s, int id, uint32_t *ub) /* bboard.hpp:474-481 */
{
    if (id < 0 || id >= POM_AGENT_COUNT) {
        *ub |= POM_UB_BAD_INDEX;
        return;
    }
    if (!s->agents[id].dead) {
        s->agents[id].dead = 1;
        s->aliveAgents--;
    }
}

static int has_bomb(const PomState *s, int x, int y) /* bboard.cpp:265-275 */
{
    for (int i = 0; i < s->bombs.count; i++) {
        int b = s->bombs.queue[bq(s, i)];
        if (bmb_x(b) == x && bmb_y(b) == y)
            return 1;
    }
    return 0;
}

static int get_bomb_index(const PomState *s, int x, int y) /* bboard.cpp:301-311 (and GetBomb :277-287) */
{
    for (int i = 0; i < s->bombs.count; i++) {
        int b = s->bombs.queue[bq(s, i)];
        if (bmb_x(b) == x && bmb_y(b) == y)
            return i;
    }
    return -1;
}

static int get_agent(const PomState *s, int x, int y) /* bboard.cpp:289-299 */
{
    for (int i = 0; i < POM_AGENT_COUNT; i++)
        if (!s->agents[i].dead && s->agents[i].x == x && s->agents[i].y == y)
            return i;
    return -1;
}

static int flag_item(int pwp) /* bboard.cpp:182-189 */
{
    if (pwp == 1) return POM_EXTRABOMB;
    if (pwp == 2) return POM_INCRRANGE;
    if (pwp == 3) return POM_KICK;
    return POM_PASSAGE;
}

static Pos desired(int x, int y, int move) /* step_utility.cpp:9-31 */
{
    Pos p = { x, y };
    if (move == POM_MOVE_UP) p.y -= 1;
    else if (move == POM_MOVE_DOWN) p.y += 1;
    else if (move == POM_MOVE_LEFT) p.x -= 1;
    else if (move == POM_MOVE_RIGHT) p.x += 1;
    return p;
}

static Pos origin_of(int x, int y, int move) /* step_utility.cpp:33-55 */
{
    Pos p = { x, y };
    if (move == POM_MOVE_DOWN) p.y -= 1;
    else if (move == POM_MOVE_UP) p.y += 1;
    else if (move == POM_MOVE_RIGHT) p.x -= 1;
    else if (move == POM_MOVE_LEFT) p.x += 1;
    return p;
}

static Pos bomb_desired(int b) /* step_utility.cpp:57-60 */
{
    return desired(bmb_x(b), bmb_y(b), bmb_dir(b));
}

/* ---- explosions: src/bboard/bboard.cpp:24-57, 111-118, 191-263 ---------- */
static void spawn_flame(PomState *s, int x, int y, int strength, uint32_t *ub);

static void explode_bomb_at(PomState *s, int i, uint32_t *ub) /* bboard.cpp:111-118 */
{
    int b = s->bombs.queue[bq(s, i)];
    int owner = bmb_id(b);
    int strength = 0;
    if (owner < POM_AGENT_COUNT) strength = s->agents[owner].bombStrength;
    else *ub |= POM_UB_BAD_INDEX;
    spawn_flame(s, bmb_x(b), bmb_y(b), strength, ub);
    /* the slot is re-read after the nested explosions (stale index) */
    owner = bmb_id(s->bombs.queue[bq(s, i)]);
    if (owner < POM_AGENT_COUNT) s->agents[owner].bombCount--;
    else *ub |= POM_UB_BAD_INDEX;
    bombs_remove_at(s, i);
}

static int spawn_flame_item(PomState *s, int x, int y, int signature, uint32_t *ub) /* bboard.cpp:24-57 */
{
    if (s->board[y][x] >= POM_AGENT0)
        kill_agent(s, s->board[y][x] - POM_AGENT0, ub);
    if (s->board[y][x] == POM_BOMB || s->board[y][x] >= POM_AGENT0) {
        for (int i = 0; i < s->bombs.count; i++) {
            if (bmb_pos(s->bombs.queue[bq(s, i)]) == (x + (y << 4))) {
                explode_bomb_at(s, i, ub);
                break;
            }
        }
    }
    if (s->board[y][x] != POM_RIGID) {
        int old = s->board[y][x];
        int was_wood = is_wood(old);
        s->board[y][x] = POM_FLAMES + signature;
        if (was_wood)
            s->board[y][x] += old & 3;
        return !was_wood;
    }
    return 0;
}

static void spawn_flame(PomState *s, int x, int y, int strength, uint32_t *ub) /* bboard.cpp:198-263 */
{
    PomFlame *f = &s->flames.queue[(s->flames.index + s->flames.count) % QN];
    f->x = x;
    f->y = y;
    f->strength = strength;
    f->timeLeft = POM_FLAME_LIFETIME;
    int signature = (uint16_t)((x + N * y) << 3);
    s->flames.count++;

    if (s->board[y][x] >= POM_AGENT0)
        kill_agent(s, s->board[y][x] - POM_AGENT0, ub);
    s->board[y][x] = POM_FLAMES + signature;

    for (int i = 1; i <= strength; i++) { /* +x */
        if (x + i >= N) break;
        if (!spawn_flame_item(s, x + i, y, signature, ub)) break;
    }
    for (int i = 1; i <= strength; i++) { /* -x */
        if (x - i < 0) break;
        if (!spawn_flame_item(s, x - i, y, signature, ub)) break;
    }
    for (int i = 1; i <= strength; i++) { /* +y */
        if (y + i >= N) break;
        if (!spawn_flame_item(s, x, y + i, signature, ub)) break;
    }
    for (int i = 1; i <= strength; i++) { /* -y */
        if (y - i < 0) break;
        if (!spawn_flame_item(s, x, y - i, signature, ub)) break;
    }
}

static void pop_flame(PomState *s) /* bboard.cpp:148-180 */
{
    const PomFlame *f = &s->flames.queue[fq(s, 0)];
    int st = f->strength, x = f->x, y = f->y;
    int signature = (uint16_t)(x + N * y);
    for (int i = -st; i <= st; i++) {
        if (!oob(x + i, y) && is_flame(s->board[y][x + i])) {
            int b = s->board[y][x + i];
            if (flame_id(b) == signature)
                s->board[y][x + i] = flag_item(b & 3);
        }
        if (!oob(x, y + i) && is_flame(s->board[y + i][x])) {
            int b = s->board[y + i][x];
            if (flame_id(b) == signature)
                s->board[y + i][x] = flag_item(b & 3);
        }
    }
    s->flames.index = (s->flames.index + 1) % QN;
    s->flames.count--;
}

static void tick_flames(PomState *s) /* step_utility.cpp:208-222 */
{
    for (int i = 0; i < s->flames.count; i++)
        s->flames.queue[fq(s, i)].timeLeft--;
    int n = s->flames.count;
    for (int i = 0; i < n; i++)
        if (s->flames.queue[fq(s, 0)].timeLeft == 0)
            pop_flame(s);
}

static void tick_bombs(PomState *s, uint32_t *ub) /* step_utility.cpp:224-245 */
{
    for (int i = 0; i < s->bombs.count; i++) {
        int q = bq(s, i);
        s->bombs.queue[q] = (int)((unsigned)s->bombs.queue[q] - (1u << 16));
    }
    int n = s->bombs.count;
    for (int i = 0; i < n && s->bombs.count > 0; i++) {
        int c = s->bombs.queue[bq(s, 0)];
        if (bmb_time(c) != 0)
            break;
        /* ExplodeTopBomb, bboard.cpp:191-196: stored strength */
        spawn_flame(s, bmb_x(c), bmb_y(c), bmb_strength(c), ub);
        /* PopBomb, bboard.cpp:93-97: reads the top again after the chain */
        int owner = bmb_id(s->bombs.queue[bq(s, 0)]);
        if (owner < POM_AGENT_COUNT) s->agents[owner].bombCount--;
        else *ub |= POM_UB_BAD_INDEX;
        s->bombs.index = (s->bombs.index + 1) % QN;
        s->bombs.count--;
    }
}

void pom_oracle_plant_bomb(void *state, int x, int y, int id, int life_time, int set_item) /* bboard.cpp:120-146 */
{
    PomState *s = (PomState *)state;
    uint32_t ub = 0;
    if (s->agents[id].bombCount >= s->agents[id].maxBombCount)
        return;
    if (s->bombs.count >= QN)
        return;
    int *b = &s->bombs.queue[(s->bombs.index + s->bombs.count) % QN];
    *b = set_field(*b, 0xF00u, (unsigned)id << 8);
    *b = set_field(*b, 0xFFu, (unsigned)x + ((unsigned)y << 4));
    *b = set_field(*b, 0xF000u, (unsigned)s->agents[id].bombStrength << 12);
    *b = set_field(*b, 0xF0000u, (unsigned)life_time << 16);
    if (set_item)
        s->board[y][x] = POM_BOMB;
    s->agents[id].bombCount++;
    s->bombs.count++;
    (void)ub;
}

static void plant_from_step(PomState *s, int id, uint32_t *ub) /* step.cpp:54 */
{
    if (s->agents[id].bombCount >= s->agents[id].maxBombCount)
        return;
    if (s->bombs.count >= QN) {
        /* the reference would wrap the queue and later overrun
         * bombDestinations[20] (step.cpp:191); refuse instead */
        *ub |= POM_UB_QUEUE_OVERFLOW;
        return;
    }
    pom_oracle_plant_bomb(s, s->agents[id].x, s->agents[id].y, id, POM_BOMB_LIFETIME + 1, 0);
}

void pom_oracle_spawn_flame(void *state, int x, int y, int strength)
{
    uint32_t ub = 0;
    spawn_flame((PomState *)state, x, y, strength, &ub);
}

/* ---- bounce-back chain: step_utility.cpp:62-128 (tail recursion as a loop) */
static void chain_reversion(PomState *s, const int32_t *moves, const Pos *dest_bombs, int agent_id, uint32_t *ub)
{
    for (int hop = 0;; hop++) {
        if (hop >= 8) {
            *ub |= POM_UB_REVERT_LOOP;
            return;
        }
        PomAgentInfo *a = &s->agents[agent_id];
        Pos o = origin_of(a->x, a->y, moves[agent_id]);
        if (oob(o.x, o.y))
            return;
        int index_origin_agent = get_agent(s, o.x, o.y);
        int bomb_dest_index = -1;
        for (int i = 0; i < s->bombs.count; i++) {
            if (dest_bombs[i].x == o.x && dest_bombs[i].y == o.y) {
                bomb_dest_index = i;
                break;
            }
        }
        a->x = o.x;
        a->y = o.y;
        s->board[o.y][o.x] = POM_AGENT0 + agent_id;

        if (index_origin_agent != -1) {
            agent_id = index_origin_agent;
            continue;
        }
        if (bomb_dest_index != -1) {
            int *b = &s->bombs.queue[bq(s, bomb_dest_index)];
            Pos bd = dest_bombs[bomb_dest_index];
            Pos ob = origin_of(bd.x, bd.y, bmb_dir(*b));
            if (ob.x == bd.x && ob.y == bd.y) {
                s->board[ob.y][ob.x] = POM_AGENT0 + agent_id;
                return;
            }
            if (oob(ob.x, ob.y)) {
                /* The bomb would be put back onto a cell outside the board: the reference writes board[y][-1] & co. there
                 * (step_utility.cpp:107-111 has no bounds test) — undefined behaviour, on this toolchain a write into the
                 * neighbouring row.  Defined fallback, as on the device: flag the tick, leave the bomb where it is. */
                *ub |= POM_UB_BAD_INDEX;
                return;
            }
            int has_agent = get_agent(s, ob.x, ob.y);
            *b = set_field(*b, 0xF00000u, 0);
            *b = set_field(*b, 0xFFu, (unsigned)ob.x + ((unsigned)ob.y << 4));
            s->board[ob.y][ob.x] = POM_BOMB;
            if (has_agent != -1) {
                agent_id = has_agent;
                continue;
            }
        }
        return;
    }
}

static int has_bomb_collision(const PomState *s, int index) /* step_utility.cpp:279-293 */
{
    int b = s->bombs.queue[bq(s, index)];
    Pos bt = bomb_desired(b);
    for (int i = index; i < s->bombs.count; i++) {
        int o = s->bombs.queue[bq(s, i)];
        Pos t = bomb_desired(o);
        if (b != o && t.x == bt.x && t.y == bt.y)
            return 1;
    }
    return 0;
}

static void resolve_bomb_collision(PomState *s, const int32_t *moves, const Pos *dest_bombs, int index, uint32_t *ub) /* step_utility.cpp:295-329 */
{
    int *b = &s->bombs.queue[bq(s, index)];
    Pos bt = bomb_desired(*b);
    int collided = 0;
    for (int i = index; i < s->bombs.count; i++) {
        int *o = &s->bombs.queue[bq(s, i)];
        Pos t = bomb_desired(*o);
        if (*b != *o && t.x == bt.x && t.y == bt.y) {
            *o = set_field(*o, 0xF00000u, 0);
            collided = 1;
        }
    }
    if (collided && bmb_dir(*b) != 0) {
        *b = set_field(*b, 0xF00000u, 0);
        int ag = get_agent(s, bmb_x(*b), bmb_y(*b));
        if (ag > -1 && moves[ag] != POM_MOVE_IDLE && moves[ag] != POM_MOVE_BOMB) {
            chain_reversion(s, moves, dest_bombs, ag, ub);
            s->board[bmb_y(*b)][bmb_x(*b)] = POM_BOMB;
        }
    }
}

/* ---- the tick: src/bboard/step.cpp:9-284 -------------------------------- */
uint32_t pom_oracle_step(void *state, const int32_t *moves)
{
    PomState *s = (PomState *)state;
    uint32_t ub = 0;

    tick_flames(s); /* step.cpp:15 */

    /* step.cpp:21-26 with step_utility.cpp:130-170; dead agents take part */
    Pos old_pos[POM_AGENT_COUNT], dest[POM_AGENT_COUNT];
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        old_pos[i].x = s->agents[i].x;
        old_pos[i].y = s->agents[i].y;
        dest[i] = desired(s->agents[i].x, s->agents[i].y, moves[i]);
    }
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        for (int j = i; j < POM_AGENT_COUNT; j++) {
            if (dest[i].x == s->agents[j].x && dest[i].y == s->agents[j].y &&
                dest[j].x == s->agents[i].x && dest[j].y == s->agents[i].y) {
                dest[i].x = s->agents[i].x;
                dest[i].y = s->agents[i].y;
                dest[j].x = s->agents[j].x;
                dest[j].y = s->agents[j].y;
            }
        }
    }

    /* ResolveDependencies, step_utility.cpp:172-205 */
    int dependency[POM_AGENT_COUNT] = { -1, -1, -1, -1 };
    int roots[POM_AGENT_COUNT] = { -1, -1, -1, -1 };
    int root_number = 0;
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        if (s->agents[i].dead) {
            roots[root_number++] = i;
            continue;
        }
        int is_root = 1;
        for (int j = 0; j < POM_AGENT_COUNT; j++) {
            if (i == j || s->agents[j].dead) continue;
            if (dest[i].x == s->agents[j].x && dest[i].y == s->agents[j].y) {
                dependency[j] = i;
                is_root = 0;
                break;
            }
        }
        if (is_root)
            roots[root_number++] = i;
    }
    const int ouroboros = root_number == 0;

    /* agent loop, step.cpp:35-185 */
    int root_idx = 0;
    int i = ouroboros ? 0 : roots[0];
    for (int n = 0; n < POM_AGENT_COUNT; n++, i = dependency[i]) {
        if (i == -1) {
            root_idx++;
            if (root_idx >= POM_AGENT_COUNT || roots[root_idx] == -1) {
                /* reference reads moves[-1]/agents[-1] here (SURVEY Q-UB1) */
                ub |= POM_UB_LOST_AGENT;
                break;
            }
            i = roots[root_idx];
        }
        const int m = moves[i];
        PomAgentInfo *a = &s->agents[i];
        if (a->dead || m == POM_MOVE_IDLE)
            continue;
        if (m == POM_MOVE_BOMB) {
            plant_from_step(s, i, &ub);
            continue;
        }
        int x = a->x, y = a->y;
        Pos d = dest[i];
        if (oob(d.x, d.y))
            continue;
        int item = s->board[d.y][d.x];
        if (ouroboros && has_bomb(s, d.x, d.y)) /* step.cpp:71-82 */
            item = POM_BOMB;

        if (is_flame(item)) { /* step.cpp:84-99 */
            kill_agent(s, i, &ub);
            if (s->board[y][x] == POM_AGENT0 + i)
                s->board[y][x] = has_bomb(s, x, y) ? POM_BOMB : POM_PASSAGE;
            continue;
        }
        /* HasDPCollision, step_utility.cpp:264-277 */
        int collide = 0;
        for (int j = 0; j < POM_AGENT_COUNT; j++) {
            if (j == i || s->agents[j].dead) continue;
            if (dest[i].x == dest[j].x && dest[i].y == dest[j].y) {
                collide = 1;
                break;
            }
        }
        if (collide)
            continue;

        if (is_powerup(item)) { /* ConsumePowerup, step_utility.cpp:247-262 */
            if (item == POM_EXTRABOMB) a->maxBombCount++;
            else if (item == POM_INCRRANGE) a->bombStrength++;
            else if (item == POM_KICK) a->canKick = 1;
            item = POM_PASSAGE;
        }

        if (item == POM_PASSAGE || (ouroboros && item >= POM_AGENT0)) { /* step.cpp:120-140 */
            if (s->board[y][x] == POM_AGENT0 + i)
                s->board[y][x] = has_bomb(s, x, y) ? POM_BOMB : POM_PASSAGE;
            s->board[d.y][d.x] = POM_AGENT0 + i;
            a->x = d.x;
            a->y = d.y;
        } else if (item == POM_BOMB) { /* step.cpp:147-184: kicker and non-kicker both step on */
            s->board[y][x] = has_bomb(s, x, y) ? POM_BOMB : POM_PASSAGE;
            s->board[d.y][d.x] = POM_AGENT0 + i;
            a->x = d.x;
            a->y = d.y;
            if (a->canKick) {
                int bi = get_bomb_index(s, d.x, d.y);
                if (bi < 0) {
                    ub |= POM_UB_NULL_BOMB; /* step.cpp:167 derefs nullptr */
                } else {
                    int *b = &s->bombs.queue[bq(s, bi)];
                    *b = set_field(*b, 0xF00000u, (unsigned)m << 20);
                }
            }
        }
    }

    /* ResetBombFlags, step_utility.cpp:331-337 */
    for (int k = 0; k < s->bombs.count; k++) {
        int *b = &s->bombs.queue[bq(s, k)];
        *b = set_field(*b, 0xF000000u, 0);
    }
    /* FillBombDestPos, step_utility.cpp:146-152 (snapshot before loop A) */
    Pos bomb_dest[QN];
    memset(bomb_dest, 0, sizeof bomb_dest);
    for (int k = 0; k < s->bombs.count && k < QN; k++)
        bomb_dest[k] = bomb_desired(s->bombs.queue[bq(s, k)]);

    /* bomb loop A, step.cpp:195-227 */
    for (int k = 0; k < s->bombs.count; k++) {
        int *b = &s->bombs.queue[bq(s, k)];
        int bx = bmb_x(*b), by = bmb_y(*b);
        Pos t = bomb_desired(*b);
        if (oob(t.x, t.y) || is_static_mov_block(s->board[t.y][t.x]) || is_agent(s->board[t.y][t.x])) {
            *b = set_field(*b, 0xF00000u, 0);
            int ag = get_agent(s, bx, by);
            if (ag > -1 && moves[ag] != POM_MOVE_IDLE && moves[ag] != POM_MOVE_BOMB &&
                !(s->agents[ag].x == old_pos[ag].x && s->agents[ag].y == old_pos[ag].y)) {
                chain_reversion(s, moves, bomb_dest, ag, &ub);
                if (get_agent(s, bx, by) == -1)
                    s->board[by][bx] = POM_BOMB;
            }
        }
    }

    /* bomb loop B, step.cpp:230-278 */
    for (int k = 0; k < s->bombs.count; k++) {
        int *b = &s->bombs.queue[bq(s, k)];
        if (bmb_dir(*b) == 0) {
            if (has_bomb_collision(s, k)) {
                resolve_bomb_collision(s, moves, bomb_dest, k, &ub);
                continue;
            }
        }
        int bx = bmb_x(*b), by = bmb_y(*b);
        Pos t = bomb_desired(*b);
        if (!oob(t.x, t.y) && !is_static_mov_block(s->board[t.y][t.x])) {
            if (has_bomb_collision(s, k)) {
                resolve_bomb_collision(s, moves, bomb_dest, k, &ub);
                continue;
            }
            *b = set_field(*b, 0xFFu, (unsigned)t.x + ((unsigned)t.y << 4));
            if (!has_bomb(s, bx, by) && s->board[by][bx] == POM_BOMB)
                s->board[by][bx] = POM_PASSAGE;
            int *cell = &s->board[t.y][t.x];
            if (is_walkable(*cell))
                *cell = POM_BOMB;
            else if (is_flame(*cell))
                explode_bomb_at(s, get_bomb_index(s, t.x, t.y), &ub);
        } else {
            *b = set_field(*b, 0xF00000u, 0);
        }
    }

    tick_bombs(s, &ub); /* step.cpp:283 */
    return ub;
}

/* ---- Environment::Step bookkeeping: src/bboard/environment.cpp:123-169 --- */
uint32_t pom_oracle_env_step(void *state, const int32_t *moves, PomEnvStatus *st)
{
    PomState *s = (PomState *)state;
    if (st->done) /* environment.cpp:125-128 */
        return 0;
    uint32_t ub = pom_oracle_step(s, moves);
    s->timeStep++; /* :150 */
    if (s->aliveAgents == 1) { /* :152-163 */
        st->done = 1;
        for (int i = 0; i < POM_AGENT_COUNT; i++)
            if (!s->agents[i].dead)
                st->winner = i;
    }
    if (s->aliveAgents == 0) { /* :164-168 */
        st->done = 1;
        st->draw = 1;
    }
    return ub;
}

/* ---- test-setup helpers mirroring State methods used by the reference's
 *      own tests: bboard.cpp:313-333 -------------------------------------- */
void pom_oracle_init_state(void *state) /* what std::make_unique<State>() yields */
{
    PomState *s = (PomState *)state;
    memset(s, 0, sizeof *s);
    s->aliveAgents = POM_AGENT_COUNT;
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        s->agents[i].maxBombCount = 1;
        s->agents[i].bombStrength = 1;
    }
    for (int i = 0; i < QN; i++)
        s->flames.queue[i].timeLeft = POM_FLAME_LIFETIME;
}

void pom_oracle_put_agent(void *state, int x, int y, int id) /* bboard.cpp:313-320 */
{
    PomState *s = (PomState *)state;
    s->board[y][x] = POM_AGENT0 + id;
    s->agents[id].x = x;
    s->agents[id].y = y;
}

void pom_oracle_put_agents_in_corners(void *state, int a0, int a1, int a2, int a3) /* bboard.cpp:322-333 */
{
    PomState *s = (PomState *)state;
    s->board[0][0] = POM_AGENT0 + a0;
    s->board[0][N - 1] = POM_AGENT0 + a1;
    s->board[N - 1][N - 1] = POM_AGENT0 + a2;
    s->board[N - 1][0] = POM_AGENT0 + a3;
    s->agents[a1].x = s->agents[a2].x = N - 1;
    s->agents[a2].y = s->agents[a3].y = N - 1;
}

void pom_oracle_kill(void *state, int id)
{
    uint32_t ub = 0;
    kill_agent((PomState *)state, id, &ub);
}

/* ---- the helpers pinned by unit_test/bboard/step_utility_test.cpp, exported for the
 *      restated [step utilities] tests (same code the tick above runs inline) -------------- */
void pom_oracle_dest_pos(const void *state, const int32_t *moves, int32_t *out_xy) /* FillDestPos, step_utility.cpp:138-144 */
{
    const PomState *s = (const PomState *)state;
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        Pos p = desired(s->agents[i].x, s->agents[i].y, moves[i]);
        out_xy[2 * i] = p.x;
        out_xy[2 * i + 1] = p.y;
    }
}

void pom_oracle_fix_switch_move(const void *state, int32_t *xy) /* FixSwitchMove, step_utility.cpp:154-170 */
{
    const PomState *s = (const PomState *)state;
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        for (int j = i; j < POM_AGENT_COUNT; j++) {
            if (xy[2 * i] == s->agents[j].x && xy[2 * i + 1] == s->agents[j].y &&
                xy[2 * j] == s->agents[i].x && xy[2 * j + 1] == s->agents[i].y) {
                xy[2 * i] = s->agents[i].x;
                xy[2 * i + 1] = s->agents[i].y;
                xy[2 * j] = s->agents[j].x;
                xy[2 * j + 1] = s->agents[j].y;
            }
        }
    }
}

int pom_oracle_resolve_dependencies(const void *state, const int32_t *xy, int32_t *dependency, int32_t *chain) /* step_utility.cpp:172-205 */
{
    const PomState *s = (const PomState *)state;
    int root_count = 0;
    for (int i = 0; i < POM_AGENT_COUNT; i++) {
        if (s->agents[i].dead) {
            chain[root_count++] = i;
            continue;
        }
        int is_root = 1;
        for (int j = 0; j < POM_AGENT_COUNT; j++) {
            if (i == j || s->agents[j].dead) continue;
            if (xy[2 * i] == s->agents[j].x && xy[2 * i + 1] == s->agents[j].y) {
                dependency[j] = i;
                is_root = 0;
                break;
            }
        }
        if (is_root)
            chain[root_count++] = i;
    }
    return root_count;
}

/* bounded batch driver for the CPU baseline leg of bench.py: steps `n` envs
 * `ticks` times with the same counter-based move stream as the device
 * (pom_rng.h) and the same auto-reset rule; returns env-steps executed */
#include "pom_rng.h"
int64_t pom_oracle_run_random(void *states, const void *initial, int n, int ticks, uint64_t seed, int first_env,
                              int tick0, int dist, int max_steps)
{
    PomState *s = (PomState *)states;
    const PomState *init = (const PomState *)initial;
    int64_t steps = 0;
    for (int t = 0; t < ticks; t++) {
        for (int e = 0; e < n; e++) {
            PomState *st = &s[e];
            int done = st->aliveAgents <= 1 || (max_steps > 0 && st->timeStep >= max_steps);
            if (done)
                *st = init[e];
            int32_t mv[4];
            pom_rng_moves(seed, (uint32_t)(first_env + e), (uint32_t)(tick0 + t), dist, mv);
            pom_oracle_step(st, mv);
            st->timeStep++;
            steps++;
        }
    }
    return steps;
}
