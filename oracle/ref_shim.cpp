/*
 * ref_shim.cpp — TEST INFRASTRUCTURE.  A C-ABI window onto the UNMODIFIED
 * reference, compiled from the sources where they lie under /root/reference
 * (never copied into this repo) into oracle/_ref/libpomref.so by oracle/Makefile.
 * Used in this container only: to validate oracle/pom_oracle.c (fuzz_diff) and
 * to generate the golden vectors under tests/golden/ (tests/golden/gen_golden.py).
 * /root/reference does not exist on the GPU box; nothing there needs this file.
 */
#include <csetjmp>
#include <csignal>
#include <cstring>
#include <new>

#include "bboard.hpp"
#include "step_utility.hpp"

using namespace bboard;

extern "C" {

int ref_state_size() { return int(sizeof(State)); }

/* what std::make_unique<State>() produces (value-init + default member initialisers) */
void ref_init_state(void* p)
{
    std::memset(p, 0, sizeof(State));
    new (p) State();
}

/*
 * bboard::Step with a padded move array: the reference reads moves[-1] when an
 * agent is lost in dependency resolution (step.cpp:36-46, SURVEY Q-UB1); the
 * pad makes that read a deterministic IDLE.
 */
void ref_step(void* state, const int* moves)
{
    Move buf[6] = {Move::IDLE, Move::IDLE, Move::IDLE, Move::IDLE, Move::IDLE, Move::IDLE};
    for (int i = 0; i < 4; i++) buf[1 + i] = Move(moves[i]);
    Step(static_cast<State*>(state), buf + 1);
}

void ref_put_agent(void* s, int x, int y, int id) { static_cast<State*>(s)->PutAgent(x, y, id); }
void ref_put_agents_in_corners(void* s, int a0, int a1, int a2, int a3)
{
    static_cast<State*>(s)->PutAgentsInCorners(a0, a1, a2, a3);
}
void ref_kill(void* s, int id) { static_cast<State*>(s)->Kill(id); }
void ref_put_item(void* s, int x, int y, int item) { static_cast<State*>(s)->board[y][x] = item; }
void ref_plant_bomb(void* s, int x, int y, int id, int lifeTime, int setItem)
{
    static_cast<State*>(s)->PlantBombModifiedLife(x, y, id, lifeTime, setItem != 0);
}
void ref_spawn_flame(void* s, int x, int y, int strength) { static_cast<State*>(s)->SpawnFlame(x, y, strength); }
void ref_set_bomb_direction(void* s, int queueOffset, int dir)
{
    SetBombDirection(static_cast<State*>(s)->bombs[queueOffset], Direction(dir));
}

/* InitBoardItems (bboard.cpp:346-382) on a fresh State.  It reads one queue slot past the woods it collected (idxSample draws from
 * [0, count], bboard.cpp:367-372) — an uninitialised stack word: callers run this in a child process (tests/golden/gen_boardgen_stats.py) */
static sigjmp_buf g_fault_jmp;
static void on_fault(int) { siglongjmp(g_fault_jmp, 1); }
/* returns 1 if the call faulted (SIGSEGV / SIGBUS inside the flag pass): the cell kinds, drawn before that pass, are in the
 * State then, the flags are not to be trusted; the caller should not reuse the process */
int ref_init_board_items(void* p, int seed)
{
    std::memset(p, 0, sizeof(State));
    new (p) State();
    struct sigaction sa, old_segv, old_bus;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fault;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    int faulted = 0;
    if (sigsetjmp(g_fault_jmp, 1) == 0) InitBoardItems(*static_cast<State*>(p), seed);
    else faulted = 1;
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return faulted;
}

/* step utilities pinned by unit_test/bboard/step_utility_test.cpp */
void ref_fill_dest_pos(void* s, const int* moves, int* outXY)
{
    Move m[4];
    Position p[4];
    for (int i = 0; i < 4; i++) m[i] = Move(moves[i]);
    util::FillDestPos(static_cast<State*>(s), m, p);
    for (int i = 0; i < 4; i++) { outXY[2 * i] = p[i].x; outXY[2 * i + 1] = p[i].y; }
}
void ref_fix_switch_move(void* s, int* xy)
{
    Position p[4];
    for (int i = 0; i < 4; i++) p[i] = {xy[2 * i], xy[2 * i + 1]};
    util::FixSwitchMove(static_cast<State*>(s), p);
    for (int i = 0; i < 4; i++) { xy[2 * i] = p[i].x; xy[2 * i + 1] = p[i].y; }
}
int ref_resolve_dependencies(void* s, const int* xy, int* dependency, int* chain)
{
    Position p[4];
    for (int i = 0; i < 4; i++) p[i] = {xy[2 * i], xy[2 * i + 1]};
    return util::ResolveDependencies(static_cast<State*>(s), p, dependency, chain);
}

}

/* ---- SimpleAgent: the reference's own policy object behind a C window ------------------------------
 * src/agents/simple_agent.cpp + src/bboard/strategy.cpp compiled unmodified.  The agent draws from a std::mt19937_64
 * seeded by random_device (simple_agent.cpp:17-22); for a reproducible comparison the shim reseeds that public member
 * and, before each act(), peeks the value the next intDist(rng) call WOULD return on a copy of the generator — act() makes
 * at most one draw — so the restatement can be given the identical draw as an input. */
#include "agents.hpp"

extern "C" {

void* ref_simple_new(int id, unsigned long long seed)
{
    agents::SimpleAgent* a = new agents::SimpleAgent();
    a->id = id;
    a->rng = std::mt19937_64(seed);
    return a;
}
void ref_simple_delete(void* p) { delete static_cast<agents::SimpleAgent*>(p); }
int ref_simple_peek_draw(void* p)
{
    agents::SimpleAgent* a = static_cast<agents::SimpleAgent*>(p);
    std::mt19937_64 r2 = a->rng;
    std::uniform_int_distribution<int> d2 = a->intDist;
    return d2(r2);
}
int ref_simple_act(void* p, const void* state) { return int(static_cast<agents::SimpleAgent*>(p)->act(static_cast<const State*>(state))); }
/* agent memory that survives between act() calls: recentPositions (4 x {x,y}, index, count), moveQueue (4 moves, index, count) */
void ref_simple_memory(void* p, int* out16)
{
    agents::SimpleAgent* a = static_cast<agents::SimpleAgent*>(p);
    for (int i = 0; i < 4; i++) {
        out16[2 * i] = a->recentPositions.queue[i].x;
        out16[2 * i + 1] = a->recentPositions.queue[i].y;
    }
    out16[8] = a->recentPositions.index;
    out16[9] = a->recentPositions.count;
    for (int i = 0; i < 4; i++) out16[10 + i] = int(a->moveQueue.queue[i]);
    out16[14] = a->moveQueue.index;
    out16[15] = a->moveQueue.count;
}
/* the strategy helpers the reference's own [strategy] tests call (unit_test/bboard/strategy_test.cpp): IsAdjacentEnemy, and FillRMap
 * with what MoveTowardsPosition makes of it — the raw map (distance | predecessor << 16 per cell) and, for every reachable cell
 * other than the source, the first move of the path to it (-1 elsewhere: the reference spins forever when asked for those) */
int ref_is_adjacent_enemy(const void* state, int id, int distance)
{
    return strategy::IsAdjacentEnemy(*static_cast<const State*>(state), id, distance) ? 1 : 0;
}
void ref_fill_rmap(const void* state, int id, int* map121, int* move_to121)
{
    strategy::RMap r;
    strategy::FillRMap(*static_cast<const State*>(state), r, id);
    for (int y = 0; y < BOARD_SIZE; y++)
        for (int x = 0; x < BOARD_SIZE; x++) {
            map121[y * BOARD_SIZE + x] = r.map[y][x];
            const bool ask = strategy::IsReachable(r, x, y) && !(x == r.source.x && y == r.source.y);
            move_to121[y * BOARD_SIZE + x] = ask ? int(strategy::MoveTowardsPosition(r, {x, y})) : -1;
        }
}
void ref_simple_set_memory(void* p, const int* in16)
{
    agents::SimpleAgent* a = static_cast<agents::SimpleAgent*>(p);
    for (int i = 0; i < 4; i++) a->recentPositions.queue[i] = {in16[2 * i], in16[2 * i + 1]};
    a->recentPositions.index = in16[8];
    a->recentPositions.count = in16[9];
    for (int i = 0; i < 4; i++) a->moveQueue.queue[i] = Move(in16[10 + i]);
    a->moveQueue.index = in16[14];
    a->moveQueue.count = in16[15];
}

}

/* ---- Environment::Step (src/bboard/environment.cpp:123-169) behind a C window: the unmodified Environment
 * with four agents that play back a move handed in before each step and note that they were asked.  Pins the restatement's
 * bookkeeping (pom_oracle_env_step): timeStep++, finished / winner / draw, finished games not stepped, act() only asked of
 * live agents.  The reference leaves the Move entry of a dead agent uninitialised (environment.cpp:130) and passes a 4-entry
 * array to Step, which reads moves[-1] on lost-agent ticks (Q-UB1): the caller (fuzz_env.c, gen_golden.py) never steps it on
 * a tick where either could matter. */
namespace {
struct ScriptedAgent : Agent {
    Move next = Move::IDLE;
    bool asked = false;
    Move act(const State*) override
    {
        asked = true;
        return next;
    }
};
struct RefEnvGame {
    Environment env;
    ScriptedAgent a[4];
};
}  // namespace

extern "C" {

void* ref_env_new(const void* start_state)
{
    RefEnvGame* g = new RefEnvGame();
    g->env.MakeGame({&g->a[0], &g->a[1], &g->a[2], &g->a[3]}, false); /* sets hasStarted; its board is replaced right away */
    std::memcpy(&g->env.GetState(), start_state, sizeof(State));
    return g;
}
void ref_env_delete(void* p) { delete static_cast<RefEnvGame*>(p); }
/* one Environment::Step(false); returns the bit mask of the agents whose act() was called */
int ref_env_step(void* p, const int* moves, void* state_out, int* done, int* winner, int* draw)
{
    RefEnvGame* g = static_cast<RefEnvGame*>(p);
    for (int i = 0; i < 4; i++) {
        g->a[i].next = Move(moves[i]);
        g->a[i].asked = false;
    }
    g->env.Step(false);
    std::memcpy(state_out, &g->env.GetState(), sizeof(State));
    *done = g->env.IsDone();
    *winner = g->env.GetWinner();
    *draw = g->env.IsDraw();
    int mask = 0;
    for (int i = 0; i < 4; i++) mask |= int(g->a[i].asked) << i;
    return mask;
}
int ref_env_last_move(void* p, int agent) { return int(static_cast<RefEnvGame*>(p)->env.GetLastMove(agent)); }

}
