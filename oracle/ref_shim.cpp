/*
 * ref_shim.cpp — TEST INFRASTRUCTURE.  A C-ABI window onto the UNMODIFIED
 * reference, compiled from the sources where they lie under /root/reference
 * (never copied into this repo) into oracle/_ref/libpomref.so by oracle/Makefile.
 * Used in this container only: to validate oracle/pom_oracle.c (fuzz_diff) and
 * to generate the golden vectors under tests/golden/ (scripts/gen_golden.py).
 * /root/reference does not exist on the GPU box; nothing there needs this file.
 */
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
