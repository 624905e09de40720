// env_game.cpp — games played through bboard::Environment (include/pom_bboard.hpp), the reference's own game-loop
// surface (include/bboard.hpp:541-644): MakeGame / Step / GetState / IsDone / IsDraw / GetWinner / GetLastMove /
// SetStepListener.  Writes a trace (start state, then per step: the moves the environment used, the state after,
// done / winner / draw) that tests/test_cpp_dropin.py replays through the oracle's Environment::Step restatement.
//
// Two builds: with -DPOM_WITH_REFERENCE_AGENTS the four players are the reference's agents::SimpleAgent, linked from its
// UNMODIFIED sources (build container only; the binary travels to the GPU box under oracle/_ref/); without it, small
// scripted agents that read the State the way the reference's agents do.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bboard.hpp"
#ifdef POM_WITH_REFERENCE_AGENTS
#include "agents.hpp"
#endif

namespace {

struct HashAgent : bboard::Agent {  // a deterministic function of what it sees: position, tick, neighbours
    bboard::Move act(const bboard::State* s) override
    {
        const bboard::AgentInfo& me = s->agents[id];
        unsigned h = unsigned(me.x * 31 + me.y * 131 + s->timeStep * 2654435761u + id * 97u + s->bombs.count * 7u);
        h ^= h >> 13;
        h *= 0x9E3779B1u;
        h ^= h >> 15;
        bboard::Move m = bboard::Move(h % 6u);
        if (m != bboard::Move::BOMB && m != bboard::Move::IDLE) {  // prefer a walkable target, like a sane agent would
            bboard::Position p = bboard::util::DesiredPosition(me.x, me.y, m);
            if (bboard::util::IsOutOfBounds(p) || !bboard::IS_WALKABLE(s->board[p.y][p.x])) m = bboard::Move((h >> 8) % 6u);
        }
        return m;
    }
};

void put(std::FILE* f, const void* p, size_t n)
{
    if (std::fwrite(p, 1, n, f) != n) std::exit(3);
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s trace.bin games [max_steps] | --board out.bin [seed]\n", argv[0]);
        return 2;
    }
    if (std::strcmp(argv[1], "--board") == 0) {  // no device involved: InitState of (seed) as 1004 bytes
        std::FILE* b = std::fopen(argv[2], "wb");
        if (!b) return 2;
        auto s = std::make_unique<bboard::State>();
        bboard::InitBoardItems(*s, argc > 3 ? std::atoi(argv[3]) : 0x1337);
        s->PutAgentsInCorners(0, 1, 2, 3);
        put(b, s.get(), sizeof(bboard::State));
        std::fclose(b);
        return 0;
    }
    std::FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 2;
    const int games = std::atoi(argv[2]), maxSteps = argc > 3 ? std::atoi(argv[3]) : 300;
    int listened = 0;
    for (int g = 0; g < games; ++g) {
#ifdef POM_WITH_REFERENCE_AGENTS
        agents::SimpleAgent a[4];
#else
        HashAgent a[4];
#endif
        bboard::Environment env;
        env.Step();  // before MakeGame: a no-op (environment.cpp:125)
        env.MakeGame({&a[0], &a[1], &a[2], &a[3]}, false);  // (true shuffles the corners from std::random_device: not for a test)
        env.SetStepListener([&listened](const bboard::Environment&) { ++listened; });
        for (bboard::AgentInfo& info : env.GetState().agents) info.canKick = (g % 3) == 0;  // as main.cpp:18-21 does
        if (g % 4 != 0) {  // other boards, other corners
            bboard::InitBoardItems(env.GetState(), 1000 + g);
            for (bboard::AgentInfo& info : env.GetState().agents) info.x = info.y = 0;  // PutAgentsInCorners relies on zeroed positions
            env.GetState().PutAgentsInCorners(g % 4, (g + 1) % 4, (g + 2) % 4, (g + 3) % 4);
        }
        const bboard::State start = env.GetState();
        const int32_t header[2] = {0x504F4D45, g};
        put(f, header, sizeof header);
        put(f, &start, sizeof start);
        int steps = 0;
        while (!env.IsDone() && env.GetState().timeStep < maxSteps) {
            bool was_dead[4];
            for (int i = 0; i < 4; ++i) was_dead[i] = env.GetState().agents[i].dead;
            env.Step(false);
            int32_t rec[8] = {1, 0, 0, 0, 0, env.IsDone(), env.GetWinner(), env.IsDraw()};
            for (int i = 0; i < 4; ++i) rec[1 + i] = was_dead[i] ? 0 : int32_t(env.GetLastMove(i));
            put(f, rec, sizeof rec);
            put(f, &env.GetState(), sizeof(bboard::State));
            ++steps;
        }
        const bboard::State before = env.GetState();
        if (env.IsDone()) env.Step(false);  // a finished game is not stepped (environment.cpp:125-128)
        if (std::memcmp(&before, &env.GetState(), sizeof before) != 0) {
            std::printf("a finished game was stepped\n");
            return 1;
        }
        const int32_t tail[8] = {2, steps, 0, 0, 0, env.IsDone(), env.GetWinner(), env.IsDraw()};
        put(f, tail, sizeof tail);
    }
    std::fclose(f);
    std::printf("env games ok: %d\n", games);
    return 0;
}
