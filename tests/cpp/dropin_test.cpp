// dropin_test.cpp — code written against pomcpp's API (make_unique<State>, State methods, bboard::Step, an
// Agent subclass reading the State) compiled against include/pom_bboard.hpp and run on the GPU through the
// C-ABI.  The first block is the reference's "Bomb Mechanics / Standard Bomb Laying" and "Bomb Kick Mechanics /
// One Agent - One Bomb" sections (unit_test/bboard/board_logic.cpp:247-257, 486-500) as they are written there.
#include <cstdio>
#include <memory>
#include <random>

#include "pom_bboard.hpp"

#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::printf("REQUIRE failed line %d: %s\n", __LINE__, #c);   \
            return 1;                                                     \
        }                                                                 \
    } while (0)

struct HarmlessLike : bboard::Agent {  // agents::HarmlessAgent's distribution, src/agents/basic_agents.cpp:28-38
    std::mt19937_64 rng{42};
    std::uniform_int_distribution<int> d{0, 4};
    bboard::Move act(const bboard::State*) override { return static_cast<bboard::Move>(d(rng)); }
};

struct BombWatcher : bboard::Agent {  // reads the State the way simple_agent.cpp / strategy.cpp do
    bboard::Move act(const bboard::State* s) override
    {
        const bboard::AgentInfo& me = s->agents[id];
        for (int i = 0; i < s->bombs.count; i++) {
            const bboard::Bomb b = s->bombs[i];
            if (bboard::BMB_POS_X(b) == me.x && bboard::BMB_POS_Y(b) == me.y) {  // standing on a bomb: walk away
                bboard::Position p = bboard::util::DesiredPosition(me.x, me.y, bboard::Move::DOWN);
                return bboard::util::IsOutOfBounds(p) ? bboard::Move::UP : bboard::Move::DOWN;
            }
        }
        return me.bombCount < me.maxBombCount ? bboard::Move::BOMB : bboard::Move::IDLE;
    }
};

int main()
{
    {
        auto s = std::make_unique<bboard::State>();
        bboard::Move id = bboard::Move::IDLE;
        bboard::Move m[4] = {id, id, id, id};
        s->PutAgentsInCorners(0, 1, 2, 3);
        m[0] = bboard::Move::BOMB;
        bboard::Step(s.get(), m);
        REQUIRE(s->board[0][0] == bboard::Item::AGENT0);
        m[0] = bboard::Move::DOWN;
        bboard::Step(s.get(), m);
        REQUIRE(s->board[0][0] == bboard::Item::BOMB);
    }
    {
        auto s = std::make_unique<bboard::State>();
        bboard::Move id = bboard::Move::IDLE;
        bboard::Move m[4] = {id, id, id, id};
        s->PutAgent(0, 1, 0);
        s->agents[0].canKick = true;
        s->PlantBomb(1, 1, 0, true);
        s->agents[0].maxBombCount = bboard::MAX_BOMBS_PER_AGENT;
        m[0] = bboard::Move::RIGHT;
        s->Kill(1, 2, 3);
        bboard::Step(s.get(), m);
        REQUIRE(s->agents[0].x == 1 && s->agents[0].y == 1 && s->board[1][1] == bboard::Item::AGENT0);
        REQUIRE(s->board[1][2] == bboard::Item::BOMB);
        for (int i = 0; i < 4; i++) {
            REQUIRE(s->board[1][2 + i] == bboard::Item::BOMB);
            bboard::Step(s.get(), m);
            m[0] = bboard::Move::IDLE;
        }
    }
    {
        const int n = 128;
        std::vector<bboard::State> start(n);
        for (auto& s : start) s.PutAgentsInCorners(0, 1, 2, 3);
        bboard::BatchEnvironment env(n);
        env.MakeGame(start.data());
        HarmlessLike h;
        BombWatcher w;
        std::array<bboard::Agent*, 4> agents = {&w, &h, &h, &w};
        for (int t = 0; t < 40; t++) env.Step(agents);
        std::vector<int32_t> done, winner, draw;
        env.Status(done, winner, draw);
        const bboard::State* st = env.GetStates();
        int finished = 0, blown_up = 0;
        for (int e = 0; e < n; e++) {
            finished += done[e];
            blown_up += st[e].agents[0].dead + st[e].agents[3].dead;
            REQUIRE(st[e].timeStep <= 40 && st[e].timeStep >= 1);
            REQUIRE(done[e] == (st[e].aliveAgents <= 1));
            if (done[e] && !draw[e]) REQUIRE(!st[e].agents[winner[e]].dead);
        }
        for (int e = 0; e < 4; e++) { /* the per-game queries agree with the batch ones */
            REQUIRE(env.IsDone(e) == (done[e] != 0) && env.IsDraw(e) == (draw[e] != 0) && env.GetWinner(e) == winner[e]);
            REQUIRE(env.GetState(e).timeStep == st[e].timeStep && env.GetState(e).aliveAgents == st[e].aliveAgents);
        }
        REQUIRE(blown_up > n);  // the watchers planted strength-1 bombs and stepped only one cell away
        std::printf("batch: %d of %d games finished, %d watchers caught by their own bombs\n", finished, n, blown_up);
    }
    {
        /* no host in the loop: boards drawn on the device, SimpleAgents and tick on the device, a fresh board per game */
        const int n = 256;
        bboard::BatchEnvironment env(n, 0, /*autoReset*/ true, /*maxSteps*/ 60, /*freshBoards*/ true, /*boardSeed*/ 5);
        env.MakeGame(uint64_t(5));
        const bboard::State* st = env.GetStates();
        int woods = 0, rigid = 0;
        for (int e = 0; e < n; e++) {
            REQUIRE(st[e].board[0][0] == bboard::Item::AGENT0 && st[e].board[10][10] == bboard::Item::AGENT0 + 2);
            REQUIRE(st[e].aliveAgents == 4 && st[e].timeStep == 0 && st[e].agents[2].x == 10 && st[e].agents[2].y == 10);
            for (int y = 0; y < 11; y++)
                for (int x = 0; x < 11; x++) {
                    woods += bboard::IS_WOOD(st[e].board[y][x]);
                    rigid += st[e].board[y][x] == bboard::Item::RIGID;
                }
        }
        REQUIRE(woods > n * 121 / 9 && woods < n * 121 / 5 && rigid > n * 121 / 9 && rigid < n * 121 / 5);  // 1/7 each
        env.StepSimpleAgents(/*seed*/ 3, /*ticks*/ 150);
        st = env.GetStates();
        uint32_t episodes[256];
        REQUIRE(pom_batch_episodes(env.Handle(), 0, n, episodes) == POM_OK);
        int restarted = 0;
        for (int e = 0; e < n; e++) {
            restarted += episodes[e] >= 2;
            REQUIRE(st[e].timeStep <= 60);
        }
        REQUIRE(restarted == n);  // 150 ticks under a 60-tick cap: every game is at least in its third round
        std::printf("generated boards: %d woods, %d rigid cells in %d games; all restarted on fresh boards\n", woods, rigid, n);
    }
    std::printf("dropin ok\n");
    return 0;
}
