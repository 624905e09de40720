// pom_bboard.hpp — C++ host surface over the C-ABI (pom_batch.h) that keeps the reference's names for the
// Step path, so code written against pomcpp's bboard.hpp — agents that read a `const State*`, game loops that
// call `bboard::Step(State*, Move*)` — compiles against this header unchanged and runs the tick on the MI355X.
//
// What is mirrored (file:line in /root/reference/include/bboard.hpp): the constants :15-27, Move :35-43,
// Direction :45-52, Item and its predicates :54-109, FixedQueue :115-188, Position :192-201, AgentInfo :228-245,
// the bit-packed Bomb and its accessors :261-335, Flame :342-347, State :356-511 (identical 1004-byte layout,
// checked below), Agent :517-533, Environment :541-644 (src/bboard/environment.cpp:48-213), the free functions
// InitBoardItems / InitState / Step / StartGame / PrintState / PrintItem :646-689 and std::hash<Position> :693-704;
// from step_utility.hpp the helpers agents call, DesiredPosition :16, OriginPosition and IsOutOfBounds :155-166.
// The standard headers bboard.hpp pulls in (:4-10) are included here too: the reference's agents rely on them
// transitively (std::cout in simple_agent.cpp:135, strategy.cpp:263).  State methods that only place things are
// host inlines; the ones that simulate (SpawnFlame, PopFlame, Explode*) live on the device path inside Step and are
// not offered as host calls.  `BatchEnvironment` is the n-game counterpart of Environment.
//
// tests/test_cpp_dropin.py compiles the reference's UNMODIFIED src/agents/*.cpp, src/bboard/strategy.cpp and
// src/main.cpp against this header (tests/cpp/shim/bboard.hpp and step_utility.hpp only include it).
//
// Not a port: there is no simulation code in this header.  Step() hands the State to libpom_batch.so.
#ifndef POM_BBOARD_HPP_
#define POM_BBOARD_HPP_

#include <algorithm>
#include <array>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pom_batch.h"
#include "pom_boardgen.h"

#include <sys/types.h>  // `uint`, which the reference's sources use unqualified (strategy.hpp:21, bboard.hpp:614)

namespace bboard {

constexpr int MOVE_COUNT = 4, AGENT_COUNT = 4, BOARD_SIZE = 11;
constexpr int BOMB_LIFETIME = 10, BOMB_DEFAULT_STRENGTH = 1, FLAME_LIFETIME = 4;
constexpr int MAX_BOMBS_PER_AGENT = 5, MAX_BOMBS = AGENT_COUNT * MAX_BOMBS_PER_AGENT;

enum class Move : int { IDLE = 0, UP, DOWN, LEFT, RIGHT, BOMB };
enum class Direction : int { IDLE = 0, UP, DOWN, LEFT, RIGHT };

enum Item : int {
    PASSAGE = POM_PASSAGE, RIGID = POM_RIGID, WOOD = POM_WOOD, BOMB = POM_BOMB, FLAMES = POM_FLAMES, FOG = POM_FOG,
    EXTRABOMB = POM_EXTRABOMB, INCRRANGE = POM_INCRRANGE, KICK = POM_KICK, AGENTDUMMY = 9,
    AGENT0 = POM_AGENT0, AGENT1, AGENT2, AGENT3
};

constexpr bool IS_WOOD(int v) { return (v >> 8) == 2; }
constexpr bool IS_POWERUP(int v) { return v > 5 && v < 9; }
constexpr bool IS_WALKABLE(int v) { return v == 0 || IS_POWERUP(v); }
constexpr bool IS_FLAME(int v) { return (v >> 16) == 4; }
constexpr bool IS_AGENT(int v) { return v >= (1 << 24); }
constexpr bool IS_STATIC_MOV_BLOCK(int v) { return v == 1 || IS_WOOD(v) || IS_POWERUP(v); }
constexpr int FLAME_ID(int v) { return (v & 0xFFFF) >> 3; }
constexpr int FLAME_POWFLAG(int v) { return v & 3; }
constexpr int WOOD_POWFLAG(int v) { return v & 3; }

template <typename T, int TSize>
struct FixedQueue {  // circular buffer; element k of the live range is queue[(index + k) % TSize]
    T queue[TSize];
    int index = 0;
    int count = 0;

    int RemainingCapacity() const { return TSize - count; }
    T& NextPos() { return queue[(index + count) % TSize]; }
    void AddElem(const T& e) { NextPos() = e; ++count; }
    T& PopElem()
    {
        T& top = queue[index % TSize];
        index = (index + 1) % TSize;
        --count;
        return top;
    }
    void RemoveAt(int at)
    {
        for (int k = at + 1; k < count; ++k) queue[(index + k - 1) % TSize] = queue[(index + k) % TSize];
        --count;
    }
    T& operator[](int k) { return queue[(index + k) % TSize]; }
    const T& operator[](int k) const { return queue[(index + k) % TSize]; }
};

struct Position {
    int x, y;
};
inline bool operator==(const Position& a, const Position& b) { return a.x == b.x && a.y == b.y; }
inline std::ostream& operator<<(std::ostream& os, const Position& p) { return os << "(" << p.x << ", " << p.y << ")"; }

struct AgentInfo {
    int x, y;
    int bombCount = 0, maxBombCount = 1, bombStrength = BOMB_DEFAULT_STRENGTH;
    bool canKick = false, dead = false;
    Position GetPos() const { return {x, y}; }
};

// one int per bomb: x[0,4) y[4,8) id[8,12) strength[12,16) time[16,20) direction[20,24) moved[24,28)
typedef int Bomb;
constexpr int BMB_POS(Bomb b) { return b & 0xFF; }
constexpr int BMB_POS_X(Bomb b) { return b & 0xF; }
constexpr int BMB_POS_Y(Bomb b) { return (b >> 4) & 0xF; }
constexpr int BMB_ID(Bomb b) { return (b >> 8) & 0xF; }
constexpr int BMB_STRENGTH(Bomb b) { return (b >> 12) & 0xF; }
constexpr int BMB_TIME(Bomb b) { return (b >> 16) & 0xF; }
constexpr int BMB_DIR(Bomb b) { return (b >> 20) & 0xF; }
constexpr int BMB_MOVED(Bomb b) { return (b >> 24) & 0xF; }
namespace detail {
inline void put_nibbles(Bomb& b, unsigned mask, unsigned v) { b = int((unsigned(b) & ~mask) + v); }
}
inline void ReduceBombTimer(Bomb& b) { b = int(unsigned(b) - (1u << 16)); }
inline void SetBombPosition(Bomb& b, int x, int y) { detail::put_nibbles(b, 0xFFu, unsigned(x) + (unsigned(y) << 4)); }
inline void SetBombID(Bomb& b, int id) { detail::put_nibbles(b, 0xF00u, unsigned(id) << 8); }
inline void SetBombStrength(Bomb& b, int s) { detail::put_nibbles(b, 0xF000u, unsigned(s) << 12); }
inline void SetBombTime(Bomb& b, int t) { detail::put_nibbles(b, 0xF0000u, unsigned(t) << 16); }
inline void SetBombDirection(Bomb& b, Direction d) { detail::put_nibbles(b, 0xF00000u, unsigned(d) << 20); }
inline void SetBombMovedFlag(Bomb& b, bool m) { detail::put_nibbles(b, 0xF000000u, unsigned(m) << 24); }

struct Flame {
    Position position;
    int timeLeft = FLAME_LIFETIME;
    int strength;
};

struct State {
    int board[BOARD_SIZE][BOARD_SIZE];  // [y][x]
    int timeStep = 0;
    int aliveAgents = AGENT_COUNT;
    AgentInfo agents[AGENT_COUNT];
    FixedQueue<Bomb, MAX_BOMBS> bombs;
    FixedQueue<Flame, MAX_BOMBS> flames;

    int& operator[](const Position& p) { return board[p.y][p.x]; }
    void PutItem(int x, int y, Item item) { board[y][x] = item; }
    void PutAgent(int x, int y, int id)
    {
        board[y][x] = AGENT0 + id;
        agents[id].x = x;
        agents[id].y = y;
    }
    void PutAgentsInCorners(int a0, int a1, int a2, int a3)
    {
        const int e = BOARD_SIZE - 1;
        board[0][0] = AGENT0 + a0;
        board[0][e] = AGENT0 + a1;
        board[e][e] = AGENT0 + a2;
        board[e][0] = AGENT0 + a3;
        agents[a1].x = agents[a2].x = e;
        agents[a2].y = agents[a3].y = e;
    }
    void Kill(int id)
    {
        if (!agents[id].dead) {
            agents[id].dead = true;
            --aliveAgents;
        }
    }
    template <typename... Rest>
    void Kill(int id, Rest... rest)
    {
        Kill(id);
        Kill(rest...);
    }
    void PlantBombModifiedLife(int x, int y, int id, int lifeTime = BOMB_LIFETIME, bool setItem = false)
    {
        if (agents[id].bombCount >= agents[id].maxBombCount) return;
        Bomb& b = bombs.NextPos();  // the slot's other bits are deliberately left as they are
        SetBombID(b, id);
        SetBombPosition(b, x, y);
        SetBombStrength(b, agents[id].bombStrength);
        SetBombTime(b, lifeTime);
        if (setItem) board[y][x] = BOMB;
        ++agents[id].bombCount;
        ++bombs.count;
    }
    void PlantBomb(int x, int y, int id, bool setItem = false) { PlantBombModifiedLife(x, y, id, BOMB_LIFETIME, setItem); }
    int GetBombIndex(int x, int y) const
    {
        for (int k = 0; k < bombs.count; ++k)
            if (BMB_POS(bombs[k]) == x + (y << 4)) return k;
        return -1;
    }
    bool HasBomb(int x, int y) const { return GetBombIndex(x, y) >= 0; }
    Bomb* GetBomb(int x, int y)
    {
        const int k = GetBombIndex(x, y);
        return k < 0 ? nullptr : &bombs[k];
    }
    int GetAgent(int x, int y) const
    {
        for (int i = 0; i < AGENT_COUNT; ++i)
            if (!agents[i].dead && agents[i].x == x && agents[i].y == y) return i;
        return -1;
    }
    static Item FlagItem(int flag) { return flag >= 1 && flag <= 3 ? Item(EXTRABOMB + flag - 1) : PASSAGE; }
};

static_assert(sizeof(State) == POM_STATE_BYTES, "bboard::State must stay the reference's 1004 bytes");
static_assert(offsetof(State, timeStep) == 484 && offsetof(State, aliveAgents) == 488, "State layout");
static_assert(offsetof(State, agents) == 492 && sizeof(AgentInfo) == 24 && offsetof(AgentInfo, canKick) == 20, "State layout");
static_assert(offsetof(State, bombs) == 588 && offsetof(State, flames) == 676 && sizeof(Flame) == 16, "State layout");

struct Agent {  // a behaviour: State in, Move out
    virtual ~Agent() {}
    int id = -1;
    virtual Move act(const State* state) = 0;
};

struct PomException : std::runtime_error {
    int code;
    PomException(int c, const char* what) : std::runtime_error(what), code(c) {}
};
inline void pom_check(int rc)
{
    if (rc != POM_OK) throw PomException(rc, pom_last_error());
}

// One simulation tick of one State, executed on the GPU (bboard.hpp:668).  `moves` holds AGENT_COUNT entries.
inline void Step(State* state, Move* moves)
{
    static_assert(sizeof(Move) == sizeof(int32_t), "Move is an int enum");
    pom_check(pom_step(state, reinterpret_cast<const int32_t*>(moves)));
}

namespace util {
inline Position DesiredPosition(int x, int y, Move m)
{
    return {x + (m == Move::RIGHT) - (m == Move::LEFT), y + (m == Move::DOWN) - (m == Move::UP)};
}
inline Position DesiredPosition(const Bomb b) { return DesiredPosition(BMB_POS_X(b), BMB_POS_Y(b), Move(BMB_DIR(b))); }
inline Position OriginPosition(int x, int y, Move m)  // one step against the move
{
    return {x - (m == Move::RIGHT) + (m == Move::LEFT), y - (m == Move::DOWN) + (m == Move::UP)};
}
inline bool IsOutOfBounds(int x, int y) { return x < 0 || y < 0 || x >= BOARD_SIZE || y >= BOARD_SIZE; }
inline bool IsOutOfBounds(const Position& p) { return IsOutOfBounds(p.x, p.y); }
}  // namespace util

// ---- start boards (bboard.hpp:646-661).  The reference draws them from std::mt19937_64 through libstdc++'s distributions
// and reads an unwritten queue slot doing so (bboard.cpp:346-382), so its stream is not reproducible; its DISTRIBUTION is
// specified in pom_boardgen.h as a pure function of (seed, env, episode), which the device generator implements
// (pom_batch_generate).  These host inlines place the board of (seed, env 0, episode 0) the same way.
inline void InitBoardItems(State& state, int seed = 0x1337)
{
    const uint32_t key = pom_board_key(uint64_t(uint32_t(seed)), 0u, 0u);
    int* const cells = &state.board[0][0];  // cell c = y * BOARD_SIZE + x
    int woods = 0;
    for (int c = 0; c < BOARD_SIZE * BOARD_SIZE; ++c) {
        const uint32_t kind = pom_mulhi32(pom_board_draw(key, uint32_t(c)), 7u);
        cells[c] = kind == 1u ? RIGID : kind == 2u ? WOOD : PASSAGE;
        woods += kind == 2u;
    }
    int left = woods, need = (woods + 1) / 2;  // selection sampling: exactly ceil(woods / 2) woods carry a flag
    for (int c = 0; c < BOARD_SIZE * BOARD_SIZE && need > 0; ++c) {
        if (cells[c] != WOOD) continue;
        if (int(pom_mulhi32(pom_board_draw(key, uint32_t(POM_BOARD_DRAW_SELECT + c)), uint32_t(left))) < need) {
            cells[c] = WOOD + 1 + int(pom_board_draw(key, uint32_t(POM_BOARD_DRAW_FLAG + c)) >> 30);
            --need;
        }
        --left;
    }
}
inline void InitState(State* state, int a0, int a1, int a2, int a3)
{
    InitBoardItems(*state);
    state->PutAgentsInCorners(a0, a1, a2, a3);
}

// ---- plain-text rendering (bboard.hpp:678-689).  Rendering is outside this repo's scope; these exist so that
// game loops written against the reference link.  Three characters per cell, no colours.
inline std::string PrintItem(int item)
{
    if (item == PASSAGE) return " . ";
    if (item == RIGID) return "[X]";
    if (IS_WOOD(item)) return "[ ]";
    if (item == BOMB) return " o ";
    if (IS_FLAME(item)) return " * ";
    if (item == EXTRABOMB) return " +b";
    if (item == INCRRANGE) return " +r";
    if (item == KICK) return " +k";
    if (IS_AGENT(item)) return " " + std::to_string(item - AGENT0) + " ";
    return " ? ";
}
inline void PrintState(State* state, bool clearConsole = false)
{
    if (clearConsole) std::cout << "\033c";
    for (int y = 0; y < BOARD_SIZE; ++y) {
        for (int x = 0; x < BOARD_SIZE; ++x) std::cout << PrintItem(state->board[y][x]);
        if (y < AGENT_COUNT) {
            const AgentInfo& a = state->agents[y];
            std::cout << "    agent " << y << (a.dead ? " dead" : "     ") << " bombs " << a.bombCount << "/" << a.maxBombCount
                      << " range " << a.bombStrength << (a.canKick ? " kick" : "");
        }
        std::cout << "\n";
    }
    std::cout << "tick " << state->timeStep << ", " << state->bombs.count << " bombs, " << state->flames.count << " flames" << std::endl;
}

// bboard.hpp:663-676: the bare loop over a caller-owned State (every agent is asked, dead or not, bboard.cpp:396-412)
inline void StartGame(State* state, Agent* agents[AGENT_COUNT], int timeSteps)
{
    Move moves[AGENT_COUNT];
    for (int t = 0; t < timeSteps; ++t) {
        for (int j = 0; j < AGENT_COUNT; ++j) moves[j] = agents[j]->act(state);
        Step(state, moves);
        PrintState(state, true);
        std::this_thread::sleep_for(std::chrono::milliseconds(80));
    }
}

// One game with the reference's own surface (bboard.hpp:541-644, environment.cpp:48-213): the tick and Environment::Step's
// bookkeeping (timeStep++, finished / winner / draw) run on the GPU, one launch per Step (pom_env_step).  GetState() is the
// host State callers may edit between steps, as main.cpp:18-21 does: every Step hands it to the device and takes the result
// back (a loop over many games wants BatchEnvironment).  Differences, all at points where the reference is undefined: a dead
// agent's Move entry is IDLE (the reference leaves it uninitialised, environment.cpp:130); MakeGame's board comes
// from InitBoardItems above.
class Environment {
public:
    Environment() : state(std::make_unique<State>()) { agents.fill(nullptr); std::fill(lastMoves, lastMoves + AGENT_COUNT, Move::IDLE); }
    Environment(const Environment&) = delete;
    Environment& operator=(const Environment&) = delete;

    void MakeGame(std::array<Agent*, AGENT_COUNT> a, bool randomizePositions = false)
    {
        InitBoardItems(*state);
        std::array<int, 4> f = {0, 1, 2, 3};
        if (randomizePositions) std::shuffle(f.begin(), f.end(), std::mt19937(std::random_device{}()));
        state->PutAgentsInCorners(f[0], f[1], f[2], f[3]);
        SetAgents(a);
        hasStarted = true;
    }
    void StartGame(int timeSteps, bool render = true, bool stepByStep = false)
    {
        state->timeStep = 0;
        while (!IsDone() && state->timeStep < timeSteps) {
            if (render) {
                Print();
                if (listener) listener(*this);
                if (stepByStep) std::cin.get();
            }
            Step(true);
        }
        Print();
        std::cout << std::endl;
        if (!IsDone()) std::cout << "Draw! Max timesteps reached " << std::endl;
        else if (IsDraw()) std::cout << "Draw! All agents are dead" << std::endl;
        else std::cout << "Finished! The winner is Agent " << GetWinner() << std::endl;
    }
    // competitiveTimeLimit: the live agents think concurrently and the environment waits 100 ms (environment.cpp:95-116)
    void Step(bool competitiveTimeLimit = false)
    {
        if (!hasStarted || finished) return;
        Move m[AGENT_COUNT] = {Move::IDLE, Move::IDLE, Move::IDLE, Move::IDLE};
        if (competitiveTimeLimit) {
            std::thread th[AGENT_COUNT];
            for (int i = 0; i < AGENT_COUNT; ++i)
                if (!state->agents[i].dead) th[i] = std::thread([&m, this, i] { m[i] = agents[i]->act(state.get()); });
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            for (int i = 0; i < AGENT_COUNT; ++i)
                if (th[i].joinable()) th[i].join();
        } else {
            for (int i = 0; i < AGENT_COUNT; ++i)
                if (!state->agents[i].dead) {  // act() is only asked of live agents, environment.cpp:139-146
                    m[i] = agents[i]->act(state.get());
                    lastMoves[i] = m[i];
                }
        }
        int32_t done = 0, winner = -1, draw = 0;
        pom_check(pom_env_step(state.get(), reinterpret_cast<const int32_t*>(m), 0, &done, &winner, &draw, nullptr));
        finished = done != 0;
        isDraw = draw != 0;
        if (winner >= 0) agentWon = winner;
    }
    void Print(bool clear = true) { (void)clear; PrintState(state.get(), true); }
    State& GetState() const { return *state; }
    void SetAgents(std::array<Agent*, AGENT_COUNT> a)
    {
        for (int i = 0; i < AGENT_COUNT; ++i) a[size_t(i)]->id = i;
        agents = a;
    }
    Agent* GetAgent(uint agentID) const { return agents[agentID]; }
    void SetStepListener(const std::function<void(const Environment&)>& f) { listener = f; }
    bool IsDone() { return finished; }
    bool IsDraw() { return isDraw; }
    int GetWinner() { return agentWon; }
    Move GetLastMove(int agentID) { return lastMoves[agentID]; }

private:
    std::unique_ptr<State> state;
    std::array<Agent*, AGENT_COUNT> agents;
    std::function<void(const Environment&)> listener;
    bool finished = false, hasStarted = false, isDraw = false;
    int agentWon = -1;
    Move lastMoves[AGENT_COUNT];
};

// n concurrent games on one device: MakeGame / Step / IsDone / IsDraw / GetWinner / GetState of Environment,
// with moves for all n games handed over at once.  Agents are asked for a move only while alive
// (environment.cpp:139-146); a dead agent's entry is IDLE.
class BatchEnvironment {
public:
    // autoReset: POM_RESET_OFF (finished games stay finished), POM_RESET_AT_START (the next Step restarts and steps them) or
    // POM_RESET_AT_END (the Step that finishes a game leaves the next game's start state behind: what agents are shown next
    // is the state their move will be applied to; LastResults() has the finished game's outcome).
    // freshBoards: a restarting game gets a newly drawn board (InitState's distribution, bboard.cpp:339-382, generated on the
    // device; include/pom_boardgen.h) instead of replaying the one handed to MakeGame
    explicit BatchEnvironment(int64_t n, int device = 0, int autoReset = POM_RESET_OFF, int maxSteps = 0, bool freshBoards = false,
                              uint64_t boardSeed = 0)
        : n_(n), states_(size_t(n))
    {
        PomBatchOptions o{};
        o.struct_size = sizeof o;
        o.device = device;
        o.mode = POM_MODE_ENV;
        o.auto_reset = autoReset;
        o.max_steps = maxSteps;
        o.fresh_boards = freshBoards;
        o.board_seed = boardSeed;
        pom_check(pom_batch_create(&h_, n, &o));
    }
    ~BatchEnvironment() { pom_batch_destroy(h_); }
    BatchEnvironment(const BatchEnvironment&) = delete;
    BatchEnvironment& operator=(const BatchEnvironment&) = delete;

    int64_t Size() const { return n_; }
    void MakeGame(const State* start) { pom_check(pom_batch_upload(h_, start, 0, n_)); }
    // MakeGame without the host: InitState for every game, drawn on the device (bboard.cpp:339-344)
    void MakeGame(uint64_t boardSeed) { pom_check(pom_batch_generate(h_, boardSeed)); }
    // Environment::Step with four agents::SimpleAgent per game, policy and tick both on the device (include/agents.hpp:55-76)
    void StepSimpleAgents(uint64_t seed, int ticks = 1) { pom_check(pom_batch_step_simple(h_, seed, ticks)); }
    void Step(const Move* moves /* [n][AGENT_COUNT] */) { pom_check(pom_batch_step(h_, reinterpret_cast<const int32_t*>(moves))); }
    // `ticks` Steps whose moves are fixed in advance: a tape Move[ticks][n][AGENT_COUNT] in DEVICE memory (replays, open-loop
    // rollouts), issued as chained launches; the tape must stay unchanged until the next call that reads results back
    void StepTape(const Move* movesDev, int ticks) { pom_check(pom_batch_step_device_many(h_, reinterpret_cast<const int32_t*>(movesDev), ticks)); }
    // ask `agents` (shared by all games) for moves on the current states, then step: Environment::Step
    void Step(const std::array<Agent*, AGENT_COUNT>& agents)
    {
        const State* st = GetStates();
        std::vector<Move> mv(size_t(n_) * AGENT_COUNT, Move::IDLE);
        for (int64_t e = 0; e < n_; ++e)
            for (int i = 0; i < AGENT_COUNT; ++i)
                if (!st[e].agents[i].dead) {
                    agents[i]->id = i;
                    mv[size_t(e) * AGENT_COUNT + i] = agents[i]->act(&st[e]);
                }
        Step(mv.data());
    }
    const State* GetStates()
    {
        pom_check(pom_batch_download(h_, states_.data(), 0, n_));
        return states_.data();
    }
    void Status(std::vector<int32_t>& done, std::vector<int32_t>& winner, std::vector<int32_t>& draw)
    {
        done.resize(size_t(n_));
        winner.resize(size_t(n_));
        draw.resize(size_t(n_));
        pom_check(pom_batch_status(h_, 0, n_, done.data(), winner.data(), draw.data(), nullptr, nullptr, nullptr));
    }
    // POM_RESET_AT_END: which games finished with the latest Step, and how (winner -1: nobody)
    void LastResults(std::vector<int32_t>& finished, std::vector<int32_t>& winner, std::vector<int32_t>& draw)
    {
        finished.resize(size_t(n_));
        winner.resize(size_t(n_));
        draw.resize(size_t(n_));
        pom_check(pom_batch_last_results(h_, 0, n_, finished.data(), winner.data(), draw.data(), nullptr, nullptr));
    }
    // Environment's per-game queries (bboard.hpp:597-636, environment.cpp:176-208) for game e; each call downloads what it
    // needs — game loops that look at every game should use GetStates() / Status() once per tick instead
    const State& GetState(int64_t e)
    {
        pom_check(pom_batch_download(h_, &states_[size_t(e)], e, 1));
        return states_[size_t(e)];
    }
    bool IsDone(int64_t e) { return Query(e, 0) != 0; }
    bool IsDraw(int64_t e) { return Query(e, 2) != 0; }
    int GetWinner(int64_t e) { return Query(e, 1); }
    PomBatch* Handle() { return h_; }

private:
    int32_t Query(int64_t e, int which)
    {
        int32_t v[3] = {0, -1, 0};
        pom_check(pom_batch_status(h_, e, 1, &v[0], &v[1], &v[2], nullptr, nullptr, nullptr));
        return v[which];
    }
    PomBatch* h_ = nullptr;
    int64_t n_;
    std::vector<State> states_;
};

}  // namespace bboard

namespace std {
template <>
struct hash<bboard::Position> {  // bboard.hpp:693-704: positions as keys of unordered containers (strategy.cpp:270)
    size_t operator()(const bboard::Position& p) const { return hash<int>()(p.x + p.y * bboard::BOARD_SIZE); }
};
}  // namespace std

#endif  // POM_BBOARD_HPP_
