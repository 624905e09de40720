/*
 * pom_emul_quad.cpp — TEST-ONLY host build of the device tick (pomcpp_amd/csrc/pom_step_body.h) in its SHIPPED shape: four
 * lanes per env (A::G = 4), i.e. the code the quad kernel runs — the agent rounds with rank and depth, the zero-byte contact /
 * clash test, the look / commit rounds of explode_long with lane r = ray r, bomb_index_wide, pack_moves_quad, the folded
 * timer decrement with its quad flag reduction — none of which the one-lane build (pom_emul.cpp) reaches.  Never linked into
 * libpom_batch.so: the product has no CPU stepper.
 *
 * Model.  Four host threads are the four lanes of a quad; each runs PomStepper over its own registers (PomLane) and a store
 * that is a private overlay (the lane's own writes since the last rendezvous) over the committed tile.  Every cross-lane
 * operation of the store interface (gor / gmin / gadd / gbcast — DPP quad permutes on the device — and sync(), a point where
 * the device's lock-step order makes earlier LDS writes of the other lanes visible) is a rendezvous of all four lanes: the
 * overlays are merged into the committed tile, the values exchanged.  What the rendezvous checks is what the device silently
 * assumes:
 *   - all four lanes arrive at the SAME operation (quad-uniform control flow around every cross-lane op);
 *   - two lanes never write different values to one address between two rendezvous (put_*: split sections write disjoint
 *     cells / slots);
 *   - a replicated write (set_*: on the device only the owner lane writes) is reached by the owner, and every other lane that
 *     reaches it holds the same value (the lanes' registers have not diverged);
 *   - at the end of the tick all four lanes hold identical registers.
 * Any violation is reported as a mismatch (POM_EMUL_QUAD_DIVERGED in the returned flags) on top of the comparison with the
 * oracle that the callers make.  Within one lane reads see the lane's own writes first (program order); another lane's
 * writes become visible at the next rendezvous — the device makes them visible earlier (at the next instruction), which
 * correct code must not rely on except where it says so with sync().
 */
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "pom_packed.h"
#include "pom_step_body.h"

namespace {

enum { OP_OR = 1, OP_MIN, OP_ADD, OP_BCAST, OP_SYNC = OP_BCAST + 4, OP_END };
enum { W_NONE = 0, W_PUT = 1, W_SET_OWNER = 2, W_SET_SHADOW = 3 };
enum { N_CELL = 124, N_SLOT = 20, N_STACK = POM_STACK_DEPTH, N_ADDR = N_CELL + 3 * N_SLOT + N_STACK + 4 };
/* one address space for the merge: cells, bombs, flames, bomb destinations, frames */
enum { A_CELL = 0, A_BOMB = N_CELL, A_FLAME = A_BOMB + N_SLOT, A_BDEST = A_FLAME + N_SLOT, A_STACK = A_BDEST + N_SLOT, A_AG1 = A_STACK + N_STACK };

struct Quad {
    int mem[N_ADDR];            /* the committed tile */
    int ov[4][N_ADDR];          /* per lane: pending writes */
    unsigned char kind[4][N_ADDR];
    int dirty_list[4][N_ADDR], n_dirty[4];
    int claim_mem[124];         /* the cell counters (loop_b_todo): committed values ... */
    int claim_add[4][124];      /* ... and what each lane has counted since the last rendezvous (atomic adds on the device: they sum) */
    int claim_clear[4];         /* a lane has cleared its share since the last rendezvous */
    int xval[2][4], xop[2][4];  /* exchange slots, double-buffered by the parity of the rendezvous count */
    std::atomic<int> arrived{0};
    std::atomic<unsigned> generation{0};
    std::atomic<int> diverged{0};
    int why = 0, why_addr = -1, why_round = -1; /* first violation: 1 different operations, 2 conflicting writes, 3 replicated write not shared by the owner, 4 registers */
    PomLane lanes[4];
    /* the job */
    uint32_t mvp_moves[4];
    int32_t moves[4];
};

struct Barrier { /* four participants, sense by generation; spins, yields when the box is oversubscribed */
    static void wait(Quad& q)
    {
        const unsigned gen = q.generation.load(std::memory_order_acquire);
        if (q.arrived.fetch_add(1, std::memory_order_acq_rel) == 3) {
            q.arrived.store(0, std::memory_order_relaxed);
            q.generation.store(gen + 1, std::memory_order_release);
            return;
        }
        int spins = 0;
        while (q.generation.load(std::memory_order_acquire) == gen)
            if (++spins > 2000) std::this_thread::yield();
    }
};

struct QuadLaneEnv {
    static constexpr int G = 4;
    Quad* q;
    int sub_;
    mutable int round = 0; /* rendezvous count of this lane (all lanes count alike while they agree) */

    int sub() const { return sub_; }
    int rd(int a) const { return q->kind[sub_][a] ? q->ov[sub_][a] : q->mem[a]; }
    void wr(int a, int v, int k) const
    {
        static const int watch = getenv("POM_EMUL_QUAD_WATCH") ? atoi(getenv("POM_EMUL_QUAD_WATCH")) : -1;
        if (a == watch) fprintf(stderr, "  lane %d round %d: write kind %d value %x\n", sub_, round, k, v);
        if (!q->kind[sub_][a]) q->dirty_list[sub_][q->n_dirty[sub_]++] = a;
        q->ov[sub_][a] = v;
        q->kind[sub_][a] = (unsigned char)k;
    }
    int set_kind() const { return sub_ == 0 ? W_SET_OWNER : W_SET_SHADOW; }

    void flag(int why, int addr) const
    {
        if (!q->diverged.exchange(1)) {
            q->why = why;
            q->why_addr = addr;
            q->why_round = round;
            if (getenv("POM_EMUL_QUAD_VERBOSE") && why != 1)
                fprintf(stderr, "quad model: address %d committed %x; lanes kind/value: %d/%x %d/%x %d/%x %d/%x\n", addr, q->mem[addr], q->kind[0][addr],
                        q->ov[0][addr], q->kind[1][addr], q->ov[1][addr], q->kind[2][addr], q->ov[2][addr], q->kind[3][addr], q->ov[3][addr]);
        }
    }
    /* merge the four overlays into the committed tile (lane 0, between the two barriers of a rendezvous) */
    void merge() const
    {
        for (int l = 0; l < 4; l++) {
            for (int i = 0; i < q->n_dirty[l]; i++) {
                const int a = q->dirty_list[l][i];
                if (q->kind[l][a] == W_SET_SHADOW) continue;
                int v = q->ov[l][a];
                for (int m = l + 1; m < 4; m++) /* two real writes to one address must agree */
                    if (q->kind[m][a] == W_PUT || q->kind[m][a] == W_SET_OWNER)
                        if (q->ov[m][a] != v) flag(2, a);
                q->mem[a] = v;
            }
        }
        for (int l = 0; l < 4; l++) {
            for (int i = 0; i < q->n_dirty[l]; i++) {
                const int a = q->dirty_list[l][i];
                /* a replicated write reached by a lane that is not the owner: the owner must have written there too, and what is
                 * committed must be what this lane believes is there */
                if (q->kind[l][a] == W_SET_SHADOW && (q->kind[0][a] == W_NONE || q->mem[a] != q->ov[l][a])) flag(3, a);
            }
        }
        for (int l = 0; l < 4; l++) {
            for (int i = 0; i < q->n_dirty[l]; i++) q->kind[l][q->dirty_list[l][i]] = W_NONE;
            q->n_dirty[l] = 0;
        }
        /* the cell counters: a lane's share is dwords l, l + 4, ... of the 31; clears first (a clear and a count of the same cell in
         * one interval would race on the device: flagged), then the counts, which add up */
        for (int l = 0; l < 4; l++) {
            if (!q->claim_clear[l]) continue;
            for (int d = l; d < 31; d += 4)
                for (int c = 4 * d; c < 4 * d + 4; c++) {
                    for (int m = 0; m < 4; m++)
                        if (q->claim_add[m][c]) flag(2, 10000 + c);
                    q->claim_mem[c] = 0;
                }
            q->claim_clear[l] = 0;
        }
        for (int l = 0; l < 4; l++)
            for (int c = 0; c < 124; c++) {
                q->claim_mem[c] += q->claim_add[l][c];
                q->claim_add[l][c] = 0;
            }
    }
    int rendezvous(int op, int v) const
    {
        const int par = round & 1;
        round++;
        q->xval[par][sub_] = v;
        q->xop[par][sub_] = op;
        Barrier::wait(*q);
        if (sub_ == 0) {
            for (int l = 1; l < 4; l++)
                if (q->xop[par][l] != op) flag(1, q->xop[par][l] * 100 + op); /* the lanes are not at the same operation */
            merge();
        }
        Barrier::wait(*q);
        const int* x = q->xval[par];
        switch (op) {
        case OP_OR: return x[0] | x[1] | x[2] | x[3];
        case OP_ADD: return x[0] + x[1] + x[2] + x[3];
        case OP_MIN: {
            int m = x[0];
            for (int l = 1; l < 4; l++) m = x[l] < m ? x[l] : m;
            return m;
        }
        case OP_BCAST: case OP_BCAST + 1: case OP_BCAST + 2: case OP_BCAST + 3: return x[op - OP_BCAST];
        default: return 0;
        }
    }
    int gor(int v) const { return rendezvous(OP_OR, v); }
    int gmin(int v) const { return rendezvous(OP_MIN, v); }
    int gadd(int v) const { return rendezvous(OP_ADD, v); }
    template <int J> int gbcast(int v) const { return rendezvous(OP_BCAST + J, v); }
    void sync() const { (void)rendezvous(OP_SYNC, 0); }

    int cell(int c) const { return rd(A_CELL + c); }
    void put_cell(int c, int v) { wr(A_CELL + c, v & 0xFF, W_PUT); }
    void set_cell(int c, int v) { wr(A_CELL + c, v & 0xFF, set_kind()); }
    int bomb(int s) const { return rd(A_BOMB + s); }
    void put_bomb(int s, int v) { wr(A_BOMB + s, v, W_PUT); }
    void set_bomb(int s, int v) { wr(A_BOMB + s, v, set_kind()); }
    int flame(int s) const { return rd(A_FLAME + s); }
    void put_flame(int s, int v) { wr(A_FLAME + s, v, W_PUT); }
    void set_flame(int s, int v) { wr(A_FLAME + s, v, set_kind()); }
    int bdest(int i) const { return rd(A_BDEST + i); }
    void put_bdest(int i, int v) { wr(A_BDEST + i, v & 0xFF, W_PUT); }
    void set_bdest(int i, int v) { wr(A_BDEST + i, v & 0xFF, set_kind()); }
    int frame(int d) const { return rd(A_STACK + d); }
    void set_frame(int d, int v) { wr(A_STACK + d, v, set_kind()); }
    int ag1(int i) const { return rd(A_AG1 + i); }
    void put_ag1(int i, int v) { wr(A_AG1 + i, v, W_PUT); }
    void set_ag1(int i, int v) { wr(A_AG1 + i, v, set_kind()); }
    void claims_clear() { q->claim_clear[sub_] = 1; }
    void claim(int c) { q->claim_add[sub_][c]++; }
    int claims(int c) const /* read after a sync(): nothing of this lane's own may be pending (the device reads what is committed) */
    {
        if (q->claim_add[sub_][c] || q->claim_clear[sub_]) flag(2, 20000 + c);
        return q->claim_mem[c];
    }
};

void run_lane(Quad& q, int sub)
{
    QuadLaneEnv env{&q, sub};
    PomLane& L = q.lanes[sub];
    PomStepper<QuadLaneEnv> st(env, L);
    /* the kernel's own entry: lane m hands in agent m's move, the quad exchanges them */
    const uint32_t mvp = st.pack_moves_quad(q.moves[sub]);
    st.step_packed(mvp);
    /* the end of the tick is a rendezvous too; a lane that gets here while others are still stepping (diverged: already
     * flagged by the operation check) keeps answering their rendezvous until all four have arrived */
    for (;;) {
        const int par = env.round & 1;
        (void)env.rendezvous(OP_END, 0);
        bool all = true;
        for (int l = 0; l < 4; l++) all = all && q.xop[par][l] == OP_END;
        if (all) break;
        if (env.round > 200000) std::abort(); /* a diverged lane that never ends: fail loudly rather than hang */
    }
}

/* three persistent helper threads = lanes 1..3; the caller is lane 0 */
struct Workers {
    std::mutex mu;
    std::condition_variable cv;
    Quad* job = nullptr;
    unsigned posted = 0;
    bool quit = false;
    std::atomic<unsigned> finished{0};
    std::thread th[3];
    Workers()
    {
        for (int k = 0; k < 3; k++)
            th[k] = std::thread([this, k] {
                unsigned seen = 0;
                for (;;) {
                    Quad* q;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return quit || posted != seen; });
                        if (quit) return;
                        seen = posted;
                        q = job;
                    }
                    run_lane(*q, k + 1);
                    finished.fetch_add(1, std::memory_order_release);
                }
            });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            quit = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void run(Quad& q)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            job = &q;
            posted++;
        }
        cv.notify_all();
        run_lane(q, 0);
        /* the helpers are past the last barrier too, but may still be looking at the exchange slots: wait until they are out */
        int spins = 0;
        while (finished.load(std::memory_order_acquire) != 3 * posted)
            if (++spins > 2000) std::this_thread::yield();
    }
};

} // namespace

extern "C" {

enum { POM_EMUL_QUAD_DIVERGED = 0x40000000u };

/* one tick through pack -> device body, four lanes per env -> unpack; same contract as pom_emul_step (pom_emul.cpp) plus the
 * POM_EMUL_QUAD_DIVERGED bit when the quad model's checks fail */
uint32_t pom_emul_quad_step(void* state_1004, const int32_t* moves, int env_mode, int max_steps, uint32_t* status_io)
{
    static Workers workers;
    static Quad q; /* one call at a time (the tests are single-threaded) */
    uint32_t rec[POM_REC_DWORDS];
    if (pom_pack_state((const int32_t*)state_1004, rec, 1)) return 0xFFFFFFFFu;
    std::memset(q.mem, 0, sizeof q.mem);
    std::memset(q.kind, 0, sizeof q.kind);
    std::memset(q.n_dirty, 0, sizeof q.n_dirty);
    std::memset(q.claim_mem, 0x55, sizeof q.claim_mem); /* whatever the last tick left there */
    std::memset(q.claim_add, 0, sizeof q.claim_add);
    std::memset(q.claim_clear, 0, sizeof q.claim_clear);
    q.diverged.store(0);
    for (int c = 0; c < POM_CELLS; c++) q.mem[A_CELL + c] = pom_rec_cell(rec, 1, c);
    for (int k = 0; k < 20; k++) {
        q.mem[A_BOMB + k] = (int)rec[POM_REC_BOMBS + k];
        q.mem[A_FLAME + k] = (int)rec[POM_REC_FLAMES + k];
    }
    for (int i = 0; i < 4; i++) q.mem[A_AG1 + i] = (int)rec[POM_REC_AGENTS + 2 * i + 1];
    int time_step = (int)rec[POM_REC_TIMESTEP];
    uint32_t status = status_io ? *status_io : 0;
    for (int l = 0; l < 4; l++) {
        uint32_t status_rec = 0;
        pom_lane_load(q.lanes[l], rec + POM_REC_AGENTS, status_rec);
        q.lanes[l].ub = 0;
        q.moves[l] = moves[l];
    }
    uint32_t extra = 0;
    if (!(env_mode && (status & POM_ST_DONE))) {
        workers.run(q);
        for (int l = 1; l < 4; l++) /* replicated registers: identical in all four lanes at the end of the tick */
            if (std::memcmp(&q.lanes[0], &q.lanes[l], sizeof(PomLane)) != 0 && !q.diverged.exchange(1)) q.why = 4;
        if (q.diverged.load()) {
            extra = POM_EMUL_QUAD_DIVERGED;
            if (getenv("POM_EMUL_QUAD_VERBOSE")) fprintf(stderr, "quad model: violation %d at address / ops %d, rendezvous %d\n", q.why, q.why_addr, q.why_round);
        }
        if (env_mode) {
            time_step++;
            status = pom_env_epilogue(q.lanes[0], time_step, max_steps, status);
        }
    }
    const PomLane& L = q.lanes[0];
    for (int r = 0; r < POM_REC_BOARD_DWORDS; r++)
        rec[POM_REC_BOARD + r] = (uint32_t)q.mem[A_CELL + 4 * r] | ((uint32_t)q.mem[A_CELL + 4 * r + 1] << 8) |
                                 ((uint32_t)q.mem[A_CELL + 4 * r + 2] << 16) | ((uint32_t)q.mem[A_CELL + 4 * r + 3] << 24);
    rec[POM_REC_TIMESTEP] = (uint32_t)time_step;
    for (int k = 0; k < 8; k++) rec[POM_REC_AGENTS + k] = pom_lane_agent_word(L, status, k, (k & 1) ? (uint32_t)q.mem[A_AG1 + (k >> 1)] : 0u);
    for (int k = 0; k < 20; k++) {
        rec[POM_REC_BOMBS + k] = (uint32_t)q.mem[A_BOMB + k];
        rec[POM_REC_FLAMES + k] = (uint32_t)q.mem[A_FLAME + k];
    }
    int32_t out[251];
    std::memset(out, 0, sizeof out);
    pom_unpack_state(rec, 1, out);
    std::memcpy(state_1004, out, POM_STATE_BYTES);
    if (status_io) *status_io = status;
    return L.ub | extra;
}

}
