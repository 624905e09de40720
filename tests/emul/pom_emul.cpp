/*
 * pom_emul.cpp — TEST-ONLY host build of the device tick (pomcpp_amd/csrc/pom_step_body.h)
 * over a plain-array store, so the kernel's logic can be fuzzed against the oracle in a
 * container without a GPU.  Lives under tests/ and is never linked into libpom_batch.so:
 * the product has no CPU stepper.
 */
#include <cstring>

#include "pom_packed.h"
#include "pom_step_body.h"

struct ArrayEnv {
    uint8_t cells[124];
    int bombs[20], flames[20], stack[POM_STACK_DEPTH];
    uint8_t bd[20];
    int cnt[124];
    int a1w[4];
    static constexpr int G = 1; /* one lane per env: the split sections degenerate to plain loops */
    int sub() const { return 0; }
    int gor(int v) const { return v; }
    int gmin(int v) const { return v; }
    int gadd(int v) const { return v; }
    template <int J> int gbcast(int v) const { return v; }
    void sync() const {}
    void put_cell(int c, int v) { cells[c] = (uint8_t)v; }
    void put_bomb(int s, int v) { bombs[s] = v; }
    void put_flame(int s, int v) { flames[s] = v; }
    void put_bdest(int i, int v) { bd[i] = (uint8_t)v; }
    int cell(int c) const { return cells[c]; }
    void set_cell(int c, int v) { cells[c] = (uint8_t)v; }
    int bomb(int s) const { return bombs[s]; }
    void set_bomb(int s, int v) { bombs[s] = v; }
    int flame(int s) const { return flames[s]; }
    void set_flame(int s, int v) { flames[s] = v; }
    int bdest(int i) const { return bd[i]; }
    void set_bdest(int i, int v) { bd[i] = (uint8_t)v; }
    int frame(int d) const { return stack[d]; }
    void set_frame(int d, int v) { stack[d] = v; }
    int ag1(int i) const { return a1w[i]; }
    void set_ag1(int i, int v) { a1w[i] = v; }
    void put_ag1(int i, int v) { a1w[i] = v; }
    void claims_clear() { std::memset(cnt, 0, sizeof cnt); }
    void claim(int c) { cnt[c]++; }
    int claims(int c) const { return cnt[c]; }
};

extern "C" {

/* The device body's agent preparation alone (PomStepper::prep_positions / prep_dependencies: FillPositions, FillDestPos,
 * FixSwitchMove, ResolveDependencies as the tick runs them), for the reference's [step utilities] vectors
 * (unit_test/bboard/step_utility_test.cpp:38-173).  dest: x, y per agent after FixSwitchMove where the body applies it (only
 * with `contact`: some destination touches another agent's cell; without it FixSwitchMove cannot change anything);
 * dependency[j] = the agent that waits for j's cell or -1; roots in visiting order, -1 padded.  Returns the number of roots,
 * or -1 if the state is not representable. */
int pom_emul_prep(const void* state_1004, const int32_t* moves, int32_t* dest_xy, int32_t* dependency, int32_t* roots_out, int32_t* contact_out)
{
    uint32_t rec[POM_REC_DWORDS];
    if (pom_pack_state((const int32_t*)state_1004, rec, 1)) return -1;
    ArrayEnv env;
    std::memset(&env, 0, sizeof env);
    PomLane L;
    uint32_t status_unused = 0;
    pom_lane_load(L, rec + POM_REC_AGENTS, status_unused);
    PomStepper<ArrayEnv> st(env, L);
    const uint32_t mvp = PomStepper<ArrayEnv>::pack_moves(moves);
    uint32_t oldp = 0, dstp = 0, dep = 0xFFFF, roots = 0x3210;
    int nroots = 4, deadmask = 0, contact = 0, clash = 0;
    st.prep_positions(mvp, oldp, dstp, deadmask, contact, clash);
    if (contact) st.prep_dependencies(mvp, oldp, deadmask, dstp, dep, roots, nroots);
    for (int i = 0; i < 4; i++) {
        dest_xy[2 * i] = (int)((dstp >> (8 * i)) & 0xF) - 1;
        dest_xy[2 * i + 1] = (int)((dstp >> (8 * i + 4)) & 0xF) - 1;
        const int d = (int)((dep >> (4 * i)) & 0xF), r = (int)((roots >> (4 * i)) & 0xF);
        dependency[i] = d == 0xF ? -1 : d;
        roots_out[i] = (i < nroots && r != 0xF) ? r : -1;
    }
    if (contact_out) *contact_out = contact | (clash << 1);
    return nroots;
}

/* The device buffers' layout (pom_packed.h: tiles of 16 envs, pom_rec_col): `n` States packed into a buffer of ceil(n/16) tiles
 * exactly as pom_pack_kernel does it, then unpacked again.  Returns 0 if every State comes back and every env's POM_REC_DWORDS dwords lie
 * inside its own tile at the documented places, else a line number. */
int pom_emul_tile_roundtrip(const void* states_1004, int n, void* out_1004)
{
    const int tiles = (n + POM_TILE_ENVS - 1) / POM_TILE_ENVS;
    uint32_t* buf = new uint32_t[(size_t)tiles * POM_TILE_DWORDS];
    uint8_t* owner = new uint8_t[(size_t)tiles * POM_TILE_DWORDS]; /* which env (mod 251, +1) wrote each dword */
    std::memset(buf, 0, (size_t)tiles * POM_TILE_DWORDS * 4);
    std::memset(owner, 0, (size_t)tiles * POM_TILE_DWORDS);
    int rc = 0;
    for (int e = 0; e < n && !rc; e++) {
        const int64_t col = pom_rec_col(e);
        if (col / POM_TILE_DWORDS != e / POM_TILE_ENVS) rc = __LINE__;
        if (pom_pack_state((const int32_t*)states_1004 + (size_t)e * 251, buf + col, POM_TILE_ENVS, e % POM_TILE_ENVS)) rc = __LINE__;
        for (int d = POM_REC_TIMESTEP; d < POM_REC_DWORDS && !rc; d++) {
            const int64_t at = (int64_t)(e / POM_TILE_ENVS) * POM_TILE_DWORDS + d * POM_TILE_ENVS + e % POM_TILE_ENVS; /* the documented place */
            if (at != col + (int64_t)d * POM_TILE_ENVS || owner[at]) rc = __LINE__;
            owner[at] = (uint8_t)(e % 251 + 1);
        }
        /* the board, by cell: cell c of env e is byte c * 16 + e % 16 of its tile */
        const uint8_t* tile_b = (const uint8_t*)(buf + (int64_t)(e / POM_TILE_ENVS) * POM_TILE_DWORDS);
        for (int c = 0; c < POM_CELLS && !rc; c++)
            if (tile_b[c * POM_TILE_ENVS + e % POM_TILE_ENVS] != (uint8_t)pom_cell_encode(((const int32_t*)states_1004)[(size_t)e * 251 + c], c)) rc = __LINE__;
    }
    for (int e = 0; e < n && !rc; e++) {
        int32_t st[251];
        std::memset(st, 0, sizeof st);
        pom_unpack_state(buf + pom_rec_col(e), POM_TILE_ENVS, st, e % POM_TILE_ENVS);
        std::memcpy((char*)out_1004 + (size_t)e * POM_STATE_BYTES, st, POM_STATE_BYTES);
    }
    delete[] buf;
    delete[] owner;
    return rc;
}

/* one tick through pack -> device body -> unpack.  status_io: the env's status byte (ENV mode).
 * returns the POM_UB_* flags of this tick, or 0xFFFFFFFF if the state is not representable */
uint32_t pom_emul_step(void* state_1004, const int32_t* moves, int env_mode, int max_steps, uint32_t* status_io)
{
    uint32_t rec[POM_REC_DWORDS];
    if (pom_pack_state((const int32_t*)state_1004, rec, 1)) return 0xFFFFFFFFu;
    ArrayEnv env;
    std::memset(&env, 0, sizeof env);
    for (int c = 0; c < POM_CELLS; c++) env.cells[c] = (uint8_t)pom_rec_cell(rec, 1, c);
    for (int k = 0; k < 20; k++) {
        env.bombs[k] = (int)rec[POM_REC_BOMBS + k];
        env.flames[k] = (int)rec[POM_REC_FLAMES + k];
    }
    for (int i = 0; i < 4; i++) env.a1w[i] = (int)rec[POM_REC_AGENTS + 2 * i + 1];
    PomLane L;
    uint32_t status_rec = 0;
    pom_lane_load(L, rec + POM_REC_AGENTS, status_rec);
    int time_step = (int)rec[POM_REC_TIMESTEP];
    uint32_t status = status_io ? *status_io : 0;
    L.ub = 0;
    if (!(env_mode && (status & POM_ST_DONE))) {
        PomStepper<ArrayEnv> st(env, L);
        st.step(moves);
        if (env_mode) {
            time_step++;
            status = pom_env_epilogue(L, time_step, max_steps, status);
        }
    }
    for (int r = 0; r < POM_REC_BOARD_DWORDS; r++)
        rec[POM_REC_BOARD + r] = (uint32_t)env.cells[4 * r] | ((uint32_t)env.cells[4 * r + 1] << 8) | ((uint32_t)env.cells[4 * r + 2] << 16) |
                                 ((uint32_t)env.cells[4 * r + 3] << 24); /* (cells 121..123 stay 0) */
    rec[POM_REC_TIMESTEP] = (uint32_t)time_step;
    for (int k = 0; k < 8; k++) rec[POM_REC_AGENTS + k] = pom_lane_agent_word(L, status, k, (k & 1) ? (uint32_t)env.a1w[k >> 1] : 0u);
    for (int k = 0; k < 20; k++) {
        rec[POM_REC_BOMBS + k] = (uint32_t)env.bombs[k];
        rec[POM_REC_FLAMES + k] = (uint32_t)env.flames[k];
    }
    int32_t out[251];
    std::memset(out, 0, sizeof out);
    pom_unpack_state(rec, 1, out);
    std::memcpy(state_1004, out, POM_STATE_BYTES);
    if (status_io) *status_io = status;
    return L.ub;
}

/* chained launches: a visit's distance from its call's first visit (pom_packed.h) */
uint32_t pom_emul_chain_visit_distance(uint32_t visit, uint32_t first) { return pom_chain_visit_distance(visit, first); }

}
