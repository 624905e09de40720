/*
 * pom_policy_emul.cpp — TEST-ONLY host build of the device policy body (pomcpp_amd/csrc/pom_policy_body.h) over plain arrays,
 * to fuzz the kernel's logic against the policy oracle without a GPU.  Never linked into the product.
 */
#include <cstring>

#include "pom_packed.h"
#include "pom_policy_body.h"

struct PolicyArrays {
    uint16_t cells[128];
    int bombs[20];
    int dmap[128];
    uint32_t sets[12];
    int who; /* which of the env's four lanes is executing */
    int member() const { return who; }
    int danger(int c) const { return dmap[c]; }
    void danger_init(int c) { dmap[c] = POM_DANGER_NONE; }
    void danger_put(int c, int t) { dmap[c] = t; }
    uint32_t setw(int k) const { return sets[k]; }
    void set_put(int k, uint32_t bits) { sets[k] = bits; }
    int cell(int c) const { return cells[c]; }
    uint32_t board_word(int k) const { return (uint32_t)cells[2 * k] | ((uint32_t)cells[2 * k + 1] << 16); }
    int bomb(int s) const { return bombs[s]; }
};

extern "C" {

/* one act() of agent `id` through pack -> device policy body; mem16 in the oracle's 16-int form (in/out).
 * returns the move, or -1 if the state is not representable */
int pom_emul_simple_act(const void* state_1004, int id, int32_t* mem16, int draw)
{
    uint32_t rec[POM_REC_DWORDS];
    if (pom_pack_state((const int32_t*)state_1004, rec, 1)) return -1;
    PolicyArrays st;
    std::memset(&st, 0, sizeof st);
    for (int r = 0; r < 61; r++) {
        st.cells[2 * r] = (uint16_t)(rec[POM_REC_BOARD + r] & 0xFFFF);
        st.cells[2 * r + 1] = (uint16_t)(rec[POM_REC_BOARD + r] >> 16);
    }
    for (int r = 61; r < 64; r++) { /* dwords 61..63 of the record follow the board in the tile: the body must mask them */
        st.cells[2 * r] = (uint16_t)(rec[r] & 0xFFFF);
        st.cells[2 * r + 1] = (uint16_t)(rec[r] >> 16);
    }
    for (int c = POM_CELLS; c < 128; c++) st.dmap[c] = (c * 7) % 3; /* rows past the map: arbitrary */
    for (int k = 0; k < 20; k++) st.bombs[k] = (int)rec[POM_REC_BOMBS + k];
    PomPolicyEnv E;
    for (int i = 0; i < 4; i++) {
        E.a0[i] = (int)rec[POM_REC_AGENTS + 2 * i];
        E.a1[i] = (int)rec[POM_REC_AGENTS + 2 * i + 1];
    }
    E.bIdx = (int)((rec[POM_REC_META] >> 8) & 0xFF);
    E.bCnt = (int)((rec[POM_REC_META] >> 16) & 0xFF);
    /* 16-int memory -> 2 dwords */
    uint32_t m0 = 0, m1 = 0;
    for (int i = 0; i < 4; i++) m0 |= (uint32_t)((mem16[2 * i] & 0xF) | ((mem16[2 * i + 1] & 0xF) << 4)) << (8 * i);
    m1 = (uint32_t)(mem16[8] & 3) | ((uint32_t)(mem16[9] & 7) << 2) | ((uint32_t)(mem16[15] & 7) << 17);
    for (int i = 0; i < 4; i++) m1 |= (uint32_t)(mem16[10 + i] & 7) << (5 + 3 * i);
    /* on the device the env's four lanes do this together, phase by phase */
    for (st.who = 0; st.who < 4; st.who++) pom_policy_prepare_clear(st);
    for (st.who = 0; st.who < 4; st.who++) pom_policy_prepare_fill(st, E);
    for (st.who = 0; st.who < 4; st.who++) pom_policy_prepare_safe(st);
    st.who = id;
    PomSimplePolicy<PolicyArrays> pol(st, E, id, m0, m1);
    const int mv = pol.act(draw);
    pom_policy_mem_unpack(pol.m0, pol.m1, mem16);
    return mv;
}

}
