/*
 * pom_policy_emul.cpp — TEST-ONLY host build of the device policy body (pomcpp_amd/csrc/pom_policy_body.h) over plain arrays,
 * to fuzz the kernel's logic against the policy oracle without a GPU.  Never linked into the product.
 */
#include <cstring>

#include "pom_packed.h"
#include "pom_policy_body.h"

struct PolicyArrays {
    uint8_t cells[128];
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
    uint32_t cells4(int k) const { return (uint32_t)cells[4 * k] | ((uint32_t)cells[4 * k + 1] << 8) | ((uint32_t)cells[4 * k + 2] << 16) | ((uint32_t)cells[4 * k + 3] << 24); }
    int bomb(int s) const { return bombs[s]; }
};

/* the four lanes of a quad at once: what PomQuadLanes (pom_kernels.h) is on the device, for the level functions of
 * pom_policy_body.h that the wave-cooperative floods are made of */
struct QW {
    uint32_t v[4];
    QW operator&(const QW& o) const { return QW{{v[0] & o.v[0], v[1] & o.v[1], v[2] & o.v[2], v[3] & o.v[3]}}; }
    QW operator|(const QW& o) const { return QW{{v[0] | o.v[0], v[1] | o.v[1], v[2] | o.v[2], v[3] | o.v[3]}}; }
    QW operator~() const { return QW{{~v[0], ~v[1], ~v[2], ~v[3]}}; }
    QW operator<<(int s) const { return QW{{v[0] << s, v[1] << s, v[2] << s, v[3] << s}}; }
    QW operator>>(int s) const { return QW{{v[0] >> s, v[1] >> s, v[2] >> s, v[3] >> s}}; }
};
struct PomQuadHost {
    typedef QW W;
    W prev(W w) const { return QW{{0u, w.v[0], w.v[1], w.v[2]}}; }
    W next(W w) const { return QW{{w.v[1], w.v[2], w.v[3], 0u}}; }
    bool any(W w) const { return (w.v[0] | w.v[1] | w.v[2] | w.v[3]) != 0; }
    W bit(int c) const
    {
        QW r{{0, 0, 0, 0}};
        r.v[c >> 5] = 1u << (c & 31);
        return r;
    }
    W col0() const { return QW{{0x00400801u, 0x00801002u, 0x01002004u, 0x00004008u}}; }
    W col10() const { return QW{{0x00200400u, 0x00400801u, 0x00801002u, 0x01002004u}}; }
    W valid() const { return QW{{~0u, ~0u, ~0u, 0x01FFFFFFu}}; }
    int lowest(W w) const
    {
        for (int k = 0; k < 4; k++)
            if (w.v[k]) return 32 * k + __builtin_ctz(w.v[k]);
        return 999;
    }
    int gates_hit(W hit, uint32_t g8) const
    {
        int m = 0;
        for (int j = 0; j < 4; j++) {
            const int g = (int)((g8 >> (8 * j)) & 0xFF);
            if (g != 0xFF && ((hit.v[g >> 5] >> (g & 31)) & 1u)) m |= 1 << j;
        }
        return m;
    }
};

/* SimpleAgent::act with the searches run through the quad-word level functions, as the kernels run them */
template <class P>
static int act_with_quad_floods(PomSimplePolicy<P>& pol, P& st, int draw)
{
    const PomQuadHost q;
    const int src = pol.src_cell();
    QW walk{{st.setw(0), st.setw(1), st.setw(2), st.setw(3)}}, agents{{st.setw(4), st.setw(5), st.setw(6), st.setw(7)}};
    walk = walk & ~q.bit(src);
    agents = agents & ~q.bit(src);
    int safe_cell = -1;
    if (pol.begin()) {
        QW front = q.bit(src), all{{0, 0, 0, 0}};
        while (pom_quad_forward_level(q, walk, agents, front, all)) {
        }
        const PomCells win = pol.window(pol.danger_);
        QW wq;
        for (int k = 0; k < 4; k++) { /* the window word by word, as the worker lanes build it */
            wq.v[k] = pom_window_word(k, pol.sx, pol.sy, pol.danger_);
            if (wq.v[k] != win.w[k]) return -2; /* a mismatch the fuzzer reports */
        }
        const QW cand = wq & all & QW{{st.setw(8), st.setw(9), st.setw(10), st.setw(11)}};
        const int low = q.lowest(cand);
        safe_cell = low == 999 ? -1 : low;
    }
    const int target = pol.pick_target(safe_cell);
    int found, hit = 0;
    uint32_t g8;
    if (pol.path_begin(target, found, g8)) {
        QW gates{{0, 0, 0, 0}};
        for (int j = 0; j < 4; j++)
            if (((g8 >> (8 * j)) & 0xFF) != 0xFF) gates = gates | q.bit((int)((g8 >> (8 * j)) & 0xFF));
        QW front = q.bit(target), seen = front;
        int r;
        while ((r = pom_quad_backward_level(q, walk, gates, g8, front, seen)) < 0) {
        }
        hit = r;
    }
    return pol.finish(pol.path_end(target, found, hit), draw);
}

extern "C" {

/* pom_window_word against PomSimplePolicy::window for every agent cell and radius: the number of words that differ */
int pom_emul_window_check(void)
{
    PolicyArrays st;
    std::memset(&st, 0, sizeof st);
    int bad = 0;
    for (int sy = 0; sy < POM_N; sy++)
        for (int sx = 0; sx < POM_N; sx++) {
            PomPolicyEnv E;
            std::memset(&E, 0, sizeof E);
            E.a0[0] = sx | (sy << 4);
            PomSimplePolicy<PolicyArrays> pol(st, E, 0, 0u, 0u);
            for (int radius = 0; radius <= 15; radius++) {
                const PomCells w = pol.window(radius);
                for (int k = 0; k < 4; k++) bad += pom_window_word(k, sx, sy, radius) != w.w[k];
            }
        }
    return bad;
}

int pom_emul_quad_floods = 0; /* tests: 1 = run the searches through the quad-word level functions */

/* one act() of agent `id` through pack -> device policy body; mem16 in the oracle's 16-int form (in/out).
 * returns the move, or -1 if the state is not representable */
int pom_emul_simple_act(const void* state_1004, int id, int32_t* mem16, int draw)
{
    uint32_t rec[POM_REC_DWORDS];
    if (pom_pack_state((const int32_t*)state_1004, rec, 1)) return -1;
    PolicyArrays st;
    std::memset(&st, 0, sizeof st);
    for (int c = 0; c < POM_CELLS; c++) st.cells[c] = (uint8_t)pom_rec_cell(rec, 1, c);
    for (int c = POM_CELLS + 3; c < 128; c++) /* dword 31 of the record follows the board in the tile: the body must mask it */
        st.cells[c] = (uint8_t)(rec[POM_REC_TIMESTEP] >> (8 * (c & 3)));
    for (int c = POM_CELLS; c < 128; c++) st.dmap[c] = (c * 7) % 3; /* rows past the map: arbitrary */
    for (int k = 0; k < 20; k++) st.bombs[k] = (int)rec[POM_REC_BOMBS + k];
    PomPolicyEnv E;
    for (int i = 0; i < 4; i++) {
        E.a0[i] = (int)rec[POM_REC_AGENTS + 2 * i];
        E.a1[i] = (int)rec[POM_REC_AGENTS + 2 * i + 1];
    }
    E.bIdx = (int)((pom_rec_meta(rec, 1) >> 8) & 0xFF);
    E.bCnt = (int)((pom_rec_meta(rec, 1) >> 16) & 0xFF);
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
    const int mv = pom_emul_quad_floods ? act_with_quad_floods(pol, st, draw) : pol.act(draw);
    pom_policy_mem_unpack(pol.m0, pol.m1, mem16);
    return mv;
}

}
