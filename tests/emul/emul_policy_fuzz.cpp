/*
 * emul_policy_fuzz.cpp — TEST-ONLY: four SimpleAgents play random games; every act() is computed by the device policy body
 * (host build) and by the policy oracle on the same state, memory and draw.  usage: emul_policy_fuzz <scenario> <acts> <seed> [quad]
 * quad: the searches run through the quad-word level functions the kernels' wave-cooperative floods are made of
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {
#include "pom_oracle.h"
#include "pom_policy_oracle.h"
#include "pom_testgen.h"
#include "pom_rng.h"
int pom_emul_simple_act(const void* state_1004, int id, int32_t* mem16, int draw);
extern int pom_emul_quad_floods;
}

int main(int argc, char** argv)
{
    int scenario = argc > 1 ? atoi(argv[1]) : 1;
    long long want = argc > 2 ? atoll(argv[2]) : 200000;
    uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
    pom_emul_quad_floods = argc > 4 && !strcmp(argv[4], "quad");
    PomTestRng rng = {pom_splitmix64(seed) | 1};
    long long acts = 0, episodes = 0, maps = 0;
    PomState st;
    while (acts < want) {
        pom_testgen_board(&st, scenario, &rng);
        if (episodes & 1) {
            st.board[0][0] = 0; st.board[0][10] = 0;
            pom_oracle_put_agent(&st, 4, 5, 0);
            pom_oracle_put_agent(&st, 6, 5, 1);
        }
        episodes++;
        PomSimpleMem mem[4];
        memset(mem, 0, sizeof mem);
        for (int t = 0; t < 300 && st.aliveAgents > 1; t++) {
            int32_t mv[4] = {0, 0, 0, 0};
            const uint64_t r = pom_rng_draw(seed, (uint32_t)episodes, (uint32_t)t);
            for (int i = 0; i < 4; i++) {
                if (st.agents[i].dead) continue;
                const int draw = (int)((((uint32_t)(r >> (16 * i)) & 0xFFFFu) * 5u) >> 16);
                int32_t m16[16];
                memcpy(m16, &mem[i], sizeof m16);
                const int m_dev = pom_emul_simple_act(&st, i, m16, draw);
                const int m_ora = pom_oracle_simple_act(&st, i, &mem[i], draw);
                acts++;
                if (m_dev != m_ora || memcmp(m16, &mem[i], sizeof m16)) {
                    printf("MISMATCH scenario %d episode %lld tick %d agent %d draw %d: device body %d, oracle %d\n", scenario, episodes, t, i,
                           draw, m_dev, m_ora);
                    const int32_t* om = (const int32_t*)&mem[i];
                    for (int k = 0; k < 16; k++) printf("  mem[%d] dev %d ora %d\n", k, m16[k], om[k]);
                    return 1;
                }
                mv[i] = m_ora;
            }
            if (pom_oracle_step(&st, mv) & ~1u) break;
            st.timeStep++;
        }
    }
    (void)maps;
    printf("emul_policy_fuzz scenario %d seed %llu: %lld act() calls, %lld games, 0 mismatches\n", scenario, (unsigned long long)seed, acts,
           episodes);
    return 0;
}
