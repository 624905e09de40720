/*
 * flood_levels.cpp — host analysis (not shipped, not a test): how many levels do the policy's two flood fills run, per agent
 * and per "wavefront" (16 envs = 64 agents in lock-step: the wavefront pays the maximum)?  Plays SimpleAgent games on
 * generated boards with the oracle tick; the act() is the device policy body built for the host with POM_LEVEL_STATS.
 * build: see tests/emul/flood_levels.sh
 */
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "pom_boardgen_oracle.h"
#include "pom_oracle.h"
#include "pom_policy_oracle.h"
#include "pom_rng.h"
int pom_emul_simple_act(const void* state_1004, int id, int32_t* mem16, int draw);
int pom_stat_fwd = 0, pom_stat_bwd = 0;
}

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 64, ticks = argc > 2 ? atoi(argv[2]) : 400;
    const int n = waves * 16;
    std::vector<PomState> st(n);
    std::vector<int32_t> mem(n * 4 * 16, 0);
    std::vector<int> ep(n, 0);
    for (int e = 0; e < n; e++) pom_oracle_boardgen(1, e, 0, &st[e]);
    std::vector<int> wf, wb, af, ab;
    long long acts = 0;
    for (int t = 0; t < ticks; t++) {
        for (int w = 0; w < waves; w++) {
            int mf = 0, mb = 0;
            for (int e = 16 * w; e < 16 * w + 16; e++) {
                PomState* s = &st[e];
                if (s->aliveAgents <= 1 || s->timeStep >= 800) {
                    pom_oracle_boardgen(1, e, ++ep[e], s);
                    memset(&mem[e * 64], 0, 64 * sizeof(int32_t));
                }
                const uint64_t r = pom_rng_draw(7, (uint32_t)e, (uint32_t)t);
                int32_t mv[4] = {0, 0, 0, 0};
                for (int i = 0; i < 4; i++) {
                    if (s->agents[i].dead) continue;
                    const int draw = (int)((((uint32_t)(r >> (16 * i)) & 0xFFFFu) * 5u) >> 16);
                    pom_stat_fwd = pom_stat_bwd = 0;
                    mv[i] = pom_emul_simple_act(s, i, &mem[(e * 4 + i) * 16], draw);
                    acts++;
                    if (t >= 100) {
                        af.push_back(pom_stat_fwd);
                        ab.push_back(pom_stat_bwd);
                    }
                    mf = std::max(mf, pom_stat_fwd);
                    mb = std::max(mb, pom_stat_bwd);
                }
                pom_oracle_step(s, mv);
                s->timeStep++;
            }
            if (t >= 100) {
                wf.push_back(mf);
                wb.push_back(mb);
            }
        }
    }
    auto stats = [](const char* name, std::vector<int>& v) {
        std::sort(v.begin(), v.end());
        double sum = 0;
        for (int x : v) sum += x;
        long long nz = 0;
        for (int x : v) nz += x > 0;
        printf("%-34s n %9zu  nonzero %5.1f %%  mean %6.2f  p50 %3d  p90 %3d  p99 %3d  max %3d\n", name, v.size(), 100.0 * nz / v.size(),
               sum / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
    };
    stats("forward levels per act()", af);
    stats("backward levels per act()", ab);
    stats("forward levels per wavefront (max)", wf);
    stats("backward levels per wavefront (max)", wb);
    printf("%lld act() calls\n", acts);
    return 0;
}
