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
    /* what the floods of one wavefront-tick would cost (VALU per wavefront) under three ways of running them — see the end */
    std::vector<int> c_now, c_seq, c_queue, it_seq, it_queue, sw_queue, c_dyn, c_dynf, jobs_f, jobs_b;
    long long acts = 0;
    for (int t = 0; t < ticks; t++) {
        for (int w = 0; w < waves; w++) {
            int mf = 0, mb = 0;
            int lf[16][4] = {}, lb[16][4] = {};
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
                    lf[e - 16 * w][i] = pom_stat_fwd;
                    lb[e - 16 * w][i] = pom_stat_bwd;
                }
                pom_oracle_step(s, mv);
                s->timeStep++;
            }
            if (t >= 100) {
                wf.push_back(mf);
                wb.push_back(mb);
                /* now: one lane per agent, a level = one dilation of a 121-bit set in 4 registers (50 / 45 VALU measured) */
                c_now.push_back(50 * mf + 45 * mb);
                /* quad-shared, member by member: the 4 lanes of an env hold one word each of ONE flood (22 VALU per level:
                 * 4 DPP word exchanges + 12 logic ops + the vote), the env's floods run one after the other, job j of all 16
                 * quads in the same loop: iterations = sum over j of the longest job j; ~30 VALU to set a job up */
                int seq = 0;
                for (int j = 0; j < 4; j++) {
                    int a = 0, b = 0;
                    for (int q = 0; q < 16; q++) {
                        a = std::max(a, lf[q][j]);
                        b = std::max(b, lb[q][j]);
                    }
                    seq += a + b;
                }
                it_seq.push_back(seq);
                c_seq.push_back(22 * seq + 8 * 30);
                /* quad-shared with a queue per quad: every quad works through its own jobs back to back; iterations = the
                 * longest queue; the job-switch path (~40 VALU: next member's source / target / gates by broadcast, sets reset)
                 * runs in every iteration in which ANY quad switches */
                int longest = 0;
                bool sw[512] = {};
                for (int q = 0; q < 16; q++) {
                    int at = 0;
                    for (int j = 0; j < 4; j++)
                        for (int pass = 0; pass < 2; pass++) {
                            const int len = pass ? lb[q][j] : lf[q][j];
                            if (!len) continue;
                            sw[at] = true;
                            at += len;
                        }
                    longest = std::max(longest, at);
                }
                int switches = 0;
                for (int k = 0; k < longest; k++) switches += sw[k];
                it_queue.push_back(longest);
                sw_queue.push_back(switches);
                c_queue.push_back(22 * longest + 40 * switches);
                /* dynamic quads: the wavefront's flood jobs (whoever's) are dealt to its 16 quads, 16 at a time, job k of a round
                 * to quad k: a round runs as long as its longest job, 22 VALU per level, ~60 VALU per round to hand the jobs out
                 * and the answers back through LDS */
                int dyn[2] = {0, 0}, nj[2] = {0, 0};
                for (int pass = 0; pass < 2; pass++) {
                    int in_round = 0, mx = 0;
                    for (int q = 0; q < 16; q++)
                        for (int j = 0; j < 4; j++) {
                            const int len = pass ? lb[q][j] : lf[q][j];
                            if (!len) continue;
                            nj[pass]++;
                            mx = std::max(mx, len);
                            if (++in_round == 16) {
                                dyn[pass] += 22 * mx + 60;
                                in_round = mx = 0;
                            }
                        }
                    if (in_round) dyn[pass] += 22 * mx + 60;
                }
                jobs_f.push_back(nj[0]);
                jobs_b.push_back(nj[1]);
                c_dyn.push_back(dyn[0] + dyn[1]);
                c_dynf.push_back(dyn[0] + 45 * mb);
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
    {
        long long hf[12] = {0}, hb[12] = {0}, nf = 0, nb = 0;
        for (int x : af) if (x) { hf[std::min(x, 11)]++; nf++; }
        for (int x : ab) if (x) { hb[std::min(x, 11)]++; nb++; }
        printf("levels of the floods that run (%% of them), 1 .. 10, 11+:\n  forward ");
        for (int k = 1; k < 12; k++) printf(" %5.1f", 100.0 * hf[k] / nf);
        printf("\n  backward");
        for (int k = 1; k < 12; k++) printf(" %5.1f", 100.0 * hb[k] / nb);
        printf("\n");
    }
    printf("\nfloods of one wavefront-tick (16 envs), estimated VALU per wavefront:\n");
    stats("now: lane = agent (50 F + 45 B)", c_now);
    stats("quad-shared, member by member", c_seq);
    stats("  its iterations", it_seq);
    stats("quad-shared, a queue per quad", c_queue);
    stats("  its iterations", it_queue);
    stats("  of them with a job switch", sw_queue);
    stats("dynamic quads, both floods", c_dyn);
    stats("dynamic quads, forward only", c_dynf);
    stats("  forward jobs per wavefront", jobs_f);
    stats("  backward jobs per wavefront", jobs_b);
    return 0;
}
