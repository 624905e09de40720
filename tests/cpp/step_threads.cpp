// tests/cpp/step_threads.cpp — pom_step from T threads, each on a State of its own, as the reference's performance test steps one
// env per std::thread (unit_test/bboard/performance_test.cpp:40-50,71-94).  Prints one JSON line: calls per second with 1 thread
// and with T threads, and an FNV digest of every final State (the caller compares them with the oracle's).
//   step_threads <threads> <calls per thread> <states.bin: threads x 1004 B> <moves.bin: threads x calls x 4 int32>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pom_batch.h"

static uint64_t fnv(const unsigned char* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int threads = atoi(argv[1]), calls = atoi(argv[2]);
    std::vector<unsigned char> states((size_t)threads * POM_STATE_BYTES);
    std::vector<int32_t> moves((size_t)threads * calls * 4);
    FILE* f = fopen(argv[3], "rb");
    if (!f || fread(states.data(), 1, states.size(), f) != states.size()) return 3;
    fclose(f);
    f = fopen(argv[4], "rb");
    if (!f || fread(moves.data(), 4, moves.size(), f) != moves.size()) return 3;
    fclose(f);
    std::vector<unsigned char> warm(states.begin(), states.begin() + POM_STATE_BYTES);
    int32_t idle[4] = {0, 0, 0, 0};
    for (int i = 0; i < 50; i++)
        if (pom_step(warm.data(), idle)) { fprintf(stderr, "pom_step: %s\n", pom_last_error()); return 4; }
    auto run = [&](int t_count, std::vector<unsigned char>& st) {
        std::vector<std::thread> th;
        std::vector<int> rc((size_t)t_count, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < t_count; k++)
            th.emplace_back([&, k] {
                for (int i = 0; i < calls && !rc[k]; i++) rc[k] = pom_step(st.data() + (size_t)k * POM_STATE_BYTES, &moves[((size_t)k * calls + i) * 4]);
            });
        for (auto& x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int k = 0; k < t_count; k++)
            if (rc[k]) { fprintf(stderr, "pom_step failed in thread %d: %s\n", k, pom_last_error()); exit(4); }
        return (double)t_count * calls / s;
    };
    std::vector<unsigned char> one(states), all(states);
    const double r1 = run(1, one);
    const double rt = run(threads, all);
    printf("{\"threads\": %d, \"calls_per_thread\": %d, \"calls_per_s_1_thread\": %.1f, \"calls_per_s_all_threads\": %.1f, \"digests\": [", threads, calls, r1, rt);
    for (int k = 0; k < threads; k++) printf("%s\"%016llx\"", k ? ", " : "", (unsigned long long)fnv(all.data() + (size_t)k * POM_STATE_BYTES, POM_STATE_BYTES));
    printf("]}\n");
    return 0;
}
