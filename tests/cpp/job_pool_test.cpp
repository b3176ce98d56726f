// job_pool_test.cpp -- the worker protocol of the multi-device index (vaq_amd/csrc/job_pool.h) on the
// CPU, meant to be built with -fsanitize=thread and -fsanitize=address (tests/test_cabi_cpu.py does
// both): G = 2..8 workers, thousands of two-phase "searches" whose exchange phase is only entered
// when every shard's first phase succeeded, shards failing at random, concurrent callers serialised
// by a mutex as vaqhip_multi's are, start / stop cycles.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "job_pool.h"

#define CHECK(c)                                                            \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

int main() {
  for (int G = 2; G <= 8; G++) {
    vaq::JobPool pool;
    pool.start(G);
    CHECK(pool.size() == G);
    std::vector<int> packed((size_t)G, 0), gathered((size_t)G * G, 0);
    std::mutex mu;  // one call at a time, like vaqhip_multi::mu
    std::atomic<int> exchanged{0}, failed{0};
    auto search = [&](unsigned seed, int fail_shard) {
      std::lock_guard<std::mutex> lk(mu);
      // phase 1: every shard "searches"; one of them may fail
      const int rc1 = pool.run([&](int g) -> int {
        if (g == fail_shard) return -7;
        packed[(size_t)g] = (int)(seed * 131u + (unsigned)g);
        return 0;
      });
      if (rc1) {
        CHECK(rc1 == -7 && pool.rc(fail_shard) == -7);
        for (int g = 0; g < G; g++) CHECK(g == fail_shard || pool.rc(g) == 0);
        failed++;
        return;  // the exchange phase is never entered: nobody waits for the failed shard
      }
      // phase 2: the "all-gather": every shard reads every shard's packed word
      const int rc2 = pool.run([&](int g) -> int {
        for (int h = 0; h < G; h++) gathered[(size_t)g * G + h] = packed[(size_t)h];
        return 0;
      });
      CHECK(rc2 == 0);
      for (int g = 0; g < G; g++)
        for (int h = 0; h < G; h++) CHECK(gathered[(size_t)g * G + h] == (int)(seed * 131u + (unsigned)h));
      exchanged++;
    };
    std::vector<std::thread> callers;
    for (int c = 0; c < 3; c++)
      callers.emplace_back([&, c] {
        std::mt19937 rng(1234u + (unsigned)c * 77u + (unsigned)G);
        for (int i = 0; i < 400; i++) {
          const int fs = (rng() % 5u == 0u) ? (int)(rng() % (unsigned)G) : -1;
          search((unsigned)rng(), fs);
        }
      });
    for (auto &t : callers) t.join();
    CHECK(exchanged + failed == 1200 && failed > 0 && exchanged > 0);
    pool.stop();
    pool.stop();  // idempotent
  }
  // a pool that is started and dropped without ever running a job
  {
    vaq::JobPool idle;
    idle.start(4);
  }
  std::puts("job_pool_test: ok");
  return 0;
}
