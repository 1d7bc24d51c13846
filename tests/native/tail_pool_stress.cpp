// Stress of csrc/tail_pool.hpp's claimed shares (host-only; built and run by tests/test_tail_pool.py, with
// -fsanitize=thread when the toolchain has it): many rounds of "post shares to the idle helpers, run the caller's own
// share, finish the others in order", with helpers that are randomly stalled BEFORE they look at their job -- the
// failure this mechanism exists for.  Every share must run exactly once per round, whoever runs it; a stalled helper's
// late look at an old job must touch nothing; no round may wait for a stalled helper that had not started.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "tail_pool.hpp"

using msm377::TailPool;

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  const int stall_every = argc > 2 ? atoi(argv[2]) : 97;  // one posted job in this many sleeps 2 ms before its claim
  TailPool pool;
  pool.numa_local = false;
  pool.start();
  struct Round {
    std::atomic<int> runs[TailPool::WORKERS + 1];
    std::atomic<int> sum{0};
  };
  uint64_t seed = 12345;
  auto rnd = [&] { seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17; return seed; };
  long stolen = 0, skipped_helpers = 0, slow_rounds = 0;
  for (int r = 0; r < rounds; r++) {
    auto rd = std::make_shared<Round>();
    for (auto& a : rd->runs) a.store(0);
    const int want = 1 + (int)(rnd() % TailPool::WORKERS);  // helpers asked for
    if (rnd() % 3 == 0) pool.prewake(200, want);
    int worker_of[TailPool::WORKERS + 1];
    const int helpers = pool.idle_workers(worker_of, want);
    skipped_helpers += want - helpers;
    auto shares = std::make_shared<TailPool::Shares>();
    auto job = [rd](int j) {
      rd->runs[j].fetch_add(1);
      volatile unsigned x = 0;
      for (int i = 0; i < 2000; i++) x += i;  // ~ a few microseconds
      rd->sum.fetch_add(j + 1);
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < helpers; j++) {
      const bool stall = stall_every > 0 && rnd() % stall_every == 0;
      // the stall sits in front of the claim: a helper that lost its CPU before it got to its job
      pool.post(worker_of[j], [shares, j, job, stall] {
        if (stall) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        int expect = 0;
        if (shares->state[j].compare_exchange_strong(expect, 1)) {
          job(j);
          shares->state[j].store(2, std::memory_order_release);
        }
      });
    }
    job(helpers);  // the caller's own share
    for (int j = helpers - 1; j >= 0; j--) {
      const int before = shares->state[j].load();
      if (!pool.finish_share(shares, j, [job, j] { job(j); })) {
        printf("FAIL: finish_share timed out in round %d\n", r);
        return 1;
      }
      if (before == 0) stolen++;
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us > 1500.0) slow_rounds++;  // a round that waited for a 2 ms stall (allowed only if the helper had claimed first: it cannot here)
    int expect_sum = 0;
    for (int j = 0; j <= helpers; j++) {
      expect_sum += j + 1;
      if (rd->runs[j].load() != 1) {
        printf("FAIL: share %d of round %d ran %d times\n", j, r, rd->runs[j].load());
        return 1;
      }
    }
    if (rd->sum.load() != expect_sum) {
      printf("FAIL: round %d sum %d != %d\n", r, rd->sum.load(), expect_sum);
      return 1;
    }
  }
  printf("ok: %d rounds, %ld shares taken over by the caller, %ld helper requests skipped (still busy), %ld rounds over 1.5 ms\n", rounds, stolen,
         skipped_helpers, slow_rounds);
  return slow_rounds > rounds / 50 ? 2 : 0;  // (scheduling noise of the test box allowed for)
}
