// Helper threads of the host tail (host-only).  The Horner pass over the window records is ~0.14 ms of serial field
// arithmetic on one thread -- 5 % of a 2^20 MSM and the one stage nothing else can hide -- so a call cuts it into up to
// eight pieces (host_tail.hip tail_horner_mt): the caller runs one, up to WORKERS = 7 helper threads the others.  The
// same threads invert the block products of the batched affine conversion.  Workers sleep on a condition variable
// between calls; a call posts its jobs and collects them in the order it needs them.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace msm377 {

struct TailPool {
  static constexpr int WORKERS = 7;
  // Workers asleep on their condition variable take 20-60 us to come back -- as long as their whole job (a piece of the
  // Horner chain is ~55 us) -- so a call that will need them ARMS the pool (prewake) once its accumulation kernel has
  // finished: the first `count` workers wake up while the GPU reduces the buckets (0.1-0.3 ms) and poll for their
  // job until the deadline, then go back to sleep.  Costs that many spinning cores for the length of the bucket
  // reduction, at most `spin_us` per call (MSM377_TAIL_SPIN_US, 0 = never spin).
  // Every worker has its own slot (job, generation counters, mutex, condition variable) on its own cache lines: posting
  // a job to a polling worker is two stores, no lock and no system call; only a sleeping worker is notified.
  struct alignas(128) Slot {
    std::function<void()> job;
    std::atomic<uint64_t> posted{0};  // generation of the last job handed to this worker
    std::atomic<uint64_t> done{0};    // generation it has finished
    std::atomic<bool> asleep{false};
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
  };
  Slot slot[WORKERS];
  std::atomic<int64_t> armed_until_ns{0};
  std::atomic<int> armed_count{0};
  std::atomic<bool> stop{false};
  bool started = false;
  static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool armed(int k) const { return k < armed_count.load(std::memory_order_relaxed) && now_ns() < armed_until_ns.load(std::memory_order_relaxed); }
  // The logical CPUs of the NUMA node the calling thread runs on (false: unknown).  The workers are kept on that node:
  // on a two-socket host a worker on the far socket reads the records and its job across the socket link.  (Why: the
  // tail stage was bimodal from one context to the next on some boxes, 0.077 / 0.112 ms; an A/B of 8 contexts each
  // way on another box showed 0.075-0.079 for all of them, so the cause is a hypothesis, not a measurement.)
  static bool local_node_cpus(cpu_set_t* set) {
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    for (int node = 0; node < 64; node++) {
      char path[96];
      snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
      FILE* f = fopen(path, "r");
      if (!f) break;
      char buf[4096];
      const bool got = fgets(buf, sizeof buf, f) != nullptr;
      fclose(f);
      if (!got) continue;
      CPU_ZERO(set);
      bool mine = false;
      for (char* p = buf; *p;) {  // "0-63,128-191"
        char* e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, set);
        mine |= cpu >= a && cpu <= b;
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
      }
      if (mine) return true;
    }
    return false;
  }
  bool numa_local = true;  // MSM377_TAIL_NUMA=0: leave the workers where the scheduler puts them
  void start() {
    if (started) return;
    started = true;
    cpu_set_t node_cpus;
    const bool pin = numa_local && local_node_cpus(&node_cpus);
    for (int k = 0; k < WORKERS; k++) {
      slot[k].th = std::thread([this, k] {
        Slot& me = slot[k];
        uint64_t seen = 0;
        for (;;) {
          while (me.posted.load() == seen) {  // (sequentially consistent against post(): one of the two sides sees the other)
            if (stop.load()) return;
            if (armed(k)) {
              __builtin_ia32_pause();
              continue;
            }
            std::unique_lock<std::mutex> lk(me.mu);
            me.asleep.store(true);
            me.cv.wait(lk, [&] { return stop.load() || me.posted.load() != seen || armed(k); });
            me.asleep.store(false);
          }
          seen = me.posted.load();
          me.job();
          me.done.store(seen, std::memory_order_release);
        }
      });
      if (pin) (void)pthread_setaffinity_np(slot[k].th.native_handle(), sizeof node_cpus, &node_cpus);
    }
  }
  void wake(Slot& sl) {
    if (!sl.asleep.load()) return;
    std::lock_guard<std::mutex> lk(sl.mu);  // with the lock: a worker between its predicate and its sleep must not miss this
    sl.cv.notify_one();
  }
  void prewake(int64_t spin_us, int count = WORKERS) {
    if (spin_us <= 0 || count <= 0) return;
    start();
    armed_count.store(std::min(count, (int)WORKERS));
    armed_until_ns.store(now_ns() + spin_us * 1000);
    for (int k = 0; k < std::min(count, (int)WORKERS); k++) wake(slot[k]);
  }
  void disarm() { armed_until_ns.store(0, std::memory_order_relaxed); }
  // The previous job of worker k must have been waited for (wait(k)): the slot's job is not read any more.
  void post(int k, std::function<void()> f) {
    Slot& sl = slot[k];
    sl.job = std::move(f);
    sl.posted.fetch_add(1);
    wake(sl);
  }
  // Spins: the job is a few tens of microseconds.  Bounded (MSM377_TAIL_WAIT_MS, default 2 s): a worker that died or was
  // never scheduled must not hang the caller -- false makes the entry point fail with MSM377_EHIP.  The job may still be
  // running then; the pool is poisoned (no further posts) so that nothing it captured by reference is reused.
  int64_t wait_limit_ns = 2000000000ll;
  std::atomic<bool> poisoned{false};
  bool wait(int k) {
    const uint64_t want = slot[k].posted.load(std::memory_order_acquire);
    if (slot[k].done.load(std::memory_order_acquire) == want) return true;
    const int64_t t0 = now_ns();
    for (uint32_t spins = 0; slot[k].done.load(std::memory_order_acquire) != want; spins++) {
      __builtin_ia32_pause();
      if ((spins & 0x3ff) == 0x3ff && now_ns() - t0 > wait_limit_ns) {
        poisoned.store(true);
        return false;
      }
    }
    return true;
  }
  // ---- shares: jobs that either a worker or the caller runs, exactly once ----
  // A helper thread that has lost its CPU (a busy box, another tenant) comes back milliseconds late; a call that waits
  // for it turns a 60 us tail into a 3 ms one, and one such step in twenty moves a benchmark's mean by 5 %.  So a call's
  // shares are CLAIMED: whoever gets to a share first -- its worker, or the caller when it reaches that share in its own
  // order and finds it untouched -- runs it, the other side skips it.  A worker that is still holding an earlier,
  // never-started job is not `idle` and gets no share until it has come back and dropped that job.
  struct Shares {
    std::atomic<int> state[WORKERS + 1];  // 0 unclaimed, 1 running, 2 done
    Shares() {
      for (auto& a : state) a.store(0, std::memory_order_relaxed);
    }
  };
  bool idle(int k) const { return slot[k].done.load(std::memory_order_acquire) == slot[k].posted.load(std::memory_order_acquire); }
  // The workers that can take a share now (at most `want`), in slot order.
  int idle_workers(int* out, int want) const {
    int n = 0;
    for (int k = 0; k < WORKERS && n < want; k++)
      if (idle(k)) out[n++] = k;
    return n;
  }
  // f must own (by value / shared_ptr) everything it touches: a late worker may still call it after the caller has
  // left -- it then finds the share claimed and returns without touching anything but `sh`.
  void post_share(int worker, std::shared_ptr<Shares> sh, int j, std::function<void()> f) {
    post(worker, [sh, j, f] {
      int expect = 0;
      if (sh->state[j].compare_exchange_strong(expect, 1)) {
        f();
        sh->state[j].store(2, std::memory_order_release);
      }
    });
  }
  // The caller's side of share j: run it if nobody has, else wait for whoever runs it (bounded like wait()).
  bool finish_share(const std::shared_ptr<Shares>& sh, int j, const std::function<void()>& f) {
    int expect = 0;
    if (sh->state[j].compare_exchange_strong(expect, 1)) {
      f();
      sh->state[j].store(2, std::memory_order_release);
      return true;
    }
    const int64_t t0 = now_ns();
    for (uint32_t spins = 0; sh->state[j].load(std::memory_order_acquire) != 2; spins++) {
      __builtin_ia32_pause();
      if ((spins & 0x3ff) == 0x3ff && now_ns() - t0 > wait_limit_ns) {
        poisoned.store(true);
        return false;
      }
    }
    return true;
  }
  ~TailPool() {
    if (!started) return;
    stop.store(true);
    armed_until_ns.store(0);
    for (Slot& sl : slot) {
      {
        std::lock_guard<std::mutex> lk(sl.mu);
        sl.cv.notify_one();
      }
      if (sl.th.joinable()) sl.th.join();
    }
  }
};

}  // namespace msm377
