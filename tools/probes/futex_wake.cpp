// The group commit's wake-up of a batch's sleeping followers, in isolation (CPU only): one futex per thread woken one after the other (what
// co_submit does) against one shared futex and FUTEX_WAKE(all).  Prints the time the waking thread is busy and the time until the last
// sleeper runs.   g++ -O2 -pthread -o /tmp/futex_wake tools/probes/futex_wake.cpp && /tmp/futex_wake 31
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <climits>
#include <cstdlib>
static long fx(void* a, int op, uint32_t v) { return syscall(SYS_futex, a, op, v, nullptr, nullptr, 0); }
static inline uint64_t now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 31, ROUNDS = 2000;
  struct alignas(64) W { std::atomic<uint32_t> w{0}; };
  std::vector<W> own(N);
  alignas(64) std::atomic<uint32_t> shared{0};
  std::atomic<int> asleep{0}, stop{0};
  std::atomic<uint64_t> last_wake{0};
  for (int mode = 0; mode < 2; mode++) {
    std::vector<std::thread> th;
    asleep = 0; stop = 0;
    for (int i = 0; i < N; i++)
      th.emplace_back([&, i, mode] {
        uint32_t gen = 0;
        for (;;) {
          asleep++;
          if (mode == 0) { while (own[i].w.load(std::memory_order_acquire) == gen) fx(&own[i].w, FUTEX_WAIT_PRIVATE, gen); gen = own[i].w.load(); }
          else { while (shared.load(std::memory_order_acquire) == gen) fx(&shared, FUTEX_WAIT_PRIVATE, gen); gen = shared.load(); }
          uint64_t t = now(), prev = last_wake.load();
          while (t > prev && !last_wake.compare_exchange_weak(prev, t)) {}
          if (stop.load()) return;
        }
      });
    uint64_t lead = 0, tail = 0;
    for (int r = 0; r < ROUNDS; r++) {
      while (asleep.load() < N) std::this_thread::yield();
      std::this_thread::sleep_for(std::chrono::microseconds(100));  // (they are in the futex by now)
      asleep = 0;
      if (r == ROUNDS - 1) stop = 1;
      const uint64_t t0 = now();
      if (mode == 0) for (int i = 0; i < N; i++) { own[i].w.fetch_add(1, std::memory_order_release); fx(&own[i].w, FUTEX_WAKE_PRIVATE, 1); }
      else { shared.fetch_add(1, std::memory_order_release); fx(&shared, FUTEX_WAKE_PRIVATE, INT_MAX); }
      const uint64_t t1 = now();
      lead += t1 - t0;
      while (asleep.load() < N && !stop.load()) std::this_thread::yield();
      if (!stop.load()) tail += last_wake.load() - t0;
    }
    for (auto& t : th) t.join();
    printf("%s: %d sleepers, leader busy %.1f us per round, last sleeper running after %.1f us\n", mode == 0 ? "one futex each, woken one by one" : "one shared futex, FUTEX_WAKE(all)", N,
           lead / 1e3 / ROUNDS, tail / 1e3 / (ROUNDS - 1));
  }
}
