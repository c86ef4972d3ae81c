// Host-side worker threads for the image builders (index.bin walk / decode, posting packing): plain std::thread, chunks handed out
// by an atomic counter.  SS_LOADER_THREADS overrides the count (default: the CPUs this process may run on, at most 32).
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <new>
#include <thread>
#include <utility>
#include <vector>

inline unsigned ss_loader_threads() {
  static const unsigned n = [] {
    if (const char* e = getenv("SS_LOADER_THREADS")) return (unsigned)std::max(1, atoi(e));
    unsigned c = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) c = std::min<unsigned>(c ? c : 1u, (unsigned)CPU_COUNT(&set));
    return std::max(1u, std::min(c ? c : 1u, 32u));
  }();
  return n;
}
// f(begin, end, worker) over [0, n) in chunks of `grain`; f must not throw.  Runs inline when one chunk (or one thread) suffices.
template <class F>
inline void ss_parallel_for(size_t n, size_t grain, F f) {
  if (n == 0) return;
  grain = std::max<size_t>(grain, 1);
  const size_t chunks = (n + grain - 1) / grain;
  const unsigned T = (unsigned)std::min<size_t>(ss_loader_threads(), chunks);
  if (T <= 1) { f((size_t)0, n, 0u); return; }
  std::atomic<size_t> next{0};
  auto work = [&](unsigned w) {
    for (;;) {
      const size_t c = next.fetch_add(1, std::memory_order_relaxed);
      if (c >= chunks) break;
      f(c * grain, std::min(n, (c + 1) * grain), w);
    }
  };
  std::vector<std::thread> th;
  th.reserve(T - 1);
  for (unsigned w = 1; w < T; w++) th.emplace_back(work, w);
  work(0);
  for (auto& t : th) t.join();
}

// vector storage that is not value-initialised: resize() of the 10^7-entry block table would otherwise zero (and page in) the whole
// array on one thread before the workers fill it
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using NoInitVec = std::vector<T, NoInitAlloc<T>>;
