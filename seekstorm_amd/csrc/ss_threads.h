// Host-side worker threads for the image builders (index.bin walk / decode, posting packing): plain std::thread, chunks handed out
// by an atomic counter.  SS_LOADER_THREADS overrides the count (default: the CPUs this process may run on, at most 32).
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
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
// f(begin, end, worker) over [0, n) in chunks of `grain`.  Runs inline when one chunk (or one thread) suffices.  A worker that throws
// (std::bad_alloc on a multi-GB index.bin) stops the hand-out of chunks; every thread is joined and the first exception is rethrown on
// the CALLER's thread (the loaders' entry points turn it into SS_ENOMEM, ss_guard below) -- never std::terminate with joinable
// threads.  Threads that cannot be created (std::system_error) leave their share to the others.
template <class F>
inline void ss_parallel_for(size_t n, size_t grain, F f) {
  if (n == 0) return;
  grain = std::max<size_t>(grain, 1);
  const size_t chunks = (n + grain - 1) / grain;
  const unsigned T = (unsigned)std::min<size_t>(ss_loader_threads(), chunks);
  if (T <= 1) { f((size_t)0, n, 0u); return; }
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::exception_ptr first;
  std::mutex first_mu;
  auto work = [&](unsigned w) {
    try {
      for (;;) {
        const size_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= chunks || failed.load(std::memory_order_relaxed)) break;
        f(c * grain, std::min(n, (c + 1) * grain), w);
      }
    } catch (...) {
      failed.store(true, std::memory_order_relaxed);
      std::lock_guard<std::mutex> g(first_mu);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  try {
    th.reserve(T - 1);
    for (unsigned w = 1; w < T; w++) th.emplace_back(work, w);
  } catch (...) {  // fewer helpers than planned: the chunks are handed out dynamically, nothing is lost
  }
  work(0);
  for (auto& t : th) t.join();
  if (first) std::rethrow_exception(first);
}
// the C ABI never throws: a loader entry point runs its body under this
template <class F>
inline int ss_guard(F f, int enomem, int eother) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return enomem;
  } catch (...) {
    return eother;
  }
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
