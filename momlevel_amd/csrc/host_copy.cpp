// Host side of the staged transfers (momlevel_amd/hostio.py): copies between the caller's pageable
// memory and our page-locked staging buffers, split over a small team of native threads.
//
// Why native: hostio used to split every 64 MiB staging piece over a Python thread pool calling
// libc's memcpy -- 8-16 futures per piece, ~7000 per call of the reference's recorded example, each
// completion taking the GIL that the upload thread, the download thread and the caller's thread
// contend for.  One foreign call per piece (GIL released for its whole duration) removes that, and
// makes more threads per piece affordable (profiles/r04_hostio_breakdown.log).
// Why streaming stores: libc's memcpy writes its destination through the cache -- every
// destination line is first READ from DRAM (read-for-ownership) -- for data that is touched exactly
// once (a result array, a staging buffer the DMA engine reads next).  Non-temporal stores skip that
// read.  Measured on the MI355X hosts this is worth little (1-5 %: glibc switches to such stores
// itself above a size threshold); it is kept because it costs nothing.
//
// No reference counterpart: the reference hands numpy arrays to numpy.  Host code only; kept out
// of momlevel_hip.hip so that the kernel sources' hash (bench.py, profiles/) does not move with it.
// (hipcc compiles every source of the library twice; there is nothing here for the gfx950 pass)
#if !defined(__HIP_DEVICE_COMPILE__)
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

#include "../../include/momlevel_hip.h"

namespace mlx {
namespace detail {  // momlevel_hip.hip (mlx_internal.hpp; not included: it pulls in the HIP runtime)
__attribute__((visibility("hidden"))) int fail(int code, const char* msg);
}  // namespace detail
}  // namespace mlx

namespace {

#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_stream_avx2(unsigned char* d, const unsigned char* s,
                                                        size_t n) {
  // head: up to the first 32-byte boundary of the destination
  size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;
  if (head > n) head = n;
  std::memcpy(d, s, head);
  d += head, s += head, n -= head;
  size_t blocks = n / 128;
  for (size_t i = 0; i < blocks; ++i) {
    __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s));
    __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 32));
    __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 64));
    __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i*>(d), a);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(d + 32), b);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(d + 64), c);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(d + 96), e);
    s += 128, d += 128;
  }
  _mm_sfence();  // the streaming stores are globally visible before the slice is reported done
  std::memcpy(d, s, n - blocks * 128);
}
bool have_streaming_stores() { return __builtin_cpu_supports("avx2"); }
#else  // any other host: libc's memcpy (which picks its own non-temporal path above a threshold)
void copy_stream_avx2(unsigned char* d, const unsigned char* s, size_t n) { std::memcpy(d, s, n); }
bool have_streaming_stores() { return false; }
#endif

void copy_slice(unsigned char* d, const unsigned char* s, size_t n, bool streaming) {
  if (streaming && n >= 4096)
    copy_stream_avx2(d, s, n);
  else
    std::memcpy(d, s, n);
}

struct Slice {
  unsigned char* d;
  const unsigned char* s;
  size_t n;
  bool streaming;
  std::atomic<int>* remaining;  // on the stack of the call that owns the slice
};

// The team: detached workers on one queue, shared by every caller (hostio's upload and download
// threads copy at the same time).  Created on first use, grown to the largest team asked for, never
// torn down: the object is leaked on purpose so that no destructor runs under threads that still
// wait on its condition variable at process exit.
struct Team {
  std::mutex m;
  std::condition_variable work, done;
  std::deque<Slice> queue;
  int workers = 0;
};
constexpr int kMaxThreads = 64;
std::atomic<Team*> g_team{nullptr};
std::mutex g_team_init;

void worker(Team* t) {
  for (;;) {
    Slice job;
    {
      std::unique_lock<std::mutex> lk(t->m);
      t->work.wait(lk, [t] { return !t->queue.empty(); });
      job = t->queue.front();
      t->queue.pop_front();
    }
    copy_slice(job.d, job.s, job.n, job.streaming);
    if (job.remaining->fetch_sub(1, std::memory_order_acq_rel) == 1) {
      // last slice of its call: the owner either has not looked yet (it will see 0) or sleeps in
      // done.wait -- taking the mutex first orders this notify after its predicate check
      std::lock_guard<std::mutex> lk(t->m);
      t->done.notify_all();
    }
  }
}

// A forked child inherits the Team's memory but none of its threads: start over (and leak the old
// object: its mutex may have been held by a thread that does not exist here).
void forget_team_in_child() {
  g_team.store(nullptr, std::memory_order_release);
  new (&g_team_init) std::mutex();
}

Team* team() {
  Team* t = g_team.load(std::memory_order_acquire);
  if (t) return t;
  std::lock_guard<std::mutex> lk(g_team_init);
  t = g_team.load(std::memory_order_acquire);
  if (!t) {
    static bool registered = false;
    if (!registered) {
      pthread_atfork(nullptr, nullptr, forget_team_in_child);
      registered = true;
    }
    t = new Team();
    g_team.store(t, std::memory_order_release);
  }
  return t;
}

}  // namespace

extern "C" int mlx_host_copy(void* dst, const void* src, size_t nbytes, int threads,
                             int streaming) {
  if (nbytes == 0) return 0;
  if (dst == nullptr || src == nullptr)
    return mlx::detail::fail(MLX_E_NULL, "dst and src must not be NULL");
  if (threads < 1 || threads > kMaxThreads)
    return mlx::detail::fail(MLX_E_SHAPE, "threads must be in 1..64");
  auto* d = static_cast<unsigned char*>(dst);
  auto* s = static_cast<const unsigned char*>(src);
  const uintptr_t da = reinterpret_cast<uintptr_t>(d), sa = reinterpret_cast<uintptr_t>(s);
  if (nbytes > UINTPTR_MAX - da || nbytes > UINTPTR_MAX - sa)
    return mlx::detail::fail(MLX_E_SHAPE, "a range wraps around the address space");
  if (da < sa + nbytes && sa < da + nbytes)
    return mlx::detail::fail(MLX_E_SHAPE, "dst and src overlap");
  const bool stream_stores = streaming != 0 && have_streaming_stores();
  // slices of whole pages, at least 1 MiB each: below that a hand-over costs more than it saves
  size_t slice = (nbytes + static_cast<size_t>(threads) - 1) / static_cast<size_t>(threads);
  if (slice < (size_t(1) << 20)) slice = size_t(1) << 20;
  slice = (slice + 4095) & ~size_t(4095);
  const int parts = static_cast<int>((nbytes + slice - 1) / slice);
  if (parts <= 1) {
    copy_slice(d, s, nbytes, stream_stores);
    return 0;
  }
  // No C++ exception leaves this extern "C" function: whatever cannot be handed to the
  // team -- no team (bad_alloc), no worker thread to be had, no room in the queue -- is copied by
  // this thread.  Slices that WERE queued point at `remaining` on this stack, so from the first
  // push_back on the function only returns once they are all done.
  Team* t = nullptr;
  try {
    t = team();
  } catch (...) {
  }
  if (t == nullptr) {
    copy_slice(d, s, nbytes, stream_stores);
    return 0;
  }
  std::atomic<int> remaining(0);
  int queued = 0;  // slices 1..queued are the team's, 0 and queued+1..parts-1 this thread's
  {
    std::lock_guard<std::mutex> lk(t->m);  // (workers cannot take a slice before this is released)
    try {
      while (t->workers < parts - 1) {
        std::thread(worker, t).detach();
        ++t->workers;
      }
    } catch (...) {  // no more threads to be had: the workers there are do it
    }
    if (t->workers > 0) {
      try {
        for (int i = 1; i < parts; ++i) {
          const size_t off = static_cast<size_t>(i) * slice;
          const size_t n = (off + slice <= nbytes) ? slice : nbytes - off;
          t->queue.push_back(Slice{d + off, s + off, n, stream_stores, &remaining});
          ++queued;
        }
      } catch (...) {  // bad_alloc in the deque: what is queued stays queued, the rest is ours
      }
    }
    remaining.store(queued, std::memory_order_release);
  }
  if (queued > 0) t->work.notify_all();
  copy_slice(d, s, slice, stream_stores);  // this thread's own share
  const size_t done_to = static_cast<size_t>(queued + 1) * slice;
  if (done_to < nbytes) copy_slice(d + done_to, s + done_to, nbytes - done_to, stream_stores);
  if (queued > 0) {
    std::unique_lock<std::mutex> lk(t->m);
    t->done.wait(lk, [&remaining] { return remaining.load(std::memory_order_acquire) == 0; });
  }
  return 0;
}
#endif  // !__HIP_DEVICE_COMPILE__
