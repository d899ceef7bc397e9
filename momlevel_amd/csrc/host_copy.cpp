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

// dst[i] = mask[i] ? NaN : src[i] over n elements of 4 or 8 bytes (bit patterns: no float ops, so
// signalling NaNs and denormals in src pass through untouched).
template <typename U>
void fill_masked_plain(U* __restrict__ d, const U* __restrict__ s,
                       const unsigned char* __restrict__ m, size_t n, U nan) {
  for (size_t i = 0; i < n; ++i) d[i] = m[i] ? nan : s[i];
}
#if defined(__x86_64__)
// avx2: 8 (4) mask bytes widened to 32-bit (64-bit) lanes, compared with zero, blended; the
// destination -- a page-locked staging buffer the DMA engine reads next, or a fresh array -- written
// with streaming stores once it is 32-byte aligned (as copy_stream_avx2 does).  The upload of a lazily
// read masked field runs through here piece by piece and must keep up with the 57 GB/s host link.
// Blocks of 32 elements by their 32 mask bytes: nothing masked (the ocean's interior) -> a plain
// streaming copy; everything masked (land, below the bottom: a third of an ocean field) -> NaNs
// stored, the source NOT READ; only mixed blocks (coast lines, the bottom's edge) take the
// widen-compare-blend path.  Round 6: the blend on every element made the masked upload of the
// reference's recorded call 10-13 % slower end to end than the plain one (it shares the copy team
// with the result downloads); with the fast paths it moves fewer bytes than the plain copy.
__attribute__((target("avx2"))) void fill_masked32_avx2(uint32_t* d, const uint32_t* s,
                                                          const unsigned char* m, size_t n) {
  const uint32_t nanv = 0x7FC00000u;
  size_t i = 0;
  while (i < n && (reinterpret_cast<uintptr_t>(d + i) & 31)) {
    d[i] = m[i] ? nanv : s[i];
    ++i;
  }
  const __m256i nan8 = _mm256_set1_epi32(static_cast<int>(nanv)), zero = _mm256_setzero_si256();
  for (; i + 32 <= n; i += 32) {
    const __m256i mb = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(m + i));
    const unsigned keep = static_cast<unsigned>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(mb, zero)));
    if (keep == 0xFFFFFFFFu) {  // no byte set: copy
      for (int k = 0; k < 32; k += 8)
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k),
                            _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + k)));
    } else if (keep == 0u) {  // every byte set: NaN, the source stays unread
      for (int k = 0; k < 32; k += 8) _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k), nan8);
    } else {
      for (int k = 0; k < 32; k += 8) {
        const __m128i m8 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(m + i + k));
        const __m256i k0 = _mm256_cmpeq_epi32(_mm256_cvtepu8_epi32(m8), zero);  // lanes: keep?
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + k));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k), _mm256_blendv_epi8(nan8, a, k0));
      }
    }
  }
  _mm_sfence();
  for (; i < n; ++i) d[i] = m[i] ? nanv : s[i];
}
__attribute__((target("avx2"))) void fill_masked64_avx2(uint64_t* d, const uint64_t* s,
                                                          const unsigned char* m, size_t n) {
  const uint64_t nanv = 0x7FF8000000000000ull;
  size_t i = 0;
  while (i < n && (reinterpret_cast<uintptr_t>(d + i) & 31)) {
    d[i] = m[i] ? nanv : s[i];
    ++i;
  }
  const __m256i nan4 = _mm256_set1_epi64x(static_cast<long long>(nanv)), zero = _mm256_setzero_si256();
  for (; i + 32 <= n; i += 32) {
    const __m256i mb = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(m + i));
    const unsigned keep = static_cast<unsigned>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(mb, zero)));
    if (keep == 0xFFFFFFFFu) {
      for (int k = 0; k < 32; k += 4)
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k),
                            _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + k)));
    } else if (keep == 0u) {
      for (int k = 0; k < 32; k += 4) _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k), nan4);
    } else {
      for (int k = 0; k < 32; k += 4) {
        uint32_t raw;
        std::memcpy(&raw, m + i + k, 4);  // 4 mask bytes
        const __m128i m4 = _mm_cvtsi32_si128(static_cast<int>(raw));
        const __m256i k0 = _mm256_cmpeq_epi64(_mm256_cvtepu8_epi64(m4), zero);
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + k));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + k), _mm256_blendv_epi8(nan4, a, k0));
      }
    }
  }
  _mm_sfence();
  for (; i < n; ++i) d[i] = m[i] ? nanv : s[i];
}
#endif
void fill_masked(unsigned char* d, const unsigned char* s, const unsigned char* m, size_t n,
                 int elem) {
  if (elem == 4) {
    auto* dd = reinterpret_cast<uint32_t*>(d);
    auto* ss = reinterpret_cast<const uint32_t*>(s);
#if defined(__x86_64__)
    if (have_streaming_stores()) return fill_masked32_avx2(dd, ss, m, n);
#endif
    fill_masked_plain<uint32_t>(dd, ss, m, n, 0x7FC00000u);
  } else {
    auto* dd = reinterpret_cast<uint64_t*>(d);
    auto* ss = reinterpret_cast<const uint64_t*>(s);
#if defined(__x86_64__)
    if (have_streaming_stores()) return fill_masked64_avx2(dd, ss, m, n);
#endif
    fill_masked_plain<uint64_t>(dd, ss, m, n, 0x7FF8000000000000ull);
  }
}

struct Slice {
  unsigned char* d;
  const unsigned char* s;
  size_t n;  // bytes; ELEMENTS when mask != nullptr
  bool streaming;
  std::atomic<int>* remaining;  // on the stack of the call that owns the slice
  const unsigned char* mask;    // nullptr: a plain copy; else dst = mask ? NaN : src
  int elem;                     // element size of a masked slice (4 or 8)
};

void run_slice(const Slice& j) {
  if (j.mask != nullptr) fill_masked(j.d, j.s, j.mask, j.n, j.elem);
  else copy_slice(j.d, j.s, j.n, j.streaming);
}

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
    run_slice(job);
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

// Run `parts` slices -- slice i covers units [i*per, min((i+1)*per, total)) -- slice 0 on the calling
// thread, the others on the team; returns when all are done.  No C++ exception leaves it:
// whatever cannot be handed to the team -- no team (bad_alloc), no worker thread to be had, no room
// in the queue -- is done by this thread.  Slices that WERE queued point at `remaining` on this
// stack, so from the first push_back on the function only returns once they are all done.
template <typename Make>
void run_parts(int parts, Make make) {
  if (parts <= 1) {
    run_slice(make(0, nullptr));
    return;
  }
  Team* t = nullptr;
  try {
    t = team();
  } catch (...) {
  }
  std::atomic<int> remaining(0);
  int queued = 0;  // slices 1..queued are the team's, 0 and queued+1..parts-1 this thread's
  if (t != nullptr) {
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
          t->queue.push_back(make(i, &remaining));
          ++queued;
        }
      } catch (...) {  // bad_alloc in the deque: what is queued stays queued, the rest is ours
      }
    }
    remaining.store(queued, std::memory_order_release);
  }
  if (queued > 0) t->work.notify_all();
  run_slice(make(0, nullptr));  // this thread's own share
  for (int i = queued + 1; i < parts; ++i) run_slice(make(i, nullptr));
  if (queued > 0) {
    std::unique_lock<std::mutex> lk(t->m);
    t->done.wait(lk, [&remaining] { return remaining.load(std::memory_order_acquire) == 0; });
  }
}

bool ranges_overlap(uintptr_t a, size_t na, uintptr_t b, size_t nb) { return a < b + nb && b < a + na; }

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
  if (ranges_overlap(da, nbytes, sa, nbytes))
    return mlx::detail::fail(MLX_E_SHAPE, "dst and src overlap");
  const bool stream_stores = streaming != 0 && have_streaming_stores();
  // slices of whole pages, at least 1 MiB each: below that a hand-over costs more than it saves
  size_t slice = (nbytes + static_cast<size_t>(threads) - 1) / static_cast<size_t>(threads);
  if (slice < (size_t(1) << 20)) slice = size_t(1) << 20;
  slice = (slice + 4095) & ~size_t(4095);
  const int parts = static_cast<int>((nbytes + slice - 1) / slice);
  run_parts(parts, [=](int i, std::atomic<int>* remaining) {
    const size_t off = static_cast<size_t>(i) * slice;
    const size_t n = (off + slice <= nbytes) ? slice : nbytes - off;
    return Slice{d + off, s + off, n, stream_stores, remaining, nullptr, 0};
  });
  return 0;
}

// dst[i] = mask[i] ? NaN : src[i] for n elements of elem_size 4 or 8 bytes (float32 / float64 bit
// patterns), split over the same team: how a numpy masked array -- a netCDF4 read -- becomes the
// NaN-filled array the reference is handed (momlevel_amd/labeled.py as_plain), at the host's memory
// bandwidth instead of numpy's single-threaded 0.7-1 GB/s.
extern "C" int mlx_host_copy_masked(void* dst, const void* src, const unsigned char* mask,
                                    size_t n, int elem_size, int threads) {
  if (n == 0) return 0;
  if (dst == nullptr || src == nullptr || mask == nullptr)
    return mlx::detail::fail(MLX_E_NULL, "dst, src and mask must not be NULL");
  if (elem_size != 4 && elem_size != 8)
    return mlx::detail::fail(MLX_E_ENUM, "elem_size must be 4 or 8");
  if (threads < 1 || threads > kMaxThreads)
    return mlx::detail::fail(MLX_E_SHAPE, "threads must be in 1..64");
  const size_t es = static_cast<size_t>(elem_size);
  if (n > SIZE_MAX / es) return mlx::detail::fail(MLX_E_SHAPE, "n * elem_size overflows");
  const size_t nbytes = n * es;
  auto* d = static_cast<unsigned char*>(dst);
  auto* s = static_cast<const unsigned char*>(src);
  const uintptr_t da = reinterpret_cast<uintptr_t>(d), sa = reinterpret_cast<uintptr_t>(s);
  const uintptr_t ma = reinterpret_cast<uintptr_t>(mask);
  if (da % es || sa % es) return mlx::detail::fail(MLX_E_ALIGN, "dst / src not element-aligned");
  if (nbytes > UINTPTR_MAX - da || nbytes > UINTPTR_MAX - sa || n > UINTPTR_MAX - ma)
    return mlx::detail::fail(MLX_E_SHAPE, "a range wraps around the address space");
  if (ranges_overlap(da, nbytes, sa, nbytes) || ranges_overlap(da, nbytes, ma, n))
    return mlx::detail::fail(MLX_E_SHAPE, "dst overlaps src or mask");
  // slices of whole pages of the mask, at least 256 Ki elements each
  size_t per = (n + static_cast<size_t>(threads) - 1) / static_cast<size_t>(threads);
  if (per < (size_t(1) << 18)) per = size_t(1) << 18;
  per = (per + 4095) & ~size_t(4095);
  const int parts = static_cast<int>((n + per - 1) / per);
  run_parts(parts, [=](int i, std::atomic<int>* remaining) {
    const size_t off = static_cast<size_t>(i) * per;
    const size_t cnt = (off + per <= n) ? per : n - off;
    return Slice{d + off * es, s + off * es, cnt, false, remaining, mask + off, elem_size};
  });
  return 0;
}

#endif  // !__HIP_DEVICE_COMPILE__
