// Device-buffer pool of libhx.
//
// Every fixed-point call builds the same objects again -- SELL-64 copies of the operators (1.5 GB at 5 M DOF), level vectors,
// recycling panels (several GB), SpGEMM scratch -- and drops them when its solver objects are destroyed, as a driver that follows
// the reference does (eigensolvers.py:161 E.destroy()).  hipMalloc / hipFree of such blocks cost milliseconds each on a quiet
// machine and many times that on a loaded one (hipFree also synchronises the device); the hierarchy set-up was measured between
// 0.16 and 0.47 s for the same 5 M-DOF problem from process to process.  Blocks of HX_POOL_MIN_BYTES or more are therefore kept
// on release and handed out again to the next request of EXACTLY the same size (the next call asks for the same sizes), up to
// HX_POOL_MAX_GIB of cached memory (default: a quarter of the device's memory; least recently released blocks go first); an allocation the driver refuses empties the
// cache and is tried once more.  Reuse is ordered like torch's caching allocator orders it: by the stream -- all libhx work of a
// process runs on the caller's current stream, and handles are destroyed after their last solve has been read by the host.
// HX_POOL=0 passes everything through.  hx_pool_trim() returns the cached blocks to the driver; hx_pool_stats() reports.
#include <chrono>
#include <cstdlib>
#include <list>
#include <map>
#include <mutex>
#include <unordered_map>

#include "hx_common.h"

namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;                         // pooled blocks handed out: size
  struct Free { void* p; size_t bytes; };
  std::list<Free> lru;                                            // cached blocks, least recently released first
  std::multimap<size_t, std::list<Free>::iterator> by_size;
  size_t cached = 0;
  int64_t n_alloc = 0, n_hit = 0, n_driver = 0;
  double driver_s = 0.0;
  bool on = true;
  size_t min_bytes = (size_t)1 << 20, max_cached = (size_t)32 << 30;
  bool init = false;
  void setup() {
    if (init) return;
    init = true;
    if (const char* e = getenv("HX_POOL")) on = atoi(e) != 0;
    if (const char* e = getenv("HX_POOL_MIN_BYTES")) min_bytes = (size_t)atoll(e);
    if (const char* e = getenv("HX_POOL_MAX_GIB")) max_cached = (size_t)(atof(e) * (double)((size_t)1 << 30));
    else {                                                         // default: a quarter of the device's memory (72 GiB of 288)
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) max_cached = total_b / 4;
      else (void)hipGetLastError();
    }
  }
  void drop(std::list<Free>::iterator it) {                       // (lock held) cached block -> driver
    auto range = by_size.equal_range(it->bytes);
    for (auto q = range.first; q != range.second; ++q) if (q->second == it) { by_size.erase(q); break; }
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipFree(it->p);
    driver_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++n_driver;
    cached -= it->bytes;
    lru.erase(it);
  }
};
Pool g_pool;
}  // namespace

hipError_t hx_pool_alloc(void** out, size_t bytes) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  P.setup();
  ++P.n_alloc;
  if (!out) return hipErrorInvalidValue;
  *out = nullptr;
  if (bytes == 0) return hipSuccess;
  const bool pooled = P.on && bytes >= P.min_bytes;
  if (pooled) {
    auto it = P.by_size.find(bytes);
    if (it != P.by_size.end()) {
      auto node = it->second;
      *out = node->p;
      P.cached -= bytes;
      P.lru.erase(node);
      P.by_size.erase(it);
      P.live[*out] = bytes;
      ++P.n_hit;
      return hipSuccess;
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess && !P.lru.empty()) {                        // out of memory with blocks in the cache: give them back, once more
    (void)hipGetLastError();
    while (!P.lru.empty()) P.drop(P.lru.begin());
    e = hipMalloc(out, bytes);
  }
  P.driver_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ++P.n_driver;
  if (e == hipSuccess && pooled) P.live[*out] = bytes;
  return e;
}

hipError_t hx_pool_free(void* p) {
  if (!p) return hipSuccess;
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  P.setup();
  auto it = P.live.find(p);
  if (it == P.live.end()) {                                       // small block (or pool off): straight to the driver
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipFree(p);
    P.driver_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++P.n_driver;
    return e;
  }
  const size_t bytes = it->second;
  P.live.erase(it);
  P.lru.push_back(Pool::Free{p, bytes});
  P.by_size.emplace(bytes, std::prev(P.lru.end()));
  P.cached += bytes;
  while (P.cached > P.max_cached && !P.lru.empty()) P.drop(P.lru.begin());
  return hipSuccess;
}

extern "C" int hx_pool_trim(void) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  while (!P.lru.empty()) P.drop(P.lru.begin());
  return HX_OK;
}

extern "C" int hx_pool_stats(int64_t* n_requests, int64_t* n_reused, int64_t* n_driver_calls, double* driver_seconds,
                             int64_t* cached_bytes, int reset) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  if (n_requests) *n_requests = P.n_alloc;
  if (n_reused) *n_reused = P.n_hit;
  if (n_driver_calls) *n_driver_calls = P.n_driver;
  if (driver_seconds) *driver_seconds = P.driver_s;
  if (cached_bytes) *cached_bytes = (int64_t)P.cached;
  if (reset) { P.n_alloc = P.n_hit = P.n_driver = 0; P.driver_s = 0.0; }
  return HX_OK;
}
