// Device-buffer pool of libhx.
//
// Every fixed-point call builds the same objects again -- SELL-64 copies of the operators (1.5 GB at 5 M DOF), level vectors,
// recycling panels (several GB), SpGEMM scratch -- and drops them when its solver objects are destroyed, as a driver that follows
// the reference does (eigensolvers.py:161 E.destroy()).  hipMalloc / hipFree of such blocks cost milliseconds each on a quiet
// machine and many times that on a loaded one (hipFree also synchronises the device); the hierarchy set-up was measured between
// 0.16 and 0.47 s for the same 5 M-DOF problem from process to process.  Blocks of HX_POOL_MIN_BYTES or more are therefore kept
// on release and handed out again to the next request of EXACTLY the same size (the next call asks for the same sizes), up to
// HX_POOL_MAX_GIB of cached memory (default: a quarter of the memory that was FREE on the device when the pool started, divided
// by the number of local ranks sharing the device -- LOCAL_WORLD_SIZE / device count; least recently released blocks go first);
// an allocation the driver refuses empties the cache and is tried once more.
// ORDERING OF REUSE.  hipFree used to synchronise the device; a cached block does not go through it, and libhx launches on more
// than one stream (the caller's stream -- possibly a non-blocking torch side stream --, the communicator's side stream for interior
// slices, the graph-replay stream).  On release a block therefore records one event on every stream that may still touch it
// (the recently seen caller streams, noted by hx_s(), and the registered internal streams); the request that takes the block out
// of the cache waits for those events on the HOST before it hands the block out, so whatever stream uses it next -- including the
// null stream of a hipMemset / hipMemcpy -- finds it idle.  Released blocks usually sit in the cache for a whole fixed-point call:
// the waits return at once.  HX_POOL=0 passes everything through.  hx_pool_trim() returns the cached blocks to the driver;
// hx_pool_stats() reports.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hx_common.h"

// the caller streams recently passed to the library (hx_s notes them) and the library's own side streams
std::atomic<hipStream_t> hx_last_stream{nullptr};

namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;                         // pooled blocks handed out: size
  struct Free { void* p; size_t bytes; std::vector<hipEvent_t> ev; };
  std::vector<hipEvent_t> spare_events;
  std::vector<hipStream_t> caller_streams;                        // up to 4 distinct caller streams, most recent last
  std::vector<hipStream_t> internal_streams;
  int64_t n_events = 0, n_waits = 0;
  void note_caller() {                                            // (lock held)
    hipStream_t s = hx_last_stream.load(std::memory_order_relaxed);
    for (size_t i = 0; i < caller_streams.size(); ++i)
      if (caller_streams[i] == s) { caller_streams.erase(caller_streams.begin() + i); break; }
    caller_streams.push_back(s);
    if (caller_streams.size() > 4) caller_streams.erase(caller_streams.begin());
  }
  void mark(Free& f) {                                            // (lock held) one event per stream that may still use the block
    note_caller();
    auto rec = [&](hipStream_t st) {
      hipEvent_t e = nullptr;
      if (!spare_events.empty()) { e = spare_events.back(); spare_events.pop_back(); }
      else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
      if (hipEventRecord(e, st) != hipSuccess) { (void)hipGetLastError(); spare_events.push_back(e); return false; }
      f.ev.push_back(e);
      ++n_events;
      return true;
    };
    bool ok = true;
    for (hipStream_t st : caller_streams) ok = rec(st) && ok;
    for (hipStream_t st : internal_streams) ok = rec(st) && ok;
    if (!ok) (void)hipDeviceSynchronize();                        // (an event could not be recorded: the old, blunt ordering)
  }
  void wait(Free& f) {                                            // (lock held) the block's last users have finished
    for (hipEvent_t e : f.ev) {
      if (hipEventSynchronize(e) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
      spare_events.push_back(e);
      ++n_waits;
    }
    f.ev.clear();
  }
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
    else {
      // default: a quarter of what is FREE now, shared between the local ranks that use this device (rehearsals run several
      // ranks per card; torch's allocator and the other ranks' pools cannot see this cache)
      size_t free_b = 0, total_b = 0;
      int ndev = 1;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); ndev = 1; }
      int local_world = 1;
      if (const char* e = getenv("LOCAL_WORLD_SIZE")) local_world = atoi(e) > 0 ? atoi(e) : 1;
      const int sharing = (local_world + ndev - 1) / ndev;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) max_cached = free_b / 4 / (size_t)(sharing > 0 ? sharing : 1);
      else (void)hipGetLastError();
    }
  }
  void drop(std::list<Free>::iterator it) {                       // (lock held) cached block -> driver
    auto range = by_size.equal_range(it->bytes);
    for (auto q = range.first; q != range.second; ++q) if (q->second == it) { by_size.erase(q); break; }
    for (hipEvent_t e : it->ev) spare_events.push_back(e);        // (hipFree synchronises the device)
    it->ev.clear();
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
      P.wait(*node);
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
  P.lru.push_back(Pool::Free{p, bytes, {}});
  P.mark(P.lru.back());
  P.by_size.emplace(bytes, std::prev(P.lru.end()));
  P.cached += bytes;
  while (P.cached > P.max_cached && !P.lru.empty()) P.drop(P.lru.begin());
  return hipSuccess;
}

// A side stream of the library (communicator: interior slices; hierarchy: graph replay) that may touch pooled blocks
void hx_pool_register_stream(hipStream_t s, bool add) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  for (size_t i = 0; i < P.internal_streams.size(); ++i)
    if (P.internal_streams[i] == s) { P.internal_streams.erase(P.internal_streams.begin() + i); break; }
  if (add) P.internal_streams.push_back(s);
  else {                                                         // a destroyed stream must not be recorded on: drop caller entries too
    hipStream_t cur = s;
    hx_last_stream.compare_exchange_strong(cur, nullptr);
    for (size_t i = 0; i < P.caller_streams.size(); ++i)
      if (P.caller_streams[i] == s) { P.caller_streams.erase(P.caller_streams.begin() + i); break; }
  }
}
// events recorded on release / waited for on reuse (bookkeeping of the ordering above; tests)
extern "C" int hx_pool_ordering_stats(int64_t* n_events, int64_t* n_waits, int64_t* max_cached_bytes) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  P.setup();
  if (n_events) *n_events = P.n_events;
  if (n_waits) *n_waits = P.n_waits;
  if (max_cached_bytes) *max_cached_bytes = (int64_t)P.max_cached;
  return HX_OK;
}

extern "C" int hx_pool_trim(void) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  while (!P.lru.empty()) P.drop(P.lru.begin());
  P.caller_streams.clear();                                       // (a caller about to destroy its stream: forget it)
  hx_last_stream.store(nullptr);
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
