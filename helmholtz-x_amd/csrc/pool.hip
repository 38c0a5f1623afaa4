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
// (the caller streams seen since the last hx_pool_trim -- hx_s() notes a stream whenever it differs from the previous call's, up
// to 8 of them, most recent last -- and the registered internal streams); the request that takes the block out
// of the cache waits for those events on the HOST before it hands the block out, so whatever stream uses it next -- including the
// null stream of a hipMemset / hipMemcpy -- finds it idle.  Released blocks usually sit in the cache for a whole fixed-point call:
// the waits return at once.  HX_POOL=0 passes everything through.  hx_pool_trim() returns the cached blocks to the driver;
// hx_pool_stats() reports.  A caller that DESTROYS a stream it has passed to the library calls hx_pool_trim() first (the set of
// noted streams is cleared there; an event must not be recorded on a dead handle).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hx_common.h"

// the caller streams recently passed to the library (hx_s notes them) and the library's own side streams
static hipStream_t const HX_NO_STREAM = (hipStream_t)(intptr_t)-1;    // "none seen yet" (nullptr is the null stream, a real one)
std::atomic<hipStream_t> hx_last_stream{(hipStream_t)(intptr_t)-1};
std::atomic<int> hx_stream_epoch{0};          // bumped whenever the pool forgets noted streams: hx_s() then notes them again

namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;                         // pooled blocks handed out: size
  struct Free { void* p; size_t bytes; std::vector<hipEvent_t> ev; };
  std::vector<hipEvent_t> spare_events;
  std::vector<hipStream_t> caller_streams;                        // up to 8 distinct caller streams, most recent last
  std::vector<hipStream_t> internal_streams;
  int64_t n_events = 0, n_waits = 0;
  void note(hipStream_t s) {                                      // (lock held)
    if (s == HX_NO_STREAM) return;
    for (hipStream_t in : internal_streams)
      if (in == s) return;                                        // a side stream of the library: marked through its own list already
    for (size_t i = 0; i < caller_streams.size(); ++i)
      if (caller_streams[i] == s) { caller_streams.erase(caller_streams.begin() + i); break; }
    caller_streams.push_back(s);
    if (caller_streams.size() > 8) { caller_streams.erase(caller_streams.begin()); hx_stream_epoch.fetch_add(1, std::memory_order_relaxed); }
  }
  void mark(Free& f) {                                            // (lock held) one event per stream that may still use the block
    note(hx_last_stream.load(std::memory_order_relaxed));
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
  bool init = false, cap_from_env = false;
  size_t free_at_start = 0;
  void setup() {
    if (init) return;
    init = true;
    if (const char* e = getenv("HX_POOL")) on = atoi(e) != 0;
    if (const char* e = getenv("HX_POOL_MIN_BYTES")) min_bytes = (size_t)atoll(e);
    if (const char* e = getenv("HX_POOL_MAX_GIB")) { max_cached = (size_t)(atof(e) * (double)((size_t)1 << 30)); cap_from_env = true; }
    else {
      // default: a quarter of what is FREE now, shared between the local ranks that use this device (rehearsals run several
      // ranks per card; torch's allocator and the other ranks' pools cannot see this cache).  The GUESS made here -- local ranks /
      // visible devices -- is wrong in one direction only when a launcher isolates one GPU per rank (ROCR_ / HIP_VISIBLE_DEVICES:
      // 8 ranks, 1 visible device each, nobody shares): a process that sees ONE device through such a mask counts as alone on
      // it.  Python's ShardContext replaces the guess by the count of local ranks with the same PCI bus id (hx_pool_set_sharing).
      size_t free_b = 0, total_b = 0;
      int ndev = 1;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); ndev = 1; }
      int local_world = 1;
      if (const char* e = getenv("LOCAL_WORLD_SIZE")) local_world = atoi(e) > 0 ? atoi(e) : 1;
      int sharing = (local_world + ndev - 1) / ndev;
      const bool masked = getenv("ROCR_VISIBLE_DEVICES") || getenv("HIP_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES");
      if (ndev == 1 && masked && !getenv("HX_POOL_ASSUME_SHARED")) sharing = 1;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
        free_at_start = free_b;
        max_cached = free_b / 4 / (size_t)(sharing > 0 ? sharing : 1);
      } else (void)hipGetLastError();
      log_cap(sharing);
    }
  }
  void log_cap(int sharing) {
    if (getenv("HX_POOL_VERBOSE"))
      fprintf(stderr, "[hx pool] cache cap %.2f GiB (%d rank(s) on this device)\n", (double)max_cached / (double)((size_t)1 << 30), sharing);
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

void hx_pool_note_stream(hipStream_t s) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  P.note(s);
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
    hx_last_stream.compare_exchange_strong(cur, HX_NO_STREAM);
    for (size_t i = 0; i < P.caller_streams.size(); ++i)
      if (P.caller_streams[i] == s) { P.caller_streams.erase(P.caller_streams.begin() + i); break; }
    hx_stream_epoch.fetch_add(1, std::memory_order_relaxed);
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

// The number of ranks that really share THIS device (the caller counted equal PCI bus ids over the local ranks): replaces the
// start-up guess unless HX_POOL_MAX_GIB fixed the cap.
extern "C" int hx_pool_set_sharing(int ranks_on_this_device) {
  HX_REQUIRE(ranks_on_this_device >= 1, "ranks_on_this_device must be >= 1");
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  P.setup();
  if (P.cap_from_env || P.free_at_start == 0) return HX_OK;
  P.max_cached = P.free_at_start / 4 / (size_t)ranks_on_this_device;
  P.log_cap(ranks_on_this_device);
  while (P.cached > P.max_cached && !P.lru.empty()) P.drop(P.lru.begin());
  return HX_OK;
}

extern "C" int hx_pool_trim(void) {
  Pool& P = g_pool;
  std::lock_guard<std::mutex> lock(P.mu);
  while (!P.lru.empty()) P.drop(P.lru.begin());
  P.caller_streams.clear(); hx_stream_epoch.fetch_add(1, std::memory_order_relaxed);                                       // (a caller about to destroy its stream: forget it)
  hx_last_stream.store(HX_NO_STREAM);
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
