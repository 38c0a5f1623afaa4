// Shared device/host helpers for libhx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hx.h"

#define HX_WAVE 64

extern thread_local char hx_errbuf[512];

static inline int hx_fail(int code, const char* what, const char* detail) {
  snprintf(hx_errbuf, sizeof(hx_errbuf), "%s: %s", what, detail ? detail : "");
  return code;
}

#define HX_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) return hx_fail(HX_ERR_HIP, #call, hipGetErrorString(e_)); \
  } while (0)

#define HX_LAUNCH_CHECK(name)                                               \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) return hx_fail(HX_ERR_HIP, name, hipGetErrorString(e_)); \
  } while (0)

#define HX_REQUIRE(cond, msg) \
  do { if (!(cond)) return hx_fail(HX_ERR_ARG, __func__, msg); } while (0)

// Device buffers of the library come from its pool (pool.hip): blocks of 1 MiB or more are kept on release and reused by the next
// request of the same size -- the next fixed-point call builds the same objects again.
hipError_t hx_pool_alloc(void** out, size_t bytes);
hipError_t hx_pool_free(void* p);
template <typename T> static inline hipError_t hx_malloc(T** p, size_t bytes) { return hx_pool_alloc((void**)p, bytes); }
static inline hipError_t hx_free(void* p) { return hx_pool_free(p); }

typedef double2 zc;  // complex128 as {re, im}: one 16-byte (dwordx4) access per element

__device__ __forceinline__ zc zmul(zc a, zc b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ zc zcmul(zc a, zc b) { /* conj(a)*b */ return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ zc zadd(zc a, zc b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ zc zsub(zc a, zc b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ zc zscale(double s, zc a) { return make_double2(s * a.x, s * a.y); }
__device__ __forceinline__ void zfma(zc& acc, zc a, zc b) {
  acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ void zcfma(zc& acc, zc a, zc b) {  // acc += conj(a)*b
  acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}
__device__ __forceinline__ zc zdiv(zc a, zc b) {
  double d = b.x * b.x + b.y * b.y;
  return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
template <int W>
__device__ __forceinline__ zc zreduce_sub(zc v) {  // sum over aligned groups of W lanes
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    v.x += __shfl_down(v.x, o, W);
    v.y += __shfl_down(v.y, o, W);
  }
  return v;
}

// ---- complex64 (float2) counterparts: the AMG V-cycle may run in single precision inside flexible GMRES
typedef float2 cc;
__device__ __forceinline__ cc zmul(cc a, cc b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cc zadd(cc a, cc b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cc zsub(cc a, cc b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cc zscale(float s, cc a) { return make_float2(s * a.x, s * a.y); }
__device__ __forceinline__ void zfma(cc& acc, cc a, cc b) {
  acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x);
  acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y);
}
// real matrix entry times complex vector entry (transfer operators of the hierarchy are real)
__device__ __forceinline__ void zfma(zc& acc, double a, zc b) { acc.x = fma(a, b.x, acc.x); acc.y = fma(a, b.y, acc.y); }
__device__ __forceinline__ void zfma(cc& acc, float a, cc b) { acc.x = fmaf(a, b.x, acc.x); acc.y = fmaf(a, b.y, acc.y); }
template <int W>
__device__ __forceinline__ cc zreduce_sub(cc v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    v.x += __shfl_down(v.x, o, W);
    v.y += __shfl_down(v.y, o, W);
  }
  return v;
}
template <typename V> __device__ __forceinline__ V czero();
template <> __device__ __forceinline__ zc czero<zc>() { return make_double2(0.0, 0.0); }
template <> __device__ __forceinline__ cc czero<cc>() { return make_float2(0.0f, 0.0f); }
template <typename V> struct real_of;
template <> struct real_of<zc> { typedef double type; };
template <> struct real_of<cc> { typedef float type; };

static inline int hx_grid(int64_t work_items, int block, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
// every launch converts its stream argument here: the pool notes which caller streams may still touch a block it caches
#include <atomic>
extern std::atomic<hipStream_t> hx_last_stream;
extern std::atomic<int> hx_stream_epoch;
void hx_pool_register_stream(hipStream_t s, bool add);      // side streams of the library (see pool.hip)
void hx_pool_note_stream(hipStream_t s);                    // (slow path: takes the pool's lock)
static inline hipStream_t hx_s(hx_stream s) {
  hipStream_t st = (hipStream_t)s;
  if (hx_last_stream.load(std::memory_order_relaxed) != st) {   // a stream other than the previous call's: remember it where it is USED
    hx_last_stream.store(st, std::memory_order_relaxed);
    // the split exchanges alternate between the caller's stream and the communicator's side stream on every launch: the two
    // streams seen last by this thread are known to the pool already (no lock on that path); only a NEW stream takes it
    static thread_local hipStream_t noted[2] = {(hipStream_t)-1, (hipStream_t)-1};
    static thread_local int epoch = -1;
    const int now = hx_stream_epoch.load(std::memory_order_relaxed);
    if (epoch != now) { epoch = now; noted[0] = noted[1] = (hipStream_t)-1; }
    if (noted[0] != st && noted[1] != st) {
      noted[1] = noted[0]; noted[0] = st;
      hx_pool_note_stream(st);
    }
  }
  return st;
}
static inline zc hx_z(const double a[2]) { return make_double2(a[0], a[1]); }

// ---- run-time options (options.hip): hx_set_option overrides, else the environment variable HX_<NAME>, else the default.
// A call site keeps a static HxOpt: the value is looked up again only when the registry's generation changed.
extern std::atomic<int> hx_option_generation;
long hx_option_lookup_int(const char* name, long dflt);
double hx_option_lookup_double(const char* name, double dflt);
bool hx_option_lookup_is(const char* name, const char* value);
bool hx_option_lookup_set(const char* name);
struct HxOpt { int gen = -1; long i = 0; double d = 0.0; };
#define HX_OPT_INT(name, dflt)                                                                   \
  ([]() -> long { static HxOpt o_; const int g_ = hx_option_generation.load(std::memory_order_relaxed);  \
                  if (o_.gen != g_) { o_.i = hx_option_lookup_int(name, dflt); o_.gen = g_; } return o_.i; }())
#define HX_OPT_DOUBLE(name, dflt)                                                                \
  ([]() -> double { static HxOpt o_; const int g_ = hx_option_generation.load(std::memory_order_relaxed); \
                    if (o_.gen != g_) { o_.d = hx_option_lookup_double(name, dflt); o_.gen = g_; } return o_.d; }())
#define HX_OPT_IS(name, value)                                                                   \
  ([]() -> bool { static HxOpt o_; const int g_ = hx_option_generation.load(std::memory_order_relaxed);   \
                  if (o_.gen != g_) { o_.i = hx_option_lookup_is(name, value) ? 1 : 0; o_.gen = g_; } return o_.i != 0; }())
#define HX_OPT_SET(name)                                                                         \
  ([]() -> bool { static HxOpt o_; const int g_ = hx_option_generation.load(std::memory_order_relaxed);   \
                  if (o_.gen != g_) { o_.i = hx_option_lookup_set(name) ? 1 : 0; o_.gen = g_; } return o_.i != 0; }())
static inline long hx_option_int(const char* name, long dflt) { return hx_option_lookup_int(name, dflt); }

// ---- option "debug_bounds" (HX_DEBUG_BOUNDS=1; off by default): entry points that follow caller-supplied index arrays check
// them on the device first -- every indexed access must stay inside the ALLOCATION that holds the indexed vector
// (hipMemGetAddressRange), so a stale or foreign index array gives HX_ERR_ARG with the offending position instead of a fault of
// the device.  One reduction + one synchronising read-back per call: for test suites and bug hunts.  (vecops.hip)
//   idx: device array of `count` int32 indices (or pointer positions ptr[0] .. ptr[nvec] when `via_ptr` != null: count is then
//   read from ptr[nvec]); `vec` / `elem_bytes`: the vector they index.
int hx_debug_check_indices(const char* who, const int32_t* idx, int64_t count, const int32_t* via_ptr, int nvec,
                           const void* vec, int elem_bytes, hipStream_t st);
#define HX_DEBUG_BOUNDS(who, idx, count, via_ptr, nvec, vec, elem_bytes, st)                                         \
  do { if (HX_OPT_SET("debug_bounds") && HX_OPT_INT("debug_bounds", 0) != 0) {                                     \
         const int rc_ = hx_debug_check_indices(who, idx, count, via_ptr, nvec, vec, elem_bytes, st);                \
         if (rc_ != HX_OK) return rc_; } } while (0)
