// SELL-64: sliced-ELLPACK companion of a CSR operator for the fine-level products (operator SpMV, smoother sweeps).
//
// Why: P1 rows are short and nearly uniform (15 entries in the interior of a tetrahedral mesh).  The CSR kernel
// (spmv.hip) gives a row to 8 lanes, so a wave reads 8 separated 128-byte chunks per step and touches most cache
// lines twice.  In SELL-64 a slice = 64 consecutive rows = one wave; entries are stored column-major inside the
// slice (entry k of row r at base + 64 k + r), padded to the longest row of the slice.  Every load instruction of a
// wave is then one fully coalesced 1 KiB (values) / 256 B (columns) transaction, each lane owns a whole row (no
// shuffle reduction, no row-pointer chasing: the slice offset is wave-uniform), and y is written as one coalesced
// 1 KiB store.  Padding on the annulus operator is ~2 % (interior rows all have 15 entries).
// Algorithmic bytes are still counted with the CSR formula nnz*20 + n*36.
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "hx_common.h"

#include "sell_internal.h"

// max over |re|, |im| of a complex128 array (non-negative floats order like their bit patterns)
__global__ void k_absmax(int64_t n, const zc* __restrict__ v, unsigned int* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const zc a = v[i];
    m = fmaxf(m, fmaxf(fabsf((float)a.x), fabsf((float)a.y)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

template <>
__device__ __forceinline__ ch czero<ch>() { return __floats2half2_rn(0.f, 0.f); }

__global__ void k_sell_widths(int32_t n, int32_t n_slices, const int32_t* __restrict__ rowptr, int64_t* __restrict__ slice_sz) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int32_t row = slice * SELL_C + lane;
  int len = row < n ? rowptr[row + 1] - rowptr[row] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
  if (lane == 0) slice_sz[slice] = (int64_t)len * SELL_C;
}

// Union of the rows' column deltas, slice by slice: every lane walks its (sorted) row, the wave repeatedly takes the
// smallest pending delta.  wu[slice] = length of the union if it stays within cap(width) = width + width/8 (at most 64),
// else 0; the list itself goes to tmp[slice * 64 ...].
__global__ void k_sell_stencil_union(int32_t n, int32_t n_slices, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                     const int64_t* __restrict__ slice_sz, int32_t* __restrict__ wu, int32_t* __restrict__ tmp) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int width = (int)(slice_sz[slice] / SELL_C);
  int cap = width + width / 8;
  if (cap > 64) cap = 64;
  const int32_t row = slice * SELL_C + lane;
  int32_t p = row < n ? rowptr[row] : 0;
  const int32_t e = row < n ? rowptr[row + 1] : 0;
  int cnt = 0;
  bool ok = width > 0;
  while (ok) {
    long long d = p < e ? (long long)col[p] - row : (1ll << 40);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(d, o, 64); d = t < d ? t : d; }
    if (d == (1ll << 40)) break;                       // every lane is through
    if (cnt == cap) { ok = false; break; }
    if (lane == 0) tmp[(int64_t)slice * 64 + cnt] = (int32_t)d;
    ++cnt;
    if (p < e && (long long)col[p] - row == d) ++p;
  }
  if (lane == 0) wu[slice] = ok ? cnt : 0;
}
// (wu[n_slices], zero on entry and ignored by the exclusive scan that follows, collects the NUMBER of stencil slices)
__global__ void k_sell_stencil_sizes(int32_t n_slices, int32_t* __restrict__ wu, int64_t* __restrict__ slice_sz) {
  for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slices; s += gridDim.x * blockDim.x)
    if (wu[s] > 0) { slice_sz[s] = (int64_t)wu[s] * SELL_C; atomicAdd(wu + n_slices, 1); }
}
__global__ void k_sell_stencil_pack(int32_t n_slices, const int32_t* __restrict__ st_ptr, const int32_t* __restrict__ tmp,
                                    int32_t* __restrict__ st_delta) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int32_t b = st_ptr[slice], w = st_ptr[slice + 1] - b;
  if (lane < w) st_delta[b + lane] = tmp[(int64_t)slice * 64 + lane];
}

// Body width and overflow tail of the slices that are no stencil slices (see hx_sell::tptr).  Candidates: the longest row (no
// tail) and every multiple of 8 below it; cost in lane-slots of the row-sum loop: 64 (8 ceil(w / 8) + w) for the body's issue slots
// and bytes, 128 per tail step (the longest overflow decides how often the wave goes round), 3 per tail entry.
__global__ void k_sell_tail_plan(int32_t n, int32_t n_slices, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ wu,
                                 int64_t* __restrict__ slice_sz, int32_t* __restrict__ tail_sz, uint16_t* __restrict__ trow,
                                 int step_cost, int gran) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int32_t row = slice * SELL_C + lane;
  const int len = row < n ? rowptr[row + 1] - rowptr[row] : 0;
  int M = len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M = max(M, __shfl_xor(M, o, 64));
  int best_w = M;
  if (!(wu && wu[slice] > 0)) {
    long best = 64l * (8 * ((M + 7) / 8) + M);
    for (int w = 8; w < M; w += gran) {
      int T = max(len - w, 0);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) T += __shfl_xor(T, o, 64);
      const int trip = M - w;
      if (trip > 63 || T > 1023) continue;
      const long cost = 64l * (8 * ((w + 7) / 8) + w) + (long)step_cost * trip + 3l * T;
      if (cost < best) { best = cost; best_w = w; }
    }
  }
  const int ex = max(len - best_w, 0);
  int incl = ex;                                   // inclusive prefix sum over the lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  trow[(int64_t)slice * SELL_C + lane] = (uint16_t)(((incl - ex) << 6) | ex);
  if (lane == 63) { tail_sz[slice] = incl; if (incl > 0) atomicAdd(tail_sz + n_slices + 1, 1); }     // (+1: behind the scan's total)
  if (lane == 0 && best_w != M) slice_sz[slice] = (int64_t)best_w * SELL_C;
}

template <typename VOUT>
__device__ __forceinline__ VOUT sell_store_value(zc v, float inv_scale) {
  if constexpr (sizeof(VOUT) == sizeof(zc)) return v;
  else if constexpr (sizeof(VOUT) == sizeof(cc)) return make_float2((float)v.x, (float)v.y);
  else return __floats2half2_rn((float)v.x * inv_scale, (float)v.y * inv_scale);
}

template <typename VOUT>
__global__ void k_sell_fill(int32_t n, int32_t n_slices, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                            const zc* __restrict__ val, const int64_t* __restrict__ slice_ptr, int32_t* __restrict__ scol,
                            VOUT* __restrict__ sval, float inv_scale, const int32_t* __restrict__ st_ptr,
                            const int32_t* __restrict__ st_delta, int32_t n_cols, const int32_t* __restrict__ tptr,
                            const uint16_t* __restrict__ trow, int32_t* __restrict__ tcol, VOUT* __restrict__ tval) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int64_t base = slice_ptr[slice];
  const int width = (int)((slice_ptr[slice + 1] - base) / SELL_C);
  const int32_t row = slice * SELL_C + lane;
  const int32_t s = row < n ? rowptr[row] : 0, len = row < n ? rowptr[row + 1] - s : 0;
  const int32_t pad_col = row < n ? row : 0;
  if (st_ptr && st_ptr[slice + 1] > st_ptr[slice]) {
    // stencil slice: entry k sits at the position of its delta in the slice's list; absent entries are zeros whose
    // (clamped) column only has to be a valid address
    const int32_t* U = st_delta + st_ptr[slice];
    int32_t p = 0;
    for (int k = 0; k < width; ++k) {
      const int64_t o = base + (int64_t)k * SELL_C + lane;
      const int64_t c = (int64_t)row + U[k];
      const bool hit = p < len && (int64_t)col[s + p] == c;
      if (scol) scol[o] = hit ? col[s + p] : (int32_t)(c < 0 ? 0 : (c >= n_cols ? n_cols - 1 : c));
      if (hit) {
        const zc v = val[s + p];
        ++p;
        if constexpr (sizeof(VOUT) == sizeof(zc)) sval[o] = v;
        else if constexpr (sizeof(VOUT) == sizeof(cc)) sval[o] = make_float2((float)v.x, (float)v.y);
        else sval[o] = __floats2half2_rn((float)v.x * inv_scale, (float)v.y * inv_scale);
      } else sval[o] = czero<VOUT>();
    }
    return;
  }
  for (int k = 0; k < width; ++k) {
    const int64_t o = base + (int64_t)k * SELL_C + lane;
    if (k < len) {
      const zc v = val[s + k];
      if (scol) scol[o] = col[s + k];
      if constexpr (sizeof(VOUT) == sizeof(zc)) sval[o] = v;
      else if constexpr (sizeof(VOUT) == sizeof(cc)) sval[o] = make_float2((float)v.x, (float)v.y);
      else sval[o] = __floats2half2_rn((float)v.x * inv_scale, (float)v.y * inv_scale);
    } else {
      if (scol) scol[o] = pad_col;
      sval[o] = czero<VOUT>();
    }
  }
  if (tptr && len > width) {                          // what the row has beyond the body: packed into the slice's tail
    const int64_t t0 = (int64_t)tptr[slice] + (trow[(int64_t)slice * SELL_C + lane] >> 6);
    for (int k = width; k < len; ++k) {
      if (scol) tcol[t0 + k - width] = col[s + k];
      tval[t0 + k - width] = sell_store_value<VOUT>(val[s + k], inv_scale);
    }
  }
}

// per slice: do all stored (padded) entries fit a 16-bit delta?  If so write them, else flag the slice as wide.
__global__ void k_sell_narrow(int32_t n_slices, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ scol,
                              int16_t* __restrict__ dcol, uint8_t* __restrict__ wide, int32_t* __restrict__ n_wide) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int64_t base = slice_ptr[slice];
  const int width = (int)((slice_ptr[slice + 1] - base) / SELL_C);
  const int32_t row = slice * SELL_C + lane;
  int m = 0;
  for (int k = 0; k < width; ++k) {
    const int32_t d = scol[base + (int64_t)k * SELL_C + lane] - row;
    m = max(m, d < 0 ? -d : d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  const bool w = m > 32767;
  if (lane == 0) { wide[slice] = w ? 1 : 0; if (w) atomicAdd(n_wide, 1); }
  if (!w)
    for (int k = 0; k < width; ++k) {
      const int64_t o = base + (int64_t)k * SELL_C + lane;
      dcol[o] = (int16_t)(scol[o] - row);
    }
}

// flag[slice] = 1 when a stored entry of the slice (padding and the zero entries of stencil slices included: their clamped
// columns are read too) addresses a column >= n_rows, i.e. the ghost block of a row-sharded operand
__global__ void k_sell_ghost_flags(int32_t n_rows, int32_t n_slices, const int64_t* __restrict__ slice_ptr,
                                   const int32_t* __restrict__ scol, uint8_t* __restrict__ flag,
                                   const int32_t* __restrict__ tptr, const int32_t* __restrict__ tcol) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (slice >= n_slices) return;
  const int64_t base = slice_ptr[slice];
  const int width = (int)((slice_ptr[slice + 1] - base) / SELL_C);
  bool g = false;
  for (int k = 0; k < width; ++k) g = g || scol[base + (int64_t)k * SELL_C + lane] >= n_rows;
  if (tptr)
    for (int32_t t = tptr[slice] + lane; t < tptr[slice + 1]; t += SELL_C) g = g || tcol[t] >= n_rows;
  const bool any = __any(g);
  if (lane == 0) flag[slice] = any ? 1 : 0;
}

extern "C" int hx_sell_destroy(hx_sell* h) {
  if (!h) return HX_OK;
  (void)hx_free(h->slice_ptr); (void)hx_free(h->col); (void)hx_free(h->dcol); (void)hx_free(h->wide); (void)hx_free(h->val_z); (void)hx_free(h->val_c);
  (void)hx_free(h->val_h); (void)hx_free(h->st_ptr); (void)hx_free(h->st_delta);
  (void)hx_free(h->list_interior); (void)hx_free(h->list_boundary); (void)hx_free(h->dot_ws);
  (void)hx_free(h->tptr); (void)hx_free(h->trow); (void)hx_free(h->tcol); (void)hx_free(h->tval_z); (void)hx_free(h->tval_c); (void)hx_free(h->tval_h);
  delete h;
  return HX_OK;
}

// Build the SELL-64 companion of a device CSR (complex128 values).  want_z / want_c select which value copies to
// keep (complex128 for the operator product, complex64 for the single-precision V-cycle; want_c == 2 stores that copy
// as scaled complex-half instead: same complex64 arithmetic, 8 instead of 12 bytes per entry).  SYNC.
extern "C" int hx_sell_create(hx_sell** out, int32_t n, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                              int want_z, int want_c, hx_stream s_) {
  HX_REQUIRE(out && n > 0 && rowptr && col && val && (want_z || want_c) && want_c >= 0 && want_c <= 2, "bad argument");
  hipStream_t s = hx_s(s_);
  hx_sell* h = new hx_sell();
  h->n = n;
  h->n_slices = (n + SELL_C - 1) / SELL_C;
  const size_t np = (size_t)h->n_slices + 1;
  {   // number of columns = largest column index + 1 (at least n)
    int32_t nnz0 = 0, cmax = -1;
    int32_t* d_max = nullptr;
    void* t0 = nullptr;
    size_t b0 = 0;
    hipError_t e0 = hipMemcpyAsync(&nnz0, rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e0 == hipSuccess) e0 = hipStreamSynchronize(s);
    if (e0 == hipSuccess && nnz0 > 0) {
      e0 = hx_malloc((void**)&d_max, sizeof(int32_t));
      if (e0 == hipSuccess) e0 = hipcub::DeviceReduce::Max(nullptr, b0, col, d_max, nnz0, s);
      if (e0 == hipSuccess) e0 = hx_malloc(&t0, b0 ? b0 : 16);
      if (e0 == hipSuccess) e0 = hipcub::DeviceReduce::Max(t0, b0, col, d_max, nnz0, s);
      if (e0 == hipSuccess) e0 = hipMemcpyAsync(&cmax, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, s);
      if (e0 == hipSuccess) e0 = hipStreamSynchronize(s);
      (void)hx_free(t0); (void)hx_free(d_max);
    }
    if (e0 != hipSuccess) { delete h; return hx_fail(HX_ERR_HIP, __func__, hipGetErrorString(e0)); }
    h->n_cols = cmax + 1 > n ? cmax + 1 : n;
    if (h->n_cols > (1 << 27)) { delete h; return hx_fail(HX_ERR_ARG, __func__, "more than 2^27 columns (32-bit byte offsets into x)"); }
  }
  if (hx_malloc((void**)&h->slice_ptr, sizeof(int64_t) * np) != hipSuccess) { delete h; return hx_fail(HX_ERR_ALLOC, __func__, "slice_ptr"); }
  {
    const hipError_t em = hipMemsetAsync(h->slice_ptr, 0, sizeof(int64_t) * np, s);
    if (em != hipSuccess) { hx_sell_destroy(h); return hx_fail(HX_ERR_HIP, "hx_sell_create: hipMemsetAsync", hipGetErrorString(em)); }
  }
  const int grid = (int)(((int64_t)h->n_slices * 64 + 255) / 256);
  k_sell_widths<<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, h->slice_ptr);
  {
    const hipError_t el = hipGetLastError();
    if (el != hipSuccess) { hx_sell_destroy(h); return hx_fail(HX_ERR_HIP, "k_sell_widths", hipGetErrorString(el)); }
  }
  // stencil slices (HX_SELL_STENCIL=0 switches the detection off)
  const bool allow_stencil = HX_OPT_INT("sell_stencil", 1) != 0;
  int32_t* d_wu = nullptr;
  int32_t* d_tmpU = nullptr;
  if (allow_stencil && h->n_slices > 0 &&
      hx_malloc((void**)&d_wu, sizeof(int32_t) * np) == hipSuccess &&
      hx_malloc((void**)&d_tmpU, sizeof(int32_t) * 64 * (size_t)h->n_slices) == hipSuccess) {
    // (an allocation failure above only skips the stencil detection -- the format is complete without it; an error of
    // these launches is caught by the checked scan that follows on the same stream)
    if (hipMemsetAsync(d_wu, 0, sizeof(int32_t) * np, s) == hipSuccess) {
      k_sell_stencil_union<<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, col, h->slice_ptr, d_wu, d_tmpU);
      k_sell_stencil_sizes<<<hx_grid(h->n_slices, 256), 256, 0, s>>>(h->n_slices, d_wu, h->slice_ptr);
    }
  } else { (void)hx_free(d_wu); (void)hx_free(d_tmpU); d_wu = nullptr; d_tmpU = nullptr; (void)hipGetLastError(); }
  // overflow tails (HX_SELL_TAIL=0: every slice as wide as its longest row): for operators WITHOUT stencil slices only --
  // unstructured meshes.  (A structured operator has a handful of irregular slices; splitting those would put the tail code
  // into its kernels for nothing: same-box runs against round 5 showed the smoother step of the 1 M-row annulus 5 % slower.)
  // An allocation failure only skips the tails: the format is complete without.
  int32_t* d_tsz = nullptr;
  const size_t tsz_bytes = sizeof(int32_t) * (np + 1) < ((size_t)1 << 20) ? ((size_t)1 << 20) : sizeof(int32_t) * (np + 1);   // (a pooled size)
  int32_t n_stencil_slices = 0;
  if (d_wu && (hipMemcpyAsync(&n_stencil_slices, d_wu + h->n_slices, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
               hipStreamSynchronize(s) != hipSuccess)) { (void)hipGetLastError(); n_stencil_slices = 1; }
  if (HX_OPT_INT("sell_tail", 1) != 0 && h->n_slices > 0 && n_stencil_slices == 0) {
    if (hx_malloc((void**)&d_tsz, tsz_bytes) == hipSuccess &&
        hx_malloc((void**)&h->trow, sizeof(uint16_t) * SELL_C * (size_t)h->n_slices) == hipSuccess &&
        hipMemsetAsync(d_tsz, 0, sizeof(int32_t) * (np + 1), s) == hipSuccess) {
      k_sell_tail_plan<<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, d_wu, h->slice_ptr, d_tsz, h->trow,
                                           (int)HX_OPT_INT("sell_tail_step", 128), (int)(HX_OPT_INT("sell_tail_gran", 8) > 0 ? HX_OPT_INT("sell_tail_gran", 8) : 8));
    } else { (void)hx_free(d_tsz); (void)hx_free(h->trow); d_tsz = nullptr; h->trow = nullptr; (void)hipGetLastError(); }
  }
  size_t bytes = 0;
  hipError_t e = hipGetLastError();               // the width / stencil / tail launches above
  if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, h->slice_ptr, h->slice_ptr, (int)np, s);
  void* tmp = nullptr;
  if (e != hipSuccess || hx_malloc(&tmp, bytes ? bytes : 16) != hipSuccess) {
    (void)hx_free(d_wu); (void)hx_free(d_tmpU); (void)hx_free(d_tsz);
    hx_sell_destroy(h);
    return hx_fail(e != hipSuccess ? HX_ERR_HIP : HX_ERR_ALLOC, "hx_sell_create: slice scan", e != hipSuccess ? hipGetErrorString(e) : "workspace");
  }
  e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, h->slice_ptr, h->slice_ptr, (int)np, s);
  int64_t total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, h->slice_ptr + h->n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hx_free(tmp);
  if (e == hipSuccess && d_wu) {        // st_ptr = exclusive scan of the union lengths; keep it only if some slice qualified
    size_t b2 = 0;
    void* t2 = nullptr;
    int32_t n_st = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, b2, d_wu, d_wu, (int)np, s);
    if (e == hipSuccess) e = hx_malloc(&t2, b2 ? b2 : 16);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(t2, b2, d_wu, d_wu, (int)np, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&n_st, d_wu + h->n_slices, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hx_free(t2);
    if (e == hipSuccess && n_st > 0 && hx_malloc((void**)&h->st_delta, sizeof(int32_t) * (size_t)n_st) == hipSuccess) {
      h->st_ptr = d_wu;
      d_wu = nullptr;
      h->n_stencil = n_st;
      k_sell_stencil_pack<<<grid, 256, 0, s>>>(h->n_slices, h->st_ptr, d_tmpU, h->st_delta);
      e = hipStreamSynchronize(s);
    }
    (void)hx_free(d_wu); (void)hx_free(d_tmpU);
    d_wu = nullptr; d_tmpU = nullptr;
  }
  if (e == hipSuccess && d_tsz) {       // tptr = exclusive scan of the tail lengths; kept only if some slice was split
    size_t b3 = 0;
    void* t3 = nullptr;
    int32_t n_tl = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, b3, d_tsz, d_tsz, (int)np, s);
    if (e == hipSuccess) e = hx_malloc(&t3, b3 ? b3 : 16);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(t3, b3, d_tsz, d_tsz, (int)np, s);
    int32_t two[2] = {0, 0};              // total tail entries (the scan's last output), slices with a tail (the plan's counter)
    if (e == hipSuccess) e = hipMemcpyAsync(two, d_tsz + h->n_slices, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hx_free(t3);
    n_tl = two[0];
    if (e == hipSuccess && n_tl > 0) {
      h->tptr = d_tsz;
      d_tsz = nullptr;
      h->n_tail = n_tl;
      h->n_split = two[1];
    } else { (void)hx_free(h->trow); h->trow = nullptr; }
    (void)hx_free(d_tsz);
    d_tsz = nullptr;
  }
  if (e != hipSuccess) { (void)hx_free(d_wu); (void)hx_free(d_tmpU); (void)hx_free(d_tsz); hx_sell_destroy(h); return hx_fail(HX_ERR_HIP, __func__, hipGetErrorString(e)); }
  h->padded = total;
  const size_t ne = (size_t)(total > 0 ? total : 1);
  if (hx_malloc((void**)&h->col, sizeof(int32_t) * ne) != hipSuccess ||
      (want_z && hx_malloc((void**)&h->val_z, sizeof(zc) * ne) != hipSuccess) ||
      (want_c == 1 && hx_malloc((void**)&h->val_c, sizeof(cc) * ne) != hipSuccess) ||
      (want_c == 2 && hx_malloc((void**)&h->val_h, sizeof(ch) * ne) != hipSuccess)) {
    hx_sell_destroy(h);
    return hx_fail(HX_ERR_ALLOC, __func__, "SELL arrays");
  }
  if (h->tptr) {
    const size_t nt = (size_t)h->n_tail;
    if (hx_malloc((void**)&h->tcol, sizeof(int32_t) * nt) != hipSuccess ||
        (want_z && hx_malloc((void**)&h->tval_z, sizeof(zc) * nt) != hipSuccess) ||
        (want_c == 1 && hx_malloc((void**)&h->tval_c, sizeof(cc) * nt) != hipSuccess) ||
        (want_c == 2 && hx_malloc((void**)&h->tval_h, sizeof(ch) * nt) != hipSuccess)) {
      hx_sell_destroy(h);
      return hx_fail(HX_ERR_ALLOC, __func__, "SELL tail arrays");
    }
  }
  if (want_z) k_sell_fill<zc><<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, col, (const zc*)val, h->slice_ptr, h->col, h->val_z, 1.f, h->st_ptr, h->st_delta, h->n_cols,
                                                    h->tptr, h->trow, h->tcol, h->tval_z);
  if (want_c == 1) k_sell_fill<cc><<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, col, (const zc*)val, h->slice_ptr,
                                                          want_z ? nullptr : h->col, h->val_c, 1.f, h->st_ptr, h->st_delta, h->n_cols,
                                                          h->tptr, h->trow, h->tcol, h->tval_c);
  if (want_c == 2) {
    // scale so that the largest component maps to 2^14: five binary orders of head-room below the fp16 maximum, and
    // entries down to ~4e-9 of the largest stay normal numbers
    unsigned int* dmax = nullptr;
    unsigned int hmax = 0;
    int32_t nnz = 0;
    e = hx_malloc((void**)&dmax, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemsetAsync(dmax, 0, sizeof(unsigned int), s);
    if (e == hipSuccess) e = hipMemcpyAsync(&nnz, rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
      k_absmax<<<hx_grid(nnz, 256), 256, 0, s>>>(nnz, (const zc*)val, dmax);
      e = hipMemcpyAsync(&hmax, dmax, sizeof(unsigned int), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hx_free(dmax);
    if (e != hipSuccess) { hx_sell_destroy(h); return hx_fail(HX_ERR_HIP, __func__, hipGetErrorString(e)); }
    float vmax;
    memcpy(&vmax, &hmax, sizeof(float));
    h->h_scale = vmax > 0.f ? vmax / 16384.f : 1.f;
    k_sell_fill<ch><<<grid, 256, 0, s>>>(n, h->n_slices, rowptr, col, (const zc*)val, h->slice_ptr,
                                           want_z ? nullptr : h->col, h->val_h, 1.f / h->h_scale, h->st_ptr, h->st_delta, h->n_cols,
                                           h->tptr, h->trow, h->tcol, h->tval_h);
  }
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) { hx_sell_destroy(h); return hx_fail(HX_ERR_HIP, __func__, hipGetErrorString(e)); }
  HX_LAUNCH_CHECK("hx_sell_create");
  // 16-bit column deltas, slice by slice (HX_SELL_INDEX=i32 keeps the 32-bit columns everywhere)
  const bool allow16 = !HX_OPT_IS("sell_index", "i32");
  if (allow16 && total > 0) {
    int32_t* d_nw = nullptr;
    bool ok = hx_malloc((void**)&h->dcol, sizeof(int16_t) * ne) == hipSuccess &&
              hx_malloc((void**)&h->wide, (size_t)h->n_slices) == hipSuccess &&
              hx_malloc((void**)&d_nw, sizeof(int32_t)) == hipSuccess;
    if (ok) {
      (void)hipMemsetAsync(d_nw, 0, sizeof(int32_t), s);
      k_sell_narrow<<<grid, 256, 0, s>>>(h->n_slices, h->slice_ptr, h->col, h->dcol, h->wide, d_nw);
      ok = hipMemcpyAsync(&h->n_wide, d_nw, sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess &&
           hipStreamSynchronize(s) == hipSuccess;
    }
    (void)hx_free(d_nw);
    // not worth it when (nearly) every slice is wide: drop the narrow copy
    if (!ok || h->n_wide > h->n_slices / 2) {
      (void)hx_free(h->dcol); (void)hx_free(h->wide);
      h->dcol = nullptr; h->wide = nullptr; h->n_wide = h->n_slices;
    }
    (void)hipGetLastError();
  } else h->n_wide = h->n_slices;
  if (h->n_cols > n && total > 0) {          // rectangular: a row-sharded local operator -> interior / boundary slice lists
    uint8_t* d_flag = nullptr;
    std::vector<uint8_t> hf((size_t)h->n_slices);
    hipError_t ef = hx_malloc((void**)&d_flag, (size_t)h->n_slices);
    if (ef == hipSuccess) {
      k_sell_ghost_flags<<<grid, 256, 0, s>>>(n, h->n_slices, h->slice_ptr, h->col, d_flag, h->tptr, h->tcol);
      ef = hipMemcpyAsync(hf.data(), d_flag, (size_t)h->n_slices, hipMemcpyDeviceToHost, s);
      if (ef == hipSuccess) ef = hipStreamSynchronize(s);
    }
    (void)hx_free(d_flag);
    if (ef == hipSuccess) {
      std::vector<int32_t> li, lb;
      for (int32_t q = 0; q < h->n_slices; ++q) (hf[q] ? lb : li).push_back(q);
      if (!li.empty() && !lb.empty() &&
          hx_malloc((void**)&h->list_interior, sizeof(int32_t) * li.size()) == hipSuccess &&
          hx_malloc((void**)&h->list_boundary, sizeof(int32_t) * lb.size()) == hipSuccess &&
          hipMemcpy(h->list_interior, li.data(), sizeof(int32_t) * li.size(), hipMemcpyHostToDevice) == hipSuccess &&
          hipMemcpy(h->list_boundary, lb.data(), sizeof(int32_t) * lb.size(), hipMemcpyHostToDevice) == hipSuccess) {
        h->n_interior = (int32_t)li.size(); h->n_boundary = (int32_t)lb.size();
      } else { (void)hx_free(h->list_interior); (void)hx_free(h->list_boundary); h->list_interior = h->list_boundary = nullptr; }
    }
    (void)hipGetLastError();
  }
  *out = h;
  return HX_OK;
}

extern "C" int hx_sell_index_bits(const hx_sell* h) { return h ? (h->dcol ? 16 : 32) : 0; }
extern "C" int hx_sell_wide_slices(const hx_sell* h) { return h ? h->n_wide : 0; }
// number of slices stored without per-entry indices (common delta list), and the total length of their lists
__global__ void k_count_stencil(int32_t n_slices, const int32_t* __restrict__ st_ptr, int32_t* __restrict__ out) {
  for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slices; s += gridDim.x * blockDim.x)
    if (st_ptr[s + 1] > st_ptr[s]) atomicAdd(out, 1);
}
extern "C" int64_t hx_sell_stencil_entries(const hx_sell* h) { return h ? (int64_t)h->n_stencil * SELL_C : 0; }
extern "C" int hx_sell_stencil_slices(const hx_sell* h) {
  if (!h || !h->st_ptr) return 0;
  int32_t* d = nullptr;
  int32_t v = 0;
  if (hx_malloc((void**)&d, sizeof(int32_t)) != hipSuccess) return -1;
  hipError_t e = hipMemset(d, 0, sizeof(int32_t));
  if (e == hipSuccess) {
    k_count_stencil<<<hx_grid(h->n_slices, 256), 256>>>(h->n_slices, h->st_ptr, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&v, d, sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hx_free(d);
  if (e != hipSuccess) { (void)hx_fail(HX_ERR_HIP, "hx_sell_stencil_slices", hipGetErrorString(e)); return -1; }
  return v;
}

extern "C" int hx_sell_info(const hx_sell* h, int32_t* n_slices, int64_t* padded_entries) {
  HX_REQUIRE(h, "null handle");
  if (n_slices) *n_slices = h->n_slices;
  if (padded_entries) *padded_entries = h->padded;
  return HX_OK;
}

extern "C" int hx_sell_tail_info(const hx_sell* h, int64_t* tail_entries, int32_t* split_slices) {
  HX_REQUIRE(h, "null handle");
  if (tail_entries) *tail_entries = h->n_tail;
  if (split_slices) *split_slices = h->n_split;
  return HX_OK;
}

// IT = int32_t: absolute columns.  IT = int16_t: column - row (see hx_sell::dcol).
template <typename IT> __device__ __forceinline__ int32_t cidx(IT c, int32_t row);
template <> __device__ __forceinline__ int32_t cidx<int32_t>(int32_t c, int32_t) { return c; }
template <> __device__ __forceinline__ int32_t cidx<int16_t>(int16_t c, int32_t row) { return row + (int32_t)c; }

// acc += v * x with v in its storage type (complex-half: the conversions ride inside mixed-precision multiply-adds)
__device__ __forceinline__ void cmul_acc(zc& acc, zc v, zc x) { zfma(acc, v, x); }
__device__ __forceinline__ void cmul_acc(cc& acc, cc v, cc x) { zfma(acc, v, x); }
__device__ __forceinline__ void cmul_acc(cc& acc, ch v, cc x) {
  const float vr = __low2float(v), vi = __high2float(v);
  acc.x = fmaf(vr, x.x, acc.x);
  acc.y = fmaf(vr, x.y, acc.y);
  acc.x = fmaf(-vi, x.y, acc.x);
  acc.y = fmaf(vi, x.x, acc.y);
}
// Addressing: every array of a slice is reached as (wave-uniform base) + (32-bit byte offset), the form the hardware adds for
// free (scalar base + vector offset + immediate); columns are below 2^27 (checked by hx_sell_create), so offsets into x fit.
template <typename T>
__device__ __forceinline__ T ld32(const void* base, uint32_t byte_off) { return *(const T*)((const char*)base + byte_off); }
__device__ __forceinline__ int32_t clampi(int32_t v, int32_t hi) { return min(max(v, 0), hi); }

// One lane = one row of the slice: sum_k val[k] * x[col[k]].  The entries are taken G at a time: all G index loads, then all
// G value loads and G gathers of x are in flight before the first multiply-add, so a slice of the usual 15 entries costs
// two memory round trips (G = 8) instead of one per pair.  The last, partial group is padded by REPEATING the last entry
// with weight zero (a multiplication, so that the compiler keeps the loads unconditional and together): no scalar tail.
// Four accumulator chains.  cb / vb: the slice's index / value arrays (wave-uniform), lane: this row's position in them.
template <typename V, typename MV, typename IT, int G>
__device__ __forceinline__ V sell_row_dot(const IT* __restrict__ cb, const MV* __restrict__ vb, int lane, const V* __restrict__ x,
                                          int width, int32_t row) {
  typedef typename real_of<V>::type R;
  V acc[4] = {czero<V>(), czero<V>(), czero<V>(), czero<V>()};
  const uint32_t lc = (uint32_t)lane * (uint32_t)sizeof(IT), lv = (uint32_t)lane * (uint32_t)sizeof(MV);
  int k0 = 0;
  for (; k0 + G <= width; k0 += G) {
    IT ci[G];
    MV v[G];
    V xv[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      ci[j] = ld32<IT>(cb, (uint32_t)(k0 + j) * (uint32_t)(SELL_C * sizeof(IT)) + lc);
      v[j] = ld32<MV>(vb, (uint32_t)(k0 + j) * (uint32_t)(SELL_C * sizeof(MV)) + lv);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) xv[j] = ld32<V>(x, (uint32_t)cidx<IT>(ci[j], row) * (uint32_t)sizeof(V));
#pragma unroll
    for (int j = 0; j < G; ++j) cmul_acc(acc[j & 3], v[j], xv[j]);
  }
  if (k0 < width) {
    IT ci[G];
    MV v[G];
    V xv[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = min(k0 + j, width - 1);
      ci[j] = ld32<IT>(cb, (uint32_t)k * (uint32_t)(SELL_C * sizeof(IT)) + lc);
      v[j] = ld32<MV>(vb, (uint32_t)k * (uint32_t)(SELL_C * sizeof(MV)) + lv);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) xv[j] = ld32<V>(x, (uint32_t)cidx<IT>(ci[j], row) * (uint32_t)sizeof(V));
#pragma unroll
    for (int j = 0; j < G; ++j) cmul_acc(acc[j & 3], v[j], zscale((R)(k0 + j < width ? 1 : 0), xv[j]));
  }
  return zadd(zadd(acc[0], acc[1]), zadd(acc[2], acc[3]));
}

// Stencil slice: column of entry k = row + U[k], clamped for the zero entries of rows that lack that neighbour.  The
// slice's delta list (at most 64 long) sits in one register across the wave (lane k holds U[k]) and is broadcast with
// v_readlane: no memory access on the way to the address of x.
template <typename V, typename MV, int G>
__device__ __forceinline__ V sell_row_dot_stencil(int32_t u, const MV* __restrict__ vb, int lane, const V* __restrict__ x, int width,
                                                  int32_t row, int32_t n /* columns */) {
  typedef typename real_of<V>::type R;
  V acc[4] = {czero<V>(), czero<V>(), czero<V>(), czero<V>()};
  const uint32_t lv = (uint32_t)lane * (uint32_t)sizeof(MV);
  int k0 = 0;
  for (; k0 + G <= width; k0 += G) {
    MV v[G];
    V xv[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      v[j] = ld32<MV>(vb, (uint32_t)(k0 + j) * (uint32_t)(SELL_C * sizeof(MV)) + lv);
      xv[j] = ld32<V>(x, (uint32_t)clampi(row + __builtin_amdgcn_readlane(u, k0 + j), n - 1) * (uint32_t)sizeof(V));
    }
#pragma unroll
    for (int j = 0; j < G; ++j) cmul_acc(acc[j & 3], v[j], xv[j]);
  }
  if (k0 < width) {
    MV v[G];
    V xv[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = min(k0 + j, width - 1);
      v[j] = ld32<MV>(vb, (uint32_t)k * (uint32_t)(SELL_C * sizeof(MV)) + lv);
      xv[j] = ld32<V>(x, (uint32_t)clampi(row + __builtin_amdgcn_readlane(u, k), n - 1) * (uint32_t)sizeof(V));
    }
#pragma unroll
    for (int j = 0; j < G; ++j) cmul_acc(acc[j & 3], v[j], zscale((R)(k0 + j < width ? 1 : 0), xv[j]));
  }
  return zadd(zadd(acc[0], acc[1]), zadd(acc[2], acc[3]));
}

// Stencil slice with the neighbourhood of x STAGED IN LDS.  The deltas of a slice fall into a few BANDS (runs of deltas less
// than SX_GAP apart: on an extruded mesh the plane below, the own plane, the plane above); since the 64 rows of a slice are
// consecutive, entry k of all rows reads the contiguous window x[row0 + U[k] .. + 63], and a whole band reads one window
// x[row0 + first .. row0 + last + 63].  The wave loads each band's window ONCE (coalesced, clamped like the gathers), keeps
// it in its LDS region and takes the 15 operands of every row from there: 3 windows of 114 + 164 + 114 entries instead of 15
// gathers of 64 through the CU's vector-memory path.  Same operands in the same order as the gather form: identical sums.
// Returns false (wave-uniform) when the slice does not fit (more than SX_NB bands, a band wider than SX_Q * 64, more than
// SX_CAP entries in all): the caller then takes the gather form.
#define SX_NB 4
#define SX_Q 4
#define SX_CAP 640
#define SX_GAP 128
template <typename V, typename MV, int G>
__device__ __forceinline__ bool sell_row_dot_stencil_lds(int32_t u, int32_t stw, const MV* __restrict__ vb, int lane,
                                                         const V* __restrict__ x, int width, int32_t row0, int32_t n, V* lds,
                                                         V& result) {
  typedef typename real_of<V>::type R;
  const int32_t prev = __shfl_up(u, 1);
  // a band = a run of ASCENDING deltas less than SX_GAP apart (the list of a row-sharded local operator is not sorted:
  // its ghost columns sit behind the owned ones whatever their global position, so a decrease also starts a band)
  const bool start = lane < stw && (lane == 0 || u - prev >= SX_GAP || u < prev);
  unsigned long long m = __ballot(start);
  const int nb = __popcll(m);
  if (nb > SX_NB || nb == 0) return false;
  int32_t bf[SX_NB], bl[SX_NB], bo[SX_NB];
  int32_t lofs = 0, total = 0;
  bool fits = true;
#pragma unroll
  for (int b = 0; b < SX_NB; ++b) {
    bf[b] = 0;
    bl[b] = 0;
    bo[b] = total;
    if (b < nb) {
      const int sl = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int el = (m ? __ffsll((long long)m) - 1 : stw) - 1;
      const int32_t f = __builtin_amdgcn_readlane(u, sl), l = __builtin_amdgcn_readlane(u, el);
      bf[b] = f;
      bl[b] = l - f + SELL_C;
      if (lane >= sl && lane <= el) lofs = total + (u - f);
      total += bl[b];
      fits = fits && bl[b] <= SX_Q * 64;
    }
  }
  if (!fits || total > SX_CAP) return false;
  // the windows: requested together, then written to LDS
  V tmp[SX_NB][SX_Q];
#pragma unroll
  for (int b = 0; b < SX_NB; ++b)
#pragma unroll
    for (int q = 0; q < SX_Q; ++q) {
      const int32_t e = lane + 64 * q;
      if (e < bl[b]) tmp[b][q] = ld32<V>(x, (uint32_t)clampi(row0 + bf[b] + e, n - 1) * (uint32_t)sizeof(V));
    }
  V acc[4] = {czero<V>(), czero<V>(), czero<V>(), czero<V>()};
  const uint32_t lv = (uint32_t)lane * (uint32_t)sizeof(MV);
  // first group of matrix values: in flight together with the windows
  MV v0[G];
#pragma unroll
  for (int j = 0; j < G; ++j) v0[j] = ld32<MV>(vb, (uint32_t)min(j, width - 1) * (uint32_t)(SELL_C * sizeof(MV)) + lv);
#pragma unroll
  for (int b = 0; b < SX_NB; ++b)
#pragma unroll
    for (int q = 0; q < SX_Q; ++q) {
      const int32_t e = lane + 64 * q;
      if (e < bl[b]) lds[bo[b] + e] = tmp[b][q];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int k = min(j, width - 1);
    const V xv = lds[__builtin_amdgcn_readlane(lofs, k) + lane];
    cmul_acc(acc[j & 3], v0[j], zscale((R)(j < width ? 1 : 0), xv));
  }
  for (int k0 = G; k0 < width; k0 += G) {
    MV v[G];
#pragma unroll
    for (int j = 0; j < G; ++j) v[j] = ld32<MV>(vb, (uint32_t)min(k0 + j, width - 1) * (uint32_t)(SELL_C * sizeof(MV)) + lv);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = min(k0 + j, width - 1);
      const V xv = lds[__builtin_amdgcn_readlane(lofs, k) + lane];
      cmul_acc(acc[j & 3], v[j], zscale((R)(k0 + j < width ? 1 : 0), xv));
    }
  }
  result = zadd(zadd(acc[0], acc[1]), zadd(acc[2], acc[3]));
  return true;
}

// MODE 0: y = alpha*A*x + beta*yin (yin may be y itself).   MODE 1: xout = xin + omega*dinv.*(b - A*xin) + mom*(xin - xprev)  (smoother step)
// MV = storage type of the matrix values (V itself, or complex-half for V = complex64: entries are then vscale * half2).
// dcol / wide: 16-bit column deltas for the slices flagged narrow (wave-uniform branch); both null -> 32-bit everywhere.
// One wave = one slice.  The slice offsets are wave-uniform (scalar loads); the row-wise operands of the update (b, x, dinv,
// xprev / yin) are requested BEFORE the row sum so that they travel with its first group of loads.
// The row's share of the slice's overflow tail (hx_sell::tptr): every lane walks its own cnt entries, two at a time; the wave
// goes round as often as its longest overflow needs.  The first pair's index and value loads are issued BEFORE the body of the
// row sum (sell_tail_begin) so that they travel with its first group; sell_tail_end gathers their operands and does the rest.
template <typename MV>
struct SellTailRow {
  uint32_t off; int cnt;
  int32_t c0, c1;
  MV v0, v1;
  const int32_t* tc; const MV* tv;
};
template <typename MV>
__device__ __forceinline__ SellTailRow<MV> sell_tail_begin(const SellTail<MV>& tl, int32_t t0, int32_t row) {
  SellTailRow<MV> r;
  const uint32_t packed = tl.trow[row];
  r.cnt = (int)(packed & 63u);
  r.off = packed >> 6;
  r.tc = tl.tcol + t0;
  r.tv = tl.tval + t0;
  r.c0 = r.c1 = 0;
  r.v0 = r.v1 = czero<MV>();
  if (r.cnt > 0) { r.c0 = r.tc[r.off]; r.v0 = r.tv[r.off]; }
  if (r.cnt > 1) { r.c1 = r.tc[r.off + 1]; r.v1 = r.tv[r.off + 1]; }
  return r;
}
template <typename V, typename MV>
__device__ __forceinline__ V sell_tail_end(const SellTailRow<MV>& r, const V* __restrict__ x) {
  V a0 = czero<V>(), a1 = czero<V>();
  {
    V x0 = czero<V>(), x1 = czero<V>();
    if (r.cnt > 0) x0 = ld32<V>(x, (uint32_t)r.c0 * (uint32_t)sizeof(V));
    if (r.cnt > 1) x1 = ld32<V>(x, (uint32_t)r.c1 * (uint32_t)sizeof(V));
    cmul_acc(a0, r.v0, x0);
    cmul_acc(a1, r.v1, x1);
  }
  for (int j = 2; __any(j < r.cnt); j += 2) {
    const bool on0 = j < r.cnt, on1 = j + 1 < r.cnt;
    int32_t c0 = 0, c1 = 0;
    MV v0 = czero<MV>(), v1 = czero<MV>();
    if (on0) { c0 = r.tc[r.off + j]; v0 = r.tv[r.off + j]; }
    if (on1) { c1 = r.tc[r.off + j + 1]; v1 = r.tv[r.off + j + 1]; }
    V x0 = czero<V>(), x1 = czero<V>();
    if (on0) x0 = ld32<V>(x, (uint32_t)c0 * (uint32_t)sizeof(V));
    if (on1) x1 = ld32<V>(x, (uint32_t)c1 * (uint32_t)sizeof(V));
    cmul_acc(a0, v0, x0);
    cmul_acc(a1, v1, x1);
  }
  return zadd(a0, a1);
}

template <typename V>
struct SellOp {            // the scalars of the update
  V alpha, beta;
  int beta_is_zero;
  typename real_of<V>::type omega, mom, vscale;
  int ablate;              // timing-only experiment hook (HX_SELL_ABLATE; results are WRONG): bit 0 no D^-1 stream, bit 1 one entry fewer per row, bits 2..5 = q: only (16 - q) / 16 of each slice's width
};

template <typename V, int MODE, typename MV, int G, bool LDSX, bool TAILS>
__global__ void __launch_bounds__(256)
k_sell(int32_t n, int32_t n_slices, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
       const int16_t* __restrict__ dcol, const uint8_t* __restrict__ wide, const int32_t* __restrict__ st_ptr,
       const int32_t* __restrict__ st_delta, int32_t n_cols, const MV* __restrict__ val, const V* __restrict__ x, V* y,
       const V* yin, const V* __restrict__ dinv, const V* __restrict__ b, const V* __restrict__ xprev, const SellOp<V> a,
       const int32_t* __restrict__ slice_list, int32_t xcd_chunk, V* __restrict__ dot_out /* MODE 0: x^T y per slice, or NULL */,
       const SellTail<MV> tl) {
  const int lane = threadIdx.x & 63;
  // n_slices counts the waves of this launch: all slices, or the entries of slice_list (interior / boundary part)
  // xcd_chunk > 0: workgroups are dealt round-robin to the eight XCDs, so workgroup g of XCD k = blockIdx.x % 8 takes the
  // g-th group of four slices of the k-th EIGHTH of the operator: neighbouring slices (which share their x windows: the
  // plane below, the own plane, the plane above) then meet in ONE L2 instead of being fetched through every XCD's
  int64_t blk = blockIdx.x;
  if (xcd_chunk > 0) blk = (int64_t)(blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3);
  else if (xcd_chunk < 0) {      // experiment knob HX_SELL_XCD_CHUNK: super-chunks of 8 * |xcd_chunk| workgroups
    const int32_t c = -xcd_chunk, super = 8 * c, sb = blockIdx.x / super, r = blockIdx.x - sb * super;
    blk = (int64_t)sb * super + (int64_t)(r & 7) * c + (r >> 3);
  }
  const int32_t wv = __builtin_amdgcn_readfirstlane((int32_t)((blk * (int64_t)blockDim.x + threadIdx.x) >> 6));
  if (wv >= n_slices) return;
  const int32_t slice = slice_list ? __builtin_amdgcn_readfirstlane(slice_list[wv]) : wv;
  const int64_t base = slice_ptr[slice];
  int32_t width = (int32_t)((slice_ptr[slice + 1] - base) >> 6) - ((a.ablate >> 1) & 1);
  if (a.ablate >> 2) width = (width * (16 - ((a.ablate >> 2) & 15))) >> 4;
  int32_t st0 = 0, stw = 0;
  if (st_ptr) { st0 = st_ptr[slice]; stw = st_ptr[slice + 1] - st0; }
  const int32_t u = lane < stw ? st_delta[st0 + lane] : 0;       // the slice's delta list, one entry per lane
  const int32_t row = slice * SELL_C + lane;
  const bool live = row < n;
  V rb = czero<V>(), rx = czero<V>(), rd = czero<V>(), rp = czero<V>();
  if (live) {
    if (MODE == 0) {
      if (!a.beta_is_zero) rb = yin[row];
      if (dot_out) rx = x[row];                                    // (the diagonal entry gathers the same line)
    } else {
      rb = b[row];
      rx = x[row];
      if (a.ablate & 1) rd.x = 1; else rd = dinv[row];
      if (a.mom != 0 && xprev) rp = xprev[row];
    }
  }
  const MV* vb = val + base;
  // overflow tail of a split slice (wave-uniform; TAILS: the operator has tails at all -- operators without stencil slices only,
  // see hx_sell_create, so the LDS variant never carries this code)
  bool has_tail = false;
  SellTailRow<MV> tr;
  if constexpr (TAILS) {
    const int32_t t0 = tl.tptr[slice];
    has_tail = tl.tptr[slice + 1] > t0;
    if (has_tail) tr = sell_tail_begin<MV>(tl, t0, row);
  }
  V acc;
  bool staged = false;
  if constexpr (LDSX) {
    __shared__ V s_x[4][SX_CAP];
    if (stw > 0) staged = sell_row_dot_stencil_lds<V, MV, (sizeof(MV) == 4 ? 16 : G)>(u, stw, vb, lane, x, width, slice * SELL_C, n_cols,
                                                                                    s_x[threadIdx.x >> 6], acc);
  }
  if (!staged)
    acc = stw > 0 ? sell_row_dot_stencil<V, MV, G>(u, vb, lane, x, width, row, n_cols)
          : (dcol && !wide[slice]) ? sell_row_dot<V, MV, int16_t, G>(dcol + base, vb, lane, x, width, row)
                                   : sell_row_dot<V, MV, int32_t, G>(col + base, vb, lane, x, width, row);
  if constexpr (TAILS) { if (has_tail) acc = zadd(acc, sell_tail_end<V, MV>(tr, x)); }
  if constexpr (sizeof(MV) != sizeof(V)) acc = zscale(a.vscale, acc);
  if (MODE == 0) {
    V r = zmul(a.alpha, acc);
    if (!a.beta_is_zero) zfma(r, a.beta, rb);
    if (live) y[row] = r;
    if (dot_out) {                 // bilinear x^T y of the slice's rows: shuffle tree over the wave, one partial per slice
      V d = live ? zmul(rx, r) : czero<V>();
      d = zreduce_sub<HX_WAVE>(d);
      if (lane == 0) dot_out[slice] = d;
    }
    return;
  }
  if (!live) return;
  {
    V xn = rx;
    zfma(xn, zscale(a.omega, rd), zsub(rb, acc));
    if (a.mom != 0) xn = zadd(xn, zscale(a.mom, xprev ? zsub(rx, rp) : rx));
    y[row] = xn;
  }
}

static int sell_grid(int32_t n_slices) { return (n_slices + 3) / 4; }   // 4 waves (slices) per 256-thread block
// one launch site.  G = 8 entries per group of the row sums (4 ties, measured)
template <typename V, int MODE, typename MV>
static void launch_sell(const hx_sell* h, const MV* val, const V* x, V* y, V alpha, V beta, int beta_is_zero, const V* dinv,
                        const V* b, const V* xprev, typename real_of<V>::type omega, typename real_of<V>::type mom,
                        typename real_of<V>::type vscale, hx_stream s, const V* yin = nullptr, int part = 0, V* dot_out = nullptr) {
  const int ablate = (int)HX_OPT_INT("sell_ablate", 0);
  const SellOp<V> op{alpha, beta, beta_is_zero, omega, mom, vscale, MODE == 1 || sizeof(V) == 8 || (ablate >> 2) ? ablate : 0};
  // x windows of the stencil slices through LDS (HX_SELL_LDS=0: gathers everywhere; =2: also for complex128 values)
  const int lds_mode = (int)HX_OPT_INT("sell_lds", 1);
  const bool use_lds = h->st_ptr && (lds_mode == 2 || (lds_mode == 1 && sizeof(MV) == 4));
  const SellTail<MV> tl = sell_tail_of<MV>(h);
  auto kern = use_lds ? k_sell<V, MODE, MV, 8, true, false> : tl.tptr ? k_sell<V, MODE, MV, 8, false, true> : k_sell<V, MODE, MV, 8, false, false>;
  // part 0: every slice; 1 / 2: the interior / boundary slices of a row-sharded local operator (hx_sell::list_*)
  const int32_t* list = part == 1 ? h->list_interior : part == 2 ? h->list_boundary : nullptr;
  const int32_t nw = part == 1 ? h->n_interior : part == 2 ? h->n_boundary : h->n_slices;
  if (nw == 0) return;
  // XCD-contiguous schedule (see k_sell).  It pays while the pass is served from the 256 MB Infinity Cache (every launch of
  // a solve re-reads the same operator): the x windows refetched through eight L2s are then a fifth of the fabric traffic
  // (1.9 M rows: smoother step 35.4 -> 29.0 us, residual 33.5 -> 26.1; 1 M rows: operator product 45.4 -> 41.5 us).  Beyond
  // that size the pass streams from HBM, the refetches come from the cache beside it and the interleaved schedule is as fast or
  // faster (5 M rows: 95 vs 98 us).  HX_SELL_XCD = 0 / 3: never / always (bit 0 complex64 kernels, bit 1 complex128).
  const int xcd_mode = (int)HX_OPT_INT("sell_xcd", -1);
  const double footprint = (double)h->padded * sizeof(MV) + (double)h->n * sizeof(V) * (MODE == 1 ? 5 : 3);
  const bool xcd = sell_grid(nw) >= 64 &&
                   (xcd_mode < 0 ? footprint <= 288e6 : (xcd_mode & (sizeof(V) == 8 ? 1 : 2)) != 0);
  // HX_SELL_XCD_CHUNK = c > 0 (experiment knob): the launch is cut into super-chunks of 8 c workgroups and each XCD takes c
  // consecutive ones of every super-chunk.  5 M rows, complex64 smoother step, same box, every setting twice
  // (profiles/r06_xcd_chunks_annulus5M.txt): interleaved 0.649 of the HBM peak, c = 4 and 16 0.648, eighths 0.614 - 0.621 -- small
  // chunks ARE the interleaved schedule as far as the memory system can tell; nothing to adopt beyond the cache-resident size.
  const int32_t want = (int32_t)HX_OPT_INT("sell_xcd_chunk", 0);
  const bool chunked = xcd && want > 0 && want < (sell_grid(nw) + 7) / 8;
  const int32_t chunk = xcd ? (chunked ? want : (sell_grid(nw) + 7) / 8) : 0;
  const int32_t xgrid = xcd ? (int32_t)(((int64_t)sell_grid(nw) + 8 * chunk - 1) / (8 * chunk)) * 8 * chunk : 0;
  kern<<<xcd ? xgrid : sell_grid(nw), 256, 0, hx_s(s)>>>(h->n, nw, h->slice_ptr, h->col, h->dcol, h->wide, h->st_ptr,
                                           h->st_delta, h->n_cols, val, x, y, yin ? yin : y, dinv, b, xprev, op, list, chunked ? -chunk : chunk, dot_out, tl);
}


// In-situ timing of the operator product (the roofline kernel of bench.py): when enabled, every k_sell<c128,0> launch is
// bracketed by a pair of HIP events on its launch stream (no synchronisation until the read-out).
#include <vector>
namespace {
// kind 0: operator product k_sell<c128,0> (hx_sell_spmv_z); kind 1: smoother step k_sell<c64,1,*> (hx_sell_smooth_c64)
struct SpmvProfile {
  bool on = false;
  int32_t min_rows = 0;
  std::vector<hipEvent_t> ev;          // 2 per launch
  size_t used = 0;
  int64_t skipped = 0;
  int64_t seen = 0;                    // (kind 1: launches offered to the sampling)
} g_prof[2];
// events are created in chunks as the launches come (a driver-length run of the 5 M-DOF bench records ~45,000 products; round 3
// stopped at 16,384 and timed the first 11 of 20 steps only); launches beyond the cap are counted in `skipped` and reported
const size_t PROF_MAX_LAUNCHES = (size_t)1 << 18, PROF_CHUNK = 4096;
static bool prof_grow(SpmvProfile& P) {
  if (P.ev.size() >= 2 * PROF_MAX_LAUNCHES) return false;
  const size_t old = P.ev.size();
  P.ev.resize(old + 2 * PROF_CHUNK);
  for (size_t i = old; i < P.ev.size(); ++i)
    if (hipEventCreate(&P.ev[i]) != hipSuccess) { (void)hipGetLastError(); P.ev.resize(i & ~(size_t)1); return i > old; }
  return true;
}
}
extern "C" int64_t hx_profile_sell_skipped(int kind) { return (kind == 0 || kind == 1) ? g_prof[kind].skipped : -1; }
extern "C" int hx_profile_sell_spmv(int enable, int32_t min_rows) {
  for (auto& P : g_prof) {
    if (enable && P.ev.empty() && !prof_grow(P)) return hx_fail(HX_ERR_HIP, "hx_profile_sell_spmv", "hipEventCreate failed");
    P.on = enable != 0;
    P.min_rows = min_rows;
    if (enable) { P.used = 0; P.skipped = 0; P.seen = 0; }
  }
  return HX_OK;
}
// number of recorded launches and the sum of their durations (SYNC: waits for the last recorded event)
extern "C" int hx_profile_sell_read(int kind, int64_t* launches, double* total_ms) {
  HX_REQUIRE(launches && total_ms && (kind == 0 || kind == 1), "bad argument");
  SpmvProfile& P = g_prof[kind];
  double tot = 0.0;
  for (size_t i = 0; i < P.used; i += 2) {
    HX_HIP(hipEventSynchronize(P.ev[i + 1]));
    float ms = 0.f;
    HX_HIP(hipEventElapsedTime(&ms, P.ev[i], P.ev[i + 1]));
    tot += ms;
  }
  *launches = (int64_t)(P.used / 2);
  *total_ms = tot;
  return HX_OK;
}
extern "C" int hx_profile_sell_spmv_read(int64_t* launches, double* total_ms) { return hx_profile_sell_read(0, launches, total_ms); }
static bool prof_begin(int kind, int32_t n, hx_stream s) {
  SpmvProfile& P = g_prof[kind];
  // The operator product (the roofline kernel) is timed at EVERY launch; the smoother step (bench.py's dominant_by_time, 2.7 times as
  // many launches) at every 8th: an event pair costs ~10 us of a launch-bound call (1 M DOF: +9 % per call with both kernels timed
  // at every launch, measured with tools/step_time_drift.py; attaching the pair to the dispatch with hipExtLaunchKernelGGL costs the
  // same).  Launches passed over by the sampling are not `skipped` (that counts launches that could not be timed).
  if (kind == 1 && P.on && n >= P.min_rows) {
    const long every = HX_OPT_INT("sell_profile_smoother_every", 8);
    if (every > 1 && (P.seen++ % every) != 0) return false;
  }
  bool rec = P.on && n >= P.min_rows && (P.used + 2 <= P.ev.size() || prof_grow(P));
  if (rec) {      // launches that are being captured into a hipGraph (V-cycle replay) cannot be bracketed by events
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(hx_s(s), &st) != hipSuccess || st != hipStreamCaptureStatusNone) rec = false;
  }
  if (P.on && !rec && n >= P.min_rows) ++P.skipped;
  if (rec && hipEventRecord(P.ev[P.used], hx_s(s)) != hipSuccess) return false;
  return rec;
}
static void prof_end(int kind, hx_stream s) {
  SpmvProfile& P = g_prof[kind];
  if (hipEventRecord(P.ev[P.used + 1], hx_s(s)) == hipSuccess) P.used += 2;
}

// ---- products by PART of a row-sharded local operator (internal; amg.hip / cocg.hip): part 0 = every slice, 1 = the slices that
// read no ghost column (may run while the halo exchange of x is in flight), 2 = the others (after it).  An operator without
// slice lists (square, or all slices of one kind) runs everything in part 1 and nothing in part 2.
int hx_sell_has_parts(const hx_sell* h) { return h && h->list_interior && h->list_boundary ? 1 : 0; }
static thread_local bool g_rec_pending[2] = {false, false};
static inline int norm_part(const hx_sell* h, int part) { return hx_sell_has_parts(h) ? part : (part == 2 ? -1 : 0); }
static inline void part_begin(int kind, const hx_sell* h, int part, hx_stream s) {
  if (part == 0 || part == 1) g_rec_pending[kind] = prof_begin(kind, h->n, s);
}
static inline void part_end(int kind, int part, hx_stream s) {
  if ((part == 0 || part == 2) && g_rec_pending[kind]) { prof_end(kind, s); g_rec_pending[kind] = false; }
}

int hx_sell_spmv_z_part(const hx_sell* h, int part, const hx_c128* x, hx_c128* y, const double alpha[2], const double beta[2],
                        hx_stream s) {
  HX_REQUIRE(h && h->val_z && x && y && alpha && beta && part >= 0 && part <= 2, "bad argument");
  const int req = part;
  part = norm_part(h, part);
  const zc a = hx_z(alpha), b = hx_z(beta);
  part_begin(0, h, req, s);
  if (part >= 0)
    launch_sell<zc, 0, zc>(h, h->val_z, (const zc*)x, (zc*)y, a, b, (b.x == 0 && b.y == 0), nullptr, nullptr, nullptr, 0.0, 0.0, 1.0, s,
                           nullptr, part);
  part_end(0, req, s);
  HX_LAUNCH_CHECK("k_sell<c128,spmv>");
  return HX_OK;
}
extern "C" int hx_sell_spmv_z(const hx_sell* h, const hx_c128* x, hx_c128* y, const double alpha[2], const double beta[2],
                              hx_stream s) {
  return hx_sell_spmv_z_part(h, 0, x, y, alpha, beta, s);
}
// y = A x and, in the same pass, *dot = x^T y (bilinear; device scalar): the p^T K p of hx_cocg without a pass over p and q.
// Every wave leaves the sum of its slice's rows, one block adds the slices in a fixed order (deterministic).
__global__ void __launch_bounds__(1024)
k_slice_dot_finish(int32_t n_slices, const zc* __restrict__ partials, zc* __restrict__ out) {
  __shared__ zc red[1024 / HX_WAVE];
  // four independent chains per thread (the loads of a trip are in flight together; a single chain made this one-block kernel
  // latency-bound: 14 us for 29 k slices), added in a fixed order
  zc v0 = make_double2(0.0, 0.0), v1 = v0, v2 = v0, v3 = v0;
  int32_t i = threadIdx.x;
  for (; i + 3 * 1024 < n_slices; i += 4 * 1024) {
    const zc a = partials[i], b = partials[i + 1024], c = partials[i + 2048], d = partials[i + 3072];
    v0 = zadd(v0, a); v1 = zadd(v1, b); v2 = zadd(v2, c); v3 = zadd(v3, d);
  }
  for (; i < n_slices; i += 1024) v0 = zadd(v0, partials[i]);
  zc v = zadd(zadd(v0, v1), zadd(v2, v3));
  v = zreduce_sub<HX_WAVE>(v);
  if ((threadIdx.x & (HX_WAVE - 1)) == 0) red[threadIdx.x / HX_WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    zc t = red[0];
    for (int w = 1; w < 1024 / HX_WAVE; ++w) t = zadd(t, red[w]);
    out[0] = t;
  }
}
int hx_sell_spmv_z_dot(const hx_sell* h, const hx_c128* x, hx_c128* y, hx_c128* dot, hx_stream s) {
  HX_REQUIRE(h && h->val_z && x && y && dot && x != y, "bad argument");
  hx_sell* m = const_cast<hx_sell*>(h);
  if (!m->dot_ws) HX_HIP(hx_malloc((void**)&m->dot_ws, sizeof(zc) * (size_t)h->n_slices));
  part_begin(0, h, 0, s);           // (in-situ timing of the operator product: this launch is one)
  launch_sell<zc, 0, zc>(h, h->val_z, (const zc*)x, (zc*)y, make_double2(1.0, 0.0), make_double2(0.0, 0.0), 1, nullptr, nullptr,
                         nullptr, 0.0, 0.0, 1.0, s, nullptr, 0, m->dot_ws);
  part_end(0, 0, s);
  HX_LAUNCH_CHECK("k_sell<c128,spmv+dot>");
  k_slice_dot_finish<<<1, 1024, 0, hx_s(s)>>>(h->n_slices, m->dot_ws, (zc*)dot);
  HX_LAUNCH_CHECK("k_slice_dot_finish");
  return HX_OK;
}
int hx_sell_residual_c64_part(const hx_sell* h, int part, const hx_c64* x, const hx_c64* b, hx_c64* r, hx_stream s) {
  HX_REQUIRE(h && (h->val_c || h->val_h) && x && b && r && part >= 0 && part <= 2, "bad argument");
  part = norm_part(h, part);
  if (part < 0) return HX_OK;
  if (h->val_h)
    launch_sell<cc, 0, ch>(h, h->val_h, (const cc*)x, (cc*)r, make_float2(-1.f, 0.f), make_float2(1.f, 0.f), 0, nullptr, nullptr,
                           nullptr, 0.f, 0.f, h->h_scale, s, (const cc*)b, part);
  else
    launch_sell<cc, 0, cc>(h, h->val_c, (const cc*)x, (cc*)r, make_float2(-1.f, 0.f), make_float2(1.f, 0.f), 0, nullptr, nullptr,
                           nullptr, 0.f, 0.f, 1.f, s, (const cc*)b, part);
  HX_LAUNCH_CHECK("k_sell<c64,residual>");
  return HX_OK;
}
int hx_sell_residual_z_part(const hx_sell* h, int part, const hx_c128* x, const hx_c128* b, hx_c128* r, hx_stream s) {
  HX_REQUIRE(h && h->val_z && x && b && r && part >= 0 && part <= 2, "bad argument");
  part = norm_part(h, part);
  if (part < 0) return HX_OK;
  launch_sell<zc, 0, zc>(h, h->val_z, (const zc*)x, (zc*)r, make_double2(-1.0, 0.0), make_double2(1.0, 0.0), 0, nullptr, nullptr,
                         nullptr, 0.0, 0.0, 1.0, s, (const zc*)b, part);
  HX_LAUNCH_CHECK("k_sell<c128,residual>");
  return HX_OK;
}
int hx_sell_smooth_c64_part(const hx_sell* h, int part, const hx_c64* dinv, const hx_c64* b, const hx_c64* xin, hx_c64* xout,
                            float omega, const hx_c64* xprev, float mom, hx_stream s) {
  HX_REQUIRE(h && (h->val_c || h->val_h) && dinv && b && xin && xout && xin != xout && part >= 0 && part <= 2, "bad argument");
  const int req = part;
  part = norm_part(h, part);
  part_begin(1, h, req, s);
  if (part >= 0) {
    if (h->val_h)
      launch_sell<cc, 1, ch>(h, h->val_h, (const cc*)xin, (cc*)xout, make_float2(0.f, 0.f), make_float2(0.f, 0.f), 1, (const cc*)dinv,
                             (const cc*)b, (const cc*)xprev, omega, mom, h->h_scale, s, nullptr, part);
    else
      launch_sell<cc, 1, cc>(h, h->val_c, (const cc*)xin, (cc*)xout, make_float2(0.f, 0.f), make_float2(0.f, 0.f), 1, (const cc*)dinv,
                             (const cc*)b, (const cc*)xprev, omega, mom, 1.f, s, nullptr, part);
  }
  part_end(1, req, s);
  HX_LAUNCH_CHECK("k_sell<c64,smooth>");
  return HX_OK;
}
int hx_sell_smooth_z_part(const hx_sell* h, int part, const hx_c128* dinv, const hx_c128* b, const hx_c128* xin, hx_c128* xout,
                          double omega, const hx_c128* xprev, double mom, hx_stream s) {
  HX_REQUIRE(h && h->val_z && dinv && b && xin && xout && xin != xout && part >= 0 && part <= 2, "bad argument");
  part = norm_part(h, part);
  if (part < 0) return HX_OK;
  launch_sell<zc, 1, zc>(h, h->val_z, (const zc*)xin, (zc*)xout, make_double2(0, 0), make_double2(0, 0), 1, (const zc*)dinv,
                         (const zc*)b, (const zc*)xprev, omega, mom, 1.0, s, nullptr, part);
  HX_LAUNCH_CHECK("k_sell<c128,smooth>");
  return HX_OK;
}
extern "C" int hx_sell_spmv_c64(const hx_sell* h, const hx_c64* x, hx_c64* y, float alpha, float beta, hx_stream s) {
  HX_REQUIRE(h && (h->val_c || h->val_h) && x && y, "bad argument");
  if (h->val_h)
    launch_sell<cc, 0, ch>(h, h->val_h, (const cc*)x, (cc*)y, make_float2(alpha, 0.f), make_float2(beta, 0.f), beta == 0.f, nullptr,
                           nullptr, nullptr, 0.f, 0.f, h->h_scale, s);
  else
    launch_sell<cc, 0, cc>(h, h->val_c, (const cc*)x, (cc*)y, make_float2(alpha, 0.f), make_float2(beta, 0.f), beta == 0.f, nullptr,
                           nullptr, nullptr, 0.f, 0.f, 1.f, s);
  HX_LAUNCH_CHECK("k_sell<c64,spmv>");
  return HX_OK;
}
// r = b - A x in one pass (internal; the V-cycle's residual: no copy of b into r first)
int hx_sell_residual_c64(const hx_sell* h, const hx_c64* x, const hx_c64* b, hx_c64* r, hx_stream s) {
  return hx_sell_residual_c64_part(h, 0, x, b, r, s);
}
int hx_sell_residual_z(const hx_sell* h, const hx_c128* x, const hx_c128* b, hx_c128* r, hx_stream s) {
  return hx_sell_residual_z_part(h, 0, x, b, r, s);
}
// smoother step (see hx_csr_jacobi_* / the Chebyshev recurrence in amg.hip); xprev may be NULL
extern "C" int hx_sell_smooth_c64(const hx_sell* h, const hx_c64* dinv, const hx_c64* b, const hx_c64* xin, hx_c64* xout,
                                  float omega, const hx_c64* xprev, float mom, hx_stream s) {
  return hx_sell_smooth_c64_part(h, 0, dinv, b, xin, xout, omega, xprev, mom, s);
}
extern "C" int hx_sell_smooth_z(const hx_sell* h, const hx_c128* dinv, const hx_c128* b, const hx_c128* xin, hx_c128* xout,
                                double omega, const hx_c128* xprev, double mom, hx_stream s) {
  return hx_sell_smooth_z_part(h, 0, dinv, b, xin, xout, omega, xprev, mom, s);
}
