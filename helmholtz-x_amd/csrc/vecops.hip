// Complex128 vector / tall-skinny panel kernels (PETSc Vec + SLEPc BV equivalents).
// All are HBM-bound streaming kernels: 16-byte (dwordx4) accesses, grid-stride, 256-thread blocks.
#include "hx_common.h"

thread_local char hx_errbuf[512] = "";

extern "C" int hx_version(void) { return 100; }
extern "C" const char* hx_last_error(void) { return hx_errbuf; }

extern "C" int hx_device_info(char* name, int name_len, int* cu_count, size_t* hbm_bytes) {
  int dev = 0;
  HX_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HX_HIP(hipGetDeviceProperties(&p, dev));
  if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return HX_OK;
}

__global__ void k_zaxpy(int64_t n, zc a, const zc* __restrict__ x, zc* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc xv = x[i], yv = y[i];
    zfma(yv, a, xv);
    y[i] = yv;
  }
}
extern "C" int hx_zaxpy(int64_t n, const double alpha[2], const hx_c128* x, hx_c128* y, hx_stream s) {
  HX_REQUIRE(n >= 0 && x && y && alpha, "null/negative argument");
  if (n == 0) return HX_OK;
  k_zaxpy<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, hx_z(alpha), (const zc*)x, (zc*)y);
  HX_LAUNCH_CHECK("k_zaxpy");
  return HX_OK;
}

__global__ void k_zscal(int64_t n, zc a, zc* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = zmul(a, x[i]);
}
extern "C" int hx_zscal(int64_t n, const double alpha[2], hx_c128* x, hx_stream s) {
  HX_REQUIRE(n >= 0 && x && alpha, "null/negative argument");
  if (n == 0) return HX_OK;
  k_zscal<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, hx_z(alpha), (zc*)x);
  HX_LAUNCH_CHECK("k_zscal");
  return HX_OK;
}

struct LinComb4 { const zc* x[4]; zc c[4]; int nt; };
__global__ void k_zlincomb(int64_t n, LinComb4 p, zc* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < p.nt) zfma(acc, p.c[t], p.x[t][i]);
    out[i] = acc;
  }
}
extern "C" int hx_zlincomb(int64_t n, int nterms, const double* h_coefs, const hx_c128* const* h_xs,
                           hx_c128* out, hx_stream s) {
  HX_REQUIRE(n >= 0 && nterms >= 1 && nterms <= 4 && h_coefs && h_xs && out, "bad argument");
  if (n == 0) return HX_OK;
  LinComb4 p;
  p.nt = nterms;
  for (int t = 0; t < 4; ++t) {
    p.x[t] = t < nterms ? (const zc*)h_xs[t] : nullptr;
    p.c[t] = t < nterms ? make_double2(h_coefs[2 * t], h_coefs[2 * t + 1]) : make_double2(0, 0);
    if (t < nterms && !p.x[t]) return hx_fail(HX_ERR_ARG, __func__, "null term");
  }
  k_zlincomb<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, p, (zc*)out);
  HX_LAUNCH_CHECK("k_zlincomb");
  return HX_OK;
}

__global__ void k_zscatter_axpy(int64_t m, zc a, const zc* __restrict__ src, const int32_t* __restrict__ slot,
                                zc* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t k = slot[i];
    zc d = dst[k];
    zfma(d, a, src[i]);
    dst[k] = d;
  }
}
// ---------------------------------------------------------------- option debug_bounds (hx_common.h)
__global__ void k_first_bad_index(int64_t n, const int32_t* __restrict__ idx, int64_t lo, int64_t hi, unsigned long long* __restrict__ first) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = idx[i];
    if (v < lo || v >= hi) atomicMin(first, (unsigned long long)i);
  }
}

int hx_debug_check_indices(const char* who, const int32_t* idx, int64_t count, const int32_t* via_ptr, int nvec,
                           const void* vec, int elem_bytes, hipStream_t st) {
  if (via_ptr) {                                  // packed vectors: the number of indices is ptr[nvec]
    int32_t last = 0;
    HX_HIP(hipMemcpyAsync(&last, via_ptr + nvec, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    count = last;
  }
  if (count <= 0) return HX_OK;
  if (!idx) return hx_fail(HX_ERR_ARG, who, "debug_bounds: null index array");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)vec) != hipSuccess) {
    (void)hipGetLastError();
    return hx_fail(HX_ERR_ARG, who, "debug_bounds: the indexed vector is not a device allocation");
  }
  const int64_t lo = -(int64_t)(((const char*)vec - (const char*)base) / elem_bytes);
  const int64_t hi = (int64_t)(((const char*)base + size - (const char*)vec) / elem_bytes);
  static unsigned long long* d_first = nullptr;
  if (!d_first) HX_HIP(hipMalloc((void**)&d_first, sizeof(unsigned long long)));
  unsigned long long first = ~0ull;
  HX_HIP(hipMemcpyAsync(d_first, &first, sizeof(first), hipMemcpyHostToDevice, st));
  k_first_bad_index<<<hx_grid(count, 256), 256, 0, st>>>(count, idx, lo, hi, d_first);
  HX_LAUNCH_CHECK("k_first_bad_index");
  HX_HIP(hipMemcpyAsync(&first, d_first, sizeof(first), hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  if (first != ~0ull) {
    int32_t bad = 0;
    HX_HIP(hipMemcpy(&bad, idx + first, sizeof(int32_t), hipMemcpyDeviceToHost));
    char msg[256];
    snprintf(msg, sizeof(msg), "debug_bounds: index array entry %llu is %d, outside the allocation of the indexed vector "
             "(elements %lld .. %lld from its start)", first, (int)bad, (long long)lo, (long long)hi - 1);
    return hx_fail(HX_ERR_ARG, who, msg);
  }
  return HX_OK;
}

extern "C" int hx_zscatter_axpy(int64_t m, const double alpha[2], const hx_c128* src, const int32_t* slot,
                                hx_c128* dst, hx_stream s) {
  HX_REQUIRE(m >= 0 && alpha && src && slot && dst, "bad argument");
  if (m == 0) return HX_OK;
  HX_DEBUG_BOUNDS("hx_zscatter_axpy", slot, m, nullptr, 0, dst, 16, hx_s(s));
  k_zscatter_axpy<<<hx_grid(m, 256), 256, 0, hx_s(s)>>>(m, hx_z(alpha), (const zc*)src, slot, (zc*)dst);
  HX_LAUNCH_CHECK("k_zscatter_axpy");
  return HX_OK;
}

// slot[e] = position of entry e of the pattern (rp, ci) inside the pattern (brp, bci) (columns sorted within rows: binary
// search inside the base row); *missing = 1 if some entry is absent.  One thread per row of the smaller pattern.
__global__ void k_locate_slots(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                               const int32_t* __restrict__ brp, const int32_t* __restrict__ bci, int32_t* __restrict__ slot,
                               int32_t* __restrict__ missing) {
  for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    const int32_t e0 = rp[r], e1 = rp[r + 1];
    if (e1 == e0) continue;
    const int32_t b0 = brp[r], b1 = brp[r + 1];
    for (int32_t e = e0; e < e1; ++e) {
      const int32_t c = ci[e];
      int32_t lo = b0, hi = b1;
      while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (bci[mid] < c) lo = mid + 1; else hi = mid; }
      if (lo < b1 && bci[lo] == c) slot[e] = lo;
      else { slot[e] = 0; atomicOr(missing, 1); }
    }
  }
}
extern "C" int hx_csr_locate_slots(int32_t n, const int32_t* rp, const int32_t* ci, const int32_t* brp, const int32_t* bci,
                                   int32_t* slot, int32_t* missing, hx_stream s) {
  HX_REQUIRE(n >= 0 && rp && brp && slot && missing, "bad argument");
  HX_HIP(hipMemsetAsync(missing, 0, sizeof(int32_t), hx_s(s)));
  if (n == 0) return HX_OK;
  k_locate_slots<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, brp, bci, slot, missing);
  HX_LAUNCH_CHECK("k_locate_slots");
  return HX_OK;
}

__global__ void k_zgather(int64_t m, const zc* __restrict__ x, const int32_t* __restrict__ idx, zc* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[idx[i]];
}
extern "C" int hx_zgather(int64_t m, const hx_c128* x, const int32_t* idx, hx_c128* y, hx_stream s) {
  HX_REQUIRE(m >= 0 && x && idx && y, "bad argument");
  if (m == 0) return HX_OK;
  HX_DEBUG_BOUNDS("hx_zgather", idx, m, nullptr, 0, x, 16, hx_s(s));
  k_zgather<<<hx_grid(m, 256), 256, 0, hx_s(s)>>>(m, (const zc*)x, idx, (zc*)y);
  HX_LAUNCH_CHECK("k_zgather");
  return HX_OK;
}

// ---------------------------------------------------------------- CGS panel: out[j] = V_j^H w, out[k] = w^H w
// Stage 1: each block owns a contiguous slab of rows and produces k+1 partial sums; columns are
// processed in register tiles of JT so V is streamed exactly once (w is re-read per tile from L2/MALL).
// Stage 2: one block sums the per-block partials in a fixed order => bitwise reproducible.
#define PD_BLOCK 256
#define PD_JT 8
#define PD_MAXBLOCKS 1024

extern "C" size_t hx_panel_dot_ws(int k) { return (size_t)PD_MAXBLOCKS * (size_t)(k + 1); }

__device__ __forceinline__ zc pd_load(const zc* p) { return *p; }
__device__ __forceinline__ zc pd_load(const cc* p) { const cc v = *p; return make_double2((double)v.x, (double)v.y); }
// PT = zc: complex128 panel.  PT = cc: a complex64 COPY of the panel (half the traffic); w and the accumulation stay
// complex128.  The coefficients then carry a 6e-8 relative error, which classical Gram-Schmidt applied twice removes:
// the subtraction w -= V h always uses the complex128 panel, so the Arnoldi relation holds for whatever h was used.
// BILIN: unconjugated products out[j] = V_j^T w and out[k] = e^T w for a separate vector e (the bilinear form of complex
// symmetric Krylov methods, hx_cocg); otherwise e is w itself and all products are conjugated (Gram-Schmidt).
// WT: type of w (complex128, or complex64 for the V-cycle output that hx_cocg consumes without widening it first).
template <typename PT, int JT, bool BILIN, typename WT = zc>
__global__ void __launch_bounds__(PD_BLOCK)
k_panel_dot_stage1(int64_t n, int k, const PT* __restrict__ V, int64_t ldv, const WT* __restrict__ w,
                   const zc* __restrict__ e, zc* __restrict__ partials, int64_t rows_per_block) {
  __shared__ zc red[PD_BLOCK / HX_WAVE][JT];
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const int lane = threadIdx.x & (HX_WAVE - 1), wv = threadIdx.x / HX_WAVE;
  auto acc_fma = [](zc& a, zc v, zc x) { if (BILIN) zfma(a, v, x); else zcfma(a, v, x); };
  for (int j0 = 0; j0 <= k; j0 += JT) {           // column k is w itself (w^H w) or the extra vector (e^T w)
    zc acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) acc[t] = make_double2(0.0, 0.0);
    int64_t i = r0 + threadIdx.x;
    for (; i + PD_BLOCK < r1; i += 2 * PD_BLOCK) {     // two rows per trip: 2 x JT independent loads in flight
      const zc wa = pd_load(w + i), wb = pd_load(w + i + PD_BLOCK);
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = j0 + t;
        if (j < k) {
          const zc va = pd_load(V + (int64_t)j * ldv + i), vb = pd_load(V + (int64_t)j * ldv + i + PD_BLOCK);
          acc_fma(acc[t], va, wa);
          acc_fma(acc[t], vb, wb);
        } else if (j == k) {
          if (BILIN) { acc_fma(acc[t], e[i], wa); acc_fma(acc[t], e[i + PD_BLOCK], wb); }
          else { zcfma(acc[t], wa, wa); zcfma(acc[t], wb, wb); }
        }
      }
    }
    for (; i < r1; i += PD_BLOCK) {
      const zc wi = pd_load(w + i);
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = j0 + t;
        if (j < k) acc_fma(acc[t], pd_load(V + (int64_t)j * ldv + i), wi);
        else if (j == k) { if (BILIN) acc_fma(acc[t], e[i], wi); else zcfma(acc[t], wi, wi); }
      }
    }
#pragma unroll
    for (int t = 0; t < JT; ++t) {
      zc v = zreduce_sub<HX_WAVE>(acc[t]);
      if (lane == 0) red[wv][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < JT && j0 + (int)threadIdx.x <= k) {
      zc v = red[0][threadIdx.x];
      for (int q = 1; q < PD_BLOCK / HX_WAVE; ++q) v = zadd(v, red[q][threadIdx.x]);
      partials[(int64_t)blockIdx.x * (k + 1) + j0 + threadIdx.x] = v;
    }
    __syncthreads();
  }
}
// One wave per output column: lane l sums partials l, l+64, ... in a fixed order, then a shuffle tree.
__global__ void __launch_bounds__(HX_WAVE)
k_panel_dot_stage2(int nblocks, int k, const zc* __restrict__ partials, zc* __restrict__ out) {
  const int j = blockIdx.x;
  zc v = make_double2(0.0, 0.0);
  for (int b = threadIdx.x; b < nblocks; b += HX_WAVE) v = zadd(v, partials[(int64_t)b * (k + 1) + j]);
  v = zreduce_sub<HX_WAVE>(v);
  if (threadIdx.x == 0) out[j] = v;
}
// number of stage-1 blocks (= partial sums per column) for vectors of length n
int hx_panel_dot_blocks(int64_t n, int64_t* rows_per_block) {
  int64_t rpb = (n + PD_MAXBLOCKS - 1) / PD_MAXBLOCKS;
  if (rpb < PD_BLOCK * 4) rpb = PD_BLOCK * 4;
  if (rows_per_block) *rows_per_block = rpb;
  return (int)((n + rpb - 1) / rpb);
}
// fixed-order sum of per-block partials (k + 1 columns) -> out
int hx_panel_dot_finish(int nblocks, int k, const hx_c128* partials, hx_c128* out, hx_stream s) {
  k_panel_dot_stage2<<<k + 1, HX_WAVE, 0, hx_s(s)>>>(nblocks, k, (const zc*)partials, (zc*)out);
  HX_LAUNCH_CHECK("k_panel_dot_stage2");
  return HX_OK;
}
template <typename PT, int JT, bool BILIN, typename WT = zc>
static int panel_dot_launch(int64_t n, int k, const PT* V, int64_t ldv, const void* w, const hx_c128* e, hx_c128* out,
                            hx_c128* partials, hx_stream s, bool finish = true) {
  HX_REQUIRE(n > 0 && k >= 0 && w && out && partials && (k == 0 || V) && ldv >= n && (!BILIN || e), "bad argument");
  int64_t rpb;
  const int nblocks = hx_panel_dot_blocks(n, &rpb);
  k_panel_dot_stage1<PT, JT, BILIN, WT><<<nblocks, PD_BLOCK, 0, hx_s(s)>>>(n, k, V, ldv, (const WT*)w, (const zc*)e,
                                                                          (zc*)partials, rpb);
  HX_LAUNCH_CHECK("k_panel_dot_stage1");
  if (!finish) return HX_OK;          // the consumer kernel sums the per-block partials itself (hx_cocg: one launch fewer)
  return hx_panel_dot_finish(nblocks, k, partials, out, s);
}
extern "C" int hx_zpanel_dot(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* w, hx_c128* out,
                             hx_c128* partials, hx_stream s) {
  return panel_dot_launch<zc, PD_JT, false>(n, k, (const zc*)V, ldv, w, nullptr, out, partials, s);
}
extern "C" int hx_zpanel_dot_c64(int64_t n, int k, const hx_c64* Vf, int64_t ldv, const hx_c128* w, hx_c128* out,
                                 hx_c128* partials, hx_stream s) {
  return panel_dot_launch<cc, PD_JT, false>(   // (a 16-wide register tile was measured: 371 vs 223 us per launch)
      n, k, (const cc*)Vf, ldv, w, nullptr, out, partials, s);
}
// Bilinear variants (internal; hx_cocg): out[j] = V_j^T w for j < k, out[k] = e^T w.
int hx_zpanel_bdot(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* w, const hx_c128* e, hx_c128* out,
                   hx_c128* partials, hx_stream s) {
  return panel_dot_launch<zc, PD_JT, true>(n, k, (const zc*)V, ldv, w, e, out, partials, s);
}
// stage 1 only: partials[b * (k + 1) + j], b < hx_panel_dot_blocks(n), summed by the consumer in the order of stage 2
int hx_zpanel_bdot_partials(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* w, const hx_c128* e,
                            hx_c128* partials, hx_stream s) {
  return panel_dot_launch<zc, PD_JT, true>(n, k, (const zc*)V, ldv, w, e, partials, partials, s, false);
}
int hx_zpanel_bdot_w32_partials(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c64* w, const hx_c128* e,
                                hx_c128* partials, hx_stream s) {
  return panel_dot_launch<zc, PD_JT, true, cc>(n, k, (const zc*)V, ldv, w, e, partials, partials, s, false);
}
// the same with w given in complex64 (panel complex128)
int hx_zpanel_bdot_w32(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c64* w, const hx_c128* e, hx_c128* out,
                       hx_c128* partials, hx_stream s) {
  return panel_dot_launch<zc, PD_JT, true, cc>(n, k, (const zc*)V, ldv, w, e, out, partials, s);
}
int hx_zpanel_bdot_c64(int64_t n, int k, const hx_c64* Vf, int64_t ldv, const hx_c128* w, const hx_c128* e, hx_c128* out,
                       hx_c128* partials, hx_stream s) {
  return panel_dot_launch<cc, PD_JT, true>(n, k, (const cc*)Vf, ldv, w, e, out, partials, s);
}

// ---------------------------------------------------------------- w -= V h
#define PS_MAXK 192
template <typename PT>
__global__ void __launch_bounds__(256)
k_panel_sub(int64_t n, int k, const PT* __restrict__ V, int64_t ldv, const zc* __restrict__ h, zc* __restrict__ w) {
  __shared__ zc hs[PS_MAXK];
  for (int j = threadIdx.x; j < k; j += blockDim.x) hs[j] = h[j];
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc acc = w[i];
    for (int j = 0; j < k; ++j) {
      zc v = pd_load(V + (int64_t)j * ldv + i), hj = hs[j];
      acc.x = fma(-v.x, hj.x, acc.x); acc.x = fma(v.y, hj.y, acc.x);
      acc.y = fma(-v.x, hj.y, acc.y); acc.y = fma(-v.y, hj.x, acc.y);
    }
    w[i] = acc;
  }
}
extern "C" int hx_zpanel_sub(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* h, hx_c128* w,
                             hx_stream s) {
  HX_REQUIRE(n > 0 && k >= 0 && k <= PS_MAXK && w && (k == 0 || (V && h)) && ldv >= n, "bad argument");
  if (k == 0) return HX_OK;
  k_panel_sub<zc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, (const zc*)V, ldv, (const zc*)h, (zc*)w);
  HX_LAUNCH_CHECK("k_panel_sub");
  return HX_OK;
}
extern "C" int hx_zpanel_sub_c64(int64_t n, int k, const hx_c64* Vf, int64_t ldv, const hx_c128* h, hx_c128* w,
                                 hx_stream s) {
  HX_REQUIRE(n > 0 && k >= 0 && k <= PS_MAXK && w && (k == 0 || (Vf && h)) && ldv >= n, "bad argument");
  if (k == 0) return HX_OK;
  k_panel_sub<cc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, (const cc*)Vf, ldv, (const zc*)h, (zc*)w);
  HX_LAUNCH_CHECK("k_panel_sub");
  return HX_OK;
}

// ---------------------------------------------------------------- Vout = V Q (restart update)
#define PM_CT 4
__global__ void __launch_bounds__(256)
k_panel_mul(int64_t n, int m, int kout, const zc* __restrict__ V, int64_t ldv, const zc* __restrict__ Q,
            zc* __restrict__ Vout, int64_t ldvout) {
  extern __shared__ zc qs[];  // m x kout, column-major
  for (int t = threadIdx.x; t < m * kout; t += blockDim.x) qs[t] = Q[t];
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    for (int c0 = 0; c0 < kout; c0 += PM_CT) {
      zc acc[PM_CT];
#pragma unroll
      for (int t = 0; t < PM_CT; ++t) acc[t] = make_double2(0.0, 0.0);
      for (int j = 0; j < m; ++j) {
        const zc v = V[(int64_t)j * ldv + i];
#pragma unroll
        for (int t = 0; t < PM_CT; ++t)
          if (c0 + t < kout) zfma(acc[t], v, qs[(c0 + t) * m + j]);
      }
#pragma unroll
      for (int t = 0; t < PM_CT; ++t)
        if (c0 + t < kout) Vout[(int64_t)(c0 + t) * ldvout + i] = acc[t];
    }
  }
}
extern "C" int hx_zpanel_mul(int64_t n, int m, int kout, const hx_c128* V, int64_t ldv, const hx_c128* Q,
                             hx_c128* Vout, int64_t ldvout, hx_stream s) {
  HX_REQUIRE(n > 0 && m > 0 && kout > 0 && V && Q && Vout && ldv >= n && ldvout >= n, "bad argument");
  size_t lds = (size_t)m * kout * sizeof(zc);
  HX_REQUIRE(lds <= 64 * 1024, "Q too large for LDS staging");
  k_panel_mul<<<hx_grid(n, 256), 256, lds, hx_s(s)>>>(n, m, kout, (const zc*)V, ldv, (const zc*)Q, (zc*)Vout, ldvout);
  HX_LAUNCH_CHECK("k_panel_mul");
  return HX_OK;
}

// ---------------------------------------------------------------- sparse dots (b_f^T y), one wave per vector
__global__ void k_sparse_dots(int nvec, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                              const zc* __restrict__ vals, const zc* __restrict__ y, zc* __restrict__ out) {
  int f = blockIdx.x;
  if (f >= nvec) return;
  zc acc = make_double2(0.0, 0.0);
  for (int p = ptr[f] + threadIdx.x; p < ptr[f + 1]; p += HX_WAVE) zfma(acc, vals[p], y[idx[p]]);
  acc = zreduce_sub<HX_WAVE>(acc);
  if (threadIdx.x == 0) out[f] = acc;
}
extern "C" int hx_zsparse_dots(int nvec, const int32_t* ptr, const int32_t* idx, const hx_c128* vals,
                               const hx_c128* y, hx_c128* out, hx_stream s) {
  HX_REQUIRE(nvec >= 0 && ptr && y && out, "bad argument");
  if (nvec == 0) return HX_OK;
  HX_DEBUG_BOUNDS("hx_zsparse_dots", idx, 0, ptr, nvec, y, 16, hx_s(s));
  k_sparse_dots<<<nvec, HX_WAVE, 0, hx_s(s)>>>(nvec, ptr, idx, (const zc*)vals, (const zc*)y, (zc*)out);
  HX_LAUNCH_CHECK("k_sparse_dots");
  return HX_OK;
}

// ---------------------------------------------------------------- sparse updates y += coef * sum_f t[f] u_f (left factors of the flame term)
// One block per vector f of the range [f0, f1): y[idx[p]] += coef * t[f] * vals[p] for its packed entries.  The coefficients
// t stay on the device (they come out of hx_zsparse_dots / an all-reduce): no host round trip per product.  Indices are unique
// inside a vector; vectors with DISJOINT supports (flames in separate regions) go in one launch, overlapping ones are issued
// one vector per launch by the caller (stream order = a fixed summation order: deterministic, no atomics).
__global__ void k_sparse_axpys(int f0, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, const zc* __restrict__ vals,
                               const zc* __restrict__ t, zc coef, zc* __restrict__ y) {
  const int f = f0 + blockIdx.x;
  const zc a = zmul(coef, t[f]);
  for (int p = ptr[f] + threadIdx.x; p < ptr[f + 1]; p += blockDim.x) {
    const int32_t i = idx[p];
    zc v = y[i];
    zfma(v, a, vals[p]);
    y[i] = v;
  }
}
extern "C" int hx_zsparse_axpys(int nvec, const int32_t* ptr, const int32_t* idx, const hx_c128* vals, const hx_c128* t,
                                const double coef[2], hx_c128* y, int disjoint, hx_stream s) {
  HX_REQUIRE(nvec >= 0 && ptr && t && coef && y, "bad argument");
  if (nvec == 0) return HX_OK;
  HX_REQUIRE(idx && vals, "bad argument");
  HX_DEBUG_BOUNDS("hx_zsparse_axpys", idx, 0, ptr, nvec, y, 16, hx_s(s));
  if (disjoint) k_sparse_axpys<<<nvec, 256, 0, hx_s(s)>>>(0, ptr, idx, (const zc*)vals, (const zc*)t, hx_z(coef), (zc*)y);
  else for (int f = 0; f < nvec; ++f) k_sparse_axpys<<<1, 256, 0, hx_s(s)>>>(f, ptr, idx, (const zc*)vals, (const zc*)t, hx_z(coef), (zc*)y);
  HX_LAUNCH_CHECK("k_sparse_axpys");
  return HX_OK;
}

// ---------------------------------------------------------------- threshold + compaction of a load vector
// flame_matrices.py:66-72: zero |Re| < tol and |Im| < tol separately, keep the non-zeros as (index, value) pairs.
// The supports are tiny compared with n, so only the packed pairs travel to the host (order fixed there by sorting
// the indices; the atomic counter only hands out slots).
__global__ void k_threshold_pack(int64_t n, const zc* __restrict__ x, double tol, int32_t cap, int32_t* __restrict__ count,
                                 int32_t* __restrict__ idx, zc* __restrict__ val) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc v = x[i];
    if (fabs(v.x) < tol) v.x = 0.0;
    if (fabs(v.y) < tol) v.y = 0.0;
    if (v.x != 0.0 || v.y != 0.0) {
      const int32_t pos = atomicAdd(count, 1);
      if (pos < cap) { idx[pos] = (int32_t)i; val[pos] = v; }
    }
  }
}
extern "C" int hx_zthreshold_pack(int64_t n, const hx_c128* x, double tol, int32_t cap, int32_t* count, int32_t* idx,
                                  hx_c128* val, hx_stream s) {
  HX_REQUIRE(n >= 0 && n < (int64_t)INT32_MAX && x && cap >= 0 && count && idx && val, "bad argument");
  HX_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), hx_s(s)));
  if (n == 0) return HX_OK;
  k_threshold_pack<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, (const zc*)x, tol, cap, count, idx, (zc*)val);
  HX_LAUNCH_CHECK("k_threshold_pack");
  return HX_OK;
}

// ---------------------------------------------------------------- small dense helpers of the solve path (were torch calls)
// y = A x for a small row-major m x n matrix on the device (the r x r Woodbury coefficient matrix of the flame term,
// r = number of flames <= 64): one block, one thread per row.
__global__ void __launch_bounds__(64)
k_small_gemv(int m, int n, const zc* __restrict__ A, const zc* __restrict__ x, zc* __restrict__ y) {
  __shared__ zc xs[64];
  if ((int)threadIdx.x < n) xs[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  const int i = threadIdx.x;
  if (i < m) {
    zc acc = make_double2(0.0, 0.0);
    for (int j = 0; j < n; ++j) zfma(acc, A[i * n + j], xs[j]);
    y[i] = acc;
  }
}
extern "C" int hx_zsmall_gemv(int m, int n, const hx_c128* A, const hx_c128* x, hx_c128* y, hx_stream s) {
  HX_REQUIRE(m >= 1 && m <= 64 && n >= 1 && n <= 64 && A && x && y && x != y, "bad argument (m, n <= 64)");
  k_small_gemv<<<1, 64, 0, hx_s(s)>>>(m, n, (const zc*)A, (const zc*)x, (zc*)y);
  HX_LAUNCH_CHECK("k_small_gemv");
  return HX_OK;
}

// dst[(i + shift) mod n] = src[i]: cyclic shift (the translated Woodbury column of identical flames on a periodic numbering)
__global__ void k_zroll(int64_t n, int64_t shift, const zc* __restrict__ src, zc* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = i + shift;
    if (j >= n) j -= n;
    dst[j] = src[i];
  }
}
extern "C" int hx_zroll(int64_t n, int64_t shift, const hx_c128* src, hx_c128* dst, hx_stream s) {
  HX_REQUIRE(n > 0 && src && dst && src != dst, "bad argument");
  shift %= n;
  if (shift < 0) shift += n;
  k_zroll<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, shift, (const zc*)src, (zc*)dst);
  HX_LAUNCH_CHECK("k_zroll");
  return HX_OK;
}

// ---- in-place inverse of a dense n x n complex128 matrix (row-major) by Gauss-Jordan elimination with partial pivoting: the
// coarsest operator of the V-cycle (n <= ~1000; applied as a dense product afterwards).  Per column: one single-block kernel finds
// the pivot, swaps the two rows, scales the pivot row; one grid-wide kernel eliminates the column from every other row.  The
// column permutation the row swaps imply is undone at the end, one thread per row.  2 n + 1 launches: ~10 ms at n = 1000 --
// once per hierarchy, where a LAPACK call on the host (round 1) or rocSOLVER through torch (round 3) stood.
__global__ void __launch_bounds__(256)
k_gj_pivot(int n, int k, zc* __restrict__ A, int32_t* __restrict__ piv, int* __restrict__ singular) {
  __shared__ double best[256];
  __shared__ int where[256];
  double b = -1.0;
  int w = k;
  for (int i = k + threadIdx.x; i < n; i += 256) {
    const zc v = A[(int64_t)i * n + k];
    const double a = v.x * v.x + v.y * v.y;
    if (a > b) { b = a; w = i; }
  }
  best[threadIdx.x] = b; where[threadIdx.x] = w;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double c = best[threadIdx.x + o];
      const int cw = where[threadIdx.x + o];
      if (c > best[threadIdx.x] || (c == best[threadIdx.x] && cw < where[threadIdx.x])) { best[threadIdx.x] = c; where[threadIdx.x] = cw; }
    }
    __syncthreads();
  }
  const int p = where[0];
  if (threadIdx.x == 0) { piv[k] = p; if (!(best[0] > 0.0)) *singular = 1; }
  if (!(best[0] > 0.0)) return;
  __shared__ zc dinv;
  if (threadIdx.x == 0) dinv = zdiv(make_double2(1.0, 0.0), A[(int64_t)p * n + k]);
  __syncthreads();
  const zc d = dinv;
  // swap rows k and p, then scale row k by 1/pivot with the pivot entry itself replaced by 1 (in-place Gauss-Jordan)
  for (int j = threadIdx.x; j < n; j += 256) {
    zc rk = A[(int64_t)p * n + j];
    if (p != k) A[(int64_t)p * n + j] = A[(int64_t)k * n + j];
    if (j == k) rk = make_double2(1.0, 0.0);
    A[(int64_t)k * n + j] = zmul(rk, d);
  }
}
__global__ void __launch_bounds__(256)
k_gj_eliminate(int n, int k, zc* __restrict__ A, const int* __restrict__ singular) {
  if (*singular) return;
  const int i = blockIdx.x;                    // one block per row
  if (i == k) return;
  __shared__ zc f_s;
  if (threadIdx.x == 0) { f_s = A[(int64_t)i * n + k]; }
  __syncthreads();
  const zc f = f_s;
  if (f.x == 0.0 && f.y == 0.0) return;
  for (int j = threadIdx.x; j < n; j += 256) {
    zc a = j == k ? make_double2(0.0, 0.0) : A[(int64_t)i * n + j];
    const zc r = A[(int64_t)k * n + j];
    a.x = fma(-f.x, r.x, a.x); a.x = fma(f.y, r.y, a.x);
    a.y = fma(-f.x, r.y, a.y); a.y = fma(-f.y, r.x, a.y);
    A[(int64_t)i * n + j] = a;
  }
}
__global__ void k_gj_unpermute(int n, zc* __restrict__ A, const int32_t* __restrict__ piv, const int* __restrict__ singular) {
  if (*singular) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = n - 1; k >= 0; --k) {
    const int p = piv[k];
    if (p != k) { const zc t = A[(int64_t)i * n + k]; A[(int64_t)i * n + k] = A[(int64_t)i * n + p]; A[(int64_t)i * n + p] = t; }
  }
}
// SYNC (reads the singularity flag).  A: n x n row-major, overwritten by its inverse.
extern "C" int hx_zdense_inverse(int n, hx_c128* A, hx_stream s) {
  HX_REQUIRE(n >= 1 && n <= 4096 && A, "bad argument (n <= 4096)");
  int32_t* piv = nullptr;
  HX_HIP(hx_malloc(&piv, sizeof(int32_t) * (size_t)(n + 1)));
  int* flag = (int*)(piv + n);
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), hx_s(s));
  for (int k = 0; k < n && e == hipSuccess; ++k) {
    k_gj_pivot<<<1, 256, 0, hx_s(s)>>>(n, k, (zc*)A, piv, flag);
    k_gj_eliminate<<<n, 256, 0, hx_s(s)>>>(n, k, (zc*)A, flag);
    if ((k & 63) == 63) e = hipGetLastError();
  }
  if (e == hipSuccess) {
    k_gj_unpermute<<<(n + 255) / 256, 256, 0, hx_s(s)>>>(n, (zc*)A, piv, flag);
    e = hipGetLastError();
  }
  int h_flag = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, hx_s(s));
  if (e == hipSuccess) e = hipStreamSynchronize(hx_s(s));
  (void)hx_free(piv);
  if (e != hipSuccess) return hx_fail(HX_ERR_HIP, "hx_zdense_inverse", hipGetErrorString(e));
  if (h_flag) return hx_fail(HX_ERR_BREAKDOWN, "hx_zdense_inverse", "singular matrix");
  return HX_OK;
}
