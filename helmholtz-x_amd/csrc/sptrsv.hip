// Level-scheduled sparse triangular solve, complex128 (PETSc MatSolve on the shift-invert path).
//
// The factors come from a host factorisation; analysis (dependency levels) runs on the host once per
// factorisation, the solves run on the device.  Rows are stored physically in (level, row) order so
// each level streams one contiguous span of col/val.  Wide levels get one launch each (LPR lanes per
// row chosen from the level's mean row length); runs of consecutive narrow levels (the long dependency
// chains of the trailing separators) are fused into ONE single-workgroup launch that walks the levels
// with workgroup barriers, which removes thousands of ~1.5 us kernel boundaries.
// Triangular solves are dependency/latency-bound, not HBM-bound.
#include <algorithm>
#include <vector>

#include "hx_common.h"

#define CHAIN_MAX_ROWS 48      // a level with <= this many rows may be fused into a chain launch
#define CHAIN_THREADS 1024
#define CHAIN_WAVES (CHAIN_THREADS / HX_WAVE)

struct Launch { int kind; int32_t l0, l1; int lpr; };  // kind 0: wide level l0; kind 1: chain [l0, l1)

struct hx_sptrsv {
  int32_t n = 0;
  int64_t nnz = 0;
  int32_t n_levels = 0;
  std::vector<Launch> plan;
  std::vector<int32_t> h_level_ptr;
  int32_t* d_level_ptr = nullptr;  // n_levels+1, offsets into the permuted row order
  int32_t* d_rows = nullptr;       // permuted position -> original row
  int32_t* d_rowptr = nullptr;     // n+1 over permuted positions
  int32_t* d_col = nullptr;
  zc* d_val = nullptr;
  zc* d_inv_diag = nullptr;        // indexed by original row
};

template <int LPR>
__global__ void __launch_bounds__(256)
k_level(int32_t p0, int32_t p1, const int32_t* __restrict__ rows, const int32_t* __restrict__ rowptr,
        const int32_t* __restrict__ col, const zc* __restrict__ val, const zc* __restrict__ inv_diag,
        const zc* __restrict__ b, zc* __restrict__ x) {
  const int sub = threadIdx.x % LPR;
  const int32_t p = p0 + (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / LPR);
  zc acc = make_double2(0.0, 0.0);
  const bool live = p < p1;
  if (live)
    for (int32_t t = rowptr[p] + sub; t < rowptr[p + 1]; t += LPR) zfma(acc, val[t], x[col[t]]);
  if (LPR > 1) acc = zreduce_sub<LPR>(acc);
  if (live && sub == 0) {
    const int32_t r = rows[p];
    x[r] = zmul(zsub(b[r], acc), inv_diag[r]);
  }
}

__global__ void __launch_bounds__(CHAIN_THREADS)
k_chain(int32_t l0, int32_t l1, const int32_t* __restrict__ level_ptr, const int32_t* __restrict__ rows,
        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const zc* __restrict__ val,
        const zc* __restrict__ inv_diag, const zc* b, zc* x) {
  const int lane = threadIdx.x & (HX_WAVE - 1), wv = threadIdx.x / HX_WAVE;
  for (int32_t l = l0; l < l1; ++l) {
    const int32_t pe = level_ptr[l + 1];
    for (int32_t p = level_ptr[l] + wv; p < pe; p += CHAIN_WAVES) {
      zc acc = make_double2(0.0, 0.0);
      for (int32_t t = rowptr[p] + lane; t < rowptr[p + 1]; t += HX_WAVE) {
        // x entries of earlier levels were written by other waves of THIS workgroup: read them with a
        // device-coherent load so no stale L1 line can be observed.
        const zc xv = make_double2(__hip_atomic_load(&x[col[t]].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   __hip_atomic_load(&x[col[t]].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        zfma(acc, val[t], xv);
      }
      acc = zreduce_sub<HX_WAVE>(acc);
      if (lane == 0) {
        const int32_t r = rows[p];
        const zc v = zmul(zsub(b[r], acc), inv_diag[r]);
        __hip_atomic_store(&x[r].x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&x[r].y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

template <typename T>
static int upload(T** dst, const T* src, size_t count) {
  HX_HIP(hx_malloc((void**)dst, std::max<size_t>(count, 1) * sizeof(T)));
  if (count) HX_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return HX_OK;
}

extern "C" int hx_sptrsv_destroy(hx_sptrsv* t) {
  if (!t) return HX_OK;
  (void)hx_free(t->d_level_ptr); (void)hx_free(t->d_rows); (void)hx_free(t->d_rowptr);
  (void)hx_free(t->d_col); (void)hx_free(t->d_val); (void)hx_free(t->d_inv_diag);
  delete t;
  return HX_OK;
}

extern "C" int hx_sptrsv_create(hx_sptrsv** out, int32_t n, int lower, const int32_t* h_rowptr,
                                const int32_t* h_col, const hx_c128* h_val, const hx_c128* h_inv_diag) {
  HX_REQUIRE(out && n > 0 && h_rowptr && h_inv_diag, "bad argument");
  const int64_t nnz = h_rowptr[n];
  HX_REQUIRE(nnz == 0 || (h_col && h_val), "null factor arrays");
  // ---- dependency levels
  std::vector<int32_t> level(n, 0);
  int32_t nlev = 0;
  if (lower) {
    for (int32_t i = 0; i < n; ++i) {
      int32_t lv = 0;
      for (int32_t t = h_rowptr[i]; t < h_rowptr[i + 1]; ++t) {
        const int32_t j = h_col[t];
        if (j >= i || j < 0) return hx_fail(HX_ERR_ARG, __func__, "L must be strictly lower triangular");
        lv = std::max(lv, level[j] + 1);
      }
      level[i] = lv;
      nlev = std::max(nlev, lv + 1);
    }
  } else {
    for (int32_t i = n - 1; i >= 0; --i) {
      int32_t lv = 0;
      for (int32_t t = h_rowptr[i]; t < h_rowptr[i + 1]; ++t) {
        const int32_t j = h_col[t];
        if (j <= i || j >= n) return hx_fail(HX_ERR_ARG, __func__, "U must be strictly upper triangular");
        lv = std::max(lv, level[j] + 1);
      }
      level[i] = lv;
      nlev = std::max(nlev, lv + 1);
    }
  }
  // ---- bucket rows by level (stable in row index) and permute the CSR into that order
  hx_sptrsv* t = new hx_sptrsv();
  t->n = n; t->nnz = nnz; t->n_levels = nlev;
  t->h_level_ptr.assign(nlev + 1, 0);
  for (int32_t i = 0; i < n; ++i) t->h_level_ptr[level[i] + 1]++;
  for (int32_t l = 0; l < nlev; ++l) t->h_level_ptr[l + 1] += t->h_level_ptr[l];
  std::vector<int32_t> rows(n), cursor(t->h_level_ptr.begin(), t->h_level_ptr.end() - 1);
  for (int32_t i = 0; i < n; ++i) rows[cursor[level[i]]++] = i;
  std::vector<int32_t> prowptr(n + 1, 0), pcol((size_t)nnz);
  std::vector<hx_c128> pval((size_t)nnz);
  for (int32_t p = 0; p < n; ++p) {
    const int32_t r = rows[p];
    const int32_t len = h_rowptr[r + 1] - h_rowptr[r];
    prowptr[p + 1] = prowptr[p] + len;
    std::copy(h_col + h_rowptr[r], h_col + h_rowptr[r + 1], pcol.begin() + prowptr[p]);
    std::copy(h_val + h_rowptr[r], h_val + h_rowptr[r + 1], pval.begin() + prowptr[p]);
  }
  // ---- launch plan
  int32_t l = 0;
  while (l < nlev) {
    const int32_t sz = t->h_level_ptr[l + 1] - t->h_level_ptr[l];
    if (sz > CHAIN_MAX_ROWS) {
      const int64_t lnnz = prowptr[t->h_level_ptr[l + 1]] - prowptr[t->h_level_ptr[l]];
      const double avg = (double)lnnz / (double)sz;
      const int lpr = avg <= 6 ? 4 : avg <= 48 ? 16 : 64;
      t->plan.push_back({0, l, l + 1, lpr});
      ++l;
    } else {
      int32_t e = l;
      while (e < nlev && t->h_level_ptr[e + 1] - t->h_level_ptr[e] <= CHAIN_MAX_ROWS) ++e;
      t->plan.push_back({1, l, e, HX_WAVE});
      l = e;
    }
  }
  int rc = upload(&t->d_level_ptr, t->h_level_ptr.data(), t->h_level_ptr.size());
  if (!rc) rc = upload(&t->d_rows, rows.data(), rows.size());
  if (!rc) rc = upload(&t->d_rowptr, prowptr.data(), prowptr.size());
  if (!rc) rc = upload(&t->d_col, pcol.data(), pcol.size());
  if (!rc) rc = upload(&t->d_val, (const zc*)pval.data(), pval.size());
  if (!rc) rc = upload(&t->d_inv_diag, (const zc*)h_inv_diag, (size_t)n);
  if (rc) { hx_sptrsv_destroy(t); return rc; }
  *out = t;
  return HX_OK;
}

extern "C" int hx_sptrsv_info(const hx_sptrsv* t, int32_t* n_levels, int32_t* n_launches, int64_t* nnz) {
  HX_REQUIRE(t, "null handle");
  if (n_levels) *n_levels = t->n_levels;
  if (n_launches) *n_launches = (int32_t)t->plan.size();
  if (nnz) *nnz = t->nnz;
  return HX_OK;
}

extern "C" int hx_sptrsv_solve(hx_sptrsv* t, const hx_c128* b_, hx_c128* x_, hx_stream s_) {
  HX_REQUIRE(t && b_ && x_, "bad argument");
  hipStream_t s = hx_s(s_);
  const zc* b = (const zc*)b_;
  zc* x = (zc*)x_;
  for (const Launch& L : t->plan) {
    if (L.kind == 1) {
      k_chain<<<1, CHAIN_THREADS, 0, s>>>(L.l0, L.l1, t->d_level_ptr, t->d_rows, t->d_rowptr, t->d_col, t->d_val,
                                          t->d_inv_diag, b, x);
    } else {
      const int32_t p0 = t->h_level_ptr[L.l0], p1 = t->h_level_ptr[L.l0 + 1];
      const int64_t threads = (int64_t)(p1 - p0) * L.lpr;
      const int grid = (int)((threads + 255) / 256);
      if (L.lpr == 4) k_level<4><<<grid, 256, 0, s>>>(p0, p1, t->d_rows, t->d_rowptr, t->d_col, t->d_val, t->d_inv_diag, b, x);
      else if (L.lpr == 16) k_level<16><<<grid, 256, 0, s>>>(p0, p1, t->d_rows, t->d_rowptr, t->d_col, t->d_val, t->d_inv_diag, b, x);
      else k_level<64><<<grid, 256, 0, s>>>(p0, p1, t->d_rows, t->d_rowptr, t->d_col, t->d_val, t->d_inv_diag, b, x);
    }
  }
  HX_LAUNCH_CHECK("hx_sptrsv_solve");
  return HX_OK;
}
