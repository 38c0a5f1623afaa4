// P1 (Lagrange degree 1) operator and load-vector assembly on simplices, vertex-centric.
//
// One thread owns one CSR row (= one mesh vertex) and gathers the element contributions of the cells
// incident to that vertex in ascending cell order.  Consequences: no atomics, bitwise reproducible
// values, integer maps (adjacency, rowptr, col) that are a pure function of the mesh, and an
// "owner computes" formulation that needs no inter-GPU exchange when rows are sharded (each GPU keeps
// one ghost layer of cells).  Element integrals are the closed forms of the reference's UFL forms
// (acoustic_matrices.py:101,108,121; flame_matrices.py:141,199,200); the small per-simplex tables
// (mass / triple-product templates, degree-2 quadrature points) are compile-time constants.
#include <hipcub/hipcub.hpp>

#include "hx_common.h"

// ------------------------------------------------------------------------------------------------ adjacency
__global__ void k_count_incidence(int64_t total, const int32_t* __restrict__ ents, int32_t* __restrict__ cnt) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&cnt[ents[t]], 1);
}
__global__ void k_fill_incidence(int64_t total, int k, const int32_t* __restrict__ ents,
                                 const int32_t* __restrict__ ptr, int32_t* __restrict__ cursor,
                                 int32_t* __restrict__ idx) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int32_t v = ents[t];
    int32_t pos = atomicAdd(&cursor[v], 1);
    idx[ptr[v] + pos] = (int32_t)(t / k);
  }
}
__global__ void k_sort_lists(int32_t n, const int32_t* __restrict__ ptr, int32_t* __restrict__ idx) {
  for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    int32_t s = ptr[v], e = ptr[v + 1];
    for (int32_t a = s + 1; a < e; ++a) {          // insertion sort: lists are short (vertex degree)
      int32_t key = idx[a], b = a - 1;
      while (b >= s && idx[b] > key) { idx[b + 1] = idx[b]; --b; }
      idx[b + 1] = key;
    }
  }
}

static int exclusive_scan_i32(int32_t* d_inout, int64_t count, hipStream_t s) {
  size_t bytes = 0;
  HX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_inout, d_inout, (int)count, s));
  void* tmp = nullptr;
  HX_HIP(hx_malloc(&tmp, bytes ? bytes : 16));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, d_inout, d_inout, (int)count, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hx_free(tmp);
  if (e != hipSuccess) return hx_fail(HX_ERR_HIP, "ExclusiveSum", hipGetErrorString(e));
  if (e2 != hipSuccess) return hx_fail(HX_ERR_HIP, "hipStreamSynchronize", hipGetErrorString(e2));
  return HX_OK;
}

extern "C" int hx_build_v2e(int32_t n_vert, int32_t n_ent, int k, const int32_t* ents, int32_t* v2e_ptr,
                            int32_t* v2e_idx, hx_stream s_) {
  HX_REQUIRE(n_vert > 0 && n_ent >= 0 && k >= 1 && k <= 16 && v2e_ptr && (n_ent == 0 || (ents && v2e_idx)),
             "bad argument");
  hipStream_t s = hx_s(s_);
  const int64_t total = (int64_t)n_ent * k;
  HX_REQUIRE(total < (int64_t)INT32_MAX, "adjacency exceeds int32 range");
  HX_HIP(hipMemsetAsync(v2e_ptr, 0, sizeof(int32_t) * ((size_t)n_vert + 1), s));
  if (total > 0) {
    k_count_incidence<<<hx_grid(total, 256), 256, 0, s>>>(total, ents, v2e_ptr);
    HX_LAUNCH_CHECK("k_count_incidence");
  }
  int rc = exclusive_scan_i32(v2e_ptr, (int64_t)n_vert + 1, s);
  if (rc) return rc;
  if (total == 0) return HX_OK;
  int32_t* cursor = nullptr;
  HX_HIP(hx_malloc((void**)&cursor, sizeof(int32_t) * (size_t)n_vert));
  (void)hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n_vert, s);
  k_fill_incidence<<<hx_grid(total, 256), 256, 0, s>>>(total, k, ents, v2e_ptr, cursor, v2e_idx);
  k_sort_lists<<<hx_grid(n_vert, 256), 256, 0, s>>>(n_vert, v2e_ptr, v2e_idx);
  hipError_t e = hipStreamSynchronize(s);
  (void)hx_free(cursor);
  if (e != hipSuccess) return hx_fail(HX_ERR_HIP, "hx_build_v2e", hipGetErrorString(e));
  HX_LAUNCH_CHECK("hx_build_v2e");
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ pattern
// Row v = sorted unique union of the vertices of the entities incident to v, written to the row's scratch region
// [k*ptr[v], k*ptr[v+1]).  A group of 16 lanes owns a row: the candidates (k per incident entity) go to the group's LDS
// region; a candidate is KEPT if no earlier candidate has the same value, and a kept candidate's position in the sorted
// list is the number of kept candidates below it -- two all-pairs passes over LDS (<= 128^2 / 16 comparisons per lane),
// no data-dependent shifting in memory.  Rows with more than PR_MAX candidates take the serial insertion walk.
#define PR_GL 16
#define PR_MAX 128
__device__ __forceinline__ void pattern_row_serial(int32_t v, int k, const int32_t* __restrict__ ents, const int32_t* __restrict__ ptr,
                                                   const int32_t* __restrict__ idx, int32_t* __restrict__ scratch,
                                                   int32_t* __restrict__ rowlen) {
  int32_t* list = scratch + (int64_t)k * ptr[v];
  int32_t len = 0;
  for (int32_t p = ptr[v]; p < ptr[v + 1]; ++p) {
    const int32_t* ev = ents + (int64_t)idx[p] * k;
    for (int j = 0; j < k; ++j) {
      int32_t c = ev[j];
      int32_t lo = 0, hi = len;                   // lower bound
      while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (list[mid] < c) lo = mid + 1; else hi = mid; }
      if (lo < len && list[lo] == c) continue;
      for (int32_t t = len; t > lo; --t) list[t] = list[t - 1];
      list[lo] = c;
      ++len;
    }
  }
  rowlen[v] = len;
}
__global__ void __launch_bounds__(256)
k_pattern_rows(int32_t n, int k, const int32_t* __restrict__ ents, const int32_t* __restrict__ ptr,
               const int32_t* __restrict__ idx, int32_t* __restrict__ scratch, int32_t* __restrict__ rowlen) {
  constexpr int GPB = 256 / PR_GL;
  __shared__ int32_t s_cand[GPB][PR_MAX];
  __shared__ uint8_t s_keep[GPB][PR_MAX];
  const int g = threadIdx.x / PR_GL, t = threadIdx.x % PR_GL;
  const int64_t n_groups = (int64_t)gridDim.x * GPB;
  for (int64_t v64 = (int64_t)blockIdx.x * GPB + g; v64 < n; v64 += n_groups) {
    const int32_t v = (int32_t)v64;
    const int32_t p0 = ptr[v], p1 = ptr[v + 1];
    const int N = (p1 - p0) * k;
    if (N > PR_MAX) {
      if (t == 0) pattern_row_serial(v, k, ents, ptr, idx, scratch, rowlen);
      continue;
    }
    for (int e = t; e < N; e += PR_GL) s_cand[g][e] = ents[(int64_t)idx[p0 + e / k] * k + e % k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = t; e < N; e += PR_GL) {
      const int32_t val = s_cand[g][e];
      bool keep = true;
      for (int f = 0; f < e; ++f) keep = keep && (s_cand[g][f] != val);
      s_keep[g][e] = keep ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int32_t* list = scratch + (int64_t)k * p0;
    int kept = 0;
    for (int e = t; e < N; e += PR_GL) {
      if (!s_keep[g][e]) continue;
      const int32_t val = s_cand[g][e];
      int pos = 0;
      for (int f = 0; f < N; ++f) pos += (s_keep[g][f] && s_cand[g][f] < val) ? 1 : 0;
      list[pos] = val;
      ++kept;
    }
    // row length = number of kept candidates (sum over the group's lanes)
#pragma unroll
    for (int o = PR_GL / 2; o > 0; o >>= 1) kept += __shfl_xor(kept, o, PR_GL);
    if (t == 0) rowlen[v] = kept;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
__global__ void k_pattern_fill(int32_t n, int k, const int32_t* __restrict__ ptr, const int32_t* __restrict__ rowptr,
                               const int32_t* __restrict__ scratch, int32_t* __restrict__ col) {
  for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const int32_t* list = scratch + (int64_t)k * ptr[v];
    int32_t s = rowptr[v], e = rowptr[v + 1];
    for (int32_t t = s; t < e; ++t) col[t] = list[t - s];
  }
}
extern "C" int hx_p1_pattern_count(int32_t n_vert, int32_t n_ent, int k, const int32_t* ents,
                                   const int32_t* v2e_ptr, const int32_t* v2e_idx, int32_t* rowptr,
                                   int32_t* scratch, int64_t* h_nnz, hx_stream s_) {
  HX_REQUIRE(n_vert > 0 && n_ent >= 0 && k >= 1 && k <= 16 && v2e_ptr && rowptr && h_nnz, "bad argument");
  hipStream_t s = hx_s(s_);
  HX_HIP(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * ((size_t)n_vert + 1), s));
  if (n_ent > 0) {
    HX_REQUIRE(ents && v2e_idx && scratch, "null argument");
    k_pattern_rows<<<hx_grid(n_vert, 256 / PR_GL, 256 * 64), 256, 0, s>>>(n_vert, k, ents, v2e_ptr, v2e_idx, scratch, rowptr);
    HX_LAUNCH_CHECK("k_pattern_rows");
  }
  int rc = exclusive_scan_i32(rowptr, (int64_t)n_vert + 1, s);
  if (rc) return rc;
  int32_t last = 0;
  HX_HIP(hipMemcpy(&last, rowptr + n_vert, sizeof(int32_t), hipMemcpyDeviceToHost));
  *h_nnz = last;
  return HX_OK;
}
extern "C" int hx_p1_pattern_fill(int32_t n_vert, int k, const int32_t* v2e_ptr, const int32_t* rowptr,
                                  const int32_t* scratch, int32_t* col, hx_stream s) {
  HX_REQUIRE(n_vert > 0 && v2e_ptr && rowptr && scratch && col, "bad argument");
  k_pattern_fill<<<hx_grid(n_vert, 128), 128, 0, hx_s(s)>>>(n_vert, k, v2e_ptr, rowptr, scratch, col);
  HX_LAUNCH_CHECK("k_pattern_fill");
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ geometry
template <int D>
__device__ __forceinline__ void p1_geometry(const double* __restrict__ x, const int32_t* v, double G[D + 1][D],
                                            double& vol) {
  double X[D + 1][D];
#pragma unroll
  for (int a = 0; a <= D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) X[a][c] = x[(int64_t)v[a] * 3 + c];
  if (D == 1) {
    double h = X[1][0] - X[0][0];
    G[1][0] = 1.0 / h;
    G[0][0] = -G[1][0];
    vol = fabs(h);
  } else if (D == 2) {
    // J[:,i] = v_{i+1} - v_0 ; rows of J^{-1} are grad(phi_1), grad(phi_2)
    double a = X[1][0] - X[0][0], b = X[2][0] - X[0][0];
    double c = X[1][1] - X[0][1], d = X[2][1] - X[0][1];
    double det = a * d - b * c, id = 1.0 / det;
    G[1][0] = d * id;  G[1][1] = -b * id;
    G[2][0] = -c * id; G[2][1] = a * id;
    G[0][0] = -(G[1][0] + G[2][0]);
    G[0][1] = -(G[1][1] + G[2][1]);
    vol = 0.5 * fabs(det);
  } else {
    double J[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 3; ++i) J[c][i] = X[i + 1][c] - X[0][c];
    double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
    // inverse = adj/det ; row i of the inverse = grad(phi_{i+1})
    G[1][0] = c00 * id;
    G[1][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    G[1][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    G[2][0] = c01 * id;
    G[2][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    G[2][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    G[3][0] = c02 * id;
    G[3][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    G[3][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
#pragma unroll
    for (int c = 0; c < 3; ++c) G[0][c] = -(G[1][c] + G[2][c] + G[3][c]);
    vol = fabs(det) / 6.0;
  }
}

__device__ __forceinline__ int32_t find_slot(const int32_t* __restrict__ col, int32_t s, int32_t e, int32_t c) {
  int32_t lo = s, hi = e;
  while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (col[mid] < c) lo = mid + 1; else hi = mid; }
  return lo;  // pattern contains c by construction
}

// ------------------------------------------------------------------------------------------------ A and C
// A group of SM_GL = 16 lanes owns one CSR row.  Round by round, each lane takes one incident cell, computes its geometry
// and leaves the cell's K contributions to this row (column, grad.grad product, mass entry; the cell's weight int c^2)
// in the group's LDS region; then every lane scans the round's contributions IN ORDER (cell ascending, local vertex
// ascending: the order a single thread walking the row would use, so the sums are bitwise those of a serial walk) and
// adds the ones that fall on the column slots it owns (slots t, t + 16, ... of the row, held in registers).  Nothing
// is accumulated in global memory; rows longer than SM_GL * SM_Q entries take the single-thread walk below.
#define SM_GL 16
#define SM_Q 4
template <int D>
__global__ void __launch_bounds__(256)
k_stiff_mass(int32_t n, const double* __restrict__ x, const int32_t* __restrict__ cells,
             const int32_t* __restrict__ v2c_ptr, const int32_t* __restrict__ v2c_idx,
             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const zc* __restrict__ cf,
             int cellwise, zc* __restrict__ A, zc* __restrict__ C) {
  constexpr int K = D + 1;
  constexpr double MT = 1.0 / ((D + 1) * (D + 2));
  constexpr int GPB = 256 / SM_GL;                       // row groups per block
  __shared__ int32_t s_col[GPB][SM_GL * K];
  __shared__ double s_gg[GPB][SM_GL * K];
  __shared__ double s_cm[GPB][SM_GL * K];
  __shared__ zc s_w[GPB][SM_GL];
  const int g = threadIdx.x / SM_GL, t = threadIdx.x % SM_GL;
  const int64_t n_groups = (int64_t)gridDim.x * GPB;
  for (int64_t i64 = (int64_t)blockIdx.x * GPB + g; i64 < n; i64 += n_groups) {
    const int32_t i = (int32_t)i64;
    const int32_t rs = rowptr[i], re = rowptr[i + 1];
    if (re - rs > SM_GL * SM_Q) {
      // long row (never on a simplicial mesh of bounded vertex degree): lane 0 walks it, accumulating in place
      if (t == 0) {
        for (int32_t u = rs; u < re; ++u) {
          if (A) A[u] = make_double2(0.0, 0.0);
          if (C) C[u] = make_double2(0.0, 0.0);
        }
        for (int32_t p = v2c_ptr[i]; p < v2c_ptr[i + 1]; ++p) {
          const int32_t c = v2c_idx[p];
          int32_t v[K];
          int j = 0;
#pragma unroll
          for (int a = 0; a < K; ++a) { v[a] = cells[(int64_t)c * K + a]; if (v[a] == i) j = a; }
          double G[K][D], vol;
          p1_geometry<D>(x, v, G, vol);
          zc w = make_double2(0.0, 0.0);
          if (A) {
            if (cellwise) { const zc cc = cf[c]; w = zscale(vol, zmul(cc, cc)); }
            else {
              zc sum = make_double2(0.0, 0.0), sq = make_double2(0.0, 0.0);
#pragma unroll
              for (int a = 0; a < K; ++a) { zc ca = cf[v[a]]; sum = zadd(sum, ca); sq = zadd(sq, zmul(ca, ca)); }
              w = zscale(vol * MT, zadd(zmul(sum, sum), sq));
            }
          }
#pragma unroll
          for (int m = 0; m < K; ++m) {
            double gg = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) gg += G[j][d] * G[m][d];
            const int32_t slot = find_slot(col, rs, re, v[m]);
            if (A) { zc a = A[slot]; a.x -= w.x * gg; a.y -= w.y * gg; A[slot] = a; }
            if (C) { zc cm = C[slot]; cm.x += vol * MT * (m == j ? 2.0 : 1.0); C[slot] = cm; }
          }
        }
      }
      continue;
    }
    int32_t mycol[SM_Q];
    zc accA[SM_Q];
    double accC[SM_Q];
#pragma unroll
    for (int q = 0; q < SM_Q; ++q) {
      const int32_t u = rs + t + q * SM_GL;
      mycol[q] = u < re ? col[u] : -1;
      accA[q] = make_double2(0.0, 0.0);
      accC[q] = 0.0;
    }
    const int32_t p0 = v2c_ptr[i], p1 = v2c_ptr[i + 1];
    for (int32_t pb = p0; pb < p1; pb += SM_GL) {
      const int nround = min(p1 - pb, SM_GL);
      if (t < nround) {
        const int32_t c = v2c_idx[pb + t];
        int32_t v[K];
        int j = 0;
#pragma unroll
        for (int a = 0; a < K; ++a) { v[a] = cells[(int64_t)c * K + a]; if (v[a] == i) j = a; }
        double G[K][D], vol;
        p1_geometry<D>(x, v, G, vol);
        zc w = make_double2(0.0, 0.0);  // int_K c^2 dx  (complex square, acoustic_matrices.py:101)
        if (A) {
          if (cellwise) { const zc cc = cf[c]; w = zscale(vol, zmul(cc, cc)); }
          else {
            zc sum = make_double2(0.0, 0.0), sq = make_double2(0.0, 0.0);
#pragma unroll
            for (int a = 0; a < K; ++a) { zc ca = cf[v[a]]; sum = zadd(sum, ca); sq = zadd(sq, zmul(ca, ca)); }
            w = zscale(vol * MT, zadd(zmul(sum, sum), sq));
          }
        }
        s_w[g][t] = w;
#pragma unroll
        for (int m = 0; m < K; ++m) {
          double gg = 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) gg += G[j][d] * G[m][d];
          s_col[g][t * K + m] = v[m];
          s_gg[g][t * K + m] = gg;
          s_cm[g][t * K + m] = vol * MT * (m == j ? 2.0 : 1.0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int e = 0; e < nround * K; ++e) {
        const int32_t cc = s_col[g][e];
#pragma unroll
        for (int q = 0; q < SM_Q; ++q) {
          if (cc == mycol[q]) {
            const double gg = s_gg[g][e];
            const zc w = s_w[g][e / K];
            accA[q].x -= w.x * gg;
            accA[q].y -= w.y * gg;
            accC[q] += s_cm[g][e];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int q = 0; q < SM_Q; ++q) {
      const int32_t u = rs + t + q * SM_GL;
      if (u < re) {
        if (A) A[u] = accA[q];
        if (C) C[u] = make_double2(accC[q], 0.0);
      }
    }
  }
}

extern "C" int hx_assemble_p1_stiff_mass(int dim, int32_t n_vert, const double* x, const int32_t* cells,
                                         const int32_t* v2c_ptr, const int32_t* v2c_idx, const int32_t* rowptr,
                                         const int32_t* col, const hx_c128* c, int c_is_cellwise,
                                         hx_c128* A_val, hx_c128* C_val, hx_stream s) {
  HX_REQUIRE(dim >= 1 && dim <= 3 && n_vert > 0 && x && cells && v2c_ptr && v2c_idx && rowptr && col, "bad argument");
  HX_REQUIRE(A_val || C_val, "nothing to assemble");
  HX_REQUIRE(!A_val || c, "A needs the sound-speed field");
  const int grid = hx_grid(n_vert, 256 / SM_GL, 256 * 64);
#define HX_GO(D) k_stiff_mass<D><<<grid, 256, 0, hx_s(s)>>>(n_vert, x, cells, v2c_ptr, v2c_idx, rowptr, col, \
                                                              (const zc*)c, c_is_cellwise, (zc*)A_val, (zc*)C_val)
  if (dim == 1) HX_GO(1); else if (dim == 2) HX_GO(2); else HX_GO(3);
#undef HX_GO
  HX_LAUNCH_CHECK("k_stiff_mass");
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ Robin B
// triple-product template on a unit-measure simplex of dimension DS: DS! a!b!c!/(DS+3)!
__device__ __forceinline__ double triple_T(int DS, int a, int b, int c) {
  const double base = DS == 0 ? 1.0 / 6.0 : DS == 1 ? 1.0 / 24.0 : DS == 2 ? 1.0 / 60.0 : 1.0 / 120.0;
  const int mult = (a == b && b == c) ? 6 : (a == b || b == c || a == c) ? 2 : 1;
  return base * mult;
}

template <int KF>  // facet vertex count = cell dim
__global__ void __launch_bounds__(128)
k_robin(int32_t n, const double* __restrict__ x, const int32_t* __restrict__ facets,
        const int32_t* __restrict__ v2f_ptr, const int32_t* __restrict__ v2f_idx, const int32_t* __restrict__ rowptr,
        const int32_t* __restrict__ col, const zc* __restrict__ cf, int cellwise,
        const int32_t* __restrict__ facet_cell, const zc* __restrict__ coef, zc* __restrict__ B) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t rs = rowptr[i], re = rowptr[i + 1];
    for (int32_t t = rs; t < re; ++t) B[t] = make_double2(0.0, 0.0);
    for (int32_t p = v2f_ptr[i]; p < v2f_ptr[i + 1]; ++p) {
      const int32_t f = v2f_idx[p];
      int32_t v[KF];
      int j = 0;
#pragma unroll
      for (int a = 0; a < KF; ++a) { v[a] = facets[(int64_t)f * KF + a]; if (v[a] == i) j = a; }
      double meas = 1.0;
      if (KF == 2) {
        double dx = x[(int64_t)v[1] * 3] - x[(int64_t)v[0] * 3], dy = x[(int64_t)v[1] * 3 + 1] - x[(int64_t)v[0] * 3 + 1];
        meas = sqrt(dx * dx + dy * dy);
      } else if (KF == 3) {
        double e1[3], e2[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          e1[d] = x[(int64_t)v[1] * 3 + d] - x[(int64_t)v[0] * 3 + d];
          e2[d] = x[(int64_t)v[2] * 3 + d] - x[(int64_t)v[0] * 3 + d];
        }
        double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        meas = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
      }
      const zc kf = zscale(meas, coef[f]);
#pragma unroll
      for (int m = 0; m < KF; ++m) {
        zc e;
        if (cellwise) {
          e = zscale((m == j ? 2.0 : 1.0) / (KF * (KF + 1)), cf[facet_cell[f]]);
        } else {
          e = make_double2(0.0, 0.0);
#pragma unroll
          for (int q = 0; q < KF; ++q) e = zadd(e, zscale(triple_T(KF - 1, q, j, m), cf[v[q]]));
        }
        const int32_t slot = find_slot(col, rs, re, v[m]);
        zc b = B[slot];
        zfma(b, kf, e);
        B[slot] = b;
      }
    }
  }
}

extern "C" int hx_assemble_p1_robin(int dim, int32_t n_vert, const double* x, const int32_t* facets,
                                    const int32_t* v2f_ptr, const int32_t* v2f_idx, const int32_t* rowptr,
                                    const int32_t* col, const hx_c128* c, int c_is_cellwise,
                                    const int32_t* facet_cell, const hx_c128* coef_per_facet, hx_c128* B_val,
                                    hx_stream s) {
  HX_REQUIRE(dim >= 1 && dim <= 3 && n_vert > 0 && x && facets && v2f_ptr && v2f_idx && rowptr && col && c &&
                 coef_per_facet && B_val, "bad argument");
  HX_REQUIRE(!c_is_cellwise || facet_cell, "cellwise c needs facet_cell");
  const int grid = hx_grid(n_vert, 128, 256 * 16);
#define HX_GO(KF) k_robin<KF><<<grid, 128, 0, hx_s(s)>>>(n_vert, x, facets, v2f_ptr, v2f_idx, rowptr, col, (const zc*)c, \
                                                          c_is_cellwise, facet_cell, (const zc*)coef_per_facet, (zc*)B_val)
  if (dim == 1) HX_GO(1); else if (dim == 2) HX_GO(2); else HX_GO(3);
#undef HX_GO
  HX_LAUNCH_CHECK("k_robin");
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ Dirichlet
__global__ void k_dirichlet(int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                            zc* __restrict__ val, const uint8_t* __restrict__ is_bc, double diag) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool rbc = is_bc[i] != 0;
    for (int32_t t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      const int32_t c = col[t];
      if (rbc) val[t] = make_double2(c == i ? diag : 0.0, 0.0);
      else if (is_bc[c]) val[t] = make_double2(0.0, 0.0);
    }
  }
}
extern "C" int hx_csr_apply_dirichlet(int32_t n, const int32_t* rowptr, const int32_t* col, hx_c128* val,
                                      const uint8_t* is_bc, double diag, hx_stream s) {
  HX_REQUIRE(n > 0 && rowptr && col && val && is_bc, "bad argument");
  k_dirichlet<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rowptr, col, (zc*)val, is_bc, diag);
  HX_LAUNCH_CHECK("k_dirichlet");
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ load vectors
template <int D>
struct Quad2;  // degree-2 simplex rule in barycentric coordinates (the rule FFCx selects for kind 2)
template <> struct Quad2<1> { static constexpr int NQ = 2; };
template <> struct Quad2<2> { static constexpr int NQ = 3; };
template <> struct Quad2<3> { static constexpr int NQ = 4; };

template <int D>
__device__ __forceinline__ double quad2_bary(int q, int a) {
  if (D == 1) { const double g = 0.28867513459481287; return (q == a) ? 0.5 + g : 0.5 - g; }
  if (D == 2) return (q == a) ? 2.0 / 3.0 : 1.0 / 6.0;
  return (q == a) ? 0.5854101966249685 : 0.1381966011250105;
}

// load-vector entry of vertex i: sum over its incident cells (ascending), optionally only those tagged `tag`
template <int D, int KIND>
__device__ __forceinline__ zc load_row(int32_t i, const double* __restrict__ x, const int32_t* __restrict__ cells,
                                       const int32_t* __restrict__ v2c_ptr, const int32_t* __restrict__ v2c_idx,
                                       const zc* __restrict__ fa, const zc* __restrict__ fb,
                                       const int32_t* __restrict__ cell_tag, int32_t tag) {
  constexpr int K = D + 1;
  zc acc = make_double2(0.0, 0.0);
  for (int32_t p = v2c_ptr[i]; p < v2c_ptr[i + 1]; ++p) {
    const int32_t c = v2c_idx[p];
    if (cell_tag && cell_tag[c] != tag) continue;
    int32_t v[K];
    int j = 0;
#pragma unroll
    for (int a = 0; a < K; ++a) { v[a] = cells[(int64_t)c * K + a]; if (v[a] == i) j = a; }
    double G[K][D], vol;
    p1_geometry<D>(x, v, G, vol);
    if (KIND == 0) {
      acc = zadd(acc, zscale(vol / K, fa[c]));
    } else if (KIND == 1) {
      zc e = make_double2(0.0, 0.0);
#pragma unroll
      for (int m = 0; m < K; ++m) {
        zc am = fa[v[m]];
#pragma unroll
        for (int l = 0; l < K; ++l) zfma(e, zscale(triple_T(D, j, m, l), am), fb[v[l]]);
      }
      acc = zadd(acc, zscale(vol, e));
    } else {
      zc sq = make_double2(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < Quad2<D>::NQ; ++q) {
        zc rq = make_double2(0.0, 0.0), wq = make_double2(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const double l = quad2_bary<D>(q, a);
          rq = zadd(rq, zscale(l, fa[v[a]]));
          wq = zadd(wq, zscale(l, fb[v[a]]));
        }
        sq = zadd(sq, zscale(1.0 / Quad2<D>::NQ, zdiv(wq, rq)));
      }
      const double gn = G[j][D == 3 ? 2 : 0];   // n_r = e_x (1D/2D), e_z (3D): flame_matrices.py:35-40
      acc = zadd(acc, zscale(vol * gn, sq));
    }
  }
  return acc;
}

template <int D, int KIND>
__global__ void __launch_bounds__(128)
k_load(int32_t n, const double* __restrict__ x, const int32_t* __restrict__ cells, const int32_t* __restrict__ v2c_ptr,
       const int32_t* __restrict__ v2c_idx, const zc* __restrict__ fa, const zc* __restrict__ fb, zc scale,
       const int32_t* __restrict__ cell_tag, int32_t tag, zc* __restrict__ out) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = zmul(scale, load_row<D, KIND>(i, x, cells, v2c_ptr, v2c_idx, fa, fb, cell_tag, tag));
}
// the same for a LIST of vertices, each with its own tag (the supports of the flames: a few hundred vertices each)
template <int D, int KIND>
__global__ void __launch_bounds__(128)
k_load_rows(int32_t n_rows, const int32_t* __restrict__ rows, const int32_t* __restrict__ row_tag, const double* __restrict__ x,
            const int32_t* __restrict__ cells, const int32_t* __restrict__ v2c_ptr, const int32_t* __restrict__ v2c_idx,
            const zc* __restrict__ fa, const zc* __restrict__ fb, zc scale, const int32_t* __restrict__ cell_tag,
            zc* __restrict__ out) {
  for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_rows; t += gridDim.x * blockDim.x)
    out[t] = zmul(scale, load_row<D, KIND>(rows[t], x, cells, v2c_ptr, v2c_idx, fa, fb, cell_tag, row_tag ? row_tag[t] : 0));
}

extern "C" int hx_assemble_p1_load(int kind, int dim, int32_t n_vert, const double* x, const int32_t* cells,
                                   const int32_t* v2c_ptr, const int32_t* v2c_idx, const hx_c128* fa,
                                   const hx_c128* fb, const double scale[2], const int32_t* cell_tag, int32_t tag,
                                   hx_c128* out, hx_stream s) {
  HX_REQUIRE(kind >= 0 && kind <= 2 && dim >= 1 && dim <= 3 && n_vert > 0 && x && cells && v2c_ptr && v2c_idx && fa &&
                 scale && out, "bad argument");
  HX_REQUIRE(kind == 0 || fb, "kind 1/2 need both fields");
  const int grid = hx_grid(n_vert, 128, 256 * 16);
  const zc sc = hx_z(scale);
#define HX_GO(D, KD) k_load<D, KD><<<grid, 128, 0, hx_s(s)>>>(n_vert, x, cells, v2c_ptr, v2c_idx, (const zc*)fa, \
                                                               (const zc*)fb, sc, cell_tag, tag, (zc*)out)
#define HX_DIM(KD) do { if (dim == 1) HX_GO(1, KD); else if (dim == 2) HX_GO(2, KD); else HX_GO(3, KD); } while (0)
  if (kind == 0) HX_DIM(0); else if (kind == 1) HX_DIM(1); else HX_DIM(2);
#undef HX_DIM
#undef HX_GO
  HX_LAUNCH_CHECK("k_load");
  return HX_OK;
}


// out[t] = entry rows[t] of the load vector of hx_assemble_p1_load, restricted (when cell_tag is given) to the cells tagged
// row_tag[t]: the flames' load vectors (flame_matrices.py:139-141) live on a few hundred vertices each.
extern "C" int hx_assemble_p1_load_rows(int kind, int dim, int32_t n_rows, const int32_t* rows, const int32_t* row_tag,
                                        const double* x, const int32_t* cells, const int32_t* v2c_ptr, const int32_t* v2c_idx,
                                        const hx_c128* fa, const hx_c128* fb, const double scale[2], const int32_t* cell_tag,
                                        hx_c128* out, hx_stream s) {
  HX_REQUIRE(kind >= 0 && kind <= 2 && dim >= 1 && dim <= 3 && n_rows >= 0 && x && cells && v2c_ptr && v2c_idx && fa && scale,
             "bad argument");
  HX_REQUIRE(kind == 0 || fb, "kind 1/2 need both fields");
  HX_REQUIRE(!cell_tag || row_tag, "tagged cells need a tag per row");
  if (n_rows == 0) return HX_OK;
  HX_REQUIRE(rows && out, "null argument");
  const int grid = hx_grid(n_rows, 128, 256 * 16);
  const zc sc = hx_z(scale);
#define HX_GO(D, KD) k_load_rows<D, KD><<<grid, 128, 0, hx_s(s)>>>(n_rows, rows, row_tag, x, cells, v2c_ptr, v2c_idx, \
                                                                    (const zc*)fa, (const zc*)fb, sc, cell_tag, (zc*)out)
#define HX_DIM(KD) do { if (dim == 1) HX_GO(1, KD); else if (dim == 2) HX_GO(2, KD); else HX_GO(3, KD); } while (0)
  if (kind == 0) HX_DIM(0); else if (kind == 1) HX_DIM(1); else HX_DIM(2);
#undef HX_DIM
#undef HX_GO
  HX_LAUNCH_CHECK("k_load_rows");
  return HX_OK;
}


// ------------------------------------------------------------------------------------------------ cell volumes / point location
template <int D>
__global__ void k_cell_volumes(int32_t nc, const double* __restrict__ x, const int32_t* __restrict__ cells,
                               double* __restrict__ vol) {
  constexpr int K = D + 1;
  for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    int32_t v[K];
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] = cells[(int64_t)c * K + a];
    double G[K][D], w;
    p1_geometry<D>(x, v, G, w);
    vol[c] = w;
  }
}
extern "C" int hx_p1_cell_volumes(int dim, int32_t n_cells, const double* x, const int32_t* cells, double* vol,
                                  hx_stream s) {
  HX_REQUIRE(dim >= 1 && dim <= 3 && n_cells > 0 && x && cells && vol, "bad argument");
  const int grid = hx_grid(n_cells, 256);
  if (dim == 1) k_cell_volumes<1><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, vol);
  else if (dim == 2) k_cell_volumes<2><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, vol);
  else k_cell_volumes<3><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, vol);
  HX_LAUNCH_CHECK("k_cell_volumes");
  return HX_OK;
}

// For every query point: the lowest-index cell whose barycentric coordinates are all >= -tol
// (DOLFINx determine_point_ownership, flame_matrices.py:144-145).  out_cell must be pre-filled with INT32_MAX.
template <int D>
__global__ void k_locate_points(int32_t nc, const double* __restrict__ x, const int32_t* __restrict__ cells,
                                int npts, const double* __restrict__ pts, double tol, int32_t* __restrict__ out_cell) {
  constexpr int K = D + 1;
  for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    int32_t v[K];
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] = cells[(int64_t)c * K + a];
    double G[K][D], w;
    p1_geometry<D>(x, v, G, w);
    double x0[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x0[d] = x[(int64_t)v[0] * 3 + d];
    for (int q = 0; q < npts; ++q) {
      // lambda_i = grad(phi_i).(p - x0) for i >= 1, lambda_0 = 1 - sum
      double lam0 = 1.0;
      bool inside = true;
#pragma unroll
      for (int i = 1; i < K; ++i) {
        double l = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) l += G[i][d] * (pts[q * 3 + d] - x0[d]);
        lam0 -= l;
        inside = inside && (l >= -tol);
      }
      if (inside && lam0 >= -tol) atomicMin(&out_cell[q], c);
    }
  }
}
extern "C" int hx_p1_locate_points(int dim, int32_t n_cells, const double* x, const int32_t* cells, int n_pts,
                                   const double* pts, double tol, int32_t* out_cell, hx_stream s) {
  HX_REQUIRE(dim >= 1 && dim <= 3 && n_cells > 0 && x && cells && n_pts > 0 && pts && out_cell, "bad argument");
  const int grid = hx_grid(n_cells, 256);
  if (dim == 1) k_locate_points<1><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, n_pts, pts, tol, out_cell);
  else if (dim == 2) k_locate_points<2><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, n_pts, pts, tol, out_cell);
  else k_locate_points<3><<<grid, 256, 0, hx_s(s)>>>(n_cells, x, cells, n_pts, pts, tol, out_cell);
  HX_LAUNCH_CHECK("k_locate_points");
  return HX_OK;
}


// ================================================================================================ P2 on tetrahedra
// Quadratic Lagrange trial/test space with P1 coefficient fields (the reference's degree = 2 runs,
// ShapeSensitivities/RijkeFFD/main_shape.py:68-83).  Same owner-computes scheme as P1: one thread per CSR row (= P2 dof:
// a mesh vertex or an edge midpoint) gathers its incident cells in ascending order -> deterministic, no atomics.
// Local dof order of a cell: 4 vertices, then the 6 edges (0,1),(0,2),(0,3),(1,2),(1,3),(2,3).  Integrals by a quadrature
// table handed in by the host (barycentric points + weights summing to 1): nq x 5 doubles for cells, nq x 4 for facets.
__device__ __constant__ int P2_TET_EDGE[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
__device__ __constant__ int P2_TRI_EDGE[3][2] = {{0, 1}, {0, 2}, {1, 2}};

// value and d/dlambda of P2 basis function b on a simplex with NV vertices
template <int NV>
__device__ __forceinline__ double p2_basis(int b, const double* lam, double* dl) {
#pragma unroll
  for (int m = 0; m < NV; ++m) dl[m] = 0.0;
  if (b < NV) { dl[b] = 4.0 * lam[b] - 1.0; return lam[b] * (2.0 * lam[b] - 1.0); }
  const int e = b - NV;
  const int a0 = NV == 4 ? P2_TET_EDGE[e][0] : P2_TRI_EDGE[e][0], a1 = NV == 4 ? P2_TET_EDGE[e][1] : P2_TRI_EDGE[e][1];
  dl[a0] = 4.0 * lam[a1];
  dl[a1] = 4.0 * lam[a0];
  return 4.0 * lam[a0] * lam[a1];
}

__global__ void __launch_bounds__(128)
k_p2_stiff_mass(int32_t n_dof, const double* __restrict__ x, const int32_t* __restrict__ cells,
                const int32_t* __restrict__ cdofs, const int32_t* __restrict__ d2c_ptr, const int32_t* __restrict__ d2c_idx,
                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const zc* __restrict__ cf,
                int nq, const double* __restrict__ quad, zc* __restrict__ A, zc* __restrict__ C) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dof; i += gridDim.x * blockDim.x) {
    const int32_t rs = rowptr[i], re = rowptr[i + 1];
    for (int32_t t = rs; t < re; ++t) { if (A) A[t] = make_double2(0, 0); if (C) C[t] = make_double2(0, 0); }
    for (int32_t p = d2c_ptr[i]; p < d2c_ptr[i + 1]; ++p) {
      const int32_t cell = d2c_idx[p];
      const int32_t* v = cells + (int64_t)cell * 4;
      const int32_t* dofs = cdofs + (int64_t)cell * 10;
      int j = 0;
      for (int t = 0; t < 10; ++t) if (dofs[t] == i) j = t;
      double G[4][3], vol;
      p1_geometry<3>(x, v, G, vol);
      zc cv[4];
      for (int a = 0; a < 4; ++a) cv[a] = cf ? cf[v[a]] : make_double2(1.0, 0.0);
      zc rowA[10];
      double rowC[10];
      for (int k = 0; k < 10; ++k) { rowA[k] = make_double2(0, 0); rowC[k] = 0.0; }
      for (int q = 0; q < nq; ++q) {
        const double* lam = quad + 5 * q;
        const double w = lam[4] * vol;
        double dl[4];
        const double pj = p2_basis<4>(j, lam, dl);
        double gj[3] = {0, 0, 0};
        for (int m = 0; m < 4; ++m) for (int c = 0; c < 3; ++c) gj[c] += dl[m] * G[m][c];
        zc cq = make_double2(0, 0);
        for (int a = 0; a < 4; ++a) { cq.x += lam[a] * cv[a].x; cq.y += lam[a] * cv[a].y; }
        const zc c2 = zmul(cq, cq);
        for (int k = 0; k < 10; ++k) {
          const double pk = p2_basis<4>(k, lam, dl);
          double gk[3] = {0, 0, 0};
          for (int m = 0; m < 4; ++m) for (int c = 0; c < 3; ++c) gk[c] += dl[m] * G[m][c];
          const double gg = w * (gj[0] * gk[0] + gj[1] * gk[1] + gj[2] * gk[2]);
          rowA[k].x -= gg * c2.x;
          rowA[k].y -= gg * c2.y;
          rowC[k] += w * pj * pk;
        }
      }
      for (int k = 0; k < 10; ++k) {
        const int32_t pos = find_slot(col, rs, re, dofs[k]);
        if (A) { A[pos].x += rowA[k].x; A[pos].y += rowA[k].y; }
        if (C) C[pos].x += rowC[k];
      }
    }
  }
}
extern "C" int hx_assemble_p2_stiff_mass(int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                                         const int32_t* d2c_ptr, const int32_t* d2c_idx, const int32_t* rowptr,
                                         const int32_t* col, const hx_c128* c, int nq, const double* quad, hx_c128* A,
                                         hx_c128* C, hx_stream s) {
  HX_REQUIRE(n_dof > 0 && x && cells && cell_dofs && d2c_ptr && d2c_idx && rowptr && col && nq > 0 && quad && (A || C),
             "bad argument");
  k_p2_stiff_mass<<<hx_grid(n_dof, 128), 128, 0, hx_s(s)>>>(n_dof, x, cells, cell_dofs, d2c_ptr, d2c_idx, rowptr, col,
                                                           (const zc*)c, nq, quad, (zc*)A, (zc*)C);
  HX_LAUNCH_CHECK("k_p2_stiff_mass");
  return HX_OK;
}

// B = sum_f coef_f int_F c phi_k phi_j ds over boundary triangles (P2 on the facet: 3 vertices + 3 edges)
__global__ void __launch_bounds__(128)
k_p2_robin(int32_t n_dof, const double* __restrict__ x, const int32_t* __restrict__ facets,
           const int32_t* __restrict__ fdofs, const int32_t* __restrict__ d2f_ptr, const int32_t* __restrict__ d2f_idx,
           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const zc* __restrict__ cf,
           const zc* __restrict__ coef, int nq, const double* __restrict__ quad, zc* __restrict__ B) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dof; i += gridDim.x * blockDim.x) {
    const int32_t rs = rowptr[i], re = rowptr[i + 1];
    for (int32_t t = rs; t < re; ++t) B[t] = make_double2(0, 0);
    for (int32_t p = d2f_ptr[i]; p < d2f_ptr[i + 1]; ++p) {
      const int32_t f = d2f_idx[p];
      const int32_t* v = facets + (int64_t)f * 3;
      const int32_t* dofs = fdofs + (int64_t)f * 6;
      int j = 0;
      for (int t = 0; t < 6; ++t) if (dofs[t] == i) j = t;
      double e1[3], e2[3];
      for (int c = 0; c < 3; ++c) { e1[c] = x[(int64_t)v[1] * 3 + c] - x[(int64_t)v[0] * 3 + c];
                                    e2[c] = x[(int64_t)v[2] * 3 + c] - x[(int64_t)v[0] * 3 + c]; }
      const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
      const double area = 0.5 * sqrt(nx * nx + ny * ny + nz * nz);
      zc cv[3];
      for (int a = 0; a < 3; ++a) cv[a] = cf[v[a]];
      zc row[6];
      for (int k = 0; k < 6; ++k) row[k] = make_double2(0, 0);
      for (int q = 0; q < nq; ++q) {
        const double* lam = quad + 4 * q;
        const double w = lam[3] * area;
        double dl[3];
        const double pj = p2_basis<3>(j, lam, dl);
        zc cq = make_double2(0, 0);
        for (int a = 0; a < 3; ++a) { cq.x += lam[a] * cv[a].x; cq.y += lam[a] * cv[a].y; }
        for (int k = 0; k < 6; ++k) {
          const double pk = p2_basis<3>(k, lam, dl);
          row[k].x += w * pj * pk * cq.x;
          row[k].y += w * pj * pk * cq.y;
        }
      }
      const zc cz = coef[f];
      for (int k = 0; k < 6; ++k) {
        const int32_t pos = find_slot(col, rs, re, dofs[k]);
        const zc t = zmul(cz, row[k]);
        B[pos].x += t.x;
        B[pos].y += t.y;
      }
    }
  }
}
extern "C" int hx_assemble_p2_robin(int32_t n_dof, const double* x, const int32_t* facets, const int32_t* facet_dofs,
                                    const int32_t* d2f_ptr, const int32_t* d2f_idx, const int32_t* rowptr,
                                    const int32_t* col, const hx_c128* c, const hx_c128* coef, int nq, const double* quad,
                                    hx_c128* B, hx_stream s) {
  HX_REQUIRE(n_dof > 0 && x && facets && facet_dofs && d2f_ptr && d2f_idx && rowptr && col && c && coef && nq > 0 && quad && B,
             "bad argument");
  k_p2_robin<<<hx_grid(n_dof, 128), 128, 0, hx_s(s)>>>(n_dof, x, facets, facet_dofs, d2f_ptr, d2f_idx, rowptr, col,
                                                      (const zc*)c, (const zc*)coef, nq, quad, (zc*)B);
  HX_LAUNCH_CHECK("k_p2_robin");
  return HX_OK;
}

// load vectors on the P2 space with P1 fields fa, fb:
//   kind 1: out_j = scale * int fa fb phi_j dx                 (flame_matrices.py:199 with fa = gamma - 1, fb = h)
//   kind 2: out_j = scale * int (d phi_j / dz) fb / fa dx      (flame_matrices.py:200 with fa = rho, fb = w; n_r = e_z)
__global__ void __launch_bounds__(128)
k_p2_load(int kind, int32_t n_dof, const double* __restrict__ x, const int32_t* __restrict__ cells,
          const int32_t* __restrict__ cdofs, const int32_t* __restrict__ d2c_ptr, const int32_t* __restrict__ d2c_idx,
          const zc* __restrict__ fa, const zc* __restrict__ fb, zc scale, int nq, const double* __restrict__ quad,
          zc* __restrict__ out) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dof; i += gridDim.x * blockDim.x) {
    zc acc = make_double2(0, 0);
    for (int32_t p = d2c_ptr[i]; p < d2c_ptr[i + 1]; ++p) {
      const int32_t cell = d2c_idx[p];
      const int32_t* v = cells + (int64_t)cell * 4;
      const int32_t* dofs = cdofs + (int64_t)cell * 10;
      int j = 0;
      for (int t = 0; t < 10; ++t) if (dofs[t] == i) j = t;
      double G[4][3], vol;
      p1_geometry<3>(x, v, G, vol);
      zc av[4], bv[4];
      for (int a = 0; a < 4; ++a) { av[a] = fa[v[a]]; bv[a] = fb[v[a]]; }
      for (int q = 0; q < nq; ++q) {
        const double* lam = quad + 5 * q;
        const double w = lam[4] * vol;
        double dl[4];
        const double pj = p2_basis<4>(j, lam, dl);
        zc aq = make_double2(0, 0), bq = make_double2(0, 0);
        for (int a = 0; a < 4; ++a) { aq.x += lam[a] * av[a].x; aq.y += lam[a] * av[a].y;
                                      bq.x += lam[a] * bv[a].x; bq.y += lam[a] * bv[a].y; }
        if (kind == 1) {
          const zc t = zmul(aq, bq);
          acc.x += w * pj * t.x;
          acc.y += w * pj * t.y;
        } else {
          double gz = 0.0;
          for (int m = 0; m < 4; ++m) gz += dl[m] * G[m][2];
          const zc t = zdiv(bq, aq);
          acc.x += w * gz * t.x;
          acc.y += w * gz * t.y;
        }
      }
    }
    out[i] = zmul(scale, acc);
  }
}
extern "C" int hx_assemble_p2_load(int kind, int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                                   const int32_t* d2c_ptr, const int32_t* d2c_idx, const hx_c128* fa, const hx_c128* fb,
                                   const double scale[2], int nq, const double* quad, hx_c128* out, hx_stream s) {
  HX_REQUIRE((kind == 1 || kind == 2) && n_dof > 0 && x && cells && cell_dofs && d2c_ptr && d2c_idx && fa && fb && scale &&
             nq > 0 && quad && out, "bad argument");
  k_p2_load<<<hx_grid(n_dof, 128), 128, 0, hx_s(s)>>>(kind, n_dof, x, cells, cell_dofs, d2c_ptr, d2c_idx, (const zc*)fa,
                                                     (const zc*)fb, hx_z(scale), nq, quad, (zc*)out);
  HX_LAUNCH_CHECK("k_p2_load");
  return HX_OK;
}

// Pointwise flames on the quadratic space (PointwiseFlameMatrix(degree = 2), flame_matrices.py:139-141):
//   out_j = scale * sum over the cells tagged `tag` of  h_cell * int_cell phi_j dx      (h: DG0, one value per cell)
// one thread per dof over its incident cells, like k_p2_load.
__global__ void __launch_bounds__(128)
k_p2_load_cells(int32_t n_dof, const double* __restrict__ x, const int32_t* __restrict__ cells, const int32_t* __restrict__ cdofs,
                const int32_t* __restrict__ d2c_ptr, const int32_t* __restrict__ d2c_idx, const zc* __restrict__ h_cell,
                const int32_t* __restrict__ cell_tag, int32_t tag, zc scale, int nq, const double* __restrict__ quad,
                zc* __restrict__ out) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dof; i += gridDim.x * blockDim.x) {
    zc acc = make_double2(0, 0);
    for (int32_t p = d2c_ptr[i]; p < d2c_ptr[i + 1]; ++p) {
      const int32_t cell = d2c_idx[p];
      if (cell_tag[cell] != tag) continue;
      const int32_t* v = cells + (int64_t)cell * 4;
      const int32_t* dofs = cdofs + (int64_t)cell * 10;
      int j = 0;
      for (int t = 0; t < 10; ++t) if (dofs[t] == i) j = t;
      double G[4][3], vol;
      p1_geometry<3>(x, v, G, vol);
      double integral = 0.0;
      for (int q = 0; q < nq; ++q) {
        const double* lam = quad + 5 * q;
        double dl[4];
        integral += lam[4] * vol * p2_basis<4>(j, lam, dl);
      }
      const zc hc = h_cell[cell];
      acc.x += integral * hc.x;
      acc.y += integral * hc.y;
    }
    out[i] = zmul(scale, acc);
  }
}
extern "C" int hx_assemble_p2_load_cells(int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                                         const int32_t* d2c_ptr, const int32_t* d2c_idx, const hx_c128* h_cell,
                                         const int32_t* cell_tag, int32_t tag, const double scale[2], int nq, const double* quad,
                                         hx_c128* out, hx_stream s) {
  HX_REQUIRE(n_dof > 0 && x && cells && cell_dofs && d2c_ptr && d2c_idx && h_cell && cell_tag && scale && nq > 0 && quad && out,
             "bad argument");
  k_p2_load_cells<<<hx_grid(n_dof, 128), 128, 0, hx_s(s)>>>(n_dof, x, cells, cell_dofs, d2c_ptr, d2c_idx, (const zc*)h_cell, cell_tag,
                                                           tag, hx_z(scale), nq, quad, (zc*)out);
  HX_LAUNCH_CHECK("k_p2_load_cells");
  return HX_OK;
}
