// DEVICE set-up passes of the smoothed-aggregation hierarchy (once per shifted operator K): strength filter, power
// iteration for rho(D^-1 L), smoothed prolongator, transpose, and the Galerkin products R K P as two sparse
// matrix-matrix products.  Round 1 ran these passes on the host (1.0 s of a 6.5 s fixed-point call at 5 M DOF, after a
// 1.5 GB device-to-host copy of K); here K never leaves HBM -- only the filtered strength graph travels to the host
// for the greedy aggregation, which is inherently sequential (and cheap: 0.04 s at 5 M rows).
//
// Replaces (with everything in amg.hip) what the reference delegates to PETSc PCLU behind ST sinvert
// (eigensolvers.py:49-50, 102-103).  All kernels are one-off integer / gather work, latency- and L2-bound, far from any
// roofline that matters; they are written for simplicity and determinism of the PATTERNS (sorted CSR rows).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "hx_common.h"

// ---------------------------------------------------------------- small helpers
template <typename T>
static int exclusive_scan_inplace(T* d, int64_t count, hipStream_t s) {
  size_t bytes = 0;
  HX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d, d, (int)count, s));
  void* tmp = nullptr;
  HX_HIP(hx_malloc(&tmp, bytes ? bytes : 16));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, d, d, (int)count, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hx_free(tmp);
  HX_HIP(e);
  return HX_OK;
}

// ---------------------------------------------------------------- diagonal of a complex CSR
__global__ void k_csr_diag(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                           const zc* __restrict__ val, zc* __restrict__ diag) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    zc d = make_double2(0.0, 0.0);
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] == i) { d = val[p]; break; }
    diag[i] = d;
  }
}
extern "C" int hx_csr_diagonal_z(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, hx_c128* diag,
                                 hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && val && diag, "bad argument");
  k_csr_diag<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, (const zc*)val, (zc*)diag);
  HX_LAUNCH_CHECK("k_csr_diag");
  return HX_OK;
}

// ---------------------------------------------------------------- strength filter
// L = sign * Re(K).  Keep the diagonal and the couplings |l_ij| >= theta sqrt(|l_ii| |l_jj|); weak couplings are lumped
// onto the diagonal.  Output Lf: compact CSR with sorted columns, diagonal entry always present.
__global__ void k_sa_sd(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const zc* __restrict__ val,
                        double sign, double* __restrict__ sd) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double d = 0.0;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] == i) { d = sign * val[p].x; break; }
    d = fabs(d);
    sd[i] = sqrt(d == 0.0 ? 1.0 : d);
  }
}
template <bool FILL>
__global__ void k_sa_filter(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                            const zc* __restrict__ val, double sign, double theta, const double* __restrict__ sd,
                            int32_t* __restrict__ out_rp, int32_t* __restrict__ out_ci, double* __restrict__ out_v) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double ti = theta * sd[i];
    if (!FILL) {
      int32_t cnt = 1;                                   // the diagonal always stays
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
        const int32_t j = ci[p];
        if (j != i && fabs(val[p].x) >= ti * sd[j]) ++cnt;
      }
      out_rp[i] = cnt;                                   // counts; scanned afterwards
      continue;
    }
    double diag = 0.0, lump = 0.0;
    int32_t q = out_rp[i], qd = -1;
    bool placed = false, stored = false;
    // (rows of a row-sharded operator list their ghost columns -- local index >= n -- in any position: only a row WITHOUT a
    // stored diagonal gets the slot inserted in front of its first larger column)
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) stored |= ci[p] == i;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      const int32_t j = ci[p];
      const double l = sign * val[p].x;
      if (j == i) { diag = l; qd = q; out_ci[q] = i; ++q; placed = true; continue; }
      if (!stored && !placed && j > i) { qd = q; out_ci[q] = i; ++q; placed = true; }     // keep columns sorted if no diagonal stored
      if (fabs(l) >= ti * sd[j]) { out_ci[q] = j; out_v[q] = l; ++q; }
      else lump += l;
    }
    if (!placed) { qd = q; out_ci[q] = i; ++q; }
    out_v[qd] = diag + lump;
  }
}
// sd_i = sqrt|l_ii| of the rows alone (row-sharded set-up: the ghost entries of sd are exchanged before the filter passes,
// which recompute the owned entries and read the ghost ones)
extern "C" int hx_sa_strength_diag(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign,
                                   double* sd, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && val && sd, "bad argument");
  k_sa_sd<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, (const zc*)val, sign, sd);
  HX_LAUNCH_CHECK("k_sa_sd");
  return HX_OK;
}
// count pass: out_rp (n + 1 entries) = row pointer of Lf, *h_nnz = its nnz; sd (n doubles) is kept for the fill pass.  SYNC.
extern "C" int hx_sa_filter_count(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign,
                                  double theta, double* sd, int32_t* out_rp, int64_t* h_nnz, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && val && sd && out_rp && h_nnz, "bad argument");
  k_sa_sd<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, (const zc*)val, sign, sd);
  HX_LAUNCH_CHECK("k_sa_sd");
  k_sa_filter<false><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, (const zc*)val, sign, theta, sd, out_rp, nullptr, nullptr);
  HX_LAUNCH_CHECK("k_sa_filter");
  HX_HIP(hipMemsetAsync(out_rp + n, 0, sizeof(int32_t), hx_s(s)));
  int rc = exclusive_scan_inplace(out_rp, (int64_t)n + 1, hx_s(s));
  if (rc) return rc;
  int32_t last = 0;
  HX_HIP(hipMemcpy(&last, out_rp + n, sizeof(int32_t), hipMemcpyDeviceToHost));
  *h_nnz = last;
  return HX_OK;
}
extern "C" int hx_sa_filter_fill(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign,
                                 double theta, const double* sd, const int32_t* out_rp, int32_t* out_ci, double* out_v,
                                 hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && val && sd && out_rp && out_ci && out_v, "bad argument");
  k_sa_filter<true><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, (const zc*)val, sign, theta, sd,
                                                          const_cast<int32_t*>(out_rp), out_ci, out_v);
  HX_LAUNCH_CHECK("k_sa_filter");
  return HX_OK;
}

// ---------------------------------------------------------------- rho(Df^-1 Lf) by power iteration
__global__ void __launch_bounds__(256)
k_dinv_spmv_norm(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const double* __restrict__ lf,
                 const double* __restrict__ v, double* __restrict__ w, double* __restrict__ partial) {
  __shared__ double red[256 / HX_WAVE];
  double acc = 0.0;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double d = 1.0, sum = 0.0;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) { if (ci[p] == i) d = lf[p]; sum = fma(lf[p], v[ci[p]], sum); }
    const double wi = sum / d;
    w[i] = wi;
    acc = fma(wi, wi, acc);
  }
#pragma unroll
  for (int o = HX_WAVE / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, HX_WAVE);
  if ((threadIdx.x & (HX_WAVE - 1)) == 0) red[threadIdx.x / HX_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double t = red[0]; for (int q = 1; q < 256 / HX_WAVE; ++q) t += red[q]; partial[blockIdx.x] = t; }
}
// v <- Df^-1 Lf v, `iters` times from the caller's start vector (v, w: n doubles, both overwritten); *h_rho = ratio of
// the last two norms (the iterates are not renormalised: 2^iters growth at most).  SYNC.
extern "C" int hx_sa_power_rho(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, double* v, double* w,
                               int iters, double* h_rho, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && lf && v && w && h_rho && iters >= 2, "bad argument");
  const int nb = hx_grid(n, 256, 1024);
  double* part = nullptr;
  HX_HIP(hx_malloc((void**)&part, sizeof(double) * 2 * (size_t)nb));
  std::vector<double> h(2 * (size_t)nb);
  double *a = v, *b = w;
  for (int it = 0; it < iters; ++it) {
    double* slot = part + (size_t)(it & 1) * nb;
    k_dinv_spmv_norm<<<nb, 256, 0, hx_s(s)>>>(n, rp, ci, lf, a, b, slot);
    std::swap(a, b);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), part, sizeof(double) * 2 * (size_t)nb, hipMemcpyDeviceToHost, hx_s(s));
  if (e == hipSuccess) e = hipStreamSynchronize(hx_s(s));
  (void)hx_free(part);
  HX_HIP(e);
  double n_last = 0.0, n_prev = 0.0;
  for (int q = 0; q < nb; ++q) { n_last += h[(size_t)((iters - 1) & 1) * nb + q]; n_prev += h[(size_t)((iters - 2) & 1) * nb + q]; }
  HX_REQUIRE(n_prev > 0.0 && n_last > 0.0, "power iteration collapsed (zero vector)");
  *h_rho = std::sqrt(n_last / n_prev);
  return HX_OK;
}

// One step w = Df^-1 Lf v of that iteration with the squared norm of w written to *d_norm2 (device): the row-sharded set-up
// exchanges the ghost entries of v between steps and sums the norms over the ranks (Lf: owned rows over [owned | ghost]).
__global__ void k_sum_partials(int nb, const double* __restrict__ part, double* __restrict__ out) {
  double t = 0.0;
  for (int q = threadIdx.x; q < nb; q += HX_WAVE) t += part[q];
#pragma unroll
  for (int o = HX_WAVE / 2; o > 0; o >>= 1) t += __shfl_down(t, o, HX_WAVE);
  if (threadIdx.x == 0) *out = t;
}
extern "C" int hx_sa_power_step(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, const double* v,
                                double* w, double* d_norm2, double* d_scratch, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && lf && v && w && d_norm2 && d_scratch, "bad argument");
  const int nb = hx_grid(n, 256, 1024);           // d_scratch: 1024 doubles
  k_dinv_spmv_norm<<<nb, 256, 0, hx_s(s)>>>(n, rp, ci, lf, v, w, d_scratch);
  HX_LAUNCH_CHECK("k_dinv_spmv_norm");
  k_sum_partials<<<1, HX_WAVE, 0, hx_s(s)>>>(nb, d_scratch, d_norm2);
  HX_LAUNCH_CHECK("k_sum_partials");
  return HX_OK;
}

// ---------------------------------------------------------------- smoothed prolongator
// P = (I - omega Df^-1 Lf) P0,  P0[i, agg(i)] = inv_sqrt_cnt[agg(i)].  Row i touches the aggregates of i and of its strong
// neighbours; the distinct aggregates are emitted in ascending order by repeated minimum selection (rows are short), the
// contributions to one aggregate are summed in entry order => deterministic.
template <bool FILL>
__global__ void k_sa_prolongator(int32_t n, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                 const double* __restrict__ lf, const int32_t* __restrict__ agg,
                                 const double* __restrict__ isc, double omega, int32_t* __restrict__ out_rp,
                                 int32_t* __restrict__ out_ci, double* __restrict__ out_v) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t b = rp[i], e = rp[i + 1], ai = agg[i];
    double dii = 1.0;
    if (FILL) { for (int32_t p = b; p < e; ++p) if (ci[p] == i) dii = lf[p]; if (dii == 0.0) dii = 1.0; }
    const double w = omega / dii;
    int32_t cur = -1, cnt = 0, q = FILL ? out_rp[i] : 0;
    while (true) {
      int32_t nxt = (ai > cur) ? ai : INT32_MAX;
      for (int32_t p = b; p < e; ++p) { const int32_t a = agg[ci[p]]; if (a > cur && a < nxt) nxt = a; }
      if (nxt == INT32_MAX) break;
      if (FILL) {
        double v = (nxt == ai) ? isc[ai] : 0.0;
        for (int32_t p = b; p < e; ++p) if (agg[ci[p]] == nxt) v -= w * lf[p] * isc[nxt];
        out_ci[q] = nxt; out_v[q] = v; ++q;
      }
      ++cnt;
      cur = nxt;
    }
    if (!FILL) out_rp[i] = cnt;
  }
}
extern "C" int hx_sa_prolongator_count(int32_t n, const int32_t* rp, const int32_t* ci, const int32_t* agg,
                                       int32_t* out_rp, int64_t* h_nnz, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && agg && out_rp && h_nnz, "bad argument");
  k_sa_prolongator<false><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, nullptr, agg, nullptr, 0.0, out_rp, nullptr, nullptr);
  HX_LAUNCH_CHECK("k_sa_prolongator");
  HX_HIP(hipMemsetAsync(out_rp + n, 0, sizeof(int32_t), hx_s(s)));
  int rc = exclusive_scan_inplace(out_rp, (int64_t)n + 1, hx_s(s));
  if (rc) return rc;
  int32_t last = 0;
  HX_HIP(hipMemcpy(&last, out_rp + n, sizeof(int32_t), hipMemcpyDeviceToHost));
  *h_nnz = last;
  return HX_OK;
}
extern "C" int hx_sa_prolongator_fill(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf,
                                      const int32_t* agg, const double* inv_sqrt_cnt, double omega, const int32_t* out_rp,
                                      int32_t* out_ci, double* out_v, hx_stream s) {
  HX_REQUIRE(n > 0 && rp && ci && lf && agg && inv_sqrt_cnt && out_rp && out_ci && out_v, "bad argument");
  k_sa_prolongator<true><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rp, ci, lf, agg, inv_sqrt_cnt, omega,
                                                               const_cast<int32_t*>(out_rp), out_ci, out_v);
  HX_LAUNCH_CHECK("k_sa_prolongator");
  return HX_OK;
}

// ---------------------------------------------------------------- transpose of a real CSR (R = P^T)
__global__ void k_expand_rows(int32_t n, const int32_t* __restrict__ rp, int32_t* __restrict__ row_of) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) row_of[p] = i;
}
__global__ void k_iota(int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}
__global__ void k_hist(int64_t nnz, const int32_t* __restrict__ ci, int32_t* __restrict__ cnt) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&cnt[ci[p]], 1);
}
__global__ void k_transpose_gather(int64_t nnz, const int32_t* __restrict__ perm, const int32_t* __restrict__ row_of,
                                   const double* __restrict__ v, int32_t* __restrict__ out_ci, double* __restrict__ out_v) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = perm[q];
    out_ci[q] = row_of[p];
    out_v[q] = v[p];
  }
}
// out = A^T for A (n_rows x n_cols, nnz entries, sorted columns): a STABLE radix sort of the column indices keeps the row
// order inside every column, so the rows of A^T come out with sorted columns.  SYNC (temporary storage).
extern "C" int hx_csr_transpose_d(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t* rp, const int32_t* ci,
                                  const double* v, int32_t* out_rp, int32_t* out_ci, double* out_v, hx_stream s) {
  HX_REQUIRE(n_rows > 0 && n_cols > 0 && nnz > 0 && nnz < (int64_t)INT32_MAX && rp && ci && v && out_rp && out_ci && out_v,
             "bad argument");
  int32_t *row_of = nullptr, *idx = nullptr, *keys_out = nullptr, *perm = nullptr;
  void* tmp = nullptr;
  size_t bytes = 0;
  int rc = HX_OK;
  hipError_t e = hx_malloc((void**)&row_of, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hx_malloc((void**)&idx, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hx_malloc((void**)&keys_out, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hx_malloc((void**)&perm, sizeof(int32_t) * (size_t)nnz);
  int bits = 1;
  while ((1ll << bits) < (int64_t)n_cols) ++bits;
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, ci, keys_out, idx, perm, (int)nnz, 0, bits, hx_s(s));
  if (e == hipSuccess) e = hx_malloc(&tmp, bytes ? bytes : 16);
  if (e == hipSuccess) {
    k_expand_rows<<<hx_grid(n_rows, 256), 256, 0, hx_s(s)>>>(n_rows, rp, row_of);
    k_iota<<<hx_grid(nnz, 256), 256, 0, hx_s(s)>>>(nnz, idx);
    e = hipMemsetAsync(out_rp, 0, sizeof(int32_t) * ((size_t)n_cols + 1), hx_s(s));
    k_hist<<<hx_grid(nnz, 256), 256, 0, hx_s(s)>>>(nnz, ci, out_rp);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, ci, keys_out, idx, perm, (int)nnz, 0, bits, hx_s(s));
    if (e == hipSuccess) {
      k_transpose_gather<<<hx_grid(nnz, 256), 256, 0, hx_s(s)>>>(nnz, perm, row_of, v, out_ci, out_v);
      e = hipGetLastError();
    }
    if (e == hipSuccess) rc = exclusive_scan_inplace(out_rp, (int64_t)n_cols + 1, hx_s(s));
    if (e == hipSuccess && rc == HX_OK) e = hipStreamSynchronize(hx_s(s));
  }
  (void)hx_free(row_of); (void)hx_free(idx); (void)hx_free(keys_out); (void)hx_free(perm); (void)hx_free(tmp);
  HX_HIP(e);
  return rc;
}

// ---------------------------------------------------------------- C = A B (sparse x sparse), Gustavson row by row
// A group of 16 lanes owns an output row.  Symbolic: the distinct columns of the row are found with an open-addressing
// hash table (keys claimed by atomicCAS; one table per group in HBM/L2 scratch, sized per row to the next power of two
// above 2 min(upper bound, columns of B)), appended to a per-row list, and the lists of all rows are ordered by ONE radix
// sort of (row, column) keys.  Numeric: every product a_ik b_kj is added at the position of j in the sorted row (binary
// search).  A group adds the contributions to one entry in the order of the A entries, so sums do not depend on timing.
#define SG 16                       // lanes per output row
struct hx_spgemm {
  int32_t nA = 0, ncolsB = 0;
  int64_t nnzC = 0;
  int32_t* rpC = nullptr;           // nA + 1
  int32_t* ciC = nullptr;           // nnzC, sorted per row
};
__global__ void k_spgemm_ub(int32_t nA, const int32_t* __restrict__ rpA, const int32_t* __restrict__ ciA,
                            const int32_t* __restrict__ rpB, int32_t ncolsB, int64_t* __restrict__ ub, int32_t* __restrict__ cap) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nA; i += gridDim.x * blockDim.x) {
    int64_t u = 0;
    for (int32_t p = rpA[i]; p < rpA[i + 1]; ++p) { const int32_t k = ciA[p]; u += rpB[k + 1] - rpB[k]; }
    if (u > ncolsB) u = ncolsB;
    ub[i] = u;
    int32_t c = 16;
    while (c < 2 * u) c <<= 1;
    cap[i] = c;
  }
}
__device__ __forceinline__ uint32_t sg_hash(int32_t c) { return (uint32_t)c * 2654435761u; }
__global__ void __launch_bounds__(256)
k_spgemm_symbolic(int32_t nA, const int32_t* __restrict__ rpA, const int32_t* __restrict__ ciA,
                  const int32_t* __restrict__ rpB, const int32_t* __restrict__ ciB, const int64_t* __restrict__ ub_off,
                  const int32_t* __restrict__ cap, int32_t cap_max, int32_t* __restrict__ tables, int32_t* __restrict__ cnt,
                  int32_t* __restrict__ lists) {
  const int lane = threadIdx.x & (SG - 1);
  const int64_t group = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / SG, n_groups = (int64_t)gridDim.x * blockDim.x / SG;
  int32_t* tab = tables + group * (int64_t)cap_max;
  for (int64_t i = group; i < nA; i += n_groups) {
    const int32_t c = cap[i];
    const uint32_t mask = (uint32_t)c - 1u;
    for (int32_t t = lane; t < c; t += SG) tab[t] = -1;
    __threadfence_block();
    int32_t* list = lists + ub_off[i];
    for (int32_t p = rpA[i]; p < rpA[i + 1]; ++p) {
      const int32_t k = ciA[p];
      for (int32_t e = rpB[k] + lane; e < rpB[k + 1]; e += SG) {
        const int32_t col = ciB[e];
        uint32_t h = sg_hash(col) & mask;
        while (true) {
          const int32_t old = atomicCAS(&tab[h], -1, col);
          if (old == -1) { list[atomicAdd(&cnt[i], 1)] = col; break; }
          if (old == col) break;
          h = (h + 1) & mask;
        }
      }
    }
    __threadfence_block();          // the table is cleared for the next row only after every lane left this one
  }
}
__global__ void k_spgemm_keys(int32_t nA, const int32_t* __restrict__ rpC, const int64_t* __restrict__ ub_off,
                              const int32_t* __restrict__ lists, uint64_t* __restrict__ keys) {
  const int lane = threadIdx.x & (SG - 1);
  const int64_t group = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / SG, n_groups = (int64_t)gridDim.x * blockDim.x / SG;
  for (int64_t i = group; i < nA; i += n_groups) {
    const int32_t b = rpC[i], len = rpC[i + 1] - b;
    const int32_t* list = lists + ub_off[i];
    for (int32_t t = lane; t < len; t += SG) keys[b + t] = ((uint64_t)i << 32) | (uint32_t)list[t];
  }
}
__global__ void k_spgemm_cols(int64_t nnz, const uint64_t* __restrict__ keys, int32_t* __restrict__ ci) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
    ci[q] = (int32_t)(keys[q] & 0xffffffffu);
}
// SYNC.  The plan keeps the pattern of C; out_rp (nA + 1, device) receives its row pointer, *h_nnz its size.
extern "C" int hx_spgemm_symbolic(hx_spgemm** out, int32_t nA, const int32_t* rpA, const int32_t* ciA, int32_t ncolsB,
                                  const int32_t* rpB, const int32_t* ciB, int32_t* out_rp, int64_t* h_nnz, hx_stream s) {
  HX_REQUIRE(out && nA > 0 && ncolsB > 0 && rpA && ciA && rpB && ciB && out_rp && h_nnz, "bad argument");
  int64_t* ub = nullptr;
  int32_t *cap = nullptr, *tables = nullptr, *lists = nullptr, *cnt = nullptr;
  uint64_t *keys = nullptr, *keys2 = nullptr;
  void* tmp = nullptr;
  hx_spgemm* P = new hx_spgemm();
  P->nA = nA; P->ncolsB = ncolsB;
  int rc = HX_OK;
  hipError_t e = hipSuccess;
  auto cleanup = [&]() {
    (void)hx_free(ub); (void)hx_free(cap); (void)hx_free(tables); (void)hx_free(lists); (void)hx_free(cnt);
    (void)hx_free(keys); (void)hx_free(keys2); (void)hx_free(tmp);
  };
  auto fail = [&](int code, const char* what, const char* detail) {
    cleanup();
    (void)hx_free(P->rpC); (void)hx_free(P->ciC);
    delete P;
    return hx_fail(code, what, detail);
  };
#define SP_HIP(call) do { e = (call); if (e != hipSuccess) return fail(HX_ERR_HIP, #call, hipGetErrorString(e)); } while (0)
  SP_HIP(hx_malloc((void**)&ub, sizeof(int64_t) * ((size_t)nA + 1)));
  SP_HIP(hx_malloc((void**)&cap, sizeof(int32_t) * (size_t)nA));
  SP_HIP(hx_malloc((void**)&cnt, sizeof(int32_t) * ((size_t)nA + 1)));
  SP_HIP(hx_malloc((void**)&P->rpC, sizeof(int32_t) * ((size_t)nA + 1)));
  k_spgemm_ub<<<hx_grid(nA, 256), 256, 0, hx_s(s)>>>(nA, rpA, ciA, rpB, ncolsB, ub, cap);
  SP_HIP(hipGetLastError());
  // largest table, total list length
  int32_t cap_max = 16;
  {
    size_t bytes = 0;
    int32_t* d_max = nullptr;
    SP_HIP(hx_malloc((void**)&d_max, sizeof(int32_t)));
    e = hipcub::DeviceReduce::Max(nullptr, bytes, cap, d_max, nA, hx_s(s));
    void* t2 = nullptr;
    if (e == hipSuccess) e = hx_malloc(&t2, bytes ? bytes : 16);
    if (e == hipSuccess) e = hipcub::DeviceReduce::Max(t2, bytes, cap, d_max, nA, hx_s(s));
    if (e == hipSuccess) e = hipMemcpyAsync(&cap_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, hx_s(s));
    if (e == hipSuccess) e = hipStreamSynchronize(hx_s(s));
    (void)hx_free(t2); (void)hx_free(d_max);
    if (e != hipSuccess) return fail(HX_ERR_HIP, "hipcub::DeviceReduce::Max", hipGetErrorString(e));
  }
  SP_HIP(hipMemsetAsync(ub + nA, 0, sizeof(int64_t), hx_s(s)));
  if ((rc = exclusive_scan_inplace(ub, (int64_t)nA + 1, hx_s(s)))) { cleanup(); (void)hx_free(P->rpC); delete P; return rc; }
  int64_t total_ub = 0;
  SP_HIP(hipMemcpy(&total_ub, ub + nA, sizeof(int64_t), hipMemcpyDeviceToHost));
  // scratch tables: as many groups as fit 1 GiB, at most 8 groups per wave slot of the chip
  int64_t n_groups = (int64_t)(1ll << 30) / ((int64_t)cap_max * 4);
  n_groups = std::max<int64_t>(16, std::min<int64_t>(n_groups, 256 * 8 * (256 / SG) / 2));
  n_groups = std::min<int64_t>(n_groups, ((int64_t)nA + 0));
  const int blocks = (int)std::max<int64_t>(1, (n_groups * SG + 255) / 256);
  n_groups = (int64_t)blocks * 256 / SG;
  SP_HIP(hx_malloc((void**)&tables, sizeof(int32_t) * (size_t)n_groups * (size_t)cap_max));
  SP_HIP(hx_malloc((void**)&lists, sizeof(int32_t) * (size_t)std::max<int64_t>(total_ub, 1)));
  SP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)nA + 1), hx_s(s)));
  k_spgemm_symbolic<<<blocks, 256, 0, hx_s(s)>>>(nA, rpA, ciA, rpB, ciB, ub, cap, cap_max, tables, cnt, lists);
  SP_HIP(hipGetLastError());
  SP_HIP(hipMemcpyAsync(P->rpC, cnt, sizeof(int32_t) * ((size_t)nA + 1), hipMemcpyDeviceToDevice, hx_s(s)));
  if ((rc = exclusive_scan_inplace(P->rpC, (int64_t)nA + 1, hx_s(s)))) { cleanup(); (void)hx_free(P->rpC); delete P; return rc; }
  int32_t nnz32 = 0;
  SP_HIP(hipMemcpy(&nnz32, P->rpC + nA, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (nnz32 <= 0) return fail(HX_ERR_ARG, "hx_spgemm_symbolic", "empty product (or more than 2^31 entries)");
  P->nnzC = nnz32;
  SP_HIP(hx_malloc((void**)&keys, sizeof(uint64_t) * (size_t)nnz32));
  SP_HIP(hx_malloc((void**)&keys2, sizeof(uint64_t) * (size_t)nnz32));
  SP_HIP(hx_malloc((void**)&P->ciC, sizeof(int32_t) * (size_t)nnz32));
  k_spgemm_keys<<<blocks, 256, 0, hx_s(s)>>>(nA, P->rpC, ub, lists, keys);
  SP_HIP(hipGetLastError());
  int row_bits = 1;
  while ((1ll << row_bits) < (int64_t)nA) ++row_bits;
  size_t bytes = 0;
  SP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, keys, keys2, (int)nnz32, 0, 32 + row_bits, hx_s(s)));
  SP_HIP(hx_malloc(&tmp, bytes ? bytes : 16));
  SP_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, bytes, keys, keys2, (int)nnz32, 0, 32 + row_bits, hx_s(s)));
  k_spgemm_cols<<<hx_grid(nnz32, 256), 256, 0, hx_s(s)>>>(nnz32, keys2, P->ciC);
  SP_HIP(hipGetLastError());
  SP_HIP(hipMemcpyAsync(out_rp, P->rpC, sizeof(int32_t) * ((size_t)nA + 1), hipMemcpyDeviceToDevice, hx_s(s)));
  SP_HIP(hipStreamSynchronize(hx_s(s)));
#undef SP_HIP
  cleanup();
  *h_nnz = P->nnzC;
  *out = P;
  return HX_OK;
}

template <bool AZ, bool BZ>
__global__ void __launch_bounds__(256)
k_spgemm_numeric(int32_t nA, const int32_t* __restrict__ rpA, const int32_t* __restrict__ ciA, const double* __restrict__ va,
                 const int32_t* __restrict__ rpB, const int32_t* __restrict__ ciB, const double* __restrict__ vb,
                 const int32_t* __restrict__ rpC, const int32_t* __restrict__ ciC, zc* __restrict__ vc) {
  const int lane = threadIdx.x & (SG - 1);
  const int64_t group = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / SG, n_groups = (int64_t)gridDim.x * blockDim.x / SG;
  for (int64_t i = group; i < nA; i += n_groups) {
    const int32_t cb = rpC[i], clen = rpC[i + 1] - cb;
    const int32_t* crow = ciC + cb;
    for (int32_t p = rpA[i]; p < rpA[i + 1]; ++p) {
      const int32_t k = ciA[p];
      const zc a = AZ ? ((const zc*)va)[p] : make_double2(va[p], 0.0);
      for (int32_t e = rpB[k] + lane; e < rpB[k + 1]; e += SG) {
        const int32_t col = ciB[e];
        const zc b = BZ ? ((const zc*)vb)[e] : make_double2(vb[e], 0.0);
        int32_t lo = 0, hi = clen - 1;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (crow[mid] < col) lo = mid + 1; else hi = mid; }
        const zc prod = zmul(a, b);
        double* dst = (double*)(vc + cb + lo);
        unsafeAtomicAdd(dst, prod.x);          // one group per row, A entries in order: the adds to one entry are issued in order
        unsafeAtomicAdd(dst + 1, prod.y);
      }
    }
  }
}
// values of C = A B on the pattern of the plan; a_complex / b_complex: value arrays are complex128 (else float64).
extern "C" int hx_spgemm_numeric(const hx_spgemm* P, const int32_t* rpA, const int32_t* ciA, const void* va, int a_complex,
                                 const int32_t* rpB, const int32_t* ciB, const void* vb, int b_complex, int32_t* out_ci,
                                 hx_c128* out_v, hx_stream s) {
  HX_REQUIRE(P && rpA && ciA && va && rpB && ciB && vb && out_ci && out_v, "bad argument");
  HX_HIP(hipMemcpyAsync(out_ci, P->ciC, sizeof(int32_t) * (size_t)P->nnzC, hipMemcpyDeviceToDevice, hx_s(s)));
  HX_HIP(hipMemsetAsync(out_v, 0, sizeof(hx_c128) * (size_t)P->nnzC, hx_s(s)));
  const int blocks = (int)std::min<int64_t>(((int64_t)P->nA * SG + 255) / 256, 256 * 32);
  const double *a = (const double*)va, *b = (const double*)vb;
  zc* c = (zc*)out_v;
  if (a_complex && b_complex) k_spgemm_numeric<true, true><<<blocks, 256, 0, hx_s(s)>>>(P->nA, rpA, ciA, a, rpB, ciB, b, P->rpC, P->ciC, c);
  else if (a_complex) k_spgemm_numeric<true, false><<<blocks, 256, 0, hx_s(s)>>>(P->nA, rpA, ciA, a, rpB, ciB, b, P->rpC, P->ciC, c);
  else if (b_complex) k_spgemm_numeric<false, true><<<blocks, 256, 0, hx_s(s)>>>(P->nA, rpA, ciA, a, rpB, ciB, b, P->rpC, P->ciC, c);
  else k_spgemm_numeric<false, false><<<blocks, 256, 0, hx_s(s)>>>(P->nA, rpA, ciA, a, rpB, ciB, b, P->rpC, P->ciC, c);
  HX_LAUNCH_CHECK("k_spgemm_numeric");
  return HX_OK;
}
extern "C" int hx_spgemm_destroy(hx_spgemm* P) {
  if (!P) return HX_OK;
  (void)hx_free(P->rpC); (void)hx_free(P->ciC);
  delete P;
  return HX_OK;
}
