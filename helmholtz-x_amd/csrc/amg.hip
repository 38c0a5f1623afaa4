// Algebraic-multigrid V-cycle + right-preconditioned GMRES for the shifted operator K (complex128).
//
// Used as the ST "inner solve" of shift-invert Arnoldi when a sparse direct factorisation of K does not fit
// (3-D, >= 1e5 DOF; the reference would call MUMPS/SuperLU_DIST behind PCLU, eigensolvers.py:49-50,102-103).
// The hierarchy (aggregates, smoothed prolongators, Galerkin operators) is set up once per K by the host side
// (amg.py + hx_host_aggregate below); everything inside a solve runs on the device and is SpMV-shaped:
// fused Jacobi sweeps, residuals, P / R products, CGS2 panels -- i.e. HBM-bound streaming kernels.
#include <cmath>
#include <complex>
#include <vector>

#include "hx_common.h"

extern "C" int hx_csr_jacobi_z(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*,
                               const hx_c128*, const hx_c128*, hx_c128*, double, hx_stream);
extern "C" int hx_zdiag_scale(int64_t, double, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);

struct Csr { int32_t rows = 0; int64_t nnz = 0; const int32_t* rp = nullptr; const int32_t* ci = nullptr; const hx_c128* va = nullptr; };

struct AmgLevel {
  int32_t n = 0;
  Csr K, P, R;                 // P: n x n_coarse, R = P^T: n_coarse x n   (absent on the coarsest level)
  const hx_c128* dinv = nullptr;
  hx_c128 *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;   // owned work vectors (b, x unused on level 0)
};

struct hx_amg {
  std::vector<AmgLevel> L;
  Csr coarse_inv;              // dense inverse of the coarsest operator, stored as CSR
  int pre = 2, post = 2;
  double omega = 0.6;
};

static const double ONE[2] = {1.0, 0.0}, ZERO[2] = {0.0, 0.0}, MONE[2] = {-1.0, 0.0};

int hx_csr_spmv_transfer(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, hx_c128*,
                         const double*, const double*, hx_stream);

static int spmv(const Csr& A, const hx_c128* x, hx_c128* y, const double* a, const double* b, hx_stream s) {
  return hx_csr_spmv_z(A.rows, A.nnz, A.rp, A.ci, A.va, x, y, a, b, 0, s);
}
static int spmv_t(const Csr& A, const hx_c128* x, hx_c128* y, const double* a, const double* b, hx_stream s) {
  return hx_csr_spmv_transfer(A.rows, A.nnz, A.rp, A.ci, A.va, x, y, a, b, s);
}

extern "C" int hx_amg_create(hx_amg** out, int n_levels, int pre, int post, double omega) {
  HX_REQUIRE(out && n_levels >= 1 && pre >= 1 && post >= 0 && omega > 0, "bad argument");
  hx_amg* h = new hx_amg();
  h->L.resize(n_levels);
  h->pre = pre; h->post = post; h->omega = omega;
  *out = h;
  return HX_OK;
}

extern "C" int hx_amg_destroy(hx_amg* h) {
  if (!h) return HX_OK;
  for (auto& l : h->L) { (void)hipFree(l.x); (void)hipFree(l.x2); (void)hipFree(l.b); (void)hipFree(l.r); }
  delete h;
  return HX_OK;
}

extern "C" int hx_amg_set_level(hx_amg* h, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK,
                                const hx_c128* vaK, const hx_c128* dinv, int32_t n_coarse, int64_t nnzP,
                                const int32_t* rpP, const int32_t* ciP, const hx_c128* vaP, int64_t nnzR,
                                const int32_t* rpR, const int32_t* ciR, const hx_c128* vaR) {
  HX_REQUIRE(h && l >= 0 && l < (int)h->L.size() && n > 0 && rpK && ciK && vaK, "bad argument");
  AmgLevel& L = h->L[l];
  L.n = n;
  L.K = {n, nnzK, rpK, ciK, vaK};
  L.dinv = dinv;
  const bool coarsest = (l == (int)h->L.size() - 1);
  if (!coarsest) {
    HX_REQUIRE(dinv && n_coarse > 0 && rpP && ciP && vaP && rpR && ciR && vaR, "missing transfer operators");
    L.P = {n, nnzP, rpP, ciP, vaP};
    L.R = {n_coarse, nnzR, rpR, ciR, vaR};
  }
  const size_t bytes = sizeof(hx_c128) * (size_t)n;
  HX_HIP(hipMalloc((void**)&L.x, bytes));
  HX_HIP(hipMalloc((void**)&L.x2, bytes));
  HX_HIP(hipMalloc((void**)&L.b, bytes));
  HX_HIP(hipMalloc((void**)&L.r, bytes));
  return HX_OK;
}

extern "C" int hx_amg_set_coarse(hx_amg* h, int32_t n_c, int64_t nnz, const int32_t* rp, const int32_t* ci,
                                 const hx_c128* va) {
  HX_REQUIRE(h && n_c > 0 && rp && ci && va && h->L.back().n == n_c, "bad argument");
  h->coarse_inv = {n_c, nnz, rp, ci, va};
  return HX_OK;
}

// x = V-cycle(b) on level l with zero initial guess.  Result is left in *xres (one of the level's buffers or xout).
static int vcycle(hx_amg* h, int l, const hx_c128* b, hx_c128* xout, hx_stream s) {
  AmgLevel& L = h->L[l];
  int rc;
  if (l == (int)h->L.size() - 1) return spmv_t(h->coarse_inv, b, xout, ONE, ZERO, s);
  hx_c128* cur = L.x;      // ping-pong between L.x and L.x2; final result copied/written to xout
  hx_c128* nxt = L.x2;
  // pre-smoothing: first sweep from x = 0 is a diagonal scaling
  if ((rc = hx_zdiag_scale(L.n, h->omega, L.dinv, b, cur, s))) return rc;
  for (int k = 1; k < h->pre; ++k) {
    if ((rc = hx_csr_jacobi_z(L.n, L.K.nnz, L.K.rp, L.K.ci, L.K.va, L.dinv, b, cur, nxt, h->omega, s))) return rc;
    std::swap(cur, nxt);
  }
  // residual r = b - K x, restrict
  HX_HIP(hipMemcpyAsync(L.r, b, sizeof(hx_c128) * (size_t)L.n, hipMemcpyDeviceToDevice, hx_s(s)));
  if ((rc = spmv(L.K, cur, L.r, MONE, ONE, s))) return rc;
  AmgLevel& Lc = h->L[l + 1];
  if ((rc = spmv_t(L.R, L.r, Lc.b, ONE, ZERO, s))) return rc;
  // coarse correction (result in Lc.r is free to use as the output buffer of the coarse level)
  hx_c128* ec = Lc.r;
  if ((rc = vcycle(h, l + 1, Lc.b, ec, s))) return rc;
  if ((rc = spmv_t(L.P, ec, cur, ONE, ONE, s))) return rc;    // x += P e_c
  // post-smoothing: every sweep reads `cur` and writes the other buffer; the last one writes xout
  for (int k = 0; k < h->post; ++k) {
    hx_c128* dst = (k == h->post - 1) ? xout : (cur == L.x ? L.x2 : L.x);
    if ((rc = hx_csr_jacobi_z(L.n, L.K.nnz, L.K.rp, L.K.ci, L.K.va, L.dinv, b, cur, dst, h->omega, s))) return rc;
    cur = dst;
  }
  if (h->post == 0)
    HX_HIP(hipMemcpyAsync(xout, cur, sizeof(hx_c128) * (size_t)L.n, hipMemcpyDeviceToDevice, hx_s(s)));
  return HX_OK;
}

extern "C" int hx_amg_vcycle(hx_amg* h, const hx_c128* b, hx_c128* x, hx_stream s) {
  HX_REQUIRE(h && b && x && b != x && !h->L.empty() && h->L[0].n > 0, "bad argument");
  HX_REQUIRE(h->coarse_inv.rp, "coarse solver not set");
  return vcycle(h, 0, b, x, s);
}

// ------------------------------------------------------------------------------------------------ GMRES
// Right-preconditioned restarted GMRES, CGS2 orthogonalisation with the fused panel kernels.
// workspace: (restart + 4) * n complex numbers + hx_panel_dot_ws(restart + 1) + (restart + 2).
extern "C" size_t hx_gmres_ws(int64_t n, int restart) {
  return (size_t)(restart + 4) * (size_t)n + hx_panel_dot_ws(restart + 1) + (size_t)restart + 2;
}

typedef std::complex<double> cd;

extern "C" int hx_gmres(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                        const hx_c128* val, const hx_c128* b, hx_c128* x, double rtol, int max_it, int restart,
                        hx_c128* ws, int* h_iters, double* h_relres, hx_stream s) {
  HX_REQUIRE(n > 0 && rowptr && col && val && b && x && ws && restart >= 1 && restart <= 120 && max_it >= 1, "bad argument");
  const Csr K{n, nnz, rowptr, col, val};
  hx_c128* V = ws;                                   // (restart+1) x n, row j = v_j
  hx_c128* z = V + (size_t)(restart + 1) * n;
  hx_c128* u = z + n;
  hx_c128* r = u + n;
  hx_c128* part = r + n;
  hx_c128* hdev = part + hx_panel_dot_ws(restart + 1);
  std::vector<cd> hh(restart + 2);
  std::vector<std::vector<cd>> H(restart + 1, std::vector<cd>(restart, cd(0)));
  std::vector<cd> cs(restart), sn(restart), g(restart + 1);
  auto dots = [&](int k, const hx_c128* w) -> int {
    int rc = hx_zpanel_dot(n, k, V, n, w, hdev, part, s);
    if (rc) return rc;
    HX_HIP(hipMemcpyAsync(hh.data(), hdev, sizeof(cd) * (size_t)(k + 1), hipMemcpyDeviceToHost, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    return HX_OK;
  };
  int rc, it = 0;
  HX_HIP(hipMemsetAsync(x, 0, sizeof(hx_c128) * (size_t)n, hx_s(s)));
  if ((rc = dots(0, b))) return rc;
  const double bnorm = std::sqrt(hh[0].real());
  if (bnorm == 0.0) { if (h_iters) *h_iters = 0; if (h_relres) *h_relres = 0.0; return HX_OK; }
  double relres = 1.0;
  bool first = true;
  while (it < max_it) {
    // r = b - K x
    HX_HIP(hipMemcpyAsync(r, b, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
    if (!first) { if ((rc = spmv(K, x, r, MONE, ONE, s))) return rc; }
    first = false;
    if ((rc = dots(0, r))) return rc;
    const double beta = std::sqrt(hh[0].real());
    relres = beta / bnorm;
    if (relres <= rtol) break;
    HX_HIP(hipMemcpyAsync(V, r, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
    const double ib[2] = {1.0 / beta, 0.0};
    if ((rc = hx_zscal(n, ib, V, s))) return rc;
    std::fill(g.begin(), g.end(), cd(0));
    g[0] = beta;
    int j = 0;
    for (; j < restart && it < max_it; ++j, ++it) {
      hx_c128* vj = V + (size_t)j * n;
      hx_c128* w = V + (size_t)(j + 1) * n;
      if (M) { if ((rc = hx_amg_vcycle(M, vj, z, s))) return rc; if ((rc = spmv(K, z, w, ONE, ZERO, s))) return rc; }
      else { if ((rc = spmv(K, vj, w, ONE, ZERO, s))) return rc; }
      const int k = j + 1;
      // classical Gram-Schmidt, always twice ("twice is enough"): one pass loses orthogonality like eps*kappa^2 of
      // the Krylov matrix, which was measured to stall GMRES here (55 instead of 29 iterations per solve).
      if ((rc = dots(k, w))) return rc;
      if ((rc = hx_zpanel_sub(n, k, V, n, hdev, w, s))) return rc;
      for (int i = 0; i < k; ++i) H[i][j] = hh[i];
      if ((rc = dots(k, w))) return rc;
      if ((rc = hx_zpanel_sub(n, k, V, n, hdev, w, s))) return rc;
      double n2 = hh[k].real();
      for (int i = 0; i < k; ++i) { n2 -= std::norm(hh[i]); H[i][j] += hh[i]; }
      if (n2 <= 1e-8 * hh[k].real()) { if ((rc = dots(0, w))) return rc; n2 = hh[0].real(); }
      const double hn = std::sqrt(n2 > 0 ? n2 : 0.0);
      H[k][j] = hn;
      if (hn > 0) { const double ih[2] = {1.0 / hn, 0.0}; if ((rc = hx_zscal(n, ih, w, s))) return rc; }
      // Givens rotations (complex)
      for (int i = 0; i < j; ++i) {
        const cd t = std::conj(cs[i]) * H[i][j] + std::conj(sn[i]) * H[i + 1][j];
        H[i + 1][j] = -sn[i] * H[i][j] + cs[i] * H[i + 1][j];
        H[i][j] = t;
      }
      const double denom = std::sqrt(std::norm(H[j][j]) + std::norm(H[j + 1][j]));
      if (denom == 0.0) return hx_fail(HX_ERR_BREAKDOWN, __func__, "zero Hessenberg column");
      cs[j] = H[j][j] / denom;            // complex "cosine"
      sn[j] = H[j + 1][j] / denom;
      H[j][j] = std::conj(cs[j]) * H[j][j] + std::conj(sn[j]) * H[j + 1][j];
      H[j + 1][j] = 0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = std::conj(cs[j]) * g[j];
      relres = std::abs(g[j + 1]) / bnorm;
      if (relres <= rtol || hn == 0.0) { ++j; ++it; break; }
    }
    // y = H^-1 g ; x += M^-1 (V y)
    std::vector<cd> y(j);
    for (int i = j - 1; i >= 0; --i) {
      cd t = g[i];
      for (int q = i + 1; q < j; ++q) t -= H[i][q] * y[q];
      y[i] = t / H[i][i];
    }
    for (int i = 0; i < j; ++i) y[i] = -y[i];          // panel_sub computes w -= V h
    HX_HIP(hipMemcpyAsync(hdev, y.data(), sizeof(cd) * (size_t)j, hipMemcpyHostToDevice, hx_s(s)));
    HX_HIP(hipMemsetAsync(u, 0, sizeof(hx_c128) * (size_t)n, hx_s(s)));
    if ((rc = hx_zpanel_sub(n, j, V, n, hdev, u, s))) return rc;
    HX_HIP(hipStreamSynchronize(hx_s(s)));             // y (host vector) must outlive the copy
    if (M) { if ((rc = hx_amg_vcycle(M, u, z, s))) return rc; if ((rc = hx_zaxpy(n, ONE, z, x, s))) return rc; }
    else { if ((rc = hx_zaxpy(n, ONE, u, x, s))) return rc; }
    if (relres <= rtol) {
      // confirm with the true residual on the next loop entry only if more iterations are allowed
      HX_HIP(hipMemcpyAsync(r, b, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      if ((rc = spmv(K, x, r, MONE, ONE, s))) return rc;
      if ((rc = dots(0, r))) return rc;
      relres = std::sqrt(hh[0].real()) / bnorm;
      if (relres <= 10 * rtol) break;
    }
  }
  if (h_iters) *h_iters = it;
  if (h_relres) *h_relres = relres;
  return HX_OK;
}

// ------------------------------------------------------------------------------------------------ aggregation (host)
// Greedy (Vanek) aggregation on the strength graph: n nodes, CSR (rp, ci) of strong off-diagonal couplings with
// weights w (|a_ij|).  Writes agg[i] in [0, n_agg).  Pure integer/graph work, one-off per hierarchy level.
extern "C" int hx_host_aggregate(int32_t n, const int32_t* rp, const int32_t* ci, const double* w, int32_t* agg,
                                 int32_t* n_agg) {
  HX_REQUIRE(n > 0 && rp && agg && n_agg, "bad argument");
  for (int32_t i = 0; i < n; ++i) agg[i] = -1;
  int32_t na = 0;
  // pass 1: roots whose whole strong neighbourhood is free
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1 || rp[i + 1] == rp[i]) continue;
    bool free_nb = true;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (agg[ci[p]] != -1) { free_nb = false; break; }
    if (!free_nb) continue;
    agg[i] = na;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) agg[ci[p]] = na;
    ++na;
  }
  // pass 2: attach leftovers to the most strongly connected pass-1 aggregate
  std::vector<int32_t> pass1(agg, agg + n);
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    double best = -1.0; int32_t tgt = -1;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      const int32_t a = pass1[ci[p]];
      if (a >= 0 && w[p] > best) { best = w[p]; tgt = a; }
    }
    if (tgt >= 0) agg[i] = tgt;
  }
  // pass 3: whatever is left forms aggregates with its free neighbours (isolated nodes become singletons)
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    agg[i] = na;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (agg[ci[p]] == -1) agg[ci[p]] = na;
    ++na;
  }
  *n_agg = na;
  return HX_OK;
}
