// Algebraic-multigrid V-cycle + flexible right-preconditioned GMRES for the shifted operator K (complex128).
//
// Used as the ST "inner solve" of shift-invert Arnoldi when a sparse direct factorisation of K does not pay
// (the reference calls MUMPS/SuperLU_DIST behind PCLU there, eigensolvers.py:49-50,102-103).
// The hierarchy (aggregates, smoothed prolongators, Galerkin operators) is set up once per K by the host side
// (amg.py + hx_host_aggregate below); everything inside a solve runs on the device and is SpMV-shaped:
// fused Jacobi sweeps, residuals, P / R products, CGS2 panels -- i.e. HBM-bound streaming kernels.
//
// Precision: the Krylov iteration, its operator products, dot products and the solution update are complex128.
// The V-cycle -- a preconditioner, i.e. only an approximation of K^-1 by construction -- may run on a complex64
// copy of the hierarchy, which halves its HBM traffic; GMRES is therefore the FLEXIBLE variant (it stores
// z_j = M^-1 v_j), which converges to the full double-precision residual whatever the preconditioner does.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <complex>
#include <thread>
#include <vector>

#include "amg_internal.h"

extern "C" int hx_csr_jacobi_z(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*,
                               const hx_c128*, const hx_c128*, hx_c128*, double, hx_stream);
extern "C" int hx_zdiag_scale(int64_t, double, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_csr_spmv_transfer(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, hx_c128*,
                         const double*, const double*, hx_stream);
int hx_csr_spmv_creal(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, float*, float, float, hx_stream);
int hx_csr_spmv_zreal(int32_t, int64_t, const int32_t*, const int32_t*, const double*, const hx_c128*, hx_c128*, double, double, hx_stream);
int hx_csr_spmv_c(int role, int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, float*, float,
                  float, hx_stream);
int hx_csr_smooth_rows_c(int32_t, int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*,
                         float*, float, const float*, float, hx_stream);
int hx_csr_smooth_rows_z(int32_t, int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, const hx_c128*,
                         const hx_c128*, hx_c128*, double, const hx_c128*, double, hx_stream);
int hx_csr_restrict_diag_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, float*, const float*, float*, float,
                           hx_stream);
int hx_csr_restrict_diag_z(int32_t, int64_t, const int32_t*, const int32_t*, const double*, const hx_c128*, hx_c128*, const hx_c128*,
                           hx_c128*, double, hx_stream);
int hx_csr_jacobi_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*,
                    const float*, float*, float, hx_stream);
int hx_cdiag_scale(int64_t, float, const float*, const float*, float*, hx_stream);
int hx_csr_smooth_z(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, const hx_c128*,
                    const hx_c128*, hx_c128*, double, const hx_c128*, double, hx_stream);
int hx_csr_smooth_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*,
                    const float*, float*, float, const float*, float, hx_stream);
extern "C" int hx_zpanel_dot_c64(int64_t, int, const hx_c64*, int64_t, const hx_c128*, hx_c128*, hx_c128*, hx_stream);
extern "C" int hx_zpanel_sub_c64(int64_t, int, const hx_c64*, int64_t, const hx_c128*, hx_c128*, hx_stream);
struct hx_sell;
extern "C" int hx_sell_spmv_z(const hx_sell*, const hx_c128*, hx_c128*, const double*, const double*, hx_stream);
extern "C" int hx_sell_spmv_c64(const hx_sell*, const hx_c64*, hx_c64*, float, float, hx_stream);
extern "C" int hx_sell_smooth_z(const hx_sell*, const hx_c128*, const hx_c128*, const hx_c128*, hx_c128*, double,
                                const hx_c128*, double, hx_stream);
extern "C" int hx_sell_smooth_c64(const hx_sell*, const hx_c64*, const hx_c64*, const hx_c64*, hx_c64*, float,
                                  const hx_c64*, float, hx_stream);
int hx_sell_residual_c64(const hx_sell*, const hx_c64*, const hx_c64*, hx_c64*, hx_stream);
int hx_sell_residual_z(const hx_sell*, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_sell_has_parts(const hx_sell*);
int hx_sell_spmv_z_part(const hx_sell*, int, const hx_c128*, hx_c128*, const double*, const double*, hx_stream);
int hx_sell_residual_c64_part(const hx_sell*, int, const hx_c64*, const hx_c64*, hx_c64*, hx_stream);
int hx_sell_residual_z_part(const hx_sell*, int, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_sell_smooth_c64_part(const hx_sell*, int, const hx_c64*, const hx_c64*, const hx_c64*, hx_c64*, float, const hx_c64*, float, hx_stream);
int hx_sell_smooth_z_part(const hx_sell*, int, const hx_c128*, const hx_c128*, const hx_c128*, hx_c128*, double, const hx_c128*, double, hx_stream);
int hx_csr_residual_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*, float*, hx_stream);
int hx_csr_residual_z(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_cast_z2c(int64_t, const hx_c128*, float*, hx_stream);
int hx_csr_norm_inf(int32_t, const int32_t*, const hx_c128*, void*, double*, hx_stream);
int hx_cast_c2z(int64_t, const float*, hx_c128*, hx_stream);

// block kernels (block.hip)
extern "C" int hx_block_width(void);
int hx_sell_spmm_c64(const hx_sell*, const hx_c64*, hx_c64*, float, float, const hx_c64*, hx_stream);
int hx_sell_smooth_blk_c64(const hx_sell*, const hx_c64*, const hx_c64*, const hx_c64*, hx_c64*, float, const hx_c64*, float, hx_stream);
int hx_csr_spmm_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, bool, const float*, float*, float, float, const float*, hx_stream);
int hx_csr_smooth_blk_c(int32_t, int64_t, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*, float*, float,
                        const float*, float, hx_stream);
int hx_cdiag_scale_blk(int64_t, float, const float*, const float*, float*, hx_stream);

static const double ONE[2] = {1.0, 0.0}, ZERO[2] = {0.0, 0.0}, MONE[2] = {-1.0, 0.0};

static void drop_graph(hx_amg* h) {
  if (h->vexec) { (void)hipGraphExecDestroy(h->vexec); h->vexec = nullptr; }
  if (h->graph_state == 1) h->graph_state = 0;          // re-capture with the new set-up
}
// block work vectors are sized from the levels' row counts when the first block solve asks for them: a level that is set again
// (hx_amg_set_level*, hx_amg_set_coarse*, hx_amg_set_sharded) invalidates them -- the next block solve allocates afresh
static void drop_blk(hx_amg* h) {
  for (auto& l : h->blk) { (void)hx_free(l.x); (void)hx_free(l.x2); (void)hx_free(l.x3); (void)hx_free(l.b); (void)hx_free(l.r); }
  h->blk.clear();
}
// ghost block of a sharded level's iterate must be current before K (or a transfer operator) is applied to it
static int halo_level(hx_amg* h, int l, void* v, int elem_bytes, hx_stream s) {
  if (!h->sharded || l >= h->n_sharded) return HX_OK;
  return h->halo(h->ctx, l, v, elem_bytes, s) == 0 ? HX_OK : hx_fail(HX_ERR_COMM, __func__, "halo exchange callback failed");
}

// A product with the level's operator whose operand needs its ghost block: exchange first, then the whole product -- or, when
// the split callbacks are set and the level's SELL-64 operator carries slice lists, start the exchange (on s), run the INTERIOR
// slices (they read no ghost column) on the side stream the exchange hands out while it travels, join, run the boundary slices
// on s.  apply(part, stream): part 0 all, 1 interior, 2 boundary.
template <typename LT, typename F>
static int with_halo(hx_amg* h, const LT& L, int l, void* v, int elem_bytes, hx_stream s, F&& apply) {
  if (!h->sharded || l >= h->n_sharded) return apply(0, s);
  int rc;
  if (h->halo_begin && h->halo_end && L.sell && hx_sell_has_parts(L.sell)) {
    hx_stream side = nullptr;
    if (h->halo_begin(h->ctx, l, v, elem_bytes, s, &side) != 0) return hx_fail(HX_ERR_COMM, __func__, "halo exchange (begin) failed");
    rc = apply(1, side);
    if (h->halo_end(h->ctx, l, s) != 0) return hx_fail(HX_ERR_COMM, __func__, "halo exchange (end) failed");   // (joins the side stream also after an error)
    if (rc) return rc;
    return apply(2, s);
  }
  if ((rc = halo_level(h, l, v, elem_bytes, s))) return rc;
  return apply(0, s);
}

// ---- per-precision kernel dispatch
static int op_kmult(const LevelT<double>& L, const double* x, double* y, double a, double b, hx_stream s);
static int op_kmult(const LevelT<float>& L, const float* x, float* y, float a, float b, hx_stream s);
static int op_spmv(const CsrT<double>& A, const double* x, double* y, double a, double b, bool transfer, hx_stream s) {
  if (A.real_vals) return hx_csr_spmv_zreal(A.rows, A.nnz, A.rp, A.ci, A.va, (const hx_c128*)x, (hx_c128*)y, a, b, s);
  const double aa[2] = {a, 0.0}, bb[2] = {b, 0.0};
  if (transfer)
    return hx_csr_spmv_transfer(A.rows, A.nnz, A.rp, A.ci, (const hx_c128*)A.va, (const hx_c128*)x, (hx_c128*)y, aa, bb, s);
  return hx_csr_spmv_z(A.rows, A.nnz, A.rp, A.ci, (const hx_c128*)A.va, (const hx_c128*)x, (hx_c128*)y, aa, bb, 0, s);
}
static int op_spmv(const CsrT<float>& A, const float* x, float* y, float a, float b, bool transfer, hx_stream s) {
  if (A.real_vals) return hx_csr_spmv_creal(A.rows, A.nnz, A.rp, A.ci, A.va, x, y, a, b, s);
  return hx_csr_spmv_c(transfer ? 1 : 2, A.rows, A.nnz, A.rp, A.ci, A.va, x, y, a, b, s);
}
// y = a K x + b y through the SELL companion when the level has one
static int op_kmult(const LevelT<double>& L, const double* x, double* y, double a, double b, hx_stream s) {
  if (L.sell) { const double aa[2] = {a, 0.0}, bb[2] = {b, 0.0};
                return hx_sell_spmv_z(L.sell, (const hx_c128*)x, (hx_c128*)y, aa, bb, s); }
  return op_spmv(L.K, x, y, a, b, false, s);
}
static int op_kmult(const LevelT<float>& L, const float* x, float* y, float a, float b, hx_stream s) {
  if (L.sell) return hx_sell_spmv_c64(L.sell, (const hx_c64*)x, (hx_c64*)y, a, b, s);
  return op_spmv(L.K, x, y, a, b, false, s);
}
// r = b - K x (one pass, no copy of b)
static int op_residual(const LevelT<double>& L, const double* x, const double* b, double* r, hx_stream s, int part = 0) {
  if (L.sell) return hx_sell_residual_z_part(L.sell, part, (const hx_c128*)x, (const hx_c128*)b, (hx_c128*)r, s);
  return hx_csr_residual_z(L.K.rows, L.K.nnz, L.K.rp, L.K.ci, (const hx_c128*)L.K.va, (const hx_c128*)x, (const hx_c128*)b, (hx_c128*)r, s);
}
static int op_residual(const LevelT<float>& L, const float* x, const float* b, float* r, hx_stream s, int part = 0) {
  if (L.sell) return hx_sell_residual_c64_part(L.sell, part, (const hx_c64*)x, (const hx_c64*)b, (hx_c64*)r, s);
  return hx_csr_residual_c(L.K.rows, L.K.nnz, L.K.rp, L.K.ci, L.K.va, x, b, r, s);
}
// xout = xin + w D^-1 (b - K xin) + mom (xin - xprev)     (xprev may be NULL)
static int op_smooth(const LevelT<double>& L, const double* b, const double* xin, double* xout, double w,
                     const double* xprev, double mom, hx_stream s, int part = 0) {
  if (L.sell) return hx_sell_smooth_z_part(L.sell, part, (const hx_c128*)L.dinv, (const hx_c128*)b, (const hx_c128*)xin, (hx_c128*)xout,
                                           w, (const hx_c128*)xprev, mom, s);
  return hx_csr_smooth_z(L.n, L.K.nnz, L.K.rp, L.K.ci, (const hx_c128*)L.K.va, (const hx_c128*)L.dinv, (const hx_c128*)b,
                         (const hx_c128*)xin, (hx_c128*)xout, w, (const hx_c128*)xprev, mom, s);
}
static int op_smooth(const LevelT<float>& L, const float* b, const float* xin, float* xout, double w, const float* xprev,
                     double mom, hx_stream s, int part = 0) {
  if (L.sell) return hx_sell_smooth_c64_part(L.sell, part, (const hx_c64*)L.dinv, (const hx_c64*)b, (const hx_c64*)xin, (hx_c64*)xout,
                                             (float)w, (const hx_c64*)xprev, (float)mom, s);
  return hx_csr_smooth_c(L.n, L.K.nnz, L.K.rp, L.K.ci, L.K.va, L.dinv, b, xin, xout, (float)w, xprev, (float)mom, s);
}
// bc = R r and, in the same launch, the coarse level's first smoothing step x1 = a0 Dc^-1 bc (real-valued transfers only)
static int op_restrict_diag(const CsrT<double>& R, const double* r, double* bc, const double* dc, double* x1, double a0, hx_stream s) {
  return hx_csr_restrict_diag_z(R.rows, R.nnz, R.rp, R.ci, R.va, (const hx_c128*)r, (hx_c128*)bc, (const hx_c128*)dc, (hx_c128*)x1, a0, s);
}
static int op_restrict_diag(const CsrT<float>& R, const float* r, float* bc, const float* dc, float* x1, double a0, hx_stream s) {
  return hx_csr_restrict_diag_c(R.rows, R.nnz, R.rp, R.ci, R.va, r, bc, dc, x1, (float)a0, s);
}
// ---- deep-halo mode: the same steps on the rows of the FIRST GHOST LAYER (stored apart in L.K1; vector rows n .. n + n_g1 - 1)
static int op_smooth_g(const LevelT<double>& L, int32_t rows, int64_t nnz, const double* b, const double* xin, double* xout, double w,
                       const double* xprev, double mom, hx_stream s) {
  return hx_csr_smooth_rows_z(rows, L.n, nnz, L.K1.rp, L.K1.ci, (const hx_c128*)L.K1.va, (const hx_c128*)L.dinv, (const hx_c128*)b,
                              (const hx_c128*)xin, (hx_c128*)xout, w, (const hx_c128*)xprev, mom, s);
}
static int op_smooth_g(const LevelT<float>& L, int32_t rows, int64_t nnz, const float* b, const float* xin, float* xout, double w,
                       const float* xprev, double mom, hx_stream s) {
  return hx_csr_smooth_rows_c(rows, L.n, nnz, L.K1.rp, L.K1.ci, L.K1.va, L.dinv, b, xin, xout, (float)w, xprev, (float)mom, s);
}
static int op_residual_g1(const LevelT<double>& L, const double* x, const double* b, double* r, hx_stream s) {
  const size_t o = 2 * (size_t)L.n;
  return hx_csr_residual_z(L.n_g1, L.nnz_g1, L.K1.rp, L.K1.ci, (const hx_c128*)L.K1.va, (const hx_c128*)x, (const hx_c128*)(b + o),
                           (hx_c128*)(r + o), s);
}
static int op_residual_g1(const LevelT<float>& L, const float* x, const float* b, float* r, hx_stream s) {
  const size_t o = 2 * (size_t)L.n;
  return hx_csr_residual_c(L.n_g1, L.nnz_g1, L.K1.rp, L.K1.ci, L.K1.va, x, b + o, r + o, s);
}
// x[off .. off + n) = w D^-1 b on a range of the extended vectors
static int op_diag_n(const LevelT<double>& L, int64_t off, int64_t n, double w, const double* b, double* x, hx_stream s) {
  return hx_zdiag_scale(n, w, (const hx_c128*)(L.dinv + 2 * off), (const hx_c128*)(b + 2 * off), (hx_c128*)(x + 2 * off), s);
}
static int op_diag_n(const LevelT<float>& L, int64_t off, int64_t n, double w, const float* b, float* x, hx_stream s) {
  return hx_cdiag_scale(n, (float)w, L.dinv + 2 * off, b + 2 * off, x + 2 * off, s);
}
static int op_diag(const LevelT<double>& L, double w, const double* b, double* x, hx_stream s) {
  return hx_zdiag_scale(L.n, w, (const hx_c128*)L.dinv, (const hx_c128*)b, (hx_c128*)x, s);
}
static int op_diag(const LevelT<float>& L, double w, const float* b, float* x, hx_stream s) {
  return hx_cdiag_scale(L.n, (float)w, L.dinv, b, x, s);
}

extern "C" int hx_amg_create(hx_amg** out, int n_levels, int pre, int post, double omega) {
  HX_REQUIRE(out && n_levels >= 1 && pre >= 0 && post >= 0 && pre + post >= 1 && omega > 0, "bad argument");
  hx_amg* h = new hx_amg();
  h->D.L.resize(n_levels);
  h->S.L.resize(n_levels);
  h->pre = pre; h->post = post; h->omega = omega;
  *out = h;
  return HX_OK;
}

extern "C" int hx_amg_destroy(hx_amg* h) {
  if (!h) return HX_OK;
  auto release = [](auto& l) {
    (void)hx_free(l.x); (void)hx_free(l.x2); (void)hx_free(l.x3); (void)hx_free(l.b); (void)hx_free(l.r);
  };
  for (auto& l : h->D.L) release(l);
  for (auto& l : h->S.L) release(l);
  drop_blk(h);
  (void)hx_free(h->rec.U); (void)hx_free(h->rec.CV); (void)hx_free(h->rec.CVf);
  (void)hx_free(h->rec.Uf); (void)hx_free(h->rec.Cd); (void)hx_free(h->rec.Einv);
  if (h->vexec) (void)hipGraphExecDestroy(h->vexec);
  if (h->ev_in) (void)hipEventDestroy(h->ev_in);
  if (h->ev_out) (void)hipEventDestroy(h->ev_out);
  if (h->gstream) { (void)hipStreamSynchronize(h->gstream); hx_pool_register_stream(h->gstream, false); (void)hipStreamDestroy(h->gstream); }
  if (h->cg_pin) (void)hipHostFree(h->cg_pin);
  for (hipEvent_t e : h->cg_ev) if (e) (void)hipEventDestroy(e);
  delete h;
  return HX_OK;
}

template <typename T>
static int set_level(bool real_transfers, HierT<T>& H, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK, const T* vaK,
                     const T* dinv, int32_t n_coarse, int64_t nnzP, const int32_t* rpP, const int32_t* ciP,
                     const T* vaP, int64_t nnzR, const int32_t* rpR, const int32_t* ciR, const T* vaR) {
  HX_REQUIRE(l >= 0 && l < (int)H.L.size() && n > 0 && rpK && ciK && vaK, "bad argument");
  LevelT<T>& L = H.L[l];
  L.n = n;
  L.p_ghost_rows = false;      // a level that is set again starts without ghost rows in P (hx_amg_set_prolongator_ghost_rows re-arms it)
  L.deep = false; L.K1 = CsrT<T>(); L.n_g1 = 0; L.nnz_g1 = 0; L.deep_layers = 0;      // ... and out of deep-halo mode (hx_amg_set_level_deep re-arms it)
  L.K.rows = n; L.K.nnz = nnzK; L.K.rp = rpK; L.K.ci = ciK; L.K.va = vaK;
  L.dinv = dinv;
  if (l != (int)H.L.size() - 1) {
    HX_REQUIRE(dinv && n_coarse > 0 && rpP && ciP && vaP && rpR && ciR && vaR, "missing transfer operators");
    L.P.rows = n; L.P.nnz = nnzP; L.P.rp = rpP; L.P.ci = ciP; L.P.va = vaP; L.P.real_vals = real_transfers;
    L.R.rows = n_coarse; L.R.nnz = nnzR; L.R.rp = rpR; L.R.ci = ciR; L.R.va = vaR; L.R.real_vals = real_transfers;
  }
  if (!L.x) {
    const size_t bytes = 2 * sizeof(T) * (size_t)n;
    HX_HIP(hx_malloc((void**)&L.x, bytes));
    HX_HIP(hx_malloc((void**)&L.x2, bytes));
    HX_HIP(hx_malloc((void**)&L.x3, bytes));
    HX_HIP(hx_malloc((void**)&L.b, bytes));
    HX_HIP(hx_malloc((void**)&L.r, bytes));
  }
  return HX_OK;
}

extern "C" int hx_amg_set_level(hx_amg* h, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK,
                                const hx_c128* vaK, const hx_c128* dinv, int32_t n_coarse, int64_t nnzP,
                                const int32_t* rpP, const int32_t* ciP, const hx_c128* vaP, int64_t nnzR,
                                const int32_t* rpR, const int32_t* ciR, const hx_c128* vaR) {
  HX_REQUIRE(h, "null handle");
  h->normK_val = nullptr;      // new operator values: the cached |K|_inf of the backward-error test is stale
  drop_blk(h);
  return set_level<double>(h->real_transfers, h->D, l, n, nnzK, rpK, ciK, (const double*)vaK, (const double*)dinv, n_coarse, nnzP, rpP, ciP,
                           (const double*)vaP, nnzR, rpR, ciR, (const double*)vaR);
}
extern "C" int hx_amg_set_level_c(hx_amg* h, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK,
                                  const hx_c64* vaK, const hx_c64* dinv, int32_t n_coarse, int64_t nnzP,
                                  const int32_t* rpP, const int32_t* ciP, const hx_c64* vaP, int64_t nnzR,
                                  const int32_t* rpR, const int32_t* ciR, const hx_c64* vaR) {
  HX_REQUIRE(h, "null handle");
  h->single = true;
  h->normK_val = nullptr;
  drop_graph(h);
  drop_blk(h);
  return set_level<float>(h->real_transfers, h->S, l, n, nnzK, rpK, ciK, (const float*)vaK, (const float*)dinv, n_coarse, nnzP, rpP, ciP,
                          (const float*)vaP, nnzR, rpR, ciR, (const float*)vaR);
}
// P / R values of the following hx_amg_set_level* calls are REAL arrays (float for the complex64 copy, double otherwise)
extern "C" int hx_amg_set_real_transfers(hx_amg* h, int flag) {
  HX_REQUIRE(h, "null handle");
  h->real_transfers = flag != 0;
  return HX_OK;
}
extern "C" int hx_amg_set_level_sell(hx_amg* h, int l, const hx_sell* sell) {
  HX_REQUIRE(h && l >= 0 && l < (int)h->D.L.size(), "bad argument");
  drop_graph(h);
  h->D.L[l].sell = sell;       // the handle must carry the value copy of the precision the V-cycle runs in
  h->S.L[l].sell = sell;
  return HX_OK;
}

extern "C" int hx_amg_set_smoother(hx_amg* h, int l, double rho) {
  HX_REQUIRE(h && l >= 0 && l < (int)h->D.L.size() && rho >= 0.0, "bad argument");
  drop_graph(h);
  h->D.L[l].rho = rho;
  h->S.L[l].rho = rho;
  return HX_OK;
}

// work vectors of a sharded level carry a ghost block behind the owned entries
template <typename T>
static int resize_ext(HierT<T>& H, int n_sharded, const int32_t* n_ext) {
  for (int l = 0; l < n_sharded; ++l) {
    LevelT<T>& L = H.L[l];
    HX_REQUIRE(L.n > 0 && n_ext[l] >= L.n, "n_ext must cover the owned rows of the level");
    (void)hx_free(L.x); (void)hx_free(L.x2); (void)hx_free(L.x3); (void)hx_free(L.r);
    L.x = L.x2 = L.x3 = L.r = nullptr;
    const size_t bytes = 2 * sizeof(T) * (size_t)n_ext[l];
    HX_HIP(hx_malloc((void**)&L.x, bytes));
    HX_HIP(hx_malloc((void**)&L.x2, bytes));
    HX_HIP(hx_malloc((void**)&L.x3, bytes));
    HX_HIP(hx_malloc((void**)&L.r, bytes));
    // (blocks from the pool are idle when handed out; the fills run on the null stream and are awaited here, so that kernels of a
    // non-blocking caller stream cannot overtake them)
    HX_HIP(hipMemsetAsync(L.x, 0, bytes, nullptr)); HX_HIP(hipMemsetAsync(L.x2, 0, bytes, nullptr));
    HX_HIP(hipMemsetAsync(L.x3, 0, bytes, nullptr)); HX_HIP(hipMemsetAsync(L.r, 0, bytes, nullptr));
    HX_HIP(hipStreamSynchronize(nullptr));
    L.n_ext = n_ext[l];
  }
  return HX_OK;
}
extern "C" int hx_amg_set_sharded(hx_amg* h, int n_sharded, const int32_t* n_ext, hx_halo_fn halo,
                                  hx_allreduce_fn allreduce, void* ctx) {
  HX_REQUIRE(h && n_ext && halo && allreduce, "bad argument");
  const size_t nl = h->single ? h->S.L.size() : h->D.L.size();
  HX_REQUIRE(n_sharded >= 1 && (size_t)n_sharded < nl, "row-sharded mode needs at least one replicated coarse level");
  drop_graph(h);
  drop_blk(h);
  h->sharded = true; h->halo = halo; h->allreduce = allreduce; h->ctx = ctx; h->n_sharded = n_sharded;
  return h->single ? resize_ext<float>(h->S, n_sharded, n_ext) : resize_ext<double>(h->D, n_sharded, n_ext);
}

// The prolongators of the first `n_levels` row-sharded levels carry n_ext ROWS: behind this rank's rows the owners' rows of its
// ghost vertices (same values, this rank's column numbering).  The prolongation then corrects the ghost block of the iterate as
// well -- it held the pre-smoothed values since the exchange in front of the residual -- so the first post-smoothing step needs
// no exchange: one halo fewer per sharded level and cycle.  Call after hx_amg_set_sharded.
extern "C" int hx_amg_set_prolongator_ghost_rows(hx_amg* h, int n_levels) {
  HX_REQUIRE(h && h->sharded && n_levels >= 0 && n_levels <= h->n_sharded, "bad argument");
  drop_graph(h);
  // the registered P must really carry n_ext rows: its row pointer was given with rows + 1 entries, so the declared entry count
  // has to be the one behind row n_ext (a P with owned rows only would make the kernel read rp / ci out of bounds)
  for (int l = 0; l < n_levels; ++l) {
    const int32_t n_ext = h->single ? h->S.L[l].n_ext : h->D.L[l].n_ext;
    const int32_t* rp = h->single ? h->S.L[l].P.rp : h->D.L[l].P.rp;
    const int64_t nnz = h->single ? h->S.L[l].P.nnz : h->D.L[l].P.nnz;
    HX_REQUIRE(rp && n_ext > 0, "level without a prolongator / without a ghost block");
    int32_t last = -1;
    HX_HIP(hipMemcpy(&last, rp + n_ext, sizeof(int32_t), hipMemcpyDeviceToHost));
    HX_REQUIRE((int64_t)last == nnz, "the prolongator of a ghost-row level must have n_ext rows (row pointer of n_ext + 1 entries)");
  }
  for (int l = 0; l < h->n_sharded; ++l) {
    const bool on = l < n_levels;
    if (h->single) { auto& L = h->S.L[l]; L.p_ghost_rows = on; L.P.rows = on ? L.n_ext : L.n; }
    else { auto& L = h->D.L[l]; L.p_ghost_rows = on; L.P.rows = on ? L.n_ext : L.n; }
  }
  return HX_OK;
}

// Deep-halo mode of row-sharded level l (call after hx_amg_set_sharded, whose n_ext[l] then counts ALL ghost layers, and after the
// level was set with an inverse diagonal of n_ext[l] entries and a prolongator of n_ext[l] rows).  layers = 2: vectors
// [owned | g1 | g2], K1 = the operator rows of layer 1 (n_g1 = n_gs rows) -- the cycle exchanges its right-hand side and its
// pre-smoothed iterate two layers deep: 3 exchanges per cycle on a level whose coarse level is sharded too (5 in the classic
// schedule), 2 where it is replicated (3).  layers = 3: vectors [owned | g1 | g2 | g3], K1 = the rows of layer 1 followed by those
// of layer 2 (n_gs rows in all) -- the right-hand side travels three layers deep, both layers are pre-smoothed redundantly and the
// second exchange goes away: 2 and 1.  Columns of K1 in the extended numbering, values in the precision of the handle's cycle, all
// fetched from their owners at set-up; the exchange callback is called with level number 32 + l.  Same arithmetic as the classic
// schedule.  Needs pre = post = 2 smoothing steps; option amg_deep_halo = 0 falls back to the classic schedule.
extern "C" int hx_amg_set_level_deep(hx_amg* h, int l, int layers, int32_t n_g1, int64_t nnz_g1, int32_t n_gs, int64_t nnz_gs,
                                     const int32_t* rp1, const int32_t* ci1, const void* va1) {
  HX_REQUIRE(h && h->sharded && l >= 0 && l < h->n_sharded && (layers == 2 || layers == 3) && n_g1 >= 0 && n_gs >= n_g1 &&
             nnz_g1 >= 0 && nnz_gs >= nnz_g1 && rp1 && (nnz_gs == 0 || (ci1 && va1)) && (layers == 3 || n_gs == n_g1), "bad argument");
  drop_graph(h);
  auto arm = [&](auto& L, auto tag) -> int {
    typedef decltype(tag) T;
    HX_REQUIRE(L.n > 0 && L.n_ext >= L.n + n_gs, "n_ext of a deep level counts owned rows + all ghost layers");
    L.K1.rows = n_gs; L.K1.nnz = nnz_gs; L.K1.rp = rp1; L.K1.ci = ci1; L.K1.va = (const T*)va1;
    L.n_g1 = n_g1; L.nnz_g1 = nnz_g1; L.deep_layers = layers; L.deep = true;
    (void)hx_free(L.b); L.b = nullptr;                     // the right-hand side now lives in an extended buffer
    const size_t bytes = 2 * sizeof(T) * (size_t)L.n_ext;
    HX_HIP(hx_malloc((void**)&L.b, bytes));
    HX_HIP(hipMemsetAsync(L.b, 0, bytes, nullptr));
    HX_HIP(hipStreamSynchronize(nullptr));
    return HX_OK;
  };
  return h->single ? arm(h->S.L[l], float()) : arm(h->D.L[l], double());
}

extern "C" int hx_amg_set_halo_overlap(hx_amg* h, hx_halo_begin_fn begin, hx_halo_end_fn end) {
  HX_REQUIRE(h && ((begin && end) || (!begin && !end)), "both callbacks or none");
  h->halo_begin = begin; h->halo_end = end;
  return HX_OK;
}

extern "C" int hx_amg_set_coarse(hx_amg* h, int32_t n_c, int64_t nnz, const int32_t* rp, const int32_t* ci,
                                 const hx_c128* va) {
  HX_REQUIRE(h && n_c > 0 && rp && ci && va && h->D.L.back().n == n_c, "bad argument");
  drop_blk(h);
  h->D.coarse_inv.rows = n_c; h->D.coarse_inv.nnz = nnz; h->D.coarse_inv.rp = rp; h->D.coarse_inv.ci = ci;
  h->D.coarse_inv.va = (const double*)va;
  return HX_OK;
}
extern "C" int hx_amg_set_coarse_c(hx_amg* h, int32_t n_c, int64_t nnz, const int32_t* rp, const int32_t* ci,
                                   const hx_c64* va) {
  HX_REQUIRE(h && n_c > 0 && rp && ci && va && h->S.L.back().n == n_c, "bad argument");
  drop_graph(h);
  drop_blk(h);
  h->S.coarse_inv.rows = n_c; h->S.coarse_inv.nnz = nnz; h->S.coarse_inv.rp = rp; h->S.coarse_inv.ci = ci;
  h->S.coarse_inv.va = (const float*)va;
  return HX_OK;
}

// Chebyshev coefficients for eigenvalues of D^-1 K in [lo, hi]: x_{k+1} = x_k + a_k D^-1 (b - K x_k) + c_k (x_k - x_{k-1})
struct Cheb { double a[4], c[4]; };
static Cheb cheb_coeffs(double rho) {
  const double lo_frac = HX_OPT_DOUBLE("cheb_lo", 0.1);   // experiment hook
  const int kind = (int)HX_OPT_INT("cheb_kind", 1);       // experiment hook
  const double hi = 1.1 * rho, lo = lo_frac * hi;
  Cheb C;
  if (kind == 4) {
    // Chebyshev polynomials of the 4th kind (Lottes 2022): no lower bound needed, d_k = c_k d_{k-1} + a_k D^-1 r_k
    for (int k = 0; k < 4; ++k) {
      C.a[k] = (8.0 * k + 4.0) / (2.0 * k + 3.0) / hi;
      C.c[k] = k ? (2.0 * k - 1.0) / (2.0 * k + 3.0) : 0.0;
    }
    return C;
  }
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
  double r_old = 1.0 / sigma;
  C.a[0] = 1.0 / theta; C.c[0] = 0.0;
  for (int k = 1; k < 4; ++k) {
    const double r_new = 1.0 / (2.0 * sigma - r_old);
    C.a[k] = 2.0 * r_new / delta;
    C.c[k] = r_new * r_old;
    r_old = r_new;
  }
  return C;
}

// xout = V-cycle(b) on level l with zero initial guess (xout must not be one of the level's x buffers or b).
// Smoother: `pre` / `post` steps of damped Jacobi (rho == 0) or of a Chebyshev polynomial in D^-1 K (rho > 0).
template <typename T> static int vcycle(hx_amg* h, HierT<T>& H, int l, const T* b, T* xout, hx_stream s, bool x1_ready = false);

// The cycle of a row-sharded level in DEEP-HALO mode (hx_amg_set_level_deep): V(2,2) with the arithmetic of the classic schedule,
// three (layers = 2) or two (layers = 3) exchanges instead of five.  Vectors are [owned | g1 | g2 (| g3)]; "valid on" says where an
// iterate equals the global one.
//   exchange b over all layers                      -> b valid everywhere
//   x1 = a0 D^-1 b on all entries                   -> x1 valid everywhere (local)
//   x2 = step(x1) on owned rows and on the rows of K1 (g1; with three layers also g2)
//   [two layers: exchange x2 two layers deep]       -> x2 valid on owned + g1 + g2 either way
//   r = b - K x2 on owned rows and on g1 rows       -> r valid on owned + g1: the restriction (owned coarse rows over
//                                                      [owned | g1] fine columns) needs no exchange of its own
//   coarse solve; [exchange of the coarse correction when the coarse level is sharded: its ordinary plan]
//   x~ = x2 + P e on owned + g1 + g2 (P carries the owners' rows of both layers)
//   x3 = step(x~) on owned rows and on g1 rows      -> valid on owned + g1
//   x4 = step(x3, x~) on owned rows                 -> the result
template <typename T>
static int vcycle_deep(hx_amg* h, HierT<T>& H, int l, const T* b_in, T* xout, hx_stream s) {
  const int DEEP = 32;                                      // plan number of a level's multi-layer exchange: 32 + level
  LevelT<T>& L = H.L[l];
  LevelT<T>& Lc = H.L[l + 1];
  int rc;
  const int EB = (int)(2 * sizeof(T));
  const bool cheb = L.rho > 0.0;
  const Cheb C = cheb ? cheb_coeffs(L.rho) : Cheb();
  const bool shc = l + 1 < h->n_sharded;
  auto deep_halo = [&](T* v) -> int {
    return h->halo(h->ctx, DEEP + l, v, EB, s) == 0 ? HX_OK : hx_fail(HX_ERR_COMM, "vcycle_deep", "multi-layer halo exchange failed");
  };
  // one smoothing step on the owned rows and on the first `g_rows` ghost-layer rows (entries g_nnz of K1)
  auto step = [&](const T* xin, T* out, int k, const T* xprev, bool zero_prev, int32_t g_rows, int64_t g_nnz) -> int {
    const double w = cheb ? C.a[k] : h->omega, mom = (cheb && k) ? C.c[k] : 0.0;
    const T* pv = zero_prev ? nullptr : xprev;
    int r1 = op_smooth(L, L.b, xin, out, w, pv, mom, s);
    if (r1 || g_rows == 0) return r1;
    return op_smooth_g(L, g_rows, g_nnz, L.b, xin, out, w, pv, mom, s);
  };
  if (b_in != L.b) HX_HIP(hipMemcpyAsync(L.b, b_in, 2 * sizeof(T) * (size_t)L.n, hipMemcpyDeviceToDevice, hx_s(s)));
  T* x1 = L.x; T* x2 = L.x2; T* x3 = L.x3;
  const double a0 = cheb ? C.a[0] : h->omega;
  // The exchange of b overlaps what needs OWNED entries only: x1 on the owned block and the interior slices of the first smoothing
  // step (split form of the exchange, as for the operator product: the exchange stays on the caller's stream, that work runs on the
  // communicator's side stream); then x1 on the ghost layers, the boundary slices and the redundantly smoothed layer rows.
  if (h->halo_begin && h->halo_end && L.sell && hx_sell_has_parts(L.sell) && HX_OPT_INT("amg_deep_overlap", 1) != 0) {
    hx_stream side = nullptr;
    if (h->halo_begin(h->ctx, DEEP + l, L.b, EB, s, &side) != 0) return hx_fail(HX_ERR_COMM, "vcycle_deep", "multi-layer halo exchange (begin) failed");
    int r1 = op_diag_n(L, 0, L.n, a0, L.b, x1, side);
    if (!r1) r1 = op_smooth(L, L.b, x1, x2, cheb ? C.a[1] : h->omega, nullptr, cheb ? C.c[1] : 0.0, side, 1);
    if (h->halo_end(h->ctx, DEEP + l, s) != 0) return hx_fail(HX_ERR_COMM, "vcycle_deep", "multi-layer halo exchange (end) failed");
    if (r1) return r1;
    if ((rc = op_diag_n(L, L.n, (int64_t)L.n_ext - L.n, a0, L.b, x1, s))) return rc;
    if ((rc = op_smooth(L, L.b, x1, x2, cheb ? C.a[1] : h->omega, nullptr, cheb ? C.c[1] : 0.0, s, 2))) return rc;
    if (L.K1.rows > 0 && (rc = op_smooth_g(L, L.K1.rows, L.K1.nnz, L.b, x1, x2, cheb ? C.a[1] : h->omega, nullptr, cheb ? C.c[1] : 0.0, s)))
      return rc;
  } else {
    if ((rc = deep_halo(L.b))) return rc;
    if ((rc = op_diag_n(L, 0, L.n_ext, a0, L.b, x1, s))) return rc;
    // pre-smoothing step 1: the iterate before x1 is the zero start vector (xprev NULL with momentum); redundantly on every row K1 holds
    if ((rc = step(x1, x2, 1, nullptr, true, L.K1.rows, L.K1.nnz))) return rc;
  }
  if (L.deep_layers == 2 && (rc = deep_halo(x2))) return rc;
  if ((rc = op_residual(L, x2, L.b, L.r, s))) return rc;
  if (L.n_g1 > 0 && (rc = op_residual_g1(L, x2, L.b, L.r, s))) return rc;
  if ((rc = op_spmv(L.R, L.r, Lc.b, (T)1, (T)0, true, s))) return rc;
  if (!shc && h->allreduce(h->ctx, Lc.b, 2 * (int64_t)Lc.n, sizeof(T) == 8 ? 1 : 0, 0, s) != 0)
    return hx_fail(HX_ERR_COMM, "vcycle_deep", "all-reduce callback failed");
  T* ec = Lc.r;
  if ((rc = vcycle<T>(h, H, l + 1, Lc.b, ec, s, false))) return rc;
  if (shc && (rc = halo_level(h, l + 1, ec, EB, s))) return rc;
  if ((rc = op_spmv(L.P, ec, x2, (T)1, (T)1, true, s))) return rc;        // x~ on owned + g1 + g2 (P.rows = n_ext; empty rows on g3)
  if ((rc = step(x2, x3, 0, nullptr, true, L.n_g1, L.nnz_g1))) return rc;  // x3 on owned + g1
  return step(x3, xout, 1, x2, false, 0, 0);                              // x4 on owned rows (xout is none of x1, x2, x3)
}

// x1_ready: the level's first pre-smoothing step (x1 = a0 D^-1 b, in L.x) was written by the restriction that produced b.
template <typename T>
static int vcycle(hx_amg* h, HierT<T>& H, int l, const T* b, T* xout, hx_stream s, bool x1_ready) {
  LevelT<T>& L = H.L[l];
  int rc;
  if (l == (int)H.L.size() - 1) return op_spmv(H.coarse_inv, b, xout, (T)1, (T)0, true, s);
  if (h->sharded && l < h->n_sharded && L.deep && L.p_ghost_rows && h->pre == 2 && h->post == 2 && xout != L.x2 && xout != L.x3 &&
      HX_OPT_INT("amg_deep_halo", 1) != 0 && HX_OPT_INT("amg_coarse_extra", 0) == 0)
    return vcycle_deep<T>(h, H, l, b, xout, s);
  const bool cheb = L.rho > 0.0;
  const Cheb C = cheb ? cheb_coeffs(L.rho) : Cheb();
  const int extra = (int)HX_OPT_INT("amg_coarse_extra", 0);   // experiment hook
  const int n_pre = h->pre + (l > 0 ? extra : 0), n_post = h->post + (l > 0 ? extra : 0);
  T* bufs[3] = {L.x, L.x2, L.x3};
  auto other = [&](const T* a, const T* c) -> T* { for (T* q : bufs) if (q != a && q != c) return q; return nullptr; };
  T* cur = L.x;
  T* prev = nullptr;
  LevelT<T>& Lc = H.L[l + 1];
  const int EB = (int)(2 * sizeof(T));                      // bytes per complex entry of this precision
  const bool sh = h->sharded && l < h->n_sharded;           // this level is row-sharded ...
  const bool shc = h->sharded && l + 1 < h->n_sharded;      // ... and so is the next one
  // restriction of the residual in `res`: a sharded coarse level computes its OWNED entries from the exchanged fine
  // residual (R: owned coarse rows over [owned | ghost] fine columns); the first replicated level sums the ranks'
  // partial right-hand sides
  // The restriction also leaves the coarse level's first smoothing step (one launch fewer per level and cycle) when that level
  // smooths at all, its right-hand side is final here (no all-reduce of partial sums follows) and the transfers are real-valued.
  const bool coarse_is_last = l + 1 == (int)H.L.size() - 1;
  const int n_pre_c = h->pre + extra;
  const bool fold = HX_OPT_INT("amg_fold_diag", 1) != 0 && !coarse_is_last && n_pre_c > 0 && L.R.real_vals && !(sh && !shc) && Lc.dinv;
  auto restrict_from = [&](T* res) -> int {
    if (shc && (rc = halo_level(h, l, res, EB, s))) return rc;
    if (fold) {
      const double a0c = Lc.rho > 0.0 ? cheb_coeffs(Lc.rho).a[0] : h->omega;
      return op_restrict_diag(L.R, res, Lc.b, Lc.dinv, Lc.x, a0c, s);
    }
    if ((rc = op_spmv(L.R, res, Lc.b, (T)1, (T)0, true, s))) return rc;
    if (sh && !shc && h->allreduce(h->ctx, Lc.b, 2 * (int64_t)Lc.n, sizeof(T) == 8 ? 1 : 0, 0, s) != 0)
      return hx_fail(HX_ERR_COMM, "vcycle", "all-reduce callback failed");
    return HX_OK;
  };
  if (n_pre == 0) {
    // no pre-smoothing: x = 0, so the residual IS b and no operator product is needed before the restriction
    if (shc) {      // (b has no ghost block: stage it in the level's residual buffer)
      HX_HIP(hipMemcpyAsync(L.r, b, 2 * sizeof(T) * (size_t)L.n, hipMemcpyDeviceToDevice, hx_s(s)));
      if ((rc = restrict_from(L.r))) return rc;
    } else if ((rc = restrict_from(const_cast<T*>(b)))) return rc;
  } else {
    // pre-smoothing from x = 0: the first step is a diagonal scaling (already in L.x when the restriction above this level wrote it)
    if (!x1_ready && (rc = op_diag(L, cheb ? C.a[0] : h->omega, b, cur, s))) return rc;
    for (int k = 1; k < n_pre; ++k) {
      T* nxt = other(cur, prev);
      // (for k = 1 the previous iterate of the Chebyshev recurrence is the zero start vector: prev == nullptr)
      if ((rc = with_halo(h, L, l, cur, EB, s, [&](int part, hx_stream st) {
            return cheb ? op_smooth(L, b, cur, nxt, C.a[k < 4 ? k : 3], prev, C.c[k < 4 ? k : 3], st, part)
                        : op_smooth(L, b, cur, nxt, h->omega, nullptr, 0.0, st, part); }))) return rc;
      prev = cur; cur = nxt;
    }
    // residual r = b - K x, restrict
    if ((rc = with_halo(h, L, l, cur, EB, s, [&](int part, hx_stream st) { return op_residual(L, cur, b, L.r, st, part); }))) return rc;
    if ((rc = restrict_from(L.r))) return rc;
  }
  T* ec = Lc.r;      // coarse correction lands in the coarse level's r buffer (free at this point)
  if ((rc = vcycle<T>(h, H, l + 1, Lc.b, ec, s, fold))) return rc;
  if (shc && (rc = halo_level(h, l + 1, ec, EB, s))) return rc;        // P: owned rows over [owned | ghost] coarse columns
  if ((rc = op_spmv(L.P, ec, cur, (T)1, n_pre == 0 ? (T)0 : (T)1, true, s))) return rc;    // x (+)= P e_c
  // post-smoothing: a fresh polynomial started from the corrected x; the last step writes xout
  prev = nullptr;
  // (P with ghost rows: the prolongation has corrected the ghost block of cur too, the first step needs no exchange)
  const bool ghosts_current = sh && L.p_ghost_rows && n_pre > 0;
  for (int k = 0; k < n_post; ++k) {
    T* dst = (k == n_post - 1) ? xout : other(cur, prev);
    auto step = [&](int part, hx_stream st) {
      return cheb ? op_smooth(L, b, cur, dst, C.a[k < 4 ? k : 3], k ? prev : nullptr, k ? C.c[k < 4 ? k : 3] : 0.0, st, part)
                  : op_smooth(L, b, cur, dst, h->omega, nullptr, 0.0, st, part); };
    if ((rc = (k == 0 && ghosts_current) ? step(0, s) : with_halo(h, L, l, cur, EB, s, step))) return rc;
    prev = cur; cur = dst;
  }
  if (n_post == 0)
    HX_HIP(hipMemcpyAsync(xout, cur, 2 * sizeof(T) * (size_t)L.n, hipMemcpyDeviceToDevice, hx_s(s)));
  return HX_OK;
}

// Request (1) / drop (0) the hipGraph replay of the complex64 V-cycle.  Ignored in row-sharded mode (the cycle calls
// back to the host for its exchanges) and for the complex128 cycle.
extern "C" int hx_amg_enable_graph(hx_amg* h, int enable) {
  HX_REQUIRE(h, "null handle");
  if (!enable) { if (h->vexec) { (void)hipGraphExecDestroy(h->vexec); h->vexec = nullptr; } h->graph_state = -1; }
  else if (h->graph_state < 0) h->graph_state = 0;
  return HX_OK;
}
// 1 if the last cycles were replayed from the captured graph
extern "C" int hx_amg_graph_active(const hx_amg* h) { return h && h->graph_state == 1 ? 1 : 0; }

// The complex64 cycle from the level-0 input buffer (L0.b) to the output buffer (L0.r), no precision casts: hx_cocg writes
// its residual into the input buffer in complex64 and reads the result as complex64.
int hx_amg_vcycle_buffers(hx_amg* h, hx_c64** b, hx_c64** z) {
  HX_REQUIRE(h && b && z, "bad argument");
  if (!h->single || h->S.L.empty() || h->S.L[0].n <= 0) { *b = nullptr; *z = nullptr; return HX_OK; }
  *b = (hx_c64*)h->S.L[0].b;
  *z = (hx_c64*)h->S.L[0].r;
  return HX_OK;
}
// Where the producer of the cycle's complex64 input may ALSO leave level 0's first smoothing step x1 = a0 D^-1 b (the diagonal
// launch of the cycle then goes away): inverse diagonal, destination and coefficient -- or dinv == NULL when the cycle cannot take
// it (complex128 cycle, sharded, captured graph, no pre-smoothing, single level, fold switched off).  The producer announces a
// written x1 with hx_amg_x1_ready; the announcement lasts for one cycle.
int hx_amg_level0_diag(hx_amg* h, const hx_c64** dinv, hx_c64** x1, float* a0) {
  HX_REQUIRE(h && dinv && x1 && a0, "bad argument");
  *dinv = nullptr; *x1 = nullptr; *a0 = 0.f;
  if (!h->single || h->sharded || h->graph_state >= 0 || h->pre < 1 || h->S.L.size() < 2 || HX_OPT_INT("amg_fold_diag", 1) == 0) return HX_OK;
  LevelT<float>& L0 = h->S.L[0];
  if (!L0.dinv || !L0.x) return HX_OK;
  *dinv = (const hx_c64*)L0.dinv; *x1 = (hx_c64*)L0.x;
  *a0 = (float)(L0.rho > 0.0 ? cheb_coeffs(L0.rho).a[0] : h->omega);
  return HX_OK;
}
void hx_amg_x1_ready(hx_amg* h, bool ready) { if (h) h->x1_ready0 = ready; }

int hx_amg_vcycle_c64(hx_amg* h, hx_stream s) {
  HX_REQUIRE(h && h->single && h->S.coarse_inv.rp && h->S.L[0].n > 0, "complex64 hierarchy not set");
  LevelT<float>& L0 = h->S.L[0];
  const bool x1_ready = h->x1_ready0 && !h->sharded && h->graph_state < 0;
  h->x1_ready0 = false;
  // on level 0 the residual buffer L0.r is consumed by the restriction before the last post-sweep writes it
  if (h->graph_state == 0 && !h->sharded) {                      // capture the cycle once (its buffers never change)
    bool ok = true;
    if (!h->gstream && (ok = hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) == hipSuccess))
      hx_pool_register_stream(h->gstream, true);
    if (ok && !h->ev_in) ok =
                          hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) == hipSuccess;
    hipGraph_t g = nullptr;
    if (ok && hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int rcap = vcycle<float>(h, h->S, 0, L0.b, L0.r, (hx_stream)h->gstream);
      const hipError_t e = hipStreamEndCapture(h->gstream, &g);
      ok = rcap == HX_OK && e == hipSuccess && g && hipGraphInstantiate(&h->vexec, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
    } else ok = false;
    (void)hipGetLastError();
    h->graph_state = ok ? 1 : -1;                                // any failure: fall back to eager launches for good
  }
  if (h->graph_state == 1 && !h->sharded) {
    const bool hop = HX_OPT_SET("amg_graph_private_stream");     // (measured variant)
    if (hop) {
      HX_HIP(hipEventRecord(h->ev_in, hx_s(s)));                 // order the graph after the producer on the caller's stream
      HX_HIP(hipStreamWaitEvent(h->gstream, h->ev_in, 0));
      HX_HIP(hipGraphLaunch(h->vexec, h->gstream));
      HX_HIP(hipEventRecord(h->ev_out, h->gstream));
      HX_HIP(hipStreamWaitEvent(hx_s(s), h->ev_out, 0));         // ... and everything that follows after the graph
    } else {
      HX_HIP(hipGraphLaunch(h->vexec, hx_s(s)));                 // replay in the caller's stream: plain stream order
    }
    return HX_OK;
  }
  return vcycle<float>(h, h->S, 0, L0.b, L0.r, s, x1_ready);
}

// ---- the complex64 V-cycle on a BLOCK of hx_block_width() interleaved right-hand sides (single GPU): the same cycle as
// vcycle<float> -- same smoother polynomials, same transfers, same coarse inverse -- with every pass shared by the block.
static int vcycle_blk(hx_amg* h, int l, const float* b, float* xout, hx_stream s) {
  HierT<float>& H = h->S;
  LevelT<float>& L = H.L[l];
  int rc;
  if (l == (int)H.L.size() - 1)
    return hx_csr_spmm_c(H.coarse_inv.rows, H.coarse_inv.nnz, H.coarse_inv.rp, H.coarse_inv.ci, H.coarse_inv.va, H.coarse_inv.real_vals,
                         b, xout, 1.f, 0.f, nullptr, s);
  const bool cheb = L.rho > 0.0;
  const Cheb C = cheb ? cheb_coeffs(L.rho) : Cheb();
  const int extra = (int)HX_OPT_INT("amg_coarse_extra", 0);
  const int n_pre = h->pre + (l > 0 ? extra : 0), n_post = h->post + (l > 0 ? extra : 0);
  hx_amg::BlkBuf& W = h->blk[l];
  hx_amg::BlkBuf& Wc = h->blk[l + 1];
  float* bufs[3] = {W.x, W.x2, W.x3};
  auto other = [&](const float* a, const float* c) -> float* { for (float* q : bufs) if (q != a && q != c) return q; return nullptr; };
  auto smooth = [&](const float* xin, float* out, double w, const float* xprev, double mom) -> int {
    if (L.sell) return hx_sell_smooth_blk_c64(L.sell, (const hx_c64*)L.dinv, (const hx_c64*)b, (const hx_c64*)xin, (hx_c64*)out, (float)w,
                                              (const hx_c64*)xprev, (float)mom, s);
    return hx_csr_smooth_blk_c(L.n, L.K.nnz, L.K.rp, L.K.ci, L.K.va, L.dinv, b, xin, out, (float)w, xprev, (float)mom, s);
  };
  float* cur = W.x;
  float* prev = nullptr;
  LevelT<float>& Lc = H.L[l + 1];
  if (n_pre == 0) {
    if ((rc = hx_csr_spmm_c(L.R.rows, L.R.nnz, L.R.rp, L.R.ci, L.R.va, L.R.real_vals, b, Wc.b, 1.f, 0.f, nullptr, s))) return rc;
  } else {
    if ((rc = hx_cdiag_scale_blk(L.n, (float)(cheb ? C.a[0] : h->omega), L.dinv, b, cur, s))) return rc;
    for (int k = 1; k < n_pre; ++k) {
      float* nxt = other(cur, prev);
      if ((rc = cheb ? smooth(cur, nxt, C.a[k < 4 ? k : 3], prev, C.c[k < 4 ? k : 3]) : smooth(cur, nxt, h->omega, nullptr, 0.0))) return rc;
      prev = cur; cur = nxt;
    }
    if (L.sell) { if ((rc = hx_sell_spmm_c64(L.sell, (const hx_c64*)cur, (hx_c64*)W.r, -1.f, 1.f, (const hx_c64*)b, s))) return rc; }
    else if ((rc = hx_csr_spmm_c(L.n, L.K.nnz, L.K.rp, L.K.ci, L.K.va, false, cur, W.r, -1.f, 1.f, b, s))) return rc;
    if ((rc = hx_csr_spmm_c(L.R.rows, L.R.nnz, L.R.rp, L.R.ci, L.R.va, L.R.real_vals, W.r, Wc.b, 1.f, 0.f, nullptr, s))) return rc;
  }
  float* ec = Wc.r;
  if ((rc = vcycle_blk(h, l + 1, Wc.b, ec, s))) return rc;
  (void)Lc;
  if ((rc = hx_csr_spmm_c(L.P.rows, L.P.nnz, L.P.rp, L.P.ci, L.P.va, L.P.real_vals, ec, cur, 1.f, n_pre == 0 ? 0.f : 1.f, nullptr, s))) return rc;
  prev = nullptr;
  for (int k = 0; k < n_post; ++k) {
    float* dst = (k == n_post - 1) ? xout : other(cur, prev);
    if ((rc = cheb ? smooth(cur, dst, C.a[k < 4 ? k : 3], k ? prev : nullptr, k ? C.c[k < 4 ? k : 3] : 0.0)
                   : smooth(cur, dst, h->omega, nullptr, 0.0))) return rc;
    prev = cur; cur = dst;
  }
  if (n_post == 0)
    HX_HIP(hipMemcpyAsync(xout, cur, 2 * sizeof(float) * (size_t)L.n * hx_block_width(), hipMemcpyDeviceToDevice, hx_s(s)));
  return HX_OK;
}
// input / output buffers of the block cycle (level 0: b and r, as for the single-vector cycle); allocates the block work vectors
int hx_amg_vcycle_blk_buffers(hx_amg* h, hx_c64** b, hx_c64** z) {
  HX_REQUIRE(h && b && z && h->single && !h->sharded && h->S.coarse_inv.rp && !h->S.L.empty() && h->S.L[0].n > 0,
             "the block cycle needs the complex64 hierarchy on one GPU");
  if (h->blk.empty()) {
    h->blk.resize(h->S.L.size());
    const size_t nb = (size_t)hx_block_width();
    for (size_t l = 0; l < h->S.L.size(); ++l) {
      const size_t bytes = 2 * sizeof(float) * (size_t)h->S.L[l].n * nb;
      hx_amg::BlkBuf& W = h->blk[l];
      HX_HIP(hx_malloc(&W.b, bytes));
      HX_HIP(hx_malloc(&W.r, bytes));
      if (l + 1 < h->S.L.size()) { HX_HIP(hx_malloc(&W.x, bytes)); HX_HIP(hx_malloc(&W.x2, bytes)); HX_HIP(hx_malloc(&W.x3, bytes)); }
    }
  }
  *b = (hx_c64*)h->blk[0].b;
  *z = (hx_c64*)h->blk[0].r;
  return HX_OK;
}
int hx_amg_vcycle_blk(hx_amg* h, hx_stream s) {
  HX_REQUIRE(h && !h->blk.empty(), "hx_amg_vcycle_blk_buffers first");
  return vcycle_blk(h, 0, h->blk[0].b, h->blk[0].r, s);
}

extern "C" int hx_amg_vcycle(hx_amg* h, const hx_c128* b, hx_c128* x, hx_stream s) {
  HX_REQUIRE(h && b && x && b != x && !h->D.L.empty(), "bad argument");
  if (h->single) {
    HX_REQUIRE(h->S.coarse_inv.rp && h->S.L[0].n > 0, "complex64 hierarchy not set");
    LevelT<float>& L0 = h->S.L[0];
    int rc;
    h->x1_ready0 = false;                                         // (this entry casts its own input: nobody wrote x1)
    if ((rc = hx_cast_z2c(L0.n, b, L0.b, s))) return rc;          // narrow the right-hand side
    if ((rc = hx_amg_vcycle_c64(h, s))) return rc;
    return hx_cast_c2z(L0.n, L0.r, x, s);                         // widen the result
  }
  HX_REQUIRE(h->D.coarse_inv.rp && h->D.L[0].n > 0, "coarse solver not set");
  return vcycle<double>(h, h->D, 0, (const double*)b, (double*)x, s);
}

// ------------------------------------------------------------------------------------------------ FGMRES
// Flexible right-preconditioned restarted GMRES on K x = b (x0 = 0); CGS2 orthogonalisation with the fused panel
// kernels.  workspace: (2*restart + 2) * n complex numbers + hx_panel_dot_ws(restart + 1) + (restart + 2).
extern "C" size_t hx_gmres_ws(int64_t n, int restart) {      // (room for up to 32 recycled rows in the dot workspace)
  return (size_t)(2 * restart + 2) * (size_t)n + hx_panel_dot_ws(restart + 1 + 32) + (size_t)restart + 2 + 32;
}
// the same for a handle with the recycling workspace (hx_amg_enable_recycling): the Arnoldi vectors then live in the handle's
// [C; V] panel and the workspace holds the Z panel, the residual and the reduction scratch only -- (restart + 1) x n less
extern "C" size_t hx_gmres_ws_recycling(int64_t n, int restart) {
  return (size_t)(restart + 1) * (size_t)n + hx_panel_dot_ws(restart + 1 + 32) + (size_t)restart + 2 + 32;
}


// xe / hbuf: row-sharded mode only (xe: owned + ghost staging vector for the K products, hbuf: inner products, all-reduced);
// n is then the OWNED row count.
//
// Subspace recycling (GCRO, fixed operator): when M carries k recycled pairs (U, C) with K U = C, C^H C = I, every
// cycle starts from the Galerkin guess x += U C^H r, the Arnoldi vectors are kept orthogonal to C as well (C simply
// forms the first k rows of the Gram-Schmidt panel) and the update becomes x += Z y - U (B y), B = C^H K Z.  All
// solves of one fixed-point call share K, so one harvest (harmonic Ritz vectors of the first solve) serves them all.
static int gmres_impl(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                      const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it,
                      int restart, hx_c128* ws, hx_c128* xe, hx_c128* hbuf, int* h_iters, double* h_relres, hx_stream s) {
  HX_REQUIRE(n > 0 && rowptr && col && val && b && x && ws && restart >= 1 && restart <= 120 && max_it >= 1, "bad argument");
  const bool sharded = xe != nullptr;
  HX_REQUIRE(!sharded || (M && M->sharded && hbuf), "row-sharded GMRES needs a sharded preconditioner handle");
  Recycle* rec = (M && M->rec.kmax > 0) ? &M->rec : nullptr;
  HX_REQUIRE(!rec || (rec->n == n && rec->restart == restart), "recycling workspace was sized for another n / restart");
  const int kc = rec ? rec->k : 0;
  auto Kx = [&](const hx_c128* in, hx_c128* out, const double* a, const double* bt) -> int {
    if (sharded) {   // stage the owned block, exchange the ghost block, multiply the local rows
      HX_HIP(hipMemcpyAsync(xe, in, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      if (Ksell && M->halo_begin && M->halo_end && hx_sell_has_parts(Ksell)) {      // interior slices while the ghost entries travel
        hx_stream side = nullptr;
        if (M->halo_begin(M->ctx, 0, xe, 16, s, &side) != 0) return hx_fail(HX_ERR_COMM, "hx_gmres", "halo exchange (begin) failed");
        const int rcx = hx_sell_spmv_z_part(Ksell, 1, xe, out, a, bt, side);
        if (M->halo_end(M->ctx, 0, s) != 0) return hx_fail(HX_ERR_COMM, "hx_gmres", "halo exchange (end) failed");
        if (rcx) return rcx;
        return hx_sell_spmv_z_part(Ksell, 2, xe, out, a, bt, s);
      }
      if (M->halo(M->ctx, 0, xe, 16, s) != 0) return hx_fail(HX_ERR_COMM, "hx_gmres", "halo exchange callback failed");
      in = xe;
    }
    if (Ksell) return hx_sell_spmv_z(Ksell, in, out, a, bt, s);       // same operator, SELL-64 layout
    return hx_csr_spmv_z(n, nnz, rowptr, col, val, in, out, a, bt, 0, s);
  };
  // Gram-Schmidt panel: rows [0, kc) = recycled C, rows kc + j = v_j.  Without recycling it lives in ws as before.
  hx_c128* Pn = rec ? rec->CV : ws;
  hx_c128* V = Pn + (size_t)kc * n;                  // (restart+1) x n, row j = v_j
  hx_c128* Z = rec ? ws : ws + (size_t)(restart + 1) * n;       // restart x n,   row j = M^-1 v_j  (hx_gmres_ws / _recycling)
  hx_c128* r = Z + (size_t)restart * n;
  hx_c128* part = r + n;
  const int kmax = rec ? rec->kmax : 0;
  hx_c128* hdev = sharded ? hbuf : part + hx_panel_dot_ws(restart + 1 + kmax);
  std::vector<cd> hh(restart + 2 + kmax);
  std::vector<std::vector<cd>> H(restart + 1, std::vector<cd>(restart, cd(0)));
  std::vector<std::vector<cd>> B(kc > 0 ? kc : 1, std::vector<cd>(restart, cd(0)));
  std::vector<cd> cs(restart), sn(restart), g(restart + 1);
  // out[0..k) = panel rows [0, k)^H w, out[k] = w^H w   (all-reduced in sharded mode).  With the recycling workspace the
  // inner products read the complex64 copy of the panel; the subtractions always use the complex128 rows.
  float* Pf = rec ? rec->CVf : nullptr;
  auto mirror = [&](const hx_c128* row) -> int {      // refresh the complex64 copy of one finished panel row
    if (!Pf) return HX_OK;
    return hx_cast_z2c(n, row, Pf + 2 * (size_t)(row - Pn), s);
  };
  auto dots = [&](const hx_c128* panel, int k, const hx_c128* w) -> int {
    int rc = (Pf && k > 0) ? hx_zpanel_dot_c64(n, k, (const hx_c64*)Pf, n, w, hdev, part, s)
                           : hx_zpanel_dot(n, k, panel, n, w, hdev, part, s);
    if (rc) return rc;
    if (sharded && M->allreduce(M->ctx, hdev, 2 * (int64_t)(k + 1), 1, 0, s) != 0)
      return hx_fail(HX_ERR_COMM, "hx_gmres", "all-reduce callback failed");
    HX_HIP(hipMemcpyAsync(hh.data(), hdev, sizeof(cd) * (size_t)(k + 1), hipMemcpyDeviceToHost, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    return HX_OK;
  };
  int rc, it = 0, m_last = 0;
  const bool trace = HX_OPT_SET("gmres_trace");     // convergence history on stderr (diagnostics)
  if (M) { M->last_backward = -1.0; M->x0_next = false; }     // (initial guesses are a feature of hx_cocg only)
  HX_HIP(hipMemsetAsync(x, 0, sizeof(hx_c128) * (size_t)n, hx_s(s)));
  if ((rc = dots(Pn, 0, b))) return rc;
  const double bnorm = std::sqrt(hh[0].real());
  if (bnorm == 0.0) { if (h_iters) *h_iters = 0; if (h_relres) *h_relres = 0.0; return HX_OK; }
  double relres = 1.0;
  bool first = true;
  std::vector<cd> tmpc(kmax + restart + 2);
  auto upload = [&](const std::vector<cd>& v, int cnt) -> int {     // small host vector -> hdev (sync: v may be reused)
    HX_HIP(hipMemcpyAsync(hdev, v.data(), sizeof(cd) * (size_t)cnt, hipMemcpyHostToDevice, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    return HX_OK;
  };
  while (it < max_it) {
    // r = b - K x   (true residual, double precision)
    HX_HIP(hipMemcpyAsync(r, b, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
    if (!first) { if ((rc = Kx(x, r, MONE, ONE))) return rc; }
    first = false;
    if (kc > 0) {   // Galerkin step on the recycled space: t = C^H r, x += U t, r -= C t
      if ((rc = dots(Pn, kc, r))) return rc;
      if ((rc = hx_zpanel_sub(n, kc, Pn, n, hdev, r, s))) return rc;
      for (int i = 0; i < kc; ++i) tmpc[i] = -hh[i];
      if ((rc = upload(tmpc, kc))) return rc;
      if ((rc = hx_zpanel_sub(n, kc, rec->U, n, hdev, x, s))) return rc;
    }
    if ((rc = dots(Pn, 0, r))) return rc;
    const double beta = std::sqrt(hh[0].real());
    relres = beta / bnorm;
    if (relres <= rtol) break;
    HX_HIP(hipMemcpyAsync(V, r, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
    const double ib[2] = {1.0 / beta, 0.0};
    if ((rc = hx_zscal(n, ib, V, s))) return rc;
    if ((rc = mirror(V))) return rc;
    std::fill(g.begin(), g.end(), cd(0));
    g[0] = beta;
    int j = 0;
    for (; j < restart && it < max_it;) {
      hx_c128* vj = V + (size_t)j * n;
      hx_c128* zj = M ? Z + (size_t)j * n : vj;
      hx_c128* w = V + (size_t)(j + 1) * n;
      if (M) { if ((rc = hx_amg_vcycle(M, vj, zj, s))) return rc; }
      if ((rc = Kx(zj, w, ONE, ZERO))) return rc;
      const int k = kc + j + 1;                        // panel rows to orthogonalise against: C and v_0..v_j
      // classical Gram-Schmidt, always twice ("twice is enough"): one pass loses orthogonality like eps*kappa^2 of
      // the Krylov matrix, which was measured to stall GMRES here (55 instead of 29 iterations per solve).
      if ((rc = dots(Pn, k, w))) return rc;
      if ((rc = hx_zpanel_sub(n, k, Pn, n, hdev, w, s))) return rc;
      for (int i = 0; i < kc; ++i) B[i][j] = hh[i];
      for (int i = 0; i <= j; ++i) H[i][j] = hh[kc + i];
      if ((rc = dots(Pn, k, w))) return rc;
      double h2sq = 0.0;
      for (int i = 0; i < k; ++i) h2sq += std::norm(hh[i]);
      // second (correction) pass: its coefficients are ~1e-7 |w|, so the update may read the complex64 mirror too
      // (relation error 6e-8 |h2|); a large correction falls back to the complex128 panel
      if (Pf && h2sq <= 1e-10 * hh[k].real()) { if ((rc = hx_zpanel_sub_c64(n, k, (const hx_c64*)Pf, n, hdev, w, s))) return rc; }
      else if ((rc = hx_zpanel_sub(n, k, Pn, n, hdev, w, s))) return rc;
      double n2 = hh[k].real() - h2sq;
      for (int i = 0; i < kc; ++i) B[i][j] += hh[i];
      for (int i = 0; i <= j; ++i) H[i][j] += hh[kc + i];
      if (n2 <= 1e-8 * hh[k].real()) { if ((rc = dots(Pn, 0, w))) return rc; n2 = hh[0].real(); }
      const double hn = std::sqrt(n2 > 0 ? n2 : 0.0);
      H[j + 1][j] = hn;
      if (rec && kc == 0) {                            // raw Hessenberg column of the current cycle, kept for the harvest
        for (int i = 0; i <= j + 1; ++i) rec->Hraw[(size_t)j * (restart + 1) + i] = H[i][j];
      }
      if (hn > 0) { const double ih[2] = {1.0 / hn, 0.0}; if ((rc = hx_zscal(n, ih, w, s))) return rc; }
      if ((rc = mirror(w))) return rc;
      // complex Givens rotations
      for (int i = 0; i < j; ++i) {
        const cd t = std::conj(cs[i]) * H[i][j] + std::conj(sn[i]) * H[i + 1][j];
        H[i + 1][j] = -sn[i] * H[i][j] + cs[i] * H[i + 1][j];
        H[i][j] = t;
      }
      const double denom = std::sqrt(std::norm(H[j][j]) + std::norm(H[j + 1][j]));
      if (denom == 0.0) return hx_fail(HX_ERR_BREAKDOWN, __func__, "zero Hessenberg column");
      cs[j] = H[j][j] / denom;
      sn[j] = H[j + 1][j] / denom;
      H[j][j] = std::conj(cs[j]) * H[j][j] + std::conj(sn[j]) * H[j + 1][j];
      H[j + 1][j] = 0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = std::conj(cs[j]) * g[j];
      relres = std::abs(g[j + 1]) / bnorm;
      if (trace) fprintf(stderr, "[hx_gmres] it %3d  relres %.3e\n", it + 1, relres);
      ++j; ++it;
      if (relres <= rtol || hn == 0.0) break;
    }
    m_last = j;
    // y = H^-1 g ;  x += Z y   (flexible: the stored preconditioned directions, no extra preconditioner call)
    std::vector<cd> y(j);
    for (int i = j - 1; i >= 0; --i) {
      cd t = g[i];
      for (int q = i + 1; q < j; ++q) t -= H[i][q] * y[q];
      y[i] = t / H[i][i];
    }
    for (int i = 0; i < j; ++i) tmpc[i] = -y[i];        // panel_sub computes x -= Z h
    if ((rc = upload(tmpc, j))) return rc;
    if ((rc = hx_zpanel_sub(n, j, M ? Z : V, n, hdev, x, s))) return rc;
    if (kc > 0) {                                       // ... - U (B y)
      for (int i = 0; i < kc; ++i) { cd t(0); for (int q = 0; q < j; ++q) t += B[i][q] * y[q]; tmpc[i] = t; }
      if ((rc = upload(tmpc, kc))) return rc;
      if ((rc = hx_zpanel_sub(n, kc, rec->U, n, hdev, x, s))) return rc;
    }
    if (relres <= rtol) {
      // accept only against the TRUE residual; otherwise keep iterating from the updated x
      HX_HIP(hipMemcpyAsync(r, b, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      if ((rc = Kx(x, r, MONE, ONE))) return rc;
      if ((rc = dots(Pn, 0, r))) return rc;
      relres = std::sqrt(hh[0].real()) / bnorm;
      if (relres <= 10 * rtol) break;
      // The Arnoldi estimate met the tolerance, the true residual did not: the attainable accuracy?  (A right-hand side
      // close to the eigenvector next to the shift has a solution 1e4..1e6 times larger than b, and eps |K| |x| then
      // exceeds rtol |b|; without this test such solves cycle until max_it.)  Accept a normwise backward error
      // |b - K x| / (|K|_inf |x| + |b|) at rounding level, as hx_cocg does.
      if (M) {
        if (M->normK_val != (const void*)val) {
          double local = 0.0;
          if ((rc = hx_csr_norm_inf(n, rowptr, val, hdev, &local, s))) return rc;
          if (sharded) {                       // maximum over the ranks' row blocks = |K|_inf of the whole operator
            const cd v(local, 0.0);
            HX_HIP(hipMemcpyAsync(hdev, &v, sizeof(cd), hipMemcpyHostToDevice, hx_s(s)));
            HX_HIP(hipStreamSynchronize(hx_s(s)));
            if (M->allreduce(M->ctx, hdev, 2, 1, 1, s) != 0) return hx_fail(HX_ERR_COMM, "hx_gmres", "all-reduce callback failed");
            HX_HIP(hipMemcpyAsync(hh.data(), hdev, sizeof(cd), hipMemcpyDeviceToHost, hx_s(s)));
            HX_HIP(hipStreamSynchronize(hx_s(s)));
            local = hh[0].real();
          }
          M->normK = local;
          M->normK_val = (const void*)val;
        }
        if ((rc = dots(Pn, 0, x))) return rc;
        const double xnorm = std::sqrt(hh[0].real());
        M->last_backward = relres * bnorm / (M->normK * xnorm + bnorm);
        if (trace) fprintf(stderr, "[hx_gmres] true relres %.3e, backward error %.3e\n", relres, M->last_backward);
        if (M->last_backward <= 10 * 2.220446049250313e-16) break;
      }
    }
  }
  // harvestable: a plain solve; the panels and Hraw describe its LAST cycle (K Z_m = V_{m+1} Hbar_m holds per cycle; a
  // restarted solve stagnates on exactly the modes worth recycling, so its last cycle is the richest one)
  if (rec) { rec->m_last = (kc == 0) ? m_last : 0; }
  if (h_iters) *h_iters = it;
  if (h_relres) *h_relres = relres;
  return HX_OK;
}

// ---- recycling: workspace, Hessenberg read-back, installation of the harvested pairs
extern "C" int hx_amg_enable_recycling(hx_amg* h, int64_t n, int k_max, int restart) {
  HX_REQUIRE(h && n > 0 && k_max >= 1 && k_max <= 32 && restart >= 2 && restart <= 120, "bad argument");
  Recycle& R = h->rec;
  if (R.kmax > 0) {          // called again: another panel size (the escalation ladder widens the Krylov space; the hierarchy stays)
    HX_REQUIRE(R.n == n, "recycling workspace was sized for another n");
    HX_HIP(hipStreamSynchronize(hx_last_stream.load(std::memory_order_relaxed)));
    (void)hx_free(R.U); (void)hx_free(R.CV); (void)hx_free(R.CVf); (void)hx_free(R.Uf); (void)hx_free(R.Cd); (void)hx_free(R.Einv);
    R.U = R.CV = R.Cd = R.Einv = nullptr; R.CVf = R.Uf = nullptr;
    R.kmax = 0;
    h->normK_val = nullptr;
  }
  HX_HIP(hx_malloc((void**)&R.U, sizeof(hx_c128) * (size_t)k_max * (size_t)n));
  HX_HIP(hx_malloc((void**)&R.CV, sizeof(hx_c128) * (size_t)(k_max + restart + 1) * (size_t)n));
  HX_HIP(hx_malloc((void**)&R.CVf, sizeof(hx_c64) * (size_t)(k_max + restart + 1) * (size_t)n));
  HX_HIP(hx_malloc((void**)&R.Uf, sizeof(hx_c64) * (size_t)k_max * (size_t)n));
  HX_HIP(hx_malloc((void**)&R.Cd, sizeof(hx_c128) * (size_t)k_max * (size_t)n));
  HX_HIP(hx_malloc((void**)&R.Einv, sizeof(hx_c128) * (size_t)k_max * (size_t)k_max));
  R.kmax = k_max; R.k = 0; R.n = n; R.restart = restart; R.m_last = 0;
  R.Hraw.assign((size_t)(restart + 1) * restart, cd(0));
  return HX_OK;
}
// (m+1) x m raw Hessenberg of the last cycle of the last solve, column-major with leading dimension m+1; *m = 0 when that
// solve is not harvestable (it already used recycled pairs).
extern "C" int hx_gmres_hessenberg(const hx_amg* h, int* m, hx_c128* H) {
  HX_REQUIRE(h && m, "bad argument");
  const Recycle& R = h->rec;
  *m = R.kmax > 0 ? R.m_last : 0;
  if (*m == 0 || !H) return HX_OK;
  for (int j = 0; j < *m; ++j)
    for (int i = 0; i <= *m; ++i) {
      const cd v = i <= j + 1 ? R.Hraw[(size_t)j * (R.restart + 1) + i] : cd(0);
      ((cd*)H)[(size_t)j * (*m + 1) + i] = v;
    }
  return HX_OK;
}
// U = Z_m Ucoef, C = V_{m+1} Ccoef (coefficients column-major: m x k and (m+1) x k, on the HOST), from the panels the
// last hx_gmres call left in `ws` / the handle.  Must directly follow that call (same ws).
extern "C" int hx_amg_set_recycle(hx_amg* h, int k, const hx_c128* Ucoef, const hx_c128* Ccoef, hx_c128* ws, hx_stream s) {
  HX_REQUIRE(h && Ucoef && Ccoef && ws, "bad argument");
  h->normK_val = nullptr;      // (pairs are harvested once per operator: re-measure |K|_inf with them)
  Recycle& R = h->rec;
  const int m = R.m_last;
  HX_REQUIRE(R.kmax > 0 && k >= 1 && k <= R.kmax && m >= k, "nothing to harvest from");
  const int64_t n = R.n;
  hx_c128* Z = ws;                                    // (layout of a handle with the recycling workspace: hx_gmres_ws_recycling)
  hx_c128* coef = nullptr;
  HX_HIP(hx_malloc((void**)&coef, sizeof(hx_c128) * (size_t)(m + 1) * k));
  int rc;
  HX_HIP(hipMemcpyAsync(coef, Ucoef, sizeof(hx_c128) * (size_t)m * k, hipMemcpyHostToDevice, hx_s(s)));
  rc = hx_zpanel_mul(n, m, k, Z, n, coef, R.U, n, s);
  if (!rc) rc = hx_cast_z2c((int64_t)k * n, R.U, R.Uf, s);
  HX_HIP(hipStreamSynchronize(hx_s(s)));
  if (!rc) {
    HX_HIP(hipMemcpyAsync(coef, Ccoef, sizeof(hx_c128) * (size_t)(m + 1) * k, hipMemcpyHostToDevice, hx_s(s)));
    // V (rows 0..m of CV, since k was 0) -> C: staged through the Z panel, which is free now
    rc = hx_zpanel_mul(n, m + 1, k, R.CV, n, coef, Z, n, s);
    if (!rc) HX_HIP(hipMemcpyAsync(R.CV, Z, sizeof(hx_c128) * (size_t)k * n, hipMemcpyDeviceToDevice, hx_s(s)));
    if (!rc) rc = hx_cast_z2c((int64_t)k * n, R.CV, R.CVf, s);
    HX_HIP(hipStreamSynchronize(hx_s(s)));
  }
  (void)hx_free(coef);
  if (rc) return rc;
  R.k = k; R.m_last = 0; R.have_E = false;
  return HX_OK;
}

extern "C" int hx_gmres(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                        const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it,
                        int restart, hx_c128* ws, int* h_iters, double* h_relres, hx_stream s) {
  HX_REQUIRE(!M || !M->sharded, "use hx_gmres_sharded with a row-sharded preconditioner");
  return gmres_impl(M, n, nnz, rowptr, col, val, Ksell, b, x, rtol, max_it, restart, ws, nullptr, nullptr, h_iters,
                    h_relres, s);
}

extern "C" int hx_gmres_sharded(hx_amg* M, int32_t n_own, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol,
                                int max_it, int restart, hx_c128* ws, hx_c128* xe, hx_c128* hbuf, int* h_iters,
                                double* h_relres, hx_stream s) {
  HX_REQUIRE(M && M->sharded && xe && hbuf, "bad argument");
  return gmres_impl(M, n_own, nnz, rowptr, col, val, Ksell, b, x, rtol, max_it, restart, ws, xe, hbuf, h_iters,
                    h_relres, s);
}

// ------------------------------------------------------------------------------------------------ aggregation (host)
// Greedy (Vanek) aggregation on the strength graph: n nodes, CSR (rp, ci) of strong off-diagonal couplings with
// weights w (|a_ij|).  Writes agg[i] in [0, n_agg).  Pure integer/graph work, one-off per hierarchy level.
template <typename W>
static int host_aggregate(int32_t n, const int32_t* rp, const int32_t* ci, const W* w, int32_t* agg, int32_t* n_agg) {
  if (!(n > 0 && rp && agg && n_agg)) return hx_fail(HX_ERR_ARG, "hx_host_aggregate", "bad argument");
  // columns >= n are ghost columns of a row-sharded graph (owned rows over [owned | ghost]): they are no neighbours here, so
  // aggregates never cross a rank boundary (uncoupled aggregation) and every rank aggregates its rows on its own
  for (int32_t i = 0; i < n; ++i) agg[i] = -1;
  int32_t na = 0;
  // pass 1: roots whose whole strong neighbourhood is free (a stored diagonal entry is not a neighbour)
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    bool free_nb = true, has_nb = false;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      if (ci[p] == i || ci[p] >= n) continue;
      has_nb = true;
      if (agg[ci[p]] != -1) { free_nb = false; break; }
    }
    if (!free_nb || !has_nb) continue;
    agg[i] = na;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] < n) agg[ci[p]] = na;
    ++na;
  }
  // pass 2: attach leftovers to the most strongly connected pass-1 aggregate
  std::vector<int32_t> pass1(agg, agg + n);
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    double best = -1.0; int32_t tgt = -1;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      if (ci[p] == i || ci[p] >= n) continue;
      const int32_t a = pass1[ci[p]];
      const double wp = std::fabs((double)w[p]);
      if (a >= 0 && wp > best) { best = wp; tgt = a; }
    }
    if (tgt >= 0) agg[i] = tgt;
  }
  // pass 3: whatever is left forms aggregates with its free neighbours (isolated nodes become singletons)
  for (int32_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    agg[i] = na;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] < n && agg[ci[p]] == -1) agg[ci[p]] = na;
    ++na;
  }
  *n_agg = na;
  return HX_OK;
}


extern "C" int hx_host_aggregate(int32_t n, const int32_t* rp, const int32_t* ci, const double* w, int32_t* agg,
                                 int32_t* n_agg) {
  return host_aggregate<double>(n, rp, ci, w, agg, n_agg);
}
// the same with float32 coupling strengths (they only rank the neighbouring aggregates of a leftover node: half the
// device-to-host bytes when the graph comes from the device set-up)
extern "C" int hx_host_aggregate_f32(int32_t n, const int32_t* rp, const int32_t* ci, const float* w, int32_t* agg,
                                     int32_t* n_agg) {
  return host_aggregate<float>(n, rp, ci, w, agg, n_agg);
}

// ------------------------------------------------------------------------------------------------ SA set-up passes (host)
// Row-parallel helper for the host passes: contiguous row blocks, one std::thread each (HX_HOST_THREADS, default
// min(16, hardware threads)); every row is produced by exactly one thread, so results do not depend on the count.
static int host_threads() {
  const int v = (int)HX_OPT_INT("host_threads", (long)std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
  return std::max(1, std::min(v, 64));
}
template <typename F> static void parallel_rows(int32_t n, F&& fn) {   // fn(thread, row_begin, row_end)
  int nt = host_threads();
  if (n < 50000) nt = 1;
  std::vector<std::thread> pool;
  auto blk = [&](int t) { return (int32_t)((int64_t)n * t / nt); };
  for (int t = 1; t < nt; ++t) pool.emplace_back([&, t] { fn(t, blk(t), blk(t + 1)); });
  fn(0, blk(0), blk(1));
  for (auto& th : pool) th.join();
}

// One sweep over the Laplacian-like matrix L (CSR, real values, positive diagonal): keep the diagonal and the strong
// off-diagonal couplings |l_ij| >= theta*sqrt(l_ii l_jj); weak couplings are lumped onto the diagonal.  Output is the
// filtered matrix Lf in compact CSR (diagonal entry included).  Call once with out_ci == NULL to get the row pointer
// (count pass), then again to fill.  Replaces ~15 NumPy passes over the nnz arrays.
extern "C" int hx_host_sa_filter(int32_t n, const int32_t* rp, const int32_t* ci, const double* lv_in, int stride,
                                 double sign, double theta, int32_t* out_rp, int32_t* out_ci, double* out_v) {
  HX_REQUIRE(n > 0 && rp && ci && lv_in && out_rp && stride >= 1, "bad argument");
  HX_REQUIRE(out_ci == nullptr || out_v != nullptr, "bad argument");
  // entry p of L is sign * lv_in[p * stride]: stride 2 reads the real parts of an interleaved complex128 array in place
  struct View { const double* p; int st; double sg; double operator[](int64_t i) const { return sg * p[i * st]; } };
  const View lv{lv_in, stride, sign};
  std::vector<double> sd(n);
  parallel_rows(n, [&](int, int32_t i0, int32_t i1) {
    for (int32_t i = i0; i < i1; ++i) {
      double d = 0.0;
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] == i) d = lv[p];
      d = std::fabs(d);
      sd[i] = std::sqrt(d == 0.0 ? 1.0 : d);
    }
  });
  if (!out_ci) {
    parallel_rows(n, [&](int, int32_t i0, int32_t i1) {
      for (int32_t i = i0; i < i1; ++i) {
        int32_t cnt = 1;                                  // the diagonal always stays
        const double ti = theta * sd[i];
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
          const int32_t j = ci[p];
          if (j != i && std::fabs(lv[p]) >= ti * sd[j]) ++cnt;
        }
        out_rp[i + 1] = cnt;
      }
    });
    out_rp[0] = 0;
    for (int32_t i = 0; i < n; ++i) out_rp[i + 1] += out_rp[i];
    return HX_OK;
  }
  parallel_rows(n, [&](int, int32_t i0, int32_t i1) {
    for (int32_t i = i0; i < i1; ++i) {
      const double ti = theta * sd[i];
      double diag = 0.0, lump = 0.0;
      int32_t q = out_rp[i], qd = -1;
      bool placed = false;
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
        const int32_t j = ci[p];
        if (j == i) { diag = lv[p]; qd = q; out_ci[q] = i; ++q; placed = true; continue; }
        if (!placed && j > i) { qd = q; out_ci[q] = i; ++q; placed = true; }      // keep columns sorted if no diagonal stored
        if (std::fabs(lv[p]) >= ti * sd[j]) { out_ci[q] = j; out_v[q] = lv[p]; ++q; }
        else lump += lv[p];
      }
      if (!placed) { qd = q; out_ci[q] = i; ++q; }
      out_v[qd] = diag + lump;
    }
  });
  return HX_OK;
}

// Smoothed prolongator P = (I - omega Df^-1 Lf) P0 with the piecewise-constant tentative P0[i, agg(i)] = 1/sqrt(|agg|).
// Rows have at most (row length of Lf) distinct aggregates; columns are merged and sorted per row.
// Count pass with out_ci == NULL.
extern "C" int hx_host_sa_prolongator(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf,
                                      const int32_t* agg, const double* inv_sqrt_cnt, double omega,
                                      int32_t* out_rp, int32_t* out_ci, double* out_v) {
  HX_REQUIRE(n > 0 && rp && ci && lf && agg && inv_sqrt_cnt && out_rp, "bad argument");
  HX_REQUIRE(out_ci == nullptr || out_v != nullptr, "bad argument");
  parallel_rows(n, [&](int, int32_t i0, int32_t i1) {
    std::vector<int32_t> cols;
    std::vector<double> vals;
    for (int32_t i = i0; i < i1; ++i) {
      cols.clear(); vals.clear();
      double dii = 1.0;
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] == i) dii = lf[p];
      if (dii == 0.0) dii = 1.0;
      auto add = [&](int32_t c, double v) {
        size_t k = 0;
        while (k < cols.size() && cols[k] < c) ++k;
        if (k < cols.size() && cols[k] == c) { vals[k] += v; return; }
        cols.insert(cols.begin() + k, c);
        vals.insert(vals.begin() + k, v);
      };
      add(agg[i], inv_sqrt_cnt[agg[i]]);
      const double w = omega / dii;
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
        const int32_t a = agg[ci[p]];
        add(a, -w * lf[p] * inv_sqrt_cnt[a]);
      }
      if (!out_ci) { out_rp[i + 1] = (int32_t)cols.size(); continue; }
      int32_t q = out_rp[i];
      for (size_t k = 0; k < cols.size(); ++k, ++q) { out_ci[q] = cols[k]; out_v[q] = vals[k]; }
    }
  });
  if (!out_ci) {
    out_rp[0] = 0;
    for (int32_t i = 0; i < n; ++i) out_rp[i + 1] += out_rp[i];
  }
  return HX_OK;
}

// Diagonal of a complex CSR matrix (0 where no diagonal entry is stored); row-parallel.  SciPy's csr_diagonal is a
// sequential pass (0.1 s at 5 M rows).
extern "C" int hx_host_csr_diagonal(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, hx_c128* out) {
  HX_REQUIRE(n > 0 && rp && ci && val && out, "bad argument");
  parallel_rows(n, [&](int, int32_t i0, int32_t i1) {
    for (int32_t i = i0; i < i1; ++i) {
      hx_c128 d = {0.0, 0.0};
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) if (ci[p] == i) { d = val[p]; break; }
      out[i] = d;
    }
  });
  return HX_OK;
}

// Power iteration for rho(Df^-1 Lf) (Lf CSR with positive diagonal): v <- Df^-1 Lf v / ||.||, `iters` times from the
// caller's start vector v (overwritten); returns the last norm ratio.  Row-parallel matvec; the norm is summed per
// row block and combined in block order.
extern "C" int hx_host_power_rho(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, double* v,
                                 int iters, double* rho) {
  HX_REQUIRE(n > 0 && rp && ci && lf && v && rho && iters >= 1, "bad argument");
  std::vector<double> w(n), part(64, 0.0);
  double nrm = 0.0;
  for (int32_t i = 0; i < n; ++i) nrm += v[i] * v[i];
  nrm = std::sqrt(nrm);
  HX_REQUIRE(nrm > 0.0, "zero start vector");
  for (int32_t i = 0; i < n; ++i) v[i] /= nrm;
  double lam = 1.0;
  for (int it = 0; it < iters; ++it) {
    std::fill(part.begin(), part.end(), 0.0);
    parallel_rows(n, [&](int t, int32_t i0, int32_t i1) {
      double acc = 0.0;
      for (int32_t i = i0; i < i1; ++i) {
        double d = 1.0, sum = 0.0;
        for (int32_t p = rp[i]; p < rp[i + 1]; ++p) { if (ci[p] == i) d = lf[p]; sum += lf[p] * v[ci[p]]; }
        w[i] = sum / d;
        acc += w[i] * w[i];
      }
      part[t] = acc;
    });
    double s2 = 0.0;
    for (double q : part) s2 += q;
    lam = std::sqrt(s2);
    if (lam == 0.0) break;
    const double inv = 1.0 / lam;
    parallel_rows(n, [&](int, int32_t i0, int32_t i1) { for (int32_t i = i0; i < i1; ++i) v[i] = w[i] * inv; });
  }
  *rho = lam;
  return HX_OK;
}


// Galerkin coarse operator Kc = R K P in ONE fused pass over the coarse rows (host, multi-threaded):
//   Kc[I, J] = sum_{i in R(I)} sum_{j in K(i)} sum_{J in P(j)} R_Ii K_ij P_jJ        (R, P real; K complex)
// Each thread owns a contiguous block of coarse rows and a private dense marker/accumulator, so no intermediate
// K P is ever materialised (SciPy's two SpGEMMs were ~55 % of the hierarchy set-up at 5 M DOF).
// Two calls: out_ci == NULL -> row pointer (count pass); then fill (columns sorted per row).
extern "C" int hx_host_galerkin_rap(int32_t n_c, const int32_t* Rrp, const int32_t* Rci, const double* Rv,
                                    const int32_t* Krp, const int32_t* Kci, const hx_c128* Kv,
                                    const int32_t* Prp, const int32_t* Pci, const double* Pv,
                                    int32_t* out_rp, int32_t* out_ci, hx_c128* out_v, int n_threads) {
  HX_REQUIRE(n_c > 0 && Rrp && Rci && Rv && Krp && Kci && Kv && Prp && Pci && Pv && out_rp, "bad argument");
  const bool fill = out_ci != nullptr;
  HX_REQUIRE(!fill || out_v, "bad argument");
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  if (n_threads > n_c) n_threads = n_c;
  std::vector<int32_t> counts(fill ? 0 : (size_t)n_c);
  auto work = [&](int t) {
    const int32_t I0 = (int32_t)((int64_t)n_c * t / n_threads), I1 = (int32_t)((int64_t)n_c * (t + 1) / n_threads);
    std::vector<int32_t> marker((size_t)n_c, -1);      // marker[J] = position of J in the current row's list
    std::vector<int32_t> cols;
    std::vector<cd> vals;
    std::vector<int32_t> order;
    for (int32_t I = I0; I < I1; ++I) {
      cols.clear(); vals.clear();
      for (int32_t a = Rrp[I]; a < Rrp[I + 1]; ++a) {
        const int32_t i = Rci[a];
        const double r = Rv[a];
        for (int32_t b = Krp[i]; b < Krp[i + 1]; ++b) {
          const int32_t j = Kci[b];
          const cd rk = r * cd(Kv[b].re, Kv[b].im);
          for (int32_t c = Prp[j]; c < Prp[j + 1]; ++c) {
            const int32_t J = Pci[c];
            int32_t pos = marker[J];
            if (pos < 0) { pos = (int32_t)cols.size(); marker[J] = pos; cols.push_back(J); if (fill) vals.push_back(cd(0)); }
            if (fill) vals[pos] += rk * Pv[c];
          }
        }
      }
      for (int32_t J : cols) marker[J] = -1;
      if (!fill) { counts[I] = (int32_t)cols.size(); continue; }
      order.resize(cols.size());
      for (size_t q = 0; q < cols.size(); ++q) order[q] = (int32_t)q;
      std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cols[x] < cols[y]; });
      int32_t o = out_rp[I];
      for (int32_t q : order) { out_ci[o] = cols[q]; out_v[o].re = vals[q].real(); out_v[o].im = vals[q].imag(); ++o; }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  if (!fill) {
    out_rp[0] = 0;
    for (int32_t I = 0; I < n_c; ++I) out_rp[I + 1] = out_rp[I] + counts[I];
  }
  return HX_OK;
}
