// Complex128 CSR SpMV, y = alpha*A*x + beta*y  (PETSc MatMult on the shift-invert Arnoldi path).
//
// HBM-bound: algorithmic traffic = nnz*(16 B value + 4 B column) + n*(4 B rowptr + 16 B x + 16 B y)
// (SURVEY.md section 8d).  Layout: LPR consecutive lanes cooperate on one row, so a 64-lane wave
// streams 64/LPR adjacent rows = one contiguous span of val/col (P1 rows are short and nearly
// uniform: 3 / 7 / ~15 nonzeros in 1D / 2D / 3D).  Values are read as one dwordx4 per lane; the x
// gather is served by L2 / Infinity Cache because FEM columns are local.  Grid-stride over rows with a
// capped grid (>> 256 workgroups) so every XCD's L2 sees a contiguous band of rows per workgroup.
#include "hx_common.h"

// Partial row product of one sub-wave lane: sum over p = s+sub, s+sub+LPR, ... of val[p] * x[col[p]].
// MV = type of the matrix values: V itself, or the real type of V for the (real) transfer operators of the hierarchy.
// complex64: same entry-per-lane sweep.  Measured and rejected on MI355X (5 M rows, round 1): lanes taking PAIRS of
// entries (one 16-byte value load + one 8-byte index load) ran at 3.1-3.4 TB/s against 3.9 TB/s for this form.
template <int LPR, typename V, typename MV>
__device__ __forceinline__ V row_dot(const int32_t* __restrict__ col, const MV* __restrict__ val,
                                     const V* __restrict__ x, int32_t s, int32_t e, int sub) {
  V acc = czero<V>();
  for (int32_t p = s + sub; p < e; p += LPR) zfma(acc, val[p], x[col[p]]);
  return acc;
}

// ROLE only separates kernel symbols in profiles: 0 = operator product on the Arnoldi / GMRES path (the roofline
// kernel), 1 = grid-transfer products (P, R) and coarse-grid solves inside the AMG V-cycle.
//
// Measured and rejected on MI355X at 1 M / 5 M rows (tools/run_spmv_modes.sh, round 1): (a) XCD-aware banding
// (workgroups with equal blockIdx % 8 sweeping a contiguous eighth of the rows) -8 %; (b) non-temporal loads of
// val/col -30 % (they split the 16-byte value load).  The plain grid-stride sweep below is the fastest form.
template <int LPR, int ROLE, typename V, typename MV = V>
__global__ void __launch_bounds__(256)
k_spmv(int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const MV* __restrict__ val,
       const V* __restrict__ x, V* y, const V* yin, V alpha, V beta, int beta_is_zero,
       const V* __restrict__ ep_d = nullptr, V* __restrict__ ep_x = nullptr, typename real_of<V>::type ep_a = 0) {
  constexpr int RPB = 256 / LPR;                       // rows per block per sweep
  const int sub = threadIdx.x % LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  // every lane of a wave runs the same number of sweeps => the sub-wave shuffles below are convergent
  const int64_t sweeps = ((int64_t)n + stride - 1) / stride;
  int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  for (int64_t it = 0; it < sweeps; ++it, row += stride) {
    V acc = czero<V>();
    if (row < n) acc = row_dot<LPR, V, MV>(col, val, x, rowptr[row], rowptr[row + 1], sub);
    if (LPR > 1) acc = zreduce_sub<LPR>(acc);
    if (row < n && sub == 0) {
      V r = zmul(alpha, acc);
      if (!beta_is_zero) zfma(r, beta, yin[row]);
      y[row] = r;
      // epilogue of a RESTRICTION (transfer role only): the first smoothing step of the coarse level from a zero start,
      // x1 = a0 D^-1 b, leaves with its right-hand side -- the arithmetic of k_diag_scale on the value just stored
      if constexpr (ROLE == 1) { if (ep_x) ep_x[row] = zscale(ep_a, zmul(ep_d[row], r)); }
    }
  }
}

// yin: the vector scaled by beta (y itself when NULL): y = alpha*A*x + beta*yin
template <typename V> struct SpmvEpilogue { const V* d = nullptr; V* x = nullptr; typename real_of<V>::type a = 0; };
template <int LPR, int ROLE, typename V, typename MV = V>
static int launch(int32_t n, const int32_t* rowptr, const int32_t* col, const MV* val, const V* x, V* y,
                  V alpha, V beta, hipStream_t s, const V* yin = nullptr, SpmvEpilogue<V> ep = SpmvEpilogue<V>()) {
  int64_t rows_per_block = 256 / LPR;
  int grid = hx_grid((int64_t)n, (int)rows_per_block, 256 * 16);
  k_spmv<LPR, ROLE, V, MV><<<grid, 256, 0, s>>>(n, rowptr, col, val, x, y, yin ? yin : y, alpha, beta, (beta.x == 0 && beta.y == 0),
                                                ep.d, ep.x, ep.a);
  HX_LAUNCH_CHECK("k_spmv");
  return HX_OK;
}

static inline int pick_lpr(int64_t nnz, int32_t n_rows) {
  const double avg = (double)nnz / (double)n_rows;
  return avg <= 2.0 ? 1 : avg <= 4.5 ? 2 : avg <= 9.0 ? 4 : avg <= 24.0 ? 8 : avg <= 64.0 ? 16 : avg <= 160 ? 32 : 64;
}

template <int ROLE, typename V, typename MV = V>
static int spmv_dispatch(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const MV* val,
                         const V* x, V* y, V a, V b, int lanes_per_row, hx_stream s, const V* yin = nullptr,
                         SpmvEpilogue<V> ep = SpmvEpilogue<V>()) {
  HX_REQUIRE(n_rows >= 0 && nnz >= 0 && rowptr && x && y && (nnz == 0 || (col && val)), "bad argument");
  if (n_rows == 0) return HX_OK;
  const int lpr = lanes_per_row ? lanes_per_row : pick_lpr(nnz, n_rows);
  switch (lpr) {
    case 1: return launch<1, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 2: return launch<2, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 4: return launch<4, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 8: return launch<8, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 16: return launch<16, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 32: return launch<32, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    case 64: return launch<64, ROLE, V, MV>(n_rows, rowptr, col, val, x, y, a, b, hx_s(s), yin, ep);
    default: return hx_fail(HX_ERR_ARG, __func__, "lanes_per_row must be 0,1,2,4,8,16,32,64");
  }
}

extern "C" int hx_csr_spmv_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                             const hx_c128* val, const hx_c128* x, hx_c128* y, const double alpha[2],
                             const double beta[2], int lanes_per_row, hx_stream s) {
  HX_REQUIRE(alpha && beta, "bad argument");
  return spmv_dispatch<0, zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)x, (zc*)y, hx_z(alpha), hx_z(beta),
                              lanes_per_row, s);
}

// y = A x together with the per-block partial sums of the BILINEAR form x^T y (square A): hx_cocg's q = K p and p^T q in one
// launch for operators that run on the CSR kernels (small, launch-bound problems; the SELL-64 product has its own fused form).
// partials[b], b < gridDim.x <= max_blocks; the consumer (k_cocg_update) adds them in a fixed order.
template <int LPR>
__global__ void __launch_bounds__(256)
k_spmv_dot(int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const zc* __restrict__ val,
           const zc* __restrict__ x, zc* __restrict__ y, zc* __restrict__ partials) {
  constexpr int RPB = 256 / LPR;
  __shared__ zc red[256 / HX_WAVE];
  const int sub = threadIdx.x % LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t sweeps = ((int64_t)n + stride - 1) / stride;
  int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  zc dot = make_double2(0.0, 0.0);
  for (int64_t it = 0; it < sweeps; ++it, row += stride) {
    zc acc = make_double2(0.0, 0.0);
    if (row < n) acc = row_dot<LPR, zc, zc>(col, val, x, rowptr[row], rowptr[row + 1], sub);
    if (LPR > 1) acc = zreduce_sub<LPR>(acc);
    if (row < n && sub == 0) { y[row] = acc; zfma(dot, x[row], acc); }
  }
  dot = zreduce_sub<HX_WAVE>(dot);
  if ((threadIdx.x & (HX_WAVE - 1)) == 0) red[threadIdx.x / HX_WAVE] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    zc v = red[0];
    for (int w = 1; w < 256 / HX_WAVE; ++w) v = zadd(v, red[w]);
    partials[blockIdx.x] = v;
  }
}
int hx_csr_spmv_z_dot(int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val, const hx_c128* x,
                      hx_c128* y, hx_c128* partials, int max_blocks, int* n_blocks, hx_stream s) {
  HX_REQUIRE(n > 0 && rowptr && col && val && x && y && partials && max_blocks >= 1 && n_blocks, "bad argument");
  const int lpr = pick_lpr(nnz, n);
#define HX_SD(L) { int grid = hx_grid((int64_t)n, 256 / L, max_blocks); *n_blocks = grid; \
    k_spmv_dot<L><<<grid, 256, 0, hx_s(s)>>>(n, rowptr, col, (const zc*)val, (const zc*)x, (zc*)y, (zc*)partials); } break
  switch (lpr) { case 1: HX_SD(1); case 2: HX_SD(2); case 4: HX_SD(4); case 8: HX_SD(8); case 16: HX_SD(16); case 32: HX_SD(32);
                 default: HX_SD(64); }
#undef HX_SD
  HX_LAUNCH_CHECK("k_spmv_dot");
  return HX_OK;
}

// internal entry points for amg.hip: ROLE 1 = transfer / coarse products; complex64 variants for the
// single-precision V-cycle (role 2 = its operator products, so profiles keep the roofline kernel separate)
int hx_csr_spmv_transfer(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                         const hx_c128* x, hx_c128* y, const double alpha[2], const double beta[2], hx_stream s) {
  return spmv_dispatch<1, zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)x, (zc*)y, hx_z(alpha), hx_z(beta), 0, s);
}
int hx_csr_spmv_c(int role, int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                  const float* x, float* y, float alpha, float beta, hx_stream s) {
  const cc a = make_float2(alpha, 0.f), b = make_float2(beta, 0.f);
  if (role == 1) return spmv_dispatch<1, cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)x, (cc*)y, a, b, 0, s);
  return spmv_dispatch<2, cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)x, (cc*)y, a, b, 0, s);
}

// transfer operators with REAL values (float for the complex64 hierarchy, double for the complex128 one): 8 instead of
// 12 (resp. 12 instead of 20) bytes per entry with the column index
int hx_csr_spmv_creal(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, const float* x,
                      float* y, float alpha, float beta, hx_stream s) {
  return spmv_dispatch<1, cc, float>(n_rows, nnz, rowptr, col, val, (const cc*)x, (cc*)y, make_float2(alpha, 0.f),
                                     make_float2(beta, 0.f), 0, s);
}
int hx_csr_spmv_zreal(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const double* val,
                      const hx_c128* x, hx_c128* y, double alpha, double beta, hx_stream s) {
  return spmv_dispatch<1, zc, double>(n_rows, nnz, rowptr, col, val, (const zc*)x, (zc*)y, make_double2(alpha, 0.0),
                                      make_double2(beta, 0.0), 0, s);
}
// restriction with the coarse level's first smoothing step as epilogue: y = R x, x1 = a0 .* d .* y  (one launch instead of two)
int hx_csr_restrict_diag_c(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, const float* x,
                           float* y, const float* d, float* x1, float a0, hx_stream s) {
  HX_REQUIRE(d && x1, "bad argument");
  SpmvEpilogue<cc> ep; ep.d = (const cc*)d; ep.x = (cc*)x1; ep.a = a0;
  return spmv_dispatch<1, cc, float>(n_rows, nnz, rowptr, col, val, (const cc*)x, (cc*)y, make_float2(1.f, 0.f), make_float2(0.f, 0.f), 0, s,
                                     nullptr, ep);
}
int hx_csr_restrict_diag_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const double* val, const hx_c128* x,
                           hx_c128* y, const hx_c128* d, hx_c128* x1, double a0, hx_stream s) {
  HX_REQUIRE(d && x1, "bad argument");
  SpmvEpilogue<zc> ep; ep.d = (const zc*)d; ep.x = (zc*)x1; ep.a = a0;
  return spmv_dispatch<1, zc, double>(n_rows, nnz, rowptr, col, val, (const zc*)x, (zc*)y, make_double2(1.0, 0.0), make_double2(0.0, 0.0), 0,
                                      s, nullptr, ep);
}
// r = b - A x in one pass (internal; the V-cycle's residual on the CSR levels)
int hx_csr_residual_c(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, const float* x,
                      const float* b, float* r, hx_stream s) {
  return spmv_dispatch<2, cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)x, (cc*)r, make_float2(-1.f, 0.f),
                              make_float2(1.f, 0.f), 0, s, (const cc*)b);
}
int hx_csr_residual_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                      const hx_c128* x, const hx_c128* b, hx_c128* r, hx_stream s) {
  return spmv_dispatch<0, zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)x, (zc*)r, make_double2(-1.0, 0.0),
                              make_double2(1.0, 0.0), 0, s, (const zc*)b);
}

// ------------------------------------------------------------------------------------------------
// Fused smoother sweep (AMG):  xout = xin + omega * dinv .* (b - A xin) + mom * (xin - xprev).
// mom = 0: damped Jacobi (xprev unused; a NULL xprev with mom != 0 means the previous iterate was 0); with the Chebyshev coefficients (omega_k, mom_k) the same kernel is one
// step of the Chebyshev three-term recurrence on D^-1 A.  Same streaming pattern and traffic as the SpMV plus one
// read of b, dinv, xin (and xprev) per row.
template <int LPR, typename V>
__global__ void __launch_bounds__(256)
k_jacobi(int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const V* __restrict__ val,
         const V* __restrict__ dinv, const V* __restrict__ b, const V* __restrict__ xin, V* __restrict__ xout,
         typename real_of<V>::type omega, const V* __restrict__ xprev, typename real_of<V>::type mom, int32_t row0 = 0) {
  // row0: the n rows of this CSR are rows row0 .. row0 + n - 1 of the vectors (the ghost-layer rows a rank smooths redundantly
  // in the deep-halo cycle: their matrix is stored on its own, their columns index the whole extended vector)
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t sweeps = ((int64_t)n + stride - 1) / stride;
  int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  for (int64_t it = 0; it < sweeps; ++it, row += stride) {
    V acc = czero<V>();
    if (row < n) acc = row_dot<LPR, V, V>(col, val, xin, rowptr[row], rowptr[row + 1], sub);
    if (LPR > 1) acc = zreduce_sub<LPR>(acc);
    if (row < n && sub == 0) {
      const int64_t g = row0 + row;
      const V r = zsub(b[g], acc);
      const V x0 = xin[g];
      V x = x0;
      zfma(x, zscale(omega, dinv[g]), r);
      if (mom != 0) x = zadd(x, zscale(mom, xprev ? zsub(x0, xprev[g]) : x0));   // xprev == NULL: previous iterate is 0
      xout[g] = x;
    }
  }
}

template <typename V>
static int jacobi_dispatch(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const V* val,
                           const V* dinv, const V* b, const V* xin, V* xout, typename real_of<V>::type omega,
                           const V* xprev, typename real_of<V>::type mom, hx_stream s, int32_t row0 = 0) {
  HX_REQUIRE(n_rows >= 0 && rowptr && col && val && dinv && b && xin && xout && xin != xout, "bad argument");
  if (n_rows == 0) return HX_OK;
  int lpr = pick_lpr(nnz, n_rows);
  if (lpr < 2) lpr = 2;
  if (lpr > 32) lpr = 32;
#define HX_GO(L) k_jacobi<L, V><<<hx_grid((int64_t)n_rows, 256 / L, 256 * 16), 256, 0, hx_s(s)>>>(                \
      n_rows, rowptr, col, val, dinv, b, xin, xout, omega, xprev, mom, row0)
  switch (lpr) {
    case 2: HX_GO(2); break;
    case 4: HX_GO(4); break;
    case 8: HX_GO(8); break;
    case 16: HX_GO(16); break;
    default: HX_GO(32); break;
  }
#undef HX_GO
  HX_LAUNCH_CHECK("k_jacobi");
  return HX_OK;
}

extern "C" int hx_csr_jacobi_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                               const hx_c128* val, const hx_c128* dinv, const hx_c128* b, const hx_c128* xin,
                               hx_c128* xout, double omega, hx_stream s) {
  return jacobi_dispatch<zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)dinv, (const zc*)b, (const zc*)xin,
                             (zc*)xout, omega, nullptr, 0.0, s);
}
// internal: three-term (Chebyshev) form; xprev may be NULL (plain damped Jacobi)
int hx_csr_smooth_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                    const hx_c128* dinv, const hx_c128* b, const hx_c128* xin, hx_c128* xout, double omega,
                    const hx_c128* xprev, double mom, hx_stream s) {
  return jacobi_dispatch<zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)dinv, (const zc*)b, (const zc*)xin,
                             (zc*)xout, omega, (const zc*)xprev, mom, s);
}
int hx_csr_jacobi_c(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                    const float* dinv, const float* b, const float* xin, float* xout, float omega, hx_stream s) {
  return jacobi_dispatch<cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)dinv, (const cc*)b, (const cc*)xin,
                             (cc*)xout, omega, nullptr, 0.f, s);
}
int hx_csr_smooth_c(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                    const float* dinv, const float* b, const float* xin, float* xout, float omega, const float* xprev,
                    float mom, hx_stream s) {
  return jacobi_dispatch<cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)dinv, (const cc*)b, (const cc*)xin,
                             (cc*)xout, omega, (const cc*)xprev, mom, s);
}

// the same step on a block of rows that starts at row0 of the vectors (deep-halo cycle: first ghost layer)
int hx_csr_smooth_rows_c(int32_t n_rows, int32_t row0, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* dinv, const float* b, const float* xin, float* xout, float omega, const float* xprev, float mom,
                         hx_stream s) {
  return jacobi_dispatch<cc>(n_rows, nnz, rowptr, col, (const cc*)val, (const cc*)dinv, (const cc*)b, (const cc*)xin, (cc*)xout, omega,
                             (const cc*)xprev, mom, s, row0);
}
int hx_csr_smooth_rows_z(int32_t n_rows, int32_t row0, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                         const hx_c128* dinv, const hx_c128* b, const hx_c128* xin, hx_c128* xout, double omega, const hx_c128* xprev,
                         double mom, hx_stream s) {
  return jacobi_dispatch<zc>(n_rows, nnz, rowptr, col, (const zc*)val, (const zc*)dinv, (const zc*)b, (const zc*)xin, (zc*)xout, omega,
                             (const zc*)xprev, mom, s, row0);
}

// x = scale .* d .* b  (first Jacobi sweep from a zero initial guess; also a diagonal preconditioner)
template <typename V>
__global__ void k_diag_scale(int64_t n, typename real_of<V>::type scale, const V* __restrict__ d, const V* __restrict__ b,
                             V* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = zscale(scale, zmul(d[i], b[i]));
}
extern "C" int hx_zdiag_scale(int64_t n, double scale, const hx_c128* d, const hx_c128* b, hx_c128* x, hx_stream s) {
  HX_REQUIRE(n >= 0 && d && b && x, "bad argument");
  if (n == 0) return HX_OK;
  k_diag_scale<zc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, scale, (const zc*)d, (const zc*)b, (zc*)x);
  HX_LAUNCH_CHECK("k_diag_scale");
  return HX_OK;
}
int hx_cdiag_scale(int64_t n, float scale, const float* d, const float* b, float* x, hx_stream s) {
  if (n == 0) return HX_OK;
  k_diag_scale<cc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, scale, (const cc*)d, (const cc*)b, (cc*)x);
  HX_LAUNCH_CHECK("k_diag_scale<c64>");
  return HX_OK;
}

// precision casts at the V-cycle boundary
__global__ void k_z2c(int64_t n, const zc* __restrict__ x, cc* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const zc v = x[i];
    y[i] = make_float2((float)v.x, (float)v.y);
  }
}
__global__ void k_c2z(int64_t n, const cc* __restrict__ x, zc* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const cc v = x[i];
    y[i] = make_double2((double)v.x, (double)v.y);
  }
}
int hx_cast_z2c(int64_t n, const hx_c128* x, float* y, hx_stream s) {
  if (n == 0) return HX_OK;
  k_z2c<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, (const zc*)x, (cc*)y);
  HX_LAUNCH_CHECK("k_z2c");
  return HX_OK;
}
int hx_cast_c2z(int64_t n, const float* x, hx_c128* y, hx_stream s) {
  if (n == 0) return HX_OK;
  k_c2z<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, (const cc*)x, (zc*)y);
  HX_LAUNCH_CHECK("k_c2z");
  return HX_OK;
}

// public complex64 entry points (hx.h): the single-precision copies used by the AMG V-cycle
extern "C" int hx_csr_spmv_c64(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c64* val,
                               const hx_c64* x, hx_c64* y, float alpha, float beta, hx_stream s) {
  return hx_csr_spmv_c(2, n_rows, nnz, rowptr, col, (const float*)val, (const float*)x, (float*)y, alpha, beta, s);
}
extern "C" int hx_csr_jacobi_c64(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                 const hx_c64* val, const hx_c64* dinv, const hx_c64* b, const hx_c64* xin, hx_c64* xout,
                                 float omega, hx_stream s) {
  return hx_csr_jacobi_c(n_rows, nnz, rowptr, col, (const float*)val, (const float*)dinv, (const float*)b,
                         (const float*)xin, (float*)xout, omega, s);
}
