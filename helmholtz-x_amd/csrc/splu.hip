// Sparse direct solver on the device: supernodal multifrontal LU of a complex128 matrix (hx_splu_*).
//
// replaces: KSP(preonly) + PC(LU) inside ST sinvert -- MatLUFactorNumeric / MatSolve / MatSolveTranspose of PETSc's LU or MUMPS
// (eigensolvers.py:49-50,102-103 -> :62,:113 of the reference).  Host analysis: splu_analyse.hip; layout: splu_internal.h.
//
// Numeric factorisation, level by level up the assembly tree (fronts of one level are independent and share every launch):
//   zero the contribution blocks -> extend-add the children (round r adds the r-th child of every front: race-free, fixed order)
//   -> per pivot block of 64 columns: k_diag (Gauss-Jordan inverse of the 64 x 64 diagonal block, partial pivoting INSIDE the
//   block, one row of the block per lane in registers), k_gemm<PANEL> (L21 <- F21 D^-1), k_gemm<INNER> (right-looking update
//   of the rest of the current 256-column outer block, L and U side) -- before an outer block starts k_gemm<LEFT> applies all
//   earlier outer blocks at once (left-looking: K = hundreds, the rank-64 updates stay inside 256 columns) -> k_gemm<SUPD>
//   F22 -= L21 U12.
// Everything GEMM-shaped is ONE kernel: C (-)= A B on column-major complex128 operands, 64 x 64 tiles, 4 waves, each wave a
// 32 x 32 block as 2 x 2 x (re, im) accumulators of v_mfma_f64_16x16x4_f64 (4 real MFMAs per complex product block; operands
// staged through LDS as interleaved complex so one ds_read_b128 feeds the re and im fragments).
// No pivoting across blocks: K = A + sigma B + sigma^2 C of a Helmholtz problem is complex symmetric with a complex shift
// (every leading block is a shifted sub-problem); tiny pivots are replaced (static pivoting) and counted, the caller checks the
// residual and refines.
//
// Solves: forward up the levels, backward down, supernodal (dense panels): small fronts run whole in one workgroup per front and
// level, large fronts step through their pivot blocks with tiled launches.  Both sweeps are "rows of a panel times a vector"
// with the 64 contiguous rows on the lanes (L is stored m x ns, U ns x m).  Child updates travel through per-front workspace
// slices (no atomics: bitwise reproducible).  Transposed solves use the same factors (U^T L^T) through column-dot kernels.
#include <algorithm>
#include <string.h>
#include <vector>

#include "hx_common.h"
#include "splu_internal.h"

#define NB SPLU_NB
#define NBO 256                 // outer block of the panel factorisation (left-looking granularity)
#define GT 64                   // GEMM tile
#define GK 16                   // GEMM k chunk
#define GLD (GT + 1)            // LDS stride (complex elements) of a staged operand chunk
#define BIG_FRONT_ELEMS (48 * 1024)   // a front whose panel has more entries than this steps through launches in the solves
#define FWD_ROWS 64             // rows per workgroup of the large-front forward update
#define BWD_COLS 256            // columns per workgroup of the large-front backward partial products
#define SOLVE_ROWS 512          // rows per workgroup of the column-dot kernels (transposed solves)
#define GV_K 64                // columns per workgroup of the large-front matrix-vector products (64 rows x GV_K columns)

typedef double v4d __attribute__((ext_vector_type(4)));

enum { K_ZERO = 0, K_EXTEND, K_DIAG, K_PANEL, K_INNER, K_LEFT, K_SUPD,
       S_FWD_SMALL, S_FWD_GATHER, S_FWD_DIAG, S_FWD_UPD, S_BWD_SMALL, S_BWD_GATHER, S_BWD_DOT, S_BWD_DIAG,
       K_INV_INIT, K_INVU_SCALE, K_INVL, K_INVU, S_GV_PART, S_GV_REDUCE };
enum { GV_IL = 0, GV_L21, GV_U12, GV_IU };

struct SpluDev {
  const int32_t* sfirst; const int64_t* sptr; const int32_t* srow; const int32_t* rel; const int32_t* sparent;
  const int64_t* loff; const int64_t* uoff; const int64_t* ioff; const int64_t* soff; const int64_t* woff; const int64_t* poff; const int64_t* voff;
  const int32_t* child_ptr; const int32_t* child_list;
  zc* F; zc* W; zc* T; zc* wk; zc* part; zc* V; int32_t* status;
  int sym;          // this factorisation runs in symmetric mode (A = A^T): U panels are transposed copies of the unscaled L panels
};

struct Launch { int kind; int64_t list; int32_t nf; int32_t grid; int32_t p0; int64_t list_sym = -1; int32_t grid_sym = 0; };   // list_sym: prefix of the symmetric-mode tiling

struct hx_splu {
  SpluSymbolic sym;
  int64_t nnz_in = 0;
  bool factored = false;
  std::vector<Launch> fplan, iplan, fwd_plan[2], bwd_plan[2];
  std::vector<int64_t> poff, voff;      // per supernode: partial sums of the large-front products / inverted diagonal blocks (-1: none)
  int64_t part_elems = 0, inv_elems = 0;
  double inv_flops = 0.0;
  // device
  int32_t *d_i32 = nullptr; int64_t* d_i64 = nullptr; int32_t* d_lists = nullptr; int64_t* d_amap = nullptr;
  int32_t *d_perm = nullptr, *d_status = nullptr;
  zc *F = nullptr, *W = nullptr, *T = nullptr, *wk = nullptr, *part = nullptr, *pb = nullptr, *xp = nullptr, *V = nullptr;
  SpluDev dev{};
  int32_t tiny_pivots = 0;
  double factor_seconds = 0.0;
  int32_t* d_mirror = nullptr;
  double* d_symcheck = nullptr;
  bool symmetric = false;
  double flops_done = 0.0;
  hipGraphExec_t sweep_graph[2] = {nullptr, nullptr};   // the two sweeps of a solve (A x = b / A^T x = b), captured at the first solve
  bool graph_failed = false;
};

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int trap_tiles(int tm, int tn) {      // tiles (ti, tj) with ti >= tj
  int c = 0;
  for (int tj = 0; tj < tn; ++tj) c += tm > tj ? tm - tj : 0;
  return c;
}
__host__ __device__ static inline int splu_tiles(int kind, int ns, int nr, int p0) {
  const int m = ns + nr;
  switch (kind) {
    case K_PANEL: { const int c0 = p0 * NB, c1 = min(ns, c0 + NB); return cdiv(m - c1, GT); }
    case K_INNER: { const int c0 = p0 * NB, c1 = min(ns, c0 + NB), cj1 = min(ns, (c0 / NBO + 1) * NBO);
                    return 2 * trap_tiles(cdiv(m - c1, GT), cdiv(cj1 - c1, GT)); }
    case K_LEFT: { const int cj0 = p0 * NBO, cj1 = min(ns, cj0 + NBO); return 2 * trap_tiles(cdiv(m - cj0, GT), cdiv(cj1 - cj0, GT)); }
    case K_SUPD: return cdiv(nr, GT) * cdiv(nr, GT);
    case K_ZERO: return cdiv(nr * nr, 256 * 16);     // nr <= 46 k: no overflow
    case K_EXTEND: return cdiv(nr, GT) * cdiv(nr, GT);
    case S_FWD_UPD: { const int c0 = p0 * NB, c1 = min(ns, c0 + NB); return cdiv(m - c1, FWD_ROWS); }
    case S_BWD_DOT: { const int c0 = p0 * NB, c1 = min(ns, c0 + NB); return cdiv(m - c1, BWD_COLS); }
    case K_INV_INIT: return cdiv(ns, NB);
    case K_INVU_SCALE: { int c = 0; for (int c1 = NB; c1 < ns; c1 += NB) c += cdiv(ns - c1, GT); return c; }
    case K_INVL: return p0;                                               // row block p0: columns 0 .. p0 * 64
    case K_INVU: { const int k = cdiv(ns, NB) - 1 - p0; return cdiv(ns - (k + 1) * NB, GT); }
    case S_GV_PART: switch (p0) { case GV_IL: case GV_IU: return cdiv(ns, 64) * cdiv(ns, GV_K);
                                  case GV_L21: return cdiv(nr, 64) * cdiv(ns, GV_K);
                                  default: return cdiv(ns, 64) * cdiv(nr, GV_K); }
    case S_GV_REDUCE: return cdiv(p0 == GV_L21 ? nr : ns, 256);
    default: return 1;
  }
}
// transposed solves use other tilings of the same two launch kinds
__host__ __device__ static inline int splu_tiles_trans(int kind, int ns, int nr, int p0) {
  const int m = ns + nr, c0 = p0 * NB, c1 = min(ns, c0 + NB);
  return cdiv(m - c1, SOLVE_ROWS);
}

// which front does workgroup b of a launch serve: prefix[f] <= b < prefix[f + 1]
__device__ __forceinline__ int find_front(const int32_t* __restrict__ prefix, int nf, int b) {
  int lo = 0, hi = nf;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

// trapezoid decode: t-th tile (ti >= tj) of a tm x tn tile grid, column by column
__device__ __forceinline__ void trap_decode(int t, int tm, int tn, int& ti, int& tj) {
  tj = 0;
  for (; tj < tn; ++tj) {
    const int c = tm > tj ? tm - tj : 0;
    if (t < c) break;
    t -= c;
  }
  ti = tj + t;
}

// ------------------------------------------------------------------------------------------------- assembly of the fronts
// max |a_ij - a_ji| and max |a_ij| over the upper entries of a matrix with a symmetric pattern (out[0], out[1]: doubles >= 0, so
// their bit patterns order like integers)
__global__ void __launch_bounds__(256) k_splu_symcheck(int64_t nnz, const int32_t* __restrict__ mirror, const zc* __restrict__ val,
                                                       unsigned long long* __restrict__ out) {
  double dmax = 0.0, vmax = 0.0;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
    const zc v = val[k];
    vmax = fmax(vmax, fmax(fabs(v.x), fabs(v.y)));
    const int32_t mk = mirror[k];
    if (mk >= 0) { const zc u = val[mk]; dmax = fmax(dmax, fmax(fabs(v.x - u.x), fabs(v.y - u.y))); }
  }
  for (int o = 32; o > 0; o >>= 1) { dmax = fmax(dmax, __shfl_xor(dmax, o)); vmax = fmax(vmax, __shfl_xor(vmax, o)); }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, (unsigned long long)__double_as_longlong(dmax));
    atomicMax(out + 1, (unsigned long long)__double_as_longlong(vmax));
  }
}

__global__ void __launch_bounds__(256) k_splu_scatter(int64_t nnz, const int64_t* __restrict__ amap, const zc* __restrict__ val,
                                                      zc* __restrict__ F, int sym) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
    const zc v = val[k];
    int64_t dest = amap[k];
    if (dest < 0) { if (sym) continue; dest = -dest - 1; }       // an entry of a U panel: symmetric mode copies those from L
    zc* d = F + dest;
    d->x += v.x; d->y += v.y;        // duplicates of a (row, col) pair do not occur in a CSR matrix: plain add
  }
}

__global__ void __launch_bounds__(256) k_splu_zero(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf) {
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int64_t nr = d.sptr[s + 1] - d.sptr[s];
  zc* S = d.T + d.soff[s];
  const int64_t base = (int64_t)(blockIdx.x - prefix[f]) * 256 * 16;
  for (int q = 0; q < 16; ++q) {
    const int64_t e = base + q * 256 + threadIdx.x;
    if (e < nr * nr) S[e] = make_double2(0.0, 0.0);
  }
}

// fronts[] holds CHILDREN here: tile (ti, tj) of the child's contribution block is added into its parent's storage
__global__ void __launch_bounds__(256) k_splu_extend(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf) {
  const int f = find_front(prefix, nf, blockIdx.x);
  const int c = fronts[f];
  const int t = blockIdx.x - prefix[f];
  const int nrc = (int)(d.sptr[c + 1] - d.sptr[c]);
  const int tm = cdiv(nrc, GT);
  int ti, tj;
  if (d.sym) trap_decode(t, tm, tm, ti, tj);      // symmetric mode: the launch holds the lower triangle of tiles only
  else { ti = t % tm; tj = t / tm; }
  const int p = d.sparent[c];
  const int nsp = d.sfirst[p + 1] - d.sfirst[p];
  const int64_t nrp = d.sptr[p + 1] - d.sptr[p], mp = nsp + nrp;
  const int32_t* rel = d.rel + d.sptr[c];
  const zc* Sc = d.T + d.soff[c];
  zc* Lp = d.F + d.loff[p];
  zc* Up = d.F + d.uoff[p];
  zc* Sp = d.T + (nrp > 0 ? d.soff[p] : 0);
  __shared__ int32_t rj[GT];
  if (threadIdx.x < GT) rj[threadIdx.x] = (tj * GT + (int)threadIdx.x < nrc) ? rel[tj * GT + threadIdx.x] : -1;
  __syncthreads();
  const int i = ti * GT + (threadIdx.x & 63);
  if (i >= nrc) return;
  const int64_t a = rel[i];
  for (int jj = threadIdx.x >> 6; jj < GT; jj += 4) {
    const int64_t b = rj[jj];
    if (b < 0) break;
    if (d.sym && b > a) continue;
    const zc v = Sc[i + (int64_t)(tj * GT + jj) * nrc];
    zc* dst;
    if (b < nsp && a >= b) dst = Lp + a + b * mp;
    else if (a < nsp && b > a) dst = Up + a + b * nsp;
    else dst = Sp + (a - nsp) + (b - nsp) * nrp;
    dst->x += v.x; dst->y += v.y;
  }
}

// ------------------------------------------------------------------------------------------------- diagonal block
// One workgroup (512 threads) per front: D = F[c0:c1, c0:c1] (lower + diagonal from Lp, strict upper from Up) -> D^-1 by
// in-place Gauss-Jordan with threshold pivoting inside the block.  The chain of 64 dependent pivots bounds the critical path
// of a whole factorisation, and both the LDS (a 16-byte store costs 13 cycles per wave) and a lone wave's dependent issue
// (~8 cycles per instruction) are slow against it, so: the matrix lives in REGISTERS (thread (a = t >> 3, q = t & 7) holds
// D[a, q + 8 j], j < 8), only the pivot row and the pivot column travel through LDS, run-time register indices are resolved
// by wave-uniform switches, and a step is ONE barrier long when the diagonal entry is an acceptable pivot
// (not small against the block's largest entry): while it eliminates with pivot k every thread already publishes what step k + 1 needs -- column
// k + 1, the row holding the next diagonal entry (unscaled: every thread scales its own multiplier by 1 / pivot instead) and,
// by one LDS atomic max per candidate row, the largest entry of column k + 1 (key = magnitude bits | row).  Only when the
// diagonal entry fails the test rows are exchanged: by RENAMING (each thread carries the logical index of the row it holds)
// plus one more barrier to publish that row.  The column permutation of the inverse is applied on the way out.  Blocks
// narrower than 64 are padded with the identity.
__device__ __forceinline__ unsigned long long pivot_key(double mag, int row) {   // mag >= 0 (NaN sorts above everything)
  return ((unsigned long long)__double_as_longlong(mag) & ~63ull) | (unsigned long long)(63 - row);
}
__device__ __forceinline__ double fast_rcp(double x) {            // 1 / x for a normal x: v_rcp_f64 + two Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
struct DiagShared {
  zc col[2][NB], rowb[2][NB];
  unsigned long long key;
  int perm[NB];
  double wmax[8];
};

// step k = 8 J + qk of the elimination: column k lives in register r[J] of the threads with q == qk, column k + 1 in r[JN]
template <int J, int JN>
__device__ __forceinline__ void diag_step(DiagShared& sh, zc (&r)[8], int& la, const int qk, const int w, const int q, const int tid,
                                          const double small2, const double tiny2, const double scale_ref, int32_t* status) {
  {
    const int k = 8 * J + qk;
    if (k >= w) return;                           // (uniform) the identity padding needs no elimination
    const int cur = k & 1;
    const zc f = sh.col[cur][la];                 // this row's entry in column k (multiplier of the elimination)
    const zc dk = sh.col[cur][k];
    zc pv = dk;
    if (!(dk.x * dk.x + dk.y * dk.y >= small2)) { // (uniform, rare) small against the block: search column k, exchange rows by renaming
      if (tid == 0) sh.key = 0ull;
      __syncthreads();
      if (q == 0 && la >= k && la < w) atomicMax(&sh.key, pivot_key(f.x * f.x + f.y * f.y, la));
      __syncthreads();
      const int p = 63 - (int)(sh.key & 63ull);
      if (p != k) {
        pv = sh.col[cur][p];
        if (la == k) la = p; else if (la == p) la = k;
        if (la == k) {                            // publish the new row k
#pragma unroll
          for (int j = 0; j < 8; ++j) sh.rowb[cur][q + 8 * j] = r[j];
        }
        if (tid == 0) { const int tp = sh.perm[k]; sh.perm[k] = sh.perm[p]; sh.perm[p] = tp; }
      }
      __syncthreads();
      const double mag = pv.x * pv.x + pv.y * pv.y;
      if (!(mag > tiny2)) {                       // static pivoting (also catches NaN)
        if (tid == 0) atomicAdd(status, 1);
        const double nrm = sqrt(mag);
        pv = nrm > 0.0 ? make_double2(pv.x / nrm * 1e-8 * scale_ref, pv.y / nrm * 1e-8 * scale_ref) : make_double2(1e-8 * scale_ref, 0.0);
        if (!(scale_ref > 0.0)) pv = make_double2(1.0, 0.0);
      }
    }
    const double inv = fast_rcp(pv.x * pv.x + pv.y * pv.y);
    const zc pinv = make_double2(pv.x * inv, -pv.y * inv);
    const zc fp = zmul(f, pinv);
    zc u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = sh.rowb[cur][q + 8 * j];
    if (la != k) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { r[j].x -= fp.x * u[j].x - fp.y * u[j].y; r[j].y -= fp.x * u[j].y + fp.y * u[j].x; }
      if (q == qk) r[J] = make_double2(-fp.x, -fp.y);       // column k becomes a column of the inverse
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = zmul(r[j], pinv);
      if (q == qk) r[J] = pinv;
    }
    if (k + 1 < w) {                              // what step k + 1 reads: column k + 1 and the row of the next diagonal entry
      if (q == ((qk + 1) & 7)) sh.col[cur ^ 1][la] = r[JN];
      if (la == k + 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh.rowb[cur ^ 1][q + 8 * j] = r[j];
      }
    }
    __syncthreads();
  }
}

template <int J>
__device__ __forceinline__ void diag_steps(DiagShared& sh, zc (&r)[8], int& la, const int w, const int q, const int tid,
                                           const double small2, const double tiny2, const double scale_ref, int32_t* status) {
#pragma unroll 1
  for (int qk = 0; qk < 7; ++qk) diag_step<J, J>(sh, r, la, qk, w, q, tid, small2, tiny2, scale_ref, status);
  diag_step<J, (J < 7 ? J + 1 : 7)>(sh, r, la, 7, w, q, tid, small2, tiny2, scale_ref, status);
}

__global__ void __launch_bounds__(512) k_splu_diag(SpluDev d, const int32_t* __restrict__ fronts, int kb) {
  __shared__ DiagShared sh;
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  const int c0 = kb * NB, w = min(ns, c0 + NB) - c0;
  const zc* Lp = d.F + d.loff[s];
  const zc* Up = d.F + d.uoff[s];
  const int tid = threadIdx.x, a = tid >> 3, q = tid & 7;
  int la = a;                                     // logical row held by this thread
  zc r[8];
  double amax = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int b = q + 8 * j;
    zc v = make_double2(a == b ? 1.0 : 0.0, 0.0);
    if (a < w && b < w) v = (a >= b) ? Lp[(c0 + a) + (int64_t)(c0 + b) * m]
                                     : (d.sym ? Lp[(c0 + b) + (int64_t)(c0 + a) * m] : Up[(c0 + a) + (int64_t)(c0 + b) * ns]);
    r[j] = v;
    if (a < w && b < w) amax = fmax(amax, fmax(fabs(v.x), fabs(v.y)));
  }
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
  if ((tid & 63) == 0) sh.wmax[tid >> 6] = amax;
  if (tid < NB) sh.perm[tid] = tid;
  if (q == 0) sh.col[0][a] = r[0];
  if (a == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sh.rowb[0][q + 8 * j] = r[j];
  }
  __syncthreads();
  double scale_ref = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) scale_ref = fmax(scale_ref, sh.wmax[i]);
  const double tiny2 = (1e-13 * scale_ref) * (1e-13 * scale_ref), small2 = (1e-5 * scale_ref) * (1e-5 * scale_ref);
  diag_steps<0>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<1>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<2>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<3>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<4>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<5>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<6>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  diag_steps<7>(sh, r, la, w, q, tid, small2, tiny2, scale_ref, d.status);
  // stored column b holds column perm[b] of D^-1 (in-place Gauss-Jordan with row exchanges)
  zc* Wd = d.W + d.ioff[s] + (int64_t)kb * NB * NB;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int b = q + 8 * j;
    const int cc = sh.perm[b];
    Wd[la + cc * NB] = (la < w && cc < w) ? r[j] : make_double2(0.0, 0.0);
  }
}

// ------------------------------------------------------------------------------------------------- GEMM  C (-)= A B
// A: M x K, B: K x N, C: M x N, all column-major complex128.  ASSIGN: C = A B (C may be A itself when N = K <= 64: every
// operand chunk is in LDS / registers before the first store).  One 64 x 64 tile per workgroup.
// MFMA mapping: D[row][col] of v_mfma_f64_16x16x4_f64 has col = lane & 15, so the tile's ROW index i (contiguous in memory)
// rides on lane & 15: the "a" operand is the B-matrix fragment (k, j), the "b" operand the A-matrix fragment (i, k).
struct GemmProb { const zc* A; const zc* B; zc* C; int64_t lda, ldb, ldc; int M, N, K, K0 = 0;    // k runs over [K0, K)
                  zc* T = nullptr; int64_t ldt = 0; };       // T != null: the A tile is also written TRANSPOSED, T[k + i ldt] = A[i, k]
enum { G_SUB = 0, G_ASSIGN = 1, G_NEG = 2 };     // C -= A B, C = A B, C = -A B

template <int MODE>
__device__ __forceinline__ void gemm_tile(const GemmProb& p, int ti, int tj, zc* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wi = wv & 1, wj = wv >> 1;
  const int row0 = ti * GT, col0 = tj * GT;
  const int lr = tid & 63, lk = tid >> 6;           // staging of A: row lr, k offset lk + 4q
  const int bk = tid & 15, bj = tid >> 4;           // staging of B: k offset bk, column bj + 16q (k is contiguous in memory)
  const bool a_ok = row0 + lr < p.M;
  const zc* Ag = p.A + row0 + lr;
  const zc* Bg = p.B + (int64_t)col0 * p.ldb;
  zc ra[4], rb[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + lk + 4 * q;
      ra[q] = (a_ok && k < p.K) ? Ag[(int64_t)k * p.lda] : make_double2(0.0, 0.0);
      const int j = bj + 16 * q;
      rb[q] = (col0 + j < p.N && k0 + bk < p.K) ? Bg[(k0 + bk) + (int64_t)j * p.ldb] : make_double2(0.0, 0.0);
    }
  };
  v4d cr[2][2], ci[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) { cr[x][y] = (v4d){0, 0, 0, 0}; ci[x][y] = (v4d){0, 0, 0, 0}; }
  load(p.K0);
  int buf = 0;
  const int fi = lane & 15, fk = lane >> 4;
  for (int k0 = p.K0; k0 < p.K; k0 += GK) {
    zc* sA = lds + buf * 2 * GK * GLD;
    zc* sB = sA + GK * GLD;
#pragma unroll
    for (int q = 0; q < 4; ++q) { sA[(lk + 4 * q) * GLD + lr] = ra[q]; sB[bk * GLD + bj + 16 * q] = rb[q]; }
    __syncthreads();
    if (p.T) {                                      // transposed copy of this chunk of A (16 consecutive k per 16 lanes: 256-byte runs)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = bj + 16 * q;
        if (row0 + i < p.M && k0 + bk < p.K) p.T[(k0 + bk) + (int64_t)(row0 + i) * p.ldt] = sA[bk * GLD + i];
      }
    }
    if (k0 + GK < p.K) load(k0 + GK);
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      zc av[2], bv[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        av[x] = sA[(kk + fk) * GLD + 32 * wi + 16 * x + fi];
        bv[x] = sB[(kk + fk) * GLD + 32 * wj + 16 * x + fi];
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          cr[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[y].x, av[x].x, cr[x][y], 0, 0, 0);
          cr[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[y].y, -av[x].y, cr[x][y], 0, 0, 0);
          ci[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[y].y, av[x].x, ci[x][y], 0, 0, 0);
          ci[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[y].x, av[x].y, ci[x][y], 0, 0, 0);
        }
    }
    buf ^= 1;
  }
  // D[row = j][col = i]: lane & 15 -> i, (lane >> 4) + 4 r -> j
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = row0 + 32 * wi + 16 * x + fi;
        const int j = col0 + 32 * wj + 16 * y + fk + 4 * r;
        if (i < p.M && j < p.N) {
          zc* c = p.C + i + (int64_t)j * p.ldc;
          if (MODE == G_ASSIGN) *c = make_double2(cr[x][y][r], ci[x][y][r]);
          else if (MODE == G_NEG) *c = make_double2(-cr[x][y][r], -ci[x][y][r]);
          else { zc v = *c; v.x -= cr[x][y][r]; v.y -= ci[x][y][r]; *c = v; }
        }
      }
}

template <int KIND>
__global__ void __launch_bounds__(256) k_splu_gemm(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf, int p0) {
  extern __shared__ zc lds[];
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  int t = blockIdx.x - prefix[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int nr = (int)(d.sptr[s + 1] - d.sptr[s]);
  const int64_t m = ns + nr;
  zc* Lp = d.F + d.loff[s];
  zc* Up = d.F + d.uoff[s];
  GemmProb p;
  int ti, tj;
  if (KIND == K_PANEL) {                         // Lp[c1:m, c0:c1] <- Lp[c1:m, c0:c1] D^-1
    const int c0 = p0 * NB, c1 = min(ns, c0 + NB);
    p.A = Lp + c1 + (int64_t)c0 * m; p.lda = m; p.C = (zc*)p.A; p.ldc = m;
    p.B = d.W + d.ioff[s] + (int64_t)p0 * NB * NB; p.ldb = NB;
    p.M = (int)m - c1; p.N = p.K = c1 - c0;
    if (d.sym) { p.T = Up + c0 + (int64_t)c1 * ns; p.ldt = ns; }       // U[c0:c1, c1:m] = (unscaled L[c1:m, c0:c1])^T
    gemm_tile<G_ASSIGN>(p, t, 0, lds);
  } else if (KIND == K_INNER || KIND == K_LEFT) {
    int r0, k0, k1, cj1;
    if (KIND == K_INNER) { k0 = p0 * NB; k1 = min(ns, k0 + NB); r0 = k1; cj1 = min(ns, (k0 / NBO + 1) * NBO); }
    else { k0 = 0; k1 = p0 * NBO; r0 = k1; cj1 = min(ns, r0 + NBO); }
    const int tall = cdiv((int)m - r0, GT), wide = cdiv(cj1 - r0, GT);
    const int per = trap_tiles(tall, wide);
    const int side = t / per; t -= side * per;
    if (side == 1 && d.sym) return;               // symmetric mode: the U panels are copies, nothing to update
    int tl, ts;                                   // tile along the long / the short (outer block) direction, tl >= ts
    trap_decode(t, tall, wide, tl, ts);
    p.K = k1 - k0;
    if (side == 0) {                              // Lp[r0:m, r0:cj1] -= Lp[r0:m, k0:k1] Up[k0:k1, r0:cj1]
      p.A = Lp + r0 + (int64_t)k0 * m; p.lda = m; p.B = Up + k0 + (int64_t)r0 * ns; p.ldb = ns;
      p.C = Lp + r0 + (int64_t)r0 * m; p.ldc = m; p.M = (int)m - r0; p.N = cj1 - r0;
      ti = tl; tj = ts;
    } else {                                      // Up[r0:cj1, r0:m] -= Lp[r0:cj1, k0:k1] Up[k0:k1, r0:m]
      p.A = Lp + r0 + (int64_t)k0 * m; p.lda = m; p.B = Up + k0 + (int64_t)r0 * ns; p.ldb = ns;
      p.C = Up + r0 + (int64_t)r0 * ns; p.ldc = ns; p.M = cj1 - r0; p.N = (int)m - r0;
      ti = ts; tj = tl;
    }
    gemm_tile<G_SUB>(p, ti, tj, lds);
  } else if (KIND == K_SUPD) {   // K_SUPD: S -= Lp[ns:m, 0:ns] Up[0:ns, ns:m]
    const int tm = cdiv(nr, GT);
    if (d.sym) trap_decode(t, tm, tm, ti, tj);    // symmetric mode: F22 is symmetric, its lower triangle is all the parent reads
    else { ti = t % tm; tj = t / tm; }
    p.A = Lp + ns; p.lda = m; p.B = Up + (int64_t)ns * ns; p.ldb = ns; p.C = d.T + d.soff[s]; p.ldc = nr; p.M = p.N = nr; p.K = ns;
    gemm_tile<G_SUB>(p, ti, tj, lds);
  } else {
    // selective inversion of a large supernode: IL = L11^-1 (unit block lower triangular) and IU = U11^-1 (block upper
    // triangular, diagonal blocks D_k^-1), ns x ns each at d.V + voff; the strictly upper BLOCKS of IL carry D_k^-1 U[k, :]
    // while IU is being built (the products never touch them)
    zc* IL = d.V + d.voff[s];
    zc* IU = IL + (int64_t)ns * ns;
    if (KIND == K_INVU_SCALE) {                  // IL[k, c1:ns] = D_k^-1 U[k, c1:ns] for every pivot block k but the last
      int k = 0, c1 = NB;
      for (; c1 < ns; c1 += NB, ++k) { const int c = cdiv(ns - c1, GT); if (t < c) break; t -= c; }
      const int c0 = k * NB;
      p.A = d.W + d.ioff[s] + (int64_t)k * NB * NB; p.lda = NB; p.B = Up + c0 + (int64_t)c1 * ns; p.ldb = ns;
      p.C = IL + c0 + (int64_t)c1 * ns; p.ldc = ns; p.M = NB; p.N = ns - c1; p.K = NB;
      gemm_tile<G_ASSIGN>(p, 0, t, lds);
    } else if (KIND == K_INVL) {                 // row block i = p0: IL[i, 0:c0] = -L[i, 0:c0] IL[0:c0, 0:c0]   (IL lower triangular)
      const int c0 = p0 * NB;
      p.A = Lp + c0; p.lda = m; p.B = IL; p.ldb = ns; p.C = IL + c0; p.ldc = ns;
      p.M = min(NB, ns - c0); p.N = c0; p.K = c0; p.K0 = t * GT;
      gemm_tile<G_NEG>(p, 0, t, lds);
    } else {                                     // row block k = nb - 1 - p0: IU[k, c1:ns] = -(D_k^-1 U[k, c1:ns]) IU[c1:ns, c1:ns]   (upper triangular)
      const int k = cdiv(ns, NB) - 1 - p0, c0 = k * NB, c1 = c0 + NB;
      p.A = IL + c0 + (int64_t)c1 * ns; p.lda = ns; p.B = IU + c1 + (int64_t)c1 * ns; p.ldb = ns; p.C = IU + c0 + (int64_t)c1 * ns; p.ldc = ns;
      p.M = NB; p.N = ns - c1; p.K = min(ns - c1, (t + 1) * GT);
      gemm_tile<G_NEG>(p, 0, t, lds);
    }
  }
}

// diagonal blocks of the inverted supernode: IL[k, k] = I, IU[k, k] = D_k^-1; one workgroup per (front, pivot block)
__global__ void __launch_bounds__(256) k_splu_inv_init(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf) {
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int k = blockIdx.x - prefix[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int c0 = k * NB, wd = min(ns, c0 + NB) - c0;
  zc* IL = d.V + d.voff[s];
  zc* IU = IL + (int64_t)ns * ns;
  const zc* Wd = d.W + d.ioff[s] + (int64_t)k * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int a = e & (NB - 1), b = e >> 6;
    if (a < wd && b < wd) {
      IL[(c0 + a) + (int64_t)(c0 + b) * ns] = make_double2(a == b ? 1.0 : 0.0, 0.0);
      IU[(c0 + a) + (int64_t)(c0 + b) * ns] = Wd[a + b * NB];
    }
  }
}

// ------------------------------------------------------------------------------------------------- solves
// out[a] = sum_k M[a, k] v[k] for the stored 64 x 64 inverse (column-major); TRANS: M^T.  All 256 threads; sv, sy in LDS.
template <bool TRANS>
__device__ __forceinline__ void block_apply(const zc* __restrict__ Wb, int w, const zc* sv, zc* sy, zc* sred) {
  const int a = threadIdx.x & 63, q = threadIdx.x >> 6;
  zc acc = make_double2(0.0, 0.0);
  if (a < w)
    for (int k = q * 16; k < min(w, q * 16 + 16); ++k) zfma(acc, TRANS ? Wb[k + a * NB] : Wb[a + k * NB], sv[k]);
  sred[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) sy[a] = zadd(zadd(sred[a], sred[a + 64]), zadd(sred[a + 128], sred[a + 192]));
  __syncthreads();
}

// rows [r0, r1) of a column-major panel (leading dimension ld) times the 64-vector sy, columns c0 .. c0 + wd:
// w[r] -= sum_k P[r, c0 + k] sy[k].  Thread (r = t & 63, q = t >> 6) takes a quarter of the columns; 64 rows per pass.
__device__ __forceinline__ void rows_times_block(const zc* __restrict__ P, int64_t ld, int64_t r0, int64_t r1, int c0, int wd,
                                                 const zc* sy, zc* w, zc* sred) {
  const int lr = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int k0 = q * 16, k1 = min(wd, k0 + 16);
  for (int64_t base = r0; base < r1; base += 64) {
    const int64_t r = base + lr;
    zc acc = make_double2(0.0, 0.0);
    if (r < r1) {
      const zc* pr = P + r + (int64_t)c0 * ld;
#pragma unroll 8
      for (int k = k0; k < k1; ++k) zfma(acc, pr[(int64_t)k * ld], sy[k]);
    }
    sred[threadIdx.x] = acc;
    __syncthreads();
    if (q == 0 && r < r1) {
      const zc t = zadd(zadd(sred[lr], sred[lr + 64]), zadd(sred[lr + 128], sred[lr + 192]));
      w[r].x -= t.x; w[r].y -= t.y;
    }
    __syncthreads();
  }
}

// pivot rows c0 .. c0 + wd of the row panel U (ns x m, leading dimension ns) times w[a0:a1]: out[k] = sum_a U[c0 + k, a] w[a].
// Thread (k = t & 63, q) takes every fourth column; the four partial sums are left in sred[q * 64 + k].
__device__ __forceinline__ void block_times_cols(const zc* __restrict__ U, int64_t ld, int c0, int wd, int64_t a0, int64_t a1,
                                                 const zc* __restrict__ w, zc* sred) {
  const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
  zc acc = make_double2(0.0, 0.0);
  if (k < wd) {
    const zc* pu = U + c0 + k;
#pragma unroll 4
    for (int64_t a = a0 + q; a < a1; a += 4) zfma(acc, pu[a * ld], w[a]);
  }
  sred[threadIdx.x] = acc;
  __syncthreads();
}

// column dots (transposed solves): out[k] = sum_a P[a, c0 + k] w[a] over a in [r0, r1); wave wv takes k = wv, wv + 4, ...
__device__ __forceinline__ void block_dots(const zc* __restrict__ P, int64_t m, int c0, int wd, int64_t r0, int64_t r1, const zc* __restrict__ w, zc* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = wv; k < wd; k += 4) {
    zc acc = make_double2(0.0, 0.0);
    const zc* pc = P + (int64_t)(c0 + k) * m;
    for (int64_t a = r0 + lane; a < r1; a += 64) zfma(acc, pc[a], w[a]);
    acc = zreduce_sub<64>(acc);
    if (lane == 0) out[k] = acc;
  }
}
// w[a] -= sum_k U[c0 + k, a] sy[k] for a in [r0, r1) (transposed forward sweep): one wave per a, lanes over k
__device__ __forceinline__ void cols_dot_block(const zc* __restrict__ U, int64_t ld, int c0, int wd, int64_t r0, int64_t r1, const zc* sy, zc* w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const zc yk = lane < wd ? sy[lane] : make_double2(0.0, 0.0);
  for (int64_t a = r0 + wv; a < r1; a += 4) {
    zc acc = make_double2(0.0, 0.0);
    if (lane < wd) zfma(acc, U[c0 + lane + a * ld], yk);
    acc = zreduce_sub<64>(acc);
    if (lane == 0) { w[a].x -= acc.x; w[a].y -= acc.y; }
  }
}

__device__ __forceinline__ void fwd_gather(const SpluDev& d, int s, int ns, int64_t m, const zc* __restrict__ pb, zc* w) {
  const int first = d.sfirst[s];
  for (int64_t i = threadIdx.x; i < m; i += blockDim.x) w[i] = i < ns ? pb[first + i] : make_double2(0.0, 0.0);
  __syncthreads();
  for (int q = d.child_ptr[s]; q < d.child_ptr[s + 1]; ++q) {
    const int c = d.child_list[q];
    const int nsc = d.sfirst[c + 1] - d.sfirst[c];
    const int64_t nrc = d.sptr[c + 1] - d.sptr[c];
    const zc* wc = d.wk + d.woff[c] + nsc;
    const int32_t* rel = d.rel + d.sptr[c];
    for (int64_t i = threadIdx.x; i < nrc; i += blockDim.x) { zc* t = w + rel[i]; const zc v = wc[i]; t->x += v.x; t->y += v.y; }
    __syncthreads();
  }
}

// (256-thread form, used by the transposed sweeps) forward sweep of one front: w[0:ns] <- y of this supernode (stays in the workspace for the backward sweep),
// w[ns:m] <- update for the ancestors.  L has unit diagonal blocks; the transposed sweep runs on U^T (diagonal blocks D^T).
__global__ void __launch_bounds__(256) k_splu_fwd_small_256(SpluDev d, const int32_t* __restrict__ fronts, int trans, const zc* __restrict__ pb) {
  __shared__ zc sv[NB], sy[NB], sred[256];
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  zc* w = d.wk + d.woff[s];
  fwd_gather(d, s, ns, m, pb, w);
  for (int kb = 0; kb * NB < ns; ++kb) {
    const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
    if (threadIdx.x < wd) sv[threadIdx.x] = w[c0 + threadIdx.x];
    __syncthreads();
    if (!trans) {
      rows_times_block(d.F + d.loff[s], m, c1, m, c0, wd, sv, w, sred);
    } else {
      block_apply<true>(d.W + d.ioff[s] + (int64_t)kb * NB * NB, wd, sv, sy, sred);
      if (threadIdx.x < wd) w[c0 + threadIdx.x] = sy[threadIdx.x];
      cols_dot_block(d.F + d.uoff[s], ns, c0, wd, c1, m, sy, w);
      __syncthreads();
    }
  }
}

// forward sweep of one front: w[0:ns] <- y of this supernode (stays in the workspace for the backward sweep),
// w[ns:m] <- update for the ancestors.  L has unit diagonal blocks; the transposed sweep runs on U^T (diagonal blocks D^T).
// 512 threads: a wave takes whole 64-row tiles (every load of a tile's 64 columns in flight at once, no reduction).
#define SM_T 512
#define SM_W (SM_T / 64)
__global__ void __launch_bounds__(SM_T) k_splu_fwd_small(SpluDev d, const int32_t* __restrict__ fronts, const zc* __restrict__ pb) {
  __shared__ zc sv[NB];
  const bool trans = false;
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  zc* w = d.wk + d.woff[s];
  fwd_gather(d, s, ns, m, pb, w);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int kb = 0; kb * NB < ns; ++kb) {
    const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
    if (threadIdx.x < wd) sv[threadIdx.x] = w[c0 + threadIdx.x];
    __syncthreads();
    if (!trans) {
      const zc* P = d.F + d.loff[s] + (int64_t)c0 * m;
      for (int64_t r = c1 + (int64_t)wv * 64 + lane; r < m; r += SM_T) {
        zc acc = make_double2(0.0, 0.0);
        const zc* pr = P + r;
#pragma unroll 16
        for (int k = 0; k < wd; ++k) zfma(acc, pr[(int64_t)k * m], sv[k]);
        w[r].x -= acc.x; w[r].y -= acc.y;
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_splu_fwd_gather(SpluDev d, const int32_t* __restrict__ fronts, const zc* __restrict__ pb) {
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  fwd_gather(d, s, ns, m, pb, d.wk + d.woff[s]);
}

// large fronts, transposed sweep only, pivot block kb: w[c0:c1] <- D^-T w[c0:c1]
__global__ void __launch_bounds__(256) k_splu_fwd_diag(SpluDev d, const int32_t* __restrict__ fronts, int kb) {
  __shared__ zc sv[NB], sy[NB], sred[256];
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int c0 = kb * NB, wd = min(ns, c0 + NB) - c0;
  zc* w = d.wk + d.woff[s];
  if (threadIdx.x < wd) sv[threadIdx.x] = w[c0 + threadIdx.x];
  __syncthreads();
  block_apply<true>(d.W + d.ioff[s] + (int64_t)kb * NB * NB, wd, sv, sy, sred);
  if (threadIdx.x < wd) w[c0 + threadIdx.x] = sy[threadIdx.x];
}

// large fronts, pivot block kb: w[c1:m] -= L[c1:m, c0:c1] w[c0:c1] (transposed: U[c0:c1, c1:m]^T)
__global__ void __launch_bounds__(256) k_splu_fwd_upd(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf,
                                                      int kb, int trans) {
  __shared__ zc sy[NB], sred[256];
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int t = blockIdx.x - prefix[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
  zc* w = d.wk + d.woff[s];
  if (threadIdx.x < wd) sy[threadIdx.x] = w[c0 + threadIdx.x];
  __syncthreads();
  if (!trans) {
    const int64_t r0 = c1 + (int64_t)t * FWD_ROWS;
    rows_times_block(d.F + d.loff[s], m, r0, min(m, r0 + FWD_ROWS), c0, wd, sy, w, sred);
  } else {
    const int64_t r0 = c1 + (int64_t)t * SOLVE_ROWS;
    cols_dot_block(d.F + d.uoff[s], ns, c0, wd, r0, min(m, r0 + SOLVE_ROWS), sy, w);
  }
}

// (256-thread form, used by the transposed sweeps) backward sweep of one front: w[ns:m] <- x of the border rows, then pivot blocks from last to first
__global__ void __launch_bounds__(256) k_splu_bwd_small_256(SpluDev d, const int32_t* __restrict__ fronts, int trans, zc* __restrict__ x) {
  __shared__ zc sv[NB], sy[NB], sred[256];
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  zc* w = d.wk + d.woff[s];
  {
    const int32_t* rows = d.srow + d.sptr[s];
    for (int64_t i = ns + threadIdx.x; i < m; i += 256) w[i] = x[rows[i - ns]];
    __syncthreads();
  }
  for (int kb = (ns - 1) / NB; kb >= 0; --kb) {
    const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
    if (!trans) {
      block_times_cols(d.F + d.uoff[s], ns, c0, wd, c1, m, w, sred);
      if (threadIdx.x < wd) {
        const int k = threadIdx.x;
        const zc t = zadd(zadd(sred[k], sred[k + 64]), zadd(sred[k + 128], sred[k + 192]));
        sv[k] = zsub(w[c0 + k], t);
      }
      __syncthreads();
      block_apply<false>(d.W + d.ioff[s] + (int64_t)kb * NB * NB, wd, sv, sy, sred);
      if (threadIdx.x < wd) w[c0 + threadIdx.x] = sy[threadIdx.x];
    } else {
      block_dots(d.F + d.loff[s], m, c0, wd, c1, m, w, sy);
      __syncthreads();
      if (threadIdx.x < wd) w[c0 + threadIdx.x] = zsub(w[c0 + threadIdx.x], sy[threadIdx.x]);
    }
    __syncthreads();
  }
  const int first = d.sfirst[s];
  for (int i = threadIdx.x; i < ns; i += 256) x[first + i] = w[i];
}

// backward sweep of one front: w[ns:m] <- x of the border rows, then pivot blocks from last to first
__global__ void __launch_bounds__(SM_T) k_splu_bwd_small(SpluDev d, const int32_t* __restrict__ fronts, zc* __restrict__ x) {
  __shared__ zc sv[NB], sred[SM_T];
  const bool trans = false;
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  zc* w = d.wk + d.woff[s];
  {
    const int32_t* rows = d.srow + d.sptr[s];
    for (int64_t i = ns + threadIdx.x; i < m; i += SM_T) w[i] = x[rows[i - ns]];
    __syncthreads();
  }
  const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int kb = (ns - 1) / NB; kb >= 0; --kb) {
    const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
    if (!trans) {
      // t = w[c0:c1] - U[c0:c1, c1:m] w[c1:m]: lane = pivot row, wave q takes every 8th column
      zc acc = make_double2(0.0, 0.0);
      if (k < wd) {
        const zc* pu = d.F + d.uoff[s] + c0 + k;
#pragma unroll 8
        for (int64_t a = c1 + q; a < m; a += SM_W) zfma(acc, pu[a * ns], w[a]);
      }
      sred[threadIdx.x] = acc;
      __syncthreads();
      if (threadIdx.x < wd) {
        zc t = sred[k];
#pragma unroll
        for (int i = 1; i < SM_W; ++i) t = zadd(t, sred[k + 64 * i]);
        sv[k] = zsub(w[c0 + k], t);
      }
      __syncthreads();
      // x = D^-1 t: row a = lane, wave q takes 8 columns
      const zc* Wb = d.W + d.ioff[s] + (int64_t)kb * NB * NB;
      zc y = make_double2(0.0, 0.0);
      if (k < wd)
        for (int j = q * 8; j < min(wd, q * 8 + 8); ++j) zfma(y, Wb[k + j * NB], sv[j]);
      sred[threadIdx.x] = y;
      __syncthreads();
      if (threadIdx.x < wd) {
        zc t = sred[k];
#pragma unroll
        for (int i = 1; i < SM_W; ++i) t = zadd(t, sred[k + 64 * i]);
        w[c0 + k] = t;
      }
    }
    __syncthreads();
  }
  const int first = d.sfirst[s];
  for (int i = threadIdx.x; i < ns; i += SM_T) x[first + i] = w[i];
}

__global__ void __launch_bounds__(256) k_splu_bwd_gather(SpluDev d, const int32_t* __restrict__ fronts, const zc* __restrict__ x) {
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  zc* w = d.wk + d.woff[s];
  const int32_t* rows = d.srow + d.sptr[s];
  for (int64_t i = ns + threadIdx.x; i < m; i += 256) w[i] = x[rows[i - ns]];
}

// large fronts, pivot block kb: partial products of U[c0:c1, c1:m] w[c1:m] (transposed: L[c1:m, c0:c1]^T w[c1:m]) per tile
__global__ void __launch_bounds__(256) k_splu_bwd_dot(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf,
                                                      int kb, int trans) {
  __shared__ zc sred[256];
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int t = blockIdx.x - prefix[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
  const zc* w = d.wk + d.woff[s];
  zc* out = d.part + d.poff[s] + (int64_t)t * NB;
  if (!trans) {
    const int64_t a0 = c1 + (int64_t)t * BWD_COLS;
    block_times_cols(d.F + d.uoff[s], ns, c0, wd, a0, min(m, a0 + BWD_COLS), w, sred);
    if (threadIdx.x < wd) {
      const int k = threadIdx.x;
      out[k] = zadd(zadd(sred[k], sred[k + 64]), zadd(sred[k + 128], sred[k + 192]));
    }
  } else {
    const int64_t r0 = c1 + (int64_t)t * SOLVE_ROWS;
    block_dots(d.F + d.loff[s], m, c0, wd, r0, min(m, r0 + SOLVE_ROWS), w, out);
  }
}

// large fronts, pivot block kb: w[c0:c1] <- D^-1 (w[c0:c1] - sum of the partial products)   (transposed: no D)
__global__ void __launch_bounds__(256) k_splu_bwd_diag(SpluDev d, const int32_t* __restrict__ fronts, int kb, int trans, zc* __restrict__ x) {
  __shared__ zc sv[NB], sy[NB], sred[256];
  const int s = fronts[blockIdx.x];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int64_t m = ns + (d.sptr[s + 1] - d.sptr[s]);
  const int c0 = kb * NB, c1 = min(ns, c0 + NB), wd = c1 - c0;
  zc* w = d.wk + d.woff[s];
  const int ntile = cdiv((int)(m - c1), trans ? SOLVE_ROWS : BWD_COLS);
  if (threadIdx.x < wd) {
    zc acc = make_double2(0.0, 0.0);
    const zc* pp = d.part + d.poff[s] + threadIdx.x;
    for (int t = 0; t < ntile; ++t) acc = zadd(acc, pp[(int64_t)t * NB]);
    sv[threadIdx.x] = zsub(w[c0 + threadIdx.x], acc);
  }
  __syncthreads();
  if (!trans) block_apply<false>(d.W + d.ioff[s] + (int64_t)kb * NB * NB, wd, sv, sy, sred);
  if (threadIdx.x < wd) {
    const zc v = trans ? sv[threadIdx.x] : sy[threadIdx.x];
    w[c0 + threadIdx.x] = v; x[d.sfirst[s] + c0 + threadIdx.x] = v;
  }
}

// ---- large fronts, A x = b: every product is "64 rows of a column-major matrix times a vector", tiled 64 rows x GV_K columns;
// the partial sums of a row's tiles are added in a fixed order by k_splu_gv_reduce (no atomics).  Per front and sweep:
//   forward   y = IL w[0:ns]            (GV_IL, lower triangular: tiles right of the diagonal are skipped)
//             w[ns:m] -= L21 y          (GV_L21)
//   backward  t = y - U12 w[ns:m]       (GV_U12)
//             x = IU t                  (GV_IU, upper triangular)
// y / t live in the ns entries behind the front's m workspace entries.
__global__ void __launch_bounds__(256) k_splu_gv_part(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf, int what) {
  __shared__ zc sred[256];
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int t = blockIdx.x - prefix[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int nr = (int)(d.sptr[s + 1] - d.sptr[s]);
  const int64_t m = ns + nr;
  const zc* w = d.wk + d.woff[s];
  const zc* P; const zc* v; int64_t ld; int R, C;
  if (what == GV_IL) { P = d.V + d.voff[s]; ld = ns; R = ns; C = ns; v = w; }
  else if (what == GV_L21) { P = d.F + d.loff[s] + ns; ld = m; R = nr; C = ns; v = w + m; }
  else if (what == GV_U12) { P = d.F + d.uoff[s] + (int64_t)ns * ns; ld = ns; R = ns; C = nr; v = w + ns; }
  else { P = d.V + d.voff[s] + (int64_t)ns * ns; ld = ns; R = ns; C = ns; v = w + m; }
  const int tr = cdiv(R, 64);
  const int ti = t % tr, c = t / tr;
  int k0 = c * GV_K, k1 = min(C, k0 + GV_K);
  if (what == GV_IL) k1 = min(k1, min(ns, (ti + 1) * 64));      // columns beyond the row tile's diagonal block are zero
  if (what == GV_IU) k0 = max(k0, ti * 64);
  if (k0 >= k1) return;                                          // (the reduction skips this tile too)
  const int lr = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int r = ti * 64 + lr;
  zc acc = make_double2(0.0, 0.0);
  if (r < R) {
    const zc* pr = P + r;
#pragma unroll 16
    for (int k = k0 + q; k < k1; k += 4) zfma(acc, pr[(int64_t)k * ld], v[k]);
  }
  sred[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && r < R)
    d.part[d.poff[s] + (int64_t)c * R + r] = zadd(zadd(sred[lr], sred[lr + 64]), zadd(sred[lr + 128], sred[lr + 192]));
}

__global__ void __launch_bounds__(256) k_splu_gv_reduce(SpluDev d, const int32_t* __restrict__ fronts, const int32_t* __restrict__ prefix, int nf, int what,
                                                        zc* __restrict__ x) {
  const int f = find_front(prefix, nf, blockIdx.x);
  const int s = fronts[f];
  const int ns = d.sfirst[s + 1] - d.sfirst[s];
  const int nr = (int)(d.sptr[s + 1] - d.sptr[s]);
  const int64_t m = ns + nr;
  zc* w = d.wk + d.woff[s];
  const int R = what == GV_L21 ? nr : ns, C = what == GV_U12 ? nr : ns;
  const int r = (blockIdx.x - prefix[f]) * 256 + threadIdx.x;
  if (r >= R) return;
  const int ti = r >> 6;
  int ca = 0, cb = cdiv(C, GV_K);
  if (what == GV_IL) cb = min(cb, (min(ns, (ti + 1) * 64) - 1) / GV_K + 1);
  if (what == GV_IU) ca = (ti * 64) / GV_K;
  zc acc = make_double2(0.0, 0.0);
  const zc* pp = d.part + d.poff[s] + r;
  for (int c = ca; c < cb; ++c) acc = zadd(acc, pp[(int64_t)c * R]);
  if (what == GV_IL) w[m + r] = acc;
  else if (what == GV_L21) w[ns + r] = zsub(w[ns + r], acc);
  else if (what == GV_U12) w[m + r] = zsub(w[m + r], acc);
  else { w[r] = acc; x[d.sfirst[s] + r] = acc; }
}

__global__ void __launch_bounds__(256) k_splu_permute(int32_t n, const int32_t* __restrict__ perm, const zc* __restrict__ in, zc* __restrict__ out, int scatter) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (scatter) out[perm[i]] = in[i]; else out[i] = in[perm[i]];
  }
}

// ------------------------------------------------------------------------------------------------- host side
typedef int (*tile_fn)(int, int, int, int);

static void add_launch(std::vector<Launch>& plan, std::vector<int32_t>& lists, int kind, const std::vector<int32_t>& fronts,
                       const SpluSymbolic& S, int p0, tile_fn tiles) {
  if (fronts.empty()) return;
  Launch L; L.kind = kind; L.list = (int64_t)lists.size(); L.nf = (int32_t)fronts.size(); L.p0 = p0;
  lists.insert(lists.end(), fronts.begin(), fronts.end());
  if (tiles) {
    int32_t acc = 0;
    lists.push_back(0);
    for (int32_t s : fronts) {
      const int ns = S.sfirst[s + 1] - S.sfirst[s], nr = (int)(S.sptr[s + 1] - S.sptr[s]);
      acc += tiles(kind, ns, nr, p0);
      lists.push_back(acc);
    }
    L.grid = acc;
    if (kind == K_SUPD || kind == K_EXTEND) {       // symmetric mode: the lower triangle of tiles only
      L.list_sym = (int64_t)lists.size();
      int32_t a2 = 0;
      lists.push_back(0);
      for (int32_t s : fronts) {
        const int tm = cdiv((int)(S.sptr[s + 1] - S.sptr[s]), GT);
        a2 += tm * (tm + 1) / 2;
        lists.push_back(a2);
      }
      L.grid_sym = a2;
    }
  } else {
    L.grid = L.nf;
  }
  if (L.grid > 0) plan.push_back(L);
}

static void build_plans(hx_splu* h, std::vector<int32_t>& lists) {
  const SpluSymbolic& S = h->sym;
  auto ns_of = [&](int s) { return S.sfirst[s + 1] - S.sfirst[s]; };
  auto nr_of = [&](int s) { return (int)(S.sptr[s + 1] - S.sptr[s]); };
  h->poff.assign(S.nsuper, 0);
  h->voff.assign(S.nsuper, -1);
  int64_t pcur = 0, vcur = 0;
  std::vector<char> big(S.nsuper, 0);
  std::vector<int32_t> all_big;
  h->inv_flops = 0.0;
  for (int s = 0; s < S.nsuper; ++s) {
    const int64_t ns = ns_of(s), nr = nr_of(s), m = ns + nr;
    if (m * ns > BIG_FRONT_ELEMS) {
      big[s] = 1; all_big.push_back(s);
      const int64_t big_dim = std::max(ns, nr);
      h->poff[s] = pcur; pcur += std::max<int64_t>(big_dim * cdiv((int)big_dim, GV_K), (int64_t)cdiv((int)m, BWD_COLS) * NB);
      h->voff[s] = vcur; vcur += 2 * ns * ns;
      h->inv_flops += 8.0 * ((double)ns * ns * ns / 3.0 + (double)NB * ns * ns / 2.0);
    }
  }
  h->part_elems = pcur; h->inv_elems = vcur;
  // ---- selective inversion of the large supernodes (all levels together: the inverses only serve the solves)
  {
    int max_nb = 0;
    for (int s : all_big) max_nb = std::max(max_nb, cdiv(ns_of(s), NB));
    add_launch(h->iplan, lists, K_INV_INIT, all_big, S, 0, splu_tiles);
    std::vector<int32_t> sel;
    for (int s : all_big) if (ns_of(s) > NB) sel.push_back(s);
    add_launch(h->iplan, lists, K_INVU_SCALE, sel, S, 0, splu_tiles);
    for (int t = 1; t < max_nb; ++t) {
      sel.clear();
      for (int s : all_big) if (cdiv(ns_of(s), NB) > t) sel.push_back(s);
      add_launch(h->iplan, lists, K_INVL, sel, S, t, splu_tiles);
      add_launch(h->iplan, lists, K_INVU, sel, S, t, splu_tiles);
    }
  }
  for (int l = 0; l < S.nlevels; ++l) {
    std::vector<int32_t> fr(S.level_list.begin() + S.level_ptr[l], S.level_list.begin() + S.level_ptr[l + 1]);
    // ---- factorisation
    std::vector<int32_t> sel;
    for (int s : fr) if (nr_of(s) > 0) sel.push_back(s);
    add_launch(h->fplan, lists, K_ZERO, sel, S, 0, splu_tiles);
    for (int r = 0;; ++r) {
      sel.clear();
      for (int s : fr) if (S.child_ptr[s + 1] - S.child_ptr[s] > r) sel.push_back(S.child_list[S.child_ptr[s] + r]);
      if (sel.empty()) break;
      add_launch(h->fplan, lists, K_EXTEND, sel, S, 0, splu_tiles);
    }
    int max_ns = 0;
    for (int s : fr) max_ns = std::max(max_ns, ns_of(s));
    for (int kb = 0; kb * NB < max_ns; ++kb) {
      const int c0 = kb * NB;
      sel.clear();
      for (int s : fr) if (ns_of(s) > c0) sel.push_back(s);
      if (c0 % NBO == 0 && c0 > 0) add_launch(h->fplan, lists, K_LEFT, sel, S, c0 / NBO, splu_tiles);
      add_launch(h->fplan, lists, K_DIAG, sel, S, kb, nullptr);
      std::vector<int32_t> sp, si;
      for (int s : sel) {
        const int ns = ns_of(s), m = ns + nr_of(s), c1 = std::min(ns, c0 + NB), cj1 = std::min(ns, (c0 / NBO + 1) * NBO);
        if (m > c1) sp.push_back(s);
        if (cj1 > c1) si.push_back(s);
      }
      add_launch(h->fplan, lists, K_PANEL, sp, S, kb, splu_tiles);
      add_launch(h->fplan, lists, K_INNER, si, S, kb, splu_tiles);
    }
    sel.clear();
    for (int s : fr) if (nr_of(s) > 0) sel.push_back(s);
    add_launch(h->fplan, lists, K_SUPD, sel, S, 0, splu_tiles);
  }
  // ---- solves: [0] = A x = b, [1] = A^T x = b
  for (int tr = 0; tr < 2; ++tr) {
    tile_fn tiles = tr ? splu_tiles_trans : splu_tiles;
    for (int l = 0; l < S.nlevels; ++l) {
      std::vector<int32_t> small, large, sel;
      for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) { const int s = S.level_list[k]; (big[s] ? large : small).push_back(s); }
      add_launch(h->fwd_plan[tr], lists, S_FWD_SMALL, small, S, 0, nullptr);
      add_launch(h->fwd_plan[tr], lists, S_FWD_GATHER, large, S, 0, nullptr);
      if (!tr) {
        std::vector<int32_t> with_border;
        for (int s : large) if (nr_of(s) > 0) with_border.push_back(s);
        add_launch(h->fwd_plan[0], lists, S_GV_PART, large, S, GV_IL, splu_tiles);
        add_launch(h->fwd_plan[0], lists, S_GV_REDUCE, large, S, GV_IL, splu_tiles);
        add_launch(h->fwd_plan[0], lists, S_GV_PART, with_border, S, GV_L21, splu_tiles);
        add_launch(h->fwd_plan[0], lists, S_GV_REDUCE, with_border, S, GV_L21, splu_tiles);
        continue;
      }
      int max_big = 0;
      for (int s : large) max_big = std::max(max_big, ns_of(s));
      for (int kb = 0; kb * NB < max_big; ++kb) {
        sel.clear();
        std::vector<int32_t> su;
        for (int s : large) if (ns_of(s) > kb * NB) { sel.push_back(s); if (ns_of(s) + nr_of(s) > std::min(ns_of(s), kb * NB + NB)) su.push_back(s); }
        if (tr) add_launch(h->fwd_plan[tr], lists, S_FWD_DIAG, sel, S, kb, nullptr);
        add_launch(h->fwd_plan[tr], lists, S_FWD_UPD, su, S, kb, tiles);
      }
    }
    for (int l = S.nlevels - 1; l >= 0; --l) {
      std::vector<int32_t> small, large, sel;
      for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) { const int s = S.level_list[k]; (big[s] ? large : small).push_back(s); }
      add_launch(h->bwd_plan[tr], lists, S_BWD_SMALL, small, S, 0, nullptr);
      add_launch(h->bwd_plan[tr], lists, S_BWD_GATHER, large, S, 0, nullptr);
      if (!tr) {
        std::vector<int32_t> with_border;
        for (int s : large) if (nr_of(s) > 0) with_border.push_back(s);
        add_launch(h->bwd_plan[0], lists, S_GV_PART, with_border, S, GV_U12, splu_tiles);
        add_launch(h->bwd_plan[0], lists, S_GV_REDUCE, large, S, GV_U12, splu_tiles);
        add_launch(h->bwd_plan[0], lists, S_GV_PART, large, S, GV_IU, splu_tiles);
        add_launch(h->bwd_plan[0], lists, S_GV_REDUCE, large, S, GV_IU, splu_tiles);
        continue;
      }
      int max_big = 0;
      for (int s : large) max_big = std::max(max_big, ns_of(s));
      for (int kb = (max_big - 1) / NB; kb >= 0 && max_big > 0; --kb) {
        sel.clear();
        std::vector<int32_t> sd;
        for (int s : large) if (ns_of(s) > kb * NB) { sel.push_back(s); if (ns_of(s) + nr_of(s) > std::min(ns_of(s), kb * NB + NB)) sd.push_back(s); }
        add_launch(h->bwd_plan[tr], lists, S_BWD_DOT, sd, S, kb, tiles);
        add_launch(h->bwd_plan[tr], lists, S_BWD_DIAG, sel, S, kb, nullptr);
      }
    }
  }
}

template <typename T>
static hipError_t upload(T** dst, const std::vector<T>& v) {
  hipError_t e = hx_malloc(dst, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  if (!v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

static void splu_release(hx_splu* h) {
  for (int t = 0; t < 2; ++t) if (h->sweep_graph[t]) { (void)hipGraphExecDestroy(h->sweep_graph[t]); h->sweep_graph[t] = nullptr; }
  void* blocks[] = {h->d_i32, h->d_i64, h->d_lists, h->d_amap, h->d_perm, h->d_status, h->F, h->W, h->T, h->wk, h->part, h->pb, h->xp, h->V, h->d_mirror, h->d_symcheck};
  for (void* b : blocks) (void)hx_free(b);
  h->d_i32 = h->d_lists = h->d_perm = h->d_status = nullptr; h->d_i64 = h->d_amap = nullptr;
  h->F = h->W = h->T = h->wk = h->part = h->pb = h->xp = h->V = nullptr; h->d_mirror = nullptr; h->d_symcheck = nullptr;
}

extern "C" int hx_splu_analyse(hx_splu** out, int32_t n, const int32_t* h_rowptr, const int32_t* h_col,
                               const double* h_coords, int dim, int leaf_size, int flags) {
  HX_REQUIRE(out && h_rowptr && h_col, "null argument");
  hx_splu* h = new hx_splu();
  const int rc = splu_analyse_host(h->sym, n, h_rowptr, h_col, h_coords, dim, leaf_size > 0 ? leaf_size : SPLU_NB, flags);
  if (rc != HX_OK) { delete h; return rc; }
  h->nnz_in = h_rowptr[n];
  *out = h;
  return HX_OK;
}

// uploads the plan; called by the first hx_splu_factor (the analysis itself needs no device)
static int splu_prepare_device(hx_splu* h) {
  if (h->d_i32) return HX_OK;
  const SpluSymbolic& S = h->sym;
  std::vector<int32_t> lists;
  build_plans(h, lists);
  // one int32 and one int64 slab
  std::vector<int32_t> i32;
  std::vector<int64_t> i64;
  auto put32 = [&](const std::vector<int32_t>& v) { const size_t o = i32.size(); i32.insert(i32.end(), v.begin(), v.end()); return o; };
  auto put64 = [&](const std::vector<int64_t>& v) { const size_t o = i64.size(); i64.insert(i64.end(), v.begin(), v.end()); return o; };
  const size_t o_sfirst = put32(S.sfirst), o_srow = put32(S.srow), o_rel = put32(S.rel), o_par = put32(S.sparent),
               o_cp = put32(S.child_ptr), o_cl = put32(S.child_list);
  const size_t o_sptr = put64(S.sptr), o_loff = put64(S.loff), o_uoff = put64(S.uoff), o_ioff = put64(S.ioff), o_soff = put64(S.soff),
               o_woff = put64(S.woff), o_poff = put64(h->poff), o_voff = put64(h->voff);
  HX_HIP(upload(&h->d_i32, i32));
  HX_HIP(upload(&h->d_i64, i64));
  HX_HIP(upload(&h->d_lists, lists));
  HX_HIP(upload(&h->d_amap, S.amap));
  HX_HIP(upload(&h->d_perm, S.perm));
  if (!S.mirror.empty()) HX_HIP(upload(&h->d_mirror, S.mirror));
  HX_HIP(hx_malloc(&h->d_symcheck, 2 * sizeof(double)));
  HX_HIP(hx_malloc(&h->d_status, 4 * sizeof(int32_t)));
  HX_HIP(hx_malloc(&h->F, std::max<int64_t>(S.factor_elems, 1) * sizeof(zc)));
  HX_HIP(hx_malloc(&h->W, std::max<int64_t>(S.inverse_elems, 1) * sizeof(zc)));
  HX_HIP(hx_malloc(&h->wk, std::max<int64_t>(S.work_elems, 1) * sizeof(zc)));
  HX_HIP(hx_malloc(&h->part, std::max<int64_t>(h->part_elems, 1) * sizeof(zc)));
  HX_HIP(hx_malloc(&h->V, std::max<int64_t>(h->inv_elems, 1) * sizeof(zc)));
  HX_HIP(hx_malloc(&h->pb, (size_t)S.n * sizeof(zc)));
  HX_HIP(hx_malloc(&h->xp, (size_t)S.n * sizeof(zc)));
  SpluDev& d = h->dev;
  d.sfirst = h->d_i32 + o_sfirst; d.srow = h->d_i32 + o_srow; d.rel = h->d_i32 + o_rel; d.sparent = h->d_i32 + o_par;
  d.child_ptr = h->d_i32 + o_cp; d.child_list = h->d_i32 + o_cl;
  d.sptr = h->d_i64 + o_sptr; d.loff = h->d_i64 + o_loff; d.uoff = h->d_i64 + o_uoff; d.ioff = h->d_i64 + o_ioff;
  d.soff = h->d_i64 + o_soff; d.woff = h->d_i64 + o_woff; d.poff = h->d_i64 + o_poff; d.voff = h->d_i64 + o_voff;
  d.F = h->F; d.W = h->W; d.T = nullptr; d.wk = h->wk; d.part = h->part; d.V = h->V; d.status = h->d_status;
  return HX_OK;
}

extern "C" int hx_splu_factor(hx_splu* h, const hx_c128* val, hx_stream stream) {
  HX_REQUIRE(h && val, "null argument");
  int rc = splu_prepare_device(h);
  if (rc != HX_OK) return rc;
  hipStream_t st = hx_s(stream);
  const SpluSymbolic& S = h->sym;
  hipEvent_t e0, e1;
  HX_HIP(hipEventCreate(&e0)); HX_HIP(hipEventCreate(&e1));
  HX_HIP(hipEventRecord(e0, st));
  // symmetric mode (A = A^T, checked on the values: complex symmetric Helmholtz operators): the U side of the factorisation is
  // not computed but copied -- half the in-panel products, half of F22, half the extend-add (option splu_symmetric=0: never)
  h->symmetric = false;
  if (h->d_mirror && HX_OPT_INT("splu_symmetric", 1) != 0) {
    HX_HIP(hipMemsetAsync(h->d_symcheck, 0, 2 * sizeof(double), st));
    k_splu_symcheck<<<hx_grid(h->nnz_in, 256), 256, 0, st>>>(h->nnz_in, h->d_mirror, (const zc*)val, (unsigned long long*)h->d_symcheck);
    HX_LAUNCH_CHECK("k_splu_symcheck");
    double chk[2] = {0.0, 0.0};
    HX_HIP(hipMemcpyAsync(chk, h->d_symcheck, sizeof(chk), hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    h->symmetric = chk[0] <= 1e-13 * chk[1];
  }
  h->dev.sym = h->symmetric ? 1 : 0;
  {
    double done = 0.0;
    for (int32_t q = 0; q < S.nsuper; ++q) {
      const double ns = S.sfirst[q + 1] - S.sfirst[q], nr = (double)(S.sptr[q + 1] - S.sptr[q]);
      done += h->symmetric ? 8.0 * (ns * ns * ns / 6.0 + nr * ns * ns / 2.0 + nr * nr * ns / 2.0) : 8.0 * (ns * ns * ns / 3.0 + nr * ns * ns + nr * nr * ns);
    }
    h->flops_done = done + h->inv_flops;
  }
  HX_HIP(hx_malloc(&h->T, std::max<int64_t>(S.temp_elems, 1) * sizeof(zc)));
  h->dev.T = h->T;
  HX_HIP(hipMemsetAsync(h->F, 0, S.factor_elems * sizeof(zc), st));
  HX_HIP(hipMemsetAsync(h->d_status, 0, 4 * sizeof(int32_t), st));
  k_splu_scatter<<<hx_grid(h->nnz_in, 256), 256, 0, st>>>(h->nnz_in, h->d_amap, (const zc*)val, h->F, h->dev.sym);
  HX_LAUNCH_CHECK("k_splu_scatter");
  const size_t gemm_lds = 2 * 2 * GK * GLD * sizeof(zc);
  static bool attr_done = false;
  if (!attr_done) {
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_PANEL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_INNER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_LEFT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_SUPD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_INVU_SCALE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_INVL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    HX_HIP(hipFuncSetAttribute((const void*)k_splu_gemm<K_INVU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds));
    attr_done = true;
  }
  for (const Launch& L : h->fplan) {
    const int32_t* fr = h->d_lists + L.list;
    const bool alt = h->symmetric && L.list_sym >= 0;
    const int32_t* px = alt ? h->d_lists + L.list_sym : fr + L.nf;
    const int32_t grid = alt ? L.grid_sym : L.grid;
    switch (L.kind) {
      case K_ZERO: k_splu_zero<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf); break;
      case K_EXTEND: k_splu_extend<<<grid, 256, 0, st>>>(h->dev, fr, px, L.nf); break;
      case K_DIAG: k_splu_diag<<<L.grid, 512, 0, st>>>(h->dev, fr, L.p0); break;
      case K_PANEL: k_splu_gemm<K_PANEL><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      case K_INNER: k_splu_gemm<K_INNER><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      case K_LEFT: k_splu_gemm<K_LEFT><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      case K_SUPD: k_splu_gemm<K_SUPD><<<grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      default: break;
    }
    HX_LAUNCH_CHECK("hx_splu_factor");
  }
  for (const Launch& L : h->iplan) {
    const int32_t* fr = h->d_lists + L.list;
    const int32_t* px = fr + L.nf;
    switch (L.kind) {
      case K_INV_INIT: k_splu_inv_init<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf); break;
      case K_INVU_SCALE: k_splu_gemm<K_INVU_SCALE><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      case K_INVL: k_splu_gemm<K_INVL><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      case K_INVU: k_splu_gemm<K_INVU><<<L.grid, 256, gemm_lds, st>>>(h->dev, fr, px, L.nf, L.p0); break;
      default: break;
    }
    HX_LAUNCH_CHECK("hx_splu_factor (selective inversion)");
  }
  HX_HIP(hipEventRecord(e1, st));
  int32_t status[4] = {0, 0, 0, 0};
  HX_HIP(hipMemcpyAsync(status, h->d_status, sizeof(status), hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  HX_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  h->factor_seconds = ms * 1e-3;
  h->tiny_pivots = status[0];
  (void)hx_free(h->T); h->T = nullptr; h->dev.T = nullptr;      // the contribution blocks are dead: back to the pool
  h->factored = true;
  return HX_OK;
}

extern "C" int hx_splu_solve(hx_splu* h, const hx_c128* b, hx_c128* x, int transposed, hx_stream stream) {
  HX_REQUIRE(h && b && x, "null argument");
  HX_REQUIRE(h->factored, "hx_splu_factor has not run");
  hipStream_t st = hx_s(stream);
  const SpluSymbolic& S = h->sym;
  // (a factorisation in symmetric mode represents a symmetric matrix exactly -- its U panels are copies of its L panels --: the
  // transposed system is the system itself, and takes the fast sweeps)
  const int tr = transposed && !h->dev.sym ? 1 : 0;
  k_splu_permute<<<hx_grid(S.n, 256), 256, 0, st>>>(S.n, h->d_perm, (const zc*)b, h->pb, 0);
  // The two sweeps are ~100 small launches on fixed internal buffers: captured once into a hipGraph and replayed (option
  // splu_graph=0: plain launches).  Only the two permutation kernels around them see the caller's vectors.
  zc* y = h->xp;
  auto sweeps = [&](hipStream_t st) -> int {
  // forward leaves y of every supernode in its workspace slice, backward turns it into x and writes the permuted solution
    for (const Launch& L : h->fwd_plan[tr]) {
      const int32_t* fr = h->d_lists + L.list;
      const int32_t* px = fr + L.nf;
      switch (L.kind) {
        case S_FWD_SMALL: if (tr) k_splu_fwd_small_256<<<L.grid, 256, 0, st>>>(h->dev, fr, tr, h->pb);
                        else k_splu_fwd_small<<<L.grid, SM_T, 0, st>>>(h->dev, fr, h->pb);
                        break;
        case S_FWD_GATHER: k_splu_fwd_gather<<<L.grid, 256, 0, st>>>(h->dev, fr, h->pb); break;
        case S_FWD_DIAG: k_splu_fwd_diag<<<L.grid, 256, 0, st>>>(h->dev, fr, L.p0); break;
        case S_FWD_UPD: k_splu_fwd_upd<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0, tr); break;
        case S_GV_PART: k_splu_gv_part<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0); break;
        case S_GV_REDUCE: k_splu_gv_reduce<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0, y); break;
        default: break;
      }
    }
    for (const Launch& L : h->bwd_plan[tr]) {
      const int32_t* fr = h->d_lists + L.list;
      const int32_t* px = fr + L.nf;
      switch (L.kind) {
        case S_BWD_SMALL: if (tr) k_splu_bwd_small_256<<<L.grid, 256, 0, st>>>(h->dev, fr, tr, y);
                        else k_splu_bwd_small<<<L.grid, SM_T, 0, st>>>(h->dev, fr, y);
                        break;
        case S_BWD_GATHER: k_splu_bwd_gather<<<L.grid, 256, 0, st>>>(h->dev, fr, y); break;
        case S_BWD_DOT: k_splu_bwd_dot<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0, tr); break;
        case S_BWD_DIAG: k_splu_bwd_diag<<<L.grid, 256, 0, st>>>(h->dev, fr, L.p0, tr, y); break;
        case S_GV_PART: k_splu_gv_part<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0); break;
        case S_GV_REDUCE: k_splu_gv_reduce<<<L.grid, 256, 0, st>>>(h->dev, fr, px, L.nf, L.p0, y); break;
        default: break;
      }
    }
  return HX_OK;
  };
  const bool want_graph = HX_OPT_INT("splu_graph", 1) != 0 && !h->graph_failed;
  if (want_graph && !h->sweep_graph[tr]) {
    // captured on a private stream (the caller's may be the null stream, which cannot capture), replayed on the caller's
    hipGraph_t g = nullptr;
    hipStream_t cs = nullptr;
    bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
    if (ok) ok = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rc = sweeps(cs);
      ok = hipStreamEndCapture(cs, &g) == hipSuccess && rc == HX_OK && g != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&h->sweep_graph[tr], g, nullptr, nullptr, 0) == hipSuccess;
    if (g) (void)hipGraphDestroy(g);
    if (cs) (void)hipStreamDestroy(cs);
    if (!ok) { (void)hipGetLastError(); h->sweep_graph[tr] = nullptr; h->graph_failed = true; }
  }
  if (want_graph && h->sweep_graph[tr]) HX_HIP(hipGraphLaunch(h->sweep_graph[tr], st));
  else { const int rc = sweeps(st); if (rc != HX_OK) return rc; }
  k_splu_permute<<<hx_grid(S.n, 256), 256, 0, st>>>(S.n, h->d_perm, y, (zc*)x, 1);
  HX_LAUNCH_CHECK("hx_splu_solve");
  return HX_OK;
}

extern "C" int hx_splu_info(const hx_splu* h, int64_t* iout, double* dout) {
  HX_REQUIRE(h && iout && dout, "null argument");
  const SpluSymbolic& S = h->sym;
  iout[0] = S.n; iout[1] = S.nsuper; iout[2] = S.nlevels; iout[3] = S.nnz_L; iout[4] = S.factor_elems;
  iout[5] = S.inverse_elems; iout[6] = S.temp_elems; iout[7] = S.work_elems; iout[8] = S.max_front; iout[9] = S.max_ns;
  iout[10] = (int64_t)S.srow.size(); iout[11] = h->nnz_in; iout[12] = h->tiny_pivots;
  iout[13] = (int64_t)h->fplan.size(); iout[14] = (int64_t)h->fwd_plan[0].size(); iout[15] = (int64_t)h->bwd_plan[0].size();
  dout[0] = S.flops; dout[1] = S.analyse_seconds; dout[2] = h->factor_seconds; dout[3] = h->inv_flops;
  dout[4] = h->flops_done; dout[5] = h->symmetric ? 1.0 : 0.0;
  iout[5] = S.inverse_elems + h->inv_elems;
  return HX_OK;
}

template <typename T>
static int copy_out(const std::vector<T>& v, void* h_out, int64_t capacity_bytes, int64_t* count) {
  if (count) *count = (int64_t)v.size();
  if (!h_out) return HX_OK;
  if ((int64_t)(v.size() * sizeof(T)) > capacity_bytes) return hx_fail(HX_ERR_ARG, "hx_splu_symbolic", "buffer too small");
  memcpy(h_out, v.data(), v.size() * sizeof(T));
  return HX_OK;
}

extern "C" int hx_splu_symbolic(const hx_splu* h, const char* name, void* h_out, int64_t capacity_bytes, int64_t* count) {
  HX_REQUIRE(h && name, "null argument");
  const SpluSymbolic& S = h->sym;
#define HX_SPLU_FIELD(f) if (!strcmp(name, #f)) return copy_out(S.f, h_out, capacity_bytes, count)
  HX_SPLU_FIELD(perm); HX_SPLU_FIELD(iperm); HX_SPLU_FIELD(sfirst); HX_SPLU_FIELD(sparent); HX_SPLU_FIELD(sptr);
  HX_SPLU_FIELD(srow); HX_SPLU_FIELD(rel); HX_SPLU_FIELD(slevel); HX_SPLU_FIELD(level_ptr); HX_SPLU_FIELD(level_list);
  HX_SPLU_FIELD(child_ptr); HX_SPLU_FIELD(child_list); HX_SPLU_FIELD(loff); HX_SPLU_FIELD(uoff); HX_SPLU_FIELD(ioff);
  HX_SPLU_FIELD(soff); HX_SPLU_FIELD(woff); HX_SPLU_FIELD(amap);
#undef HX_SPLU_FIELD
  return hx_fail(HX_ERR_ARG, "hx_splu_symbolic", "unknown field");
}

extern "C" int hx_splu_destroy(hx_splu* h) {
  if (!h) return HX_OK;
  splu_release(h);
  delete h;
  return HX_OK;
}
