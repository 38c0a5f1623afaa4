// Deflated COCG on a BLOCK of right-hand sides: hx_block_width() independent recurrences that share every pass over the matrix.
//
// What the reference gets for free from its factorisation -- any number of right-hand sides per MatSolve (eigensolvers.py:50,103;
// the flame columns of flame_matrices.py:170-177, the nev eigenvectors of the last solve) -- the iterative inner solver paid
// vector by vector.  Here the columns of B run the single-reduction recurrence of cocg.hip side by side: one block V-cycle
// (amg.hip: vcycle_blk), one block operator product (block.hip: k_sell_blk), one fused pass for all inner products, one fused
// update.  Every column keeps its OWN scalars (gamma, delta, alpha, beta, deflation coefficients): these are NB independent
// Krylov iterations, not a block-Krylov method -- no rank-deficiency handling is needed and a column that has converged is
// simply masked out of the updates.  Single GPU, SELL-64 fine level, complex64 V-cycle.
#include <cmath>
#include <cstdlib>

#include "amg_internal.h"

struct hx_sell;
extern "C" int hx_block_width(void);
extern "C" int hx_sell_spmm_z(const hx_sell*, const hx_c128*, hx_c128*, const double*, const double*, hx_stream);
int hx_sell_residual_blk_z(const hx_sell*, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_amg_vcycle_blk_buffers(hx_amg*, hx_c64**, hx_c64**);
int hx_amg_vcycle_blk(hx_amg*, hx_stream);
int hx_sell_spmm_z_x32(const hx_sell*, const hx_c64*, hx_c128*, hx_stream);

#define NB 4
#define BK_MAXK 32
#define BK_SCW 80            // scalars per column: [0, k + 3) reduction results, then the state below
#define BK_RHO 40
#define BK_BETA 41
#define BK_ALPHA 42
#define BK_T 44              // [44, 44 + k): deflation coefficients
#define BK_BLOCKS 512        // stage-1 blocks of the reductions

namespace {
__device__ __forceinline__ zc wide(zc v) { return v; }
__device__ __forceinline__ zc wide(cc v) { return make_double2((double)v.x, (double)v.y); }

// Stage 1 of all inner products of one iteration, per column c of the block:
//   j < k : P_j^T u_c        (P: k rows of length n -- the panel C = K U in complex128, or U itself in complex64)
//   EXTRA == 1:  j = k: r_c^T u_c,  k + 1: v_c^T u_c,  k + 2: |r_c|^2
//   EXTRA == 2:  j = k: |u_c|^2  (norms of a block vector)
// partials[(block * ncol + j) * NB + c].  ROW mapping (one lane per row, JT x NB accumulators): the panel -- the largest stream of
// this pass -- is then read once, coalesced; the element mapping of the update kernel was measured here and lost (376 vs 154 us at
// 1.9 M rows: four lanes sharing every panel entry quadruple the load instructions of the panel).
template <typename PT, typename UT, int EXTRA, int JT>
__global__ void __launch_bounds__(256)
k_bdots(int64_t n, int k, const PT* __restrict__ P, int64_t ld, const UT* __restrict__ u, const zc* __restrict__ r,
        const zc* __restrict__ v, zc* __restrict__ partials, int64_t rows_per_block) {
  __shared__ zc red[256 / HX_WAVE][JT][NB];
  const int ncol = k + (EXTRA == 1 ? 3 : EXTRA == 2 ? 1 : 0);
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const int lane = threadIdx.x & (HX_WAVE - 1), wv = threadIdx.x / HX_WAVE;
  for (int j0 = 0; j0 < ncol; j0 += JT) {
    zc acc[JT][NB];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int c = 0; c < NB; ++c) acc[t][c] = make_double2(0.0, 0.0);
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
      zc uc[NB];
      if constexpr (sizeof(UT) == 8) {               // complex64 block: two 16-byte loads
        const float4* q = reinterpret_cast<const float4*>(u + i * NB);
        const float4 a = q[0], b = q[1];
        uc[0] = make_double2(a.x, a.y); uc[1] = make_double2(a.z, a.w); uc[2] = make_double2(b.x, b.y); uc[3] = make_double2(b.z, b.w);
      } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) uc[c] = wide(u[i * NB + c]);
      }
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = j0 + t;
        if (j < k) {
          const zc pj = wide(P[(int64_t)j * ld + i]);
#pragma unroll
          for (int c = 0; c < NB; ++c) zfma(acc[t][c], pj, uc[c]);
        } else if (EXTRA == 1 && j == k) {
#pragma unroll
          for (int c = 0; c < NB; ++c) zfma(acc[t][c], r[i * NB + c], uc[c]);
        } else if (EXTRA == 1 && j == k + 1) {
#pragma unroll
          for (int c = 0; c < NB; ++c) zfma(acc[t][c], v[i * NB + c], uc[c]);
        } else if (EXTRA == 1 && j == k + 2) {
#pragma unroll
          for (int c = 0; c < NB; ++c) { const zc ri = r[i * NB + c]; acc[t][c].x = fma(ri.x, ri.x, acc[t][c].x); acc[t][c].x = fma(ri.y, ri.y, acc[t][c].x); }
        } else if (EXTRA == 2 && j == k) {
#pragma unroll
          for (int c = 0; c < NB; ++c) { acc[t][c].x = fma(uc[c].x, uc[c].x, acc[t][c].x); acc[t][c].x = fma(uc[c].y, uc[c].y, acc[t][c].x); }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const zc w = zreduce_sub<HX_WAVE>(acc[t][c]);
        if (lane == 0) red[wv][t][c] = w;
      }
    __syncthreads();
    if (threadIdx.x < JT * NB) {
      const int t = threadIdx.x / NB, c = threadIdx.x % NB;
      if (j0 + t < ncol) {
        zc w = red[0][t][c];
        for (int q = 1; q < 256 / HX_WAVE; ++q) w = zadd(w, red[q][t][c]);
        partials[((int64_t)blockIdx.x * ncol + j0 + t) * NB + c] = w;
      }
    }
    __syncthreads();
  }
}
// Stage 2: one wave per (j, c): fixed order over the blocks -> sc[c * BK_SCW + j]
__global__ void __launch_bounds__(HX_WAVE)
k_bdots_finish(int nblocks, int ncol, const zc* __restrict__ partials, zc* __restrict__ sc) {
  const int j = blockIdx.x / NB, c = blockIdx.x % NB;
  zc v = make_double2(0.0, 0.0);
  for (int b = threadIdx.x; b < nblocks; b += HX_WAVE) v = zadd(v, partials[((int64_t)b * ncol + j) * NB + c]);
  v = zreduce_sub<HX_WAVE>(v);
  if (threadIdx.x == 0) sc[c * BK_SCW + j] = v;
}

// per column: t = +-E^-1 c and (mode 0) the scalars of the recurrence.  mode 1: Galerkin start (t = -E^-1 U^T r).
// Columns outside `mask` get alpha = beta = 0 and t = 0: their vectors are left alone by the update.
__global__ void __launch_bounds__(64 * NB)
k_bcg_small(int k, const zc* __restrict__ Einv, zc* __restrict__ sc, int first, int mode, unsigned mask) {
  const int c = threadIdx.x / 64, i = threadIdx.x % 64;
  __shared__ zc s_c[NB][BK_MAXK], s_t[NB][BK_MAXK];
  zc* my = sc + c * BK_SCW;
  const bool on = (mask >> c) & 1u;
  if (i < k) s_c[c][i] = my[i];
  __syncthreads();
  if (i < k) {
    zc t = make_double2(0.0, 0.0);
    for (int j = 0; j < k; ++j) zfma(t, Einv[i * k + j], s_c[c][j]);
    if (mode == 1) t = make_double2(-t.x, -t.y);
    if (!on) t = make_double2(0.0, 0.0);
    s_t[c][i] = t;
    my[BK_T + i] = t;
  }
  __syncthreads();
  if (i == 0 && mode == 0) {
    zc beta = make_double2(0.0, 0.0), alpha = make_double2(0.0, 0.0);
    if (on) {
      const zc gamma = my[k];
      zc delta = my[k + 1];
      for (int j = 0; j < k; ++j) delta = zsub(delta, zmul(s_t[c][j], s_c[c][j]));
      if (first) alpha = zdiv(gamma, delta);
      else {
        beta = zdiv(gamma, my[BK_RHO]);
        alpha = zdiv(gamma, zsub(delta, zdiv(zmul(beta, gamma), my[BK_ALPHA])));
      }
      my[BK_RHO] = gamma;
      my[BK_ALPHA] = alpha;
    }
    my[BK_BETA] = beta;
    if (!on) my[BK_ALPHA + 1] = make_double2(0.0, 0.0);   // (alpha of a masked column is read through `mask` below, never from here)
  }
}

// p = (u - U t) + beta p ;  s = (v - C t) + beta s ;  x += alpha p ;  r -= alpha s ; complex64 copy of r -- per column.  ELEMENT
// mapping (one lane per entry of the interleaved block: every block array streams coalesced; the four lanes of a row read the same
// panel entries U_j[i], C_j[i]).  Masked columns are not touched.
template <typename UT>
__global__ void __launch_bounds__(256)
k_bcg_update(int64_t n, int k, const cc* __restrict__ Uf, const zc* __restrict__ Cd, int64_t ld, const UT* __restrict__ u,
             const zc* __restrict__ v, zc* __restrict__ p, zc* __restrict__ sv, zc* __restrict__ x, zc* __restrict__ r,
             cc* __restrict__ r32, const zc* __restrict__ sc, unsigned mask) {
  __shared__ zc ts[NB][BK_MAXK];
  __shared__ zc s_ab[NB][2];
  if (k > 0 && (int)threadIdx.x < NB * k) ts[threadIdx.x / k][threadIdx.x % k] = sc[(threadIdx.x / k) * BK_SCW + BK_T + threadIdx.x % k];
  if (threadIdx.x < NB) { s_ab[threadIdx.x][0] = sc[threadIdx.x * BK_SCW + BK_ALPHA]; s_ab[threadIdx.x][1] = sc[threadIdx.x * BK_SCW + BK_BETA]; }
  __syncthreads();
  const int c = threadIdx.x % NB;                       // (blockDim and the grid stride are multiples of NB)
  if (!((mask >> c) & 1u)) return;
  const zc alpha = s_ab[c][0], beta = s_ab[c][1];
  const bool has_beta = beta.x != 0.0 || beta.y != 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * NB; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / NB;
    zc pn = wide(u[e]), sn = v[e];
    for (int j = 0; j < k; ++j) {
      const cc uf = Uf[(int64_t)j * ld + i];
      const zc uj = make_double2((double)uf.x, (double)uf.y), cj = Cd[(int64_t)j * ld + i], tj = ts[c][j];
      pn.x = fma(-uj.x, tj.x, pn.x); pn.x = fma(uj.y, tj.y, pn.x);
      pn.y = fma(-uj.x, tj.y, pn.y); pn.y = fma(-uj.y, tj.x, pn.y);
      sn.x = fma(-cj.x, tj.x, sn.x); sn.x = fma(cj.y, tj.y, sn.x);
      sn.y = fma(-cj.x, tj.y, sn.y); sn.y = fma(-cj.y, tj.x, sn.y);
    }
    if (has_beta) { zfma(pn, beta, p[e]); zfma(sn, beta, sv[e]); }
    p[e] = pn; sv[e] = sn;
    zc xv = x[e], rv = r[e];
    zfma(xv, alpha, pn);
    rv.x = fma(-alpha.x, sn.x, rv.x); rv.x = fma(alpha.y, sn.y, rv.x);
    rv.y = fma(-alpha.x, sn.y, rv.y); rv.y = fma(-alpha.y, sn.x, rv.y);
    x[e] = xv; r[e] = rv;
    r32[e] = make_float2((float)rv.x, (float)rv.y);
  }
}

// w_c += SIGN * sum_j t_jc P_j   (Galerkin start: x -= U t with t = -E^-1 U^T r, r += C t); element mapping as above
template <typename PT, int SIGN>
__global__ void __launch_bounds__(256)
k_bpanel_axpy(int64_t n, int k, const PT* __restrict__ P, int64_t ld, zc* __restrict__ w, const zc* __restrict__ sc, unsigned mask) {
  __shared__ zc ts[NB][BK_MAXK];
  if (k > 0 && (int)threadIdx.x < NB * k) ts[threadIdx.x / k][threadIdx.x % k] = sc[(threadIdx.x / k) * BK_SCW + BK_T + threadIdx.x % k];
  __syncthreads();
  const int c = threadIdx.x % NB;
  if (!((mask >> c) & 1u)) return;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * NB; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / NB;
    zc acc = w[e];
    for (int j = 0; j < k; ++j) {
      const zc pj = wide(P[(int64_t)j * ld + i]), tj = ts[c][j];
      acc.x = fma((double)SIGN * pj.x, tj.x, acc.x); acc.x = fma(-(double)SIGN * pj.y, tj.y, acc.x);
      acc.y = fma((double)SIGN * pj.x, tj.y, acc.y); acc.y = fma((double)SIGN * pj.y, tj.x, acc.y);
    }
    w[e] = acc;
  }
}
__global__ void k_bz2c(int64_t m, const zc* __restrict__ a, cc* __restrict__ b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x)
    b[i] = make_float2((float)a[i].x, (float)a[i].y);
}
}  // namespace

// complex128 entries of workspace hx_cocg_block needs for vectors of length n
extern "C" size_t hx_cocg_block_ws(int64_t n) {
  return 4 * (size_t)n * NB + (size_t)BK_BLOCKS * (BK_MAXK + 3) * NB + (size_t)NB * BK_SCW + 64;
}

// Solve K X = B for the first `ncols` (<= hx_block_width()) columns of the interleaved block B (n x NB); the other columns of X
// come back zero.  Ksell must carry complex128 values.  h_iters / h_relres: per column (NB entries): iterations and the TRUE
// relative residual at exit.  Columns that do not reach 10 rtol are reported as they are (the caller decides); HX_ERR_BREAKDOWN
// on a non-finite recurrence.  Needs the deflation pairs of the handle prepared (any earlier hx_cocg solve does it), or none.
extern "C" int hx_cocg_block(hx_amg* M, int32_t n, const hx_sell* Ksell, int ncols, const hx_c128* B, hx_c128* X, double rtol,
                             int max_it, hx_c128* ws, int* h_iters, double* h_relres, hx_stream s) {
  HX_REQUIRE(M && n > 0 && Ksell && B && X && ws && ncols >= 1 && ncols <= NB && max_it >= 1 && h_iters && h_relres, "bad argument");
  HX_REQUIRE(M->single && !M->sharded && M->pre == M->post, "block COCG: complex64 V-cycle on one GPU, symmetric smoothing");
  Recycle* rec = (M->rec.kmax > 0 && M->rec.k > 0) ? &M->rec : nullptr;
  HX_REQUIRE(!rec || (rec->n == n && rec->have_E), "deflation pairs not prepared for this operator (run one hx_cocg solve first)");
  const int k = rec ? rec->k : 0;
  HX_REQUIRE(k + 3 <= BK_MAXK + 3 && k <= BK_MAXK, "too many deflation pairs");
  int rc;
  const size_t nn = (size_t)n * NB;
  hx_c128 *r = ws, *p = r + nn, *sv = p + nn, *v = sv + nn, *part = v + nn;
  hx_c128* sc = part + (size_t)BK_BLOCKS * (BK_MAXK + 3) * NB;
  hx_c64 *r32 = nullptr, *z32 = nullptr;
  if ((rc = hx_amg_vcycle_blk_buffers(M, &r32, &z32))) return rc;
  int64_t rpb = ((int64_t)n + BK_BLOCKS - 1) / BK_BLOCKS;
  if (rpb < 1024) rpb = 1024;
  const int nblocks = (int)(((int64_t)n + rpb - 1) / rpb);
  const bool trace = HX_OPT_SET("gmres_trace");
  cd host[NB * BK_SCW];
  auto fetch_col_slot = [&](int slot, double* out) -> int {      // real part of scalar `slot` of every column (SYNC)
    HX_HIP(hipMemcpyAsync(host, sc, sizeof(cd) * (size_t)NB * BK_SCW, hipMemcpyDeviceToHost, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    for (int c = 0; c < NB; ++c) out[c] = host[c * BK_SCW + slot].real();
    return HX_OK;
  };
  auto norms = [&](const hx_c128* w, double* out) -> int {       // |w_c|^2 (SYNC)
    k_bdots<zc, zc, 2, 4><<<nblocks, 256, 0, hx_s(s)>>>(n, 0, nullptr, n, (const zc*)w, nullptr, nullptr, (zc*)part, rpb);
    HX_LAUNCH_CHECK("k_bdots<norm>");
    k_bdots_finish<<<1 * NB, HX_WAVE, 0, hx_s(s)>>>(nblocks, 1, (const zc*)part, (zc*)sc);
    HX_LAUNCH_CHECK("k_bdots_finish");
    return fetch_col_slot(0, out);
  };
  double bn2[NB], bnorm[NB], relres[NB], best[NB];
  int its[NB], since_best[NB];
  if ((rc = norms(B, bn2))) return rc;
  unsigned live = 0;                       // columns that take part at all
  for (int c = 0; c < NB; ++c) {
    bnorm[c] = std::sqrt(bn2[c]);
    its[c] = 0; since_best[c] = 0;
    relres[c] = 0.0;
    if (c < ncols && bnorm[c] > 0.0 && std::isfinite(bnorm[c])) { live |= 1u << c; relres[c] = 1.0; }
    else if (c < ncols && !std::isfinite(bnorm[c])) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg_block", "right-hand side is not finite");
    best[c] = relres[c];
  }
  HX_HIP(hipMemsetAsync(X, 0, sizeof(hx_c128) * nn, hx_s(s)));
  HX_HIP(hipMemcpyAsync(r, B, sizeof(hx_c128) * nn, hipMemcpyDeviceToDevice, hx_s(s)));
  unsigned open = live;                    // columns whose TRUE residual is not yet accepted
  int total = 0;
  for (int cycle = 0; cycle < 4 && open && total < max_it; ++cycle) {
    if (cycle > 0) {                       // true residuals of what the recurrences left
      if ((rc = hx_sell_residual_blk_z(Ksell, X, B, r, s))) return rc;
      double r2[NB];
      if ((rc = norms(r, r2))) return rc;
      for (int c = 0; c < NB; ++c) if ((live >> c) & 1u) {
        relres[c] = std::sqrt(r2[c]) / bnorm[c];
        if (!std::isfinite(relres[c])) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg_block", "non-finite residual");
        if (relres[c] <= 10 * rtol) open &= ~(1u << c);
        if (trace) fprintf(stderr, "[hx_cocg_block] cycle %d column %d true relres %.3e\n", cycle, c, relres[c]);
      }
      if (!open) break;
    }
    unsigned act = open;                   // columns iterating in this cycle
    if (k > 0) {                           // Galerkin correction on span(U): x += U E^-1 U^T r, r -= C E^-1 U^T r
      k_bdots<cc, zc, 0, 4><<<nblocks, 256, 0, hx_s(s)>>>(n, k, (const cc*)rec->Uf, n, (const zc*)r, nullptr, nullptr, (zc*)part, rpb);
      HX_LAUNCH_CHECK("k_bdots<U>");
      k_bdots_finish<<<k * NB, HX_WAVE, 0, hx_s(s)>>>(nblocks, k, (const zc*)part, (zc*)sc);
      k_bcg_small<<<1, 64 * NB, 0, hx_s(s)>>>(k, (const zc*)rec->Einv, (zc*)sc, 1, 1, act);
      k_bpanel_axpy<cc, -1><<<hx_grid((int64_t)nn, 256), 256, 0, hx_s(s)>>>(n, k, (const cc*)rec->Uf, n, (zc*)X, (const zc*)sc, act);
      k_bpanel_axpy<zc, +1><<<hx_grid((int64_t)nn, 256), 256, 0, hx_s(s)>>>(n, k, (const zc*)rec->Cd, n, (zc*)r, (const zc*)sc, act);
      HX_LAUNCH_CHECK("block Galerkin start");
    }
    k_bz2c<<<hx_grid((int64_t)nn, 256), 256, 0, hx_s(s)>>>((int64_t)nn, (const zc*)r, (cc*)r32);
    HX_LAUNCH_CHECK("k_bz2c");
    bool first = true;
    for (int c = 0; c < NB; ++c) { best[c] = relres[c]; since_best[c] = 0; }
    int upd = 0;
    while (act && total < max_it) {
      if ((rc = hx_amg_vcycle_blk(M, s))) return rc;                              // u = M r (complex64 block, in z32)
      if ((rc = hx_sell_spmm_z_x32(Ksell, z32, v, s))) return rc;                 // v = K u, u read as complex64 (exact)
      k_bdots<zc, cc, 1, 4><<<nblocks, 256, 0, hx_s(s)>>>(n, k, rec ? (const zc*)rec->Cd : nullptr, n, (const cc*)z32, (const zc*)r,
                                                          (const zc*)v, (zc*)part, rpb);
      HX_LAUNCH_CHECK("k_bdots");
      k_bdots_finish<<<(k + 3) * NB, HX_WAVE, 0, hx_s(s)>>>(nblocks, k + 3, (const zc*)part, (zc*)sc);
      HX_LAUNCH_CHECK("k_bdots_finish");
      // |r_c|^2 of the residuals this iteration started from: judged BEFORE the update is queued
      if (upd > 0 || k > 0 || cycle > 0) {
        double r2[NB];
        if ((rc = fetch_col_slot(k + 2, r2))) return rc;
        for (int c = 0; c < NB; ++c) if ((act >> c) & 1u) {
          relres[c] = std::sqrt(r2[c]) / bnorm[c];
          if (upd > 0) ++its[c];
          if (trace) fprintf(stderr, "[hx_cocg_block] it %3d column %d relres %.3e\n", its[c], c, relres[c]);
          if (!std::isfinite(relres[c])) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg_block", "breakdown (p^T K p = 0 or r^T z = 0)");
          if (relres[c] <= rtol) { act &= ~(1u << c); continue; }
          if (relres[c] < best[c]) { best[c] = relres[c]; since_best[c] = 0; }
          else if (++since_best[c] >= 6 || (since_best[c] >= 3 && best[c] <= 1e3 * rtol)) act &= ~(1u << c);   // stalls: the next cycle judges
          else if (relres[c] > 1e6 * best[c]) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg_block", "diverging recurrence");
        }
        if (!act) break;
      }
      k_bcg_small<<<1, 64 * NB, 0, hx_s(s)>>>(k, rec ? (const zc*)rec->Einv : nullptr, (zc*)sc, first ? 1 : 0, 0, act);
      HX_LAUNCH_CHECK("k_bcg_small");
      k_bcg_update<cc><<<hx_grid((int64_t)nn, 256), 256, 0, hx_s(s)>>>(n, k, rec ? (const cc*)rec->Uf : nullptr, rec ? (const zc*)rec->Cd : nullptr, n,
                                                             (const cc*)z32, (const zc*)v, (zc*)p, (zc*)sv, (zc*)X, (zc*)r, (cc*)r32,
                                                             (const zc*)sc, act);
      HX_LAUNCH_CHECK("k_bcg_update");
      first = false;
      ++upd; ++total;
    }
  }
  if (open) {                              // limits reached: report the true residuals of what we have
    if ((rc = hx_sell_residual_blk_z(Ksell, X, B, r, s))) return rc;
    double r2[NB];
    if ((rc = norms(r, r2))) return rc;
    for (int c = 0; c < NB; ++c) if ((live >> c) & 1u) relres[c] = std::sqrt(r2[c]) / bnorm[c];
  }
  for (int c = 0; c < NB; ++c) { h_iters[c] = its[c]; h_relres[c] = relres[c]; }
  return HX_OK;
}
