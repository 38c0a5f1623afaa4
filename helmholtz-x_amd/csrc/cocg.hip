// Deflated, preconditioned COCG (conjugate orthogonal conjugate gradients) for the shifted operator K (complex128).
//
// K = A + sigma B + sigma^2 C of the shift-invert solve (eigensolvers.py:49-50,102-103 -> PCLU in the reference) is
// complex SYMMETRIC (K^T = K, not Hermitian), and the V-cycle with equal pre/post Chebyshev smoothing and R = P^T is a
// complex symmetric operator too.  COCG is CG in the bilinear form x^T y: three-term recurrences, so no Gram-Schmidt
// panel -- the CGS2 passes were 36 % of the kernel time of a 5 M-DOF fixed-point call with flexible GMRES.  Per
// iteration: 1 operator product, 1 V-cycle, 2 reductions and 2 fused vector updates.
//
// Deflation replaces the subspace recycling of hx_gmres: with pairs (U, C = K U) and E = U^T K U every search direction
// is kept K-conjugate to span(U),  p = z - U E^-1 C^T z + beta p,  and the iteration starts from the Galerkin guess
// x0 = U E^-1 U^T b.  The recurrence needs K U = C to hold to working precision: with complex64 MIRRORS of the harvested
// pairs the residual leaves the deflated space at the 1e-8 level and the iteration diverges from there (measured).
// So U is ROUNDED to complex64 once (the rounded vectors are exactly representable, and any basis of nearly the same
// space deflates as well), and C = K U is recomputed from the rounded vectors in complex128: an exactly consistent pair
// whose U half streams 8 bytes per entry.  All scalars (rho, alpha, beta, the k deflation coefficients) live on the
// device; the host reads one number per iteration (|r|^2) for the stopping test.
//
// A breakdown (p^T K p = 0 or r^T z = 0: possible in the bilinear form) or a stall is reported as HX_ERR_BREAKDOWN; the
// caller then repeats the solve with hx_gmres.
#include <cmath>
#include <cstdlib>

#include "amg_internal.h"

struct hx_sell;
extern "C" int hx_sell_spmv_z(const hx_sell*, const hx_c128*, hx_c128*, const double*, const double*, hx_stream);
int hx_sell_residual_z(const hx_sell*, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_sell_has_parts(const hx_sell*);
int hx_sell_spmv_z_part(const hx_sell*, int, const hx_c128*, hx_c128*, const double*, const double*, hx_stream);
int hx_sell_residual_z_part(const hx_sell*, int, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_csr_residual_z(int32_t, int64_t, const int32_t*, const int32_t*, const hx_c128*, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_zpanel_bdot(int64_t, int, const hx_c128*, int64_t, const hx_c128*, const hx_c128*, hx_c128*, hx_c128*, hx_stream);
int hx_cast_c2z(int64_t, const float*, hx_c128*, hx_stream);
int hx_zpanel_bdot_c64(int64_t, int, const hx_c64*, int64_t, const hx_c128*, const hx_c128*, hx_c128*, hx_c128*, hx_stream);
int hx_zpanel_bdot_w32(int64_t, int, const hx_c128*, int64_t, const hx_c64*, const hx_c128*, hx_c128*, hx_c128*, hx_stream);
int hx_zpanel_bdot_partials(int64_t, int, const hx_c128*, int64_t, const hx_c128*, const hx_c128*, hx_c128*, hx_stream);
int hx_zpanel_bdot_w32_partials(int64_t, int, const hx_c128*, int64_t, const hx_c64*, const hx_c128*, hx_c128*, hx_stream);
int hx_cast_z2c(int64_t, const hx_c128*, float*, hx_stream);
int hx_amg_vcycle_buffers(hx_amg*, hx_c64**, hx_c64**);
int hx_amg_vcycle_c64(hx_amg*, hx_stream);
int hx_amg_level0_diag(hx_amg*, const hx_c64**, hx_c64**, float*);
void hx_amg_x1_ready(hx_amg*, bool);
int hx_panel_dot_blocks(int64_t n, int64_t* rows_per_block);
int hx_panel_dot_finish(int nblocks, int k, const hx_c128* partials, hx_c128* out, hx_stream s);
int hx_sell_spmv_z_dot(const hx_sell* h, const hx_c128* x, hx_c128* y, hx_c128* dot, hx_stream s);
int hx_csr_spmv_z_dot(int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val, const hx_c128* x,
                      hx_c128* y, hx_c128* partials, int max_blocks, int* n_blocks, hx_stream s);

static const double ONE[2] = {1.0, 0.0}, ZERO[2] = {0.0, 0.0};

// Scalar block on the device (complex128 entries): OUT = raw reduction results (at the FRONT: in row-sharded mode the
// all-reduce callback sums the first `count` entries), ST = state, T = deflation coefficients.
#define CG_OUT 0      // [0, 33): dot results
#define CG_RHO 34     // r^T z of the current iterate
#define CG_BETA 35
#define CG_TMP 36      // [36, 38): scratch
#define CG_ALPHA 38    // single-reduction form: alpha of the previous iteration
#define CG_T 40       // [40, 72): t = E^-1 C^T z
#define CG_SCALARS 72
#define CG_MAXK 32

// x += alpha p, r -= alpha q with alpha = rho / (p^T q) read from the device; per-block partial sums of |r|^2.
__global__ void __launch_bounds__(256)
k_cocg_update(int64_t n, const zc* __restrict__ p, const zc* __restrict__ q, zc* __restrict__ x, zc* __restrict__ r,
              cc* __restrict__ r32 /* may be NULL: complex64 copy of the new residual = the V-cycle's input */,
              const zc* __restrict__ sc, zc* __restrict__ partials, int64_t rows_per_block,
              const zc* __restrict__ pq_partials /* may be NULL */, int pq_blocks,
              const cc* __restrict__ dinv0 = nullptr /* with x1_0: also the cycle's first smoothing step x1 = a0 D^-1 r */,
              cc* __restrict__ x1_0 = nullptr, float a0 = 0.f) {
  __shared__ double red[256 / HX_WAVE];
  __shared__ zc s_pq;
  // p^T q: reduced already (sc[CG_OUT]), or -- single-GPU path -- the stage-1 partials of its panel dot, summed here by every
  // block in the fixed order of k_panel_dot_stage2 (lanes stride over the blocks, then the shuffle tree): same bits
  zc pq;
  if (pq_partials) {
    if (threadIdx.x < HX_WAVE) {
      zc v = make_double2(0.0, 0.0);
      for (int bq = threadIdx.x; bq < pq_blocks; bq += HX_WAVE) v = zadd(v, pq_partials[bq]);
      v = zreduce_sub<HX_WAVE>(v);
      if (threadIdx.x == 0) s_pq = v;
    }
    __syncthreads();
    pq = s_pq;
  } else pq = sc[CG_OUT];
  const zc alpha = zdiv(sc[CG_RHO], pq);
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  double acc = 0.0;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
    zc xv = x[i], rv = r[i];
    const zc pv = p[i], qv = q[i];
    zfma(xv, alpha, pv);
    rv.x = fma(-alpha.x, qv.x, rv.x); rv.x = fma(alpha.y, qv.y, rv.x);
    rv.y = fma(-alpha.x, qv.y, rv.y); rv.y = fma(-alpha.y, qv.x, rv.y);
    x[i] = xv; r[i] = rv;
    if (r32) {
      const cc rs = make_float2((float)rv.x, (float)rv.y);
      r32[i] = rs;
      if (x1_0) x1_0[i] = zscale(a0, zmul(dinv0[i], rs));        // (the arithmetic of k_diag_scale<c64> on the value just stored)
    }
    acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
  }
#pragma unroll
  for (int o = HX_WAVE / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, HX_WAVE);
  if ((threadIdx.x & (HX_WAVE - 1)) == 0) red[threadIdx.x / HX_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = red[0];
    for (int w = 1; w < 256 / HX_WAVE; ++w) v += red[w];
    partials[blockIdx.x] = make_double2(v, 0.0);
  }
}

// After the reduction of [C^T z ; r^T z]: beta = rho_new / rho (0 on the first step), rho = rho_new, t = sign * E^-1 c.
// mode 0: direction update (sign +, with beta); mode 1: Galerkin start (t = -E^-1 U^T b so that x = 0 - U t ... ), no rho.
// Launched with 64 threads (partials == NULL: the reduced scalars are in sc) or with one wave per column, at most 16 waves.
__global__ void __launch_bounds__(1024)
k_cocg_small(int k, const zc* __restrict__ Einv, zc* __restrict__ sc, int first, int mode,
             const zc* __restrict__ partials /* may be NULL */, int nblocks) {
  const int i = threadIdx.x;
  __shared__ zc s_out[CG_MAXK + 1];
  if (partials) {
    // the k + 1 reductions arrive as per-block partials (stage 1 of the panel dot): summed here, one wave per column, in the
    // fixed order of k_panel_dot_stage2 (lanes stride over the blocks, then the shuffle tree) -- same bits, one launch fewer
    const int lane = i & (HX_WAVE - 1), nw = blockDim.x / HX_WAVE;
    for (int j = i / HX_WAVE; j <= k; j += nw) {
      zc v = make_double2(0.0, 0.0);
      for (int b = lane; b < nblocks; b += HX_WAVE) v = zadd(v, partials[(int64_t)b * (k + 1) + j]);
      v = zreduce_sub<HX_WAVE>(v);
      if (lane == 0) { s_out[j] = v; sc[CG_OUT + j] = v; }
    }
  } else if (i <= k) s_out[i] = sc[CG_OUT + i];
  __syncthreads();
  if (i < k) {
    zc t = make_double2(0.0, 0.0);
    for (int j = 0; j < k; ++j) zfma(t, Einv[i * k + j], s_out[j]);
    if (mode == 1) t = make_double2(-t.x, -t.y);
    sc[CG_T + i] = t;
  }
  if (i == 0 && mode == 0) {
    const zc rho_new = s_out[k];
    sc[CG_BETA] = first ? make_double2(0.0, 0.0) : zdiv(rho_new, sc[CG_RHO]);
    sc[CG_RHO] = rho_new;
  }
}

__device__ __forceinline__ zc widen(zc v) { return v; }
__device__ __forceinline__ zc widen(cc v) { return make_double2((double)v.x, (double)v.y); }
// p = z - sum_j t_j U_j + beta p   (U complex64; z complex128 or the V-cycle's complex64 output; t, beta from the scalar block)
template <typename ZT>
__global__ void __launch_bounds__(256)
k_cocg_direction(int64_t n, int k, const cc* __restrict__ Uf, int64_t ldu, const ZT* __restrict__ z, zc* __restrict__ p,
                 const zc* __restrict__ sc) {
  __shared__ zc ts[CG_MAXK];
  if ((int)threadIdx.x < k) ts[threadIdx.x] = sc[CG_T + threadIdx.x];
  __syncthreads();
  const zc beta = sc[CG_BETA];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc acc = widen(z[i]);
    if (beta.x != 0.0 || beta.y != 0.0) zfma(acc, beta, p[i]);
    for (int j = 0; j < k; ++j) {
      const cc u = Uf[(int64_t)j * ldu + i];
      const zc v = make_double2((double)u.x, (double)u.y), tj = ts[j];
      acc.x = fma(-v.x, tj.x, acc.x); acc.x = fma(v.y, tj.y, acc.x);
      acc.y = fma(-v.x, tj.y, acc.y); acc.y = fma(-v.y, tj.x, acc.y);
    }
    p[i] = acc;
  }
}

// x -= sum_j t_j U_j (SIGN = -1, U complex64)  /  r += sum_j t_j C_j (SIGN = +1, C = K U complex128: the residual that goes
// with the start correction of x, without an operator product)
template <typename PT, int SIGN>
__global__ void __launch_bounds__(256)
k_cocg_start(int64_t n, int k, const PT* __restrict__ U, int64_t ldu, zc* __restrict__ x, const zc* __restrict__ sc) {
  __shared__ zc ts[CG_MAXK];
  if ((int)threadIdx.x < k) ts[threadIdx.x] = sc[CG_T + threadIdx.x];
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc acc = x[i];
    for (int j = 0; j < k; ++j) {
      const zc v = widen(U[(int64_t)j * ldu + i]), tj = ts[j];
      acc.x = fma((double)SIGN * v.x, tj.x, acc.x); acc.x = fma(-(double)SIGN * v.y, tj.y, acc.x);
      acc.y = fma((double)SIGN * v.x, tj.y, acc.y); acc.y = fma((double)SIGN * v.y, tj.x, acc.y);
    }
    x[i] = acc;
  }
}

// ---- single-reduction (Chronopoulos-Gear) form of the deflated recurrence, used by the row-sharded solve: ONE all-reduce per
// iteration instead of three.  With u = M r and v = K u (the product is applied to the preconditioned residual, not to the
// direction), one fused pass leaves
//     c = C^T u (k entries),  gamma = r^T u,  nu = v^T u,  |r|^2
// and everything else follows from scalars:  t = E^-1 c,  the deflated quantities are u~ = u - U t and K u~ = v - C t (C = K U
// holds to working precision, see the header), so  delta = (K u~)^T u~ = nu - t^T c  (K^T = K gives v^T U = c^T, and
// t^T E t = t^T c),  beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old), and
//     p = u~ + beta p,   s = K u~ + beta s (= K p by recurrence),   x += alpha p,   r -= alpha s.
// Price: the vector s, and C is streamed twice (dots and update) -- per-rank vectors are short where this form is used and the
// all-reduces it removes are pure latency (two of ~15-20 us per inner iteration of ~0.2 ms on eight ranks).
template <typename UT, int JT>
__global__ void __launch_bounds__(256)
k_cocg_sr_dots(int64_t n, int k, const zc* __restrict__ Cd, int64_t ldc, const UT* __restrict__ u, const zc* __restrict__ r,
               const zc* __restrict__ v, zc* __restrict__ partials, int64_t rows_per_block) {
  __shared__ zc red[256 / HX_WAVE][JT];
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const int lane = threadIdx.x & (HX_WAVE - 1), wv = threadIdx.x / HX_WAVE;
  const int ncol = k + 3;                        // C_0 .. C_{k-1}, r, v (all bilinear with u), |r|^2
  for (int j0 = 0; j0 < ncol; j0 += JT) {
    zc acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) acc[t] = make_double2(0.0, 0.0);
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
      const zc ui = widen(u[i]);
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = j0 + t;
        if (j < k) zfma(acc[t], Cd[(int64_t)j * ldc + i], ui);
        else if (j == k) zfma(acc[t], r[i], ui);
        else if (j == k + 1) zfma(acc[t], v[i], ui);
        else if (j == k + 2) { const zc ri = r[i]; acc[t].x = fma(ri.x, ri.x, acc[t].x); acc[t].x = fma(ri.y, ri.y, acc[t].x); }
      }
    }
#pragma unroll
    for (int t = 0; t < JT; ++t) {
      zc w = zreduce_sub<HX_WAVE>(acc[t]);
      if (lane == 0) red[wv][t] = w;
    }
    __syncthreads();
    if (threadIdx.x < JT && j0 + (int)threadIdx.x < ncol) {
      zc w = red[0][threadIdx.x];
      for (int q = 1; q < 256 / HX_WAVE; ++q) w = zadd(w, red[q][threadIdx.x]);
      partials[(int64_t)blockIdx.x * ncol + j0 + threadIdx.x] = w;
    }
    __syncthreads();
  }
}

// scalars of one iteration from the (all-)reduced [c ; gamma ; nu ; |r|^2] in sc[0 .. k+2]
__global__ void __launch_bounds__(HX_WAVE)
k_cocg_sr_small(int k, const zc* __restrict__ Einv, zc* __restrict__ sc, int first) {
  const int i = threadIdx.x;
  __shared__ zc s_c[CG_MAXK], s_t[CG_MAXK];
  if (i < k) s_c[i] = sc[CG_OUT + i];
  __syncthreads();
  if (i < k) {
    zc t = make_double2(0.0, 0.0);
    for (int j = 0; j < k; ++j) zfma(t, Einv[i * k + j], s_c[j]);
    s_t[i] = t;
    sc[CG_T + i] = t;
  }
  __syncthreads();
  if (i == 0) {
    const zc gamma = sc[CG_OUT + k];
    zc delta = sc[CG_OUT + k + 1];
    for (int j = 0; j < k; ++j) { const zc tc = zmul(s_t[j], s_c[j]); delta = zsub(delta, tc); }
    zc beta = make_double2(0.0, 0.0), alpha;
    if (first) alpha = zdiv(gamma, delta);
    else {
      beta = zdiv(gamma, sc[CG_RHO]);
      alpha = zdiv(gamma, zsub(delta, zdiv(zmul(beta, gamma), sc[CG_ALPHA])));
    }
    sc[CG_BETA] = beta;
    sc[CG_ALPHA] = alpha;
    sc[CG_RHO] = gamma;
  }
}

// p = (u - U t) + beta p ;  s = (v - C t) + beta s ;  x += alpha p ;  r -= alpha s  (+ the complex64 copy of the new residual)
template <typename UT>
__global__ void __launch_bounds__(256)
k_cocg_sr_update(int64_t n, int k, const cc* __restrict__ Uf, const zc* __restrict__ Cd, int64_t ld, const UT* __restrict__ u,
                 const zc* __restrict__ v, zc* __restrict__ p, zc* __restrict__ sv, zc* __restrict__ x, zc* __restrict__ r,
                 cc* __restrict__ r32, const zc* __restrict__ sc) {
  __shared__ zc ts[CG_MAXK];
  if ((int)threadIdx.x < k) ts[threadIdx.x] = sc[CG_T + threadIdx.x];
  __syncthreads();
  const zc beta = sc[CG_BETA], alpha = sc[CG_ALPHA];
  const bool has_beta = beta.x != 0.0 || beta.y != 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    zc pn = widen(u[i]), sn = v[i];
    for (int j = 0; j < k; ++j) {
      const cc uf = Uf[(int64_t)j * ld + i];
      const zc uj = make_double2((double)uf.x, (double)uf.y), cj = Cd[(int64_t)j * ld + i], tj = ts[j];
      pn.x = fma(-uj.x, tj.x, pn.x); pn.x = fma(uj.y, tj.y, pn.x);
      pn.y = fma(-uj.x, tj.y, pn.y); pn.y = fma(-uj.y, tj.x, pn.y);
      sn.x = fma(-cj.x, tj.x, sn.x); sn.x = fma(cj.y, tj.y, sn.x);
      sn.y = fma(-cj.x, tj.y, sn.y); sn.y = fma(-cj.y, tj.x, sn.y);
    }
    if (has_beta) { zfma(pn, beta, p[i]); zfma(sn, beta, sv[i]); }
    p[i] = pn; sv[i] = sn;
    zc xv = x[i], rv = r[i];
    zfma(xv, alpha, pn);
    rv.x = fma(-alpha.x, sn.x, rv.x); rv.x = fma(alpha.y, sn.y, rv.x);
    rv.y = fma(-alpha.x, sn.y, rv.y); rv.y = fma(-alpha.y, sn.x, rv.y);
    x[i] = xv; r[i] = rv;
    if (r32) r32[i] = make_float2((float)rv.x, (float)rv.y);
  }
}

// max_i sum_j |K_ij| (non-negative doubles order like their bit patterns)
__global__ void k_csr_norm_inf(int32_t n, const int32_t* __restrict__ rp, const zc* __restrict__ val,
                               unsigned long long* __restrict__ out) {
  double m = 0.0;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) s += hypot(val[p].x, val[p].y);
    m = fmax(m, s);
  }
#pragma unroll
  for (int o = HX_WAVE / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, HX_WAVE));
  if ((threadIdx.x & (HX_WAVE - 1)) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// ||K||_inf of a device CSR (SYNC; d_scratch: 8 bytes of device memory)
int hx_csr_norm_inf(int32_t n, const int32_t* rowptr, const hx_c128* val, void* d_scratch, double* h_out, hx_stream s) {
  unsigned long long* d = (unsigned long long*)d_scratch;
  HX_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), hx_s(s)));
  k_csr_norm_inf<<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, rowptr, (const zc*)val, d);
  HX_LAUNCH_CHECK("k_csr_norm_inf");
  unsigned long long bits = 0;
  HX_HIP(hipMemcpyAsync(&bits, d, sizeof(bits), hipMemcpyDeviceToHost, hx_s(s)));
  HX_HIP(hipStreamSynchronize(hx_s(s)));
  memcpy(h_out, &bits, sizeof(double));
  return HX_OK;
}

extern "C" size_t hx_cocg_ws(int64_t n) {
  return 5 * (size_t)n + hx_panel_dot_ws(CG_MAXK + 1) + CG_SCALARS + 8;
}

// small dense complex inverse (Gauss-Jordan with partial pivoting), row-major k x k; returns false if singular
static bool invert_small(std::vector<cd>& A, int k) {
  std::vector<cd> I((size_t)k * k, cd(0));
  for (int i = 0; i < k; ++i) I[(size_t)i * k + i] = 1.0;
  for (int c = 0; c < k; ++c) {
    int piv = c;
    for (int r = c + 1; r < k; ++r) if (std::abs(A[(size_t)r * k + c]) > std::abs(A[(size_t)piv * k + c])) piv = r;
    if (std::abs(A[(size_t)piv * k + c]) == 0.0) return false;
    if (piv != c) for (int j = 0; j < k; ++j) { std::swap(A[(size_t)piv * k + j], A[(size_t)c * k + j]); std::swap(I[(size_t)piv * k + j], I[(size_t)c * k + j]); }
    const cd d = 1.0 / A[(size_t)c * k + c];
    for (int j = 0; j < k; ++j) { A[(size_t)c * k + j] *= d; I[(size_t)c * k + j] *= d; }
    for (int r = 0; r < k; ++r) {
      if (r == c) continue;
      const cd f = A[(size_t)r * k + c];
      if (f == cd(0)) continue;
      for (int j = 0; j < k; ++j) { A[(size_t)r * k + j] -= f * A[(size_t)c * k + j]; I[(size_t)r * k + j] -= f * I[(size_t)c * k + j]; }
    }
  }
  A.swap(I);
  return true;
}

static int cocg_impl(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                     const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it, hx_c128* ws,
                     hx_c128* xe, hx_c128* hbuf, int* h_iters, double* h_relres, hx_stream s) {
  // one-shot switch (hx_amg_next_solve_uses_x): consumed by this call whatever its outcome -- read and cleared BEFORE any
  // argument check can return, so that a rejected call never leaves it armed for an unrelated later solve
  const bool given_x0 = M && M->x0_next;
  if (M) M->x0_next = false;
  HX_REQUIRE(M && n > 0 && rowptr && col && val && b && x && ws && max_it >= 1, "bad argument");
  HX_REQUIRE(M->pre == M->post, "hx_cocg needs a symmetric V-cycle (pre == post smoothing steps)");
  const bool sharded = xe != nullptr;
  HX_REQUIRE(!sharded || (M->sharded && hbuf), "row-sharded COCG needs a sharded preconditioner handle");
  HX_REQUIRE(sharded || !M->sharded, "use hx_cocg_sharded with a row-sharded preconditioner");
  Recycle* rec = (M->rec.kmax > 0 && M->rec.k > 0) ? &M->rec : nullptr;
  HX_REQUIRE(!rec || rec->n == n, "deflation pairs were harvested for another n");
  const int k = rec ? rec->k : 0;
  HX_REQUIRE(k <= CG_MAXK, "too many deflation pairs");
  int rc = HX_OK, rc0 = HX_OK;
  hx_c128 *r = ws, *p = r + n, *q = p + n, *z = q + n, *tmp = z + n;
  // complex64 cycle: the residual goes into the cycle's input buffer as complex64 (written by the update kernel) and its
  // result is consumed as complex64: no cast passes, half the bytes for z in the two passes that read it
  hx_c64 *r32 = nullptr, *z32 = nullptr;
  if ((rc0 = hx_amg_vcycle_buffers(M, &r32, &z32))) return rc0;
  // level 0's first smoothing step written by the update kernel (one launch fewer per inner iteration; single GPU, complex64 cycle)
  const hx_c64* d0 = nullptr; hx_c64* x10 = nullptr; float a00 = 0.f;
  if (r32 && !sharded && (rc0 = hx_amg_level0_diag(M, &d0, &x10, &a00))) return rc0;
  hx_amg_x1_ready(M, false);
  hx_c128* part = tmp + n;
  hx_c128* sc = sharded ? hbuf : part + hx_panel_dot_ws(CG_MAXK + 1);
  int64_t rpb;
  const int nblocks = hx_panel_dot_blocks(n, &rpb);
  // row-sharded products: the operand is staged in xe (owned block + ghost block); with the split callbacks the interior
  // slices of the product run while the ghost entries travel
  const bool split = sharded && Ksell && M->halo_begin && M->halo_end && hx_sell_has_parts(Ksell);
  auto Kx = [&](const hx_c128* in, hx_c128* out, const double* a, const double* bt) -> int {
    if (sharded) {
      HX_HIP(hipMemcpyAsync(xe, in, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      if (split) {
        hx_stream side = nullptr;
        if (M->halo_begin(M->ctx, 0, xe, 16, s, &side) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (begin) failed");
        const int rcx = hx_sell_spmv_z_part(Ksell, 1, xe, out, a, bt, side);
        if (M->halo_end(M->ctx, 0, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (end) failed");
        if (rcx) return rcx;
        return hx_sell_spmv_z_part(Ksell, 2, xe, out, a, bt, s);
      }
      if (M->halo(M->ctx, 0, xe, 16, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange callback failed");
      in = xe;
    }
    if (Ksell) return hx_sell_spmv_z(Ksell, in, out, a, bt, s);
    return hx_csr_spmv_z(n, nnz, rowptr, col, val, in, out, a, bt, 0, s);
  };
  auto Kx_staged = [&](hx_c128* out) -> int {                        // out = K xe, the owned block of xe staged by the caller
    if (split) {
      hx_stream side = nullptr;
      if (M->halo_begin(M->ctx, 0, xe, 16, s, &side) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (begin) failed");
      const int rcx = hx_sell_spmv_z_part(Ksell, 1, xe, out, ONE, ZERO, side);
      if (M->halo_end(M->ctx, 0, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (end) failed");
      if (rcx) return rcx;
      return hx_sell_spmv_z_part(Ksell, 2, xe, out, ONE, ZERO, s);
    }
    if (M->halo(M->ctx, 0, xe, 16, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange callback failed");
    if (Ksell) return hx_sell_spmv_z(Ksell, xe, out, ONE, ZERO, s);
    return hx_csr_spmv_z(n, nnz, rowptr, col, val, xe, out, ONE, ZERO, 0, s);
  };
  auto Kres = [&](const hx_c128* in, hx_c128* out) -> int {          // out = b - K in
    if (sharded) {
      HX_HIP(hipMemcpyAsync(xe, in, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      if (split) {
        hx_stream side = nullptr;
        if (M->halo_begin(M->ctx, 0, xe, 16, s, &side) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (begin) failed");
        const int rcx = hx_sell_residual_z_part(Ksell, 1, xe, b, out, side);
        if (M->halo_end(M->ctx, 0, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange (end) failed");
        if (rcx) return rcx;
        return hx_sell_residual_z_part(Ksell, 2, xe, b, out, s);
      }
      if (M->halo(M->ctx, 0, xe, 16, s) != 0) return hx_fail(HX_ERR_COMM, "hx_cocg", "halo exchange callback failed");
      in = xe;
    }
    if (Ksell) return hx_sell_residual_z(Ksell, in, b, out, s);
    return hx_csr_residual_z(n, nnz, rowptr, col, val, in, b, out, s);
  };
  auto reduce = [&](int count, int op = 0) -> int {       // sum (op 1: max) of the first `count` scalars over the ranks
    if (sharded && M->allreduce(M->ctx, sc, 2 * (int64_t)count, 1, op, s) != 0)
      return hx_fail(HX_ERR_COMM, "hx_cocg", "all-reduce callback failed");
    return HX_OK;
  };
  cd host[CG_MAXK + 2];
  const bool fuse = !sharded && HX_OPT_INT("cocg_fuse_reductions", 1) != 0;
  const int PD_NB = 1024;                       // PD_MAXBLOCKS of vecops.hip: room of one column of partials
  auto fetch = [&](int count) -> int {        // first `count` scalars -> host (SYNC)
    HX_HIP(hipMemcpyAsync(host, sc, sizeof(cd) * (size_t)count, hipMemcpyDeviceToHost, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    return HX_OK;
  };
  const bool trace = HX_OPT_SET("gmres_trace");
  // single-reduction form of the recurrence (see k_cocg_sr_dots): row-sharded solves (1, default), never (0), also on one GPU (2)
  const int sr_opt = (int)HX_OPT_INT("cocg_single_reduction", 1);
  const bool single = (sr_opt == 2 || (sr_opt == 1 && sharded)) && k + 3 <= 33;
  const int pq_env = (int)HX_OPT_INT("cocg_pq_in_product", 1);

  // ---- once per harvested set: Cd = K Ut (Ut = the complex64-rounded U), E = Cd^T Ut (bilinear) and its inverse
  if (rec && !rec->have_E) {
    std::vector<cd> E((size_t)k * k);
    for (int j = 0; j < k; ++j) {
      if ((rc = hx_cast_c2z(n, rec->Uf + 2 * (size_t)j * n, tmp, s))) return rc;      // Ut_j widened (exact)
      if ((rc = Kx(tmp, rec->Cd + (size_t)j * n, ONE, ZERO))) return rc;
    }
    for (int j = 0; j < k; ++j) {             // column j: Cd_i^T Ut_j for all i
      if ((rc = hx_cast_c2z(n, rec->Uf + 2 * (size_t)j * n, tmp, s))) return rc;
      if ((rc = hx_zpanel_bdot(n, k, rec->Cd, n, tmp, tmp, sc, part, s))) return rc;
      if ((rc = reduce(k))) return rc;
      if ((rc = fetch(k))) return rc;
      for (int i = 0; i < k; ++i) E[(size_t)i * k + j] = host[i];     // E[i][j] = (K Ut_i)^T Ut_j, symmetric for K^T = K
    }
    if (!invert_small(E, k)) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "singular deflation matrix U^T K U");
    HX_HIP(hipMemcpyAsync(rec->Einv, E.data(), sizeof(cd) * (size_t)k * k, hipMemcpyHostToDevice, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    rec->have_E = true;
  }

  // ---- |b|
  if ((rc = hx_zpanel_dot(n, 0, nullptr, n, b, sc, part, s))) return rc;
  if ((rc = reduce(1))) return rc;
  if ((rc = fetch(1))) return rc;
  const double bnorm = std::sqrt(host[0].real());
  if (!(bnorm > 0.0)) {
    HX_HIP(hipMemsetAsync(x, 0, sizeof(hx_c128) * (size_t)n, hx_s(s)));
    if (h_iters) *h_iters = 0;
    if (h_relres) *h_relres = 0.0;
    return bnorm == 0.0 ? HX_OK : hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "right-hand side is not finite");
  }

  int it = 0;
  double relres = 1.0;
  bool converged = false, x_zero = true;
  M->last_backward = -1.0;
  auto true_residual = [&]() -> int {        // r = b - K x, relres = |r| / |b|  (SYNC)
    if (x_zero) {                            // r = b: the norm is |b| itself -- no pass, no reduction, no host round trip
      HX_HIP(hipMemcpyAsync(r, b, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
      relres = 1.0;
      return HX_OK;
    }
    if ((rc = Kres(x, r))) return rc;        // one pass, no copy of b first
    if ((rc = hx_zpanel_dot(n, 0, nullptr, n, r, sc, part, s))) return rc;
    if ((rc = reduce(1))) return rc;
    if ((rc = fetch(1))) return rc;
    relres = std::sqrt(host[0].real()) / bnorm;
    return HX_OK;
  };
  // normwise backward error |r| / (|K|_inf |x| + |b|) for a residual of relative size rel (SYNC); |K|_inf once per operator
  const double BE_FLOOR = 10 * 2.220446049250313e-16;
  auto backward_error = [&](double rel) -> int {
    if (M->normK_val != (const void*)val) {
      double local = 0.0;
      if ((rc = hx_csr_norm_inf(n, rowptr, val, sc + CG_TMP, &local, s))) return rc;
      if (sharded) {                                            // maximum over the ranks' row blocks = |K|_inf of the whole operator
        cd v(local, 0.0);
        HX_HIP(hipMemcpyAsync(sc + CG_TMP, &v, sizeof(cd), hipMemcpyHostToDevice, hx_s(s)));
        HX_HIP(hipMemcpyAsync(sc, sc + CG_TMP, sizeof(cd), hipMemcpyDeviceToDevice, hx_s(s)));
        HX_HIP(hipStreamSynchronize(hx_s(s)));
        if ((rc = reduce(1, 1))) return rc;
        if ((rc = fetch(1))) return rc;
        local = host[0].real();
      }
      M->normK = local;
      M->normK_val = (const void*)val;
    }
    if ((rc = hx_zpanel_dot(n, 0, nullptr, n, x, sc + CG_OUT + 1, part, s))) return rc;     // slot 0 may hold p^T q
    HX_HIP(hipMemcpyAsync(host, sc + CG_OUT + 1, sizeof(cd), hipMemcpyDeviceToHost, hx_s(s)));
    HX_HIP(hipStreamSynchronize(hx_s(s)));
    double x2 = host[0].real();
    if (sharded) {
      HX_HIP(hipMemcpyAsync(sc, sc + CG_OUT + 1, sizeof(cd), hipMemcpyDeviceToDevice, hx_s(s)));
      if ((rc = reduce(1))) return rc;
      if ((rc = fetch(1))) return rc;
      x2 = host[0].real();
    }
    const double xnorm = std::sqrt(x2);
    M->last_backward = rel * bnorm / (M->normK * xnorm + bnorm);
    if (trace) fprintf(stderr, "[hx_cocg] backward error %.3e  (|K| %.3e |x| %.3e |b| %.3e)\n", M->last_backward, M->normK, xnorm, bnorm);
    return HX_OK;
  };
  // initial guess: zero, or -- requested for this one solve by hx_amg_next_solve_uses_x -- what the caller left in x (a
  // related solution); a guess that does not even bring the residual below a tenth of |b| is dropped
  bool have_r = false;                       // r already holds b - K x of the accepted guess: cycle 0 does not measure it again
  if (given_x0) {
    x_zero = false;
    if ((rc = true_residual())) return rc;
    if (trace) fprintf(stderr, "[hx_cocg] initial guess: true relres %.3e\n", relres);
    if (!std::isfinite(relres) || relres > 0.1) x_zero = true;
    else have_r = true;
  }
  if (x_zero) HX_HIP(hipMemsetAsync(x, 0, sizeof(hx_c128) * (size_t)n, hx_s(s)));
  // A cycle = one run of the recurrences.  It ends when the recurrence residual meets the tolerance, when it reaches
  // the attainable accuracy (a right-hand side close to the eigenvector next to the shift has a solution 1e4..1e5
  // times larger than b, so eps * cond bounds the residual near 1e-11 |b|: progress stops there and the deflated
  // recurrence would drift away again), or when it starts to grow.  Each new cycle measures the TRUE residual and
  // accepts it within 10 rtol (as hx_gmres does), or when the normwise backward error is at rounding level (the
  // attainable accuracy: what a backward-stable direct solve would deliver); otherwise it re-projects on span(U) and
  // restarts from x.
  for (int cycle = 0; cycle < 4 && it < max_it; ++cycle) {
    if (!(cycle == 0 && have_r) && (rc = true_residual())) return rc;
    if (trace) fprintf(stderr, "[hx_cocg] cycle %d start  true relres %.3e\n", cycle, relres);
    if (!std::isfinite(relres)) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "non-finite residual");
    if (relres <= (cycle == 0 ? rtol : 10 * rtol)) { converged = true; break; }
    if (cycle > 0) {
      // The recurrence stopped improving above the tolerance: is this the attainable accuracy?  A backward-stable direct
      // solve (what the reference's PCLU delivers) leaves |r| ~ eps |K| |x|, which exceeds rtol |b| when the shift sits
      // next to an eigenvalue and x is 1e6 times larger than b.  Accept a normwise backward error at rounding level
      // (measured floors sit at 1e-16).
      if ((rc = backward_error(relres))) return rc;
      if (M->last_backward <= BE_FLOOR) { converged = true; break; }
    }
    bool need_r0 = false;
    if (k > 0) {     // Galerkin correction on span(U): x += U E^-1 U^T r, so that U^T r = 0 again
      if ((rc = hx_zpanel_bdot_c64(n, k, (const hx_c64*)rec->Uf, n, r, r, sc, part, s))) return rc;   // U^T r (entry k unused)
      if ((rc = reduce(k))) return rc;
      k_cocg_small<<<1, HX_WAVE, 0, hx_s(s)>>>(k, (const zc*)rec->Einv, (zc*)sc, 1, 1, nullptr, 0);
      HX_LAUNCH_CHECK("k_cocg_small");
      k_cocg_start<cc, -1><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, (const cc*)rec->Uf, n, (zc*)x, (const zc*)sc);
      HX_LAUNCH_CHECK("k_cocg_start");
      if (cycle == 0 && x_zero) {
        // x was zero, so x = U t' and r = b - K U t' = b - C t' exactly (C = K U in complex128): update r from the panel
        // C (k columns) instead of an operator product
        k_cocg_start<zc, +1><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, (const zc*)rec->Cd, n, (zc*)r, (const zc*)sc);
        HX_LAUNCH_CHECK("k_cocg_start<r>");
        x_zero = false;
        if (single) need_r0 = true;          // |r_0| arrives with the first fused reduction of the recurrence
        else {
          if ((rc = hx_zpanel_dot(n, 0, nullptr, n, r, sc, part, s))) return rc;
          if ((rc = reduce(1))) return rc;
          if ((rc = fetch(1))) return rc;
          relres = std::sqrt(host[0].real()) / bnorm;
        }
      } else {
        x_zero = false;
        if ((rc = true_residual())) return rc;
      }
      if (!std::isfinite(relres)) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "non-finite residual");
    }
    if (r32) { if ((rc = hx_cast_z2c(n, r, (float*)r32, s))) return rc; }     // later residuals: written by k_cocg_update
    hx_amg_x1_ready(M, false);                  // (the cycle's input was just rewritten by a cast: an x1 left by an earlier update is stale)
    bool first = true;
    double best = relres, prev = relres;
    int since_best = 0;
    // One residual norm judged (in iteration order): leave = true ends this run of the recurrences.
    bool leave = false;
    auto judge = [&](double r2) -> int {
      ++it;
      relres = std::sqrt(r2) / bnorm;
      if (trace) fprintf(stderr, "[hx_cocg] it %3d  relres %.3e\n", it, relres);
      if (h_iters) *h_iters = it;
      if (h_relres) *h_relres = relres;
      if (!std::isfinite(relres)) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "breakdown (p^T K p = 0 or r^T z = 0)");
      if (relres <= rtol) { leave = true; return HX_OK; }
      if (relres <= 1e3 * rtol && relres > 0.5 * prev) {               // progress slows down near the goal: attainable accuracy?
        if ((rc = backward_error(relres))) return rc;
        if (M->last_backward <= BE_FLOOR) { leave = true; return HX_OK; }
      }
      prev = relres;
      if (relres < best) { best = relres; since_best = 0; }
      else {
        ++since_best;
        if (since_best >= 3 && best <= 1e3 * rtol) leave = true;       // drifting away from a nearly converged iterate
        else if (since_best >= 6) leave = true;                         // stagnation: let the next cycle judge it (backward error)
        else if (relres > 1e6 * best) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "diverging recurrence");
      }
      return HX_OK;
    };
    // The host does not wait for |r_j| while the residual is still far from every stopping rule.  The partials of the update
    // (row-sharded: the all-reduced norm -- every rank judges the same number) land in one of two pinned areas; while the last JUDGED residual is above LAG_MARGIN * rtol (and the recurrence is
    // not stalling) iteration j + 1 is queued first and |r_j| is read after it -- the queue never drains on the host round trip
    // (42 us per inner iteration at 1 M DOF).  None of the rules can fire on |r_j| in that regime except "converged" after a
    // single-step drop by more than LAG_MARGIN; the recurrence then simply ends one (converged) iteration later.  Closer to the
    // tolerance every norm is awaited as before.  HX_COCG_LAG=0: always wait.
    const int lag_env = (int)HX_OPT_INT("cocg_lag", 1);
    const double LAG_MARGIN = 1e4;
    const bool may_lag = lag_env != 0 && !trace;
    if (!M->cg_pin) {
      HX_HIP(hipHostMalloc(&M->cg_pin, 2 * sizeof(cd) * (size_t)PD_NB, hipHostMallocDefault));
      for (int e = 0; e < 2; ++e) HX_HIP(hipEventCreateWithFlags(&M->cg_ev[e], hipEventDisableTiming));
    }
    auto arrived = [&](int slot, double* r2) -> int {      // waits for the partials of one update, sums them in the order of stage 2
      HX_HIP(hipEventSynchronize(M->cg_ev[slot]));
      const cd* hp = (const cd*)M->cg_pin + (size_t)slot * PD_NB;
      if (!fuse) { *r2 = hp[0].real(); return HX_OK; }      // (two-stage form: the reduced -- row-sharded: all-reduced -- norm itself)
      double lane[HX_WAVE];
      for (int l = 0; l < HX_WAVE; ++l) lane[l] = 0.0;
      for (int bq = 0; bq < nblocks; ++bq) lane[bq % HX_WAVE] += hp[bq].real();
      for (int o = HX_WAVE / 2; o > 0; o >>= 1) for (int l = 0; l < o; ++l) lane[l] += lane[l + o];
      *r2 = lane[0];
      return HX_OK;
    };
    if (single) {
      // ---- single-reduction form: per iteration one V-cycle, one product (of the preconditioned residual), ONE all-reduce
      // of k + 3 scalars; |r_j|^2 rides with it, so the norm of the residual the iteration STARTED from is known before its
      // update is queued -- far from the tolerance it is read one iteration late (the queue does not drain), near it the host
      // waits and leaves before the update (x_j and r_j stay consistent; the V-cycle and the product of that last iteration
      // are the price of not sending the norm on its own).
      hx_c128* sv = q;                           // s = K p by recurrence
      hx_c128* v = tmp;                          // v = K u
      auto arrived_sr = [&](int slot, double* r2) -> int {
        HX_HIP(hipEventSynchronize(M->cg_ev[slot]));
        *r2 = ((const cd*)M->cg_pin + (size_t)slot * PD_NB)[0].real();
        return HX_OK;
      };
      int queued = it, upd = 0, pending = -1;
      while (queued < max_it) {
        if (r32) { if ((rc = hx_amg_vcycle_c64(M, s))) return rc; }
        else if ((rc = hx_amg_vcycle(M, r, z, s))) return rc;
        // v = K u
        if (sharded) {
          if (r32) { if ((rc = hx_cast_c2z(n, (const float*)z32, xe, s))) return rc; }
          else HX_HIP(hipMemcpyAsync(xe, z, sizeof(hx_c128) * (size_t)n, hipMemcpyDeviceToDevice, hx_s(s)));
          if ((rc = Kx_staged(v))) return rc;
        } else {
          if (r32) { if ((rc = hx_cast_c2z(n, (const float*)z32, z, s))) return rc; }
          if ((rc = Kx(z, v, ONE, ZERO))) return rc;
        }
        if (r32) k_cocg_sr_dots<cc, 8><<<nblocks, 256, 0, hx_s(s)>>>(n, k, rec ? (const zc*)rec->Cd : nullptr, n, (const cc*)z32,
                                                                     (const zc*)r, (const zc*)v, (zc*)part, rpb);
        else k_cocg_sr_dots<zc, 8><<<nblocks, 256, 0, hx_s(s)>>>(n, k, rec ? (const zc*)rec->Cd : nullptr, n, (const zc*)z,
                                                                 (const zc*)r, (const zc*)v, (zc*)part, rpb);
        HX_LAUNCH_CHECK("k_cocg_sr_dots");
        if ((rc = hx_panel_dot_finish(nblocks, k + 2, part, sc, s))) return rc;
        if ((rc = reduce(k + 3))) return rc;
        const int slot = upd & 1;
        if (upd > 0 || need_r0) {
          HX_HIP(hipMemcpyAsync((cd*)M->cg_pin + (size_t)slot * PD_NB, sc + k + 2, sizeof(cd), hipMemcpyDeviceToHost, hx_s(s)));
          HX_HIP(hipEventRecord(M->cg_ev[slot], hx_s(s)));
        }
        k_cocg_sr_small<<<1, HX_WAVE, 0, hx_s(s)>>>(k, rec ? (const zc*)rec->Einv : nullptr, (zc*)sc, first ? 1 : 0);
        HX_LAUNCH_CHECK("k_cocg_sr_small");
        double r2;
        if (need_r0) {                 // residual of the Galerkin start: known only now (no update has been counted yet)
          need_r0 = false;
          if ((rc = arrived_sr(slot, &r2))) return rc;
          relres = std::sqrt(r2) / bnorm;
          if (!std::isfinite(relres)) return hx_fail(HX_ERR_BREAKDOWN, "hx_cocg", "non-finite residual");
          if (trace) fprintf(stderr, "[hx_cocg] start  relres %.3e\n", relres);
          best = prev = relres;
          if (relres <= rtol) break;   // (the next cycle measures the true residual and accepts it)
        }
        if (pending >= 0) {            // the norm the previous iteration left unread
          if ((rc = arrived_sr(pending, &r2))) return rc;
          pending = -1;
          if ((rc = judge(r2))) return rc;
          if (leave) {                 // (a drop by more than LAG_MARGIN in one step) x is one update further: judge that one too, stop
            if ((rc = arrived_sr(slot, &r2))) return rc;
            if ((rc = judge(r2))) return rc;
            leave = true;
            break;
          }
        }
        if (upd > 0) {
          const bool lag = may_lag && relres > LAG_MARGIN * rtol && since_best <= 3 && queued + 1 < max_it;
          if (lag) pending = slot;
          else {
            if ((rc = arrived_sr(slot, &r2))) return rc;
            if ((rc = judge(r2))) return rc;
            if (leave) break;          // r_j meets the stopping rule: x_j is the iterate that goes with it
          }
        }
        if (r32) k_cocg_sr_update<cc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, rec ? (const cc*)rec->Uf : nullptr,
            rec ? (const zc*)rec->Cd : nullptr, n, (const cc*)z32, (const zc*)v, (zc*)p, (zc*)sv, (zc*)x, (zc*)r, (cc*)r32, (const zc*)sc);
        else k_cocg_sr_update<zc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, rec ? (const cc*)rec->Uf : nullptr,
            rec ? (const zc*)rec->Cd : nullptr, n, (const zc*)z, (const zc*)v, (zc*)p, (zc*)sv, (zc*)x, (zc*)r, (cc*)r32, (const zc*)sc);
        HX_LAUNCH_CHECK("k_cocg_sr_update");
        first = false;
        x_zero = false;
        ++queued; ++upd;
      }
      if (pending >= 0) {
        double r2;
        if ((rc = arrived_sr(pending, &r2))) return rc;
        if ((rc = judge(r2))) return rc;
      }
    } else {
    int queued = it;                 // iterations queued (it = iterations judged)
    int pending = -1;                // landing area of a queued, not yet judged iteration
    int carry = -1;                  // row-sharded: landing area of an iteration whose |r|^2 is still to be all-reduced
    while (queued < max_it) {
      // z = M r ;  [c ; rho] = [C^T z ; r^T z] ;  t = E^-1 c, beta ;  p = z - U t + beta p
      // single GPU: the consumers of the three reductions of an iteration sum the stage-1 partials themselves (k_cocg_small,
      // k_cocg_update, the host for |r|^2) in the order of stage 2: three launches fewer, same bits.  Row-sharded: the reduced
      // scalars have to exist on the device for the all-reduce, so the two-stage form stays.
      if (r32) {
        if ((rc = hx_amg_vcycle_c64(M, s))) return rc;
        if ((rc = fuse ? hx_zpanel_bdot_w32_partials(n, k, k > 0 ? rec->Cd : nullptr, n, z32, r, part, s)
                       : hx_zpanel_bdot_w32(n, k, k > 0 ? rec->Cd : nullptr, n, z32, r, sc, part, s))) return rc;
      } else {
        if ((rc = hx_amg_vcycle(M, r, z, s))) return rc;
        if ((rc = fuse ? hx_zpanel_bdot_partials(n, k, k > 0 ? rec->Cd : nullptr, n, z, r, part, s)
                       : hx_zpanel_bdot(n, k, k > 0 ? rec->Cd : nullptr, n, z, r, sc, part, s))) return rc;
      }
      if (carry >= 0) {              // the local |r|^2 of the iteration before rides with this all-reduce (entry k + 1), then travels
        if ((rc = reduce(k + 2))) return rc;
        HX_HIP(hipMemcpyAsync((cd*)M->cg_pin + (size_t)carry * PD_NB, sc + k + 1, sizeof(cd), hipMemcpyDeviceToHost, hx_s(s)));
        HX_HIP(hipEventRecord(M->cg_ev[carry], hx_s(s)));
        carry = -1;
      } else if ((rc = reduce(k + 1))) return rc;
      k_cocg_small<<<1, fuse ? HX_WAVE * std::min(k + 1, 16) : HX_WAVE, 0, hx_s(s)>>>(
          k, rec ? (const zc*)rec->Einv : nullptr, (zc*)sc, first ? 1 : 0, 0, fuse ? (const zc*)part : nullptr, nblocks);
      HX_LAUNCH_CHECK("k_cocg_small");
      if (r32) k_cocg_direction<cc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, rec ? (const cc*)rec->Uf : nullptr, n,
                                                                          (const cc*)z32, (zc*)p, (const zc*)sc);
      else k_cocg_direction<zc><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, k, rec ? (const cc*)rec->Uf : nullptr, n, (const zc*)z,
                                                                      (zc*)p, (const zc*)sc);
      HX_LAUNCH_CHECK("k_cocg_direction");
      first = false;
      // q = K p ;  alpha = rho / p^T q ;  x += alpha p, r -= alpha q, |r|^2
      // single GPU with a SELL-64 operator: p^T q comes out of the product itself (every wave leaves the sum of its slice, one
      // block adds the slices): no pass over p and q -- 0.388 -> 0.377 ms per inner iteration at 1 M DOF.  At 5 M DOF the
      // product is 1.5 us slower for it and the pass it replaces (36 us) ran out of the Infinity Cache; rounds 4-5 kept the panel
      // dot above 2 M rows (1.4445 vs 1.4474 ms per iteration, inside the noise of one run each); three alternating same-box
      // pairs in round 6: 2246 / 2246 / 2229 ms per call with the panel dot, 2221 / 2225 / 2213 ms with the fused sum -- every
      // size now.  HX_COCG_PQ_IN_PRODUCT=0: never (2: the former "at every size", same as the default now).
      const bool pq_in_product = !sharded && Ksell && pq_env != 0;
      // operators on the CSR kernels (small problems, every launch counts): the product leaves the per-block partials of p^T q,
      // the update sums them like those of the panel dot it replaces (one launch fewer per inner iteration)
      const bool pq_in_csr_product = !sharded && !Ksell && fuse && pq_env != 0;
      int pq_blocks = nblocks;
      if (pq_in_product) {
        if ((rc = hx_sell_spmv_z_dot(Ksell, p, q, sc + CG_OUT, s))) return rc;
      } else if (pq_in_csr_product) {
        if ((rc = hx_csr_spmv_z_dot(n, nnz, rowptr, col, val, p, q, part, PD_NB, &pq_blocks, s))) return rc;
      } else {
        if ((rc = Kx(p, q, ONE, ZERO))) return rc;
        if ((rc = fuse ? hx_zpanel_bdot_partials(n, 0, nullptr, n, q, p, part, s)
                       : hx_zpanel_bdot(n, 0, nullptr, n, q, p, sc, part, s))) return rc;
        if ((rc = reduce(1))) return rc;
      }
      hx_c128* part_r = part + PD_NB;             // |r|^2 partials of the update: behind the p^T q partials it reads
      k_cocg_update<<<nblocks, 256, 0, hx_s(s)>>>(n, (const zc*)p, (const zc*)q, (zc*)x, (zc*)r, (cc*)r32, (const zc*)sc,
                                                  (zc*)part_r, rpb, fuse && !pq_in_product ? (const zc*)part : nullptr, pq_blocks,
                                                  (const cc*)d0, (cc*)x10, a00);
      HX_LAUNCH_CHECK("k_cocg_update");
      if (d0) hx_amg_x1_ready(M, true);           // the next cycle skips its diagonal launch (a cycle that never comes: no harm)
      x_zero = false;
      double r2;
      const int slot = queued & 1;
      ++queued;
      // |r|^2 of the update just queued -> its landing area: second stage on the device and the sum over the ranks in the
      // two-stage form (one scalar travels), the per-block partials themselves in the single-GPU form
      auto norm_travels = [&]() -> int {
        if (!fuse) {
          if ((rc = hx_panel_dot_finish(nblocks, 0, part_r, sc, s))) return rc;
          if ((rc = reduce(1))) return rc;
        }
        HX_HIP(hipMemcpyAsync((cd*)M->cg_pin + (size_t)slot * PD_NB, fuse ? part_r : sc, sizeof(cd) * (size_t)(fuse ? nblocks : 1),
                              hipMemcpyDeviceToHost, hx_s(s)));
        HX_HIP(hipEventRecord(M->cg_ev[slot], hx_s(s)));
        return HX_OK;
      };
      if (pending >= 0) {            // the iteration before this one: its norm arrived long ago
        if ((rc = arrived(pending, &r2))) return rc;
        pending = -1;
        if ((rc = judge(r2))) return rc;
        if (leave) {                 // (a drop by more than LAG_MARGIN in one step) x is one iteration further: judge that one, stop
          if ((rc = norm_travels())) return rc;
          if ((rc = arrived(slot, &r2))) return rc;
          if ((rc = judge(r2))) return rc;
          leave = true;
          break;
        }
      }
      const bool lag = may_lag && relres > LAG_MARGIN * rtol && since_best <= 3 && queued < max_it;
      if (lag && sharded) {
        // row-sharded: no all-reduce of its own -- the rank's sum waits in entry k + 1 of the scalar block and is reduced together
        // with [C^T z ; r^T z] of the next iteration (3.9 instead of 4.5 all-reduces per inner iteration)
        if ((rc = hx_panel_dot_finish(nblocks, 0, part_r, sc + k + 1, s))) return rc;
        carry = pending = slot;
        continue;
      }
      if ((rc = norm_travels())) return rc;
      if (lag) { pending = slot; continue; }
      if ((rc = arrived(slot, &r2))) return rc;
      if ((rc = judge(r2))) return rc;
      if (leave) break;
    }
    if (pending >= 0) {              // (iteration limit reached with one norm outstanding)
      double r2;
      if ((rc = arrived(pending, &r2))) return rc;
      if ((rc = judge(r2))) return rc;
    }
    }
  }
  if (!converged) {                           // iteration / cycle limit: report the true residual of what we have
    if ((rc = true_residual())) return rc;
  }
  if (h_iters) *h_iters = it;
  if (h_relres) *h_relres = relres;
  return HX_OK;
}

// normwise backward error |b - K x| / (|K| |x| + |b|) of the last hx_cocg solve that ended above its tolerance; -1 if it
// was not measured (the solve met the tolerance)
extern "C" int hx_cocg_backward_error(const hx_amg* M, double* out) {
  HX_REQUIRE(M && out, "bad argument");
  *out = M->last_backward;
  return HX_OK;
}

// The next hx_cocg / hx_cocg_sharded call on this handle starts from the contents of its x argument instead of zero
// (one solve only; ignored by hx_gmres).
extern "C" int hx_amg_next_solve_uses_x(hx_amg* M) {
  HX_REQUIRE(M, "null handle");
  M->x0_next = true;
  return HX_OK;
}

extern "C" int hx_cocg(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                       const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it, hx_c128* ws,
                       int* h_iters, double* h_relres, hx_stream s) {
  return cocg_impl(M, n, nnz, rowptr, col, val, Ksell, b, x, rtol, max_it, ws, nullptr, nullptr, h_iters, h_relres, s);
}
extern "C" int hx_cocg_sharded(hx_amg* M, int32_t n_own, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                               const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol,
                               int max_it, hx_c128* ws, hx_c128* xe, hx_c128* hbuf, int* h_iters, double* h_relres,
                               hx_stream s) {
  HX_REQUIRE(M && M->sharded && xe && hbuf, "bad argument");
  return cocg_impl(M, n_own, nnz, rowptr, col, val, Ksell, b, x, rtol, max_it, ws, xe, hbuf, h_iters, h_relres, s);
}
