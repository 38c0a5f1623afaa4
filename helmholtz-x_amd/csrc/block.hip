// Blocks of right-hand sides: SELL-64 / CSR products, smoother steps and the V-cycle for NB vectors at once.
//
// The reference's direct factorisation serves any number of right-hand sides with one pass over the factors
// (eigensolvers.py:50,103 -> MatSolve; the 16 flame columns of flame_matrices.py:170-177 and the nev eigenvectors of the final
// solve are all solves with the SAME shifted operator).  The iterative inner solver paid one full pass over the matrix per
// vector.  Here NB vectors are interleaved row by row (X[row * NB + c]): a matrix entry -- value and index -- is loaded ONCE
// and multiplies NB operands that sit next to each other in memory, so
//   * the matrix stream of a pass is shared by NB vectors (fine smoother step: 60 B of matrix + 40 B of vectors per row and
//     vector -> 15 + 40 at NB = 4; complex128 operator product: 240 + 32 -> 60 + 32);
//   * a gather fetches NB * sizeof(V) contiguous bytes per lane (32 B complex64 / 64 B complex128 at NB = 4: whole sectors)
//     instead of 8 / 16 -- on unstructured numberings the single-vector kernels are bound by the CU's vector-memory path
//     (profiles/r04_pmc_sell_kernels_micca_real_r2.txt: 1300 L1 accesses per slice), the block form issues a quarter of the
//     gather instructions per vector.
// One wave = one slice, one lane = one row with NB accumulators; stencil / 16-bit-delta / 32-bit index formats as in sell.hip
// (no LDS staging: the block gathers are sector-sized already).  CSR levels: LPR lanes per row, NB accumulators per lane.
#include "amg_internal.h"
#include "sell_internal.h"

namespace {
template <typename IT> __device__ __forceinline__ int32_t b_cidx(IT c, int32_t row);
template <> __device__ __forceinline__ int32_t b_cidx<int32_t>(int32_t c, int32_t) { return c; }
template <> __device__ __forceinline__ int32_t b_cidx<int16_t>(int16_t c, int32_t row) { return row + (int32_t)c; }
__device__ __forceinline__ void b_acc(zc& acc, zc v, zc x) { zfma(acc, v, x); }
__device__ __forceinline__ void b_acc(cc& acc, cc v, cc x) { zfma(acc, v, x); }
__device__ __forceinline__ void b_acc(cc& acc, ch v, cc x) {
  const float vr = __low2float(v), vi = __high2float(v);
  acc.x = fmaf(vr, x.x, acc.x);
  acc.y = fmaf(vr, x.y, acc.y);
  acc.x = fmaf(-vi, x.y, acc.x);
  acc.y = fmaf(vi, x.x, acc.y);
}
__device__ __forceinline__ void b_acc(zc& acc, double v, zc x) { acc.x = fma(v, x.x, acc.x); acc.y = fma(v, x.y, acc.y); }
__device__ __forceinline__ void b_acc(cc& acc, float v, cc x) { acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); }
template <typename T> __device__ __forceinline__ T b_zero();
template <> __device__ __forceinline__ zc b_zero<zc>() { return make_double2(0.0, 0.0); }
template <> __device__ __forceinline__ cc b_zero<cc>() { return make_float2(0.f, 0.f); }

template <typename V>
struct BlkOp {
  V alpha, beta;
  int beta_is_zero;
  typename real_of<V>::type omega, mom, vscale;
};

// row sums of one lane for NB interleaved vectors: acc[c] += val_k * x[col_k * NB + c].  KIND 0: stencil (col = row + delta_k,
// delta list in register u across the wave), 1: 16-bit deltas, 2: 32-bit columns.
__device__ __forceinline__ zc b_wide(zc v) { return v; }
__device__ __forceinline__ cc b_wide(cc v) { return v; }
template <typename V> __device__ __forceinline__ V b_conv(zc v);
template <> __device__ __forceinline__ zc b_conv<zc>(zc v) { return v; }
template <typename V> __device__ __forceinline__ V b_conv(cc v);
template <> __device__ __forceinline__ cc b_conv<cc>(cc v) { return v; }
template <> __device__ __forceinline__ zc b_conv<zc>(cc v) { return make_double2((double)v.x, (double)v.y); }

// NB = 4 interleaved entries of one row: 32 bytes (complex64: two 16-byte accesses) or 64 bytes (complex128: four).  Written out
// so that the compiler emits dwordx4 accesses -- element-wise complex64 loads come out as four dwordx2 instructions, i.e. four
// trips through the vector-memory path per gather instead of two (measured: the first version of the block smoother took 180 us
// where this one takes ~100 us on 1.9 M unstructured rows).
template <typename V, typename XT, int NB>
__device__ __forceinline__ void blk_load(const XT* __restrict__ p, V (&out)[NB]) {
  static_assert(NB == 4, "block width 4");
  if constexpr (sizeof(XT) == 8) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1];
    out[0] = b_conv<V>(make_float2(a.x, a.y)); out[1] = b_conv<V>(make_float2(a.z, a.w));
    out[2] = b_conv<V>(make_float2(b.x, b.y)); out[3] = b_conv<V>(make_float2(b.z, b.w));
  } else {
#pragma unroll
    for (int c = 0; c < NB; ++c) out[c] = b_conv<V>(p[c]);
  }
}
template <typename V, int NB>
__device__ __forceinline__ void blk_store(V* __restrict__ p, const V (&v)[NB]) {
  static_assert(NB == 4, "block width 4");
  if constexpr (sizeof(V) == 8) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
    q[1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
  } else {
#pragma unroll
    for (int c = 0; c < NB; ++c) p[c] = v[c];
  }
}

// XT: storage type of the operand block (V itself, or complex64 for a complex128 product whose operand is the V-cycle's output)
template <typename V, typename MV, typename XT, int NB, int G, int KIND>
__device__ __forceinline__ void blk_row_dot(V (&acc)[NB], int32_t u, const void* __restrict__ cb, const MV* __restrict__ vb, int lane,
                                            const XT* __restrict__ x, int width, int32_t row, int32_t n_cols) {
  typedef typename real_of<V>::type R;
  for (int k0 = 0; k0 < width; k0 += G) {
    MV v[G];
    int32_t col[G];
    R w[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = min(k0 + j, width - 1);
      w[j] = (R)(k0 + j < width ? 1 : 0);
      v[j] = vb[(int64_t)k * SELL_C + lane];
      if (KIND == 0) col[j] = min(max(row + __builtin_amdgcn_readlane(u, k), 0), n_cols - 1);
      else if (KIND == 1) col[j] = b_cidx<int16_t>(((const int16_t*)cb)[(int64_t)k * SELL_C + lane], row);
      else col[j] = ((const int32_t*)cb)[(int64_t)k * SELL_C + lane];
    }
    V xv[G][NB];
#pragma unroll
    for (int j = 0; j < G; ++j) blk_load<V, XT, NB>(x + (int64_t)col[j] * NB, xv[j]);
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int c = 0; c < NB; ++c) b_acc(acc[c], v[j], zscale(w[j], xv[j][c]));
  }
}

// MODE 0: Y = alpha K X + beta Yin.   MODE 1: Xout = Xin + omega D^-1 (B - K Xin) + mom (Xin - Xprev)   (D^-1 per ROW, shared)
template <typename V, int MODE, typename MV, int NB, int G, typename XT>
__global__ void __launch_bounds__(256)
k_sell_blk(int32_t n, int32_t n_slices, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
           const int16_t* __restrict__ dcol, const uint8_t* __restrict__ wide, const int32_t* __restrict__ st_ptr,
           const int32_t* __restrict__ st_delta, int32_t n_cols, const MV* __restrict__ val, const XT* __restrict__ x, V* y,
           const V* yin, const V* __restrict__ dinv, const V* __restrict__ b, const V* __restrict__ xprev, const BlkOp<V> a,
           const SellTail<MV> tl) {
  const int lane = threadIdx.x & 63;
  const int32_t slice = __builtin_amdgcn_readfirstlane((int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6));
  if (slice >= n_slices) return;
  const int64_t base = slice_ptr[slice];
  const int32_t width = (int32_t)((slice_ptr[slice + 1] - base) >> 6);
  int32_t st0 = 0, stw = 0;
  if (st_ptr) { st0 = st_ptr[slice]; stw = st_ptr[slice + 1] - st0; }
  const int32_t u = lane < stw ? st_delta[st0 + lane] : 0;
  const int32_t row = slice * SELL_C + lane;
  const bool live = row < n;
  V acc[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = b_zero<V>();
  const MV* vb = val + base;
  if (stw > 0) blk_row_dot<V, MV, XT, NB, G, 0>(acc, u, nullptr, vb, lane, x, width, row, n_cols);
  else if (dcol && !wide[slice]) blk_row_dot<V, MV, XT, NB, G, 1>(acc, 0, dcol + base, vb, lane, x, width, row, n_cols);
  else blk_row_dot<V, MV, XT, NB, G, 2>(acc, 0, col + base, vb, lane, x, width, row, n_cols);
  if (tl.tptr && stw == 0) {                          // overflow tail of a split slice (hx_sell::tptr): the lane's own entries
    const int32_t t0 = tl.tptr[slice];
    if (tl.tptr[slice + 1] > t0) {
      const uint32_t pk = tl.trow[row];
      const int cnt = (int)(pk & 63u);
      const int32_t* tc = tl.tcol + t0 + (pk >> 6);
      const MV* tv = tl.tval + t0 + (pk >> 6);
      for (int j = 0; __any(j < cnt); ++j)
        if (j < cnt) {
          V xv[NB];
          blk_load<V, XT, NB>(x + (int64_t)tc[j] * NB, xv);
          const MV v = tv[j];
#pragma unroll
          for (int c = 0; c < NB; ++c) b_acc(acc[c], v, xv[c]);
        }
    }
  }
  if (!live) return;
  const int64_t o = (int64_t)row * NB;
  if constexpr (sizeof(MV) != sizeof(V)) {
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = zscale(a.vscale, acc[c]);
  }
  V out[NB];
  if (MODE == 0) {
    V yi[NB];
    if (!a.beta_is_zero) blk_load<V, V, NB>(yin + o, yi);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      V r = zmul(a.alpha, acc[c]);
      if (!a.beta_is_zero) zfma(r, a.beta, yi[c]);
      out[c] = r;
    }
    blk_store<V, NB>(y + o, out);
    return;
  }
  const V d = zscale(a.omega, dinv[row]);
  V x0[NB], bb[NB], xp[NB];
  blk_load<V, XT, NB>(x + o, x0);
  blk_load<V, V, NB>(b + o, bb);
  const bool has_prev = a.mom != 0 && xprev;
  if (has_prev) blk_load<V, V, NB>(xprev + o, xp);
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    V xn = x0[c];
    zfma(xn, d, zsub(bb[c], acc[c]));
    if (a.mom != 0) xn = zadd(xn, zscale(a.mom, has_prev ? zsub(x0[c], xp[c]) : x0[c]));
    out[c] = xn;
  }
  blk_store<V, NB>(y + o, out);
}

template <typename V, int MODE, typename MV, int NB, typename XT = V>
int launch_sell_blk(const hx_sell* h, const MV* val, const XT* x, V* y, V alpha, V beta, const V* yin, const V* dinv, const V* b,
                    const V* xprev, typename real_of<V>::type omega, typename real_of<V>::type mom, typename real_of<V>::type vscale,
                    hx_stream s) {
  const BlkOp<V> op{alpha, beta, (beta.x == 0 && beta.y == 0) ? 1 : 0, omega, mom, vscale};
  // entries per group of the row sums: 4 for complex64 (measured at 1.9 M unstructured rows: smoother step 155 / 141 / 170 us for
  // groups of 2 / 4 / 8), 2 for complex128 (registers)
  constexpr int G = sizeof(V) == 16 ? 2 : 4;
  k_sell_blk<V, MODE, MV, NB, G, XT><<<(h->n_slices + 3) / 4, 256, 0, hx_s(s)>>>(h->n, h->n_slices, h->slice_ptr, h->col, h->dcol, h->wide,
                                                                           h->st_ptr, h->st_delta, h->n_cols, val, x, y,
                                                                           yin ? yin : y, dinv, b, xprev, op, sell_tail_of<MV>(h));
  HX_LAUNCH_CHECK("k_sell_blk");
  return HX_OK;
}

// ---- CSR levels: LPR lanes per row, NB accumulators per lane.  MV: complex values, or REAL values (transfer operators).
template <int LPR, typename V, typename MV, int NB, int MODE>
__global__ void __launch_bounds__(256)
k_csr_blk(int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const MV* __restrict__ val,
          const V* __restrict__ x, V* y, const V* yin, const V* __restrict__ dinv, const V* __restrict__ b,
          const V* __restrict__ xprev, const BlkOp<V> a) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t sweeps = ((int64_t)n + stride - 1) / stride;
  int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  for (int64_t it = 0; it < sweeps; ++it, row += stride) {
    V acc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = b_zero<V>();
    if (row < n) {
      const int32_t e = rowptr[row + 1];
      for (int32_t p = rowptr[row] + sub; p < e; p += LPR) {
        const MV v = val[p];
        V xv[NB];
        blk_load<V, V, NB>(x + (int64_t)col[p] * NB, xv);
#pragma unroll
        for (int c = 0; c < NB; ++c) b_acc(acc[c], v, xv[c]);
      }
    }
    if (LPR > 1) {
#pragma unroll
      for (int c = 0; c < NB; ++c) acc[c] = zreduce_sub<LPR>(acc[c]);
    }
    if (row < n && sub == 0) {
      const int64_t o = row * NB;
      V out[NB];
      if (MODE == 0) {
        V yi[NB];
        if (!a.beta_is_zero) blk_load<V, V, NB>(yin + o, yi);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          V r = zmul(a.alpha, acc[c]);
          if (!a.beta_is_zero) zfma(r, a.beta, yi[c]);
          out[c] = r;
        }
      } else {
        const V d = zscale(a.omega, dinv[row]);
        V x0[NB], bb[NB], xp[NB];
        blk_load<V, V, NB>(x + o, x0);
        blk_load<V, V, NB>(b + o, bb);
        const bool has_prev = a.mom != 0 && xprev;
        if (has_prev) blk_load<V, V, NB>(xprev + o, xp);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          V xn = x0[c];
          zfma(xn, d, zsub(bb[c], acc[c]));
          if (a.mom != 0) xn = zadd(xn, zscale(a.mom, has_prev ? zsub(x0[c], xp[c]) : x0[c]));
          out[c] = xn;
        }
      }
      blk_store<V, NB>(y + o, out);
    }
  }
}
inline int blk_pick_lpr(int64_t nnz, int32_t n_rows) {
  const double avg = (double)nnz / (double)(n_rows > 0 ? n_rows : 1);
  return avg <= 3.0 ? 1 : avg <= 6.0 ? 2 : avg <= 12.0 ? 4 : avg <= 32.0 ? 8 : avg <= 96.0 ? 16 : avg <= 256 ? 32 : 64;
}
template <typename V, typename MV, int NB, int MODE>
int launch_csr_blk(int32_t n, int64_t nnz, const int32_t* rp, const int32_t* ci, const MV* va, const V* x, V* y, V alpha, V beta,
                   const V* yin, const V* dinv, const V* b, const V* xprev, typename real_of<V>::type omega,
                   typename real_of<V>::type mom, hx_stream s) {
  if (n == 0) return HX_OK;
  const BlkOp<V> op{alpha, beta, (beta.x == 0 && beta.y == 0) ? 1 : 0, omega, mom, 1};
#define HX_GO(L)                                                                                                             \
  k_csr_blk<L, V, MV, NB, MODE><<<hx_grid((int64_t)n, 256 / L, 256 * 16), 256, 0, hx_s(s)>>>(n, rp, ci, va, x, y, yin ? yin : y, dinv, \
                                                                                             b, xprev, op)
  switch (blk_pick_lpr(nnz, n)) {
    case 1: HX_GO(1); break;
    case 2: HX_GO(2); break;
    case 4: HX_GO(4); break;
    case 8: HX_GO(8); break;
    case 16: HX_GO(16); break;
    case 32: HX_GO(32); break;
    default: HX_GO(64); break;
  }
#undef HX_GO
  HX_LAUNCH_CHECK("k_csr_blk");
  return HX_OK;
}

// X = w D^-1 B (first smoothing step from a zero iterate), D^-1 per row
template <typename V, int NB>
__global__ void k_diag_blk(int64_t n, typename real_of<V>::type w, const V* __restrict__ dinv, const V* __restrict__ b, V* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * NB; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = zscale(w, zmul(dinv[i / NB], b[i]));
}

// ---- interleave / de-interleave: rows of a (nb x n) row-major panel <-> columns of the (n x NB) block vector (columns >= nb: zero)
template <int NB>
__global__ void k_blk_pack(int64_t n, int nb, const zc* __restrict__ panel, int64_t ld, zc* __restrict__ blk) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
    for (int c = 0; c < NB; ++c) blk[i * NB + c] = c < nb ? panel[(int64_t)c * ld + i] : make_double2(0.0, 0.0);
}
template <int NB>
__global__ void k_blk_unpack(int64_t n, int nb, const zc* __restrict__ blk, zc* __restrict__ panel, int64_t ld) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
    for (int c = 0; c < NB; ++c) if (c < nb) panel[(int64_t)c * ld + i] = blk[i * NB + c];
}
}  // namespace

#define HXB 4      // block width of the library (one instantiation; the matrix share of a pass is already a quarter)
extern "C" int hx_block_width(void) { return HXB; }

// ---- SELL-64 block entry points
extern "C" int hx_sell_spmm_z(const hx_sell* h, const hx_c128* X, hx_c128* Y, const double alpha[2], const double beta[2], hx_stream s) {
  HX_REQUIRE(h && h->val_z && X && Y && X != Y && alpha && beta, "bad argument (complex128 values needed)");
  return launch_sell_blk<zc, 0, zc, HXB>(h, h->val_z, (const zc*)X, (zc*)Y, hx_z(alpha), hx_z(beta), nullptr, nullptr, nullptr, nullptr,
                                         0.0, 0.0, 1.0, s);
}
// the same with the operand block given in complex64 (the V-cycle's output: exactly representable, half the gather bytes)
int hx_sell_spmm_z_x32(const hx_sell* h, const hx_c64* X, hx_c128* Y, hx_stream s) {
  HX_REQUIRE(h && h->val_z && X && Y, "bad argument (complex128 values needed)");
  return launch_sell_blk<zc, 0, zc, HXB, cc>(h, h->val_z, (const cc*)X, (zc*)Y, make_double2(1.0, 0.0), make_double2(0.0, 0.0), nullptr,
                                             nullptr, nullptr, nullptr, 0.0, 0.0, 1.0, s);
}
// R = B - K X
int hx_sell_residual_blk_z(const hx_sell* h, const hx_c128* X, const hx_c128* B, hx_c128* R, hx_stream s) {
  HX_REQUIRE(h && h->val_z && X && B && R && X != R, "bad argument");
  return launch_sell_blk<zc, 0, zc, HXB>(h, h->val_z, (const zc*)X, (zc*)R, make_double2(-1.0, 0.0), make_double2(1.0, 0.0),
                                         (const zc*)B, nullptr, nullptr, nullptr, 0.0, 0.0, 1.0, s);
}
// complex64 arithmetic on the complex-half (or complex64) copy: Y = alpha K X + beta Yin (Yin == NULL: Y itself)
int hx_sell_spmm_c64(const hx_sell* h, const hx_c64* X, hx_c64* Y, float alpha, float beta, const hx_c64* Yin, hx_stream s) {
  HX_REQUIRE(h && (h->val_h || h->val_c) && X && Y && X != Y, "bad argument (complex64 / complex-half values needed)");
  const cc a = make_float2(alpha, 0.f), b = make_float2(beta, 0.f);
  if (h->val_h)
    return launch_sell_blk<cc, 0, ch, HXB>(h, h->val_h, (const cc*)X, (cc*)Y, a, b, (const cc*)Yin, nullptr, nullptr, nullptr, 0.f, 0.f,
                                           h->h_scale, s);
  return launch_sell_blk<cc, 0, cc, HXB>(h, h->val_c, (const cc*)X, (cc*)Y, a, b, (const cc*)Yin, nullptr, nullptr, nullptr, 0.f, 0.f, 1.f, s);
}
int hx_sell_smooth_blk_c64(const hx_sell* h, const hx_c64* dinv, const hx_c64* B, const hx_c64* Xin, hx_c64* Xout, float omega,
                           const hx_c64* Xprev, float mom, hx_stream s) {
  HX_REQUIRE(h && (h->val_h || h->val_c) && dinv && B && Xin && Xout && Xin != Xout, "bad argument");
  const cc z = make_float2(0.f, 0.f);
  if (h->val_h)
    return launch_sell_blk<cc, 1, ch, HXB>(h, h->val_h, (const cc*)Xin, (cc*)Xout, z, z, nullptr, (const cc*)dinv, (const cc*)B,
                                           (const cc*)Xprev, omega, mom, h->h_scale, s);
  return launch_sell_blk<cc, 1, cc, HXB>(h, h->val_c, (const cc*)Xin, (cc*)Xout, z, z, nullptr, (const cc*)dinv, (const cc*)B,
                                         (const cc*)Xprev, omega, mom, 1.f, s);
}

// ---- CSR block entry points (internal)
int hx_csr_spmm_z(int32_t n, int64_t nnz, const int32_t* rp, const int32_t* ci, const hx_c128* va, const hx_c128* X, hx_c128* Y,
                  double alpha, double beta, const hx_c128* Yin, hx_stream s) {
  return launch_csr_blk<zc, zc, HXB, 0>(n, nnz, rp, ci, (const zc*)va, (const zc*)X, (zc*)Y, make_double2(alpha, 0.0),
                                        make_double2(beta, 0.0), (const zc*)Yin, nullptr, nullptr, nullptr, 0.0, 0.0, s);
}
int hx_csr_spmm_c(int32_t n, int64_t nnz, const int32_t* rp, const int32_t* ci, const float* va, bool real_vals, const float* X,
                  float* Y, float alpha, float beta, const float* Yin, hx_stream s) {
  const cc a = make_float2(alpha, 0.f), b = make_float2(beta, 0.f);
  if (real_vals)
    return launch_csr_blk<cc, float, HXB, 0>(n, nnz, rp, ci, va, (const cc*)X, (cc*)Y, a, b, (const cc*)Yin, nullptr, nullptr, nullptr,
                                             0.f, 0.f, s);
  return launch_csr_blk<cc, cc, HXB, 0>(n, nnz, rp, ci, (const cc*)va, (const cc*)X, (cc*)Y, a, b, (const cc*)Yin, nullptr, nullptr,
                                        nullptr, 0.f, 0.f, s);
}
int hx_csr_smooth_blk_c(int32_t n, int64_t nnz, const int32_t* rp, const int32_t* ci, const float* va, const float* dinv, const float* B,
                        const float* Xin, float* Xout, float omega, const float* Xprev, float mom, hx_stream s) {
  const cc z = make_float2(0.f, 0.f);
  return launch_csr_blk<cc, cc, HXB, 1>(n, nnz, rp, ci, (const cc*)va, (const cc*)Xin, (cc*)Xout, z, z, nullptr, (const cc*)dinv,
                                        (const cc*)B, (const cc*)Xprev, omega, mom, s);
}
int hx_cdiag_scale_blk(int64_t n, float w, const float* dinv, const float* B, float* X, hx_stream s) {
  if (n == 0) return HX_OK;
  k_diag_blk<cc, HXB><<<hx_grid(n * HXB, 256), 256, 0, hx_s(s)>>>(n, w, (const cc*)dinv, (const cc*)B, (cc*)X);
  HX_LAUNCH_CHECK("k_diag_blk");
  return HX_OK;
}
// rows [0, nb) of a row-major panel (leading dimension ld) <-> the interleaved block vector of width hx_block_width()
extern "C" int hx_zblock_pack(int64_t n, int nb, const hx_c128* panel, int64_t ld, hx_c128* blk, hx_stream s) {
  HX_REQUIRE(n > 0 && nb >= 1 && nb <= HXB && panel && blk && ld >= n, "bad argument");
  k_blk_pack<HXB><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, nb, (const zc*)panel, ld, (zc*)blk);
  HX_LAUNCH_CHECK("k_blk_pack");
  return HX_OK;
}
extern "C" int hx_zblock_unpack(int64_t n, int nb, const hx_c128* blk, hx_c128* panel, int64_t ld, hx_stream s) {
  HX_REQUIRE(n > 0 && nb >= 1 && nb <= HXB && panel && blk && ld >= n, "bad argument");
  k_blk_unpack<HXB><<<hx_grid(n, 256), 256, 0, hx_s(s)>>>(n, nb, (const zc*)blk, (zc*)panel, ld);
  HX_LAUNCH_CHECK("k_blk_unpack");
  return HX_OK;
}
