// Boundary functional of the Hadamard-form shape derivative (reference: helmholtz_x/shape_derivatives.py:12-37):
//
//     dω/dV  =  ∫_Γ (V · n)  div( p̄† c² ∇p ) ds ,        G := div(q c² ∇p) = (c² ∇q + 2 q c ∇c) · ∇p + q c² Δp ,   q = conj(p†)
//
// with p the direct and p† the (normalised) adjoint eigenvector, c the sound speed and V a nodal displacement field of the
// boundary.  The reference assembles it with UFL / FFCx (`assemble_scalar(form(inner(V_ffd, normal) * G_neu * ds(tag)))`);
// here one thread owns a boundary facet (a triangle of a tetrahedral mesh), evaluates G from the ADJACENT cell's basis
// functions at the points of a 12-point degree-6 rule (exact for every polynomial the P1 / P2 spaces produce here: degree 5)
// and writes the facet's contribution; the caller sums the contributions in order (deterministic).  P1: ∇p is constant on the
// cell and Δp = 0; P2 (the reference's RijkeFFD runs use degree 2): ∇p is linear, Δp constant.  One-off post-processing
// work on O(n^(2/3)) facets: nowhere near a roofline, written for clarity.
#include "hx_common.h"

namespace {
__device__ __forceinline__ zc zs(double a, zc b) { return make_double2(a * b.x, a * b.y); }
struct C3 { zc x, y, z; };
__device__ __forceinline__ C3 c3zero() { return C3{make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)}; }
__device__ __forceinline__ void c3axpy(C3& a, zc s, const double g[3]) {      // a += s * g (g real)
  a.x.x = fma(s.x, g[0], a.x.x); a.x.y = fma(s.y, g[0], a.x.y);
  a.y.x = fma(s.x, g[1], a.y.x); a.y.y = fma(s.y, g[1], a.y.y);
  a.z.x = fma(s.x, g[2], a.z.x); a.z.y = fma(s.y, g[2], a.z.y);
}
__device__ __forceinline__ zc c3dot(const C3& a, const C3& b) {               // bilinear (no conjugate)
  zc r = zmul(a.x, b.x);
  zfma(r, a.y, b.y);
  zfma(r, a.z, b.z);
  return r;
}
__constant__ int TET_EDGE[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

// value, gradient and Laplacian at barycentric point lam of a P1 (degree 1: 4 dofs) or P2 (10 dofs) field on one tetrahedron
__device__ void eval_field(int degree, const zc* u, const double lam[4], const double gl[4][3], zc& val, C3& grad, zc& lap) {
  val = make_double2(0, 0);
  lap = make_double2(0, 0);
  grad = c3zero();
  if (degree == 1) {
    for (int i = 0; i < 4; ++i) { zfma(val, lam[i], u[i]); c3axpy(grad, u[i], gl[i]); }
    return;
  }
  for (int i = 0; i < 4; ++i) {
    zfma(val, lam[i] * (2 * lam[i] - 1), u[i]);
    c3axpy(grad, zs(4 * lam[i] - 1, u[i]), gl[i]);
    zfma(lap, 4 * (gl[i][0] * gl[i][0] + gl[i][1] * gl[i][1] + gl[i][2] * gl[i][2]), u[i]);
  }
  for (int e = 0; e < 6; ++e) {
    const int i = TET_EDGE[e][0], j = TET_EDGE[e][1];
    const zc w = u[4 + e];
    zfma(val, 4 * lam[i] * lam[j], w);
    c3axpy(grad, zs(4 * lam[i], w), gl[j]);
    c3axpy(grad, zs(4 * lam[j], w), gl[i]);
    zfma(lap, 8 * (gl[i][0] * gl[j][0] + gl[i][1] * gl[j][1] + gl[i][2] * gl[j][2]), w);
  }
}

__global__ void k_shape_facets(int degree, int32_t nf, const double* __restrict__ x, const int32_t* __restrict__ cells,
                               const int32_t* __restrict__ cell_dofs, const int32_t* __restrict__ facets,
                               const int32_t* __restrict__ facet_cell, const zc* __restrict__ p, const zc* __restrict__ q,
                               const zc* __restrict__ c, int c_cellwise, const double* __restrict__ V, int nq,
                               const double* __restrict__ quad, zc* __restrict__ out) {
  for (int32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int32_t cell = facet_cell[f];
    int32_t v[4];
    double X[4][3];
    for (int i = 0; i < 4; ++i) {
      v[i] = cells[4 * (int64_t)cell + i];
      for (int d = 0; d < 3; ++d) X[i][d] = x[3 * (int64_t)v[i] + d];
    }
    // gradients of the barycentric coordinates: rows of the inverse of [x1-x0, x2-x0, x3-x0]^T
    double J[3][3], gl[4][3];
    for (int r = 0; r < 3; ++r) for (int d = 0; d < 3; ++d) J[r][d] = X[r + 1][d] - X[0][d];
    const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                       J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    const double id = 1.0 / det;
    // columns of J^-1 (J rows = edge vectors): grad lam_{r+1} = column r of J^-1
    gl[1][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id; gl[1][1] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id; gl[1][2] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
    gl[2][0] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; gl[2][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; gl[2][2] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    gl[3][0] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id; gl[3][1] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id; gl[3][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    for (int d = 0; d < 3; ++d) gl[0][d] = -(gl[1][d] + gl[2][d] + gl[3][d]);
    // the facet inside its cell: local indices of its three vertices, the opposite vertex, outward unit normal, area
    int loc[3], opp = 0;
    bool used[4] = {false, false, false, false};
    for (int k = 0; k < 3; ++k) {
      const int32_t fv = facets[3 * (int64_t)f + k];
      loc[k] = 0;
      for (int i = 0; i < 4; ++i) if (v[i] == fv) loc[k] = i;
      used[loc[k]] = true;
    }
    for (int i = 0; i < 4; ++i) if (!used[i]) opp = i;
    double e1[3], e2[3], nrm[3];
    for (int d = 0; d < 3; ++d) { e1[d] = X[loc[1]][d] - X[loc[0]][d]; e2[d] = X[loc[2]][d] - X[loc[0]][d]; }
    nrm[0] = e1[1] * e2[2] - e1[2] * e2[1]; nrm[1] = e1[2] * e2[0] - e1[0] * e2[2]; nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
    const double len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    double sgn = 0.0;
    for (int d = 0; d < 3; ++d) sgn += nrm[d] * (X[loc[0]][d] - X[opp][d]);
    const double s = (sgn >= 0 ? 1.0 : -1.0) / len;
    for (int d = 0; d < 3; ++d) nrm[d] *= s;
    const double area = 0.5 * len;
    // dof values of the adjacent cell
    zc pu[10], qu[10], cu[4];
    const int nd = degree == 1 ? 4 : 10;
    for (int i = 0; i < nd; ++i) {
      const int32_t dof = degree == 1 ? v[i] : cell_dofs[10 * (int64_t)cell + i];
      pu[i] = p[dof];
      qu[i] = q[dof];
    }
    for (int i = 0; i < 4; ++i) cu[i] = c_cellwise ? c[cell] : c[v[i]];
    double Vn[3];                                       // V . n at the facet's vertices
    for (int k = 0; k < 3; ++k) {
      const int32_t fv = v[loc[k]];
      Vn[k] = V[3 * (int64_t)fv] * nrm[0] + V[3 * (int64_t)fv + 1] * nrm[1] + V[3 * (int64_t)fv + 2] * nrm[2];
    }
    zc acc = make_double2(0, 0);
    for (int g = 0; g < nq; ++g) {
      const double* Q = quad + 4 * g;
      double lam[4] = {0, 0, 0, 0};
      lam[loc[0]] = Q[0]; lam[loc[1]] = Q[1]; lam[loc[2]] = Q[2];
      zc pv, qv, cv, pl, ql, cl;
      C3 gp, gq, gc;
      eval_field(degree, pu, lam, gl, pv, gp, pl);
      eval_field(degree, qu, lam, gl, qv, gq, ql);
      eval_field(1, cu, lam, gl, cv, gc, cl);
      const zc c2 = zmul(cv, cv), qc2 = zmul(zmul(qv, cv), make_double2(2.0, 0.0));
      C3 t;                                             // grad(q c^2) = c^2 grad q + 2 q c grad c
      t.x = zadd(zmul(c2, gq.x), zmul(qc2, gc.x));
      t.y = zadd(zmul(c2, gq.y), zmul(qc2, gc.y));
      t.z = zadd(zmul(c2, gq.z), zmul(qc2, gc.z));
      zc G = c3dot(t, gp);
      zfma(G, zmul(qv, c2), pl);
      const double vn = Q[0] * Vn[0] + Q[1] * Vn[1] + Q[2] * Vn[2];
      zfma(acc, Q[3] * vn, G);
    }
    out[f] = zs(area, acc);
  }
}
}  // namespace

extern "C" int hx_shape_derivative_facets(int degree, int32_t nf, const double* x, const int32_t* cells,
                                          const int32_t* cell_dofs, const int32_t* facets, const int32_t* facet_cell,
                                          const hx_c128* p, const hx_c128* q, const hx_c128* c, int c_is_cellwise,
                                          const double* V, int nq, const double* quad, hx_c128* out, hx_stream s) {
  HX_REQUIRE((degree == 1 || degree == 2) && nf > 0 && x && cells && facets && facet_cell && p && q && c && V && nq > 0 &&
             quad && out && (degree == 1 || cell_dofs), "bad argument");
  k_shape_facets<<<hx_grid(nf, 128), 128, 0, hx_s(s)>>>(degree, nf, x, cells, cell_dofs, facets, facet_cell, (const zc*)p,
                                                        (const zc*)q, (const zc*)c, c_is_cellwise, V, nq, quad, (zc*)out);
  HX_LAUNCH_CHECK("k_shape_facets");
  return HX_OK;
}
