/* hx.h -- C ABI of libhx.so: the MI355X (gfx950) hot path of helmholtz-x.
 *
 * The reference (ekremekc/helmholtz-x) has no C/FFI boundary of its own: every flop of its hot path
 * runs inside DOLFINx / PETSc / SLEPc, reached from Python.  This header is
 * therefore the boundary a maintainer would bind in place of those library calls; each entry point
 * cites the reference line(s) whose delegated arithmetic it replaces (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *  - plain C, no exceptions: every function returns 0 on success, a negative hx_status otherwise;
 *    hx_last_error() gives a thread-local message.
 *  - all array arguments are DEVICE pointers unless the name starts with h_ (host).
 *  - complex128 is interleaved {re, im} doubles (PETSc complex scalar layout); scalars are passed as
 *    const double[2].
 *  - indices are int32 (the reference uses np.int32 / PetscInt32, flame_matrices.py:70).
 *  - `stream` is a hipStream_t passed as void*; NULL = the default stream.  Nothing synchronises
 *    unless documented ("SYNC").  One calling thread per device, as in the reference
 *    (single-threaded Python per MPI rank).
 *  - buffers are caller-owned; opaque handles have explicit create / destroy.
 */
#ifndef HX_H
#define HX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } hx_c128;
typedef struct { float re, im; } hx_c64;   /* complex64: single-precision copy of the AMG hierarchy only */
typedef void* hx_stream;

enum hx_status {
  HX_OK = 0,
  HX_ERR_ARG = -1,      /* bad argument (null pointer, negative size, unsupported dim) */
  HX_ERR_HIP = -2,      /* a HIP runtime call failed; see hx_last_error() */
  HX_ERR_NOGPU = -3,    /* no gfx950 device visible */
  HX_ERR_ALLOC = -4,
  HX_ERR_BREAKDOWN = -5, /* numerical breakdown (zero pivot / zero norm) */
  HX_ERR_COMM = -6       /* an exchange callback of the row-sharded mode reported failure */
};

/* Switches of the library (experiment / diagnostics hooks; none is needed for normal use).  Each can be given as an environment
 * variable or, at run time, through hx_set_option (below), which takes precedence; the environment is consulted again whenever
 * an option is set or removed.
 *   HX_SELL_STENCIL=0 / HX_SELL_INDEX=i32 / HX_SELL_LDS=0|2   SELL-64 format and kernel variants (hx_sell_create, k_sell)
 *   HX_POOL=0, HX_POOL_MAX_GIB, HX_POOL_MIN_BYTES                 device-buffer pool off / cap of the cached bytes / smallest pooled block
 *   HX_SELL_TAIL=0, HX_SELL_TAIL_STEP, HX_SELL_TAIL_GRAN       overflow tails of ragged slices off (every slice as wide as its longest row) / cost of a tail step
 *                                                             in the split's model (128) / granularity of the body widths it may choose (8)
 *   HX_SELL_XCD=0|3, HX_SELL_XCD_CHUNK, HX_SELL_ABLATE         XCD-contiguous slice schedule never / always (default: by footprint) / in chunks of that many workgroups;
 *                                                             timing-only ablations
 *   HX_COCG_PQ_IN_PRODUCT=0                                   hx_cocg: p^T K p from a panel dot instead of from the product itself
 *   HX_AMG_FOLD_DIAG=0                                        the first smoothing step of a level (x1 = a0 D^-1 b) as a launch of its own again (default: written by the
 *                                                             restriction that produced b, on the fine level by hx_cocg's update kernel)
 *   HX_AMG_DEEP_HALO=0                                        row-sharded levels in deep-halo mode run the classic schedule (one exchange per smoothing step) anyway
 *   HX_AMG_DEEP_OVERLAP=0                                     the deep-halo cycle's multi-layer exchange of b blocks (default: split -- x1 on the owned block and the
 *                                                             interior slices of the first smoothing step run on the side stream while it travels; SELL-64 levels)
 *   HX_COCG_LAG=0                                             hx_cocg: wait for every residual norm (default: read one iteration late while far from the tolerance)
 *   HX_COCG_FUSE_REDUCTIONS=0                                  hx_cocg: two-stage reductions on one GPU too
 *   HX_COCG_SINGLE_REDUCTION=0|2                               single-reduction (Chronopoulos-Gear) recurrence never / also on one GPU (default: row-sharded solves)
 *   HX_CHEB_LO, HX_CHEB_KIND, HX_AMG_COARSE_EXTRA             Chebyshev bounds / kind, extra coarse smoothing steps
 *   HX_AMG_GRAPH_PRIVATE_STREAM                               hipGraph replay of the V-cycle on a private stream
 *   HX_GMRES_TRACE=1                                          residual histories of hx_gmres / hx_cocg on stderr
 *   HX_HOST_THREADS                                           threads of the hx_host_* set-up passes
 *   HX_RCCL_LIB                                               librccl to bind in hx_comm_* (default: the loaded one / librccl.so.1)
 *   HX_DEBUG_BOUNDS=1                                         entry points that follow caller-supplied index arrays (hx_zgather, hx_zscatter_axpy, hx_zsparse_dots,
 *                                                             hx_zsparse_axpys, the send / receive lists of hx_comm_exchange) first check on the device that every
 *                                                             indexed access stays inside the allocation holding the indexed vector: HX_ERR_ARG with the offending
 *                                                             position instead of a device fault.  One reduction + a synchronising read-back per call (tests) */
/* Run-time options (csrc/options.hip).  Every switch listed above is ALSO an option: its name is the environment variable's
 * without the HX_ prefix, in lower case ("sell_stencil", "cocg_lag", "cocg_single_reduction", ...).  hx_set_option(name, value)
 * overrides the environment from the next library call on (value NULL or "" removes the override); nothing is latched at first
 * use any more, so a test or a driver can switch a kernel variant between two calls of one process.  hx_get_option returns the
 * value in force as text (HX_ERR_ARG when neither an override nor the environment variable exists).  The pool's start-up
 * parameters (HX_POOL*) and HX_RCCL_LIB are read once, when the pool / the RCCL binding is first used.
 * replaces: nothing in the reference (its solver options are set in code, eigensolvers.py:43-60); PETSc's options database is the
 * analogue. */
int hx_set_option(const char* name, const char* value);
int hx_get_option(const char* name, char* out, int out_len);
/* Device-buffer pool (csrc/pool.hip): buffers of 1 MiB or more that a handle releases (hx_sell_destroy, hx_amg_destroy, the scratch
 * of the set-up passes) are kept and handed to the next request of exactly the same size -- every fixed-point call of a driver
 * builds the same objects again (the reference: E.destroy() and a new EPS per iteration, eigensolvers.py:161) -- instead of
 * going through hipFree / hipMalloc, whose cost varies from milliseconds to tenths of a second with the load of the machine.
 * HX_POOL=0 switches it off, HX_POOL_MAX_GIB caps the cached bytes (default: a quarter of the memory FREE on the device when the
 * pool starts, divided among the local ranks sharing the device: hx_pool_set_sharing), HX_POOL_MIN_BYTES the smallest
 * pooled block.  Reuse is ordered: a released block records an event on every stream that may still touch it (the caller streams
 * passed to the library since the last hx_pool_trim -- up to 8 -- and the library's own side streams) and the request that takes it out of the cache waits for
 * them on the host -- the device-wide synchronisation hipFree used to provide, without the device-wide part.  (A caller that
 * destroys a stream it passed to the library calls hx_pool_trim first.)
 * hx_pool_trim returns every cached block to the driver (call it when another allocator of the process runs out of memory);
 * hx_pool_stats: requests, requests served from the cache, calls that went to the driver and the seconds spent in them, bytes
 * cached now (reset != 0 clears the counters); hx_pool_ordering_stats: events recorded on release, events waited for on reuse,
 * the cap in bytes. */
int hx_pool_trim(void);
/* The number of local ranks that share THIS device (counted by the caller, e.g. equal PCI bus ids over the node's ranks): the
 * cap becomes (free memory at start) / 4 / that number.  Without it the pool guesses LOCAL_WORLD_SIZE / visible devices, and a
 * process that sees ONE device through ROCR_/HIP_VISIBLE_DEVICES counts as alone on it.  No effect when HX_POOL_MAX_GIB is set.
 * HX_POOL_VERBOSE=1 prints the cap chosen.  replaces: nothing in the reference (PETSc allocates through the host malloc). */
int hx_pool_set_sharing(int ranks_on_this_device);
int hx_pool_ordering_stats(int64_t* n_events, int64_t* n_waits, int64_t* max_cached_bytes);
int hx_pool_stats(int64_t* n_requests, int64_t* n_reused, int64_t* n_driver_calls, double* driver_seconds, int64_t* cached_bytes,
                  int reset);
int hx_version(void);
const char* hx_last_error(void);
/* SYNC. Device name, compute-unit count and HBM bytes of the current device. */
int hx_device_info(char* name, int name_len, int* cu_count, size_t* hbm_bytes);

/* ------------------------------------------------------------------ vectors (PETSc Vec / SLEPc BV)
 * replaces: VecAXPY / VecScale reached from petsc4py_utils.py:4-34, eigensolvers.py:174-176;
 *           BVDotVec / BVMultVec / VecNorm inside EPSSolve/PEPSolve (eigensolvers.py:62,113).   */
int hx_zaxpy(int64_t n, const double alpha[2], const hx_c128* x, hx_c128* y, hx_stream s);
int hx_zscal(int64_t n, const double alpha[2], hx_c128* x, hx_stream s);
/* out = sum_t coef[t] * xs[t], nterms <= 4; h_xs is a HOST array of device pointers. */
int hx_zlincomb(int64_t n, int nterms, const double* h_coefs, const hx_c128* const* h_xs,
                hx_c128* out, hx_stream s);
/* dst[slot[i]] += alpha * src[i]; slots must be unique. */
int hx_zscatter_axpy(int64_t m, const double alpha[2], const hx_c128* src, const int32_t* slot,
                     hx_c128* dst, hx_stream s);
/* slot[e] = position of entry e of the CSR pattern (rp, ci) inside the pattern (brp, bci) of the same n rows, columns sorted
 * within rows; *missing (device) = 1 if some entry is absent.  Replaces MatAXPY's DIFFERENT_NONZERO_PATTERN merge behind
 * `A + omega*B` (eigensolvers.py:174-176,240) by a slot map computed once per pattern pair. */
int hx_csr_locate_slots(int32_t n, const int32_t* rp, const int32_t* ci, const int32_t* brp, const int32_t* bci,
                        int32_t* slot, int32_t* missing, hx_stream s);
/* y[i] = x[idx[i]] (gather) */
int hx_zgather(int64_t m, const hx_c128* x, const int32_t* idx, hx_c128* y, hx_stream s);
/* Classical Gram-Schmidt panel: out[j] = V_j^H w (j < k), out[k] = w^H w.  V is column-major with
 * leading dimension ldv.  `partials` is workspace of hx_panel_dot_ws(k) complex numbers.
 * Deterministic two-stage reduction (no atomics). */
size_t hx_panel_dot_ws(int k);
int hx_zpanel_dot(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* w, hx_c128* out,
                  hx_c128* partials, hx_stream s);
/* Same inner products from a complex64 COPY of the panel (half the HBM traffic; w and the accumulation stay
 * complex128).  For Gram-Schmidt applied twice: the coefficients carry a 6e-8 relative error that the second pass
 * removes, while w -= V h keeps using the complex128 panel, so the Arnoldi relation holds exactly. */
int hx_zpanel_dot_c64(int64_t n, int k, const hx_c64* Vf, int64_t ldv, const hx_c128* w, hx_c128* out,
                      hx_c128* partials, hx_stream s);
/* w -= V[:, :k] h   (h: k complex numbers on the device) */
int hx_zpanel_sub(int64_t n, int k, const hx_c128* V, int64_t ldv, const hx_c128* h, hx_c128* w,
                  hx_stream s);
/* The same update from the complex64 copy of the panel: only for a correction pass whose coefficients are tiny
 * (|h| <= 1e-5 |w|): the Arnoldi relation then keeps an error of 6e-8 |h|, below the working precision. */
int hx_zpanel_sub_c64(int64_t n, int k, const hx_c64* Vf, int64_t ldv, const hx_c128* h, hx_c128* w, hx_stream s);
/* Vout[:, :kout] = V[:, :m] Q,  Q (m x kout) column-major on the device (Krylov-Schur restart). */
int hx_zpanel_mul(int64_t n, int m, int kout, const hx_c128* V, int64_t ldv, const hx_c128* Q,
                  hx_c128* Vout, int64_t ldvout, hx_stream s);
/* Small dense helpers of the solve path (round 4: these were torch calls).  hx_zsmall_gemv: y = A x for a row-major m x n matrix
 * on the device, m, n <= 64 (the r x r coefficient matrix of the Woodbury update of the flame term, flame_matrices.py:170-177 keeps
 * r = number of flames).  hx_zroll: dst[(i + shift) mod n] = src[i].  hx_zdense_inverse (SYNC): in-place inverse of a row-major
 * n x n matrix, n <= 4096, Gauss-Jordan with partial pivoting -- the coarsest operator of the V-cycle; HX_ERR_BREAKDOWN if singular.
 * replaces: the coarse LU inside the reference's PCLU (eigensolvers.py:50,103 via MUMPS) for that level. */
int hx_zsmall_gemv(int m, int n, const hx_c128* A, const hx_c128* x, hx_c128* y, hx_stream s);
int hx_zroll(int64_t n, int64_t shift, const hx_c128* src, hx_c128* dst, hx_stream s);
int hx_zdense_inverse(int n, hx_c128* A, hx_stream s);
/* Sparse dots t[f] = sum_i vals[ptr[f]+i] * y[idx[ptr[f]+i]]  (the b_f^T y of the rank-1 flame
 * operator kept factored; replaces the explicit D of flame_matrices.py:91-92,176,233). */
int hx_zsparse_dots(int nvec, const int32_t* ptr, const int32_t* idx, const hx_c128* vals,
                    const hx_c128* y, hx_c128* out, hx_stream s);
/* The other half of a product with the factored flame term: y[idx[p]] += coef * t[f] * vals[p] over the packed entries of the
 * left factors a_f (same packing as hx_zsparse_dots), the coefficients t[f] = b_f^T x read from DEVICE memory.  disjoint != 0:
 * the supports of the nvec vectors do not overlap (one launch); 0: one launch per vector in order (deterministic sums).
 * replaces: the rows of the explicit D in MatMult (flame_matrices.py:91-92,176,233 + petsc4py_utils.py:85). */
int hx_zsparse_axpys(int nvec, const int32_t* ptr, const int32_t* idx, const hx_c128* vals, const hx_c128* t,
                     const double coef[2], hx_c128* y, int disjoint, hx_stream s);

/* Threshold + pack a load vector (flame_matrices.py:66-72): entries with |Re| < tol / |Im| < tol zeroed separately,
 * surviving (index, value) pairs written to idx/val (capacity cap) in arbitrary order, *count = number found
 * (may exceed cap: then only cap pairs were stored). */
int hx_zthreshold_pack(int64_t n, const hx_c128* x, double tol, int32_t cap, int32_t* count, int32_t* idx,
                       hx_c128* val, hx_stream s);

/* ------------------------------------------------------------------ CSR SpMV (PETSc MatMult)
 * replaces: MatMult inside EPSSolve/PEPSolve (eigensolvers.py:62,113), Mat.mult
 * (petsc4py_utils.py:85,95).  y = alpha*A*x + beta*y.  lanes_per_row: 0 = choose from nnz/n_rows,
 * else one of 1,2,4,8,16,32,64. */
int hx_csr_spmv_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                  const hx_c128* val, const hx_c128* x, hx_c128* y, const double alpha[2],
                  const double beta[2], int lanes_per_row, hx_stream s);
/* y = alpha*A^T*x + beta*y is not provided: the operators of this path are complex symmetric or are
 * transposed once on the host (B_adj = B^H, acoustic_matrices.py:112-114). */

/* ------------------------------------------------------------------ SELL-64 companion of a CSR operator
 * Sliced ELLPACK with slice height 64 (= one wavefront): entries column-major inside a slice, padded to the slice's
 * longest row.  For the short, nearly uniform rows of P1 operators every wave-level load is one coalesced 1 KiB
 * transaction and a lane owns a whole row (no shuffle reduction, no row-pointer chase).  Same products as
 * hx_csr_spmv_z / hx_csr_jacobi_*; the handle owns its arrays.  hx_sell_create is SYNC.
 * want_c64: 0 none, 1 complex64 value copy for the *_c64 entry points, 2 the same copy STORED as scaled complex-half
 * (8 instead of 12 bytes per entry, arithmetic still complex64) -- for the V-cycle preconditioner only. */
typedef struct hx_sell hx_sell;
int hx_sell_create(hx_sell** out, int32_t n, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
                   int want_c128, int want_c64, hx_stream s);
int hx_sell_info(const hx_sell* h, int32_t* n_slices, int64_t* padded_entries);
/* Index compression, decided slice by slice at creation: a slice whose stored entries all satisfy |column - row| <= 32767
 * (banded numberings) is streamed through 16-bit column DELTAS, two index bytes per entry less; the remaining "wide"
 * slices (e.g. the rows closing a periodic ring) keep 32-bit columns.  hx_sell_index_bits: 16 if any slice is narrow,
 * else 32; hx_sell_wide_slices: number of slices that stayed on 32-bit columns. */
int hx_sell_index_bits(const hx_sell* h);
int hx_sell_wide_slices(const hx_sell* h);
/* Stencil slices, also decided at creation: when the 64 rows of a slice draw their columns from ONE list of deltas
 * (column - row) at most 1/8 longer than the longest row -- structured or extruded meshes numbered along a grid line --
 * the slice stores that list once and no per-entry index at all (entry k of a row is column row + delta_k; a row that
 * lacks it holds a zero).  Lossless; hx_sell_stencil_slices: number of such slices (SYNC). */
int hx_sell_stencil_slices(const hx_sell* h);
int64_t hx_sell_stencil_entries(const hx_sell* h);      /* stored entries that carry no index */
/* Overflow tails (unstructured meshes): a slice may keep a body narrower than its longest row and store the longer rows'
 * remaining entries packed in a tail that the same kernel adds after the body (option sell_tail / HX_SELL_TAIL, default 1).
 * tail_entries: entries stored in tails (with 32-bit columns; hx_sell_info counts the bodies only); split_slices: slices
 * that have one.  Either pointer may be NULL. */
int hx_sell_tail_info(const hx_sell* h, int64_t* tail_entries, int32_t* split_slices);
int hx_sell_spmv_z(const hx_sell* h, const hx_c128* x, hx_c128* y, const double alpha[2], const double beta[2],
                   hx_stream s);
int hx_sell_spmv_c64(const hx_sell* h, const hx_c64* x, hx_c64* y, float alpha, float beta, hx_stream s);
/* xout = xin + omega*dinv.*(b - A xin) + mom*(xin - xprev); xprev may be NULL (previous iterate = 0 when mom != 0) */
int hx_sell_smooth_z(const hx_sell* h, const hx_c128* dinv, const hx_c128* b, const hx_c128* xin, hx_c128* xout,
                     double omega, const hx_c128* xprev, double mom, hx_stream s);
int hx_sell_smooth_c64(const hx_sell* h, const hx_c64* dinv, const hx_c64* b, const hx_c64* xin, hx_c64* xout,
                       float omega, const hx_c64* xprev, float mom, hx_stream s);
int hx_sell_destroy(hx_sell* h);
/* In-situ timing of hx_sell_spmv_z (the operator product = the roofline kernel): once enabled, every launch on an operator
 * with at least min_rows rows is bracketed by HIP events on its launch stream (up to 262144 launches, events created in
 * chunks as needed, no synchronisation); the read-out waits for the recorded events and returns their count and summed
 * duration; hx_profile_sell_skipped: launches that qualified but were not bracketed (cap reached, or inside a graph capture). */
int hx_profile_sell_spmv(int enable, int32_t min_rows);
int hx_profile_sell_spmv_read(int64_t* launches, double* total_ms);
/* the same for kind 0 = hx_sell_spmv_z, kind 1 = hx_sell_smooth_c64 (the V-cycle smoother step, the kernel with the largest
 * share of the run time) on operators with at least min_rows rows.  Kind 1 is SAMPLED: every 8th qualifying launch carries the
 * event pair (option "sell_profile_smoother_every"; an event pair costs ~10 us of a launch-bound call and the smoother is launched
 * 2.7 times as often as the product) -- launches passed over by the sampling do not count as skipped. */
int hx_profile_sell_read(int kind, int64_t* launches, double* total_ms);
int64_t hx_profile_sell_skipped(int kind);

/* ------------------------------------------------------------------ P1 assembly (DOLFINx assemble_*)
 * Vertex-centric "owner computes" assembly: one CSR row per vertex, gathered from the cells incident
 * to that vertex in ascending cell order => deterministic, no atomics, no inter-GPU exchange.
 * `ents` are simplices as (n_ent x k) int32 vertex ids (k = 2,3,4 for cells; facets use k = dim). */

/* vertex -> incident entity adjacency (CSR, entity ids ascending).  SYNC (uses a device scan). */
int hx_build_v2e(int32_t n_vert, int32_t n_ent, int k, const int32_t* ents, int32_t* v2e_ptr,
                 int32_t* v2e_idx, hx_stream s);
/* CSR pattern of the P1 operator on those entities: sorted unique columns per row.
 * Step 1 (SYNC): writes rowptr (n_vert+1) and returns nnz in *h_nnz; `scratch` holds k*k*n_ent int32.
 * Step 2: writes col (nnz). */
int hx_p1_pattern_count(int32_t n_vert, int32_t n_ent, int k, const int32_t* ents,
                        const int32_t* v2e_ptr, const int32_t* v2e_idx, int32_t* rowptr,
                        int32_t* scratch, int64_t* h_nnz, hx_stream s);
int hx_p1_pattern_fill(int32_t n_vert, int k, const int32_t* v2e_ptr, const int32_t* rowptr,
                       const int32_t* scratch, int32_t* col, hx_stream s);
/* A_jk = -int c^2 grad(phi_k).grad(phi_j) dx and C_jk = int phi_k phi_j dx on the same pattern
 * (acoustic_matrices.py:101-105, 121-125).  c is nodal (P1, n_vert) or cellwise (DG0, n_cells).
 * Either of A_val / C_val may be NULL. dim = 1,2,3 (cells have dim+1 vertices; x is n_vert x 3). */
int hx_assemble_p1_stiff_mass(int dim, int32_t n_vert, const double* x, const int32_t* cells,
                              const int32_t* v2c_ptr, const int32_t* v2c_idx, const int32_t* rowptr,
                              const int32_t* col, const hx_c128* c, int c_is_cellwise,
                              hx_c128* A_val, hx_c128* C_val, hx_stream s);
/* B_jk = sum_facets coef_f int_F c phi_k phi_j ds, coef_f = i/Z_tag (acoustic_matrices.py:71,83,95,
 * 108-110).  facets are (n_facets x dim) vertex ids; c nodal, or cellwise through facet_cell. */
int hx_assemble_p1_robin(int dim, int32_t n_vert, const double* x, const int32_t* facets,
                         const int32_t* v2f_ptr, const int32_t* v2f_idx, const int32_t* rowptr,
                         const int32_t* col, const hx_c128* c, int c_is_cellwise,
                         const int32_t* facet_cell, const hx_c128* coef_per_facet, hx_c128* B_val,
                         hx_stream s);
/* DOLFINx assemble_matrix(bcs=...) semantics: zero rows+cols flagged in is_bc, diagonal = diag. */
int hx_csr_apply_dirichlet(int32_t n, const int32_t* rowptr, const int32_t* col, hx_c128* val,
                           const uint8_t* is_bc, double diag, hx_stream s);
/* Load vectors (DOLFINx assemble_vector; flame_matrices.py:63,141,199-200):
 *  kind 0: out_j = scale * int f_cell phi_j dx                 (f DG0: fa = f_cell (n_cells), fb unused)
 *  kind 1: out_j = scale * int fa fb phi_j dx                  (fa, fb nodal P1; exact cubic rule)
 *  kind 2: out_k = scale * int (n_r . grad phi_k) fb / fa dx   (fa = rho, fb = w nodal; degree-2 rule)
 * n_r = e_x (dim 1,2) or e_z (dim 3), flame_matrices.py:35-40.
 * cell_tag (may be NULL): when given, only cells with cell_tag[c] == tag contribute (the dx(flame) subdomain
 * measure of flame_matrices.py:137,141). */
int hx_assemble_p1_load(int kind, int dim, int32_t n_vert, const double* x, const int32_t* cells,
                        const int32_t* v2c_ptr, const int32_t* v2c_idx, const hx_c128* fa,
                        const hx_c128* fb, const double scale[2], const int32_t* cell_tag, int32_t tag,
                        hx_c128* out, hx_stream s);
/* The same entries for a LIST of vertices: out[t] = entry rows[t], restricted (cell_tag given) to the cells tagged
 * row_tag[t].  The load vector of a flame (dx(flame), flame_matrices.py:139-141) lives on the few hundred vertices of its
 * subdomain, so all flames are assembled by one launch over the union of their supports instead of one sweep of the mesh
 * per flame. */
int hx_assemble_p1_load_rows(int kind, int dim, int32_t n_rows, const int32_t* rows, const int32_t* row_tag,
                             const double* x, const int32_t* cells, const int32_t* v2c_ptr, const int32_t* v2c_idx,
                             const hx_c128* fa, const hx_c128* fb, const double scale[2], const int32_t* cell_tag,
                             hx_c128* out, hx_stream s);

/* |K_c| of every cell (assemble_scalar(Constant(1)*dx(tag)) in parameters_utils.py:215-216,235-236). */
int hx_p1_cell_volumes(int dim, int32_t n_cells, const double* x, const int32_t* cells, double* vol,
                       hx_stream s);
/* Lowest-index cell containing each query point (pts: n_pts x 3), barycentric tolerance tol; replaces
 * determine_point_ownership (flame_matrices.py:144-145).  out_cell must be pre-filled with INT32_MAX. */
int hx_p1_locate_points(int dim, int32_t n_cells, const double* x, const int32_t* cells, int n_pts,
                        const double* pts, double tol, int32_t* out_cell, hx_stream s);

/* ---- P2 (quadratic Lagrange) on tetrahedra with P1 coefficient fields: the reference's degree = 2 runs
 * (acoustic_matrices.py:25-125 and flame_matrices.py:199-200 with functionspace(mesh, ("Lagrange", 2));
 * ShapeSensitivities/RijkeFFD/main_shape.py:68-83).  One CSR row per P2 dof (vertices first, then one dof per edge);
 * cell_dofs: n_cells x 10 (4 vertices, then the edges (0,1),(0,2),(0,3),(1,2),(1,3),(2,3)); facet_dofs: n_facets x 6
 * (3 vertices, then the edges (0,1),(0,2),(1,2)).  The dof -> cell adjacency and the CSR pattern come from
 * hx_build_v2e / hx_p1_pattern_count / hx_p1_pattern_fill with k = 10 (resp. 6).  quad: barycentric quadrature table on
 * the device, nq x 5 doubles (4 barycentric coordinates + weight, weights summing to 1) for cells, nq x 4 for facets. */
int hx_assemble_p2_stiff_mass(int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                              const int32_t* d2c_ptr, const int32_t* d2c_idx, const int32_t* rowptr, const int32_t* col,
                              const hx_c128* c /* P1 nodal, may be NULL (= 1) */, int nq, const double* quad,
                              hx_c128* A /* may be NULL */, hx_c128* C /* may be NULL */, hx_stream s);
int hx_assemble_p2_robin(int32_t n_dof, const double* x, const int32_t* facets, const int32_t* facet_dofs,
                         const int32_t* d2f_ptr, const int32_t* d2f_idx, const int32_t* rowptr, const int32_t* col,
                         const hx_c128* c, const hx_c128* coef_per_facet, int nq, const double* quad, hx_c128* B,
                         hx_stream s);
/* kind 1: out_j = scale * int fa fb phi_j dx;  kind 2: out_j = scale * int (d phi_j / dz) fb / fa dx   (fa, fb P1 nodal) */
int hx_assemble_p2_load(int kind, int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                        const int32_t* d2c_ptr, const int32_t* d2c_idx, const hx_c128* fa, const hx_c128* fb,
                        const double scale[2], int nq, const double* quad, hx_c128* out, hx_stream s);
/* Pointwise flames on the quadratic space: out_j = scale * sum_{cells with cell_tag == tag} h_cell int_cell phi_j dx  (h DG0).
 * replaces: flame_matrices.py:139-141 -- assemble_vector(form((gamma - 1) q_0 / u_b inner(h, phi_j) dx(flame))) with degree = 2 */
int hx_assemble_p2_load_cells(int32_t n_dof, const double* x, const int32_t* cells, const int32_t* cell_dofs, const int32_t* d2c_ptr,
                              const int32_t* d2c_idx, const hx_c128* h_cell, const int32_t* cell_tag, int32_t tag,
                              const double scale[2], int nq, const double* quad, hx_c128* out, hx_stream s);

/* ------------------------------------------------------------------ sparse triangular solves (MatSolve)
 * replaces: KSP(preonly)+PC(LU) MatSolve inside ST sinvert (eigensolvers.py:49-50,102-103 -> :62,:113).
 * The factors come from a host factorisation (Pr*A*Pc = L*U); each triangle is passed as strictly
 * triangular CSR + inverse diagonal.  Level-scheduled on the device. */
typedef struct hx_sptrsv hx_sptrsv;
/* h_* are HOST arrays (analysed on the host, then uploaded).  lower != 0 for L. SYNC. */
int hx_sptrsv_create(hx_sptrsv** out, int32_t n, int lower, const int32_t* h_rowptr,
                     const int32_t* h_col, const hx_c128* h_val, const hx_c128* h_inv_diag);
int hx_sptrsv_info(const hx_sptrsv* t, int32_t* n_levels, int32_t* n_launches, int64_t* nnz);
/* x = T^{-1} b  (b, x device vectors of length n; may alias) */
int hx_sptrsv_solve(hx_sptrsv* t, const hx_c128* b, hx_c128* x, hx_stream s);
int hx_sptrsv_destroy(hx_sptrsv* t);

/* ------------------------------------------------------------------ sparse direct solver (device multifrontal LU)
 * replaces: KSP(preonly) + PC(LU): MatLUFactorSymbolic / MatLUFactorNumeric / MatSolve / MatSolveTranspose behind ST sinvert
 * (eigensolvers.py:49-50,102-103 -> :62,:113), i.e. PETSc's LU or MUMPS under -st_pc_factor_mat_solver_type.
 * Supernodal multifrontal LU of a complex128 CSR matrix (pattern symmetrised for the analysis), nested-dissection ordering,
 * fronts factorised on the device with f64 MFMA (csrc/splu.hip); partial pivoting inside 64 x 64 diagonal blocks, static
 * pivoting otherwise (replaced pivots are counted in hx_splu_info: check the residual, refine).  Solves are supernodal
 * (dense panels), bitwise reproducible; the transposed solve uses the same factors. */
typedef struct hx_splu hx_splu;
/* HOST analysis (no device needed): h_rowptr / h_col = CSR pattern, h_coords = n x dim vertex coordinates of the unknowns
 * (inertial nested dissection) or NULL (level-set nested dissection on the graph); leaf_size <= 0: default 64;
 * flags bit 0: the pattern is structurally symmetric with sorted rows (the analysis then skips forming A + A^T).  The first
 * levels of the dissection and the entry map run on HX_HOST_THREADS host threads (default 8).  SYNC. */
int hx_splu_analyse(hx_splu** out, int32_t n, const int32_t* h_rowptr, const int32_t* h_col, const double* h_coords,
                    int dim, int leaf_size, int flags);
/* iout[16]: n, supernodes, levels, nnz(L) (= nnz(U), diagonal included), factor / inverse / temporary / workspace elements,
 * largest front, widest supernode, border rows, nnz of the input, replaced pivots of the last factorisation, launches of
 * the factorisation / forward / backward sweeps;  dout[8]: real flops of the (unsymmetric) factorisation, seconds of the analysis
 * (host), seconds of the last numeric factorisation (HIP events), flops of the selective inversion, flops the last
 * factorisation actually executed, 1.0 when it ran in symmetric mode. */
int hx_splu_info(const hx_splu* h, int64_t* iout, double* dout);
/* the symbolic structure by field name (tests, diagnostics): perm, iperm, sfirst, sparent, sptr, srow, rel, slevel, level_ptr,
 * level_list, child_ptr, child_list, loff, uoff, ioff, soff, woff, amap.  h_out NULL: only *count.  HOST buffers. */
int hx_splu_symbolic(const hx_splu* h, const char* name, void* h_out, int64_t capacity_bytes, int64_t* count);
/* numeric factorisation of the values `val` (device, in the CSR order of the analysed pattern).  May be repeated with new
 * values.  A matrix with a symmetric pattern (flags bit 0 of the analysis) whose VALUES are symmetric too (checked on the device,
 * 1e-13 relative: the complex symmetric K = A + sigma B + sigma^2 C) is factorised in symmetric mode: the U panels are transposed
 * copies of the unscaled L panels, half the products (option splu_symmetric=0: never).  SYNC (reads back the pivot counter). */
int hx_splu_factor(hx_splu* h, const hx_c128* val, hx_stream s);
/* x = A^-1 b (transposed != 0: A^-T b; after a factorisation in symmetric mode that is the same system and takes the same fast
 * sweeps); b, x device vectors of length n, may alias. */
int hx_splu_solve(hx_splu* h, const hx_c128* b, hx_c128* x, int transposed, hx_stream s);
int hx_splu_destroy(hx_splu* h);

/* ------------------------------------------------------------------ AMG-preconditioned GMRES (iterative MatSolve)
 * replaces: the same KSP/PC solve as above for 3-D problems where a sparse direct factorisation does not fit
 * (the reference switches PCLU to MUMPS/SuperLU_DIST there).  Hierarchy arrays are device pointers owned by
 * the caller and must outlive the handle. */
/* xout = xin + omega * dinv .* (b - A xin)   (fused damped-Jacobi sweep; xin != xout) */
int hx_csr_jacobi_z(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                    const hx_c128* val, const hx_c128* dinv, const hx_c128* b, const hx_c128* xin,
                    hx_c128* xout, double omega, hx_stream s);
/* complex64 forms of the two kernels above (y = alpha*A*x + beta*y with real alpha, beta). */
int hx_csr_spmv_c64(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                    const hx_c64* val, const hx_c64* x, hx_c64* y, float alpha, float beta, hx_stream s);
int hx_csr_jacobi_c64(int32_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                      const hx_c64* val, const hx_c64* dinv, const hx_c64* b, const hx_c64* xin,
                      hx_c64* xout, float omega, hx_stream s);
/* x = scale * d .* b */
int hx_zdiag_scale(int64_t n, double scale, const hx_c128* d, const hx_c128* b, hx_c128* x, hx_stream s);
typedef struct hx_amg hx_amg;
int hx_amg_create(hx_amg** out, int n_levels, int pre, int post, double omega);
/* level l (0 = finest): operator K_l, inverse diagonal, prolongator P_l (n x n_coarse) and restriction
 * R_l = P_l^T (n_coarse x n); transfer operators are ignored on the coarsest level. */
int hx_amg_set_level(hx_amg* h, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK,
                     const hx_c128* vaK, const hx_c128* dinv, int32_t n_coarse, int64_t nnzP,
                     const int32_t* rpP, const int32_t* ciP, const hx_c128* vaP, int64_t nnzR,
                     const int32_t* rpR, const int32_t* ciR, const hx_c128* vaR);
/* The prolongators and restrictions of a smoothed-aggregation hierarchy are REAL: after hx_amg_set_real_transfers(h, 1) the
 * vaP / vaR arguments of hx_amg_set_level / hx_amg_set_level_c are real arrays (double, resp. float: cast the pointer), which
 * the transfer products stream at 12 / 8 instead of 20 / 12 bytes per entry. */
int hx_amg_set_real_transfers(hx_amg* h, int flag);
/* Optional SELL-64 companion of K_l (must hold the value copy of the precision the V-cycle runs in): the smoother
 * steps and the residual product of that level then use the SELL kernels. */
int hx_amg_set_level_sell(hx_amg* h, int l, const hx_sell* sell);
/* Smoother of level l: rho = 0 -> `pre`/`post` damped-Jacobi sweeps with the weight given to hx_amg_create;
 * rho > 0 (estimate of the spectral radius of D^-1 K_l) -> Chebyshev polynomial of degree pre/post on D^-1 K_l
 * targeting [0.11 rho, 1.1 rho], evaluated by the same fused kernel as a three-term recurrence. */
int hx_amg_set_smoother(hx_amg* h, int l, double rho);
/* dense inverse of the coarsest operator, passed as a (dense) CSR */
int hx_amg_set_coarse(hx_amg* h, int32_t n_c, int64_t nnz, const int32_t* rp, const int32_t* ci,
                      const hx_c128* va);
/* Optional complex64 copy of every level (same integer arrays may be shared).  Once set, hx_amg_vcycle narrows
 * its input, runs the whole cycle in single precision (half the HBM traffic) and widens the result; use it only
 * as the preconditioner of hx_gmres, which is flexible and measures residuals in double precision. */
int hx_amg_set_level_c(hx_amg* h, int l, int32_t n, int64_t nnzK, const int32_t* rpK, const int32_t* ciK,
                       const hx_c64* vaK, const hx_c64* dinv, int32_t n_coarse, int64_t nnzP,
                       const int32_t* rpP, const int32_t* ciP, const hx_c64* vaP, int64_t nnzR,
                       const int32_t* rpR, const int32_t* ciR, const hx_c64* vaR);
int hx_amg_set_coarse_c(hx_amg* h, int32_t n_c, int64_t nnz, const int32_t* rp, const int32_t* ci,
                        const hx_c64* va);
/* hipGraph replay of the complex64 V-cycle: all buffers of the cycle belong to the handle, so its ~45 kernel launches are
 * captured once (on a private stream, ordered against the caller's stream by events) and replayed by one hipGraphLaunch
 * per cycle.  For small / mid-size problems where the cycle is launch-bound; falls back to eager launches if the
 * capture fails; not used in row-sharded mode.  hx_amg_graph_active: 1 once cycles run from the graph. */
int hx_amg_enable_graph(hx_amg* h, int enable);
int hx_amg_graph_active(const hx_amg* h);
/* x = one V-cycle applied to b from a zero initial guess (b != x) */
int hx_amg_vcycle(hx_amg* h, const hx_c128* b, hx_c128* x, hx_stream s);
int hx_amg_destroy(hx_amg* h);
/* FLEXIBLE right-preconditioned restarted GMRES on K x = b (x0 = 0), CGS2 with the fused panel kernels; M may be NULL.
 * ws: hx_gmres_ws(n, restart) complex numbers.  SYNC (one small D2H per orthogonalisation pass). */
size_t hx_gmres_ws(int64_t n, int restart);
/* workspace of hx_gmres / hx_gmres_sharded / hx_amg_set_recycle for a handle with hx_amg_enable_recycling: the Arnoldi vectors
 * live in the handle's [C; V] panel, so the workspace is (restart + 1) x n entries smaller (3.8 GiB at 5 M rows, restart 50) */
size_t hx_gmres_ws_recycling(int64_t n, int restart);
int hx_gmres(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
             const hx_c128* val, const hx_sell* Ksell /* optional SELL-64 companion of the same K */,
             const hx_c128* b, hx_c128* x, double rtol, int max_it, int restart,
             hx_c128* ws, int* h_iters, double* h_relres, hx_stream s);
/* ---- Krylov subspace recycling across the solves of one fixed-point call (all of them share K; the reference instead
 * re-factorises A - D(omega_k) every iteration, eigensolvers.py:165-178).  GCRO with a fixed operator: k pairs (U, C),
 * K U = C, C^H C = I, harvested once as the harmonic Ritz vectors of the first solve.  hx_gmres / hx_gmres_sharded
 * then start every cycle from the Galerkin guess on span(U) and keep the Arnoldi vectors orthogonal to C, which
 * removes the initial plateau caused by the few near-null modes of the preconditioned operator.
 *   hx_amg_enable_recycling  allocate the [C; V] panel and U inside the handle (n = rows of the local operator); called again it
 *                            re-sizes them for another k_max / restart and drops the harvested pairs (the hierarchy stays). SYNC then.
 *   hx_gmres_hessenberg      raw (m+1) x m Hessenberg of the last cycle of the last solve (column-major, ld m+1);
 *                            *m = 0 if that solve already used recycled pairs.  Call with H = NULL to query m.
 *   hx_amg_set_recycle       U = Z_m Ucoef, C = V_{m+1} Ccoef (HOST coefficients, column-major m x k and (m+1) x k);
 *                            must directly follow the harvested solve, with the same ws. */
int hx_amg_enable_recycling(hx_amg* h, int64_t n, int k_max, int restart);
int hx_gmres_hessenberg(const hx_amg* h, int* m, hx_c128* H);
int hx_amg_set_recycle(hx_amg* h, int k, const hx_c128* Ucoef, const hx_c128* Ccoef, hx_c128* ws, hx_stream s);
/* ---- Short-recurrence alternative to hx_gmres for COMPLEX SYMMETRIC K (K^T = K: every shifted operator of this path
 * except the Bloch-reduced ones) with a complex symmetric preconditioner (V-cycle with pre == post smoothing steps and
 * R = P^T): deflated preconditioned COCG -- CG in the bilinear form x^T y.  No Gram-Schmidt panel: per iteration one
 * operator product, one V-cycle, two reductions and two fused vector updates; all scalars stay on the device, the host
 * reads |r|^2 once per iteration.  When the handle carries harvested pairs (hx_amg_set_recycle: K U = C) they are used
 * as a deflation space: x0 = U E^-1 U^T b and p = z - U E^-1 C^T z + beta p with E = U^T K U (complex64 mirrors of U and
 * C are read: the direction may be perturbed freely, x and r stay consistent).  Convergence is accepted against the TRUE
 * residual.  Returns HX_ERR_BREAKDOWN on a breakdown of the bilinear form or a stall: repeat the solve with hx_gmres.
 * ws: hx_cocg_ws(n) complex numbers (the hx_gmres workspace is large enough).  SYNC. */
size_t hx_cocg_ws(int64_t n);
/* Warm start: the NEXT hx_cocg / hx_cocg_sharded call on this handle takes the contents of its x argument as the initial
 * guess instead of zero (one solve; a guess whose residual exceeds |b| / 10 is dropped).  Used for sequences of related
 * right-hand sides -- the Woodbury columns K^-1 u_f of identical flames (flame_matrices.py:170-177 loops over them). */
int hx_amg_next_solve_uses_x(hx_amg* M);

/* When the recurrence stops improving above the tolerance, hx_cocg measures the normwise backward error
 * |b - K x| / (|K|_inf |x| + |b|) and accepts the iterate below 10 eps: what a backward-stable direct solve (the
 * reference's PCLU) delivers when the shift sits next to an eigenvalue and |x| >> |b|.  The value of the last such solve
 * (-1: the tolerance was met, nothing measured).  hx_gmres applies the same test when its Arnoldi estimate meets the
 * tolerance but the true residual does not, and reports through the same getter: */
int hx_cocg_backward_error(const hx_amg* M, double* out);
int hx_cocg(hx_amg* M, int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const hx_c128* val,
            const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it, hx_c128* ws, int* h_iters,
            double* h_relres, hx_stream s);
/* ---- Blocks of right-hand sides (csrc/block.hip, csrc/cocg_block.hip; single GPU).  What the reference gets from its direct
 * factorisation -- one MatSolve serves any number of right-hand sides (eigensolvers.py:50,103; the flame columns of
 * flame_matrices.py:170-177 are 16 solves with one operator) -- the iterative route gets by running hx_block_width() (= 4)
 * INDEPENDENT deflated COCG recurrences side by side: the vectors of a block are interleaved row by row (X[row * 4 + c]), every
 * matrix entry is loaded once per pass and multiplies four adjacent operands (V-cycle smoother: 60 + 40 -> 15 + 40 bytes per
 * row and vector; gathers of 32 / 64 contiguous bytes per lane).  hx_zblock_pack / _unpack convert between nb <= 4 rows of a
 * row-major panel and the interleaved block (missing columns: zero).  hx_sell_spmm_z: Y = alpha K X + beta Y on blocks.
 * hx_cocg_block: K X = B for the first ncols columns; needs a handle whose complex64 V-cycle and SELL-64 fine level are set and
 * whose deflation pairs (if any) were prepared by an earlier hx_cocg solve; h_iters / h_relres: 4 entries (iterations and TRUE
 * relative residual per column -- a column above 10 rtol is reported as it is, the caller decides); ws: hx_cocg_block_ws(n)
 * complex128 entries. */
int hx_block_width(void);
int hx_zblock_pack(int64_t n, int nb, const hx_c128* panel, int64_t ld, hx_c128* blk, hx_stream s);
int hx_zblock_unpack(int64_t n, int nb, const hx_c128* blk, hx_c128* panel, int64_t ld, hx_stream s);
int hx_sell_spmm_z(const hx_sell* h, const hx_c128* X, hx_c128* Y, const double alpha[2], const double beta[2], hx_stream s);
size_t hx_cocg_block_ws(int64_t n);
int hx_cocg_block(hx_amg* M, int32_t n, const hx_sell* Ksell, int ncols, const hx_c128* B, hx_c128* X, double rtol, int max_it,
                  hx_c128* ws, int* h_iters, double* h_relres, hx_stream s);
/* ---- Row-sharded mode (one process per GPU).  Replaces what the reference gets from PETSc's
 * parallel Mat / KSP / PC under mpirun (eigensolvers.py:43-62 with a distributed A; run lines
 * numerical_examples/AnnularCombustor/Micca/fullAnnulus/runAll.sh:9-10): the first `n_sharded` levels of the hierarchy
 * hold THIS rank's rows of K_l over [owned | ghost] columns (hx_amg_set_level* with n = owned rows; P_l: owned rows over
 * the [owned | ghost] columns of level l + 1; R_l: owned rows of level l + 1 over the [owned | ghost] columns of level l,
 * i.e. the owner computes its coarse right-hand side from exchanged fine residuals), every coarser level is replicated
 * (the restriction onto the first replicated level sums the ranks' partial right-hand sides with one all-reduce).
 * The two exchange steps go through callbacks, both enqueued on / ordered against stream `s`:
 *   halo(ctx, level, x_ext, elem_bytes, s)   fill the ghost block [n_own_l, n_ext_l) of the level-l vector x_ext (elements of
 *                                            elem_bytes = 8: complex64, 16: complex128) from the owning ranks
 *   allreduce(ctx, buf, count, dtype, op, s) reduce `count` REAL numbers in place over all ranks (dtype 0: float32,
 *                                            1: float64; op 0: sum, 1: max); complex data = 2 reals per entry
 * Production: hx_comm_halo / hx_comm_allreduce with an hx_comm as ctx (RCCL issued by the library, below).  The caller
 * may supply its own (the gloo rehearsal path of the tests wraps torch.distributed).  Both return 0 on success. */
typedef int (*hx_halo_fn)(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s);
typedef int (*hx_allreduce_fn)(void* ctx, void* buf, int64_t count, int dtype, int op, hx_stream s);
/* Call after every level is set.  n_ext[l] = owned + ghost length of level l < n_sharded (1 <= n_sharded < levels): the
 * handle re-allocates its level-l work vectors with that length. */
int hx_amg_set_sharded(hx_amg* h, int n_sharded, const int32_t* n_ext, hx_halo_fn halo, hx_allreduce_fn allreduce,
                       void* ctx);
/* Deep-halo mode of row-sharded level l (after hx_amg_set_sharded -- whose n_ext[l] then counts owned rows + ALL ghost layers -- after
 * hx_amg_set_prolongator_ghost_rows, with the level set with an inverse diagonal of n_ext[l] entries and a prolongator of n_ext[l]
 * rows).  K1 = operator rows of the level's ghost layers (sorted CSR on the device, columns in the extended numbering
 * [owned | layer 1 | layer 2 (| layer 3)], values in the precision of the handle's cycle), fetched from their owners at set-up:
 * layers = 2: rows of layer 1 only (n_gs = n_g1); layers = 3: rows of layer 1 (the first n_g1 rows / nnz_g1 entries) followed by
 * those of layer 2 (n_gs rows / nnz_gs entries in all).  The cycle exchanges the level's right-hand side over all layers in ONE
 * exchange (the callback is called with level number 32 + l), smooths the layer rows redundantly and forms the residual of layer 1
 * itself, so restriction and post-smoothing need no exchange; with two layers the pre-smoothed iterate is exchanged once more.
 * Exchanges per cycle and level: 5 classic, 3 / 2 (coarse level sharded / replicated) with two layers, 2 / 1 with three.  Same
 * arithmetic.  pre = post = 2 only; option amg_deep_halo = 0 falls back to the classic schedule.  With the split callbacks set
 * (hx_amg_set_halo_overlap) and slice lists on the level's SELL-64 operator, the exchange of the right-hand side is split like the
 * operator product's: the owned part of x1 and the interior slices of the first smoothing step run while it travels (option
 * amg_deep_overlap = 0: blocking).
 * replaces: the ghost updates inside PCApply of the reference's distributed preconditioner (eigensolvers.py:49-50,102-103 under
 * mpirun, fullAnnulus/runAll.sh:10-11). */
int hx_amg_set_level_deep(hx_amg* h, int l, int layers, int32_t n_g1, int64_t nnz_g1, int32_t n_gs, int64_t nnz_gs, const int32_t* rp1,
                          const int32_t* ci1, const void* va1);
/* Optional split form of the halo exchange.  begin(ctx, level, x_ext, elem_bytes, s, &interior) enqueues the exchange ON s,
 * ordered after the work enqueued there, and hands out a side stream `interior` that waits for the same point of s;
 * end(ctx, level, s) makes s wait for the side stream.  With both set, every product with a SELL-64 operator of a sharded level
 * (smoother steps, residuals, the operator product of hx_cocg_sharded / hx_gmres_sharded) launches its interior slices -- the
 * ones that read no ghost column -- on `interior` between the two and its boundary slices on s after: the exchange travels over
 * xGMI while most of the product streams from HBM, and every operation of the communicator (split and blocking exchanges,
 * all-reduces) stays on ONE stream in program order -- RCCL orders the operations of a communicator only within a stream.
 * hx_comm_halo_begin / hx_comm_halo_end implement it.  (PETSc overlaps the same way: VecScatterBegin, local part of MatMult,
 * VecScatterEnd, off-diagonal part.) */
typedef int (*hx_halo_begin_fn)(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s, hx_stream* interior);
typedef int (*hx_halo_end_fn)(void* ctx, int level, hx_stream s);
int hx_amg_set_halo_overlap(hx_amg* h, hx_halo_begin_fn begin, hx_halo_end_fn end);
/* The prolongators of the first n_levels row-sharded levels were given with n_ext ROWS (behind this rank's rows the owners' rows of
 * its ghost vertices): the prolongation then also corrects the ghost block of the iterate, and the first post-smoothing step of a
 * cycle runs without an exchange -- one halo fewer per sharded level and cycle.  After hx_amg_set_sharded.  (PETSc: the ghost
 * update MatMult would do in front of that step, /root/reference/helmholtz_x/eigensolvers.py:62,113 via KSPSolve.) */
int hx_amg_set_prolongator_ghost_rows(hx_amg* h, int n_levels);
/* ---- RCCL communicator owned by the library (comm.hip).  librccl is bound at run time (the copy the process already
 * loaded, else librccl.so.1; HX_RCCL_LIB overrides).  Rank 0 draws an id (hx_comm_unique_id) and ships its
 * HX_COMM_ID_BYTES bytes to the other ranks by whatever channel the host side has (torch.distributed broadcast); every
 * rank then calls hx_comm_create (collective, its GPU current).  Exchange plans are set once per hierarchy level from HOST
 * arrays; hx_comm_exchange / hx_comm_halo / hx_comm_allreduce enqueue ncclSend/ncclRecv groups and ncclAllReduce on the
 * caller's stream -- no host synchronisation.
 * replaces: VecScatterBegin/End inside MatMult and MPI_Allreduce inside BVDot / VecNorm of the reference's PETSc / SLEPc
 * objects (reached from eigensolvers.py:62, 113); dolfinx_utils.py:187-207 are its only hand-written collectives. */
#define HX_COMM_ID_BYTES 128
typedef struct hx_comm hx_comm;
int hx_comm_unique_id(void* out, int nbytes);
int hx_comm_create(hx_comm** out, int rank, int world, const void* id, int nbytes);
int hx_comm_destroy(hx_comm* c);
const char* hx_comm_library(void);        /* which librccl was bound ("" before the first use) */
/* plan: send_peer[ns], send_ptr[ns + 1] into send_idx (indices into the source buffer, per peer) and recv_peer[nr],
 * recv_off[nr], recv_cnt[nr] (element range of the destination buffer filled by that peer).  HOST arrays, copied. */
int hx_comm_set_plan(hx_comm* c, int plan, int ns, const int32_t* send_peer, const int64_t* send_ptr,
                     const int32_t* send_idx, int nr, const int32_t* recv_peer, const int64_t* recv_off,
                     const int64_t* recv_cnt);
int hx_comm_exchange(hx_comm* c, int plan, const void* src, void* dst, int elem_bytes, hx_stream s);
int hx_comm_halo(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s);          /* an hx_halo_fn */
int hx_comm_halo_begin(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s, hx_stream* interior); /* an hx_halo_begin_fn */
int hx_comm_halo_end(void* ctx, int level, hx_stream s);                                   /* ... and its hx_halo_end_fn */
int64_t hx_comm_overlapped(const hx_comm* c);   /* exchanges that ran split since the last counter reset */
int64_t hx_comm_stream_switches(const hx_comm* c); /* operations that arrived on another stream than their predecessor (product path: 0) */
int hx_comm_allreduce(void* ctx, void* buf, int64_t count, int dtype, int op, hx_stream s); /* an hx_allreduce_fn */
/* recv[p * bytes ... (p + 1) * bytes) <- rank p's `send` block of `bytes` bytes, on every rank (one group of ncclSend / ncclRecv on
 * the caller's stream): the right-hand sides of the replicated direct solve on the row-sharded path (slepc.ReplicatedLUSolver;
 * the reference: MUMPS' centralised right-hand side under mpirun, eigensolvers.py:49-50). */
int hx_comm_allgather(void* ctx, const void* send, void* recv, int64_t bytes, hx_stream s);
int hx_comm_counters(const hx_comm* c, int64_t* n_exchange, int64_t* n_allreduce, int reset);
/* hx_gmres on the row-sharded operator: K = local rows over [owned | ghost] columns, b / x / ws sized by n_own;
 * xe (n_ext complex128: staging vector for the K products) and hbuf (restart + 2 + k_max complex128: inner products, all-reduced)
 * are caller-owned device arrays.  Every rank must call it collectively. */
int hx_gmres_sharded(hx_amg* M, int32_t n_own, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                     const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it,
                     int restart, hx_c128* ws, hx_c128* xe, hx_c128* hbuf, int* h_iters, double* h_relres, hx_stream s);
/* hx_cocg on the row-sharded operator (same buffers as hx_gmres_sharded; hbuf needs at least 72 entries). */
int hx_cocg_sharded(hx_amg* M, int32_t n_own, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                    const hx_c128* val, const hx_sell* Ksell, const hx_c128* b, hx_c128* x, double rtol, int max_it,
                    hx_c128* ws, hx_c128* xe, hx_c128* hbuf, int* h_iters, double* h_relres, hx_stream s);
/* ---- DEVICE set-up passes of the smoothed-aggregation hierarchy (once per K; all arrays are device pointers, K never
 * leaves HBM).  replaces: PCSetUp behind ST sinvert -- the analysis + numeric factorisation MUMPS / SuperLU_DIST run for the
 * reference at every E.solve() (eigensolvers.py:49-50, 62, 102-103, 113); here once per fixed-point call.  Same mathematics as the hx_host_* passes below, which remain for the row-sharded mode.
 *   hx_csr_diagonal_z       diagonal of a complex CSR (0 where none is stored)
 *   hx_sa_filter_*          Lf = strength-filtered L = sign*Re(K): |l_ij| >= theta sqrt(|l_ii l_jj|) kept, weak couplings
 *                           lumped on the diagonal; count pass (SYNC) fills out_rp (n+1) and *h_nnz, sd = n doubles kept
 *                           for the fill pass
 *   hx_sa_power_rho         rho(Df^-1 Lf) from `iters` >= 2 power iterations starting at v (v, w: n doubles, overwritten). SYNC
 *   hx_sa_prolongator_*     P = (I - omega Df^-1 Lf) P0, P0[i, agg(i)] = inv_sqrt_cnt[agg(i)]; sorted columns; count is SYNC
 *   hx_csr_transpose_d      out = A^T of a real CSR with sorted columns (stable radix sort of the column indices). SYNC
 *   hx_spgemm_*             C = A B row by row (hash-based symbolic pass + one radix sort for the sorted pattern, then a
 *                           numeric pass by binary search); values real (float64) or complex128 per operand, C complex128.
 *                           The Galerkin operator is R (K P): two products. */
int hx_csr_diagonal_z(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, hx_c128* diag, hx_stream s);
/* row-sharded set-up (owned rows over [owned | ghost] columns): hx_sa_strength_diag fills sd for the owned rows so that its
 * ghost entries can be exchanged before the filter passes; hx_sa_power_step is ONE step w = Df^-1 Lf v of the power
 * iteration with |w|^2 left on the device (d_scratch: 1024 doubles), the ghost entries of v being exchanged between steps */
int hx_sa_strength_diag(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign, double* sd,
                        hx_stream s);
int hx_sa_power_step(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, const double* v, double* w,
                     double* d_norm2, double* d_scratch, hx_stream s);
int hx_sa_filter_count(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign, double theta,
                       double* sd, int32_t* out_rp, int64_t* h_nnz, hx_stream s);
int hx_sa_filter_fill(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, double sign, double theta,
                      const double* sd, const int32_t* out_rp, int32_t* out_ci, double* out_v, hx_stream s);
int hx_sa_power_rho(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, double* v, double* w, int iters,
                    double* h_rho, hx_stream s);
int hx_sa_prolongator_count(int32_t n, const int32_t* rp, const int32_t* ci, const int32_t* agg, int32_t* out_rp,
                            int64_t* h_nnz, hx_stream s);
int hx_sa_prolongator_fill(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, const int32_t* agg,
                           const double* inv_sqrt_cnt, double omega, const int32_t* out_rp, int32_t* out_ci,
                           double* out_v, hx_stream s);
int hx_csr_transpose_d(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t* rp, const int32_t* ci, const double* v,
                       int32_t* out_rp, int32_t* out_ci, double* out_v, hx_stream s);
typedef struct hx_spgemm hx_spgemm;
/* SYNC.  out_rp: nA + 1 entries (device); the plan keeps the sorted pattern of C for the numeric pass. */
int hx_spgemm_symbolic(hx_spgemm** plan, int32_t nA, const int32_t* rpA, const int32_t* ciA, int32_t ncolsB,
                       const int32_t* rpB, const int32_t* ciB, int32_t* out_rp, int64_t* h_nnz, hx_stream s);
int hx_spgemm_numeric(const hx_spgemm* plan, const int32_t* rpA, const int32_t* ciA, const void* va, int a_complex,
                      const int32_t* rpB, const int32_t* ciB, const void* vb, int b_complex, int32_t* out_ci,
                      hx_c128* out_v, hx_stream s);
int hx_spgemm_destroy(hx_spgemm* plan);
/* ---- Shape-derivative boundary functional.
 * replaces: shape_derivatives.py:12-37 -- `assemble_scalar(form(inner(V_ffd, normal) * div(p_adj_conj * c**2 * grad(p_dir)) *
 * ds(tag)))`, the UFL / FFCx boundary integral of the reference's shapeDerivativesFFD.  Per boundary facet (triangle of a
 * tetrahedral mesh, listed with its adjacent cell) out[f] = int_f (V . n) [(c^2 grad q + 2 q c grad c) . grad p + q c^2 lap p] ds
 * with q = conj(p_adj); p, q: P1 (degree 1: vertex values) or P2 (degree 2: dofs as in cell_dofs, 4 vertices + 6 edges) dof
 * arrays, c: P1 nodal or cell-wise values, V: nodal displacement vectors (n x 3), quad: nq x 4 barycentric points + weights
 * (sum 1) of a rule exact to degree 5.  The caller sums out[] (fixed order). */
int hx_shape_derivative_facets(int degree, int32_t nf, const double* x, const int32_t* cells, const int32_t* cell_dofs,
                               const int32_t* facets, const int32_t* facet_cell, const hx_c128* p, const hx_c128* q,
                               const hx_c128* c, int c_is_cellwise, const double* V, int nq, const double* quad,
                               hx_c128* out, hx_stream s);
/* HOST set-up passes of the smoothed-aggregation hierarchy (graph / integer work, once per K):
 * hx_host_sa_filter: strength filter |l_ij| >= theta*sqrt(l_ii l_jj) with weak couplings lumped on the diagonal;
 *   L is the Laplacian-like real part (CSR), entry p = sign * h_lv[p * stride] (stride 2 = the real parts of an
 *   interleaved complex128 value array, read in place); output = filtered matrix in compact CSR.  Call with h_out_ci == NULL
 *   for the count pass (fills h_out_rp), then again to fill.
 * hx_host_aggregate: greedy aggregation on that graph (diagonal entries are ignored, |w| is the coupling strength).
 * hx_host_sa_prolongator: P = (I - omega Df^-1 Lf) P0 with P0[i, agg(i)] = inv_sqrt_cnt[agg(i)]; same two-call protocol. */
int hx_host_sa_filter(int32_t n, const int32_t* h_rp, const int32_t* h_ci, const double* h_lv, int stride, double sign,
                      double theta, int32_t* h_out_rp, int32_t* h_out_ci, double* h_out_v);
int hx_host_aggregate(int32_t n, const int32_t* h_rp, const int32_t* h_ci, const double* h_w,
                      int32_t* h_agg, int32_t* h_n_agg);
int hx_host_aggregate_f32(int32_t n, const int32_t* h_rp, const int32_t* h_ci, const float* h_w,
                          int32_t* h_agg, int32_t* h_n_agg);      /* float32 strengths (device set-up: half the D2H bytes) */
int hx_host_sa_prolongator(int32_t n, const int32_t* h_rp, const int32_t* h_ci, const double* h_lf,
                           const int32_t* h_agg, const double* h_inv_sqrt_cnt, double omega,
                           int32_t* h_out_rp, int32_t* h_out_ci, double* h_out_v);
/* diagonal of a complex CSR matrix (0 where none is stored); row-parallel */
int hx_host_csr_diagonal(int32_t n, const int32_t* rp, const int32_t* ci, const hx_c128* val, hx_c128* out);
/* rho(Df^-1 Lf) by `iters` power iterations from the start vector v (length n, overwritten); row-parallel. */
int hx_host_power_rho(int32_t n, const int32_t* rp, const int32_t* ci, const double* lf, double* v, int iters,
                      double* rho);
/* HOST, multi-threaded: Galerkin product Kc = R K P (R, P real CSR; K complex CSR) fused over coarse rows, no
 * intermediate K P.  Same two-call protocol (h_out_ci == NULL: count pass); columns sorted per row. */
int hx_host_galerkin_rap(int32_t n_c, const int32_t* h_Rrp, const int32_t* h_Rci, const double* h_Rv,
                         const int32_t* h_Krp, const int32_t* h_Kci, const hx_c128* h_Kv,
                         const int32_t* h_Prp, const int32_t* h_Pci, const double* h_Pv,
                         int32_t* h_out_rp, int32_t* h_out_ci, hx_c128* h_out_v, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* HX_H */
