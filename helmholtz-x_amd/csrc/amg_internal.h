// Private structures of the AMG handle, shared by amg.hip (V-cycle, flexible GMRES) and cocg.hip (deflated COCG).
#pragma once
#include <complex>
#include <vector>

#include "hx_common.h"

struct hx_sell;

// T = double: complex128 hierarchy; T = float: complex64 copy of it (values interleaved re, im: 2 T per entry).
// va: interleaved complex values (2 T per entry), or -- real_vals -- one T per entry (transfer operators are real)
template <typename T> struct CsrT { int32_t rows = 0; int64_t nnz = 0; const int32_t* rp = nullptr; const int32_t* ci = nullptr; const T* va = nullptr; bool real_vals = false; };
template <typename T> struct LevelT {
  int32_t n = 0;
  CsrT<T> K, P, R;             // P: n x n_coarse, R = P^T: n_coarse x n   (absent on the coarsest level)
  const T* dinv = nullptr;
  double rho = 0.0;            // estimate of rho(D^-1 K) for the Chebyshev smoother (0: damped Jacobi with h->omega)
  const hx_sell* sell = nullptr;   // optional SELL-64 companion of K (fine level): used for the K products
  T *x = nullptr, *x2 = nullptr, *x3 = nullptr, *b = nullptr, *r = nullptr;   // owned work vectors
  int32_t n_ext = 0;           // row-sharded level: owned + ghost length of x, x2, x3, r (0: not sharded)
  bool p_ghost_rows = false;   // row-sharded level: P carries n_ext rows (the owners' rows of the ghost vertices behind the owned ones)
  // deep-halo mode of a row-sharded level (hx_amg_set_level_deep): the vectors carry TWO ghost layers [owned | g1 | g2] (n_ext counts
  // both), K1 holds the operator rows of the first layer (n_g1 rows over the whole extended numbering), dinv covers n_ext entries
  CsrT<T> K1;                  // rows of layer 1 first (n_g1 rows, nnz_g1 entries), then -- three layers -- those of layer 2
  int32_t n_g1 = 0;
  int64_t nnz_g1 = 0;
  int deep_layers = 0;         // 2: vectors [owned | g1 | g2], K1 = rows of g1;  3: [owned | g1 | g2 | g3], K1 = rows of g1 and g2
  bool deep = false;
};
template <typename T> struct HierT { std::vector<LevelT<T>> L; CsrT<T> coarse_inv; };

typedef std::complex<double> cd;
// Recycled pairs of the GCRO variant of hx_gmres (K U = C, C^H C = I) and the panel that holds [C; V].
struct Recycle {
  int kmax = 0, k = 0, restart = 0, m_last = 0;
  int64_t n = 0;
  hx_c128* U = nullptr;        // kmax x n
  hx_c128* CV = nullptr;       // (kmax + restart + 1) x n : rows [0, k) = C, then the Arnoldi vectors
  float* CVf = nullptr;        // complex64 copy of CV, read by the inner-product passes (hx_zpanel_dot_c64)
  // hx_cocg deflates with the pairs (Ut, Cd): Ut = U ROUNDED to complex64 (kept in that format: exactly representable),
  // Cd = K Ut recomputed in complex128 -- an exactly consistent pair, which the K-conjugacy of the recurrence needs
  float* Uf = nullptr;         // Ut, kmax x n complex64
  hx_c128* Cd = nullptr;       // K Ut, kmax x n complex128 (built on first use)
  hx_c128* Einv = nullptr;     // device copy of (Ut^T K Ut)^-1, k x k row-major (built on first use)
  bool have_E = false;
  std::vector<cd> Hraw;        // raw Hessenberg of the last first cycle, column-major, ld = restart + 1
};
// Row-sharded mode (one process per GPU): the first `n_sharded` levels hold this rank's rows over [owned | ghost] columns,
// every coarser level is replicated.  The two exchange steps go through callbacks (hx_comm_halo / hx_comm_allreduce:
// RCCL issued by the library on the kernels' stream; or the caller's own), see include/hx.h.
struct hx_amg {
  HierT<double> D;
  HierT<float> S;
  bool single = false;         // run the V-cycle on the complex64 copy
  bool real_transfers = false; // hx_amg_set_level* receive P / R values as REAL arrays
  int pre = 2, post = 2;
  double omega = 0.6;
  bool sharded = false;
  hx_halo_fn halo = nullptr;
  hx_allreduce_fn allreduce = nullptr;
  hx_halo_begin_fn halo_begin = nullptr;      // optional split form: the exchange overlaps the interior slices of the product
  hx_halo_end_fn halo_end = nullptr;
  void* ctx = nullptr;
  int n_sharded = 0;           // number of leading row-sharded levels
  Recycle rec;
  // hx_cocg: ||K||_inf of the operator last seen (by value pointer) and the backward error of the last solve that ended
  // above its tolerance (-1: not measured)
  double normK = 0.0;
  const void* normK_val = nullptr;
  double last_backward = -1.0;
  bool x0_next = false;        // the next hx_cocg call takes the contents of x as its initial guess (hx_amg_next_solve_uses_x)
  bool x1_ready0 = false;      // the producer of the next cycle's input (k_cocg_update) also wrote level 0's first smoothing step
  // hipGraph of the complex64 V-cycle (all buffers of the cycle are owned by the handle, so the ~45 launches of one cycle
  // are captured once and replayed): for small and mid-size problems, where the cycle is launch-bound
  int graph_state = -1;        // -1 off, 0 requested (capture at the next cycle), 1 ready
  hipGraphExec_t vexec = nullptr;
  hipStream_t gstream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  // hx_cocg (single GPU): two pinned landing areas for the per-block partials of |r|^2 and the events that mark their arrival,
  // so that the host can read the norm of iteration j while iteration j + 1 is already queued
  void* cg_pin = nullptr;
  hipEvent_t cg_ev[2] = {nullptr, nullptr};
  // work vectors of the BLOCK V-cycle (block.hip / hx_amg_vcycle_blk: hx_block_width() interleaved complex64 vectors per level),
  // allocated at the first block cycle
  struct BlkBuf { float *x = nullptr, *x2 = nullptr, *x3 = nullptr, *b = nullptr, *r = nullptr; };
  std::vector<BlkBuf> blk;
};
