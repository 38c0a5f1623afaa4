// Sparse direct solver of libhx (hx_splu_*): what the host analysis hands to the device factorisation.
//
// Supernodal multifrontal LU of a structurally symmetric(ised) complex128 matrix, nested-dissection ordered:
//   * a supernode s owns the pivot columns [first, first + ns) of the permuted matrix and a sorted list of nr border rows;
//     its front is the (ns + nr)^2 matrix [F11 F12; F21 F22] over (own columns ++ border);
//   * block elimination with 64 x 64 pivot blocks: F = L U with unit diagonal BLOCKS in L and the (updated) diagonal blocks D_k
//     themselves on the diagonal of U, so L[r, k] = F[r, k] D_k^-1 and U[k, j] = F[k, j] is never transformed.  The factor
//     keeps two column-major panels per supernode:
//       Lp[a, b] (m x ns, leading dimension m)  = L(front row a, pivot b)     for a >= b (strict upper triangle of the top block unused),
//       Up[b, a] (ns x m, leading dimension ns) = U(pivot b, front column a)  for a >  b (lower triangle + diagonal of its left block unused),
//     and D_k^-1 (Gauss-Jordan with partial pivoting inside the block), NB x NB column-major per pivot block, in a separate array;
//   * F22 (the contribution block) is temporary: it lives in an arena until the parent front has absorbed it.
#pragma once
#include <stdint.h>
#include <vector>

#define SPLU_NB 64          // pivot block: one LDS-resident diagonal block, one step of the panel recurrences

struct SpluSymbolic {
  int32_t n = 0, nsuper = 0, nlevels = 0;
  std::vector<int32_t> perm, iperm;          // perm[new] = old, iperm[old] = new
  std::vector<int32_t> sfirst;               // nsuper + 1: first pivot column of a supernode
  std::vector<int32_t> sparent;              // -1: a root
  std::vector<int64_t> sptr;                 // nsuper + 1: border rows of a supernode in srow / rel
  std::vector<int32_t> srow;                 // border rows (permuted numbering, ascending)
  std::vector<int32_t> rel;                  // parallel to srow: position of that row in the PARENT's front (0 .. m_parent)
  std::vector<int32_t> slevel;               // height above the leaves
  std::vector<int32_t> level_ptr, level_list;  // supernodes grouped by level
  std::vector<int32_t> child_ptr, child_list;  // children in ascending order (the order of the extend-add)
  std::vector<int64_t> loff, uoff;           // element offsets of Lp / Up in the factor arena
  std::vector<int64_t> ioff;                 // element offset of the diagonal-block inverses (NB * NB per pivot block)
  std::vector<int64_t> soff;                 // element offset of the contribution block in the temporary arena
  std::vector<int64_t> woff;                 // offset of the front's slice of the solve workspace (m + ns entries)
  int64_t factor_elems = 0, inverse_elems = 0, temp_elems = 0, work_elems = 0;
  int64_t nnz_L = 0;                         // entries of L incl. diagonal (= entries of U)
  double flops = 0.0;                        // real flops of the numeric factorisation (complex multiply-add = 8)
  int32_t max_front = 0, max_ns = 0;
  double analyse_seconds = 0.0;
  std::vector<int64_t> amap;                 // per stored entry of the input CSR: destination in the factor arena (>= 0: in an L panel;
                                             // < 0: -(destination + 1) in a U panel)
  std::vector<int32_t> mirror;               // symmetric pattern: per UPPER entry (i, j) the position of (j, i); empty otherwise
};

// splu_analyse.hip (host only)
// flags: bit 0 = the pattern is structurally symmetric with sorted rows (skips forming A + A^T)
int splu_analyse_host(SpluSymbolic& S, int32_t n, const int32_t* rowptr, const int32_t* col, const double* coords, int dim,
                      int leaf_size, int flags);
