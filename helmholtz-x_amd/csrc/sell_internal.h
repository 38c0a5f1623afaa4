// SELL-64 handle and the per-entry helpers shared by sell.hip (one right-hand side) and block.hip (blocks of right-hand sides).
#pragma once
#include <hip/hip_fp16.h>

#include "hx_common.h"

#define SELL_C 64

struct hx_sell {
  int32_t n = 0, n_slices = 0;
  int32_t n_cols = 0;                // largest column + 1 (the operator may be rectangular: row-sharded local rows over
                                     // [owned | ghost] columns); bound of the clamped columns of stencil slices
  int64_t padded = 0;                // stored entries (multiple of 64)
  int64_t* slice_ptr = nullptr;      // n_slices + 1 offsets (in entries)
  int32_t* col = nullptr;
  // 16-bit column DELTAS (column - row), used slice by slice: a slice whose stored entries all satisfy
  // |column - row| <= 32767 (banded numberings: structured grids, RCM-ordered meshes) streams 2 instead of 4 index bytes
  // per entry; the others (e.g. the rows that close a periodic ring) keep the 32-bit columns (`wide[slice]` = 1).
  int16_t* dcol = nullptr;
  uint8_t* wide = nullptr;
  int32_t n_wide = 0;
  zc* val_z = nullptr;               // complex128 values (may be null)
  cc* val_c = nullptr;               // complex64 values (may be null)
  // complex-half STORAGE of the values for the complex64 kernels (8 instead of 12 bytes per entry with the int32
  // column): value = h_scale * half2.  Only for the V-cycle, a preconditioner inside flexible GMRES -- measured:
  // identical iteration counts with fp16-rounded fine-level values (bf16: +3..9 %).  May be null.
  __half2* val_h = nullptr;
  float h_scale = 1.0f;
  // STENCIL slices: when the 64 rows of a slice draw their columns from one common list of deltas (column - row) --
  // structured and extruded meshes numbered along a grid line -- the slice stores that list once (st_delta[st_ptr[s] ..
  // st_ptr[s+1])) and NO per-entry index: entry k of a row belongs to column row + delta_k, rows that lack it hold a
  // zero.  Decided slice by slice (union of the rows' delta lists at most 1/8 longer than the longest row); the others
  // keep their 16- / 32-bit indices.  Lossless, like the 16-bit deltas.
  int32_t* st_ptr = nullptr;         // n_slices + 1, or null when no slice qualified
  int32_t* st_delta = nullptr;
  int32_t n_stencil = 0;
  // Row-sharded local operator (more columns than rows: [owned | ghost]): the slices that read NO ghost column ("interior")
  // and the others ("boundary"), as two slice lists.  A product can then run its interior slices while the halo exchange
  // of x is still in flight on another stream and its boundary slices after it (hx_sell_*_part).  Null for square operators.
  int32_t* list_interior = nullptr;
  int32_t* list_boundary = nullptr;
  int32_t n_interior = 0, n_boundary = 0;
  // OVERFLOW TAIL: on an unstructured mesh the longest of 64 rows sets the slice's width (24 % padding on tetrahedra).  A slice
  // may instead keep a BODY of `width` entries per row (a multiple of the kernels' group of 8) and store what its longer rows
  // have beyond that packed, row after row, in the tail arrays: entries tptr[s] .. tptr[s+1] belong to slice s, row r of the
  // slice owns (trow >> 6) .. + (trow & 63) of them.  The kernel that owns the slice adds them right after the body (each lane
  // walks its own few entries): no second launch, no second pass over y.  Null when no slice was split (structured meshes).
  int32_t* tptr = nullptr;           // n_slices + 1
  uint16_t* trow = nullptr;          // 64 per slice
  int32_t* tcol = nullptr;           // absolute columns
  zc* tval_z = nullptr;
  cc* tval_c = nullptr;
  __half2* tval_h = nullptr;         // (same scale as val_h)
  int64_t n_tail = 0;
  int32_t n_split = 0;               // slices with a tail
  zc* dot_ws = nullptr;              // one partial sum per slice (hx_sell_spmv_z_dot; allocated on first use)
};
typedef __half2 ch;
template <typename MV> struct SellTail {   // kernel argument (all null: no tails)
  const int32_t* tptr; const uint16_t* trow; const int32_t* tcol; const MV* tval;
};
template <typename MV> inline SellTail<MV> sell_tail_of(const hx_sell* h);
template <> inline SellTail<zc> sell_tail_of<zc>(const hx_sell* h) { return {h->tval_z ? h->tptr : nullptr, h->trow, h->tcol, h->tval_z}; }
template <> inline SellTail<cc> sell_tail_of<cc>(const hx_sell* h) { return {h->tval_c ? h->tptr : nullptr, h->trow, h->tcol, h->tval_c}; }
template <> inline SellTail<__half2> sell_tail_of<__half2>(const hx_sell* h) { return {h->tval_h ? h->tptr : nullptr, h->trow, h->tcol, h->tval_h}; }
__device__ __forceinline__ zc mload(const zc* p) { return *p; }
__device__ __forceinline__ cc mload(const cc* p) { return *p; }
__device__ __forceinline__ cc mload(const ch* p) { return __half22float2(*p); }

