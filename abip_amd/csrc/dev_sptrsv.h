// dev_sptrsv.h -- device side of the direct KKT back-end: x = P' L^-T D^-1 L^-1 P b
// (reference: _ldl_solve, linsys/direct.c:172-198 -> LDL_perm / lsolve / dsolve / ltsolve / permt,
//  external/ldl/ldl.c:357-550).  The factor is computed once on the host (host_setup.cpp); here both
// triangular solves run in GATHER form over level sets:
//   forward  (L z = b):  z_i = b_i - sum_{j<i} L_ij z_j      rows of L   (CSR)
//   backward (L' x = z): x_j = z_j - sum_{i>j} L_ij x_i      columns of L (CSC)
// A level is a set of rows whose dependencies are all in earlier levels.  Small systems run the sparse part of the
// solve in ONE 1024-thread workgroup (dev_ldl.h: levels separated by workgroup barriers, no launches in between);
// large systems launch wide levels as grids and runs of thin levels as single-workgroup segments.
#pragma once
#include "dev_common.h"

namespace abip {

constexpr int TBS = 1024; // threads of the single-workgroup triangular kernels

struct Tri { // entries in level order: position r of lev_rows owns [ptr[r], ptr[r+1])
  const int *ptr, *idx;
  const double *val;
  const int *lev_ptr, *lev_rows, *lev_g;
  int nlev;
};

// rows [a, b) of one level, `g` lanes per row (power of two <= 64), executed by `nthr` threads starting at `tid`
__device__ __forceinline__ void tri_level(const Tri &T, double *x, int a, int b, int g, int tid, int nthr) {
  const int ngrp = nthr / g, grp = tid / g, q = tid % g;
  for (int base = a; base < b; base += ngrp) {
    const int r = base + grp;
    double acc = 0.0;
    int row = -1;
    if (r < b) {
      row = T.lev_rows[r];
      const int e = T.ptr[r + 1];
      for (int k = T.ptr[r] + q; k < e; k += g) acc += T.val[k] * x[T.idx[k]];
    }
    for (int off = g >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row >= 0 && q == 0) x[row] -= acc;
  }
}

// segmented variant for systems too large for one workgroup
static __global__ __launch_bounds__(BS) void k_perm_in(const int *__restrict__ Pmap, const double *__restrict__ b, double *__restrict__ x, int N, const Ctl *ctl) {
  if (ctl->halt) return;
  for (int j = blockIdx.x * BS + threadIdx.x; j < N; j += gridDim.x * BS) x[j] = b[Pmap[j]];
}
static __global__ __launch_bounds__(BS) void k_perm_out(const int *__restrict__ Pmap, double *__restrict__ b, const double *__restrict__ x, int N, const Ctl *ctl) {
  if (ctl->halt) return;
  for (int j = blockIdx.x * BS + threadIdx.x; j < N; j += gridDim.x * BS) b[Pmap[j]] = x[j];
}
static __global__ __launch_bounds__(BS) void k_dscale(double *__restrict__ x, const double *__restrict__ D, int N, const Ctl *ctl) {
  if (ctl->halt) return;
  for (int j = blockIdx.x * BS + threadIdx.x; j < N; j += gridDim.x * BS) x[j] /= D[j];
}
// one wide level over the grid: the level's rows are contiguous positions of the level-ordered storage, i.e. a CSR matrix of
// its own, and x[row] -= (that matrix) * x is the CSR-stream SpMV of the PCG path (dev_common.h: coalesced value/index stream,
// products staged in LDS, rows reduced from LDS).  Gathers touch earlier levels only, every row is written by one lane.
static __global__ __launch_bounds__(BS) void k_tri_wide(Tri T, const int4 *__restrict__ rbd, int nrb, double *x, const Ctl *ctl) {
  if (ctl->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  const Csr M{T.ptr, T.idx, T.val, rbd, nrb, 0, nullptr, nullptr, nullptr, nullptr, 0}; // (no sliced image)
  spmv_stream<1>(M, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
                 [&](int pos, double(&acc)[1]) { const int row = T.lev_rows[pos]; x[row] -= acc[0]; });
}
} // namespace abip
