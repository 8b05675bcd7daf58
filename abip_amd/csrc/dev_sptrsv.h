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
// A wide level of the BACKWARD sweep with the dense tail's part of x RESIDENT IN LDS (round 6; the design of qcp_pcg.h kq_pcg_Aty_lds, which runs the same-shaped product
// A'y in 20 us where the CSR-stream kernel takes 30).  The rows of such a level are head columns of L; what they gather is mostly x2 -- the solution on the last T
// pivots (C5: all of it: L11 = I, every entry of a head column lies in L21) -- and T doubles fit the LDS of a CU up to T = 16 384.  With x2 in LDS those gathers never
// leave the CU: k_tri_wide runs at the rate at which a CU keeps L2 gathers in flight (one request per non-zero: DESIGN section 4), this kernel at the rate of the
// stream of L's entries.  One 1024-thread workgroup per CU; G lanes per row (16 for short rows, 64 for long ones), four loads in flight per lane for each of two rows,
// the next pair's extents requested before the current pair is reduced.  An entry outside the tail window (an earlier head level) is gathered from global memory.
// A row's products add up lane-strided, then by a G-lane butterfly: a fixed order.
template <int G>
static __global__ __launch_bounds__(1024) void k_tri_wide_lds(Tri T, int a, int b, double *x, int t0, int tl, const Ctl *ctl) {
  extern __shared__ double gl[];
  if (ctl->halt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < tl; i += 1024) gl[i] = x[t0 + i];
  __syncthreads();
  constexpr int GPW = 64 / G;
  const int gidx = ((int)blockIdx.x * 16 + wave) * GPW + lane / G, gl_ = lane % G, TG = (int)gridDim.x * 16 * GPW;
  auto extent = [&](int r, int &s, int &e) { s = 0; e = 0; if (r < b) { s = T.ptr[r]; e = T.ptr[r + 1]; } };
  auto gat = [&](int c) { const int k = c - t0; return (k >= 0 && k < tl) ? gl[k] : x[c]; };
  auto finish = [&](int r, double acc) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, G);
    if (gl_ == 0 && r < b) { const int row = T.lev_rows[r]; x[row] -= acc; }
  };
  int r = a + gidx, sa, ea, sb, eb;
  extent(r, sa, ea); extent(r + TG, sb, eb);
  while (r < b) {
    int na, ma, nb2, mb;
    extent(r + 2 * TG, na, ma); extent(r + 3 * TG, nb2, mb);
    double acca = 0.0, accb = 0.0;
    for (int qa = sa + gl_, qb = sb + gl_; qa < ea || qb < eb; qa += 4 * G, qb += 4 * G) {
      double va[4], vb[4]; int ca[4], cb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q1 = qa + k * G, q2 = qb + k * G;
        const bool o1 = q1 < ea, o2 = q2 < eb;
        va[k] = o1 ? T.val[q1] : 0.0; ca[k] = o1 ? T.idx[q1] : t0;
        vb[k] = o2 ? T.val[q2] : 0.0; cb[k] = o2 ? T.idx[q2] : t0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { acca += va[k] * gat(ca[k]); accb += vb[k] * gat(cb[k]); }
    }
    finish(r, acca); finish(r + TG, accb);
    r += 2 * TG; sa = na; ea = ma; sb = nb2; eb = mb;
  }
}
} // namespace abip
