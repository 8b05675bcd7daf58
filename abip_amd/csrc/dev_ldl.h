// dev_ldl.h -- the direct KKT back-end on the device: sparse head + dense tail.
//
//   P K P' = L D L',   L = [[L11, 0], [L21, L22]]      (reference: LDL_numeric / _ldl_solve, linsys/direct.c:172-270)
//
// The host (host_setup.cpp: factor_upper) factors the sparse head (L11, L21, D1) with the reference's up-looking algorithm and
// hands over S = K22 - L21 D1 L21', the Schur complement onto the last T pivots.  With a fill-reducing ordering that trailing
// block is (nearly) dense and, level-scheduled, would be T one-row levels -- the sequential part of the solve.  Here S is
// factored densely on the device (blocked right-looking LDL', 64x64 blocks, no pivoting: K is quasi-definite), W = inv(L22) is
// formed once, and every solve applies the tail as two dense triangular mat-vecs:
//
//   forward :  levels over [L11; L21]           (gather form, dev_sptrsv.h)   z1, w = b2 - L21 z1
//              t  = D2^-1 (W w)                 k_tail_mv, one wavefront per row, HBM/Infinity-Cache stream of W
//   backward:  x2 = W' t                        k_tail_mv on the stored transpose
//              levels over the head columns     x1 = D1^-1 z1 - L11'.. - L21' x2
//   large tails (T >= 2048): x2 = M w with M = W' D2^-1 W formed once, lower triangle streamed ONCE per solve (k_tail_sym; half the bytes of the two mat-vecs)
//
// Set-up kernels: the two that carry the flops (the trailing update of the dense LDL' and the Schur complement's rank-k update) run on the matrix cores
// (v_mfma_f64_16x16x4_f64, 64 x 64 tiles through LDS); the rest is LDS-tiled fp64 FMA code.
#pragma once
#include "host_par.h"
#include <algorithm>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdlib>

#include "dev_host_util.h"
#include "dev_sptrsv.h"
#include "dev_tail.h"
#include "host_setup.h"

namespace abip {

constexpr int DB = 64;  // dense block edge
constexpr int DH = 32;  // K-slice staged through LDS per step

// acc[a][b] += sum_q A[ty*4+a][q] * (TRANSB ? B[tx*4+b][q] : B[q][tx*4+b]),  q in [0, DH)
template <bool TRANSB>
__device__ __forceinline__ void tile_mac(double (&acc)[4][4], const double (*As)[DH + 1], const double *Bs, int ty, int tx) {
#pragma unroll 4
  for (int q = 0; q < DH; ++q) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = As[ty * 4 + i][q];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = TRANSB ? Bs[(tx * 4 + i) * (DH + 1) + q] : Bs[q * (DB + 1) + tx * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
  }
}

// 64 x DH slice (columns [q0, q0+DH)) of a row-major 64x64 tile -> LDS [64][DH+1]
__device__ __forceinline__ void load_rows_slice(double (*dst)[DH + 1], const double *src, long ld, int q0, int tid) {
  for (int e = tid; e < DB * DH; e += 256) { const int r = e / DH, q = e % DH; dst[r][q] = src[(long)r * ld + q0 + q]; }
}
// DH x 64 slice (rows [q0, q0+DH)) -> LDS [DH][DB+1]
__device__ __forceinline__ void load_cols_slice(double *dst, const double *src, long ld, int q0, int tid) {
  for (int e = tid; e < DH * DB; e += 256) { const int q = e / DB, c = e % DB; dst[q * (DB + 1) + c] = src[(long)(q0 + q) * ld + c]; }
}

// (1) factor the diagonal block at (k0, k0) in place; pivots -> Dt[k0..], inverse of its unit-lower factor -> Linv (64x64)
static __global__ __launch_bounds__(256) void k_dldl_diag(double *S, int ld, int k0, double *Dt, double *Linv, int *fail) {
  __shared__ double a[DB][DB + 1];
  const int tid = threadIdx.x, r = tid >> 2, c4 = tid & 3;
  for (int e = tid; e < DB * DB; e += 256) a[e / DB][e % DB] = S[(long)(k0 + e / DB) * ld + k0 + e % DB];
  // right-looking; column c keeps l_rc d_c until the end, so no entry is read and written in the same step
  for (int c = 0; c < DB - 1; ++c) {
    __syncthreads();
    if (r > c) {
      const double lr = a[r][c] / a[c][c];
      for (int cc = c + 1 + ((c4 - (c + 1)) & 3); cc <= r; cc += 4) a[r][cc] -= lr * a[cc][c];
    }
  }
  __syncthreads();
  if (tid < DB) { const double d = a[tid][tid]; Dt[k0 + tid] = d; if (d == 0.0 || !isfinite(d)) *fail = 1; }
  for (int e = tid; e < DB * DB; e += 256) { const int i = e / DB, j = e % DB; if (i > j) a[i][j] /= a[j][j]; }
  __syncthreads();
  for (int e = tid; e < DB * DB; e += 256) { const int i = e / DB, j = e % DB; if (i > j) S[(long)(k0 + i) * ld + k0 + j] = a[i][j]; }
  // column j of inv(L): x_j = 1, x_i = -sum_{c=j}^{i-1} L_ic x_c, kept in the unused upper triangle as a[j][i]
  if (tid < DB) {
    const int j = tid;
    for (int i = j + 1; i < DB; ++i) { double s = a[i][j]; for (int c = j + 1; c < i; ++c) s += a[i][c] * a[j][c]; a[j][i] = -s; }
  }
  __syncthreads();
  for (int e = tid; e < DB * DB; e += 256) { const int i = e / DB, j = e % DB; Linv[e] = i > j ? a[j][i] : (i == j ? 1.0 : 0.0); }
}

// (2) panel below the diagonal block: P = S[i, k] inv(Lkk)' (= L[i,k] Dk) -> LD; L[i,k] = P / Dk -> S
static __global__ __launch_bounds__(256) void k_dldl_panel(double *S, int ld, int k0, const double *Dt, const double *Linv, double *LD) {
  __shared__ double As[DB][DH + 1], Bs[DB * (DH + 1)];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const long i0 = k0 + DB + (long)blockIdx.x * DB;
  double acc[4][4] = {};
  for (int q0 = 0; q0 < DB; q0 += DH) {
    __syncthreads();
    load_rows_slice(As, S + i0 * ld + k0, ld, q0, tid);
    load_rows_slice((double(*)[DH + 1])Bs, Linv, DB, q0, tid); // Linv[c][q], used transposed
    __syncthreads();
    tile_mac<true>(acc, As, Bs, ty, tx);
  }
  __syncthreads();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const int r = ty * 4 + i, c = tx * 4 + j;
      LD[(i0 + r) * DB + c] = acc[i][j];
      S[(i0 + r) * ld + k0 + c] = acc[i][j] / Dt[k0 + c];
    }
}

// (3) trailing update: S[i, j] -= LD[i] * L[j, k]'   for tile pairs i >= j > k
static __global__ __launch_bounds__(256) void k_dldl_update(double *S, int ld, int k0, const double *LD) {
  __shared__ double As[DB][DH + 1], Bs[DB * (DH + 1)];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
  while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const long i0 = k0 + DB + (long)bi * DB, j0 = k0 + DB + (long)bj * DB;
  double acc[4][4] = {};
  for (int q0 = 0; q0 < DB; q0 += DH) {
    __syncthreads();
    load_rows_slice(As, LD + i0 * DB, DB, q0, tid);
    load_rows_slice((double(*)[DH + 1])Bs, S + j0 * ld + k0, ld, q0, tid);
    __syncthreads();
    tile_mac<true>(acc, As, Bs, ty, tx);
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) S[(i0 + ty * 4 + i) * ld + j0 + tx * 4 + j] -= acc[i][j];
}

// ---- the same tile products on the matrix cores (v_mfma_f64_16x16x4_f64) -----------------------------------------------------------
// C (64 x 64) += As (64 x DH) Bs' (64 x DH): wavefront w of the 256-thread workgroup owns the 32 x 32 quadrant (w >> 1, w & 1) as 2 x 2 blocks of 16 x 16.
// Operand maps (MI355X_MICROARCH.md): lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; it receives D[(l >> 4) + 4 r][l & 15], r = 0..3.
// LDS rows are padded to DHP = DH + 4 doubles: (l & 15) * 36 + (l >> 4) hits every 8-byte bank pair twice -- the minimum for a 512-byte wavefront read
// (stride DH + 1, right for the scalar tiles above, would stack four lanes on one bank here).
constexpr int DHP = DH + 4;
typedef double v4d __attribute__((ext_vector_type(4)));
struct MfmaTile { v4d c[2][2]; };
__device__ __forceinline__ void mfma_zero(MfmaTile &t) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) t.c[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
}
__device__ __forceinline__ void load_rows_slice_p(double (*dst)[DHP], const double *src, long ld, int q0, int tid) {
  for (int e = tid; e < DB * DH; e += 256) { const int r = e / DH, q = e % DH; dst[r][q] = src[(long)r * ld + q0 + q]; }
}
__device__ __forceinline__ void tile_mac_mfma(MfmaTile &t, const double (*As)[DHP], const double (*Bs)[DHP], int wave, int lane) {
  const int r0 = (wave >> 1) * 32 + (lane & 15), c0 = (wave & 1) * 32 + (lane & 15), kq = lane >> 4;
#pragma unroll
  for (int k4 = 0; k4 < DH; k4 += 4) {
    const double a0 = As[r0][k4 + kq], a1 = As[r0 + 16][k4 + kq], b0 = Bs[c0][k4 + kq], b1 = Bs[c0 + 16][k4 + kq];
    t.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, t.c[0][0], 0, 0, 0);
    t.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, t.c[0][1], 0, 0, 0);
    t.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, t.c[1][0], 0, 0, 0);
    t.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t.c[1][1], 0, 0, 0);
  }
}
// visit the 16 results of this lane: f(row, col, value) with (row, col) inside the 64 x 64 tile
template <class F>
__device__ __forceinline__ void mfma_foreach(const MfmaTile &t, int wave, int lane, F f) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) f((wave >> 1) * 32 + 16 * a + (lane >> 4) + 4 * r, (wave & 1) * 32 + 16 * b + (lane & 15), t.c[a][b][r]);
}
// (3m) trailing update on the matrix cores: S[i, j] -= LD[i] * L[j, k]'   for tile pairs i >= j > k
static __global__ __launch_bounds__(256) void k_dldl_update_mfma(double *S, int ld, int k0, const double *LD) {
  __shared__ double As[DB][DHP], Bs[DB][DHP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
  while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const long i0 = k0 + DB + (long)bi * DB, j0 = k0 + DB + (long)bj * DB;
  MfmaTile t; mfma_zero(t);
  for (int q0 = 0; q0 < DB; q0 += DH) {
    __syncthreads();
    load_rows_slice_p(As, LD + i0 * DB, DB, q0, tid);
    load_rows_slice_p(Bs, S + j0 * ld + k0, ld, q0, tid);
    __syncthreads();
    tile_mac_mfma(t, As, Bs, wave, lane);
  }
  mfma_foreach(t, wave, lane, [&](int r, int c, double v) { S[(i0 + r) * ld + j0 + c] -= v; });
}
// (0m) rank-kc update of the lower tile pairs on the matrix cores: S[i, j] -= PD[i, :] P[j, :]'
static __global__ __launch_bounds__(256) void k_schur_sub_mfma(double *S, int ld, const double *__restrict__ PD, const double *__restrict__ P, int ldp, int kc) {
  __shared__ double As[DB][DHP], Bs[DB][DHP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
  while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const long i0 = (long)bi * DB, j0 = (long)bj * DB;
  MfmaTile t; mfma_zero(t);
  for (int q0 = 0; q0 < kc; q0 += DH) {
    __syncthreads();
    load_rows_slice_p(As, PD + i0 * ldp, ldp, q0, tid);
    load_rows_slice_p(Bs, P + j0 * ldp, ldp, q0, tid);
    __syncthreads();
    tile_mac_mfma(t, As, Bs, wave, lane);
  }
  mfma_foreach(t, wave, lane, [&](int r, int c, double v) { S[(i0 + r) * ld + j0 + c] -= v; });
}

// (0) when the host leaves the product to the device (LdlHost::dev_schur): S -= L21 D1 L21' by dense panels of L21.
// Panel fill: P[r][c - c0] = L21[r, c] and PD = P D1 for the head columns [c0, c0 + kc); one wavefront per column (the panels are zeroed first).
static __global__ __launch_bounds__(256) void k_l21_panel(const long *__restrict__ ptr, const int *__restrict__ row, const double *__restrict__ val,
                                                          const double *__restrict__ Dh, int c0, int kc, int ldp, double *__restrict__ P, double *__restrict__ PD) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= kc) return;
  const int c = c0 + col;
  const double d = Dh[c];
  for (long q = ptr[c] + lane; q < ptr[c + 1]; q += 64) { const long o = (long)row[q] * ldp + col; const double v = val[q]; P[o] = v; PD[o] = v * d; }
}
// K22 from its triplets into the zeroed T x T block (each entry appears once)
static __global__ __launch_bounds__(256) void k_scatter_k22(const int *__restrict__ row, const int *__restrict__ col, const double *__restrict__ val, int n, double *__restrict__ S, int T) {
  for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) S[(long)row[q] * T + col[q]] = val[q];
}
// The same product for a SPARSE L21, row by row: one wavefront owns tail row r, keeps S[r, 0..r] -= ... in an LDS accumulator of T doubles and walks the
// row's entries (c, l_rc) in order; for each it adds l_rc D_c l_r'c for the entries r' <= r of column c (a prefix of the column's list: rows ascend), one lane
// per entry -- no two lanes of a step touch the same r', no atomics, a fixed order: deterministic.  cpos[e] = position of row entry e in the column lists.
static __global__ __launch_bounds__(64) void k_schur_rows(double *__restrict__ S, int T, const long *__restrict__ rptr, const int *__restrict__ rcol, const long *__restrict__ cpos,
                                                          const long *__restrict__ cptr, const int *__restrict__ crow, const double *__restrict__ cval, const double *__restrict__ Dh) {
  extern __shared__ double acc[];
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int q = lane; q <= r; q += 64) acc[q] = 0.0;
  __syncthreads();
  for (long e = rptr[r]; e < rptr[r + 1]; ++e) {
    const int c = rcol[e];
    const long last = cpos[e], first = cptr[c];
    const double f = cval[last] * Dh[c];
    for (long q = first + lane; q <= last; q += 64) acc[crow[q]] += f * cval[q];
    __syncthreads(); // (one wavefront: orders this column's LDS updates before the next column's)
  }
  double *Srow = S + (long)r * T;
  for (int q = lane; q <= r; q += 64) Srow[q] -= acc[q];
}
// rank-kc update of the lower tile pairs: S[i, j] -= PD[i, :] P[j, :]'
static __global__ __launch_bounds__(256) void k_schur_sub(double *S, int ld, const double *__restrict__ PD, const double *__restrict__ P, int ldp, int kc) {
  __shared__ double As[DB][DH + 1], Bs[DB * (DH + 1)];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bi + 1) * (bi + 2) / 2 <= (long)blockIdx.x) ++bi;
  while ((long)bi * (bi + 1) / 2 > (long)blockIdx.x) --bi;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const long i0 = (long)bi * DB, j0 = (long)bj * DB;
  double acc[4][4] = {};
  for (int q0 = 0; q0 < kc; q0 += DH) {
    __syncthreads();
    load_rows_slice(As, PD + i0 * ldp, ldp, q0, tid);
    load_rows_slice((double(*)[DH + 1])Bs, P + j0 * ldp, ldp, q0, tid);
    __syncthreads();
    tile_mac<true>(acc, As, Bs, ty, tx);
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) S[(i0 + ty * 4 + i) * ld + j0 + tx * 4 + j] -= acc[i][j];
}

// (4) block row kb of W = inv(L22):  W[kb, j] = -inv(L_kb,kb) * sum_{i=j}^{kb-1} L[kb, i] W[i, j],  W[kb, kb] = inv(L_kb,kb)
static __global__ __launch_bounds__(256) void k_dtri_inv_row(const double *L, int ld, int kb, const double *Linv_all, double *W) {
  constexpr int NA = DB * (DH + 1), NC = DB * (DB + 1);
  __shared__ double sm[NA + NC];
  double(*As)[DH + 1] = (double(*)[DH + 1])sm;
  double *Bs = sm + NA;                              // accumulation phase: DH x (DB+1) slice of W[i, j]
  double(*Cs)[DB + 1] = (double(*)[DB + 1])(sm + NA); // afterwards: the accumulated 64x64 product
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int j = blockIdx.x;
  const double *Lk = Linv_all + (long)kb * DB * DB;
  double *out = W + (long)kb * DB * ld + (long)j * DB;
  if (j == kb) { for (int e = tid; e < DB * DB; e += 256) out[(long)(e / DB) * ld + e % DB] = Lk[e]; return; }
  double acc[4][4] = {};
  for (int i = j; i < kb; ++i) {
    const double *A = L + (long)kb * DB * ld + (long)i * DB;       // L[kb, i]
    const double *B = W + (long)i * DB * ld + (long)j * DB;        // W[i, j] (block rows < kb are final)
    for (int q0 = 0; q0 < DB; q0 += DH) {
      __syncthreads();
      load_rows_slice(As, A, ld, q0, tid);
      load_cols_slice(Bs, B, ld, q0, tid);
      __syncthreads();
      tile_mac<false>(acc, As, Bs, ty, tx);
    }
  }
  __syncthreads();
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) Cs[ty * 4 + a][tx * 4 + b] = acc[a][b];
  double res[4][4] = {};
  for (int q0 = 0; q0 < DB; q0 += DH) {
    __syncthreads();
    load_rows_slice(As, Lk, DB, q0, tid);
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < DH; ++q) {
      double a[4], b[4];
      for (int x = 0; x < 4; ++x) { a[x] = As[ty * 4 + x][q]; b[x] = Cs[q0 + q][tx * 4 + x]; }
      for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) res[x][y] += a[x] * b[y];
    }
  }
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) out[(long)(ty * 4 + a) * ld + tx * 4 + b] = -res[a][b];
}

// (4m) the inverse by halves on the matrix cores: inv [[L11, 0], [L21, L22]] = [[W11, 0], [-W22 L21 W11, W22]] -- the two halves are independent, the corner is two
// products, and every level of the recursion is a handful of large launches instead of one small launch per block row (C5, T = 10 048: 0.19 s -> see profiles/r02z).
// C = alpha A B on row-major blocks whose dimensions are multiples of 64: C is M x N (grid N/64 x M/64), A is M x K, B is K x N.  triA: A is lower-triangular
// (K == M), row block bi stops at k = 64 (bi + 1); triB: B is lower-triangular (K == N), column block bj starts at k = 64 bj.
// triA == 2: A is UPPER-triangular (row block bi starts at k = 64 bi) and only the blocks bj <= bi of C are formed; bscale: row k of B is divided by bscale[k].
static __global__ __launch_bounds__(256) void k_dgemm_mfma(double *__restrict__ C, long ldc, const double *__restrict__ A, long lda, const double *__restrict__ B, long ldb, int K,
                                                           double alpha, int triA, int triB, const double *__restrict__ bscale) {
  __shared__ double As[DB][DHP], Bs[DH][DB + 8]; // B slice k-major: lane l reads Bs[k4 + (l >> 4)][c0 + (l & 15)]: (l >> 4) * 72 + (l & 15) -> every bank pair twice
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (triA == 2 && bj > bi) return;
  const int k0 = triA == 2 ? bi * DB : (triB ? bj * DB : 0), k1 = triA == 1 ? (K < (bi + 1) * DB ? K : (bi + 1) * DB) : K;
  const double *Ab = A + (long)bi * DB * lda, *Bb = B + (long)bj * DB;
  const int r0 = (wave >> 1) * 32 + (lane & 15), c0 = (wave & 1) * 32 + (lane & 15), kq = lane >> 4;
  MfmaTile t; mfma_zero(t);
  for (int q0 = k0; q0 < k1; q0 += DH) {
    __syncthreads();
    load_rows_slice_p(As, Ab, lda, q0, tid);
    for (int e = tid; e < DH * DB; e += 256) { const int q = e / DB, c = e % DB; const double bv = Bb[(long)(q0 + q) * ldb + c]; Bs[q][c] = bscale ? bv / bscale[q0 + q] : bv; }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < DH; k4 += 4) {
      const double a0 = As[r0][k4 + kq], a1 = As[r0 + 16][k4 + kq], b0 = Bs[k4 + kq][c0], b1 = Bs[k4 + kq][c0 + 16];
      t.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, t.c[0][0], 0, 0, 0);
      t.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, t.c[0][1], 0, 0, 0);
      t.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, t.c[1][0], 0, 0, 0);
      t.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t.c[1][1], 0, 0, 0);
    }
  }
  double *Cb = C + (long)bi * DB * ldc + (long)bj * DB;
  mfma_foreach(t, wave, lane, [&](int r, int c, double v) { Cb[(long)r * ldc + c] = alpha * v; });
}
// the diagonal blocks of W: inv(L_kk) as k_dldl_diag left them
static __global__ __launch_bounds__(256) void k_dtri_inv_diag(const double *__restrict__ Linv_all, double *__restrict__ W, int ld) {
  const int kb = blockIdx.x;
  const double *Lk = Linv_all + (long)kb * DB * DB;
  double *out = W + (long)kb * DB * ld + (long)kb * DB;
  for (int e = threadIdx.x; e < DB * DB; e += 256) out[(long)(e / DB) * ld + e % DB] = Lk[e];
}

// (5) Wt = W' (lower triangle of W -> upper triangle of Wt), 64x64 tiles
static __global__ __launch_bounds__(256) void k_dtranspose_lower(const double *W, double *Wt, int ld) {
  __shared__ double t[DB][DB + 1];
  const int bi = blockIdx.y, bj = blockIdx.x, tid = threadIdx.x;
  if (bj > bi) return;
  for (int e = tid; e < DB * DB; e += 256) t[e / DB][e % DB] = W[((long)bi * DB + e / DB) * ld + (long)bj * DB + e % DB];
  __syncthreads();
  for (int e = tid; e < DB * DB; e += 256) Wt[((long)bj * DB + e / DB) * ld + (long)bi * DB + e % DB] = t[e % DB][e / DB];
}

// ---- solve time ---------------------------------------------------------------------------------------------------------
// out[r] = (sum_c M[r, c] v[c]) / (dsc ? dsc[r] : 1),  c in [0, r] (lower) or [r, T) (upper); one wavefront per row,
// rows dealt round-robin so that the triangle's long and short rows mix in every workgroup
// (xh, Dh, nh): optionally also xh[j] /= Dh[j] for j < nh -- the head's D^-1 of the segmented path rides on the second mat-vec
static __global__ __launch_bounds__(BS) void k_tail_mv(const double *__restrict__ M, int ld, int T, int upper, const double *__restrict__ v,
                                                       double *__restrict__ out, const double *__restrict__ dsc, const Ctl *ctl,
                                                       double *__restrict__ xh, const double *__restrict__ Dh, int nh) {
  if (ctl->halt) return;
  for (int j = blockIdx.x * BS + threadIdx.x; j < nh; j += gridDim.x * BS) xh[j] /= Dh[j];
  const int lane = threadIdx.x & 63, wave = blockIdx.x * (BS / 64) + (threadIdx.x >> 6), nw = gridDim.x * (BS / 64);
  for (int r = wave; r < T; r += nw) {
    const int lo = upper ? r : 0, hi = upper ? T : r + 1;
    const double2 *row2 = reinterpret_cast<const double2 *>(M + (long)r * ld); // rows start 512-byte aligned (ld % 64 == 0)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // 16-byte loads: lane q of an iteration owns entries 2q, 2q+1 of a 128-entry slice; four slices in flight
    int c = (lo & ~127) + 2 * lane;
    auto mac = [&](int cc, double &acc) {
      const double2 m = row2[cc >> 1];
      if (cc >= lo && cc < hi) acc += m.x * v[cc];
      if (cc + 1 >= lo && cc + 1 < hi) acc += m.y * v[cc + 1];
    };
    auto mac_in = [&](int cc, double &acc) { const double2 m = row2[cc >> 1]; acc += m.x * v[cc]; acc += m.y * v[cc + 1]; }; // fully inside [lo, hi)
    if (c < hi) mac(c, a0);
    c += 128;
    for (; c + 3 * 128 + 1 < hi; c += 512) { mac_in(c, a0); mac_in(c + 128, a1); mac_in(c + 256, a2); mac_in(c + 384, a3); }
    for (; c < hi; c += 128) mac(c, a0);
    double s = (a0 + a1) + (a2 + a3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[r] = dsc ? s / dsc[r] : s;
  }
}

// (the tail as ONE symmetric mat-vec: dev_tail.h)

// ---- small systems: the sparse part of the solve in ONE 1024-thread workgroup ---------------------------------------------------
// A level costs one workgroup barrier plus the latency of whatever it loads after the barrier.  Only the gathers x[idx] depend
// on the previous level, so (i) x lives in LDS when it fits (N <= XL_MAX), (ii) the level table is copied to LDS up front and
// (iii) the row extents and first matrix entry of level l+1 are fetched while level l is being reduced.
constexpr int MAXLEV_LDS = 1024;
constexpr int XL_MAX = 16384;

struct RowPre { int row, s, e, i0; double v0; };
__device__ __forceinline__ RowPre row_pre(const Tri &T, int a, int b, int g, int tid) {
  RowPre p; p.row = -1; p.s = 0; p.e = 0; p.i0 = 0; p.v0 = 0.0;
  const int r = a + tid / g;
  if (r < b) {
    p.row = T.lev_rows[r]; p.s = T.ptr[r] + tid % g; p.e = T.ptr[r + 1];
    if (p.s < p.e) { p.v0 = T.val[p.s]; p.i0 = T.idx[p.s]; }
  }
  return p;
}

// levels [l0, l1) of T
__device__ __forceinline__ void run_levels(const Tri &T, double *x, int *s_lp, int *s_lg, int tid, int l0, int l1) {
  if (l1 <= l0) return;
  const int nl = l1 - l0;
  const bool meta = nl <= MAXLEV_LDS;
  if (meta) {
    for (int l = tid; l <= nl; l += TBS) s_lp[l] = T.lev_ptr[l0 + l];
    for (int l = tid; l < nl; l += TBS) s_lg[l] = T.lev_g[l0 + l];
    __syncthreads();
  }
  auto lp = [&](int l) { return meta ? s_lp[l] : T.lev_ptr[l0 + l]; };
  auto lg = [&](int l) { return meta ? s_lg[l] : T.lev_g[l0 + l]; };
  int a = lp(0), b = lp(1), g = lg(0);
  RowPre cur = row_pre(T, a, b, g, tid);
  for (int l = 0; l < nl; ++l) {
    int nb = b, ng = 1;
    RowPre nxt = cur;
    if (l + 1 < nl) { nb = lp(l + 2); ng = lg(l + 1); nxt = row_pre(T, b, nb, ng, tid); }
    {
      double acc = 0.0;
      if (cur.s < cur.e) {
        acc = cur.v0 * x[cur.i0];
        for (int k = cur.s + g; k < cur.e; k += g) acc += T.val[k] * x[T.idx[k]];
      }
      for (int off = g >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (cur.row >= 0 && tid % g == 0) x[cur.row] -= acc;
    }
    const int ngrp = TBS / g;
    if (a + ngrp < b) tri_level(T, x, a + ngrp, b, g, tid, TBS); // rows beyond the first 1024/g of a wide level
    __syncthreads();
    a = b; b = nb; g = ng; cur = nxt;
  }
}

// a run of thin levels [l0, l1) of a larger system in one workgroup (x in global memory)
static __global__ __launch_bounds__(TBS) void k_tri_thin(Tri T, double *x, int l0, int l1, const Ctl *ctl) {
  if (ctl->halt) return;
  __shared__ int s_lp[MAXLEV_LDS + 1], s_lg[MAXLEV_LDS];
  run_levels(T, x, s_lp, s_lg, threadIdx.x, l0, l1);
}

// Optional neighbours of the solve folded into the one-workgroup kernels (saves two launches per ADMM iteration on small LPs):
// pre(b, tid) builds the right-hand side in b before the permutation gathers it, post(b, tid) consumes the solution after the
// inverse permutation has scattered it.  Both run on all TBS threads of the single workgroup.
struct NoFuse {
  static constexpr bool active = false;
  __device__ bool pre(double *, int) const { return true; } // false: leave the kernel
  __device__ void post(const double *, int) const {}
};

// sum over the 1024-thread workgroup, result to every thread
__device__ __forceinline__ double tbs_sum(double v, double *red /* TBS/64 */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int wv = 0; wv < TBS / 64; ++wv) t += red[wv];
  return t;
}

// FWD: x = P b, L-solve over the sparse rows.  BWD: head D^-1, L'-solve over the head columns, b = P' x.  Both when there is no tail.
template <bool XL, bool FWD, bool BWD, class Fuse>
static __global__ __launch_bounds__(TBS) void k_ldl_small(Tri F, Tri B, const int *__restrict__ Pmap, const double *__restrict__ D, double *b, double *xg,
                                                          int t0, int N, const Ctl *ctl, Fuse fz) {
  if (ctl->halt) return;
  extern __shared__ double x_lds[];
  __shared__ int s_lp[MAXLEV_LDS + 1], s_lg[MAXLEV_LDS];
  double *x = XL ? x_lds : xg;
  const int tid = threadIdx.x;
  if (FWD) {
    if (Fuse::active) { if (!fz.pre(b, tid)) return; __syncthreads(); }
    for (int j = tid; j < N; j += TBS) x[j] = b[Pmap[j]];
    __syncthreads();
    run_levels(F, x, s_lp, s_lg, tid, 0, F.nlev);
    if (!BWD) { if (XL) for (int j = tid; j < N; j += TBS) xg[j] = x[j]; return; }
    for (int j = tid; j < t0; j += TBS) x[j] /= D[j];
  } else {
    for (int j = tid; j < N; j += TBS) { const double v = xg[j]; x[j] = j < t0 ? v / D[j] : v; }
  }
  __syncthreads();
  run_levels(B, x, s_lp, s_lg, tid, 0, B.nlev);
  for (int j = tid; j < N; j += TBS) b[Pmap[j]] = x[j];
  if (Fuse::active) { __syncthreads(); fz.post(b, tid); }
}

// lower triangle of a symmetric T x T array -> its upper triangle (64 x 64 tiles)
static __global__ __launch_bounds__(256) void k_dsym_fill_upper(double *M, int ld) {
  __shared__ double t[DB][DB + 1];
  const int bi = blockIdx.y, bj = blockIdx.x, tid = threadIdx.x;
  if (bj > bi) return;
  for (int e = tid; e < DB * DB; e += 256) t[e / DB][e % DB] = M[((long)bi * DB + e / DB) * ld + (long)bj * DB + e % DB];
  __syncthreads();
  for (int e = tid; e < DB * DB; e += 256) {
    const int r = e / DB, c = e % DB; // entry (bj*64 + r, bi*64 + c) of the upper part = entry (bi*64 + c, bj*64 + r) of the lower one
    if (bi != bj || c > r) M[((long)bj * DB + r) * ld + (long)bi * DB + c] = t[c][r];
  }
}

namespace hostutil {
// inv(S) of a symmetric positive definite T x T matrix (T % 64 == 0) given by its LOWER triangle in a device array that is overwritten; the full
// symmetric inverse is left in `out` (T x T, row-major).  The same kernels as the dense tail of the direct back-end: blocked LDL', inverse of the
// unit-lower factor, M = W' D^-1 W on the matrix cores.  Returns 0, or -1 (allocation failure / a zero or non-finite pivot).
inline int dense_spd_inverse(double *S, int T, hipStream_t s, DBuf<double> &out) {
  const int nt = T / DB;
  DBuf<double> Linv, LD, W, Dt; DBuf<int> flag;
  const std::vector<int> zero(1, 0);
  auto drop = [&]() { Linv.release(); LD.release(); W.release(); Dt.release(); flag.release(); };
  if (Linv.alloc((size_t)nt * DB * DB) || LD.alloc((size_t)T * DB) || W.alloc((size_t)T * T) || Dt.alloc(T) || flag.upload(zero, s) || out.alloc((size_t)T * T)) { drop(); return -1; }
  if (hipMemsetAsync(W.p, 0, sizeof(double) * (size_t)T * T, s) != hipSuccess) { drop(); return -1; }
  for (int kb = 0; kb < nt; ++kb) {
    const int k0 = kb * DB, rem = nt - kb - 1;
    hipLaunchKernelGGL(k_dldl_diag, dim3(1), dim3(256), 0, s, S, T, k0, Dt.p, Linv.p + (size_t)kb * DB * DB, flag.p);
    if (rem > 0) {
      hipLaunchKernelGGL(k_dldl_panel, dim3(rem), dim3(256), 0, s, S, T, k0, (const double *)Dt.p, (const double *)(Linv.p + (size_t)kb * DB * DB), LD.p);
      hipLaunchKernelGGL(k_dldl_update_mfma, dim3(rem * (rem + 1) / 2), dim3(256), 0, s, S, T, k0, (const double *)LD.p);
    }
  }
  for (int kb = 0; kb < nt; ++kb) hipLaunchKernelGGL(k_dtri_inv_row, dim3(kb + 1), dim3(256), 0, s, (const double *)S, T, kb, (const double *)Linv.p, W.p);
  // S (its factor is no longer needed) takes W'
  hipLaunchKernelGGL(k_dtranspose_lower, dim3(nt, nt), dim3(256), 0, s, (const double *)W.p, S, T);
  hipLaunchKernelGGL(k_dgemm_mfma, dim3(nt, nt), dim3(256), 0, s, out.p, (long)T, (const double *)S, (long)T, (const double *)W.p, (long)T, T, 1.0, 2, 0, (const double *)Dt.p);
  hipLaunchKernelGGL(k_dsym_fill_upper, dim3(nt, nt), dim3(256), 0, s, out.p, T);
  int bad = 0;
  const bool fail = hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess;
  drop();
  return (fail || bad) ? -1 : 0;
}
} // namespace hostutil

namespace hostutil {

struct DevLdl {
  DevTri F, B;
  DBuf<int> Pmap, flag;
  DBuf<double> D, xw, W, Wt, tmp;
  DBuf<double> Msym, rowpart, colpart; // the tail as one symmetric mat-vec (k_tail_sym): M = W' D2^-1 W, W and W' released
  SymPlan sym;       // how the triangle is dealt to the wavefronts (dev_tail.h)
  int n_sym_tiles = 0; // > 0: the symmetric form is in use (= workgroups of k_tail_sym)
  bool small = false, xl = false; // one-workgroup sparse part; x in LDS
  int bwd_lds = 0, n_cu = 0;      // backward wide levels with the tail's solution in LDS (dev_sptrsv.h k_tri_wide_lds): 0 off, else the LDS bytes; one workgroup per CU
  int N = 0, t0 = 0, T = 0;
  long lnnz = 0;

  // more than 64 KB of dynamic LDS has to be asked for, per kernel instantiation
  template <class Fuse>
  bool allow_lds() const {
    const int bytes = (int)(sizeof(double) * (size_t)N);
    return hipFuncSetAttribute((const void *)k_ldl_small<true, true, true, Fuse>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
           hipFuncSetAttribute((const void *)k_ldl_small<true, true, false, Fuse>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
           hipFuncSetAttribute((const void *)k_ldl_small<true, false, true, Fuse>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  }

  // pmap[k] = position in the caller's rhs vector of pivot k.  Returns 0, or -1 (allocation / zero pivot).
  int setup(const host::LdlHost &H, const std::vector<int> &pmap, hipStream_t s) {
    N = H.N; t0 = H.t0; T = H.T; lnnz = H.lnnz;
    const bool tms = getenv("ABIP_HIP_SETUP_TIMES") != nullptr;
    auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq = clk();
    auto lap = [&](const char *what) { if (tms) { (void)hipStreamSynchronize(s); const double t = clk(); printf("[setup]   device: %s %.3f s\n", what, t - tq); tq = t; } };
    if (Pmap.upload(pmap, s) || D.upload(H.D, s) || xw.alloc(N)) return -1;
    // the dense tail first: the host may still be building the level-ordered forms of the head (LdlHost::forms_job), which nothing below needs
    const int rt = T > 0 ? setup_tail(H, s, tms) : 0;
    H.wait_forms();
    if (rt) return rt;
    tq = clk();
    // small systems: the whole sparse part in one workgroup (x in LDS), whatever the shape of the levels
    const bool one_wg = N <= XL_MAX && H.fwd.idx.size() <= 32768 && H.bwd.idx.size() <= 32768;
    if (F.upload(H.fwd, s, one_wg) || B.upload(H.bwd, s, one_wg)) return -1;
    small = F.single_workgroup() && B.single_workgroup() && N <= 65536;
    xl = small && N <= XL_MAX;
    if (xl && !allow_lds<NoFuse>()) xl = false;
    lap("upload of the sparse head (forward / backward forms)");
    // backward wide levels: the tail's solution x2 resident in LDS where it fits and the levels are large enough to be bound by their gathers (ABIP_HIP_TRI_LDS=0 / 1 forces)
    bwd_lds = 0;
    if (!small && T > 0 && T <= 16384) {
      const char *e = getenv("ABIP_HIP_TRI_LDS");
      long wide_nnz = 0;
      for (const Segment &sg : B.segs) if (sg.wide) wide_nnz += (long)(sg.mean_len * (sg.b - sg.a));
      int dev = 0, cus = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
      n_cu = cus;
      if (cus > 0 && (e ? atoi(e) != 0 : wide_nnz >= 1000000)) {
        const int bytes = (int)(sizeof(double) * (size_t)T);
        bool ok = true;
        if (bytes > 48 * 1024)
          ok = hipFuncSetAttribute((const void *)k_tri_wide_lds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute((const void *)k_tri_wide_lds<64>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        if (ok) bwd_lds = bytes; else (void)hipGetLastError();
      }
    }
    return 0;
  }
  int setup_tail(const host::LdlHost &H, hipStream_t s, bool tms) {
    auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq = clk();
    auto lap = [&](const char *what) { if (tms) { (void)hipStreamSynchronize(s); const double t = clk(); printf("[setup]   device: %s %.3f s\n", what, t - tq); tq = t; } };
#ifdef ABIP_HIP_TEST_HOOKS // fault injection exists only in the library variant the tests build (libabip_hip_hooks.so): the shipped one ignores these variables
    if (getenv("ABIP_HIP_TAIL_FAIL")) return -1; // pretend the dense set-up failed (the callers fall back to T = 0)
#endif
    const int nt = T / DB;
    DBuf<double> Linv, LD;
    const std::vector<int> zero(1, 0);
    if (H.S.empty()) { // dev_schur: K22 arrives as triplets -- zero the block on the device and scatter them (no 8 T^2-byte host array, no upload of zeros)
      DBuf<int> kr, kc; DBuf<double> kv;
      const int nk = (int)H.k22_val.size();
      int badk = Wt.alloc((size_t)T * T) || hipMemsetAsync(Wt.p, 0, sizeof(double) * (size_t)T * T, s) != hipSuccess || kr.upload(H.k22_row, s) || kc.upload(H.k22_col, s) || kv.upload(H.k22_val, s);
      if (!badk && nk > 0) hipLaunchKernelGGL(k_scatter_k22, dim3(std::max(1, std::min(2048, (nk + 255) / 256))), dim3(256), 0, s, (const int *)kr.p, (const int *)kc.p, (const double *)kv.p, nk, Wt.p, T);
      if (!badk) badk = hipStreamSynchronize(s) != hipSuccess;
      kr.release(); kc.release(); kv.release();
      if (badk) return -1;
    } else if (Wt.upload(H.S, s)) return -1;
    if (W.alloc((size_t)T * T) || tmp.alloc(T) || Linv.alloc((size_t)nt * DB * DB) || LD.alloc((size_t)T * DB) || flag.upload(zero, s)) return -1;
    if (hipMemsetAsync(W.p, 0, sizeof(double) * (size_t)T * T, s) != hipSuccess) return -1;
    double *S = Wt.p, *Dt = D.p + t0;
    lap("upload of S (K22), allocations");
    const bool use_mfma = !(getenv("ABIP_HIP_MFMA") && atoi(getenv("ABIP_HIP_MFMA")) == 0); // the dense set-up products on the matrix cores (0: the LDS-tiled FMA kernels)
    if (H.dev_schur && H.schur_rows) { // sparse L21: the row-wise kernel (LDS accumulator of T doubles per wavefront)
      // L21 by tail row: (head column, position in the column form) of every entry, columns ascending inside a row (host_par.h: a stable bucket pass)
      std::vector<int> rp32;
      std::vector<long> cposv(H.l21_row.size());
      std::vector<int> rcol(H.l21_row.size());
      host::par_bucket((long)t0, H.l21_ptr.data(), (long)T, rp32, 1000000, [&](long, long q) { return (long)H.l21_row[q]; }, [&](long dst, long c, long q) { rcol[dst] = (int)c; cposv[dst] = q; });
      std::vector<long> rptr(rp32.begin(), rp32.end());
      DBuf<long> drp, dcp, dpos; DBuf<int> drc, dcr; DBuf<double> dcv;
      auto drop = [&]() { drp.release(); dcp.release(); dpos.release(); drc.release(); dcr.release(); dcv.release(); };
      if (drp.upload(rptr, s) || drc.upload(rcol, s) || dpos.upload(cposv, s) || dcp.upload(H.l21_ptr, s) || dcr.upload(H.l21_row, s) || dcv.upload(H.l21_val, s)) { drop(); return -1; }
      const size_t lds = sizeof(double) * (size_t)T;
      if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_schur_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { drop(); return -1; }
      hipLaunchKernelGGL(k_schur_rows, dim3(T), dim3(64), lds, s, S, T, (const long *)drp.p, (const int *)drc.p, (const long *)dpos.p, (const long *)dcp.p, (const int *)dcr.p,
                         (const double *)dcv.p, (const double *)D.p);
      if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { drop(); return -1; }
      drop();
    } else if (H.dev_schur) { // S arrived as K22: subtract L21 D1 L21' here, a panel of head columns at a time
      constexpr int KC = 2048;
      DBuf<long> lp; DBuf<int> lr; DBuf<double> lv, P, PD;
      auto drop = [&]() { lp.release(); lr.release(); lv.release(); P.release(); PD.release(); };
      if (lp.upload(H.l21_ptr, s) || lr.upload(H.l21_row, s) || lv.upload(H.l21_val, s) || P.alloc((size_t)T * KC) || PD.alloc((size_t)T * KC)) { drop(); return -1; }
      for (int c0 = 0; c0 < t0; c0 += KC) {
        const int kc = std::min(KC, t0 - c0), kcp = (kc + DH - 1) / DH * DH;
        if (H.l21_ptr[c0 + kc] == H.l21_ptr[c0]) continue; // no tail entries in these columns
        if (hipMemsetAsync(P.p, 0, sizeof(double) * (size_t)T * KC, s) != hipSuccess || hipMemsetAsync(PD.p, 0, sizeof(double) * (size_t)T * KC, s) != hipSuccess) { drop(); return -1; }
        hipLaunchKernelGGL(k_l21_panel, dim3((kc + 3) / 4), dim3(256), 0, s, (const long *)lp.p, (const int *)lr.p, (const double *)lv.p, (const double *)D.p, c0, kc, KC, P.p, PD.p);
        if (use_mfma) hipLaunchKernelGGL(k_schur_sub_mfma, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, S, T, (const double *)PD.p, (const double *)P.p, KC, kcp);
        else hipLaunchKernelGGL(k_schur_sub, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, S, T, (const double *)PD.p, (const double *)P.p, KC, kcp);
      }
      if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { drop(); return -1; }
      drop();
    }
    lap("Schur complement on the device");
    for (int kb = 0; kb < nt; ++kb) {
      const int k0 = kb * DB, rem = nt - kb - 1;
      hipLaunchKernelGGL(k_dldl_diag, dim3(1), dim3(256), 0, s, S, T, k0, Dt, Linv.p + (size_t)kb * DB * DB, flag.p);
      if (rem > 0) {
        hipLaunchKernelGGL(k_dldl_panel, dim3(rem), dim3(256), 0, s, S, T, k0, (const double *)Dt, (const double *)(Linv.p + (size_t)kb * DB * DB), LD.p);
        if (use_mfma) hipLaunchKernelGGL(k_dldl_update_mfma, dim3(rem * (rem + 1) / 2), dim3(256), 0, s, S, T, k0, (const double *)LD.p);
        else hipLaunchKernelGGL(k_dldl_update, dim3(rem * (rem + 1) / 2), dim3(256), 0, s, S, T, k0, (const double *)LD.p);
      }
    }
    lap("dense LDL' of the tail");
    if (use_mfma && nt >= 4) { // by halves (k_dgemm_mfma): nodes of the recursion in level order; Tm = L21 W11, then W21 = -W22 Tm
      struct Node { int lo, mid, hi, level; };
      std::vector<Node> nodes;
      std::function<int(int, int)> rec = [&](int lo, int hi) -> int {
        if (hi - lo == 1) return 0;
        const int mid = (lo + hi) / 2;
        const int lv = std::max(rec(lo, mid), rec(mid, hi)) + 1;
        nodes.push_back(Node{lo, mid, hi, lv});
        return lv;
      };
      rec(0, nt);
      std::stable_sort(nodes.begin(), nodes.end(), [](const Node &a, const Node &b) { return a.level < b.level; });
      DBuf<double> Tm;
      const long half = (long)((nt + 1) / 2) * DB;
      if (Tm.alloc((size_t)half * half)) return -1;
      hipLaunchKernelGGL(k_dtri_inv_diag, dim3(nt), dim3(256), 0, s, (const double *)Linv.p, W.p, T);
      for (const Node &nd : nodes) {
        const int mb = nd.hi - nd.mid, nb = nd.mid - nd.lo; // block rows / block columns of the corner
        const long Mr = (long)mb * DB, Nc = (long)nb * DB;
        const double *L21 = S + (long)nd.mid * DB * T + (long)nd.lo * DB;
        const double *W11 = W.p + (long)nd.lo * DB * T + (long)nd.lo * DB, *W22 = W.p + (long)nd.mid * DB * T + (long)nd.mid * DB;
        double *W21 = W.p + (long)nd.mid * DB * T + (long)nd.lo * DB;
        hipLaunchKernelGGL(k_dgemm_mfma, dim3(nb, mb), dim3(256), 0, s, Tm.p, Nc, L21, (long)T, W11, (long)T, (int)Nc, 1.0, 0, 1, (const double *)nullptr);
        hipLaunchKernelGGL(k_dgemm_mfma, dim3(nb, mb), dim3(256), 0, s, W21, (long)T, W22, (long)T, (const double *)Tm.p, Nc, (int)Mr, -1.0, 1, 0, (const double *)nullptr);
      }
      if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { Tm.release(); return -1; }
      Tm.release();
    } else
    for (int kb = 0; kb < nt; ++kb) hipLaunchKernelGGL(k_dtri_inv_row, dim3(kb + 1), dim3(256), 0, s, (const double *)S, T, kb, (const double *)Linv.p, W.p);
    // L22 is no longer needed: its buffer takes the transpose
    hipLaunchKernelGGL(k_dtranspose_lower, dim3(nt, nt), dim3(256), 0, s, (const double *)W.p, Wt.p, T);
    lap("inverse of L22 and its transpose");
    int bad = 0;
    if (hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return -1;
    Linv.release(); LD.release();
    if (bad) return -1;
    // large tails: M = W' D2^-1 W once, and every solve streams its lower triangle once (k_tail_sym) instead of W and W'.  Below ~2 000 pivots the two mat-vecs are
    // latency-bound and spread over more workgroups: kept.  ABIP_HIP_TAIL_SYM=0 / 1 forces either.
    const char *se = getenv("ABIP_HIP_TAIL_SYM");
    const bool want_sym = use_mfma && (se ? atoi(se) != 0 : T >= 2048);
    if (want_sym) {
      // wavefronts of the stream: eight per CU (two workgroups of four; ABIP_HIP_TAIL_WAVES names another number)
      const char *we = getenv("ABIP_HIP_TAIL_WAVES");
      const bool planned = sym.make(T, we && atoi(we) > 0 ? atoi(we) : 2048);
      if (!planned || Msym.alloc((size_t)T * T) || rowpart.alloc((size_t)sym.ncc * T) || colpart.alloc((size_t)sym.slots * SYC)) {
        Msym.release(); rowpart.release(); colpart.release(); (void)hipGetLastError(); // no room: the two mat-vecs stay
      } else {
        hipLaunchKernelGGL(k_dgemm_mfma, dim3(nt, nt), dim3(256), 0, s, Msym.p, (long)T, (const double *)Wt.p, (long)T, (const double *)W.p, (long)T, T, 1.0, 2, 0, (const double *)Dt);
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return -1;
        n_sym_tiles = sym.nwv / 4;
        // before W and W' go: M v against W' D2^-1 W v on one pseudo-random v.  The explicit inverse of S must not cost the solve more than the set-up guard
        // allows (1e-8); if it does, the two mat-vecs stay.
        bool keep = false;
        {
          std::vector<double> hv, ha((size_t)T), hb((size_t)T);
          host::guard_rhs(T, hv);
          DBuf<double> dv, da, db;
          Ctl *zc = nullptr;
          if (!dv.upload(hv, s) && !da.alloc(T) && !db.alloc(T) && hipMalloc((void **)&zc, sizeof(Ctl)) == hipSuccess && hipMemsetAsync(zc, 0, sizeof(Ctl), s) == hipSuccess) {
            const int grid = std::max(1, std::min(MAXNB, (T + BS / 64 - 1) / (BS / 64)));
            hipLaunchKernelGGL(k_tail_mv, dim3(grid), dim3(BS), 0, s, (const double *)W.p, T, T, 0, (const double *)dv.p, tmp.p, (const double *)Dt, (const Ctl *)zc, (double *)nullptr, (const double *)nullptr, 0);
            hipLaunchKernelGGL(k_tail_mv, dim3(grid), dim3(BS), 0, s, (const double *)Wt.p, T, T, 1, (const double *)tmp.p, da.p, (const double *)nullptr, (const Ctl *)zc, (double *)nullptr, (const double *)nullptr, 0);
            hipLaunchKernelGGL((k_tail_sym<2>), dim3(n_sym_tiles), dim3(256), 0, s, (const double *)Msym.p, T, (const double *)dv.p, rowpart.p, colpart.p, sym.args(), (const Ctl *)zc);
            hipLaunchKernelGGL(k_tail_sym_fin, dim3(nt), dim3(1024), 0, s, (const double *)rowpart.p, (const double *)colpart.p, sym.args(), db.p, (const Ctl *)zc, (double *)nullptr, (const double *)nullptr, 0);
            if (hipMemcpyAsync(ha.data(), da.p, sizeof(double) * T, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemcpyAsync(hb.data(), db.p, sizeof(double) * T, hipMemcpyDeviceToHost, s) == hipSuccess &&
                hipStreamSynchronize(s) == hipSuccess) {
              double num = 0.0, den = 0.0;
              for (int i = 0; i < T; ++i) { num += (ha[i] - hb[i]) * (ha[i] - hb[i]); den += ha[i] * ha[i]; }
              const double dev = std::sqrt(num) / std::max(std::sqrt(den), 1e-300);
              keep = dev == dev && dev <= 1e-9;
#ifdef ABIP_HIP_TEST_HOOKS
              if (getenv("ABIP_HIP_TAIL_SYM_FAIL")) keep = false; // pretend the explicit inverse lost the accuracy
#endif
              if (tms) printf("[setup]   device: M v against W' D2^-1 W v: relative difference %.2e (%s)\n", dev, keep ? "M kept" : "the two mat-vecs stay");
            }
          }
          (void)hipGetLastError();
          dv.release(); da.release(); db.release();
          if (zc) (void)hipFree(zc);
        }
        if (keep) { W.release(); Wt.release(); tmp.release(); }
        else { n_sym_tiles = 0; Msym.release(); rowpart.release(); colpart.release(); }
      }
      lap("M = W' D2^-1 W (the tail as one symmetric mat-vec)");
    }
    return 0;
  }

  // enqueue rhs <- K^-1 rhs; `launch(kernel, grid, block, lds_bytes, args...)` is the caller's launcher (profiling classes differ)
  template <class LaunchFn, class Fuse = NoFuse>
  void enqueue(LaunchFn &&launch, double *rhs, const Ctl *ctl, int NB, Fuse fz = Fuse{}) const {
    auto tail = [&](bool scale_head) {
      if (T == 0) return;
      if (n_sym_tiles > 0) {
        launch(k_tail_sym<2>, n_sym_tiles, 256, (size_t)0, (const double *)Msym.p, T, (const double *)(xw.p + t0), rowpart.p, colpart.p, sym.args(), ctl);
        launch(k_tail_sym_fin, std::max(T / 64, std::min(256, scale_head ? (t0 + 1023) / 1024 : 1)), 1024, (size_t)0, (const double *)rowpart.p, (const double *)colpart.p, sym.args(), xw.p + t0, ctl,
               xw.p, (const double *)D.p, scale_head ? t0 : 0);
        return;
      }
      const int grid = std::max(1, std::min(MAXNB, (T + BS / 64 - 1) / (BS / 64)));
      launch(k_tail_mv, grid, BS, (size_t)0, (const double *)W.p, T, T, 0, (const double *)(xw.p + t0), tmp.p, (const double *)(D.p + t0), ctl,
             (double *)nullptr, (const double *)nullptr, 0);
      launch(k_tail_mv, grid, BS, (size_t)0, (const double *)Wt.p, T, T, 1, (const double *)tmp.p, xw.p + t0, (const double *)nullptr, ctl,
             xw.p, (const double *)D.p, scale_head ? t0 : 0);
    };
    if (small) {
      const size_t sh = xl ? sizeof(double) * (size_t)N : 0;
      const int *pm = Pmap.p; const double *dd = D.p;
      if (T == 0) {
        if (xl) launch(k_ldl_small<true, true, true, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
        else launch(k_ldl_small<false, true, true, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
        return;
      }
      if (xl) launch(k_ldl_small<true, true, false, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
      else launch(k_ldl_small<false, true, false, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
      tail(false);
      if (xl) launch(k_ldl_small<true, false, true, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
      else launch(k_ldl_small<false, false, true, Fuse>, 1, TBS, sh, F.view(), B.view(), pm, dd, rhs, xw.p, t0, N, ctl, fz);
      return;
    }
    const int gN = std::max(1, std::min(NB, (N + BS - 1) / BS));
    launch(k_perm_in, gN, BS, (size_t)0, (const int *)Pmap.p, (const double *)rhs, xw.p, N, ctl);
    auto run = [&](const DevTri &Tr, bool backward) {
      for (const Segment &sg : Tr.segs) {
        if (sg.wide && backward && bwd_lds > 0) {
          if (sg.mean_len <= 96.0) launch(k_tri_wide_lds<16>, n_cu, 1024, (size_t)bwd_lds, Tr.view(), sg.a, sg.b, xw.p, t0, T, ctl);
          else launch(k_tri_wide_lds<64>, n_cu, 1024, (size_t)bwd_lds, Tr.view(), sg.a, sg.b, xw.p, t0, T, ctl);
        } else if (sg.wide) launch(k_tri_wide, std::max(1, std::min(NB, sg.nrb)), BS, (size_t)0, Tr.view(), (const int4 *)Tr.rbd.p + sg.rb0, sg.nrb, xw.p, ctl);
        else launch(k_tri_thin, 1, TBS, (size_t)0, Tr.view(), xw.p, sg.l0, sg.l1, ctl);
      }
    };
    run(F, false);
    tail(true);
    if (T == 0) launch(k_dscale, gN, BS, (size_t)0, xw.p, (const double *)D.p, t0, ctl);
    run(B, true);
    launch(k_perm_out, gN, BS, (size_t)0, (const int *)Pmap.p, rhs, (const double *)xw.p, N, ctl);
  }

  void release() {
    F.release(); B.release(); Pmap.release(); flag.release(); D.release(); xw.release(); W.release(); Wt.release(); tmp.release();
    Msym.release(); rowpart.release(); colpart.release(); n_sym_tiles = 0;
  }
};

} // namespace hostutil
} // namespace abip
