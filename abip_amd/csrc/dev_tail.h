// dev_tail.h -- the dense tail of the direct KKT back-end as ONE symmetric mat-vec (solve time; set-up is in dev_ldl.h).
//
//   x2 = W' D2^-1 W w = M w,  M = inv(S) formed once at set-up (k_dgemm_mfma), its LOWER triangle stored row-major with leading dimension ld.
//
// Reference: the two triangular sweeps over the last T pivots of LDL_lsolve / LDL_ltsolve (external/ldl/ldl.c:357-457) and, on the conic path, of
// QDLDL_solve (external/qdldl/src/qdldl.c:236-281).  Every stored entry serves two outputs (M[r][c] w[c] into row r, M[r][c] w[r] into row c), so a solve
// streams the triangle ONCE: 4 T^2 bytes instead of the 8 T^2 of two triangular mat-vecs (C5, T = 10 048: 404 MB per KKT solve).
//
// The kernel is a pure HBM stream and is built like one (round 6; round 5's form reduced every row across the wavefront as it went -- six dependent
// cross-lane steps per 4 KB of matrix -- and sat at 0.48 of HBM peak):
//   * the triangle is cut into column chunks of SYC = 512 columns and each chunk into WAVE-TILES of SYR = 16 consecutive rows; the wave-tiles, taken chunk
//     by chunk and top to bottom, form one list of Nw entries and wavefront q of NWV owns the contiguous range [q Nw / NWV, (q + 1) Nw / NWV): every
//     wavefront streams the same number of bytes (+- one wave-tile of 64 KB), walking DOWN a column chunk;
//   * a lane owns 8 of the chunk's columns (four 16-byte loads per row: a wavefront reads 4 KB of a row with each sweep) -- its column sums stay in registers
//     for as long as the wavefront stays in the chunk and go out once, as 16-byte stores, when it leaves (slot q + cc of `colpart`: the sequence of
//     (wavefront, chunk) pairs is monotone in both, so the sum is a unique, dense slot number);
//   * the 16 row sums of a wave-tile stay in registers too (one accumulator per row and lane), no cross-lane traffic inside the loop: all 64 loads of a
//     wave-tile are independent; they are reduced together afterwards by a transposing butterfly (8 + 4 + 2 + 1 + 1 + 1 = 17 cross-lane steps for 16 rows
//     instead of 96) and leave as `rowpart[chunk][row]`;
//   * w[r] of the row being swept is wave-uniform: a scalar load.
// k_tail_sym_fin adds, for output i, the row partials of the chunks left of the diagonal and the column partials of every wavefront that walked through i's
// chunk, in a fixed order: no atomics, the same bits every time.
// Algorithmic bytes per launch (the tail's share of B_solve_direct): 4 T (T + 1) + the two partial tables written and read once,
// 8 (ncc T + (NWV + ncc) 512) each way.
#pragma once
#include "dev_common.h"

namespace abip {

constexpr int SYC = 512; // columns of a chunk
constexpr int SYR = 16;  // rows whose sums are reduced together (a wave-tile)
constexpr int SYU = 4;   // rows of a unit: what a wavefront requests together, and the grain of the partition

struct SymArgs { int T, ncc, nu, nwv; int pre[64 + 1]; int qlo[64], qhi[64]; };
struct SymPlan : SymArgs {    // host side: how the triangle is dealt to the wavefronts (T % 64 == 0)
  int slots = 0;
  static long first_unit(long q, long nu, long nwv) { return q * nu / nwv; }
  bool make(int T_, int waves_wanted) {
    T = T_; ncc = (T + SYC - 1) / SYC;
    if (T <= 0 || T % SYR || ncc > 64) return false;
    pre[0] = 0;
    for (int cc = 0; cc < ncc; ++cc) pre[cc + 1] = pre[cc] + (T - SYC * cc) / SYU; // units of chunk cc: its rows [512 cc, T), four at a time
    nu = pre[ncc];
    nwv = std::max(4, std::min(waves_wanted, std::max(nu / 4, 4)) / 4 * 4); // (at least four units = one 16-row tile per wavefront: small tails take fewer, longer walks)
    slots = nwv + ncc;
    // the wavefronts whose range meets chunk cc (a contiguous run: the ranges are in list order)
    int q = 0;
    for (int cc = 0; cc < ncc; ++cc) {
      while (q + 1 < nwv && first_unit(q + 1, nu, nwv) <= pre[cc]) ++q;
      qlo[cc] = q;
      int e = q;
      while (e + 1 < nwv && first_unit(e + 1, nu, nwv) < pre[cc + 1]) ++e;
      qhi[cc] = e;
    }
    for (int cc = ncc; cc < 64; ++cc) { qlo[cc] = 0; qhi[cc] = -1; }
    return true;
  }
  const SymArgs &args() const { return *this; }
};

// 16 per-lane partial sums of 16 rows -> the rows' totals: afterwards every lane holds the total of row (lane >> 2) in a[0].
__device__ __forceinline__ void sym_reduce16(double (&a)[SYR], int lane) {
#pragma unroll
  for (int n = SYR / 2, off = 32; n >= 1; n >>= 1, off >>= 1) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
      const double send = hi ? a[k] : a[k + n], keep = hi ? a[k + n] : a[k];
      a[k] = keep + __shfl_xor(send, off, 64);
    }
  }
  a[0] += __shfl_xor(a[0], 2, 64);
  a[0] += __shfl_xor(a[0], 1, 64);
}

// SYU consecutive rows of a chunk as a wavefront holds them: 16 loads of 16 bytes per lane, and w at those rows (wave-uniform: scalar loads)
struct SymRows { double2 m[SYU][4]; double wr[SYU]; };
// (a lane whose columns lie beyond the diagonal entry of a row reads the row's last pair instead: rows are 512-byte aligned, c0 <= r, and the loads stay inside the
// row; below the chunk's first 512 rows the clamp never bites)
__device__ __forceinline__ void sym_load(SymRows &b, const double *__restrict__ M, int ld, const double *__restrict__ w, int r, int c0, int lane) {
#pragma unroll
  for (int i = 0; i < SYU; ++i) {
    const double2 *row2 = reinterpret_cast<const double2 *>(M + (long)(r + i) * ld + c0);
    const int last = (r + i - c0) >> 1; // the pair that holds the diagonal entry
    b.wr[i] = w[r + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) b.m[i][k] = row2[min(64 * k + lane, last)];
  }
}
// rows strictly below the diagonal of the chunk (r >= c0 + 512): every entry counts for its row and for its column
__device__ __forceinline__ void sym_fma(const SymRows &b, double *acc /* SYU row sums */, double2 (&ca)[4], const double2 (&wc)[4]) {
#pragma unroll
  for (int i = 0; i < SYU; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += b.m[i][k].x * wc[k].x; s += b.m[i][k].y * wc[k].y; ca[k].x += b.m[i][k].x * b.wr[i]; ca[k].y += b.m[i][k].y * b.wr[i]; }
    acc[i] = s;
  }
  // (the column sums are wanted HERE: without this the compiler postpones their FMAs to the end of the wave-tile and parks the matrix entries in scratch memory until then)
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ca[k].x), "+v"(ca[k].y));
}
// rows the diagonal runs through (the first 512 of a chunk): entry (r, c) counts for row r when c <= r and for column c when c < r; what lies right of the diagonal
// was never read
__device__ __forceinline__ void sym_fma_diag(const SymRows &b, double *acc, double2 (&ca)[4], const double2 (&wc)[4], int r, int c0, int lane) {
#pragma unroll
  for (int i = 0; i < SYU; ++i) {
    const int rr = r + i;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 128 * k + 2 * lane;
      const double2 m = b.m[i][k];
      const double mxr = (c <= rr) ? m.x : 0.0, myr = (c + 1 <= rr) ? m.y : 0.0, mxc = (c < rr) ? m.x : 0.0, myc = (c + 1 < rr) ? m.y : 0.0;
      s += mxr * wc[k].x; s += myr * wc[k].y; ca[k].x += mxc * b.wr[i]; ca[k].y += myc * b.wr[i];
    }
    acc[i] = s;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ca[k].x), "+v"(ca[k].y));
}

// Rows [r, rend) of chunk cc (multiples of four), all of them above (DIAG) or all of them below the chunk's 512th row.  Four rows are requested while the four before
// them are consumed (two register buffers; the scheduler is told not to move anything across the phases: left alone it hoists all 64 loads of a wave-tile and
// spills).  The request runs ahead across wave-tiles: the cross-lane reduction of a tile's 16 row sums overlaps the next tile's first loads.
template <bool DIAG>
__device__ __forceinline__ void sym_stream(const double *__restrict__ M, int ld, const double *__restrict__ w, double *__restrict__ rowpart, int T, int cc, int r, int rend, int lane,
                                           double2 (&ca)[4], const double2 (&wc)[4]) {
  const int c0 = cc * SYC;
  SymRows bufA, bufB;
  sym_load(bufA, M, ld, w, r, c0, lane);
  while (r < rend) {
    const int nr = min(SYR, rend - r); // a multiple of four; 16 unless this is the range's last tile
    const bool more = r + SYR < rend;
    double acc[SYR];
#pragma unroll
    for (int i = 0; i < SYR; ++i) acc[i] = 0.0;
#pragma unroll
    for (int g = 0; g < SYR / SYU; g += 2) {
      // (a request is never conditional -- behind the range's end it asks for the tile's own first rows once more and nobody uses them: a conditional request
      // would make the compiler keep the buffers in scratch memory)
      sym_load(bufB, M, ld, w, (g + 1) * SYU < nr ? r + (g + 1) * SYU : r, c0, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (g * SYU < nr) {
        if (DIAG) sym_fma_diag(bufA, acc + g * SYU, ca, wc, r + g * SYU, c0, lane);
        else sym_fma(bufA, acc + g * SYU, ca, wc);
      }
      __builtin_amdgcn_sched_barrier(0);
      sym_load(bufA, M, ld, w, (g + 2) * SYU < nr ? r + (g + 2) * SYU : ((g + 2) * SYU == SYR && more ? r + SYR : r), c0, lane); // (the last group of a tile requests the next tile's first rows)
      __builtin_amdgcn_sched_barrier(0);
      if ((g + 1) * SYU < nr) {
        if (DIAG) sym_fma_diag(bufB, acc + (g + 1) * SYU, ca, wc, r + (g + 1) * SYU, c0, lane);
        else sym_fma(bufB, acc + (g + 1) * SYU, ca, wc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    sym_reduce16(acc, lane);
    if ((lane & 3) == 0 && (lane >> 2) < nr) rowpart[(long)cc * T + r + (lane >> 2)] = acc[0];
    r += nr;
  }
}

// MINW wavefronts per SIMD bound the registers.
template <int MINW>
static __global__ __launch_bounds__(256, MINW) void k_tail_sym(const double *__restrict__ M, int ld, const double *__restrict__ w, double *__restrict__ rowpart,
                                                                double *__restrict__ colpart, const SymArgs sa, const Ctl *ctl) {
  if (ctl->halt) return;
  const int lane = threadIdx.x & 63, T = sa.T;
  const int q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))); // this wavefront's number: a scalar
  if (q >= sa.nwv) return;
  int u = (int)((long)q * sa.nu / sa.nwv);
  const int u1 = (int)((long)(q + 1) * sa.nu / sa.nwv);
  if (u >= u1) return;
  int cc = 0;
  while (sa.pre[cc + 1] <= u) ++cc;
  while (u < u1) {
    const int c0 = cc * SYC, pre_cc = sa.pre[cc];
    const int uend = min(u1, sa.pre[cc + 1]);
    const int r = c0 + SYU * (u - pre_cc), rend = c0 + SYU * (uend - pre_cc);
    double2 wc[4], ca[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 128 * k + 2 * lane;
      wc[k] = c + 1 < T ? *reinterpret_cast<const double2 *>(w + c) : make_double2(c < T ? w[c] : 0.0, 0.0);
      ca[k] = make_double2(0.0, 0.0);
    }
    const int rdiag = c0 + SYC; // rows below it lie strictly under the diagonal: no masks
    if (r < rdiag) sym_stream<true>(M, ld, w, rowpart, T, cc, r, min(rend, rdiag), lane, ca, wc); // the chunk's top: the diagonal runs through these rows
    if (rend > rdiag) sym_stream<false>(M, ld, w, rowpart, T, cc, max(r, rdiag), rend, lane, ca, wc);
    double2 *cp = reinterpret_cast<double2 *>(colpart + (long)(q + cc) * SYC);
#pragma unroll
    for (int k = 0; k < 4; ++k) cp[64 * k + lane] = ca[k];
    u = uend;
    ++cc;
  }
}

// x2[i] = sum over the chunks cc <= i / 512 of rowpart[cc][i] + sum over the wavefronts that walked through i's chunk (qlo .. qhi: the host's table) of their column
// partials; then (xh, Dh, nh): optionally xh[j] /= Dh[j] for j < nh -- the head's D^-1 of the segmented path rides on this launch.  One 1024-thread workgroup per 64
// outputs: sixteen groups of lanes share the partials of an output and are added in a fixed order.
static __global__ __launch_bounds__(1024) void k_tail_sym_fin(const double *__restrict__ rowpart, const double *__restrict__ colpart, const SymArgs sa, double *__restrict__ x2,
                                                              const Ctl *ctl, double *__restrict__ xh, const double *__restrict__ Dh, int nh) {
  if (ctl->halt) return;
  __shared__ double ps[16][64];
  const int tid = threadIdx.x, l = tid & 63, g = tid >> 6, T = sa.T;
  for (int j = blockIdx.x * 1024 + tid; j < nh; j += gridDim.x * 1024) xh[j] /= Dh[j];
  for (int ib = blockIdx.x; ib * 64 < T; ib += gridDim.x) {
    const int i = ib * 64 + l, cc = (ib * 64) / SYC;
    const int qlo = sa.qlo[cc], qhi = sa.qhi[cc];
    double s = 0.0;
    for (int c2 = g; c2 <= cc; c2 += 16) s += rowpart[(long)c2 * T + i];
    const double *cp = colpart + (long)cc * SYC + (i - cc * SYC);
    int qq = qlo + g;
    for (; qq + 48 <= qhi; qq += 64) { // four partials in flight
      const double v0 = cp[(long)qq * SYC], v1 = cp[(long)(qq + 16) * SYC], v2 = cp[(long)(qq + 32) * SYC], v3 = cp[(long)(qq + 48) * SYC];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; qq <= qhi; qq += 16) s += cp[(long)qq * SYC];
    ps[g][l] = s;
    __syncthreads();
    if (g == 0) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += ps[k][l];
      x2[i] = t;
    }
    __syncthreads();
  }
}

} // namespace abip
