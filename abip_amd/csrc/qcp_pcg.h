// qcp_pcg.h -- the conic path's indirect KKT back-end (settings.linsys_solver = 3): Jacobi-PCG in y-space on the device.
//
//   K z = g,  K = [[-rho_y I, -A], [-A', H]],  H = Q + rho_x I                             (the system the direct back-end factors)
//   <=>  (rho_y I + A H^-1 A') z_y = -g_y - A H^-1 g_x,     z_x = H^-1 (g_x + A' z_y)        (H diagonal: Q absent or diagonal)
//
// What upstream has, and why this is NOT a restatement of it: the reference's own PCG for the generic conic formulation
// (qcp_pcg, src/abip-qcp/source/linsys.c:755-851, on the n-space matrix R_x + Q + A' R_y^-1 A, mat_vec :722-748) cannot be reached
// -- ABIP(solve_linsys) sends prob_type QCP to the LP-shaped `pcg` with mismatched dimensions (linsys.c:1158-1165) --
// and it would not work if it could: with the default rho_y = 1e-6 that matrix has condition number ~1e8 and n iterations (its cap)
// do not converge in double precision (restated on the CPU and tried (the test tree's CPU checker): residual oscillating between
// 1e-1 and 1e+2 on the 2 x 8 toy problem of test/test_abip_install.m).  The formulation that does work is the one the reference uses
// for its specialised problems, `pcg` (linsys.c:629-716): the y-space Schur complement rho I + A A', here with H^-1 in the middle.
// DEFINITION (the test tree's CPU checker restates the same steps; nothing here depends on it):
//   * pcg of linsys.c:629-716 on G = rho_y I + A H^-1 A' with the Jacobi preconditioner M_i = 1 / (rho_y + sum_j A_ij^2 / H_jj),
//     warm start y0, r = b - G y0, at most m iterations, stop after an update once |r|_2 < tol (linsys.c:683);
//   * warm start and tolerance as the projection prepares them for its PCG branch (abip.c:206-218): y0 = (u + tau r)_y,
//     tol = max(0.2 min(min(|Ax - b tau|_inf, |Qx - A'y + c tau - s|_inf of the last residual check; +inf before the first),
//                       |(u + tau r)[0:n]|_inf / (iter + 1)^1.5), 1e-12);  the set-up solve runs cold with tol = 1e-12 (abip.c:899);
//   * the projection around the solve is the generic one (abip.c:192-255);  a non-diagonal Q is refused (use linsys_solver = 1).
//
// Device form: the LP path's PCG (dev_kernels.h) with H^-1 between the two products: A'p_new = A'z + beta A'p_old, so every product
// gathers z (complete after the update kernel) and beta comes from the update's partials, re-reduced by the consumer.
#pragma once
#include "qcp_kernels.h"

namespace abip {

enum QPcgSlot : int { PQ_RM0 = 0, PQ_RM1, PQ_ZR0, PQ_ZR1, PQ_PG, PQ_WM, PQ_COUNT }; // RM*: |r|^2; WM: a max slot

template <int NS>
__device__ __forceinline__ void read_partials_max(const double *part, const int (&slots)[NS], int nb, double (&out)[NS], double *sm /* NS * WAVES */) {
  constexpr int PER = MAXNB / BS;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double t = 0.0; // the reduced quantities are absolute values
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int i = threadIdx.x + u * BS; if (i < nb) t = fmax(t, part[slots[s] * MAXNB + i]); }
    out[s] = wave_max(t);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sm[s * WAVES + wave] = out[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) { double t = sm[s * WAVES]; for (int w = 1; w < WAVES; ++w) t = fmax(t, sm[s * WAVES + w]); out[s] = t; }
}

struct QPcgVec { double *y0, *r, *z, *p, *Gp, *tn; const double *Minv, *Hinv; }; // m-space (tn: n-space H^-1 A'p)

// b = -g_y - A (H^-1 g_x) into the y block (the PCG's right-hand side); warm start y0 = (u + tau r)_y and the partial of
// |(u + tau r)[0:n]|_inf (abip.c:208-215: the first n entries of the (m + n)-vector)
__global__ __launch_bounds__(BS, 8) void kq_pcg_prep(Csr A, double *__restrict__ rhs, const double *__restrict__ u, const double *__restrict__ rv, int warm,
                                                  QDims d, QPcgVec v, double *ppart, Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  if (blockIdx.x == 0 && threadIdx.x == 0) { hc->cg_it = 0; hc->cg_done = 0; }
  const double *gx = rhs + d.MP;
  spmv_stream<1>(
      A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * (gx[c] * v.Hinv[c]); },
      [&](int i, double(&acc)[1]) { rhs[i] = -rhs[i] - acc[0]; });
  double wm[1] = {0.0};
  if (warm) {
    const double tau = u[d.MP + d.n];
    const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
    for (int i = t0; i < d.m; i += stride) v.y0[i] = u[i] + tau * rv[i];
    for (int e = t0; e < d.n; e += stride) { // all of y, and the first n - m entries of x when n > m
      const int q = e < d.m ? e : d.MP + (e - d.m);
      wm[0] = fmax(wm[0], fabs(u[q] + tau * rv[q]));
    }
  }
  const int ws[1] = {PQ_WM};
  write_partials_max<1>(ppart, ws, wm, sm);
}

// (launch bounds: 8 waves per SIMD = 64 VGPRs, so that a full grid of 2048 workgroups is resident at once -- at 68 VGPRs only 1792 are, and the
//  remaining 256 run as a second round: kq_pcg_Aty<false> took 180 us instead of 90 on the LASSO protocol's operator)
// tn = H^-1 (A' y).  INIT: y = y0.  Loop: the convergence test and beta from the update's partials, then tn = H^-1 (A' z) + beta tn.
template <bool INIT>
__global__ __launch_bounds__(BS, 8) void kq_pcg_Aty(Csr At, QPcgVec v, int max_its, double *ppart, int nb, Ctl *hc) {
  if (hc->halt || hc->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[2 * WAVES];
  double beta = 0.0;
  if (!INIT) {
    const int it = hc->cg_it, par = it & 1;
    double s2[2];
    const int sl[2] = {PQ_RM0 + par, PQ_ZR0 + par};
    read_partials<2>(ppart, sl, nb, s2, sm);
    const bool done = (it > 0 && sqrt(s2[0]) < hc->cg_tol) || it >= max_its || s2[0] == 0.0; // linsys.c:683; nothing to do for a zero residual
    if (done) { if (blockIdx.x == 0 && threadIdx.x == 0) hc->cg_done = 1; return; }
    beta = it == 0 ? 0.0 : s2[1] / hc->zr_hist[par ^ 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { hc->it_cur = it; hc->beta_cur = beta; hc->zr_cur = s2[1]; hc->zr_hist[par] = s2[1]; }
  }
  const double *g = INIT ? v.y0 : v.z;
  spmv_stream<1>(
      At, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * g[c]; },
      [&](int j, double(&acc)[1]) { const double t = acc[0] * v.Hinv[j]; v.tn[j] = (INIT || beta == 0.0) ? t : t + beta * v.tn[j]; });
}
// The same product with the gathered m-vector RESIDENT IN LDS (m <= 16 384: 128 KB): kq_pcg_Aty runs at the rate at which a CU keeps L2 gathers in flight (one
// request per non-zero); with y / z in LDS the gathers never leave the CU and what remains is the stream of A'.  One 1024-thread workgroup per
// CU (the vector takes most of its LDS); G lanes per row (16 for short rows, 64 for long ones), four loads in flight per lane, the next row's extent requested
// before the current row is reduced.  Same entry test and beta as kq_pcg_Aty; a row's products add up lane-strided, then by a G-lane butterfly (deterministic).
template <bool INIT, int G>
__global__ __launch_bounds__(1024) void kq_pcg_Aty_lds(Csr At, QPcgVec v, int m, int max_its, const double *__restrict__ ppart, int nb, Ctl *hc) {
  extern __shared__ double gl[];
  __shared__ double sm16[2][16];
  __shared__ int s_gate;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the gate is read ONCE per workgroup: workgroup 0 of this very launch may raise cg_done below, and a late wavefront that saw 1 where its
  // siblings saw 0 would leave them alone at the barriers
  if (tid == 0) s_gate = (hc->halt || hc->cg_done) ? 1 : 0;
  __syncthreads();
  if (s_gate) return;
  const double *g = INIT ? v.y0 : v.z;
  for (int i = tid; i < m; i += 1024) gl[i] = g[i];
  double beta = 0.0;
  if (!INIT) {
    const int it = hc->cg_it, par = it & 1;
    double a0 = 0.0, a1 = 0.0;
    for (int i = tid; i < nb; i += 1024) { a0 += ppart[(PQ_RM0 + par) * MAXNB + i]; a1 += ppart[(PQ_ZR0 + par) * MAXNB + i]; }
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); }
    if (lane == 0) { sm16[0][wave] = a0; sm16[1][wave] = a1; }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s0 += sm16[0][k]; s1 += sm16[1][k]; }
    const bool done = (it > 0 && sqrt(s0) < hc->cg_tol) || it >= max_its || s0 == 0.0; // linsys.c:683
    if (done) { if (blockIdx.x == 0 && tid == 0) hc->cg_done = 1; return; }
    beta = it == 0 ? 0.0 : s1 / hc->zr_hist[par ^ 1];
    if (blockIdx.x == 0 && tid == 0) { hc->it_cur = it; hc->beta_cur = beta; hc->zr_cur = s1; hc->zr_hist[par] = s1; }
  }
  __syncthreads(); // gl is complete
  constexpr int GPW = 64 / G;
  const int gidx = (blockIdx.x * 16 + wave) * GPW + lane / G, gl_ = lane % G, TG = gridDim.x * 16 * GPW, n = At.nrows;
  // two rows per lane group at a time (eight loads in flight per lane); the next pair's extents are requested before the current pair is reduced
  auto extent = [&](int row, int &s, int &e) { s = 0; e = 0; if (row < n) { s = At.ptr[row]; e = At.ptr[row + 1]; } };
  auto finish = [&](int row, double acc) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, G);
    if (gl_ == 0 && row < n) { const double t = acc * v.Hinv[row]; v.tn[row] = (INIT || beta == 0.0) ? t : t + beta * v.tn[row]; }
  };
  int r = gidx, sa, ea, sb, eb;
  extent(r, sa, ea); extent(r + TG, sb, eb);
  while (r < n) {
    int na, ma, nb2, mb;
    extent(r + 2 * TG, na, ma); extent(r + 3 * TG, nb2, mb);
    double acca = 0.0, accb = 0.0;
    for (int qa = sa + gl_, qb = sb + gl_; qa < ea || qb < eb; qa += 4 * G, qb += 4 * G) {
      double a[4], b[4]; int ca[4], cb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q1 = qa + k * G, q2 = qb + k * G;
        const bool o1 = q1 < ea, o2 = q2 < eb;
        a[k] = o1 ? At.val[q1] : 0.0; ca[k] = o1 ? At.idx[q1] : 0;
        b[k] = o2 ? At.val[q2] : 0.0; cb[k] = o2 ? At.idx[q2] : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { acca += a[k] * gl[ca[k]]; accb += b[k] * gl[cb[k]]; }
    }
    finish(r, acca); finish(r + TG, accb);
    r += 2 * TG; sa = na; ea = ma; sb = nb2; eb = mb;
  }
}
// INIT: r = b - (rho_y y0 + A tn); y = y0; z = M r; p = z; partials |r|^2, z'r; the tolerance.
// Loop: p = z + beta p; Gp = rho_y p + A tn; partial p'Gp.
template <bool INIT>
__global__ __launch_bounds__(BS, 8) void kq_pcg_Gp(Csr A, QPcgVec v, double *__restrict__ ysol /* rhs y block: b in, y out */, double rho_y,
                                                double tol_host, double iter_pow, double *ppart, int nb, Ctl *hc) {
  if (hc->halt || hc->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[2 * WAVES];
  if (INIT) {
    double wm[1];
    const int s1[1] = {PQ_WM};
    read_partials_max<1>(ppart, s1, nb, wm, sm);
    if (blockIdx.x == 0 && threadIdx.x == 0) hc->cg_tol = fmax(0.2 * fmin(tol_host, wm[0] / iter_pow), 1e-12); // abip.c:213-217
    double a2[2] = {0.0, 0.0};
    spmv_stream<1>(
        A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * v.tn[c]; },
        [&](int i, double(&acc)[1]) {
          const double y0 = v.y0[i];
          const double ri = ysol[i] - (y0 * rho_y + acc[0]), zi = ri * v.Minv[i];
          ysol[i] = y0; v.r[i] = ri; v.z[i] = zi; v.p[i] = zi;
          a2[0] += ri * ri; a2[1] += zi * ri;
        });
    const int ws[2] = {PQ_RM0, PQ_ZR0};
    write_partials<2>(ppart, ws, a2, sm);
  } else {
    const double beta = hc->beta_cur;
    double acc1[1] = {0.0};
    spmv_stream<1>(
        A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * v.tn[c]; },
        [&](int i, double(&acc)[1]) {
          const double pn = beta == 0.0 ? v.z[i] : v.z[i] + beta * v.p[i];
          const double gp = pn * rho_y + acc[0];
          v.p[i] = pn; v.Gp[i] = gp;
          acc1[0] += pn * gp;
        });
    const int ws[1] = {PQ_PG};
    write_partials<1>(ppart, ws, acc1, sm);
  }
}
// no warm start (the set-up solve): y = 0, r = b, z = M r, p = z
__global__ __launch_bounds__(BS) void kq_pcg_init_cold(QPcgVec v, double *__restrict__ ysol, int m, double tol, double *ppart, Ctl *hc) {
  if (hc->halt) return;
  __shared__ double sm[2 * WAVES];
  if (blockIdx.x == 0 && threadIdx.x == 0) hc->cg_tol = tol;
  double a2[2] = {0.0, 0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    const double ri = ysol[i], zi = ri * v.Minv[i];
    ysol[i] = 0.0; v.r[i] = ri; v.z[i] = zi; v.p[i] = zi;
    a2[0] += ri * ri; a2[1] += zi * ri;
  }
  const int ws[2] = {PQ_RM0, PQ_ZR0};
  write_partials<2>(ppart, ws, a2, sm);
}
// y += alpha p; r -= alpha Gp; z = M r; partials |r|^2, z'r of the next parity (linsys.c:676-700)
__global__ __launch_bounds__(BS) void kq_pcg_update(QPcgVec v, double *__restrict__ ysol, int m, double *ppart, int nb, Ctl *hc) {
  if (hc->halt || hc->cg_done) return;
  __shared__ double sm[2 * WAVES];
  double pg[1];
  const int rs[1] = {PQ_PG};
  read_partials<1>(ppart, rs, nb, pg, sm);
  const int it = hc->it_cur;
  const double alpha = hc->zr_cur / pg[0];
  double a2[2] = {0.0, 0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    ysol[i] += alpha * v.p[i];
    const double ri = v.r[i] - alpha * v.Gp[i], zi = ri * v.Minv[i];
    v.r[i] = ri; v.z[i] = zi;
    a2[0] += ri * ri; a2[1] += zi * ri;
  }
  const int par = (it + 1) & 1;
  const int ws[2] = {PQ_RM0 + par, PQ_ZR0 + par};
  write_partials<2>(ppart, ws, a2, sm);
  if (blockIdx.x == 0 && threadIdx.x == 0) hc->cg_it = it + 1;
}
// z_x = H^-1 (g_x + A' z_y); runs once the PCG has converged (re-checks: the last update of a chunk has no product behind it)
__global__ __launch_bounds__(BS, 8) void kq_pcg_post(Csr At, double *__restrict__ rhs, QPcgVec v, int max_its, QDims d, double *ppart, int nb, Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  if (!hc->cg_done) {
    const int it = hc->cg_it, par = it & 1;
    double rr[1];
    const int s1[1] = {PQ_RM0 + par};
    read_partials<1>(ppart, s1, nb, rr, sm);
    const bool done = (it > 0 && sqrt(rr[0]) < hc->cg_tol) || it >= max_its || rr[0] == 0.0;
    if (!done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) hc->cg_done = 1;
  }
  double *gx = rhs + d.MP;
  spmv_stream<1>(
      At, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * rhs[c]; },
      [&](int j, double(&acc)[1]) { gx[j] = (gx[j] + acc[0]) * v.Hinv[j]; });
}

} // namespace abip
