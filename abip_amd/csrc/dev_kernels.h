// dev_kernels.h -- the HIP kernels of the ABIP-LP hot path (gfx950, fp64, wave64).
//
// One inner ADMM iteration (reference: src/abip-lp/src/abip.c:2131-2215) is the kernel chain
//
//   k_rhs                      project_lin_sys prologue, abip.c:552-558
//   [indirect] k_cg_init_At, k_cg_init_A, { k_cg_spmv_At, k_cg_spmv_A, k_cg_update }*, k_post_At
//                              solve_lin_sys / pcg, linsys/indirect.c:321-434
//   [direct]   k_sptrsv_*      _ldl_solve, linsys/direct.c:172-198 ; then k_post_dot
//   k_admm_update              u_t tau recovery abip.c:560, project_barrier 717-748, update_dual_vars
//                              567-584, restart sums 602-606, compute_avg 649-656 (+ the reductions the
//                              NEXT k_rhs and the inner stopping test need)
//   k_q_A, k_q_At              iterate_Q_norm_resd 1976-1996 and calc_residuals 385-453 in one pass each
//   k_finalize                 partials -> scalars for the host's once-per-iteration control read
//
// All reductions go through the partials table (dev_common.h); no atomics, no fences.
#pragma once
#include "dev_common.h"
#include "dev_peer.h"
#include "lp_scalars.h"

namespace abip {

struct Dims { int m, n, MP; }; // MP = offset of the x block inside an l-vector; tau sits at MP+n

#define ABIP_GATE_HALT(ctl) do { if ((ctl)->halt) return; } while (0)

// ---------------------------------------------------------------------------------------------
// S_WG <- sum_y rho*(u+v)*g + sum_x (u+v)*g        (the part of abip.c:557's dot that does not
// depend on tau; k_rhs adds the tau term analytically: (w - t h)'g = w'g - t g_th)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BS) void k_dot_wg(const double *__restrict__ u, const double *__restrict__ v,
                                               const double *__restrict__ g, double rho, Dims d, double *part, double xw) {
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) acc[0] += rho * (u[i] + v[i]) * g[i];
  for (int j = t0; j < d.n; j += stride) acc[0] += xw * ((u[d.MP + j] + v[d.MP + j]) * g[d.MP + j]);
  const int slots[1] = {S_WG};
  write_partials<1>(part, slots, acc, sm);
}

// ||y||^2 of an l-vector's y block -> S_BN (setup solve only; k_rhs produces it in the loop)
__global__ __launch_bounds__(BS) void k_norm_y(const double *__restrict__ x, Dims d, double *part) {
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < d.m; i += gridDim.x * BS) acc[0] += x[i] * x[i];
  const int slots[1] = {S_BN};
  write_partials<1>(part, slots, acc, sm);
}

// ---------------------------------------------------------------------------------------------
// k_rhs: u_t <- rhs of the KKT system (abip.c:552-558).
//   u_t = u + v; u_t[y] *= rho; u_t[0:l-1) -= tau~ h; u_t[0:l-1) -= (u_t'g/(g_th+1)) h; u_t[x] *= -1
// ---------------------------------------------------------------------------------------------
// `aty` / `pair` (PCG back-end, one GPU; both or neither): A'u_y of the current iterate is already known -- the previous iteration's back-substitution left it
// (k_post_At) -- so the pairs (rhs_x[j], (A's)[j]) that k_cg_init_A gathers are written here and k_cg_init_At's product is not repeated.
__global__ __launch_bounds__(BS) void k_rhs(const double *__restrict__ u, const double *__restrict__ v, double *__restrict__ ut,
                                            const double *__restrict__ h, double rho, double g_th, Dims d,
                                            double *part, int nb, const Ctl *ctl, const double *gs,
                                            const double *__restrict__ aty, double2 *__restrict__ pair) {
  ABIP_GATE_HALT(ctl);
  __shared__ double sm[WAVES];
  double wg[1];
  const int rs[1] = {S_WG};
  get_scalars<1>(part, rs, nb, wg, sm, gs);
  const int tail = d.MP + d.n;
  const double tsum = u[tail] + v[tail];
  const double coef = (wg[0] - tsum * g_th) / (g_th + 1.0);
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) {
    double t = (u[i] + v[i]) * rho;
    t += -tsum * h[i];
    t += -coef * h[i];
    ut[i] = t;
    acc[0] += t * t;
  }
  for (int j = t0; j < d.n; j += stride) {
    double t = u[d.MP + j] + v[d.MP + j];
    t += -tsum * h[d.MP + j];
    t += -coef * h[d.MP + j];
    ut[d.MP + j] = -t;
    if (pair) pair[j] = make_double2(-t, aty[j]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ut[tail] = tsum;
  const int ws[1] = {S_BN};
  write_partials<1>(part, ws, acc, sm);
}

// ---------------------------------------------------------------------------------------------
// PCG set-up (indirect.c:345-365).  tmp = A' s ; then in ONE pass over the rows of A:
//   b_i = rhs_y[i] + (A rhs_x)_i      (indirect.c:415)
//   r_i = b_i - ((A tmp)_i + rho s_i) (indirect.c:352-354)      x0 = s
//   z_i = M_i r_i ; p = z             (indirect.c:364-365)
// Both products gather at the same column, so k_cg_init_At leaves (rhs_x[j], tmp[j]) side by side and k_cg_init_A fetches the pair
// with ONE 16-byte gather per non-zero: two separate 8-byte gathers cost a second trip through the texture path (measured on C4:
// 82 us and 4.6x the algorithmic traffic against ~35 us for every other product of the same matrix).  The two sums stay separate,
// exactly as the reference forms them.  Without a warm start (s == nullptr): r = b, x0 = 0 (indirect.c:347-348), one plain gather.
// ---------------------------------------------------------------------------------------------
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_cg_init_At(Csr At /* rows = columns of A */, const double *__restrict__ s, const double *__restrict__ bx,
                                                   double2 *__restrict__ pair, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  spmv_rows<1, SELL>(
      At, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * s[c]; },
      [&](int row, double(&acc)[1]) { pair[row] = make_double2(bx[row], acc[0]); });
}
// sharded: T holds A's summed over the ranks -> the same pairs
__global__ __launch_bounds__(BS) void k_cg_init_pair(const double *__restrict__ T, const double *__restrict__ bx, double2 *__restrict__ pair, int n, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  for (int j = blockIdx.x * BS + threadIdx.x; j < n; j += gridDim.x * BS) pair[j] = make_double2(bx[j], T[j]);
}

template <bool DIST, bool SELL> // DIST: also the partial of z'z (= ||p||^2 of the first direction) for the sharded path
__global__ __launch_bounds__(BS, SELL ? 5 : 8) void k_cg_init_A(Csr A, double *__restrict__ rhs /* l-vector: y in/out (x0), x read */,
                                                  const double2 *__restrict__ pair /* (rhs_x, A's) per column; unused without a warm start */,
                                                  const double *__restrict__ s, const double *__restrict__ Minv, double *__restrict__ r, double *__restrict__ z,
                                                  double *__restrict__ p, double rho, double tol_factor, Dims d,
                                                  double *part, int nb, Ctl *ctl, const double *gs) {
  ABIP_GATE_HALT(ctl);
  __shared__ double lds[2 * CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[3 * WAVES];
  double bn[1];
  const int rs[1] = {S_BN};
  get_scalars<1>(part, rs, nb, bn, sm, gs);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double tol = sqrt(bn[0]) * tol_factor; // indirect.c:406-409
    tol = fmax(tol, 1e-7);
    ctl->cg_tol = fmax(tol, 1e-9);         // indirect.c:418
    ctl->cg_it = 0;
    ctl->cg_done = 0;
  }
  const double *bx = rhs + d.MP;
  double acc2[3] = {0.0, 0.0, 0.0};
  if (s) {
    spmv_rows<2, SELL>(
        A, lds, lptr, sm, [&](int c, double a, double(&pr)[2]) { const double2 t = pair[c]; pr[0] = a * t.x; pr[1] = a * t.y; },
        [&](int i, double(&acc)[2]) {
          const double si = s[i];
          const double b = rhs[i] + acc[0];
          const double ri = b - (acc[1] + rho * si);
          const double zi = ri * Minv[i];
          rhs[i] = si; r[i] = ri; z[i] = zi; p[i] = zi;
          acc2[0] += ri * ri; acc2[1] += zi * ri;
          if (DIST) acc2[2] += zi * zi;
        });
  } else {
    spmv_rows<1, SELL>(
        A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * bx[c]; },
        [&](int i, double(&acc)[1]) {
          const double ri = rhs[i] + acc[0];
          const double zi = ri * Minv[i];
          rhs[i] = 0.0; r[i] = ri; z[i] = zi; p[i] = zi;
          acc2[0] += ri * ri; acc2[1] += zi * ri;
          if (DIST) acc2[2] += zi * zi;
        });
  }
  if (DIST) {
    const int ws[3] = {S_RR0, S_ZR0, S_ZZ};
    write_partials<3>(part, ws, acc2, sm);
  } else {
    const int ws[2] = {S_RR0, S_ZR0};
    double a2[2] = {acc2[0], acc2[1]};
    write_partials<2>(part, ws, a2, sm);
  }
}

// Convergence test shared by the first SpMV of an iteration and by the post-solve kernel.
// Every block evaluates it on the same partials, so all blocks agree (idempotent flag write).
__device__ __forceinline__ bool cg_converged(Ctl *ctl, const double *part, int nb, int max_its, double *sm, int &it, double &zr,
                                             const double *gs = nullptr) {
  it = ctl->cg_it;
  const int par = it & 1;
  double v[2];
  const int rs[2] = {S_RR0 + par, S_ZR0 + par};
  get_scalars<2>(part, rs, nb, v, sm, gs);
  zr = v[1];
  const double nr = sqrt(v[0]), tol = ctl->cg_tol;
  bool done = (it == 0) ? (nr < fmin(tol, 1e-18)) : (nr < tol); // indirect.c:359, 375
  if (it >= max_its) done = true;                               // indirect.c:368
  return done;
}

// tmp = A' p_new with p_new = z + beta p (indirect.c:386-387 folded into 216).  Gathering two m-vectors per non-zero
// costs a second pass through the texture path, so the identity A'(z + beta p) = A'z + beta (A'p) is used instead:
// tmp still holds A'p of the previous iteration (it is rebuilt from scratch, beta = 0, at every solve).
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_cg_spmv_At(Csr At, const double *__restrict__ z,
                                                   double *__restrict__ tmp, int max_its, double *part, int nb, Ctl *ctl, Stamp *st) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  stamp_begin(st);
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[2 * WAVES];
  double beta = 0.0;
  bool first = true, ran = false;
  // the convergence test on the previous update's partials runs while the first row block's stream is in flight
  auto pre = [&]() -> bool {
    int it; double zr;
    if (cg_converged(ctl, part, nb, max_its, sm, it, zr)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) ctl->cg_done = 1;
      return false;
    }
    const int par = it & 1;
    first = (it == 0);
    beta = first ? 0.0 : zr / ctl->zr_hist[par ^ 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->it_cur = it; ctl->beta_cur = beta; ctl->zr_cur = zr; ctl->zr_hist[par] = zr; }
    ran = true;
    return true;
  };
  spmv_rows<1, SELL>(
      At, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * z[c]; },
      [&](int row, double(&acc)[1]) { tmp[row] = first ? acc[0] : acc[0] + beta * tmp[row]; }, pre);
  if (ran) stamp_end(st); // a launch that found the PCG converged leaves t1 == 0: not counted
}

// p <- z + beta p ; Gp = A tmp + rho p ; S_PG <- p'Gp            (indirect.c:214-219, 371)
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_cg_spmv_A(Csr A, const double *__restrict__ tmp, const double *__restrict__ z,
                                                  double *__restrict__ p, double *__restrict__ Gp, double rho, double *part, const Ctl *ctl, Stamp *st) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  stamp_begin(st);
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  const double beta = ctl->beta_cur;
  double acc1[1] = {0.0};
  spmv_rows<1, SELL>(
      A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * tmp[c]; },
      [&](int i, double(&acc)[1]) {
        const double pn = z[i] + beta * p[i];
        const double gp = acc[0] + rho * pn;
        p[i] = pn; Gp[i] = gp;
        acc1[0] += pn * gp;
      });
  const int ws[1] = {S_PG};
  write_partials<1>(part, ws, acc1, sm);
  stamp_end(st);
}

// x += alpha p ; r -= alpha Gp ; z = M r ; S_RR, S_ZR (next parity)          (indirect.c:371-385)
template <bool DIST> // DIST (sharded): p'Gp = rho ||p||^2 + ||A'p||^2 from replicated data, plus the partials of z'z and z'p
__global__ __launch_bounds__(BS) void k_cg_update(double *__restrict__ x, double *__restrict__ r, double *__restrict__ z,
                                                  const double *__restrict__ p, const double *__restrict__ Gp,
                                                  const double *__restrict__ Minv, int m, double rho, double *part, int nb, Ctl *ctl, const double *gs) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  __shared__ double sm[4 * WAVES];
  double pg[1];
  if (DIST) {
    const int rs[1] = {S_TT};
    read_partials<1>(part, rs, nb, pg, sm); // ||A'p||^2: every rank holds the same A'p and reduces it on the same grid
    pg[0] += rho * ctl->pp_cur;
  } else {
    const int rs[1] = {S_PG};
    get_scalars<1>(part, rs, nb, pg, sm, gs);
  }
  const int it = ctl->it_cur;
  const double alpha = ctl->zr_cur / pg[0];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    const double pi = p[i];
    x[i] += alpha * pi;
    const double ri = r[i] - alpha * Gp[i];
    const double zi = ri * Minv[i];
    r[i] = ri; z[i] = zi;
    acc[0] += ri * ri; acc[1] += zi * ri;
    if (DIST) { acc[2] += zi * zi; acc[3] += zi * pi; }
  }
  const int par = (it + 1) & 1;
  // this kernel runs on a smaller grid than the SpMVs (dispatching NB workgroups costs more than its work): the
  // partial entries of the workgroups that do not exist are zeroed so that consumers can keep summing nb entries
  if (DIST) {
    const int ws[4] = {S_RR0 + par, S_ZR0 + par, S_ZZ, S_ZP};
    write_partials<4>(part, ws, acc, sm);
    if (threadIdx.x == 0)
      for (int e = blockIdx.x + gridDim.x; e < nb; e += gridDim.x)
        for (int q = 0; q < 4; ++q) part[ws[q] * MAXNB + e] = 0.0;
  } else {
    const int ws[2] = {S_RR0 + par, S_ZR0 + par};
    double a2[2] = {acc[0], acc[1]};
    write_partials<2>(part, ws, a2, sm);
    if (threadIdx.x == 0)
      for (int e = blockIdx.x + gridDim.x; e < nb; e += gridDim.x) { part[ws[0] * MAXNB + e] = 0.0; part[ws[1] * MAXNB + e] = 0.0; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->cg_it = it + 1;
}

// Post-solve: rhs_x <- A' rhs_y - rhs_x (indirect.c:419-420) and S_DH <- rhs[0:l-1)'h (abip.c:560).
// Runs only once the CG has converged; re-checks convergence itself because the last update of a
// chunk has no SpMV behind it.
// `aty` (or null): the product A'rhs_y itself is kept -- with v_y == 0 the iterate's new y block IS rhs_y (abip.c:731-734), so the stopping test's A'u_y
// and the next solve's warm-start product A's (indirect.c:345-350) are this very vector, row for row the same sums (solver.hip: aty_valid).
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_post_At(Csr At, double *__restrict__ rhs, const double *__restrict__ h, Dims d,
                                                int max_its, double *part, int nb, Ctl *ctl, double *__restrict__ aty) {
  ABIP_GATE_HALT(ctl);
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[2 * WAVES];
  if (!ctl->cg_done) {
    int it; double zr;
    if (!cg_converged(ctl, part, nb, max_its, sm, it, zr)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->cg_done = 1;
  }
  double *bx = rhs + d.MP;
  const double *hx = h + d.MP;
  double acc1[1] = {0.0};
  spmv_rows<1, SELL>(
      At, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * rhs[c]; },
      [&](int j, double(&acc)[1]) {
        const double v = acc[0] - bx[j];
        bx[j] = v;
        if (aty) aty[j] = acc[0];
        acc1[0] += v * hx[j];
      });
  for (int i = blockIdx.x * BS + threadIdx.x; i < d.m; i += gridDim.x * BS) acc1[0] += rhs[i] * h[i];
  const int ws[1] = {S_DH};
  write_partials<1>(part, ws, acc1, sm);
}

// Direct back-end: only the dot with h is left to do after the triangular solves.
__global__ __launch_bounds__(BS) void k_post_dot(const double *__restrict__ rhs, const double *__restrict__ h, Dims d, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) acc[0] += rhs[i] * h[i];
  for (int j = t0; j < d.n; j += stride) acc[0] += rhs[d.MP + j] * h[d.MP + j];
  const int ws[1] = {S_DH};
  write_partials<1>(part, ws, acc, sm);
}

// ---------------------------------------------------------------------------------------------
// k_admm_update: everything element-wise between the KKT solve and the stopping test.
// ---------------------------------------------------------------------------------------------
struct UpdArgs {
  double *u, *v, *ut;
  double *u_avg, *v_avg, *u_sum, *v_sum, *u_avgc, *v_avgc;
  const double *g, *b, *c;
  double alpha, mu_over_beta, rho, dom;
  double xw;       // weight of the replicated (x, tau) entries in the reductions: 1 on a single GPU / rank 0, 0 on the other ranks
  const double *gs; // all-reduced scalars (multi-GPU) or null
  int half_update; // abip.c:2143-2149
  int fuse_avg;    // 1: also do compute_avg + the statistics (no restart this iteration)
  int avg_stats;   // 1: (j+1)%10==0 -> statistics of the averaged iterate too (abip.c:2000)
};

// statistics of one (u, v) element pair; tail = the tau/kappa entry (in the norms, not in the dots)
struct Stat { double wg, nu, nv, cx, by, nua, nva, cxa, bya; };

__device__ __forceinline__ void avg_and_stats_y(const UpdArgs &a, int i, double un, double vn, Stat &st) {
  const double us = a.u_sum[i] + un, vs = a.v_sum[i] + vn; // compute_avg, abip.c:649-656
  a.u_sum[i] = us; a.v_sum[i] = vs;
  const double ua = us / a.dom, va = vs / a.dom;
  a.u_avgc[i] = ua; a.v_avgc[i] = va;
  st.wg += a.rho * (un + vn) * a.g[i];
  st.nu += un * un; st.nv += vn * vn; st.by += a.b[i] * un;
  if (a.avg_stats) { st.nua += ua * ua; st.nva += va * va; st.bya += a.b[i] * ua; }
}
__device__ __forceinline__ void avg_and_stats_x(const UpdArgs &a, int q /* MP + j */, int j, bool tail, double un, double vn, Stat &st) {
  const double us = a.u_sum[q] + un, vs = a.v_sum[q] + vn;
  a.u_sum[q] = us; a.v_sum[q] = vs;
  const double ua = us / a.dom, va = vs / a.dom;
  a.u_avgc[q] = ua; a.v_avgc[q] = va;
  st.nu += a.xw * (un * un); st.nv += a.xw * (vn * vn);
  if (a.avg_stats) { st.nua += a.xw * (ua * ua); st.nva += a.xw * (va * va); }
  if (!tail) {
    st.wg += a.xw * ((un + vn) * a.g[q]);
    st.cx += a.xw * (a.c[j] * un);
    if (a.avg_stats) st.cxa += a.xw * (a.c[j] * ua);
  }
}
__device__ __forceinline__ void prox_x(const UpdArgs &a, int q, double utq, double &un, double &vn) {
  const double uo = a.u[q], vo = a.v[q];
  if (!a.half_update) {
    const double t = a.alpha * utq + (1.0 - a.alpha) * uo - vo; // abip.c:738
    const double hlf = t / 2;
    un = hlf + sqrt(hlf * hlf + a.mu_over_beta);                // abip.c:743-744: the barrier prox ("orthant clip")
    vn = vo + (un - a.alpha * utq - (1.0 - a.alpha) * uo);      // abip.c:580
  } else {
    double vh = vo + 0.5 * (uo - utq);                          // abip.c:675
    const double hlf = (utq - vh) / 2;                          // abip.c:695,700
    un = hlf + sqrt(hlf * hlf + a.mu_over_beta);
    vn = vh + (un - utq);                                       // abip.c:707
  }
}

__global__ __launch_bounds__(BS) void k_admm_update(UpdArgs a, Dims d, double *part, int nb, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double sm[9 * WAVES];
  double dh[1];
  const int rs[1] = {S_DH};
  get_scalars<1>(part, rs, nb, dh, sm, a.gs);
  Stat st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) {
    double un, vn;
    const double uti = a.ut[i];
    if (!a.half_update) { vn = a.v[i]; un = uti - vn; }                                    // abip.c:731-734
    else { double vh = a.v[i] + 0.5 * (a.u[i] - uti); un = uti - vh; vn = vh + (un - uti); } // abip.c:675,695,707
    a.u[i] = un; a.v[i] = vn;
    a.u_avg[i] += un; a.v_avg[i] += vn;                                                    // abip.c:602-606
    if (a.fuse_avg) avg_and_stats_y(a, i, un, vn, st);
  }
  for (int j = t0; j < d.n; j += stride) {
    const int q = d.MP + j;
    double un, vn;
    prox_x(a, q, a.ut[q], un, vn);
    a.u[q] = un; a.v[q] = vn;
    a.u_avg[q] += un; a.v_avg[q] += vn;
    if (a.fuse_avg) avg_and_stats_x(a, q, j, false, un, vn, st);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { // the tau / kappa entry
    const int q = d.MP + d.n;
    const double utq = a.ut[q] + dh[0];      // abip.c:560
    a.ut[q] = utq;
    double un, vn;
    prox_x(a, q, utq, un, vn);
    a.u[q] = un; a.v[q] = vn;
    a.u_avg[q] += un; a.v_avg[q] += vn;
    if (a.fuse_avg) avg_and_stats_x(a, q, d.n, true, un, vn, st);
  }
  if (a.fuse_avg) {
    double vals[9] = {st.wg, st.nu, st.nv, st.cx, st.by, st.nua, st.nva, st.cxa, st.bya};
    const int ws[9] = {S_WG, S_NU, S_NV, S_CX, S_BY, S_NUA, S_NVA, S_CXA, S_BYA};
    write_partials<9>(part, ws, vals, sm);
  }
}

// restart from the running mean (abip.c:613-627): u,v <- u_avg/fre, v_avg/fre ; sums zeroed
__global__ __launch_bounds__(BS) void k_restart_apply(double *u, double *v, double *u_avg, double *v_avg, double fre, int len) {
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) {
    const double a = u_avg[i] / fre, b = v_avg[i] / fre;
    u[i] = a; v[i] = b; u_avg[i] = 0.0; v_avg[i] = 0.0;
  }
}

// compute_avg + statistics as a separate pass (used on restart iterations, where (u,v) change after the prox)
__global__ __launch_bounds__(BS) void k_avg_stats(UpdArgs a, Dims d, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double sm[9 * WAVES];
  Stat st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) avg_and_stats_y(a, i, a.u[i], a.v[i], st);
  for (int j = t0; j < d.n; j += stride) avg_and_stats_x(a, d.MP + j, j, false, a.u[d.MP + j], a.v[d.MP + j], st);
  if (blockIdx.x == 0 && threadIdx.x == 0) avg_and_stats_x(a, d.MP + d.n, d.n, true, a.u[d.MP + d.n], a.v[d.MP + d.n], st);
  double vals[9] = {st.wg, st.nu, st.nv, st.cx, st.by, st.nua, st.nva, st.cxa, st.bya};
  const int ws[9] = {S_WG, S_NU, S_NV, S_CX, S_BY, S_NUA, S_NVA, S_CXA, S_BYA};
  write_partials<9>(part, ws, vals, sm);
}

// ---------------------------------------------------------------------------------------------
// Residual SpMV pair: the inner stopping metric (abip.c:1976-1992) and the D/E-weighted outer
// residuals (abip.c:407-413, 443-449) from ONE pass over each matrix; pr/dr are never stored.
// ---------------------------------------------------------------------------------------------
// bodies: (vb, vgrid) = this workgroup's index / count among those working on the product
template <bool SELL>
__device__ __forceinline__ void d_q_A(const Csr &A, const double *__restrict__ uu /* l-vector */, const double *__restrict__ b,
                                      const double *__restrict__ wD /* D_i/(sc_b*scale) or null */, const Dims &d, int slot0, double *part,
                                      double *lds, int *lptr, double *sm, int vb, int vgrid) {
  const double *x = uu + d.MP;
  const double tau = uu[d.MP + d.n];
  double acc3[3] = {0.0, 0.0, 0.0};
  spmv_rows<1, SELL>(
      A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
      [&](int i, double(&acc)[1]) {
        const double pri = acc[0], e = pri - b[i] * tau;
        double sc = wD ? wD[i] : 1.0;
        sc = sc * sc;
        acc3[0] += e * e; acc3[1] += (e * e) * sc; acc3[2] += (pri * pri) * sc;
      },
      [] { return true; }, vb, vgrid);
  const int ws[3] = {slot0, slot0 + 1, slot0 + 2};
  write_partials<3>(part, ws, acc3, sm, vb);
}
template <bool SELL>
__device__ __forceinline__ void d_q_At(const Csr &At, const double *__restrict__ uu, const double *__restrict__ vv, const double *__restrict__ c,
                                       const double *__restrict__ wE /* E_j/(sc_c*scale) or null */, const Dims &d, int slot0, double *part,
                                       double *lds, int *lptr, double *sm, int vb, int vgrid) {
  const double *s = vv + d.MP;
  const double tau = uu[d.MP + d.n];
  double acc3[3] = {0.0, 0.0, 0.0};
  spmv_rows<1, SELL>(
      At, lds, lptr, sm, [&](int cidx, double a, double(&pr)[1]) { pr[0] = a * uu[cidx]; },
      [&](int j, double(&acc)[1]) {
        const double drj = acc[0] + s[j], e = drj - c[j] * tau;
        double sc = wE ? wE[j] : 1.0;
        sc = sc * sc;
        acc3[0] += e * e; acc3[1] += (e * e) * sc; acc3[2] += (drj * drj) * sc;
      },
      [] { return true; }, vb, vgrid);
  const int ws[3] = {slot0, slot0 + 1, slot0 + 2};
  write_partials<3>(part, ws, acc3, sm, vb);
}
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_q_A(Csr A, const double *__restrict__ uu, const double *__restrict__ b, const double *__restrict__ wD, Dims d, int slot0,
                                            double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[3 * WAVES];
  d_q_A<SELL>(A, uu, b, wD, d, slot0, part, lds, lptr, sm, (int)blockIdx.x, (int)gridDim.x);
}
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_q_At(Csr At, const double *__restrict__ uu, const double *__restrict__ vv, const double *__restrict__ c,
                                             const double *__restrict__ wE, Dims d, int slot0, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[3 * WAVES];
  d_q_At<SELL>(At, uu, vv, c, wE, d, slot0, part, lds, lptr, sm, (int)blockIdx.x, (int)gridDim.x);
}
// both residual products of the stopping test in one launch: workgroups [0, nbA) take A u_x, the rest A'u_y (independent products,
// each with its own nbA partial entries per slot)
template <bool SELLA, bool SELLT>
__global__ __launch_bounds__(BS, (SELLA || SELLT) ? 6 : 8) void k_q_both(Csr A, Csr At, const double *__restrict__ uu, const double *__restrict__ vv, const double *__restrict__ b,
                                               const double *__restrict__ c, const double *__restrict__ wD, const double *__restrict__ wE, Dims d,
                                               int slotA, int slotAt, int nbA, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[3 * WAVES];
  if ((int)blockIdx.x < nbA) d_q_A<SELLA>(A, uu, b, wD, d, slotA, part, lds, lptr, sm, (int)blockIdx.x, nbA);
  else d_q_At<SELLT>(At, uu, vv, c, wE, d, slotAt, part, lds, lptr, sm, (int)blockIdx.x - nbA, (int)gridDim.x - nbA);
}

// the same launch when A'u_y is at hand (k_post_At's `aty`): workgroups [0, nbA) take A u_x as before, the rest walk the n dual residuals element-wise --
// d_q_At's row epilogue on the stored product: the same per-element numbers, but DEALT TO THE WORKGROUPS DIFFERENTLY (a strided walk here, row blocks there), so the
// three folded sums S_QD, S_RD, S_NATY agree with ABIP_HIP_ATY=0 to rounding, not bit for bit.  The iterate does not depend on them; the exit test and
// the final check do, so at an exact tie the two settings could stop an iteration apart -- none of the fixtures has such a tie (the tests hold both to the same
// counts), and each setting by itself is deterministic.
template <bool SELLA>
__global__ __launch_bounds__(BS, SELLA ? 6 : 8) void k_q_A_aty(Csr A, const double *__restrict__ aty, const double *__restrict__ uu, const double *__restrict__ vv, const double *__restrict__ b,
                                                const double *__restrict__ c, const double *__restrict__ wD, const double *__restrict__ wE, Dims d,
                                                int slotA, int slotAt, int nbA, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[3 * WAVES];
  if ((int)blockIdx.x < nbA) { d_q_A<SELLA>(A, uu, b, wD, d, slotA, part, lds, lptr, sm, (int)blockIdx.x, nbA); return; }
  const int vb = (int)blockIdx.x - nbA, vgrid = (int)gridDim.x - nbA;
  const double *s = vv + d.MP;
  const double tau = uu[d.MP + d.n];
  double acc3[3] = {0.0, 0.0, 0.0};
  for (int j = vb * BS + threadIdx.x; j < d.n; j += vgrid * BS) {
    const double drj = aty[j] + s[j], e = drj - c[j] * tau;
    double sc = wE ? wE[j] : 1.0;
    sc = sc * sc;
    acc3[0] += e * e; acc3[1] += (e * e) * sc; acc3[2] += (drj * drj) * sc;
  }
  const int ws[3] = {slotAt, slotAt + 1, slotAt + 2};
  write_partials<3>(part, ws, acc3, sm, vb);
}

struct XcdFinal { // the final_check branch of abip.c:2190-2213 evaluated on the device (calc_residuals + has_converged on the finalised sums)
  int on, pfeasopt, ipm_pos;
  double eps, den, nm_b, nm_c;
  long k0, max_admm;
};

// the finalised sums of one iterate as lp_scalars.h reads them (o = the out[] array; ac: the averaged iterate's)
__host__ __device__ inline LpSums x_sums(const double *o, int ac) {
  LpSums s;
  s.ut = ac ? o[82] : o[80]; s.vt = ac ? o[83] : o[81];
  s.rp = ac ? o[S_RPA] : o[S_RP]; s.nax = ac ? o[S_NAXA] : o[S_NAX];
  s.rd = ac ? o[S_RDA] : o[S_RD]; s.naty = ac ? o[S_NATYA] : o[S_NATY];
  s.by = ac ? o[S_BYA] : o[S_BY]; s.cx = ac ? o[S_CXA] : o[S_CX];
  return s;
}
// calc_residuals (abip.c:458-535) + has_converged (1613-1641) on finalised sums; ac = avg_criterion of this iteration
__device__ __forceinline__ int x_converged(const double *o, int ac, const XcdFinal &f, long ipm_iter, long k) {
  LpResid r;
  lp_residuals(x_sums(o, ac), f.den, f.nm_b, f.nm_c, r);
  return lp_converged(r, f.eps, f.pfeasopt, ipm_iter, k) != 0;
}

// One block: fold the listed slots into ctl->out[slot] and append the tau/kappa entries the host needs.
struct FinArgs {
  int nslots; int slots[40]; const double *u, *v, *ua, *va; const double *gs; /* non-null: take the already all-reduced values */
  // decide = 1 (the finalize that closes an ADMM iteration): evaluate the inner-loop exit test on the device and raise ctl->halt
  // when it holds, so that iterations enqueued behind this one fall through; the kernel itself is then gated on halt too
  int decide = 0, avg_stats = 0;
  double thr = 0.0, sentinel = 0.0; // gamma * mu; Qres_avg when no averaged statistics were taken (= max_admm_iters, abip.c:1957)
};
// (a device function so that the one-workgroup solve kernel of the NEXT iteration can run it as its prologue, dev_ldl.h / LpSolveFuse)
__device__ __forceinline__ void d_finalize(const FinArgs &f, const Dims &d, const double *part, int nb, Ctl *ctl) {
  // one wavefront per slot, all loads of a lane in flight at once
  if (f.decide && (ctl->halt || !ctl->cg_done)) return; // cg_done: permanently set for the direct back-end
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PER = MAXNB / 64;
  const int nwaves = blockDim.x >> 6; // launched with 1024 threads: one round of loads covers 16 slots
  for (int s = wave; s < f.nslots; s += nwaves) {
    const int slot = f.slots[s];
    if (f.gs) { if (lane == 0) ctl->out[slot] = f.gs[slot]; continue; }
    double t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = lane + u * 64;
      t[u] = (i < nb) ? part[slot * MAXNB + i] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < PER; ++u) acc += t[u];
    acc = wave_sum(acc);
    if (lane == 0) ctl->out[slot] = acc;
  }
  if (threadIdx.x == 0) {
    const int q = d.MP + d.n;
    ctl->out[80] = f.u[q]; ctl->out[81] = f.v[q];
    ctl->out[82] = f.ua ? f.ua[q] : 0.0; ctl->out[83] = f.va ? f.va[q] : 0.0;
  }
  if (!f.decide) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double *o = ctl->out;
    double Qres = o[S_QP] + o[S_QD];
    const double gap = o[S_BY] - o[S_CX] - o[81];
    Qres = __dadd_rn(Qres, __dmul_rn(gap, gap)); // un-fused, like the host code this mirrors: ties between the two candidates must stay ties
    const double norm = 1 + sqrt(o[S_NU] + o[S_NV]);
    double Qres_avg = f.sentinel, norm_avg = 1;
    if (f.avg_stats) {
      Qres_avg = o[S_QPA] + o[S_QDA];
      const double gap_a = o[S_BYA] - o[S_CXA] - o[83];
      Qres_avg = __dadd_rn(Qres_avg, __dmul_rn(gap_a, gap_a));
      norm_avg = 1 + sqrt(o[S_NUA] + o[S_NVA]);
    }
    const double ma = sqrt(Qres_avg) / norm_avg, mc = sqrt(Qres) / norm;
    const int ac = ma < mc ? 1 : 0;
    const double metric = ac ? ma : mc;
    ctl->metric = metric; ctl->avg_crit = ac; ctl->it_count = ctl->it_count + 1;
    if (metric < f.thr) ctl->halt = 1;
  }
}
__global__ __launch_bounds__(1024) void k_finalize(FinArgs f, Dims d, const double *part, int nb, Ctl *ctl) { d_finalize(f, d, part, nb, ctl); }

// The finalize of a STREAMED iteration (solver.hip: admm_stream_pcg -- the host enqueues iteration j + 1 before it has read the verdict of iteration j, so nothing
// the next iteration depends on may be left to the host):
//   * the PCG had not converged inside the launches enqueued for it: halt = 2 ("stalled") -- everything enqueued behind falls through, the host adds PCG iterations
//     and enqueues the rest of this iteration again;
//   * the exit test of abip.c:2173 holds: halt = 1 (d_finalize);
//   * final_check (abip.c:2190-2213): calc_residuals + has_converged on the finalised sums, the iteration limits: halt = 3 -- the host extracts the solution;
//   * in every case the control block is copied to a pinned host mirror (no copy launch between two iterations; the host waits for an event behind this kernel).
struct FinStream { Ctl *mirror; XcdFinal fc; long ipm_iter; int ipm_last; /* i + 1 >= max_ipm_iters */ };
__global__ __launch_bounds__(1024) void k_finalize_stream(FinArgs f, FinStream fs, Dims d, const double *part, int nb, Ctl *ctl) {
  const int halt0 = ctl->halt, done0 = ctl->cg_done, it0 = ctl->it_count;
  __syncthreads(); // (everybody has read the state thread 0 may overwrite)
  if (!halt0 && !done0) { if (threadIdx.x == 0) ctl->halt = 2; }
  else d_finalize(f, d, part, nb, ctl);
  __syncthreads();
  if (threadIdx.x == 0 && fs.fc.on && ctl->it_count != it0 && !ctl->halt) {
    const long k = fs.fc.k0 + 1; // admm_iter as abip.c:2190 sees it
    if (x_converged(ctl->out, ctl->avg_crit, fs.fc, fs.ipm_iter, k) || k + 1 >= fs.fc.max_admm || fs.ipm_last) ctl->halt = 3;
  }
  __syncthreads();
  if (fs.mirror) {
    const double *src = reinterpret_cast<const double *>(ctl);
    double *dst = reinterpret_cast<double *>(fs.mirror);
    for (int q = threadIdx.x; q < (int)(sizeof(Ctl) / sizeof(double)); q += blockDim.x) dst[q] = src[q];
  }
}

// ---------------------------------------------------------------------------------------------
// outer-iteration element-wise kernels
// ---------------------------------------------------------------------------------------------
// reinitialize_vars, abip.c:996-1075 (x block and tau entry: indices m..l-1)
__global__ __launch_bounds__(BS) void k_reinit(double *u, double *v, double sigma, int indx, Dims d) {
  const int cnt = d.n + 1;
  const double sq = (indx == 1) ? sqrt(sigma) : sqrt(1.0 / sigma);
  for (int j = blockIdx.x * BS + threadIdx.x; j < cnt; j += gridDim.x * BS) {
    const int q = d.MP + j;
    if (indx == 0) { if (u[q] > v[q]) v[q] = sigma * v[q]; else u[q] = sigma * u[q]; }
    else { u[q] = sq * u[q]; v[q] = sq * v[q]; }
  }
}
// LOQO statistics, abip.c:962-965: sum and min of u_i v_i over i >= m.  One block per partial; the min is
// folded by the host from the per-block minima (exact, order-free).
__global__ __launch_bounds__(BS) void k_xs(const double *u, const double *v, Dims d, double *part) {
  __shared__ double sm[WAVES];
  __shared__ double smin[WAVES];
  const int cnt = d.n + 1;
  double acc[1] = {0.0};
  double mn = 1e+10;
  for (int j = blockIdx.x * BS + threadIdx.x; j < cnt; j += gridDim.x * BS) {
    const double x = u[d.MP + j] * v[d.MP + j];
    acc[0] += x; mn = fmin(mn, x);
  }
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
  if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = mn;
  const int ws[1] = {S_XS};
  write_partials<1>(part, ws, acc, sm);
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = smin[0];
    for (int w = 1; w < WAVES; ++w) t = fmin(t, smin[w]);
    part[S_XMIN * MAXNB + blockIdx.x] = t;
  }
}
__global__ void k_min_fold(const double *part, int nb, Ctl *ctl) { // single thread: tiny
  double t = 1e+10;
  for (int i = 0; i < nb; ++i) t = fmin(t, part[S_XMIN * MAXNB + i]);
  ctl->out[S_XMIN] = t;
}
// x block of an l-vector *= -1 (abip.c:1923)
__global__ __launch_bounds__(BS) void k_neg_x(double *g, Dims d) {
  for (int j = blockIdx.x * BS + threadIdx.x; j < d.n; j += gridDim.x * BS) g[d.MP + j] = -g[d.MP + j];
}
__global__ __launch_bounds__(BS) void k_dot_full(const double *a, const double *b, Dims d, int slot, double *part, double xw) { // over l-1 entries
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) acc[0] += a[i] * b[i];
  for (int j = t0; j < d.n; j += stride) acc[0] += xw * (a[d.MP + j] * b[d.MP + j]);
  const int ws[1] = {slot};
  write_partials<1>(part, ws, acc, sm);
}
// half_update clean-up on an inner break (abip.c:2175-2185)
__global__ __launch_bounds__(BS) void k_clip_v(double *v, Dims d) {
  const int len = d.MP + d.n + 1;
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) {
    const bool live = (i < d.m) || (i >= d.MP);
    if (live && v[i] < 0) v[i] = 1e-6;
  }
}

// ---------------------------------------------------------------------------------------------
// Barzilai-Borwein look-ahead (adaptive.c:101-178)
// ---------------------------------------------------------------------------------------------
// one ADMM step on scratch vectors: (ut, u_prev, v_prev) -> (u, v) with penalty beta_prev
// ctl != null (the streamed search, solver.hip: adaptive_search_stream): the penalty is mu / ctl->bb_prev (the device keeps beta_prev), the step runs only behind a
// solve that has converged -- else it raises halt = 2 ("stalled": the host adds PCG iterations and enqueues the rest again) -- and leaves `which` in bb_stage.
__global__ __launch_bounds__(BS) void k_adapt_step(const double *ut_in, double *ut_tail_fix, const double *__restrict__ up,
                                                   const double *__restrict__ vp, double *__restrict__ u, double *__restrict__ v,
                                                   double alpha, double mu_over_beta, Dims d, const double *part, int nb, const double *gs,
                                                   Ctl *ctl, double mu, int which) {
  if (ctl) {
    ABIP_GATE_HALT(ctl);
    if (!ctl->cg_done) { if (blockIdx.x == 0 && threadIdx.x == 0) ctl->halt = 2; return; }
    mu_over_beta = mu / ctl->bb_prev;
  }
  __shared__ double sm[WAVES];
  double dh[1];
  const int rs[1] = {S_DH};
  get_scalars<1>(part, rs, nb, dh, sm, gs);
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) u[i] = ut_in[i] - vp[i]; // adaptive.c:101-104 (v[0:m) is left untouched, :118-121)
  for (int j = t0; j <= d.n; j += stride) {
    const int q = d.MP + j;
    double utq = ut_in[q];
    if (j == d.n) { utq += dh[0]; ut_tail_fix[q] = utq; }           // adaptive.c:99
    const double t = alpha * utq + (1 - alpha) * up[q] - vp[q];
    const double hlf = t / 2;
    const double un = hlf + sqrt(hlf * hlf + mu_over_beta);
    u[q] = un;
    v[q] = vp[q] + (un - alpha * utq - (1 - alpha) * up[q]);
  }
  if (ctl && blockIdx.x == 0 && threadIdx.x == 0) { ctl->bb_stage = which; ctl->bb_cg[which - 1] = ctl->cg_it; ctl->bb_cg_total += ctl->cg_it; }
}
// the five inner products of the difference vectors, formed on the fly (adaptive.c:154-174)
__global__ __launch_bounds__(BS) void k_adapt_dots(const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ un,
                                                   const double *__restrict__ vn, const double *__restrict__ vp, double alpha, Dims d, double *part,
                                                   double xw) {
  __shared__ double sm[5 * WAVES];
  double a[5] = {0, 0, 0, 0, 0};
  const int len = d.MP + d.n + 1;
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) {
    if (i >= d.m && i < d.MP) continue; // padding
    const double wgt = (i < d.m) ? 1.0 : xw;
    const double dut = 2.0 * v[i] + un[i] - u[i] - vn[i] - vp[i];
    const double du = u[i] - un[i];
    const double dv = (un[i] - u[i]) * (alpha - 1.0) + vn[i] - v[i];
    a[0] += wgt * (dut * dut); a[1] += wgt * (dut * dv); a[2] += wgt * (du * du); a[3] += wgt * (dv * dv); a[4] += wgt * (du * dv);
  }
  const int ws[5] = {S_A0, S_A1, S_A2, S_A3, S_A4};
  write_partials<5>(part, ws, a, sm);
}
// v_prev[x,tau] = (mu/beta)/u_prev  (adaptive.c:238-241); y block copied by the caller
__global__ __launch_bounds__(BS) void k_adapt_vprev(double *vp, const double *up, double mu_over_beta, Dims d) {
  for (int j = blockIdx.x * BS + threadIdx.x; j <= d.n; j += gridDim.x * BS) vp[d.MP + j] = mu_over_beta / up[d.MP + j];
}

// The decision of one look-ahead on the device (adaptive.c:170-247; lp_scalars.h: lp_bb_beta -- the host's arithmetic, contraction off): the five sums, the spectral
// step, beta_prev for the next look-ahead; halt = 4 once the search is over (stop, or the look-back is used up).  The control block is mirrored in every case (a
// stalled look-ahead reaches this kernel with halt = 2: the host reads that from the mirror).
__global__ __launch_bounds__(1024) void k_adapt_decide(const double *part, int nb, Ctl *ctl, double eps_cor, double eps_pen, int lookback, Ctl *mirror, int skip_ok) {
  const int halt0 = ctl->halt;
  __syncthreads();
  if (!halt0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PER = MAXNB / 64;
    if (wave < 5) { // (d_finalize's fold: one wavefront per slot, lane order)
      const int slot = S_A0 + wave;
      double t[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) { const int i = lane + u * 64; t[u] = (i < nb) ? part[slot * MAXNB + i] : 0.0; }
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < PER; ++u) acc += t[u];
      acc = wave_sum(acc);
      if (lane == 0) ctl->out[slot] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double *o = ctl->out;
      double beta = 0.0;
      const int act = lp_bb_beta(o[S_A0], o[S_A1], o[S_A2], o[S_A3], o[S_A4], eps_cor, eps_pen, ctl->bb_prev, beta);
      ctl->bb_act = act; ctl->bb_beta = beta;
      if (act == 1) ctl->bb_prev = beta;
      ctl->bb_it = ctl->bb_it + 1; ctl->bb_stage = 0; ctl->bb_skip = 0;
      if (act == 0 || ctl->bb_it >= lookback) ctl->halt = 4;
      else if (act == 2 && skip_ok) { ctl->bb_skip = 1; ctl->halt = 5; } // the next look-ahead's first half is this one's second: see k_adapt_next
    }
  }
  __syncthreads();
  if (mirror) {
    const double *src = reinterpret_cast<const double *>(ctl);
    double *dst = reinterpret_cast<double *>(mirror);
    for (int q = threadIdx.x; q < (int)(sizeof(Ctl) / sizeof(double)); q += blockDim.x) dst[q] = src[q];
  }
}
// Behind the first step of a look-ahead: if that first half was skipped (halt 5: its results were moved over from the previous look-ahead, k_adapt_next) the
// second half may run -- halt is lowered here, by a launch of its own, so that no kernel both reads and writes it -- and the skipped solve is counted as the
// reference counts it (it solves again and takes the same number of PCG iterations).
__global__ void k_adapt_resume(Ctl *ctl) {
  if (ctl->halt == 5) { ctl->halt = 0; ctl->bb_stage = 1; ctl->bb_cg[0] = ctl->bb_cg[1]; ctl->bb_cg_total += ctl->bb_cg[1]; ctl->bb_cg_skipped += ctl->bb_cg[1]; }
}
// (u_prev, v_prev) for the next look-ahead as the last decision wants them (adaptive.c:230-247): u_prev = u; v_prev = v, its (x, tau) part rebuilt as
// (mu / beta_prev) / u_prev when the penalty changed.
// When it did NOT change (act 2: beta = beta_prev, by far the most frequent outcome -- profiles/r05h_c4_bb_trace.txt), the next look-ahead's first half is this
// one's second half over again: the same right-hand side from (u, v), the same warm start u, the same tolerance -- the reference runs that solve a second time
// and gets the same bits.  Here its results are moved instead -- u_t <- u_t,next, u <- u_next, v[x, tau] <- v_next[x, tau] (the y block of v is never written:
// adaptive.c:118-121): k_adapt_decide has raised bb_skip with halt = 5, every gated kernel of the next unit's first half falls through, and k_adapt_resume
// behind them lowers halt again.
__global__ __launch_bounds__(BS) void k_adapt_next(double *__restrict__ up, double *__restrict__ vp, double *__restrict__ u, double *__restrict__ v,
                                                   double *__restrict__ ut, const double *__restrict__ utn, const double *__restrict__ un, const double *__restrict__ vn,
                                                   double mu, Dims d, const Ctl *ctl) {
  if (ctl->halt && ctl->halt != 5) return;
  const int act = ctl->bb_act;
  const double mob = mu / ctl->bb_prev;
  const bool skip = ctl->bb_skip != 0;
  const int len = d.MP + d.n + 1;
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) {
    if (i >= d.m && i < d.MP) continue; // padding
    const double ui = u[i], vi = v[i];
    up[i] = ui;
    vp[i] = (act == 1 && i >= d.MP) ? mob / ui : vi;
    if (skip) {
      ut[i] = utn[i];
      u[i] = un[i];
      if (i >= d.MP) v[i] = vn[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU (rows of A sharded over ranks).  m-space is local, n-space replicated; a rank's
// A_g' y_g is a PARTIAL n-vector that the host all-reduces (RCCL) together with the packed scalars in `gs`.
// ---------------------------------------------------------------------------------------------
// partials -> local scalars gs[slot] (one wavefront per slot); the all-reduce then sums gs over the ranks
// out = M x (overwrites), gated: mode 0 always, 1 while the PCG runs, 2 once it has converged
struct FoldArgs { int nslots; int slots[40]; };
// partials -> one number per slot in gs (the packed scalars that ride along with an all-reduce); one wavefront per slot
__device__ __forceinline__ void fold_slots(const FoldArgs &f, const double *part, int nb, double *gs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PER = MAXNB / 64, GRP = 8; // eight loads in flight per lane at a time: the fold rides inside an SpMV kernel and must not set its register count
  for (int s = wave; s < f.nslots; s += WAVES) {
    const int slot = f.slots[s];
    double acc = 0.0;
#pragma unroll 1
    for (int u0 = 0; u0 < PER; u0 += GRP) {
      double t[GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int i = lane + (u0 + u) * 64;
        t[u] = (i < nb) ? part[slot * MAXNB + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GRP; ++u) acc += t[u];
    }
    acc = wave_sum(acc);
    if (lane == 0) gs[slot] = acc;
  }
}
__global__ __launch_bounds__(BS) void k_fold(FoldArgs f, const double *part, int nb, double *gs) { fold_slots(f, part, nb, gs); }
// out = M x on the rows of this rank; FOLD: the last workgroup also folds the partials of the previous kernel into gs.
// pp.on (the peer-mapped transport, dev_peer.h): `out` is one rank's contribution to an all-reduce, so every row goes straight into the inbox of the rank that
// reduces it (step 1 of the exchange: no local vector, no copy launch) -- and so do the folded scalars, which ride behind the vector at index `scal0`.
template <bool FOLD, bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_spmv_set_t(Csr M, const double *__restrict__ x, double *__restrict__ out, int mode, const Ctl *ctl,
                                                   FoldArgs f, const double *part, int nb, double *gs, Stamp *st, PeerPush pp, long scal0) {
  ABIP_GATE_HALT(ctl);
  if (FOLD && blockIdx.x == gridDim.x - 1) { // (before the gates: the convergence test needs the sums)
    fold_slots(f, part, nb, gs);
    if (pp.on && !(mode == 1 && ctl->cg_done) && !(mode == 2 && !ctl->cg_done)) {
      __syncthreads();
      for (int q = threadIdx.x; q < f.nslots; q += BS) d_peer_put(pp, scal0 + f.slots[q], gs[f.slots[q]]);
    }
  }
  if (mode == 1 && ctl->cg_done) return;
  if (mode == 2 && !ctl->cg_done) return;
  stamp_begin(st);
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  if (pp.on) {
    spmv_rows<1, SELL>(
        M, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
        [&](int row, double(&acc)[1]) { d_peer_put(pp, (long)row, acc[0]); });
    d_peer_push_done(pp);
  } else {
    spmv_rows<1, SELL>(
        M, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
        [&](int row, double(&acc)[1]) { out[row] = acc[0]; });
  }
  stamp_end(st);
}
// steps 2 + 3 of the exchange behind such a producer, under the producer's own gates (every rank takes them alike: the control state is replicated)
__global__ __launch_bounds__(256) void k_peer_reduce_gather_gated(PeerCtx c, double *buf, long count, unsigned long long epoch, int mode, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (mode == 1 && ctl->cg_done) return;
  if (mode == 2 && !ctl->cg_done) return;
  d_peer_reduce_gather(c, buf, count, epoch);
}
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_spmv_set(Csr M, const double *__restrict__ x, double *__restrict__ out, int mode, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (mode == 1 && ctl->cg_done) return;
  if (mode == 2 && !ctl->cg_done) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  spmv_rows<1, SELL>(
      M, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
      [&](int row, double(&acc)[1]) { out[row] = acc[0]; });
}
// After the all-reduce of [A_g'z_g | rr | zr]: the convergence decision and beta (identical on every rank) and, when
// `vec`, tmp = T + beta*tmp  (k_cg_spmv_At's second half).  vec == 0: decision only (end of a chunk).
__global__ __launch_bounds__(BS) void k_dist_cg_step(const double *__restrict__ T, double *__restrict__ tmp, int n, int max_its, int vec,
                                                     const double *gs, double *part, Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  int it; double zr;
  if (cg_converged(ctl, nullptr, 0, max_its, nullptr, it, zr, gs)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->cg_done = 1;
    return;
  }
  if (!vec) return;
  __shared__ double sm[WAVES];
  const int par = it & 1;
  const double beta = (it == 0) ? 0.0 : zr / ctl->zr_hist[par ^ 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->it_cur = it; ctl->beta_cur = beta; ctl->zr_cur = zr; ctl->zr_hist[par] = zr;
    // ||p||^2 of the direction p = z + beta p_old about to be formed (z'z and z'p_old arrived summed with this all-reduce)
    ctl->pp_cur = (it == 0) ? gs[S_ZZ] : gs[S_ZZ] + 2.0 * beta * gs[S_ZP] + beta * beta * ctl->pp_cur;
  }
  double acc[1] = {0.0};
  for (int j = blockIdx.x * BS + threadIdx.x; j < n; j += gridDim.x * BS) {
    const double t = (it == 0) ? T[j] : T[j] + beta * tmp[j];
    tmp[j] = t;
    acc[0] += t * t; // ||A'p||^2, replicated
  }
  const int ws[1] = {S_TT};
  write_partials<1>(part, ws, acc, sm);
}
// rhs_x <- T - rhs_x with T = A'rhs_y all-reduced; S_DH partial (x part weighted: replicated)      (indirect.c:419-420, abip.c:560)
__global__ __launch_bounds__(BS) void k_dist_post(const double *__restrict__ T, double *__restrict__ rhs, const double *__restrict__ h, Dims d,
                                                  double xw, double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int j = t0; j < d.n; j += stride) {
    const double v = T[j] - rhs[d.MP + j];
    rhs[d.MP + j] = v;
    acc[0] += xw * (v * h[d.MP + j]);
  }
  for (int i = t0; i < d.m; i += stride) acc[0] += rhs[i] * h[i];
  const int ws[1] = {S_DH};
  write_partials<1>(part, ws, acc, sm);
}
// ---- sharded PCG, COLUMN form of the solve (ABIP_HIP_DIST_CG=cols; solver.hip: enqueue_cg_*).  The iteration around the solve keeps the row
// blocks; inside the solve the m-space is gathered and REPLICATED and A is used by column blocks: A'p is local, A(A'p) = sum_g A_g (A_g'p) is the
// one exchange -- m doubles instead of the row form's n -- and p'Gp, |r|^2, z'r are sums over replicated vectors (no exchange).  The kernels
// that do not touch A are the single-GPU ones (k_cg_spmv_At on the column block, k_cg_update); below: the pieces around the exchanges. ----
// dst[off + i] = src[i]: a rank's rows into their place of a zeroed whole vector (summed over the ranks = the whole vector); gate: 2 = only once the PCG is done
__global__ __launch_bounds__(BS) void k_cols_place(const double *__restrict__ src, int len, int off, double *__restrict__ dst, int gate, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (gate == 2 && !ctl->cg_done) return;
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) dst[off + i] = src[i];
}
// dst[i] = src[off + i]: the rank's rows out of a replicated vector, once the PCG is done
__global__ __launch_bounds__(BS) void k_cols_take(const double *__restrict__ src, int len, int off, double *__restrict__ dst, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  for (int i = blockIdx.x * BS + threadIdx.x; i < len; i += gridDim.x * BS) dst[i] = src[off + i];
}
// d = b_x - t over the rank's columns (t = A_g's, the warm start's share; null without one)
__global__ __launch_bounds__(BS) void k_cols_diff(const double *__restrict__ bx, const double *__restrict__ t, double *__restrict__ dd, int n, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  for (int j = blockIdx.x * BS + threadIdx.x; j < n; j += gridDim.x * BS) dd[j] = t ? bx[j] - t[j] : bx[j];
}
// after the exchange of buf = sum_g A_g (b_x - A_g's): r = b_y + buf - rho s, z = M r, x0 = s; |r|^2, z'r; tolerance and counters (k_cg_init_A's second half)
__global__ __launch_bounds__(BS) void k_cols_init_fin(const double *__restrict__ by, const double *__restrict__ buf, const double *__restrict__ s, const double *__restrict__ Minv,
                                                      double *__restrict__ x, double *__restrict__ r, double *__restrict__ z, double *__restrict__ p, double rho,
                                                      double tol_factor, int m, double *part, Ctl *ctl, const double *gs) {
  ABIP_GATE_HALT(ctl);
  __shared__ double sm[2 * WAVES];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double tol = sqrt(gs[S_BN]) * tol_factor; // indirect.c:406-409 (|b_y| before the accumulation: summed over the row blocks)
    tol = fmax(tol, 1e-7);
    ctl->cg_tol = fmax(tol, 1e-9);
    ctl->cg_it = 0;
    ctl->cg_done = 0;
  }
  double a2[2] = {0.0, 0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    const double si = s ? s[i] : 0.0;
    const double ri = s ? (by[i] + buf[i]) - rho * si : by[i] + buf[i];
    const double zi = ri * Minv[i];
    x[i] = si; r[i] = ri; z[i] = zi; p[i] = zi;
    a2[0] += ri * ri; a2[1] += zi * ri;
  }
  const int ws[2] = {S_RR0, S_ZR0};
  write_partials<2>(part, ws, a2, sm);
}
// after the exchange of buf = sum_g A_g tmp_g: p = z + beta p, Gp = buf + rho p, p'Gp (k_cg_spmv_A's row epilogue)
__global__ __launch_bounds__(BS) void k_cols_Gp_fin(const double *__restrict__ buf, const double *__restrict__ z, double *__restrict__ p, double *__restrict__ Gp, double rho, int m,
                                                    double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  __shared__ double sm[WAVES];
  const double beta = ctl->beta_cur;
  double acc1[1] = {0.0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    const double pn = z[i] + beta * p[i];
    const double gp = buf[i] + rho * pn;
    p[i] = pn; Gp[i] = gp;
    acc1[0] += pn * gp;
  }
  const int ws[1] = {S_PG};
  write_partials<1>(part, ws, acc1, sm);
}
// the convergence decision at the end of a chunk (the product kernel of the next iteration would take it; there is none)
__global__ __launch_bounds__(BS) void k_cols_decide(int max_its, double *part, int nb, Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (ctl->cg_done) return;
  __shared__ double sm[2 * WAVES];
  int it; double zr;
  if (cg_converged(ctl, part, nb, max_its, sm, it, zr) && threadIdx.x == 0) ctl->cg_done = 1;
}

// dual residual sums from T = A'y all-reduced (k_q_At without the SpMV)
__global__ __launch_bounds__(BS) void k_dist_q(const double *__restrict__ T, const double *__restrict__ uu, const double *__restrict__ vv,
                                               const double *__restrict__ c, const double *__restrict__ wE, Dims d, int slot0, double xw,
                                               double *part, const Ctl *ctl) {
  ABIP_GATE_HALT(ctl);
  if (!ctl->cg_done) return;
  __shared__ double sm[3 * WAVES];
  const double tau = uu[d.MP + d.n];
  double acc3[3] = {0.0, 0.0, 0.0};
  for (int j = blockIdx.x * BS + threadIdx.x; j < d.n; j += gridDim.x * BS) {
    const double drj = T[j] + vv[d.MP + j], e = drj - c[j] * tau;
    double sc = wE ? wE[j] : 1.0;
    sc = sc * sc;
    acc3[0] += xw * (e * e); acc3[1] += xw * ((e * e) * sc); acc3[2] += xw * ((drj * drj) * sc);
  }
  const int ws[3] = {slot0, slot0 + 1, slot0 + 2};
  write_partials<3>(part, ws, acc3, sm);
}

// plain y += A x for the unit-level ABI and the direct back-end's accumulations
template <bool SELL>
__global__ __launch_bounds__(BS, SELL ? 6 : 8) void k_spmv_acc(Csr M, const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  spmv_rows<1, SELL>(
      M, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
      [&](int row, double(&acc)[1]) { y[row] += acc[0]; });
}

} // namespace abip
