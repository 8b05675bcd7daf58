// qcp_solver.hip -- the conic (ABIP-QCP) path of libabip_hip.so: abip_qcp() == abip(d, sol, info, K) of the reference
// (src/abip-qcp/source/abip.c:1335-1371) for the generic QCP formulation with the QDLDL-class direct solver.
// Host: scaling (qcp_config.c:91-491), KKT assembly (699-748) + LDL' (host_setup.cpp), the loop of abip.c:1129-1246 with its
// scalar decisions.  Device: everything per iteration (qcp_kernels.h, dev_sptrsv.h); one control read per inner iteration.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/abip_qcp.h"
#include "dev_host_util.h"
#include "dev_ldl.h"
#include "qcp_kernels.h"
#include "qcp_pcg.h"
#include "qcp_dist.h"
#include "dist_internal.h"
#include "qcp_work.h"
#include "qcp_formulations.h"

using namespace abip;
using namespace abip::hostutil;
using namespace abip::qcp;

namespace {

constexpr double EPS_TOL = 1E-18;
inline double safediv_pos(double x, double y) { return y < EPS_TOL ? x / EPS_TOL : x / y; }

void hcsr_from(const HMat &M, host::HostCsr &out, bool transpose_to_rows) {
  // transpose_to_rows = false: CSC read as CSR of M' (ncols rows); true: explicit CSR of M (nrows rows)
  const int nnz = M.p[M.n];
  if (!transpose_to_rows) {
    out.nrows = M.n; out.ncols = M.m; out.ptr = M.p; out.idx = M.i; out.val = M.x;
  } else {
    out.nrows = M.m; out.ncols = M.n;
    (void)nnz;
    host::par_transpose((long)M.m, (long)M.n, M.p.data(), M.i.data(), M.x.data(), out.ptr, out.idx, out.val); // (host_par.h: the counting sort on a few threads, entry for entry the same result)
  }
  host::build_row_blocks(out, CHUNK);
}

void enqueue_solve(QWk *w, double *rhs) { // _ldl_solve, linsys.c:309-316
  w->ldl.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, w->stream, a...); }, rhs, w->lp_ctl, w->NB);
}

// several GPUs: in-place sum over the ranks, on the solver's stream
int ar(QWk *w, double *buf, size_t count) { w->n_allreduce++; return dist_allreduce(buf, count, w->stream, w->hstage); }
// the sums kq_finalize (decide = 0) left in ctl->out for `slots`: exchange them, write the totals back
int exchange_sums(QWk *w, std::initializer_list<int> slots) {
  QPack pk; pk.n = 0;
  for (int sl : slots) pk.slots[pk.n++] = sl;
  double *out = reinterpret_cast<double *>(w->ctl.p); // QCtl::out is its first member
  QLAUNCH(w, kq_dist_pack, 1, 64, pk, (const double *)out, w->gsbuf.p, (int)Q_COUNT);
  if (ar(w, w->gsbuf.p, Q_COUNT)) return -1;
  QLAUNCH(w, kq_dist_unpack, 1, 64, pk, (const double *)w->gsbuf.p, out);
  return 0;
}

// K z = rhs by y-space PCG (qcp_pcg.h).  warm: y0 = (u + tau r)_y and the tolerance of abip.c:213-217 (tol_host = min of the two residual
// norms of the last check); otherwise y0 = 0 and tol as given (the set-up solve, abip.c:899).  Synchronises with the host once per chunk
// of iterations.  Returns the CG iterations used, < 0 on a device error.
int solve_pcg(QWk *w, double *rhs, bool warm, int iter, double tol_host) {
  const QDims d{w->m, w->n, w->MP, w->kind == 0 ? 1 : (w->kind == 1 ? 2 : 0), w->wy};
  const QCPSettings *st = w->st;
  QPcgVec v{w->cg_x0.p, w->cg_r.p, w->cg_z.p, w->cg_p.p, w->cg_Gp.p, w->cg_tm.p, w->cg_M.p, w->cg_H.p};
  Ctl *hc = w->lp_ctl;
  const int max_its = w->m;
  double *buf = w->arbuf.p; // several GPUs: the exchange area, [0, m) products, [m, m + world) the max lanes of the warm start
  const double ipow = std::pow((double)iter + 1.0, 1.5);
  auto aty = [&](bool init) { // tn = H^-1 A' (y0 | z): the m-vector in LDS where it fits (one 1024-thread workgroup per CU), else the streaming kernel
    if (w->aty_lds) {
      const size_t lds = sizeof(double) * (size_t)w->m;
      const double *pp = w->cg_part.p;
      if (w->aty_lds == 16) {
        if (init) hipLaunchKernelGGL((kq_pcg_Aty_lds<true, 16>), dim3(w->n_cu), dim3(1024), lds, w->stream, w->dAt.view(), v, w->m, max_its, pp, w->NB, hc);
        else hipLaunchKernelGGL((kq_pcg_Aty_lds<false, 16>), dim3(w->n_cu), dim3(1024), lds, w->stream, w->dAt.view(), v, w->m, max_its, pp, w->NB, hc);
      } else {
        if (init) hipLaunchKernelGGL((kq_pcg_Aty_lds<true, 64>), dim3(w->n_cu), dim3(1024), lds, w->stream, w->dAt.view(), v, w->m, max_its, pp, w->NB, hc);
        else hipLaunchKernelGGL((kq_pcg_Aty_lds<false, 64>), dim3(w->n_cu), dim3(1024), lds, w->stream, w->dAt.view(), v, w->m, max_its, pp, w->NB, hc);
      }
      return;
    }
    if (init) QLAUNCH(w, kq_pcg_Aty<true>, w->NB, BS, w->dAt.view(), v, max_its, w->cg_part.p, w->NB, hc);
    else QLAUNCH(w, kq_pcg_Aty<false>, w->NB, BS, w->dAt.view(), v, max_its, w->cg_part.p, w->NB, hc);
  };
  if (w->dist) { // qcp_dist.h: products of the rank's column block, all-reduced, then the element-wise halves on the replicated m-space
    QLAUNCH(w, kq_prod_A, w->NB, BS, w->dA.view(), (const double *)(rhs + w->MP), (const double *)w->cg_H.p, buf, 0, (const Ctl *)hc);
    QLAUNCH(w, kq_dist_prep_warm, w->NB, BS, (const double *)w->u.p, (const double *)w->r.p, warm ? 1 : 0, d, w->n0, w->n_glob - w->m, v, w->cg_part.p, hc);
    QLAUNCH(w, kq_dist_pack_max, 1, BS, (const double *)w->cg_part.p, (int)PQ_WM, 1, w->NB, buf + w->m, w->rank, w->world);
    if (ar(w, buf, (size_t)w->m + w->world)) return -1;
    QLAUNCH(w, kq_dist_prep_fin, w->NB, BS, rhs, (const double *)buf, w->m, (const Ctl *)hc);
  } else QLAUNCH(w, kq_pcg_prep, w->NB, BS, w->dA.view(), rhs, (const double *)w->u.p, (const double *)w->r.p, warm ? 1 : 0, d, v, w->cg_part.p, hc);
  if (warm) {
    aty(true);
    if (w->dist) {
      QLAUNCH(w, kq_prod_A, w->NB, BS, w->dA.view(), (const double *)w->cg_tm.p, (const double *)nullptr, buf, 1, (const Ctl *)hc);
      if (ar(w, buf, (size_t)w->m)) return -1;
      QLAUNCH(w, kq_dist_Gp_fin<true>, w->NB, BS, v, rhs, (const double *)buf, st->rho_y, tol_host, ipow, (const double *)(buf + w->m), w->world, w->m, w->cg_part.p, hc);
    } else QLAUNCH(w, kq_pcg_Gp<true>, w->NB, BS, w->dA.view(), v, rhs, st->rho_y, tol_host, ipow, w->cg_part.p, w->NB, hc);
  } else {
    QLAUNCH(w, kq_pcg_init_cold, w->NB, BS, v, rhs, w->m, tol_host, w->cg_part.p, hc);
  }
  int chunk = std::max(2, w->last_cg + std::max(2, w->last_cg >> 3));
  for (;;) {
    for (int q = 0; q < chunk; ++q) {
      aty(false);
      if (w->dist) {
        QLAUNCH(w, kq_prod_A, w->NB, BS, w->dA.view(), (const double *)w->cg_tm.p, (const double *)nullptr, buf, 1, (const Ctl *)hc);
        if (ar(w, buf, (size_t)w->m)) return -1;
        QLAUNCH(w, kq_dist_Gp_fin<false>, w->NB, BS, v, rhs, (const double *)buf, st->rho_y, 0.0, 1.0, (const double *)(buf + w->m), w->world, w->m, w->cg_part.p, hc);
      } else QLAUNCH(w, kq_pcg_Gp<false>, w->NB, BS, w->dA.view(), v, rhs, st->rho_y, 0.0, 1.0, w->cg_part.p, w->NB, hc);
      QLAUNCH(w, kq_pcg_update, w->NB, BS, v, rhs, w->m, w->cg_part.p, w->NB, hc);
    }
    QLAUNCH(w, kq_pcg_post, w->NB, BS, w->dAt.view(), rhs, v, max_its, d, w->cg_part.p, w->NB, hc);
    HIP_OK(hipMemcpyAsync(w->hlp, hc, sizeof(Ctl), hipMemcpyDeviceToHost, w->stream));
    HIP_OK(hipStreamSynchronize(w->stream));
    if (w->hlp->cg_done || w->hlp->halt) break;
    chunk = std::max(4, chunk);
  }
  const int zero = 0;
  HIP_OK(hipMemcpyAsync(&hc->cg_done, &zero, sizeof(int), hipMemcpyHostToDevice, w->stream)); // the kernels of the rest of the iteration are not gated on it, the next solve is
  w->last_cg = w->hlp->cg_it;
  if (iter >= 0) { w->tot_cg += w->hlp->cg_it; w->cg_solves++; }
  return w->hlp->cg_it;
}

// the device raised the halt flag at the inner exit: lower it (and its mirror) for the next inner loop
int clear_halt(QWk *w) {
  HIP_OK(hipMemsetAsync(&w->lp_ctl->halt, 0, sizeof(int), w->stream));
  HIP_OK(hipMemsetAsync(&w->ctl.p->halted, 0, sizeof(int), w->stream));
  w->hctl->halted = 0;
  return 0;
}
int read_ctl(QWk *w) {
  HIP_OK(hipMemcpyAsync(w->hctl, w->ctl.p, sizeof(QCtl), hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  return 0;
}
void finalize(QWk *w, std::initializer_list<int> slots, std::initializer_list<int> both_halves, double tol_inner = -1.0) {
  QFin f; f.nslots = 0;
  for (int s : slots) { f.slots[f.nslots] = s; f.second_half[f.nslots] = 0; for (int bsl : both_halves) if (bsl == s) f.second_half[f.nslots] = 1; ++f.nslots; }
  f.norm_u = w->kind == 0 ? 1 : (w->kind == 1 ? 2 : 0);
  if (tol_inner >= 0) { f.decide = 1; f.tol_inner = tol_inner; f.u_tau = w->u.p + w->MP + w->n; f.vo_tau = w->vo.p + w->MP + w->n; }
  if (w->dist && f.decide) { // local sums -> exchange -> the decision from the totals (every rank decides alike)
    QFin g = f; g.decide = 0;
    QLAUNCH(w, kq_finalize, 1, 1024, g, (const double *)w->part.p, w->NB, w->ctl.p, w->lp_ctl);
    if (exchange_sums(w, slots)) { w->dist_failed = true; return; }
    QLAUNCH(w, kq_decide, 1, 64, f, w->ctl.p, w->lp_ctl);
    return;
  }
  QLAUNCH(w, kq_finalize, 1, 1024, f, (const double *)w->part.p, w->NB, w->ctl.p, w->lp_ctl);
}

double adjust_barrier(QWk *w, const QResid &r) { // abip.c:994-1071
  const QCPSettings *st = w->st;
  double sigma = 0.8, gamma;
  const double ratio = w->mu / std::min(std::min(st->eps_p, st->eps_d), st->eps_g);
  if (ratio > 50 && ratio <= 100) gamma = 1.5;
  else if (ratio > 10 && ratio <= 50) gamma = 1.3;
  else if (ratio > 5 && ratio <= 10) gamma = 1.2;
  else if (ratio > 1 && ratio <= 5) gamma = 1.1;
  else if (ratio > 0.5 && ratio <= 1) gamma = 1;
  else if (ratio > 0.05 && ratio <= 0.5) gamma = 0.9;
  else if (ratio > 0.005 && ratio <= 0.05) gamma = 0.8;
  else if (ratio > 0.0005 && ratio <= 0.005) gamma = 0.7;
  else if (ratio > 0.00005 && ratio <= 0.0005) gamma = 0.6;
  else gamma = 0.5;
  const double mr = r.error_ratio;
  if (mr > 22) gamma *= 4.4;
  else if (mr > 18 && mr <= 22) gamma *= 4.2;
  else if (mr > 15 && mr <= 18) gamma *= 4;
  else if (mr > 12 && mr <= 15) gamma *= 3.8;
  else if (mr > 8 && mr <= 12) gamma *= 3.6;
  else if (mr > 6 && mr <= 8) { sigma = 0.81; gamma *= 3.4; }
  else if (mr > 4 && mr <= 6) { sigma = 0.82; gamma *= 3.4; }
  else if (mr > 3 && mr <= 4) { sigma = 0.83; gamma *= 3.2; }
  else if (mr > 2 && mr <= 3) { sigma = 0.85; gamma *= 2.8; }
  else if (mr > 1.5 && mr <= 2) { sigma = 0.85; gamma *= 2.6; }
  else if (mr < 1.5) { sigma = 0.85; gamma *= 2.4; }
  sigma *= 0.2;
  w->mu = sigma * w->mu;
  return gamma * std::pow(w->mu, st->psi);
}

int has_converged(const QWk *w, const QResid &r, int ipm_iter, int admm_iter) { // abip.c:750-777
  const QCPSettings *st = w->st;
  if (r.res_pri < st->eps_p && r.res_dual < st->eps_d && r.rel_gap < st->eps_g) return 1;
  if (r.res_dif < st->err_dif * std::max(std::max(st->eps_p, st->eps_d), st->eps_g)) return 2;
  if (r.res_unbdd < st->eps_unb && ipm_iter > 0 && admm_iter > 0) return -1;
  if (r.res_infeas < st->eps_inf && ipm_iter > 0 && admm_iter > 0) return -2;
  return 0;
}

int calc_residuals(QWk *w, QResid &r, int ipm_iter, int admm_iter) { // qcp_config.c:562-691 (sums from kq_resid)
  if (admm_iter && r.last_admm_iter == admm_iter) return 0;
  r.last_ipm_iter = ipm_iter; r.last_admm_iter = admm_iter;
  const QDims d{w->m, w->n, w->MP, w->kind == 0 ? 1 : (w->kind == 1 ? 2 : 0), w->wy};
  QLAUNCH(w, kq_resid, w->NB, BS, (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->bd.p, (const double *)w->cd.p,
          (const double *)w->Dd.p, (const double *)w->Ed.p, (const double *)w->Ax.p, (const double *)w->ATy.p, (const double *)w->Qx.p, d, w->part.p);
  finalize(w, {Q_S0, Q_S1, Q_S2, Q_S3, Q_S4, Q_S5, Q_M0, Q_M1, Q_M2, Q_M3, Q_M4, Q_M5}, {});
  if (w->dist) { // sums: packed exchange; inf-norms: one lane per rank in a sum exchange, then the maximum over the lanes
    if (exchange_sums(w, {Q_S0, Q_S1, Q_S2, Q_S3, Q_S4, Q_S5})) return -1;
    double *mx = w->gsbuf.p + Q_COUNT;
    QLAUNCH(w, kq_dist_pack_max, 1, BS, (const double *)w->part.p, (int)Q_M0, 6, w->NB, mx, w->rank, w->world);
    if (ar(w, mx, (size_t)6 * w->world)) return -1;
    QLAUNCH(w, kq_dist_unpack_max, 1, 64, (const double *)mx, (int)Q_M0, 6, w->world, reinterpret_cast<double *>(w->ctl.p));
  }
  if (w->kind == 0) {
    const LassoForm &L = w->ls;
    QLasso ql{L.dm, L.dn, std::sqrt(L.sc_cone2), L.sc_b, L.sc_c, L.lambda, L.Dd.p, L.Ed.p, L.yd.p, w->n0};
    QLAUNCH(w, kq_resid_lasso, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->Ax.p, (const double *)w->ATy.p, ql, d, w->part.p);
    finalize(w, {Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5}, {});
    if (w->dist && exchange_sums(w, {Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5})) return -1;
  }
  if (w->kind == 1) {
    const SvmForm &V = w->sv;
    QSvm qs{V.dm, V.dn, V.sc, V.sc_b, V.sc_c, V.lambda, V.Dd.p, V.Ed.p, V.wEd.p};
    QLAUNCH(w, kq_resid_svm, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->Ax.p, (const double *)w->ATy.p, qs, d, w->part.p);
    finalize(w, {Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5}, {});
  }
  double tails[2];
  HIP_OK(hipMemcpyAsync(&tails[0], w->u.p + w->MP + w->n, sizeof(double), hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipMemcpyAsync(&tails[1], w->vo.p + w->MP + w->n, sizeof(double), hipMemcpyDeviceToHost, w->stream));
  if (read_ctl(w)) return -1;
  const double *o = w->hctl->out;
  const QCPSettings *st = w->st;
  if (w->kind == 0) { // calc_lasso_residuals, lasso_config.c:358-503
    const LassoForm &L = w->ls;
    r.tau = tails[0];
    r.Ax_b_norm = o[Q_M0]; r.Qx_ATy_c_s_norm = o[Q_M3]; // (for the PCG tolerance of abip.c:213-217 only)
    double ny = 0; for (double t : L.y) ny += t * t;
    const double this_pr = std::sqrt(o[Q_L0]) / std::max(std::sqrt(ny), 1.0);
    const double this_dr = std::sqrt(o[Q_L1]) / (std::sqrt((double)(2 * L.dn)) * L.lambda);
    const double P = 0.5 * o[Q_L2] + L.lambda * o[Q_L3];
    const double this_gap = std::fabs(P + 0.5 * o[Q_L4] - o[Q_L5]) / (1 + std::fabs(P));
    r.dobj = -0.5 * o[Q_L4] + o[Q_L5]; r.pobj = P;
    r.res_dif = std::max(std::max(std::fabs(this_pr - r.res_pri), std::fabs(this_dr - r.res_dual)), std::fabs(this_gap - r.rel_gap));
    r.res_pri = this_pr; r.res_dual = this_dr; r.rel_gap = this_gap;
    r.error_ratio = std::max(r.res_pri / st->eps_p, std::max(r.res_dual / st->eps_d, r.rel_gap / st->eps_g));
    const double ctu = o[Q_S2], btu = o[Q_S1]; // c'u_x, b'u_y; |A u_x|_2 and |A'u_y + v_o|_2 from the generic sums (D = E = 1 here)
    r.res_unbdd = ctu < 0 ? std::sqrt(o[Q_S0]) / (-ctu) : INFINITY;
    r.res_infeas = btu > 0 ? std::sqrt(o[Q_S5]) / btu : INFINITY;
    return 0;
  }
  if (w->kind == 1) { // calc_svm_residuals, svm_config.c:445-561
    const SvmForm &V = w->sv;
    const double C = V.lambda;
    r.tau = tails[0];
    r.Ax_b_norm = o[Q_M0]; r.Qx_ATy_c_s_norm = o[Q_M3]; // (for the PCG tolerance of abip.c:213-217 only)
    const double this_pr = std::sqrt(o[Q_L0]) / std::sqrt((double)V.dm);
    const double this_dr = std::sqrt(o[Q_L1]) / (std::sqrt((double)V.dm) * C);
    r.dobj = o[Q_L2] - 0.5 * o[Q_L5]; r.pobj = C * o[Q_L3] + 0.5 * o[Q_L4];
    const double this_gap = std::fabs(r.dobj - r.pobj) / (1 + std::fabs(r.pobj));
    r.res_dif = std::max(std::max(std::fabs(this_pr - r.res_pri), std::fabs(this_dr - r.res_dual)), std::fabs(this_gap - r.rel_gap));
    r.res_pri = this_pr; r.res_dual = this_dr; r.rel_gap = this_gap;
    r.error_ratio = std::max(r.res_pri / st->eps_p, std::max(r.res_dual / st->eps_d, r.rel_gap / st->eps_g));
    const double ctu = o[Q_S2], btu = o[Q_S1];
    r.res_unbdd = ctu < 0 ? std::sqrt(o[Q_S0]) / (-ctu) : INFINITY;
    r.res_infeas = btu > 0 ? std::sqrt(o[Q_S5]) / btu : INFINITY;
    return 0;
  }
  r.tau = std::fabs(tails[0]);
  r.kap = std::fabs(tails[1]) / (st->normalize ? (st->scale * w->sc_c * w->sc_b) : 1);
  r.Ax_b_norm = o[Q_M0];
  const double this_pr = o[Q_M1] / (w->sc_b + std::max(o[Q_M2], w->sc_b * w->nm_inf_b));
  const double xQx_2 = w->hasQ ? (o[Q_S3] / (r.tau * r.tau)) / (2 * w->sc_b * w->sc_c) : 0.0;
  r.Qx_ATy_c_s_norm = o[Q_M3];
  const double this_dr = o[Q_M4] / (w->sc_c + std::max(w->sc_c * w->nm_inf_c, o[Q_M5]));
  const double cTx = (o[Q_S2] / r.tau) / (w->sc_b * w->sc_c), bTy = (o[Q_S1] / r.tau) / (w->sc_b * w->sc_c);
  const double this_gap = std::fabs(2 * xQx_2 + cTx - bTy) / (1 + std::max(2 * xQx_2, std::max(std::fabs(cTx), std::fabs(bTy))));
  r.pobj = xQx_2 + cTx; r.dobj = -xQx_2 + bTy;
  r.res_dif = std::max(std::max(std::fabs(this_pr - r.res_pri), std::fabs(this_dr - r.res_dual)), std::fabs(this_gap - r.rel_gap));
  r.res_pri = this_pr; r.res_dual = this_dr; r.rel_gap = this_gap;
  r.error_ratio = std::max(r.res_pri / st->eps_p, std::max(r.res_dual / st->eps_d, r.rel_gap / st->eps_g));
  const double ctu = o[Q_S2], btu = o[Q_S1];
  r.res_unbdd = ctu < 0 ? std::max(std::sqrt(o[Q_S4]), std::sqrt(o[Q_S0])) / (-ctu) : INFINITY;
  r.res_infeas = btu > 0 ? std::sqrt(o[Q_S5]) / btu : INFINITY;
  return 0;
}

qcp_int fail(QCPInfo *info, const char *msg) {
  if (info) { info->status_val = -4; strcpy(info->status, "Failure"); info->ipm_iter = -1; info->admm_iter = -1; info->pobj = info->dobj = NAN; }
  printf("Failure:%s\n", msg);
  return -4;
}

} // namespace

extern "C" {

// unit-level access to the cone kernel (parity tests at the branch boundaries of cones.c:130-248)
int abip_hip_qcp_cone_prox(int kind, double *x, const double *tmp, double lambda, int len) {
  if (!x || !tmp || len < (kind == 0 ? 1 : 2) || (kind != 0 && kind != 1)) return -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { printf("ERROR: no usable HIP device: libabip_hip has no CPU fallback\n"); return -2; }
  DBuf<double> u, rel; DBuf<int> meta;
  std::vector<double> hu(x, x + len), hr(tmp, tmp + len);
  std::vector<int> hm = {0, len, kind};
  if (u.upload(hu, nullptr) || rel.upload(hr, nullptr) || meta.upload(hm, nullptr)) return -3;
  QCones C{meta.p, meta.p + 1, meta.p + 2, 1};
  if (len > QC_BIG) hipLaunchKernelGGL(kq_cones<true>, dim3(1), dim3(QC_TB), 0, nullptr, C, 0, u.p, (const double *)rel.p, lambda, 0, (const Ctl *)nullptr);
  else hipLaunchKernelGGL(kq_cones<false>, dim3(1), dim3(BS), 0, nullptr, C, 0, u.p, (const double *)rel.p, lambda, 0, (const Ctl *)nullptr);
  const int rc = (hipMemcpy(x, u.p, sizeof(double) * len, hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess) ? 0 : -4;
  u.release(); rel.release(); meta.release();
  return rc;
}

// Column ranges of the sharded conic path: bounds[g] .. bounds[g+1] are rank g's columns (world + 1 entries out); pure host code.
// 0 ok, -1 a rotated cone of fewer than 3 entries, -2 fewer blocks than ranks, -3 bad arguments.
// Pure host code (no device needed): the formulation front end and the scaling exactly as abip_qcp applies them (qcp_formulations.h), then two products with the
// scaled, materialised operator.  For CPU-side parity tests of the host logic (tests/test_qcp_host_cpu.py).  Any output pointer may be NULL.
int abip_hip_qcp_host_probe(const QCPData *d, const QCPCone *K, const double *x_in, const double *y_in, double *Ax_out, double *Aty_out, double *b_out, double *c_out,
                            double *scal4, int *dims2) {
  if (!d || !K || !d->stgs || !d->A || !d->b) return -1;
  const QCPSettings *st = d->stgs;
  const int kind = st->prob_type;
  if (kind < 0 || kind > 3 || (kind == 2 && !d->c) || (kind != 2 && (d->m <= 0 || d->n <= 0 || !(d->lambda > 0)))) return -1;
  if ((kind == 0 && (!st->normalize || !st->scale_E)) || ((kind == 1 || kind == 3) && !st->normalize) || (kind == 1 && !st->scale_E)) return -1;
  const int m = kind == 0 ? d->m + 1 : (kind == 1 ? d->m + d->n + 1 : d->m);
  const int n = kind == 0 ? 2 + 2 * d->n + d->m : (kind == 1 ? 4 + 3 * d->n + 2 * d->m : (kind == 3 ? 1 + d->n + 2 * d->m : d->n));
  if (dims2) { dims2[0] = m; dims2[1] = n; }
  QWk W; QWk *w = &W;
  w->kind = kind; w->m = m; w->n = n; w->st = st; w->hasQ = kind == 3 || (kind == 2 && d->Q != nullptr);
  w->kkt_rho_x = (kind == 0 || kind == 1) ? 1.0 : st->rho_x;
  if (kind == 0) build_lasso(w, d);
  else if (kind == 1) { if (!build_svm(w, d)) return -2; }
  else if (kind == 3) build_svmqp(w, d, K);
  else {
    w->sparsity = (((long long)d->A->p[n] / std::max(1LL, (long long)m * (long long)n)) < 0.05);
    copy_in(w->A, d->A);
    if (w->hasQ) copy_in(w->Q, d->Q);
    w->nm_inf_b = vnrminf(d->b, m); w->nm_inf_c = vnrminf(d->c, n);
    scale_data(w, d, K);
  }
  const HMat &A = w->A;
  if (A.m != m || A.n != n) return -3;
  if (Ax_out && x_in) { std::fill(Ax_out, Ax_out + m, 0.0); for (int j = 0; j < n; ++j) for (int q = A.p[j]; q < A.p[j + 1]; ++q) Ax_out[A.i[q]] += A.x[q] * x_in[j]; }
  if (Aty_out && y_in) for (int j = 0; j < n; ++j) { double t = 0.0; for (int q = A.p[j]; q < A.p[j + 1]; ++q) t += A.x[q] * y_in[A.i[q]]; Aty_out[j] = t; }
  if (b_out) std::copy(w->b.begin(), w->b.begin() + m, b_out);
  if (c_out) std::copy(w->c.begin(), w->c.begin() + n, c_out);
  if (scal4) { scal4[0] = w->sc_b; scal4[1] = w->sc_c; scal4[2] = (double)A.p[n]; scal4[3] = (double)w->sparsity; }
  return 0;
}

int abip_hip_qcp_dist_partition(const QCPMatrix *A, const QCPCone *K, int world, int *bounds_out) {
  if (!A || !K || !bounds_out || world < 1 || A->n < 1) return -3;
  std::vector<int> bounds;
  const int rc = column_bounds(A->n, A->p, K, world, bounds, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc != 0) return rc;
  for (int g = 0; g <= world; ++g) bounds_out[g] = bounds[g];
  return 0;
}

void abip_hip_qcp_last_stats(double *out8) { for (int q = 0; q < 8; ++q) out8[q] = g_stats[q]; }
void abip_hip_qcp_phase_times(double *out5) { for (int q = 0; q < 5; ++q) out5[q] = g_phase[q]; }

void abip_qcp_set_default_settings(QCPData *d) { // util.c:203-255
  QCPSettings *s = d->stgs;
  const double nz = d->A ? d->A->p[d->n] : 0, sparsity = nz / ((double)d->m * d->n);
  s->normalize = 1; s->scale_E = 1; s->scale_bc = 1; s->max_ipm_iters = 500; s->max_admm_iters = 10000000;
  s->eps = s->eps_p = s->eps_d = s->eps_g = s->eps_inf = s->eps_unb = 1e-3; s->alpha = 1.8; s->cg_rate = 2.0;
  s->use_indirect = 0; s->scale = 1.0; s->rho_y = 1e-6; s->rho_x = 1; s->rho_tau = 1; s->verbose = 1; s->err_dif = 0;
  s->inner_check_period = 500; s->outer_check_period = 1;
  s->linsys_solver = ((double)d->m * d->n > 1e12) ? 3 : (sparsity > 0.4 ? 5 : 1);
  s->prob_type = 2; // what the mex gateway sets before calling abip() (abip_qcp_mex.c:436)
  s->time_limit = INFINITY; s->psi = 1; s->origin_scaling = 1; s->ruiz_scaling = 1; s->pc_scaling = 0;
}

qcp_int abip_qcp(const QCPData *d, QCPSolution *sol, QCPInfo *info, QCPCone *K) {
  if (!d || !sol || !info || !K) return fail(info, "ABIP_NULL input");
  if (!d->stgs) return fail(info, "ABIP_NULL input");
  const int kind = d->stgs->prob_type; // abip.c:1341-1348: 0 LASSO, 1 SVM, 2 QCP, 3 SVMQP
  if (kind < 0 || kind > 3) return fail(info, "prob_type must be 0 (LASSO), 1 (SVM as an SOCP), 2 (generic QCP) or 3 (SVM as a QP)");
  if (!d->A || !d->b || (kind == 2 && !d->c)) return fail(info, "the device path needs A, b and c");
  // linsys_solver: 3 = PCG; 0 (MKL-DSS), 1 (QDLDL), 2 (CSparse Cholesky), 4 (PARDISO), 5 (LAPACK dense Cholesky) are the reference's exact factorisations of the
  // same KKT system (linsys.c:1129-1170; the default rule of util.c:237-243 picks 5 for dense data): all of them are served by the device LDL'
  if (d->stgs->linsys_solver < 0 || d->stgs->linsys_solver > 5) { printf("\nlinsys solver type error\n"); return fail(info, "linsys_solver must be 0 .. 5"); }
  const QCPSettings *st = d->stgs;
  if (kind == 0) { // LASSO: data = (X, y, lambda) as abip_ml_mex.c:117-160 hands them over
    if (d->m <= 0 || d->n <= 0 || !(d->lambda > 0)) return fail(info, "LASSO needs a non-empty X and lambda > 0");
    // the reference scales the data whatever `normalize` says and un-scales only when it is set, and divides by E = 0 when scale_E = 0
    // (lasso_config.c:147-156, 300-310; abip.c:580-582): neither combination returns a usable beta
    if (!st->normalize || !st->scale_E) return fail(info, "the LASSO formulation needs normalize = 1 and scale_E = 1");
    if ((long long)2 + 2LL * d->n + d->m > 2147483647LL || 1LL + d->m + 2LL * d->A->p[d->n] > 2147483647LL) return fail(info, "problem too large for 32-bit indices");
  }
  if (kind == 3 || kind == 1) { // SVM: data = (X, labels y, lambda) as abip_ml_mex.c:117-160 hands them over
    if (d->m <= 0 || d->n <= 0 || !(d->lambda > 0)) return fail(info, "SVM needs a non-empty X and lambda > 0");
    if (!st->normalize) return fail(info, "the SVM formulation needs normalize = 1"); // (as for LASSO: scaled unconditionally, un-scaled only when set)
    if (kind == 1 && !st->scale_E) return fail(info, "the SVM-SOCP formulation needs scale_E = 1"); // (E = 0 otherwise, svm_config.c:296-316)
    if ((long long)4 + 3LL * d->n + 2LL * d->m > 2147483647LL || 1LL + 3LL * d->n + 4LL * d->m + 2LL * d->A->p[d->n] > 2147483647LL)
      return fail(info, "problem too large for 32-bit indices"); // (columns / non-zeros of the materialised operator)
  }
  const int m = kind == 0 ? d->m + 1 : (kind == 1 ? d->m + d->n + 1 : d->m);
  const int n = kind == 0 ? 2 + 2 * d->n + d->m : (kind == 1 ? 4 + 3 * d->n + 2 * d->m : (kind == 3 ? 1 + d->n + 2 * d->m : d->n));
  { // validate, abip.c:779-832 ; cones.c:37-81
    long dims = (long)K->l + K->z + K->f;
    for (int i = 0; K->q && i < K->qsize; ++i) dims += K->q[i];
    for (int i = 0; K->rq && i < K->rqsize; ++i) dims += K->rq[i];
    if (n <= 0) { printf("n must be greater than 0; n = %li\n", (long)n); return fail(info, "could not initialize work"); }
    if (m > n) { printf("WARN: m larger than n, problem likely degenerate\n"); return fail(info, "could not initialize work"); }
    if (dims != n) { printf("cone dimensions %li not equal to num rows in A = n = %li\n", dims, (long)n); return fail(info, "could not initialize work"); }
    if (st->max_ipm_iters <= 0 || st->max_admm_iters <= 0 || st->eps_p <= 0 || st->eps_d <= 0 || st->eps_g <= 0 || st->eps_inf <= 0 ||
        st->eps_unb <= 0 || st->alpha <= 0 || st->alpha >= 2 || st->rho_y <= 0)
      return fail(info, "could not initialize work");
  }
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { printf("ERROR: no usable HIP device: libabip_hip has no CPU fallback\n"); return fail(info, "could not initialize work"); }
  const double t_init = now_ms();
  const bool tms = getenv("ABIP_HIP_SETUP_TIMES") != nullptr;
  double t_ph = t_init;
  auto phase = [&](const char *what) { if (tms) { const double t = now_ms(); printf("[setup] conic: %s %.3f s\n", what, (t - t_ph) / 1e3); t_ph = t; } };
  QWk W; QWk *w = &W;
  w->kind = kind; w->m = m; w->n = n; w->st = st; w->hasQ = kind == 3 || (kind == 2 && d->Q != nullptr);
  w->kkt_rho_x = (kind == 0 || kind == 1) ? 1.0 : st->rho_x; // (lasso_config.c:652-708 and svm_config.c:725-806 hard-code rho_x = 1 in the solve)
  if (kind == 0) build_lasso(w, d);
  else if (kind == 1) { if (!build_svm(w, d)) return fail(info, "SVM-SOCP: X has an all-zero feature column (or only zero-label entries in it); drop it or use prob_type 3"); }
  else if (kind == 3) build_svmqp(w, d, K);
  else {
    // integer division, qcp_config.c:22 -- taken in 64 bits: the reference's 32-bit m * n overflows from m * n = 2^31 on (and divides by
    // zero at m = 32768, n = 131072); for every size the reference can run, the quotient below is the same number
    w->sparsity = (((long long)d->A->p[n] / std::max(1LL, (long long)m * (long long)n)) < 0.05);
    copy_in(w->A, d->A);
    if (w->hasQ) copy_in(w->Q, d->Q);
    w->nm_inf_b = vnrminf(d->b, m); w->nm_inf_c = vnrminf(d->c, n);
    scale_data(w, d, K);
  }
  // (a rank that gives up inside a sharded solve aborts the communicator first: its peers would otherwise wait in their next collective for ever)
  auto bail = [&](const char *msg) { if (w->dist) dist_abort_from(msg); release(w); return fail(info, msg); };
  // ---- several GPUs (dist_internal.h): this rank's column block, cut at cone boundaries (qcp_dist.h).  Served with the PCG back-end for the generic
  // formulation, the LASSO front end and the SVM-QP front end; everything else runs as independent replicas (the direct back-end does not shard).
  const DistInfo di = dist_info();
  QCPCone Kloc = *K;
  std::vector<int> kq_loc, krq_loc;
  std::vector<double> Mpre; // the Jacobi preconditioner sums over ALL columns: formed before the block is cut out
  int nl = n;
  w->n_glob = n;
  const long nnz_glob = w->A.p[n];
  if (di.kind != 0 && (kind == 2 || kind == 0 || kind == 3) && st->linsys_solver == 3) { // (not the SVM-SOCP: its residuals pair entries of different columns)
    w->dist = true; w->rank = di.rank; w->world = di.world; w->wy = di.rank == 0 ? 1.0 : 0.0;
    std::vector<int> bounds, qs, qe, rs, re;
    int f0 = 0;
    {
      std::vector<int> colp(w->A.p.begin(), w->A.p.end());
      const int rc = column_bounds(n, colp.data(), K, di.world, bounds, &qs, &qe, &rs, &re, &f0);
      if (rc == -1) return bail("sharded conic path: rotated cones of fewer than 3 entries are not served");
      if (rc != 0) return bail("sharded conic path: fewer column blocks (cones, free / zero / orthant entries) than ranks");
    }
    const int z0 = f0 + K->f, l0 = z0 + K->z;
    const int n0 = bounds[di.rank], n1 = bounds[di.rank + 1];
    if (n1 <= n0) return bail("sharded conic path: empty column block");
    w->n0 = n0; nl = n1 - n0;
    for (size_t q = 0; q < qs.size(); ++q) if (qs[q] >= n0 && qe[q] <= n1) kq_loc.push_back(qe[q] - qs[q]);
    for (size_t q = 0; q < rs.size(); ++q) if (rs[q] >= n0 && re[q] <= n1) krq_loc.push_back(re[q] - rs[q]);
    auto overlap = [&](int a, int b) { return std::max(0, std::min(b, n1) - std::max(a, n0)); };
    Kloc.q = kq_loc.empty() ? nullptr : kq_loc.data(); Kloc.qsize = (int)kq_loc.size();
    Kloc.rq = krq_loc.empty() ? nullptr : krq_loc.data(); Kloc.rqsize = (int)krq_loc.size();
    Kloc.f = overlap(f0, z0); Kloc.z = overlap(z0, l0); Kloc.l = overlap(l0, n);
    // the preconditioner M_i = 1 / (rho_y + sum_j A_ij^2 / H_jj) over all columns (replicated, like every m-space quantity)
    Mpre.assign(m, st->rho_y);
    {
      std::vector<double> Hd(n, w->kkt_rho_x);
      if (w->hasQ) for (int j = 0; j < n; ++j) for (int q = w->Q.p[j]; q < w->Q.p[j + 1]; ++q) {
        if (w->Q.i[q] == j) Hd[j] += w->Q.x[q];
        else if (w->Q.x[q] != 0.0) return bail("linsys_solver = 3 (PCG) needs Q absent or diagonal; use linsys_solver = 1");
      }
      for (int j = 0; j < n; ++j) for (int q = w->A.p[j]; q < w->A.p[j + 1]; ++q) Mpre[w->A.i[q]] += w->A.x[q] * w->A.x[q] * (1.0 / Hd[j]);
      for (int i = 0; i < m; ++i) Mpre[i] = 1.0 / Mpre[i];
    }
    // cut the block out: A_g, the diagonal block of Q, c_g, E_g
    auto cut_cols = [&](HMat &M, bool square) {
      HMat G; G.m = square ? nl : M.m; G.n = nl; G.p.assign(nl + 1, 0);
      const int base = M.p[n0];
      for (int j = 0; j <= nl; ++j) G.p[j] = M.p[n0 + j] - base;
      G.i.assign(M.i.begin() + base, M.i.begin() + M.p[n1]); G.x.assign(M.x.begin() + base, M.x.begin() + M.p[n1]);
      if (square) for (int &r : G.i) r -= n0;
      M = std::move(G);
    };
    cut_cols(w->A, false);
    if (w->hasQ) cut_cols(w->Q, true);
    w->Efull = w->E;
    w->c = std::vector<double>(w->c.begin() + n0, w->c.begin() + n1);
    w->E = std::vector<double>(w->E.begin() + n0, w->E.begin() + n1);
    w->n = nl;
  }
  const QCPCone *KK = &Kloc;
  phase("formulation + scaling (+ column block)");
  w->MP = ((m + 31) / 32) * 32;
  w->LV = ((w->MP + nl + 1 + 31) / 32) * 32;
  if (hipStreamCreate(&w->stream) != hipSuccess) return bail("hipStreamCreate failed");
  if (hipEventCreate(&w->ev_a) != hipSuccess || hipEventCreate(&w->ev_b) != hipSuccess) return bail("hipEventCreate failed");
  for (hipEvent_t &e : w->ev_ph) if (hipEventCreate(&e) != hipSuccess) return bail("hipEventCreate failed");
  { // matrices
    host::HostCsr hAt, hA, hQ;
    // the column form IS the work's copy of A: lent to the upload instead of copied (22 M entries on the LASSO protocol)
    hAt.nrows = w->A.n; hAt.ncols = w->A.m; hAt.ptr.swap(w->A.p); hAt.idx.swap(w->A.i); hAt.val.swap(w->A.x);
    host::build_row_blocks(hAt, CHUNK);
    bool up_bad = w->dAt.upload(hAt, w->stream) != 0;
    hAt.ptr.swap(w->A.p); hAt.idx.swap(w->A.i); hAt.val.swap(w->A.x);
    // the row form: large operators are transposed on the device from the column form just uploaded (dev_transpose.hip: the same arrays entry for entry, no second
    // pass over the matrix on the host and half the upload); small ones, or a device transpose that found no room, take the host's counting sort
    const long nnzA = w->A.p[w->A.n];
    const char *dt = getenv("ABIP_HIP_DEV_TRANSPOSE"); // 0: never, 1: always (tests)
    bool on_dev = !up_bad && nnzA > 0 && (dt ? atoi(dt) != 0 : nnzA >= 2000000);
    if (on_dev && w->dA.from_columns(w->dAt, w->A.m, nnzA, CHUNK, w->stream)) on_dev = false;
    if (!up_bad && !on_dev) { hcsr_from(w->A, hA, true); up_bad = w->dA.upload(hA, w->stream) != 0; }
    if (up_bad) return bail("device allocation failure");
    long nrb = std::max<long>(std::max<long>(w->dAt.nrb, w->dA.nrb), (std::max(m, nl) + 4 * BS - 1) / (4 * BS));
    if (w->hasQ) { hcsr_from(w->Q, hQ, false); if (w->dQ.upload(hQ, w->stream)) return bail("device allocation failure"); nrb = std::max<long>(nrb, w->dQ.nrb); }
    if (w->dist) // the replicated m-space kernels must sum in the same order on every rank: a grid that depends on global quantities only
      nrb = std::max<long>(((long)nnz_glob / w->world + CHUNK - 1) / CHUNK + 1, (std::max<long>(m, (long)n / w->world + 1) + 4 * BS - 1) / (4 * BS));
    const long per = (nrb + MAXNB - 1) / MAXNB;
    w->NB = (int)std::max<long>(1, (nrb + per - 1) / per);
  }
  phase("row / column forms of the operator, row blocks, upload");
  w->pcg = st->linsys_solver == 3;
  if (w->pcg) { // H = rho_x I + Q must be diagonal; Jacobi preconditioner M_i = 1 / (rho_y + sum_j A_ij^2 / H_jj)  (qcp_pcg.h)
    std::vector<double> Hinv(nl, w->kkt_rho_x), M(m, st->rho_y);
    if (w->hasQ) for (int j = 0; j < nl; ++j) for (int q = w->Q.p[j]; q < w->Q.p[j + 1]; ++q) {
      if (w->Q.i[q] == j) Hinv[j] += w->Q.x[q];
      else if (w->Q.x[q] != 0.0) return bail("linsys_solver = 3 (PCG) needs Q absent or diagonal; use linsys_solver = 1");
    }
    for (int j = 0; j < nl; ++j) Hinv[j] = 1.0 / Hinv[j];
    for (int j = 0; j < nl; ++j) for (int q = w->A.p[j]; q < w->A.p[j + 1]; ++q) M[w->A.i[q]] += w->A.x[q] * w->A.x[q] * Hinv[j];
    for (int i = 0; i < m; ++i) M[i] = 1.0 / M[i];
    if (w->dist) { // (the sums above saw this rank's columns only)
      M = Mpre;
      if (w->arbuf.alloc((size_t)m + w->world) || w->gsbuf.alloc((size_t)Q_COUNT + 6 * (size_t)w->world) ||
          hipMemsetAsync(w->arbuf.p, 0, sizeof(double) * ((size_t)m + w->world), w->stream) != hipSuccess)
        return bail("init_lin_sys_work failure");
    }
    if (w->cg_M.upload(M, w->stream) || w->cg_H.upload(Hinv, w->stream) || w->cg_x0.alloc(m) || w->cg_r.alloc(m) || w->cg_z.alloc(m) || w->cg_p.alloc(m) || w->cg_Gp.alloc(m) ||
        w->cg_tm.alloc(nl) || w->cg_part.alloc((size_t)PQ_COUNT * MAXNB) || hipMemsetAsync(w->cg_part.p, 0, sizeof(double) * PQ_COUNT * MAXNB, w->stream) != hipSuccess ||
        hipMemsetAsync(w->cg_tm.p, 0, sizeof(double) * nl, w->stream) != hipSuccess || hipHostMalloc((void **)&w->hlp, sizeof(Ctl), hipHostMallocDefault) != hipSuccess)
      return bail("init_lin_sys_work failure");
    if (hipMalloc((void **)&w->lp_ctl, sizeof(Ctl)) != hipSuccess || hipMemsetAsync(w->lp_ctl, 0, sizeof(Ctl), w->stream) != hipSuccess) return bail("allocation failure");
    { // A' y with y resident in LDS (kq_pcg_Aty_lds): where the m-vector fits and the product is large enough to be bound by its gathers.  ABIP_HIP_ATY_LDS=0 / 1 forces.
      const char *e = getenv("ABIP_HIP_ATY_LDS");
      const long nnzl = w->A.p[nl];
      int dev = 0, cus = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
      w->n_cu = cus;
      const bool fits = cus > 0 && m <= 16384 && nl >= 1;
      // Sharded: every rank must run the SAME kernel -- the two A'y kernels add |r|^2 and z'r in different orders, so ranks on different kernels would
      // differ in the last bit of beta and of the exit test, leave the PCG loop in different chunks and mismatch their collectives.  The choice therefore
      // reads global quantities only (non-zeros and columns of the whole matrix per rank), never this rank's block.
      const long nnz_dec = w->dist ? nnz_glob / w->world : nnzl;
      const long col_dec = w->dist ? std::max<long>(1, (long)w->n_glob / w->world) : (long)nl;
      (void)nnzl;
      if (fits && (e ? atoi(e) != 0 : nnz_dec >= 2000000)) { // (break-even measured at ~1e6 non-zeros, profiles/r02zj_*)
        const int G = (double)nnz_dec / (double)col_dec <= 96.0 ? 16 : 64;
        const int bytes = (int)(sizeof(double) * (size_t)m);
        bool ok = true;
        if (bytes > 48 * 1024) {
          if (G == 16) ok = hipFuncSetAttribute((const void *)kq_pcg_Aty_lds<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                            hipFuncSetAttribute((const void *)kq_pcg_Aty_lds<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
          else ok = hipFuncSetAttribute((const void *)kq_pcg_Aty_lds<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                    hipFuncSetAttribute((const void *)kq_pcg_Aty_lds<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        }
        if (ok) w->aty_lds = G; else (void)hipGetLastError();
      }
    }
  } else { // KKT upper triangle (qcp_config.c:699-748) -> LDL' -> level-scheduled device factors
    const int N = m + nl;
    const double rho_y = st->rho_y, rho_x = w->kkt_rho_x;
    std::vector<int> Kp(N + 1), Ki; std::vector<double> Kx;
    if (!w->hasQ) { // column m + i = -A(:, i) on top of rho_x: the extents are known up front, the columns filled by ranges of equal work (host_par.h)
      const long kk = (long)m + w->A.p[nl] + nl;
      Ki.resize(kk); Kx.resize(kk);
      for (int i = 0; i < m; ++i) { Kp[i] = i; Ki[i] = i; Kx[i] = -rho_y; }
      for (int i = 0; i <= nl; ++i) Kp[m + i] = m + w->A.p[i] + i;
      host::par_by_entries(w->A.p.data(), (long)nl, 500000, [&](long lo, long hi, int) {
        for (long i = lo; i < hi; ++i) {
          long dst = Kp[m + i];
          for (int j = w->A.p[i]; j < w->A.p[i + 1]; ++j, ++dst) { Ki[dst] = w->A.i[j]; Kx[dst] = -w->A.x[j]; }
          Ki[dst] = m + (int)i; Kx[dst] = rho_x;
        }
      });
    } else {
      Ki.reserve(N + w->A.p[nl] + w->Q.p[nl]); Kx.reserve(Ki.capacity());
      for (int i = 0; i < m; ++i) { Kp[i] = (int)Ki.size(); Ki.push_back(i); Kx.push_back(-rho_y); }
      for (int i = 0; i < nl; ++i) {
        Kp[m + i] = (int)Ki.size();
        for (int j = w->A.p[i]; j < w->A.p[i + 1]; ++j) { Ki.push_back(w->A.i[j]); Kx.push_back(-w->A.x[j]); }
        if (w->Q.p[i] == w->Q.p[i + 1]) { Ki.push_back(m + i); Kx.push_back(rho_x); }
        else for (int j = w->Q.p[i]; j < w->Q.p[i + 1]; ++j) {
          if (w->Q.i[j] > i) continue;
          const double t = (w->Q.i[j] == i) ? w->Q.x[j] + rho_x : w->Q.x[j];
          if (t == 0) continue; // cs_dropzeros
          Ki.push_back(m + w->Q.i[j]); Kx.push_back(t);
        }
      }
      Kp[N] = (int)Ki.size();
    }
    host::LdlHost F;
    std::vector<int> pmap(N);
    // x block diagonal and the Schur complement onto the y block dense anyway (pairs of non-zeros per column >= a quarter of the m (m - 1) / 2 entries):
    // eliminate the x block first, as the reference's reduced systems do, and skip the minimum-degree pass (ABIP_HIP_ORDER=md forces it)
    std::vector<int> order;
    {
      bool diagH = true;
      if (w->hasQ) for (int j = 0; j < nl && diagH; ++j) for (int q = w->Q.p[j]; q < w->Q.p[j + 1]; ++q) if (w->Q.i[q] != j && w->Q.x[q] != 0.0) { diagH = false; break; }
      double pairs = 0;
      for (int j = 0; j < nl; ++j) { const double kj = w->A.p[j + 1] - w->A.p[j]; pairs += 0.5 * kj * (kj - 1); }
      const char *e = getenv("ABIP_HIP_ORDER");
      const char *tm_ = getenv("ABIP_HIP_TAIL_MAX"), *tr_ = getenv("ABIP_HIP_TAIL");
      const bool tail_fits = m + 64 <= (tm_ ? atoi(tm_) : 24576) && !(tr_ && atoi(tr_) >= 0); // the whole y block must fit the dense tail (and the tail be chosen automatically)
      if (diagH && tail_fits && m >= 256 && pairs >= 0.125 * (double)m * (double)(m - 1) && !(e && !strcmp(e, "md"))) {
        order.resize(N);
        for (int j = 0; j < nl; ++j) order[j] = m + j;
        for (int i = 0; i < m; ++i) order[nl + i] = i;
      }
    }
    auto set_up = [&](int tail_request) -> int {
      if (tail_request != -2) host::set_tail_request(tail_request);
      host::set_order_hint(order.empty() || tail_request == 0 ? nullptr : &order); // (the fall-back without a dense tail wants a fill-reducing order)
      const int rc = host::factor_upper(N, Kp, Ki, Kx, F);
      host::set_order_hint(nullptr);
      host::set_tail_request(-2);
      if (rc < 0) return -2;
      for (int q = 0; q < N; ++q) pmap[q] = F.P[q] < m ? F.P[q] : w->MP + (F.P[q] - m);
      const double td = now_ms();
      const int rd = w->ldl.setup(F, pmap, w->stream) ? -1 : 0;
      if (getenv("ABIP_HIP_SETUP_TIMES")) printf("[setup] device part (uploads, Schur panels, dense LDL' + inverse of the tail) %.3f s\n", (now_ms() - td) / 1e3);
      return rd;
    };
    if (hipMalloc((void **)&w->lp_ctl, sizeof(Ctl)) != hipSuccess || hipMemsetAsync(w->lp_ctl, 0, sizeof(Ctl), w->stream) != hipSuccess) return bail("allocation failure");
    // set-up guard (see solver.hip: abip_init): one known right-hand side through the factor, ||K z - rhs|| checked on the host
    auto residual = [&]() -> double {
      std::vector<double> rhs, lv(w->LV, 0.0), z(N);
      host::guard_rhs(N, rhs);
      for (int i = 0; i < N; ++i) lv[i < m ? i : w->MP + (i - m)] = rhs[i];
      DBuf<double> tmp;
      if (tmp.alloc(w->LV)) return 1e300;
      double out = 1e300;
      if (hipMemcpyAsync(tmp.p, lv.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream) == hipSuccess) {
        w->ldl.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, w->stream, a...); }, tmp.p, (const Ctl *)w->lp_ctl, w->NB);
        if (hipMemcpyAsync(lv.data(), tmp.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream) == hipSuccess && hipStreamSynchronize(w->stream) == hipSuccess) {
          for (int i = 0; i < N; ++i) z[i] = lv[i < m ? i : w->MP + (i - m)];
          out = host::sym_upper_residual(N, Kp, Ki, Kx, z, rhs);
#ifdef ABIP_HIP_TEST_HOOKS
          if (getenv("ABIP_HIP_TAIL_RESID_FAIL") && F.T > 0) out = 1.0; // pretend the tail is too ill-conditioned for its explicit inverse
#endif
        }
      }
      tmp.release();
      return out;
    };
    constexpr double kGuardTol = 1e-8;
    int rc = set_up(-2);
    double res = rc == 0 ? residual() : 1e300;
    bool ok = rc == 0 && res <= kGuardTol;
    if (!ok && rc != -2 && F.T > 0) { // no dense tail then
      if (st->verbose) printf("dense tail rejected (T = %d, set-up residual %.2e): using the level-scheduled factor\n", F.T, res);
      (void)hipGetLastError();
      w->ldl.release();
      ok = set_up(0) == 0 && residual() <= kGuardTol;
    }
    if (!ok) { printf("\nerror in LDL factorization\n"); return bail("init_lin_sys_work failure"); }
  }
  phase("KKT back-end (preconditioner, or assembly + LDL')");
  DBuf<double> *lv[] = {&w->u, &w->v, &w->vo, &w->ut, &w->rel, &w->r, &w->p};
  for (auto *b : lv) { if (b->alloc(w->LV)) return bail("work memory allocation failure"); if (hipMemsetAsync(b->p, 0, sizeof(double) * w->LV, w->stream) != hipSuccess) return bail("memset failure"); }
  if (w->bd.upload(w->b, w->stream) || w->cd.upload(w->c, w->stream) || w->Dd.upload(w->D, w->stream) || w->Ed.upload(w->E, w->stream) || w->Ax.alloc(m) ||
      w->ATy.alloc(nl) || w->Qx.alloc(nl) || w->part.alloc((size_t)2 * Q_COUNT * MAXNB) || w->ctl.alloc(1))
    return bail("work memory allocation failure");
  if (kind == 1 && (w->sv.Dd.upload(w->sv.D, w->stream) || w->sv.Ed.upload(w->sv.E, w->stream) || w->sv.wEd.upload(w->sv.wE, w->stream))) return bail("work memory allocation failure");
  if (kind == 0 && (w->ls.Dd.upload(w->ls.D, w->stream) || w->ls.Ed.upload(w->ls.E, w->stream) || w->ls.yd.upload(w->ls.y, w->stream))) return bail("work memory allocation failure");
  if (hipMemsetAsync(w->part.p, 0, sizeof(double) * 2 * Q_COUNT * MAXNB, w->stream) != hipSuccess || hipMemsetAsync(w->Qx.p, 0, sizeof(double) * nl, w->stream) != hipSuccess ||
      hipMemsetAsync(w->ctl.p, 0, sizeof(QCtl), w->stream) != hipSuccess || hipHostMalloc((void **)&w->hctl, sizeof(QCtl), hipHostMallocDefault) != hipSuccess)
    return bail("work memory allocation failure");
  // cone layout (abip.c:355-409, same `count` walk) and the start point (update_work, abip.c:912-985)
  std::vector<int> xkind(nl, XK_NONE), c_off, c_len, c_kind;
  std::vector<double> hu(w->LV, 0.0);
  {
    int count = 0;
    double *x = hu.data() + w->MP;
    for (int i = 0; KK->q && i < KK->qsize; ++i) {
      const int len = KK->q[i];
      if (len == 0) continue;
      if (len == 1) xkind[count] = XK_ORTHANT;
      else { for (int t = 0; t < len; ++t) xkind[count + t] = XK_CONE; c_off.push_back(count); c_len.push_back(len); c_kind.push_back(0); }
      for (int t = 0; t < len; ++t) x[count + t] = 0;
      x[count] = 1;
      count += len;
    }
    for (int i = 0; KK->rq && i < KK->rqsize; ++i) {
      const int len = KK->rq[i];
      if (len < 3) continue; // sic: count is not advanced (abip.c:379-381, 944-946)
      for (int t = 0; t < len; ++t) { xkind[count + t] = XK_CONE; x[count + t] = 0; }
      c_off.push_back(count); c_len.push_back(len); c_kind.push_back(1);
      x[count] = 1; x[count + 1] = 1;
      count += len;
    }
    for (int t = 0; t < KK->f && count + t < nl; ++t) { xkind[count + t] = XK_FREE; x[count + t] = 0; }
    count += KK->f;
    for (int t = 0; t < KK->z && count + t < nl; ++t) { xkind[count + t] = XK_ZERO; x[count + t] = 0; }
    count += KK->z;
    for (int t = 0; t < KK->l && count + t < nl; ++t) { xkind[count + t] = XK_ORTHANT; x[count + t] = 1; }
    hu[w->MP + nl] = 1.0;
  }
  w->ncones = (int)c_off.size();
  { // small cones first (one wavefront each), then the large ones (one workgroup each)
    std::vector<int> ord(w->ncones);
    for (int q = 0; q < w->ncones; ++q) ord[q] = q;
    std::stable_partition(ord.begin(), ord.end(), [&](int q) { return c_len[q] <= QC_BIG; });
    std::vector<int> o2(w->ncones), l2(w->ncones), k2(w->ncones);
    w->nsmall = 0;
    for (int q = 0; q < w->ncones; ++q) { o2[q] = c_off[ord[q]]; l2[q] = c_len[ord[q]]; k2[q] = c_kind[ord[q]]; if (l2[q] <= QC_BIG) w->nsmall++; }
    c_off.swap(o2); c_len.swap(l2); c_kind.swap(k2);
  }
  if (w->xkind.upload(xkind, w->stream) || w->c_off.upload(c_off, w->stream) || w->c_len.upload(c_len, w->stream) || w->c_kind.upload(c_kind, w->stream))
    return bail("work memory allocation failure");
  if (hipMemcpyAsync(w->u.p, hu.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream) != hipSuccess ||
      hipMemcpyAsync(w->v.p, hu.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream) != hipSuccess)
    return bail("upload failure");
  // pre_calculate, abip.c:886-910: r = K^-1 (b ; c)  [the solve negates the first block of (-b ; c)] ; a = rho_tau + r' (rho o r)
  {
    std::vector<double> hr(w->LV, 0.0);
    for (int i = 0; i < m; ++i) hr[i] = w->b[i]; // -(-b)
    for (int j = 0; j < nl; ++j) hr[w->MP + j] = w->c[j];
    if (hipMemcpyAsync(w->r.p, hr.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream) != hipSuccess) return bail("upload failure");
    if (w->pcg) { if (solve_pcg(w, w->r.p, false, -1, 1e-12) < 0) return bail("device failure in pre_calculate"); } // abip.c:899
    else enqueue_solve(w, w->r.p);
    if (hipMemcpyAsync(hr.data(), w->r.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess)
      return bail("device failure in pre_calculate");
    double acc = 0;
    for (int i = 0; i < m; ++i) acc += (hr[i] * st->rho_y) * hr[i];
    acc *= w->wy;
    for (int j = 0; j < nl; ++j) acc += (hr[w->MP + j] * st->rho_x) * hr[w->MP + j];
    if (w->dist) { // the x block is sharded: sum the shares (through the exchange area: one double)
      if (hipMemcpyAsync(w->gsbuf.p, &acc, sizeof(double), hipMemcpyHostToDevice, w->stream) != hipSuccess || ar(w, w->gsbuf.p, 1) ||
          hipMemcpyAsync(&acc, w->gsbuf.p, sizeof(double), hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess)
        return bail("device failure in pre_calculate");
    }
    w->a_quad = st->rho_tau + acc;
  }
  phase("work vectors, cone tables, the set-up solve");
  info->setup_time = now_ms() - t_init;
  const double t0 = now_ms();
  const double time_limit_left = 1e3 * st->time_limit - info->setup_time;
  // several GPUs: the ranks must take the same branch and their clocks differ -- with a finite limit the verdicts are summed over the ranks (one more
  // collective per test; none when time_limit is infinite, the default)
  bool time_failed = false;
  auto over_time = [&]() -> bool {
    const bool mine = (now_ms() - t0) > time_limit_left;
    if (!w->dist || !std::isfinite(st->time_limit)) return mine;
    double flag = mine ? 1.0 : 0.0;
    if (hipMemcpyAsync(w->gsbuf.p, &flag, sizeof(double), hipMemcpyHostToDevice, w->stream) != hipSuccess || ar(w, w->gsbuf.p, 1) ||
        hipMemcpyAsync(&flag, w->gsbuf.p, sizeof(double), hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) { time_failed = true; return true; }
    return flag > 0;
  };

  QResid r;
  info->status_val = 0;
  double tol_inner = 4 * std::pow(w->mu, st->psi);
  const QDims dm{m, nl, w->MP, w->kind == 0 ? 1 : (w->kind == 1 ? 2 : 0), w->wy};
  int i = 0, j = 0, k = 0;
  bool finished = false;
  auto get_solution = [&](int ipm_iter, int admm_iter) -> int { // abip.c:559-587
    std::vector<double> hu2(w->LV), hv2(w->LV);
    if (hipMemcpyAsync(hu2.data(), w->u.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
        hipMemcpyAsync(hv2.data(), w->v.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess)
      return -1;
    // the conic (x, y, s) of the formulation; what the caller receives depends on the formulation (below)
    const int nx = w->dist ? n : nl; // several GPUs: x and s are gathered into the whole problem's order, every rank returns all of them
    std::vector<double> X(nx), Y(m), S(nx);
    if (w->dist) {
      DBuf<double> gx;
      if (gx.alloc((size_t)2 * n)) return -1;
      int bad = hipMemsetAsync(gx.p, 0, sizeof(double) * 2 * (size_t)n, w->stream) != hipSuccess;
      if (!bad) {
        QLAUNCH(w, kq_dist_place, w->NB, BS, (const double *)(w->u.p + w->MP), nl, w->n0, gx.p);
        QLAUNCH(w, kq_dist_place, w->NB, BS, (const double *)(w->v.p + w->MP), nl, w->n0, gx.p + n);
        bad = ar(w, gx.p, (size_t)2 * n) || hipMemcpyAsync(X.data(), gx.p, sizeof(double) * n, hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
              hipMemcpyAsync(S.data(), gx.p + n, sizeof(double) * n, hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess;
      }
      gx.release();
      if (bad) return -1;
    } else for (int q = 0; q < nl; ++q) { X[q] = hu2[w->MP + q]; S[q] = hv2[w->MP + q]; }
    const std::vector<double> &Eu = w->dist ? w->Efull : w->E;
    for (int q = 0; q < m; ++q) Y[q] = hu2[q];
    const int sv = info->status_val;
    if (sv == 0 || sv == 1 || sv == 2) {
      const double sc = safediv_pos(1.0, r.tau);
      for (int q = 0; q < nx; ++q) { X[q] *= sc; S[q] *= sc; }
      for (int q = 0; q < m; ++q) Y[q] *= sc;
      if (sv == 0 || sv == 2) { strcpy(info->status, "Solved/Inaccurate"); info->status_val = 2; } else { strcpy(info->status, "Solved"); info->status_val = 1; }
    } else if (sv == -2 || sv == -7) {
      const double bty = r.dobj * r.tau;
      for (int q = 0; q < m; ++q) Y[q] *= 1 / bty;
      for (int q = 0; q < nx; ++q) { S[q] *= 1 / bty; X[q] = NAN; }
      strcpy(info->status, "Infeasible"); info->status_val = -2;
    } else {
      const double ctx = r.pobj * r.tau;
      for (int q = 0; q < nx; ++q) { X[q] *= -1 / ctx; S[q] = NAN; }
      for (int q = 0; q < m; ++q) Y[q] = NAN;
      strcpy(info->status, "Unbounded"); info->status_val = -1;
    }
    if (kind == 0) { // un_scaling_lasso_sol, lasso_config.c:296-311: beta = E o (beta+ - beta-) / sc_b is all the caller gets (x: dn entries)
      const LassoForm &L = w->ls;
      if (!sol->x) sol->x = (qcp_float *)malloc(sizeof(qcp_float) * L.dn);
      for (int jx = 0; jx < L.dn; ++jx) sol->x[jx] = (X[L.dm + 2 + jx] + (-1) * X[L.dm + L.dn + 2 + jx]) * L.E[jx] * (1 / L.sc_b);
    } else if (kind == 1) { // un_scaling_svm_sol, svm_config.c:410-440: x: w (dn), y: b (1), s: xi (dm)
      const SvmForm &V = w->sv;
      if (!sol->x) sol->x = (qcp_float *)malloc(sizeof(qcp_float) * V.dn);
      if (!sol->y) sol->y = (qcp_float *)malloc(sizeof(qcp_float));
      if (!sol->s) sol->s = (qcp_float *)malloc(sizeof(qcp_float) * V.dm);
      for (int q = 0; q < V.dn; ++q) sol->x[q] = (X[V.dn + 2 + q] + (-1) * X[2 * V.dn + 3 + q]) * V.E[q] * (1 / V.sc_b);
      sol->y[0] = (X[2 * V.dn + 2] - X[3 * V.dn + 3]) * V.E[V.dn] / V.sc_b;
      for (int q = 0; q < V.dm; ++q) sol->s[q] = X[3 * V.dn + 4 + q] * (1 / (V.sc_b * V.sc_c));
    } else if (kind == 3) { // un_scaling_svmqp_sol, svm_qp_config.c:595-619: x / (E sc_b), then x: w (dn), y: b (1), s: xi (dm)
      const SvmForm &V = w->sv;
      for (int q = 0; q < nx; ++q) X[q] /= (Eu[q] * w->sc_b);
      if (!sol->x) sol->x = (qcp_float *)malloc(sizeof(qcp_float) * V.dn);
      if (!sol->y) sol->y = (qcp_float *)malloc(sizeof(qcp_float));
      if (!sol->s) sol->s = (qcp_float *)malloc(sizeof(qcp_float) * V.dm);
      for (int q = 0; q < V.dn; ++q) sol->x[q] = X[q];
      sol->y[0] = X[V.dn];
      for (int q = 0; q < V.dm; ++q) sol->s[q] = X[V.dn + 1 + q];
    } else {
      if (st->normalize) { // un_scaling_qcp_sol, qcp_config.c:496-513
        for (int q = 0; q < nx; ++q) X[q] /= (Eu[q] * w->sc_b);
        for (int q = 0; q < m; ++q) Y[q] /= (w->D[q] * w->sc_c);
        for (int q = 0; q < nx; ++q) S[q] *= Eu[q] / (w->sc_c * st->scale);
      }
      if (!sol->x) sol->x = (qcp_float *)malloc(sizeof(qcp_float) * nx);
      if (!sol->y) sol->y = (qcp_float *)malloc(sizeof(qcp_float) * m);
      if (!sol->s) sol->s = (qcp_float *)malloc(sizeof(qcp_float) * nx);
      for (int q = 0; q < nx; ++q) { sol->x[q] = X[q]; sol->s[q] = S[q]; }
      for (int q = 0; q < m; ++q) sol->y[q] = Y[q];
    }
    info->ipm_iter = ipm_iter + 1; info->admm_iter = admm_iter;
    info->res_infeas = r.res_infeas; info->res_unbdd = r.res_unbdd;
    if (info->status_val == 1 || info->status_val == 2) { info->rel_gap = r.rel_gap; info->res_pri = r.res_pri; info->res_dual = r.res_dual; info->pobj = r.pobj; info->dobj = r.dobj; }
    else if (info->status_val == -1) { info->rel_gap = info->res_pri = info->res_dual = NAN; info->pobj = info->dobj = -INFINITY; }
    else { info->rel_gap = info->res_pri = info->res_dual = NAN; info->pobj = info->dobj = INFINITY; }
    info->solve_time = now_ms() - t0;
    return 0;
  };
  auto stop_now = [&](int ii) { const bool ot = over_time(); return (double)k + 1 >= (double)st->max_admm_iters * st->max_ipm_iters || ii + 1 >= st->max_ipm_iters || ot; };

  const Ctl *hc = w->lp_ctl;
  const bool batch_ok = !(getenv("ABIP_HIP_BATCH") && atoi(getenv("ABIP_HIP_BATCH")) == 0);
  const bool qmerge = !(getenv("ABIP_HIP_QMERGE") && atoi(getenv("ABIP_HIP_QMERGE")) == 0);
  int seen = 0; // QCtl.it_count at the last control read
  bool pcg_failed = false;
  // one inner iteration (abip.c:1120-1160), everything on the stream; `timed` brackets the KKT solve with events
  auto enqueue_iteration = [&](int kk, bool timed) {
    // projection, abip.c:186-255
    if (timed) (void)hipEventRecord(w->ev_ph[0], w->stream);
    QLAUNCH(w, kq_rhs, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->r.p, w->p.p, st->rho_y, st->rho_x, dm, w->part.p, hc);
    if (timed) (void)hipEventRecord(w->ev_a, w->stream);
    if (w->pcg) { if (solve_pcg(w, w->p.p, true, kk, std::min(r.Ax_b_norm, r.Qx_ATy_c_s_norm)) < 0) pcg_failed = true; } // abip.c:206-224
    else enqueue_solve(w, w->p.p);
    if (timed) (void)hipEventRecord(w->ev_b, w->stream);
    QLAUNCH(w, kq_dots, w->NB, BS, (const double *)w->r.p, (const double *)w->p.p, st->rho_y, st->rho_x, dm, w->part.p, hc);
    if (w->hasQ) QLAUNCH(w, kq_Qp, w->NB, BS, w->dQ.view(), (const double *)w->p.p, dm, w->part.p, hc);
    if (timed) (void)hipEventRecord(w->ev_ph[1], w->stream); // end of project_lin_sys
    QProxArgs pa;
    pa.u = w->u.p; pa.v = w->v.p; pa.ut = w->ut.p; pa.rel = w->rel.p; pa.p = w->p.p; pa.r = w->r.p; pa.xkind = w->xkind.p;
    pa.alpha = st->alpha; pa.lambda = w->mu / w->beta; pa.rho_x = st->rho_x; pa.rho_tau = st->rho_tau; pa.a_quad = w->a_quad; pa.iter_pos = kk > 0; pa.hasQ = w->hasQ;
    const double *gsc = nullptr;
    if (w->dist) { // r'mu, r'(rho o p), p_x'Qp run over the sharded n-space: reduce locally, exchange, hand the totals to kq_ut_prox
      finalize(w, {Q_T0, Q_T1, Q_PG}, {});
      if (exchange_sums(w, {Q_T0, Q_T1, Q_PG})) w->dist_failed = true;
      gsc = reinterpret_cast<const double *>(w->ctl.p);
    }
    QLAUNCH(w, kq_ut_prox, w->NB, BS, pa, dm, (const double *)w->part.p, w->NB, w->ctl.p, hc, gsc);
    if (w->ncones) {
      const double lam = (w->mu / w->beta) / st->rho_x;
      if (w->nsmall) { QCones C{w->c_off.p, w->c_len.p, w->c_kind.p, w->nsmall}; QLAUNCH(w, kq_cones<false>, (w->nsmall + WAVES - 1) / WAVES, BS, C, 0, w->u.p, (const double *)w->rel.p, lam, w->MP, hc); }
      if (w->ncones > w->nsmall) { QCones C{w->c_off.p, w->c_len.p, w->c_kind.p, w->ncones}; QLAUNCH(w, kq_cones<true>, w->ncones - w->nsmall, QC_TB, C, w->nsmall, w->u.p, (const double *)w->rel.p, lam, w->MP, hc); }
    }
    if (timed) (void)hipEventRecord(w->ev_ph[2], w->stream); // end of solve_barrier_subproblem
    QLAUNCH(w, kq_dual, w->NB, BS, w->u.p, (const double *)w->rel.p, w->v.p, w->vo.p, st->rho_y, st->rho_x, st->rho_tau, dm, (const QCtl *)w->ctl.p, hc);
    if (timed) (void)hipEventRecord(w->ev_ph[3], w->stream); // end of the dual update ("updating work")
    // inner stopping test, qcp_config.c:518-557 (the tau entries and the comparison with tol_inner happen in kq_finalize)
    if (w->dist) { // A'u_y is local; A u_x = sum over the ranks of the block products, then the y-block sums on the replicated result
      QLAUNCH(w, kq_inner_At, w->NB, BS, w->dAt.view(), (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->cd.p, w->ATy.p, w->Qx.p, w->hasQ ? 0 : 1, dm, w->part.p, hc);
      QLAUNCH(w, kq_prod_A, w->NB, BS, w->dA.view(), (const double *)(w->u.p + w->MP), (const double *)nullptr, w->arbuf.p, 0, hc);
      if (ar(w, w->arbuf.p, (size_t)m)) w->dist_failed = true;
      QLAUNCH(w, kq_dist_inner_A_fin, w->NB, BS, (const double *)w->arbuf.p, (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->bd.p, w->Ax.p, dm, w->part.p, hc);
    } else if (qmerge) {
      QLAUNCH(w, kq_inner_both, 2 * w->NB, BS, w->dA.view(), w->dAt.view(), (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->bd.p,
              (const double *)w->cd.p, w->Ax.p, w->ATy.p, w->Qx.p, w->hasQ ? 0 : 1, dm, w->NB, w->part.p, hc);
    } else {
      QLAUNCH(w, kq_inner_A, w->NB, BS, w->dA.view(), (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->bd.p, w->Ax.p, dm, w->part.p, hc);
      QLAUNCH(w, kq_inner_At, w->NB, BS, w->dAt.view(), (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->cd.p, w->ATy.p, w->Qx.p, w->hasQ ? 0 : 1, dm, w->part.p, hc);
    }
    if (w->hasQ) QLAUNCH(w, kq_inner_Q, w->NB, BS, w->dQ.view(), (const double *)w->u.p, (const double *)w->vo.p, (const double *)w->cd.p, (const double *)w->ATy.p, w->Qx.p, dm, w->part.p, hc);
    finalize(w, {Q_D1, Q_D2, Q_D3, Q_E1, Q_E2, Q_E3}, {Q_D1, Q_D3, Q_E1, Q_E2, Q_E3}, tol_inner);
    if (timed) (void)hipEventRecord(w->ev_ph[4], w->stream); // end of err_inner
  };

  for (i = 0; i < st->max_ipm_iters && !finished; ++i) {
    int batch = 2;
    for (j = 0; j < st->max_admm_iters;) {
      // Between residual checks the host has nothing to decide but the inner exit, and that is found on the device: enqueue a
      // batch of iterations and read the control block once (the iterations behind the exit fall through on the halt flag).
      int nb = 1;
      if (batch_ok && !w->pcg && r.error_ratio > 8) { // (the PCG back-end returns to the host inside every solve: no batching)
        const int to_check = st->inner_check_period - (j % st->inner_check_period); // the iteration with (j+1) % period == 0 closes a batch
        nb = std::max(1, std::min(std::min(batch, to_check), (int)st->max_admm_iters - j));
      }
      for (int q = 0; q < nb; ++q) enqueue_iteration(k + q, q == 0);
      if (read_ctl(w) || pcg_failed || w->dist_failed) return bail("device error in the inner iteration");
      { float ms = 0.f; if (hipEventElapsedTime(&ms, w->ev_a, w->ev_b) == hipSuccess) { w->lin_ms += ms; w->lin_n++; } }
      { float ph[4]; bool ok = true;
        for (int q = 0; q < 4; ++q) ok = ok && hipEventElapsedTime(&ph[q], w->ev_ph[q], w->ev_ph[q + 1]) == hipSuccess;
        if (ok) { for (int q = 0; q < 4; ++q) w->ph_ms[q] += ph[q]; w->ph_n++; } else (void)hipGetLastError(); }
      const int ran = w->hctl->it_count - seen;
      seen = w->hctl->it_count;
      if (ran < 1 || ran > nb) return bail("device error in the inner iteration");
      k += ran;
      const int j_last = j + ran - 1;
      const bool halted = w->hctl->halted != 0;
      if (halted && clear_halt(w)) return bail("device error in the inner iteration");
      { const bool ot = over_time(); if (time_failed) return bail("collective failure"); if (halted || ot) { j = j_last; break; } } // err_inner < tol_inner, abip.c:1147
      if ((j_last + 1) % st->inner_check_period == 0 || r.error_ratio <= 8) {
        { const double tr0 = now_ms(); const int rc = calc_residuals(w, r, i, k); w->res_ms += now_ms() - tr0; if (rc) return bail("device error in calc_residuals"); }
        if ((info->status_val = has_converged(w, r, i, k)) != 0 || stop_now(i)) {
          if (get_solution(i, k)) return bail("device error in get_solution");
          finished = true;
          j = j_last;
          break;
        }
      }
      j = j_last + 1;
      batch = std::min(32, batch * 2);
    }
    if (finished) break;
    if (w->sparsity || (i + 1) % st->outer_check_period == 0) {
      { const double tr0 = now_ms(); const int rc = calc_residuals(w, r, i, k); w->res_ms += now_ms() - tr0; if (rc) return bail("device error in calc_residuals"); }
      if ((info->status_val = has_converged(w, r, i, k)) != 0 || stop_now(i)) {
        if (get_solution(i, k)) return bail("device error in get_solution");
        finished = true;
        break;
      }
    }
    tol_inner = adjust_barrier(w, r);
  }
  { // per-phase totals in the reference's order and units (seconds; abip.c:1196-1201)
    const double sc = w->ph_n ? (double)k / (double)w->ph_n * 1e-3 : 0.0;
    g_phase[0] = w->ph_ms[0] * sc; g_phase[1] = w->ph_ms[1] * sc; g_phase[2] = w->res_ms * 1e-3; g_phase[3] = w->ph_ms[3] * sc; g_phase[4] = w->ph_ms[2] * sc;
    if (st->verbose)
      printf("\ntotal time of project_lin_sys: %.2es\ntotal time of solve_barrier_subproblem: %.2es\ntotal time of calculate res: %.2es\ntotal time of calculate err_inner: %.2es\n"
             "total time of updating work: %.2es\n(device phases: hipEvents around one iteration per control read, %ld of %d iterations, scaled)\n",
             g_phase[0], g_phase[1], g_phase[2], g_phase[3], g_phase[4], w->ph_n, (int)k);
  }
  info->avg_linsys_time = w->lin_n ? w->lin_ms / (double)w->lin_n : 0; info->avg_cg_iters = w->cg_solves ? (double)w->tot_cg / (double)w->cg_solves : 0; // ms per solve, as lin_sys_time_per_iter (abip.c:1228)
  g_stats[0] = w->ldl.N; g_stats[1] = w->ldl.T; g_stats[2] = (double)w->ldl.lnnz; g_stats[3] = w->ldl.F.nlev; g_stats[4] = w->ldl.B.nlev;
  g_stats[5] = (double)w->lin_n; g_stats[6] = w->lin_ms; g_stats[7] = w->dist ? (double)w->n_allreduce : (double)(w->ldl.F.idx.n + w->ldl.B.idx.n); // (sharded runs: collectives issued)
  release(w);
  return info->status_val;
}

// Pure host code: how the dense tail's triangle is dealt to the wavefronts of its stream (dev_tail.h SymPlan).  out4 = {column chunks, units of four rows, wavefronts,
// column-partial slots}; pre (chunks + 1), qlo / qhi (chunks) as the kernels get them.  For tests/test_tail_plan_cpu.py.  Returns 0, -1 where there is no plan.
__attribute__((visibility("default"))) int abip_hip_tail_plan(int T, int waves, int *out4, int *pre, int *qlo, int *qhi) { // (declared in abip_hip.h, which this file does not include: the export is marked here)
  if (!out4) return -1;
  abip::SymPlan p;
  if (!p.make(T, waves)) return -1;
  out4[0] = p.ncc; out4[1] = p.nu; out4[2] = p.nwv; out4[3] = p.slots;
  if (pre) std::copy(p.pre, p.pre + p.ncc + 1, pre);
  if (qlo) std::copy(p.qlo, p.qlo + p.ncc, qlo);
  if (qhi) std::copy(p.qhi, p.qhi + p.ncc, qhi);
  return 0;
}
} // extern "C"
