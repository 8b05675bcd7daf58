// qcp_work.h -- host-side state of one conic solve (qcp_solver.hip) and what the formulation front ends (qcp_formulations.h) fill in.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/abip_qcp.h"
#include "dev_host_util.h"
#include "dev_ldl.h"
#include "qcp_kernels.h"

namespace abip {
namespace qcp {
using namespace abip::hostutil;

struct HMat { int m = 0, n = 0; std::vector<int> p, i; std::vector<double> x; };
inline void copy_in(HMat &dst, const QCPMatrix *src) {
  dst.m = src->m; dst.n = src->n;
  const int nnz = src->p[src->n];
  dst.p.assign(src->p, src->p + src->n + 1); dst.i.assign(src->i, src->i + nnz); dst.x.assign(src->x, src->x + nnz);
}
inline double vnrminf(const double *a, long n) { double mx = 0; for (long k = 0; k < n; ++k) { const double t = std::fabs(a[k]); if (t >= mx) mx = t; } return mx; }

struct QResid { // struct ABIP_RESIDUALS, abip.h:182-207
  int last_ipm_iter = -1, last_admm_iter = -1;
  double res_pri = 1e8, res_dual = 1e8, rel_gap = 1e8, res_infeas = 0, res_unbdd = 0, pobj = 0, dobj = 0, tau = 0, kap = 0, res_dif = 0,
         error_ratio = 1e8, Ax_b_norm = INFINITY, Qx_ATy_c_s_norm = INFINITY; // (the two norms: unknown until the first residual check -- qcp_pcg.h)
};

// The LASSO reformulation (lasso_config.c): what the front end keeps of the caller's data and of its own scaling
struct LassoForm {
  int dm = 0, dn = 0; // rows (samples) and columns (features) of the data matrix X
  double lambda = 0, sc = 1, sc_b = 1, sc_c = 1, sc_cone1 = 1, sc_cone2 = 1;
  std::vector<double> D, E, y;
  DBuf<double> Dd, Ed, yd;
};

// The SVM reformulations (svm_qp_config.c; svm_config.c): data dimensions and what the un-scaling needs
struct SvmForm {
  int dm = 0, dn = 0; double lambda = 0;
  double sc = 1, sc_b = 1, sc_c = 1, sc_cone1 = 1, sc_cone2 = 1; // SVM-SOCP only (svm_config.c:63-107)
  std::vector<double> D, E, wE;
  DBuf<double> Dd, Ed, wEd;
};

struct QWk {
  SvmForm sv;
  int kind = 2; // enum problem_type as abip() maps settings.prob_type (abip.c:1341-1348): 0 LASSO, 1 SVM as an SOCP, 2 generic QCP, 3 SVM as a QP
  LassoForm ls;
  double kkt_rho_x = 1; // the rho_x the KKT system is assembled with (the LASSO solve hard-codes 1, lasso_config.c:652-708)
  int m = 0, n = 0, MP = 0, LV = 0, NB = 1;
  const QCPSettings *st = nullptr;
  bool hasQ = false;
  int sparsity = 0;
  HMat A, Q;
  std::vector<double> D, E, b, c;
  double sc_b = 1, sc_c = 1, nm_inf_b = 0, nm_inf_c = 0, a_quad = 0, mu = 1, beta = 1;
  hipStream_t stream = nullptr;
  DevCsr dA, dAt, dQ;
  DBuf<double> u, v, vo, ut, rel, r, p, bd, cd, Dd, Ed, Ax, ATy, Qx, part;
  DBuf<int> xkind, c_off, c_len, c_kind;
  DBuf<QCtl> ctl;
  QCtl *hctl = nullptr;
  DevLdl ldl;
  hipEvent_t ev_a = nullptr, ev_b = nullptr; // bracket of the KKT solve of the current iteration (avg_linsys_time)
  double lin_ms = 0; long lin_n = 0;
  // the reference's per-phase timers (abip.c:1084-1093, printed at 1196-1201): project_lin_sys, solve_barrier_subproblem, updating work, err_inner --
  // hipEvents at the phase boundaries of the ONE iteration per control read that is bracketed anyway (sampled; totals = mean x iterations);
  // calc_residuals by the host clock around its calls (it ends in a synchronisation)
  hipEvent_t ev_ph[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double ph_ms[4] = {0, 0, 0, 0}; long ph_n = 0; double res_ms = 0;
  int ncones = 0, nsmall = 0; // cone table: the nsmall cones of <= QC_BIG entries first
  Ctl *lp_ctl = nullptr; // dev_sptrsv kernels are gated on an LP-style control block (halt flag): a zeroed one
  // indirect back-end (linsys_solver = 3, qcp_pcg.h)
  bool pcg = false;
  DBuf<double> cg_x0, cg_r, cg_z, cg_p, cg_Gp, cg_tm, cg_M, cg_H, cg_part; // m-space: y0, r, z, p, Gp, M; n-space: tn (cg_tm), H^-1
  Ctl *hlp = nullptr;    // pinned mirror of lp_ctl
  int last_cg = 8; long tot_cg = 0, cg_solves = 0;
  int aty_lds = 0, n_cu = 0; // qcp_pcg.h: kq_pcg_Aty_lds with 16 or 64 lanes per row (0: the streaming kernel), one workgroup per CU
  // several GPUs: this rank's column block [n0, n0 + n) of the n_glob columns (qcp_dist.h); m-space is replicated
  bool dist = false;
  int rank = 0, world = 1, n_glob = 0, n0 = 0;
  double wy = 1.0;                 // weight of the sums over the replicated y block: 1 on rank 0, 0 elsewhere
  std::vector<double> Efull;       // column scale of the whole problem (the gathered solution is un-scaled on every rank)
  DBuf<double> arbuf, gsbuf;       // exchange areas: m + world doubles (products + the max lanes of the warm start); Q_COUNT + 6 world (packed sums + max lanes)
  std::vector<double> hstage;      // staging of the callback transport
  long n_allreduce = 0;            // collectives issued (bench, tests)
  bool dist_failed = false;       // a collective failed inside an enqueue-only helper
};

#define QLAUNCH(w, kern, grid, block, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (w)->stream, __VA_ARGS__)

inline double g_stats[8] = {0};
inline double g_phase[5] = {0}; // seconds: project_lin_sys, solve_barrier_subproblem, calc_residuals, err_inner, updating work (the order the reference prints them in)

inline void release(QWk *w) {
  if (w->ev_a) (void)hipEventDestroy(w->ev_a);
  if (w->ev_b) (void)hipEventDestroy(w->ev_b);
  w->ev_a = w->ev_b = nullptr;
  for (hipEvent_t &e : w->ev_ph) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  w->dA.release(); w->dAt.release(); w->dQ.release();
  w->ls.Dd.release(); w->ls.Ed.release(); w->ls.yd.release();
  w->sv.Dd.release(); w->sv.Ed.release(); w->sv.wEd.release();
  DBuf<double> *bufs[] = {&w->u, &w->v, &w->vo, &w->ut, &w->rel, &w->r, &w->p, &w->bd, &w->cd, &w->Dd, &w->Ed, &w->Ax, &w->ATy, &w->Qx, &w->part};
  for (auto *b : bufs) b->release();
  w->xkind.release(); w->c_off.release(); w->c_len.release(); w->c_kind.release(); w->ctl.release();
  w->ldl.release();
  w->arbuf.release(); w->gsbuf.release();
  { DBuf<double> *cb[] = {&w->cg_x0, &w->cg_r, &w->cg_z, &w->cg_p, &w->cg_Gp, &w->cg_tm, &w->cg_M, &w->cg_H, &w->cg_part}; for (auto *b : cb) b->release(); }
  if (w->hlp) (void)hipHostFree(w->hlp);
  if (w->lp_ctl) (void)hipFree(w->lp_ctl);
  if (w->hctl) (void)hipHostFree(w->hctl);
  if (w->stream) (void)hipStreamDestroy(w->stream);
}

} // namespace qcp
} // namespace abip
