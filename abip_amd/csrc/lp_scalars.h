// lp_scalars.h -- the scalar decisions of ABIP-LP's outer iteration, ONE source for the host state machine (solver.hip) and for the
// persistent launch that takes them on the device (dev_xcd.h).
//
// Every function here is compiled for both sides with floating-point contraction OFF: the host (x86-64, no FMA in the default target) and
// the device (which would fuse a*b+c) must come to the same bits, because these numbers steer control flow -- has the solve converged, which
// mu comes next, which penalty the Barzilai-Borwein search settles on.  +, -, *, /, sqrt are correctly rounded on both sides; nothing else
// is used (no pow, no exp: the one rule that needs pow -- update_barrier_dynamic_2 -- is tabulated by the host, see xcd_mu_table).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define ABIP_HD __host__ __device__ inline
#else
#define ABIP_HD inline
#endif

namespace abip {

// ABIPResiduals, src/abip-lp/include/abip.h:178-195 (the fields the outer iteration reads)
struct LpResid {
  double res_pri, res_dual, rel_gap, res_infeas, res_unbdd, ct_x_by_tau, bt_y_by_tau, tau, kap;
};

// what calc_residuals needs of the finalised sums (Ctl::out / the persistent launch's copy of it); index = Slot (dev_common.h), 80..83 = the tau / kappa entries
struct LpSums {
  double ut, vt, rp, nax, rd, naty, by, cx;
};

ABIP_HD double lp_safediv(double x, double y) { return y < 1E-18 ? x / 1E-18 : x / y; } // SAFEDIV_POS, glbopts.h:157-160

// calc_residuals, abip.c:458-535 on sums the stopping test has already formed (D- / E-weighted squares, b'y, c'x)
ABIP_HD void lp_residuals(const LpSums &s, double den /* scale * sc_c * sc_b, or 1 */, double nm_b, double nm_c, LpResid &r) {
#pragma clang fp contract(off)
  r.tau = fabs(s.ut);
  r.kap = fabs(s.vt) / den;
  const double nmpr_tau = sqrt(s.rp), nm_A_x_tau = sqrt(s.nax);
  const double nmdr_tau = sqrt(s.rd), nm_At_ys_tau = sqrt(s.naty);
  r.bt_y_by_tau = s.by / den;
  r.ct_x_by_tau = s.cx / den;
  const double nan_ = NAN;
  r.res_infeas = r.bt_y_by_tau > 0 ? nm_b * nm_At_ys_tau / r.bt_y_by_tau : nan_;
  r.res_unbdd = r.ct_x_by_tau < 0 ? nm_c * nm_A_x_tau / -r.ct_x_by_tau : nan_;
  const double bt_y = lp_safediv(r.bt_y_by_tau, r.tau), ct_x = lp_safediv(r.ct_x_by_tau, r.tau);
  r.res_pri = lp_safediv(nmpr_tau / (1 + nm_b), r.tau);
  r.res_dual = lp_safediv(nmdr_tau / (1 + nm_c), r.tau);
  r.rel_gap = fabs(ct_x - bt_y) / (1 + fabs(ct_x) + fabs(bt_y));
}

// has_converged, abip.c:1613-1641: 1 solved, -1 unbounded, -2 infeasible, 0 go on (the values of glbopts.h:22-31)
ABIP_HD int lp_converged(const LpResid &r, double eps, int pfeasopt, long ipm_iter, long admm_iter) {
  if (r.res_pri < eps && (r.res_dual < eps || pfeasopt) && r.rel_gap < eps) return 1;
  if (r.res_unbdd < eps && ipm_iter > 0 && admm_iter > 0) return -1;
  if (r.res_infeas < eps && ipm_iter > 0 && admm_iter > 0) return -2;
  return 0;
}

// which barrier rule the outer iteration applies (abip.c:2251-2277); `dyn_sigma` is stgs->dynamic_sigma, which the rule itself overwrites
enum LpMuRule { LP_MU_NONE = 0, LP_MU_LOQO = 1, LP_MU_TABLE = 2 /* update_barrier, the "tedious" table */, LP_MU_DYN2 = 3 };
ABIP_HD int lp_mu_rule(int hybrid_mu, double dyn_sigma_second, double hybrid_thresh, double eps, double mu, double &dyn_sigma) {
#pragma clang fp contract(off)
  if (hybrid_mu) {
    if (dyn_sigma_second > 0.0 && mu < hybrid_thresh * eps) { dyn_sigma = dyn_sigma_second; return LP_MU_LOQO; }
    if (dyn_sigma_second == 0.0 && mu < hybrid_thresh * eps) { dyn_sigma = dyn_sigma_second; return LP_MU_TABLE; }
    if (dyn_sigma < 0.0) return LP_MU_DYN2;
    return LP_MU_NONE;
  }
  if (dyn_sigma == 0.0) return LP_MU_TABLE;
  if (dyn_sigma < 0.0) return LP_MU_DYN2;
  return LP_MU_LOQO;
}

// update_barrier_dynamic (the LOQO rule), abip.c:930-977, from the sum and the minimum of u_i v_i over the n + 1 entries i >= m; the factor mu is multiplied by
ABIP_HD double lp_loqo_sigma(double xs_sum, double xs_min, long n_plus_1, double dyn_sigma) {
#pragma clang fp contract(off)
  const double xs = xs_sum / (double)n_plus_1;
  const double ksi = xs_min / xs;
  double sigma = fmin(0.05 * (1 - ksi) / ksi, 2.0);
  sigma = fmax(0.1 * sigma * sigma * sigma, dyn_sigma);
  return sigma;
}

// one look-ahead of the Barzilai-Borwein search, adaptive.c:170-229: the spectral step from the five inner products of the difference vectors.
// Returns what the search does next: 0 = stop (beta is the mean of the last two), 1 = go on from (u, v) with v_prev rebuilt for the new penalty
// (adaptive.c:230-242), 2 = go on from (u, v) as they are (adaptive.c:243-247)
ABIP_HD int lp_bb_beta(double utut, double utv, double uu, double vv, double uv, double eps_cor, double eps_pen, double beta_prev, double &beta) {
#pragma clang fp contract(off)
  const double norm_ut = sqrt(utut), norm_u = sqrt(uu), norm_v = sqrt(vv);
  const double alpha_SD = vv / utv, alpha_MG = utv / utut, gamma_SD = vv / uv, gamma_MG = uv / uu;
  const double alpha_ss = (2 * alpha_MG > alpha_SD) ? alpha_MG : alpha_SD - 0.5 * alpha_MG;
  const double gamma_ss = (2 * gamma_MG > gamma_SD) ? gamma_MG : gamma_SD - 0.5 * gamma_MG;
  const double alpha_cor = utv / (norm_v * norm_ut), gamma_cor = uv / (norm_v * norm_u);
  if (alpha_cor > eps_cor && gamma_cor > eps_cor) beta = sqrt(alpha_ss * gamma_ss);
  else if (alpha_cor > eps_cor && gamma_cor <= eps_cor) beta = alpha_ss;
  else if (alpha_cor <= eps_cor && gamma_cor > eps_cor) beta = gamma_ss;
  else beta = beta_prev;
  const double diff = fabs(beta - beta_prev);
  if (diff > 0 && diff <= eps_pen) { beta = (beta + beta_prev) / 2; return 0; }
  if (diff > eps_pen) return 1;
  return 2;
}

} // namespace abip
