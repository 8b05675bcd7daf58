// qcp_formulations.h -- host front ends of the conic path: the scaling of the generic formulation (qcp_config.c:91-491) and the
// three specialised formulations the reference reaches through abip_ml (spe_problem vtable, src/abip-qcp/include/abip.h:27-60):
// LASSO (lasso_config.c), SVM as an SOCP (svm_config.c), SVM as a QP (svm_qp_config.c).  Each front end restates the formulation and
// its scaling rule and leaves in the work struct the scaled operator as ONE sparse matrix plus (b, c) -- what the device path runs on.
#pragma once
#include <algorithm>
#include <cmath>

#include "host_par.h"
#include "qcp_work.h"

namespace abip {
namespace qcp {

constexpr double kMinScale = 1e-3, kMaxScale = 1e3; // qcp_config.c:2-3

// ---- scaling, qcp_config.c:91-491 -----------------------------------------------------------------------------
inline void cone_average(std::vector<double> &E, const QCPCone *k) { // :194-212
  int count = 0;
  auto avg = [&](int len) { double y = 0; for (int j = 0; j < len; ++j) y += E[count + j]; y /= len; for (int j = 0; j < len; ++j) E[count + j] = y; count += len; };
  if (k->q) for (int i = 0; i < k->qsize; ++i) avg(k->q[i]);
  if (k->rq) for (int i = 0; i < k->rqsize; ++i) avg(k->rq[i]);
}
inline void apply_pass(QWk *w, std::vector<double> &Dp, std::vector<double> &Ep) { // :214-262
  const int m = w->m, n = w->n;
  const double min_row = kMinScale * std::sqrt((double)n), max_row = kMaxScale * std::sqrt((double)n);
  const double min_col = kMinScale * std::sqrt((double)m), max_col = kMaxScale * std::sqrt((double)m);
  HMat &A = w->A, &Q = w->Q;
  for (int i = 0; i < m; ++i) { if (Dp[i] < min_row) Dp[i] = 1; else if (Dp[i] > max_row) Dp[i] = max_row; }
  // (column ranges on a few host threads, host_par.h: every entry is touched by one thread, the arithmetic per entry is the sequential code's)
  host::par_by_entries(A.p.data(), n, 400000, [&](long lo, long hi, int) {
    for (long i = lo; i < hi; ++i) {
      if (Ep[i] < min_col) Ep[i] = 1; else if (Ep[i] > max_col) Ep[i] = max_col;
      for (int j = A.p[i]; j < A.p[i + 1]; ++j) A.x[j] /= Ep[i];
    }
  });
  if (w->hasQ) {
    for (int i = 0; i < n; ++i) for (int j = Q.p[i]; j < Q.p[i + 1]; ++j) Q.x[j] /= Ep[i];
    for (int q = 0; q < Q.p[n]; ++q) Q.x[q] /= Ep[Q.i[q]];
  }
  host::par_ranges((long)A.p[n], 400000, [&](long lo, long hi, int) { for (long q = lo; q < hi; ++q) A.x[q] /= Dp[A.i[q]]; });
  for (int i = 0; i < n; ++i) w->E[i] *= Ep[i];
  for (int i = 0; i < m; ++i) w->D[i] *= Dp[i];
}
// the ruiz / origin / pc passes over (A, Q) of a view of w->m rows and w->n columns; D_hat and E_hat accumulate in w->D, w->E
// (qcp_config.c:130-460; svm_qp_config.c:199-556 runs the same passes over its first dn + 1 columns)
inline void scale_passes(QWk *w, const QCPCone *k) {
  const int m = w->m, n = w->n;
  HMat &A = w->A, &Q = w->Q;
  w->D.assign(m, 1.0); w->E.assign(n, 1.0);
  std::vector<double> Ep(n), E1(n), E2(n), Dp(m);
  auto col_inf = [](const HMat &M, int j) { double mx = 0; for (int q = M.p[j]; q < M.p[j + 1]; ++q) { const double t = std::fabs(M.x[q]); if (t >= mx) mx = t; } return mx; };
  if (w->st->ruiz_scaling) {
    for (int it = 0; it < 10; ++it) {
      std::fill(E2.begin(), E2.end(), 0.0); std::fill(Dp.begin(), Dp.end(), 0.0);
      host::par_by_entries(A.p.data(), n, 400000, [&](long lo, long hi, int) { for (long j = lo; j < hi; ++j) E1[j] = (A.p[j] == A.p[j + 1]) ? 0 : std::sqrt(col_inf(A, (int)j)); });
      if (w->hasQ) for (int j = 0; j < n; ++j) E2[j] = (Q.p[j] == Q.p[j + 1]) ? 0 : std::sqrt(col_inf(Q, j));
      for (int j = 0; j < n; ++j) Ep[j] = E1[j] < E2[j] ? E2[j] : E1[j];
      cone_average(Ep, k);
      { // row maxima: a maximum does not depend on the order, so every thread takes a range of entries into a table of its own and the tables are folded
        const int T = (int)std::max<long>(1, std::min<long>(host::par_threads(), (long)A.p[n] / host::par_grain(400000)));
        if (T <= 1) { for (int q = 0; q < A.p[n]; ++q) if (Dp[A.i[q]] < std::fabs(A.x[q])) Dp[A.i[q]] = std::fabs(A.x[q]); }
        else {
          std::vector<std::vector<double>> part(T, std::vector<double>(m, 0.0));
          host::par_ranges_T((long)A.p[n], T, [&](long lo, long hi, int t) { std::vector<double> &P = part[t]; for (long q = lo; q < hi; ++q) { const double a = std::fabs(A.x[q]); if (P[A.i[q]] < a) P[A.i[q]] = a; } });
          for (int t = 0; t < T; ++t) for (int i = 0; i < m; ++i) if (Dp[i] < part[t][i]) Dp[i] = part[t][i];
        }
      }
      for (int i = 0; i < m; ++i) Dp[i] = std::sqrt(Dp[i]);
      apply_pass(w, Dp, Ep);
    }
  }
  if (w->st->origin_scaling) {
    std::fill(E1.begin(), E1.end(), 0.0); std::fill(E2.begin(), E2.end(), 0.0); std::fill(Dp.begin(), Dp.end(), 0.0);
    for (int j = 0; j < n; ++j) { for (int q = A.p[j]; q < A.p[j + 1]; ++q) E1[j] += A.x[q] * A.x[q]; E1[j] = std::sqrt(E1[j]); }
    if (w->hasQ) for (int j = 0; j < n; ++j) { for (int q = Q.p[j]; q < Q.p[j + 1]; ++q) E2[j] += Q.x[q] * Q.x[q]; E2[j] = std::sqrt(E2[j]); }
    for (int j = 0; j < n; ++j) Ep[j] = std::sqrt(E1[j] < E2[j] ? E2[j] : E1[j]);
    cone_average(Ep, k);
    for (int q = 0; q < A.p[n]; ++q) Dp[A.i[q]] += A.x[q] * A.x[q];
    for (int i = 0; i < m; ++i) Dp[i] = std::sqrt(std::sqrt(Dp[i]));
    apply_pass(w, Dp, Ep);
  }
  if (w->st->pc_scaling) {
    std::fill(E1.begin(), E1.end(), 0.0); std::fill(E2.begin(), E2.end(), 0.0); std::fill(Dp.begin(), Dp.end(), 0.0);
    for (int j = 0; j < n; ++j) { for (int q = A.p[j]; q < A.p[j + 1]; ++q) E1[j] += std::fabs(A.x[q]); E1[j] = std::sqrt(E1[j]); }
    if (w->hasQ) for (int j = 0; j < n; ++j) { for (int q = Q.p[j]; q < Q.p[j + 1]; ++q) E2[j] += std::fabs(Q.x[q]); E2[j] = std::sqrt(E2[j]); }
    for (int j = 0; j < n; ++j) Ep[j] = E1[j] < E2[j] ? E2[j] : E1[j];
    cone_average(Ep, k);
    for (int q = 0; q < A.p[n]; ++q) Dp[A.i[q]] += std::fabs(A.x[q]);
    for (int i = 0; i < m; ++i) Dp[i] = std::sqrt(Dp[i]);
    apply_pass(w, Dp, Ep);
  }
}
inline void scale_data(QWk *w, const QCPData *d, const QCPCone *k) {
  const int m = w->m, n = w->n;
  w->b.assign(d->b, d->b + m); w->c.assign(d->c, d->c + n);
  scale_passes(w, k);
  double ss = 0;
  for (double t : w->c) ss += t * t;
  double sb = 0;
  for (double t : w->b) sb += t * t;
  double sc = std::sqrt(std::sqrt(ss + sb)); // :462-463
  for (int i = 0; i < m; ++i) w->b[i] /= w->D[i];
  for (int j = 0; j < n; ++j) w->c[j] /= w->E[j];
  if (sc < kMinScale) sc = 1; else if (sc > kMaxScale) sc = kMaxScale;
  w->sc_b = 1 / sc; w->sc_c = 1 / sc;
  for (int i = 0; i < m; ++i) w->b[i] *= w->sc_b * w->st->scale;
  for (int j = 0; j < n; ++j) w->c[j] *= w->sc_c * w->st->scale;
}

// ---- LASSO front end: init_lasso + scaling_lasso_data, lasso_config.c:8-260 -----------------------------------------------
// min 1/2 |X beta - y|^2 + lambda |beta|_1 as the conic problem over (x0, x1, z (dm), beta+ (dn), beta- (dn)):
//   row 0: x0 = 1; rows 1..dm: z + X beta+ - X beta- = y; (x0, x1, z) in one rotated cone; beta+- >= 0; cost 2 x1 + lambda 1'(beta+ + beta-).
// The reference applies this operator matrix-free (lasso_A_times / lasso_AT_times, :99-128) and solves the KKT system through a reduced
// dm x dm or dn x dn system (:506-556, 652-708); here the scaled operator is materialised once as a sparse matrix (nnz = 1 + dm + 2 nnz(X))
// and handed to the conic path's own kernels and KKT back-ends -- the same linear maps, one code path on the device.
inline void build_lasso(QWk *w, const QCPData *d) {
  LassoForm &L = w->ls;
  const int dm = d->m, dn = d->n;
  L.dm = dm; L.dn = dn; L.lambda = d->lambda;
  const int p = dm + 1, q = 2 + 2 * dn + dm;
  const QCPMatrix *X = d->A;
  const int xnnz = X->p[dn];
  w->sparsity = (((double)xnnz / ((double)dm * (double)dn)) < 0.1); // :21
  if (w->sparsity) { // :36-51
    L.sc = 2; L.sc_c = 1 / L.lambda; L.sc_cone2 = L.lambda / dm * 80; L.sc_cone1 = 0.8 / L.sc_c / L.sc_cone2; L.sc_b = L.sc_c * 300 * L.lambda / dm;
  } else {
    L.sc = dm < dn ? 4 : 1; L.sc_c = 1 / L.lambda; L.sc_b = L.sc_c; L.sc_cone2 = 0.8; L.sc_cone1 = 1 / L.sc_c;
  }
  std::vector<double> xs(X->x, X->x + xnnz);
  std::vector<double> &E = L.E, &D = L.D;
  E.assign(dn, 0.0); D.assign(dm, 0.0);
  const double sqm = std::sqrt((double)dm);
  if (w->st->scale_E) { // :156-210
    if (w->sparsity) {
      double avg = 0, avg1 = 0;
      for (int i = 0; i < dn; ++i) { for (int j = X->p[i]; j < X->p[i + 1]; ++j) E[i] += xs[j] * xs[j]; avg += std::sqrt(E[i]); }
      avg /= dn;
      for (int i = 0; i < dn; ++i) {
        E[i] = avg / std::sqrt(E[i] + 1e-4) / L.sc;
        if (E[i] > 1000 * sqm) E[i] = 1000 * sqm;
        if (E[i] < 0.001 * sqm) E[i] = 1;
        if (E[i] > 50) E[i] = 50;
        avg1 += E[i];
      }
      avg1 /= dn;
      for (int i = 0; i < dn; ++i) E[i] = avg1 / E[i] / L.sc;
    } else {
      for (int i = 0; i < dn; ++i) {
        for (int j = X->p[i]; j < X->p[i + 1]; ++j) E[i] += xs[j] * xs[j];
        E[i] = std::sqrt(E[i]);
        if (E[i] > 1000 * sqm) E[i] = 1000 * sqm;
        if (E[i] < 0.001 * sqm) E[i] = 1;
        if (E[i] > 7) E[i] = 7;
        E[i] = 1 / (E[i] * L.sc);
      }
    }
    host::par_by_entries(X->p, (long)dn, 400000, [&](long lo, long hi, int) { for (long i = lo; i < hi; ++i) for (int j = X->p[i]; j < X->p[i + 1]; ++j) xs[j] *= E[i]; });
  }
  for (int k = 0; k < xnnz; ++k) D[X->i[k]] += xs[k] * xs[k]; // :212-230
  double avg = 0;
  for (int i = 0; i < dm; ++i) avg += std::sqrt(2 * D[i] + L.sc_cone2);
  avg /= dm;
  for (int i = 0; i < dm; ++i) D[i] = avg / std::sqrt(2 * D[i] + L.sc_cone2);
  host::par_ranges((long)xnnz, 400000, [&](long lo, long hi, int) { for (long k = lo; k < hi; ++k) xs[k] *= D[X->i[k]]; });
  L.y.assign(d->b, d->b + dm);
  w->b.assign(p, 0.0); w->c.assign(q, 0.0); // :232-250
  w->b[0] = L.sc_cone1;
  for (int i = 0; i < dm; ++i) w->b[1 + i] = d->b[i] * D[i];
  for (double &t : w->b) t *= L.sc_b;
  w->c[1] = L.sc_cone1 * L.sc_cone2;
  for (int i = 0; i < dn; ++i) { w->c[dm + 2 + i] = E[i] * L.lambda; w->c[dm + 2 + dn + i] = E[i] * L.lambda; }
  for (double &t : w->c) t *= L.sc_c;
  // the operator of lasso_A_times (:99-110) as a p x q CSC matrix
  HMat &A = w->A;
  A.m = p; A.n = q; A.p.assign(q + 1, 0);
  A.i.assign((size_t)1 + dm + 2 * (size_t)xnnz, 0); A.x.assign(A.i.size(), 0.0);
  const double sq2 = std::sqrt(L.sc_cone2);
  A.i[0] = 0; A.x[0] = 1.0; A.p[1] = 1; // column 0
  A.p[2] = 1;                           // column 1 is empty
  for (int i = 0; i < dm; ++i) { A.i[1 + i] = 1 + i; A.x[1 + i] = D[i] * sq2; A.p[3 + i] = 2 + i; }
  for (int sign = 0; sign < 2; ++sign) { // [X~ | -X~]: the column pointers first, then the two copies filled by column ranges (host_par.h)
    const int base = 1 + dm + sign * xnnz;
    for (int j = 0; j < dn; ++j) A.p[dm + 2 + sign * dn + j + 1] = base + (int)X->p[j + 1];
    host::par_by_entries(X->p, (long)dn, 400000, [&](long lo, long hi, int) {
      for (long j = lo; j < hi; ++j) for (int k = X->p[j]; k < X->p[j + 1]; ++k) { A.i[base + k] = 1 + (int)X->i[k]; A.x[base + k] = sign ? -xs[k] : xs[k]; }
    });
  }
  w->D.assign(p, 1.0); w->E.assign(q, 1.0); w->sc_b = 1; w->sc_c = 1; // (neutral for the generic sums kq_resid still provides: certificates)
}

// ---- SVM-SOCP front end: init_svm + scaling_svm_data, svm_config.c:8-171, 281-391 -------------------------------------------------
// x = (x0, x1, r (dn), w+ (dn), b+, w- (dn), b-, xi (dm), t (dm)); (x0, x1, r) in one rotated cone, the rest >= 0; dm + dn + 1 rows:
//   row 0: x0 = const;  rows 1..dm: diag(y)(X (w+ - w-) + (b+ - b-)) + xi - t = 1;  rows dm+1..: r tied to w+ - w-;  cost x1 + C 1'xi.
// As for LASSO the operator of svm_A_times (:177-199) is materialised (nnz = 1 + 3 dn + 2 dm + 2 (nnz(X) + dm)) and the conic path's own
// KKT back-ends replace the block elimination of :725-806 (same linear system, rho_x = 1 as hard-coded there).
// The scale constants are a table of heuristics in (dm, dn, lambda); the reference leaves them uninitialised when dm == 10 dn or
// 10 dm == dn (no branch taken, :63-107) and never assigns sc_cone2 when dm > 10 dn with dn < 10 (:85-89): the boundaries are closed
// towards the outer branches here and sc_cone2 starts from sc_cone1 in that corner.
inline void svm_constants(int m, int n, double lambda, SvmForm &V) {
  const double l2 = std::log(2 * lambda) / std::log(10.0), l5 = std::log(5 * lambda) / std::log(10.0);
  V.sc = 1; V.sc_b = 1;
  if (((long long)m < 10LL * n) && (10LL * m > (long long)n)) {
    V.sc_c = std::max(0.45, std::pow(7.5, -l2) * 2); V.sc_cone1 = std::max(3.0, l2 * 4 + 4); V.sc_cone2 = V.sc_cone1;
  } else if (10LL * m <= (long long)n) {
    V.sc_cone2 = std::max(3.0, l2 * 2 + 2);
    if (lambda >= 1) { V.sc_c = std::max(0.2, std::pow(0.2, l2) * 7.5); V.sc_cone1 = V.sc_cone2; }
    else { V.sc_c = std::pow(0.3, l2) * 3; V.sc_cone1 = std::max(0.4, l2 * 0.2 + 0.8); }
  } else {
    if (n < 10) {
      V.sc_c = 1 / lambda; V.sc_cone1 = 6; V.sc_cone2 = 6;
      if (lambda < 0.002) V.sc_cone2 = V.sc_cone2 - 3 * std::log(lambda * 500) / std::log(10.0);
    } else if (lambda >= 1) { V.sc_c = 1 / lambda; V.sc_cone1 = 6; V.sc_cone2 = lambda; }
    else {
      V.sc_c = std::min(std::pow(5, -l5) * 4, 300.0); V.sc_b = std::max(0.1, l5 * 0.2 + 0.9); V.sc_cone1 = std::max(0.05, l5 * 0.3 + 0.7); V.sc_cone2 = -l5 * 2 + 6;
      if (lambda < 0.002) V.sc_cone2 = V.sc_cone2 - 3 * std::log(lambda * 500) / std::log(10.0);
    }
  }
}
inline bool build_svm(QWk *w, const QCPData *d) { // false: a feature column of X is identically zero
  for (int j = 0; j < d->n; ++j) { // the scaling divides by every column's 2-norm (svm_config.c:300-308): the reference goes on with inf / NaN and never converges
    bool nz = false;
    for (int t = d->A->p[j]; t < d->A->p[j + 1] && !nz; ++t) nz = d->A->x[t] * d->b[d->A->i[t]] != 0.0;
    if (!nz) return false;
  }
  SvmForm &V = w->sv;
  const int dm = d->m, dn = d->n, n1 = dn + 1, p = dm + dn + 1, q = 4 + 3 * dn + 2 * dm;
  V.dm = dm; V.dn = dn; V.lambda = d->lambda;
  const QCPMatrix *X = d->A;
  const int xnnz = X->p[dn];
  w->sparsity = (((double)xnnz / ((double)dm * (double)dn)) < 0.05); // :20
  svm_constants(dm, dn, V.lambda, V);
  // data_A = [diag(y) X, y] (:109-133), then column and row equilibration (:300-341)
  std::vector<double> xs((size_t)xnnz + dm);
  for (int t = 0; t < xnnz; ++t) xs[t] = X->x[t] * d->b[X->i[t]];
  for (int i = 0; i < dm; ++i) xs[xnnz + i] = d->b[i];
  auto col_lo = [&](int j) { return j < dn ? X->p[j] : xnnz; };
  auto col_hi = [&](int j) { return j < dn ? X->p[j + 1] : xnnz + dm; };
  auto row_of = [&](int t) { return t < xnnz ? X->i[t] : t - xnnz; };
  std::vector<double> &E = V.E, &D = V.D;
  E.assign(n1, 0.0); D.assign(dm, 0.0);
  double avg = 0;
  if (w->st->scale_E) {
    for (int j = 0; j < n1; ++j) { for (int t = col_lo(j); t < col_hi(j); ++t) E[j] += xs[t] * xs[t]; E[j] = std::sqrt(E[j]); avg += E[j]; }
    avg /= n1;
    for (int j = 0; j < n1; ++j) E[j] = avg / E[j];
    for (int j = 0; j < n1; ++j) for (int t = col_lo(j); t < col_hi(j); ++t) xs[t] *= E[j];
  }
  for (int t = 0; t < xnnz + dm; ++t) D[row_of(t)] += xs[t] * xs[t];
  avg = 0;
  for (int i = 0; i < dm; ++i) avg += std::sqrt(D[i]);
  avg /= dm;
  for (int i = 0; i < dm; ++i) D[i] = avg / std::sqrt(D[i]);
  for (int t = 0; t < xnnz + dm; ++t) xs[t] *= D[row_of(t)];
  std::vector<double> wD(dn), &wE = V.wE;
  wE.assign(dn, 0.0);
  for (int j = 0; j < dn; ++j) { const double F = 1 / std::sqrt(1 + 2 * E[j] * E[j]); wD[j] = F * -std::sqrt(V.sc_cone1); wE[j] = E[j] * F; } // :343-345, 374-380
  w->b.assign(p, 0.0); w->c.assign(q, 0.0); // :347-364
  w->b[0] = V.sc_cone2;
  for (int i = 0; i < dm; ++i) w->b[1 + i] = D[i];
  for (double &t : w->b) t *= V.sc_b;
  w->c[1] = V.sc_c * V.sc_cone1 * V.sc_cone2;
  for (int i = 0; i < dm; ++i) w->c[3 * dn + 4 + i] = V.lambda * V.sc_c / V.sc;
  // the operator of svm_A_times (:177-199) as a p x q CSC matrix
  HMat &A = w->A;
  A.m = p; A.n = q; A.p.assign(q + 1, 0); A.i.clear(); A.x.clear();
  A.i.reserve((size_t)1 + 3 * dn + 2 * dm + 2 * ((size_t)xnnz + dm)); A.x.reserve(A.i.capacity());
  int col = 0;
  auto close = [&]() { A.p[++col] = (int)A.i.size(); };
  A.i.push_back(0); A.x.push_back(1.0); close(); // x0
  close();                                        // x1: empty
  for (int j = 0; j < dn; ++j) { A.i.push_back(1 + dm + j); A.x.push_back(wD[j]); close(); } // r
  for (int sign = 0; sign < 2; ++sign) {
    const double sg = sign ? -1.0 : 1.0;
    for (int j = 0; j < dn; ++j) { // w+ / w-
      for (int t = X->p[j]; t < X->p[j + 1]; ++t) { A.i.push_back(1 + X->i[t]); A.x.push_back(sg * xs[t]); }
      A.i.push_back(1 + dm + j); A.x.push_back(-sg * wE[j]);
      close();
    }
    for (int i = 0; i < dm; ++i) { A.i.push_back(1 + i); A.x.push_back(sg * xs[xnnz + i]); } // b+ / b-
    close();
  }
  for (int i = 0; i < dm; ++i) { A.i.push_back(1 + i); A.x.push_back(D[i] * (1 / V.sc)); close(); } // xi
  for (int i = 0; i < dm; ++i) { A.i.push_back(1 + i); A.x.push_back(-D[i]); close(); }             // t
  w->D.assign(p, 1.0); w->E.assign(q, 1.0); w->sc_b = 1; w->sc_c = 1; // (neutral for the generic sums kq_resid still provides: certificates)
  return true;
}

// ---- SVM-QP front end: init_svmqp + scaling_svmqp_data, svm_qp_config.c:8-124, 195-590 -------------------------------------
// x = (w (dn), b, xi (dm), t (dm)); w, b free, xi, t >= 0;  min 1/2 |w|^2 + 1/(dm lambda) 1'xi  s.t.  diag(y) (X w + b) + xi - t = 1.
// The reference keeps the data block B~ = D^-1 diag(y) [X, 1] E^-1 and applies the +-D^-1 identity columns on the fly (:129-147); here the
// whole dm x q operator [B~, D^-1, -D^-1] is materialised and the generic conic path (kernels, KKT back-ends, residuals) runs on it.
// The caller's X is left untouched (the reference folds the labels into it in place, :84-86).
inline void build_svmqp(QWk *w, const QCPData *d, const QCPCone *k) {
  SvmForm &V = w->sv;
  const int dm = d->m, dn = d->n, q = 1 + dn + 2 * dm, n1 = dn + 1;
  V.dm = dm; V.dn = dn; V.lambda = d->lambda;
  const QCPMatrix *X = d->A;
  const int xnnz = X->p[dn];
  w->sparsity = (((double)xnnz / ((double)dm * (double)dn)) < 0.05); // :37
  HMat &B = w->A, &Q = w->Q;
  B.m = dm; B.n = n1; B.p.assign(X->p, X->p + dn + 1); B.p.push_back(xnnz + dm);
  B.i.assign(X->i, X->i + xnnz); B.x.resize((size_t)xnnz + dm);
  for (int t = 0; t < xnnz; ++t) B.x[t] = X->x[t] * d->b[X->i[t]];
  for (int i = 0; i < dm; ++i) { B.i.push_back(i); B.x[xnnz + i] = d->b[i]; }
  Q.m = q; Q.n = q; Q.p.assign(q + 1, dn); Q.i.resize(dn); Q.x.assign(dn, 1.0); // :19-35
  for (int i = 0; i < dn; ++i) { Q.i[i] = i; Q.p[i] = i; }
  w->b.assign(dm, 1.0); w->c.assign(q, 0.0);
  for (int i = 0; i < dm; ++i) w->c[dn + 1 + i] = 1.0 / (dm * V.lambda);
  w->nm_inf_b = vnrminf(w->b.data(), dm); w->nm_inf_c = vnrminf(w->c.data(), q); // abip.c:875-876
  w->n = n1; scale_passes(w, k); w->n = q; // (the passes see the data block only)
  w->E.resize(q, 1.0);                      // :107-110
  double ss = 0, sb = 0;
  for (double t : w->c) ss += t * t;
  for (double t : w->b) sb += t * t;
  double sc = std::sqrt(std::sqrt(ss + sb)); // :549-550
  for (int i = 0; i < dm; ++i) w->b[i] /= w->D[i];
  for (int j = 0; j < n1; ++j) w->c[j] /= w->E[j];
  if (sc < kMinScale) sc = 1; else if (sc > kMaxScale) sc = kMaxScale;
  w->sc_b = 1 / sc; w->sc_c = 1 / sc;
  for (double &t : w->b) t *= w->sc_b * w->st->scale;
  for (double &t : w->c) t *= w->sc_c * w->st->scale;
  // [B~, D^-1, -D^-1]
  B.n = q; B.p.resize(q + 1);
  B.i.reserve(B.i.size() + 2 * (size_t)dm); B.x.reserve(B.x.size() + 2 * (size_t)dm);
  for (int sign = 0; sign < 2; ++sign)
    for (int i = 0; i < dm; ++i) { B.i.push_back(i); B.x.push_back((sign ? -1.0 : 1.0) * (1 / w->D[i])); B.p[n1 + sign * dm + i + 1] = (int)B.i.size(); }
}

// ---- several GPUs: column blocks of the sharded conic path (qcp_dist.h) --------------------------------------------------------
// Cut points are allowed behind every cone and anywhere inside the free / zero / orthant blocks; block g ends at the first allowed cut at or
// beyond the g-th share of the weight (non-zeros + 1 per column), always leaving a cut for every rank still to come.
// colp: the n + 1 column pointers of A.  Optionally returns the cone extents and the start of the free block.  0 ok; -1 a rotated cone of
// fewer than 3 entries (the reference's layout walk skips those without advancing, abip.c:379-381: not served sharded); -2 fewer blocks than ranks.
inline int column_bounds(int n, const int *colp, const QCPCone *K, int world, std::vector<int> &bounds, std::vector<int> *qs, std::vector<int> *qe,
                         std::vector<int> *rs, std::vector<int> *re, int *f0_out) {
  std::vector<int> cuts;
  int pos = 0;
  for (int i = 0; K->q && i < K->qsize; ++i) { if (K->q[i] <= 0) continue; if (qs) qs->push_back(pos); pos += K->q[i]; if (qe) qe->push_back(pos); cuts.push_back(pos); }
  for (int i = 0; K->rq && i < K->rqsize; ++i) { if (K->rq[i] < 3) return -1; if (rs) rs->push_back(pos); pos += K->rq[i]; if (re) re->push_back(pos); cuts.push_back(pos); }
  if (f0_out) *f0_out = pos;
  for (int t = pos + 1; t <= n; ++t) cuts.push_back(t);
  std::vector<double> Wt(n + 1, 0.0);
  for (int j = 0; j < n; ++j) Wt[j + 1] = Wt[j] + (double)(colp[j + 1] - colp[j]) + 1.0;
  bounds.assign(world + 1, 0);
  bounds[world] = n;
  size_t ci = 0;
  for (int g = 1; g < world; ++g) {
    const double target = Wt[n] * g / world;
    while (ci < cuts.size() && (cuts[ci] <= bounds[g - 1] || (Wt[cuts[ci]] < target && cuts.size() - ci > (size_t)(world - g)))) ++ci;
    if (ci >= cuts.size() || cuts[ci] >= n) return -2;
    bounds[g] = cuts[ci++];
  }
  return bounds[world - 1] < n ? 0 : -2;
}

} // namespace qcp
} // namespace abip
