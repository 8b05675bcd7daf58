// host_setup.cpp -- host-side, once-per-init work: data validation, the reference's scaling of A,
// the 32-bit CSR/CSC images the kernels stream, their row blocks, the Jacobi preconditioner, and the
// sparse LDL' factorisation + level schedule for the direct back-end.
#include "host_setup.h"
#include "host_par.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>
#include <numeric>

namespace abip {
namespace host {

namespace {
constexpr double kMinScale = 1e-3; // linsys/common.c:4
constexpr double kMaxScale = 1e3;  // linsys/common.c:5

inline double clamp_scale(double e, double lo, double hi) { // common.c:224-229 and its repeats
  if (e < lo) return 1.0;
  if (e > hi) return hi;
  return e;
}
inline double col_norm1(const double *v, abip_int len) { double s = 0; for (abip_int i = 0; i < len; ++i) s += std::fabs(v[i]); return s; }
inline double col_norm2(const double *v, abip_int len) { double s = 0; for (abip_int i = 0; i < len; ++i) s += v[i] * v[i]; return std::sqrt(s); }
inline double col_norminf(const double *v, abip_int len) { double mx = 0; for (abip_int i = 0; i < len; ++i) { double t = std::fabs(v[i]); if (t >= mx) mx = t; } return mx; }
inline void col_scale(double *v, double sc, abip_int len) { for (abip_int i = 0; i < len; ++i) v[i] *= sc; }
} // namespace

int validate(const ABIPData *d) {
  const ABIPSettings *s = d->stgs;
  if (d->m <= 0 || d->n <= 0) { printf("m and n must both be greater than 0; m = %li, n = %li\n", (long)d->m, (long)d->n); return -1; }
  if (d->m > d->n) { printf("WARN: m larger than n, problem likely degenerate\n"); return -1; }
  const ABIPMatrix *A = d->A;
  if (!A || !A->x || !A->i || !A->p) { printf("ERROR: incomplete data!\n"); printf("invalid linear system input data\n"); return -1; }
  for (abip_int i = 0; i < A->n; ++i) {
    if (A->p[i] == A->p[i + 1]) printf("WARN: the %li-th column empty!\n", (long)i);
    else if (A->p[i] > A->p[i + 1]) { printf("ERROR: the column pointers decreases!\n"); printf("invalid linear system input data\n"); return -1; }
  }
  const abip_int nnz = A->p[A->n];
  if (((double)nnz / A->m > A->n) || nnz <= 0) {
    printf("ERROR: the number of nonzeros in A = %li, outside of valid range!\n", (long)nnz);
    printf("invalid linear system input data\n");
    return -1;
  }
  abip_int rmax = 0;
  for (abip_int i = 0; i < nnz; ++i) if (A->i[i] > rmax) rmax = A->i[i];
  if (rmax > A->m - 1) { printf("ERROR: the number of rows in A is inconsistent with input dimension!\n"); printf("invalid linear system input data\n"); return -1; }
  if (nnz >= 2147483647L || A->n >= 2147483647L - 4096) { printf("ERROR: problem too large for the 32-bit device index space\n"); return -1; }
  if (s->max_ipm_iters <= 0) { printf("max_ipm_iters must be positive\n"); return -1; }
  if (s->max_admm_iters <= 0) { printf("max_admm_iters must be positive\n"); return -1; }
  if (s->eps <= 0) { printf("eps tolerance must be positive\n"); return -1; }
  if (s->alpha <= 0 || s->alpha >= 2) { printf("alpha must be in (0,2)\n"); return -1; }
  if (s->rho_y <= 0) { printf("rho_y must be positive (1e-3 works well).\n"); return -1; }
  if (s->scale <= 0) { printf("scale must be positive (1 works well).\n"); return -1; }
  if (s->eps_cor <= 0) { printf("eps_cor tolerance must be positive.\n"); return -1; }
  if (s->eps_pen <= 0) { printf("eps_pen tolerance must be positive.\n"); return -1; }
  if (s->adaptive_lookback <= 0) { printf("adaptive_lookback must be positive.\n"); return -1; }
  if (s->hybrid_mu > 0 && s->dynamic_sigma >= 0) { printf("when use hybrid mu strategy, dynamic_sigma must be negative.\n"); return -1; }
  return 0;
}

void normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, std::vector<double> &D, std::vector<double> &E,
                 double *mean_norm_row, double *mean_norm_col) {
  const abip_int m = A->m, n = A->n;
  std::vector<double> Dk(m, 0.0), Ek(n, 0.0), Dt(m, 0.0);
  std::vector<double> D_pc(m, 1.0), E_pc(n, 1.0), D_or(m, 1.0), E_or(n, 1.0), D_rz(m, 1.0), E_rz(n, 1.0), D_qp(m, 1.0), E_qp(n, 1.0);
  const double min_row = kMinScale * std::sqrt((double)n), max_row = kMaxScale * std::sqrt((double)n); // common.c:172-175
  const double min_col = kMinScale * std::sqrt((double)m), max_col = kMaxScale * std::sqrt((double)m);
  // A few host threads (host_par.h): a thread owns a range of columns (or of entries for the element-wise divisions); the sums down a ROW keep their sequential
  // pass (their order is the reference's), the row maxima of the Ruiz passes are folded from per-thread tables (a maximum has no order).
  auto rows_div = [&](const std::vector<double> &R) { par_ranges((long)A->p[n], 400000, [&](long lo, long hi, int) { for (long q = lo; q < hi; ++q) A->x[q] /= R[A->i[q]]; }); };
  auto cols = [&](auto body) { par_by_entries(A->p, (long)n, 400000, [&](long lo, long hi, int) { for (long j = lo; j < hi; ++j) body((abip_int)j); }); };

  if (stgs->pc_ruiz_rescale) { // common.c:217-266
    cols([&](abip_int j) {
      const abip_int len = A->p[j + 1] - A->p[j];
      const double e = clamp_scale(std::sqrt(col_norm1(&A->x[A->p[j]], len)), min_col, max_col);
      col_scale(&A->x[A->p[j]], 1.0 / e, len);
      E_pc[j] = e;
    });
    std::fill(Dk.begin(), Dk.end(), 0.0);
    for (abip_int q = 0; q < A->p[n]; ++q) Dk[A->i[q]] += std::fabs(A->x[q]);
    for (abip_int i = 0; i < m; ++i) D_pc[i] = clamp_scale(std::sqrt(Dk[i]), min_row, max_row);
    rows_div(D_pc);
    printf("Done the pc rescaling!\n");
  }
  if (stgs->origin_rescale) { // common.c:279-327
    for (abip_int j = 0; j < n; ++j) {
      const abip_int len = A->p[j + 1] - A->p[j];
      const double e = clamp_scale(col_norm2(&A->x[A->p[j]], len), min_col, max_col);
      col_scale(&A->x[A->p[j]], 1.0 / e, len);
      E_or[j] = e;
    }
    std::fill(Dk.begin(), Dk.end(), 0.0);
    for (abip_int q = 0; q < A->p[n]; ++q) Dk[A->i[q]] += A->x[q] * A->x[q];
    for (abip_int i = 0; i < m; ++i) D_or[i] = clamp_scale(std::sqrt(Dk[i]), min_row, max_row);
    rows_div(D_or);
    printf("Done the origin rescaling!\n");
  }
  if (stgs->pc_ruiz_rescale) { // common.c:339-413
    for (abip_int it = 0; it < stgs->ruiz_iter; ++it) {
      cols([&](abip_int j) {
        const abip_int len = A->p[j + 1] - A->p[j];
        const double e = clamp_scale(std::sqrt(col_norminf(&A->x[A->p[j]], len)), min_col, max_col);
        col_scale(&A->x[A->p[j]], 1.0 / e, len);
        Ek[j] = e;
      });
      std::fill(Dk.begin(), Dk.end(), 0.0);
      {
        const long nz = (long)A->p[n];
        const int T = (int)std::max<long>(1, std::min<long>(par_threads(), nz / par_grain(400000)));
        if (T <= 1) { for (abip_int q = 0; q < A->p[n]; ++q) { const double w = std::fabs(A->x[q]); if (w >= Dk[A->i[q]]) Dk[A->i[q]] = w; } }
        else {
          std::vector<std::vector<double>> part(T, std::vector<double>(m, 0.0));
          par_ranges_T(nz, T, [&](long lo, long hi, int t) { std::vector<double> &P = part[t]; for (long q = lo; q < hi; ++q) { const double w = std::fabs(A->x[q]); if (w >= P[A->i[q]]) P[A->i[q]] = w; } });
          for (int t = 0; t < T; ++t) for (abip_int i = 0; i < m; ++i) if (part[t][i] >= Dk[i]) Dk[i] = part[t][i];
        }
      }
      for (abip_int i = 0; i < m; ++i) Dk[i] = clamp_scale(std::sqrt(Dk[i]), min_row, max_row);
      rows_div(Dk);
      for (abip_int j = 0; j < n; ++j) E_rz[j] = E_rz[j] * Ek[j];
      for (abip_int i = 0; i < m; ++i) D_rz[i] = D_rz[i] * Dk[i];
    }
    printf("Done the ruiz rescaling!\n");
  }
  if (stgs->qp_rescale) { // common.c:415-499
    std::fill(D_qp.begin(), D_qp.end(), 0.0);
    for (abip_int j = 0; j < n; ++j) {
      const abip_int len = A->p[j + 1] - A->p[j];
      double *col = &A->x[A->p[j]];
      double e = col_norminf(col, len), ref = e;
      for (abip_int i = 0; i < len; ++i) { const double t = std::fabs(col[i]); if (t <= ref && t > 0) ref = t; } // linalg.c:126-144
      e = clamp_scale(std::sqrt(ref) * std::sqrt(e), min_col, max_col);
      col_scale(col, 1.0 / e, len);
      E_qp[j] = e;
    }
    for (abip_int q = 0; q < A->p[n]; ++q) { const double w = std::fabs(A->x[q]); if (w >= D_qp[A->i[q]]) D_qp[A->i[q]] = w; }
    Dt = D_qp;
    for (abip_int q = 0; q < A->p[n]; ++q) { const double w = std::fabs(A->x[q]); if (w <= Dt[A->i[q]] && w > 0) Dt[A->i[q]] = w; }
    for (abip_int i = 0; i < m; ++i) D_qp[i] = clamp_scale(std::sqrt(D_qp[i] * Dt[i]), min_row, max_row);
    rows_div(D_qp);
    printf("Done the QP rescaling\n");
  }
  D.resize(m); E.resize(n);
  for (abip_int i = 0; i < m; ++i) D[i] = D_pc[i] * D_rz[i] * D_or[i] * D_qp[i]; // common.c:512-520
  for (abip_int j = 0; j < n; ++j) E[j] = E_pc[j] * E_rz[j] * E_or[j] * E_qp[j];

  std::fill(Dk.begin(), Dk.end(), 0.0); // common.c:523-545
  for (abip_int q = 0; q < A->p[n]; ++q) Dk[A->i[q]] += A->x[q] * A->x[q];
  *mean_norm_row = 0.0;
  for (abip_int i = 0; i < m; ++i) *mean_norm_row += std::sqrt(Dk[i]) / m;
  *mean_norm_col = 0.0;
  for (abip_int j = 0; j < n; ++j) *mean_norm_col += col_norm2(&A->x[A->p[j]], A->p[j + 1] - A->p[j]) / n;
  if (stgs->scale != 1) col_scale(A->x, stgs->scale, A->p[n]); // common.c:547-550
}

void un_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, const std::vector<double> &D, const std::vector<double> &E) {
  const abip_int n = A->n;
  for (abip_int q = 0; q < A->p[n]; ++q) A->x[q] *= D[A->i[q]];
  for (abip_int j = 0; j < n; ++j) col_scale(&A->x[A->p[j]], E[j] / stgs->scale, A->p[j + 1] - A->p[j]);
}

void csc_as_csr(const ABIPMatrix *A, HostCsr &out) {
  const abip_int n = A->n, nnz = A->p[n];
  out.nrows = (int)n; out.ncols = (int)A->m;
  out.ptr.resize(n + 1); out.idx.resize(nnz); out.val.assign(A->x, A->x + nnz);
  for (abip_int j = 0; j <= n; ++j) out.ptr[j] = (int)A->p[j];
  for (abip_int q = 0; q < nnz; ++q) out.idx[q] = (int)A->i[q];
}

void transpose_to_csr(const ABIPMatrix *A, HostCsr &out) {
  const abip_int m = A->m, n = A->n, nnz = A->p[n];
  out.nrows = (int)m; out.ncols = (int)n;
  (void)nnz;
  par_transpose((long)m, (long)n, A->p, A->i, A->x, out.ptr, out.idx, out.val); // (host_par.h)
}

void build_row_blocks(HostCsr &M, int chunk) {
  M.rb.clear();
  M.rb.push_back(0);
  int r = 0;
  while (r < M.nrows) {
    int nn = 0, rows = 0, e = r;
    while (e < M.nrows) {
      const int len = M.ptr[e + 1] - M.ptr[e];
      if (rows > 0 && (nn + len > chunk || rows >= chunk)) break;
      nn += len; ++rows; ++e;
      if (nn > chunk) break; // a single long row forms its own block
    }
    M.rb.push_back(e);
    r = e;
  }
}

bool build_sell(HostCsr &M, double max_pad, long min_nnz) {
  M.sval.clear(); M.sidx.clear(); M.slen.clear(); M.soff.clear();
  const long nnz = M.ptr[M.nrows];
  const char *e = getenv("ABIP_HIP_SELL");
  if (e && atoi(e) == 0) return false;
  if (nnz < min_nnz && !(e && atoi(e) == 2)) return false; // (ABIP_HIP_SELL=2: tests force it on small matrices)
  const int ns = (M.nrows + 63) / 64;
  long stored = 0;
  std::vector<int> slen(ns, 0);
  for (int s = 0; s < ns; ++s) {
    int len = 0;
    for (int r = s * 64; r < std::min(M.nrows, s * 64 + 64); ++r) len = std::max(len, M.ptr[r + 1] - M.ptr[r]);
    slen[s] = len; stored += (long)len * 64;
  }
  if ((double)stored > (1.0 + max_pad) * (double)nnz + 64.0 * 8) return false;
  M.slen = slen; M.soff.resize(ns); M.sval.assign(stored, 0.0); M.sidx.assign(stored, 0);
  long off = 0;
  for (int s = 0; s < ns; ++s) {
    M.soff[s] = off;
    for (int l = 0; l < 64; ++l) {
      const int r = s * 64 + l;
      const int a = r < M.nrows ? M.ptr[r] : 0, b = r < M.nrows ? M.ptr[r + 1] : 0;
      const int fill = b > a ? M.idx[b - 1] : 0; // padding gathers a column the row gathers anyway (value 0)
      for (int k = 0; k < slen[s]; ++k) {
        const long q = off + (long)k * 64 + l;
        if (a + k < b) { M.sval[q] = M.val[a + k]; M.sidx[q] = M.idx[a + k]; } else M.sidx[q] = fill;
      }
    }
    off += (long)slen[s] * 64;
  }
  return true;
}

void jacobi_preconditioner(const ABIPMatrix *A, std::vector<double> &Minv) {
  Minv.assign(A->m, 0.0);
  for (abip_int q = 0; q < A->p[A->n]; ++q) Minv[A->i[q]] += A->x[q] * A->x[q];
  for (abip_int i = 0; i < A->m; ++i) Minv[i] = 1 / Minv[i];
}

// ------------------------------------------------------------------------------------------------
// direct back-end: ordering + LDL'
// ------------------------------------------------------------------------------------------------
namespace {

// Minimum-degree ordering on the quotient graph with LAZY degree updates.  Eliminating p changes the external degree of its
// neighbours i by at least -1 and to at least |Lp| - 1, so instead of recomputing every neighbour after every pivot (the cost of a
// classical minimum-degree code, and still the cost of AMD's approximate update when rows sit in hundreds of elements, as the rows of
// a LASSO / least-squares KKT matrix do) a neighbour only gets that LOWER BOUND and a "stale" mark.  A stale node is recomputed --
// exactly, by marking the members of its elements, or by the sum-of-elements bound when that would be too long -- only when it
// reaches the front of the degree lists; a node that is fresh there is a true minimum.  Element absorption as usual; no
// supervariables.  The reference calls SuiteSparse AMD here (direct.c:106-119); any symmetric permutation is admissible because K
// is quasi-definite.
void min_degree(int N, const std::vector<int> &Gp, const std::vector<int> &Gi, std::vector<int> &perm) {
  std::vector<std::vector<int>> adjv(N), adje(N), elem(N);
  std::vector<int> deg(N), mark(N, -1), head(N + 1, -1), nxt(N, -1), prv(N, -1);
  std::vector<char> elim(N, 0), dead(N, 0), stale(N, 0);
  for (int i = 0; i < N; ++i) { adjv[i].assign(Gi.begin() + Gp[i], Gi.begin() + Gp[i + 1]); deg[i] = (int)adjv[i].size(); }
  auto ins = [&](int x) { const int d = deg[x]; nxt[x] = head[d]; prv[x] = -1; if (head[d] >= 0) prv[head[d]] = x; head[d] = x; };
  auto del = [&](int x) { const int d = deg[x]; if (prv[x] >= 0) nxt[prv[x]] = nxt[x]; else head[d] = nxt[x]; if (nxt[x] >= 0) prv[nxt[x]] = prv[x]; };
  for (int i = N - 1; i >= 0; --i) ins(i);
  perm.resize(N);
  int stamp = 0, mindeg = 0;
  std::vector<int> Lp;
  // external degree of x among the nodes not yet eliminated (k of them gone); prunes x's lists on the way
  auto refresh = [&](int x, int k) {
    auto &ae = adje[x];
    size_t keep = 0;
    long bound = 0;
    for (int e : ae) if (!dead[e]) { ae[keep++] = e; bound += (long)elem[e].size() - 1; }
    ae.resize(keep);
    auto &av = adjv[x];
    av.erase(std::remove_if(av.begin(), av.end(), [&](int y) { return elim[y] != 0; }), av.end());
    const long cap = N - k - 1;
    long d;
    if (bound > 4L * cap + 64) d = std::min<long>(cap, bound + (long)av.size()); // long element lists: the sum bound (it saturates)
    else {
      const int st = ++stamp; mark[x] = st;
      d = 0;
      for (int y : av) if (mark[y] != st) { mark[y] = st; ++d; }
      for (int e : ae) for (int y : elem[e]) if (mark[y] != st) { mark[y] = st; ++d; }
    }
    return (int)std::min<long>(d, cap);
  };
  for (int k = 0; k < N; ++k) {
    int p = -1;
    for (;;) {
      while (mindeg <= N && head[mindeg] < 0) ++mindeg;
      if (mindeg > N) { mindeg = 0; continue; } // (cannot happen while nodes remain; never index past the lists)
      p = head[mindeg];
      if (!stale[p]) break;
      del(p);
      deg[p] = refresh(p, k);
      stale[p] = 0;
      ins(p);
      // normally at or above mindeg (the stored value was a lower bound) -- but a value that came from the saturating sum bound,
      // or from a cap that has shrunk since, can be above the true degree: follow it down
      if (deg[p] < mindeg) mindeg = deg[p];
    }
    if (N - k > 64 && (double)mindeg >= 0.7 * (double)(N - k - 1)) {
      // what is left is (close to) a clique: any order fills it in completely.  Finish in (bound) order; this block becomes
      // the dense tail of the factor.
      for (int dgr = mindeg; dgr <= N && k < N; ++dgr)
        for (int x = head[dgr]; x >= 0; x = nxt[x]) perm[k++] = x;
      break;
    }
    del(p);
    elim[p] = 1; perm[k] = p;
    ++stamp; mark[p] = stamp;
    Lp.clear();
    for (int x : adjv[p]) if (!elim[x] && mark[x] != stamp) { mark[x] = stamp; Lp.push_back(x); }
    for (int e : adje[p]) {
      if (dead[e]) continue;
      for (int x : elem[e]) if (!elim[x] && mark[x] != stamp) { mark[x] = stamp; Lp.push_back(x); }
      dead[e] = 1; std::vector<int>().swap(elem[e]);
    }
    std::vector<int>().swap(adjv[p]); std::vector<int>().swap(adje[p]);
    const int lp = (int)Lp.size();
    for (int x : Lp) {
      del(x);
      adje[x].push_back(p);
      deg[x] = std::max(std::max(deg[x] - 1, lp - 1), 0);
      stale[x] = 1;
      ins(x);
      if (deg[x] < mindeg) mindeg = deg[x];
    }
    elem[p] = Lp;
  }
}

inline int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// ABIP_HIP_TAIL: -1 / unset = choose by density, 0 = no dense tail, T > 0 = force (rounded down to a multiple of 64)
int g_tail_request = -2;
int tail_request() {
  if (g_tail_request != -2) return g_tail_request;
  const char *e = getenv("ABIP_HIP_TAIL");
  return e ? atoi(e) : -1;
}
int tail_cap() { const char *e = getenv("ABIP_HIP_TAIL_MAX"); return e ? atoi(e) : 24576; } // W and W' take 16 T^2 bytes of HBM (9.7 GB at the cap)

void level_sets(int N, TriHost &T, bool backward) {
  std::vector<int> lev(N, 0);
  int maxlev = 0;
  if (!backward) {
    for (int i = 0; i < N; ++i) { int l = 0; for (int q = T.ptr[i]; q < T.ptr[i + 1]; ++q) l = std::max(l, lev[T.idx[q]] + 1); lev[i] = l; maxlev = std::max(maxlev, l); }
  } else {
    for (int j = N - 1; j >= 0; --j) { int l = 0; for (int q = T.ptr[j]; q < T.ptr[j + 1]; ++q) l = std::max(l, lev[T.idx[q]] + 1); lev[j] = l; maxlev = std::max(maxlev, l); }
  }
  // bucket rows that have work (level >= 1 <=> at least one entry)
  std::vector<int> cnt(maxlev + 2, 0);
  for (int i = 0; i < N; ++i) if (lev[i] >= 1) cnt[lev[i]]++;
  T.lev_ptr.assign(1, 0);
  for (int l = 1; l <= maxlev; ++l) T.lev_ptr.push_back(T.lev_ptr.back() + cnt[l]);
  T.lev_rows.resize(T.lev_ptr.back());
  std::vector<int> pos(T.lev_ptr.begin(), T.lev_ptr.end());
  for (int i = 0; i < N; ++i) if (lev[i] >= 1) T.lev_rows[pos[lev[i] - 1]++] = i;
  const int nlev = (int)T.lev_ptr.size() - 1;
  T.lev_g.resize(nlev);
  for (int l = 0; l < nlev; ++l) {
    const int rows = T.lev_ptr[l + 1] - T.lev_ptr[l];
    long nn = 0;
    for (int r = T.lev_ptr[l]; r < T.lev_ptr[l + 1]; ++r) nn += T.ptr[T.lev_rows[r] + 1] - T.ptr[T.lev_rows[r]];
    int g = std::min(64, pow2_ceil((int)((nn + rows - 1) / std::max(rows, 1))));
    while (g > 1 && (long)rows * g > 4096) g >>= 1;
    T.lev_g[l] = std::max(g, 1);
  }
  // store the entries in level order: position r of lev_rows owns [ptr[r], ptr[r+1]), so that the device reads a row's extent
  // without first looking its index up
  const int npos = (int)T.lev_rows.size();
  std::vector<int> p2(npos + 1, 0), i2;
  std::vector<double> v2;
  for (int r = 0; r < npos; ++r) p2[r + 1] = p2[r] + (T.ptr[T.lev_rows[r] + 1] - T.ptr[T.lev_rows[r]]);
  i2.resize(p2[npos]); v2.resize(p2[npos]);
  par_by_entries(p2.data(), (long)npos, 500000, [&](long r0, long r1, int) {
    for (long r = r0; r < r1; ++r) {
      const int row = T.lev_rows[r];
      std::copy(T.idx.begin() + T.ptr[row], T.idx.begin() + T.ptr[row + 1], i2.begin() + p2[r]);
      std::copy(T.val.begin() + T.ptr[row], T.val.begin() + T.ptr[row + 1], v2.begin() + p2[r]);
    }
  });
  T.ptr.swap(p2); T.idx.swap(i2); T.val.swap(v2);
}

} // namespace

void set_tail_request(int t) { g_tail_request = t; }
namespace { const std::vector<int> *g_order_hint = nullptr; }
void set_order_hint(const std::vector<int> *P) { g_order_hint = P; }

int host_solve(const LdlHost &F, std::vector<double> &b) {
  F.wait_forms();
  const int N = F.N, t0 = F.t0, T = F.T;
  if ((int)b.size() != N) return -1;
  std::vector<double> x(N);
  for (int k = 0; k < N; ++k) x[k] = b[F.P[k]];
  auto sweep = [&](const TriHost &Tr) { // levels in order; position r of lev_rows owns [ptr[r], ptr[r+1])
    for (size_t r = 0; r < Tr.lev_rows.size(); ++r) {
      double acc = 0.0;
      for (int q = Tr.ptr[r]; q < Tr.ptr[r + 1]; ++q) acc += Tr.val[q] * x[Tr.idx[q]];
      x[Tr.lev_rows[r]] -= acc;
    }
  };
  sweep(F.fwd);
  std::vector<double> D(F.D);
  if (T > 0) { // dense LDL' of the Schur complement, then the two triangular solves of the tail (the device applies inv(L22) instead)
    std::vector<double> S(F.S);
    if (F.dev_schur) { LdlHost G; G.t0 = t0; G.T = T; G.D = F.D; G.S.swap(S); G.k22_row = F.k22_row; G.k22_col = F.k22_col; G.k22_val = F.k22_val; G.l21_ptr = F.l21_ptr; G.l21_row = F.l21_row; G.l21_val = F.l21_val; complete_schur_on_host(G); S.swap(G.S); }
    for (int c = 0; c < T; ++c) {
      const double d = S[(size_t)c * T + c];
      if (d == 0.0) return -2;
      D[t0 + c] = d;
      for (int r = c + 1; r < T; ++r) S[(size_t)r * T + c] /= d;
      for (int r = c + 1; r < T; ++r) {
        const double lr = S[(size_t)r * T + c] * d;
        for (int cc = c + 1; cc <= r; ++cc) S[(size_t)r * T + cc] -= lr * S[(size_t)cc * T + c];
      }
    }
    for (int r = 0; r < T; ++r) { double acc = 0.0; for (int c = 0; c < r; ++c) acc += S[(size_t)r * T + c] * x[t0 + c]; x[t0 + r] -= acc; }
    for (int r = 0; r < T; ++r) x[t0 + r] /= D[t0 + r];
    for (int r = T - 1; r >= 0; --r) { double acc = 0.0; for (int c = r + 1; c < T; ++c) acc += S[(size_t)c * T + r] * x[t0 + c]; x[t0 + r] -= acc; }
  }
  for (int k = 0; k < t0; ++k) x[k] /= D[k];
  sweep(F.bwd);
  for (int k = 0; k < N; ++k) b[F.P[k]] = x[k];
  return 0;
}

void kkt_upper(const ABIPMatrix *A, double rho_y, std::vector<int> &Kp, std::vector<int> &Ki, std::vector<double> &Kx) {
  const int m = (int)A->m, n = (int)A->n, N = m + n;
  const long nnzA = (long)A->p[n];
  // upper triangle of K by columns (direct.c:49-104)
  Kp.assign(N + 1, 0); Ki.assign(N + nnzA, 0); Kx.assign(N + nnzA, 0.0);
  long kk = 0;
  for (int i = 0; i < m; ++i) { Kp[i] = (int)kk; Ki[kk] = i; Kx[kk] = rho_y; ++kk; }
  for (int j = 0; j < n; ++j) {
    Kp[m + j] = (int)kk;
    for (abip_int q = A->p[j]; q < A->p[j + 1]; ++q) { Ki[kk] = (int)A->i[q]; Kx[kk] = A->x[q]; ++kk; }
    Ki[kk] = m + j; Kx[kk] = -1.0; ++kk;
  }
  Kp[N] = (int)kk;
}
int factor_kkt(const ABIPMatrix *A, double rho_y, LdlHost &out) {
  std::vector<int> Kp, Ki;
  std::vector<double> Kx;
  kkt_upper(A, rho_y, Kp, Ki, Kx);
  return factor_upper((int)(A->m + A->n), Kp, Ki, Kx, out);
}
double sym_upper_residual(int N, const std::vector<int> &Kp, const std::vector<int> &Ki, const std::vector<double> &Kx,
                          const std::vector<double> &z, const std::vector<double> &rhs) {
  std::vector<double> y(N, 0.0);
  for (int j = 0; j < N; ++j)
    for (int q = Kp[j]; q < Kp[j + 1]; ++q) {
      const int i = Ki[q];
      y[i] += Kx[q] * z[j];
      if (i != j) y[j] += Kx[q] * z[i];
    }
  double num = 0, den = 0;
  for (int i = 0; i < N; ++i) { const double e = y[i] - rhs[i]; num += e * e; den += rhs[i] * rhs[i]; }
  const double r = std::sqrt(num) / std::max(std::sqrt(den), 1e-300);
  return (r == r) ? r : 1e300; // NaN -> fail
}
void guard_rhs(int N, std::vector<double> &rhs) {
  rhs.resize(N);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < N; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; rhs[i] = (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
}

namespace { int g_dev_schur_req = -1; }
void set_dev_schur_request(int v) { g_dev_schur_req = v; }
static int dev_schur_request() {
  if (g_dev_schur_req >= 0) return g_dev_schur_req;
  const char *e = getenv("ABIP_HIP_DEV_SCHUR");
  return e ? atoi(e) : -1;
}
void complete_schur_on_host(LdlHost &F) {
  const int t0 = F.t0, T = F.T;
  if (F.S.empty() && T > 0) { // K22 from its triplets
    F.S.assign((size_t)T * T, 0.0);
    for (size_t q = 0; q < F.k22_val.size(); ++q) F.S[(size_t)F.k22_row[q] * T + F.k22_col[q]] += F.k22_val[q];
  }
  if ((long)F.l21_ptr.size() == (long)t0 + 1)
    for (int c = 0; c < t0; ++c) {
      const double d = F.D[c];
      for (long a = F.l21_ptr[c]; a < F.l21_ptr[c + 1]; ++a) {
        const double la = F.l21_val[a] * d;
        double *Srow = F.S.data() + (size_t)F.l21_row[a] * T;
        for (long b = F.l21_ptr[c]; b <= a; ++b) Srow[F.l21_row[b]] -= la * F.l21_val[b];
      }
    }
  F.dev_schur = false;
}

int factor_upper(int N, const std::vector<int> &Kp, const std::vector<int> &Ki, const std::vector<double> &Kx, LdlHost &out) {
  out.wait_forms(); out.forms_job = std::shared_future<void>(); // (a second factorisation into the same object: the first one's forms thread is through)
  out.N = N;
  const long kk = Kp[N];
  const bool tm = getenv("ABIP_HIP_SETUP_TIMES") != nullptr;
  auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tq = clk();
  if (g_order_hint && (int)g_order_hint->size() == N) { out.P = *g_order_hint; if (tm) printf("[setup] ordering: the caller's elimination order\n"); }
  else { // symmetric adjacency (no diagonal) for the minimum-degree pass (not built when the caller brings the order: two passes over the matrix)
    std::vector<int> Gp(N + 1, 0);
    for (int j = 0; j < N; ++j) for (int q = Kp[j]; q < Kp[j + 1]; ++q) if (Ki[q] != j) { Gp[Ki[q] + 1]++; Gp[j + 1]++; }
    for (int i = 0; i < N; ++i) Gp[i + 1] += Gp[i];
    std::vector<int> Gi(Gp[N]), pos(Gp.begin(), Gp.end() - 1);
    for (int j = 0; j < N; ++j) for (int q = Kp[j]; q < Kp[j + 1]; ++q) if (Ki[q] != j) { Gi[pos[Ki[q]]++] = j; Gi[pos[j]++] = Ki[q]; }
    min_degree(N, Gp, Gi, out.P);
  }
  if (tm) { printf("[setup] ordering %.3f s\n", clk() - tq); tq = clk(); }
  std::vector<int> Pinv(N);
  for (int i = 0; i < N; ++i) Pinv[out.P[i]] = i;
  // C = upper triangle of P K P' by columns (cs_symperm, direct.c:259-260): a stable bucket pass over the entries of K (host_par.h)
  std::vector<int> Cp, Ci(kk);
  std::vector<double> Cx(kk);
  par_bucket((long)N, Kp.data(), (long)N, Cp, 1000000, [&](long j, long q) { return (long)std::max(Pinv[Ki[q]], Pinv[j]); },
             [&](long dst, long j, long q) { Ci[dst] = std::min(Pinv[Ki[q]], Pinv[j]); Cx[dst] = Kx[q]; });
  // elimination tree
  std::vector<int> parent(N, -1), anc(N, -1), flag(N, -1), lnz(N, 0);
  for (int j = 0; j < N; ++j)
    for (int q = Cp[j]; q < Cp[j + 1]; ++q) {
      int r = Ci[q];
      while (r != -1 && r < j) { const int nx = anc[r]; anc[r] = j; if (nx == -1) parent[r] = j; r = nx; }
    }

  // ---- head / dense-tail split -----------------------------------------------------------------------------------
  // With a fill-reducing ordering the last pivots form a (nearly) dense trailing block whose rows are one level each:
  // the sequential part of a level-scheduled solve.  Columns >= t0 are therefore not kept as a sparse factor: the host
  // computes only the Schur complement S onto them (sparse arithmetic), the device factors S densely and applies
  // inv(L22) as two dense triangular mat-vecs (dev_ldl.h).  (Chosen from the tree alone, before the column counts: see below.)
  int T = 0;
  {
    const int Tmax = std::min(N - 1, tail_cap());
    const int req = tail_request();
    if (req > 0) T = std::min(req, Tmax) / 64 * 64;
    else if (req < 0 && N >= 256) {
      // The levels of the forward (and backward) solve over the head [0, t0) are the heights of the elimination forest
      // restricted to it.  Pick T to minimise  levels * (cost of a level) + bytes of the two dense triangles / bandwidth.
      std::vector<int> hgt(N, 0), pm(N + 1, 0);
      for (int i = 0; i < N; ++i) if (parent[i] >= 0) hgt[parent[i]] = std::max(hgt[parent[i]], hgt[i] + 1);
      for (int i = 0; i < N; ++i) pm[i + 1] = std::max(pm[i], hgt[i] + 1);
      // us per level (one-workgroup path with x in LDS, or a launch per level on larger systems), us per mat-vec launch, bytes per us
      const char *ce = getenv("ABIP_HIP_CLEV");
      const double c_lev = ce ? atof(ce) : (N <= 16384 ? 1.5 : 6.0), c_mv = 3.0, bw = 3.0e6;
      double best = 2.0 * c_lev * pm[N];
      for (int t = 64; t <= Tmax; t += 64) {
        const double cost = 2.0 * c_lev * pm[N - t] + 2.0 * (c_mv + 4.0 * (double)t * t / bw);
        if (cost < best) { best = cost; T = t; }
      }
    }
  }
  const int t0 = N - T;
  // column counts (LDL_symbolic, ldl.c:70-120): row j's pattern = the tree paths from its entries up to j.  Nothing ever reads the pattern of a LARGE dense tail
  // (it is factored as a dense block on the device), and walking it is T^2 / 2 steps (C5: 5e7 of the 5.45e7 entries of L, 0.1 s): from T = 2048 on the walks stop
  // at the first tail column and the tail is counted as the dense triangle it is stored as.
  const bool skip_tail_pattern = T >= 2048;
  const int lim = skip_tail_pattern ? t0 : N;
  for (int j = 0; j < N; ++j) {
    flag[j] = j;
    for (int q = Cp[j]; q < Cp[j + 1]; ++q) { int r = Ci[q]; while (r < j && r < lim && flag[r] != j) { lnz[r]++; flag[r] = j; r = parent[r]; } }
  }
  std::vector<long> Lp(N + 1, 0);
  for (int j = 0; j < N; ++j) Lp[j + 1] = Lp[j] + lnz[j];
  out.lnnz = Lp[N] + (skip_tail_pattern ? (long)T * (T - 1) / 2 : 0L);
  if (tm) { printf("[setup] symbolic %.3f s (nnz(L) = %ld%s)\n", clk() - tq, out.lnnz, skip_tail_pattern ? ", the tail counted as dense" : ""); tq = clk(); }
  out.S.clear(); out.k22_row.clear(); out.k22_col.clear(); out.k22_val.clear();
  out.t0 = t0; out.T = T;

  // numeric, up-looking: row k of L by a sparse triangular solve against the leading block.  For a tail row only the head
  // columns take part; what is left in Y on the tail positions is row k of S.
  const long Lhead = Lp[t0];
  std::vector<int> Li(std::max<long>(Lhead, 1));
  std::vector<double> Lx(std::max<long>(Lhead, 1));
  out.D.assign(N, 0.0);
  std::vector<double> Y(N, 0.0);
  std::vector<int> stack(N), pat(N), fill(N, 0), hfill;
  std::fill(flag.begin(), flag.end(), -1);
  if (t0 == 0) hfill = fill;
  for (int k = 0; k < N; ++k) {
    int top = N; flag[k] = k;
    double dk = 0.0;
    for (int q = Cp[k]; q < Cp[k + 1]; ++q) {
      int r = Ci[q];
      if (r == k) { dk += Cx[q]; continue; }
      if (r >= t0) { out.k22_row.push_back(k - t0); out.k22_col.push_back(r - t0); out.k22_val.push_back(Cx[q]); continue; } // tail-tail: an entry of K22 (its etree path stays in the tail)
      Y[r] += Cx[q];
      int len = 0;
      while (r < t0 && flag[r] != k) { pat[len++] = r; flag[r] = k; r = parent[r]; } // (a tail column's ancestors are tail columns, and those are skipped below: the walk ends at the head's edge)
      while (len > 0) stack[--top] = pat[--len];
    }
    if (k == t0) hfill = fill; // from here on only tail rows arrive: a head column's first hfill entries are its head rows
    const bool tail_row = k >= t0;
    for (; top < N; ++top) {
      const int c = stack[top];
      if (c >= t0) continue;       // tail column: its contribution is applied on the device
      const double yc = Y[c];
      Y[c] = 0.0;
      const long e = Lp[c] + fill[c];
      // a tail row needs the head rows of column c only: the products of its tail entries are the Schur complement's L21 D1 L21',
      // accumulated afterwards (complete_schur_on_host, or the device)
      const long eu = tail_row ? Lp[c] + hfill[c] : e;
      for (long q = Lp[c]; q < eu; ++q) Y[Li[q]] -= Lx[q] * yc;
      const double lkc = yc / out.D[c];
      if (!tail_row) dk -= lkc * yc;
      Li[e] = k; Lx[e] = lkc; fill[c]++;
    }
    if (k >= t0) { out.k22_row.push_back(k - t0); out.k22_col.push_back(k - t0); out.k22_val.push_back(dk); continue; }
    out.D[k] = dk;
    if (dk == 0.0) return -1;
  }
  if (tm) { printf("[setup] numeric (head + L21, T = %d) %.3f s\n", T, clk() - tq); tq = clk(); }
  out.dev_schur = false; out.l21_ptr.clear(); out.l21_row.clear(); out.l21_val.clear();
  if (T > 0) { // L21 by head column, and who multiplies it out
    if ((int)hfill.size() != N) hfill = fill;
    out.l21_ptr.assign(t0 + 1, 0);
    double cost = 0.0;
    for (int c = 0; c < t0; ++c) { const long kc = fill[c] - hfill[c]; out.l21_ptr[c + 1] = out.l21_ptr[c] + kc; cost += 0.5 * (double)kc * (double)(kc + 1); }
    out.l21_row.resize(out.l21_ptr[t0]); out.l21_val.resize(out.l21_ptr[t0]);
    for (int c = 0; c < t0; ++c) {
      long dst = out.l21_ptr[c];
      for (long q = Lp[c] + hfill[c]; q < Lp[c] + fill[c]; ++q, ++dst) { out.l21_row[dst] = Li[q] - t0; out.l21_val[dst] = Lx[q]; }
    }
    // who multiplies it out: the host's scattered updates of S run at ~2.4e8 multiply-adds per second (measured: 1.17e8 in 0.49 s on C5's
    // KKT matrix), the device's dense panels at ~3 TFLOP/s over T^2 x (head columns that reach the tail) whatever their sparsity
    long ncol = 0;
    for (int c = 0; c < t0; ++c) ncol += out.l21_ptr[c + 1] > out.l21_ptr[c];
    // measured (profiles/r02x, r02z): host 1.17e8 multiply-adds in 0.49 s; dense panels on the matrix cores 5056^2 x 35002 in 0.058 s (0.077 s with the FMA tiles);
    // row-wise sparse kernel 1.8e8 wavefront steps in 0.28 s (LASSO protocol), 6e6 in 0.04 s with its uploads (C5)
    const double host_s = cost / 2.4e8, dev_s = 0.02 + (double)T * (double)T * (double)ncol / 1.5e13;
    // the device's row-wise sparse form: one wavefront step per 64 entries of a column prefix, ~512 wavefronts at a time, ~0.8 us per step, plus the upload of L21 twice
    double steps = 0;
    for (int c = 0; c < t0; ++c) { const double kc = (double)(out.l21_ptr[c + 1] - out.l21_ptr[c]); steps += kc * (1.0 + kc / 128.0); }
    const double rows_s = T <= 20480 ? 0.01 + steps / 512.0 * 0.8e-6 + 32.0 * (double)out.l21_ptr[t0] / 5.0e9 : 1e30;
    const int req = dev_schur_request();
    out.dev_schur = req == 1 || req == 2 || (req < 0 && host_s > 0.1 && std::min(dev_s, rows_s) < host_s);
    out.schur_rows = out.dev_schur && T <= 20480 && (req == 2 || (req != 1 && rows_s < dev_s));
    if (!out.dev_schur) {
      try { complete_schur_on_host(out); }
      catch (const std::bad_alloc &) { return -3; } // no room for the dense block on the host
    }
    if (tm) { printf("[setup] Schur complement L21 D L21' (%.2e multiply-adds): %s %.3f s\n", cost, out.dev_schur ? (out.schur_rows ? "left to the device (row-wise, sparse)" : "left to the device (dense panels)") : "host", clk() - tq); tq = clk(); }
  }
  // backward form = CSC of the head columns of L (all rows); forward form = CSR of [L11; L21].  The two are independent: the backward one is built on
  // a thread of its own while this one transposes (host_par.h) and levels the forward one.  Nothing of the dense tail's set-up reads them: with a tail
  // and a large head the whole job runs behind the caller's back (LdlHost::forms_job), the device starts on the Schur complement meanwhile
  struct Head { std::vector<long> Lp; std::vector<int> Li; std::vector<double> Lx; };
  auto head = std::make_shared<Head>();
  head->Lp.swap(Lp); head->Li.swap(Li); head->Lx.swap(Lx);
  LdlHost *o = &out;
  const char *fe = getenv("ABIP_HIP_FORMS_ASYNC"); // 0: never, 1: always (tests), unset: large heads in front of a dense tail
  const bool async = fe ? atoi(fe) != 0 : (T > 0 && Lhead >= 1000000);
  auto forms = [head, o, N, t0, Lhead, tm, clk, async]() {
    const double ts = clk();
    const std::vector<long> &Lp = head->Lp; const std::vector<int> &Li = head->Li; const std::vector<double> &Lx = head->Lx;
    std::thread back([&] {
      o->bwd.ptr.assign(N + 1, 0);
      for (int j = 0; j <= N; ++j) o->bwd.ptr[j] = (int)Lp[std::min(j, t0)];
      o->bwd.idx.assign(Li.begin(), Li.begin() + Lhead); o->bwd.val.assign(Lx.begin(), Lx.begin() + Lhead);
      level_sets(N, o->bwd, true);
    });
    par_transpose(N, t0, Lp.data(), Li.data(), Lx.data(), o->fwd.ptr, o->fwd.idx, o->fwd.val);
    level_sets(N, o->fwd, false);
    back.join();
    if (tm) printf("[setup] forward / backward forms + level sets %.3f s%s\n", clk() - ts, async ? " (beside the device's work on the tail)" : "");
  };
  if (async) out.forms_job = std::async(std::launch::async, forms).share();
  else forms();
  return 0;
}

} // namespace host
} // namespace abip
