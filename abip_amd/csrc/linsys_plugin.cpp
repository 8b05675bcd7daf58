// libabip_hip_linsys.so: the reference's linear-system plug-in interface (src/abip-lp/include/linsys.h:10-91) over the device
// back-ends of this library -- the symbols linsys/direct.c and linsys/indirect.c define, so that the reference's own abip.c links
// against the GPU in place of either (include/abip_linsys.h).  The work behind the opaque handle is an
// ordinary ABIPWork set up on the matrix as the reference hands it over (already scaled: normalize off), of which only the KKT
// back-end and the two SpMV images are used; host vectors in, host vectors out.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/abip.h"
#include "../../include/abip_hip.h"
#include "../../include/abip_linsys.h"
#include "host_setup.h"

struct ABIP_LIN_SYS_WORK {
  ABIPWork *w = nullptr;
  ABIPSettings stgs{};   // private copy: normalize = 0 (A arrives scaled), verbose = 0
  int linsys = 0;        // ABIP_HIP_LINSYS_DIRECT / _INDIRECT, fixed at init
  double total_solve_ms = 0.0;
  long tot_cg_its = 0;
};

namespace {
void *(*g_alloc)(size_t) = malloc;
void (*g_free)(void *) = free;
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
char *new_str() { char *s = (char *)g_alloc(128); if (s) s[0] = 0; return s; }
} // namespace

extern "C" {

void abip_hip_linsys_set_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *)) {
  g_alloc = alloc_fn ? alloc_fn : malloc;
  g_free = free_fn ? free_fn : free;
}

char *abip_get_lin_sys_method(const ABIPMatrix *A, const ABIPSettings *stgs) { // direct.c:5-13, indirect.c:8-18
  char *s = new_str();
  if (!s) return s;
  if (abip_hip_get_linsys() == ABIP_HIP_LINSYS_INDIRECT) snprintf(s, 128, "sparse-indirect on MI355X, nnz in A = %li, CG tol ~ 1/iter^(%2.2f)", (long)A->p[A->n], stgs->cg_rate);
  else snprintf(s, 128, "sparse-direct on MI355X, nnz in A = %li", (long)A->p[A->n]);
  return s;
}

char *abip_get_lin_sys_summary(ABIPLinSysWork *p, const ABIPInfo *info) { // direct.c:15-26, indirect.c:20-34
  char *s = new_str();
  if (!s || !p) return s;
  const double per = p->total_solve_ms / (double)(info->admm_iter + 1) / 1e3;
  if (p->linsys == ABIP_HIP_LINSYS_INDIRECT) snprintf(s, 128, "\tLin-sys: avg # CG iterations: %2.2f, avg solve time: %1.2es\n", (double)p->tot_cg_its / (double)(info->admm_iter + 1), per);
  else snprintf(s, 128, "\tLin-sys: nnz in L factor: %li, avg solve time: %1.2es\n", (long)abip_hip_get_scalar(p->w, "lnnz"), per);
  p->tot_cg_its = 0;
  p->total_solve_ms = 0.0;
  return s;
}

void abip_free_lin_sys_work(ABIPLinSysWork *p) {
  if (!p) return;
  if (p->w) abip_finish(p->w);
  delete p;
}
void abip_free_lin_sys_work_pds(ABIPLinSysWork *p, ABIPMatrix *) { abip_free_lin_sys_work(p); }

ABIPLinSysWork *abip_init_lin_sys_work(const ABIPMatrix *A, const ABIPSettings *stgs) {
  if (!A || !stgs) return nullptr;
  ABIPLinSysWork *p = new ABIPLinSysWork();
  p->stgs = *stgs;
  p->stgs.normalize = 0;
  p->stgs.verbose = 0;
  p->linsys = abip_hip_get_linsys();
  // abip_init validates and keeps a whole problem: b and c play no part in the back-end
  std::vector<abip_float> zb((size_t)A->m, 0.0), zc((size_t)A->n, 0.0);
  ABIPData d{};
  d.m = A->m; d.n = A->n; d.A = const_cast<ABIPMatrix *>(A); d.b = zb.data(); d.c = zc.data(); d.sp = 0.0; d.stgs = &p->stgs;
  ABIPInfo info{};
  const int copy_before = abip_hip_get_copy_a_matrix();
  abip_hip_set_copy_a_matrix(1); // the caller's A is const here
  p->w = abip_init(&d, &info);
  abip_hip_set_copy_a_matrix(copy_before); // (a process-wide switch: leave it as the host program had it)
  if (!p->w) { delete p; return nullptr; }
  return p;
}

abip_int abip_solve_lin_sys(const ABIPMatrix *, const ABIPSettings *, ABIPLinSysWork *p, abip_float *b, const abip_float *s, abip_int iter) {
  if (!p || !p->w || !b) return -1;
  const double t0 = now_ms();
  const abip_int its = abip_hip_kkt_solve(p->w, b, s, iter);
  p->total_solve_ms += now_ms() - t0;
  if (its < 0) return -1;
  if (iter >= 0) p->tot_cg_its += (long)its; // indirect.c:422-425
  return 0;
}

// The interface has no error channel.  A device failure fills y with NaN: the reference's own loop then reports "Failure" through its
// NaN handling (abip.c:219-277) instead of the host process (Matlab) being killed.
void abip_accum_by_Atrans(const ABIPMatrix *A, ABIPLinSysWork *p, const abip_float *x, abip_float *y) {
  if (!p || !p->w || abip_hip_accum_by_Atrans(p->w, x, y) != 0) {
    fprintf(stderr, "abip_hip linsys plug-in: accum_by_Atrans failed on the device\n");
    if (A && y) for (abip_int j = 0; j < A->n; ++j) y[j] = NAN;
  }
}
void abip_accum_by_A(const ABIPMatrix *A, ABIPLinSysWork *p, const abip_float *x, abip_float *y) {
  if (!p || !p->w || abip_hip_accum_by_A(p->w, x, y) != 0) {
    fprintf(stderr, "abip_hip linsys plug-in: accum_by_A failed on the device\n");
    if (A && y) for (abip_int i = 0; i < A->m; ++i) y[i] = NAN;
  }
}

void abip_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, ABIPScaling *scal) { // ABIP(_normalize_A), linsys/common.c:150-565
  std::vector<double> D, E;
  abip::host::normalize_A(A, stgs, D, E, &scal->mean_norm_row_A, &scal->mean_norm_col_A);
  scal->D = (abip_float *)g_alloc(sizeof(abip_float) * D.size());
  scal->E = (abip_float *)g_alloc(sizeof(abip_float) * E.size());
  if (!scal->D || !scal->E) { fprintf(stderr, "abip_hip linsys plug-in: the allocator returned NULL for the scaling vectors\n"); abort(); } // the caller dereferences them unconditionally (abip.c)
  memcpy(scal->D, D.data(), sizeof(abip_float) * D.size());
  memcpy(scal->E, E.data(), sizeof(abip_float) * E.size());
}
void abip_un_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, const ABIPScaling *scal) { // linsys/common.c:569-594
  const std::vector<double> D(scal->D, scal->D + A->m), E(scal->E, scal->E + A->n);
  abip::host::un_normalize_A(A, stgs, D, E);
}

} // extern "C"
