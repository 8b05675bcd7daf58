/*
 * abip_linsys.h -- the reference's LINEAR-SYSTEM PLUG-IN interface over the MI355X back-ends.
 *
 * The reference selects its KKT back-end at link time: src/abip-lp/src/abip.c calls the functions of
 * src/abip-lp/include/linsys.h:10-91 and is linked with ONE of linsys/direct.c (LDL'; compile_direct.m:62-78) or linsys/indirect.c (PCG; compile_indirect.m).
 * lib/libabip_hip_linsys.so is a third such variant: it defines exactly the symbols those two files define, so the
 * reference's own abip.c (+ linalg.c, adaptive.c, normalize.c, util.c, cs.c, ctrlc.c, abip_version.c and
 * linsys/common.c, all unchanged) links against it in place of direct.c + ldl.c + the AMD sources:
 *
 *   symbol (ABIP(x) = abip_x, glbopts.h:10-12)    replaces
 *   abip_init_lin_sys_work ........ direct.c:273-303 / indirect.c:282-318   A (scaled) -> device: host LDL' + device triangular solves, or PCG
 *   abip_solve_lin_sys ............ direct.c:305-328 / indirect.c:393-434   b (m+n) <- K^-1 b, K = [[rho_y I, A], [A', -I]]
 *   abip_accum_by_A / _Atrans ..... direct.c:200-208 / indirect.c:222-242   y += A x, y += A' x (device SpMV)
 *   abip_normalize_A / un_ ........ direct.c:210-216 / indirect.c:244-261   the scaling of linsys/common.c:150-594 (host)
 *   abip_get_lin_sys_method / _summary, abip_free_lin_sys_work(_pds)        direct.c:5-47, indirect.c:8-34,141-203
 *
 * Everything else of the reference (the ADMM loop, the barrier prox, the mu rules) then runs on the CPU as before and
 * every solve crosses PCIe twice: this is the smallest possible change to the reference, not the fast path -- the fast
 * path is the whole loop on the device behind abip_init / abip_solve (include/abip.h).  The back-end is chosen by
 * ABIP_HIP_LINSYS=direct|indirect (default direct) when abip_init_lin_sys_work runs.
 *
 * The library is built for DLONG (abip_int = long, the mex default).  Strings and the D / E vectors handed to the caller
 * are released by the caller with abip_free (glbopts.h:52-79): malloc / free unless abip_hip_linsys_set_allocator says
 * otherwise (a MATLAB_MEX_FILE build must pass mxMalloc / mxFree).
 */
#ifndef ABIP_LINSYS_PLUGIN_H
#define ABIP_LINSYS_PLUGIN_H

#include <stddef.h>

#include "abip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct ABIP_LIN_SYS_WORK ABIPLinSysWork; /* opaque, as in src/abip-lp/include/abip.h:12 */
typedef struct ABIP_SCALING {                    /* src/abip-lp/include/abip.h:107-114 */
  abip_float *D;
  abip_float *E;
  abip_float mean_norm_row_A;
  abip_float mean_norm_col_A;
} ABIPScaling;

ABIPLinSysWork *abip_init_lin_sys_work(const ABIPMatrix *A, const ABIPSettings *stgs);
abip_int abip_solve_lin_sys(const ABIPMatrix *A, const ABIPSettings *stgs, ABIPLinSysWork *p, abip_float *b, const abip_float *s, abip_int iter);
void abip_free_lin_sys_work(ABIPLinSysWork *p);
void abip_free_lin_sys_work_pds(ABIPLinSysWork *p, ABIPMatrix *A);
void abip_accum_by_Atrans(const ABIPMatrix *A, ABIPLinSysWork *p, const abip_float *x, abip_float *y);
void abip_accum_by_A(const ABIPMatrix *A, ABIPLinSysWork *p, const abip_float *x, abip_float *y);
char *abip_get_lin_sys_method(const ABIPMatrix *A, const ABIPSettings *stgs);
char *abip_get_lin_sys_summary(ABIPLinSysWork *p, const ABIPInfo *info);
void abip_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, ABIPScaling *scal);
void abip_un_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, const ABIPScaling *scal);

/* not in linsys.h: the allocator the caller's abip_free pairs with (default malloc / free) */
void abip_hip_linsys_set_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *));

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
