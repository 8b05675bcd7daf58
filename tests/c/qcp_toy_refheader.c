/* The reference's literal toy problem (test/test_abip_install.m:32-43) through abip() -- compiled against the REFERENCE'S OWN conic header
 * (src/abip-qcp/include/abip.h, from where it lies: oracle/Makefile, target ref) and linked against libabip_hip_qcp.so.  Checks, with the compiler, that
 * the library's struct layouts are the reference's, and on the GPU that the entry under the reference's name reproduces the recorded output
 * (10 IPM / 91 ADMM iterations, pobj -0.984063813). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "abip.h"
void abip_set_default_settings(ABIPData *d); /* source/util.c:203-255 (declared in the reference's util.h, which pulls more than this program needs) */

int main(void) {
  /* min 1/2 x'Qx + c'x, Ax = b, x in K: A 2 x 8 dense, Q = I, K.q = [3], K.rq = [3], K.f = 1, K.l = 1 */
  const double Ad[2][8] = {{1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 1, 2, 3, 1, 2}};
  ABIPMatrix A, Q;
  ABIPData d;
  ABIPSettings st;
  ABIPCone K;
  ABIPSolution sol;
  ABIPInfo info;
  abip_float Ax[16], Qx[8], b[2] = {4.0, 3.0}, c[8] = {1, 0, 2, 1, 4, 2, 3, 0};
  abip_int Ai[16], Ap[9], Qi[8], Qp[9], q1[1] = {3}, rq1[1] = {3};
  int nz = 0, j, i;
  memset(&d, 0, sizeof d); memset(&st, 0, sizeof st); memset(&K, 0, sizeof K); memset(&sol, 0, sizeof sol); memset(&info, 0, sizeof info);
  for (j = 0; j < 8; ++j) {
    Ap[j] = nz;
    for (i = 0; i < 2; ++i) if (Ad[i][j] != 0.0) { Ax[nz] = Ad[i][j]; Ai[nz] = i; ++nz; }
    Qp[j] = j; Qi[j] = j; Qx[j] = 1.0;
  }
  Ap[8] = nz; Qp[8] = 8;
  A.m = 2; A.n = 8; A.x = Ax; A.i = Ai; A.p = Ap;
  Q.m = 8; Q.n = 8; Q.x = Qx; Q.i = Qi; Q.p = Qp;
  d.m = 2; d.n = 8; d.A = &A; d.Q = &Q; d.b = b; d.c = c; d.stgs = &st;
  abip_set_default_settings(&d);
  st.verbose = 0; st.linsys_solver = 1; st.prob_type = 2; /* the generic conic formulation, as abip_qcp_mex.c sets it */
  st.eps = st.eps_p = st.eps_d = st.eps_g = st.eps_inf = st.eps_unb = 1e-6;
  K.q = q1; K.qsize = 1; K.rq = rq1; K.rqsize = 1; K.f = 1; K.z = 0; K.l = 1;
  {
    const abip_int status = abip(&d, &sol, &info, &K);
    printf("status %ld ipm %ld admm %ld pobj %.9f dobj %.9f x0 %.6f\n", (long)status, (long)info.ipm_iter, (long)info.admm_iter, (double)info.pobj, (double)info.dobj, sol.x ? (double)sol.x[0] : 0.0);
  }
  return 0;
}
