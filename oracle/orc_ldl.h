/*
 * orc_ldl.h -- TEST INFRASTRUCTURE (shared by the LP and QCP oracles): fill-reducing ordering and sparse LDL'
 * of a symmetric quasi-definite matrix given by its upper triangle in CSC form, plus the permuted solve.
 * Own code: exact minimum-degree on the quotient graph (the reference calls SuiteSparse AMD:
 * src/abip-lp/linsys/direct.c:106-119, src/abip-qcp/source/linsys.c:270-297) and an up-looking factorisation
 * (the reference: external/ldl/ldl.c LDL_symbolic/LDL_numeric, external/qdldl QDLDL_etree/QDLDL_factor).
 */
#ifndef ORC_LDL_H
#define ORC_LDL_H
#include <stdlib.h>
#include <string.h>

typedef long orc_I;
typedef double orc_F;
#define I orc_I
#define F orc_F

typedef struct { I *v; I len, cap; } IVec;
static void iv_push(IVec *a, I x) {
  if (a->len == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (I *)realloc(a->v, sizeof(I) * a->cap); }
  a->v[a->len++] = x;
}

/* Exact minimum (external) degree ordering with element absorption.
 * Input: symmetric pattern without diagonal as adjacency (Gp, Gi) of N nodes. */
static void min_degree_order(I N, const I *Gp, const I *Gi, I *perm) {
  IVec *adjv = (IVec *)calloc((size_t)N, sizeof(IVec)); /* variable neighbours */
  IVec *adje = (IVec *)calloc((size_t)N, sizeof(IVec)); /* element neighbours  */
  IVec *elem = (IVec *)calloc((size_t)N, sizeof(IVec)); /* variables of element e */
  I *deg = (I *)malloc(sizeof(I) * N), *mark = (I *)malloc(sizeof(I) * N);
  char *elim = (char *)calloc((size_t)N, 1), *dead_e = (char *)calloc((size_t)N, 1);
  /* degree buckets as doubly linked lists */
  I *head = (I *)malloc(sizeof(I) * (N + 1)), *nxt = (I *)malloc(sizeof(I) * N), *prv = (I *)malloc(sizeof(I) * N);
  I i, k, stamp = 0, mindeg = 0;
  for (i = 0; i <= N; ++i) head[i] = -1;
  for (i = 0; i < N; ++i) {
    mark[i] = -1;
    for (I q = Gp[i]; q < Gp[i + 1]; ++q) if (Gi[q] != i) iv_push(&adjv[i], Gi[q]);
    deg[i] = adjv[i].len;
  }
#define BUCKET_INS(x) do { I d_ = deg[x]; nxt[x] = head[d_]; prv[x] = -1; if (head[d_] >= 0) prv[head[d_]] = (x); head[d_] = (x); } while (0)
#define BUCKET_DEL(x) do { I d_ = deg[x]; if (prv[x] >= 0) nxt[prv[x]] = nxt[x]; else head[d_] = nxt[x]; if (nxt[x] >= 0) prv[nxt[x]] = prv[x]; } while (0)
  for (i = N - 1; i >= 0; --i) BUCKET_INS(i);
  for (k = 0; k < N; ++k) {
    while (mindeg <= N && head[mindeg] < 0) ++mindeg;
    I p = head[mindeg];
    BUCKET_DEL(p);
    elim[p] = 1; perm[k] = p;
    /* new element p: union of variable neighbours and the variables of adjacent elements */
    ++stamp; mark[p] = stamp;
    IVec Lp = {0, 0, 0};
    for (I q = 0; q < adjv[p].len; ++q) { I x = adjv[p].v[q]; if (!elim[x] && mark[x] != stamp) { mark[x] = stamp; iv_push(&Lp, x); } }
    for (I q = 0; q < adje[p].len; ++q) {
      I e = adje[p].v[q]; if (dead_e[e]) continue;
      for (I t = 0; t < elem[e].len; ++t) { I x = elem[e].v[t]; if (!elim[x] && mark[x] != stamp) { mark[x] = stamp; iv_push(&Lp, x); } }
      dead_e[e] = 1; free(elem[e].v); elem[e].v = 0; elem[e].len = elem[e].cap = 0; /* absorbed */
    }
    free(adjv[p].v); adjv[p].v = 0; adjv[p].len = adjv[p].cap = 0;
    free(adje[p].v); adje[p].v = 0; adje[p].len = adje[p].cap = 0;
    elem[p] = Lp;
    /* update every variable of the new element */
    for (I q = 0; q < Lp.len; ++q) {
      I x = Lp.v[q];
      /* prune variable list: drop eliminated nodes and members of Lp (now reached through element p) */
      I w_ = 0;
      for (I t = 0; t < adjv[x].len; ++t) { I y = adjv[x].v[t]; if (!elim[y] && mark[y] != stamp) adjv[x].v[w_++] = y; }
      adjv[x].len = w_;
      w_ = 0;
      for (I t = 0; t < adje[x].len; ++t) { I e = adje[x].v[t]; if (!dead_e[e]) adje[x].v[w_++] = e; }
      adje[x].len = w_;
      iv_push(&adje[x], p);
    }
    for (I q = 0; q < Lp.len; ++q) { /* exact external degree */
      I x = Lp.v[q];
      BUCKET_DEL(x);
      I st2 = ++stamp; mark[x] = st2; I d = 0;
      for (I t = 0; t < adjv[x].len; ++t) { I y = adjv[x].v[t]; if (mark[y] != st2) { mark[y] = st2; ++d; } }
      for (I t = 0; t < adje[x].len; ++t) {
        I e = adje[x].v[t];
        for (I s = 0; s < elem[e].len; ++s) { I y = elem[e].v[s]; if (!elim[y] && mark[y] != st2) { mark[y] = st2; ++d; } }
      }
      deg[x] = d;
      BUCKET_INS(x);
      if (d < mindeg) mindeg = d;
    }
    /* re-stamp members of Lp so later "mark[y] != stamp" tests in this step stay valid: not needed past here */
  }
#undef BUCKET_INS
#undef BUCKET_DEL
  for (i = 0; i < N; ++i) { free(adjv[i].v); free(adje[i].v); free(elem[i].v); }
  free(adjv); free(adje); free(elem); free(deg); free(mark); free(elim); free(dead_e); free(head); free(nxt); free(prv);
}


/* LDL' = P K P'.  K: upper triangle (row <= col) by columns, N x N.  Outputs malloc'ed: P (N), Lp (N+1), Li/Lx
 * (strictly lower L by columns), D (N).  Returns 0, or -1 on a zero pivot. */
static int orc_ldl_factor(I N, const I *Kp, const I *Ki, const F *Kx, I **P_out, I **Lp_out, I **Li_out, F **Lx_out, F **D_out) {
  I i, j, q;
  const I kk = Kp[N];
  I *Gp = (I *)calloc((size_t)N + 1, sizeof(I));
  for (j = 0; j < N; ++j) for (q = Kp[j]; q < Kp[j + 1]; ++q) if (Ki[q] != j) { Gp[Ki[q] + 1]++; Gp[j + 1]++; }
  for (i = 0; i < N; ++i) Gp[i + 1] += Gp[i];
  I *Gi = (I *)malloc(sizeof(I) * (Gp[N] > 0 ? Gp[N] : 1)), *pos = (I *)malloc(sizeof(I) * N);
  for (i = 0; i < N; ++i) pos[i] = Gp[i];
  for (j = 0; j < N; ++j) for (q = Kp[j]; q < Kp[j + 1]; ++q) if (Ki[q] != j) { Gi[pos[Ki[q]]++] = j; Gi[pos[j]++] = Ki[q]; }
  I *P = (I *)malloc(sizeof(I) * N);
  min_degree_order(N, Gp, Gi, P);
  free(Gp); free(Gi); free(pos);
  I *Pinv = (I *)malloc(sizeof(I) * N);
  for (i = 0; i < N; ++i) Pinv[P[i]] = i;
  I *Cp = (I *)calloc((size_t)N + 1, sizeof(I));
  for (j = 0; j < N; ++j) for (q = Kp[j]; q < Kp[j + 1]; ++q) { I a = Pinv[Ki[q]], b = Pinv[j]; Cp[(a > b ? a : b) + 1]++; }
  for (i = 0; i < N; ++i) Cp[i + 1] += Cp[i];
  I *Ci = (I *)malloc(sizeof(I) * (kk > 0 ? kk : 1)); F *Cx = (F *)malloc(sizeof(F) * (kk > 0 ? kk : 1)); I *cpos = (I *)malloc(sizeof(I) * N);
  for (i = 0; i < N; ++i) cpos[i] = Cp[i];
  for (j = 0; j < N; ++j) for (q = Kp[j]; q < Kp[j + 1]; ++q) {
    I a = Pinv[Ki[q]], b = Pinv[j]; I r_ = a < b ? a : b, c_ = a < b ? b : a;
    Ci[cpos[c_]] = r_; Cx[cpos[c_]] = Kx[q]; cpos[c_]++;
  }
  free(cpos); free(Pinv);
  I *parent = (I *)malloc(sizeof(I) * N), *anc = (I *)malloc(sizeof(I) * N), *flag = (I *)malloc(sizeof(I) * N), *lnz = (I *)calloc((size_t)N, sizeof(I));
  for (j = 0; j < N; ++j) {
    parent[j] = -1; anc[j] = -1;
    for (q = Cp[j]; q < Cp[j + 1]; ++q) {
      I r_ = Ci[q];
      while (r_ != -1 && r_ < j) { I nx = anc[r_]; anc[r_] = j; if (nx == -1) parent[r_] = j; r_ = nx; }
    }
  }
  for (j = 0; j < N; ++j) {
    flag[j] = j;
    for (q = Cp[j]; q < Cp[j + 1]; ++q) { I r_ = Ci[q]; while (r_ < j && flag[r_] != j) { lnz[r_]++; flag[r_] = j; r_ = parent[r_]; } }
  }
  I *Lp = (I *)malloc(sizeof(I) * (N + 1));
  Lp[0] = 0;
  for (j = 0; j < N; ++j) Lp[j + 1] = Lp[j] + lnz[j];
  I Lnnz = Lp[N];
  I *Li = (I *)malloc(sizeof(I) * (Lnnz > 0 ? Lnnz : 1)); F *Lx = (F *)malloc(sizeof(F) * (Lnnz > 0 ? Lnnz : 1));
  F *Dg = (F *)malloc(sizeof(F) * N);
  F *Y = (F *)calloc((size_t)N, sizeof(F)); I *stack = (I *)malloc(sizeof(I) * N), *pat = (I *)malloc(sizeof(I) * N), *fill = (I *)calloc((size_t)N, sizeof(I));
  int ok = 1;
  for (I k = 0; k < N && ok; ++k) {
    I top = N; flag[k] = k; F dk = 0.0;
    for (q = Cp[k]; q < Cp[k + 1]; ++q) {
      I r_ = Ci[q];
      if (r_ == k) { dk += Cx[q]; continue; }
      Y[r_] += Cx[q];
      I len = 0;
      while (flag[r_] != k) { pat[len++] = r_; flag[r_] = k; r_ = parent[r_]; }
      while (len > 0) stack[--top] = pat[--len];
    }
    for (; top < N; ++top) {
      I c_ = stack[top]; F yc = Y[c_]; Y[c_] = 0.0;
      I e_ = Lp[c_] + fill[c_];
      for (q = Lp[c_]; q < e_; ++q) Y[Li[q]] -= Lx[q] * yc;
      F lkc = yc / Dg[c_];
      dk -= lkc * yc;
      Li[e_] = k; Lx[e_] = lkc; fill[c_]++;
    }
    Dg[k] = dk;
    if (dk == 0.0) ok = 0;
  }
  free(Y); free(stack); free(pat); free(fill); free(parent); free(anc); free(flag); free(lnz); free(Cp); free(Ci); free(Cx);
  *P_out = P; *Lp_out = Lp; *Li_out = Li; *Lx_out = Lx; *D_out = Dg;
  return ok ? 0 : -1;
}
/* x = P' L^-T D^-1 L^-1 P b, in place on b; bp is scratch (N) */
static void orc_ldl_solve(I N, const I *P, const I *Lp, const I *Li, const F *Lx, const F *Dg, F *b, F *bp) {
  I j, q;
  for (j = 0; j < N; ++j) bp[j] = b[P[j]];
  for (j = 0; j < N; ++j) for (q = Lp[j]; q < Lp[j + 1]; ++q) bp[Li[q]] -= Lx[q] * bp[j];
  for (j = 0; j < N; ++j) bp[j] /= Dg[j];
  for (j = N - 1; j >= 0; --j) for (q = Lp[j]; q < Lp[j + 1]; ++q) bp[j] -= Lx[q] * bp[Li[q]];
  for (j = 0; j < N; ++j) b[P[j]] = bp[j];
}
#undef I
#undef F
#endif
