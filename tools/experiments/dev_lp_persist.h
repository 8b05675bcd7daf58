// EXPERIMENT, NOT PART OF THE LIBRARY (moved out of abip_amd/csrc after measurement; last built in commit 915fa21, where solver.hip
// still carries the host side `enqueue_persist`).  Result on MI355X: correct and bit-identical between batched / stepwise / strided use
// (commit 915fa21's tests/test_gpu_parity.py::test_persistent_iteration_kernel), but ~100 us per iteration on the C2 surrogate against 58 us
// for the launch-per-kernel path (profiles/r02d_*), and the multi-workgroup form became unreliable once the kernel's scratch use grew
// (960 B per lane after the Csr struct gained the SELL fields: the grid barriers need every workgroup resident, which a scratch-limited
// dispatch does not guarantee -- it deadlocked).  Kept as the record of what was tried and as a starting point for an all-LDS variant.
//
// dev_lp_persist.h -- Netlib-class LPs on the direct back-end: a whole batch of inner ADMM iterations in ONE launch.
//
// When the sparse part of the LDL' solve fits one workgroup (DevLdl::small; every Netlib LP does), an iteration of the launch-per-kernel
// path is 6-7 launches of a few microseconds each: launch latency, not work.  Here workgroup 0 (1024 threads, the solve vector in LDS)
// runs the complete iteration -- rhs build, permutation, forward levels, [dense tail], backward levels, u_t'h, barrier prox / dual update /
// averages, both residual products of the stopping test, the exit test -- and loops over the iterations of the batch without leaving the
// kernel (reference loop: src/abip-lp/src/abip.c:2131-2215; solve: linsys/direct.c:172-198, external/ldl/ldl.c:357-550).  Nothing in
// that chain needs another workgroup except the two dense triangular mat-vecs of the tail (dev_ldl.h: T x T doubles each, megabytes):
// for those, G - 1 helper workgroups of the same launch stream W / W' beside workgroup 0, with three grid barriers per iteration
// (w ready -> t complete -> x2 complete).  The hand-offs carry only the T-vectors; they use agent-scope (sc1) loads and stores and a
// monotonic arrival counter, no cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility).  With no tail the launch is one
// workgroup and has no grid synchronisation at all.
//
// One kernel serves the batched and the stepwise (one iteration per launch: final_check, restart iterations, half_update, ABIP_HIP_BATCH=0)
// modes, so the two are bit-identical by construction.  All reductions are taken over the one workgroup in a fixed order; the persistent
// grid of the other kernels of such a solve is NB = 1 so that every partial-sum slot has exactly one entry.
#pragma once
#include "dev_kernels.h"
#include "dev_ldl.h"

namespace abip {

struct PersistArgs {
  UpdArgs upd;            // vectors and scalars of k_admm_update (dom, fuse_avg, avg_stats are set per iteration by the kernel)
  const double *h;
  double g_th;
  Dims d;
  Csr A, At;
  const double *wD, *wE;  // D_i / (sc_b scale), E_j / (sc_c scale), or null without normalisation
  double *part;
  Ctl *ctl;
  // factor: sparse head in one workgroup, dense tail
  Tri F, B;
  const int *Pmap;
  const double *D;
  double *xg;             // N doubles of global scratch (the solve vector when it does not fit LDS; always the tail's hand-off buffer)
  int t0, N, T;
  const double *W, *Wt;
  double *ttmp;           // T doubles: t = D2^-1 W w
  // batch
  int nb;                 // iterations to run at most (the exit test may end the batch earlier)
  long j0;                // inner-iteration index of the first one: dom = j + 1, averaged statistics when (j + 1) % 10 == 0
  int restart;            // nb == 1 only: this iteration restarts from the running mean (abip.c:608-627)
  double restart_fre;
  int LV;
  FinArgs fin[2];         // the finalize that closes an iteration: [0] plain, [1] with the averaged iterate's statistics (every 10th)
  double *rp;             // [12][128]: per-workgroup partial sums of the residual products (pass x matrix x 3 sums), G > 1 only
  unsigned long long *dbg; // optional (ABIP_HIP_PERSIST_DEBUG): wall-clock ticks of workgroup 0 per phase, summed over iterations
  unsigned *sync;         // [0] arrival counter of the grid barrier, [1] go flag of the current iteration (zeroed by the host per launch)
};

// ---- agent-scope accesses for the few vectors that cross workgroups ---------------------------------------------------------------
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every thread of every workgroup calls it; *epoch counts this workgroup's arrivals
__device__ __forceinline__ void grid_barrier(unsigned *cnt, unsigned G, unsigned &epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's (write-through) stores have left the CU
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = G * epoch;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// sum NS values over the 1024-thread workgroup, fixed order, result to every thread
template <int NS>
__device__ __forceinline__ void wg_sum(double (&v)[NS], double *red /* NS * TBS/64 */) {
  constexpr int NWV = TBS / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[s] += __shfl_xor(v[s], off, 64);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) red[s * NWV + wave] = v[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double t = red[s * NWV];
#pragma unroll
    for (int wv = 1; wv < NWV; ++wv) t += red[s * NWV + wv];
    v[s] = t;
  }
}

// CSR-stream product of one workgroup over ALL row blocks of M (the row blocks of dev_common.h: <= CHUNK non-zeros and rows, or one
// longer row): one non-zero per thread, products staged in LDS, `lpr` lanes reduce a row.  prod(col, a) -> product, rowf(row, sum).
template <class ProdF, class RowF>
__device__ __forceinline__ void spmv_wg(const Csr &M, double *lds /* CHUNK */, int *lptr /* CHUNK + 1 */, double *red, ProdF prod, RowF rowf, int b0 = 0, int bstride = 1) {
  const int tid = threadIdx.x;
  for (int b = b0; b < M.nrb; b += bstride) {
    const int4 d = M.rbd[b];
    const int r0 = d.x, k0 = d.z, k1 = d.w, nn = k1 - k0, R = d.y - d.x;
    if (nn <= CHUNK) {
      for (int t = tid; t <= R; t += TBS) lptr[t] = M.ptr[r0 + t] - k0;
      if (tid < nn) lds[tid] = prod(M.idx[k0 + tid], M.val[k0 + tid]);
      __syncthreads();
      int lpr = pow2_floor(TBS / (R > 0 ? R : 1));
      if (lpr > 64) lpr = 64;
      const int ngrp = TBS / lpr, grp = tid / lpr, q = tid % lpr;
      for (int base = 0; base < R; base += ngrp) {
        const int r = base + grp;
        double acc = 0.0;
        if (r < R) {
          const int s = lptr[r], e = lptr[r + 1];
          for (int k = s + q; k < e; k += lpr) acc += lds[k];
        }
        for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (r < R && q == 0) rowf(r0 + r, acc);
      }
      __syncthreads();
    } else { // one long row
      double acc[1] = {0.0};
      for (int k = k0 + tid; k < k1; k += TBS) acc[0] += prod(M.idx[k], M.val[k]);
      wg_sum<1>(acc, red);
      if (tid == 0) rowf(r0, acc[0]);
      __syncthreads();
    }
  }
}

// rows [vb-th share] of the dense triangular mat-vec of dev_ldl.h (k_tail_mv) with the vector operands crossing workgroups:
// out[r] = (sum_c M[r, c] v[c]) / (dsc ? dsc[r] : 1); one wavefront per row, rows dealt round-robin over all waves of the grid
// (v: the operand in THIS workgroup's LDS -- staged once per mat-vec with agent-scope loads, then read T/2 times per row from LDS)
__device__ __attribute__((noinline)) void tail_mv_wg(const double *__restrict__ M, int T, int upper, const double *v, double *out, const double *__restrict__ dsc, int vb, int vgrid) {
  const int lane = threadIdx.x & 63, wave = vb * (TBS / 64) + (threadIdx.x >> 6), nw = vgrid * (TBS / 64);
  for (int r = wave; r < T; r += nw) {
    const int lo = upper ? r : 0, hi = upper ? T : r + 1;
    const double2 *row2 = reinterpret_cast<const double2 *>(M + (long)r * T);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = (lo & ~127) + 2 * lane;
    auto mac = [&](int cc, double &acc) {
      const double2 m = row2[cc >> 1];
      if (cc >= lo && cc < hi) acc += m.x * v[cc];
      if (cc + 1 >= lo && cc + 1 < hi) acc += m.y * v[cc + 1];
    };
    auto mac_in = [&](int cc, double &acc) { const double2 m = row2[cc >> 1]; acc += m.x * v[cc]; acc += m.y * v[cc + 1]; };
    if (c < hi) mac(c, a0);
    c += 128;
    for (; c + 3 * 128 + 1 < hi; c += 512) { mac_in(c, a0); mac_in(c + 128, a1); mac_in(c + 256, a2); mac_in(c + 384, a3); }
    for (; c < hi; c += 128) mac(c, a0);
    double s = (a0 + a1) + (a2 + a3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) st_agent(out + r, dsc ? s / dsc[r] : s);
  }
}

// the residual products of one iterate (abip.c:1976-1992, 407-413, 443-449) over row blocks b0, b0 + bstride, ... of A and A':
// six sums -> out6 (every thread).  us: the l-vector uu in LDS, ss: the x block of vv in LDS.
struct ResidIn { Csr A, At; const double *b, *c, *wD, *wE; Dims d; };
__device__ __attribute__((noinline)) void resid_products(const ResidIn a, const double *us, const double *ss, double *s_prod, int *s_lptr, double *s_red,
                                               int b0, int bstride, double (&out6)[6]) {
  const Dims d = a.d;
  const double tau = us[d.MP + d.n];
  const double *xv = us + d.MP;
  double accA[3] = {0.0, 0.0, 0.0}, accT[3] = {0.0, 0.0, 0.0};
  spmv_wg(a.A, s_prod, s_lptr, s_red, [&](int c, double av) { return av * xv[c]; },
          [&](int i, double pri) {
            const double e = pri - a.b[i] * tau;
            double sc = a.wD ? a.wD[i] : 1.0;
            sc = sc * sc;
            accA[0] += e * e; accA[1] += (e * e) * sc; accA[2] += (pri * pri) * sc;
          }, b0, bstride);
  spmv_wg(a.At, s_prod, s_lptr, s_red, [&](int c, double av) { return av * us[c]; },
          [&](int jj, double aty) {
            const double drj = aty + ss[jj], e = drj - a.c[jj] * tau;
            double sc = a.wE ? a.wE[jj] : 1.0;
            sc = sc * sc;
            accT[0] += e * e; accT[1] += (e * e) * sc; accT[2] += (drj * drj) * sc;
          }, b0, bstride);
  out6[0] = accA[0]; out6[1] = accA[1]; out6[2] = accA[2]; out6[3] = accT[0]; out6[4] = accT[1]; out6[5] = accT[2];
  wg_sum<6>(out6, s_red);
}
// stage uu (LV entries used: y block, x block, tau) and the x block of vv into LDS with agent-scope loads
__device__ __attribute__((noinline)) void stage_iterate(const Dims d, const double *uu, const double *vv, double *us, double *ss) {
  const int tid = threadIdx.x;
  for (int i = tid; i < d.m; i += TBS) us[i] = ld_agent(uu + i);
  for (int jj = tid; jj <= d.n; jj += TBS) us[d.MP + jj] = ld_agent(uu + d.MP + jj);
  for (int jj = tid; jj < d.n; jj += TBS) ss[jj] = ld_agent(vv + d.MP + jj);
  __syncthreads();
}

// dynamic LDS: xs[max(N, T)] (workgroup 0: the solve vector; helpers: the mat-vec operand) | us[LV] | ss[n]
template <bool XL>
static __global__ __launch_bounds__(TBS) void k_lp_persist(PersistArgs a) {
  extern __shared__ double x_lds[];
  __shared__ int s_lp[MAXLEV_LDS + 1], s_lg[MAXLEV_LDS];
  __shared__ double s_prod[CHUNK];
  __shared__ int s_lptr[CHUNK + 1];
  __shared__ double s_red[9 * (TBS / 64)];
  __shared__ int s_go;
  const int tid = threadIdx.x;
  const bool lead = blockIdx.x == 0;
  const unsigned G = gridDim.x;
  const int g = (int)blockIdx.x;
  const bool multi = G > 1;
  const bool tail = a.T > 0;
  unsigned epoch = 0;
  const Dims d = a.d;
  const int q_tau = d.MP + d.n;
  double *x = x_lds;
  double *us = x_lds + (a.N > a.T ? a.N : a.T);
  double *ss = us + a.LV;
  double *tvec = a.xg + a.t0; // the tail part of the solve vector in global memory: w on the way in, x2 on the way out
  const ResidIn rin{a.A, a.At, a.upd.b, a.upd.c, a.wD, a.wE, a.d};
  auto stage_vec = [&](double *dst, const double *src, int len) { for (int k = tid; k < len; k += TBS) dst[k] = ld_agent(src + k); __syncthreads(); };
  // the residual products of this iteration's iterate(s), this workgroup's share of the row blocks -> rp (helpers and workgroup 0 alike)
  auto products_shared = [&](int avg_stats) {
    for (int pass = 0; pass <= avg_stats; ++pass) {
      stage_iterate(d, pass ? a.upd.u_avgc : a.upd.u, pass ? a.upd.v_avgc : a.upd.v, us, ss);
      double o6[6];
      resid_products(rin, us, ss, s_prod, s_lptr, s_red, g, (int)G, o6);
      if (tid < 6) st_agent(a.rp + (size_t)(pass * 6 + tid) * 128 + g, o6[tid]);
      __syncthreads();
    }
  };

  if (!lead) { // helper workgroups: their share of the dense mat-vecs and of the residual products of every iteration
    for (long j = a.j0;; ++j) {
      grid_barrier(a.sync, G, epoch);                                                                            // S1: w is ready (or the batch is over)
      if (!__hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
      if (tail) {
        stage_vec(x, tvec, a.T);
        tail_mv_wg(a.W, a.T, 0, x, a.ttmp, a.D + a.t0, g, (int)G);
        grid_barrier(a.sync, G, epoch);                                                                          // S2: t complete
        stage_vec(x, a.ttmp, a.T);
        tail_mv_wg(a.Wt, a.T, 1, x, tvec, nullptr, g, (int)G);
        grid_barrier(a.sync, G, epoch);                                                                          // S3: x2 complete
      }
      grid_barrier(a.sync, G, epoch);                                                                            // S4: (u, v) of this iteration are out
      products_shared(((j + 1) % 10 == 0) ? 1 : 0);
      grid_barrier(a.sync, G, epoch);                                                                            // S5: partial sums are out
    }
  }

  unsigned long long tk = a.dbg ? wall_clock64() : 0;
#define PH(i) do { if (a.dbg && tid == 0) { const unsigned long long n_ = wall_clock64(); a.dbg[i] += n_ - tk; tk = n_; } } while (0)
  for (int q = 0;; ++q) {
    // ---- loop control: the exit test of the previous iteration (d_finalize below) may have raised halt
    if (tid == 0) {
      s_go = (q < a.nb && !a.ctl->halt) ? 1 : 0;
      if (multi) __hip_atomic_store(a.sync + 1, (unsigned)s_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int go = s_go;
    if (!go) {
      if (multi) grid_barrier(a.sync, G, epoch); // release the helpers (they read go = 0 behind S1)
      return;
    }
    const long j = a.j0 + q;
    const int avg_stats = ((j + 1) % 10 == 0) ? 1 : 0; // abip.c:2000
    double *__restrict__ ut = a.upd.ut;

    // ---- project_lin_sys prologue (abip.c:552-558): u_t <- rhs of the KKT system
    {
      const double *__restrict__ pu = a.upd.u, *__restrict__ pv = a.upd.v, *__restrict__ ph = a.h;
      const double wg0 = a.part[S_WG * MAXNB];
      const double tsum = pu[q_tau] + pv[q_tau];
      const double coef = (wg0 - tsum * a.g_th) / (a.g_th + 1.0);
      for (int i = tid; i < d.m; i += TBS) {
        double t = (pu[i] + pv[i]) * a.upd.rho;
        t += -tsum * ph[i];
        t += -coef * ph[i];
        ut[i] = t;
      }
      for (int jj = tid; jj < d.n; jj += TBS) {
        double t = pu[d.MP + jj] + pv[d.MP + jj];
        t += -tsum * ph[d.MP + jj];
        t += -coef * ph[d.MP + jj];
        ut[d.MP + jj] = -t;
      }
      if (tid == 0) ut[q_tau] = tsum;
      __syncthreads();
    }
    PH(0);
    // ---- _ldl_solve (direct.c:172-198): x = P b, forward levels, [dense tail], D^-1, backward levels, b = P' x
    for (int k = tid; k < a.N; k += TBS) x[k] = ut[a.Pmap[k]];
    __syncthreads();
    run_levels(a.F, x, s_lp, s_lg, tid, 0, a.F.nlev);
    PH(1);
    if (tail) {
      for (int k = a.t0 + tid; k < a.N; k += TBS) st_agent(a.xg + k, x[k]);
      grid_barrier(a.sync, G, epoch);                                                                            // S1
      tail_mv_wg(a.W, a.T, 0, x + a.t0, a.ttmp, a.D + a.t0, 0, (int)G);
      grid_barrier(a.sync, G, epoch);                                                                            // S2
      stage_vec(x + a.t0, a.ttmp, a.T);
      tail_mv_wg(a.Wt, a.T, 1, x + a.t0, tvec, nullptr, 0, (int)G);
      grid_barrier(a.sync, G, epoch);                                                                            // S3
      stage_vec(x + a.t0, tvec, a.T);
    } else if (multi) grid_barrier(a.sync, G, epoch);                                                            // S1 (go flag only)
    PH(2);
    for (int k = tid; k < a.t0; k += TBS) x[k] /= a.D[k];
    __syncthreads();
    run_levels(a.B, x, s_lp, s_lg, tid, 0, a.B.nlev);
    for (int k = tid; k < a.N; k += TBS) ut[a.Pmap[k]] = x[k];
    __syncthreads();
    PH(3);
    // ---- u_t[0:l-1)'h (abip.c:560)
    double dh;
    {
      const double *__restrict__ ph = a.h;
      double s1[1] = {0.0};
      for (int i = tid; i < d.m; i += TBS) s1[0] += ut[i] * ph[i];
      for (int jj = tid; jj < d.n; jj += TBS) s1[0] += ut[d.MP + jj] * ph[d.MP + jj];
      wg_sum<1>(s1, s_red);
      dh = s1[0];
      if (tid == 0) a.part[S_DH * MAXNB] = dh;
    }
    PH(4);
    // ---- barrier prox, dual update, restart sums, running averages and their statistics (k_admm_update)
    UpdArgs ua = a.upd;
    ua.dom = (double)(j + 1); ua.fuse_avg = a.restart ? 0 : 1; ua.avg_stats = avg_stats;
    Stat st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    { // (the element loops load everything an element needs before they store: the arrays are distinct, which only __restrict__ can tell the compiler)
      double *__restrict__ pu = ua.u, *__restrict__ pv = ua.v, *__restrict__ pua = ua.u_avg, *__restrict__ pva = ua.v_avg;
      double *__restrict__ pus = ua.u_sum, *__restrict__ pvs = ua.v_sum, *__restrict__ puc = ua.u_avgc, *__restrict__ pvc = ua.v_avgc;
      const double *__restrict__ pg = ua.g, *__restrict__ pb = ua.b, *__restrict__ pc = ua.c;
      for (int i = tid; i < d.m; i += TBS) {
        const double uti = ut[i], vo = pv[i], uo = pu[i], a1 = pua[i], a2 = pva[i], s1 = pus[i], s2 = pvs[i], gi = pg[i], bi = pb[i];
        double un, vn;
        if (!ua.half_update) { vn = vo; un = uti - vn; }
        else { double vh = vo + 0.5 * (uo - uti); un = uti - vh; vn = vh + (un - uti); }
        st_agent(pu + i, un); st_agent(pv + i, vn); // (write-through: the other workgroups read the iterate for the residual products)
        pua[i] = a1 + un; pva[i] = a2 + vn;
        if (ua.fuse_avg) { // avg_and_stats_y
          const double us_ = s1 + un, vs_ = s2 + vn;
          pus[i] = us_; pvs[i] = vs_;
          const double uav = us_ / ua.dom, vav = vs_ / ua.dom;
          st_agent(puc + i, uav); st_agent(pvc + i, vav);
          st.wg += ua.rho * (un + vn) * gi;
          st.nu += un * un; st.nv += vn * vn; st.by += bi * un;
          if (ua.avg_stats) { st.nua += uav * uav; st.nva += vav * vav; st.bya += bi * uav; }
        }
      }
      for (int jj = tid; jj < d.n; jj += TBS) {
        const int qq = d.MP + jj;
        const double utq = ut[qq], vo = pv[qq], uo = pu[qq], a1 = pua[qq], a2 = pva[qq], s1 = pus[qq], s2 = pvs[qq], gq = pg[qq], cj = pc[jj];
        double un, vn;
        if (!ua.half_update) {
          const double t = ua.alpha * utq + (1.0 - ua.alpha) * uo - vo;
          const double hlf = t / 2;
          un = hlf + sqrt(hlf * hlf + ua.mu_over_beta);
          vn = vo + (un - ua.alpha * utq - (1.0 - ua.alpha) * uo);
        } else {
          double vh = vo + 0.5 * (uo - utq);
          const double hlf = (utq - vh) / 2;
          un = hlf + sqrt(hlf * hlf + ua.mu_over_beta);
          vn = vh + (un - utq);
        }
        st_agent(pu + qq, un); st_agent(pv + qq, vn);
        pua[qq] = a1 + un; pva[qq] = a2 + vn;
        if (ua.fuse_avg) { // avg_and_stats_x
          const double us_ = s1 + un, vs_ = s2 + vn;
          pus[qq] = us_; pvs[qq] = vs_;
          const double uav = us_ / ua.dom, vav = vs_ / ua.dom;
          st_agent(puc + qq, uav); st_agent(pvc + qq, vav);
          st.nu += ua.xw * (un * un); st.nv += ua.xw * (vn * vn);
          if (ua.avg_stats) { st.nua += ua.xw * (uav * uav); st.nva += ua.xw * (vav * vav); }
          st.wg += ua.xw * ((un + vn) * gq);
          st.cx += ua.xw * (cj * un);
          if (ua.avg_stats) st.cxa += ua.xw * (cj * uav);
        }
      }
    }
    if (tid == 0) { // the tau / kappa entry
      const double utq = ut[q_tau] + dh;
      ut[q_tau] = utq;
      double un, vn;
      prox_x(ua, q_tau, utq, un, vn);
      st_agent(ua.u + q_tau, un); st_agent(ua.v + q_tau, vn);
      ua.u_avg[q_tau] += un; ua.v_avg[q_tau] += vn;
      if (ua.fuse_avg) { avg_and_stats_x(ua, q_tau, d.n, true, un, vn, st); st_agent(ua.u_avgc + q_tau, ua.u_avgc[q_tau]); st_agent(ua.v_avgc + q_tau, ua.v_avgc[q_tau]); }
    }
    __syncthreads();
    if (a.restart) { // abip.c:613-627, then compute_avg on the restarted iterate
      for (int i = tid; i < a.LV; i += TBS) {
        const double ra = ua.u_avg[i] / a.restart_fre, rb = ua.v_avg[i] / a.restart_fre;
        ua.u[i] = ra; ua.v[i] = rb; ua.u_avg[i] = 0.0; ua.v_avg[i] = 0.0;
      }
      __syncthreads();
      ua.fuse_avg = 1;
      for (int i = tid; i < d.m; i += TBS) avg_and_stats_y(ua, i, ua.u[i], ua.v[i], st);
      for (int jj = tid; jj < d.n; jj += TBS) avg_and_stats_x(ua, d.MP + jj, jj, false, ua.u[d.MP + jj], ua.v[d.MP + jj], st);
      if (tid == 0) avg_and_stats_x(ua, q_tau, d.n, true, ua.u[q_tau], ua.v[q_tau], st);
      if (multi) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // (rare path, plain stores: push them out for the other workgroups)
    }
    {
      double vals[9] = {st.wg, st.nu, st.nv, st.cx, st.by, st.nua, st.nva, st.cxa, st.bya};
      wg_sum<9>(vals, s_red);
      if (tid == 0) {
        const int ws[9] = {S_WG, S_NU, S_NV, S_CX, S_BY, S_NUA, S_NVA, S_CXA, S_BYA};
        for (int s = 0; s < 9; ++s) a.part[ws[s] * MAXNB] = vals[s];
      }
    }
    __syncthreads();
    PH(5);
    // ---- iterate_Q_norm_resd + calc_residuals sums (k_q_A / k_q_At): current iterate, and the averaged one every 10th iteration
    if (multi) {
      grid_barrier(a.sync, G, epoch);                                                                            // S4: the iterate (written through above) is out
      products_shared(avg_stats);
      grid_barrier(a.sync, G, epoch);                                                                            // S5
      const int nsum = 6 * (avg_stats + 1);
      for (int e = tid; e < nsum * (int)G; e += TBS) s_prod[e] = ld_agent(a.rp + (size_t)(e / (int)G) * 128 + e % (int)G); // all loads in flight at once
      __syncthreads();
      if (tid < nsum) { // each sum over the workgroups in a fixed order
        double t = 0.0;
        for (unsigned gg = 0; gg < G; ++gg) t += s_prod[tid * (int)G + gg];
        const int pass = tid / 6, k6 = tid % 6;
        const int slot = (k6 < 3 ? (pass ? S_QPA : S_QP) : (pass ? S_QDA : S_QD)) + k6 % 3;
        a.part[slot * MAXNB] = t;
      }
    } else {
      for (int pass = 0; pass <= avg_stats; ++pass) {
        const double *uu = pass ? ua.u_avgc : ua.u, *vv = pass ? ua.v_avgc : ua.v;
        double o6[6];
        resid_products(rin, uu, vv + d.MP, s_prod, s_lptr, s_red, 0, 1, o6); // one workgroup: gather straight from global memory (own stores)
        if (tid < 6) a.part[((tid < 3 ? (pass ? S_QPA : S_QP) : (pass ? S_QDA : S_QD)) + tid % 3) * MAXNB] = o6[tid];
      }
    }
    __syncthreads();
    PH(6);
    // ---- partials -> ctl->out, the inner-loop exit test (abip.c:2027-2050, 2173)
    d_finalize(a.fin[avg_stats], d, a.part, 1, a.ctl);
    __syncthreads();
    PH(7);
  }
}

#undef PH
} // namespace abip
