// qcp_dist.h -- the conic path over several GPUs: COLUMN blocks (one process per GPU, dist_internal.h).
//
//   A = [A_1 ... A_G] by columns, cut at cone boundaries (a cone lives on one rank; free / zero / orthant entries split anywhere).
//   n-space (x, s, c, E, the cones, H^-1, the PCG's t_n) is SHARDED: rank g owns the columns [n0_g, n1_g);
//   m-space (y, b, D, every PCG vector) and tau, kappa, mu are REPLICATED and stay bit-identical on all ranks.
//   A'y is local (A_g'y).  A x = sum_g A_g x_g is the one exchange: an all-reduce of m doubles -- per PCG iteration exactly one, and none
//   for the PCG's scalars (p'Gp, |r|^2, z'r are sums over the replicated m-space: every rank computes the same numbers).
//   Sums that run over n-space (r'mu, the inner test's norms, the residual sums) are all-reduced as one packed table per use; sums over the
//   replicated m-space enter them with weight QDims::wy (1 on rank 0, 0 elsewhere) so that the reduced total counts them once.
//   Maxima (the residuals' inf-norms, the PCG tolerance's |u + tau r|_inf) travel as one slot per rank in a sum all-reduce.
// Why columns and not the LP path's rows: the conic workloads have n >> m (C5: n = 100 002, m = 10 001), so the exchanged vector is the short
// one (80 KB instead of 800 KB), the cone step needs no exchange at all, and the PCG's scalar reductions vanish.
//
// This header holds the kernels that differ from the single-GPU path: products written to the exchange buffer, and the element-wise
// halves that run after the all-reduce.  The rest of the iteration runs the single-GPU kernels on the rank's block.
#pragma once
#include "qcp_pcg.h"

namespace abip {

// out[i] = sum_j A[i, j] x[j] (sc ? sc[j] : 1) over the rank's columns; gated like the kernel it replaces
__global__ __launch_bounds__(BS, 8) void kq_prod_A(Csr A, const double *__restrict__ x, const double *__restrict__ sc, double *__restrict__ out, int gate_cg, const Ctl *hc) {
  if (hc->halt || (gate_cg && hc->cg_done)) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  if (sc) spmv_stream<1>(A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * (x[c] * sc[c]); }, [&](int i, double(&acc)[1]) { out[i] = acc[0]; });
  else spmv_stream<1>(A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; }, [&](int i, double(&acc)[1]) { out[i] = acc[0]; });
}

// kq_pcg_prep without its product: warm start y0 = (u + tau r)_y and the rank's share of |(u + tau r)[0:n]|_inf (abip.c:208-215: all of y, and
// the x entries whose GLOBAL index is below n - m)
__global__ __launch_bounds__(BS) void kq_dist_prep_warm(const double *__restrict__ u, const double *__restrict__ rv, int warm, QDims d, int n0, int xlim, QPcgVec v, double *ppart, Ctl *hc) {
  if (hc->halt) return;
  __shared__ double sm[WAVES];
  if (blockIdx.x == 0 && threadIdx.x == 0) { hc->cg_it = 0; hc->cg_done = 0; }
  double wm[1] = {0.0};
  if (warm) {
    const double tau = u[d.MP + d.n];
    const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
    for (int i = t0; i < d.m; i += stride) { const double t = u[i] + tau * rv[i]; v.y0[i] = t; wm[0] = fmax(wm[0], fabs(t)); }
    for (int j = t0; j < d.n && n0 + j < xlim; j += stride) { const int q = d.MP + j; wm[0] = fmax(wm[0], fabs(u[q] + tau * rv[q])); }
  }
  const int ws[1] = {PQ_WM};
  write_partials_max<1>(ppart, ws, wm, sm);
}
// one workgroup: the rank's maximum of `nslot` max-slots of a partials table into its own lane of the exchange area, zeros into the others
__global__ __launch_bounds__(BS) void kq_dist_pack_max(const double *part, int base_slot, int nslot, int nb, double *dst, int rank, int world) {
  __shared__ double sm[WAVES];
  for (int s = 0; s < nslot; ++s) {
    double t = 0.0;
    for (int i = threadIdx.x; i < nb; i += BS) t = fmax(t, part[(size_t)(base_slot + s) * MAXNB + i]);
    t = wave_max(t);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { double r = sm[0]; for (int wv = 1; wv < WAVES; ++wv) r = fmax(r, sm[wv]); for (int g = 0; g < world; ++g) dst[(size_t)s * world + g] = g == rank ? r : 0.0; }
  }
}
// after the all-reduce of buf = sum_g A_g H^-1 g_x: the PCG's right-hand side -g_y - buf in the y block
__global__ __launch_bounds__(BS) void kq_dist_prep_fin(double *__restrict__ rhs, const double *__restrict__ buf, int m, const Ctl *hc) {
  if (hc->halt) return;
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) rhs[i] = -rhs[i] - buf[i];
}
// kq_pcg_Gp after the all-reduce of buf = sum_g A_g t_n: the element-wise half and its partials (replicated m-space: no second exchange)
template <bool INIT>
__global__ __launch_bounds__(BS) void kq_dist_Gp_fin(QPcgVec v, double *__restrict__ ysol, const double *__restrict__ buf, double rho_y, double tol_host, double iter_pow,
                                                      const double *wmv, int world, int m, double *ppart, Ctl *hc) {
  if (hc->halt || hc->cg_done) return;
  __shared__ double sm[2 * WAVES];
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  if (INIT) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      double wm = 0.0;
      for (int g = 0; g < world; ++g) wm = fmax(wm, wmv[g]);
      hc->cg_tol = fmax(0.2 * fmin(tol_host, wm / iter_pow), 1e-12); // abip.c:213-217
    }
    double a2[2] = {0.0, 0.0};
    for (int i = t0; i < m; i += stride) {
      const double y0 = v.y0[i];
      const double ri = ysol[i] - (y0 * rho_y + buf[i]), zi = ri * v.Minv[i];
      ysol[i] = y0; v.r[i] = ri; v.z[i] = zi; v.p[i] = zi;
      a2[0] += ri * ri; a2[1] += zi * ri;
    }
    const int ws[2] = {PQ_RM0, PQ_ZR0};
    write_partials<2>(ppart, ws, a2, sm);
  } else {
    const double beta = hc->beta_cur;
    double acc1[1] = {0.0};
    for (int i = t0; i < m; i += stride) {
      const double pn = beta == 0.0 ? v.z[i] : v.z[i] + beta * v.p[i];
      const double gp = pn * rho_y + buf[i];
      v.p[i] = pn; v.Gp[i] = gp;
      acc1[0] += pn * gp;
    }
    const int ws[1] = {PQ_PG};
    write_partials<1>(ppart, ws, acc1, sm);
  }
}
// dq_inner_A after the all-reduce of buf = sum_g A_g u_x: A x and the y-block sums of the inner test (weight wy: replicated)
__global__ __launch_bounds__(BS) void kq_dist_inner_A_fin(const double *__restrict__ buf, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ b,
                                                           double *__restrict__ Ax, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double sm[5 * WAVES];
  const double tau = u[d.MP + d.n];
  double a5[5] = {0, 0, 0, 0, 0};
  for (int i = blockIdx.x * BS + threadIdx.x; i < d.m; i += gridDim.x * BS) {
    const double mu = buf[i], qu = mu + (-tau) * b[i], dv = qu - vo[i];
    Ax[i] = mu;
    a5[0] += u[i] * mu; a5[1] += u[i] * b[i]; a5[2] += dv * dv; a5[3] += d.norm_u ? u[i] * u[i] : qu * qu; a5[4] += vo[i] * vo[i];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) a5[q] *= d.wy;
  const int ws[5] = {Q_D1, Q_D2, Q_E1, Q_E2, Q_E3};
  write_partials<5>(part, ws, a5, sm);
}
// packed scalar exchange: gs[s] = requested ? out[s] : 0 before the all-reduce, out[s] = gs[s] after it
struct QPack { int n; int slots[24]; };
__global__ void kq_dist_pack(QPack p, const double *out, double *gs, int count) {
  for (int s = threadIdx.x; s < count; s += blockDim.x) gs[s] = 0.0;
  __syncthreads();
  if ((int)threadIdx.x < p.n) gs[p.slots[threadIdx.x]] = out[p.slots[threadIdx.x]];
}
__global__ void kq_dist_unpack(QPack p, const double *gs, double *out) {
  if ((int)threadIdx.x < p.n) out[p.slots[threadIdx.x]] = gs[p.slots[threadIdx.x]];
}
// the residual check's inf-norms: slot s of rank g sits at mx[s * world + g] after the exchange
__global__ void kq_dist_unpack_max(const double *mx, int base_slot, int nslot, int world, double *out) {
  if ((int)threadIdx.x < nslot) { double r = 0.0; for (int g = 0; g < world; ++g) r = fmax(r, mx[(size_t)threadIdx.x * world + g]); out[base_slot + threadIdx.x] = r; }
}
// gather of a sharded n-vector for the caller: the rank's block into its place of a zeroed global vector (then summed over the ranks)
__global__ __launch_bounds__(BS) void kq_dist_place(const double *__restrict__ src, int n_loc, int n0, double *__restrict__ dst) {
  for (int j = blockIdx.x * BS + threadIdx.x; j < n_loc; j += gridDim.x * BS) dst[n0 + j] = src[j];
}

} // namespace abip
