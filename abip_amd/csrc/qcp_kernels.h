// qcp_kernels.h -- HIP kernels of the conic (ABIP-QCP) inner iteration (reference: src/abip-qcp/source/abip.c:1130-1157):
//
//   kq_rhs           mu = rho o (u+v); rhs = (-mu_y, mu_x); partial r'mu                  abip.c:198-202,222 ; qcp_config.c:873
//   (LDL' solve)     dev_sptrsv.h on the QCP KKT matrix                                   linsys.c:309-316, qdldl.c:236-281
//   kq_dots, kq_Qp   partial r'(rho o p), p_x'Q p_x                                       abip.c:226-239
//   kq_ut_prox       tau~ from the scalar quadratic; u_t = p - tau~ r; over-relaxed point; y, tau, orthant/free/zero
//                    blocks of the barrier sub-problem                                    abip.c:241-248, 336-353, 390-409 ; cones.c:255-288
//   kq_cones         SOC / rotated-SOC barrier prox, one wavefront (workgroup if large) per cone  cones.c:130-248
//   kq_dual          v = u - rel_ut ; v_origin = rho o v                                  abip.c:314-324, 1143-1144
//   kq_inner_*       Mu = (A x ; -A'y + Q x) and the sums of the inner stopping test      qcp_config.c:518-557
//   kq_resid         inf-norm residuals, objectives, certificates from the stored A x, A'y, Q x   qcp_config.c:562-691
//   kq_finalize      partials -> scalars (sum or max)
#pragma once
#include "dev_common.h"

namespace abip {

struct QDims { int m, n, MP; int norm_u; double wy; }; // norm_u: the inner test normalises by 1 + |u| + |v_o| (1: lasso_config.c:343-345) or 1 + |(u, v_o)| (2: svm_config.c:266-268) instead of 1 + |Qu| + |v_o| (0: qcp_config.c:549-551);
// wy: weight of the sums over the y block -- 1 on a single GPU; with the columns sharded over several (qcp_dist.h) the y block is replicated and only rank 0 counts it

enum QSlot : int { // reuse of the partials table; *_MAX slots are reduced with max
  Q_T0 = 0, Q_T1, Q_PG,                       // r'mu, r'(rho o p), p_x'Qp
  Q_D1, Q_D2, Q_D3, Q_E1, Q_E2, Q_E3,         // inner test: u'Mu, u_y'b, u_x'c, |Qu - v_o|^2, |Qu|^2, |v_o|^2 (tau entry added by the host)
  Q_S0, Q_S1, Q_S2, Q_S3, Q_S4, Q_S5,         // |D o Ax|^2, b'u_y, c'u_x, u_x'Qx, |E o Qx|^2, |E o (A'y + v_o)|^2
  Q_M0, Q_M1, Q_M2, Q_M3, Q_M4, Q_M5,         // max slots: |Ax/t-b|, D|Ax/t-b|, D|Ax/t|, |R|, E|R|, E|Qx/t|
  Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5,         // LASSO residuals (kq_resid_lasso): |pr|^2, |dr|^2, x'x, 1'(beta+ + beta-), z'z, y'z;  SVM (kq_resid_svm): |pr|^2, |dr|^2, 1'y, 1'xi, |w|^2, |B'y|^2
  Q_COUNT
};
__host__ __device__ inline bool qslot_is_max(int s) { return s >= Q_M0 && s <= Q_M5; }

struct QCtl {
  double out[32]; double tau_t;
  double u_tail;    // new tau entry of u from kq_ut_prox; committed by kq_dual (every workgroup of kq_ut_prox still reads the old one)
  double err_inner; // inner stopping metric of the last completed iteration (qcp_config.c:518-557), evaluated by kq_finalize
  int it_count;     // inner iterations completed since the start of the solve
  int halted;       // mirror of the halt flag the gated kernels test (the host reads this block once per batch)
};

__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
  return x;
}
template <int NS>
__device__ __forceinline__ void write_partials_max(double *part, const int (&slots)[NS], double (&v)[NS], double *sm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] = wave_max(v[s]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sm[s * WAVES + wave] = v[s];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double t = sm[s * WAVES];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) t = fmax(t, sm[s * WAVES + w]);
      part[slots[s] * MAXNB + blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BS) void kq_rhs(const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ r,
                                             double *__restrict__ p, double rho_y, double rho_x, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) { const double mu = (u[i] + v[i]) * rho_y; p[i] = -mu; acc[0] += r[i] * mu; }
  acc[0] *= d.wy;
  for (int j = t0; j < d.n; j += stride) { const int q = d.MP + j; const double mu = (u[q] + v[q]) * rho_x; p[q] = mu; acc[0] += r[q] * mu; }
  const int ws[1] = {Q_T0};
  write_partials<1>(part, ws, acc, sm);
}
__global__ __launch_bounds__(BS) void kq_dots(const double *__restrict__ r, const double *__restrict__ p, double rho_y, double rho_x, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double sm[WAVES];
  double acc[1] = {0.0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) acc[0] += r[i] * (p[i] * rho_y);
  acc[0] *= d.wy;
  for (int j = t0; j < d.n; j += stride) acc[0] += r[d.MP + j] * (p[d.MP + j] * rho_x);
  const int ws[1] = {Q_T1};
  write_partials<1>(part, ws, acc, sm);
}
__global__ __launch_bounds__(BS) void kq_Qp(Csr Q, const double *__restrict__ p, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[WAVES];
  const double *px = p + d.MP;
  double acc1[1] = {0.0};
  spmv_stream<1>(
      Q, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * px[c]; },
      [&](int i, double(&acc)[1]) { acc1[0] += px[i] * acc[0]; });
  const int ws[1] = {Q_PG};
  write_partials<1>(part, ws, acc1, sm);
}

// x-entry kinds (host-built from the cone description, abip.c:358-409)
enum : int { XK_ORTHANT = 0, XK_FREE = 1, XK_ZERO = 2, XK_CONE = 3, XK_NONE = 4 };

__device__ __forceinline__ double orthant_prox(double t, double lambda) { // cones.c:279-288
  if (t >= 0) return (t + sqrt(t * t + 4 * lambda)) / 2;
  return 2 * lambda / (-t * (1 + sqrt(1 + 4 * lambda / (t * t))));
}

struct QProxArgs {
  double *u, *v, *ut, *rel;
  const double *p, *r;
  const int *xkind;
  double alpha, lambda, rho_x, rho_tau, a_quad;
  int iter_pos; // iter > 0
  int hasQ;
};
// gsc: null on a single GPU (the three sums are re-reduced from the partials); with sharded columns the table of all-reduced sums (ctl->out)
__global__ __launch_bounds__(BS) void kq_ut_prox(QProxArgs a, QDims d, const double *part, int nb, QCtl *ctl, const Ctl *hc, const double *gsc) {
  if (hc->halt) return;
  __shared__ double sm[3 * WAVES];
  double s3[3];
  const int rs[3] = {Q_T0, Q_T1, Q_PG};
  if (gsc) { s3[0] = gsc[Q_T0]; s3[1] = gsc[Q_T1]; s3[2] = gsc[Q_PG]; }
  else read_partials<3>(part, rs, nb, s3, sm);
  const int tail = d.MP + d.n;
  const double eta = a.rho_tau * (a.u[tail] + a.v[tail]);
  const double bq = s3[0] - 2 * s3[1] - eta;               // abip.c:229-230
  const double cq = a.hasQ ? -s3[2] : -0.0;                 // abip.c:239
  const double tt = a.iter_pos ? (-bq + sqrt(fmax(0.0, bq * bq - 4 * a.a_quad * cq))) / (2 * a.a_quad) : 1.0; // abip.c:241-245
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) {
    const double uti = a.p[i] + (-tt) * a.r[i];             // abip.c:247-248
    const double rl = uti * a.alpha + (1 - a.alpha) * a.u[i] - a.v[i]; // abip.c:336-342
    a.ut[i] = uti; a.rel[i] = rl; a.u[i] = rl;              // abip.c:352
  }
  const double lam_x = a.lambda / a.rho_x;
  for (int j = t0; j < d.n; j += stride) {
    const int q = d.MP + j;
    const double uti = a.p[q] + (-tt) * a.r[q];
    const double rl = uti * a.alpha + (1 - a.alpha) * a.u[q] - a.v[q];
    a.ut[q] = uti; a.rel[q] = rl;
    const int kd = a.xkind[j];
    if (kd == XK_ORTHANT) a.u[q] = orthant_prox(rl, lam_x);
    else if (kd == XK_FREE) a.u[q] = rl;
    else if (kd == XK_ZERO) a.u[q] = 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double rl = tt * a.alpha + (1 - a.alpha) * a.u[tail] - a.v[tail];
    a.ut[tail] = tt; a.rel[tail] = rl;
    // abip.c:348-350.  NOT stored into u[tail] here: the other workgroups of this launch read u[tail] for eta above, and nothing
    // orders their reads before this write; kq_dual, the next kernel to touch the tau entry, commits it.
    ctl->u_tail = (rl + sqrt(rl * rl + 4 * a.lambda / a.rho_tau)) / 2;
    ctl->tau_t = tt;
  }
}

// SOC / RSOC barrier prox: ||tail||^2 by a reduction, closed form for the head entries and one scale factor for the tail.
// Small cones: one wavefront per cone.  Large cones (> QC_BIG entries): one 1024-thread workgroup per cone.
struct QCones { const int *off, *len, *kind; int n; }; // kind 0 SOC, 1 RSOC
constexpr int QC_BIG = 2048;
constexpr int QC_TB = 1024;

// closed forms of cones.c:130-161 (SOC) and cones.c:169-248 (RSOC) given sq = ||tail||^2
__device__ __forceinline__ void cone_closed_form(int kind, const double *__restrict__ rel, const double *u, int off, double sq, double lambda,
                                                 double &x0, double &x1, double &sc) {
  x0 = 0; x1 = 0; sc = 0;
  if (kind == 0) {
    const double a = rel[off];
    if (fabs(a) <= 1e-9) { x0 = sqrt(2 * lambda + sq / 4); sc = 0.5; }
    else {
      const double w8 = 8 * lambda - a * a + sq;
      const double rr = 16 * a * a / (w8 + sqrt(w8 * w8 + 32 * a * a * lambda));
      const double s1 = (rr - sqrt(rr * (rr + 8))) / 2, s2 = (rr + sqrt(rr * (rr + 8))) / 2;
      const double s = a > 0 ? s2 : s1;
      x0 = (s + 2) * a / s; sc = (s + 2) / (s + 4);
    }
  } else {
    const double ze = rel[off], zn = rel[off + 1];
    if (ze + zn == 0) {
      x1 = (-ze + sqrt(ze * ze + 4 * lambda + sq)) / 2;
      x0 = u[off] + ze; // sic (cones.c:183)
      sc = 0.5;
    } else {
      double w, s;
      const double dd = 2 * ze * zn - sq;
      if (dd < 0) { const double q = -dd / (2 * lambda); w = (2 * (ze + zn) * (ze + zn) / lambda) / q / (1 + 4 / q + sqrt(1 + (4 * (ze * ze + zn * zn + sq) / lambda + 16) / q / q)); }
      else { const double q = dd / (2 * lambda); w = q * (1 - 4 / q + sqrt(1 + (4 * (ze * ze + zn * zn + sq) / lambda + 16) / q / q)) / 2; }
      if (ze + zn > 0) {
        s = (w + sqrt(w * (w + 4))) / 2;
        x0 = (ze * (s + 1) * (s + 1) + zn * (s + 1)) / (s * (s + 2)); x1 = (zn * (s + 1) * (s + 1) + ze * (s + 1)) / (s * (s + 2)); sc = (s + 1) / (s + 2);
      } else if (w > 10) {
        s = 2 / (w + 2 + sqrt(w * (w + 4)));
        x0 = (ze * s * s + zn * s) / ((s - 1) * (s + 1)); x1 = (zn * s * s + ze * s) / ((s - 1) * (s + 1)); sc = s / (s + 1);
      } else {
        s = (w - sqrt(w * (w + 4))) / 2;
        x0 = (ze * (s + 1) * (s + 1) + zn * (s + 1)) / (s * (s + 2)); x1 = (zn * (s + 1) * (s + 1) + ze * (s + 1)) / (s * (s + 2)); sc = (s + 1) / (s + 2);
      }
    }
  }
}

// cones [first, C.n) of the (small-first) cone table
template <bool BIG>
__global__ __launch_bounds__(BIG ? QC_TB : BS) void kq_cones(QCones C, int first, double *__restrict__ u, const double *__restrict__ rel, double lambda, int MP, const Ctl *hc) {
  if (hc && hc->halt) return;
  constexpr int NT = BIG ? QC_TB : 64;
  const int cone = first + (BIG ? (int)blockIdx.x : (int)((blockIdx.x * BS + threadIdx.x) >> 6));
  const int t = BIG ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  if (cone >= C.n) return;
  const int off = MP + C.off[cone], len = C.len[cone], kind = C.kind[cone];
  const int h = (kind == 0) ? 1 : 2; // head entries
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = h + t;
  for (; k + 3 * NT < len; k += 4 * NT) {
    const double a = rel[off + k], b = rel[off + k + NT], c = rel[off + k + 2 * NT], d = rel[off + k + 3 * NT];
    s0 += a * a; s1 += b * b; s2 += c * c; s3 += d * d;
  }
  for (; k < len; k += NT) { const double a = rel[off + k]; s0 += a * a; }
  double sq = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if (BIG) {
    __shared__ double sm[QC_TB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sq;
    __syncthreads();
    sq = 0.0;
#pragma unroll
    for (int wv = 0; wv < QC_TB / 64; ++wv) sq += sm[wv];
  }
  double x0, x1, sc;
  cone_closed_form(kind, rel, u, off, sq, lambda, x0, x1, sc);
  if (BIG) __syncthreads(); // every thread has read u[off] before it is overwritten
  if (t == 0) { u[off] = x0; if (kind != 0) u[off + 1] = x1; }
  for (int q = h + t; q < len; q += NT) u[off + q] = rel[off + q] * sc;
}

__global__ __launch_bounds__(BS) void kq_dual(double *__restrict__ u, const double *__restrict__ rel, double *__restrict__ v, double *__restrict__ vo,
                                              double rho_y, double rho_x, double rho_tau, QDims d, const QCtl *qc, const Ctl *hc) {
  if (hc->halt) return;
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) { const double t = u[i] - rel[i]; v[i] = t; vo[i] = t * rho_y; }
  for (int j = t0; j <= d.n; j += stride) {
    const int q = d.MP + j;
    double uq = u[q];
    if (j == d.n) { uq = qc->u_tail; u[q] = uq; } // the tau entry kq_ut_prox computed
    const double t = uq - rel[q];
    v[q] = t; vo[q] = t * (j == d.n ? rho_tau : rho_x);
  }
}

// ---- inner stopping test (qcp_config.c:518-557); the tau entry of Qu is completed by the host from the sums ----
// bodies; (vb, vgrid) = this workgroup's index / count among those working on the product
__device__ __forceinline__ void dq_inner_A(const Csr &A, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ b,
                                           double *__restrict__ Ax, const QDims &d, double *part, double *lds, int *lptr, double *sm, int vb, int vgrid) {
  const double *x = u + d.MP;
  const double tau = u[d.MP + d.n];
  double a5[5] = {0, 0, 0, 0, 0};
  spmv_stream<1>(
      A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; },
      [&](int i, double(&acc)[1]) {
        const double mu = acc[0], qu = mu + (-tau) * b[i], dv = qu - vo[i];
        Ax[i] = mu;
        a5[0] += u[i] * mu; a5[1] += u[i] * b[i]; a5[2] += dv * dv; a5[3] += d.norm_u ? u[i] * u[i] : qu * qu; a5[4] += vo[i] * vo[i];
      },
      [] { return true; }, vb, vgrid);
  const int ws[5] = {Q_D1, Q_D2, Q_E1, Q_E2, Q_E3};
  write_partials<5>(part, ws, a5, sm, vb);
}
__device__ __forceinline__ void dq_inner_At(const Csr &At, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ c,
                                            double *__restrict__ ATy, double *__restrict__ Qx, int finish, const QDims &d, double *part,
                                            double *lds, int *lptr, double *sm, int vb, int vgrid) {
  const double tau = u[d.MP + d.n];
  double a5[5] = {0, 0, 0, 0, 0};
  spmv_stream<1>(
      At, lds, lptr, sm, [&](int cc, double a, double(&pr)[1]) { pr[0] = a * u[cc]; },
      [&](int j, double(&acc)[1]) {
        ATy[j] = acc[0];
        if (finish) {
          const int q = d.MP + j;
          const double mu = -acc[0], qu = mu + tau * c[j], dv = qu - vo[q];
          Qx[j] = 0.0;
          a5[0] += u[q] * mu; a5[1] += u[q] * c[j]; a5[2] += dv * dv; a5[3] += d.norm_u ? u[q] * u[q] : qu * qu; a5[4] += vo[q] * vo[q];
        }
      },
      [] { return true; }, vb, vgrid);
  if (finish) {
    if (vb == 0 && threadIdx.x == 0) { const double t = vo[d.MP + d.n]; a5[4] += t * t * d.wy; }
    const int ws[5] = {Q_D1 + 0, Q_D3, Q_E1, Q_E2, Q_E3};
    // the y rows already own slots D1, E1..E3: use the second half of the table (offset Q_COUNT) for the x rows
    double *part2 = part + (size_t)Q_COUNT * MAXNB;
    write_partials<5>(part2, ws, a5, sm, vb);
  }
}
__global__ __launch_bounds__(BS) void kq_inner_A(Csr A, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ b,
                                                 double *__restrict__ Ax, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[5 * WAVES];
  dq_inner_A(A, u, vo, b, Ax, d, part, lds, lptr, sm, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(BS) void kq_inner_At(Csr At, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ c,
                                                  double *__restrict__ ATy, double *__restrict__ Qx, int finish, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[5 * WAVES];
  dq_inner_At(At, u, vo, c, ATy, Qx, finish, d, part, lds, lptr, sm, (int)blockIdx.x, (int)gridDim.x);
}
// the two products of the inner stopping test in one launch: workgroups [0, nbA) take A u_x, the rest A'u_y
__global__ __launch_bounds__(BS) void kq_inner_both(Csr A, Csr At, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ b,
                                                    const double *__restrict__ c, double *__restrict__ Ax, double *__restrict__ ATy, double *__restrict__ Qx,
                                                    int finish, QDims d, int nbA, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[5 * WAVES];
  if ((int)blockIdx.x < nbA) dq_inner_A(A, u, vo, b, Ax, d, part, lds, lptr, sm, (int)blockIdx.x, nbA);
  else dq_inner_At(At, u, vo, c, ATy, Qx, finish, d, part, lds, lptr, sm, (int)blockIdx.x - nbA, (int)gridDim.x - nbA);
}
__global__ __launch_bounds__(BS) void kq_inner_Q(Csr Q, const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ c,
                                                 const double *__restrict__ ATy, double *__restrict__ Qx, QDims d, double *part, const Ctl *hc) {
  if (hc->halt) return;
  __shared__ double lds[CHUNK];
  __shared__ int lptr[CHUNK + 1];
  __shared__ double sm[5 * WAVES];
  const double *x = u + d.MP;
  const double tau = u[d.MP + d.n];
  double a5[5] = {0, 0, 0, 0, 0};
  spmv_stream<1>(
      Q, lds, lptr, sm, [&](int cc, double a, double(&pr)[1]) { pr[0] = a * x[cc]; },
      [&](int j, double(&acc)[1]) {
        const int q = d.MP + j;
        Qx[j] = acc[0];
        const double mu = -ATy[j] + acc[0], qu = mu + tau * c[j], dv = qu - vo[q];
        a5[0] += u[q] * mu; a5[1] += u[q] * c[j]; a5[2] += dv * dv; a5[3] += qu * qu; a5[4] += vo[q] * vo[q];
      });
  if (blockIdx.x == 0 && threadIdx.x == 0) { const double t = vo[d.MP + d.n]; a5[4] += t * t * d.wy; }
  const int ws[5] = {Q_D1 + 0, Q_D3, Q_E1, Q_E2, Q_E3};
  double *part2 = part + (size_t)Q_COUNT * MAXNB;
  write_partials<5>(part2, ws, a5, sm);
}

// ---- residuals (qcp_config.c:562-691) from the stored products; x = u_x / tau etc. folded into the formulas ----
__global__ __launch_bounds__(BS) void kq_resid(const double *__restrict__ u, const double *__restrict__ vo, const double *__restrict__ b,
                                               const double *__restrict__ c, const double *__restrict__ Dv, const double *__restrict__ Ev,
                                               const double *__restrict__ Ax, const double *__restrict__ ATy, const double *__restrict__ Qx,
                                               QDims d, double *part) {
  __shared__ double sm[6 * WAVES];
  const double tau = fabs(u[d.MP + d.n]), it = 1 / tau;
  double s6[6] = {0, 0, 0, 0, 0, 0}, m6[6] = {0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  for (int i = t0; i < d.m; i += stride) {
    const double ax = Ax[i] * it, e = ax - b[i], Di = Dv[i];
    m6[0] = fmax(m6[0], fabs(e)); m6[1] = fmax(m6[1], fabs(e * Di)); m6[2] = fmax(m6[2], fabs(ax * Di));
    const double dax = Di * Ax[i];
    s6[0] += dax * dax; s6[1] += b[i] * u[i];
  }
  s6[0] *= d.wy; s6[1] *= d.wy;
  for (int j = t0; j < d.n; j += stride) {
    const int q = d.MP + j;
    const double qx = Qx[j] * it, aty = ATy[j] * it, s = vo[q] * it, Ej = Ev[j];
    const double R = qx - aty + c[j] - s;
    m6[3] = fmax(m6[3], fabs(R)); m6[4] = fmax(m6[4], fabs(R * Ej)); m6[5] = fmax(m6[5], fabs(qx * Ej));
    const double eq = Ej * Qx[j], ea = Ej * (ATy[j] + vo[q]);
    s6[2] += c[j] * u[q]; s6[3] += u[q] * Qx[j]; s6[4] += eq * eq; s6[5] += ea * ea;
  }
  const int ws[6] = {Q_S0, Q_S1, Q_S2, Q_S3, Q_S4, Q_S5};
  write_partials<6>(part, ws, s6, sm);
  const int wm[6] = {Q_M0, Q_M1, Q_M2, Q_M3, Q_M4, Q_M5};
  __syncthreads();
  write_partials_max<6>(part, wm, m6, sm);
}

// ---- residuals of the LASSO reformulation (lasso_config.c:358-460) from the stored products of the scaled operator ----
// With X~ = D X E the scaled data block: X (E o u_b) = D^-1 (X~ u_b) and X'(D o u_y) = E^-1 (X~' u_y), so the products with the caller's
// un-scaled X that the reference forms are the stored A u_x (rows 1..dm) and A'u_y (the beta columns) divided by D resp. E:
//   pr_i = (A u_x)_{1+i} / (D_i tau sc_b) - y_i,   dr_j = (v_j + (A'u_y)_j) / (E_j tau sc_c) - lambda   (both beta blocks),
//   x = sqrt(sc_cone2) u_z / (tau sc_b),  beta+- = E o u_b / (tau sc_b),  z = D o u_y[1:] / (tau sc_c).
// Several GPUs (qcp_dist.h): x-block indices below are GLOBAL columns; the rank owns [n0, n0 + d.n) of them and adds only what it owns, the sums
// over the replicated y block and over A u_x (replicated after its exchange) carry the weight d.wy.  Single GPU: n0 = 0, every column owned.
struct QLasso { int dm, dn; double sqrt_sc2, sc_b, sc_c, lambda; const double *D, *E, *y; int n0; };
__global__ __launch_bounds__(BS) void kq_resid_lasso(const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ Ax,
                                                     const double *__restrict__ ATy, QLasso L, QDims d, double *part) {
  __shared__ double sm[6 * WAVES];
  const double tau = u[d.MP + d.n], ib = 1.0 / (tau * L.sc_b), ic = 1.0 / (tau * L.sc_c);
  double s6[6] = {0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  const double *ux = u + d.MP - L.n0, *vx = v + d.MP - L.n0, *ATg = ATy - L.n0; // indexed by global column
  const int g0 = L.n0, g1 = L.n0 + d.n;
  for (int i = t0; i < L.dm; i += stride) {
    const double Di = L.D[i], yi = L.y[i];
    const double pr = Ax[1 + i] / Di * ib - yi, z = u[1 + i] * (Di * ic);
    s6[0] += pr * pr * d.wy; s6[4] += z * z * d.wy; s6[5] += yi * z * d.wy;
    const int gq = 2 + i;
    if (gq >= g0 && gq < g1) { const double x = ux[gq] * (L.sqrt_sc2 * ib); s6[2] += x * x; }
  }
  for (int e = t0; e < 2 * L.dn; e += stride) {
    const int q = L.dm + 2 + e;
    if (q < g0 || q >= g1) continue;
    const double Ej = L.E[e < L.dn ? e : e - L.dn];
    const double dr = (vx[q] + ATg[q]) / Ej * ic - L.lambda;
    s6[1] += dr * dr; s6[3] += ux[q] * (Ej * ib);
  }
  const int ws[6] = {Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5};
  write_partials<6>(part, ws, s6, sm);
}

// ---- residuals of the SVM-SOCP reformulation (svm_config.c:445-561) from the stored products of the scaled operator ----
// x-block layout: x0, x1, r (dn), w+ (dn), b+, w- (dn), b-, xi (dm), t (dm).  With A~ = D [diag(y) X, y] E the scaled data block:
//   pr_i = (A u_x)_{1+i} / (D_i tau sc_b) - 1            (= xi + data_A (w, b) - t - 1: the xi and t columns carry D / sc and -D),
//   (B'y)_j = ((A'u_y)_{w+_j} + wE_j u_y[1+dm+j]) / (E_j tau sc_c),   y = D o u_y[1:1+dm] / (tau sc_c),
//   dr = (y - s2 ; y + s1 - C) with s1, s2 the duals of xi, t;  w = E o (u_w+ - u_w-) / (tau sc_b);  xi = u_xi / (tau sc sc_b).
struct QSvm { int dm, dn; double sc, sc_b, sc_c, C; const double *D, *E, *wE; };
__global__ __launch_bounds__(BS) void kq_resid_svm(const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ Ax,
                                                   const double *__restrict__ ATy, QSvm V, QDims d, double *part) {
  __shared__ double sm[6 * WAVES];
  const double tau = u[d.MP + d.n], ib = 1.0 / (tau * V.sc_b), ic = 1.0 / (tau * V.sc_c);
  double s6[6] = {0, 0, 0, 0, 0, 0};
  const int stride = gridDim.x * BS, t0 = blockIdx.x * BS + threadIdx.x;
  // (single GPU only: w+ / w- and xi / t pair entries of different columns, which a column cut would separate -- the sharded path serves the
  //  LASSO and the QP formulation of the SVM, not this one)
  const double *ux = u + d.MP, *vx = v + d.MP;
  const int o_xi = 3 * V.dn + 4, o_t = o_xi + V.dm;
  for (int i = t0; i < V.dm; i += stride) {
    const double Di = V.D[i];
    const double pr = Ax[1 + i] / Di * ib - 1.0, y = u[1 + i] * (Di * ic), s2 = vx[o_t + i] * ic, s1 = vx[o_xi + i] * (V.sc * ic);
    const double a = y - s2, b = y + s1 - V.C;
    s6[0] += pr * pr; s6[1] += a * a + b * b; s6[2] += y; s6[3] += ux[o_xi + i] * (ib / V.sc);
  }
  for (int j = t0; j < V.dn; j += stride) {
    const double Ej = V.E[j];
    const double wj = (ux[V.dn + 2 + j] - ux[2 * V.dn + 3 + j]) * (Ej * ib);
    const double bty = (ATy[V.dn + 2 + j] + V.wE[j] * u[1 + V.dm + j]) / Ej * ic;
    s6[4] += wj * wj; s6[5] += bty * bty;
  }
  const int ws[6] = {Q_L0, Q_L1, Q_L2, Q_L3, Q_L4, Q_L5};
  write_partials<6>(part, ws, s6, sm);
}

// one block: partials (both halves of the table) -> ctl->out
struct QFin {
  int nslots; int slots[24]; int second_half[24];
  // decide = 1 (the finalize that closes an inner iteration): complete the inner stopping metric with the tau entries, count the
  // iteration and raise the halt flag when the metric is below tol_inner, so that iterations enqueued behind it fall through
  int decide = 0;
  int norm_u = 0; // as QDims::norm_u
  double tol_inner = 0.0;
  const double *u_tau = nullptr, *vo_tau = nullptr;
};
// complete the inner stopping metric with the tau entries, count the iteration, raise the halt flag (one thread)
__device__ __forceinline__ void q_decide(const QFin &f, QCtl *ctl, Ctl *hc) { // un-fused arithmetic: the host code this replaces was compiled without FMA contraction
  const double *o = ctl->out;
  const double tau = *f.u_tau, vot = *f.vo_tau;
  const double qut = -o[Q_D1] / tau + o[Q_D2] - o[Q_D3];
  const double dq = qut - vot;
  const double t2 = f.norm_u ? tau : qut;
  const double e1 = __dadd_rn(o[Q_E1], __dmul_rn(dq, dq)), e2 = __dadd_rn(o[Q_E2], __dmul_rn(t2, t2)), e3 = o[Q_E3];
  const double err = f.norm_u == 2 ? sqrt(e1) / (1 + sqrt(e2 + e3)) : sqrt(e1) / (1 + sqrt(e2) + sqrt(e3));
  ctl->err_inner = err; ctl->it_count = ctl->it_count + 1;
  if (err < f.tol_inner) { hc->halt = 1; ctl->halted = 1; }
}
__global__ __launch_bounds__(1024) void kq_finalize(QFin f, const double *part, int nb, QCtl *ctl, Ctl *hc) {
  if (f.decide && hc->halt) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  constexpr int PER = MAXNB / 64;
  for (int s = wave; s < f.nslots; s += nwaves) {
    const int slot = f.slots[s];
    const bool mx = qslot_is_max(slot);
    double acc = 0.0;
    for (int half = 0; half <= f.second_half[s]; ++half) {
      const double *pp = part + ((size_t)half * Q_COUNT + slot) * MAXNB;
      double t[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) { const int i = lane + u * 64; t[u] = (i < nb) ? pp[i] : 0.0; }
      double a2 = 0.0;
#pragma unroll
      for (int u = 0; u < PER; ++u) a2 = mx ? fmax(a2, t[u]) : a2 + t[u];
      a2 = mx ? wave_max(a2) : wave_sum(a2);
      acc = mx ? fmax(acc, a2) : acc + a2;
    }
    if (lane == 0) ctl->out[slot] = acc;
  }
  if (!f.decide) return;
  __syncthreads();
  if (threadIdx.x == 0) q_decide(f, ctl, hc);
}
// sharded columns: the sums were reduced locally (decide = 0), exchanged, and written back to ctl->out; this closes the iteration
__global__ void kq_decide(QFin f, QCtl *ctl, Ctl *hc) {
  if (hc->halt) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) q_decide(f, ctl, hc);
}

} // namespace abip
