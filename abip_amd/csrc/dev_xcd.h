// dev_xcd.h -- the whole inner ADMM loop of a cache-resident LP as ONE persistent launch on ONE XCD (gfx950).
//
// Why.  On Netlib / pds-class LPs a kernel of the launch path (dev_kernels.h) does 2-5 us of work and the boundary to the next one costs
// 2-3 us more; a PCG iteration is three such kernels, an ADMM iteration of the direct back-end six.  Here G = 32 workgroups -- the 32 CUs of
// one XCD, so that one L2 serves every hand-over -- run `max_iters` ADMM iterations (abip.c:2131-2215) inside one launch:
//
//   * the rows of A and of A' are cut into G contiguous slices (balanced by non-zeros); workgroup g OWNS rows [mb[g], mb[g+1]) of the
//     m-space and [nb[g], nb[g+1]) of the n-space: it alone reads and writes those entries of u, v, the running sums, the PCG vectors;
//   * a slice's non-zeros live in REGISTERS for the whole launch (thread t holds entries t, t + 1024, ...: NZ per thread), the PCG's
//     vectors too; nothing of the matrix is re-read per product;
//   * a product needs the whole operand vector: the owners PUBLISH their entries as 16-byte granules {lo32, tag, hi32, tag} (one plain
//     dwordx4 store each) and every consumer GATHERS the entries its non-zeros name with L1-bypassing 8-byte loads, polling the data
//     itself until both halves carry the tag of this exchange.  No flag, no fence, no grid barrier: tools/xcd_probe.hip measured
//     1.2 us per such exchange on one XCD against 6.1 us for release-fence + counter + acquire-fence (profiles/r03a_*);
//   * reductions ride along: every workgroup publishes its partial sums as granules of the same exchange and every workgroup adds the G
//     partials in rank order -- all workgroups hold bit-identical scalars and take the same decisions (PCG exit, inner-loop exit);
//   * the tag is the running number of the exchange, the buffers alternate with its parity: a workgroup can be at most one exchange
//     ahead of the slowest one, so a buffer is never overwritten while somebody may still read it.
//
// Placement.  The launch has 256 workgroups with > 80 KB of LDS each: one per CU, hence exactly 32 on every XCD.  The ones whose
// HW_REG_XCC_ID is not 0 return at once; the others draw a ticket (= rank).  Every poll is bounded and gives up through xstat[0]
// (the host then fails loudly), so a placement that breaks the assumption cannot hang the device.
//
// Arithmetic = the launch path's (dev_kernels.h) entry for entry; sums are taken in a different order (per row: entry order; per
// reduction: thread, wavefront, rank order), and p'Gp is formed as rho |p|^2 + |A'p|^2 (the sharded path's identity, saving one
// exchange per PCG iteration).
#pragma once
#include "dev_kernels.h"

namespace abip {

constexpr int XTB = 1024;          // threads per workgroup
constexpr int XWAVES = XTB / 64;
constexpr int XG = 32;             // workgroups taking part = CUs of one XCD
constexpr int XKS = 16;            // scalar granules per workgroup per exchange
constexpr int XHB = 15;            // ... the last of them is the heartbeat: "this rank has reached exchange <tag>"
constexpr int XSPIN = 1 << 22;     // polling rounds before a wavefront gives up (a round is ~1 us)
constexpr int XCD_LDS_MIN = 84 * 1024; // more than half a CU's LDS: one workgroup per CU

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long xu64;

struct XcdFinal { // the final_check branch of abip.c:2190-2213 evaluated on the device (calc_residuals + has_converged on the finalised sums)
  int on, pfeasopt, ipm_pos;
  double eps, den, nm_b, nm_c;
  long k0, max_admm;
};

struct XcdArgs {
  const int *Ap, *Ai; const double *Ax; // CSR of A  (m rows, gathers the n-space)
  const int *Tp, *Ti; const double *Tx; // CSR of A' (n rows, gathers the m-space)
  const int *mb, *nb;                   // G + 1 row bounds each
  int G, m, n, MP;
  UpdArgs upd;                          // u, v, ut, sums, g, b, c, alpha, mu/beta, rho, half_update (dom / avg_stats are set per iteration)
  const double *h, *wD, *wE;
  const double *Mjac;                   // PCG: Jacobi preconditioner (m)
  const double *Minv; long ldM;         // direct: inv(rho I + A A'), dense row-major
  double g_th;
  u32x4 *xn0, *xn1, *xm0, *xm1, *sc;    // exchange areas: 2 parities x (n_pad | m_pad | XG * XKS) granules
  int n_pad, m_pad;
  unsigned tag0;
  unsigned *tickets; unsigned ticket_base;
  Ctl *ctl; int *xstat;                 // xstat[0]: a poll gave up; xstat[1]: exchanges used by this launch
  long j0; int max_iters;               // inner index of the first iteration; iterations to run unless the exit test holds earlier
  double thr, sentinel;                 // gamma * mu; Qres_avg when no averaged statistics were taken
  const double *tolf; int cg_max_its;   // PCG: tolerance factor per iteration of this launch (host-computed: indirect.c:406-407)
  XcdFinal fc;
};

__device__ __forceinline__ unsigned x_xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// Addresses are formed as (uniform base pointer) + (32-bit byte offset) everywhere: the base stays in scalar registers and an
// element's offset is ONE vector register shared by every array it indexes (signed indices would cost a 64-bit address pair each).
typedef __amdgpu_buffer_rsrc_t xrsrc; // buffer resource: 4 scalar registers describe an exchange area; an access is (resource, 32-bit byte offset)
__device__ __forceinline__ xrsrc x_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void x_put(xrsrc r, unsigned off /* bytes */, double v, unsigned tag) { // one plain 16-byte store: the line stays in this XCD's L2
  u32x4 g; g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(g, r, (int)off, 0, 0);
}
__device__ __forceinline__ u32x4 x_ld(xrsrc r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16 /* sc1: past the L1, served by the L2 */); }
__device__ __forceinline__ bool x_ok(const u32x4 &g, unsigned tag) { return g.y == tag && g.w == tag; } // both 8-byte halves are of this exchange
__device__ __forceinline__ double x_val(const u32x4 &g) { return __hiloint2double((int)g.z, (int)g.x); }
// (the byte offset is formed in 32 bits: only then can it ride in the instruction's 32-bit offset register beside a scalar base)
template <class T> __device__ __forceinline__ T &x_at(T *base, unsigned i) { return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(unsigned)(i * (unsigned)sizeof(T))); }
template <class T> __device__ __forceinline__ const T &x_at(const T *base, unsigned i) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(unsigned)(i * (unsigned)sizeof(T))); }

// the same value in every lane, told to the compiler (scalar registers, uniform branches)
__device__ __forceinline__ double x_uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// wavefront sum on the DPP network (no LDS round trips: __shfl_down is a ds_bpermute per step); fixed order; the total lands in lane 63
template <int CTRL, int ROWS>
__device__ __forceinline__ double x_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x_wave_sum63(double v) {
  v += x_dpp<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
  v += x_dpp<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
  v += x_dpp<0x124, 0xf>(v); // row_ror:4
  v += x_dpp<0x128, 0xf>(v); // row_ror:8   -> every lane holds its row's sum
  v += x_dpp<0x142, 0xa>(v); // row_bcast:15 into rows 1, 3
  v += x_dpp<0x143, 0xc>(v); // row_bcast:31 into rows 2, 3
  return v;
}
__device__ __forceinline__ double x_wave_total(double v) { // every lane gets the total
  v = x_wave_sum63(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

struct XWait { // one thread's view of the exchange in flight
  xrsrc n0, n1, m0, m1, sc;
  unsigned tag;
  int *xstat;
  bool dead;
  int site, rank; // which wait of the iteration this is / whose: recorded when a wavefront gives up
};
// spin bookkeeping of one WAVEFRONT (its lanes poll together until all of them have their data): true = keep polling
__device__ __forceinline__ bool x_spin(XWait &w, int &spins, unsigned found, int what) {
  ++spins;
  if ((spins & 255) != 0) return true;
  int dead = 0;
  if ((threadIdx.x & 63) == 0) {
    if (spins > XSPIN && atomicCAS(w.xstat + 6, 0, 1) == 0) { // the first wavefront to give up leaves the record
      w.xstat[2] = (int)w.tag; w.xstat[3] = w.rank; w.xstat[4] = w.site; w.xstat[5] = (int)threadIdx.x; w.xstat[7] = (int)found; w.xstat[6] = 2 + what;
      __hip_atomic_store(w.xstat, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    dead = __hip_atomic_load(w.xstat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  dead = __builtin_amdgcn_readfirstlane(dead);
  if (dead) { w.dead = true; return false; }
  return true;
}

constexpr int x_rounds(int K) { return (K + XWAVES - 1) / XWAVES; } // scalar granules a thread may have to fetch

// Wait for (and fetch) what this thread needs of the exchange: the entries its A-slice non-zeros name in NVA n-space vectors, the entries
// its A'-slice non-zeros name in NVT m-space vectors, and the scalar granules of K partial sums: wavefront k (+ XWAVES r) fetches scalar k
// of rank `lane` -- the wavefront then adds the G partials on its own (x_sum_scalars).  All loads of a round are in flight together; the
// lanes of a wavefront poll together until all of them have everything (no divergent exits around the polling loads).
// ai / ti: BYTE offsets of the granules (16 * index), padded to NZ valid entries per thread (padding repeats an entry: always there).
template <int NZ, int NVA, int NVT, int K>
__device__ __forceinline__ void x_wait(XWait &w, int G, const unsigned (&ai)[NZ], const unsigned (&ti)[NZ], double (&va)[NVA > 0 ? NVA : 1][NZ],
                                       double (&vt)[NVT > 0 ? NVT : 1][NZ], double (&sv)[K > 0 ? x_rounds(K) : 1]) {
  constexpr int NS = K > 0 ? x_rounds(K) : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Every exchange is a full rendez-vous: the last wavefront also waits for the heartbeat granule of EVERY rank.  Without it a workgroup whose
  // non-zeros name entries of a few ranks only could run two exchanges ahead of a rank it does not depend on and overwrite a buffer (same
  // parity) that rank is still reading.
  const bool beat = (wave == XWAVES - 1) && lane < G;
  int spins = 0;
  // Only granules that have not arrived are asked for again, and a wavefront that has to ask again first sleeps a little: 32 k threads
  // re-polling everything flood the L2's request queues (~0.6 M requests a round), and the one rank everybody waits for -- its loads, its
  // spills, its stores -- queues behind that flood (measured: milliseconds per exchange).
  u32x4 a0[NVA > 0 ? NZ : 1], a1[NVA > 1 ? NZ : 1], c0[NVT > 0 ? NZ : 1], c1[NVT > 1 ? NZ : 1], sg[NS > 0 ? NS : 1], hb;
  const unsigned none = w.tag ^ 1u; // "not there yet"
  hb.y = beat ? none : w.tag; hb.w = w.tag; hb.x = 0; hb.z = 0;
#pragma unroll
  for (int u = 0; u < NZ; ++u) {
    if (NVA > 0) { a0[u].x = 0; a0[u].y = none; a0[u].z = 0; a0[u].w = none; }
    if (NVA > 1) { a1[u].x = 0; a1[u].y = none; a1[u].z = 0; a1[u].w = none; }
    if (NVT > 0) { c0[u].x = 0; c0[u].y = none; c0[u].z = 0; c0[u].w = none; }
    if (NVT > 1) { c1[u].x = 0; c1[u].y = none; c1[u].z = 0; c1[u].w = none; }
  }
#pragma unroll
  for (int r = 0; r < NS; ++r) {
    const int k = wave + r * XWAVES;
    sg[r].x = 0; sg[r].z = 0; sg[r].w = w.tag; sg[r].y = (k < K && lane < G) ? none : w.tag;
  }
  for (;;) {
    asm volatile("" ::: "memory"); // (the loads below are issued anew every round)
    if (hb.y != w.tag || hb.w != w.tag) hb = x_ld(w.sc, (unsigned)(lane * XKS + XHB) * 16u);
#pragma unroll
    for (int u = 0; u < NZ; ++u) {
      if (NVA > 0) { if (!x_ok(a0[u], w.tag)) a0[u] = x_ld(w.n0, ai[u]); }
      if (NVA > 1) { if (!x_ok(a1[u], w.tag)) a1[u] = x_ld(w.n1, ai[u]); }
      if (NVT > 0) { if (!x_ok(c0[u], w.tag)) c0[u] = x_ld(w.m0, ti[u]); }
      if (NVT > 1) { if (!x_ok(c1[u], w.tag)) c1[u] = x_ld(w.m1, ti[u]); }
    }
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      const int k = wave + r * XWAVES;
      if (!x_ok(sg[r], w.tag)) sg[r] = x_ld(w.sc, (unsigned)(lane * XKS + k) * 16u);
    }
    bool ok = x_ok(hb, w.tag);
#pragma unroll
    for (int u = 0; u < NZ; ++u) {
      if (NVA > 0) ok = ok & x_ok(a0[u], w.tag);
      if (NVA > 1) ok = ok & x_ok(a1[u], w.tag);
      if (NVT > 0) ok = ok & x_ok(c0[u], w.tag);
      if (NVT > 1) ok = ok & x_ok(c1[u], w.tag);
    }
#pragma unroll
    for (int r = 0; r < NS; ++r) ok = ok & x_ok(sg[r], w.tag);
    if (__all(ok ? 1 : 0)) {
#pragma unroll
      for (int u = 0; u < NZ; ++u) {
        if (NVA > 0) va[0][u] = x_val(a0[u]);
        if (NVA > 1) va[1][u] = x_val(a1[u]);
        if (NVT > 0) vt[0][u] = x_val(c0[u]);
        if (NVT > 1) vt[1][u] = x_val(c1[u]);
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) sv[r] = x_val(sg[r]); // (0.0 where this lane fetched nothing)
      return;
    }
    { // for the post-mortem: the first lane that still misses something speaks for the wavefront
      unsigned found = 0; int what = ok ? 0 : 7;
      if (!x_ok(hb, w.tag)) { found = hb.y; what = 1; }
#pragma unroll
      for (int r = NS - 1; r >= 0; --r) if (!x_ok(sg[r], w.tag)) { found = sg[r].y; what = 2; }
#pragma unroll
      for (int u = NZ - 1; u >= 0; --u) {
        if (NVA > 0 && !x_ok(a0[u], w.tag)) { found = a0[u].y; what = 300 + u + (int)(ai[u] >> 4) * 1000; }
        if (NVA > 1 && !x_ok(a1[u], w.tag)) { found = a1[u].y; what = 500 + u + (int)(ai[u] >> 4) * 1000; }
        if (NVT > 0 && !x_ok(c0[u], w.tag)) { found = c0[u].y; what = 400 + u + (int)(ti[u] >> 4) * 1000; }
        if (NVT > 1 && !x_ok(c1[u], w.tag)) { found = c1[u].y; what = 600 + u + (int)(ti[u] >> 4) * 1000; }
      }
      const unsigned long long miss = __ballot(ok ? 0 : 1);
      const int src = miss ? (int)__builtin_ctzll(miss) : 0;
      found = (unsigned)__builtin_amdgcn_readlane((int)found, src); what = __builtin_amdgcn_readlane(what, src);
      if (!x_spin(w, spins, found, what)) return;
      __builtin_amdgcn_s_sleep(8); // ~0.25 us
    }
  }
}

// products of one slice -> LDS, then every owned row adds its entries in entry order
template <int NZ, int R>
__device__ __forceinline__ void x_rows(double *prod, const double (&mat)[NZ], const double (&vec)[NZ], int cnt, const int (&s)[R], const int (&e)[R], double (&out)[R]) {
  __syncthreads(); // the readers of the previous use are done
#pragma unroll
  for (int u = 0; u < NZ; ++u)
    if (u < cnt) prod[threadIdx.x + u * XTB] = mat[u] * vec[u];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; ++q) {
    double acc = 0.0;
    for (int k = s[q]; k < e[q]; ++k) acc += prod[k];
    out[q] = acc;
  }
}

// workgroup partial sums of K values -> granules 0..K-1 of this rank's scalar area; `red` = K * XWAVES doubles of LDS
template <int K>
__device__ __forceinline__ void x_pub_scalars(double (&v)[K], double *red, xrsrc sc, unsigned sc_off /* byte offset of this rank's granules */, unsigned tag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = x_wave_sum63(v[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * XWAVES + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < x_rounds(K); ++r) { // wavefront k adds the XWAVES partials of scalar k
    const int k = wave + r * XWAVES;
    if (k < K) {
      const double s = x_wave_sum63(lane < XWAVES ? red[k * XWAVES + lane] : 0.0);
      if (lane == 63) x_put(sc, sc_off + (unsigned)k * 16u, s, tag);
    }
  }
}
// x_wait left scalar k of rank `lane` in wavefront k (+ XWAVES r): totals in a fixed order, the same bits in every workgroup
template <int K>
__device__ __forceinline__ void x_sum_scalars(const double (&sv)[x_rounds(K)], double *tot, double (&out)[K]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < x_rounds(K); ++r) {
    const int k = wave + r * XWAVES;
    if (k < K) {
      const double s = x_wave_sum63(sv[r]);
      if (lane == 63) tot[k] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = x_uni(tot[k]);
}

// calc_residuals (abip.c:458-535) + has_converged (1613-1641) on finalised sums; o = the out[] array, ac = avg_criterion of this iteration
__device__ __forceinline__ int x_converged(const double *o, int ac, const XcdFinal &f, long k) {
  const double ut = ac ? o[82] : o[80], vt = ac ? o[83] : o[81];
  const double tau = fabs(ut);
  (void)vt;
  const double nmpr = sqrt(ac ? o[S_RPA] : o[S_RP]), nmax = sqrt(ac ? o[S_NAXA] : o[S_NAX]);
  const double nmdr = sqrt(ac ? o[S_RDA] : o[S_RD]), nmaty = sqrt(ac ? o[S_NATYA] : o[S_NATY]);
  const double by_t = (ac ? o[S_BYA] : o[S_BY]) / f.den, cx_t = (ac ? o[S_CXA] : o[S_CX]) / f.den;
  const double nan_ = __longlong_as_double(0x7ff8000000000000LL);
  const double res_infeas = by_t > 0 ? f.nm_b * nmaty / by_t : nan_;
  const double res_unbdd = cx_t < 0 ? f.nm_c * nmax / -cx_t : nan_;
  auto sdiv = [](double x, double y) { return y < 1E-18 ? x / 1E-18 : x / y; };
  const double by = sdiv(by_t, tau), cx = sdiv(cx_t, tau);
  const double res_pri = sdiv(nmpr / (1 + f.nm_b), tau), res_dual = sdiv(nmdr / (1 + f.nm_c), tau);
  const double rel_gap = fabs(cx - by) / (1 + fabs(cx) + fabs(by));
  if (res_pri < f.eps && (res_dual < f.eps || f.pfeasopt) && rel_gap < f.eps) return 1;
  if (res_unbdd < f.eps && f.ipm_pos && k > 0) return 1;
  if (res_infeas < f.eps && f.ipm_pos && k > 0) return 1;
  return 0;
}

// compute_avg + statistics of one element (dev_kernels.h: avg_and_stats_y / _x, prox_x) with 32-bit unsigned indices
__device__ __forceinline__ void xs_y(const UpdArgs &a, unsigned i, double un, double vn, Stat &st) {
  const double us = x_at(a.u_sum, i) + un, vs = x_at(a.v_sum, i) + vn; // compute_avg, abip.c:649-656
  x_at(a.u_sum, i) = us; x_at(a.v_sum, i) = vs;
  const double ua = us / a.dom, va = vs / a.dom;
  x_at(a.u_avgc, i) = ua; x_at(a.v_avgc, i) = va;
  const double bi = x_at(a.b, i);
  st.wg += a.rho * (un + vn) * x_at(a.g, i);
  st.nu += un * un; st.nv += vn * vn; st.by += bi * un;
  if (a.avg_stats) { st.nua += ua * ua; st.nva += va * va; st.bya += bi * ua; }
}
__device__ __forceinline__ void xs_x(const UpdArgs &a, unsigned q /* MP + j */, unsigned j, bool tail, double un, double vn, Stat &st) {
  const double us = x_at(a.u_sum, q) + un, vs = x_at(a.v_sum, q) + vn;
  x_at(a.u_sum, q) = us; x_at(a.v_sum, q) = vs;
  const double ua = us / a.dom, va = vs / a.dom;
  x_at(a.u_avgc, q) = ua; x_at(a.v_avgc, q) = va;
  st.nu += a.xw * (un * un); st.nv += a.xw * (vn * vn);
  if (a.avg_stats) { st.nua += a.xw * (ua * ua); st.nva += a.xw * (va * va); }
  if (!tail) {
    const double cj = x_at(a.c, j);
    st.wg += a.xw * ((un + vn) * x_at(a.g, q));
    st.cx += a.xw * (cj * un);
    if (a.avg_stats) st.cxa += a.xw * (cj * ua);
  }
}
__device__ __forceinline__ void x_prox(const UpdArgs &a, double uo, double vo, double utq, double &un, double &vn) {
  if (!a.half_update) {
    const double t = a.alpha * utq + (1.0 - a.alpha) * uo - vo; // abip.c:738
    const double hlf = t / 2;
    un = hlf + sqrt(hlf * hlf + a.mu_over_beta);                // abip.c:743-744
    vn = vo + (un - a.alpha * utq - (1.0 - a.alpha) * uo);      // abip.c:580
  } else {
    double vh = vo + 0.5 * (uo - utq);                          // abip.c:675
    const double hlf = (utq - vh) / 2;                          // abip.c:695,700
    un = hlf + sqrt(hlf * hlf + a.mu_over_beta);
    vn = vh + (un - utq);                                       // abip.c:707
  }
}

template <int NZ, int RM, int RN, bool PCG>
__global__ __launch_bounds__(XTB) void k_lp_xcd(const XcdArgs a) {
  extern __shared__ double xl[];
  double *prod = xl;                 // NZ * XTB
  double *tot = prod + NZ * XTB;     // XKS
  double *red = tot + XKS;           // XWAVES * XKS
  double *outs = red + XWAVES * XKS; // 96: the finalised sums (every workgroup holds the same)
  double *wv = outs + 96;            // direct: the whole right-hand side w (m_pad), then the products of the owned rows (RM * XTB)
  __shared__ int s_rank;
  const unsigned t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  if (t == 0) {
    int r = -1;
    if (x_xcc_id() == 0) {
      r = (int)(atomicAdd(a.tickets, 1u) - a.ticket_base);
      if (r < 0 || r >= a.G) r = -1;
    }
    s_rank = r;
  }
  __syncthreads();
  const int rank = __builtin_amdgcn_readfirstlane(s_rank); // (uniform, and the compiler may know it: everything derived from it lives in scalar registers)
  if (rank < 0) return;
  const int G = a.G;
  const unsigned MP = (unsigned)a.MP;
  const unsigned m0 = (unsigned)a.mb[rank], m1 = (unsigned)a.mb[rank + 1], n0 = (unsigned)a.nb[rank], n1 = (unsigned)a.nb[rank + 1];
  const int ka0 = a.Ap[m0], ka1 = a.Ap[m1], kt0 = a.Tp[n0], kt1 = a.Tp[n1];
  // ---- the two matrix slices, in registers for the whole launch (indices as byte offsets of the granules they gather) ----
  double ax[NZ], tx[NZ];
  unsigned ai[NZ], ti[NZ];
  int na = 0, nt = 0;
#pragma unroll
  for (int u = 0; u < NZ; ++u) {
    const unsigned k = t + u * XTB;
    ax[u] = 0.0; tx[u] = 0.0; ai[u] = 0; ti[u] = 0;
    if (ka0 + (int)k < ka1) { ax[u] = x_at(a.Ax, ka0 + k); ai[u] = 16u * (unsigned)x_at(a.Ai, ka0 + k); na = u + 1; }
    if (kt0 + (int)k < kt1) { tx[u] = x_at(a.Tx, kt0 + k); ti[u] = 16u * (unsigned)x_at(a.Ti, kt0 + k); nt = u + 1; }
  }
#pragma unroll
  for (int u = 0; u < NZ; ++u) { // padding: gather an entry that is certainly published whenever the vector is (the product is never used)
    if (u >= na) ai[u] = na > 0 ? ai[0] : 0u;
    if (u >= nt) ti[u] = nt > 0 ? ti[0] : 0u;
  }
  int sa[RM], ea[RM], st[RN], et[RN];
#pragma unroll
  for (int q = 0; q < RM; ++q) { const unsigned i = m0 + t + q * XTB; sa[q] = 0; ea[q] = 0; if (i < m1) { sa[q] = x_at(a.Ap, i) - ka0; ea[q] = x_at(a.Ap, i + 1) - ka0; } }
#pragma unroll
  for (int q = 0; q < RN; ++q) { const unsigned j = n0 + t + q * XTB; st[q] = 0; et[q] = 0; if (j < n1) { st[q] = x_at(a.Tp, j) - kt0; et[q] = x_at(a.Tp, j + 1) - kt0; } }

  UpdArgs up = a.upd;
  up.fuse_avg = 1; up.xw = 1.0; up.gs = nullptr;
  const double rho = up.rho;
  const unsigned tail = MP + (unsigned)a.n;
  unsigned tag = a.tag0;
  XWait w; w.xstat = a.xstat; w.dead = false; w.site = 0; w.rank = rank;
  xrsrc pn0, pn1, pm0, pm1, psc; // the exchange areas of the parity in use
  const unsigned sc_off = (unsigned)rank * XKS * 16u;
  auto open = [&]() { // next exchange: tag and the buffers of its parity
    tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(tag + 1u)); // (uniform by construction; the loops' give-up exits hide that from the compiler)
    const size_t par = tag & 1u;
    w.tag = tag;
    pn0 = x_rsrc(a.xn0 + par * a.n_pad, 16u * (unsigned)a.n_pad); pn1 = x_rsrc(a.xn1 + par * a.n_pad, 16u * (unsigned)a.n_pad);
    pm0 = x_rsrc(a.xm0 + par * a.m_pad, 16u * (unsigned)a.m_pad); pm1 = x_rsrc(a.xm1 + par * a.m_pad, 16u * (unsigned)a.m_pad);
    psc = x_rsrc(a.sc + par * (size_t)(XG * XKS), 16u * XG * XKS);
    w.n0 = pn0; w.n1 = pn1; w.m0 = pm0; w.m1 = pm1; w.sc = psc;
    if (t == 0) {
      x_put(psc, sc_off + XHB * 16u, 0.0, tag); // heartbeat: everything this rank read of the exchange before last is in its registers
      a.xstat[8 + 2 * rank] = (int)tag; a.xstat[9 + 2 * rank] = w.site; // (post-mortem: where every rank was when a wait gave up)
    }
  };
  double dumA[1][NZ], dumT[1][NZ], dumS[1];

  // ---- prologue: S_WG and the tau entries for the first right-hand side; A'u_y, the warm start's product (PCG) ----
  double aty[RN]; // (A'u_y)_j of the owned columns: left by the stopping test, used by the next solve's set-up
  double wg, u_tau, v_tau;
#pragma unroll
  for (int q = 0; q < RN; ++q) aty[q] = 0.0;
  {
    open();
    double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + t + q * XTB;
      if (i < m1) { const double uy = x_at(up.u, i); p[0] += rho * (uy + x_at(up.v, i)) * x_at(up.g, i); if (PCG) x_put(pm0, i * 16u, uy, tag); }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j = n0 + t + q * XTB;
      if (j < n1) p[0] += (x_at(up.u, MP + j) + x_at(up.v, MP + j)) * x_at(up.g, MP + j);
    }
    if (rank == 0 && t == 0) { p[1] = x_at(up.u, tail); p[2] = x_at(up.v, tail); }
    x_pub_scalars<3>(p, red, psc, sc_off, tag);
    double vt[1][NZ], sv[x_rounds(3)];
    w.site = 1;
    if (PCG) x_wait<NZ, 0, 1, 3>(w, G, ai, ti, dumA, vt, sv);
    else x_wait<NZ, 0, 0, 3>(w, G, ai, ti, dumA, dumT, sv);
    if (w.dead) return;
    if (PCG) x_rows<NZ, RN>(prod, tx, vt[0], nt, st, et, aty);
    double s3[3];
    x_sum_scalars<3>(sv, tot, s3);
    wg = s3[0]; u_tau = s3[1]; v_tau = s3[2];
  }

  if (t < 96) outs[t] = a.ctl->out[t]; // slots this launch does not refresh keep what the last finalize left (as on the launch path)
  int ran = 0, halt = 0, last_cg = 0;
  long cg_total = 0;
  double metric = 0.0;
  int avg_crit = 0;
  for (int it = 0; it < a.max_iters; ++it) {
    // (an index the optimiser cannot see through: otherwise it hoists the element addresses of an iteration out of this loop and spills them)
    unsigned tb = t; asm volatile("" : "+v"(tb));
    const long j = a.j0 + it;
    const bool avg_stats = ((j + 1) % 10 == 0); // abip.c:2000
    up.dom = (double)(j + 1);
    up.avg_stats = avg_stats ? 1 : 0;
    // ---- right-hand side (k_rhs, abip.c:552-558) ----
    const double tsum = u_tau + v_tau;
    const double coef = (wg - tsum * a.g_th) / (a.g_th + 1.0);
    double rhs_y[RM], rhs_x[RN];
    double bn[1] = {0.0};
    open();
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      rhs_y[q] = 0.0;
      if (i < m1) {
        const double hi = x_at(a.h, i);
        double r = (x_at(up.u, i) + x_at(up.v, i)) * rho;
        r += -tsum * hi;
        r += -coef * hi;
        rhs_y[q] = r;
        bn[0] += r * r;
      }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j2 = n0 + tb + q * XTB;
      rhs_x[q] = 0.0;
      if (j2 < n1) {
        const double hj = x_at(a.h, MP + j2);
        double r = x_at(up.u, MP + j2) + x_at(up.v, MP + j2);
        r += -tsum * hj;
        r += -coef * hj;
        rhs_x[q] = -r;
        x_put(pn0, j2 * 16u, -r, tag);
        if (PCG) x_put(pn1, j2 * 16u, aty[q], tag);
      }
    }
    double y[RM]; // the y block of the solution
    if (PCG) {
      // ---- PCG set-up (k_cg_init_A, indirect.c:345-365, 415) ----
      x_pub_scalars<1>(bn, red, psc, sc_off, tag);
      double va[2][NZ], sv1[1];
      w.site = 3; x_wait<NZ, 2, 0, 1>(w, G, ai, ti, va, dumT, sv1);
      if (w.dead) return;
      double sA[RM], sB[RM];
      x_rows<NZ, RM>(prod, ax, va[0], na, sa, ea, sA);
      x_rows<NZ, RM>(prod, ax, va[1], na, sa, ea, sB);
      double bnS[1];
      x_sum_scalars<1>(sv1, tot, bnS);
      double tol = sqrt(bnS[0]) * a.tolf[it]; // indirect.c:406-409, 418
      tol = fmax(tol, 1e-7);
      tol = fmax(tol, 1e-9);
      double cr[RM], cz[RM], cp[RM], cx[RM], Mj[RM], ctmp[RN];
      double rz[2] = {0.0, 0.0};
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        cr[q] = 0.0; cz[q] = 0.0; cp[q] = 0.0; cx[q] = 0.0; Mj[q] = 0.0;
        if (i < m1) {
          const double si = x_at(up.u, i);
          const double b = rhs_y[q] + sA[q];
          const double ri = b - (sB[q] + rho * si);
          Mj[q] = x_at(a.Mjac, i);
          const double zi = ri * Mj[q];
          cr[q] = ri; cz[q] = zi; cp[q] = zi; cx[q] = si;
          rz[0] += ri * ri; rz[1] += zi * ri;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) ctmp[q] = 0.0;
      int cgit = 0;
      double zr_prev = 0.0;
      for (;;) {
        // ---- z and (|r|^2, z'r) out; convergence test; tmp = A'z + beta tmp (k_cg_spmv_At) ----
        open();
#pragma unroll
        for (int q = 0; q < RM; ++q) { const unsigned i = m0 + t + q * XTB; if (i < m1) x_put(pm0, i * 16u, cz[q], tag); }
        x_pub_scalars<2>(rz, red, psc, sc_off, tag);
        double vt[1][NZ], sv2[1];
        w.site = 4; x_wait<NZ, 0, 1, 2>(w, G, ai, ti, dumA, vt, sv2);
        if (w.dead) return;
        double tq[RN];
        x_rows<NZ, RN>(prod, tx, vt[0], nt, st, et, tq);
        double rzS[2];
        x_sum_scalars<2>(sv2, tot, rzS);
        const double nr = sqrt(rzS[0]);
        bool done = (cgit == 0) ? (nr < fmin(tol, 1e-18)) : (nr < tol); // indirect.c:359, 375
        if (cgit >= a.cg_max_its) done = true;                          // indirect.c:368
        if (done) break;
        const double beta = (cgit == 0) ? 0.0 : rzS[1] / zr_prev;
        zr_prev = rzS[1];
        double tp[2] = {0.0, 0.0};
        open();
#pragma unroll
        for (int q = 0; q < RN; ++q) {
          const unsigned j2 = n0 + t + q * XTB;
          if (j2 < n1) {
            const double v = (cgit == 0) ? tq[q] : tq[q] + beta * ctmp[q];
            ctmp[q] = v;
            tp[0] += v * v;
            x_put(pn0, j2 * 16u, v, tag);
          }
        }
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          const unsigned i = m0 + t + q * XTB;
          if (i < m1) { const double pn = cz[q] + beta * cp[q]; cp[q] = pn; tp[1] += pn * pn; }
        }
        // ---- tmp and (|A'p|^2, |p|^2) out; Gp = A tmp + rho p; alpha; x, r, z (k_cg_spmv_A + k_cg_update) ----
        x_pub_scalars<2>(tp, red, psc, sc_off, tag);
        double va1[1][NZ];
        w.site = 5; x_wait<NZ, 1, 0, 2>(w, G, ai, ti, va1, dumT, sv2);
        if (w.dead) return;
        double gq[RM];
        x_rows<NZ, RM>(prod, ax, va1[0], na, sa, ea, gq);
        double tpS[2];
        x_sum_scalars<2>(sv2, tot, tpS);
        const double pGp = rho * tpS[1] + tpS[0];
        const double alpha = rzS[1] / pGp;
        rz[0] = 0.0; rz[1] = 0.0;
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          const unsigned i = m0 + t + q * XTB;
          if (i < m1) {
            const double gp = gq[q] + rho * cp[q];
            cx[q] += alpha * cp[q];
            const double ri = cr[q] - alpha * gp;
            const double zi = ri * Mj[q];
            cr[q] = ri; cz[q] = zi;
            rz[0] += ri * ri; rz[1] += zi * ri;
          }
        }
        ++cgit;
      }
      last_cg = cgit; cg_total += cgit;
#pragma unroll
      for (int q = 0; q < RM; ++q) y[q] = cx[q];
    } else {
      // ---- direct: w = rhs_y + A rhs_x; y = inv(rho I + A A') w; (x follows below) ----
      double va1[1][NZ];
      w.site = 6; x_wait<NZ, 1, 0, 0>(w, G, ai, ti, va1, dumT, dumS);
      if (w.dead) return;
      double sA[RM];
      x_rows<NZ, RM>(prod, ax, va1[0], na, sa, ea, sA);
      open();
#pragma unroll
      for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; if (i < m1) x_put(pm0, i * 16u, rhs_y[q] + sA[q], tag); }
      w.site = 12; x_wait<NZ, 0, 0, 0>(w, G, ai, ti, dumA, dumT, dumS); // (the rendez-vous)
      if (w.dead) return;
      { // every workgroup needs the whole w
        int spins = 0;
        for (unsigned i0 = wave * 64u; i0 < (unsigned)a.m; i0 += XTB) { // (wave-uniform trip count: the lanes of a wavefront poll together)
          const unsigned i = i0 + lane;
          u32x4 g; g.y = tag; g.w = tag; g.x = 0; g.z = 0;
          for (;;) {
            asm volatile("" ::: "memory");
            if (i < (unsigned)a.m) g = x_ld(w.m0, i * 16u);
            if (__all(x_ok(g, tag) ? 1 : 0)) break;
            if (!x_spin(w, spins, g.y, 8)) break;
          }
          if (w.dead) break;
          if (i < (unsigned)a.m) wv[i] = x_val(g);
        }
        if (w.dead) return;
      }
      __syncthreads();
      double *yv = wv + a.m_pad;
      for (unsigned r = m0 + wave; r < m1; r += XWAVES) { // one wavefront per owned row
        const double *row = a.Minv + (long)r * a.ldM;
        double acc = 0.0;
        for (unsigned c = lane; c < (unsigned)a.m; c += 64) acc += x_at(row, c) * wv[c];
        acc = x_wave_sum63(acc);
        if (lane == 63) yv[r - m0] = acc;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; y[q] = (i < m1) ? yv[i - m0] : 0.0; }
    }
    // ---- back-substitution x = A'y - rhs_x (indirect.c:419-420) and u_t'h (abip.c:560) ----
    double dh[1] = {0.0};
    open();
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      if (i < m1) { x_put(pm0, i * 16u, y[q], tag); x_at(up.ut, i) = y[q]; dh[0] += y[q] * x_at(a.h, i); }
    }
    double zx[RN];
    {
      double vt[1][NZ];
      w.site = 7; x_wait<NZ, 0, 1, 0>(w, G, ai, ti, dumA, vt, dumS);
      if (w.dead) return;
      double tq[RN];
      x_rows<NZ, RN>(prod, tx, vt[0], nt, st, et, tq);
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        zx[q] = 0.0;
        if (j2 < n1) { zx[q] = tq[q] - rhs_x[q]; dh[0] += zx[q] * x_at(a.h, MP + j2); }
      }
    }
    open();
    x_pub_scalars<1>(dh, red, psc, sc_off, tag);
    double svd[1];
    w.site = 8; x_wait<NZ, 0, 0, 1>(w, G, ai, ti, dumA, dumT, svd);
    if (w.dead) return;
    double dhS[1];
    x_sum_scalars<1>(svd, tot, dhS);
    // ---- element-wise update (k_admm_update): barrier prox, dual update, running sums, averages, statistics ----
    Stat sst = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double tau4[4] = {0.0, 0.0, 0.0, 0.0};
    open();
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      if (i < m1) {
        double un, vn;
        const double uti = y[q];
        if (!up.half_update) { vn = x_at(up.v, i); un = uti - vn; }
        else { double vh = x_at(up.v, i) + 0.5 * (x_at(up.u, i) - uti); un = uti - vh; vn = vh + (un - uti); }
        x_at(up.u, i) = un; x_at(up.v, i) = vn;
        x_at(up.u_avg, i) += un; x_at(up.v_avg, i) += vn;
        xs_y(up, i, un, vn, sst);
        x_put(pm0, i * 16u, un, tag);
        if (avg_stats) x_put(pm1, i * 16u, x_at(up.u_avgc, i), tag);
      }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j2 = n0 + tb + q * XTB;
      if (j2 < n1) {
        const unsigned qq = MP + j2;
        x_at(up.ut, qq) = zx[q];
        double un, vn;
        x_prox(up, x_at(up.u, qq), x_at(up.v, qq), zx[q], un, vn);
        x_at(up.u, qq) = un; x_at(up.v, qq) = vn;
        x_at(up.u_avg, qq) += un; x_at(up.v_avg, qq) += vn;
        xs_x(up, qq, j2, false, un, vn, sst);
        x_put(pn0, j2 * 16u, un, tag);
        if (avg_stats) x_put(pn1, j2 * 16u, x_at(up.u_avgc, qq), tag);
      }
    }
    if (rank == 0 && t == 0) { // the tau / kappa entry
      const double utq = tsum + dhS[0];
      x_at(up.ut, tail) = utq;
      double un, vn;
      x_prox(up, x_at(up.u, tail), x_at(up.v, tail), utq, un, vn);
      x_at(up.u, tail) = un; x_at(up.v, tail) = vn;
      x_at(up.u_avg, tail) += un; x_at(up.v_avg, tail) += vn;
      xs_x(up, tail, (unsigned)a.n, true, un, vn, sst);
      tau4[0] = un; tau4[1] = vn; tau4[2] = x_at(up.u_avgc, tail); tau4[3] = x_at(up.v_avgc, tail);
    }
    double s13[13] = {sst.wg, sst.nu, sst.nv, sst.cx, sst.by, sst.nua, sst.nva, sst.cxa, sst.bya, tau4[0], tau4[1], tau4[2], tau4[3]};
    x_pub_scalars<13>(s13, red, psc, sc_off, tag);
    // ---- stopping-test products (k_q_both): A u_x and A'u_y, residual sums; A'u_y is also the next solve's warm-start product ----
    double q6[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double S13[13];
    {
      double va1[1][NZ], vt[1][NZ], sv13[x_rounds(13)];
      w.site = 9; x_wait<NZ, 1, 1, 13>(w, G, ai, ti, va1, vt, sv13);
      if (w.dead) return;
      double pri[RM];
      x_rows<NZ, RM>(prod, ax, va1[0], na, sa, ea, pri);
      x_rows<NZ, RN>(prod, tx, vt[0], nt, st, et, aty);
      x_sum_scalars<13>(sv13, tot, S13);
      const double tau = S13[9];
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double e = pri[q] - x_at(up.b, i) * tau;
          double sc = a.wD ? x_at(a.wD, i) : 1.0;
          sc = sc * sc;
          q6[0] += e * e; q6[1] += (e * e) * sc; q6[2] += (pri[q] * pri[q]) * sc;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const double drj = aty[q] + x_at(up.v, MP + j2), e = drj - x_at(up.c, j2) * tau;
          double sc = a.wE ? x_at(a.wE, j2) : 1.0;
          sc = sc * sc;
          q6[3] += e * e; q6[4] += (e * e) * sc; q6[5] += (drj * drj) * sc;
        }
      }
    }
    if (avg_stats) { // the same on the averaged iterate (its entries went out with the same exchange)
      double va1[1][NZ], vt[1][NZ];
      XWait w2 = w; w2.n0 = w.n1; w2.m0 = w.m1; w2.site = 20;
      w.site = 10; x_wait<NZ, 1, 1, 0>(w2, G, ai, ti, va1, vt, dumS);
      if (w2.dead) return;
      double pri[RM], atya[RN];
      x_rows<NZ, RM>(prod, ax, va1[0], na, sa, ea, pri);
      x_rows<NZ, RN>(prod, tx, vt[0], nt, st, et, atya);
      const double tau = S13[11];
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double e = pri[q] - x_at(up.b, i) * tau;
          double sc = a.wD ? x_at(a.wD, i) : 1.0;
          sc = sc * sc;
          q6[6] += e * e; q6[7] += (e * e) * sc; q6[8] += (pri[q] * pri[q]) * sc;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const double drj = atya[q] + x_at(up.v_avgc, MP + j2), e = drj - x_at(up.c, j2) * tau;
          double sc = a.wE ? x_at(a.wE, j2) : 1.0;
          sc = sc * sc;
          q6[9] += e * e; q6[10] += (e * e) * sc; q6[11] += (drj * drj) * sc;
        }
      }
    }
    open();
    x_pub_scalars<12>(q6, red, psc, sc_off, tag);
    double sv12[x_rounds(12)];
    w.site = 11; x_wait<NZ, 0, 0, 12>(w, G, ai, ti, dumA, dumT, sv12);
    if (w.dead) return;
    double Q[12];
    x_sum_scalars<12>(sv12, tot, Q);
    // ---- finalize (d_finalize): the inner-loop exit test, iterate_Q_norm_resd abip.c:2027-2050 and the comparison of abip.c:2173 ----
    wg = S13[0]; u_tau = S13[9]; v_tau = S13[10];
    if (t == 0) {
      double *o = outs;
      o[S_WG] = S13[0]; o[S_NU] = S13[1]; o[S_NV] = S13[2]; o[S_CX] = S13[3]; o[S_BY] = S13[4];
      o[S_QP] = Q[0]; o[S_RP] = Q[1]; o[S_NAX] = Q[2]; o[S_QD] = Q[3]; o[S_RD] = Q[4]; o[S_NATY] = Q[5];
      o[80] = S13[9]; o[81] = S13[10]; o[82] = S13[11]; o[83] = S13[12];
      if (avg_stats) {
        o[S_NUA] = S13[5]; o[S_NVA] = S13[6]; o[S_CXA] = S13[7]; o[S_BYA] = S13[8];
        o[S_QPA] = Q[6]; o[S_RPA] = Q[7]; o[S_NAXA] = Q[8]; o[S_QDA] = Q[9]; o[S_RDA] = Q[10]; o[S_NATYA] = Q[11];
      }
    }
    {
      double Qres = Q[0] + Q[3];
      const double gap = S13[4] - S13[3] - S13[10];
      Qres = __dadd_rn(Qres, __dmul_rn(gap, gap));
      const double norm = 1 + sqrt(S13[1] + S13[2]);
      double Qres_avg = a.sentinel, norm_avg = 1;
      if (avg_stats) {
        Qres_avg = Q[6] + Q[9];
        const double gap_a = S13[8] - S13[7] - S13[12];
        Qres_avg = __dadd_rn(Qres_avg, __dmul_rn(gap_a, gap_a));
        norm_avg = 1 + sqrt(S13[5] + S13[6]);
      }
      const double ma = sqrt(Qres_avg) / norm_avg, mc = sqrt(Qres) / norm;
      avg_crit = ma < mc ? 1 : 0;
      metric = avg_crit ? ma : mc;
    }
    ++ran;
    if (metric < a.thr) { halt = 1; break; }
    if (a.fc.on) {
      __syncthreads(); // outs complete
      const long k = a.fc.k0 + ran;
      if (x_converged(outs, avg_crit, a.fc, k) || k + 1 >= a.fc.max_admm) { halt = 2; break; }
    }
  }
  __syncthreads();
  if (rank == 0 && t < 96 && ran > 0) a.ctl->out[t] = outs[t];
  if (rank == 0 && t == 0) {
    Ctl *c = a.ctl;
    c->metric = metric; c->avg_crit = avg_crit; c->it_count = c->it_count + ran; c->halt = halt;
    c->cg_it = last_cg; c->cg_done = 1;
    c->xcd_cg_total = cg_total;
    a.xstat[1] = (int)(tag - a.tag0);
  }
}

} // namespace abip
