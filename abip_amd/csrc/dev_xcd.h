// dev_xcd.h -- the whole inner ADMM loop of a cache-resident LP as ONE persistent launch on ONE XCD (gfx950).
//
// Why.  On Netlib / pds-class LPs a kernel of the launch path (dev_kernels.h) does 2-5 us of work and the boundary to the next one costs
// 2-3 us more; a PCG iteration is three such kernels, an ADMM iteration of the direct back-end six.  Here G = 32 workgroups -- the 32 CUs of
// one XCD, so that ONE L2 is the point of coherence for every hand-over -- run `max_iters` ADMM iterations (abip.c:2131-2215) in one launch:
//
//   * the rows of A and of A' are cut into G contiguous slices (balanced by non-zeros and rows); workgroup g OWNS rows [mb[g], mb[g+1]) of
//     the m-space and [nb[g], nb[g+1]) of the n-space: it alone reads and writes those entries of u, v, the running sums, the PCG vectors
//     (the PCG vectors live in registers for the whole solve);
//   * a product needs the whole operand vector.  One EXCHANGE: the owners store their entries into an exchange area (plain stores: the lines
//     stay in this XCD's L2), wait until the L2 has acknowledged them (s_waitcnt vmcnt(0)), meet at a workgroup barrier, and then publish
//     the workgroup's partial sums of the exchange as 16-byte granules {lo32, tag, hi32, tag} -- those granules are the FLAG: a consumer that
//     sees rank r's granule with the tag of this exchange knows r's entries are in the L2.  Wavefront k of every workgroup polls scalar k of
//     all G ranks (L1-bypassing loads, 32 lanes, a short sleep between rounds: no flood of the L2's request queues), adds them in rank
//     order -- every workgroup gets bit-identical sums and takes the same decisions -- and after one more barrier all threads gather the
//     entries their non-zeros name with L1-bypassing loads.  No atomics, no agent-scope fences (a release / acquire pair costs 6 us here,
//     tools/xcd_probe.hip, profiles/r03a_*), no grid barrier;
//   * the tag is the running number of the exchange and the areas alternate with its parity.  Every exchange is a rendez-vous of all ranks
//     (everybody waits for everybody's granules), so a workgroup is at most one exchange ahead of the slowest one and an area is never
//     overwritten while somebody may still read it.
//
// (First built with the tag on every entry and all threads polling their own entries: 32 k threads re-asking flooded the L2 queues and the
// rank everybody waited for -- its loads, its spills, its stores -- queued behind them: milliseconds per exchange under load.)
//
// Placement.  The launch has 256 workgroups with > 80 KB of LDS each: one per CU, hence exactly 32 on every XCD.  The ones whose
// HW_REG_XCC_ID is not 0 return at once; the others draw a ticket (= rank).  Every poll is bounded and gives up through xstat[0]
// (the host then fails loudly), so a placement that breaks the assumption cannot hang the device.
//
// Arithmetic = the launch path's (dev_kernels.h) entry for entry; sums are taken in a different order (per row: entry order; per
// reduction: lane, wavefront, rank order), and p'Gp is formed as rho |p|^2 + |A'p|^2 (the sharded path's identity, saving one
// exchange per PCG iteration).
#pragma once
#include "dev_kernels.h"
#include "lp_scalars.h"

namespace abip {

#ifndef XCD_TB
#define XCD_TB 512
#endif
constexpr int XTB = XCD_TB;        // threads per workgroup: 8 wavefronts, 2 per SIMD -> 256 VGPRs each.  Round 3 ran 768 (168 VGPRs, no scratch then); with the outer
                                   // iterations and the search inside the kernel 768 spills (13 scratch loads per iteration of the direct variant), 512 does not:
                                   // c2 317.7 -> 276.5 ms, c3 6.03 -> 5.88 s per solve, same iteration and PCG counts (profiles/r04f_threads_per_workgroup.txt)
constexpr int XWAVES = XTB / 64;
constexpr int XG = 256;            // most workgroups taking part: the CUs of one XCD (32), or of 2, 4, 8 XCDs
constexpr int XQ = XG / 64;        // flags a polling lane looks after
constexpr int XKS = 16;            // granule slots per workgroup per exchange: sums 0..11 (wavefront k handles sum k), slots 12..15 the tau entries
// Slot k of rank r is granule k * XG + r: the granules a polling wavefront reads (slot k of every rank) are contiguous -- four lanes to a 64-byte request, 32
// requests per round of 128 ranks.  (Until round 5 rank r's slots sat side by side, r * XKS + k: one request per lane, and across XCDs the polls were a third of
// the exchange's traffic at the memory side.)
constexpr unsigned XSLOT = 16u * XG; // bytes from a rank's slot k to its slot k + 1
constexpr int XSTAT_N = 1024;       // ints of the status / post-mortem record
#ifndef XCD_SLEEP
#define XCD_SLEEP 2
#endif
constexpr int XSLEEP = XCD_SLEEP;   // s_sleep between two polling rounds of a wavefront (units of 64 clocks)
#ifndef XCD_POLL_DELAY
#define XCD_POLL_DELAY 22
#endif
#ifndef XCD_POLL_DELAY1
#define XCD_POLL_DELAY1 0
#endif
// s_sleep units before a collect's first polling round.  Across XCDs every polling round is 2 x 128 requests per workgroup to the memory side, where the exchange's own
// written-through stores and gathers queue too, and nothing can have arrived for the first ~1.5 us: the rounds that cannot succeed only stand in the way
// (c3: 0 -> 1 870, 12 -> 1 947, 20-24 -> 1 975-2 000, 28 -> 1 935, 32 -> 1 900, 40 -> 1 800 it/s; two rounds in flight half a round trip apart: 1 700 --
// profiles/r05zq_*, r05zr_*, r05zs_*)
constexpr int XPOLL_DELAY = XCD_POLL_DELAY, XPOLL_DELAY1 = XCD_POLL_DELAY1;
constexpr int XSPIN = 1 << 17;     // polling rounds before a wavefront gives up (a round is ~1 us: about a tenth of a second, then the launch path takes over)
constexpr int XCD_LDS_MIN = 84 * 1024; // more than half a CU's LDS: one workgroup per CU
static_assert(XTB % 64 == 0 && XWAVES >= 4 && XWAVES <= 16, "wavefront k handles the sums k, k + XWAVES, ... of an exchange (at most 12 sums; the granule slots 12..15 carry the tau entries)");

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// (XcdFinal, x_sums, x_converged: dev_kernels.h -- the launch path's streamed iterations take the same decisions on the device)

// A launch that SPANS OUTER ITERATIONS (round 4; abip.c:2217-2293 and src/adaptive.c:87-251 on the device).  With `on` the kernel does not return when
// the inner loop's exit test holds: every workgroup takes the reference's scalar decisions itself, from sums that are bit-identical in all of them
// (lp_scalars.h: calc_residuals, has_converged, the mu rule), rescales its own entries (reinitialize_vars), runs the Barzilai-Borwein search -- up to
// adaptive_lookback pairs of look-ahead steps through the same projection code, on the scratch vectors a_* -- zeroes its running sums and starts the next
// inner loop.  It hands the loop back to the host (XcdOut, XR_*) only when the host has something to do: the solve has converged or hit a limit
// (the host extracts the solution), the iteration budget of the call or the wall-clock slice is used up, a restart is due (abip.c:608-627: the launch
// path runs that one iteration), or the outer end needs a rule that is not on the device (the "tedious" table, an averaged iterate that won the stopping test).
struct XcdOuter {
  int on;
  int phase;            // entry point: 0 = iteration (k, j) of the inner loop, 1 = the outer end of abip.c:2217 (Ctl::out holds the iterate's sums)
  int avg_crit;         // stgs->avg_criterion at entry
  int adaptive, lookback, hybrid_mu;
  int bb_reuse;         // 1: a look-ahead whose penalty did not change hands its second step to the next one (ABIP_HIP_BB_REUSE=0: every look-ahead solves twice, as the reference does)
  long i, fre_old;
  double mu, beta, sigma, gamma, dyn_sigma;
  long max_ipm, inner_stopper, restart_thresh, restart_fre;
  long max_steps;       // ADMM iterations this launch may run (abip_hip_step's budget)
  double eps_cor, eps_pen, hybrid_thresh, dyn_sigma_second;
  unsigned long long slice_ticks; // wall-clock budget of the launch in ticks of wall_clock64 (100 MHz): looked at once per outer iteration
  const double *mu_tab; int mu_tab_n; // mu after 1, 2, ... applications of update_barrier_dynamic_2 to the entry mu (pow() stays on the host)
  int log_cap; double *log;           // one row of XLOG_W doubles per outer iteration closed here (print_summary's columns)
  double *a_up, *a_vp, *a_u, *a_v, *a_un, *a_vn; // adaptive.c:13-32 (u_prev, v_prev, u, v, u_next, v_next; ut / ut_next and the three differences never leave registers)
};
constexpr int XLOG_W = 12; // i, k, mu, res_pri, res_dual, rel_gap, c'x / tau-free, b'y, tau, kap, ticks since launch, spare
enum { XM_MAIN = 0, XM_BB1 = 1, XM_BB2 = 2 };                  // whose projection is running: the ADMM iteration's, the first / second look-ahead step's
enum { XS_PROJECT = 0, XS_OUTER_END = 1, XS_OUTER_BEGIN = 2 }; // what the launch does next
enum { CS_MU = 0, CS_BETA, CS_DYNS, CS_OI, CS_FRE, CS_BBPREV, CS_BUPT, CS_BVPT, CS_BUT, CS_BVT, CS_RUT, CS_RVT, CS_ODONE, CS_DYN2, CS_LOGN, CS_BBTOT, CS_BBIT, CS_AV, CS_PHASE, CS_REASON, CS_T0, CS_CGSKIP }; // cs[]: see the kernel
enum { XR_BATCH = 0 /* one batch of iterations, the exit test or the batch's end */, XR_STEPS = 1, XR_RESTART = 2, XR_FINAL = 3 /* the final check holds */,
       XR_HOST_OUTER = 4 /* this outer end is the host's */, XR_SLICE = 5, XR_MAXIPM = 6 };

struct XcdArgs {
  const int *Ap, *Ai; const double *Ax; // CSR of A  (m rows, gathers the n-space)
  const int *Tp, *Ti; const double *Tx; // CSR of A' (n rows, gathers the m-space)
  const int *mb, *nb;                   // G + 1 row bounds each
  int G, m, n, MP;
  UpdArgs upd;                          // u, v, ut, sums, g, b, c, alpha, mu/beta, rho, half_update (dom / avg_stats are set per iteration)
  const double *h, *wD, *wE;
  const double *hAh;                    // h_y + A h_x (m): the y-side weights of u_t'h (see the back-substitution)
  const double *Mjac;                   // PCG: Jacobi preconditioner (m)
  const double *Minv; long ldM;         // direct: inv(rho I + A A'), dense row-major
  int minv_lds_rows;                    // direct: how many of a workgroup's rows of it fit its LDS (the first ones; the others stream from the L2)
  double g_th;
  double *xn0, *xn1, *xm0, *xm1;        // exchange areas: 2 parities x (n_pad | m_pad) doubles
  double *xnv;                          // direct: the new v_x beside the new u_x (2 parities x n_pad): every rank forms the next right-hand side's x entries it gathers itself
  u32x4 *sc;                            // 2 parities x XG x XKS granules: the partial sums = the flags
  int n_pad, m_pad;
  unsigned tag0;
  unsigned *tickets; unsigned ticket_base;
  Ctl *ctl; int *xstat;                 // xstat[0]: a poll gave up; xstat[1]: exchanges used by this launch
  long j0; int max_iters;               // inner index of the first iteration; iterations to run unless the exit test holds earlier
  double thr, sentinel;                 // gamma * mu; Qres_avg when no averaged statistics were taken
  const double *tolf; int cg_max_its;   // PCG: tolerance factor per iteration of this launch (host-computed: indirect.c:406-407; a launch that spans outer iterations forms 0.1 / (k + 1)^2 itself)
  // solve-only mode (the set-up solve, the Barzilai-Borwein look-ahead: solve_lin_sys on a caller's vector, abip.c:552-560 without the prox): K z = srhs in place,
  // warm start = the y block of swarm (or null), u_t'h left in the partial table as the launch path's kernels expect it
  int solve_only; double *srhs; const double *swarm; double *part; int npart;
  int nxcd;                             // XCDs whose workgroups take a ticket: 1 (G = 32), or 2, 4, 8 (G = 32 nxcd, the CROSS variants)
  int desert;                           // fault injection (libabip_hip_hooks.so only): the rank that leaves at once, so that every wait gives up; -1 otherwise
  XcdFinal fc;
  XcdOuter outer;
};

__device__ __forceinline__ unsigned x_xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// Addresses are (scalar base) + (32-bit byte offset) everywhere: the exchange areas through buffer resources (4 scalar registers), the
// owned entries of the l-vectors through x_at -- an element's offset is ONE vector register shared by every array it indexes.
typedef __amdgpu_buffer_rsrc_t xrsrc;
__device__ __forceinline__ xrsrc x_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
// SA: cache policy of the exchange's stores.  0 = plain: the line stays in this XCD's L2, where the other 31 CUs find it.  16 (sc1) = written through,
// for the launch spread over several XCDs: their L2s are not coherent with each other (tools/xcd_probe.hip: st sc1 / ld sc1 is exact across XCDs,
// 1.7 us per exchange against 1.2).
template <int SA>
__device__ __forceinline__ void x_putd(xrsrc r, unsigned off /* bytes */, double v) {
  u32x2 g; g.x = (unsigned)__double2loint(v); g.y = (unsigned)__double2hiint(v);
  __builtin_amdgcn_raw_buffer_store_b64(g, r, (int)off, 0, SA);
}
__device__ __forceinline__ double x_ldd(xrsrc r, unsigned off) { // sc1: past the L1, served by the L2
  const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 16);
  return __hiloint2double((int)g.y, (int)g.x);
}
template <int SA>
__device__ __forceinline__ void x_putg(xrsrc r, unsigned off, double v, unsigned tag) { // a granule: one 16-byte store
  u32x4 g; g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(g, r, (int)off, 0, SA);
}
__device__ __forceinline__ u32x4 x_ldg(xrsrc r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16); }
__device__ __forceinline__ bool x_ok(const u32x4 &g, unsigned tag) { return g.y == tag && g.w == tag; } // both 8-byte halves are of this exchange
__device__ __forceinline__ double x_val(const u32x4 &g) { return __hiloint2double((int)g.z, (int)g.x); }
// (the byte offset is formed in 32 bits: only then can it ride in the instruction's 32-bit offset register beside a scalar base)
template <class T> __device__ __forceinline__ T &x_at(T *base, unsigned i) { return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(unsigned)(i * (unsigned)sizeof(T))); }
template <class T> __device__ __forceinline__ const T &x_at(const T *base, unsigned i) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(unsigned)(i * (unsigned)sizeof(T))); }

// the wavefront's index in its workgroup as the scalar it is (threadIdx.x >> 6 alone is a vector value to the compiler: every test on it an exec-mask dance,
// every address formed from it a vector register for the whole launch)
__device__ __forceinline__ int x_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// the same value in every lane, told to the compiler (scalar registers, uniform branches)
__device__ __forceinline__ double x_uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// wavefront sum on the DPP network (no LDS round trips: __shfl_down is a ds_bpermute per step); fixed order; the total lands in lane 63
// (No "old" value for the lanes a step does not write: with every row written and bound_ctrl set the compiler drops it, and the two row_bcast steps leave the rows
// they skip undefined -- nothing that reaches lane 63 reads them.  Handing 0 in as the old value cost two v_mov per step: 250 of the 1 450 instructions of a PCG
// iteration of the persistent launch.)
template <int CTRL, int ROWS>
__device__ __forceinline__ double x_dpp(double v) {
  int lo, hi;
  if (ROWS == 0xf) {
    lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  } else {
    lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, ROWS, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, ROWS, 0xf, false);
  }
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

// the sum of the first XWAVES (<= 16) lanes' values: the additions x_wave_sum63 makes of them when every other lane holds 0 (its further steps add zeros), without
// those steps; the total lands in each of the first XWAVES lanes (the mirrors pair lane i with lane 7 - i / 15 - i: no direction of rotation to get wrong)
__device__ __forceinline__ double x_sum_first_lanes(double v) {
  v += x_dpp<0xB1, 0xf>(v);                   // quad_perm [1,0,3,2]
  v += x_dpp<0x4E, 0xf>(v);                   // quad_perm [2,3,0,1]
  if (XWAVES > 4) v += x_dpp<0x141, 0xf>(v);  // row_half_mirror
  if (XWAVES > 8) v += x_dpp<0x140, 0xf>(v);  // row_mirror
  return v;
}

struct XWait { // the exchange in flight
  xrsrc n0, n1, m0, m1, sc;
  unsigned tag;
  int *xstat;
  bool dead;
  int site, rank; // which exchange of the iteration this is / whose: recorded when a wavefront gives up
  double *ldead;  // LDS: != 0 once a wavefront of this workgroup has given up (or seen another workgroup do so) -- how the wavefronts that do not poll learn of it
};
// spin bookkeeping of one WAVEFRONT (its lanes poll together): true = keep polling
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
  if (dead) { w.dead = true; if ((threadIdx.x & 63) == 0) *w.ldead = 1.0; return false; }
  return true;
}

// ---- one exchange, in four steps ----------------------------------------------------------------------------------------------------
// (1) the owners x_putd their entries;
// (2) x_publish<K>: this thread's stores are in the L2; workgroup partial sums of K values; their granules go out = the flag of this rank;
// (3) x_collect<K>: wavefront k polls scalar k of every rank, adds them in rank order; barrier: from here on every rank's entries are there;
// (4) x_gather: the entries this thread's non-zeros name.
template <int K, int SA>
__device__ __forceinline__ void x_publish(double (&v)[K], double *red, xrsrc sc, unsigned sc_off /* byte offset of this rank's granule of slot 0 */, unsigned tag) {
  static_assert(K >= 1 && K <= 12, "at most 12 sums per exchange (granule slots 12..15 carry the tau entries)");
  const int lane = threadIdx.x & 63, wave = x_wave();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = x_wave_sum63(v[k]); // (while the stores travel)
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * XWAVES + wave] = v[k];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my entries have been acknowledged by the L2
  __syncthreads(); // ... and so have everybody's of this workgroup
#pragma unroll
  for (int kk = 0; kk < K; kk += XWAVES) { // wavefront k adds the XWAVES partials of scalar k (and k + XWAVES) and raises the flag
    const int k = kk + wave;
    if (k < K) {
      const double s = x_sum_first_lanes(lane < XWAVES ? red[k * XWAVES + lane] : 0.0);
      if (lane == 0) x_putg<SA>(sc, sc_off + (unsigned)k * XSLOT, s, tag);
    }
  }
}
template <int K, bool WIDE /* more than 64 ranks (several XCDs): several loads per polling round */>
__device__ __forceinline__ void x_collect(XWait &w, int G, double *tot, double (&out)[K]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int kk = 0; kk < K; kk += XWAVES) {
  // The scalar this wavefront looks after in this pass.  On one XCD the LAST wavefronts poll: the first ones own the workgroup's elements and may have stores in
  // flight, behind which a poll's load would wait (c2: +1.7 %).  Across XCDs the first ones do, as ever: there a poll is 2 x 64 loads past the L2 per round, and
  // waiting behind the own written-through stores is the cheapest way not to ask before anything can have arrived (the last wavefronts polling at once: c3 -4 %).
  const int wave = kk + (WIDE ? x_wave() : XWAVES - 1 - x_wave());
  if (wave < K) {
    u32x4 g[XQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) { g[q].x = 0; g[q].y = w.tag; g[q].z = 0; g[q].w = w.tag; }
    int spins = 0;
    if (WIDE ? XPOLL_DELAY > 0 : XPOLL_DELAY1 > 0) __builtin_amdgcn_s_sleep(WIDE ? XPOLL_DELAY : XPOLL_DELAY1);
    for (;;) {
      asm volatile("" ::: "memory"); // (the loads are issued anew every round)
      // all the loads of a round first, the tags afterwards: a test behind each load (`ok = ok && ...`) makes the compiler wait for it before the next one goes
      // out -- four memory round trips per polling round on 256 workgroups instead of one
      bool ok = true;
      if (!WIDE) { // one XCD's worth of ranks: one load per round
        if (lane < G) g[0] = x_ldg(w.sc, (unsigned)wave * XSLOT + (unsigned)lane * 16u);
        ok = x_ok(g[0], w.tag);
      } else {
#pragma unroll
        for (int q = 0; q < XQ; ++q)
          if (q * 64 < G && q * 64 + lane < G) g[q] = x_ldg(w.sc, (unsigned)wave * XSLOT + (unsigned)(q * 64 + lane) * 16u);
#pragma unroll
        for (int q = 0; q < XQ; ++q) ok = ok & x_ok(g[q], w.tag); // (slots past G keep the tag they were initialised with)
      }
      if (__all(ok ? 1 : 0)) break;
      const unsigned long long miss = __ballot(ok ? 0 : 1);
      const int src = miss ? (int)__builtin_ctzll(miss) : 0;
      if (!x_spin(w, spins, (unsigned)__builtin_amdgcn_readlane((int)g[0].y, src), 1000 + src)) break;
      __builtin_amdgcn_s_sleep(XSLEEP);
    }
    double s = 0.0; // rank order: 64 ranks at a time, lane order inside
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      if (q * 64 >= G) break;
      s += x_wave_sum63(q * 64 + lane < G ? x_val(g[q]) : 0.0);
    }
    if (lane == 63) tot[wave] = s;
  }
  }
  __syncthreads();
  if (*w.ldead != 0.0) w.dead = true; // (a polling wavefront gave up -- it, or whoever it saw in xstat: everybody leaves.  Read from the LDS: the global flag costs an L2 round trip per exchange)
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = x_uni(tot[k]);
}
// An exchange that carries entries but no sum: the flag alone.  The same rendez-vous (stores acknowledged, barrier, one granule per rank; wavefront 0 polls
// them all; barrier) without the partial sums around it.
template <int SA>
__device__ __forceinline__ void x_flag(xrsrc sc, unsigned sc_off, unsigned tag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) x_putg<SA>(sc, sc_off, 0.0, tag);
}
template <bool WIDE>
__device__ __forceinline__ void x_wait(XWait &w, int G) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    int spins = 0;
    for (;;) {
      asm volatile("" ::: "memory"); // (the loads are issued anew every round)
      u32x4 g[XQ];
#pragma unroll
      for (int q = 0; q < XQ; ++q) { g[q].x = 0; g[q].y = w.tag; g[q].z = 0; g[q].w = w.tag; }
      bool ok = true;
      if (!WIDE) {
        if (lane < G) g[0] = x_ldg(w.sc, (unsigned)lane * 16u);
        ok = x_ok(g[0], w.tag);
      } else {
#pragma unroll
        for (int q = 0; q < XQ; ++q) // (all the loads of a round first: see x_collect)
          if (q * 64 < G && q * 64 + lane < G) g[q] = x_ldg(w.sc, (unsigned)(q * 64 + lane) * 16u);
#pragma unroll
        for (int q = 0; q < XQ; ++q) ok = ok & x_ok(g[q], w.tag);
      }
      const unsigned seen = g[0].y;
      if (__all(ok ? 1 : 0)) break;
      const unsigned long long miss = __ballot(ok ? 0 : 1);
      const int src = miss ? (int)__builtin_ctzll(miss) : 0;
      if (!x_spin(w, spins, (unsigned)__builtin_amdgcn_readlane((int)seen, src), 1000 + src)) break;
      __builtin_amdgcn_s_sleep(XSLEEP);
    }
  }
  __syncthreads();
  if (*w.ldead != 0.0) w.dead = true; // (see x_collect)
}
template <int NZ>
__device__ __forceinline__ void x_gather(xrsrc r, const unsigned (&idx)[NZ], double (&v)[NZ]) {
#pragma unroll
  for (int u = 0; u < NZ; ++u) v[u] = x_ldd(r, idx[u]);
}

// products of one slice -> LDS, then every owned row adds its entries in entry order
template <int NZ, int R>
__device__ __forceinline__ void x_rows(double *prod2 /* two buffers of NZ * XTB */, int &flip, const double (&mat)[NZ], const double (&vec)[NZ], int cnt, const int (&s)[R], const int (&e)[R], double (&out)[R]) {
  // the two buffers alternate: whoever writes buffer X again has passed the barrier of the call in between, which every thread reaches only after
  // it has finished reading X -- one barrier per product instead of two
  double *prod = prod2 + (flip ? NZ * XTB : 0);
  flip ^= 1;
#pragma unroll
  for (int u = 0; u < NZ; ++u)
    if (u < cnt) prod[threadIdx.x + u * XTB] = mat[u] * vec[u];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; ++q) { // entry order; four LDS reads in flight at a time
    double acc = 0.0;
    int k = s[q];
    const int ke = e[q];
    for (; k + 4 <= ke; k += 4) {
      const double p0 = prod[k], p1 = prod[k + 1], p2 = prod[k + 2], p3 = prod[k + 3];
      acc += p0; acc += p1; acc += p2; acc += p3;
    }
    for (; k < ke; ++k) acc += prod[k];
    out[q] = acc;
  }
}
// the values of a slice's non-zeros (thread t: entries t, t + XTB, ...): read again for every product -- coalesced, cache-resident --
// instead of living in 2 NZ registers for the whole launch
template <int NZ>
__device__ __forceinline__ void x_mat(const double *g, int cnt, double (&v)[NZ]) {
#pragma unroll
  for (int u = 0; u < NZ; ++u) v[u] = (u < cnt) ? x_at(g, threadIdx.x + (unsigned)u * XTB) : 0.0;
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
// the same with the element's old values already in registers (every load of the update is issued before its first store); the averaged u entry is returned
struct XLd { double u, v, ua, va, us, vs, g, bc; }; // u, v, the restart sums, the running sums, g, and b_i or c_j of one element
__device__ __forceinline__ double xs_y2(const UpdArgs &a, const XLd &L, double un, double vn, Stat &st) {
  const double us = L.us + un, vs = L.vs + vn; // compute_avg, abip.c:649-656
  const double ua = us / a.dom, va = vs / a.dom;
  const double bi = L.bc;
  st.wg += a.rho * (un + vn) * L.g;
  st.nu += un * un; st.nv += vn * vn; st.by += bi * un;
  if (a.avg_stats) { st.nua += ua * ua; st.nva += va * va; st.bya += bi * ua; }
  return ua;
}
__device__ __forceinline__ double xs_x2(const UpdArgs &a, const XLd &L, bool tail, double un, double vn, Stat &st) {
  const double us = L.us + un, vs = L.vs + vn;
  const double ua = us / a.dom, va = vs / a.dom;
  st.nu += a.xw * (un * un); st.nv += a.xw * (vn * vn);
  if (a.avg_stats) { st.nua += a.xw * (ua * ua); st.nva += a.xw * (va * va); }
  if (!tail) {
    const double cj = L.bc;
    st.wg += a.xw * ((un + vn) * L.g);
    st.cx += a.xw * (cj * un);
    if (a.avg_stats) st.cxa += a.xw * (cj * ua);
  }
  return ua;
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

#ifdef XCD_PROF // developer build: where the time of the PCG loop goes (ticks of the 100 MHz wall clock, rank 0 thread 0)
#define XP_DECL unsigned long long xp_t = 0, xp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define XP_START xp_t = wall_clock64();
#define XP_LAP(k) { const unsigned long long xp_n = wall_clock64(); xp_acc[k] += xp_n - xp_t; xp_t = xp_n; }
#define XP_DUMP if (rank == 0 && t == 0) { for (int q = 0; q < 8; ++q) a.xstat[600 + q] = (int)xp_acc[q]; for (int q = 0; q < 8; ++q) a.xstat[610 + q] = (int)xq_acc[q]; }
// coarse counters of a launch that spans outer iterations: [0] the launch, [1] ADMM iterations, [2] look-ahead steps of the search, [3] outer end / begin,
// [4] / [5] the PCG loops inside [1] / [2], [6] / [7] their PCG iterations
#define XQ_DECL unsigned long long xq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xq_t = 0, xq_c = 0, xq_0 = wall_clock64();
#define XQ_MARK(v) v = wall_clock64();
#define XQ_ADD(k, v) xq_acc[k] += wall_clock64() - v;
#define XQ_CNT(k, n) xq_acc[k] += (unsigned long long)(n);
#else
#define XP_DECL
#define XP_START
#define XP_LAP(k)
#define XP_DUMP
#define XQ_DECL
#define XQ_MARK(v)
#define XQ_ADD(k, v)
#define XQ_CNT(k, n)
#endif


// the smallest of one value per thread, in every thread (outer iterations only: a handful of calls per solve)
__device__ __forceinline__ double x_block_min(double v, double *mnb /* XWAVES doubles of their own */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  __syncthreads(); // (the previous call's readers are through)
  if ((threadIdx.x & 63) == 0) mnb[x_wave()] = v;
  __syncthreads();
  double m = mnb[0];
#pragma unroll
  for (int w = 1; w < XWAVES; ++w) m = fmin(m, mnb[w]);
  return x_uni(m);
}

// one look-ahead step of the Barzilai-Borwein search on one (x, tau) entry (adaptive.c:106-121; dev_kernels.h k_adapt_step)
__device__ __forceinline__ void x_bbstep(double alpha, double mu_over_beta, double utq, double uo, double vo, double &un, double &vn) {
  const double t = alpha * utq + (1 - alpha) * uo - vo;
  const double hlf = t / 2;
  un = hlf + sqrt(hlf * hlf + mu_over_beta);
  vn = vo + (un - alpha * utq - (1 - alpha) * uo);
}
// the five inner products of the difference vectors, one entry's terms (adaptive.c:154-174; dev_kernels.h k_adapt_dots)
__device__ __forceinline__ void x_bbdots(double alpha, double u, double v, double un, double vn, double vp, double (&a)[5]) {
  const double dut = 2.0 * v + un - u - vn - vp;
  const double du = u - un;
  const double dv = (un - u) * (alpha - 1.0) + vn - v;
  a[0] += dut * dut; a[1] += dut * dv; a[2] += du * du; a[3] += dv * dv; a[4] += du * dv;
}
// reinitialize_vars, abip.c:996-1075, one (x, tau) entry: indx 0, then 1 when the search follows (two passes of the reference over the same entry)
__device__ __forceinline__ void x_reinit01(double sigma, bool also1, double &u, double &v) {
  if (u > v) v = sigma * v; else u = sigma * u;
  if (also1) { const double sq = sqrt(sigma); u = sq * u; v = sq * v; }
}

template <int NZ, int RM, int RN, bool PCG, bool CROSS = false> // CROSS: the ranks sit on several XCDs (stores of the exchanges written through)
__global__ __launch_bounds__(XTB) void k_lp_xcd(const XcdArgs a) {
  constexpr int SA = CROSS ? 16 : 0;
  extern __shared__ double xl[];
  double *prod = xl;                 // 2 x NZ * XTB
  double *tot = prod + 2 * NZ * XTB; // XKS
  double *red = tot + XKS;           // XWAVES * XKS
  double *outs = red + XWAVES * XKS; // 96: the finalised sums (every workgroup holds the same)
  double *mnb = outs + 96;           // 16: x_block_min
  double *cs = mnb + 16;             // 32: the loop state only the outer iterations touch (CS_*; the same in every workgroup): out of the registers the inner loop needs
  double *wv = cs + 32;              // direct: the whole right-hand side w (m_pad), the products of the owned rows (RM * XTB), then rows of inv(rho I + A A')
  double *mrow = wv + a.m_pad + RM * XTB;
  __shared__ int s_rank;
  const unsigned t = threadIdx.x;
  const int lane = t & 63, wave = x_wave();
  if (t == 0) {
    int r = -1;
    if (x_xcc_id() < (unsigned)a.nxcd) {
      r = (int)(atomicAdd(a.tickets, 1u) - a.ticket_base);
      if (r < 0 || r >= a.G) r = -1;
    }
    s_rank = r;
    mnb[15] = 0.0; // XWait::ldead
    cs[CS_MU] = a.outer.mu; cs[CS_BETA] = a.outer.beta; cs[CS_DYNS] = a.outer.dyn_sigma; cs[CS_OI] = (double)a.outer.i; cs[CS_FRE] = (double)a.outer.fre_old;
    cs[CS_BBPREV] = 1.0; cs[CS_BUPT] = 0.0; cs[CS_BVPT] = 0.0; cs[CS_BUT] = 0.0; cs[CS_BVT] = 0.0; cs[CS_RUT] = 0.0; cs[CS_RVT] = 0.0;
    cs[CS_ODONE] = 0.0; cs[CS_DYN2] = 0.0; cs[CS_LOGN] = 0.0; cs[CS_BBTOT] = 0.0; cs[CS_BBIT] = 0.0; cs[CS_AV] = 0.0; cs[CS_PHASE] = 0.0; cs[CS_REASON] = (double)XR_BATCH;
    cs[CS_T0] = __longlong_as_double(wall_clock64());
    cs[CS_CGSKIP] = 0.0;
  }
  __syncthreads();
  const int rank = __builtin_amdgcn_readfirstlane(s_rank); // (uniform, and the compiler may know it: everything derived from it lives in scalar registers)
  if (rank < 0 || rank == a.desert) return;
  const int G = a.G;
  const unsigned MP = (unsigned)a.MP;
  const unsigned m0 = (unsigned)a.mb[rank], m1 = (unsigned)a.mb[rank + 1], n0 = (unsigned)a.nb[rank], n1 = (unsigned)a.nb[rank + 1];
  const int ka0 = a.Ap[m0], ka1 = a.Ap[m1], kt0 = a.Tp[n0], kt1 = a.Tp[n1];
  // ---- the two slices: the positions their non-zeros gather from (byte offsets), in registers for the whole launch ----
  unsigned ai[NZ], ti[NZ];
  int na = 0, nt = 0;
#pragma unroll
  for (int u = 0; u < NZ; ++u) {
    const unsigned k = t + u * XTB;
    ai[u] = 0; ti[u] = 0;
    if (ka0 + (int)k < ka1) { ai[u] = 8u * (unsigned)x_at(a.Ai, ka0 + k); na = u + 1; }
    if (kt0 + (int)k < kt1) { ti[u] = 8u * (unsigned)x_at(a.Ti, kt0 + k); nt = u + 1; }
  }
  const double *gA = a.Ax + ka0, *gT = a.Tx + kt0; // the slices' values
#ifdef XCD_HALO_PROBE // developer build (make exp EXPNAME=haloprobe EXPDEF=-DXCD_HALO_PROBE): what a halo form of the PCG would ADD to the first exchange's gather phase --
  // two more slices of the same size gathered from the m-space and summed by rows (a rank would form A'p for every column its rows touch: ~3x its own slice on c3)
  unsigned ti2[NZ], ti3[NZ];
#pragma unroll
  for (int u = 0; u < NZ; ++u) { ti2[u] = 8u * ((ti[u] / 8u + 37u) % (unsigned)a.m); ti3[u] = 8u * ((ti[u] / 8u + (unsigned)a.m / 2u + 5u) % (unsigned)a.m); }
  double hp_sink = 0.0;
#endif
  int sa[RM], ea[RM], st[RN], et[RN];
#pragma unroll
  for (int q = 0; q < RM; ++q) { const unsigned i = m0 + t + q * XTB; sa[q] = 0; ea[q] = 0; if (i < m1) { sa[q] = x_at(a.Ap, i) - ka0; ea[q] = x_at(a.Ap, i + 1) - ka0; } }
#pragma unroll
  for (int q = 0; q < RN; ++q) { const unsigned j = n0 + t + q * XTB; st[q] = 0; et[q] = 0; if (j < n1) { st[q] = x_at(a.Tp, j) - kt0; et[q] = x_at(a.Tp, j + 1) - kt0; } }

  UpdArgs up = a.upd;
  up.fuse_avg = 1; up.xw = 1.0; up.gs = nullptr;
  const double rho = up.rho;
  const unsigned tail = MP + (unsigned)a.n;
  const bool solo = a.solve_only != 0;
  const bool whole = a.outer.on != 0 && !solo;
  unsigned tag = a.tag0;
  int flip = 0;
  XWait w; w.xstat = a.xstat; w.dead = false; w.site = 0; w.rank = rank; w.ldead = mnb + 15; // (x_block_min uses the first XWAVES of the 16)
  xrsrc pn0, pn1, pm0, pm1, psc; // the exchange areas of the parity in use
  xrsrc pnv;                     // (direct) v_x of the update's exchange
  const unsigned sc_off = (unsigned)rank * 16u; // this rank's granule of slot 0 (slot k: + k * XSLOT)
  auto open = [&](int site) { // next exchange: tag and the areas of its parity
    tag = (unsigned)__builtin_amdgcn_readfirstlane((int)(tag + 1u)); // (uniform by construction; the loops' give-up exits hide that from the compiler)
    const size_t par = tag & 1u;
    w.tag = tag; w.site = site;
    pn0 = x_rsrc(a.xn0 + par * a.n_pad, 8u * (unsigned)a.n_pad); pn1 = x_rsrc(a.xn1 + par * a.n_pad, 8u * (unsigned)a.n_pad);
    pm0 = x_rsrc(a.xm0 + par * a.m_pad, 8u * (unsigned)a.m_pad); pm1 = x_rsrc(a.xm1 + par * a.m_pad, 8u * (unsigned)a.m_pad);
    psc = x_rsrc(a.sc + par * (size_t)(XG * XKS), 16u * XG * XKS);
    if (!PCG) pnv = x_rsrc(a.xnv + par * a.n_pad, 8u * (unsigned)a.n_pad);
    w.n0 = pn0; w.n1 = pn1; w.m0 = pm0; w.m1 = pm1; w.sc = psc;
    if (t == 0) { a.xstat[8 + 2 * rank] = (int)tag; a.xstat[9 + 2 * rank] = site; } // (post-mortem: where every rank was when a wait gave up)
  };

  if (!PCG) { // this workgroup's first rows of the dense inverse stay in the LDS for the whole launch
    // (small systems read every row out to 64 XMR columns, see the dense product: rows this workgroup does not fill and the pad behind the last one are zero)
    for (unsigned c = (unsigned)a.m + t; c < (unsigned)a.m_pad; c += XTB) wv[c] = 0.0; // the pad of w stays zero: the all-gather writes [0, m)
    if (a.m_pad <= 64 * 16) { for (unsigned c = t; c < (unsigned)a.minv_lds_rows * (unsigned)a.m_pad + (64u * 16u - (unsigned)a.m_pad); c += XTB) mrow[c] = 0.0; __syncthreads(); }
    for (unsigned rl = wave; (int)rl < a.minv_lds_rows && m0 + rl < m1; rl += XWAVES) {
      const double *row = a.Minv + (long)(m0 + rl) * a.ldM;
      for (unsigned c = lane; c < (unsigned)a.m_pad; c += 64) mrow[(size_t)rl * a.m_pad + c] = x_at(row, c);
    }
  }
  // ... and, while a row is at most XMR doubles per lane, each wavefront keeps ONE more row in registers and its lanes hold their share of w for all of a
  // wavefront's rows (c2: 19 rows in the LDS + 7 in registers = all 26 of a workgroup, nothing streamed from the L2 inside the loop)
  constexpr int XMR = 16;
  const bool dsmall = !PCG && a.m_pad <= 64 * XMR;
  double mreg[XMR];
#pragma unroll
  for (int k = 0; k < XMR; ++k) mreg[k] = 0.0;
  const unsigned rl_reg = (unsigned)a.minv_lds_rows + (unsigned)wave; // (local index of the row this wavefront keeps)
  if (dsmall && m0 + rl_reg < m1) {
    const double *row = a.Minv + (long)(m0 + rl_reg) * a.ldM;
#pragma unroll
    for (int k = 0; k < XMR; ++k) { const unsigned c = 64u * k + lane; if (c < (unsigned)a.m) mreg[k] = x_at(row, c); }
  }
  if (t < 96) outs[t] = a.ctl->out[t]; // slots this launch does not refresh keep what the last finalize left (as on the launch path)
  XP_DECL
  XQ_DECL

  // ---- the loop state of abip.c:2102-2294 (uniform: every workgroup carries the same values and takes the same branches).  What only the outer
  //      iterations touch lives in the LDS (cs[]: written by thread 0, read by everybody after a barrier); the inner loop's share in registers ----
  const XcdOuter &xo = a.outer;
  auto csr = [&](int k) { return x_uni(cs[k]); };
  auto csw = [&](int k, double v) { if (t == 0) cs[k] = v; };
  int mode = XM_MAIN, stage = (whole && xo.phase == 1) ? XS_OUTER_END : XS_PROJECT;
  long jj = a.j0;
  double thr = a.thr;
  int final_check = a.fc.on;
  int ran = 0, halt = 0, last_cg = 0, avg_stats_last = 0;
  bool stats_valid = false;
  long cg_total = 0;
  double metric = 0.0;
  int avg_crit = whole ? xo.avg_crit : 0;
  bool need_pre = true;
  double aty[RN]; // (A'u_y)_j of the owned columns: left by the stopping test (or by the exchange below), used by the next solve's set-up
  double wg = 0.0, u_tau = 0.0, v_tau = 0.0;
#pragma unroll
  for (int q = 0; q < RN; ++q) aty[q] = 0.0;
#ifdef XCD_RSUM // developer build (make exp EXPNAME=rsum EXPDEF=-DXCD_RSUM): the restart sums and the running sums of the owned elements live in registers for
  // the whole launch instead of four read-modify-write streams per element and iteration; the arrays are written once, when the launch ends
  double Rua_y[RM], Rva_y[RM], Rus_y[RM], Rvs_y[RM], Rua_x[RN], Rva_x[RN], Rus_x[RN], Rvs_x[RN], Rtl[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < RM; ++q) {
    const unsigned i = m0 + t + q * XTB;
    Rua_y[q] = 0.0; Rva_y[q] = 0.0; Rus_y[q] = 0.0; Rvs_y[q] = 0.0;
    if (i < m1 && !solo) { Rua_y[q] = x_at(up.u_avg, i); Rva_y[q] = x_at(up.v_avg, i); Rus_y[q] = x_at(up.u_sum, i); Rvs_y[q] = x_at(up.v_sum, i); }
  }
#pragma unroll
  for (int q = 0; q < RN; ++q) {
    const unsigned j2 = n0 + t + q * XTB;
    Rua_x[q] = 0.0; Rva_x[q] = 0.0; Rus_x[q] = 0.0; Rvs_x[q] = 0.0;
    if (j2 < n1 && !solo) { const unsigned qq = MP + j2; Rua_x[q] = x_at(up.u_avg, qq); Rva_x[q] = x_at(up.v_avg, qq); Rus_x[q] = x_at(up.u_sum, qq); Rvs_x[q] = x_at(up.v_sum, qq); }
  }
  if (rank == 0 && t == 0 && !solo) { Rtl[0] = x_at(up.u_avg, tail); Rtl[1] = x_at(up.v_avg, tail); Rtl[2] = x_at(up.u_sum, tail); Rtl[3] = x_at(up.v_sum, tail); }
#endif

  for (;;) {
    // (an index the optimiser cannot see through: otherwise it hoists the element addresses of an iteration out of this loop and spills them)
    unsigned tb = t; asm volatile("" : "+v"(tb));
    XQ_MARK(xq_c)
    // ================================================================================================================================
    // outer end (abip.c:2217-2293): time, final_check, residuals, convergence, mu, reinitialize_vars; then the search or the next outer iteration
    // ================================================================================================================================
    if (__builtin_expect(stage == XS_OUTER_END, 0)) {
      csw(CS_PHASE, 1.0);
      if (ran > 0 && ran >= xo.max_steps) { csw(CS_REASON, (double)XR_STEPS); break; } // abip_hip_step hands back right after the last iteration it was asked for: the outer end waits for the next call
      // avg_criterion (abip.c:2042, 2048): the averaged iterate won the last stopping test -- residuals, the LOQO products and reinitialize_vars then work on
      // (u_avgcon, v_avgcon) (abip.c:473-484, 940-946, 1007-1039), the search still starts from (u, v) (adaptive.c:87-88), and the next inner loop from the average (abip.c:2125-2129)
      const bool av = avg_crit != 0;
      double *RU = av ? up.u_avgc : up.u, *RV = av ? up.v_avgc : up.v;
      double mu = csr(CS_MU);
      const long oi = (long)csr(CS_OI);
      int log_n = (int)csr(CS_LOGN), dyn2_used = (int)csr(CS_DYN2);
      const unsigned long long t_launch = (unsigned long long)__double_as_longlong(csr(CS_T0));
      // one exchange: has rank 0 seen the slice's end; sum and minimum of u_i v_i over i >= m (abip.c:962-965)
      open(11);
      double p2[2] = {0.0, 0.0};
      double mn = 1e+10;
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) { const double x = x_at(RU, MP + j2) * x_at(RV, MP + j2); p2[1] += x; mn = fmin(mn, x); }
      }
      if (rank == 0 && t == 0) {
        const double x = x_at(RU, tail) * x_at(RV, tail);
        p2[1] += x; mn = fmin(mn, x);
        p2[0] = ((unsigned long long)wall_clock64() - t_launch > xo.slice_ticks) ? 1.0 : 0.0;
      }
      mn = x_block_min(mn, mnb);
      if (t == 0) x_putg<SA>(psc, sc_off + 12u * XSLOT, mn, tag); // (acknowledged before this rank's flags go out)
      x_publish<2, SA>(p2, red, psc, sc_off, tag);
      double P2[2];
      x_collect<2, CROSS>(w, G, tot, P2);
      if (__builtin_expect(w.dead, 0)) return;
      double gmin = 1e+10;
      if ((int)t < G) gmin = x_val(x_ldg(psc, 12u * XSLOT + t * 16u));
      gmin = x_block_min(gmin, mnb);
      if (P2[0] > 0.0 || log_n >= xo.log_cap) { csw(CS_REASON, (double)XR_SLICE); break; } // (nothing of this outer end has been applied: the next launch enters here again.
                                                                                          //  A full log -- LOQO rule on a tiny LP: no mu-table limit -- ends the launch the same way: the host prints its rows)
      if (mu < a.fc.eps) final_check = 1;                          // abip.c:2224-2227
      __syncthreads();                                             // outs complete
      LpResid r;
      lp_residuals(x_sums(outs, av ? 1 : 0), a.fc.den, a.fc.nm_b, a.fc.nm_c, r);
      const long kk = a.fc.k0 + ran;
      if (rank == 0 && t == 0 && xo.log && log_n < xo.log_cap) {
        double *row = xo.log + (size_t)log_n * XLOG_W;
        row[0] = (double)oi; row[1] = (double)kk; row[2] = mu; row[3] = r.res_pri; row[4] = r.res_dual; row[5] = r.rel_gap;
        row[6] = r.ct_x_by_tau; row[7] = r.bt_y_by_tau; row[8] = r.tau; row[9] = r.kap;
        row[10] = (double)((unsigned long long)wall_clock64() - t_launch); row[11] = 0.0;
      }
      if (lp_converged(r, a.fc.eps, a.fc.pfeasopt, oi, kk) != 0 || kk + 1 >= a.fc.max_admm) { csw(CS_REASON, (double)XR_HOST_OUTER); break; } // the host extracts the solution
      double ds = csr(CS_DYNS);
      const int rule = __builtin_amdgcn_readfirstlane(lp_mu_rule(xo.hybrid_mu, xo.dyn_sigma_second, xo.hybrid_thresh, a.fc.eps, mu, ds));
      if (rule == LP_MU_TABLE) { csw(CS_REASON, (double)XR_HOST_OUTER); break; }
      if (rule == LP_MU_LOQO) {
        if (gmin <= 0.0) { csw(CS_REASON, (double)XR_HOST_OUTER); break; } // the reference asserts here: the host reports it
        mu = mu * lp_loqo_sigma(P2[1], gmin, (long)a.n + 1, ds);
      } else if (rule == LP_MU_DYN2) {
        if (dyn2_used >= xo.mu_tab_n) { csw(CS_REASON, (double)XR_SLICE); break; }
        mu = xo.mu_tab[dyn2_used];
        ++dyn2_used;
      }
      if (log_n < xo.log_cap) ++log_n;
      mu = x_uni(mu);
      // reinitialize_vars(0), and (1) when the search follows (abip.c:2279-2283)
      const bool search = xo.adaptive != 0;
      u_tau = outs[80]; v_tau = outs[81];
      double r_ut = av ? outs[82] : outs[80], r_vt = av ? outs[83] : outs[81];
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          double uu = x_at(RU, MP + j2), vv = x_at(RV, MP + j2);
          x_reinit01(xo.sigma, search, uu, vv);
          x_at(RU, MP + j2) = uu; x_at(RV, MP + j2) = vv;
          if (search) { x_at(xo.a_up, MP + j2) = av ? x_at(up.u, MP + j2) : uu; x_at(xo.a_vp, MP + j2) = av ? x_at(up.v, MP + j2) : vv; } // u_prev, v_prev <- u, v (adaptive.c:87-88)
        }
      }
      x_reinit01(xo.sigma, search, r_ut, r_vt);
      r_ut = x_uni(r_ut); r_vt = x_uni(r_vt);
      if (rank == 0 && t == 0) { x_at(RU, tail) = r_ut; x_at(RV, tail) = r_vt; }
      if (!av) { u_tau = r_ut; v_tau = r_vt; }
      stats_valid = false;
      csw(CS_MU, mu); csw(CS_DYNS, ds); csw(CS_LOGN, (double)log_n); csw(CS_DYN2, (double)dyn2_used); csw(CS_ODONE, csr(CS_ODONE) + 1.0);
      csw(CS_RUT, r_ut); csw(CS_RVT, r_vt); csw(CS_AV, av ? 1.0 : 0.0);
      if (search && xo.lookback > 0) {
        csw(CS_BETA, 1.0); // abip.c:2284
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          const unsigned i = m0 + tb + q * XTB;
          if (i < m1) { x_at(xo.a_up, i) = x_at(up.u, i); x_at(xo.a_vp, i) = x_at(up.v, i); }
        }
        csw(CS_BUPT, u_tau); csw(CS_BVPT, v_tau); csw(CS_BBPREV, 1.0); csw(CS_BBIT, 0.0);
        mode = XM_BB1; need_pre = true; stage = XS_PROJECT;
      } else {
        if (search) { // (adaptive.c:90: the loop does not run, beta stays 0; reinitialize_vars(2))
          csw(CS_BETA, 0.0);
          const double sq = sqrt(1.0 / xo.sigma);
#pragma unroll
          for (int q = 0; q < RN; ++q) { const unsigned j2 = n0 + tb + q * XTB; if (j2 < n1) { x_at(RU, MP + j2) = sq * x_at(RU, MP + j2); x_at(RV, MP + j2) = sq * x_at(RV, MP + j2); } }
          r_ut = x_uni(sq * r_ut); r_vt = x_uni(sq * r_vt);
          if (rank == 0 && t == 0) { x_at(RU, tail) = r_ut; x_at(RV, tail) = r_vt; }
          csw(CS_RUT, r_ut); csw(CS_RVT, r_vt);
        }
        stage = XS_OUTER_BEGIN;
      }
    }
    // ================================================================================================================================
    // outer begin (abip.c:2102-2129)
    // ================================================================================================================================
    if (__builtin_expect(stage == XS_OUTER_BEGIN, 0)) {
      __syncthreads(); // (cs[] as the block above, or the search's last look-ahead, left it)
      const long oi = (long)csr(CS_OI) + 1;
      const double mu = csr(CS_MU), beta = csr(CS_BETA);
      const bool av = csr(CS_AV) != 0.0;
      const double r_ut = csr(CS_RUT), r_vt = csr(CS_RVT);
      __syncthreads(); // (everybody has read what thread 0 overwrites next)
      csw(CS_OI, (double)oi); csw(CS_PHASE, 2.0);
      if (oi >= xo.max_ipm) { csw(CS_REASON, (double)XR_MAXIPM); break; } // abip.c:2296: the host closes the solve
      csw(CS_FRE, 0.0);
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          x_at(up.u_avg, i) = 0.0; x_at(up.v_avg, i) = 0.0; x_at(up.u_sum, i) = 0.0; x_at(up.v_sum, i) = 0.0;
          if (av) { x_at(up.u, i) = x_at(up.u_avgc, i); x_at(up.v, i) = x_at(up.v_avgc, i); } // abip.c:2125-2129
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          x_at(up.u_avg, MP + j2) = 0.0; x_at(up.v_avg, MP + j2) = 0.0; x_at(up.u_sum, MP + j2) = 0.0; x_at(up.v_sum, MP + j2) = 0.0;
          if (av) { x_at(up.u, MP + j2) = x_at(up.u_avgc, MP + j2); x_at(up.v, MP + j2) = x_at(up.v_avgc, MP + j2); }
        }
      }
      if (rank == 0 && t == 0) {
        x_at(up.u_avg, tail) = 0.0; x_at(up.v_avg, tail) = 0.0; x_at(up.u_sum, tail) = 0.0; x_at(up.v_sum, tail) = 0.0;
        if (av) { x_at(up.u, tail) = r_ut; x_at(up.v, tail) = r_vt; }
      }
#ifdef XCD_RSUM
#pragma unroll
      for (int q = 0; q < RM; ++q) { Rua_y[q] = 0.0; Rva_y[q] = 0.0; Rus_y[q] = 0.0; Rvs_y[q] = 0.0; }
#pragma unroll
      for (int q = 0; q < RN; ++q) { Rua_x[q] = 0.0; Rva_x[q] = 0.0; Rus_x[q] = 0.0; Rvs_x[q] = 0.0; }
      Rtl[0] = 0.0; Rtl[1] = 0.0; Rtl[2] = 0.0; Rtl[3] = 0.0;
#endif
      jj = 0;
      up.mu_over_beta = x_uni(mu / beta);
      thr = x_uni(xo.gamma * mu);
      mode = XM_MAIN; need_pre = true; stage = XS_PROJECT;
      __syncthreads(); // (cs[] complete before anybody reads it again)
    }
    XQ_ADD(3, xq_c)
    // ---- may this launch start another ADMM iteration? (abip_hip_step's budget; abip.c:608-609: a restart is the launch path's) ----
    if (mode == XM_MAIN && !solo) {
      if (whole) {
        csw(CS_PHASE, 0.0);
        if (ran >= xo.max_steps) { csw(CS_REASON, (double)XR_STEPS); break; }
        const long kk = a.fc.k0 + ran;
        if (kk >= xo.restart_thresh && (jj + 1 - (long)csr(CS_FRE)) % xo.restart_fre == 0) { csw(CS_REASON, (double)XR_RESTART); break; }
      } else if (ran >= a.max_iters) break;
    }
    const double *srcU = (mode == XM_MAIN) ? up.u : (mode == XM_BB1 ? xo.a_up : xo.a_u); // the iterate the projection starts from
    const double *srcV = (mode == XM_MAIN) ? up.v : (mode == XM_BB1 ? xo.a_vp : xo.a_v);

    // ================================================================================================================================
    // S_WG and the tau entries for the right-hand side; A'u_y, the warm start's product (PCG): the first projection of a launch, of an inner loop, of a look-ahead step
    // ================================================================================================================================
    if (__builtin_expect(need_pre, 0)) {
      open(1);
      double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double uy = solo ? (a.swarm ? x_at(a.swarm, i) : 0.0) : x_at(srcU, i);
          if (!solo) p[0] += rho * (uy + x_at(srcV, i)) * x_at(up.g, i);
          if (PCG) x_putd<SA>(pm0, i * 8u, uy);
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j = n0 + tb + q * XTB;
        if (j < n1 && !solo) p[0] += (x_at(srcU, MP + j) + x_at(srcV, MP + j)) * x_at(up.g, MP + j);
      }
      if (rank == 0 && t == 0 && !solo && mode == XM_MAIN) { p[1] = x_at(up.u, tail); p[2] = x_at(up.v, tail); }
      x_publish<3, SA>(p, red, psc, sc_off, tag);
      double s3[3];
      x_collect<3, CROSS>(w, G, tot, s3);
      if (__builtin_expect(w.dead, 0)) return;
      if (PCG) {
        double tx[NZ], vt[NZ];
        x_mat<NZ>(gT, nt, tx);
        x_gather<NZ>(pm0, ti, vt);
        x_rows<NZ, RN>(prod, flip, tx, vt, nt, st, et, aty);
      }
      wg = s3[0];
      if (mode == XM_MAIN) { u_tau = s3[1]; v_tau = s3[2]; }
      need_pre = false;
    }

    // ================================================================================================================================
    // The inner loop proper (abip.c:2131-2215): one projection per trip and, for an ADMM iteration, everything up to its exit test.  A look-ahead
    // step of the search (and the solve-only mode) leaves after the projection; what they do with it follows the loop.
    // ================================================================================================================================
    double y[RM], zx[RN], dhS[1], tsum; // what a projection leaves behind: the solution's y and x blocks, u_t'h, the tau entry of the right-hand side
    double rhs_y[RM], rhs_x[RN];        // the right-hand side of the projection in flight (k_rhs, abip.c:552-558)
    // the entries this workgroup owns, from the iterate (srcU, srcV) -- or the caller's vector in the solve-only mode --; the x block goes out to the exchange
    // area of the exchange that is open (and with it, for the PCG back-end, A'u_y of the warm start)
    auto build_rhs = [&](unsigned tb_, double ts_, double cf_, double (&bn_)[1]) {
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb_ + q * XTB;
        rhs_y[q] = 0.0;
        if (i < m1) {
          double r;
          if (solo) r = x_at(a.srhs, i);
          else {
            const double hi = x_at(a.h, i);
            r = (x_at(srcU, i) + x_at(srcV, i)) * rho;
            r += -ts_ * hi;
            r += -cf_ * hi;
          }
          rhs_y[q] = r;
          bn_[0] += r * r;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb_ + q * XTB;
        rhs_x[q] = 0.0;
        if (j2 < n1) {
          double r;
          if (solo) r = -x_at(a.srhs, MP + j2);
          else {
            const double hj = x_at(a.h, MP + j2);
            r = x_at(srcU, MP + j2) + x_at(srcV, MP + j2);
            r += -ts_ * hj;
            r += -cf_ * hj;
          }
          rhs_x[q] = -r;
          x_putd<SA>(pn0, j2 * 8u, -r);
          if (PCG) x_putd<SA>(pn1, j2 * 8u, aty[q]);
        }
      }
    };
    bool have_rhs = false;
    bool have_w = false; // (direct) w of this trip went out with the previous iteration's stopping-test sums: the trip starts at the all-gather
    int leave = 0;
    double hA[NZ];       // (direct) h_x at the columns this thread's non-zeros of A gather: constant
#pragma unroll
    for (int u = 0; u < NZ; ++u) hA[u] = (!PCG && u < na) ? x_at(a.h + MP, ai[u] >> 3) : 0.0;
    double hy[RM], hx[RN]; // h of the owned elements: constant, read by three phases of every trip
    double hy2[RM];        // h_y + (A h_x) of the owned rows
#pragma unroll
    for (int q = 0; q < RM; ++q) { const unsigned i = m0 + t + q * XTB; hy[q] = i < m1 ? x_at(a.h, i) : 0.0; hy2[q] = i < m1 ? x_at(a.hAh, i) : 0.0; }
#pragma unroll
    for (int q = 0; q < RN; ++q) { const unsigned j2 = n0 + t + q * XTB; hx[q] = j2 < n1 ? x_at(a.h, MP + j2) : 0.0; }
    for (;;) {
    unsigned tb = t; asm volatile("" : "+v"(tb)); // (see the outer loop)
    XQ_MARK(xq_t)
    const bool avg_stats = ((jj + 1) % 10 == 0); // abip.c:2000
    if (mode == XM_MAIN) { up.dom = (double)(jj + 1); up.avg_stats = avg_stats ? 1 : 0; }
    // ---- right-hand side (k_rhs, abip.c:552-558) ----
    tsum = (mode == XM_MAIN) ? (u_tau + v_tau) : (mode == XM_BB1 ? (csr(CS_BUPT) + csr(CS_BVPT)) : (csr(CS_BUT) + csr(CS_BVT)));
    const double coef = (wg - tsum * a.g_th) / (a.g_th + 1.0);
    if (!PCG) { XP_START }
    double bn[1] = {0.0}, bnS[1] = {0.0};
    if (!have_rhs && !have_w) {
      open(2);
      build_rhs(tb, tsum, coef, bn);
      x_publish<1, SA>(bn, red, psc, sc_off, tag);
      x_collect<1, CROSS>(w, G, tot, bnS);
      if (__builtin_expect(w.dead, 0)) return;
    }
    have_rhs = false; // (the right-hand side this trip works on was built, and handed round, by the previous iteration's last exchange -- see there)
    if (PCG) {
      // ---- PCG set-up (k_cg_init_A, indirect.c:345-365, 415) ----
      double sA[RM], sB[RM];
      {
        double ax[NZ], va[NZ];
        x_mat<NZ>(gA, na, ax);
        x_gather<NZ>(pn0, ai, va);
        x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, sA);
        x_gather<NZ>(pn1, ai, va);
        x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, sB);
      }
      double tolf;
      if (whole) { const double kp = (double)(a.fc.k0 + ran + 1); tolf = 1e-1 / (kp * kp); } // indirect.c:406-407 with cg_rate = 2 (the host admits nothing else here); the search solves with iter = k (adaptive.c:98)
      else tolf = a.tolf[ran];
      double tol = sqrt(bnS[0]) * tolf; // indirect.c:406-409, 418
      tol = fmax(tol, 1e-7);
      tol = fmax(tol, 1e-9);
      double cr[RM], cz[RM], cp[RM], cx[RM], Mj[RM], ctmp[RN];
      double rz[2] = {0.0, 0.0};
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        cr[q] = 0.0; cz[q] = 0.0; cp[q] = 0.0; cx[q] = 0.0; Mj[q] = 0.0;
        if (i < m1) {
          const double si = solo ? (a.swarm ? x_at(a.swarm, i) : 0.0) : x_at(srcU, i);
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
      XQ_MARK(xq_c)
      XP_START
      for (;;) {
        // ---- z and (|r|^2, z'r) out; convergence test; tmp = A'z + beta tmp (k_cg_spmv_At) ----
        open(3);
#pragma unroll
        for (int q = 0; q < RM; ++q) { const unsigned i = m0 + t + q * XTB; if (i < m1) x_putd<SA>(pm0, i * 8u, cz[q]); }
        XP_LAP(0)
        x_publish<2, SA>(rz, red, psc, sc_off, tag);
        XP_LAP(1)
        double tx[NZ];
        x_mat<NZ>(gT, nt, tx); // (on their way while the flags are awaited)
        double rzS[2];
        x_collect<2, CROSS>(w, G, tot, rzS);
        if (__builtin_expect(w.dead, 0)) return;
        XP_LAP(2)
        const double nr = sqrt(rzS[0]);
        bool done = (cgit == 0) ? (nr < fmin(tol, 1e-18)) : (nr < tol); // indirect.c:359, 375
        if (cgit >= a.cg_max_its) done = true;                          // indirect.c:368
        if (done) break;
        double tq[RN];
        {
          double vt[NZ];
          x_gather<NZ>(pm0, ti, vt);
#ifdef XCD_HALO_PROBE
          double v2[NZ], v3[NZ], d2[RN], d3[RN];
          x_gather<NZ>(pm0, ti2, v2); // (all three gathers in flight together, as a halo form would issue them)
          x_gather<NZ>(pm0, ti3, v3);
#endif
          XP_LAP(3)
          x_rows<NZ, RN>(prod, flip, tx, vt, nt, st, et, tq);
#ifdef XCD_HALO_PROBE
          x_rows<NZ, RN>(prod, flip, tx, v2, nt, st, et, d2);
          x_rows<NZ, RN>(prod, flip, tx, v3, nt, st, et, d3);
#pragma unroll
          for (int q = 0; q < RN; ++q) hp_sink += d2[q] + d3[q];
#endif
        }
        XP_LAP(4)
        const double cbeta = (cgit == 0) ? 0.0 : rzS[1] / zr_prev;
        zr_prev = rzS[1];
        double tp[2] = {0.0, 0.0};
        open(4);
#pragma unroll
        for (int q = 0; q < RN; ++q) {
          const unsigned j2 = n0 + t + q * XTB;
          if (j2 < n1) {
            const double v = (cgit == 0) ? tq[q] : tq[q] + cbeta * ctmp[q];
            ctmp[q] = v;
            tp[0] += v * v;
            x_putd<SA>(pn0, j2 * 8u, v);
          }
        }
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          const unsigned i = m0 + t + q * XTB;
          if (i < m1) { const double pn = cz[q] + cbeta * cp[q]; cp[q] = pn; tp[1] += pn * pn; }
        }
        // ---- tmp and (|A'p|^2, |p|^2) out; Gp = A tmp + rho p; alpha; x, r, z (k_cg_spmv_A + k_cg_update) ----
        XP_LAP(0)
        x_publish<2, SA>(tp, red, psc, sc_off, tag);
        XP_LAP(1)
        double ax[NZ];
        x_mat<NZ>(gA, na, ax);
        double tpS[2];
        x_collect<2, CROSS>(w, G, tot, tpS);
        if (__builtin_expect(w.dead, 0)) return;
        XP_LAP(2)
        double gq[RM];
        {
          double va[NZ];
          x_gather<NZ>(pn0, ai, va);
          XP_LAP(3)
          x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, gq);
        }
        XP_LAP(4)
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
      XP_LAP(5)
      XQ_ADD(mode == XM_MAIN ? 4 : 5, xq_c)
      XQ_CNT(mode == XM_MAIN ? 7 : 6, cgit)
      last_cg = cgit; cg_total += cgit;
#pragma unroll
      for (int q = 0; q < RM; ++q) y[q] = cx[q];
    } else {
      // ---- direct: w = rhs_y + A rhs_x; y = inv(rho I + A A') w; (x follows below) ----
      if (!have_w) { // (the first trip of a loop, a look-ahead step, the solve-only mode: later trips find w out already -- see the end of the trip)
        double sA[RM];
        {
          double ax[NZ], va[NZ];
          x_mat<NZ>(gA, na, ax);
          x_gather<NZ>(pn0, ai, va);
          x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, sA);
        }
        XP_LAP(0)
        open(5);
#pragma unroll
        for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; if (i < m1) x_putd<SA>(pm0, i * 8u, rhs_y[q] + sA[q]); }
        x_flag<SA>(psc, sc_off, tag);
        x_wait<CROSS>(w, G);
        if (__builtin_expect(w.dead, 0)) return;
      }
      have_w = false;
      XP_LAP(1)
      for (unsigned i0 = t; i0 < (unsigned)a.m; i0 += 4 * XTB) { // every workgroup needs the whole w (four loads in flight)
        double tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned i = i0 + (unsigned)u * XTB; tv[u] = i < (unsigned)a.m ? x_ldd(pm0, i * 8u) : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned i = i0 + (unsigned)u * XTB; if (i < (unsigned)a.m) wv[i] = tv[u]; }
      }
      __syncthreads();
      XP_LAP(2)
      double *yv = wv + a.m_pad;
      // one wavefront per owned row; a lane adds its columns lane, lane + 64, ... in that order whichever way the row is held, then the wavefront sum
      auto stream_row = [&](unsigned r) { // a row from the L2 (8 loads in flight)
        const double *row = a.Minv + (long)r * a.ldM;
        double acc = 0.0;
        for (unsigned c0 = 0; c0 < (unsigned)a.m_pad; c0 += 512) { // (to m_pad: the pad of w is zero, the row's finite -- exact zeros, added in the same order; no guard per element)
          double mv[8], wq[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const bool in = c0 + 64u * u < (unsigned)a.m_pad; const unsigned c = c0 + 64u * u + lane; mv[u] = in ? x_at(row, c) : 0.0; wq[u] = in ? wv[c] : 0.0; } // (uniform)
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += mv[u] * wq[u];
        }
        return acc;
      };
      const unsigned nrows = m1 - m0, nlds = min((unsigned)a.minv_lds_rows, nrows);
      if (dsmall) {
        // Branch-free: a lane's column 64 k + lane may lie beyond m (the last k of a row, every k >= m_pad / 64).  A guard per element makes every LDS read a
        // basic block of its own -- read, wait, multiply, sixteen times in a row: 3.5 us of a 17.5 us iteration on c2 (profiles/r04l_*).  Instead every lane
        // reads all XMR columns at their natural addresses (one base register per row, the column in the instruction's offset field) and its entry of w is
        // ZERO beyond m: the term is an exact zero, the sum keeps its bits, and the reads go out back to back.  What lies behind a row's end is the next row,
        // or the zeroed pad behind the last one (finite either way: 0 x finite = 0).
        const unsigned mlast = (unsigned)a.m - 1u;
        double wreg[XMR]; // this lane's columns of w: read once, used for every row of the wavefront
#pragma unroll
        for (int k = 0; k < XMR; ++k) { const double wk = wv[64u * k + lane]; wreg[k] = 64u * k + lane <= mlast ? wk : 0.0; } // (behind w: y of the last trip, discarded)
        constexpr int KC = 4; // columns per lane whose reads are in flight together
        unsigned rl = wave;
        for (; rl + XWAVES < nlds; rl += 2 * XWAVES) { // rows resident in the LDS, two at a time (independent chains)
          const double *r0 = mrow + (size_t)rl * a.m_pad + lane, *r1 = r0 + (size_t)XWAVES * a.m_pad;
          double a0 = 0.0, a1 = 0.0;
#pragma unroll
          for (int k0 = 0; k0 < XMR; k0 += KC) {
            double v0[KC], v1[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) { v0[k] = r0[64 * (k0 + k)]; v1[k] = r1[64 * (k0 + k)]; }
#pragma unroll
            for (int k = 0; k < KC; ++k) { a0 += v0[k] * wreg[k0 + k]; a1 += v1[k] * wreg[k0 + k]; }
          }
          a0 = x_wave_sum63(a0); a1 = x_wave_sum63(a1);
          if (lane == 63) { yv[rl] = a0; yv[rl + XWAVES] = a1; }
        }
        if (rl < nlds) {
          const double *r0 = mrow + (size_t)rl * a.m_pad + lane;
          double a0 = 0.0;
#pragma unroll
          for (int k0 = 0; k0 < XMR; k0 += 2 * KC) {
            double v0[2 * KC];
#pragma unroll
            for (int k = 0; k < 2 * KC; ++k) v0[k] = r0[64 * (k0 + k)];
#pragma unroll
            for (int k = 0; k < 2 * KC; ++k) a0 += v0[k] * wreg[k0 + k];
          }
          a0 = x_wave_sum63(a0);
          if (lane == 63) yv[rl] = a0;
        }
        if (rl_reg < nrows) { // the row in registers (zero beyond m, like w)
          double a0 = 0.0;
#pragma unroll
          for (int k = 0; k < XMR; ++k) a0 += mreg[k] * wreg[k];
          a0 = x_wave_sum63(a0);
          if (lane == 63) yv[rl_reg] = a0;
        }
        for (unsigned r2 = (unsigned)a.minv_lds_rows + XWAVES + wave; r2 < nrows; r2 += XWAVES) { // whatever is left streams from the L2
          const double acc = x_wave_sum63(stream_row(m0 + r2));
          if (lane == 63) yv[r2] = acc;
        }
      } else {
        for (unsigned rl = wave; rl < nrows; rl += XWAVES) {
          double acc = 0.0;
          if (rl < nlds) { // four columns per lane in flight; the pad of w is zero and the row's is finite: the terms behind m are exact zeros, added in the same order
            const double *row = mrow + (size_t)rl * a.m_pad + lane, *wl = wv + lane;
            for (unsigned c0 = 0; c0 < (unsigned)a.m_pad; c0 += 256) {
              double rv[4], wq[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) { const bool in = c0 + 64u * u < (unsigned)a.m_pad; rv[u] = in ? row[c0 + 64u * u] : 0.0; wq[u] = in ? wl[c0 + 64u * u] : 0.0; } // (uniform)
#pragma unroll
              for (int u = 0; u < 4; ++u) acc += rv[u] * wq[u];
            }
          } else acc = stream_row(m0 + rl);
          acc = x_wave_sum63(acc);
          if (lane == 63) yv[rl] = acc;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; y[q] = (i < m1) ? yv[i - m0] : 0.0; }
      XP_LAP(3)
    }
    // ---- back-substitution x = A'y - rhs_x (indirect.c:419-420) and u_t'h (abip.c:560) ----
    // u_t'h = y'h_y + (A'y - rhs_x)'h_x = y'(h_y + A h_x) - rhs_x'h_x: with A h_x formed once per solve (XcdArgs::hAh) every term is known BEFORE this exchange,
    // so the sum rides on it -- its granule is the flag -- and the scalar-only exchange round 4 ran behind the gather (one rendez-vous of a direct iteration's
    // five, ~1.2 us of c2's 15.2) is gone.  The sum is regrouped, like every sum of this kernel; the reference adds u_t[i] h[i] in index order.
    double dh[1] = {0.0};
    open(6);
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      if (i < m1) {
        x_putd<SA>(pm0, i * 8u, y[q]);
        if (solo) x_at(a.srhs, i) = y[q];
        else if (mode == XM_MAIN) x_at(up.ut, i) = y[q];
        dh[0] += y[q] * hy2[q];
      }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) dh[0] -= rhs_x[q] * hx[q]; // (zero beyond the owned columns)
    // (one XCD) The update's loads -- the old (u, v), the running sums, g, b / c, the scale factors of the owned elements: nothing the solve changes -- go out HERE,
    // behind this rank's flags: their round trip to the L2 overlaps with the wait for the other ranks and with the gather.  (Tried before the polling moved to the
    // wavefronts that own no elements: the polls then waited behind these loads, a loss.  Across XCDs the first wavefronts still poll: there the loads stay put.)
    const bool upd_next = (mode == XM_MAIN) && !solo;
    double wsy[RM], wsx[RN];
    x_publish<1, SA>(dh, red, psc, sc_off, tag);
    XLd Ly[RM], Lx[RN], Lt;
    // What the phases behind the update need of the owned elements rides in registers from here: the scale factors with this batch of loads, the new
    // (u, v) and the averaged v as they are computed -- the stopping test and the next right-hand side then start without a round trip to the L2 of their own
    // (the arrays are written all the same: everything outside this loop reads them there).
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      Ly[q] = XLd{0, 0, 0, 0, 0, 0, 0, 0};
      wsy[q] = 1.0;
      if (upd_next && i < m1) {
        Ly[q].v = x_at(up.v, i); if (up.half_update) Ly[q].u = x_at(up.u, i);
#ifdef XCD_RSUM
        Ly[q].ua = Rua_y[q]; Ly[q].va = Rva_y[q]; Ly[q].us = Rus_y[q]; Ly[q].vs = Rvs_y[q];
#else
        Ly[q].ua = x_at(up.u_avg, i); Ly[q].va = x_at(up.v_avg, i); Ly[q].us = x_at(up.u_sum, i); Ly[q].vs = x_at(up.v_sum, i);
#endif
        Ly[q].g = x_at(up.g, i); Ly[q].bc = x_at(up.b, i);
        if (a.wD) wsy[q] = x_at(a.wD, i);
      }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j2 = n0 + tb + q * XTB;
      Lx[q] = XLd{0, 0, 0, 0, 0, 0, 0, 0};
      wsx[q] = 1.0;
      if (upd_next && j2 < n1) {
        const unsigned qq = MP + j2;
        Lx[q].u = x_at(up.u, qq); Lx[q].v = x_at(up.v, qq);
#ifdef XCD_RSUM
        Lx[q].ua = Rua_x[q]; Lx[q].va = Rva_x[q]; Lx[q].us = Rus_x[q]; Lx[q].vs = Rvs_x[q];
#else
        Lx[q].ua = x_at(up.u_avg, qq); Lx[q].va = x_at(up.v_avg, qq); Lx[q].us = x_at(up.u_sum, qq); Lx[q].vs = x_at(up.v_sum, qq);
#endif
        Lx[q].g = x_at(up.g, qq); Lx[q].bc = x_at(up.c, j2);
        if (a.wE) wsx[q] = x_at(a.wE, j2);
      }
    }
    Lt = XLd{0, 0, 0, 0, 0, 0, 0, 0};
    if (upd_next && rank == 0 && t == 0) {
      Lt.u = x_at(up.u, tail); Lt.v = x_at(up.v, tail);
#ifdef XCD_RSUM
      Lt.ua = Rtl[0]; Lt.va = Rtl[1]; Lt.us = Rtl[2]; Lt.vs = Rtl[3];
#else
      Lt.ua = x_at(up.u_avg, tail); Lt.va = x_at(up.v_avg, tail); Lt.us = x_at(up.u_sum, tail); Lt.vs = x_at(up.v_sum, tail);
#endif
    }
    {
      double tx[NZ];
      x_mat<NZ>(gT, nt, tx);
      x_collect<1, CROSS>(w, G, tot, dhS);
      if (__builtin_expect(w.dead, 0)) return;
      double vt[NZ], tq[RN];
      x_gather<NZ>(pm0, ti, vt);
      x_rows<NZ, RN>(prod, flip, tx, vt, nt, st, et, tq);
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        zx[q] = 0.0;
        if (j2 < n1) zx[q] = tq[q] - rhs_x[q];
      }
    }
    if (!PCG) { XP_LAP(4) }
    if (__builtin_expect(solo || mode != XM_MAIN, 0)) break; // (not an ADMM iteration: see below the loop)
    // ---- element-wise update (k_admm_update): barrier prox, dual update, running sums, averages, statistics ----
    // Every load of the phase is issued before its first store: on this chip loads and stores share one in-order counter (vmcnt), so a load issued behind a
    // store is not handed over before that store has been acknowledged by the L2 -- read-modify-write stream after read-modify-write stream (the round-3
    // form: nine of them per element, which the compiler must keep in order because the arrays may alias) is a chain of that many L2 round trips.
    Stat sst = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double tau4[4] = {0.0, 0.0, 0.0, 0.0};
    double uay[RM], uax[RN], utq_t = 0.0; // the averaged u of the owned elements, the tau entry of u_t: kept for the stores behind the flags
#pragma unroll
    for (int q = 0; q < RM; ++q) uay[q] = 0.0;
#pragma unroll
    for (int q = 0; q < RN; ++q) uax[q] = 0.0;
    open(8);
    // (the old values of the owned elements -- Ly, Lx, Lt, the scale factors -- were requested behind the back-substitution's flags: see there)
    double nuy[RM], nvy[RM], nux[RN], nvx[RN], vacx[RN];
#pragma unroll
    for (int q = 0; q < RM; ++q) { nuy[q] = 0.0; nvy[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < RN; ++q) { nux[q] = 0.0; nvx[q] = 0.0; vacx[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + tb + q * XTB;
      if (i < m1) {
        double un, vn;
        const double uti = y[q];
        if (!up.half_update) { vn = Ly[q].v; un = uti - vn; }
        else { double vh = Ly[q].v + 0.5 * (Ly[q].u - uti); un = uti - vh; vn = vh + (un - uti); }
        const double ua = xs_y2(up, Ly[q], un, vn, sst);
        nuy[q] = un; nvy[q] = vn; uay[q] = ua;
        x_putd<SA>(pm0, i * 8u, un);
        if (avg_stats) x_putd<SA>(pm1, i * 8u, ua);
      }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j2 = n0 + tb + q * XTB;
      if (j2 < n1) {
        double un, vn;
        x_prox(up, Lx[q].u, Lx[q].v, zx[q], un, vn);
        const double ua = xs_x2(up, Lx[q], false, un, vn, sst);
        const double vac = (Lx[q].vs + vn) / up.dom;
        nux[q] = un; nvx[q] = vn; vacx[q] = vac; uax[q] = ua;
        x_putd<SA>(pn0, j2 * 8u, un);
        if (!PCG) x_putd<SA>(pnv, j2 * 8u, vn);
        if (avg_stats) x_putd<SA>(pn1, j2 * 8u, ua);
      }
    }
    if (rank == 0 && t == 0) { // the tau / kappa entry
      const double utq = tsum + dhS[0];
      double un, vn;
      x_prox(up, Lt.u, Lt.v, utq, un, vn);
      const double ua = xs_x2(up, Lt, true, un, vn, sst), va = (Lt.vs + vn) / up.dom;
      utq_t = utq;
      tau4[0] = un; tau4[1] = vn; tau4[2] = ua; tau4[3] = va;
      // the tau / kappa entries are nobody's sum: they travel as granules of rank 0 behind the sums' (slots 12..15), acknowledged before its flags go out
#pragma unroll
      for (int q = 0; q < 4; ++q) x_putg<SA>(psc, sc_off + (unsigned)(12 + q) * XSLOT, tau4[q], tag);
    }
    // (an exchange costs per scalar it carries -- a wavefront sum in every wavefront, a polling wavefront: the averaged iterate's four sums go
    //  out only on the iterations that test it, one in ten)
    // The state's own stores -- eight arrays per element, nobody else's business -- go out BEHIND this rank's flags: in front of them (round 4) the flags waited for
    // their acknowledgements too (one in-order counter), for nothing.  The polling wavefronts are the workgroup's last ones, which own no elements.
    // (One XCD only: c2 +1.7 %.  Across XCDs the first wavefronts poll -- see x_collect -- and would wait behind these stores: there the round-4 order stays.)
    auto store_state = [&]() {
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double un = nuy[q], vn = nvy[q];
          x_at(up.u, i) = un; x_at(up.v, i) = vn;
#ifdef XCD_RSUM
          Rua_y[q] = Ly[q].ua + un; Rva_y[q] = Ly[q].va + vn; Rus_y[q] = Ly[q].us + un; Rvs_y[q] = Ly[q].vs + vn;
#else
          x_at(up.u_avg, i) = Ly[q].ua + un; x_at(up.v_avg, i) = Ly[q].va + vn;
          x_at(up.u_sum, i) = Ly[q].us + un; x_at(up.v_sum, i) = Ly[q].vs + vn;
#endif
          x_at(up.u_avgc, i) = uay[q]; x_at(up.v_avgc, i) = (Ly[q].vs + vn) / up.dom;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const unsigned qq = MP + j2;
          const double un = nux[q], vn = nvx[q];
          x_at(up.ut, qq) = zx[q];
          x_at(up.u, qq) = un; x_at(up.v, qq) = vn;
#ifdef XCD_RSUM
          Rua_x[q] = Lx[q].ua + un; Rva_x[q] = Lx[q].va + vn; Rus_x[q] = Lx[q].us + un; Rvs_x[q] = Lx[q].vs + vn;
#else
          x_at(up.u_avg, qq) = Lx[q].ua + un; x_at(up.v_avg, qq) = Lx[q].va + vn;
          x_at(up.u_sum, qq) = Lx[q].us + un; x_at(up.v_sum, qq) = Lx[q].vs + vn;
#endif
          x_at(up.u_avgc, qq) = uax[q]; x_at(up.v_avgc, qq) = vacx[q];
        }
      }
      if (rank == 0 && t == 0) {
        x_at(up.ut, tail) = utq_t;
        x_at(up.u, tail) = tau4[0]; x_at(up.v, tail) = tau4[1];
#ifdef XCD_RSUM
        Rtl[0] = Lt.ua + tau4[0]; Rtl[1] = Lt.va + tau4[1]; Rtl[2] = Lt.us + tau4[0]; Rtl[3] = Lt.vs + tau4[1];
#else
        x_at(up.u_avg, tail) = Lt.ua + tau4[0]; x_at(up.v_avg, tail) = Lt.va + tau4[1];
        x_at(up.u_sum, tail) = Lt.us + tau4[0]; x_at(up.v_sum, tail) = Lt.vs + tau4[1];
#endif
        x_at(up.u_avgc, tail) = tau4[2]; x_at(up.v_avgc, tail) = tau4[3];
      }
    };
    double q6[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double S13[13];
    if (avg_stats) {
      double s9[9] = {sst.wg, sst.nu, sst.nv, sst.cx, sst.by, sst.nua, sst.nva, sst.cxa, sst.bya}, S9[9];
      if (CROSS) store_state();
      x_publish<9, SA>(s9, red, psc, sc_off, tag);
      if (!CROSS) store_state();
      x_collect<9, CROSS>(w, G, tot, S9);
#pragma unroll
      for (int q = 0; q < 9; ++q) S13[q] = S9[q];
    } else {
      double s5[5] = {sst.wg, sst.nu, sst.nv, sst.cx, sst.by}, S5[5];
      if (CROSS) store_state();
      x_publish<5, SA>(s5, red, psc, sc_off, tag);
      if (!CROSS) store_state();
      x_collect<5, CROSS>(w, G, tot, S5);
#pragma unroll
      for (int q = 0; q < 5; ++q) S13[q] = S5[q];
      S13[5] = 0.0; S13[6] = 0.0; S13[7] = 0.0; S13[8] = 0.0;
    }
    if (__builtin_expect(w.dead, 0)) return;
    // ---- stopping-test products (k_q_both): A u_x and A'u_y, residual sums; A'u_y is also the next solve's warm-start product ----
    // One round trip for everything the phase reads from the L2: rank 0's tau granules (there since its flags are), the matrix values and the gathered entries
    // of both products; b, c, the scale factors, v and the averaged v of the owned elements come from the update in registers.
    double axk[NZ], uxg[NZ], vxg[NZ]; // (direct) A's values, the gathered new u_x and v_x: the next right-hand side's x entries are formed from them below
    {
      u32x4 tg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) tg[q] = x_ldg(psc, (unsigned)(12 + q) * XSLOT); // (rank 0's)
      double ax[NZ], va[NZ], tx[NZ], vt[NZ];
      x_mat<NZ>(gA, na, ax);
      x_gather<NZ>(pn0, ai, va);
      if (!PCG) x_gather<NZ>(pnv, ai, vxg);
      x_mat<NZ>(gT, nt, tx);
      x_gather<NZ>(pm0, ti, vt);
#pragma unroll
      for (int u = 0; u < NZ; ++u) { axk[u] = ax[u]; uxg[u] = va[u]; if (PCG) vxg[u] = 0.0; }
      double pri[RM];
      x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, pri);
      x_rows<NZ, RN>(prod, flip, tx, vt, nt, st, et, aty);
#pragma unroll
      for (int q = 0; q < 4; ++q) S13[9 + q] = x_uni(x_val(tg[q]));
      const double tau = S13[9];
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double e = pri[q] - Ly[q].bc * tau;
          const double sc = wsy[q] * wsy[q];
          q6[0] += e * e; q6[1] += (e * e) * sc; q6[2] += (pri[q] * pri[q]) * sc;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const double drj = aty[q] + nvx[q], e = drj - Lx[q].bc * tau;
          const double sc = wsx[q] * wsx[q];
          q6[3] += e * e; q6[4] += (e * e) * sc; q6[5] += (drj * drj) * sc;
        }
      }
    }
    if (avg_stats) { // the same on the averaged iterate (its entries went out with the same exchange)
      double pri[RM], atya[RN];
      {
        double ax[NZ], va[NZ], tx[NZ], vt[NZ];
        x_mat<NZ>(gA, na, ax);
        x_gather<NZ>(pn1, ai, va);
        x_mat<NZ>(gT, nt, tx);
        x_gather<NZ>(pm1, ti, vt);
        x_rows<NZ, RM>(prod, flip, ax, va, na, sa, ea, pri);
        x_rows<NZ, RN>(prod, flip, tx, vt, nt, st, et, atya);
      }
      const double tau = S13[11];
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double e = pri[q] - Ly[q].bc * tau;
          const double sc = wsy[q] * wsy[q];
          q6[6] += e * e; q6[7] += (e * e) * sc; q6[8] += (pri[q] * pri[q]) * sc;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const double drj = atya[q] + vacx[q], e = drj - Lx[q].bc * tau;
          const double sc = wsx[q] * wsx[q];
          q6[9] += e * e; q6[10] += (e * e) * sc; q6[11] += (drj * drj) * sc;
        }
      }
    }
    if (!PCG) { XP_LAP(5) }
    double Q[12];
    if (!PCG) {
      // Direct back-end: the NEXT iteration's w = rhs_y + A rhs_x rides on this exchange with the sums of the stopping test.  Everything the next right-hand side
      // needs (S_WG, the tau entries, the new iterate) is known since the update's exchange; its y entries are the owner's, and the x entries a row of A gathers
      // are formed by the gathering thread itself from the new (u_x, v_x) it gathered for the stopping test (v_x went out beside u_x) and from h at those columns --
      // k_rhs's expressions on the same numbers, the same bits as the owner's.  So the exchange that handed rhs_x round in round 4 is gone: three rendez-vous per
      // iteration (w + stopping-test sums, y + u_t'h, the update).  If the exit test below ends the inner loop, w is simply not used.
      const double ts_n = S13[9] + S13[10];
      const double cf_n = (S13[0] - ts_n * a.g_th) / (a.g_th + 1.0);
      // (build_rhs with the iterate and h in registers: the same expressions, no loads)
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        rhs_y[q] = 0.0;
        if (i < m1) {
          double r = (nuy[q] + nvy[q]) * rho;
          r += -ts_n * hy[q];
          r += -cf_n * hy[q];
          rhs_y[q] = r;
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        rhs_x[q] = 0.0;
        if (j2 < n1) {
          double r = nux[q] + nvx[q];
          r += -ts_n * hx[q];
          r += -cf_n * hx[q];
          rhs_x[q] = -r;
        }
      }
      double rxg[NZ], sA[RM];
#pragma unroll
      for (int u = 0; u < NZ; ++u) {
        double r = uxg[u] + vxg[u];
        r += -ts_n * hA[u];
        r += -cf_n * hA[u];
        rxg[u] = -r;
      }
      x_rows<NZ, RM>(prod, flip, axk, rxg, na, sa, ea, sA);
      open(5);
#pragma unroll
      for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; if (i < m1) x_putd<SA>(pm0, i * 8u, rhs_y[q] + sA[q]); }
      have_w = true;
    } else open(9);
    if (avg_stats) {
      x_publish<12, SA>(q6, red, psc, sc_off, tag);
      x_collect<12, CROSS>(w, G, tot, Q);
    } else {
      double q[6] = {q6[0], q6[1], q6[2], q6[3], q6[4], q6[5]}, Q6[6];
      x_publish<6, SA>(q, red, psc, sc_off, tag);
      x_collect<6, CROSS>(w, G, tot, Q6);
#pragma unroll
      for (int k = 0; k < 6; ++k) { Q[k] = Q6[k]; Q[6 + k] = 0.0; }
    }
    if (__builtin_expect(w.dead, 0)) return;
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
    stats_valid = true; avg_stats_last = avg_stats ? 1 : 0;
    if (!PCG) { XP_LAP(6) }
    XQ_ADD(1, xq_t)
    if (__builtin_expect(metric < thr, 0)) { // abip.c:2173-2188: the inner loop is left (j is not advanced)
      if (!whole) { halt = 1; leave = 1; break; }
      stage = XS_OUTER_END;
      break;
    }
    if (__builtin_expect(final_check != 0, 0)) { // abip.c:2190-2213
      __syncthreads(); // outs complete
      const long k = a.fc.k0 + ran;
      const long oi = (long)csr(CS_OI);
      if (x_converged(outs, avg_crit, a.fc, oi, k) || k + 1 >= a.fc.max_admm || (whole && oi + 1 >= xo.max_ipm)) { halt = 2; csw(CS_REASON, (double)XR_FINAL); if (whole) ++jj; leave = 1; break; }
    }
    ++jj;
    if (whole) { // abip.c:2131; abip_hip_step's budget; abip.c:608-609: a restart is the launch path's
      if (__builtin_expect(jj >= xo.inner_stopper, 0)) { stage = XS_OUTER_END; break; }
      if (__builtin_expect(ran >= xo.max_steps, 0)) { csw(CS_REASON, (double)XR_STEPS); leave = 1; break; }
      const long kk = a.fc.k0 + ran;
      if (__builtin_expect(kk >= xo.restart_thresh, 0)) {
        if ((jj + 1 - (long)csr(CS_FRE)) % xo.restart_fre == 0) { csw(CS_REASON, (double)XR_RESTART); leave = 1; break; }
      }
    } else if (ran >= a.max_iters) { leave = 1; break; }
    } // (the inner loop)
    if (leave) break;
    if (__builtin_expect(solo, 0)) { // the solution goes back in place; u_t'h where the launch path's consumers re-reduce the partial table
#pragma unroll
      for (int q = 0; q < RN; ++q) { const unsigned j2 = n0 + tb + q * XTB; if (j2 < n1) x_at(a.srhs, MP + j2) = zx[q]; }
      if (rank == 0) for (unsigned e = t; e < (unsigned)a.npart; e += XTB) a.part[S_DH * MAXNB + e] = (e == 0) ? dhS[0] : 0.0;
      break;
    }

    // ================================================================================================================================
    // look-ahead steps of the Barzilai-Borwein search (adaptive.c:101-251): the prox on scratch vectors, the five inner products, the next penalty
    // ================================================================================================================================
    if (__builtin_expect(mode != XM_MAIN, 0)) {
      const double mu = csr(CS_MU);
      double bb_prev = csr(CS_BBPREV);
      const double bup_t = csr(CS_BUPT), bvp_t = csr(CS_BVPT);
      const double mob = mu / bb_prev, al = up.alpha;
      const double utq_t = tsum + dhS[0]; // adaptive.c:99, 131
      if (mode == XM_BB1) { // (u_prev, v_prev) -> (u, v)
#pragma unroll
        for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; if (i < m1) x_at(xo.a_u, i) = y[q] - x_at(xo.a_vp, i); }
#pragma unroll
        for (int q = 0; q < RN; ++q) {
          const unsigned j2 = n0 + tb + q * XTB;
          if (j2 < n1) {
            double un, vn;
            x_bbstep(al, mob, zx[q], x_at(xo.a_up, MP + j2), x_at(xo.a_vp, MP + j2), un, vn);
            x_at(xo.a_u, MP + j2) = un; x_at(xo.a_v, MP + j2) = vn;
          }
        }
        double un, vn;
        x_bbstep(al, mob, utq_t, bup_t, bvp_t, un, vn);
        csw(CS_BUT, un); csw(CS_BVT, vn);
        mode = XM_BB2; need_pre = true;
        XQ_ADD(2, xq_t)
        continue;
      }
      // (u, v) -> (u_next, v_next), and the inner products of the three difference vectors
      const double bu_t = csr(CS_BUT), bv_t = csr(CS_BVT);
      double d5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < RM; ++q) {
        const unsigned i = m0 + tb + q * XTB;
        if (i < m1) {
          const double vv = x_at(xo.a_v, i), vnn = x_at(xo.a_vn, i); // (the y blocks of v and v_next are never written: adaptive.c:118-121, 147-150)
          const double un = y[q] - vv;
          x_at(xo.a_un, i) = un;
          x_bbdots(al, x_at(xo.a_u, i), vv, un, vnn, x_at(xo.a_vp, i), d5);
        }
      }
#pragma unroll
      for (int q = 0; q < RN; ++q) {
        const unsigned j2 = n0 + tb + q * XTB;
        if (j2 < n1) {
          const double uo = x_at(xo.a_u, MP + j2), vo = x_at(xo.a_v, MP + j2);
          double un, vn;
          x_bbstep(al, mob, zx[q], uo, vo, un, vn);
          x_at(xo.a_un, MP + j2) = un; x_at(xo.a_vn, MP + j2) = vn;
          x_bbdots(al, uo, vo, un, vn, x_at(xo.a_vp, MP + j2), d5);
        }
      }
      double bun_t, bvn_t; // the tau entry of (u_next, v_next)
      x_bbstep(al, mob, utq_t, bu_t, bv_t, bun_t, bvn_t);
      if (rank == 0 && t == 0) x_bbdots(al, bu_t, bv_t, bun_t, bvn_t, bvp_t, d5);
      open(12);
      x_publish<5, SA>(d5, red, psc, sc_off, tag);
      double D5[5];
      x_collect<5, CROSS>(w, G, tot, D5);
      if (__builtin_expect(w.dead, 0)) return;
      const int bb_it = (int)csr(CS_BBIT);
      const double bb_tot = csr(CS_BBTOT);
      __syncthreads(); // (everybody has read the state thread 0 overwrites below)
      csw(CS_BBTOT, bb_tot + 1.0);
      double bnew = 0.0;
      const int act = __builtin_amdgcn_readfirstlane(lp_bb_beta(D5[0], D5[1], D5[2], D5[3], D5[4], xo.eps_cor, xo.eps_pen, bb_prev, bnew));
      bnew = x_uni(bnew);
      if (act != 0 && bb_it + 1 < xo.lookback) { // go on from (u, v) (adaptive.c:230-247)
        if (act == 1) bb_prev = bnew;
        const double mob2 = mu / bb_prev;
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          const unsigned i = m0 + tb + q * XTB;
          if (i < m1) { x_at(xo.a_up, i) = x_at(xo.a_u, i); x_at(xo.a_vp, i) = x_at(xo.a_v, i); }
        }
#pragma unroll
        for (int q = 0; q < RN; ++q) {
          const unsigned j2 = n0 + tb + q * XTB;
          if (j2 < n1) {
            const double uo = x_at(xo.a_u, MP + j2);
            x_at(xo.a_up, MP + j2) = uo;
            x_at(xo.a_vp, MP + j2) = (act == 1) ? mob2 / uo : x_at(xo.a_v, MP + j2);
          }
        }
        csw(CS_BBPREV, bb_prev); csw(CS_BUPT, bu_t); csw(CS_BVPT, (act == 1) ? mob2 / bu_t : bv_t); csw(CS_BBIT, (double)(bb_it + 1));
        if (act == 2 && xo.bb_reuse) {
          // The penalty did not change (beta = beta_prev: by far the most frequent outcome).  The next look-ahead's first step would start from (u_prev, v_prev) = (u, v)
          // with the same beta_prev, the same warm start and the same tolerance: it is this look-ahead's second step over again -- the reference runs that solve a
          // second time and gets the same bits.  Its results are moved instead: u <- u_next, v[x, tau] <- v_next[x, tau] (the y block of v is never written,
          // adaptive.c:118-121), and the loop goes on at the second step.  The skipped solve is counted as the reference counts it.
#pragma unroll
          for (int q = 0; q < RM; ++q) { const unsigned i = m0 + tb + q * XTB; if (i < m1) x_at(xo.a_u, i) = x_at(xo.a_un, i); }
#pragma unroll
          for (int q = 0; q < RN; ++q) {
            const unsigned j2 = n0 + tb + q * XTB;
            if (j2 < n1) { x_at(xo.a_u, MP + j2) = x_at(xo.a_un, MP + j2); x_at(xo.a_v, MP + j2) = x_at(xo.a_vn, MP + j2); }
          }
          csw(CS_BUT, bun_t); csw(CS_BVT, bvn_t);
          cg_total += last_cg;
          csw(CS_CGSKIP, csr(CS_CGSKIP) + (double)last_cg);
          mode = XM_BB2; need_pre = true;
          XQ_ADD(2, xq_t)
          continue;
        }
        mode = XM_BB1; need_pre = true;
        XQ_ADD(2, xq_t)
        continue;
      }
      // the search is over (adaptive.c:253): w->beta; reinitialize_vars(2) (abip.c:2291) on the vectors the outer end rescaled
      csw(CS_BETA, bnew);
      {
        const bool av = csr(CS_AV) != 0.0;
        double *RU = av ? up.u_avgc : up.u, *RV = av ? up.v_avgc : up.v;
        const double sq = sqrt(1.0 / xo.sigma);
#pragma unroll
        for (int q = 0; q < RN; ++q) { const unsigned j2 = n0 + tb + q * XTB; if (j2 < n1) { x_at(RU, MP + j2) = sq * x_at(RU, MP + j2); x_at(RV, MP + j2) = sq * x_at(RV, MP + j2); } }
        const double r_ut = x_uni(sq * csr(CS_RUT)), r_vt = x_uni(sq * csr(CS_RVT));
        if (rank == 0 && t == 0) { x_at(RU, tail) = r_ut; x_at(RV, tail) = r_vt; }
        __syncthreads(); // (every wavefront has read CS_RUT / CS_RVT before thread 0 rewrites them)
        csw(CS_RUT, r_ut); csw(CS_RVT, r_vt);
        if (!av) { u_tau = r_ut; v_tau = r_vt; }
      }
      stage = XS_OUTER_BEGIN;
      XQ_ADD(2, xq_t)
      continue;
    }
  }
  XQ_ADD(0, xq_0)
  __syncthreads();
  if (rank == 0 && t < 96 && ran > 0) a.ctl->out[t] = outs[t];
  if (rank == 0 && t == 0) {
    Ctl *c = a.ctl;
    if (!solo) { c->metric = metric; c->avg_crit = avg_crit; c->it_count = c->it_count + ran; c->halt = halt; }
    c->cg_it = last_cg; c->cg_done = 1;
    c->xcd_cg_total = cg_total;
    if (whole) {
      XcdOut &o = c->xo;
      o.phase = (int)cs[CS_PHASE]; o.reason = (int)cs[CS_REASON]; o.final_check = final_check; o.avg_crit = avg_crit; o.stats_valid = stats_valid ? 1 : 0; o.avg_stats = avg_stats_last;
      o.outer_done = (int)cs[CS_ODONE]; o.log_n = (int)cs[CS_LOGN]; o.last_cg = last_cg; o.bb_lookaheads = (int)cs[CS_BBTOT];
      o.i = (long)cs[CS_OI]; o.j = jj; o.k = a.fc.k0 + ran; o.ran = ran; o.solves = (long)ran + 2 * (long)cs[CS_BBTOT]; o.cg_total = cg_total; o.cg_skipped = (long)cs[CS_CGSKIP];
      o.mu = cs[CS_MU]; o.beta = cs[CS_BETA]; o.dyn_sigma = cs[CS_DYNS];
    }
    a.xstat[1] = (int)(tag - a.tag0);
  }
#ifdef XCD_HALO_PROBE
  if (hp_sink == 123.456789) a.xstat[1] = -1; // (keeps the probe's sums alive)
#endif
#ifdef XCD_RSUM
  if (!solo) { // the sums back to their arrays (the host reads them: restarts, the averaged iterate, the next launch)
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const unsigned i = m0 + t + q * XTB;
      if (i < m1) { x_at(up.u_avg, i) = Rua_y[q]; x_at(up.v_avg, i) = Rva_y[q]; x_at(up.u_sum, i) = Rus_y[q]; x_at(up.v_sum, i) = Rvs_y[q]; }
    }
#pragma unroll
    for (int q = 0; q < RN; ++q) {
      const unsigned j2 = n0 + t + q * XTB;
      if (j2 < n1) { const unsigned qq = MP + j2; x_at(up.u_avg, qq) = Rua_x[q]; x_at(up.v_avg, qq) = Rva_x[q]; x_at(up.u_sum, qq) = Rus_x[q]; x_at(up.v_sum, qq) = Rvs_x[q]; }
    }
    if (rank == 0 && t == 0) { x_at(up.u_avg, tail) = Rtl[0]; x_at(up.v_avg, tail) = Rtl[1]; x_at(up.u_sum, tail) = Rtl[2]; x_at(up.v_sum, tail) = Rtl[3]; }
  }
#endif
  XP_DUMP
}

} // namespace abip
