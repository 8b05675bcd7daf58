// solver.hip -- libabip_hip.so: the reference's C entry points (include/abip.h) and the device-level ABI
// (include/abip_hip.h) on top of the HIP kernels in dev_kernels.h / dev_sptrsv.h.
//
// Control structure.  The reference's solver is one nested loop on one CPU thread
// (src/abip-lp/src/abip.c:2102-2294).  Here every vector lives in HBM for the whole solve; the host only
//   * enqueues the kernel chain of one inner ADMM iteration on a private stream,
//   * reads back ONE small control block per iteration (the finalised reductions), and
//   * takes the reference's scalar decisions (inner stop, convergence, mu update, BB penalty).
// Inside the PCG loop nothing comes back to the host: convergence is decided on the device and the remaining
// kernels of an enqueued chunk turn into no-ops (dev_kernels.h, cg_converged).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/abip.h"
#include "../../include/abip_hip.h"
#include "dev_kernels.h"
#include "dev_ldl.h"
#include "dev_xcd.h"
#include "dist_internal.h"
#include "host_setup.h"
#include "dev_host_util.h"

using namespace abip;
using namespace abip::hostutil;

namespace {

int g_linsys = -1; // -1: not chosen yet -> environment / default
int g_copy_a = -1; // -1: environment / default (copy)
// The reference scales the caller's A in place unless it is built with COPYAMATRIX (abip.c:1799-1807), which its mex build is
// (make_abip.m:13): Matlab owns A.  A shared library cannot know which build it replaces, so the safe behaviour is the default:
// work on a private copy of the values and never touch the caller's matrix; abip_hip_set_copy_a_matrix(0) / ABIP_HIP_COPYAMATRIX=0
// gives the in-place scaling (and the un-scaling in abip_finish, abip.c:2310-2317) of the plain C build.
bool copy_a_matrix() {
  if (g_copy_a >= 0) return g_copy_a != 0;
  const char *e = getenv("ABIP_HIP_COPYAMATRIX");
  return !(e && atoi(e) == 0);
}

// ---- multi-GPU context (one process per GPU; set before abip_init) --------------------------------------
// RCCL is bound with dlopen so that the library neither needs it for single-GPU use nor clashes with the copy a host
// program (e.g. PyTorch) may already have loaded under the same soname.
typedef void *rcclComm_t;
struct Uid128 { char b[128]; }; // ncclUniqueId
struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(rcclComm_t *, int, Uid128, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*CommAbort)(rcclComm_t) = nullptr;       // optional
  int (*CommCount)(rcclComm_t, int *) = nullptr; // optional
};
struct PeerHost { // dev_peer.h: the hand-rolled exchange over peer-mapped mailboxes
  abip::PeerCtx ctx{};
  void *mine = nullptr; void *mapped[abip::PEER_MAX] = {nullptr};
  unsigned *sync = nullptr; int *hstatus = nullptr;
  unsigned long long epoch = 0;
  long cap = 0;
  bool fused = true;      // the product kernels of the sharded PCG push their rows themselves (ABIP_HIP_PEER_FUSED=0: every all-reduce is the stand-alone kernel)
  bool fine = false;      // the mailbox is fine-grained device memory (else plain device memory: see dev_peer.h "Coherence")
  bool cross_device = false; // some peer's mailbox lives on another device (a real node; false when the ranks share one GPU)
};
struct DistCtx {
  int kind = 0; // 0 none, 1 RCCL, 2 host callback (tests), 3 peer-mapped mailboxes (dev_peer.h)
  PeerHost peer;
  int rank = 0, world = 1;
  rcclComm_t comm = nullptr;
  abip_hip_allreduce_fn fn = nullptr;
  void *fn_ctx = nullptr;
  RcclApi api;
};
DistCtx g_dist;
bool g_dist_aborted = false;

// A rank that fails inside a sharded solve stops issuing collectives; its peers would wait in ncclAllReduce for ever.  Abort the
// communicator so that they fail too, and leave tearing the job down to the launcher: the caller must exit non-zero, not retry.
void dist_abort(const char *why) {
  if (g_dist.kind != 1 || !g_dist.comm || g_dist_aborted) return;
  fprintf(stderr, "abip_hip: rank %d aborts the RCCL communicator (%s); exit non-zero, do not retry in-process\n", g_dist.rank, why);
  if (g_dist.api.CommAbort) g_dist.api.CommAbort(g_dist.comm);
  g_dist.comm = nullptr; g_dist_aborted = true;
}

bool load_rccl(RcclApi &a) {
  if (a.handle) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *nm : names) { a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (a.handle) break; }
  if (!a.handle) { fprintf(stderr, "abip_hip: cannot load librccl (%s)\n", dlerror()); return false; }
  a.GetUniqueId = (int (*)(void *))dlsym(a.handle, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(rcclComm_t *, int, Uid128, int))dlsym(a.handle, "ncclCommInitRank");
  a.AllReduce = (int (*)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t))dlsym(a.handle, "ncclAllReduce");
  a.CommDestroy = (int (*)(rcclComm_t))dlsym(a.handle, "ncclCommDestroy");
  a.CommAbort = (int (*)(rcclComm_t))dlsym(a.handle, "ncclCommAbort");
  a.CommCount = (int (*)(rcclComm_t, int *))dlsym(a.handle, "ncclCommCount");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) { fprintf(stderr, "abip_hip: librccl lacks a required symbol\n"); return false; }
  return true;
}

int chosen_linsys() {
  if (g_linsys >= 0) return g_linsys;
  const char *e = getenv("ABIP_HIP_LINSYS");
  if (e && (!strcmp(e, "indirect") || !strcmp(e, "pcg") || !strcmp(e, "1"))) return ABIP_HIP_LINSYS_INDIRECT;
  return ABIP_HIP_LINSYS_DIRECT;
}

struct Resid : abip::LpResid { // ABIPResiduals, include/abip.h:178-195 (the numbers: lp_scalars.h)
  abip_int last_ipm_iter = -1, last_admm_iter = -1;
  Resid() : abip::LpResid{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
};

enum Phase { PH_IDLE, PH_OUTER_BEGIN, PH_INNER, PH_OUTER_END, PH_DONE };

const char *kHeader[] = {" ipm iter ", " admm iter ", "     mu ", " pri res ", " dua res ", " rel gap ", " pri obj ", " dua obj ", " kap/tau ", " time (s)"};
constexpr int kHSpace = 9, kHeaderLen = 10, kLineLen = 150;
void print_line(char c) { for (int i = 0; i < kLineLen; ++i) putchar(c); putchar('\n'); }

constexpr double EPS_TOL = 1E-18; // glbopts.h:157
inline double safediv_pos(double x, double y) { return y < EPS_TOL ? x / EPS_TOL : x / y; }

} // namespace

namespace abip { // dist_internal.h: shared with the conic solver
DistInfo dist_info() { return DistInfo{g_dist.kind, g_dist.rank, g_dist.world}; }
void dist_abort_from(const char *why) { dist_abort(why); }
int dist_allreduce(double *buf, size_t count, hipStream_t s, std::vector<double> &hstage) {
  if (g_dist.kind == 1) {
    if (!g_dist.comm) return -1; // aborted
    const int rc = g_dist.api.AllReduce(buf, buf, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, g_dist.comm, s);
    if (rc != 0) { fprintf(stderr, "abip_hip: ncclAllReduce failed (%d)\n", rc); return -1; }
    return 0;
  }
  if (g_dist.kind == 3) { // dev_peer.h: one launch, every chunk reduced in one place in rank order
    PeerHost &p = g_dist.peer;
    if ((long)count > p.cap) { fprintf(stderr, "abip_hip: all-reduce of %zu doubles exceeds the mailbox capacity %ld (abip_hip_dist_peer_prepare)\n", count, p.cap); return -1; }
    if (*p.hstatus) { fprintf(stderr, "abip_hip: rank %d: an earlier peer exchange gave up waiting for rank %d\n", g_dist.rank, *p.hstatus - 1); return -1; }
    ++p.epoch;
    hipLaunchKernelGGL(abip::k_peer_allreduce, dim3(64), dim3(256), 0, s, p.ctx, buf, (long)count, p.epoch);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (g_dist.kind != 2) return -1;
  // host-staged callback (test backend): D2H, reduce on the host through the caller's collective, H2D
  hstage.resize(count);
  HIP_OK(hipMemcpyAsync(hstage.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  g_dist.fn(g_dist.fn_ctx, hstage.data(), (long)count);
  HIP_OK(hipMemcpyAsync(buf, hstage.data(), sizeof(double) * count, hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}
} // namespace abip

// One-XCD persistent launch for cache-resident LPs (dev_xcd.h): what abip_init prepares when the problem fits it.
struct XcdPlan {
  bool on = false;
  int G = 32, nxcd = 1, NZ = 0, RM = 0, RN = 0;
  unsigned tickets_used = 0; // tickets drawn by the launches so far (G per launch)
  abip::hostutil::DBuf<int> mb, nb, xstat;
  abip::hostutil::DBuf<unsigned> tickets;
  abip::hostutil::DBuf<double> xn0, xn1, xm0, xm1, xnv; // (xnv: direct variant, the new v_x beside the new u_x)
  abip::hostutil::DBuf<abip::u32x4> sc;
  abip::hostutil::DBuf<double> tolf, Minv;
  double *htolf = nullptr; int *hstat = nullptr; // pinned
  long ldM = 0;
  int n_pad = 0, m_pad = 0, max_batch = 2048, minv_lds_rows = 0;
  unsigned tag = 0, launches = 0;
  size_t lds = 0;
  const void *kern = nullptr;
  long batches = 0, exchanges = 0;
  // launches that span outer iterations (dev_xcd.h XcdOuter; ABIP_HIP_XCD_OUTER=0: one batch of inner iterations per launch, as in round 3)
  bool outer = true;
  static constexpr int MU_TAB = 64, LOG_CAP = 64;
  abip::hostutil::DBuf<double> mu_tab, xlog;
  double *hmu_tab = nullptr, *hlog = nullptr; // pinned
  long outer_done = 0, lookaheads = 0, whole_launches = 0;
  double its_per_ms = 0.0;   // measured rate of the launches so far: sizes the next launch's iteration budget (a launch is kept to about a second)
  double ticks_per_ms = 1e5; // wall_clock64
  // what a launch needs to be abandoned: the iterate and its running sums as they were when it started (restored when a wait gives up; the launch path goes on from there)
  abip::hostutil::DBuf<double> snap; size_t snap_len = 0;
  long giveups = 0;
  int desert_at = -1; // fault injection (libabip_hip_hooks.so): the launch number whose last rank leaves at once
  void release() {
    mb.release(); nb.release(); xstat.release(); tickets.release(); xn0.release(); xn1.release(); xm0.release(); xm1.release(); xnv.release(); sc.release(); tolf.release(); Minv.release();
    mu_tab.release(); xlog.release(); snap.release();
    if (htolf) (void)hipHostFree(htolf);
    if (hstat) (void)hipHostFree(hstat);
    if (hmu_tab) (void)hipHostFree(hmu_tab);
    if (hlog) (void)hipHostFree(hlog);
    htolf = nullptr; hstat = nullptr; hmu_tab = nullptr; hlog = nullptr; on = false;
  }
};

struct ABIP_WORK {
  // ---- problem / settings ------------------------------------------------------------------
  abip_int m = 0, n = 0;
  int MP = 0, LV = 0; // x offset inside an l-vector, allocated length
  int NB = 1;         // persistent grid size == partials per slot
  int linsys = ABIP_HIP_LINSYS_DIRECT;
  ABIPSettings *stgs = nullptr;
  ABIPMatrix *A = nullptr; // the matrix being scaled: the caller's (in place, until abip_finish) or Aown; null once a private copy is no longer needed
  ABIPMatrix Aown{}; std::vector<double> Aown_x; // private copy of the values (COPYAMATRIX behaviour, the default)
  double sp = 0;
  std::vector<double> D, E; // host copies of the scalings
  double mean_norm_row_A = 0, mean_norm_col_A = 0;
  double sc_b = 1, sc_c = 1, nm_b = 0, nm_c = 0, g_th = 0;
  // ---- algorithm scalars (struct ABIP_WORK of the reference) ----------------------------
  double sigma = 0, gamma = 0, mu = 1, beta = 1;
  abip_int final_check = 0, double_check = 0, fre_old = 0;
  // ---- device state --------------------------------------------------------------------------
  hipStream_t stream = nullptr;
  DevCsr dAt; // CSC of A read as CSR of A' (n rows): y_n = A' x_m
  DevCsr dA;  // explicit CSR of A (m rows):          y_m = A x_n
  DBuf<double> u, v, ut, u_avg, v_avg, u_sum, v_sum, u_avgc, v_avgc, h, g, b, c, wD, wE;
  DBuf<double> cg_p, cg_r, cg_Gp, cg_z, cg_M, cg_tmp; // indirect.h:14-29
  DBuf<double> cg_pair; // 2 n: (rhs_x[j], (A's)[j]) side by side for the one-gather PCG set-up (k_cg_init_A)
  // A'u_y of the current iterate, kept by the back-substitution (k_post_At) of the iteration that produced it: with v_y == 0 the new y block is the solve's
  // (abip.c:731-734), so the stopping test's A'u_y and the next solve's warm-start product are that vector -- one product per iteration instead of three.
  // Valid from a plain iteration of the launch path (one GPU, PCG, no restart, no half update) until something else writes u_y.
  DBuf<double> aty; bool aty_valid = false, aty_on = true;
  DBuf<double> aty_bb; // the same for the look-ahead solves of the streamed Barzilai-Borwein search (its own buffer: `aty` stays good for the iteration behind the search)
  DBuf<double> hAh; // persistent launch: h_y + A h_x (m): u_t'h = y'(h_y + A h_x) - rhs_x'h_x is known BEFORE the back-substitution's exchange and rides on it (dev_xcd.h)
  // streamed iterations of the launch path (admm_stream_pcg): two pinned mirrors of the control block, written by k_finalize_stream, and the events behind them
  Ctl *hmir[2] = {nullptr, nullptr}; hipEvent_t mir_ev[2] = {nullptr, nullptr}; bool stream_on = true, bb_stream_on = true;
  long stream_stalls = 0, stream_iters = 0; int cg_steady = 0; // (iterations since the PCG count last changed)
  DBuf<double> a_up, a_vp, a_ut, a_u, a_v, a_utn, a_un, a_vn; // adaptive.c:13-32 (the three delta vectors are never stored)
  DBuf<double> part;
  DBuf<Ctl> ctl;
  Ctl *hctl = nullptr; // pinned mirror
  // direct
  DevLdl ldl;
  XcdPlan xcd;
  bool xcd_solves = true; // ABIP_HIP_XCD_SOLVES=0: the stand-alone solves (set-up, BB look-ahead) stay on the launch path
  bool host_outer_once = false; // the persistent launch handed this outer end (abip.c:2217-2293) to the host
  // ---- loop state (locals of ABIP(solve)) ------------------------------------------------
  Phase phase = PH_IDLE;
  abip_int i = 0, j = 0, k = 0, inner_stopper = 0;
  bool wg_valid = false;
  int it_seen = 0;    // ctl.it_count at the last control read
  int batch = 4;      // iterations enqueued per control read (direct back-end)
  bool fuse_small = false; // small direct systems on one GPU: rhs build and u_t'h inside the solve kernels
  bool batch_ok = true; // ABIP_HIP_BATCH=0 forces one control read per iteration
  Resid r;
  abip_int status = 0;
  double t_solve0 = 0, cpu0 = 0;
  bool stats_valid = false, avg_stats_valid = false; // ctl.out holds the sums of the current (averaged) iterate
  int last_cg_its = 6;
  double factor_resid = 0; // set-up guard of the direct back-end: ||K z - rhs|| / ||rhs|| of one known right-hand side
  int cg_enq = 0; // CG iterations enqueued so far for the solve in flight
  long tot_cg_its = 0, tot_solves = 0, tot_cg_skipped = 0; // (skipped: counted in tot_cg_its as the reference counts them, not executed -- look-ahead solves handed over)
  // solution staged on the host by finish_solution()
  std::vector<double> sol_x, sol_y, sol_s;
  ABIPInfo last_info;
  bool have_solution = false;
  // ---- multi-GPU (rows [row0, row0+m) of the global problem live on this rank) ------------------
  bool dist = false;
  int rank = 0, world = 1;
  abip_int m_glob = 0, row0 = 0;
  double xwt = 1.0;         // weight of replicated (x, tau) terms in reductions: 1 on rank 0, else 0
  DBuf<double> T;           // all-reduce buffer: [n-vector | S_COUNT scalars]
  double *gs = nullptr;     // = T.p + n_pad when dist, else null
  size_t n_pad = 0;
  std::vector<double> hstage; // host staging for the callback backend
  bool vy_zero = true;      // v[0:m) == 0 (cold start, no half_update): A'u_y == A'u_t,y, one all-reduce less per iteration
  // the PCG in its COLUMN form (ABIP_HIP_DIST_CG=cols): inside the solve the m-space is gathered and replicated, A is used by the column block
  // [col0, col0 + ncol); one exchange of m_glob doubles per PCG iteration instead of n (dev_kernels.h: k_cols_*)
  bool cg_cols = false;
  abip_int col0 = 0, ncol = 0;
  DevCsr dAc, dAct;         // the column block: CSR of A_g (m_glob rows, local column indices), CSR of A_g' (ncol rows)
  DBuf<double> cc_b, cc_y, cc_r, cc_z, cc_p, cc_Gp, cc_M, cc_tmp, cc_d, cc_buf; // m_glob: b_y, y (x0 in), r, z, p, Gp, M; ncol: A_g'p, b_x - A_g's; exchange area (2 m_glob)
  ABIPMatrix Aloc{};        // this rank's row block in CSC (owned)
  std::vector<double> Aloc_x; std::vector<abip_int> Aloc_i, Aloc_p;
  // ---- profiling -----------------------------------------------------------------------------
  unsigned prof_mask = 0;
  struct Ev { hipEvent_t a, b; int cls; int tag; };
  int ev_tag = -1; // CG iteration index of the launches being enqueued (-1: always effective)
  std::vector<Ev> ev_pool; size_t ev_used = 0;
  AbipHipProfile prof{};
  // device-side launch stamps (dev_common.h Stamp): a ring of records, harvested with the once-per-iteration control read
  static constexpr size_t ST_RING = 1u << 15;
  unsigned stamp_mask = 0;
  DBuf<Stamp> stamps; Stamp *hstamps = nullptr;
  std::vector<signed char> st_cls; // class of the launch that owns a slot
  size_t st_head = 0, st_tail = 0; // next free slot / first slot not yet harvested (monotonic; slot = counter % ST_RING)
  double tick_ms = 1e-5;           // one wall-clock tick in milliseconds
  std::vector<double> scratch; // host scratch (LV)
};

namespace {

typedef ABIP_WORK W;

// ------------------------------------------------------------------------------------------------
// launch helper: optional hipEvent bracket per kernel class
// ------------------------------------------------------------------------------------------------
template <class K, class... Args>
inline void launch_lds(W *w, int cls, K kern, int grid, int block, size_t lds_bytes, Args... args) {
  const bool timed = (w->prof_mask >> cls) & 1u;
  W::Ev *ev = nullptr;
  if (timed) {
    if (w->ev_used == w->ev_pool.size()) {
      W::Ev e; e.cls = cls;
      (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
      w->ev_pool.push_back(e);
    }
    ev = &w->ev_pool[w->ev_used++];
    ev->cls = cls; ev->tag = w->ev_tag;
    (void)hipEventRecord(ev->a, w->stream);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds_bytes, w->stream, args...);
  if (timed) (void)hipEventRecord(ev->b, w->stream);
}
template <class K, class... Args>
inline void launch(W *w, int cls, K kern, int grid, int block, Args... args) { launch_lds(w, cls, kern, grid, block, (size_t)0, args...); }

void harvest_events(W *w) { // call only after the stream has been synchronised
  for (size_t q = 0; q < w->ev_used; ++q) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, w->ev_pool[q].a, w->ev_pool[q].b) != hipSuccess) continue;
    const W::Ev &e = w->ev_pool[q];
    // a PCG launch tagged with iteration index t did work iff t < (iterations the device actually ran)
    if (e.cls == ABIP_HIP_K_CLASSES) { w->prof.allreduce_ms += ms; continue; }
    if (e.tag >= 0 && e.tag >= w->hctl->cg_it) { w->prof.noop_ms += ms; w->prof.noop_launches++; }
    else { w->prof.ms[e.cls] += ms; w->prof.launches[e.cls]++; }
  }
  w->ev_used = 0;
}

Stamp *next_stamp(W *w, int cls) { // a fresh record for the launch about to be enqueued, or null when this class is not being stamped
  if (!((w->stamp_mask >> cls) & 1u) || !w->stamps.p || w->st_head - w->st_tail >= W::ST_RING) return nullptr;
  const size_t slot = w->st_head++ % W::ST_RING;
  w->st_cls[slot] = (signed char)cls;
  return w->stamps.p + slot;
}
int enqueue_stamp_readback(W *w, size_t *lo, size_t *hi) { // D2H of the records written since the last harvest (the stream is synchronised by the caller)
  *lo = w->st_tail; *hi = w->st_head;
  for (size_t a = *lo; a < *hi;) {
    const size_t slot = a % W::ST_RING, len = std::min(*hi - a, W::ST_RING - slot);
    HIP_OK(hipMemcpyAsync(w->hstamps + slot, w->stamps.p + slot, len * sizeof(Stamp), hipMemcpyDeviceToHost, w->stream));
    HIP_OK(hipMemsetAsync(w->stamps.p + slot, 0, len * sizeof(Stamp), w->stream)); // ready for reuse
    a += len;
  }
  return 0;
}
void harvest_stamps(W *w, size_t lo, size_t hi) {
  // developer: ABIP_HIP_STAMP_DUMP=<file> appends every stamped launch in enqueue order -- class, begin tick, end tick (0 0: it returned at a gate) -- so that the
  // idle time BETWEEN two launches can be read off outside the profiler (scripts/stamp_gaps.py)
  static FILE *dump = getenv("ABIP_HIP_STAMP_DUMP") ? fopen(getenv("ABIP_HIP_STAMP_DUMP"), "a") : nullptr;
  for (size_t a = lo; a < hi; ++a) {
    const size_t slot = a % W::ST_RING;
    const Stamp &r = w->hstamps[slot];
    if (dump) fprintf(dump, "%d %llu %llu\n", (int)w->st_cls[slot], r.t1 ? (unsigned long long)~r.t0_inv : 0ull, (unsigned long long)r.t1);
    if (r.t1 == 0) { w->prof.stamp_noop_launches++; continue; } // returned at a gate (enqueued past PCG convergence)
    const unsigned long long t0 = ~r.t0_inv;
    if (r.t1 <= t0) continue;
    const int cls = w->st_cls[slot];
    w->prof.stamp_ms[cls] += (double)(r.t1 - t0) * w->tick_ms;
    w->prof.stamp_launches[cls]++;
  }
  w->st_tail = hi;
}

int sync_ctl(W *w) { // the once-per-iteration control read
  size_t lo = 0, hi = 0;
  if (w->stamp_mask && enqueue_stamp_readback(w, &lo, &hi)) return -1;
  HIP_OK(hipMemcpyAsync(w->hctl, w->ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  harvest_events(w);
  if (hi > lo) harvest_stamps(w, lo, hi);
  if (w->dist && g_dist.kind == 3 && *g_dist.peer.hstatus) { fprintf(stderr, "abip_hip: rank %d: a peer exchange gave up waiting for rank %d\n", g_dist.rank, *g_dist.peer.hstatus - 1); return -1; }
  return 0;
}

// the kernel instantiation for the layout a matrix carries (dev_common.h spmv_rows)
#define PICK(kern, M) ((M).nslices > 0 ? kern<true> : kern<false>)
#define PICK2(kern, T1, M) ((M).nslices > 0 ? kern<T1, true> : kern<T1, false>)

inline Dims dims(const W *w) { return Dims{(int)w->m, (int)w->n, w->MP}; }

// ------------------------------------------------------------------------------------------------
// multi-GPU: in-place sum over the ranks of `count` doubles at device pointer `buf`, ordered on the solver's stream
// ------------------------------------------------------------------------------------------------
int allreduce_dev(W *w, double *buf, size_t count) {
  if (!w->dist) return 0;
  w->prof.allreduce_calls++; w->prof.allreduce_bytes += (double)(sizeof(double) * count);
  W::Ev *ev = nullptr;
  if (((w->prof_mask >> ABIP_HIP_K_CLASSES) & 1u) && g_dist.kind == 1) { // RCCL: bracket the collective on the solver's stream (the host-staged test transport synchronises by itself)
    if (w->ev_used == w->ev_pool.size()) { W::Ev e; e.cls = ABIP_HIP_K_CLASSES; (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); w->ev_pool.push_back(e); }
    ev = &w->ev_pool[w->ev_used++];
    ev->cls = ABIP_HIP_K_CLASSES; ev->tag = -1; // (class "one past the kernels": the collective)
    (void)hipEventRecord(ev->a, w->stream);
  }
  const int rc = abip::dist_allreduce(buf, count, w->stream, w->hstage);
  if (ev) (void)hipEventRecord(ev->b, w->stream);
  return rc;
}
// The peer-mapped transport (dev_peer.h) lets a producer kernel push its result straight into the reducing ranks' mailboxes: peer_push_begin hands the kernel
// what it needs (and takes the epoch of this exchange); peer_finish enqueues steps 2 + 3 behind it under the producer's gate `mode`.  pp.on == 0: any other
// transport -- the producer stores locally and the caller runs the collective as before.
// INVARIANT of a fused exchange: the producer pushes the rows [0, n) (columns form: [0, m_glob)) and the scalar slots its fold list names, nothing else.
// The padding T[n, n_pad) and the gs slots outside the list keep whatever earlier exchanges with other counts left in the inboxes: after the reduce they are
// UNDEFINED (with RCCL or the stand-alone all-reduce they are sums of zeros).  Nothing reads them: k_dist_cg_step / k_dist_post / k_dist_q stop at n and take
// only the slots folded for them; a new consumer of T beyond n, or of an unfolded slot, must not run behind a fused exchange.
PeerPush peer_push_begin(W *w, size_t count) {
  PeerPush pp{};
  pp.on = 0;
  if (!w->dist || g_dist.kind != 3 || *g_dist.peer.hstatus || (long)count > g_dist.peer.cap || !g_dist.peer.fused) return pp;
  pp.on = 1; pp.c = g_dist.peer.ctx; pp.chunk = abip::peer_chunk((long)count, g_dist.world); pp.epoch = ++g_dist.peer.epoch;
  return pp;
}
int peer_finish(W *w, const PeerPush &pp, double *buf, size_t count, int mode) {
  w->prof.allreduce_calls++; w->prof.allreduce_bytes += (double)(sizeof(double) * count);
  hipLaunchKernelGGL(abip::k_peer_reduce_gather_gated, dim3(64), dim3(256), 0, w->stream, pp.c, buf, (long)count, pp.epoch, mode, (const Ctl *)w->ctl.p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
void enqueue_fold(W *w, std::initializer_list<int> slots) {
  FoldArgs f; f.nslots = 0;
  for (int sl : slots) f.slots[f.nslots++] = sl;
  launch(w, ABIP_HIP_K_VEC, k_fold, 1, BS, f, (const double *)w->part.p, w->NB, w->gs);
}
inline int allreduce_scalars(W *w) { return allreduce_dev(w, w->gs, (size_t)S_COUNT); }
inline int allreduce_vec_and_scalars(W *w) { return allreduce_dev(w, w->T.p, w->n_pad + S_COUNT); }

// ------------------------------------------------------------------------------------------------
// KKT solve on an l-vector already holding the rhs.  S_BN must hold ||rhs_y||^2 (indirect).
// enqueue-only pieces + a synchronising driver
// ------------------------------------------------------------------------------------------------
double cg_tol_factor(const W *w, abip_int iter) { // indirect.c:406-407
  return iter < 0 ? 1e-9 : 1e-1 / std::pow((double)iter + 1, w->stgs->cg_rate);
}

// PCG pieces.  Single GPU: everything stays on the device (dev_kernels.h).  Multi-GPU (w->dist): the rank's A_g' x_g is
// a partial n-vector in T[0:n); it is all-reduced together with the packed scalars in T[n:] (one collective), and the
// second half of each step runs on the summed data.  Per CG iteration: ONE collective (vector + packed scalars): p'Gp is
// rebuilt from rho ||p||^2 (recurrence on the summed z'z, z'p) + ||A'p||^2 (replicated), not reduced on its own.
int enqueue_cg_begin(W *w, double *rhs, const double *warm, abip_int iter, bool pairs_ready = false) {
  const Dims d = dims(w);
  w->cg_enq = 0;
  const Ctl *ctl = w->ctl.p;
  const double *bx = rhs + w->MP;
  double2 *pair = (double2 *)w->cg_pair.p;
  if (!w->dist) {
    if (warm && !pairs_ready) launch(w, ABIP_HIP_K_CG_EDGE, PICK(k_cg_init_At, w->dAt), w->NB, BS, w->dAt.view(), warm, bx, pair, ctl); // (pairs_ready: k_rhs wrote them from the kept A'u_y)
    launch(w, ABIP_HIP_K_CG_EDGE, PICK2(k_cg_init_A, false, w->dA), w->NB, BS, w->dA.view(), rhs, (const double2 *)pair, warm, (const double *)w->cg_M.p,
           w->cg_r.p, w->cg_z.p, w->cg_p.p, w->stgs->rho_y, cg_tol_factor(w, iter), d, w->part.p, w->NB, w->ctl.p, (const double *)nullptr);
    return 0;
  }
  if (w->cg_cols) { // gather b_y (and the warm start) over the row blocks, then the set-up products by column blocks
    const int mg = (int)w->m_glob, gm = std::max(1, std::min(w->NB, (mg + BS - 1) / BS)), gn = std::max(1, std::min(w->NB, (int)((w->ncol + BS - 1) / BS)));
    enqueue_fold(w, {S_BN});
    if (hipMemsetAsync(w->cc_buf.p, 0, sizeof(double) * 2 * (size_t)mg, w->stream) != hipSuccess) return -1;
    launch(w, ABIP_HIP_K_VEC, k_cols_place, gm, BS, (const double *)rhs, (int)w->m, (int)w->row0, w->cc_buf.p, 0, ctl);
    if (warm) launch(w, ABIP_HIP_K_VEC, k_cols_place, gm, BS, warm, (int)w->m, (int)w->row0, w->cc_buf.p + mg, 0, ctl);
    if (allreduce_dev(w, w->cc_buf.p, (warm ? 2 : 1) * (size_t)mg) || allreduce_scalars(w)) return -1;
    if (hipMemcpyAsync(w->cc_b.p, w->cc_buf.p, sizeof(double) * mg, hipMemcpyDeviceToDevice, w->stream) != hipSuccess) return -1;
    if (warm && hipMemcpyAsync(w->cc_y.p, w->cc_buf.p + mg, sizeof(double) * mg, hipMemcpyDeviceToDevice, w->stream) != hipSuccess) return -1;
    const double *bxl = bx + w->col0; // the rank's columns of the (replicated) x block
    if (warm) launch(w, ABIP_HIP_K_CG_EDGE, k_spmv_set<false>, w->NB, BS, w->dAct.view(), (const double *)w->cc_y.p, w->cc_tmp.p, 0, ctl);
    launch(w, ABIP_HIP_K_VEC, k_cols_diff, gn, BS, bxl, warm ? (const double *)w->cc_tmp.p : (const double *)nullptr, w->cc_d.p, (int)w->ncol, ctl);
    launch(w, ABIP_HIP_K_CG_EDGE, k_spmv_set<false>, w->NB, BS, w->dAc.view(), (const double *)w->cc_d.p, w->cc_buf.p, 0, ctl);
    if (allreduce_dev(w, w->cc_buf.p, (size_t)mg)) return -1;
    launch(w, ABIP_HIP_K_CG_VEC, k_cols_init_fin, w->NB, BS, (const double *)w->cc_b.p, (const double *)w->cc_buf.p, warm ? (const double *)w->cc_y.p : (const double *)nullptr,
           (const double *)w->cc_M.p, w->cc_y.p, w->cc_r.p, w->cc_z.p, w->cc_p.p, w->stgs->rho_y, cg_tol_factor(w, iter), mg, w->part.p, w->ctl.p, (const double *)w->gs);
    return 0;
  }
  enqueue_fold(w, {S_BN});
  if (warm) {
    launch(w, ABIP_HIP_K_CG_EDGE, PICK(k_spmv_set, w->dAt), w->NB, BS, w->dAt.view(), warm, w->T.p, 0, ctl);
    if (allreduce_vec_and_scalars(w)) return -1;
    launch(w, ABIP_HIP_K_VEC, k_cg_init_pair, w->NB, BS, (const double *)w->T.p, bx, pair, (int)w->n, ctl);
  } else if (allreduce_scalars(w)) return -1;
  launch(w, ABIP_HIP_K_CG_EDGE, PICK2(k_cg_init_A, true, w->dA), w->NB, BS, w->dA.view(), rhs, (const double2 *)pair, warm, (const double *)w->cg_M.p,
         w->cg_r.p, w->cg_z.p, w->cg_p.p, w->stgs->rho_y, cg_tol_factor(w, iter), d, w->part.p, w->NB, w->ctl.p, (const double *)w->gs);
  return 0;
}
int enqueue_cg_chunk(W *w, double *rhs, int its) {
  const int max_its = (int)w->m_glob; // indirect.c:418: at most m iterations
  const int gvec = std::max(1, std::min(w->NB, (int)((w->m + 2 * BS - 1) / (2 * BS))));
  if (w->cg_cols) {
    const int mg = (int)w->m_glob, gv = std::max(1, std::min(w->NB, (mg + 2 * BS - 1) / (2 * BS)));
    for (int q = 0; q < its; ++q) {
      w->ev_tag = w->cg_enq++;
      launch(w, ABIP_HIP_K_SPMV_AT, k_cg_spmv_At<false>, w->NB, BS, w->dAct.view(), (const double *)w->cc_z.p, w->cc_tmp.p, max_its, w->part.p, w->NB, w->ctl.p, next_stamp(w, ABIP_HIP_K_SPMV_AT));
      const PeerPush pp = peer_push_begin(w, (size_t)mg);
      if (pp.on) {
        launch(w, ABIP_HIP_K_SPMV_A, k_spmv_set_t<false, false>, w->NB, BS, w->dAc.view(), (const double *)w->cc_tmp.p, w->cc_buf.p, 1, (const Ctl *)w->ctl.p,
               FoldArgs{}, (const double *)w->part.p, w->NB, w->gs, (Stamp *)nullptr, pp, 0L);
        if (peer_finish(w, pp, w->cc_buf.p, (size_t)mg, 1)) return -1;
      } else {
        launch(w, ABIP_HIP_K_SPMV_A, k_spmv_set<false>, w->NB, BS, w->dAc.view(), (const double *)w->cc_tmp.p, w->cc_buf.p, 1, (const Ctl *)w->ctl.p);
        if (allreduce_dev(w, w->cc_buf.p, (size_t)mg)) return -1;
      }
      launch(w, ABIP_HIP_K_CG_VEC, k_cols_Gp_fin, w->NB, BS, (const double *)w->cc_buf.p, (const double *)w->cc_z.p, w->cc_p.p, w->cc_Gp.p, w->stgs->rho_y, mg, w->part.p, (const Ctl *)w->ctl.p);
      launch(w, ABIP_HIP_K_CG_VEC, k_cg_update<false>, gv, BS, w->cc_y.p, w->cc_r.p, w->cc_z.p, (const double *)w->cc_p.p, (const double *)w->cc_Gp.p,
             (const double *)w->cc_M.p, mg, w->stgs->rho_y, w->part.p, w->NB, w->ctl.p, (const double *)nullptr);
    }
    w->ev_tag = -1;
    return 0;
  }
  for (int q = 0; q < its; ++q) {
    w->ev_tag = w->cg_enq++;
    if (!w->dist) {
      launch(w, ABIP_HIP_K_SPMV_AT, PICK(k_cg_spmv_At, w->dAt), w->NB, BS, w->dAt.view(), (const double *)w->cg_z.p, w->cg_tmp.p,
             max_its, w->part.p, w->NB, w->ctl.p, next_stamp(w, ABIP_HIP_K_SPMV_AT));
    } else {
      FoldArgs fo; fo.nslots = 0;
      for (int sl : {S_RR0, S_RR1, S_ZR0, S_ZR1, S_ZZ, S_ZP}) fo.slots[fo.nslots++] = sl;
      const PeerPush pp = peer_push_begin(w, w->n_pad + S_COUNT);
      launch(w, ABIP_HIP_K_SPMV_AT, PICK2(k_spmv_set_t, true, w->dAt), w->NB, BS, w->dAt.view(), (const double *)w->cg_z.p, w->T.p, 1, (const Ctl *)w->ctl.p,
             fo, (const double *)w->part.p, w->NB, w->gs, next_stamp(w, ABIP_HIP_K_SPMV_AT), pp, (long)w->n_pad);
      if (pp.on ? peer_finish(w, pp, w->T.p, w->n_pad + S_COUNT, 1) : allreduce_vec_and_scalars(w)) return -1;
      launch(w, ABIP_HIP_K_CG_VEC, k_dist_cg_step, w->NB, BS, (const double *)w->T.p, w->cg_tmp.p, (int)w->n, max_its, 1, (const double *)w->gs, w->part.p, w->ctl.p);
    }
    launch(w, ABIP_HIP_K_SPMV_A, PICK(k_cg_spmv_A, w->dA), w->NB, BS, w->dA.view(), (const double *)w->cg_tmp.p, (const double *)w->cg_z.p, w->cg_p.p,
           w->cg_Gp.p, w->stgs->rho_y, w->part.p, (const Ctl *)w->ctl.p, next_stamp(w, ABIP_HIP_K_SPMV_A));
    // sharded: p'Gp = rho ||p||^2 + ||A'p||^2 needs no collective of its own (k_dist_cg_step left both pieces behind)
    if (w->dist)
      launch(w, ABIP_HIP_K_CG_VEC, k_cg_update<true>, gvec, BS, rhs, w->cg_r.p, w->cg_z.p, (const double *)w->cg_p.p, (const double *)w->cg_Gp.p,
             (const double *)w->cg_M.p, (int)w->m, w->stgs->rho_y, w->part.p, w->NB, w->ctl.p, (const double *)w->gs);
    else
      launch(w, ABIP_HIP_K_CG_VEC, k_cg_update<false>, gvec, BS, rhs, w->cg_r.p, w->cg_z.p, (const double *)w->cg_p.p, (const double *)w->cg_Gp.p,
             (const double *)w->cg_M.p, (int)w->m, w->stgs->rho_y, w->part.p, w->NB, w->ctl.p, (const double *)w->gs);
  }
  w->ev_tag = -1;
  return 0;
}
int enqueue_cg_post(W *w, double *rhs, double *keep_aty = nullptr) {
  if (!w->dist) {
    launch(w, ABIP_HIP_K_CG_EDGE, PICK(k_post_At, w->dAt), w->NB, BS, w->dAt.view(), rhs, (const double *)w->h.p, dims(w), (int)w->m_glob, w->part.p, w->NB, w->ctl.p, keep_aty);
    return 0;
  }
  if (w->cg_cols) { // decision on the replicated |r|^2; then y back to the row block, A_g'y into its place of T, and the row form's tail
    const int mg = (int)w->m_glob, gm = std::max(1, std::min(w->NB, (mg + BS - 1) / BS)), gn = std::max(1, std::min(w->NB, (int)((w->ncol + BS - 1) / BS)));
    launch(w, ABIP_HIP_K_CG_VEC, k_cols_decide, 1, BS, mg, w->part.p, w->NB, w->ctl.p);
    launch(w, ABIP_HIP_K_VEC, k_cols_take, gm, BS, (const double *)w->cc_y.p, (int)w->m, (int)w->row0, rhs, (const Ctl *)w->ctl.p);
    launch(w, ABIP_HIP_K_CG_EDGE, k_spmv_set<false>, w->NB, BS, w->dAct.view(), (const double *)w->cc_y.p, w->cc_d.p, 2, (const Ctl *)w->ctl.p);
    if (hipMemsetAsync(w->T.p, 0, sizeof(double) * w->n_pad, w->stream) != hipSuccess) return -1;
    launch(w, ABIP_HIP_K_VEC, k_cols_place, gn, BS, (const double *)w->cc_d.p, (int)w->ncol, (int)w->col0, w->T.p, 2, (const Ctl *)w->ctl.p);
    if (allreduce_vec_and_scalars(w)) return -1;
    launch(w, ABIP_HIP_K_CG_EDGE, k_dist_post, w->NB, BS, (const double *)w->T.p, rhs, (const double *)w->h.p, dims(w), w->xwt, w->part.p, (const Ctl *)w->ctl.p);
    enqueue_fold(w, {S_DH});
    if (allreduce_scalars(w)) return -1;
    return 0;
  }
  // late convergence decision on the summed ||r||^2, then (only if converged) the back-substitution A'y
  enqueue_fold(w, {S_RR0, S_RR1, S_ZR0, S_ZR1});
  if (allreduce_scalars(w)) return -1;
  launch(w, ABIP_HIP_K_CG_VEC, k_dist_cg_step, 1, BS, (const double *)w->T.p, w->cg_tmp.p, (int)w->n, (int)w->m_glob, 0, (const double *)w->gs, w->part.p, w->ctl.p);
  launch(w, ABIP_HIP_K_CG_EDGE, PICK(k_spmv_set, w->dAt), w->NB, BS, w->dAt.view(), (const double *)rhs, w->T.p, 2, (const Ctl *)w->ctl.p);
  if (allreduce_vec_and_scalars(w)) return -1;
  launch(w, ABIP_HIP_K_CG_EDGE, k_dist_post, w->NB, BS, (const double *)w->T.p, rhs, (const double *)w->h.p, dims(w), w->xwt, w->part.p, (const Ctl *)w->ctl.p);
  enqueue_fold(w, {S_DH});
  if (allreduce_scalars(w)) return -1;
  return 0;
}
// project_lin_sys around a small direct solve (abip.c:552-560), folded into the one-workgroup solve kernels: pre = k_rhs,
// post = k_post_dot.  Same arithmetic per element; the two reductions run over one 1024-thread workgroup.
struct LpSolveFuse {
  static constexpr bool active = true;
  const double *u, *v, *h;
  double rho, g_th;
  Dims d;
  double *part;
  int nb;
  Ctl *ctl;
  int has_fin;   // 1: first close the PREVIOUS iteration (its k_finalize, exit test included) -- one launch less per iteration
  FinArgs fin;
  __device__ bool pre(double *ut, int tid) const {
    if (has_fin) {
      d_finalize(fin, d, part, nb, ctl);
      __syncthreads();
      if (__hip_atomic_load(&ctl->halt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false; // the exit test held: this iteration does not run
    }
    __shared__ double red[TBS / 64];
    double s = 0.0;
    for (int i = tid; i < nb; i += TBS) s += part[S_WG * MAXNB + i];
    const double wg = tbs_sum(s, red);
    const int tail = d.MP + d.n;
    const double tsum = u[tail] + v[tail];
    const double coef = (wg - tsum * g_th) / (g_th + 1.0);
    for (int i = tid; i < d.m; i += TBS) {
      double t = (u[i] + v[i]) * rho;
      t += -tsum * h[i];
      t += -coef * h[i];
      ut[i] = t;
    }
    for (int j = tid; j < d.n; j += TBS) {
      double t = u[d.MP + j] + v[d.MP + j];
      t += -tsum * h[d.MP + j];
      t += -coef * h[d.MP + j];
      ut[d.MP + j] = -t;
    }
    if (tid == 0) ut[tail] = tsum;
    return true;
  }
  __device__ void post(const double *rhs, int tid) const {
    __shared__ double red[TBS / 64];
    double s = 0.0;
    for (int i = tid; i < d.m; i += TBS) s += rhs[i] * h[i];
    for (int j = tid; j < d.n; j += TBS) s += rhs[d.MP + j] * h[d.MP + j];
    const double dh = tbs_sum(s, red);
    for (int e = tid; e < nb; e += TBS) part[S_DH * MAXNB + e] = (e == 0) ? dh : 0.0;
  }
};

void enqueue_direct(W *w, double *rhs) {
  const Ctl *ctl = w->ctl.p;
  w->ldl.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { launch_lds(w, ABIP_HIP_K_SPTRSV, kern, grid, block, lds, a...); }, rhs, ctl, w->NB);
  launch(w, ABIP_HIP_K_VEC, k_post_dot, w->NB, BS, (const double *)rhs, (const double *)w->h.p, dims(w), w->part.p, ctl);
}

// CG iterations to enqueue blind.  Single GPU: generous (a launch past convergence costs ~3.6 us).  Multi-GPU: tight --
// a collective past convergence cannot be gated on the device, so over-enqueueing costs real all-reduces.
int next_chunk(const W *w) {
  if (w->dist) return std::max(1, std::min((int)w->m_glob, w->last_cg_its - 1));
  return std::max(2, std::min((int)w->m_glob, w->last_cg_its + std::max(2, w->last_cg_its >> 3)));
}

// Solve K z = rhs in place and leave S_DH = z[0:l-1)'h; synchronises with the host (used outside the hot loop:
// the set-up solve for g and the BB look-ahead).  Returns CG iterations, <0 on error.
int xcd_solve(W *w, double *rhs, const double *warm, abip_int iter);
int kkt_solve_sync(W *w, double *rhs, const double *warm, abip_int iter) {
  w->tot_solves++;
  w->prof.kkt_solves++;
  if (w->xcd.on && w->xcd_solves) { // cache-resident LP: the solve as one persistent launch too (the BB look-ahead is dozens of solves per outer iteration)
    const int its = xcd_solve(w, rhs, warm, iter);
    if (its == -1) return -1;
    if (its >= 0) {
      if (w->linsys == ABIP_HIP_LINSYS_INDIRECT) { w->last_cg_its = its; if (iter >= 0) { w->tot_cg_its += its; w->prof.cg_iters += its; } }
      return its;
    } // (-2: the launch was abandoned and rhs restored -- solve it below)
  }
  if (w->linsys == ABIP_HIP_LINSYS_DIRECT) {
    enqueue_direct(w, rhs);
    if (sync_ctl(w)) return -1;
    return 0;
  }
  if (enqueue_cg_begin(w, rhs, warm, iter)) return -1;
  int chunk = next_chunk(w);
  for (;;) {
    if (enqueue_cg_chunk(w, rhs, chunk)) return -1;
    if (enqueue_cg_post(w, rhs)) return -1;
    if (sync_ctl(w)) return -1;
    if (w->hctl->cg_done) break;
    chunk = w->dist ? 2 : std::max(4, chunk);
  }
  const int its = w->hctl->cg_it;
  w->last_cg_its = its;
  if (iter >= 0) { w->tot_cg_its += its; w->prof.cg_iters += its; } // indirect.c:422-425
  return its;
}

// ------------------------------------------------------------------------------------------------
// residuals from the finalised sums (calc_residuals, abip.c:458-535)
// ------------------------------------------------------------------------------------------------
void calc_residuals(W *w, abip_int ipm_iter, abip_int admm_iter) {
  Resid &r = w->r;
  if (admm_iter && r.last_admm_iter == admm_iter) return;
  r.last_ipm_iter = ipm_iter; r.last_admm_iter = admm_iter;
  const double den = w->stgs->normalize ? (w->stgs->scale * w->sc_c * w->sc_b) : 1.0;
  lp_residuals(x_sums(w->hctl->out, w->stgs->avg_criterion != 0), den, w->nm_b, w->nm_c, r); // (one source with the persistent launch: lp_scalars.h)
}

abip_int has_converged(const W *w, abip_int ipm_iter, abip_int admm_iter) { // abip.c:1613-1641
  return (abip_int)lp_converged(w->r, w->stgs->eps, (int)w->stgs->pfeasopt, (long)ipm_iter, (long)admm_iter); // ABIP_SOLVED 1, ABIP_UNBOUNDED -1, ABIP_INFEASIBLE -2
}

// ------------------------------------------------------------------------------------------------
// statistics pass on the current iterate(s): the two residual SpMVs + finalise + control read
// ------------------------------------------------------------------------------------------------
int enqueue_q_and_finalize(W *w, bool avg_stats, bool T_holds_Aty, bool decide = true, FinArgs *defer = nullptr, const double *aty = nullptr, const FinStream *fs = nullptr) {
  const Dims d = dims(w);
  const Ctl *ctl = w->ctl.p;
  const double *wD = w->stgs->normalize ? w->wD.p : nullptr, *wE = w->stgs->normalize ? w->wE.p : nullptr;
  FinArgs f;
  int ns = 0;
  const int base[] = {S_NU, S_NV, S_CX, S_BY, S_QP, S_RP, S_NAX, S_QD, S_RD, S_NATY};
  const int extra[] = {S_NUA, S_NVA, S_CXA, S_BYA, S_QPA, S_RPA, S_NAXA, S_QDA, S_RDA, S_NATYA};
  for (int s : base) f.slots[ns++] = s;
  if (!w->dist && aty) { // A'u_y is at hand (k_post_At kept it): only A u_x is a product, the dual residuals are element-wise
    launch(w, ABIP_HIP_K_QNORM, PICK(k_q_A_aty, w->dA), 2 * w->NB, BS, w->dA.view(), aty, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->b.p,
           (const double *)w->c.p, wD, wE, d, (int)S_QP, (int)S_QD, w->NB, w->part.p, ctl);
  } else if (!w->dist) {
    launch(w, ABIP_HIP_K_QNORM, (w->dA.nslices > 0 ? (w->dAt.nslices > 0 ? k_q_both<true, true> : k_q_both<true, false>) : (w->dAt.nslices > 0 ? k_q_both<false, true> : k_q_both<false, false>)), 2 * w->NB, BS, w->dA.view(), w->dAt.view(), (const double *)w->u.p, (const double *)w->v.p, (const double *)w->b.p,
           (const double *)w->c.p, wD, wE, d, (int)S_QP, (int)S_QD, w->NB, w->part.p, ctl);
  } else {
    launch(w, ABIP_HIP_K_QNORM, PICK(k_q_A, w->dA), w->NB, BS, w->dA.view(), (const double *)w->u.p, (const double *)w->b.p, wD, d, (int)S_QP, w->part.p, ctl);
    if (!T_holds_Aty) { // A'u_y differs from the A'u_t,y the back-substitution left in T (v_y != 0): one more partial + all-reduce
      launch(w, ABIP_HIP_K_QNORM, PICK(k_spmv_set, w->dAt), w->NB, BS, w->dAt.view(), (const double *)w->u.p, w->T.p, 2, ctl);
      if (allreduce_vec_and_scalars(w)) return -1;
    }
    launch(w, ABIP_HIP_K_QNORM, k_dist_q, w->NB, BS, (const double *)w->T.p, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->c.p, wE, d,
           (int)S_QD, w->xwt, w->part.p, ctl);
  }
  if (avg_stats) {
    if (!w->dist) {
      launch(w, ABIP_HIP_K_QNORM, (w->dA.nslices > 0 ? (w->dAt.nslices > 0 ? k_q_both<true, true> : k_q_both<true, false>) : (w->dAt.nslices > 0 ? k_q_both<false, true> : k_q_both<false, false>)), 2 * w->NB, BS, w->dA.view(), w->dAt.view(), (const double *)w->u_avgc.p, (const double *)w->v_avgc.p, (const double *)w->b.p,
             (const double *)w->c.p, wD, wE, d, (int)S_QPA, (int)S_QDA, w->NB, w->part.p, ctl);
    } else {
      launch(w, ABIP_HIP_K_QNORM, PICK(k_q_A, w->dA), w->NB, BS, w->dA.view(), (const double *)w->u_avgc.p, (const double *)w->b.p, wD, d, (int)S_QPA, w->part.p, ctl);
      launch(w, ABIP_HIP_K_QNORM, PICK(k_spmv_set, w->dAt), w->NB, BS, w->dAt.view(), (const double *)w->u_avgc.p, w->T.p, 2, ctl);
      if (allreduce_vec_and_scalars(w)) return -1;
      launch(w, ABIP_HIP_K_QNORM, k_dist_q, w->NB, BS, (const double *)w->T.p, (const double *)w->u_avgc.p, (const double *)w->v_avgc.p, (const double *)w->c.p, wE, d,
             (int)S_QDA, w->xwt, w->part.p, ctl);
    }
    for (int s : extra) f.slots[ns++] = s;
  }
  f.nslots = ns;
  f.u = w->u.p; f.v = w->v.p; f.ua = w->u_avgc.p; f.va = w->v_avgc.p; f.gs = w->gs;
  f.decide = decide ? 1 : 0; f.avg_stats = avg_stats ? 1 : 0; f.thr = w->gamma * w->mu; f.sentinel = (double)w->stgs->max_admm_iters;
  if (w->dist) {
    FoldArgs fo; fo.nslots = ns;
    for (int q = 0; q < ns; ++q) fo.slots[q] = f.slots[q];
    fo.slots[fo.nslots++] = S_WG; // the next k_rhs needs it summed over the ranks as well
    launch(w, ABIP_HIP_K_VEC, k_fold, 1, BS, fo, (const double *)w->part.p, w->NB, w->gs);
    if (allreduce_scalars(w)) return -1;
  }
  if (defer) { *defer = f; return 0; } // the next iteration's solve kernel runs it as its prologue
  if (fs) launch(w, ABIP_HIP_K_VEC, k_finalize_stream, 1, 1024, f, *fs, d, (const double *)w->part.p, w->NB, w->ctl.p);
  else launch(w, ABIP_HIP_K_VEC, k_finalize, 1, 1024, f, d, (const double *)w->part.p, w->NB, w->ctl.p);
  return 0;
}

UpdArgs upd_args(W *w, bool fuse_avg, bool avg_stats, abip_int j) {
  UpdArgs a;
  a.u = w->u.p; a.v = w->v.p; a.ut = w->ut.p;
  a.u_avg = w->u_avg.p; a.v_avg = w->v_avg.p; a.u_sum = w->u_sum.p; a.v_sum = w->v_sum.p; a.u_avgc = w->u_avgc.p; a.v_avgc = w->v_avgc.p;
  a.g = w->g.p; a.b = w->b.p; a.c = w->c.p;
  a.alpha = w->stgs->alpha; a.mu_over_beta = w->mu / w->beta; a.rho = w->stgs->rho_y; a.dom = (double)(j + 1);
  a.xw = w->xwt; a.gs = w->gs;
  a.half_update = (int)w->stgs->half_update; a.fuse_avg = fuse_avg ? 1 : 0; a.avg_stats = avg_stats ? 1 : 0;
  return a;
}

// ------------------------------------------------------------------------------------------------
// one inner ADMM iteration (abip.c:2133-2173 up to and including the stopping metric); returns the metric
// ------------------------------------------------------------------------------------------------
// everything of ADMM iteration (k, j) up to and including the finalize that evaluates the exit test (direct back-end)
// prev_fin: the finalize of the previous iteration, still pending (fused path); defer: hand this iteration's finalize to the next one
int enqueue_iteration_direct(W *w, abip_int j, bool restart, const FinArgs *prev_fin = nullptr, FinArgs *defer = nullptr) {
  const Dims d = dims(w);
  const Ctl *ctl = w->ctl.p;
  ABIPSettings *st = w->stgs;
  const bool avg_stats = ((j + 1) % 10 == 0); // abip.c:2000
  if (w->fuse_small) { // k_rhs and k_post_dot ride inside the one-workgroup solve kernels
    LpSolveFuse fz{w->u.p, w->v.p, w->h.p, st->rho_y, w->g_th, d, w->part.p, w->NB, w->ctl.p, prev_fin ? 1 : 0, prev_fin ? *prev_fin : FinArgs{}};
    w->ldl.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { launch_lds(w, ABIP_HIP_K_SPTRSV, kern, grid, block, lds, a...); }, w->ut.p, ctl, w->NB, fz);
  } else {
    launch(w, ABIP_HIP_K_VEC, k_rhs, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, w->ut.p, (const double *)w->h.p, st->rho_y, w->g_th, d,
           w->part.p, w->NB, ctl, (const double *)w->gs, (const double *)nullptr, (double2 *)nullptr);
    enqueue_direct(w, w->ut.p);
  }
  launch(w, ABIP_HIP_K_VEC, k_admm_update, w->NB, BS, upd_args(w, !restart, avg_stats, j), d, w->part.p, w->NB, ctl);
  if (restart) {
    launch(w, ABIP_HIP_K_VEC, k_restart_apply, w->NB, BS, w->u.p, w->v.p, w->u_avg.p, w->v_avg.p, (double)st->restart_fre, w->LV);
    launch(w, ABIP_HIP_K_VEC, k_avg_stats, w->NB, BS, upd_args(w, true, avg_stats, j), d, w->part.p, ctl);
  }
  return enqueue_q_and_finalize(w, avg_stats, false, true, defer);
}
inline bool restart_due(const W *w, abip_int k, abip_int j) { // abip.c:608-609
  return !(k < w->stgs->restart_thresh || (j + 1 - w->fre_old) % w->stgs->restart_fre != 0);
}
// The exit test has been evaluated on the device; take its verdict (one source for the batched and the stepwise path).
inline void take_verdict(W *w, double *metric_out) {
  w->stgs->avg_criterion = w->hctl->avg_crit; // abip.c:2042,2048
  *metric_out = w->hctl->metric;
  w->it_seen = w->hctl->it_count;
}
int clear_halt(W *w) {
  if (!w->hctl->halt) return 0;
  HIP_OK(hipMemsetAsync(&w->ctl.p->halt, 0, sizeof(int), w->stream));
  w->hctl->halt = 0;
  return 0;
}

// Direct back-end on one GPU: enqueue up to nb iterations back to back with no host round trip in between.  k_finalize raises
// ctl->halt at the iteration whose exit test holds and everything enqueued behind it falls through, so the trajectory is the
// one the stepwise loop produces.  *done = iterations that ran.
int admm_batch_direct(W *w, int nb, int *done, double *metric_out) {
  if (!w->wg_valid) launch(w, ABIP_HIP_K_VEC, k_dot_wg, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->g.p, w->stgs->rho_y, dims(w), w->part.p, w->xwt);
  FinArgs pend, cur;
  bool have = false;
  for (int q = 0; q < nb; ++q) {
    const bool defer = w->fuse_small && q + 1 < nb; // the last iteration of the batch closes itself
    if (enqueue_iteration_direct(w, w->j + q, false, have ? &pend : nullptr, defer ? &cur : nullptr)) return -1;
    have = defer; pend = cur;
  }
  if (sync_ctl(w)) return -1;
  *done = w->hctl->it_count - w->it_seen;
  if (*done < 1 || *done > nb) return -1;
  take_verdict(w, metric_out);
  w->tot_solves += *done; w->prof.kkt_solves += *done; w->prof.admm_iters += *done;
  w->wg_valid = true;
  w->stats_valid = true; w->avg_stats_valid = ((w->j + *done) % 10 == 0);
  return 0;
}

int admm_iteration(W *w, double *metric_out) {
  const Dims d = dims(w);
  const Ctl *ctl = w->ctl.p;
  ABIPSettings *st = w->stgs;
  if (!w->wg_valid) {
    launch(w, ABIP_HIP_K_VEC, k_dot_wg, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->g.p, st->rho_y, d, w->part.p, w->xwt);
    if (w->dist) { enqueue_fold(w, {S_WG}); if (allreduce_scalars(w)) return -1; }
  }
  const bool avg_stats = ((w->j + 1) % 10 == 0);  // abip.c:2000
  const bool restart = restart_due(w, w->k, w->j);
  w->tot_solves++;
  w->prof.kkt_solves++;
  if (w->linsys == ABIP_HIP_LINSYS_DIRECT) {
    if (enqueue_iteration_direct(w, w->j, restart) || sync_ctl(w)) return -1;
  } else {
    // one GPU, v_y == 0, a plain iteration: the back-substitution's A'u_t,y is kept and serves the stopping test and the next solve's set-up (W::aty)
    const bool keep = w->aty_on && !w->dist && w->vy_zero && !restart && !st->half_update && w->aty.p;
    const bool have = keep && w->aty_valid;
    launch(w, ABIP_HIP_K_VEC, k_rhs, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, w->ut.p, (const double *)w->h.p, st->rho_y, w->g_th, d,
           w->part.p, w->NB, ctl, (const double *)w->gs, have ? (const double *)w->aty.p : (const double *)nullptr, have ? (double2 *)w->cg_pair.p : (double2 *)nullptr);
    int err = 0;
    auto tail = [&]() {
      launch(w, ABIP_HIP_K_VEC, k_admm_update, w->NB, BS, upd_args(w, !restart, avg_stats, w->j), d, w->part.p, w->NB, ctl);
      if (restart) {
        launch(w, ABIP_HIP_K_VEC, k_restart_apply, w->NB, BS, w->u.p, w->v.p, w->u_avg.p, w->v_avg.p, (double)st->restart_fre, w->LV);
        launch(w, ABIP_HIP_K_VEC, k_avg_stats, w->NB, BS, upd_args(w, true, avg_stats, w->j), d, w->part.p, ctl);
      }
      // T still holds A'u_t,y from the back-substitution; it equals A'u_y iff v_y == 0 and (u, v) were not replaced by the restart mean
      if (enqueue_q_and_finalize(w, avg_stats, w->vy_zero && !restart, true, nullptr, keep ? (const double *)w->aty.p : (const double *)nullptr)) err = -1;
    };
    if (enqueue_cg_begin(w, w->ut.p, w->u.p, w->k, have)) return -1; // warm start = current u[0:m), abip.c:559
    w->aty_valid = false;
    int chunk = next_chunk(w);
    for (;;) {
      if (enqueue_cg_chunk(w, w->ut.p, chunk)) return -1;
      if (enqueue_cg_post(w, w->ut.p, keep ? w->aty.p : nullptr)) return -1;
      tail();
      if (err || sync_ctl(w)) return -1;
      if (w->hctl->cg_done) break;
      chunk = w->dist ? 2 : std::max(4, chunk); // not converged inside the chunk: everything behind it was a no-op; go on
    }
    w->last_cg_its = w->hctl->cg_it;
    w->tot_cg_its += w->hctl->cg_it;
    w->prof.cg_iters += w->hctl->cg_it;
    w->aty_valid = keep;
  }
  if (restart) w->fre_old = st->restart_fre; // abip.c:627
  w->wg_valid = true;
  w->stats_valid = true; w->avg_stats_valid = avg_stats;
  w->prof.admm_iters++;
  take_verdict(w, metric_out); // iterate_Q_norm_resd, abip.c:1951-2051 (scalar part, evaluated by k_finalize)
  return clear_halt(w);
}

// ------------------------------------------------------------------------------------------------
// PCG back-end on one GPU, LPs beyond the caches (C4): iterations STREAMED -- iteration q + 1 is enqueued before the host has seen the verdict of iteration q,
// so the device never waits for the host between two iterations (round 4: one control read per iteration = ~27 us of idle device + a 4.6 us copy launch).
// What makes that safe: everything an iteration needs is either known to the host in advance (j, k, mu, beta, the PCG tolerance factor) or decided on the
// device by k_finalize_stream -- exit test (halt 1), final check (halt 3), "the PCG did not converge inside the launches enqueued for it" (halt 2) -- and every
// kernel of the path falls through once halt is raised.  A stalled iteration is resumed where it stopped (more PCG iterations, then its tail again); the
// chain that was enqueued behind it ran as no-ops and is enqueued again.  The verdicts arrive through two pinned mirrors of the control block.
// Same kernels, same arguments, same order as admm_iteration: the trajectory is bit for bit the one the stepwise loop produces.
// *ran = iterations completed; *why = 0 budget used up, 1 exit test held at the last one, 3 final check held at the last one.
// ------------------------------------------------------------------------------------------------
bool stream_ok(const W *w) {
  return w->stream_on && w->hmir[0] && w->linsys == ABIP_HIP_LINSYS_INDIRECT && !w->dist && !w->xcd.on && !w->stgs->half_update && !(w->prof_mask & ~(1u << ABIP_HIP_K_CLASSES));
}
int admm_stream_pcg(W *w, long nmax, long *ran, double *metric_out, int *why) {
  const Dims d = dims(w);
  const Ctl *ctl = w->ctl.p;
  ABIPSettings *st = w->stgs;
  *ran = 0; *why = 0;
  if (!w->wg_valid) launch(w, ABIP_HIP_K_VEC, k_dot_wg, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, (const double *)w->g.p, st->rho_y, d, w->part.p, w->xwt);
  const abip_int j0 = w->j, k0 = w->k;
  const bool keep = w->aty_on && w->vy_zero && w->aty.p;
  FinStream fs;
  fs.fc.on = w->final_check ? 1 : 0; fs.fc.pfeasopt = (int)st->pfeasopt; fs.fc.ipm_pos = w->i > 0 ? 1 : 0;
  fs.fc.eps = st->eps; fs.fc.den = st->normalize ? (st->scale * w->sc_c * w->sc_b) : 1.0; fs.fc.nm_b = w->nm_b; fs.fc.nm_c = w->nm_c;
  fs.fc.max_admm = (long)st->max_admm_iters; fs.ipm_iter = (long)w->i; fs.ipm_last = (w->i + 1 >= st->max_ipm_iters) ? 1 : 0;
  auto tail = [&](long q) -> int { // back-substitution, update, stopping test, the streamed finalize + its event
    const bool avg_stats = ((j0 + q + 1) % 10 == 0);
    if (enqueue_cg_post(w, w->ut.p, keep ? w->aty.p : nullptr)) return -1;
    launch(w, ABIP_HIP_K_VEC, k_admm_update, w->NB, BS, upd_args(w, true, avg_stats, j0 + q), d, w->part.p, w->NB, ctl);
    fs.mirror = w->hmir[q & 1]; fs.fc.k0 = (long)(k0 + q);
    if (enqueue_q_and_finalize(w, avg_stats, false, true, nullptr, keep ? (const double *)w->aty.p : (const double *)nullptr, &fs)) return -1;
    return hipEventRecord(w->mir_ev[q & 1], w->stream) == hipSuccess ? 0 : -1;
  };
  // PCG iterations enqueued blind.  From one ADMM iteration to the next the count hardly moves (C4, 400 steps: unchanged 378 times, +1 four times, jumps only behind
  // an outer iteration: profiles/r05e_c4_cg_counts.txt), a launch past convergence costs ~2 us x 3 kernels, a stall ~0.2 ms once: enqueue exactly the last count.
  // ... unless it moved within the last eight iterations (the first iterations of an inner loop: 11, 11, 12, 12, 12, 13, ...): then one more.
  const char *fb = getenv("ABIP_HIP_STREAM_BLIND"); // tests: a fixed (too small) count, so that iterations stall and are resumed
  const int forced = fb ? std::max(1, atoi(fb)) : 0;
  auto blind = [&]() { return forced ? forced : std::max(2, std::min((int)w->m_glob, w->last_cg_its + (w->cg_steady < 8 ? 1 : 0))); };
  int chunk = blind();
  auto enqueue_iter = [&](long q, bool have) -> int {
    launch(w, ABIP_HIP_K_VEC, k_rhs, w->NB, BS, (const double *)w->u.p, (const double *)w->v.p, w->ut.p, (const double *)w->h.p, st->rho_y, w->g_th, d,
           w->part.p, w->NB, ctl, (const double *)w->gs, have ? (const double *)w->aty.p : (const double *)nullptr, have ? (double2 *)w->cg_pair.p : (double2 *)nullptr);
    if (enqueue_cg_begin(w, w->ut.p, w->u.p, k0 + q, have)) return -1; // warm start = current u[0:m), abip.c:559
    if (enqueue_cg_chunk(w, w->ut.p, chunk)) return -1;
    return tail(q);
  };
  long enq = 0, done = 0;
  while (done < nmax) {
    while (enq < nmax && enq - done < 2) { // (the first iteration of a call may find A'u_y kept by its predecessor; every later one does when `keep`)
      if (enqueue_iter(enq, enq == 0 ? (keep && w->aty_valid) : keep)) return -1;
      ++enq;
    }
    w->aty_valid = false;
    const int sl = (int)(done & 1);
    if (hipEventSynchronize(w->mir_ev[sl]) != hipSuccess) return -1;
    const Ctl &hm = *w->hmir[sl];
    if (hm.halt == 2) { // stalled: the PCG needs more iterations than were enqueued.  Resume it; what was enqueued behind ran as no-ops.
      ++w->stream_stalls;
      HIP_OK(hipMemsetAsync(&w->ctl.p->halt, 0, sizeof(int), w->stream));
      chunk = forced ? forced : std::max(4, chunk);
      if (enqueue_cg_chunk(w, w->ut.p, chunk) || tail(done)) return -1;
      enq = done + 1;
      continue;
    }
    memcpy(w->hctl, &hm, sizeof(Ctl));
    if (w->hctl->it_count != w->it_seen + 1) { fprintf(stderr, "abip_hip: streamed iteration %ld: the device reports %d completed iterations, expected %d\n", (long)(k0 + done), w->hctl->it_count, w->it_seen + 1); return -1; }
    w->cg_steady = (w->hctl->cg_it == w->last_cg_its) ? w->cg_steady + 1 : 0;
    w->last_cg_its = w->hctl->cg_it; w->tot_cg_its += w->hctl->cg_it; w->prof.cg_iters += w->hctl->cg_it;
    chunk = blind();
    w->tot_solves++; w->prof.kkt_solves++; w->prof.admm_iters++; ++w->stream_iters;
    take_verdict(w, metric_out);
    ++done;
    if (w->hctl->halt) { *why = w->hctl->halt; break; } // 1: exit test; 3: final check.  Whatever was enqueued behind falls through.
  }
  *ran = done;
  w->wg_valid = true; w->stats_valid = true; w->avg_stats_valid = ((j0 + done) % 10 == 0);
  w->aty_valid = keep && done > 0;
  if (w->stamp_mask) { // (device-side launch stamps of bench.py's roofline leg: read back once per call)
    size_t lo = 0, hi = 0;
    if (enqueue_stamp_readback(w, &lo, &hi)) return -1;
    HIP_OK(hipStreamSynchronize(w->stream));
    if (hi > lo) harvest_stamps(w, lo, hi);
  }
  return clear_halt(w);
}


// ------------------------------------------------------------------------------------------------
// One-XCD persistent launch (dev_xcd.h): set-up and one batch of iterations
// ------------------------------------------------------------------------------------------------
struct XcdVariant { int nz, rm, rn; const void *pcg, *direct, *pcg2 /* spread over several XCDs */, *direct2; };
const XcdVariant kXcdVariants[] = {
#define XV(a, b, c) {a, b, c, (const void *)k_lp_xcd<a, b, c, true>, (const void *)k_lp_xcd<a, b, c, false>, (const void *)k_lp_xcd<a, b, c, true, true>, (const void *)k_lp_xcd<a, b, c, false, true>}
    XV(2, 1, 1), XV(4, 1, 2), XV(6, 1, 2), XV(8, 2, 4),
#undef XV
};

// contiguous slices of the rows of M, balanced by (non-zeros + alpha per row); slices may be empty when there are fewer rows than workgroups
void xcd_slices(const host::HostCsr &M, int G, double alpha, std::vector<int> &bounds, long *max_nnz, int *max_rows, int *max_len) {
  const int rows = M.nrows;
  const long nnz = M.ptr[rows];
  bounds.assign(G + 1, 0);
  bounds[G] = rows;
  for (int g = 1; g < G; ++g) {
    const double target = ((double)nnz + alpha * rows) * g / G;
    int lo = bounds[g - 1], hi = rows;
    while (lo < hi) { const int mid = (lo + hi) / 2; if ((double)M.ptr[mid] + alpha * mid < target) lo = mid + 1; else hi = mid; }
    bounds[g] = lo;
  }
  *max_nnz = 0; *max_rows = 0; *max_len = 0;
  for (int g = 0; g < G; ++g) {
    *max_nnz = std::max<long>(*max_nnz, M.ptr[bounds[g + 1]] - M.ptr[bounds[g]]);
    *max_rows = std::max(*max_rows, bounds[g + 1] - bounds[g]);
  }
  for (int r = 0; r < rows; ++r) *max_len = std::max(*max_len, M.ptr[r + 1] - M.ptr[r]);
}
// the row weight that lets the slices fit the smallest kernel variant: a thread holds NZ non-zeros and R rows, both cost registers
void xcd_best_slices(const host::HostCsr &M, int G, std::vector<int> &bounds, long *max_nnz, int *max_rows, int *max_len, double row_cost = 0.0) {
  if (row_cost > 0.0) { xcd_slices(M, G, row_cost, bounds, max_nnz, max_rows, max_len); return; } // (direct: a row also costs a row of the dense inverse)
  double best_cost = 1e300;
  for (double alpha : {0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 16.0}) {
    std::vector<int> b; long nz; int rr, ll;
    xcd_slices(M, G, alpha, b, &nz, &rr, &ll);
    const double cost = 3.0 * std::ceil((double)nz / XTB) + 4.0 * std::ceil((double)rr / XTB) + 1e-6 * ((double)nz + rr);
    if (cost < best_cost) { best_cost = cost; bounds = b; *max_nnz = nz; *max_rows = rr; *max_len = ll; }
  }
}

// The host half of the decision (no device needed: abip_hip_xcd_plan exports it for the CPU tests): how many workgroups, which slices of A and A'
// they own, which kernel variant holds them, how the LDS is divided.  false = the problem stays on the launch path.
bool xcd_plan(XcdPlan &x, const host::HostCsr &hA, const host::HostCsr &hAt, bool pcg, std::vector<int> &mb, std::vector<int> &nb, long *nzA, long *nzT, int *rA, int *rT,
              int *lA, int *lT) {
  const long m = hA.nrows, n = hAt.nrows;
  // direct: inv(rho I + A A') is kept dense -- 8 m^2 bytes read per iteration, m^3 flops of set-up.  Staircase LPs on eight XCDs against the launch path
  // (profiles/r05zzk_*): m = 4000: 24.8 k / 6.7 k it/s, solve to 1e-6 with the set-up 0.34 / 0.45 s; 5000: 18.0 k / 6.4 k, 0.60 / 0.85 s; 6000: 13.6 k / 6.1 k,
  // 0.84 / 0.99 s; 8000: 7.7 k / 4.4 k, 1.59 / 1.53 s -- the set-up eats the lead there (ABIP_HIP_XCD_MMAX moves the limit; 4096 until the launch went to eight XCDs)
  { const char *e = getenv("ABIP_HIP_XCD_MMAX"); if (!pcg && m > (e ? atol(e) : 6144L)) return false; }
  // How many XCDs.  More XCDs shrink a slice -- the gathers and row sums of an exchange -- and cost ~0.5 us per exchange for stores written through to where the
  // other XCDs' loads find them, plus polls that go to the memory side.  PCG back-end, window rates with 32 / 64 / 128 / 256 workgroups on the last kernels of
  // round 5 (scripts/xcd_g_sweep.py, profiles/r05zzb_*, r05zzc_*): 40.6 k non-zeros 4 177 / 3 557 / 3 479 / 3 154 it/s; 56 k 9 763 / 9 143 / 10 566 / 10 328;
  // 97.8 k 2 267 / 2 583 / 2 977 / 2 877; c3 (136 k) - / 2 100 / 2 512 / 2 557 (whole solve 5.31 / 4.49 / 4.29 s); 140 k - / 6 204 / 7 556 / 8 638;
  // 260 k - / - / 4 649 / 4 967; 321 k - / - / 1 904 / 2 196; 500 k - / - / - / 3 358 (the launch path on the last three: 3 050, 859, 2 076)
  // -> one XCD up to 48 k non-zeros, four up to 120 k, eight beyond, as far as a variant fits (~8e5; two never win; ABIP_HIP_XCD_G forces 32 .. 256, either back-end).  Until the polls were
  // tamed (slot-major granules, the first round delayed) eight XCDs lost to four on c3 and the launch stopped at four.
  x.G = 32;
  {
    const char *e = getenv("ABIP_HIP_XCD_G");
    const int ge = e ? atoi(e) : 0;
    if (ge == 32 || ge == 64 || ge == 128 || ge == 256) x.G = ge;
    else if (pcg) {
      const long nnz = hA.ptr[hA.nrows];
      x.G = nnz <= 48000L ? 32 : nnz <= 120000L ? 128 : 256;
    } else x.G = m <= 1024 ? 32 : m <= 1500 ? 128 : 256; // direct: the rows of the dense inverse dominate beyond m = 1024 (there the dense product changes form too). Staircase LPs,
                                                         // 32 / 128 / 256 workgroups on the last kernels of round 5 (profiles/r05zzf_*): m = 816: 92.4 / 74.5 / 70.6 k it/s, 1000: 77.6 / 71.1 / 67.6,
                                                         // 1200: 50.6 / 67.3 / 66.6, 1400: - / 66.4 / 66.2, 1600: 31.3 / 62.1 / 65.8, 2000: - / 51.1 / 63.7, 3000: - / 32.7 / 45.5, 4000: - / 22.1 / 33.1
  }
  x.nxcd = x.G / 32;
  xcd_best_slices(hA, x.G, mb, nzA, rA, lA, pcg ? 0.0 : (double)m);
  xcd_best_slices(hAt, x.G, nb, nzT, rT, lT);
  if (std::max(*lA, *lT) > 512) return false; // rows are added up by one thread each
  const XcdVariant *pick = nullptr;
  for (const XcdVariant &v : kXcdVariants)
    if (std::max(*nzA, *nzT) <= (long)v.nz * XTB && *rA <= v.rm * XTB && *rT <= v.rn * XTB) { pick = &v; break; }
  if (!pick) return false;
  // (whatever fits a variant stays here: 5.0e5 non-zeros, 6 per thread on 256 workgroups, 3 352 it/s against the launch path's 2 018; 7.4e5, 8 per thread, 2 405 against 1 639 --
  // until round 5 more than 4 per thread went to the launch path, which had caught up with the four-XCD launch there)
  x.NZ = pick->nz; x.RM = pick->rm; x.RN = pick->rn;
  x.kern = pcg ? (x.G > 32 ? pick->pcg2 : pick->pcg) : (x.G > 32 ? pick->direct2 : pick->direct);
  x.n_pad = (int)((n + 63) / 64 * 64); x.m_pad = (int)((m + 63) / 64 * 64);
  size_t words = 2 * (size_t)x.NZ * XTB + XKS + XWAVES * XKS + 96 + 16 + 32;
  if (!pcg) words += (size_t)x.m_pad + (size_t)x.RM * XTB;
  if (!pcg && x.m_pad <= 64 * 16) words += (size_t)(64 * 16 - x.m_pad); // the zeroed pad behind the last LDS row of the dense inverse (dev_xcd.h: every row is read out to 1024 columns)
  constexpr size_t kLdsWords = (160 * 1024 - 512) / sizeof(double); // (the kernel's one static word lives in the remaining 512 bytes)
  if (words > kLdsWords) return false;
  x.minv_lds_rows = 0;
  if (!pcg) { // what is left of the LDS keeps rows of the dense inverse
    const size_t room = kLdsWords - words;
    x.minv_lds_rows = (int)std::min<size_t>(room / (size_t)x.m_pad, (size_t)*rA);
    words += (size_t)x.minv_lds_rows * x.m_pad;
  }
  x.lds = std::max<size_t>(words * sizeof(double), (size_t)XCD_LDS_MIN);
  return x.lds <= 160 * 1024;
}

// Decide whether the LP runs its inner loop as the one-XCD persistent launch and prepare it.  Never fails the set-up: when the problem does
// not fit (or ABIP_HIP_XCD=0) the launch path stays in charge.
void xcd_setup(W *w, const host::HostCsr &hA, const host::HostCsr &hAt) {
  XcdPlan &x = w->xcd;
  x.on = false;
  { const char *e = getenv("ABIP_HIP_XCD"); if (e && atoi(e) == 0) return; }
  if (w->dist) return;
  hipDeviceProp_t prop; int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
  if (prop.multiProcessorCount != 256 || !strstr(prop.gcnArchName, "gfx950")) return; // 8 XCDs x 32 CUs is what the placement argument needs
  const bool pcg = (w->linsys == ABIP_HIP_LINSYS_INDIRECT);
  std::vector<int> mb, nb;
  long nzA = 0, nzT = 0; int rA = 0, rT = 0, lA = 0, lT = 0;
  if (!pcg && !w->A) return;
  if (!xcd_plan(x, hA, hAt, pcg, mb, nb, &nzA, &nzT, &rA, &rT, &lA, &lT)) return;
  // (the cap is per kernel function, not per work: two works of one variant with different LDS sizes must not lower it under each other)
  const bool chatty = getenv("ABIP_HIP_XCD_VERBOSE") != nullptr;
  if (hipFuncSetAttribute(x.kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) { (void)hipGetLastError(); if (chatty) printf("[xcd] off: the LDS cap could not be raised\n"); return; }
  { // one workgroup per CU must be able to be resident at all: 256 workgroups on 256 CUs, every one of them waited for by the others
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, x.kern, XTB, x.lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); if (chatty) printf("[xcd] off: occupancy query says %d workgroups per CU\n", per_cu); return; }
  }
  const std::vector<int> zero2(XSTAT_N, 0);
  const std::vector<unsigned> zero1(1, 0u);
  bool bad = x.mb.upload(mb, w->stream) || x.nb.upload(nb, w->stream) || x.xstat.upload(zero2, w->stream) || x.tickets.upload(zero1, w->stream) ||
             x.xn0.alloc(2 * (size_t)x.n_pad) || x.xn1.alloc(2 * (size_t)x.n_pad) || x.xm0.alloc(2 * (size_t)x.m_pad) || x.xm1.alloc(2 * (size_t)x.m_pad) ||
             (!pcg && x.xnv.alloc(2 * (size_t)x.n_pad)) ||
             x.sc.alloc(2 * (size_t)XG * XKS) || x.tolf.alloc(x.max_batch) || x.mu_tab.alloc(XcdPlan::MU_TAB) || x.xlog.alloc((size_t)XcdPlan::LOG_CAP * XLOG_W) ||
             x.snap.alloc(9 * (size_t)w->LV);
  x.snap_len = (size_t)w->LV;
  if (!bad && !pcg) bad = hipMemsetAsync(x.xnv.p, 0, sizeof(double) * 2 * x.n_pad, w->stream) != hipSuccess;
  if (!bad) bad = hipMemsetAsync(x.xn0.p, 0, sizeof(double) * 2 * x.n_pad, w->stream) != hipSuccess || hipMemsetAsync(x.xn1.p, 0, sizeof(double) * 2 * x.n_pad, w->stream) != hipSuccess ||
                  hipMemsetAsync(x.xm0.p, 0, sizeof(double) * 2 * x.m_pad, w->stream) != hipSuccess || hipMemsetAsync(x.xm1.p, 0, sizeof(double) * 2 * x.m_pad, w->stream) != hipSuccess ||
                  hipMemsetAsync(x.sc.p, 0, sizeof(u32x4) * 2 * XG * XKS, w->stream) != hipSuccess;
  if (!bad) bad = hipHostMalloc((void **)&x.htolf, sizeof(double) * x.max_batch, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&x.hstat, XSTAT_N * sizeof(int), hipHostMallocDefault) != hipSuccess ||
                  hipHostMalloc((void **)&x.hmu_tab, sizeof(double) * XcdPlan::MU_TAB, hipHostMallocDefault) != hipSuccess ||
                  hipHostMalloc((void **)&x.hlog, sizeof(double) * XcdPlan::LOG_CAP * XLOG_W, hipHostMallocDefault) != hipSuccess;
  if (bad) { (void)hipGetLastError(); x.release(); return; }
  if (!pcg) { // direct: the x block (-I) eliminated first, the y block's Schur complement rho I + A A' inverted densely on the device
    const int m = (int)w->m, T = x.m_pad;
    std::vector<double> S((size_t)T * T, 0.0);
    for (int i = 0; i < T; ++i) S[(size_t)i * T + i] = i < m ? w->stgs->rho_y : 1.0;
    for (int j = 0; j < hAt.nrows; ++j) // column j of A: all pairs of its entries (rows ascending)
      for (int p = hAt.ptr[j]; p < hAt.ptr[j + 1]; ++p)
        for (int q = hAt.ptr[j]; q <= p; ++q) S[(size_t)hAt.idx[p] * T + hAt.idx[q]] += hAt.val[p] * hAt.val[q];
    DBuf<double> dS;
    if (dS.upload(S, w->stream) || dense_spd_inverse(dS.p, T, w->stream, x.Minv)) { dS.release(); (void)hipGetLastError(); x.release(); return; }
    dS.release();
    x.ldM = T;
    // guard: (rho I + A A') (Minv v) against v on one pseudo-random vector (host arithmetic on the downloaded inverse)
    std::vector<double> Mh((size_t)T * T), v, y(m, 0.0), t(hAt.nrows, 0.0), r(m, 0.0);
    host::guard_rhs(m, v);
    if (hipMemcpyAsync(Mh.data(), x.Minv.p, sizeof(double) * (size_t)T * T, hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) { (void)hipGetLastError(); x.release(); return; }
    for (int i = 0; i < m; ++i) { double acc = 0.0; for (int c = 0; c < m; ++c) acc += Mh[(size_t)i * T + c] * v[c]; y[i] = acc; }
    for (int j = 0; j < hAt.nrows; ++j) { double acc = 0.0; for (int p = hAt.ptr[j]; p < hAt.ptr[j + 1]; ++p) acc += hAt.val[p] * y[hAt.idx[p]]; t[j] = acc; }
    for (int i = 0; i < m; ++i) r[i] = w->stgs->rho_y * y[i] - v[i];
    for (int j = 0; j < hAt.nrows; ++j) for (int p = hAt.ptr[j]; p < hAt.ptr[j + 1]; ++p) r[hAt.idx[p]] += hAt.val[p] * t[j];
    double num = 0.0, den = 0.0;
    for (int i = 0; i < m; ++i) { num += r[i] * r[i]; den += v[i] * v[i]; }
    const double res = std::sqrt(num) / std::max(std::sqrt(den), 1e-300);
    if (getenv("ABIP_HIP_XCD_VERBOSE")) printf("[xcd] dense inverse of rho I + A A' (%d x %d): residual %.2e\n", m, m, res);
    if (!(res <= 1e-9)) { x.release(); return; }
  }
  // h_y + A h_x (see abip_hip_solve_begin): zero like h itself until a solve begins (the unit-level abip_hip_kkt_solve may run before that)
  if (w->hAh.alloc(w->m) || hipMemsetAsync(w->hAh.p, 0, sizeof(double) * w->m, w->stream) != hipSuccess) { (void)hipGetLastError(); x.release(); return; }
  x.tag = 0; x.launches = 0; x.tickets_used = 0;
  x.on = true;
  { const char *e = getenv("ABIP_HIP_XCD_OUTER"); x.outer = !(e && atoi(e) == 0); }
  { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) x.ticks_per_ms = (double)khz; }
#ifdef ABIP_HIP_TEST_HOOKS
  { const char *e = getenv("ABIP_HIP_XCD_GIVEUP_AT"); x.desert_at = e ? atoi(e) : -1; }
#endif
  // stand-alone solves (set-up, BB look-ahead) through the persistent kernel: worth it for the PCG back-end (a launch-path solve is 3 launches per PCG
  // iteration); the direct back-end's launch-path solve is 4 launches in all and wins (ABIP_HIP_XCD_SOLVES=0 / 1 forces either)
  { const char *e = getenv("ABIP_HIP_XCD_SOLVES"); w->xcd_solves = e ? atoi(e) != 0 : pcg; }
  if (getenv("ABIP_HIP_XCD_VERBOSE"))
    printf("[xcd] %d workgroups on %d XCD(s): slices of A  <= %ld nnz, %d rows (longest row %d); of A' <= %ld nnz, %d rows (longest %d); NZ %d RM %d RN %d, LDS %zu B (%d rows of the dense inverse)\n", x.G, x.nxcd, nzA, rA, lA, nzT, rT, lT,
           x.NZ, x.RM, x.RN, x.lds, x.minv_lds_rows);
}

// ---- launching the persistent kernel ---------------------------------------------------------------------------------------------
// Persistent launches are serialised process-wide: a launch is 256 workgroups that wait for each other, and two of them dealt onto the CUs at the same
// time (two works driven from two threads) could each hold CUs the other needs.  Held from the launch to the read-back that ends it.
std::mutex g_xcd_mutex;

DBuf<double> *xcd_state_vecs(W *w, int q) {
  DBuf<double> *v[] = {&w->u, &w->v, &w->u_avg, &w->v_avg, &w->u_sum, &w->v_sum, &w->u_avgc, &w->v_avgc};
  return v[q];
}
// Everything of an XcdArgs that does not depend on what the launch is asked to do
void xcd_fill(W *w, XcdArgs &a) {
  XcdPlan &x = w->xcd;
  ABIPSettings *st = w->stgs;
  const bool pcg = (w->linsys == ABIP_HIP_LINSYS_INDIRECT);
  a.Ap = w->dA.ptr.p; a.Ai = w->dA.idx.p; a.Ax = w->dA.val.p;
  a.Tp = w->dAt.ptr.p; a.Ti = w->dAt.idx.p; a.Tx = w->dAt.val.p;
  a.mb = x.mb.p; a.nb = x.nb.p;
  a.G = x.G; a.m = (int)w->m; a.n = (int)w->n; a.MP = w->MP;
  a.upd = upd_args(w, true, false, w->j);
  a.h = w->h.p; a.hAh = w->hAh.p; a.wD = st->normalize ? w->wD.p : nullptr; a.wE = st->normalize ? w->wE.p : nullptr;
  a.Mjac = pcg ? w->cg_M.p : nullptr; a.Minv = x.Minv.p; a.ldM = x.ldM; a.minv_lds_rows = x.minv_lds_rows;
  a.g_th = w->g_th;
  a.xn0 = x.xn0.p; a.xn1 = x.xn1.p; a.xm0 = x.xm0.p; a.xm1 = x.xm1.p; a.xnv = x.xnv.p; a.sc = x.sc.p; a.n_pad = x.n_pad; a.m_pad = x.m_pad;
  a.nxcd = x.nxcd;
  a.ctl = w->ctl.p; a.xstat = x.xstat.p;
  a.j0 = (long)w->j;
  a.thr = w->gamma * w->mu; a.sentinel = (double)st->max_admm_iters;
  a.tolf = x.tolf.p; a.cg_max_its = (int)w->m_glob;
  a.part = w->part.p; a.npart = w->NB;
  a.fc.on = w->final_check ? 1 : 0; a.fc.pfeasopt = (int)st->pfeasopt; a.fc.ipm_pos = w->i > 0 ? 1 : 0;
  a.fc.eps = st->eps; a.fc.den = st->normalize ? (st->scale * w->sc_c * w->sc_b) : 1.0; a.fc.nm_b = w->nm_b; a.fc.nm_c = w->nm_c;
  a.fc.k0 = (long)w->k; a.fc.max_admm = (long)st->max_admm_iters;
  a.outer.on = 0; a.outer.i = (long)w->i;
  a.desert = -1;
}
// Launch, wait, read the control block back.  0 = the launch ran to its end; 1 = a wait inside it gave up (another kernel held CUs the launch needed, or the
// placement the protocol relies on did not come about): the iterate is back to what it was before the launch, the persistent launch is switched off for
// this work and the caller goes on along the launch path; < 0 = device error.  `vec` (solve-only launches): the one vector the launch overwrites.
int xcd_launch(W *w, XcdArgs &a, double *vec) {
  XcdPlan &x = w->xcd;
  if (x.tag > 0x70000000u) { // tags only ever grow within the life of the buffers: start over long before they wrap
    HIP_OK(hipMemsetAsync(x.sc.p, 0, sizeof(u32x4) * 2 * XG * XKS, w->stream)); // (only the flags carry tags)
    x.tag = 0;
  }
  a.tag0 = x.tag;
  // the way back: the state the launch is about to change
  if (vec) HIP_OK(hipMemcpyAsync(x.snap.p, vec, sizeof(double) * x.snap_len, hipMemcpyDeviceToDevice, w->stream));
  else for (int q = 0; q < 8; ++q) HIP_OK(hipMemcpyAsync(x.snap.p + (size_t)(q + 1) * x.snap_len, xcd_state_vecs(w, q)->p, sizeof(double) * x.snap_len, hipMemcpyDeviceToDevice, w->stream));
  void *params[] = {&a};
  const double t_launch = now_ms();
  W::Ev *ev = nullptr;
  if (((w->prof_mask >> ABIP_HIP_K_XCD) & 1u) && !vec) { // (iteration launches only: the stand-alone solves are not iterations)
    if (w->ev_used == w->ev_pool.size()) { W::Ev e; e.cls = ABIP_HIP_K_XCD; (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); w->ev_pool.push_back(e); }
    ev = &w->ev_pool[w->ev_used++];
    ev->cls = ABIP_HIP_K_XCD; ev->tag = -1;
    (void)hipEventRecord(ev->a, w->stream);
  }
  {
    std::lock_guard<std::mutex> lock(g_xcd_mutex);
    a.tickets = x.tickets.p; a.ticket_base = x.tickets_used;
    x.tickets_used += (unsigned)x.G;
    if (x.desert_at >= 0 && (int)x.launches == x.desert_at) a.desert = x.G - 1;
    HIP_OK(hipLaunchKernel(x.kern, dim3(256), dim3(XTB), params, x.lds, w->stream));
    if (ev) (void)hipEventRecord(ev->b, w->stream);
    x.launches++;
    HIP_OK(hipMemcpyAsync(x.hstat, x.xstat.p, XSTAT_N * sizeof(int), hipMemcpyDeviceToHost, w->stream));
    if (sync_ctl(w)) return -1;
  }
  if (!x.hstat[0]) {
    x.tag += (unsigned)x.hstat[1]; x.exchanges += x.hstat[1];
    return 0;
  }
  // ---- a wait gave up: post-mortem, restore, leave the persistent launch for good ----
  x.giveups++;
  fprintf(stderr, "abip_hip: persistent launch %u abandoned after %.1f ms (a wait for exchange %d gave up: rank %d, wait site %d; %d of %d ranks had opened it) -- the launch path takes over\n",
          x.launches - 1, now_ms() - t_launch, x.hstat[2], x.hstat[3], x.hstat[4],
          [&] { int c = 0; for (int g = 0; g < x.G; ++g) c += (x.hstat[8 + 2 * g] == x.hstat[2]); return c; }(), x.G);
  if (getenv("ABIP_HIP_XCD_VERBOSE")) {
    for (int g = 0; g < x.G; ++g) fprintf(stderr, "  rank %3d: last exchange opened %d (wait site %d)\n", g, x.hstat[8 + 2 * g], x.hstat[9 + 2 * g]);
  }
  if (vec) HIP_OK(hipMemcpyAsync(vec, x.snap.p, sizeof(double) * x.snap_len, hipMemcpyDeviceToDevice, w->stream));
  else for (int q = 0; q < 8; ++q) HIP_OK(hipMemcpyAsync(xcd_state_vecs(w, q)->p, x.snap.p + (size_t)(q + 1) * x.snap_len, sizeof(double) * x.snap_len, hipMemcpyDeviceToDevice, w->stream));
  HIP_OK(hipMemsetAsync(x.xstat.p, 0, XSTAT_N * sizeof(int), w->stream)); // (a later work on the same buffers must not see the flag)
  HIP_OK(hipStreamSynchronize(w->stream));
  x.on = false;
  w->wg_valid = false;
  return 1;
}

// K z = rhs in place by ONE launch of the persistent kernel in its solve-only mode (warm: l-vector whose y block starts the PCG, or null);
// leaves u_t'h in the partial table like the launch path's post-solve kernels.  Returns the PCG iterations (0 for the direct back-end), < 0 on error,
// -2 when the launch was abandoned (rhs is intact: the caller solves on the launch path).
int xcd_solve(W *w, double *rhs, const double *warm, abip_int iter) {
  XcdPlan &x = w->xcd;
  XcdArgs a{};
  xcd_fill(w, a);
  a.max_iters = 1; a.solve_only = 1; a.srhs = rhs; a.swarm = warm;
  if (w->linsys == ABIP_HIP_LINSYS_INDIRECT) {
    x.htolf[0] = cg_tol_factor(w, iter);
    HIP_OK(hipMemcpyAsync(x.tolf.p, x.htolf, sizeof(double), hipMemcpyHostToDevice, w->stream));
  }
  const int rc = xcd_launch(w, a, rhs);
  if (rc) return rc < 0 ? -1 : -2;
  return w->linsys == ABIP_HIP_LINSYS_INDIRECT ? w->hctl->cg_it : 0;
}

// Run up to nb ADMM iterations (k, j), (k+1, j+1), ... as one launch; *ran = iterations that ran (the exit test, or the final check, stops it).
// Returns as xcd_launch (1: abandoned, nothing ran).
int xcd_batch(W *w, int nb, int *ran, double *metric_out) {
  XcdPlan &x = w->xcd;
  const bool pcg = (w->linsys == ABIP_HIP_LINSYS_INDIRECT);
  nb = std::min(nb, x.max_batch);
  XcdArgs a{};
  xcd_fill(w, a);
  a.max_iters = nb;
  if (pcg) {
    for (int q = 0; q < nb; ++q) x.htolf[q] = cg_tol_factor(w, w->k + q);
    HIP_OK(hipMemcpyAsync(x.tolf.p, x.htolf, sizeof(double) * nb, hipMemcpyHostToDevice, w->stream));
  }
  const int rc = xcd_launch(w, a, nullptr);
  if (rc) return rc;
  x.batches++;
#ifdef XCD_PROF
  { static long acc[8] = {0}; for (int q = 0; q < 8; ++q) acc[q] += (unsigned)x.hstat[600 + q];
    fprintf(stderr, "[xcd prof] cumulative us: %.0f %.0f %.0f %.0f %.0f %.0f %.0f (PCG loop: put, publish, collect, gather, rows, tail | direct: rhs+E1, E_w, all-gather, dense, E_y+E_dh, update+E_u, q+E_fin)\n", acc[0] * 0.01, acc[1] * 0.01, acc[2] * 0.01, acc[3] * 0.01, acc[4] * 0.01, acc[5] * 0.01, acc[6] * 0.01); }
#endif
  *ran = w->hctl->it_count - w->it_seen;
  if (*ran < 1 || *ran > nb) return -1;
  take_verdict(w, metric_out);
  w->tot_solves += *ran; w->prof.kkt_solves += *ran; w->prof.admm_iters += *ran;
  if (pcg) { w->last_cg_its = w->hctl->cg_it; w->tot_cg_its += w->hctl->xcd_cg_total; w->prof.cg_iters += w->hctl->xcd_cg_total; }
  w->wg_valid = false; // the launch path's partial table does not hold S_WG
  w->stats_valid = true; w->avg_stats_valid = ((w->j + *ran) % 10 == 0);
  return 0;
}

// ---- a launch that spans outer iterations (dev_xcd.h XcdOuter) -------------------------------------------------------------------
// May the loop of abip.c:2102-2294 run inside the kernel for this work and these settings?  What stays outside: half_update (its clean-up pass),
// the inner caps of dense problems (pow on the device), a PCG tolerance schedule other than 1 / k^2.
bool xcd_outer_ok(const W *w) {
  const ABIPSettings *st = w->stgs;
  if (!w->xcd.on || !w->xcd.outer || w->dist || st->half_update) return false;
  if (std::min(w->sp, st->sparsity_ratio) > 0.2) return false;                              // abip.c:2104-2115: inner_stopper = max_admm_iters only then
  if (w->linsys == ABIP_HIP_LINSYS_INDIRECT && st->cg_rate != 2.0) return false;           // indirect.c:406-407
  if (st->restart_fre <= 0) return false;
  return true;
}
void print_summary_row(const W *w, const double *row);
// One launch from the inner loop (phase 0: iteration (k, j) is next) or from a pending outer end (phase 1).  Updates the host's copy of the loop state from
// what the launch hands back.  Returns as xcd_launch; *ran = ADMM iterations that ran; *reason = XR_*.
int xcd_run(W *w, int phase, long max_steps, long *ran, int *reason) {
  XcdPlan &x = w->xcd;
  ABIPSettings *st = w->stgs;
  XcdArgs a{};
  xcd_fill(w, a);
  a.max_iters = 0;
  XcdOuter &o = a.outer;
  o.on = 1; o.phase = phase; o.avg_crit = (int)st->avg_criterion;
  o.adaptive = (int)st->adaptive; o.lookback = (int)st->adaptive_lookback; o.hybrid_mu = (int)st->hybrid_mu;
  o.i = (long)w->i; o.fre_old = (long)w->fre_old;
  o.mu = w->mu; o.beta = w->beta; o.sigma = w->sigma; o.gamma = w->gamma; o.dyn_sigma = st->dynamic_sigma;
  o.max_ipm = (long)st->max_ipm_iters; o.inner_stopper = (long)w->inner_stopper; o.restart_thresh = (long)st->restart_thresh; o.restart_fre = (long)st->restart_fre;
  // a launch is kept to about a second of device time: the iteration budget from the rate of the launches so far (a first guess from the size), the
  // wall-clock slice looked at once per outer iteration -- also what lets max_time (abip.c:2217-2221) take effect between launches
  const double slice_ms = 1000.0;
  long budget = x.its_per_ms > 0.0 ? (long)std::max(64.0, x.its_per_ms * slice_ms) : ((long)w->dA.val.n + (long)w->m * (w->linsys == ABIP_HIP_LINSYS_DIRECT ? (long)w->m : 0L) < 200000L ? 8192L : 1024L);
  if (!w->batch_ok) budget = 1; // ABIP_HIP_BATCH=0: one control read per iteration (the same kernel, the same bits)
  o.max_steps = std::max<long>(1, std::min(max_steps, budget));
  const double left_ms = std::max(1.0, (st->max_time - ((double)clock() - w->cpu0) / CLOCKS_PER_SEC) * 1e3);
  o.slice_ticks = (unsigned long long)(std::min(slice_ms, left_ms) * x.ticks_per_ms);
  o.eps_cor = st->eps_cor; o.eps_pen = st->eps_pen; o.hybrid_thresh = st->hybrid_thresh; o.dyn_sigma_second = st->dynamic_sigma_second;
  { const char *e = getenv("ABIP_HIP_BB_REUSE"); o.bb_reuse = (e && atoi(e) == 0) ? 0 : 1; }
  { // update_barrier_dynamic_2 (abip.c:982-992) applied 1, 2, ... times to the current mu: pow() stays on the host
    double mu = w->mu;
    for (int q = 0; q < XcdPlan::MU_TAB; ++q) { mu *= std::min(st->dynamic_x * mu, std::pow(mu, st->dynamic_sigma)); x.hmu_tab[q] = mu; }
    HIP_OK(hipMemcpyAsync(x.mu_tab.p, x.hmu_tab, sizeof(double) * XcdPlan::MU_TAB, hipMemcpyHostToDevice, w->stream));
  }
  o.mu_tab = x.mu_tab.p; o.mu_tab_n = XcdPlan::MU_TAB;
  o.log = x.xlog.p; o.log_cap = XcdPlan::LOG_CAP;
  o.a_up = w->a_up.p; o.a_vp = w->a_vp.p; o.a_u = w->a_u.p; o.a_v = w->a_v.p; o.a_un = w->a_un.p; o.a_vn = w->a_vn.p;
  const double t0 = now_ms();
  const int rc = xcd_launch(w, a, nullptr);
  if (rc) return rc;
  const double dt = now_ms() - t0;
  x.whole_launches++;
  const XcdOut &r = w->hctl->xo;
  *ran = r.ran; *reason = r.reason;
#ifdef XCD_PROF
  { static long lap[8] = {0}; for (int q = 0; q < 8; ++q) lap[q] += (unsigned)x.hstat[600 + q];
    fprintf(stderr, "[xcd prof] laps, cumulative us: %.0f %.0f %.0f %.0f %.0f %.0f %.0f (PCG loop: put, publish, collect, gather, rows, tail | direct: rhs+E1, E_w, all-gather, dense, E_y+E_dh, update+E_u, q+E_fin)\n", lap[0] * 0.01, lap[1] * 0.01, lap[2] * 0.01, lap[3] * 0.01, lap[4] * 0.01, lap[5] * 0.01, lap[6] * 0.01); }
  { static double acc[8] = {0}; for (int q = 0; q < 8; ++q) acc[q] += (unsigned)x.hstat[610 + q];
    fprintf(stderr, "[xcd prof] cumulative: launches %.0f us | ADMM iterations %.0f us (their PCG loops %.0f us, %.0f PCG iterations) | look-ahead steps %.0f us (PCG loops %.0f us, %.0f PCG iterations) | outer end / begin %.0f us\n",
            acc[0] * 0.01, acc[1] * 0.01, acc[4] * 0.01, acc[7], acc[2] * 0.01, acc[5] * 0.01, acc[6], acc[3] * 0.01); }
#endif
  if (getenv("ABIP_HIP_XCD_VERBOSE")) printf("[xcd] launch %u (entry phase %d, budget %ld): %.3f ms, reason %d, phase %d, ran %ld, outer iterations closed %d (look-aheads %d), i %ld j %ld k %ld mu %.3e beta %.6f avg_crit %d final_check %d\n",
                                             x.launches - 1, phase, o.max_steps, dt, r.reason, r.phase, r.ran, r.outer_done, r.bb_lookaheads, r.i, r.j, r.k, r.mu, r.beta, r.avg_crit, r.final_check);
  if (r.ran < 0 || r.ran > o.max_steps) return -1;
  if (r.ran >= 64 && dt > 0.0) x.its_per_ms = (double)r.ran / dt;
  if (st->verbose && r.log_n > 0) { // the rows print_summary would have written, one per outer iteration closed on the device
    HIP_OK(hipMemcpyAsync(x.hlog, x.xlog.p, sizeof(double) * (size_t)r.log_n * XLOG_W, hipMemcpyDeviceToHost, w->stream));
    HIP_OK(hipStreamSynchronize(w->stream));
    for (int q = 0; q < r.log_n; ++q) { x.hlog[(size_t)q * XLOG_W + 10] = (t0 - w->t_solve0) + x.hlog[(size_t)q * XLOG_W + 10] / x.ticks_per_ms; print_summary_row(w, x.hlog + (size_t)q * XLOG_W); }
  }
  // ---- the loop state as the launch left it ----
  w->i = (abip_int)r.i; w->j = (abip_int)r.j; w->k = (abip_int)r.k;
  w->mu = r.mu; w->beta = r.beta; st->dynamic_sigma = r.dyn_sigma; w->final_check = r.final_check;
  if (r.outer_done > 0) w->fre_old = 0;
  if (r.ran > 0) { st->avg_criterion = w->hctl->avg_crit; w->it_seen = w->hctl->it_count; } // abip.c:2042, 2048
  if (r.ran > 0 || r.outer_done > 0) { w->stats_valid = r.stats_valid != 0; w->avg_stats_valid = r.avg_stats != 0; } // (a launch that changed nothing leaves the host's sums as valid as they were)
  w->wg_valid = false;
  w->r.last_admm_iter = -1;
  w->tot_solves += r.solves; w->prof.kkt_solves += r.solves; w->prof.admm_iters += r.ran;
  if (w->linsys == ABIP_HIP_LINSYS_INDIRECT) { w->last_cg_its = r.last_cg; w->tot_cg_its += r.cg_total; w->prof.cg_iters += r.cg_total; w->tot_cg_skipped += r.cg_skipped; w->prof.cg_iters_skipped += r.cg_skipped; }
  x.outer_done += r.outer_done; x.lookaheads += r.bb_lookaheads;
  w->phase = r.phase == 0 ? PH_INNER : (r.phase == 1 ? PH_OUTER_END : PH_OUTER_BEGIN);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// barrier-parameter strategies (abip.c:753-992) -- host scalars; the LOQO rule needs one device reduction
// ------------------------------------------------------------------------------------------------
double gamma_ladder(double ratio, double top) {
  if (ratio > 10.0) return top;
  if (ratio > 1.0) return 1.0;
  if (ratio > 0.5) return 0.9;
  if (ratio > 0.1) return 0.8;
  if (ratio > 0.05) return 0.7;
  if (ratio > 0.01) return 0.6;
  if (ratio > 0.005) return 0.5;
  if (ratio > 0.001) return 0.4;
  return 0.3;
}
void update_barrier(W *w) { // "tedious" table, abip.c:753-921
  const Resid &r = w->r;
  double sigma, gamma;
  const double ratio = w->mu / w->stgs->eps;
  const double err_ratio = std::max(std::max(r.res_pri, r.res_dual), r.rel_gap) / w->stgs->eps;
  const double mx = std::max(w->sp, w->stgs->sparsity_ratio), mn = std::min(w->sp, w->stgs->sparsity_ratio);
  if (mx > 0.4 || mn > 0.1) {
    gamma = gamma_ladder(ratio, 2.0);
    if (err_ratio > 6 && err_ratio <= 10) sigma = 0.5;
    else if (err_ratio > 3 && err_ratio <= 6) { sigma = 0.6; gamma *= 0.8; }
    else if (err_ratio > 1 && err_ratio <= 3) { w->final_check = 1; gamma *= 0.4; sigma = ratio < 0.1 ? 0.8 : 0.7; }
    else sigma = w->sigma;
  } else {
    gamma = gamma_ladder(ratio, 3.0);
    if (err_ratio > 6 && err_ratio <= 10) { sigma = 0.82; gamma *= 0.8; }
    else if (err_ratio > 4 && err_ratio <= 6) { sigma = 0.84; gamma *= 0.6; }
    else if (err_ratio > 3 && err_ratio <= 4) { sigma = 0.85; gamma *= 0.5; w->final_check = 1; }
    else if (err_ratio > 1 && err_ratio <= 3) {
      w->final_check = 1;
      if (ratio < 0.1) {
        if (w->double_check) { sigma = 0.9; gamma *= 0.4; w->double_check = 0; }
        else { sigma = 1.0; gamma *= 0.1; w->double_check = 1; }
      } else { sigma = 0.88; gamma *= 0.4; }
    } else sigma = w->sigma;
  }
  w->mu *= sigma; w->sigma = sigma; w->gamma = gamma;
}
int update_barrier_dynamic(W *w) { // LOQO, abip.c:930-977
  const bool avg = w->stgs->avg_criterion != 0;
  const double *uu = avg ? w->u_avgc.p : w->u.p, *vv = avg ? w->v_avgc.p : w->v.p;
  launch(w, ABIP_HIP_K_VEC, k_xs, w->NB, BS, uu, vv, dims(w), w->part.p);
  launch(w, ABIP_HIP_K_VEC, k_min_fold, 1, 1, (const double *)w->part.p, w->NB, w->ctl.p);
  FinArgs f; f.nslots = 1; f.slots[0] = S_XS; f.u = w->u.p; f.v = w->v.p; f.ua = w->u_avgc.p; f.va = w->v_avgc.p;
  f.gs = nullptr; // (x, tau) are replicated: every rank computes the same sum and minimum, nothing to exchange
  launch(w, ABIP_HIP_K_VEC, k_finalize, 1, 1024, f, dims(w), (const double *)w->part.p, w->NB, w->ctl.p);
  if (sync_ctl(w)) return -2;
  double xs = w->hctl->out[S_XS];
  const double minxs = w->hctl->out[S_XMIN];
  if (minxs <= 0.0) { printf("Invalid xisi < 0 \n"); return -1; } // the reference asserts here
  w->mu *= lp_loqo_sigma(xs, minxs, (long)w->n + 1, w->stgs->dynamic_sigma); // (one source with the persistent launch: lp_scalars.h)
  return 0;
}
void update_barrier_dynamic_2(W *w) { // abip.c:982-992 (reads dynamic_sigma as the exponent)
  w->mu *= std::min(w->stgs->dynamic_x * w->mu, std::pow(w->mu, w->stgs->dynamic_sigma));
}
void reinitialize_vars(W *w, int indx) { // abip.c:996-1075
  const bool avg = w->stgs->avg_criterion != 0;
  launch(w, ABIP_HIP_K_VEC, k_reinit, w->NB, BS, avg ? w->u_avgc.p : w->u.p, avg ? w->v_avgc.p : w->v.p, w->sigma, indx, dims(w));
  w->wg_valid = false; w->stats_valid = false;
}

// ------------------------------------------------------------------------------------------------
// Barzilai-Borwein penalty search (update_adapt_params, adaptive.c:34-256)
// ------------------------------------------------------------------------------------------------
int lin_projection(W *w, double *ut, const double *u, const double *v, abip_int iter) { // abip.c:552-560 on scratch vectors
  const Dims d = dims(w);
  launch(w, ABIP_HIP_K_VEC, k_dot_wg, w->NB, BS, u, v, (const double *)w->g.p, w->stgs->rho_y, d, w->part.p, w->xwt);
  if (w->dist) { enqueue_fold(w, {S_WG}); if (allreduce_scalars(w)) return -1; }
  launch(w, ABIP_HIP_K_VEC, k_rhs, w->NB, BS, u, v, ut, (const double *)w->h.p, w->stgs->rho_y, w->g_th, d, w->part.p, w->NB, (const Ctl *)w->ctl.p,
         (const double *)w->gs, (const double *)nullptr, (double2 *)nullptr);
  return kkt_solve_sync(w, ut, u, iter) < 0 ? -1 : 0; // S_DH is left for k_adapt_step
}
// The search with its decisions on the device (PCG back-end, one GPU, LPs beyond the caches -- the C4 path): round 4 read the control block
// back after every look-ahead solve (is the PCG through?) and after every look-ahead (lp_bb_beta on the host) -- ~80 round trips of ~25 us in the driver's
// 20-step window.  Here a look-ahead is ONE unit of launches -- both projections, both steps, the five inner products, k_adapt_decide (lp_bb_beta from
// lp_scalars.h, beta_prev kept in the control block), k_adapt_next -- with ONE control read per look-ahead: a solve whose PCG needs more iterations than were
// enqueued stalls the unit (halt 2; bb_stage says which solve), a look-ahead that hands its second step on is halt 5, the end of the search halt 4.  Same
// kernels and arithmetic as adaptive_search below: the same bits.
int adaptive_search_stream(W *w, abip_int iter) {
  const Dims d = dims(w);
  ABIPSettings *st = w->stgs;
  const size_t bytes = sizeof(double) * (size_t)w->LV;
  const int lookback = (int)st->adaptive_lookback;
  if (lookback <= 0) { w->beta = 0.0; return 0; } // adaptive.c:90: the loop does not run
  HIP_OK(hipMemcpyAsync(w->a_up.p, w->u.p, bytes, hipMemcpyDeviceToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(w->a_vp.p, w->v.p, bytes, hipMemcpyDeviceToDevice, w->stream));
  { // bb_prev = 1, the counters zero (one small copy: the fields sit side by side at the end of the control block)
    static_assert(offsetof(Ctl, bb_cg_skipped) + sizeof(int) - offsetof(Ctl, bb_prev) == 56, "layout of the search's fields");
    struct { double prev, beta; int act, it, stage, cg0, cg1, pad; long tot; int skip, pad2; } init = {1.0, 0.0, 0, 0, 0, 0, 0, 0, 0L, 0, 0};
    static_assert(sizeof(init) == 56, "layout of the search's fields");
    HIP_OK(hipMemcpyAsync(&w->ctl.p->bb_prev, &init, sizeof(init), hipMemcpyHostToDevice, w->stream));
    HIP_OK(hipStreamSynchronize(w->stream)); // (the source is on this stack frame)
  }
  const char *fb = getenv("ABIP_HIP_STREAM_BLIND"); // tests: a fixed (too small) count, so that look-aheads stall and are resumed
  const int bb_forced = fb ? std::max(1, atoi(fb)) : 0;
  int chunk[2] = {bb_forced ? bb_forced : next_chunk(w), bb_forced ? bb_forced : next_chunk(w)};
  static const bool bb_trace = getenv("ABIP_HIP_BB_TRACE") != nullptr; // developer: the PCG counts of every look-ahead pair on stderr
  const bool bb_reuse = !(getenv("ABIP_HIP_BB_REUSE") && atoi(getenv("ABIP_HIP_BB_REUSE")) == 0); // 0: every look-ahead solves twice, as the reference does (A / B, tests)
  // A look-ahead solve's warm start is the y block of the step before it, and with v_y = 0 that block is the previous solve's y as it stands (adaptive.c:101-104:
  // u_y = u_t,y - 0): its A'y is at hand from that solve's back-substitution, exactly as between iterations (k_post_At keeps it: W::aty_bb), and k_rhs writes
  // the set-up's gather pairs from it -- no k_cg_init_At launch.  The search's very first solve starts from u itself: there the iteration's own `aty` serves.
  const bool keep = w->aty_on && w->vy_zero && !st->half_update && w->aty_bb.p;
  const bool first_have = keep && w->aty_valid;
  auto projection = [&](double *ut, const double *u, const double *v, const double *aty) -> int { // abip.c:552-559 on scratch vectors, the PCG's first `chunk` iterations
    launch(w, ABIP_HIP_K_VEC, k_dot_wg, w->NB, BS, u, v, (const double *)w->g.p, st->rho_y, d, w->part.p, w->xwt);
    launch(w, ABIP_HIP_K_VEC, k_rhs, w->NB, BS, u, v, ut, (const double *)w->h.p, st->rho_y, w->g_th, d, w->part.p, w->NB, (const Ctl *)w->ctl.p,
           (const double *)w->gs, aty, aty ? (double2 *)w->cg_pair.p : (double2 *)nullptr);
    return enqueue_cg_begin(w, ut, u, iter, aty != nullptr);
  };
  double *const kbb = keep ? w->aty_bb.p : nullptr;
  long done = 0;
  // a unit from one of its entry points: 0 = the top, 1 = behind the first solve's PCG chunk, 2 = behind the second solve's, 3 = the second half alone (the
  // first one was handed over by the previous look-ahead: k_adapt_resume lowers halt 5 and counts the solve that is not repeated)
  auto unit = [&](int from) -> int {
    if (from == 0) { if (projection(w->a_ut.p, w->a_up.p, w->a_vp.p, (done == 0 && first_have) ? (const double *)w->aty.p : nullptr) || enqueue_cg_chunk(w, w->a_ut.p, chunk[0])) return -1; }
    if (from <= 1) {
      if (enqueue_cg_post(w, w->a_ut.p, kbb)) return -1;
      launch(w, ABIP_HIP_K_VEC, k_adapt_step, w->NB, BS, (const double *)w->a_ut.p, w->a_ut.p, (const double *)w->a_up.p, (const double *)w->a_vp.p,
             w->a_u.p, w->a_v.p, st->alpha, 0.0, d, (const double *)w->part.p, w->NB, (const double *)w->gs, w->ctl.p, w->mu, 1);
    }
    if (from == 3) launch(w, ABIP_HIP_K_VEC, k_adapt_resume, 1, 1, w->ctl.p);
    if (from <= 1 || from == 3) { if (projection(w->a_utn.p, w->a_u.p, w->a_v.p, kbb) || enqueue_cg_chunk(w, w->a_utn.p, chunk[1])) return -1; }
    if (enqueue_cg_post(w, w->a_utn.p, kbb)) return -1;
    launch(w, ABIP_HIP_K_VEC, k_adapt_step, w->NB, BS, (const double *)w->a_utn.p, w->a_utn.p, (const double *)w->a_u.p, (const double *)w->a_v.p,
           w->a_un.p, w->a_vn.p, st->alpha, 0.0, d, (const double *)w->part.p, w->NB, (const double *)w->gs, w->ctl.p, w->mu, 2);
    launch(w, ABIP_HIP_K_VEC, k_adapt_dots, w->NB, BS, (const double *)w->a_u.p, (const double *)w->a_v.p, (const double *)w->a_un.p,
           (const double *)w->a_vn.p, (const double *)w->a_vp.p, st->alpha, d, w->part.p, w->xwt);
    launch(w, ABIP_HIP_K_VEC, k_adapt_decide, 1, 1024, (const double *)w->part.p, w->NB, w->ctl.p, st->eps_cor, st->eps_pen, lookback, w->hmir[0], bb_reuse ? 1 : 0);
    if (hipEventRecord(w->mir_ev[0], w->stream) != hipSuccess) return -1;
    launch(w, ABIP_HIP_K_VEC, k_adapt_next, w->NB, BS, w->a_up.p, w->a_vp.p, w->a_u.p, w->a_v.p, w->a_ut.p, (const double *)w->a_utn.p, (const double *)w->a_un.p, (const double *)w->a_vn.p,
           w->mu, d, (const Ctl *)w->ctl.p);
    return 0;
  };
  // One unit in flight: the host reads a look-ahead's verdict before it enqueues the next one, because the verdict decides the next unit's SHAPE -- handed a
  // second step (four look-aheads of five) it has no first half at all.  (Round 5 first enqueued units two deep, whole: the ~45 launches of a first half that
  // then fell through cost ~0.1 ms, four times the ~25 us the device now idles while the verdict travels.)
  const Ctl *hm = w->hmir[0];
  int from = 0;
  for (;;) {
    if (unit(from)) return -1;
    for (;;) {
      if (hipEventSynchronize(w->mir_ev[0]) != hipSuccess) return -1;
      if (hm->halt != 2) break;
      // a solve of this unit stalled: more PCG iterations, then the rest of the unit
      ++w->stream_stalls;
      const int which = hm->bb_stage == 0 ? 0 : 1;
      HIP_OK(hipMemsetAsync(&w->ctl.p->halt, 0, sizeof(int), w->stream));
      chunk[which] = bb_forced ? bb_forced : std::max(4, chunk[which]);
      if (enqueue_cg_chunk(w, which == 0 ? w->a_ut.p : w->a_utn.p, chunk[which]) || unit(which + 1)) return -1;
    }
    ++done;
    if (bb_trace) fprintf(stderr, "[bb] k %ld look-ahead %ld: PCG %d + %d (enqueued %d + %d) act %d beta %.6g%s\n", (long)iter, done, hm->bb_cg[0], hm->bb_cg[1], chunk[0], chunk[1], hm->bb_act, hm->bb_beta, from == 3 ? "  (first half handed over)" : "");
    if (hm->halt == 4) break; // the search is over (adaptive.c:221-229, or the look-back used up)
    if ((hm->halt != 0 && hm->halt != 5) || hm->bb_it != (int)done) { fprintf(stderr, "abip_hip: streamed search: unexpected state (halt %d, look-ahead %d of %ld)\n", hm->halt, hm->bb_it, done); return -1; }
    from = hm->halt == 5 ? 3 : 0;
    // blind PCG counts of the next pair (profiles/r05h_c4_bb_trace.txt): behind an unchanged penalty the second solve takes the previous second solve's count or one
    // less; behind a changed one both take more.  A launch past convergence costs ~6 us per PCG iteration, a stalled unit a round trip + the launches behind it.
    // The first change of a search (beta_prev 1 -> ~3) nearly doubles the counts; a later one adds a fifth to a quarter.
    const int base = hm->bb_cg[1], margin = (hm->bb_act != 1) ? 1 : (done == 1 ? (3 * base) / 4 + 2 : std::max(3, (base + 3) / 4 + 1));
    for (int q = 0; q < 2; ++q) chunk[q] = bb_forced ? bb_forced : std::max(2, std::min((int)w->m_glob, base + margin));
  }
  w->beta = hm->bb_beta;
  w->tot_solves += 2 * done; w->prof.kkt_solves += 2 * done;
  w->tot_cg_its += hm->bb_cg_total; w->prof.cg_iters += hm->bb_cg_total;
  w->tot_cg_skipped += hm->bb_cg_skipped; w->prof.cg_iters_skipped += hm->bb_cg_skipped;
  w->last_cg_its = hm->bb_cg[1];
  memcpy(w->hctl, hm, sizeof(Ctl));
  w->wg_valid = false;
  return clear_halt(w);
}
int adaptive_search(W *w, abip_int iter) {
  if (stream_ok(w) && w->bb_stream_on) return adaptive_search_stream(w, iter);
  const Dims d = dims(w);
  const size_t bytes = sizeof(double) * (size_t)w->LV;
  ABIPSettings *st = w->stgs;
  double beta_prev = 1.0, beta = 0.0;
  HIP_OK(hipMemcpyAsync(w->a_up.p, w->u.p, bytes, hipMemcpyDeviceToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(w->a_vp.p, w->v.p, bytes, hipMemcpyDeviceToDevice, w->stream));
  for (abip_int it = 0; it < st->adaptive_lookback; ++it) {
    if (lin_projection(w, w->a_ut.p, w->a_up.p, w->a_vp.p, iter)) return -1;
    launch(w, ABIP_HIP_K_VEC, k_adapt_step, w->NB, BS, (const double *)w->a_ut.p, w->a_ut.p, (const double *)w->a_up.p, (const double *)w->a_vp.p,
           w->a_u.p, w->a_v.p, st->alpha, w->mu / beta_prev, d, (const double *)w->part.p, w->NB, (const double *)w->gs, (Ctl *)nullptr, 0.0, 0);
    if (lin_projection(w, w->a_utn.p, w->a_u.p, w->a_v.p, iter)) return -1;
    launch(w, ABIP_HIP_K_VEC, k_adapt_step, w->NB, BS, (const double *)w->a_utn.p, w->a_utn.p, (const double *)w->a_u.p, (const double *)w->a_v.p,
           w->a_un.p, w->a_vn.p, st->alpha, w->mu / beta_prev, d, (const double *)w->part.p, w->NB, (const double *)w->gs, (Ctl *)nullptr, 0.0, 0);
    launch(w, ABIP_HIP_K_VEC, k_adapt_dots, w->NB, BS, (const double *)w->a_u.p, (const double *)w->a_v.p, (const double *)w->a_un.p,
           (const double *)w->a_vn.p, (const double *)w->a_vp.p, st->alpha, d, w->part.p, w->xwt);
    if (w->dist) { enqueue_fold(w, {S_A0, S_A1, S_A2, S_A3, S_A4}); if (allreduce_scalars(w)) return -1; }
    FinArgs f; f.nslots = 5;
    const int sl[5] = {S_A0, S_A1, S_A2, S_A3, S_A4};
    for (int q = 0; q < 5; ++q) f.slots[q] = sl[q];
    f.u = w->u.p; f.v = w->v.p; f.ua = nullptr; f.va = nullptr; f.gs = w->gs;
    launch(w, ABIP_HIP_K_VEC, k_finalize, 1, 1024, f, d, (const double *)w->part.p, w->NB, w->ctl.p);
    if (sync_ctl(w)) return -1;
    const double utut = w->hctl->out[S_A0], utv = w->hctl->out[S_A1], uu = w->hctl->out[S_A2], vv = w->hctl->out[S_A3], uv = w->hctl->out[S_A4];
    const int act = lp_bb_beta(utut, utv, uu, vv, uv, st->eps_cor, st->eps_pen, beta_prev, beta); // adaptive.c:170-229 (lp_scalars.h)
    if (act == 0) break;
    else if (act == 1) {
      beta_prev = beta;
      HIP_OK(hipMemcpyAsync(w->a_up.p, w->a_u.p, bytes, hipMemcpyDeviceToDevice, w->stream));
      HIP_OK(hipMemcpyAsync(w->a_vp.p, w->a_v.p, sizeof(double) * (size_t)w->m, hipMemcpyDeviceToDevice, w->stream));
      launch(w, ABIP_HIP_K_VEC, k_adapt_vprev, w->NB, BS, w->a_vp.p, (const double *)w->a_up.p, w->mu / beta_prev, d);
    } else {
      HIP_OK(hipMemcpyAsync(w->a_up.p, w->a_u.p, bytes, hipMemcpyDeviceToDevice, w->stream));
      HIP_OK(hipMemcpyAsync(w->a_vp.p, w->a_v.p, bytes, hipMemcpyDeviceToDevice, w->stream));
    }
  }
  w->beta = beta;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// solution extraction (get_solution / get_info, abip.c:1296-1414) -- host side on downloaded iterates
// ------------------------------------------------------------------------------------------------
bool st_solved(abip_int s) { return s == ABIP_SOLVED || s == ABIP_SOLVED_INACCURATE; }
bool st_infeas(abip_int s) { return s == ABIP_INFEASIBLE || s == ABIP_INFEASIBLE_INACCURATE; }
bool st_unbdd(abip_int s) { return s == ABIP_UNBOUNDED || s == ABIP_UNBOUNDED_INACCURATE; }

// multi-GPU helpers off the hot path: sum a few host scalars over the ranks; assemble the full y from the row blocks
int allreduce_host(W *w, double *vals, int cnt) {
  if (!w->dist) return 0;
  DBuf<double> tmp;
  if (tmp.alloc(cnt)) return -1;
  HIP_OK(hipMemcpyAsync(tmp.p, vals, sizeof(double) * cnt, hipMemcpyHostToDevice, w->stream));
  if (allreduce_dev(w, tmp.p, cnt)) return -1;
  HIP_OK(hipMemcpyAsync(vals, tmp.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  tmp.release();
  return 0;
}
int gather_rows(W *w, const double *local /* m */, std::vector<double> &full /* m_glob */) {
  full.assign(w->m_glob, 0.0);
  std::copy(local, local + w->m, full.begin() + w->row0);
  if (!w->dist) return 0;
  DBuf<double> tmp;
  if (tmp.alloc(w->m_glob)) return -1;
  HIP_OK(hipMemcpyAsync(tmp.p, full.data(), sizeof(double) * w->m_glob, hipMemcpyHostToDevice, w->stream));
  if (allreduce_dev(w, tmp.p, w->m_glob)) return -1;
  HIP_OK(hipMemcpyAsync(full.data(), tmp.p, sizeof(double) * w->m_glob, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  tmp.release();
  return 0;
}

int ensure_stats(W *w) { // make ctl.out describe the CURRENT iterate and the averaged one (off the hot path:
                         // only after an out-of-band change of (u, v) such as the half-update clip, or for a mid-run snapshot)
  if (w->stats_valid && (!w->stgs->avg_criterion || w->avg_stats_valid)) return 0;
  std::vector<double> hu(w->LV), hua(w->LV), hb(w->m), hc(w->n);
  HIP_OK(hipMemcpyAsync(hu.data(), w->u.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipMemcpyAsync(hua.data(), w->u_avgc.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipMemcpyAsync(hb.data(), w->b.p, sizeof(double) * w->m, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipMemcpyAsync(hc.data(), w->c.p, sizeof(double) * w->n, hipMemcpyDeviceToHost, w->stream));
  // the residual SpMV pair is gated on (!halt && cg_done): force both
  const int zero = 0, one = 1;
  HIP_OK(hipMemcpyAsync(&w->ctl.p->halt, &zero, sizeof(int), hipMemcpyHostToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(&w->ctl.p->cg_done, &one, sizeof(int), hipMemcpyHostToDevice, w->stream));
  if (enqueue_q_and_finalize(w, true, false, /*decide=*/false)) return -1; // statistics only: not an ADMM iteration
  if (sync_ctl(w)) return -1;
  double *o = w->hctl->out;
  double by = 0, cx = 0, bya = 0, cxa = 0;
  for (abip_int i = 0; i < w->m; ++i) { by += hb[i] * hu[i]; bya += hb[i] * hua[i]; }
  for (abip_int j = 0; j < w->n; ++j) { cx += hc[j] * hu[w->MP + j]; cxa += hc[j] * hua[w->MP + j]; }
  { double yy[2] = {by, bya}; if (allreduce_host(w, yy, 2)) return -1; by = yy[0]; bya = yy[1]; } // y is sharded, x replicated
  o[S_BY] = by; o[S_CX] = cx; o[S_BYA] = bya; o[S_CXA] = cxa;
  w->stats_valid = true; w->avg_stats_valid = true;
  w->r.last_admm_iter = -1;
  return 0;
}

void fail_fill(W *w, ABIPInfo *info, abip_int status_val, const char *ststr) { // populate_on_failure, abip.c:219-277
  if (info) {
    info->res_pri = NAN; info->res_dual = NAN; info->rel_gap = NAN; info->res_infeas = NAN; info->res_unbdd = NAN;
    info->pobj = NAN; info->dobj = NAN; info->ipm_iter = -1; info->admm_iter = -1; info->status_val = status_val; info->solve_time = NAN;
    strcpy(info->status, ststr);
  }
  if (w) {
    w->sol_x.assign(w->n, NAN); w->sol_y.assign(w->m_glob, NAN); w->sol_s.assign(w->n, NAN);
    w->have_solution = true;
    if (info) w->last_info = *info;
  }
}

int finish_solution(W *w, ABIPInfo *info, abip_int ipm_iter, abip_int admm_iter) {
  const abip_int m = w->m, n = w->n, l = w->m_glob + n + 1;
  ABIPSettings *st = w->stgs;
  if (ensure_stats(w)) return -1;
  calc_residuals(w, ipm_iter, admm_iter);
  Resid &r = w->r;
  const bool avg = st->avg_criterion != 0;
  std::vector<double> hu(w->LV), hv(w->LV);
  HIP_OK(hipMemcpyAsync(hu.data(), avg ? w->u_avgc.p : w->u.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipMemcpyAsync(hv.data(), avg ? w->v_avgc.p : w->v.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  w->sol_x.assign(hu.begin() + w->MP, hu.begin() + w->MP + n);
  if (gather_rows(w, hu.data(), w->sol_y)) return -1;
  w->sol_s.assign(hv.begin() + w->MP, hv.begin() + w->MP + n);
  int kind; // 0 solved, 1 indeterminate, 2 infeasible, 3 unbounded
  if (info->status_val == ABIP_UNFINISHED) {
    double nrm2 = 0, ny = 0;
    for (abip_int i = 0; i < m; ++i) ny += hu[i] * hu[i];
    if (allreduce_host(w, &ny, 1)) return -1;
    nrm2 = ny;
    for (abip_int j = 0; j <= n; ++j) nrm2 += hu[w->MP + j] * hu[w->MP + j];
    if (r.tau > 1e-9 && r.tau > r.kap) kind = 0;
    else if (std::sqrt(nrm2) < 1e-9 * std::sqrt((double)l)) kind = 1;
    else if (-r.bt_y_by_tau < r.ct_x_by_tau) kind = 2;
    else kind = 3;
  } else if (st_solved(info->status_val)) kind = 0;
  else if (st_infeas(info->status_val)) kind = 2;
  else kind = 3;
  const bool inacc = (info->status_val == 0);
  auto scale = [](std::vector<double> &a, double sc) { for (double &x : a) x *= sc; };
  if (kind == 0) {
    const double sc = safediv_pos(1.0, r.tau);
    scale(w->sol_x, sc); scale(w->sol_y, sc); scale(w->sol_s, sc);
    strcpy(info->status, inacc ? "Solved/Inaccurate" : "Solved");
    info->status_val = inacc ? ABIP_SOLVED_INACCURATE : ABIP_SOLVED;
  } else if (kind == 1) {
    strcpy(info->status, "Indeterminate");
    scale(w->sol_x, NAN); scale(w->sol_y, NAN); scale(w->sol_s, NAN);
    info->status_val = ABIP_INDETERMINATE;
  } else if (kind == 2) {
    scale(w->sol_y, 1 / r.bt_y_by_tau); scale(w->sol_s, 1 / r.bt_y_by_tau); scale(w->sol_x, NAN);
    strcpy(info->status, inacc ? "Infeasible/Inaccurate" : "Infeasible");
    info->status_val = inacc ? ABIP_INFEASIBLE_INACCURATE : ABIP_INFEASIBLE;
  } else {
    scale(w->sol_x, -1 / r.ct_x_by_tau); scale(w->sol_y, NAN); scale(w->sol_s, NAN);
    strcpy(info->status, inacc ? "Unbounded/Inaccurate" : "Unbounded");
    info->status_val = inacc ? ABIP_UNBOUNDED_INACCURATE : ABIP_UNBOUNDED;
  }
  if (st->normalize) { // un_normalize_sol, normalize.c:133-158
    for (abip_int j = 0; j < n; ++j) w->sol_x[j] /= (w->E[j] * w->sc_b);
    for (abip_int i = 0; i < w->m_glob; ++i) w->sol_y[i] /= (w->D[i] * w->sc_c);
    for (abip_int j = 0; j < n; ++j) w->sol_s[j] *= w->E[j] / (w->sc_c * st->scale);
  }
  info->ipm_iter = ipm_iter + 1; info->admm_iter = admm_iter + 1; // get_info, abip.c:1296-1340
  info->res_infeas = r.res_infeas; info->res_unbdd = r.res_unbdd;
  if (st_solved(info->status_val)) {
    info->rel_gap = r.rel_gap; info->res_pri = r.res_pri; info->res_dual = r.res_dual;
    info->pobj = r.ct_x_by_tau / r.tau; info->dobj = r.bt_y_by_tau / r.tau;
  } else if (st_unbdd(info->status_val)) {
    info->rel_gap = NAN; info->res_pri = NAN; info->res_dual = NAN; info->pobj = -INFINITY; info->dobj = -INFINITY;
  } else if (st_infeas(info->status_val)) {
    info->rel_gap = NAN; info->res_pri = NAN; info->res_dual = NAN; info->pobj = INFINITY; info->dobj = INFINITY;
  }
  info->solve_time = now_ms() - w->t_solve0;
  w->have_solution = true;
  w->last_info = *info;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// printing (abip.c:159-215, 1418-1607) -- same rows and columns as the reference
// ------------------------------------------------------------------------------------------------
void print_init_header(const ABIPData *d, int linsys) {
  const ABIPSettings *s = d->stgs;
  print_line('-');
  printf("\tABIP v%s - First-Order Interior-Point Solver (MI355X-native HIP back-end)\n", abip_version());
  print_line('-');
  if (linsys == ABIP_HIP_LINSYS_INDIRECT) printf("Lin-sys: sparse-indirect (device PCG), nnz in A = %li, CG tol ~ 1/iter^(%2.2f)\n", (long)d->A->p[d->A->n], s->cg_rate);
  else printf("Lin-sys: sparse-direct (host LDL', device SpTRSV), nnz in A = %li\n", (long)d->A->p[d->A->n]);
  if (s->normalize)
    printf("eps = %.2e, alpha = %.2f, max_ipm_iters = %i, max_admm_iters = %i, normalize = %i\nscale = %2.2f, adaptive = %i, adaptive_lookback = %i, rho_y = %.2e\n",
           s->eps, s->alpha, (int)s->max_ipm_iters, (int)s->max_admm_iters, (int)s->normalize, s->scale, (int)s->adaptive, (int)s->adaptive_lookback, s->rho_y);
  else
    printf("eps = %.2e, alpha = %.2f, max_ipm_iters = %i, max_admm_iters = %i, normalize = %i\nadaptive = %i, adaptive_lookback = %i, rho_y = %.2e\n",
           s->eps, s->alpha, (int)s->max_ipm_iters, (int)s->max_admm_iters, (int)s->normalize, (int)s->adaptive, (int)s->adaptive_lookback, s->rho_y);
  printf("Variables n = %i, constraints m = %i\n", (int)d->n, (int)d->m);
}
void print_header(const W *w) {
  if (w->stgs->warm_start) printf("ABIP using variable warm-starting\n");
  print_line('-');
  for (int q = 0; q < kHeaderLen - 1; ++q) printf("%s|", kHeader[q]);
  printf("%s\n", kHeader[kHeaderLen - 1]);
  print_line('-');
}
void print_summary(const W *w, abip_int i, abip_int j) {
  const Resid &r = w->r;
  printf("%*i|", (int)strlen(kHeader[0]), (int)i);
  printf("%*i|", (int)strlen(kHeader[1]), (int)j);
  printf("%*.2e|", (int)strlen(kHeader[2]), w->mu);
  printf("%*.2e|", kHSpace, r.res_pri); printf("%*.2e|", kHSpace, r.res_dual); printf("%*.2e|", kHSpace, r.rel_gap);
  printf("%*.2e|", kHSpace, safediv_pos(r.ct_x_by_tau, r.tau)); printf("%*.2e|", kHSpace, safediv_pos(r.bt_y_by_tau, r.tau));
  printf("%*.2e|", kHSpace, safediv_pos(r.kap, r.tau));
  printf("%*.2e ", kHSpace, (now_ms() - w->t_solve0) / 1e3);
  printf("\n");
}
void print_summary_row(const W *w, const double *row) { // a row of the persistent launch's outer-iteration log (dev_xcd.h XLOG_W), print_summary's columns
  (void)w;
  printf("%*i|", (int)strlen(kHeader[0]), (int)row[0]);
  printf("%*i|", (int)strlen(kHeader[1]), (int)row[1]);
  printf("%*.2e|", (int)strlen(kHeader[2]), row[2]);
  printf("%*.2e|", kHSpace, row[3]); printf("%*.2e|", kHSpace, row[4]); printf("%*.2e|", kHSpace, row[5]);
  printf("%*.2e|", kHSpace, safediv_pos(row[6], row[8])); printf("%*.2e|", kHSpace, safediv_pos(row[7], row[8]));
  printf("%*.2e|", kHSpace, safediv_pos(row[9], row[8]));
  printf("%*.2e ", kHSpace, row[10] / 1e3);
  printf("\n");
}
void print_footer(const W *w, const ABIPInfo *info) {
  print_line('-');
  printf("Status: %s\n", info->status);
  if (info->ipm_iter + 1 == w->stgs->max_ipm_iters) printf("Hit max_ipm_iters, solution may be inaccurate\n");
  if (info->admm_iter + 1 >= w->stgs->max_admm_iters) printf("Hit max_admm_iters, solution may be inaccurate\n");
  printf("Timing: Solve time: %1.2es\n", info->solve_time / 1e3);
  if (w->linsys == ABIP_HIP_LINSYS_INDIRECT) printf("\tLin-sys: avg # CG iterations: %2.2f\n", (double)w->tot_cg_its / (info->admm_iter + 1));
  else printf("\tLin-sys: nnz in L factor: %li\n", (long)(w->ldl.lnnz + w->m + w->n));
  print_line('-');
  if (st_infeas(info->status_val)) { printf("Certificate of primal infeasibility:\n|A'y + s|_2 * |b|_2 = %.4e\n", info->res_infeas); }
  else if (st_unbdd(info->status_val)) { printf("Certificate of dual infeasibility:\n|Ax|_2 * |c|_2 = %.4e\n", info->res_unbdd); }
  else {
    printf("Error metrics:\n");
    printf("primal res: |Ax - b|_2 / (1 + |b|_2) = %.4e\n", info->res_pri);
    printf("dual res: |A'y + s - c|_2 / (1 + |c|_2) = %.4e\n", info->res_dual);
    printf("rel gap: |c'x - b'y| / (1 + |c'x| + |b'y|) = %.4e\n", info->rel_gap);
    print_line('-');
    printf("c'x = %.4e, b'y = %.4e\n", info->pobj, info->dobj);
  }
  print_line('=');
}

void free_work(W *w) {
  if (!w) return;
  w->dAt.release(); w->dA.release(); w->dAc.release(); w->dAct.release();
  { DBuf<double> *cb[] = {&w->cc_b, &w->cc_y, &w->cc_r, &w->cc_z, &w->cc_p, &w->cc_Gp, &w->cc_M, &w->cc_tmp, &w->cc_d, &w->cc_buf}; for (auto *b : cb) b->release(); }
  DBuf<double> *bufs[] = {&w->u, &w->v, &w->ut, &w->u_avg, &w->v_avg, &w->u_sum, &w->v_sum, &w->u_avgc, &w->v_avgc, &w->h, &w->g, &w->b, &w->c,
                          &w->wD, &w->wE, &w->cg_p, &w->cg_r, &w->cg_Gp, &w->cg_z, &w->cg_M, &w->cg_tmp, &w->cg_pair, &w->a_up, &w->a_vp, &w->a_ut, &w->a_u,
                          &w->a_v, &w->a_utn, &w->a_un, &w->a_vn, &w->part, &w->aty, &w->aty_bb, &w->hAh};
  for (auto *b : bufs) b->release();
  w->ctl.release(); w->ldl.release(); w->T.release(); w->xcd.release();
  if (w->hctl) (void)hipHostFree(w->hctl);
  for (int q = 0; q < 2; ++q) { if (w->hmir[q]) (void)hipHostFree(w->hmir[q]); if (w->mir_ev[q]) (void)hipEventDestroy(w->mir_ev[q]); }
  w->stamps.release();
  if (w->hstamps) (void)hipHostFree(w->hstamps);
  for (auto &e : w->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
}

int upload_lvec(W *w, DBuf<double> &dst, const double *y, const double *x, double tail) { // host [y|x|tau] pieces -> padded device layout
  std::vector<double> &s = w->scratch;
  s.assign(w->LV, 0.0);
  if (y) std::copy(y, y + w->m, s.begin());
  if (x) std::copy(x, x + w->n, s.begin() + w->MP);
  s[w->MP + w->n] = tail;
  HIP_OK(hipMemcpyAsync(dst.p, s.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  return 0;
}

} // namespace

// ==================================================================================================
// public C ABI
// ==================================================================================================
extern "C" {

const char *abip_version(void) { return ABIP_VERSION; }

void abip_hip_set_linsys(int which) { g_linsys = which ? ABIP_HIP_LINSYS_INDIRECT : ABIP_HIP_LINSYS_DIRECT; }
void abip_hip_set_copy_a_matrix(int on) { g_copy_a = on < 0 ? -1 : (on ? 1 : 0); } // < 0: back to the environment / default
int abip_hip_get_copy_a_matrix(void) { return g_copy_a; } // -1: not set by the program
int abip_hip_get_linsys(void) { return chosen_linsys(); }

int abip_hip_device_info(char *name, int name_len, long *total_mem_bytes, int *num_cu) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return -1;
  hipDeviceProp_t p;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -2;
  if (name && name_len > 0) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
  if (total_mem_bytes) *total_mem_bytes = (long)p.totalGlobalMem;
  if (num_cu) *num_cu = p.multiProcessorCount;
  return 0;
}

void abip_set_default_settings(ABIPData *d) { // util.c:288-329
  ABIPSettings *s = d->stgs;
  s->max_ipm_iters = 500; s->max_admm_iters = 1000000; s->eps = 1e-3; s->alpha = 1.8; s->cg_rate = 2.0;
  s->normalize = 1; s->scale = 1.0; s->rho_y = 1e-3; s->sparsity_ratio = 0.01;
  s->adaptive = 1; s->eps_cor = 0.2; s->eps_pen = 0.1; s->adaptive_lookback = 20;
  s->dynamic_x = 0.8; s->dynamic_eta = 1.1; s->restart_fre = 1000; s->restart_thresh = 100000;
  s->origin_rescale = 0; s->pc_ruiz_rescale = 1; s->qp_rescale = 0; s->ruiz_iter = 10;
  s->hybrid_mu = 1; s->dynamic_sigma = -1.0; s->hybrid_thresh = 1000; s->dynamic_sigma_second = 0.5;
  s->half_update = 0; s->avg_criterion = 0; s->verbose = 1; s->warm_start = 0;
}

void abip_free_data(ABIPData *d) { // util.c:229-262
  if (!d) return;
  if (d->b) free(d->b);
  if (d->c) free(d->c);
  if (d->stgs) free(d->stgs);
  if (d->A) { if (d->A->x) free(d->A->x); if (d->A->i) free(d->A->i); if (d->A->p) free(d->A->p); free(d->A); }
  free(d);
}
void abip_free_sol(ABIPSolution *sol) { // util.c:264-283
  if (!sol) return;
  if (sol->x) free(sol->x);
  if (sol->y) free(sol->y);
  if (sol->s) free(sol->s);
  free(sol);
}

ABIPWork *abip_init(const ABIPData *d, ABIPInfo *info) { // abip.c:2341-2388 + init_work 1739-1839
  if (!d || !info) { printf("ERROR: Missing ABIPData or ABIPInfo input\n"); return nullptr; }
  if (host::validate(d) < 0) { printf("ERROR: Validation returned failure\n"); return nullptr; }
  char devname[128];
  if (abip_hip_device_info(devname, sizeof(devname), nullptr, nullptr) != 0) {
    printf("ERROR: no usable HIP device: libabip_hip has no CPU fallback\n");
    return nullptr;
  }
  const double t0 = now_ms();
  W *w = new W();
  w->linsys = chosen_linsys();
  { const char *e = getenv("ABIP_HIP_BATCH"); w->batch_ok = !(e && atoi(e) == 0); }
  if (d->stgs->verbose) print_init_header(d, w->linsys);
  w->stgs = d->stgs; w->n = d->n; w->A = d->A; w->sp = d->sp;
  const bool own_copy = copy_a_matrix();
  if (own_copy) { // scaling changes the values only: the index arrays are shared read-only
    w->Aown = *d->A;
    w->Aown_x.assign(d->A->x, d->A->x + d->A->p[d->n]);
    w->Aown.x = w->Aown_x.data();
    w->A = &w->Aown;
  }
  w->m_glob = d->m; w->m = d->m; w->row0 = 0;
  const abip_int n = d->n;
  auto fail = [&](const char *msg) -> ABIPWork * { printf("ERROR: %s\n", msg); free_work(w); return nullptr; };
  if (hipStreamCreate(&w->stream) != hipSuccess) return fail("hipStreamCreate failed");
  if (w->stgs->normalize) host::normalize_A(w->A, w->stgs, w->D, w->E, &w->mean_norm_row_A, &w->mean_norm_col_A);
  // multi-GPU: the PCG back-end keeps only this rank's block of rows (the direct back-end does not shard: replicas)
  w->dist = (g_dist.kind != 0) && (w->linsys == ABIP_HIP_LINSYS_INDIRECT);
  const ABIPMatrix *Ause = w->A;
  if (w->dist) {
    w->rank = g_dist.rank; w->world = g_dist.world; w->xwt = (w->rank == 0) ? 1.0 : 0.0;
    const abip_int mg = d->m, nnz = w->A->p[n];
    if (mg < w->world) return fail("fewer rows than ranks");
    std::vector<abip_int> bounds(w->world + 1);
    if (abip_hip_dist_partition(w->A, w->world, bounds.data()) != 0) return fail("row partition failed");
    const abip_int r0 = bounds[w->rank], r1 = bounds[w->rank + 1];
    (void)nnz;
    w->row0 = r0; w->m = r1 - r0;
    w->Aloc_p.assign(n + 1, 0);
    for (abip_int j = 0; j < n; ++j) {
      abip_int c = 0;
      for (abip_int q = w->A->p[j]; q < w->A->p[j + 1]; ++q) c += (w->A->i[q] >= r0 && w->A->i[q] < r1);
      w->Aloc_p[j + 1] = w->Aloc_p[j] + c;
    }
    w->Aloc_i.resize(std::max<abip_int>(w->Aloc_p[n], 1)); w->Aloc_x.resize(std::max<abip_int>(w->Aloc_p[n], 1));
    for (abip_int j = 0, t = 0; j < n; ++j)
      for (abip_int q = w->A->p[j]; q < w->A->p[j + 1]; ++q)
        if (w->A->i[q] >= r0 && w->A->i[q] < r1) { w->Aloc_i[t] = w->A->i[q] - r0; w->Aloc_x[t] = w->A->x[q]; ++t; }
    w->Aloc.x = w->Aloc_x.data(); w->Aloc.i = w->Aloc_i.data(); w->Aloc.p = w->Aloc_p.data(); w->Aloc.m = w->m; w->Aloc.n = n;
    Ause = &w->Aloc;
    // form of the sharded solve: rows (north_star's wording -- row blocks, one all-reduce of the A'-partials + packed scalars per PCG iteration) or columns
    // (inside the solve the m-space is gathered and replicated, the exchange per PCG iteration is the m-vector -- C4: 1.6 MB against 4 MB; the iteration around
    // the solve keeps its row blocks either way).  Neither has run on two GPUs yet; bench.py --gpus N measures both in one invocation.
    // Default since round 5: the form that exchanges less -- columns whenever m < n; ABIP_HIP_DIST_CG=rows|cols names one.
    { const char *e = getenv("ABIP_HIP_DIST_CG"); w->cg_cols = e ? !strcmp(e, "cols") : (mg < n); }
    if (w->cg_cols) { // column block of the scaled matrix, balanced by non-zeros (+1 per column), and the whole Jacobi preconditioner
      if (n < w->world) return fail("fewer columns than ranks");
      std::vector<abip_int> cb(w->world + 1, 0);
      cb[w->world] = n;
      for (int g = 1; g < w->world; ++g) {
        const double target = (double)(nnz + n) * g / w->world;
        abip_int lo = cb[g - 1] + 1, hi = n - (w->world - g);
        while (lo < hi) { const abip_int mid = (lo + hi) / 2; if ((double)(w->A->p[mid] + mid) < target) lo = mid + 1; else hi = mid; }
        cb[g] = lo;
      }
      w->col0 = cb[w->rank]; w->ncol = cb[w->rank + 1] - cb[w->rank];
      std::vector<abip_int> cp(w->ncol + 1);
      const abip_int base = w->A->p[w->col0];
      for (abip_int j = 0; j <= w->ncol; ++j) cp[j] = w->A->p[w->col0 + j] - base;
      ABIPMatrix Ac{}; Ac.x = w->A->x + base; Ac.i = w->A->i + base; Ac.p = cp.data(); Ac.m = mg; Ac.n = w->ncol;
      host::HostCsr hct, hc;
      host::csc_as_csr(&Ac, hct); host::build_row_blocks(hct, CHUNK);
      host::transpose_to_csr(&Ac, hc); host::build_row_blocks(hc, CHUNK);
      std::vector<double> Mfull;
      host::jacobi_preconditioner(w->A, Mfull);
      if (w->dAct.upload(hct, w->stream) || w->dAc.upload(hc, w->stream) || w->cc_M.upload(Mfull, w->stream) || w->cc_b.alloc(mg) || w->cc_y.alloc(mg) || w->cc_r.alloc(mg) ||
          w->cc_z.alloc(mg) || w->cc_p.alloc(mg) || w->cc_Gp.alloc(mg) || w->cc_tmp.alloc(w->ncol) || w->cc_d.alloc(w->ncol) || w->cc_buf.alloc(2 * (size_t)mg))
        return fail("device allocation failure (column block)");
      if (hipMemsetAsync(w->cc_tmp.p, 0, sizeof(double) * std::max<abip_int>(w->ncol, 1), w->stream) != hipSuccess) return fail("memset failure");
    }
    w->n_pad = ((size_t)n + 31) / 32 * 32; // scalars start 256-byte aligned behind the n-vector
    if (w->T.alloc(w->n_pad + S_COUNT)) return fail("work memory allocation failure");
    if (hipMemsetAsync(w->T.p, 0, sizeof(double) * (w->n_pad + S_COUNT), w->stream) != hipSuccess) return fail("memset failure");
    w->gs = w->T.p + w->n_pad;
  }
  const abip_int m = w->m; // rows on this device
  w->MP = (int)(((m + 31) / 32) * 32);
  w->LV = w->MP + (int)n + 1;
  w->LV = ((w->LV + 31) / 32) * 32;
  w->NB = 1; // fixed below, once the row blocks are known
  // device images of the (scaled) matrix
  host::HostCsr hAt, hA;
  host::csc_as_csr(Ause, hAt); host::build_row_blocks(hAt, CHUNK); host::build_sell(hAt);
  host::transpose_to_csr(Ause, hA); host::build_row_blocks(hA, CHUNK); host::build_sell(hA);
  if (w->dAt.upload(hAt, w->stream) || w->dA.upload(hA, w->stream)) return fail("device allocation failure (matrix)");
  { // persistent grid: every kernel uses the same NB (== partials per slot).  Sized so that the SpMV kernels give each
    // workgroup the same whole number of row blocks (no tail), at most MAXNB (8 workgroups per CU on 256 CUs).
    long nrb = std::max<long>(std::max<long>(w->dAt.nrb, w->dA.nrb), (std::max<long>(m, n) + 4 * BS - 1) / (4 * BS));
    // sharded: NB also fixes the order in which the REPLICATED n-space reductions (||A'p||^2, the LOQO sum and minimum) add up,
    // and those must come out bit-identical on every rank (they steer alpha, mu and with them the host's control flow): derive
    // it from global quantities only, never from this rank's row block
    if (w->dist) nrb = std::max<long>(((long)w->A->p[n] / w->world + CHUNK - 1) / CHUNK + 1, (std::max<long>((long)w->m_glob / w->world + 1, n) + 4 * BS - 1) / (4 * BS));
    const long per = (nrb + MAXNB - 1) / MAXNB;
    w->NB = (int)std::max<long>(1, (nrb + per - 1) / per);
  }
  DBuf<double> *lvecs[] = {&w->u, &w->v, &w->ut, &w->u_avg, &w->v_avg, &w->u_sum, &w->v_sum, &w->u_avgc, &w->v_avgc, &w->h, &w->g,
                           &w->a_up, &w->a_vp, &w->a_ut, &w->a_u, &w->a_v, &w->a_utn, &w->a_un, &w->a_vn};
  for (auto *b : lvecs) {
    if (b->alloc(w->LV)) return fail("work memory allocation failure");
    if (hipMemsetAsync(b->p, 0, sizeof(double) * w->LV, w->stream) != hipSuccess) return fail("memset failure");
  }
  if (w->b.alloc(m) || w->c.alloc(n) || w->wD.alloc(m) || w->wE.alloc(n) || w->part.alloc((size_t)S_COUNT * MAXNB) || w->ctl.alloc(1))
    return fail("work memory allocation failure");
  if (hipMemsetAsync(w->part.p, 0, sizeof(double) * S_COUNT * MAXNB, w->stream) != hipSuccess) return fail("memset failure");
  if (hipMemsetAsync(w->ctl.p, 0, sizeof(Ctl), w->stream) != hipSuccess) return fail("memset failure");
  if (hipHostMalloc((void **)&w->hctl, sizeof(Ctl), hipHostMallocDefault) != hipSuccess) return fail("pinned allocation failure");
  memset(w->hctl, 0, sizeof(Ctl));
  if (w->linsys == ABIP_HIP_LINSYS_INDIRECT) { // init_lin_sys_work, indirect.c:282-318
    std::vector<double> Minv;
    host::jacobi_preconditioner(Ause, Minv);
    if (w->cg_M.upload(Minv, w->stream) || w->cg_p.alloc(m) || w->cg_r.alloc(m) || w->cg_Gp.alloc(m) || w->cg_z.alloc(m) || w->cg_tmp.alloc(n) || w->cg_pair.alloc(2 * (size_t)n))
      return fail("init_lin_sys_work failure");
    { const char *e = getenv("ABIP_HIP_ATY"); w->aty_on = !(e && atoi(e) == 0); } // ABIP_HIP_ATY=0: every product formed where the reference forms it (A / B, tests)
    if (!w->dist && w->aty_on && (w->aty.alloc(n) || w->aty_bb.alloc(n))) return fail("init_lin_sys_work failure");
    { const char *e = getenv("ABIP_HIP_STREAM"); w->stream_on = !(e && atoi(e) == 0); } // ABIP_HIP_STREAM=0: one control read per iteration, as in round 4
    { const char *e = getenv("ABIP_HIP_STREAM_BB"); w->bb_stream_on = !(e && atoi(e) == 0); } // ABIP_HIP_STREAM_BB=0: the Barzilai-Borwein search driven by the host, as in round 4
    if (!w->dist && w->stream_on) {
      for (int q = 0; q < 2; ++q) {
        if (hipHostMalloc((void **)&w->hmir[q], sizeof(Ctl), hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&w->mir_ev[q], hipEventDisableTiming) != hipSuccess)
          return fail("pinned allocation failure");
        memset(w->hmir[q], 0, sizeof(Ctl));
      }
    }
    if (hipMemsetAsync(w->cg_tmp.p, 0, sizeof(double) * n, w->stream) != hipSuccess) return fail("memset failure");
  } else { // init_lin_sys_work / factorize, direct.c:218-303
    host::LdlHost F;
    std::vector<int> Kp, Ki;
    std::vector<double> Kx;
    host::kkt_upper(w->A, w->stgs->rho_y, Kp, Ki, Kx);
    const int N = (int)(m + n);
    std::vector<int> pmap(N);
    auto set_up = [&](int tail_request) -> int { // factor on the host, upload, build the dense tail on the device; < 0: cannot
      if (tail_request != -2) host::set_tail_request(tail_request);
      const int rc = host::factor_upper(N, Kp, Ki, Kx, F);
      host::set_tail_request(-2);
      if (rc < 0) return -2;
      for (int q = 0; q < N; ++q) pmap[q] = F.P[q] < (int)m ? F.P[q] : w->MP + (F.P[q] - (int)m);
      return w->ldl.setup(F, pmap, w->stream) ? -1 : 0;
    };
    // set-up guard: solve one known right-hand side with the factor as the hot loop will and check ||K z - rhs|| on the host
    auto residual = [&]() -> double {
      std::vector<double> rhs, lv(w->LV, 0.0), z(N);
      host::guard_rhs(N, rhs);
      for (int i = 0; i < N; ++i) lv[i < (int)m ? i : w->MP + (i - (int)m)] = rhs[i];
      if (hipMemcpyAsync(w->a_ut.p, lv.data(), sizeof(double) * w->LV, hipMemcpyHostToDevice, w->stream) != hipSuccess) return 1e300;
      w->ldl.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { launch_lds(w, ABIP_HIP_K_SPTRSV, kern, grid, block, lds, a...); }, w->a_ut.p, (const Ctl *)w->ctl.p, w->NB);
      if (hipMemcpyAsync(lv.data(), w->a_ut.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) return 1e300;
      for (int i = 0; i < N; ++i) z[i] = lv[i < (int)m ? i : w->MP + (i - (int)m)];
#ifdef ABIP_HIP_TEST_HOOKS
      if (getenv("ABIP_HIP_TAIL_RESID_FAIL") && F.T > 0) return 1.0; // pretend the tail is too ill-conditioned for inv(L22)
#endif
      return host::sym_upper_residual(N, Kp, Ki, Kx, z, rhs);
    };
    constexpr double kGuardTol = 1e-8; // healthy factors sit at 1e-16 ... 1e-11
    int rc = set_up(-2);
    double res = rc == 0 ? residual() : 1e300;
    if (rc == -2 || (rc != 0 && F.T == 0)) return fail("init_lin_sys_work failure");
    if (rc != 0 || res > kGuardTol) {
      // the dense tail could not be set up (no room for its two T x T triangles, a pivot the un-pivoted dense LDL' cannot take) or its
      // explicit inverse lost the solve's accuracy: fall back to the plain level-scheduled factor
      if (F.T == 0) return fail("init_lin_sys_work failure (KKT solve residual above 1e-8)");
      if (w->stgs->verbose) printf("dense tail rejected (T = %d, set-up residual %.2e): using the level-scheduled factor\n", F.T, res);
      (void)hipGetLastError();
      w->ldl.release();
      if (set_up(0) != 0) return fail("init_lin_sys_work failure");
      res = residual();
      if (res > kGuardTol) return fail("init_lin_sys_work failure (KKT solve residual above 1e-8)");
    }
    w->factor_resid = res;
    { const char *e = getenv("ABIP_HIP_FUSE"); w->fuse_small = w->ldl.small && !w->dist && !(e && atoi(e) == 0) && (!w->ldl.xl || w->ldl.allow_lds<LpSolveFuse>()); }
    const int one = 1; // the post-solve kernels are gated on cg_done: permanently set for the direct back-end
    if (hipMemcpyAsync(&w->ctl.p->cg_done, &one, sizeof(int), hipMemcpyHostToDevice, w->stream) != hipSuccess) return fail("memcpy failure");
  }
  xcd_setup(w, hA, hAt);
  if (hipStreamSynchronize(w->stream) != hipSuccess) return fail("device set-up failed");
  if (own_copy) { w->A = nullptr; std::vector<double>().swap(w->Aown_x); } // everything lives on the device now
  info->setup_time = now_ms() - t0;
  if (d->stgs->verbose) printf("Setup time: %1.2es\n", info->setup_time / 1e3);
  return w;
}

abip_int abip_hip_solve_begin(ABIPWork *w, const ABIPData *d, const ABIPSolution *sol, ABIPInfo *info) { // update_work, abip.c:1843-1927
  if (!w || !d || !info || !d->b || !d->c) { printf("ERROR: ABIP_NULL input\n"); return ABIP_FAILED; }
  ABIPSettings *st = w->stgs;
  const abip_int m = w->m, n = w->n, mg = w->m_glob, r0 = w->row0; // m: rows on this device, mg: rows of the problem
  w->t_solve0 = now_ms(); w->cpu0 = (double)clock();
  info->status_val = ABIP_UNFINISHED; w->status = 0;
  w->r = Resid();
  std::vector<double> hb(d->b, d->b + mg), hc(d->c, d->c + n);
  auto nrm = [](const std::vector<double> &v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); };
  w->nm_b = nrm(hb); w->nm_c = nrm(hc);
  if (st->normalize) { // normalize_b_c, normalize.c:11-40
    for (abip_int j = 0; j < n; ++j) hc[j] /= w->E[j];
    w->sc_c = w->mean_norm_row_A / std::max(nrm(hc), 1e-3);
    for (abip_int i = 0; i < mg; ++i) hb[i] /= w->D[i];
    w->sc_b = w->mean_norm_col_A / std::max(nrm(hb), 1e-3);
    for (abip_int j = 0; j < n; ++j) hc[j] *= w->sc_c * st->scale;
    for (abip_int i = 0; i < mg; ++i) hb[i] *= w->sc_b * st->scale;
  } else { w->sc_b = 1; w->sc_c = 1; }
  const double mx = std::max(w->sp, st->sparsity_ratio), mn = std::min(w->sp, st->sparsity_ratio); // abip.c:1886-1900
  if (mx > 0.4 || (mn > 0.1 && mn < 0.2)) { w->sigma = 0.3; w->gamma = 2.0; }
  else if (mn > 0.2) { w->sigma = 0.5; w->gamma = 3.0; }
  else { w->sigma = 0.8; w->gamma = 3.0; }
  w->final_check = 0; w->double_check = 0; w->mu = 1.0; w->beta = 1.0;
  HIP_OK(hipMemcpyAsync(w->b.p, hb.data() + r0, sizeof(double) * m, hipMemcpyHostToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(w->c.p, hc.data(), sizeof(double) * n, hipMemcpyHostToDevice, w->stream));
  if (st->normalize) { // weights of the un-scaled residual norms, abip.c:409,445
    std::vector<double> wD(m), wE(n);
    for (abip_int i = 0; i < m; ++i) wD[i] = w->D[r0 + i] / (w->sc_b * st->scale);
    for (abip_int j = 0; j < n; ++j) wE[j] = w->E[j] / (w->sc_c * st->scale);
    HIP_OK(hipMemcpyAsync(w->wD.p, wD.data(), sizeof(double) * m, hipMemcpyHostToDevice, w->stream));
    HIP_OK(hipMemcpyAsync(w->wE.p, wE.data(), sizeof(double) * n, hipMemcpyHostToDevice, w->stream));
    HIP_OK(hipStreamSynchronize(w->stream));
  }
  // start point
  {
    std::vector<double> uy(m, 0.0), ux(n), vy(m, 0.0), vx(n);
    double ut = std::sqrt(w->mu / w->beta), vt = ut;
    w->vy_zero = !st->half_update;
    if (st->warm_start && sol && sol->x && sol->y && sol->s) { // warm_start_vars, abip.c:307-357 (quirk kept: the loop overwrites the guess)
      w->vy_zero = false;
      for (abip_int i = 0; i < m; ++i) { const double y = sol->y[r0 + i]; uy[i] = (y != y) ? 0.0 : std::sqrt(w->mu / w->beta); vy[i] = std::sqrt(w->mu / w->beta); }
      for (abip_int j = 0; j < n; ++j) { ux[j] = std::sqrt(w->mu / w->beta); const double s = sol->s[j]; vx[j] = (s != s) ? 0.0 : std::sqrt(w->mu / w->beta); }
      ut = std::sqrt(w->mu / w->beta); vt = std::sqrt(w->mu / w->beta);
      if (st->normalize) { // normalize_warm_start, normalize.c:101-128
        for (abip_int j = 0; j < n; ++j) ux[j] *= (w->E[j] * w->sc_b);
        for (abip_int i = 0; i < m; ++i) uy[i] *= (w->D[r0 + i] * w->sc_c);
        for (abip_int j = 0; j < n; ++j) vx[j] /= (w->E[j] / (w->sc_c * st->scale));
      }
    } else { // cold_start_vars, abip.c:361-381
      std::fill(ux.begin(), ux.end(), std::sqrt(w->mu / w->beta));
      std::fill(vx.begin(), vx.end(), std::sqrt(w->mu / w->beta));
    }
    if (upload_lvec(w, w->u, uy.data(), ux.data(), ut) || upload_lvec(w, w->v, vy.data(), vx.data(), vt)) return ABIP_FAILED;
  }
  // h = (-b, c); g = K^-1 h with the x block negated; g_th = h'g   (abip.c:1917-1924)
  {
    std::vector<double> hy(m);
    for (abip_int i = 0; i < m; ++i) hy[i] = -hb[r0 + i];
    if (upload_lvec(w, w->h, hy.data(), hc.data(), 0.0) || upload_lvec(w, w->g, hy.data(), hc.data(), 0.0)) return ABIP_FAILED;
    if (w->xcd.on) { // h_y + A h_x for the persistent launch (before its first launch: the set-up solve below)
      HIP_OK(hipMemcpyAsync(w->hAh.p, w->h.p, sizeof(double) * m, hipMemcpyDeviceToDevice, w->stream));
      launch(w, ABIP_HIP_K_SPMV_A, PICK(k_spmv_acc, w->dA), w->NB, BS, w->dA.view(), (const double *)(w->h.p + w->MP), w->hAh.p);
    }
    launch(w, ABIP_HIP_K_VEC, k_norm_y, w->NB, BS, (const double *)w->g.p, dims(w), w->part.p);
    if (kkt_solve_sync(w, w->g.p, nullptr, -1) < 0) return ABIP_FAILED;
    launch(w, ABIP_HIP_K_VEC, k_neg_x, w->NB, BS, w->g.p, dims(w));
    launch(w, ABIP_HIP_K_VEC, k_dot_full, w->NB, BS, (const double *)w->h.p, (const double *)w->g.p, dims(w), (int)S_T0, w->part.p, w->xwt);
    if (w->dist) { enqueue_fold(w, {S_T0}); if (allreduce_scalars(w)) return ABIP_FAILED; }
    FinArgs f; f.nslots = 1; f.slots[0] = S_T0; f.u = w->u.p; f.v = w->v.p; f.ua = nullptr; f.va = nullptr; f.gs = w->gs;
    launch(w, ABIP_HIP_K_VEC, k_finalize, 1, 1024, f, dims(w), (const double *)w->part.p, w->NB, w->ctl.p);
    if (sync_ctl(w)) return ABIP_FAILED;
    w->g_th = w->hctl->out[S_T0];
  }
  w->i = 0; w->j = 0; w->k = 0; w->phase = PH_OUTER_BEGIN; w->wg_valid = false; w->stats_valid = false; w->have_solution = false; w->aty_valid = false;
  w->tot_cg_its = 0; w->tot_solves = 0; w->tot_cg_skipped = 0; w->last_cg_its = 6;
  if (st->verbose) print_header(w);
  return 0;
}

abip_int abip_hip_step(ABIPWork *w, abip_int max_admm_steps, abip_int *steps_done, ABIPInfo *info) { // the loops of abip.c:2102-2294
  abip_int steps = 0;
  if (steps_done) *steps_done = 0;
  if (!w || !info || w->phase == PH_IDLE) return 1;
  ABIPSettings *st = w->stgs;
  const size_t lbytes = sizeof(double) * (size_t)w->LV;
  auto done = [&](abip_int rc) { if (steps_done) *steps_done = steps; return rc; };
  auto hard_fail = [&](const char *msg) { // failure(), abip.c:282-303
    fail_fill(w, info, ABIP_FAILED, "Failure");
    printf("Failure:%s\n", msg);
    if (w->dist) dist_abort(msg);
    w->phase = PH_DONE;
    return done(1);
  };
  for (;;) {
    switch (w->phase) {
      case PH_DONE: return done(1);
      case PH_IDLE: return done(1);
      case PH_OUTER_BEGIN: {
        if (w->i >= st->max_ipm_iters) { info->status_val = w->status; w->phase = PH_DONE; return done(1); } // abip.c:2296
        const double mn = std::min(w->sp, st->sparsity_ratio); // abip.c:2104-2115
        if (mn > 0.5) w->inner_stopper = (int)std::round(std::pow(w->mu, -0.35));
        else if (mn > 0.2) w->inner_stopper = (int)std::round(std::pow(w->mu, -1));
        else w->inner_stopper = st->max_admm_iters;
        w->fre_old = 0;
        if (hipMemsetAsync(w->u_avg.p, 0, lbytes, w->stream) != hipSuccess || hipMemsetAsync(w->v_avg.p, 0, lbytes, w->stream) != hipSuccess ||
            hipMemsetAsync(w->u_sum.p, 0, lbytes, w->stream) != hipSuccess || hipMemsetAsync(w->v_sum.p, 0, lbytes, w->stream) != hipSuccess)
          return hard_fail("device memset");
        if (st->avg_criterion) { // abip.c:2125-2129
          if (hipMemcpyAsync(w->u.p, w->u_avgc.p, lbytes, hipMemcpyDeviceToDevice, w->stream) != hipSuccess ||
              hipMemcpyAsync(w->v.p, w->v_avgc.p, lbytes, hipMemcpyDeviceToDevice, w->stream) != hipSuccess)
            return hard_fail("device copy");
          w->wg_valid = false; w->aty_valid = false;
        }
        w->j = 0;
        w->batch = 4;
        w->phase = PH_INNER;
        break;
      }
      case PH_INNER: {
        if (w->j >= w->inner_stopper) { w->phase = PH_OUTER_END; break; }
        if (steps >= max_admm_steps) return done(0);
        double metric = 0;
        if (xcd_outer_ok(w) && !restart_due(w, w->k, w->j)) { // cache-resident LP: the loop goes on INSIDE one persistent launch, across outer iterations, until the host is needed
          long ran = 0; int why = 0;
          w->aty_valid = false;
          const int rc = xcd_run(w, 0, (long)(max_admm_steps - steps), &ran, &why);
          if (rc < 0) return hard_fail("error in project_lin_sys");
          if (rc > 0) break; // abandoned (nothing ran, the iterate is as it was): the launch path goes on from here
          steps += (abip_int)ran;
          if (w->hctl->halt && clear_halt(w)) return hard_fail("device memset");
          if (why == XR_HOST_OUTER || ((why == XR_SLICE || why == XR_STEPS) && ran == 0 && w->hctl->xo.outer_done == 0 && w->phase == PH_OUTER_END)) w->host_outer_once = true;
          if (why == XR_FINAL) { // abip.c:2190-2213 for the last iteration that ran (the device found the earlier ones unconverged)
            calc_residuals(w, w->i, w->k);
            if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters || w->i + 1 >= st->max_ipm_iters) {
              if (st->verbose && w->k > 0) print_summary(w, w->i, w->k);
              if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
              if (st->verbose) print_footer(w, info);
              w->phase = PH_DONE;
              return done(1);
            }
          }
          break;
        }
        if (w->xcd.on) { // cache-resident LP: iterations (k, j) ... as ONE persistent launch on one XCD, up to the next decision the host has to take
          long nb = std::min<long>({(long)w->xcd.max_batch, (long)(max_admm_steps - steps), (long)(w->inner_stopper - w->j)});
          if (!w->batch_ok) nb = std::min<long>(nb, 1);
          if (w->final_check && w->i + 1 >= st->max_ipm_iters) nb = std::min<long>(nb, 1);
          for (long q = 0; q < nb; ++q) if (restart_due(w, w->k + q, w->j + q)) { nb = q; break; }
          if (nb >= 1) {
            int ran = 0;
            w->aty_valid = false;
            const int rcb = xcd_batch(w, (int)nb, &ran, &metric);
            if (rcb < 0) return hard_fail("error in project_lin_sys");
            if (rcb > 0) break; // abandoned: the launch path goes on from the same iterate
            steps += ran; w->k += ran;
            const int why = w->hctl->halt;
            if (why && clear_halt(w)) return hard_fail("device memset");
            if (why == 1) { // the exit test held at the last iteration that ran (abip.c:2173-2188)
              if (st->half_update) { launch(w, ABIP_HIP_K_VEC, k_clip_v, w->NB, BS, w->v.p, dims(w)); w->stats_valid = false; }
              w->j += ran - 1;
              w->phase = PH_OUTER_END;
              break;
            }
            if (w->final_check) { // abip.c:2190-2213 for the last iteration that ran (the device found the earlier ones unconverged)
              calc_residuals(w, w->i, w->k);
              if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters || w->i + 1 >= st->max_ipm_iters) {
                if (st->verbose && w->k > 0) print_summary(w, w->i, w->k);
                if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
                if (st->verbose) print_footer(w, info);
                w->phase = PH_DONE;
                return done(1);
              }
            }
            w->j += ran;
            break;
          }
        }
        if (w->linsys == ABIP_HIP_LINSYS_DIRECT && !w->dist && !w->final_check && !st->half_update && w->batch_ok) {
          // no host decision is needed between these iterations: run them as one batch (the device finds the exit, see admm_batch_direct)
          long nb = std::min<long>({(long)w->batch, (long)(max_admm_steps - steps), (long)(w->inner_stopper - w->j)});
          for (long q = 0; q < nb; ++q) if (restart_due(w, w->k + q, w->j + q)) { nb = q; break; }
          if (nb >= 2) {
            int ran = 0;
            if (admm_batch_direct(w, (int)nb, &ran, &metric)) return hard_fail("error in project_lin_sys");
            steps += ran; w->k += ran;
            if (w->hctl->halt) { // the exit test held at the last iteration that ran
              if (clear_halt(w)) return hard_fail("device memset");
              w->j += ran - 1;
              w->phase = PH_OUTER_END;
            } else {
              w->j += ran;
              w->batch = std::min(32, w->batch * 2);
            }
            break;
          }
        }
        if (stream_ok(w) && !restart_due(w, w->k, w->j)) { // PCG on one GPU beyond the caches: iterations streamed, the host one verdict behind the device
          long nb = std::min<long>({(long)(max_admm_steps - steps), (long)(w->inner_stopper - w->j), 1L << 20});
          for (long q = 1; q < nb; ++q) if (restart_due(w, w->k + q, w->j + q)) { nb = q; break; }
          long ran = 0; int why = 0;
          if (admm_stream_pcg(w, nb, &ran, &metric, &why) || ran < 1) return hard_fail("error in project_lin_sys");
          steps += (abip_int)ran; w->k += (abip_int)ran;
          if (why == 1) { w->j += (abip_int)ran - 1; w->phase = PH_OUTER_END; break; } // abip.c:2173-2188: the exit test held at the last iteration (j is not advanced)
          if (why == 3 || w->final_check) { // abip.c:2190-2213 for the last iteration that ran (the device found the earlier ones unconverged)
            calc_residuals(w, w->i, w->k);
            if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters || w->i + 1 >= st->max_ipm_iters) {
              if (st->verbose && w->k > 0) print_summary(w, w->i, w->k);
              if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
              if (st->verbose) print_footer(w, info);
              w->phase = PH_DONE;
              return done(1);
            }
          }
          w->j += (abip_int)ran;
          break;
        }
        if (admm_iteration(w, &metric)) return hard_fail("error in project_lin_sys");
        ++steps;
        w->k += 1;
        if (metric < w->gamma * w->mu) { // abip.c:2173-2188
          if (st->half_update) { launch(w, ABIP_HIP_K_VEC, k_clip_v, w->NB, BS, w->v.p, dims(w)); w->wg_valid = false; w->stats_valid = false; }
          w->phase = PH_OUTER_END;
          break;
        }
        if (w->final_check) { // abip.c:2190-2213
          calc_residuals(w, w->i, w->k);
          if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters || w->i + 1 >= st->max_ipm_iters) {
            if (st->verbose && w->k > 0) print_summary(w, w->i, w->k);
            if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
            if (st->verbose) print_footer(w, info);
            w->phase = PH_DONE;
            return done(1);
          }
        }
        w->j += 1;
        break;
      }
      case PH_OUTER_END: {
        if (steps > 0 && steps >= max_admm_steps) return done(0); // hand back right after the last requested iteration
        const double elapsed = ((double)clock() - w->cpu0) / CLOCKS_PER_SEC; // abip.c:2217-2221
        if (!w->host_outer_once && xcd_outer_ok(w) && w->stats_valid && (!st->avg_criterion || w->avg_stats_valid) && elapsed <= st->max_time) { // the outer end, and what follows it, inside the persistent launch
          long ran = 0; int why = 0;
          w->aty_valid = false;
          const int rc = xcd_run(w, 1, (long)(max_admm_steps - steps), &ran, &why);
          if (rc < 0) return hard_fail("error in project_lin_sys");
          if (rc == 0) {
            steps += (abip_int)ran;
            if (w->hctl->halt && clear_halt(w)) return hard_fail("device memset");
            if (why == XR_HOST_OUTER || (ran == 0 && w->hctl->xo.outer_done == 0 && w->phase == PH_OUTER_END)) w->host_outer_once = true;
            if (why == XR_FINAL) {
              calc_residuals(w, w->i, w->k);
              if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters || w->i + 1 >= st->max_ipm_iters) {
                if (st->verbose && w->k > 0) print_summary(w, w->i, w->k);
                if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
                if (st->verbose) print_footer(w, info);
                w->phase = PH_DONE;
                return done(1);
              }
            }
            break;
          } // (> 0: abandoned before anything was applied -- the host's outer end below)
        }
        w->host_outer_once = false;
        double over = elapsed > st->max_time ? 1.0 : 0.0;
        if (w->dist && allreduce_host(w, &over, 1)) return hard_fail("collective failure"); // every rank must take the same branch
        if (over > 0) { printf("Timelimit reached. \n"); st->max_admm_iters = (abip_int)(w->k * 1.05); }
        if (w->mu < st->eps) w->final_check = 1;
        if (!w->stats_valid) { if (ensure_stats(w)) return hard_fail("device error in calc_residuals"); }
        calc_residuals(w, w->i, w->k);
        if (st->verbose) print_summary(w, w->i, w->k);
        if ((info->status_val = has_converged(w, w->i, w->k)) != 0 || w->k + 1 >= st->max_admm_iters) {
          if (finish_solution(w, info, w->i, w->k)) return hard_fail("device error in get_solution");
          if (st->verbose) print_footer(w, info);
          w->phase = PH_DONE;
          return done(1);
        }
        w->status = info->status_val;
        int rc = 0;
        switch (lp_mu_rule((int)st->hybrid_mu, st->dynamic_sigma_second, st->hybrid_thresh, st->eps, w->mu, st->dynamic_sigma)) { // abip.c:2251-2277 (lp_scalars.h)
          case LP_MU_LOQO: rc = update_barrier_dynamic(w); break;
          case LP_MU_TABLE: update_barrier(w); break;
          case LP_MU_DYN2: update_barrier_dynamic_2(w); break;
          default: break;
        }
        if (rc) return hard_fail("invalid complementarity products in the LOQO barrier update");
        reinitialize_vars(w, 0);
        if (st->adaptive) { // abip.c:2281-2293
          reinitialize_vars(w, 1);
          w->beta = 1;
          if (adaptive_search(w, w->k) < 0) return hard_fail("error in adaptive");
          reinitialize_vars(w, 2);
        }
        w->i += 1;
        w->phase = PH_OUTER_BEGIN;
        break;
      }
    }
  }
}

abip_int abip_hip_solve_end(ABIPWork *w, ABIPSolution *sol, ABIPInfo *info) {
  if (!w || !sol || !info) return ABIP_FAILED;
  if (!w->have_solution) { // called before termination: extract the current iterate the way get_solution would
    ABIPInfo tmp = *info;
    tmp.status_val = ABIP_UNFINISHED;
    if (finish_solution(w, &tmp, w->i, w->k)) return ABIP_FAILED;
    *info = tmp;
    w->have_solution = false; // a later, real termination overwrites this snapshot
  } else {
    const double setup = info->setup_time;
    *info = w->last_info;
    info->setup_time = setup;
  }
  if (!sol->x) sol->x = (abip_float *)malloc(sizeof(abip_float) * w->n);
  if (!sol->y) sol->y = (abip_float *)malloc(sizeof(abip_float) * w->m_glob);
  if (!sol->s) sol->s = (abip_float *)malloc(sizeof(abip_float) * w->n);
  if (!sol->x || !sol->y || !sol->s) return ABIP_FAILED;
  memcpy(sol->x, w->sol_x.data(), sizeof(double) * w->n);
  memcpy(sol->y, w->sol_y.data(), sizeof(double) * w->m_glob);
  memcpy(sol->s, w->sol_s.data(), sizeof(double) * w->n);
  return info->status_val;
}

abip_int abip_solve(ABIPWork *w, const ABIPData *d, ABIPSolution *sol, ABIPInfo *info) { // abip.c:2056-2297
  if (!d || !sol || !info || !w || !d->b || !d->c) { printf("ERROR: ABIP_NULL input\n"); return ABIP_FAILED; }
  const double setup = info->setup_time;
  if (abip_hip_solve_begin(w, d, sol, info) != 0) {
    fail_fill(w, info, ABIP_FAILED, "Failure");
    printf("Failure:%s\n", "error in update_work");
    abip_hip_solve_end(w, sol, info);
    return ABIP_FAILED;
  }
  while (!abip_hip_step(w, std::numeric_limits<abip_int>::max() / 2, nullptr, info)) {}
  if (w->have_solution) abip_hip_solve_end(w, sol, info);
  info->setup_time = setup;
  return info->status_val;
}

void abip_finish(ABIPWork *w) { // abip.c:2301-2337
  if (!w) return;
  if (w->stgs && w->stgs->normalize && w->A) host::un_normalize_A(w->A, w->stgs, w->D, w->E);
  free_work(w);
}

abip_int abip_main(const ABIPData *d, ABIPSolution *sol, ABIPInfo *info) { // abip.c:2393-2422
  abip_int status;
  ABIPWork *w = abip_init(d, info);
  if (w) { abip_solve(w, d, sol, info); status = info->status_val; }
  else {
    status = ABIP_FAILED;
    if (info) {
      info->res_pri = NAN; info->res_dual = NAN; info->rel_gap = NAN; info->res_infeas = NAN; info->res_unbdd = NAN; info->pobj = NAN; info->dobj = NAN;
      info->ipm_iter = -1; info->admm_iter = -1; info->status_val = status; info->solve_time = NAN; strcpy(info->status, "Failure");
    }
    if (sol && d) { // populate_on_failure, abip.c:249-274
      if (d->n > 0) { if (!sol->x) sol->x = (abip_float *)malloc(sizeof(abip_float) * d->n); if (!sol->s) sol->s = (abip_float *)malloc(sizeof(abip_float) * d->n);
        for (abip_int j = 0; j < d->n; ++j) { sol->x[j] = NAN; sol->s[j] = NAN; } }
      if (d->m > 0) { if (!sol->y) sol->y = (abip_float *)malloc(sizeof(abip_float) * d->m); for (abip_int i = 0; i < d->m; ++i) sol->y[i] = NAN; }
    }
    printf("Failure:%s\n", "could not initialize work");
  }
  abip_finish(w);
  return status;
}

// ---- unit-level device access --------------------------------------------------------------------
abip_int abip_hip_accum_by_A(ABIPWork *w, const abip_float *x, abip_float *y) { // on a sharded solve: this rank's rows
  if (!w || !x || !y) return -1;
  double *dx = w->a_up.p, *dy = w->a_vp.p; // scratch vectors of the BB search (LV >= m, n each; not live outside it): no allocation per call
  HIP_OK(hipMemcpyAsync(dx, x, sizeof(double) * w->n, hipMemcpyHostToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(dy, y, sizeof(double) * w->m, hipMemcpyHostToDevice, w->stream));
  launch(w, ABIP_HIP_K_SPMV_A, PICK(k_spmv_acc, w->dA), w->NB, BS, w->dA.view(), (const double *)dx, dy);
  HIP_OK(hipMemcpyAsync(y, dy, sizeof(double) * w->m, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  harvest_events(w);
  return 0;
}
abip_int abip_hip_accum_by_Atrans(ABIPWork *w, const abip_float *x, abip_float *y) {
  if (!w || !x || !y) return -1;
  double *dx = w->a_up.p, *dy = w->a_vp.p;
  HIP_OK(hipMemcpyAsync(dx, x, sizeof(double) * w->m, hipMemcpyHostToDevice, w->stream));
  HIP_OK(hipMemcpyAsync(dy, y, sizeof(double) * w->n, hipMemcpyHostToDevice, w->stream));
  launch(w, ABIP_HIP_K_SPMV_AT, PICK(k_spmv_acc, w->dAt), w->NB, BS, w->dAt.view(), (const double *)dx, dy);
  HIP_OK(hipMemcpyAsync(y, dy, sizeof(double) * w->n, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  harvest_events(w);
  return 0;
}
abip_int abip_hip_kkt_solve(ABIPWork *w, abip_float *rhs, const abip_float *warm, abip_int iter) {
  if (!w || !rhs) return -1;
  // sharded: rhs / warm are this rank's [y rows | x] pieces; every rank must call
  // a_ut holds the rhs, a_u the warm start (scratch vectors of the BB search; not live outside it)
  if (upload_lvec(w, w->a_ut, rhs, rhs + w->m, 0.0)) return -1;
  if (warm && upload_lvec(w, w->a_u, warm, nullptr, 0.0)) return -1;
  launch(w, ABIP_HIP_K_VEC, k_norm_y, w->NB, BS, (const double *)w->a_ut.p, dims(w), w->part.p);
  const int its = kkt_solve_sync(w, w->a_ut.p, warm ? w->a_u.p : nullptr, iter);
  if (its < 0) return -1;
  std::vector<double> hbuf(w->LV);
  HIP_OK(hipMemcpyAsync(hbuf.data(), w->a_ut.p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
  HIP_OK(hipStreamSynchronize(w->stream));
  std::copy(hbuf.begin(), hbuf.begin() + w->m, rhs);
  std::copy(hbuf.begin() + w->MP, hbuf.begin() + w->MP + w->n, rhs + w->m);
  return its;
}
abip_int abip_hip_get_vector(ABIPWork *w, const char *name, abip_float *out, abip_int cap) {
  if (!w || !name || !out) return -1;
  const abip_int m = w->m, n = w->n, l = m + n + 1;
  struct { const char *nm; DBuf<double> *b; } lv[] = {{"u", &w->u}, {"v", &w->v}, {"u_t", &w->ut}, {"h", &w->h}, {"g", &w->g},
                                                      {"u_avgcon", &w->u_avgc}, {"v_avgcon", &w->v_avgc}};
  for (auto &e : lv)
    if (!strcmp(name, e.nm)) {
      const bool full = strcmp(name, "h") && strcmp(name, "g");
      const abip_int len = full ? l : l - 1;
      if (cap < len) return -1;
      std::vector<double> hbuf(w->LV);
      HIP_OK(hipMemcpyAsync(hbuf.data(), e.b->p, sizeof(double) * w->LV, hipMemcpyDeviceToHost, w->stream));
      HIP_OK(hipStreamSynchronize(w->stream));
      std::copy(hbuf.begin(), hbuf.begin() + m, out);
      std::copy(hbuf.begin() + w->MP, hbuf.begin() + w->MP + n + (full ? 1 : 0), out + m);
      return len;
    }
  struct { const char *nm; const double *p; abip_int len; } sv[] = {{"b", w->b.p, m}, {"c", w->c.p, n}, {"Ax", w->dAt.val.p, (abip_int)w->dAt.val.n}};
  for (auto &e : sv)
    if (!strcmp(name, e.nm)) {
      if (cap < e.len) return -1;
      HIP_OK(hipMemcpyAsync(out, e.p, sizeof(double) * e.len, hipMemcpyDeviceToHost, w->stream));
      HIP_OK(hipStreamSynchronize(w->stream));
      return e.len;
    }
  if (!strcmp(name, "D")) { // all rows of the problem (the scaling is computed from the full matrix on every rank), not only this rank's
    if (w->D.empty() || cap < (abip_int)w->D.size()) return -1;
    std::copy(w->D.begin(), w->D.end(), out);
    return (abip_int)w->D.size();
  }
  if (!strcmp(name, "E")) { if (w->E.empty() || cap < (abip_int)w->E.size()) return -1; std::copy(w->E.begin(), w->E.end(), out); return n; }
  return -1;
}
abip_float abip_hip_get_scalar(ABIPWork *w, const char *name) {
  if (!w || !name) return NAN;
#define RET(nm, val) if (!strcmp(name, nm)) return (abip_float)(val);
  RET("mu", w->mu) RET("beta", w->beta) RET("sigma", w->sigma) RET("gamma", w->gamma) RET("g_th", w->g_th)
  RET("sc_b", w->sc_b) RET("sc_c", w->sc_c) RET("nm_b", w->nm_b) RET("nm_c", w->nm_c) RET("tot_cg_its", w->tot_cg_its) RET("tot_cg_skipped", w->tot_cg_skipped)
  RET("lnnz", w->ldl.lnnz) RET("levels_fwd", w->ldl.F.nlev) RET("levels_bwd", w->ldl.B.nlev) RET("tail", w->ldl.T) RET("admm_iter", w->k) RET("ipm_iter", w->i)
  RET("sell_At", w->dAt.nslices) RET("sell_A", w->dA.nslices) RET("nb", w->NB) RET("dist_cols", w->cg_cols ? 1 : 0) RET("small_solve", w->ldl.small ? 1 : 0) RET("factor_resid", w->factor_resid)
  RET("xcd", w->xcd.on ? 1 : 0) RET("xcd_nz", w->xcd.NZ) RET("xcd_g", w->xcd.on ? w->xcd.G : 0) RET("xcd_batches", w->xcd.batches) RET("xcd_exchanges", w->xcd.exchanges)
  RET("xcd_launches", w->xcd.launches) RET("xcd_outer", (w->xcd.on && w->xcd.outer) ? 1 : 0) RET("xcd_outer_done", w->xcd.outer_done) RET("xcd_lookaheads", w->xcd.lookaheads)
  RET("xcd_whole_launches", w->xcd.whole_launches) RET("xcd_giveups", w->xcd.giveups) RET("stream_stalls", w->stream_stalls) RET("stream_iters", w->stream_iters) RET("aty_valid", w->aty_valid ? 1.0 : 0.0)
#undef RET
  return NAN;
}

// pure host code: order + factor K = [[rho I, A],[A', -I]] the way abip_init does (tail = -1 automatic, 0 none, T forced), solve
// K z = rhs with it on the host and return the factor's shape; rhs (m+n) is overwritten with z
int abip_hip_host_factor_solve(const ABIPMatrix *A, double rho_y, int tail, double *rhs, double *stats8) {
  if (!A || !rhs || !stats8) return -1;
  host::LdlHost F;
  host::set_tail_request(tail);
  const int rc = host::factor_kkt(A, rho_y, F);
  host::set_tail_request(-2);
  if (rc < 0) return -2;
  F.wait_forms();
  if (!getenv("ABIP_HIP_HOST_FACTOR_ONLY")) { // (set by scripts/host_setup_time.py: times the set-up passes of a large matrix without the host's dense LDL' of its tail)
    std::vector<double> b(rhs, rhs + F.N);
    if (host::host_solve(F, b)) return -3;
    std::copy(b.begin(), b.end(), rhs);
  }
  stats8[0] = F.N; stats8[1] = (double)F.lnnz; stats8[2] = F.T; stats8[3] = (double)F.fwd.lev_ptr.size() - 1; stats8[4] = (double)F.bwd.lev_ptr.size() - 1;
  stats8[5] = (double)F.fwd.idx.size(); stats8[6] = 0; stats8[7] = 0;
  return 0;
}

// Unit-level access to the direct back-end's factorisation as both paths use it (LP: K = [[rho I, A],[A', -I]]; conic: qcp_config.c:699-748): any
// symmetric quasi-definite K given by its UPPER triangle in CSC form (32-bit indices).  on_device = 0: ordering + head factor + Schur complement + dense tail
// all on the host (host::host_solve; runs without a GPU); 1: the head on the host, the dense tail factored and every solve applied on the device
// (DevLdl::setup / enqueue).  tail: -1 automatic, 0 none, T > 0 forced.  rhs (N) is overwritten with K^-1 rhs.  stats4 = {T, nnz(L), forward levels, backward levels}.
// Pure host code (runs without a GPU): the plan of the persistent launch for an LP with the sparsity pattern (m, n, Ap, Ai) of A (CSC, as in ABIPMatrix)
// and back-end linsys -- out8 = {admitted 0/1, workgroups, XCDs, NZ, RM, RN, LDS bytes, rows of the dense inverse kept in LDS}; mb (workgroups + 1) /
// nb (workgroups + 1) receive the row boundaries of the slices of A / A' when not NULL (room for 257 each).  For tests/test_xcd_plan_cpu.py.
int abip_hip_xcd_plan(abip_int m, abip_int n, const abip_int *Ap, const abip_int *Ai, int linsys, double *out8, int *mb_out, int *nb_out) {
  if (m <= 0 || n <= 0 || !Ap || !Ai || !out8) return -1;
  std::vector<abip_float> ones((size_t)std::max<abip_int>(Ap[n], 1), 1.0);
  ABIPMatrix A; A.m = m; A.n = n; A.p = const_cast<abip_int *>(Ap); A.i = const_cast<abip_int *>(Ai); A.x = ones.data();
  host::HostCsr hA, hAt;
  host::transpose_to_csr(&A, hA);
  host::csc_as_csr(&A, hAt);
  XcdPlan x;
  std::vector<int> mb, nb;
  long nzA = 0, nzT = 0; int rA = 0, rT = 0, lA = 0, lT = 0;
  const bool ok = xcd_plan(x, hA, hAt, linsys == ABIP_HIP_LINSYS_INDIRECT, mb, nb, &nzA, &nzT, &rA, &rT, &lA, &lT);
  out8[0] = ok ? 1 : 0; out8[1] = x.G; out8[2] = x.nxcd; out8[3] = ok ? x.NZ : 0; out8[4] = ok ? x.RM : 0; out8[5] = ok ? x.RN : 0; out8[6] = ok ? (double)x.lds : 0; out8[7] = ok ? x.minv_lds_rows : 0;
  if (ok && mb_out) std::copy(mb.begin(), mb.end(), mb_out);
  if (ok && nb_out) std::copy(nb.begin(), nb.end(), nb_out);
  return 0;
}
int abip_hip_csc_to_csr(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int *out_ptr, int *out_col, double *out_val) {
  if (nrows <= 0 || ncols <= 0 || !Ap || !Ai || !Ax || !out_ptr || !out_col || !out_val || Ap[ncols] <= 0) return -1;
  char devname[128];
  if (abip_hip_device_info(devname, sizeof(devname), nullptr, nullptr) != 0) return -4;
  const long nnz = Ap[ncols];
  hipStream_t st = nullptr;
  if (hipStreamCreate(&st) != hipSuccess) return -4;
  DBuf<int> cp, ri, op, oc; DBuf<double> cx, ov;
  int ret = 0;
  if (cp.alloc((size_t)ncols + 1) || ri.alloc(nnz) || cx.alloc(nnz) || op.alloc((size_t)nrows + 1) || oc.alloc(nnz) || ov.alloc(nnz) ||
      hipMemcpyAsync(cp.p, Ap, sizeof(int) * ((size_t)ncols + 1), hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(ri.p, Ai, sizeof(int) * nnz, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(cx.p, Ax, sizeof(double) * nnz, hipMemcpyHostToDevice, st) != hipSuccess) ret = -5;
  if (!ret && dev_csc_to_csr(nrows, ncols, nnz, cp.p, ri.p, cx.p, op.p, oc.p, ov.p, st)) ret = -6;
  if (!ret && (hipMemcpyAsync(out_ptr, op.p, sizeof(int) * ((size_t)nrows + 1), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(out_col, oc.p, sizeof(int) * nnz, hipMemcpyDeviceToHost, st) != hipSuccess ||
               hipMemcpyAsync(out_val, ov.p, sizeof(double) * nnz, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) ret = -7;
  (void)hipGetLastError();
  cp.release(); ri.release(); cx.release(); op.release(); oc.release(); ov.release();
  (void)hipStreamDestroy(st);
  return ret;
}
int abip_hip_ldl_solve(int N, const int *Kp, const int *Ki, const double *Kx, int tail, int on_device, double *rhs, double *stats4) {
  if (N <= 0 || !Kp || !Ki || !Kx || !rhs) return -1;
  std::vector<int> kp(Kp, Kp + N + 1), ki(Ki, Ki + Kp[N]);
  std::vector<double> kx(Kx, Kx + Kp[N]);
  host::LdlHost F;
  host::set_tail_request(tail);
  const int rc = host::factor_upper(N, kp, ki, kx, F);
  host::set_tail_request(-2);
  if (rc < 0) return -2;
  F.wait_forms();
  if (stats4) { stats4[0] = F.T; stats4[1] = (double)F.lnnz; stats4[2] = (double)F.fwd.lev_ptr.size() - 1; stats4[3] = (double)F.bwd.lev_ptr.size() - 1; }
  if (!on_device) {
    if (F.dev_schur) host::complete_schur_on_host(F);
    std::vector<double> b(rhs, rhs + N);
    if (host::host_solve(F, b)) return -3; // (takes and returns the caller's order)
    std::copy(b.begin(), b.end(), rhs);
    return 0;
  }
  char devname[128];
  if (abip_hip_device_info(devname, sizeof(devname), nullptr, nullptr) != 0) return -4;
  hipStream_t st = nullptr;
  if (hipStreamCreate(&st) != hipSuccess) return -4;
  DevLdl L;
  DBuf<double> dv; DBuf<Ctl> ctl;
  std::vector<int> pmap(F.P.begin(), F.P.end());
  int ret = 0;
  if (L.setup(F, pmap, st) || dv.alloc(N) || ctl.alloc(1) || hipMemsetAsync(ctl.p, 0, sizeof(Ctl), st) != hipSuccess ||
      hipMemcpyAsync(dv.p, rhs, sizeof(double) * N, hipMemcpyHostToDevice, st) != hipSuccess) ret = -5;
  if (!ret) {
    L.enqueue([&](auto kern, int grid, int block, size_t lds, auto... a) { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a...); }, dv.p, (const Ctl *)ctl.p, 256);
    if (hipMemcpyAsync(rhs, dv.p, sizeof(double) * N, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) ret = -6;
  }
  L.release(); dv.release(); ctl.release();
  (void)hipStreamDestroy(st);
  return ret;
}

// pure host code: ABIP(_normalize_A) as abip_init applies it (A scaled in place; D has m entries, E has n, means = {row, col})
int abip_hip_host_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, double *D, double *E, double *means2) {
  if (!A || !stgs || !D || !E || !means2) return -1;
  std::vector<double> d, e;
  host::normalize_A(A, stgs, d, e, &means2[0], &means2[1]);
  std::copy(d.begin(), d.end(), D); std::copy(e.begin(), e.end(), E);
  return 0;
}

int abip_hip_dist_partition(const ABIPMatrix *A, int world, abip_int *bounds) { // pure host code
  if (!A || !bounds || world < 1 || A->m < world) return -1;
  const abip_int mg = A->m, nnz = A->p[A->n];
  std::vector<abip_int> rcnt(mg + 1, 0);
  for (abip_int q = 0; q < nnz; ++q) rcnt[A->i[q] + 1]++;
  for (abip_int i = 0; i < mg; ++i) rcnt[i + 1] += rcnt[i];
  bounds[0] = 0; bounds[world] = mg;
  for (int g = 1; g < world; ++g) { // first row of rank g: balance non-zeros (+1 per row), keep >= 1 row per rank
    const double target = (double)(nnz + mg) * g / world;
    abip_int lo = 0, hi = mg;
    while (lo < hi) { const abip_int mid = (lo + hi) / 2; if ((double)(rcnt[mid] + mid) < target) lo = mid + 1; else hi = mid; }
    bounds[g] = std::min<abip_int>(std::max<abip_int>(lo, bounds[g - 1] + 1), mg - (world - g));
  }
  return 0;
}
int abip_hip_dist_get_unique_id(void *out128) {
  if (!out128 || !load_rccl(g_dist.api)) return -1;
  return g_dist.api.GetUniqueId(out128) == 0 ? 0 : -2;
}
int abip_hip_dist_init_rccl(int rank, int world, const void *unique_id128) {
  if (world < 1 || rank < 0 || rank >= world || !unique_id128) return -1;
  // The ranks take their control decisions (PCG exit, inner / outer stopping tests) from their own copies of the all-reduced data: every rank must receive the SAME
  // bits.  Ring, tree and the all-pairs schedules reduce each chunk in one place and hand the result round; a one-shot "every rank sums all peers itself" kernel
  // (MSCCL++) adds in a rank-dependent order and may not.  Off unless the user has set it.
  setenv("RCCL_MSCCLPP_ENABLE", "0", 0);
  if (!load_rccl(g_dist.api)) return -2;
  Uid128 id;
  memcpy(id.b, unique_id128, 128);
  rcclComm_t comm = nullptr;
  const int rc = g_dist.api.CommInitRank(&comm, world, id, rank);
  if (rc != 0 || !comm) { fprintf(stderr, "abip_hip: ncclCommInitRank failed (%d)\n", rc); return -3; }
  g_dist.kind = 1; g_dist.rank = rank; g_dist.world = world; g_dist.comm = comm;
  return 0;
}
int abip_hip_dist_init_callback(int rank, int world, abip_hip_allreduce_fn fn, void *ctx) {
  if (world < 1 || rank < 0 || rank >= world || !fn) return -1;
  g_dist.kind = 2; g_dist.rank = rank; g_dist.world = world; g_dist.fn = fn; g_dist.fn_ctx = ctx;
  return 0;
}
// ---- the peer-mapped transport (dev_peer.h).  Step 1 on every rank: allocate the mailbox, hand out its IPC handle (64 bytes); the host program gathers the
// handles of all ranks (any collective it has); step 2: map the peers' mailboxes.  cap_doubles = the longest vector the solve will all-reduce
// (LP, row form: n + 64 padded to 32, + the packed scalars; abip_hip_dist_peer_capacity gives a safe figure).
long abip_hip_dist_peer_capacity(long m, long n) { return ((std::max(m, n) + 63) / 32 * 32) * 2 + 4096; }
int abip_hip_dist_peer_prepare(long cap_doubles, void *handle_out64) {
  if (cap_doubles < 1 || !handle_out64) return -1;
  PeerHost &p = g_dist.peer;
  if (p.mine) return -2;
  auto undo = [&](int rc) { // nothing half-made is left behind: a later call starts from scratch
    if (p.mine) (void)hipFree(p.mine);
    if (p.sync) (void)hipFree(p.sync);
    if (p.hstatus) (void)hipHostFree(p.hstatus);
    p = PeerHost();
    (void)hipGetLastError();
    return rc;
  };
  const size_t bytes = sizeof(double) * (size_t)abip::peer_mailbox_doubles(cap_doubles);
  hipIpcMemHandle_t h;
  // Fine-grained device memory where the runtime can export it over IPC (other agents write it while this agent's kernels poll it); else plain device memory --
  // every access to a mailbox is a system-scope access either way (dev_peer.h).  ABIP_HIP_PEER_FINE=0 skips the attempt.
  const char *fe = getenv("ABIP_HIP_PEER_FINE");
  if (!(fe && atoi(fe) == 0) && hipExtMallocWithFlags(&p.mine, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
    if (hipIpcGetMemHandle(&h, p.mine) == hipSuccess) p.fine = true;
    else { (void)hipFree(p.mine); p.mine = nullptr; (void)hipGetLastError(); }
  } else { p.mine = nullptr; (void)hipGetLastError(); }
  if (!p.mine) {
    if (hipMalloc(&p.mine, bytes) != hipSuccess) return undo(-3);
    if (hipIpcGetMemHandle(&h, p.mine) != hipSuccess) { fprintf(stderr, "abip_hip: hipIpcGetMemHandle failed (%s)\n", hipGetErrorString(hipGetLastError())); return undo(-4); }
  }
  if (hipMemset(p.mine, 0, bytes) != hipSuccess) return undo(-3);
  if (hipMalloc((void **)&p.sync, 64) != hipSuccess || hipMemset(p.sync, 0, 64) != hipSuccess) return undo(-3);
  if (hipHostMalloc((void **)&p.hstatus, sizeof(int), hipHostMallocMapped) != hipSuccess) return undo(-3);
  *p.hstatus = 0;
  p.cap = cap_doubles;
  { const char *e = getenv("ABIP_HIP_PEER_FUSED"); p.fused = !(e && atoi(e) == 0); }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  memcpy(handle_out64, &h, 64);
  if (hipDeviceSynchronize() != hipSuccess) return undo(-3);
  return 0;
}
int abip_hip_dist_init_peer(int rank, int world, const void *handles /* world x 64 bytes, in rank order */) {
  PeerHost &p = g_dist.peer;
  if (world < 1 || world > abip::PEER_MAX || rank < 0 || rank >= world || !handles || !p.mine) return -1;
  for (int r = 0; r < world; ++r) {
    if (r == rank) { p.mapped[r] = p.mine; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char *)handles + 64 * (size_t)r, 64);
    if (hipIpcOpenMemHandle(&p.mapped[r], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      fprintf(stderr, "abip_hip: rank %d cannot map the mailbox of rank %d (%s)\n", rank, r, hipGetErrorString(hipGetLastError()));
      return -2;
    }
  }
  // Coherence: peers on OTHER devices will write into this rank's mailbox while its kernels poll and read it.  HIP promises that to work for
  // fine-grained memory only; a mailbox that had to fall back to plain (coarse-grained) device memory may serve stale lines to its owner whatever the scope of
  // the loads.  Ranks that share one device (the one-GPU dry runs) go through the same L2 and are fine.  So: a coarse-grained mailbox with a peer on another
  // device is REFUSED (-4; the caller keeps RCCL) unless ABIP_HIP_PEER_COARSE_OK=1 says the experiment is wanted.
  {
    int mydev = -1;
    (void)hipGetDevice(&mydev);
    bool cross = false;
    for (int r = 0; r < world; ++r) {
      if (r == rank) continue;
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, p.mapped[r]) == hipSuccess) { if (at.device != mydev) cross = true; }
      else (void)hipGetLastError();
    }
#ifdef ABIP_HIP_TEST_HOOKS
    if (getenv("ABIP_HIP_PEER_PRETEND_CROSS")) cross = true; // one-GPU boxes: behave as if the peers sat on other devices
#endif
    p.cross_device = cross;
    const char *ok = getenv("ABIP_HIP_PEER_COARSE_OK");
    if (cross && !p.fine && !(ok && atoi(ok) != 0)) {
      fprintf(stderr, "abip_hip: rank %d: the mailbox could not be allocated as fine-grained memory and a peer sits on another device: the peer-mapped transport is "
                      "refused (no coherence guarantee); use the RCCL transport (ABIP_HIP_PEER_COARSE_OK=1 overrides)\n", rank);
      for (int r = 0; r < world; ++r) if (p.mapped[r] && p.mapped[r] != p.mine) { (void)hipIpcCloseMemHandle(p.mapped[r]); p.mapped[r] = nullptr; }
      return -4;
    }
  }
  p.ctx.rank = rank; p.ctx.world = world; p.ctx.cap = p.cap; p.ctx.sync = p.sync;
  for (int r = 0; r < world; ++r) p.ctx.mail[r] = (double *)p.mapped[r];
  int *dstatus = nullptr;
  if (hipHostGetDevicePointer((void **)&dstatus, p.hstatus, 0) != hipSuccess) return -3;
  p.ctx.status = dstatus;
  p.ctx.wait_ticks = abip::XP_WAIT_TICKS;
  if (const char *e = getenv("ABIP_HIP_PEER_WAIT_MS")) { const long ms = atol(e); if (ms > 0) p.ctx.wait_ticks = (unsigned long long)ms * 100000ull; }
  p.epoch = 0;
  g_dist.kind = 3; g_dist.rank = rank; g_dist.world = world;
  return 0;
}
int abip_hip_dist_comm_count(void) { // ranks of the live communicator as RCCL reports them (callback transport: the world it was given; none: 0)
  if (g_dist.kind == 2 || g_dist.kind == 3) return g_dist.world;
  if (g_dist.kind != 1 || !g_dist.comm) return 0;
  int cnt = g_dist.world;
  if (g_dist.api.CommCount && g_dist.api.CommCount(g_dist.comm, &cnt) != 0) return -1;
  return cnt;
}
void abip_hip_dist_finalize(void) {
  if (g_dist.kind == 1 && g_dist.comm) g_dist.api.CommDestroy(g_dist.comm);
  {
    PeerHost &p = g_dist.peer;
    if (p.mine) {
      (void)hipDeviceSynchronize();
      for (int r = 0; r < abip::PEER_MAX; ++r) if (p.mapped[r] && p.mapped[r] != p.mine) (void)hipIpcCloseMemHandle(p.mapped[r]);
      (void)hipFree(p.mine); (void)hipFree(p.sync); (void)hipHostFree(p.hstatus);
      p = PeerHost();
    }
  }
  g_dist_aborted = false;
  g_dist.kind = 0; g_dist.rank = 0; g_dist.world = 1; g_dist.comm = nullptr; g_dist.fn = nullptr; g_dist.fn_ctx = nullptr;
}
void abip_hip_dist_rows(ABIPWork *w, abip_int *row0, abip_int *row1) {
  if (!w) return;
  if (row0) *row0 = w->row0;
  if (row1) *row1 = w->row0 + w->m;
}

void abip_hip_profile_enable(ABIPWork *w, unsigned mask) { if (w) w->prof_mask = mask; }
int abip_hip_profile_enable_stamps(ABIPWork *w, unsigned mask) { // classes ABIP_HIP_K_SPMV_AT / _A only (the kernels that take a Stamp)
  if (!w) return -1;
  mask &= (1u << ABIP_HIP_K_SPMV_AT) | (1u << ABIP_HIP_K_SPMV_A);
  if (mask && !w->stamps.p) {
    if (w->stamps.alloc(W::ST_RING)) return -1;
    if (hipMemsetAsync(w->stamps.p, 0, sizeof(Stamp) * W::ST_RING, w->stream) != hipSuccess) return -1;
    if (hipHostMalloc((void **)&w->hstamps, sizeof(Stamp) * W::ST_RING, hipHostMallocDefault) != hipSuccess) return -1;
    w->st_cls.assign(W::ST_RING, 0);
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) w->tick_ms = 1.0 / (double)khz;
  }
  if (!mask && w->stamp_mask) { (void)hipStreamSynchronize(w->stream); w->st_tail = w->st_head; if (w->stamps.p) (void)hipMemsetAsync(w->stamps.p, 0, sizeof(Stamp) * W::ST_RING, w->stream); }
  w->stamp_mask = mask;
  return 0;
}
void abip_hip_profile_read(ABIPWork *w, AbipHipProfile *out, int reset) {
  if (!w || !out) return;
  *out = w->prof;
  if (reset) w->prof = AbipHipProfile{};
}
void abip_hip_sync(ABIPWork *w) { if (w && w->stream) { (void)hipStreamSynchronize(w->stream); harvest_events(w); } }

} // extern "C"
