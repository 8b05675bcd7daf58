// dev_peer.h -- a hand-rolled, DETERMINISTIC all-reduce over peer-mapped buffers ("one-shot direct reduce-scatter + direct all-gather"), the
// third transport of the sharded solve beside RCCL and the host callback.  One process per GPU; every rank owns one device allocation -- its MAILBOX -- that
// all the others have mapped (hipIpcOpenMemHandle: over xGMI on a node, plain device memory when several test ranks share one GPU).
//
// PUSH-based since round 5 (round 4 pulled: every rank copied its whole buffer into its own box and then READ 1/W of every peer's box --
// a full local copy plus round-trip reads over xGMI).  Nobody reads remote memory any more: data and flags are WRITTEN into the consumer's mailbox (posted
// writes, one direction per link) and every wait polls LOCAL memory.
//
//   mailbox of rank r = [ head: contrib[q], reduced[q] epoch counters, one 64-byte line each, written by rank q
//                       | inbox:  W slots of `chunk` doubles -- slot q = rank q's share of chunk r (written by q)
//                       | gather: the reduced vector, chunk q written by rank q ]
//
// One all-reduce of `count` doubles at `buf`, epoch e (every rank makes the same sequence of calls); chunk = ceil(count / W) rounded to 32:
//   1. PUSH    rank q stores element i of its contribution into inbox slot q of rank i / chunk; when its last workgroup is through, contrib[q] <- e in every
//              mailbox (system-scope release behind the stores).  The producer kernel itself can do this instead of writing its result locally: d_peer_push_rows is
//              what the product kernels of the sharded PCG call from their row epilogue (k_spmv_set_t), so the partial A'z never exists as a local vector and the
//              all-reduce costs ONE more launch (step 2 + 3), not a copy kernel and a collective.
//   2. REDUCE  rank r waits (locally) for contrib[q] >= e of every q, then forms chunk r = slot 0 + slot 1 + ... IN RANK ORDER: each element is reduced in exactly
//              one place, so every rank ends up with the same bits whatever arrives first (the ranks take their control decisions from their own copies; an
//              "every rank adds all peers itself" kernel would not guarantee that).  The sums go to buf and into chunk r of every peer's gather area; reduced[r] <- e.
//   3. GATHER  every rank waits (locally) for reduced[q] >= e and copies chunk q of its own gather area into buf.
// Reuse without a second set of buffers: rank q writes inbox slot q of rank r for epoch e + 1 only after it has finished epoch e, which needed reduced[r] --
// sent after r had read its inbox; and r writes chunk r of p's gather area for epoch e + 1 only after p's contribution of epoch e + 1 arrived, which p sends
// after it copied its gather area of epoch e.
// Traffic per rank and call: (W - 1) / W count doubles out (contributions) + (W - 1) chunk doubles out (its reduced chunk to everybody), nothing read remotely:
// every byte crosses a link once, all W - 1 links of a GPU busy at the same time -- the shape SURVEY 8(e) prices for xGMI.
//
// Coherence.  The mailbox is device memory other agents write while kernels of this agent run: every store into a mailbox and every load out of one is a
// system-scope access (sc0 sc1: written through / served past the L1 and the non-coherent L2 lines), the flags are release / acquire at system scope, and the
// allocation is fine-grained where the runtime can export such memory over IPC (solver.hip: abip_hip_dist_peer_prepare; plain device memory otherwise).
// EXPERIMENTAL until a node has checked it bit for bit against RCCL: only one-GPU leases were available (2 - 3 ranks on one GPU: tests/test_gpu_dist.py).
// A wait that sees no progress for XP_WAIT_TICKS of the wall clock raises *status (host-mapped; first culprit wins) and every wait of the launch ends: the
// host then fails the collective, and the transport stays failed -- ranks that have lost step with each other cannot be re-synchronised from one side.
#pragma once
#include <hip/hip_runtime.h>

namespace abip {

constexpr int PEER_MAX = 8;            // ranks of one node
constexpr int PEER_HEAD = 2 * PEER_MAX * 8; // doubles in front of the inbox: 2 x PEER_MAX flags, 64 bytes apart
constexpr int PEER_PAD = 32 * PEER_MAX;     // slack of the inbox / gather areas for the rounding of a chunk
constexpr unsigned long long XP_WAIT_TICKS = 300000000ull; // ticks of wall_clock64 (100 MHz) before a wait gives up: 3 s (PeerCtx::wait_ticks; ABIP_HIP_PEER_WAIT_MS names another limit --
                                                           // several ranks sharing ONE device take turns on it, and a rank whose peer is not scheduled for a while must not fail the dry run)

struct PeerCtx {
  int rank, world;
  long cap;                          // doubles an all-reduce may carry
  double *mail[PEER_MAX];            // every rank's mailbox as mapped here (mail[rank] = this rank's own allocation)
  unsigned *sync;                    // this rank's own: [0] workgroups through the push, [1] through the reduction (reset by the last one)
  int *status;                       // host-mapped: != 0 after a wait gave up (1 + the rank it waited for)
  unsigned long long wait_ticks;     // how long a wait may see no progress (XP_WAIT_TICKS unless ABIP_HIP_PEER_WAIT_MS says otherwise)
};
inline long peer_mailbox_doubles(long cap) { return PEER_HEAD + 2 * (cap + PEER_PAD); }
__host__ __device__ inline long peer_chunk(long count, int W) { return ((count + W - 1) / W + 31) / 32 * 32; } // (the same on every rank: derived from count and W only)

__device__ __forceinline__ unsigned long long *peer_flag(const PeerCtx &c, int owner, int which /* 0 contrib, 1 reduced */, int from) {
  return reinterpret_cast<unsigned long long *>(c.mail[owner]) + (size_t)(which * PEER_MAX + from) * 8;
}
__device__ __forceinline__ double *peer_inbox(const PeerCtx &c, int owner) { return c.mail[owner] + PEER_HEAD; }
__device__ __forceinline__ double *peer_gather(const PeerCtx &c, int owner) { return c.mail[owner] + PEER_HEAD + c.cap + PEER_PAD; }
// system-scope data accesses (see "Coherence" above)
__device__ __forceinline__ void peer_st(double *p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double peer_ld(const double *p) { return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }

// wait for flag `which` from rank `from` in MY mailbox; true = it arrived, false = gave up (or another wait of this launch did)
__device__ __forceinline__ bool peer_wait(const PeerCtx &c, int which, int from, unsigned long long epoch) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(peer_flag(c, c.rank, which, from), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      if ((++spins & 255u) == 0 && (__hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || (unsigned long long)wall_clock64() - t0 > c.wait_ticks)) {
        int expect = 0; // the first culprit stays on record
        (void)__hip_atomic_compare_exchange_strong(c.status, &expect, 1 + from, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // (system scope, every wavefront: nothing read before the flag is used behind it)
  __syncthreads();
  return ok;
}
// The last workgroup of the grid to arrive raises this rank's flag `which` in EVERY mailbox (everything the grid stored before is released with it).
__device__ __forceinline__ void peer_arrive(const PeerCtx &c, int slot, int which, unsigned long long epoch) {
  __threadfence_system(); // (every wavefront: its own stores written through before the barrier lets thread 0 count the workgroup in)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(c.sync + slot, 1u) + 1u;
    if (done == gridDim.x) {
      c.sync[slot] = 0;
      __threadfence_system();
      for (int r = 0; r < c.world; ++r) __hip_atomic_store(peer_flag(c, r, which, c.rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- step 1 as a producer kernel's row epilogue: element `i` of this rank's contribution goes straight into its owner's inbox ----
struct PeerPush {
  int on;                            // 0: the kernel stores locally as before (RCCL / host-callback transports, one GPU)
  PeerCtx c;
  long chunk;
  unsigned long long epoch;
};
__device__ __forceinline__ void d_peer_put(const PeerPush &p, long i, double v) {
  int r = 0;
#pragma unroll
  for (int q = 1; q < PEER_MAX; ++q) r += (q < p.c.world && i >= (long)q * p.chunk) ? 1 : 0;
  double *dst = peer_inbox(p.c, r) + (long)p.c.rank * p.chunk + (i - (long)r * p.chunk);
  if (r == p.c.rank) *dst = v; else peer_st(dst, v);
}
__device__ __forceinline__ void d_peer_push_done(const PeerPush &p) { peer_arrive(p.c, 0, 0, p.epoch); } // every workgroup of the producer's grid, once, at its end

// ---- step 1 on its own (a contribution that already exists as a local vector) ----
__device__ __forceinline__ void d_peer_push(const PeerCtx &c, const double *buf, long count, unsigned long long epoch) {
  PeerPush p; p.on = 1; p.c = c; p.chunk = peer_chunk(count, c.world); p.epoch = epoch;
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = t0; i < count; i += stride) d_peer_put(p, i, buf[i]);
  d_peer_push_done(p);
}
// ---- steps 2 and 3 ----
__device__ __forceinline__ void d_peer_reduce_gather(const PeerCtx &c, double *buf, long count, unsigned long long epoch) {
  const int W = c.world, me = c.rank;
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long chunk = peer_chunk(count, W);
  const long c0 = min((long)me * chunk, count), c1 = min(c0 + chunk, count);
  for (int q = 0; q < W; ++q) if (!peer_wait(c, 0, q, epoch)) return; // (my own flag too: my share is complete only when my last producer workgroup has arrived)
  const double *in = peer_inbox(c, me);
  for (long i = c0 + t0; i < c1; i += stride) {
    const long k = i - c0;
    double s = (me == 0) ? in[k] : peer_ld(in + k);
    for (int q = 1; q < W; ++q) s += (q == me) ? in[(long)q * chunk + k] : peer_ld(in + (long)q * chunk + k);
    buf[i] = s;
    for (int r = 0; r < W; ++r) if (r != me) peer_st(peer_gather(c, r) + i, s);
  }
  peer_arrive(c, 1, 1, epoch);
  const double *ga = peer_gather(c, me);
  for (int q = 0; q < W; ++q) {
    if (q == me) continue;
    if (!peer_wait(c, 1, q, epoch)) return;
    const long q0 = min((long)q * chunk, count), q1 = min(q0 + chunk, count);
    for (long i = q0 + t0; i < q1; i += stride) buf[i] = peer_ld(ga + i);
  }
}

// the stand-alone forms the solver enqueues: the whole all-reduce where it called ncclAllReduce, or steps 2 + 3 behind a producer kernel that pushed
__global__ __launch_bounds__(256) void k_peer_allreduce(PeerCtx c, double *buf, long count, unsigned long long epoch) {
  d_peer_push(c, buf, count, epoch);
  d_peer_reduce_gather(c, buf, count, epoch);
}
__global__ __launch_bounds__(256) void k_peer_reduce_gather(PeerCtx c, double *buf, long count, unsigned long long epoch) { d_peer_reduce_gather(c, buf, count, epoch); }

} // namespace abip
