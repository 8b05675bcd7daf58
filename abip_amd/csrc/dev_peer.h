// dev_peer.h -- a hand-rolled, DETERMINISTIC all-reduce over peer-mapped buffers ("one-shot direct reduce-scatter + direct all-gather"), the
// third transport of the sharded solve beside RCCL and the host callback.  One process per GPU; every rank owns one device allocation -- its MAILBOX -- that
// all the others have mapped (hipIpcOpenMemHandle: over xGMI on a node, plain device memory when several test ranks share one GPU).
//
//   mailbox = [ head: ready_contrib, ready_reduced (epoch counters) | box: this rank's contribution, `cap` doubles | red: the chunk this rank reduces ]
//
// One all-reduce of `count` doubles at `buf`, epoch e (every rank makes the same sequence of calls):
//   1. every rank copies buf into its own box; when its last workgroup is through, ready_contrib <- e                         (release, system scope)
//   2. rank r waits for ready_contrib >= e of every peer, then forms chunk r = sum over the ranks q = 0, 1, ... IN THAT ORDER of box_q[chunk r]: each
//      element is reduced in exactly one place, so every rank ends up with the same bits whatever arrives first (the ranks take their control decisions
//      from their own copies; an "every rank adds all peers itself" kernel would not guarantee that).  The sum goes to red and to buf; ready_reduced <- e
//   3. every rank waits for ready_reduced >= e of peer q and copies red_q into chunk q of buf: the all-gather.
// A box is not written again before every peer has read it (a rank reaches step 1 of epoch e + 1 only behind step 3 of epoch e, which waited for every
// peer's step 2), and the same argument protects red.  Traffic per rank: count doubles out of every peer's box (1/W of each) + (W-1)/W count from the reds:
// every byte crosses a link once in each direction, all W-1 links of a GPU busy at the same time -- the shape SURVEY 8(e) prices for xGMI.
//
// The device function below is what a producer kernel's epilogue can call instead of returning (its grid must be co-resident: the workgroups wait for
// each other through an atomic counter); k_peer_allreduce is the stand-alone form the solver enqueues where it called ncclAllReduce.
// A wait that sees no progress for XP_SPIN polls raises *status (host-mapped) and every wait of the launch ends: the host then fails the collective.
#pragma once
#include <hip/hip_runtime.h>

namespace abip {

constexpr int PEER_MAX = 8;          // ranks of one node
constexpr int PEER_HEAD = 32;        // doubles in front of the box (the two counters, padded to 256 bytes)
constexpr long XP_SPIN = 1L << 24;   // polls before a wait gives up (seconds)

struct PeerCtx {
  int rank, world;
  long cap;                          // doubles a box holds
  double *mail[PEER_MAX];            // every rank's mailbox as mapped here (mail[rank] = this rank's own allocation)
  unsigned *sync;                    // this rank's own: [0] workgroups through step 1, [1] through step 2 (reset by the last one), [2] .. spare
  int *status;                       // host-mapped: != 0 after a wait gave up
};

__device__ __forceinline__ unsigned long long *peer_flag(const PeerCtx &c, int r, int which) { return reinterpret_cast<unsigned long long *>(c.mail[r]) + which; }
__device__ __forceinline__ double *peer_box(const PeerCtx &c, int r) { return c.mail[r] + PEER_HEAD; }
__device__ __forceinline__ double *peer_red(const PeerCtx &c, int r) { return c.mail[r] + PEER_HEAD + c.cap; }

// true = the flag arrived; false = gave up (or another wait of this launch did)
__device__ __forceinline__ bool peer_wait(const PeerCtx &c, int r, int which, unsigned long long epoch) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    long spins = 0;
    while (__hip_atomic_load(peer_flag(c, r, which), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      if ((++spins & 1023) == 0 && (__hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || spins > XP_SPIN)) {
        __hip_atomic_store(c.status, 1 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // (system scope, every wavefront: nothing cached from before the flag is read behind it)
  __syncthreads();
  return ok;
}
// the last workgroup of the grid to arrive publishes the flag (everything the grid wrote before is released with it)
__device__ __forceinline__ void peer_arrive(const PeerCtx &c, int slot, int which, unsigned long long epoch) {
  __threadfence_system(); // (every wavefront: its own stores written back before the barrier lets thread 0 count the workgroup in)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(c.sync + slot, 1u) + 1u;
    if (done == gridDim.x) {
      c.sync[slot] = 0;
      __hip_atomic_store(peer_flag(c, c.rank, which), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__device__ __forceinline__ void d_peer_allreduce(const PeerCtx &c, double *buf, long count, unsigned long long epoch) {
  const int W = c.world, me = c.rank;
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long chunk = ((count + W - 1) / W + 31) / 32 * 32; // (the same on every rank: derived from count and W only)
  // 1. my contribution
  double *mybox = peer_box(c, me);
  for (long i = t0; i < count; i += stride) mybox[i] = buf[i];
  peer_arrive(c, 0, 0, epoch);
  // 2. my chunk, summed in rank order
  const long c0 = std::min((long)me * chunk, count), c1 = std::min(c0 + chunk, count);
  for (int q = 0; q < W; ++q) if (q != me && !peer_wait(c, q, 0, epoch)) return;
  if (!peer_wait(c, me, 0, epoch)) return; // (my own box is complete only when my last workgroup has arrived)
  double *myred = peer_red(c, me);
  for (long i = c0 + t0; i < c1; i += stride) {
    double s = peer_box(c, 0)[i];
    for (int q = 1; q < W; ++q) s += peer_box(c, q)[i];
    myred[i - c0] = s;
    buf[i] = s;
  }
  peer_arrive(c, 1, 1, epoch);
  // 3. everybody else's chunk
  for (int q = 0; q < W; ++q) {
    if (q == me) continue;
    if (!peer_wait(c, q, 1, epoch)) return;
    const long q0 = std::min((long)q * chunk, count), q1 = std::min(q0 + chunk, count);
    const double *rq = peer_red(c, q);
    for (long i = q0 + t0; i < q1; i += stride) buf[i] = rq[i - q0];
  }
}

__global__ __launch_bounds__(256) void k_peer_allreduce(PeerCtx c, double *buf, long count, unsigned long long epoch) { d_peer_allreduce(c, buf, count, epoch); }

} // namespace abip
