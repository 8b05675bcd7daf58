// dev_common.h -- shared device-side building blocks for the ABIP hot path on gfx950.
//
// Conventions
//  * every kernel is launched with BS = 256 threads (4 wavefronts of 64) and a
//    persistent grid of NB <= MAXNB blocks that grid-strides over its work;
//  * a reduction never uses atomics: the producer kernel writes one partial per block
//    into a slot of the partials table, and the CONSUMER kernel re-reduces those NB
//    numbers in a fixed order at its start (every block gets the bit-identical scalar).
//    That costs one 8 KB L2 read per block instead of a kernel boundary or a fence,
//    and makes every run bit-reproducible;
//  * "l-vectors" (u, v, u_t, ... of length m+n+1, reference layout [y | x | tau],
//    src/abip-lp/src/abip.c:368-378) are stored as [y(m) | pad | x(n) | tau] with the
//    x block starting at the 256-byte aligned offset MP so that both blocks vectorise.
#pragma once
#include <hip/hip_runtime.h>

namespace abip {

constexpr int BS = 256;        // threads per block
constexpr int WAVES = BS / 64; // wavefronts per block
constexpr int CHUNK = 1024;    // non-zeros staged through LDS per row block (4 per thread)
constexpr int MAXNB = 2048;    // upper bound on the persistent grid == partials per slot

// CSR view (A' is the CSC of A read as CSR; A is the explicit transpose, indirect.c:81-139).
struct Csr {
  const int *ptr;    // nrows+1
  const int *idx;    // nnz
  const double *val; // nnz
  const int4 *rbd;   // row-block descriptors {first row, last row + 1, first nnz, last nnz + 1} (host-built:
                     // <= CHUNK non-zeros and <= CHUNK rows per block, or exactly one longer row)
  int nrb;
  int nrows;
  // optional sliced-ELL image of the same matrix (SELL-64, natural row order; host_setup: build_sell): slice s holds rows 64 s .. 64 s + 63
  // column-major, padded to its longest row.  One lane owns one row: no LDS, no shuffles, no barriers, a row's products add up in column
  // order.  Built only where the padding is small (regular row lengths: e.g. the CSC columns of C4's A); nslices == 0 otherwise.
  const double *sval;
  const int *sidx;
  const long *soff; // first stored entry of a slice
  const int *slen;  // entries per lane in a slice
  int nslices;
};

// partial-sum slots (each MAXNB doubles)
enum Slot : int {
  S_WG = 0,   // sum rho*(u+v)_y*g_y + (u+v)_x*g_x          (next rhs, abip.c:557)
  S_BN,       // ||rhs_y||^2                                  (cg_tol, indirect.c:406)
  S_RR0, S_RR1, // ||r||^2, ping-pong by CG iteration parity   (indirect.c:359,375)
  S_ZR0, S_ZR1, // z'r                                         (indirect.c:263-280)
  S_PG,       // p'Gp                                         (indirect.c:371)
  S_DH,       // u_t[0:l-1)'h                                 (abip.c:560)
  S_NU, S_NV, S_CX, S_BY,         // ||u||^2, ||v||^2, c'x, b'y    (abip.c:1993-1996)
  S_NUA, S_NVA, S_CXA, S_BYA,     // the same for the averaged iterate (abip.c:2031-2036)
  S_QP, S_RP, S_NAX,              // sum (Ax-b tau)^2, D-weighted, ||D Ax||^2   (abip.c:1984-1987, 407-413)
  S_QD, S_RD, S_NATY,             // sum (A'y+s-c tau)^2, E-weighted, ||E(A'y+s)||^2 (abip.c:1989-1992, 443-449)
  S_QPA, S_RPA, S_NAXA, S_QDA, S_RDA, S_NATYA, // averaged iterate
  S_XS, S_XMIN,                   // sum u_i v_i, min u_i v_i       (abip.c:962-965)
  S_A0, S_A1, S_A2, S_A3, S_A4,   // BB search: utut, utv, uu, vv, uv (adaptive.c:170-174)
  S_T0, S_T1,                     // scratch dots
  S_ZZ, S_ZP, S_TT,               // sharded PCG: z'z, z'p_old (summed over ranks), ||A'p||^2 (replicated) -> p'Gp without a second collective
  S_COUNT
};

// What a persistent launch that spans outer iterations (dev_xcd.h, XcdOuter) hands back: the loop state of abip.c:2102-2294 at the point it stopped
struct XcdOut {
  int phase;        // where the host goes on: 0 = inner loop (iteration (k, j) is next), 1 = outer end pending (abip.c:2217), 2 = outer begin pending with i already advanced (abip.c:2102)
  int reason;       // XR_* (dev_xcd.h): why the launch ended
  int final_check, avg_crit;
  int stats_valid;  // Ctl::out describes the current iterate (the last thing the launch did was an ADMM iteration)
  int avg_stats;    // ... and holds the averaged iterate's sums too
  int outer_done;   // outer iterations closed inside the launch (mu update, reinitialisation, Barzilai-Borwein search)
  int log_n;        // rows written to the outer-iteration log
  int last_cg, bb_lookaheads;
  long i, j, k;     // the loop variables of abip.c:2102, 2131 and the running iteration count
  long ran;         // ADMM iterations of this launch
  long solves;      // KKT solves of this launch (iterations + look-ahead solves of the search)
  long cg_total;    // PCG iterations of all of them
  long cg_skipped;  // ... of which counted as the reference counts them but not executed (look-ahead solves handed over, XcdOuter::bb_reuse)
  double mu, beta, dyn_sigma;
};

// device-resident control block (also copied to the host once per ADMM iteration)
struct Ctl {
  int halt;      // 1: every gated kernel returns immediately
  int cg_done;   // set by the first kernel that sees ||r|| < tol (idempotent: every block decides alike)
  int cg_it;     // CG updates completed so far (written by the update kernel, read by the next SpMV)
  int it_cur;    // iteration index the current SpMV pair works on (written by spmv_At, read by the others)
  double beta_cur; // z'r / z'r_old of the current iteration
  double zr_cur;   // z'r entering the current iteration
  double zr_hist[2];
  double cg_tol;
  double out[96];  // finalised reductions, index = Slot
  // inner-loop exit test, evaluated by k_finalize (iterate_Q_norm_resd, abip.c:2027-2050, and the comparison of abip.c:2173)
  double metric;   // min(sqrt(Qres)/norm, sqrt(Qres_avg)/norm_avg)
  int avg_crit;    // 1: the averaged iterate gave the smaller value
  int it_count;    // ADMM iterations completed since abip_init (never reset)
  double pp_cur;   // sharded PCG: ||p||^2 of the current direction, by the recurrence ||z + beta p||^2 = z'z + 2 beta z'p + beta^2 ||p||^2
  long xcd_cg_total; // one-XCD persistent launch (dev_xcd.h): PCG iterations of all the ADMM iterations it ran
  XcdOut xo;         // ... and, when the launch spans outer iterations, the loop state it stopped in
  // the Barzilai-Borwein search of the launch path with its decisions on the device (solver.hip: adaptive_search_stream; adaptive.c:87-251)
  double bb_prev;    // beta_prev of the search in flight
  double bb_beta;    // beta as the last look-ahead left it (w->beta when the search ends)
  int bb_act;        // what lp_bb_beta decided last: 0 stop, 1 go on with v_prev rebuilt for the new penalty, 2 go on as is
  int bb_it;         // look-aheads completed
  int bb_stage;      // inside a look-ahead: 0 before the first step, 1 the first step is done, 2 the second
  int bb_cg[2];      // PCG iterations of the last first / second solve
  long bb_cg_total;  // ... and of all solves of this search
  int bb_skip;       // 1: the look-ahead in flight starts from the previous one's second step (its first solve would repeat that solve bit for bit: k_adapt_next)
  int bb_cg_skipped; // PCG iterations of bb_cg_total that were counted, not executed: the solves a look-ahead took over from its predecessor (k_adapt_resume)
};

// Device-side launch timing (bench.py's roofline leg, inside the timed region: no hipEvent records, no second pass).  A kernel handed a
// Stamp notes when its first sampled workgroup began and when its last sampled one ended, in ticks of the constant-rate wall clock
// (hipDeviceAttributeWallClockRate).  Every 64th workgroup is sampled (one atomic each: ~32 per launch of the full grid); the begin
// tick is stored inverted so that a zeroed record serves both atomicMax.  A launch that returns at one of its gates leaves t1 == 0.
struct Stamp { unsigned long long t0_inv, t1; };
__device__ __forceinline__ void stamp_begin(Stamp *st) {
  if (st && threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicMax(&st->t0_inv, ~(unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_end(Stamp *st) {
  if (st && threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicMax(&st->t1, (unsigned long long)wall_clock64());
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// Sum NS per-thread values over the block; every thread receives the totals (fixed order).
template <int NS>
__device__ __forceinline__ void block_sum(double (&v)[NS], double *sm /* NS*WAVES */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s) v[s] = wave_sum(v[s]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sm[s * WAVES + wave] = v[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double t = sm[s * WAVES];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) t += sm[s * WAVES + w];
    v[s] = t;
  }
}

template <int NS>
__device__ __forceinline__ void write_partials(double *part, const int (&slots)[NS], double (&v)[NS], double *sm, int vb) {
  block_sum<NS>(v, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) part[slots[s] * MAXNB + vb] = v[s];
  }
}
template <int NS>
__device__ __forceinline__ void write_partials(double *part, const int (&slots)[NS], double (&v)[NS], double *sm) {
  write_partials<NS>(part, slots, v, sm, (int)blockIdx.x);
}

// Re-reduce the nb per-block partials of NS slots; all threads of all blocks get identical totals.
// All MAXNB/BS loads of a thread are issued before the first add: a rolled loop would serialise them into that
// many dependent L2 round trips (measured: ~8 us of a 40 us kernel).
template <int NS>
__device__ __forceinline__ void read_partials(const double *part, const int (&slots)[NS], int nb, double (&out)[NS], double *sm) {
  constexpr int PER = MAXNB / BS;
  double t[NS][PER];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = threadIdx.x + u * BS;
      t[s][u] = (i < nb) ? part[slots[s] * MAXNB + i] : 0.0;
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < PER; ++u) acc += t[s][u];
    out[s] = acc;
  }
  block_sum<NS>(out, sm);
}

// Consumers take their scalars either from the partials (single GPU) or, when the rows of A are sharded over several
// GPUs, from the table `gs` that k_fold filled and the all-reduce summed over the ranks.
template <int NS>
__device__ __forceinline__ void get_scalars(const double *part, const int (&slots)[NS], int nb, double (&out)[NS], double *sm, const double *gs) {
  if (gs) {
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = gs[slots[s]];
  } else {
    read_partials<NS>(part, slots, nb, out, sm);
  }
}

__device__ __forceinline__ int pow2_floor(int x) { return x <= 1 ? 1 : 1 << (31 - __clz(x)); }

// CSR-stream SpMV skeleton.  For every row block: all 256 threads stream the block's
// non-zeros (coalesced: lane k reads entry k) and write NV products per entry into LDS;
// then groups of `lpr` lanes reduce one row each from LDS and one lane per row runs the
// row epilogue.  A block holding a single row longer than CHUNK is reduced by the whole
// workgroup instead.  prod(col, a, out[NV]) forms the products; rowf(row, acc[NV]) consumes a row.
//
// Latency hiding (the kernel is a chain of dependent memory round trips, not a bandwidth stream, unless
// enough of them overlap): the row-block descriptor {r0, r1, k0, k1} is ONE 16-byte load; the block's row
// pointers are staged into LDS by the same round trip that fetches values and indices; and the NEXT
// block's descriptor, values and indices are requested before the current block's LDS reduction starts, so
// only the gather itself sits on the critical path of an iteration.
// (Double-buffering the LDS stage to drop the second barrier was measured: 24 KB of LDS per workgroup costs more
// occupancy than the barrier costs time -- 36 -> 44 us per SpMV on C4 -- so the stage is single-buffered.)
// `pre` (optional) runs once per workgroup AFTER the first block's descriptor, values and indices have been requested and before
// anything is consumed: a kernel's own entry test (reduce the previous kernel's partials, decide "converged?") then overlaps with
// those loads instead of preceding them.  It returns false to abandon the kernel; it may use sm and workgroup barriers.
// (vb, vgrid): the workgroup's index and count among the workgroups working on M -- blockIdx.x / gridDim.x unless one launch
// carries two products side by side (k_q_both).
template <int NV, class ProdF, class RowF, class PreF>
__device__ __forceinline__ void spmv_stream(const Csr M, double *lds /* NV*CHUNK */, int *lptr /* CHUNK+1 */, double *sm /* NV*WAVES */,
                                            ProdF prod, RowF rowf, PreF pre, int vb, int vgrid) {
  int b = vb;
  if (b >= M.nrb) { (void)pre(); return; }
  int4 d = M.rbd[b];
  double a[4];
  int c[4];
  auto fetch = [&](const int4 &dd, double(&aa)[4], int(&cc)[4]) {
    const int nn = dd.w - dd.z;
    if (nn > CHUNK) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = threadIdx.x + u * BS;
      aa[u] = 0.0; cc[u] = 0;
      // (non-temporal loads for the once-read matrix stream were measured: +6 us per SpMV, no less gather re-fetch)
      if (k < nn) { aa[u] = M.val[dd.z + k]; cc[u] = M.idx[dd.z + k]; }
    }
  };
  fetch(d, a, c);
  if (!pre()) return;
  for (;;) {
    const int bn = b + vgrid;
    const bool has_next = bn < M.nrb;
    int4 dn = d;
    if (has_next) dn = M.rbd[bn];
    const int r0 = d.x, k0 = d.z, k1 = d.w;
    const int nn = k1 - k0, R = d.y - d.x;
    double an[4];
    int cn[4];
    if (nn <= CHUNK) {
      for (int t = threadIdx.x; t <= R; t += BS) lptr[t] = M.ptr[r0 + t] - k0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = threadIdx.x + u * BS;
        if (k < nn) {
          double pr[NV];
          prod(c[u], a[u], pr);
#pragma unroll
          for (int v = 0; v < NV; ++v) lds[v * CHUNK + k] = pr[v];
        }
      }
      __syncthreads();
      if (has_next) fetch(dn, an, cn);
      int lpr = pow2_floor(BS / (R > 0 ? R : 1));
      if (lpr > 64) lpr = 64;
      const int ngrp = BS / lpr, grp = threadIdx.x / lpr, q = threadIdx.x % lpr;
      for (int base = 0; base < R; base += ngrp) {
        const int r = base + grp;
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
        if (r < R) {
          const int s = lptr[r], e = lptr[r + 1];
          for (int k = s + q; k < e; k += lpr) {
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += lds[v * CHUNK + k];
          }
        }
        for (int off = lpr >> 1; off > 0; off >>= 1) {
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[v] += __shfl_xor(acc[v], off, 64);
        }
        if (r < R && q == 0) rowf(r0 + r, acc);
      }
      __syncthreads();
    } else { // one long row
      double acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = 0.0;
      for (int k = k0 + threadIdx.x; k < k1; k += BS) {
        double pr[NV];
        prod(M.idx[k], M.val[k], pr);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += pr[v];
      }
      if (has_next) fetch(dn, an, cn);
      block_sum<NV>(acc, sm);
      if (threadIdx.x == 0) rowf(r0, acc);
      __syncthreads();
    }
    if (!has_next) break;
    b = bn; d = dn;
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = an[u]; c[u] = cn[u]; }
  }
}

template <int NV, class ProdF, class RowF, class PreF>
__device__ __forceinline__ void spmv_stream(const Csr M, double *lds, int *lptr, double *sm, ProdF prod, RowF rowf, PreF pre) {
  spmv_stream<NV>(M, lds, lptr, sm, prod, rowf, pre, (int)blockIdx.x, (int)gridDim.x);
}
template <int NV, class ProdF, class RowF>
__device__ __forceinline__ void spmv_stream(const Csr M, double *lds, int *lptr, double *sm, ProdF prod, RowF rowf) {
  spmv_stream<NV>(M, lds, lptr, sm, prod, rowf, [] { return true; }, (int)blockIdx.x, (int)gridDim.x);
}

// SELL-64 product with the interface of spmv_stream: prod(col, a, out[NV]), rowf(row, acc[NV]) by the lane that owns the row, pre() once per
// workgroup before anything is consumed.  Measured on C4's A' (tools/sell_probe.hip): 27 us against 33 us for the CSR-stream kernel; on
// matrices whose natural-order slices pad badly (C4's A: Poisson row lengths, +33 %) the stream kernel stays the faster one.
template <int NV, class ProdF, class RowF, class PreF>
__device__ __forceinline__ void spmv_sell(const Csr M, ProdF prod, RowF rowf, PreF pre, int vb, int vgrid) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  int s = vb * WAVES + (threadIdx.x >> 6);
  const int ns = vgrid * WAVES;
  long base = 0;
  int len = 0;
  if (s < M.nslices) { base = M.soff[s]; len = M.slen[s]; }
  if (!pre()) return;
  while (s < M.nslices) {
    const int sn = s + ns;
    long nbase = 0; int nlen = 0;
    if (sn < M.nslices) { nbase = M.soff[sn]; nlen = M.slen[sn]; } // next slice's extent: in flight during this slice's gathers
    const double *v = M.sval + base + lane;
    const int *ix = M.sidx + base + lane;
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.0;
    int k = 0;
    for (; k + U <= len; k += U) {
      double a[U]; int c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = v[(long)(k + u) * 64]; c[u] = ix[(long)(k + u) * 64]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        double pr[NV];
        prod(c[u], a[u], pr);
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] += pr[q];
      }
    }
    if (k < len) {
      double a[U]; int c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = 0.0; c[u] = 0; if (k + u < len) { a[u] = v[(long)(k + u) * 64]; c[u] = ix[(long)(k + u) * 64]; } }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (k + u < len) {
          double pr[NV];
          prod(c[u], a[u], pr);
#pragma unroll
          for (int q = 0; q < NV; ++q) acc[q] += pr[q];
        }
      }
    }
    const int row = s * 64 + lane;
    if (row < M.nrows) rowf(row, acc);
    s = sn; base = nbase; len = nlen;
  }
}
// the product in the layout chosen at COMPILE time (a kernel that carried both paths would pay for the wider one in registers: the
// stream path fits 64 VGPRs = 8 waves per SIMD = the whole persistent grid resident; host code picks kern<true> when M.nslices > 0)
template <int NV, bool SELL, class ProdF, class RowF, class PreF>
__device__ __forceinline__ void spmv_rows(const Csr M, double *lds, int *lptr, double *sm, ProdF prod, RowF rowf, PreF pre, int vb, int vgrid) {
  if (SELL) spmv_sell<NV>(M, prod, rowf, pre, vb, vgrid);
  else spmv_stream<NV>(M, lds, lptr, sm, prod, rowf, pre, vb, vgrid);
}
template <int NV, bool SELL, class ProdF, class RowF, class PreF>
__device__ __forceinline__ void spmv_rows(const Csr M, double *lds, int *lptr, double *sm, ProdF prod, RowF rowf, PreF pre) {
  spmv_rows<NV, SELL>(M, lds, lptr, sm, prod, rowf, pre, (int)blockIdx.x, (int)gridDim.x);
}
template <int NV, bool SELL, class ProdF, class RowF>
__device__ __forceinline__ void spmv_rows(const Csr M, double *lds, int *lptr, double *sm, ProdF prod, RowF rowf) {
  spmv_rows<NV, SELL>(M, lds, lptr, sm, prod, rowf, [] { return true; }, (int)blockIdx.x, (int)gridDim.x);
}

} // namespace abip
