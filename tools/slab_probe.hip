// tools/slab_probe.hip -- developer micro-benchmark (not part of the library): an SpMV that stages x in LDS.
//
// Why: on the C4-shaped product every random 8-byte gather pulls a 128-byte line L2 -> L1, so 5 M gathers move 640 MB
// through the L1 fill path (measured: 21 us, ~30 TB/s = the L2's rate).  Here a workgroup owns a block of rows, walks the
// column range in slabs of W doubles, copies each slab of x into LDS once (coalesced) and gathers from LDS.  Per CU the fill
// path then carries |x| / G + its share of the matrix stream instead of 128 B per non-zero.
//   * the rows of a workgroup are dealt to its waves in contiguous runs; a wave's non-zeros are stored as ONE stream sorted by
//     (slab, row) and loaded into registers in one burst at kernel start (all HBM traffic in flight at once);
//   * row sums live in LDS; rows are wave-private, LDS operations of one wave execute in order => no atomics, fixed summation order;
//   * entries of one (slab, row) that sit in the same 64-entry chunk are combined with a segmented shuffle reduction whose
//     distances the host packed into the index word: word = col_in_slab (14 bits) | row_in_workgroup (12) << 14 | run (6) << 26;
//   * G > 1 splits the columns over G workgroups per row block; each writes a partial y (summed by the consumer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)


struct SlabDev {
  const double *val; const uint32_t *pk;
  const int *part_row; // P+1
  const long *wave_off; const int *wave_nch; const int *wptr; // per (p,g,w)
  int P, G, S, SG, W, C, R, cb /* bits of col_in_slab */, rb /* bits of the row */;
};

struct SlabHost {
  int R, C, W, S, G, SG, P, NW, CH;
  std::vector<double> val; std::vector<uint32_t> pk;
  std::vector<int> part_row; std::vector<long> wave_off; std::vector<int> wave_nch, wptr;
  bool ok = true; long padded = 0;
};

static SlabHost build(const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<double> &val, int R, int C, int W, int G, int P, int NW, int CH, int RCAP, int cb) {
  SlabHost H; H.R = R; H.C = C; H.W = W; H.G = G; H.P = P; H.NW = NW; H.CH = CH;
  H.S = (C + W - 1) / W; H.SG = (H.S + G - 1) / G;
  const long nnz = ptr[R];
  H.part_row.assign(P + 1, R);
  { // contiguous row blocks: greedy filling under a cost cap (nnz + rows) and the row cap; smallest cost cap that needs <= P blocks
    auto fill = [&](long cap, std::vector<int> *out) {
      int r = 0, np = 0;
      while (r < R) {
        if (out) (*out)[np] = r;
        long c = 0; int cnt = 0;
        while (r < R && cnt < RCAP && (c + (ptr[r + 1] - ptr[r]) + 1 <= cap || cnt == 0)) { c += ptr[r + 1] - ptr[r] + 1; ++r; ++cnt; }
        ++np;
        if (np > P) return np;
      }
      if (out) for (int q = np; q <= P; ++q) (*out)[q] = R;
      return np;
    };
    long lo = 1, hi = nnz + R;
    while (lo < hi) { const long mid = (lo + hi) / 2; if (fill(mid, nullptr) <= P) hi = mid; else lo = mid + 1; }
    if (fill(lo, &H.part_row) > P) { H.ok = false; return H; }
  }
  H.wave_off.assign((size_t)P * G * NW, 0); H.wave_nch.assign((size_t)P * G * NW, 0); H.wptr.assign((size_t)P * G * NW * (H.SG + 1), 0);
  struct E { int slab, row, col; double v; };
  std::vector<E> ent;
  for (int p = 0; p < P; ++p) {
    const int r0 = H.part_row[p], r1 = H.part_row[p + 1];
    for (int g = 0; g < G; ++g) {
      const int c0 = g * H.SG * W, c1 = std::min(C, (g + 1) * H.SG * W);
      // entries of this (p, g) per row
      std::vector<int> cnt(r1 - r0 + 1, 0);
      for (int r = r0; r < r1; ++r) { int c = 0; for (int k = ptr[r]; k < ptr[r + 1]; ++k) c += (idx[k] >= c0 && idx[k] < c1); cnt[r - r0 + 1] = cnt[r - r0] + c; }
      const int tot = cnt[r1 - r0];
      int rr = r0;
      for (int w = 0; w < NW; ++w) { // rows [rr, re) for wave w: contiguous, balanced by entries (+1 per row)
        const double target = (double)(tot + (r1 - r0)) * (w + 1) / NW;
        int re = rr;
        while (re < r1 && ((double)(cnt[re - r0 + 1] + (re - r0 + 1)) <= target || (w == NW - 1))) ++re;
        if (w == NW - 1) re = r1;
        ent.clear();
        for (int r = rr; r < re; ++r)
          for (int k = ptr[r]; k < ptr[r + 1]; ++k)
            if (idx[k] >= c0 && idx[k] < c1) ent.push_back({(idx[k] - c0) / W, r - r0, (idx[k] - c0) % W, val[k]});
        std::stable_sort(ent.begin(), ent.end(), [](const E &a, const E &b) { return a.slab != b.slab ? a.slab < b.slab : a.row < b.row; });
        const size_t wi = ((size_t)p * G + g) * NW + w;
        const int nch = ((int)ent.size() + 63) / 64 + CH; // CH chunks of slack: the prefetch reads past the end of a slab's entries
        H.wave_off[wi] = (long)H.val.size(); H.wave_nch[wi] = nch;
        int *wp = &H.wptr[wi * (H.SG + 1)];
        { int e = 0; for (int s = 0; s <= H.SG; ++s) { while (e < (int)ent.size() && ent[e].slab < s) ++e; wp[s] = e; } }
        for (int e = 0; e < nch * 64; ++e) {
          if (e < (int)ent.size()) {
            int run = 0; // following entries of the same (slab, row) inside this 64-entry chunk
            const int seg0 = wp[ent[e].slab]; // chunks of 64 are counted from the first entry of the slab
            while (e + run + 1 < (int)ent.size() && (e + run + 1 - seg0) / 64 == (e - seg0) / 64 && ent[e + run + 1].slab == ent[e].slab && ent[e + run + 1].row == ent[e].row) ++run;
            H.val.push_back(ent[e].v);
            H.pk.push_back((uint32_t)ent[e].col | ((uint32_t)ent[e].row << cb) | ((uint32_t)run << 26));
          } else { H.val.push_back(0.0); H.pk.push_back(0u); ++H.padded; }
        }
        rr = re;
      }
    }
  }
  return H;
}

// NW waves; KMAX chunks of 64 entries per wave and slab prefetched PF slabs ahead; XPT double2 per thread per slab (W = 2 * NW*64 * XPT)
template <int NW, int KMAX, int XPT, bool FILL_ONLY>
__global__ __launch_bounds__(NW * 64) void k_slab(SlabDev M, const double *__restrict__ x, double *__restrict__ y /* G partial vectors of R */) {
  extern __shared__ double smem[];
  double *xs = smem, *acc = smem + M.W;
  constexpr int NT = NW * 64, PF = 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware: blockIdx round-robins over the 8 XCDs; workgroups of the same column group share an XCD's L2 copy of that part of x
  const int g = (blockIdx.x % 8) % M.G, p = (blockIdx.x / 8) * (8 / M.G) + (blockIdx.x % 8) / M.G;
  if (p >= M.P) return;
  const size_t wi = ((size_t)p * M.G + g) * NW + w;
  const double *wval = M.val + M.wave_off[wi];
  const uint32_t *wpk = M.pk + M.wave_off[wi];
  const int row0 = M.part_row[p], nrows = M.part_row[p + 1] - row0;
  for (int r = tid; r < nrows; r += NT) acc[r] = 0.0;
  const int s0 = g * M.SG, s1 = min(M.S, s0 + M.SG), ns = s1 - s0;
  const int *wp = M.wptr + wi * (M.SG + 1);
  double2 xr[XPT];
  auto load_slab = [&](int s) { // unconditional loads (clamped), so that the outstanding-load count is static
    const long base = (long)s * M.W;
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      long i = base + 2 * (tid + NT * q);
      const bool in = i + 1 < M.C;
      if (!in) i = 0;
      xr[q] = *reinterpret_cast<const double2 *>(x + i);
      if (!in) { xr[q].x = 0.0; xr[q].y = 0.0; }
    }
  };
  double bv[PF][KMAX]; uint32_t bk[PF][KMAX];
  auto load_stream = [&](int q /* slab index inside the group, may run past the end: wptr is clamped by the host's slack */, double(&v)[KMAX], uint32_t(&k)[KMAX]) {
    const int lo = wp[min(q, ns)];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) { v[c] = wval[lo + 64 * c + lane]; k[c] = wpk[lo + 64 * c + lane]; }
  };
  load_slab(s0);
#pragma unroll
  for (int f = 0; f < PF; ++f) load_stream(f, bv[f], bk[f]);
  double chk = 0.0;
#pragma unroll 1
  for (int s = s0; s < s1; ++s) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < XPT; ++q) *reinterpret_cast<double2 *>(xs + 2 * (tid + NT * q)) = xr[q];
    __syncthreads();
    load_slab(min(s + 1, s1 - 1));
    if (FILL_ONLY) { chk += xs[(tid * 7) % M.W]; continue; }
    const int lo = wp[s - s0], cnt = wp[s - s0 + 1] - lo;
    auto body = [&](int c, double vv, uint32_t k) {
      const int e = 64 * c + lane;
      const bool act = e < cnt;
      const int col = k & ((1u << M.cb) - 1), row = (k >> M.cb) & ((1u << M.rb) - 1), run = act ? (int)(k >> 26) : 0;
      double t = act ? vv * xs[col] : 0.0;
#pragma unroll 1
      for (int o = 1; o < 64; o <<= 1) {
        if (!__any(run >= o)) break;
        const double u = __shfl_down(t, o, 64);
        if (run >= o) t += u;
      }
      const int prun = __shfl_up(run, 1, 64);
      if (act && (lane == 0 || prun == 0)) acc[row] += t;
    };
#pragma unroll
    for (int c = 0; c < KMAX; ++c)
      if (64 * c < cnt) body(c, bv[0][c], bk[0][c]);
    for (int c = KMAX; 64 * c < cnt; ++c) { // a slab with more entries than the prefetch holds
      const int e = min(lo + 64 * c + lane, lo + cnt - 1);
      body(c, wval[e], wpk[e]);
    }
#pragma unroll
    for (int f = 0; f + 1 < PF; ++f) {
#pragma unroll
      for (int c = 0; c < KMAX; ++c) { bv[f][c] = bv[f + 1][c]; bk[f][c] = bk[f + 1][c]; }
    }
    load_stream(s - s0 + PF, bv[PF - 1], bk[PF - 1]);
  }
  __syncthreads();
  if (FILL_ONLY) { if (chk == 12345.678) y[tid] = chk; return; }
  double *yp = y + (size_t)g * M.R + row0;
  for (int r = tid; r < nrows; r += NT) yp[r] = acc[r];
}

template <class K>
float time_kernel(K launch, int reps = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / reps;
}

template <class T> T *upload(const std::vector<T> &h) { T *d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

template <int NW, int KMAX, int XPT>
void run_case(const char *label, const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<double> &val, int R, int C, int G, int P, int RCAP = 4096) {
  const int W = 2 * NW * 64 * XPT;
  int cb = 0; while ((1 << cb) < W) ++cb;
  int rbits = 0; while ((1 << rbits) < RCAP) ++rbits;
  if (cb + rbits > 26) { printf("  %s: bit budget\n", label); return; }
  SlabHost H = build(ptr, idx, val, R, C, W, G, P, NW, KMAX + 1, RCAP, cb);
  if (!H.ok) { printf("  %-46s NW=%d W=%d G=%d P=%d: does not fit\n", label, NW, W, G, P); return; }
  SlabDev M; M.val = upload(H.val); M.pk = upload(H.pk); M.part_row = upload(H.part_row); M.wave_off = upload(H.wave_off); M.wave_nch = upload(H.wave_nch); M.wptr = upload(H.wptr);
  M.P = P; M.G = G; M.S = H.S; M.SG = H.SG; M.W = W; M.C = C; M.R = R; M.cb = cb; M.rb = rbits;
  std::vector<double> hx(C); std::mt19937_64 rng(7); for (auto &q : hx) q = (double)(rng() % 2001) / 1000.0 - 1.0;
  double *dx = upload(hx), *dy; CK(hipMalloc(&dy, (size_t)G * R * 8)); CK(hipMemset(dy, 0, (size_t)G * R * 8));
  const size_t lds = (size_t)(W + RCAP) * 8;
  if (lds > 163840) { printf("  %s: LDS\n", label); return; }
  auto kf = k_slab<NW, KMAX, XPT, false>; auto k0 = k_slab<NW, KMAX, XPT, true>;
  CK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = ((P * G + 7) / 8) * 8;
  const float t_fill = time_kernel([&] { hipLaunchKernelGGL(k0, dim3(grid), dim3(NW * 64), lds, 0, M, dx, dy); });
  const float t_full = time_kernel([&] { hipLaunchKernelGGL(kf, dim3(grid), dim3(NW * 64), lds, 0, M, dx, dy); });
  CK(hipGetLastError());
  std::vector<double> hy((size_t)G * R); CK(hipMemcpy(hy.data(), dy, hy.size() * 8, hipMemcpyDeviceToHost));
  double err = 0, nrm = 0;
  for (int r = 0; r < R; ++r) {
    double ref = 0; for (int k = ptr[r]; k < ptr[r + 1]; ++k) ref += val[k] * hx[idx[k]];
    double got = 0; for (int g = 0; g < G; ++g) got += hy[(size_t)g * R + r];
    err = std::max(err, std::fabs(got - ref)); nrm = std::max(nrm, std::fabs(ref));
  }
  int mxs = 0; for (size_t q = 0; q < H.wave_nch.size(); ++q) for (int t = 0; t < H.SG; ++t) mxs = std::max(mxs, H.wptr[q * (H.SG + 1) + t + 1] - H.wptr[q * (H.SG + 1) + t]);
  printf("  %-46s NW=%2d KMAX=%d (max entries per wave and slab %d) W=%5d G=%d P=%3d S=%2d pad %4.1f%%: fill-only %7.2f us, full %7.2f us, max err %.2e (|y| %.2e)\n", label, NW, KMAX, mxs, W, G, P, H.S,
         100.0 * H.padded / H.val.size(), t_fill, t_full, err, nrm);
  hipFree((void *)M.val); hipFree((void *)M.pk); hipFree((void *)M.part_row); hipFree((void *)M.wave_off); hipFree((void *)M.wave_nch); hipFree((void *)M.wptr); hipFree(dx); hipFree(dy);
}

int main() {
  const int m = 200000, n = 500000, per = 16;
  std::mt19937_64 rng(1);
  // At (CSC of A read as CSR): n rows, gathers an m-vector
  std::vector<int> ptrT(n + 1, 0), idxT; std::vector<double> valT;
  for (int j = 0; j < n; ++j) {
    if (j < m) { idxT.push_back(j); valT.push_back(1.0); }
    else { std::vector<int> r; while ((int)r.size() < per) { int q = rng() % m; if (std::find(r.begin(), r.end(), q) == r.end()) r.push_back(q); } std::sort(r.begin(), r.end());
      for (int q : r) { idxT.push_back(q); valT.push_back((double)(rng() % 2001) / 1000.0 - 1.0); } }
    ptrT[j + 1] = (int)idxT.size();
  }
  const long nnz = idxT.size();
  // A (explicit transpose): m rows, gathers an n-vector
  std::vector<int> ptrA(m + 1, 0), idxA(nnz); std::vector<double> valA(nnz);
  for (long k = 0; k < nnz; ++k) ptrA[idxT[k] + 1]++;
  for (int i = 0; i < m; ++i) ptrA[i + 1] += ptrA[i];
  { std::vector<int> pos(ptrA.begin(), ptrA.end() - 1);
    for (int j = 0; j < n; ++j) for (int k = ptrT[j]; k < ptrT[j + 1]; ++k) { const int q = pos[idxT[k]]++; idxA[q] = j; valA[q] = valT[k]; } }
  printf("C4-shaped matrix: m %d n %d nnz %ld\n", m, n, nnz);
  printf("A' z (500k rows, x = 1.6 MB)\n");
  run_case<16, 2, 6>("At G=1 P=256", ptrT, idxT, valT, n, m, 1, 256);
  run_case<16, 2, 4>("At G=1 P=256 (W 8192, 8192 rows)", ptrT, idxT, valT, n, m, 1, 256, 8192);
  run_case<16, 1, 4>("At G=1 P=256 (W 8192, 8192 rows)", ptrT, idxT, valT, n, m, 1, 256, 8192);
  run_case<8, 3, 12>("At G=1 P=256 (8 waves)", ptrT, idxT, valT, n, m, 1, 256);
  run_case<16, 2, 6>("At G=2 P=128", ptrT, idxT, valT, n, m, 2, 128);
  run_case<16, 1, 6>("At G=1 P=512", ptrT, idxT, valT, n, m, 1, 512);
  run_case<16, 2, 6>("At G=2 P=256 (512 wgs)", ptrT, idxT, valT, n, m, 2, 256);
  printf("A tmp (200k rows, x = 4 MB)\n");
  run_case<16, 1, 6>("A  G=1 P=256", ptrA, idxA, valA, m, n, 1, 256);
  run_case<16, 2, 6>("A  G=2 P=128", ptrA, idxA, valA, m, n, 2, 128);
  run_case<16, 3, 6>("A  G=4 P=64", ptrA, idxA, valA, m, n, 4, 64);
  run_case<16, 2, 6>("A  G=4 P=64", ptrA, idxA, valA, m, n, 4, 64);
  run_case<16, 2, 4>("A  G=4 P=64 (W 8192)", ptrA, idxA, valA, m, n, 4, 64);
  run_case<8, 4, 12>("A  G=4 P=64 (8 waves)", ptrA, idxA, valA, m, n, 4, 64);
  run_case<16, 3, 6>("A  G=8 P=64 (512 wgs)", ptrA, idxA, valA, m, n, 8, 64);
  run_case<16, 2, 6>("A  G=4 P=128 (512 wgs)", ptrA, idxA, valA, m, n, 4, 128);
  run_case<16, 3, 4>("A  G=8 P=32 (8192 rows)", ptrA, idxA, valA, m, n, 8, 32, 8192);
  return 0;
}
