// tools/sell_probe.hip -- developer micro-benchmark (not part of the library): the C4-shaped products in a sliced-ELL layout
// (SELL-64-sigma): 64 consecutive (length-sorted inside a window of SIGMA rows) rows form a slice stored column-major, one lane
// owns one row, no LDS, no shuffles, no barriers; compared with the structure-free stream+gather floor of spmv_probe.hip (24.8 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

struct Sell { const double *val; const int *idx; const long *soff; const int *slen /* steps of 64 entries */; const int *perm /* row of each (slice, lane / T) */; int nslices; };

template <int T, int U>
__global__ __launch_bounds__(256) void k_sell(Sell M, const double *__restrict__ x, double *__restrict__ y, double beta) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < M.nslices; s += nw) {
    const long base = M.soff[s];
    const int len = M.slen[s];
    const int row = M.perm[(long)s * (64 / T) + lane / T];
    const double *v = M.val + base + lane;
    const int *ix = M.idx + base + lane;
    double acc = 0.0;
    int k = 0;
    for (; k + U <= len; k += U) {
      double a[U]; int c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = v[(long)(k + u) * 64]; c[u] = ix[(long)(k + u) * 64]; }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += a[u] * x[c[u]];
    }
    if (k < len) {
      double a[U]; int c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = 0.0; c[u] = 0; if (k + u < len) { a[u] = v[(long)(k + u) * 64]; c[u] = ix[(long)(k + u) * 64]; } }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += a[u] * x[c[u]];
    }
#pragma unroll
    for (int o = T >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row >= 0 && (lane % T) == 0) y[row] = acc + beta * y[row];
  }
}

struct SellHost { std::vector<double> val; std::vector<int> idx, slen, perm; std::vector<long> soff; long padded = 0; };
static SellHost build(const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<double> &val, int R, int sigma, int T) {
  SellHost H;
  const int RS = 64 / T; // rows per slice
  std::vector<int> order(R);
  std::iota(order.begin(), order.end(), 0);
  for (int w0 = 0; w0 < R; w0 += sigma) {
    const int w1 = std::min(R, w0 + sigma);
    std::stable_sort(order.begin() + w0, order.begin() + w1, [&](int a, int b) { return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b]; });
  }
  const int ns = (R + RS - 1) / RS;
  H.slen.resize(ns); H.soff.resize(ns); H.perm.assign((size_t)ns * RS, -1);
  for (int s = 0; s < ns; ++s) {
    int len = 0;
    for (int l = 0; l < RS && s * RS + l < R; ++l) { const int r = order[s * RS + l]; H.perm[(size_t)s * RS + l] = r; len = std::max(len, ptr[r + 1] - ptr[r]); }
    const int steps = (len + T - 1) / T;
    H.slen[s] = steps; H.soff[s] = (long)H.val.size();
    for (int j = 0; j < steps; ++j)
      for (int l = 0; l < 64; ++l) {
        const int rl = l / T, t = l % T, k = j * T + t;
        const int r = s * RS + rl < R ? order[s * RS + rl] : -1;
        if (r >= 0 && ptr[r] + k < ptr[r + 1]) { H.val.push_back(val[ptr[r] + k]); H.idx.push_back(idx[ptr[r] + k]); }
        else { H.val.push_back(0.0); H.idx.push_back(r >= 0 && ptr[r + 1] > ptr[r] ? idx[ptr[r + 1] - 1] : 0); ++H.padded; }
      }
  }
  return H;
}
template <class T> T *upload(const std::vector<T> &h) { T *d; CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
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

template <int T>
void run(const char *label, const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<double> &val, int R, int C, int sigma) {
  SellHost H = build(ptr, idx, val, R, sigma, T);
  Sell M; M.val = upload(H.val); M.idx = upload(H.idx); M.soff = upload(H.soff); M.slen = upload(H.slen); M.perm = upload(H.perm); M.nslices = (int)H.slen.size();
  std::vector<double> hx(C); std::mt19937_64 rng(7); for (auto &q : hx) q = (double)(rng() % 2001) / 1000.0 - 1.0;
  double *dx = upload(hx), *dy; CK(hipMalloc(&dy, (size_t)R * 8)); CK(hipMemset(dy, 0, (size_t)R * 8));
  int mxl = 0; for (int q : H.slen) mxl = std::max(mxl, q);
  printf("  %s T=%d sigma %d: %d slices, max steps %d, stored %zu (padding %.1f%%)\n", label, T, sigma, M.nslices, mxl, H.val.size(), 100.0 * H.padded / H.val.size());
  for (int grid : {1024, 2048, 4096}) {
    const float t4 = time_kernel([&] { hipLaunchKernelGGL((k_sell<T, 4>), dim3(grid), dim3(256), 0, 0, M, dx, dy, 0.0); });
    const float t8 = time_kernel([&] { hipLaunchKernelGGL((k_sell<T, 8>), dim3(grid), dim3(256), 0, 0, M, dx, dy, 0.0); });
    const float t16 = time_kernel([&] { hipLaunchKernelGGL((k_sell<T, 16>), dim3(grid), dim3(256), 0, 0, M, dx, dy, 0.0); });
    printf("    grid %5d: U=4 %7.2f us  U=8 %7.2f us  U=16 %7.2f us\n", grid, t4, t8, t16);
  }
  std::vector<double> hy(R); CK(hipMemcpy(hy.data(), dy, (size_t)R * 8, hipMemcpyDeviceToHost));
  double err = 0;
  for (int r = 0; r < R; ++r) { double ref = 0; for (int k = ptr[r]; k < ptr[r + 1]; ++k) ref += val[k] * hx[idx[k]]; err = std::max(err, std::fabs(ref - hy[r])); }
  printf("    max err %.2e\n", err);
  hipFree((void *)M.val); hipFree((void *)M.idx); hipFree((void *)M.soff); hipFree((void *)M.slen); hipFree((void *)M.perm); hipFree(dx); hipFree(dy);
}

int main() {
  const int m = 200000, n = 500000, per = 16;
  std::mt19937_64 rng(1);
  std::vector<int> ptrT(n + 1, 0), idxT; std::vector<double> valT;
  for (int j = 0; j < n; ++j) {
    if (j < m) { idxT.push_back(j); valT.push_back(1.0); }
    else { std::vector<int> r; while ((int)r.size() < per) { int q = rng() % m; if (std::find(r.begin(), r.end(), q) == r.end()) r.push_back(q); } std::sort(r.begin(), r.end());
      for (int q : r) { idxT.push_back(q); valT.push_back((double)(rng() % 2001) / 1000.0 - 1.0); } }
    ptrT[j + 1] = (int)idxT.size();
  }
  const long nnz = idxT.size();
  std::vector<int> ptrA(m + 1, 0), idxA(nnz); std::vector<double> valA(nnz);
  for (long k = 0; k < nnz; ++k) ptrA[idxT[k] + 1]++;
  for (int i = 0; i < m; ++i) ptrA[i + 1] += ptrA[i];
  { std::vector<int> pos(ptrA.begin(), ptrA.end() - 1);
    for (int j = 0; j < n; ++j) for (int k = ptrT[j]; k < ptrT[j + 1]; ++k) { const int q = pos[idxT[k]]++; idxA[q] = j; valA[q] = valT[k]; } }
  printf("C4-shaped matrix: m %d n %d nnz %ld\n", m, n, nnz);
  run<1>("A' z (500k rows, x 1.6 MB)", ptrT, idxT, valT, n, m, 64);
  run<2>("A' z (500k rows, x 1.6 MB)", ptrT, idxT, valT, n, m, 64);
  run<4>("A' z (500k rows, x 1.6 MB)", ptrT, idxT, valT, n, m, 64);
  run<1>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 200000);
  run<2>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 200000);
  run<4>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 200000);
  run<4>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 2048);
  run<8>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 200000);
  run<8>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 2048);
  run<16>("A tmp (200k rows, x 4 MB)", ptrA, idxA, valA, m, n, 2048);
  { // the same A product with the columns folded into [0, m): is the 4 MB gather target (vs 1.6 MB) what costs?
    std::vector<int> idxF(idxA); for (auto &q : idxF) q %= m;
    run<4>("A tmp, columns folded mod m (x 1.6 MB)", ptrA, idxF, valA, m, m, 2048);
    run<8>("A tmp, columns folded mod m (x 1.6 MB)", ptrA, idxF, valA, m, m, 2048);
  }
  return 0;
}
