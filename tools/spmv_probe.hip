// tools/spmv_probe.hip -- developer micro-benchmark (not part of the library): where does the C4 SpMV time go?
// Builds the C4-shaped CSC-as-CSR matrix (200k one-entry rows + 300k 16-entry rows, random columns < 200k) and times
//   V0  pure stream of val+idx (no gather), per-thread partial sums
//   V1  stream + gather x[idx] (8 B from a 1.6 MB vector), per-thread partial sums (no row structure)
//   V2  stream + gather with sorted-by-line indices inside each wave (upper bound for locality)
//   V3  gather only (idx stream + gather)
// with 8-byte and 16-byte per-lane loads and several grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

template <int MODE, int W>  // W = elements per lane per load (1: 8B val/4B idx, 2: 16B/8B)
__global__ __launch_bounds__(256) void probe(const double* __restrict__ val, const int* __restrict__ idx, const double* __restrict__ x, long nnz, double* out) {
  double acc = 0.0;
  const long stride = (long)gridDim.x * 256 * W;
  for (long k = ((long)blockIdx.x * 256 + threadIdx.x) * W; k + W <= nnz; k += stride) {
    if (W == 1) {
      double a = (MODE == 3) ? 1.0 : val[k];
      int c = idx[k];
      acc += (MODE == 0) ? a * (double)c : a * x[c];
    } else {
      double2 a = (MODE == 3) ? make_double2(1.0, 1.0) : *reinterpret_cast<const double2*>(val + k);
      int2 c = *reinterpret_cast<const int2*>(idx + k);
      acc += (MODE == 0) ? a.x * (double)c.x + a.y * (double)c.y : a.x * x[c.x] + a.y * x[c.y];
    }
  }
  // unrolled x4 variant is left to the compiler; keep it simple
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

template <int MODE, int W, int U>  // U independent loads in flight per thread
__global__ __launch_bounds__(256) void probe_u(const double* __restrict__ val, const int* __restrict__ idx, const double* __restrict__ x, long nnz, double* out) {
  double acc = 0.0;
  const long tile = (long)256 * W * U;
  for (long base = (long)blockIdx.x * tile; base + tile <= nnz; base += (long)gridDim.x * tile) {
    double a[U * W]; int c[U * W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long k = base + ((long)u * 256 + threadIdx.x) * W;
#pragma unroll
      for (int w = 0; w < W; ++w) { a[u * W + w] = (MODE == 3) ? 1.0 : val[k + w]; c[u * W + w] = idx[k + w]; }
    }
#pragma unroll
    for (int q = 0; q < U * W; ++q) acc += (MODE == 0) ? a[q] * (double)c[q] : a[q] * x[c[q]];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

// V5: a real SpMV y = At z + beta y on the same matrix, CSR-vector form: VW lanes per row, U rows in flight per lane group,
// no LDS, no workgroup barrier
template <int VW, int U>
__global__ __launch_bounds__(256) void spmv_vec(const int* __restrict__ ptr, const int* __restrict__ idx, const double* __restrict__ val,
                                                const double* __restrict__ x, double* __restrict__ y, int nrows, double beta) {
  const int lane = threadIdx.x % VW;
  const long ngrp = (long)gridDim.x * 256 / VW;
  for (long r0 = ((long)blockIdx.x * 256 + threadIdx.x) / VW; r0 < nrows; r0 += ngrp * U) {
    int s[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long r = r0 + u * ngrp; s[u] = r < nrows ? ptr[r] : 0; e[u] = r < nrows ? ptr[r + 1] : 0; }
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc[u] = 0.0; for (int k = s[u] + lane; k < e[u]; k += VW) acc[u] += val[k] * x[idx[k]]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      for (int off = VW >> 1; off > 0; off >>= 1) acc[u] += __shfl_xor(acc[u], off, 64);
      const long r = r0 + u * ngrp;
      if (lane == 0 && r < nrows) y[r] = acc[u] + beta * y[r];
    }
  }
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

int main() {
  const int m = 200000, n = 500000, per = 16;
  std::mt19937_64 rng(1);
  std::vector<int> idx; std::vector<double> val;
  for (int j = 0; j < m; ++j) { idx.push_back(j); val.push_back(1.0); }
  for (int j = m; j < n; ++j) { std::vector<int> r(per); for (auto& q : r) q = rng() % m; std::sort(r.begin(), r.end()); for (int q : r) { idx.push_back(q); val.push_back(0.5); } }
  const long nnz = idx.size();
  std::vector<int> idx_sorted(idx);
  for (long k = 0; k + 64 <= nnz; k += 64) std::sort(idx_sorted.begin() + k, idx_sorted.begin() + k + 64);
  std::vector<int> idx_seq(nnz); for (long k = 0; k < nnz; ++k) idx_seq[k] = (int)(k % m);
  std::vector<int> idx_big(nnz); for (long k = 0; k < nnz; ++k) idx_big[k] = (int)(rng() % n); // gather target of n doubles (4 MB): the A tmp product
  double *dv, *dx, *dout, *dxb; int *di, *dis, *diq, *dib;
  CK(hipMalloc(&dv, nnz * 8)); CK(hipMalloc(&di, nnz * 4)); CK(hipMalloc(&dis, nnz * 4)); CK(hipMalloc(&diq, nnz * 4)); CK(hipMalloc(&dx, (size_t)m * 8)); CK(hipMalloc(&dout, 1 << 20));
  CK(hipMalloc(&dib, nnz * 4)); CK(hipMalloc(&dxb, (size_t)n * 8)); CK(hipMemcpy(dib, idx_big.data(), nnz * 4, hipMemcpyHostToDevice));
  { std::vector<double> hb(n, 1.0); CK(hipMemcpy(dxb, hb.data(), (size_t)n * 8, hipMemcpyHostToDevice)); }
  CK(hipMemcpy(dv, val.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(di, idx.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dis, idx_sorted.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(diq, idx_seq.data(), nnz * 4, hipMemcpyHostToDevice));
  std::vector<double> hx(m, 1.0); CK(hipMemcpy(dx, hx.data(), (size_t)m * 8, hipMemcpyHostToDevice));
  std::vector<int> ptr(n + 1, 0);
  for (int j = 0; j < n; ++j) ptr[j + 1] = ptr[j] + (j < m ? 1 : per);
  int *dp; double *dy;
  CK(hipMalloc(&dp, (size_t)(n + 1) * 4)); CK(hipMalloc(&dy, (size_t)n * 8)); CK(hipMemset(dy, 0, (size_t)n * 8));
  CK(hipMemcpy(dp, ptr.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
  printf("nnz %ld  stream bytes %.1f MB\n", nnz, nnz * 12 / 1e6);
  for (int grid : {2048, 8192, 32768}) {
    printf("grid %d\n", grid);
#define RUN(label, kern, ip) printf("  %-34s %8.2f us\n", label, time_kernel([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dv, ip, dx, nnz, dout); }));
    RUN("V0 stream only, 8B loads", (probe<0, 1>), di)
    RUN("V0 stream only, 16B loads", (probe<0, 2>), di)
    RUN("V0 stream only, 8B x4 in flight", (probe_u<0, 1, 4>), di)
    RUN("V0 stream only, 16B x4 in flight", (probe_u<0, 2, 4>), di)
    RUN("V1 stream+gather random, 8B", (probe<1, 1>), di)
    RUN("V1 stream+gather random, 8B x4", (probe_u<1, 1, 4>), di)
    RUN("V1 stream+gather random, 16B x4", (probe_u<1, 2, 4>), di)
    RUN("V1 stream+gather random, 8B x8", (probe_u<1, 1, 8>), di)
    RUN("V2 gather wave-sorted idx, 8B x4", (probe_u<1, 1, 4>), dis)
    RUN("V2' gather sequential idx, 8B x4", (probe_u<1, 1, 4>), diq)
    RUN("V3 idx+gather only, 8B x4", (probe_u<3, 1, 4>), di)
#define RUNB(label, kern) printf("  %-34s %8.2f us\n", label, time_kernel([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dv, dib, dxb, nnz, dout); }));
    RUNB("V1b stream+gather from 4 MB, 8B x4", (probe_u<1, 1, 4>))
    RUNB("V1b stream+gather from 4 MB, 8B x8", (probe_u<1, 1, 8>))
    RUNB("V3b idx+gather from 4 MB only, x4", (probe_u<3, 1, 4>))
#define RUNV(label, kern) printf("  %-34s %8.2f us\n", label, time_kernel([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dp, di, dv, dx, dy, n, 0.5); }));
    RUNV("V5 CSR-vector 16 lanes/row, U=1", (spmv_vec<16, 1>))
    RUNV("V5 CSR-vector 16 lanes/row, U=2", (spmv_vec<16, 2>))
    RUNV("V5 CSR-vector 16 lanes/row, U=4", (spmv_vec<16, 4>))
    RUNV("V5 CSR-vector 8 lanes/row, U=4", (spmv_vec<8, 4>))
    RUNV("V5 CSR-vector 4 lanes/row, U=4", (spmv_vec<4, 4>))
  }
  return 0;
}
