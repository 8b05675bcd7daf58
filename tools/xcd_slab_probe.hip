// xcd_slab_probe.hip -- developer probe, not part of the library: does an XCD-LOCAL gather pay on the C4-shaped product y = A x?
//
// Today (k_cg_spmv_A): every workgroup gathers from the whole 4 MB n-vector; each XCD's 4 MB L2 holds that vector AND sees the 60 MB matrix stream go by,
// L2 hit rate 0.74-0.88 (profiles/r02b_pmc_counters.json).  Here the columns are cut into 8 slabs, one per XCD: the workgroups of XCD k multiply, for ALL rows,
// only the non-zeros whose column lies in slab k (a CSR matrix of its own per slab), so an XCD's L2 has to hold 0.5 MB of x; the 8 partial row sums go to
// memory and a second kernel adds them in XCD order (fixed order: deterministic).  Price: 8 row-pointer arrays instead of 1 and 2 x 8 x 8m bytes of partials.
// Both forms run through the library's own row-block SpMV (dev_common.h spmv_stream), same row blocks of <= 1024 non-zeros.
//
//   tools/xcd_slab_probe [m n per_col]      default 200000 500000 16 (the C4 generator's shape: identity block + per_col random rows per further column)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../abip_amd/csrc/dev_common.h"
using namespace abip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct HostCsr { int rows = 0; std::vector<int> ptr, idx; std::vector<double> val; std::vector<int> rbd; int nrb = 0; };
static void row_blocks(HostCsr &M) { // as host_setup.cpp build_row_blocks: <= CHUNK non-zeros and rows per block, a longer row alone
  std::vector<int> rb{0};
  int r = 0;
  while (r < M.rows) {
    int nn = 0, rows = 0, e = r;
    while (e < M.rows) { const int len = M.ptr[e + 1] - M.ptr[e]; if (rows > 0 && (nn + len > CHUNK || rows >= CHUNK)) break; nn += len; ++rows; ++e; if (nn > CHUNK) break; }
    rb.push_back(e); r = e;
  }
  M.nrb = (int)rb.size() - 1;
  M.rbd.resize(4 * (size_t)M.nrb);
  for (int q = 0; q < M.nrb; ++q) { M.rbd[4 * q] = rb[q]; M.rbd[4 * q + 1] = rb[q + 1]; M.rbd[4 * q + 2] = M.ptr[rb[q]]; M.rbd[4 * q + 3] = M.ptr[rb[q + 1]]; }
}
struct DevCsr { int *ptr, *idx, *rbd; double *val; int nrb, rows; };
static DevCsr upload(const HostCsr &M) {
  DevCsr d; d.nrb = M.nrb; d.rows = M.rows;
  CK(hipMalloc((void **)&d.ptr, 4 * M.ptr.size())); CK(hipMalloc((void **)&d.idx, 4 * std::max<size_t>(M.idx.size(), 1))); CK(hipMalloc((void **)&d.val, 8 * std::max<size_t>(M.val.size(), 1)));
  CK(hipMalloc((void **)&d.rbd, 4 * std::max<size_t>(M.rbd.size(), 4)));
  CK(hipMemcpy(d.ptr, M.ptr.data(), 4 * M.ptr.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d.idx, M.idx.data(), 4 * M.idx.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d.val, M.val.data(), 8 * M.val.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d.rbd, M.rbd.data(), 4 * M.rbd.size(), hipMemcpyHostToDevice));
  return d;
}
static Csr view(const DevCsr &d) { return Csr{d.ptr, d.idx, d.val, (const int4 *)d.rbd, d.nrb, d.rows, nullptr, nullptr, nullptr, nullptr, 0}; }

__global__ __launch_bounds__(BS, 8) void k_base(Csr A, const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double lds[CHUNK]; __shared__ int lptr[CHUNK + 1]; __shared__ double sm[WAVES];
  spmv_stream<1>(A, lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; }, [&](int r, double(&acc)[1]) { y[r] = acc[0]; }, [] { return true; });
}
struct Slabs { Csr M[8]; };
// workgroup b works on slab (b mod 8) -- the dispatcher deals workgroups round-robin over the XCDs, so that is its XCD (counted: `off` = workgroups for which it is not)
__global__ __launch_bounds__(BS, 8) void k_slab(Slabs S, const double *__restrict__ x, double *__restrict__ part, int m, int *off) {
  __shared__ double lds[CHUNK]; __shared__ int lptr[CHUNK + 1]; __shared__ double sm[WAVES];
  const int k = blockIdx.x & 7;
  if (threadIdx.x == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); if ((int)(xcc & 0xf) != k) atomicAdd(off, 1); }
  double *yk = part + (size_t)k * m;
  spmv_stream<1>(S.M[k], lds, lptr, sm, [&](int c, double a, double(&pr)[1]) { pr[0] = a * x[c]; }, [&](int r, double(&acc)[1]) { yk[r] = acc[0]; }, [] { return true; },
                 (int)(blockIdx.x >> 3), (int)(gridDim.x >> 3));
}
__global__ __launch_bounds__(BS) void k_sum8(const double *__restrict__ part, double *__restrict__ y, int m) {
  for (int i = blockIdx.x * BS + threadIdx.x; i < m; i += gridDim.x * BS) {
    double s = part[i];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[(size_t)k * m + i];
    y[i] = s;
  }
}

int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 200000, n = argc > 2 ? atoi(argv[2]) : 500000, pc = argc > 3 ? atoi(argv[3]) : 16;
  // the C4 generator's shape: columns 0..m-1 identity, each further column pc distinct random rows
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ull; };
  std::vector<std::vector<std::pair<int, double>>> rows(m);
  for (int i = 0; i < m; ++i) rows[i].push_back({i, 1.0});
  for (int j = m; j < n; ++j) {
    std::vector<int> picked;
    while ((int)picked.size() < pc) { const int r = (int)(rnd() % (uint64_t)m); if (std::find(picked.begin(), picked.end(), r) == picked.end()) picked.push_back(r); }
    for (int r : picked) rows[r].push_back({j, (double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1});
  }
  HostCsr A; A.rows = m; A.ptr.assign(m + 1, 0);
  for (int i = 0; i < m; ++i) { std::sort(rows[i].begin(), rows[i].end()); A.ptr[i + 1] = A.ptr[i] + (int)rows[i].size(); }
  A.idx.resize(A.ptr[m]); A.val.resize(A.ptr[m]);
  for (int i = 0; i < m; ++i) for (size_t q = 0; q < rows[i].size(); ++q) { A.idx[A.ptr[i] + q] = rows[i][q].first; A.val[A.ptr[i] + q] = rows[i][q].second; }
  row_blocks(A);
  const long nnz = A.ptr[m];
  HostCsr Sl[8];
  const int w = (n + 7) / 8;
  for (int k = 0; k < 8; ++k) {
    Sl[k].rows = m; Sl[k].ptr.assign(m + 1, 0);
    for (int i = 0; i < m; ++i) { int c = 0; for (auto &e : rows[i]) c += (e.first / w == k); Sl[k].ptr[i + 1] = Sl[k].ptr[i] + c; }
    Sl[k].idx.resize(Sl[k].ptr[m]); Sl[k].val.resize(Sl[k].ptr[m]);
    for (int i = 0; i < m; ++i) { int p = Sl[k].ptr[i]; for (auto &e : rows[i]) if (e.first / w == k) { Sl[k].idx[p] = e.first; Sl[k].val[p] = e.second; ++p; } }
    row_blocks(Sl[k]);
  }
  std::vector<double> x(n), yref(m, 0.0);
  for (int j = 0; j < n; ++j) x[j] = 1.0 + 1e-3 * (j % 97);
  for (int i = 0; i < m; ++i) { double acc = 0; for (auto &e : rows[i]) acc += e.second * x[e.first]; yref[i] = acc; }
  DevCsr dA = upload(A); Slabs S; DevCsr dS[8];
  for (int k = 0; k < 8; ++k) { dS[k] = upload(Sl[k]); S.M[k] = view(dS[k]); }
  double *dx, *dy, *dpart; int *doff;
  CK(hipMalloc((void **)&dx, 8 * (size_t)n)); CK(hipMalloc((void **)&dy, 8 * (size_t)m)); CK(hipMalloc((void **)&dpart, 8 * (size_t)m * 8)); CK(hipMalloc((void **)&doff, 4));
  CK(hipMemcpy(dx, x.data(), 8 * (size_t)n, hipMemcpyHostToDevice)); CK(hipMemset(doff, 0, 4));
  const int NBLK = 2048, reps = 200;
  auto check = [&](const char *what) {
    std::vector<double> y(m); CK(hipMemcpy(y.data(), dy, 8 * (size_t)m, hipMemcpyDeviceToHost));
    double num = 0, den = 0; for (int i = 0; i < m; ++i) { num += (y[i] - yref[i]) * (y[i] - yref[i]); den += yref[i] * yref[i]; }
    printf("  %-34s relative error %.2e\n", what, std::sqrt(num / den));
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn) { for (int q = 0; q < 20; ++q) fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); for (int q = 0; q < reps; ++q) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / reps; };
  printf("y = A x, A %d x %d, %ld non-zeros (C4 shape), row blocks of <= %d non-zeros: %d for A, %d .. per slab\n", m, n, nnz, CHUNK, A.nrb, Sl[0].nrb);
  const double t_base = timeit([&] { hipLaunchKernelGGL(k_base, dim3(NBLK), dim3(BS), 0, 0, view(dA), (const double *)dx, dy); });
  check("one product over all columns:");
  const double t_slab = timeit([&] { hipLaunchKernelGGL(k_slab, dim3(NBLK), dim3(BS), 0, 0, S, (const double *)dx, dpart, m, doff); });
  const double t_sum = timeit([&] { hipLaunchKernelGGL(k_sum8, dim3(512), dim3(BS), 0, 0, (const double *)dpart, dy, m); });
  const double t_both = timeit([&] { hipLaunchKernelGGL(k_slab, dim3(NBLK), dim3(BS), 0, 0, S, (const double *)dx, dpart, m, doff); hipLaunchKernelGGL(k_sum8, dim3(512), dim3(BS), 0, 0, (const double *)dpart, dy, m); });
  check("8 column slabs + sum in XCD order:");
  int off = 0; CK(hipMemcpy(&off, doff, 4, hipMemcpyDeviceToHost));
  const double bytes = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * n + 16.0 * m;
  printf("  one product over all columns   %7.2f us   (%.0f GB/s of B_spmv = %.1f MB)\n", t_base, bytes / t_base / 1e3, bytes / 1e6);
  printf("  8 XCD-local column slabs       %7.2f us  + sum of the partials %5.2f us;  back to back %7.2f us  -> %+.1f %% against the one product\n", t_slab, t_sum, t_both, 100.0 * (t_base - t_both) / t_base);
  printf("  workgroups that were NOT on the XCD of their slab: %d of %d launches x %d\n", off, 3 * reps + 60, NBLK);
  return 0;
}
