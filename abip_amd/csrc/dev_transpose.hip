// dev_transpose.hip -- the row form of a large operator built ON the device from its column form, instead of a host counting sort + a second upload
// (set-up of the conic path: the LASSO protocol's operator has 22.5 M entries, the host transposes them in ~0.1 s and uploads 270 MB more).
//
//   CSC (cp, ri, cx), already resident  ->  CSR (out_ptr, out_col, out_val), the entries of a row in ascending column order
//
// which is exactly what the host's stable counting sort (host_par.h: par_transpose; the reference: indirect.c:81-139) produces: a STABLE sort of the entry
// numbers q by row index keeps q ascending inside a row, and q ascending is column ascending.  The sort is rocPRIM's radix sort through hipcub (least
// significant digit first, stable by construction) over the bits the row indices need; the rest are three small kernels.  Plumbing, not a hot kernel: it runs
// once per set-up.  tests/test_gpu_qcp.py::test_device_transpose_equals_the_host compares it with the host form entry for entry.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace abip {
namespace hostutil {

namespace {
__global__ __launch_bounds__(256) void k_iota(int *v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] = (int)i;
}
// out_ptr[i] = first position of the sorted keys that is >= i  (i = 0 .. nrows)
__global__ __launch_bounds__(256) void k_row_ptr(const int *keys, long n, int nrows, int *out_ptr) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i <= nrows; i += (long)gridDim.x * blockDim.x) {
    long lo = 0, hi = n;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (keys[mid] < (int)i) lo = mid + 1; else hi = mid; }
    out_ptr[i] = (int)lo;
  }
}
// entry k of the row form = entry q = order[k] of the column form; its column = the j with cp[j] <= q < cp[j + 1]
__global__ __launch_bounds__(256) void k_gather(const int *order, long n, const int *cp, int ncols, const double *cx, int *out_col, double *out_val) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int q = order[k];
    int lo = 0, hi = ncols; // invariant: cp[lo] <= q < cp[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cp[mid] <= q) lo = mid; else hi = mid; }
    out_col[k] = lo;
    out_val[k] = cx[q];
  }
}
} // namespace

// 0 = done (enqueued on s and synchronised: the scratch is freed before returning); != 0: nothing usable was written, the caller takes the host route
int dev_csc_to_csr(int nrows, int ncols, long nnz, const int *cp, const int *ri, const double *cx, int *out_ptr, int *out_col, double *out_val, hipStream_t s) {
  if (nnz <= 0 || nnz >= (1L << 31) || nrows <= 0 || ncols <= 0) return -1;
  int bits = 1;
  while ((1L << bits) < (long)nrows) ++bits;
  int *keys = nullptr, *qin = nullptr, *qout = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  auto drop = [&]() { if (keys) (void)hipFree(keys); if (qin) (void)hipFree(qin); if (qout) (void)hipFree(qout); if (tmp) (void)hipFree(tmp); };
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ri, keys, qin, qout, (int)nnz, 0, bits, s) != hipSuccess) return -2;
  if (hipMalloc((void **)&keys, sizeof(int) * nnz) != hipSuccess || hipMalloc((void **)&qin, sizeof(int) * nnz) != hipSuccess || hipMalloc((void **)&qout, sizeof(int) * nnz) != hipSuccess ||
      hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)) != hipSuccess) { drop(); (void)hipGetLastError(); return -3; }
  const int grid = (int)std::min<long>(4096, (nnz + 255) / 256);
  hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, s, qin, nnz);
  if (hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, ri, keys, qin, qout, (int)nnz, 0, bits, s) != hipSuccess) { drop(); (void)hipGetLastError(); return -4; }
  hipLaunchKernelGGL(k_row_ptr, dim3(std::max(1, std::min(1024, (nrows + 256) / 256))), dim3(256), 0, s, (const int *)keys, nnz, nrows, out_ptr);
  hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, (const int *)qout, nnz, cp, ncols, cx, out_col, out_val);
  const bool ok = hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
  drop();
  return ok ? 0 : -5;
}

} // namespace hostutil
} // namespace abip
