// rocsparse_probe.hip -- CALIBRATION ONLY: the vendor's CSR SpMV on the matrices the C4 record is quoted on, for comparison with
// k_cg_spmv_A / k_cg_spmv_At (abip_amd/csrc/dev_kernels.h).  Never linked into the product; not part of the test suite.
//
//   tools/rocsparse_probe A.bin        A.bin = int64 rows, cols, nnz | int32 ptr[rows+1] | int32 idx[nnz] | double val[nnz]   (scripts/rocsparse_compare.py writes it)
//
// For every algorithm the generic API offers (adaptive = the default of hipSPARSE / PyTorch, rowsplit ("stream"), LRB, nnzsplit): analysis once, then the mean of 200
// y = A x over hipEvents, and the implied fraction of 8 TB/s on SURVEY 8(d)'s B_spmv bytes.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define RSCHK(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { fprintf(stderr, "%s: rocsparse status %d\n", #x, (int)s_); return -1.0; } } while (0)

#pragma clang diagnostic ignored "-Wdeprecated-declarations"

static double run(rocsparse_handle h, rocsparse_spmv_alg alg, long rows, long cols, long nnz, int *dptr, int *didx, double *dval, double *dx, double *dy, int reps) {
  rocsparse_spmat_descr A;
  rocsparse_dnvec_descr x, y;
  RSCHK(rocsparse_create_csr_descr(&A, rows, cols, nnz, dptr, didx, dval, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
  RSCHK(rocsparse_create_dnvec_descr(&x, cols, dx, rocsparse_datatype_f64_r));
  RSCHK(rocsparse_create_dnvec_descr(&y, rows, dy, rocsparse_datatype_f64_r));
  const double alpha = 1.0, beta = 0.0;
  size_t bytes = 0;
  RSCHK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
  void *buf = nullptr;
  HIPCHK(hipMalloc(&buf, bytes ? bytes : 8));
  RSCHK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_preprocess, &bytes, buf));
  for (int q = 0; q < 20; ++q) RSCHK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &bytes, buf));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0, 0));
  for (int q = 0; q < reps; ++q) RSCHK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &bytes, buf));
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  HIPCHK(hipFree(buf));
  rocsparse_destroy_spmat_descr(A); rocsparse_destroy_dnvec_descr(x); rocsparse_destroy_dnvec_descr(y);
  return 1e3 * ms / reps;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s A.bin\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  long hd[3];
  if (fread(hd, sizeof(long), 3, f) != 3) return 2;
  const long rows = hd[0], cols = hd[1], nnz = hd[2];
  std::vector<int> ptr(rows + 1), idx(nnz);
  std::vector<double> val(nnz), x(cols);
  if (fread(ptr.data(), 4, rows + 1, f) != (size_t)(rows + 1) || fread(idx.data(), 4, nnz, f) != (size_t)nnz || fread(val.data(), 8, nnz, f) != (size_t)nnz) return 2;
  fclose(f);
  for (long j = 0; j < cols; ++j) x[j] = 1.0 + 1e-3 * (double)(j % 97);
  int *dptr, *didx; double *dval, *dx, *dy;
  HIPCHK(hipMalloc((void **)&dptr, 4 * (rows + 1))); HIPCHK(hipMalloc((void **)&didx, 4 * nnz)); HIPCHK(hipMalloc((void **)&dval, 8 * nnz));
  HIPCHK(hipMalloc((void **)&dx, 8 * cols)); HIPCHK(hipMalloc((void **)&dy, 8 * rows));
  HIPCHK(hipMemcpy(dptr, ptr.data(), 4 * (rows + 1), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(didx, idx.data(), 4 * nnz, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dval, val.data(), 8 * nnz, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dx, x.data(), 8 * cols, hipMemcpyHostToDevice));
  rocsparse_handle h;
  if (rocsparse_create_handle(&h) != rocsparse_status_success) return 2;
  const double bytes = 12.0 * nnz + 4.0 * (rows + 1) + 8.0 * cols + 16.0 * rows; // SURVEY 8(d) B_spmv(R, C, nnz)
  printf("%s: %ld x %ld, %ld non-zeros; B_spmv = %.1f MB\n", argv[1], rows, cols, nnz, bytes / 1e6);
  const struct { rocsparse_spmv_alg a; const char *n; } algs[] = {{rocsparse_spmv_alg_default, "default"}, {rocsparse_spmv_alg_csr_adaptive, "csr_adaptive"},
      {rocsparse_spmv_alg_csr_rowsplit, "csr_rowsplit (stream)"}, {rocsparse_spmv_alg_csr_lrb, "csr_lrb"}, {rocsparse_spmv_alg_csr_nnzsplit, "csr_nnzsplit"}};
  for (auto &al : algs) {
    const double us = run(h, al.a, rows, cols, nnz, dptr, didx, dval, dx, dy, 200);
    if (us < 0) { printf("  %-24s not available\n", al.n); continue; }
    printf("  %-24s %8.2f us per y = A x   %6.1f GB/s of B_spmv   %.3f of 8 TB/s\n", al.n, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
  }
  rocsparse_destroy_handle(h);
  return 0;
}
