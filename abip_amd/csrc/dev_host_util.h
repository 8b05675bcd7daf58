// dev_host_util.h -- host-side helpers shared by the LP and QCP drivers: device buffers, CSR / triangular-factor
// uploads, the launch plan of the level-scheduled triangular solves.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <ctime>
#include <vector>

#include "dev_common.h"
#include "dev_sptrsv.h"
#include "host_setup.h"

#define HIP_OK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      fprintf(stderr, "abip_hip: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
      return -1;                                                                                      \
    }                                                                                                 \
  } while (0)

namespace abip {
namespace hostutil {

inline double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }

template <class T>
struct DBuf { // device buffer
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t cnt) { n = cnt; if (!cnt) cnt = 1; HIP_OK(hipMalloc((void **)&p, cnt * sizeof(T))); return 0; }
  int upload(const std::vector<T> &h, hipStream_t s) {
    if (alloc(h.size())) return -1;
    if (!h.empty()) HIP_OK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// dev_transpose.hip: CSC -> CSR on the device (a stable sort of the entry numbers by row); 0 = done and synchronised
int dev_csc_to_csr(int nrows, int ncols, long nnz, const int *cp, const int *ri, const double *cx, int *out_ptr, int *out_col, double *out_val, hipStream_t s);

struct DevCsr {
  DBuf<int> ptr, idx, rbd; // rbd: 4 ints per row block, read as int4
  DBuf<double> val;
  DBuf<double> sval; DBuf<int> sidx, slen; DBuf<long> soff; // optional SELL-64 image
  int nrows = 0, nrb = 0, nslices = 0;
  int upload(const host::HostCsr &h, hipStream_t s) {
    nrows = h.nrows; nrb = (int)h.rb.size() - 1;
    std::vector<int> d4((size_t)4 * std::max(nrb, 1), 0);
    for (int q = 0; q < nrb; ++q) { d4[4 * q] = h.rb[q]; d4[4 * q + 1] = h.rb[q + 1]; d4[4 * q + 2] = h.ptr[h.rb[q]]; d4[4 * q + 3] = h.ptr[h.rb[q + 1]]; }
    if (ptr.upload(h.ptr, s) || idx.upload(h.idx, s) || val.upload(h.val, s) || rbd.upload(d4, s)) return -1;
    nslices = 0;
    if (!h.slen.empty()) {
      if (sval.upload(h.sval, s) || sidx.upload(h.sidx, s) || slen.upload(h.slen, s) || soff.upload(h.soff, s)) return -1;
      nslices = (int)h.slen.size();
    }
    return 0;
  }
  // the row form of an operator whose column form C (= the CSR of its transpose: ncols rows) is already resident: transposed on the device (dev_transpose.hip),
  // only the row pointers come back for the row blocks.  != 0: nothing kept, the caller builds and uploads the host's form instead
  int from_columns(const DevCsr &C, int nrows_, long nnz, int chunk, hipStream_t s) {
    nrows = nrows_; nslices = 0;
    if (ptr.alloc((size_t)nrows + 1) || idx.alloc((size_t)nnz) || val.alloc((size_t)nnz)) { release(); return -1; }
#ifdef ABIP_HIP_TEST_HOOKS
    if (getenv("ABIP_HIP_DEV_TRANSPOSE_FAIL")) { release(); return -1; } // as if the library sort (hipcub, dev_transpose.hip) had refused: the caller's host transpose takes over
#endif
    if (dev_csc_to_csr(nrows, C.nrows, nnz, C.ptr.p, C.idx.p, C.val.p, ptr.p, idx.p, val.p, s)) { release(); (void)hipGetLastError(); return -1; }
    host::HostCsr h;
    h.nrows = nrows; h.ptr.resize((size_t)nrows + 1);
    if (hipMemcpyAsync(h.ptr.data(), ptr.p, sizeof(int) * ((size_t)nrows + 1), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { release(); (void)hipGetLastError(); return -1; }
    host::build_row_blocks(h, chunk);
    nrb = (int)h.rb.size() - 1;
    std::vector<int> d4((size_t)4 * std::max(nrb, 1), 0);
    for (int q = 0; q < nrb; ++q) { d4[4 * q] = h.rb[q]; d4[4 * q + 1] = h.rb[q + 1]; d4[4 * q + 2] = h.ptr[h.rb[q]]; d4[4 * q + 3] = h.ptr[h.rb[q + 1]]; }
    if (rbd.upload(d4, s)) { release(); return -1; }
    return 0;
  }
  Csr view() const { return Csr{ptr.p, idx.p, val.p, (const int4 *)rbd.p, nrb, nrows, sval.p, sidx.p, soff.p, slen.p, nslices}; }
  void release() { ptr.release(); idx.release(); rbd.release(); val.release(); sval.release(); sidx.release(); slen.release(); soff.release(); nslices = 0; }
};

struct Segment { bool wide; int l0, l1; int rb0 = 0, nrb = 0; int a = 0, b = 0; double mean_len = 0.0; }; // wide: one level streamed over the grid, row blocks [rb0, rb0+nrb); positions [a, b), mean entries per row

struct DevTri {
  DBuf<int> ptr, idx, lev_ptr, lev_rows, lev_g, rbd;
  DBuf<double> val;
  int nlev = 0;
  std::vector<Segment> segs; // launch plan: runs of thin levels in one workgroup, wide levels as CSR-stream grids
  // all_thin: run every level inside one workgroup (small systems: the launch of a streamed level costs more than it saves)
  int upload(const host::TriHost &h, hipStream_t s, bool all_thin = false) {
    nlev = (int)h.lev_ptr.size() - 1;
    // a level is "wide" when it has enough rows or non-zeros to fill the chip; its positions are cut into row blocks
    // of <= CHUNK non-zeros / rows for the CSR-stream kernel (same greedy rule as host::build_row_blocks)
    segs.clear();
    std::vector<int> d4;
    int l = 0;
    auto is_wide = [&](int lv) {
      if (all_thin) return false;
      const int a = h.lev_ptr[lv], b = h.lev_ptr[lv + 1];
      return (b - a) >= 2048 || (h.ptr[b] - h.ptr[a]) >= 4096; // one workgroup chews ~1k non-zeros per microsecond at best
    };
    while (l < nlev) {
      if (is_wide(l)) {
        Segment sg{true, l, l + 1, (int)d4.size() / 4, 0};
        const int a = h.lev_ptr[l], b = h.lev_ptr[l + 1];
        int r = a;
        while (r < b) {
          int nn = 0, rows = 0, e = r;
          while (e < b) {
            const int len = h.ptr[e + 1] - h.ptr[e];
            if (rows > 0 && (nn + len > CHUNK || rows >= CHUNK)) break;
            nn += len; ++rows; ++e;
            if (nn > CHUNK) break;
          }
          d4.push_back(r); d4.push_back(e); d4.push_back(h.ptr[r]); d4.push_back(h.ptr[e]);
          r = e;
        }
        sg.nrb = (int)d4.size() / 4 - sg.rb0;
        sg.a = a; sg.b = b; sg.mean_len = (double)(h.ptr[b] - h.ptr[a]) / std::max(1, b - a);
        segs.push_back(sg);
        ++l;
      } else {
        int e = l;
        while (e < nlev && !is_wide(e)) ++e;
        segs.push_back(Segment{false, l, e, 0, 0});
        l = e;
      }
    }
    if (d4.empty()) d4.assign(4, 0);
    if (ptr.upload(h.ptr, s) || idx.upload(h.idx, s) || val.upload(h.val, s) || lev_ptr.upload(h.lev_ptr, s) ||
        lev_rows.upload(h.lev_rows, s) || lev_g.upload(h.lev_g, s) || rbd.upload(d4, s)) return -1;
    return 0;
  }
  bool single_workgroup() const { return segs.empty() || (segs.size() == 1 && !segs[0].wide); }
  Tri view() const { return Tri{ptr.p, idx.p, val.p, lev_ptr.p, lev_rows.p, lev_g.p, nlev}; }
  void release() { ptr.release(); idx.release(); val.release(); lev_ptr.release(); lev_rows.release(); lev_g.release(); rbd.release(); }
};

} // namespace hostutil
} // namespace abip
