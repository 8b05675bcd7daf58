// host_par.h -- a few host threads for the once-per-init passes over the matrix (scaling, transposes, materialising an operator): VERDICT r3 item 7.
// A small pool of detached workers (created at first use, ABIP_HIP_HOST_THREADS caps it, default min(hardware threads, 16)); a region is a set of
// contiguous index ranges, one per worker.  Everything parallelised with it is ORDER-PRESERVING: a range owns its outputs (columns, or rows through a row
// map), every sum runs over the same elements in the same order as the sequential code -- the results are bit-identical whatever the thread count
// (tests/test_host_factor_cpu.py, tests/test_qcp_host_cpu.py run with 1 and with several threads).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace abip {
namespace host {

class Pool {
 public:
  static Pool &get() { static Pool *p = new Pool(); return *p; } // (never destroyed: its workers are detached and may outlive static destruction)
  int threads() const { return nthreads_; }
  // run fn(t) for t = 0 .. T-1 (T <= threads()), the caller takes t = 0; returns when all are done
  void run(int T, const std::function<void(int)> &fn) {
    if (T <= 1) { fn(0); return; }
    std::unique_lock<std::mutex> region(region_mutex_); // one region at a time (several works may set up concurrently)
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; want_ = T; pending_ = T - 1; ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  Pool() {
    const char *e = getenv("ABIP_HIP_HOST_THREADS");
    int v = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads_ = std::max(1, std::min(v, 64));
    for (int t = 1; t < nthreads_; ++t) std::thread([this, t] { work(t); }).detach();
  }
  void work(int t) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)> *fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (t < want_) fn = fn_;
      }
      if (fn) {
        (*fn)(t);
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int nthreads_ = 1;
  std::mutex m_, region_mutex_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  int want_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
};

// (tests: ABIP_HIP_HOST_GRAIN_DIV divides every grain, so that small matrices take the threaded paths too)
inline long par_grain(long g) { static const long div = [] { const char *e = getenv("ABIP_HIP_HOST_GRAIN_DIV"); const long v = e ? atol(e) : 1; return v < 1 ? 1L : v; }(); return std::max<long>(1, g / div); }
// fn(lo, hi, t) over [0, n) cut into equal parts; fewer than `grain` items per part: fewer parts (one: inline)
template <class F>
inline void par_ranges(long n, long grain, F fn) {
  const int T = (int)std::max<long>(1, std::min<long>(Pool::get().threads(), n / par_grain(grain)));
  if (T <= 1) { fn(0L, n, 0); return; }
  Pool::get().run(T, [&](int t) { fn(n * t / T, n * (t + 1) / T, t); });
}
// the same with the number of parts GIVEN (T <= par_threads()): for callers that size per-part tables first -- going through a grain again may come out
// with a different count (ABIP_HIP_HOST_GRAIN_DIV on a tiny matrix: ADVICE r4) and index past the tables
template <class F>
inline void par_ranges_T(long n, int T, F fn) {
  T = std::max(1, std::min(T, Pool::get().threads()));
  if (T <= 1) { fn(0L, n, 0); return; }
  Pool::get().run(T, [&](int t) { fn(n * t / T, n * (t + 1) / T, t); });
}
// the same for the columns (rows) of a compressed matrix, cut so that every part holds about the same number of entries: ptr has n + 1 entries
template <class P, class F>
inline void par_by_entries(const P *ptr, long n, long grain_entries, F fn) {
  const long nnz = (long)ptr[n] - (long)ptr[0];
  const int T = (int)std::max<long>(1, std::min<long>(Pool::get().threads(), nnz / par_grain(grain_entries)));
  if (T <= 1 || n < T) { fn(0L, n, 0); return; }
  std::vector<long> cut(T + 1, n);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    const long target = (long)ptr[0] + nnz * t / T;
    cut[t] = std::max<long>(cut[t - 1], std::lower_bound(ptr, ptr + n + 1, (P)target) - ptr);
    if (cut[t] > n) cut[t] = n;
  }
  Pool::get().run(T, [&](int t) { if (cut[t] < cut[t + 1]) fn(cut[t], cut[t + 1], t); });
}
inline int par_threads() { return Pool::get().threads(); }

// A stable bucket pass over the entries of a compressed matrix (ncols columns, cp): entry q of column j goes to bucket key(j, q) in [0, nb); inside a bucket the
// entries keep the order of the sequential double loop (j ascending, q ascending).  out_ptr (nb + 1) receives the bucket extents, emit(dst, j, q) stores
// entry q at slot dst.  Threads own column ranges for the counts and for the scatter (a bucket's slots are dealt to the threads in column order), so no two
// threads write the same slot and the result is the sequential counting sort's, entry for entry.
template <class PI, class Key, class Emit>
inline void par_bucket(long ncols, const PI *cp, long nb, std::vector<int> &out_ptr, long grain, Key key, Emit emit) {
  const long nnz = (long)cp[ncols] - (long)cp[0];
  out_ptr.assign(nb + 1, 0);
  const int T = (int)std::max<long>(1, std::min<long>(std::min<long>(par_threads(), ncols), nnz / par_grain(grain)));
  if (T <= 1) {
    for (long j = 0; j < ncols; ++j) for (long q = cp[j]; q < (long)cp[j + 1]; ++q) out_ptr[key(j, q) + 1]++;
    for (long i = 0; i < nb; ++i) out_ptr[i + 1] += out_ptr[i];
    std::vector<int> pos(out_ptr.begin(), out_ptr.end() - 1);
    for (long j = 0; j < ncols; ++j) for (long q = cp[j]; q < (long)cp[j + 1]; ++q) emit((long)pos[key(j, q)]++, j, q);
    return;
  }
  std::vector<long> cut(T + 1, ncols);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) cut[t] = std::max<long>(cut[t - 1], std::lower_bound(cp, cp + ncols + 1, (PI)((long)cp[0] + nnz * t / T)) - cp);
  std::vector<std::vector<int>> cnt(T); // entries of bucket i among thread t's columns
  Pool::get().run(T, [&](int t) { cnt[t].assign(nb, 0); for (long j = cut[t]; j < cut[t + 1]; ++j) for (long q = cp[j]; q < (long)cp[j + 1]; ++q) cnt[t][key(j, q)]++; });
  par_ranges(nb, 1 << 16, [&](long lo, long hi, int) { for (long i = lo; i < hi; ++i) { int s = 0; for (int t = 0; t < T; ++t) { const int c = cnt[t][i]; cnt[t][i] = s; s += c; } out_ptr[i + 1] = s; } }); // -> offset of thread t inside bucket i
  for (long i = 0; i < nb; ++i) out_ptr[i + 1] += out_ptr[i];
  Pool::get().run(T, [&](int t) {
    std::vector<int> &off = cnt[t];
    for (long j = cut[t]; j < cut[t + 1]; ++j)
      for (long q = cp[j]; q < (long)cp[j + 1]; ++q) { const long i = key(j, q); emit((long)out_ptr[i] + off[i]++, j, q); }
  });
}

// CSC (ncols columns, row indices ri, nrows rows) -> CSR: out_ptr (nrows + 1), out_col, out_val, entries of a row in ascending column order -- what the
// sequential counting sort produces, entry for entry.
template <class PI, class RI>
inline void par_transpose(long nrows, long ncols, const PI *cp, const RI *ri, const double *cx, std::vector<int> &out_ptr, std::vector<int> &out_col, std::vector<double> &out_val) {
  const long nnz = (long)cp[ncols];
  out_col.resize(nnz); out_val.resize(nnz);
  par_bucket(ncols, cp, nrows, out_ptr, 2000000, [&](long, long q) { return (long)ri[q]; }, [&](long dst, long j, long q) { out_col[dst] = (int)j; out_val[dst] = cx[q]; });
}

} // namespace host
} // namespace abip
