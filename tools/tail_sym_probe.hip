// tail_sym_probe -- the dense tail's symmetric mat-vec (abip_amd/csrc/dev_tail.h) on its own: correctness against a naive kernel and the time of
// every (rows requested together, wavefronts per SIMD, wavefronts of the stream) variant.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/tail_sym_probe tools/tail_sym_probe.hip ;  tools/tail_sym_probe [T] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../abip_amd/csrc/dev_tail.h"
using namespace abip;
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(double *M, int T, long ld) { // lower triangle pseudo-random in [-1, 1) / T, NaN above the diagonal (must never be used)
  for (long r = blockIdx.x; r < T; r += gridDim.x)
    for (long c = threadIdx.x; c < ld; c += blockDim.x) {
      unsigned long long h = (unsigned long long)(r * 1000003ll + c) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
      M[r * ld + c] = c <= r ? ((double)(h >> 11) / 9007199254740992.0 * 2.0 - 1.0) / T : __longlong_as_double(0x7ff8000000000000ll);
    }
}
__global__ void k_naive(const double *M, long ld, int T, const double *w, double *x) { // one workgroup per output
  __shared__ double sm[256];
  const int i = blockIdx.x;
  double s = 0.0;
  for (int c = threadIdx.x; c <= i; c += 256) s += M[(long)i * ld + c] * w[c];
  for (int r = i + 1 + threadIdx.x; r < T; r += 256) s += M[(long)r * ld + i] * w[r];
  sm[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) x[i] = sm[0];
}
// yardsticks: (a) a plain contiguous read of `bytes` (16 bytes per lane, 8 loads in flight, grid-stride); (b) the tail kernel's own ADDRESS PATTERN without its arithmetic
// (every wavefront walks down a 4 KB-wide column chunk, rows ld * 8 bytes apart, 16 loads in flight) -- what the memory system gives this pattern at best
__global__ __launch_bounds__(256) void k_read_contig(const double2 *__restrict__ p, long n2, double *out) {
  double a = 0.0;
  const long stride = (long)gridDim.x * 256 * 8;
  for (long i = (long)blockIdx.x * 256 * 8 + threadIdx.x; i + 7 * 256 < n2; i += stride) {
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[i + k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) a += v[k].x + v[k].y;
  }
  if (a == 123.456) out[0] = a;
}
__global__ __launch_bounds__(256, 2) void k_read_pattern(const double *__restrict__ M, int ld, const SymArgs sa, double *out) {
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (q >= sa.nwv) return;
  int u = (int)((long)q * sa.nu / sa.nwv);
  const int u1 = (int)((long)(q + 1) * sa.nu / sa.nwv);
  int cc = 0;
  while (sa.pre[cc + 1] <= u) ++cc;
  double a = 0.0;
  while (u < u1) {
    const int c0 = cc * SYC, uend = min(u1, sa.pre[cc + 1]);
    int r = c0 + SYU * (u - sa.pre[cc]);
    const int rend = c0 + SYU * (uend - sa.pre[cc]);
    for (; r < rend; r += 4) {
      double2 v[16];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 *row2 = reinterpret_cast<const double2 *>(M + (long)(r + i) * ld + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * i + k] = row2[min(64 * k + lane, (r + i - c0) >> 1)];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) a += v[k].x;
    }
    u = uend; ++cc;
  }
  if (a == 123.456) out[0] = a;
}
template <class F>
void yard(const char *name, double bytes, int reps, F launch) {
  hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
  float tot = 0.f, best = 1e30f;
  for (int it = 0; it < reps + 2; ++it) {
    OK(hipEventRecord(a)); launch(); OK(hipEventRecord(b)); OK(hipEventSynchronize(b));
    float ms; OK(hipEventElapsedTime(&ms, a, b));
    if (it >= 2) { tot += ms; best = std::min(best, ms); }
  }
  printf("%-58s avg %7.2f us  best %7.2f us  = %6.0f GB/s avg / %6.0f best (%.3f of 8 TB/s)\n", name, 1e3 * tot / reps, 1e3 * best, bytes / (tot / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9, bytes / (tot / reps * 1e-3) / 1e9 / 8000.0);
  fflush(stdout);
}

template <int MINW>
void run(const char *name, const double *M, int T, const double *w, double *rowpart, double *colpart, double *x, const Ctl *ctl, int waves, const std::vector<double> &ref, int reps) {
  SymPlan pl;
  if (!pl.make(T, waves)) { printf("%s: no plan\n", name); return; }
  const SymArgs sa = pl.args();
  hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
  float best = 1e30f, tot = 0.f, fin = 0.f;
  for (int it = 0; it < reps + 2; ++it) {
    OK(hipEventRecord(a));
    hipLaunchKernelGGL((k_tail_sym<MINW>), dim3(pl.nwv / 4), dim3(256), 0, 0, M, T, w, rowpart, colpart, sa, ctl);
    OK(hipEventRecord(b));
    OK(hipEventSynchronize(b));
    float ms; OK(hipEventElapsedTime(&ms, a, b));
    OK(hipEventRecord(a));
    hipLaunchKernelGGL(k_tail_sym_fin, dim3(T / 64), dim3(1024), 0, 0, (const double *)rowpart, (const double *)colpart, sa, x, ctl, (double *)nullptr, (const double *)nullptr, 0);
    OK(hipEventRecord(b));
    OK(hipEventSynchronize(b));
    float ms2; OK(hipEventElapsedTime(&ms2, a, b));
    if (it >= 2) { best = std::min(best, ms); tot += ms; fin += ms2; }
  }
  std::vector<double> hx(T);
  OK(hipMemcpy(hx.data(), x, sizeof(double) * T, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (int i = 0; i < T; ++i) { num += (hx[i] - ref[i]) * (hx[i] - ref[i]); den += ref[i] * ref[i]; }
  const double bytes = 4.0 * T * (T + 1.0);
  printf("%-22s waves %5d (units %d): avg %7.2f us  best %7.2f us  = %6.0f GB/s avg / %6.0f best (%.3f of 8 TB/s);  fin %5.2f us;  rel err %.2e\n", name, pl.nwv, pl.nu, 1e3 * tot / reps,
         1e3 * best, bytes / (tot / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9, bytes / (tot / reps * 1e-3) / 1e9 / 8000.0, 1e3 * fin / reps, std::sqrt(num / den));
  fflush(stdout);
}
int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 10048, reps = argc > 2 ? atoi(argv[2]) : 30;
  const long ld = T;
  double *M, *w, *rowpart, *colpart, *x, *xr; Ctl *ctl;
  OK(hipMalloc(&M, sizeof(double) * ld * T)); OK(hipMalloc(&w, sizeof(double) * T)); OK(hipMalloc(&x, sizeof(double) * T)); OK(hipMalloc(&xr, sizeof(double) * T));
  OK(hipMalloc(&rowpart, sizeof(double) * 64 * T)); OK(hipMalloc(&colpart, sizeof(double) * (16384 + 64) * SYC)); OK(hipMalloc(&ctl, sizeof(Ctl)));
  OK(hipMemset(ctl, 0, sizeof(Ctl)));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, M, T, ld);
  std::vector<double> hw(T), ref(T);
  for (int i = 0; i < T; ++i) hw[i] = std::sin(0.37 * i) + 0.25;
  OK(hipMemcpy(w, hw.data(), sizeof(double) * T, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_naive, dim3(T), dim3(256), 0, 0, (const double *)M, ld, T, (const double *)w, xr);
  OK(hipMemcpy(ref.data(), xr, sizeof(double) * T, hipMemcpyDeviceToHost));
  printf("T = %d: lower triangle %.1f MB\n", T, 4.0 * T * (T + 1.0) / 1e6);
  {
    const double tri = 4.0 * T * (T + 1.0);
    const long n2 = (long)(tri / 16.0);
    for (int grid : {1024, 2048, 4096})
      yard(("yardstick: contiguous read of the same bytes, grid " + std::to_string(grid)).c_str(), tri, reps, [&] { hipLaunchKernelGGL(k_read_contig, dim3(grid), dim3(256), 0, 0, (const double2 *)M, n2, x); });
    for (int waves : {2048, 4096}) {
      SymPlan pl; pl.make(T, waves);
      const SymArgs sa = pl.args();
      yard(("yardstick: the tail's address pattern, loads only, waves " + std::to_string(pl.nwv)).c_str(), tri, reps, [&] { hipLaunchKernelGGL(k_read_pattern, dim3(pl.nwv / 4), dim3(256), 0, 0, (const double *)M, T, sa, x); });
    }
  }
  for (int waves : {1024, 1280, 1536, 2048}) {
    run<2>("2 waves/SIMD", M, T, w, rowpart, colpart, x, ctl, waves, ref, reps);

  }
  return 0;
}
