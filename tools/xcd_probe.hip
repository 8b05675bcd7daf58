// tools/xcd_probe.hip -- developer micro-benchmark (go / no-go for a single-XCD persistent ADMM kernel).
//
// Question: what does one "exchange phase" cost when all participating workgroups sit on ONE XCD (one L2)?  A phase = every
// workgroup computes its slice of an N-vector from the whole previous vector (held in LDS), publishes the slice, and gathers the
// whole new vector back into LDS.  That is the all-gather a sharded SpMV / dense mat-vec needs between dependent steps.
//
//   census        : where do 256 one-per-CU workgroups land (XCC id, SE/SH/CU)?
//   exchange/bar  : plain|sc1 stores, vmcnt(0), workgroup barrier, one atomic arrive (agent or workgroup scope), sc1 poll,
//                   then loads of flavour plain|sc1|nt|sys
//   exchange/tag  : 16-byte granules {lo32, tag, hi32, tag} written by one dwordx4 store, consumers poll the data itself
//
// Every variant is verified bit for bit against a host model (a stale read changes the result) and bails out on a spin limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 1024;
constexpr int NMAX = 4096; // vector length bound (doubles)

__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__device__ __forceinline__ unsigned hw_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }

__global__ __launch_bounds__(TB) void k_census(unsigned *out) {
  extern __shared__ double dyn[];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc_id(); out[2 * blockIdx.x + 1] = hw_id(); }
  if (dyn[threadIdx.x] == 123.456) out[0] = 0; // keep the LDS allocation
}

struct Sync { unsigned tickets; unsigned pad0[31]; unsigned arrive; unsigned pad1[31]; unsigned bailed; unsigned nranks; };

// rank of this workgroup among the G participants on XCD `target` (or among all workgroups when target < 0); -1: not taking part
__device__ __forceinline__ int take_rank(Sync *sy, int target, int G, int *sh) {
  if (threadIdx.x == 0) {
    int r = -1;
    if (target < 0 || (int)xcc_id() == target) {
      r = (int)atomicAdd(&sy->tickets, 1u);
      if (r >= G) r = -1;
    }
    *sh = r;
  }
  __syncthreads();
  return *sh;
}

__device__ __forceinline__ double model(const double *x, int i, int N) { return 0.5 * x[(int)(((long)i * 7 + 3) % N)] + 0.25 * x[i] + 1.0; }

template <int LD> __device__ __forceinline__ double ld8(const double *p) {
  if (LD == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (LD == 2) return __builtin_nontemporal_load(p);
  if (LD == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return *(const volatile double *)p;
}
template <int ST> __device__ __forceinline__ void st8(double *p, double v) {
  if (ST == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (ST == 2) __builtin_nontemporal_store(v, p);
  else *(volatile double *)p = v;
}

// ---- variant "bar": counter barrier + flavoured loads -------------------------------------------------------------
// FENCE: 0 none (single-XCD hypothesis: L2 is the coherence point), 1 agent release before arrive + agent acquire after the poll
template <int LD, int ST, int WGSCOPE, int FENCE>
__global__ __launch_bounds__(TB) void k_bar(Sync *sy, double *buf /* 2*NMAX */, double *result, int N, int G, int phases, int target) {
  extern __shared__ double dyn[];
  double *x = dyn; // N
  __shared__ int sh;
  const int r = take_rank(sy, target, G, &sh);
  if (r < 0) return;
  const int per = (N + G - 1) / G, i0 = r * per, i1 = min(N, i0 + per);
  for (int i = threadIdx.x; i < N; i += TB) x[i] = (double)(i % 17) * 0.125;
  __syncthreads();
  for (int p = 0; p < phases; ++p) {
    double *b = buf + (size_t)(p & 1) * NMAX;
    for (int i = i0 + threadIdx.x; i < i1; i += TB) st8<ST>(b + i, model(x, i, N));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (WGSCOPE) __hip_atomic_fetch_add(&sy->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&sy->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(p + 1) * (unsigned)G;
      long spins = 0;
      while (__hip_atomic_load(&sy->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 2000000) { sy->bailed = 1; break; }
      }
      if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (__hip_atomic_load(&sy->bailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    double t[NMAX / TB];
#pragma unroll
    for (int u = 0; u < NMAX / TB; ++u) { const int i = threadIdx.x + u * TB; t[u] = (i < N) ? ld8<LD>(b + i) : 0.0; }
#pragma unroll
    for (int u = 0; u < NMAX / TB; ++u) { const int i = threadIdx.x + u * TB; if (i < N) x[i] = t[u]; }
    __syncthreads();
  }
  if (r == 0) for (int i = threadIdx.x; i < N; i += TB) result[i] = x[i];
}

// ---- variant "tag": data-tagged 16-byte granules ---------------------------------------------------------------------
template <int SC1LD> __device__ __forceinline__ void ld16x4(const u32x4 *p0, const u32x4 *p1, const u32x4 *p2, const u32x4 *p3, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
  if (SC1LD)
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %4, off nt\n\tglobal_load_dwordx4 %1, %5, off nt\n\tglobal_load_dwordx4 %2, %6, off nt\n\tglobal_load_dwordx4 %3, %7, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
template <int SC1ST> __device__ __forceinline__ void st16(u32x4 *p, u32x4 v) {
  if (SC1ST) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ bool tag_ok(const u32x4 &g, unsigned tag) { return g.y == tag && g.w == tag; }
__device__ __forceinline__ double unpack(const u32x4 &g) { return __hiloint2double((int)g.z, (int)g.x); }

template <int SC1LD, int SC1ST>
__global__ __launch_bounds__(TB) void k_tag(Sync *sy, u32x4 *buf /* 2*NMAX granules */, double *result, int N, int G, int phases, int target) {
  extern __shared__ double dyn[];
  double *x = dyn;
  __shared__ int sh;
  __shared__ int bail;
  const int r = take_rank(sy, target, G, &sh);
  if (r < 0) return;
  if (threadIdx.x == 0) bail = 0;
  const int per = (N + G - 1) / G, i0 = r * per, i1 = min(N, i0 + per);
  for (int i = threadIdx.x; i < N; i += TB) x[i] = (double)(i % 17) * 0.125;
  __syncthreads();
  for (int p = 0; p < phases; ++p) {
    u32x4 *b = buf + (size_t)(p & 1) * NMAX;
    const unsigned tag = (unsigned)(p + 1);
    for (int i = i0 + threadIdx.x; i < i1; i += TB) {
      const double v = model(x, i, N);
      u32x4 g; g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
      st16<SC1ST>(b + i, g);
    }
    __syncthreads(); // everybody has read x of the previous phase before it is overwritten below
    const int j0 = threadIdx.x, j1 = threadIdx.x + TB, j2 = threadIdx.x + 2 * TB, j3 = threadIdx.x + 3 * TB;
    const bool n0 = j0 < N, n1 = j1 < N, n2 = j2 < N, n3 = j3 < N;
    u32x4 a, bb, c, d;
    long spins = 0;
    for (;;) {
      ld16x4<SC1LD>(b + (n0 ? j0 : 0), b + (n1 ? j1 : 0), b + (n2 ? j2 : 0), b + (n3 ? j3 : 0), a, bb, c, d);
      const bool ok = (!n0 || tag_ok(a, tag)) && (!n1 || tag_ok(bb, tag)) && (!n2 || tag_ok(c, tag)) && (!n3 || tag_ok(d, tag));
      if (ok) break;
      if (++spins > 200000) { bail = 1; sy->bailed = 1; break; }
    }
    if (n0) x[j0] = unpack(a);
    if (n1) x[j1] = unpack(bb);
    if (n2) x[j2] = unpack(c);
    if (n3) x[j3] = unpack(d);
    __syncthreads();
    if (bail) return;
  }
  if (r == 0) for (int i = threadIdx.x; i < N; i += TB) result[i] = x[i];
}

// host model
static void host_model(std::vector<double> &x, int N, int phases) {
  std::vector<double> y(N);
  for (int i = 0; i < N; ++i) x[i] = (double)(i % 17) * 0.125;
  for (int p = 0; p < phases; ++p) {
    for (int i = 0; i < N; ++i) y[i] = 0.5 * x[(int)(((long)i * 7 + 3) % N)] + 0.25 * x[i] + 1.0;
    x.swap(y);
  }
}

struct Ctx { Sync *sy; void *buf; double *res; hipEvent_t a, b; hipStream_t st; };

int main(int argc, char **argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  Ctx c;
  CK(hipMalloc(&c.sy, sizeof(Sync))); CK(hipMalloc(&c.buf, 2 * NMAX * 16)); CK(hipMalloc(&c.res, NMAX * 8));
  CK(hipEventCreate(&c.a)); CK(hipEventCreate(&c.b)); CK(hipStreamCreate(&c.st));
  const size_t lds = 100 * 1024;

  // ---- census ----
  {
    unsigned *out; CK(hipMalloc(&out, 8 * 1024));
    CK(hipFuncSetAttribute((const void *)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int grid : {256, 64}) {
      hipLaunchKernelGGL(k_census, dim3(grid), dim3(TB), lds, c.st, out);
      CK(hipStreamSynchronize(c.st));
      std::vector<unsigned> h(2 * grid); CK(hipMemcpy(h.data(), out, 8 * grid, hipMemcpyDeviceToHost));
      int cnt[16] = {0};
      for (int b = 0; b < grid; ++b) cnt[h[2 * b] & 15]++;
      printf("census grid %d (1 WG per CU by LDS): per-XCC counts:", grid);
      for (int q = 0; q < 8; ++q) printf(" %d", cnt[q]);
      printf("   first 16 blocks -> xcc:");
      for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b]);
      printf("\n");
    }
    // CU-masked streams: which mask confines a launch to one XCD?
    for (int scheme = 0; scheme < 2; ++scheme) {
      for (int k = 0; k < 8; ++k) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int bit = 0; bit < 256; ++bit) {
          const bool on = scheme == 0 ? (bit / 32 == k) : (bit % 8 == k);
          if (on) mask[bit / 32] |= 1u << (bit % 32);
        }
        hipStream_t ms;
        hipError_t e = hipExtStreamCreateWithCUMask(&ms, 8, mask);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(k_census, dim3(32), dim3(TB), lds, ms, out);
        CK(hipStreamSynchronize(ms));
        std::vector<unsigned> h(64); CK(hipMemcpy(h.data(), out, 8 * 32, hipMemcpyDeviceToHost));
        int cnt[16] = {0};
        for (int b = 0; b < 32; ++b) cnt[h[2 * b] & 15]++;
        printf("CU mask scheme %s k=%d: 32 WGs per-XCC counts:", scheme == 0 ? "bits[32k,32k+32)" : "bits = k mod 8", k);
        for (int q = 0; q < 8; ++q) printf(" %d", cnt[q]);
        printf("\n");
        CK(hipStreamDestroy(ms));
      }
    }
    CK(hipFree(out));
  }

  const int phases = (argc > 1) ? atoi(argv[1]) : 2000;
  auto bench = [&](const char *name, auto kern, int N, int G, int target, int grid) {
    std::vector<double> ref(N); host_model(ref, N, phases);
    float best = 1e30f; int bad = 0; unsigned bailed = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(c.sy, 0, sizeof(Sync), c.st));
      CK(hipMemsetAsync(c.buf, 0, 2 * NMAX * 16, c.st));
      CK(hipMemsetAsync(c.res, 0, NMAX * 8, c.st));
      CK(hipEventRecord(c.a, c.st));
      kern(grid, lds, c.st, N, G, phases, target);
      CK(hipEventRecord(c.b, c.st)); CK(hipEventSynchronize(c.b));
      float ms; CK(hipEventElapsedTime(&ms, c.a, c.b));
      Sync hs; CK(hipMemcpy(&hs, c.sy, sizeof(Sync), hipMemcpyDeviceToHost));
      bailed |= hs.bailed;
      std::vector<double> got(N); CK(hipMemcpy(got.data(), c.res, N * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < N; ++i) bad += (memcmp(&got[i], &ref[i], 8) != 0);
      if (ms < best) best = ms;
    }
    printf("%-44s N %4d G %2d %s: %7.3f us per phase%s%s\n", name, N, G, target >= 0 ? "one XCD " : "all XCDs", best * 1000.f / phases,
           bailed ? "  BAILED" : "", bad ? "  MISMATCH (stale reads)" : "  exact");
    fflush(stdout);
  };
#define BAR(LD, ST, WGS, FE) [&](int grid, size_t l, hipStream_t s, int N, int G, int ph, int tg) { CK(hipFuncSetAttribute((const void *)k_bar<LD, ST, WGS, FE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l)); hipLaunchKernelGGL((k_bar<LD, ST, WGS, FE>), dim3(grid), dim3(TB), l, s, c.sy, (double *)c.buf, c.res, N, G, ph, tg); }
#define TAG(L, S) [&](int grid, size_t l, hipStream_t s, int N, int G, int ph, int tg) { CK(hipFuncSetAttribute((const void *)k_tag<L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l)); hipLaunchKernelGGL((k_tag<L, S>), dim3(grid), dim3(TB), l, s, c.sy, (u32x4 *)c.buf, c.res, N, G, ph, tg); }
  for (int N : {2696, 1024}) {
    for (int G : {32, 16, 8}) {
      bench("bar: fences (release/acquire agent), plain", BAR(0, 0, 0, 1), N, G, 0, 256);
      bench("bar: st plain, ld sc1, atomic agent", BAR(1, 0, 0, 0), N, G, 0, 256);
      bench("bar: st sc1,   ld sc1, atomic agent", BAR(1, 1, 0, 0), N, G, 0, 256);
      bench("bar: st plain, ld nt,  atomic agent", BAR(2, 0, 0, 0), N, G, 0, 256);
      bench("bar: st plain, ld sys, atomic agent", BAR(3, 0, 0, 0), N, G, 0, 256);
      bench("bar: st plain, ld sc1, atomic workgroup", BAR(1, 0, 1, 0), N, G, 0, 256);
      bench("bar: st plain, ld plain (expected stale)", BAR(0, 0, 0, 0), N, G, 0, 256);
      bench("tag: st sc1,   ld sc1", TAG(1, 1), N, G, 0, 256);
      bench("tag: st plain, ld sc1", TAG(1, 0), N, G, 0, 256);
      bench("tag: st plain, ld nt", TAG(0, 0), N, G, 0, 256);
      bench("tag: st sc1,   ld nt", TAG(0, 1), N, G, 0, 256);
    }
    // the same protocols across XCDs (no gating): what the cross-XCD variant costs
    bench("bar: fences, plain [cross-XCD]", BAR(0, 0, 0, 1), N, 32, -1, 32);
    bench("bar: st sc1, ld sc1 [cross-XCD]", BAR(1, 1, 0, 0), N, 32, -1, 32);
    bench("tag: st sc1, ld sc1 [cross-XCD]", TAG(1, 1), N, 32, -1, 32);
  }
  return 0;
}
