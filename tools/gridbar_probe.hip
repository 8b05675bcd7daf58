// tools/gridbar_probe.hip -- developer micro-benchmark: what does a grid-wide barrier cost on this part?
// One counter in global memory, every workgroup's thread 0 does release-fence + atomicAdd and polls (bounded) until all arrived.
// Compared with the cost of ending a kernel and starting the next one (empty kernels back to back on one stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

__global__ __launch_bounds__(256) void k_rounds(unsigned *cnt, int rounds, double *sink, int *bailed, int sleep) {
  const unsigned G = gridDim.x;
  double acc = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    acc = acc * 1.0000001 + 1.0; // token work
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(cnt, 1u);
      const unsigned want = (unsigned)(r + 1) * G;
      long spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (sleep) __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000) { *bailed = 1; break; } // never hang the box
      }
    }
    __syncthreads();
  }
  if (acc == -1.0) sink[0] = acc;
}
__global__ void k_empty(double *sink) { if (sink == nullptr) sink[0] = 1; }

int main() {
  unsigned *cnt; double *sink; int *bailed;
  CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&bailed, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int rounds = 2000;
  for (int sleep : {0, 1})
    for (int G : {32, 64, 128, 256, 512, 1024}) {
      CK(hipMemset(cnt, 0, 4)); CK(hipMemset(bailed, 0, 4));
      hipLaunchKernelGGL(k_rounds, dim3(G), dim3(256), 0, 0, cnt, 10, sink, bailed, sleep); // warm
      CK(hipDeviceSynchronize());
      CK(hipMemset(cnt, 0, 4));
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_rounds, dim3(G), dim3(256), 0, 0, cnt, rounds, sink, bailed, sleep);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      int hb = 0; CK(hipMemcpy(&hb, bailed, 4, hipMemcpyDeviceToHost));
      printf("grid barrier, %4d workgroups, s_sleep %d: %7.2f us per round%s\n", G, sleep, ms * 1000.f / rounds, hb ? "  (BAILED: not all workgroups resident?)" : "");
    }
  CK(hipEventRecord(a));
  for (int r = 0; r < 2000; ++r) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("empty kernel boundary (256 workgroups back to back): %7.2f us per kernel\n", ms * 1000.f / 2000);
  // the same 2000 kernels as ONE hipGraph launch
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int r = 0; r < 2000; ++r) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  CK(hipGraphLaunch(exec, st));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  printf("the same 2000 kernels captured in one hipGraph:            %7.2f us per kernel\n", ms * 1000.f / 2000);
  for (int nodes : {3, 12, 48}) { // small graphs launched back to back: the per-launch cost
    hipGraph_t g2; hipGraphExec_t e2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < nodes; ++r) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink);
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e2, st)); CK(hipStreamSynchronize(st));
    const int reps = 1200 / nodes;
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(e2, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%d-kernel hipGraph launched %d times back to back: %7.2f us per kernel (%.2f us per graph)\n", nodes, reps, ms * 1000.f / (reps * nodes), ms * 1000.f / reps);
  }
  for (int G : {1, 16}) {
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 2000; ++r) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, st, sink);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("empty kernel boundary (%d workgroup(s) back to back): %7.2f us per kernel\n", G, ms * 1000.f / 2000);
  }
  return 0;
}
