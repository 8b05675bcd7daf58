// dist_internal.h -- the multi-GPU context of libabip_hip.so as the two solvers see it (solver.hip owns it: RCCL bound with dlopen, or the
// host callback of the tests; include/abip_hip.h "multi-GPU").  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace abip {
struct DistInfo { int kind, rank, world; }; // kind 0: none, 1: RCCL, 2: host callback
DistInfo dist_info();
// In-place sum over the ranks of `count` doubles at device pointer `buf`, ordered on stream `s` (the callback transport stages through
// `hstage` and synchronises the stream).  0 on success; every rank must make the same sequence of calls.
int dist_allreduce(double *buf, size_t count, hipStream_t s, std::vector<double> &hstage);
// A rank that fails inside a sharded solve aborts the communicator so that its peers fail too instead of waiting for ever.
void dist_abort_from(const char *why);
} // namespace abip
