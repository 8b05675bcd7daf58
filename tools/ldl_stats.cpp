// tools/ldl_stats.cpp -- host-only: factor the KKT matrix of an LP read from a raw dump and print the shape of L
// (levels, tail density).  g++ -O2 -std=c++17 -Iinclude -Iabip_amd/csrc tools/ldl_stats.cpp abip_amd/csrc/host_setup.cpp -o /tmp/ldl_stats
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "host_setup.h"
using namespace abip;
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  long hdr[3];
  if (fread(hdr, 8, 3, f) != 3) return 1;
  const long m = hdr[0], n = hdr[1], nnz = hdr[2];
  std::vector<abip_int> p(n + 1), i(nnz);
  std::vector<double> x(nnz);
  if (fread(p.data(), 8, n + 1, f) != (size_t)(n + 1) || fread(i.data(), 8, nnz, f) != (size_t)nnz || fread(x.data(), 8, nnz, f) != (size_t)nnz) return 1;
  ABIPMatrix A{x.data(), i.data(), p.data(), (abip_int)m, (abip_int)n};
  host::LdlHost F;
  const double rho = argc > 2 ? atof(argv[2]) : 1e-3;
  if (host::factor_kkt(&A, rho, F) < 0) { printf("factor failed\n"); return 1; }
  const int N = F.N;
  printf("t0 %d T %d head nnz %d\n", F.t0, F.T, (int)F.bwd.idx.size());
  printf("N %d Lnnz %ld levF %zu levB %zu\n", N, F.lnnz, F.fwd.lev_ptr.size() - 1, F.bwd.lev_ptr.size() - 1);
  // level histogram forward
  auto hist = [&](const host::TriHost &T, const char *nm) {
    const int nl = (int)T.lev_ptr.size() - 1;
    int thin = 0; long thinrows = 0;
    for (int l = 0; l < nl; ++l) { const int r = T.lev_ptr[l + 1] - T.lev_ptr[l]; if (r < 8) { ++thin; thinrows += r; } }
    printf("  %s: %d levels, %d with <8 rows (%ld rows); first levels:", nm, nl, thin, thinrows);
    for (int l = 0; l < nl && l < 40; ++l) printf(" %d:%d", T.lev_ptr[l + 1] - T.lev_ptr[l], T.ptr[T.lev_ptr[l + 1]] - T.ptr[T.lev_ptr[l]]);
    printf("\n");
  };
  hist(F.fwd, "fwd"); hist(F.bwd, "bwd");
  return 0;
}
