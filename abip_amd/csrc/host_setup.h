// host_setup.h -- one-off host-side preparation for the device solver (runs once per abip_init).
#pragma once
#include <future>
#include <vector>

#include "../../include/abip.h"

namespace abip {
namespace host {

// A <- D^-1 A E^-1 * scale, in place (reference: ABIP(_normalize_A), linsys/common.c:150-565).
void normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, std::vector<double> &D, std::vector<double> &E,
                 double *mean_norm_row, double *mean_norm_col);
// inverse of the above (reference: ABIP(_un_normalize_A), linsys/common.c:569-594)
void un_normalize_A(ABIPMatrix *A, const ABIPSettings *stgs, const std::vector<double> &D, const std::vector<double> &E);
// validate_lin_sys (linsys/common.c:45-95) + validate (src/abip.c:1646-1734); prints the reference's messages
int validate(const ABIPData *d);

// CSR with 32-bit indices for the device + the row blocks the CSR-stream kernels walk.
struct HostCsr {
  int nrows = 0, ncols = 0;
  std::vector<int> ptr, idx;
  std::vector<double> val;
  std::vector<int> rb; // row-block boundaries
  // optional SELL-64 image (natural row order), see dev_common.h Csr; empty when not built
  std::vector<double> sval; std::vector<int> sidx, slen; std::vector<long> soff;
};
// CSC arrays of A read as the CSR of A' (n rows)
void csc_as_csr(const ABIPMatrix *A, HostCsr &out);
// explicit transpose: CSR of A (m rows), columns ascending inside a row (reference: indirect.c:81-139)
void transpose_to_csr(const ABIPMatrix *A, HostCsr &out);
// greedy row blocks: <= chunk non-zeros and <= chunk rows per block; a longer row gets a block of its own
void build_row_blocks(HostCsr &M, int chunk);
// SELL-64 image if the natural-order slices pad the matrix by at most `max_pad` (fraction of nnz) and it is large enough to matter; true if built
bool build_sell(HostCsr &M, double max_pad = 0.12, long min_nnz = 1 << 17);
// M_i = 1 / sum_j A_ij^2 (reference: get_preconditioner, indirect.c:36-79)
void jacobi_preconditioner(const ABIPMatrix *A, std::vector<double> &Minv);

// LDL' = P K P' of K = [[rho_y I, A],[A', -I]] (reference: form_kkt / factorize, linsys/direct.c:49-104,218-270),
// delivered in the two gather forms the device triangular solves use, with their level sets.
struct TriHost {
  std::vector<int> lev_ptr, lev_rows, lev_g; // level sets (only rows with at least one entry), lanes per row
  // entries of the rows of L (forward) / columns of L (backward) in level order: position r of lev_rows owns [ptr[r], ptr[r+1])
  std::vector<int> ptr, idx;
  std::vector<double> val;
};
struct LdlHost {
  int N = 0;
  long lnnz = 0;         // strictly-lower non-zeros of the complete factor (head + tail), as LDL_symbolic counts them
  std::vector<int> P;    // P[k] = original KKT index of pivot k
  std::vector<double> D; // pivots of the head [0, t0); the tail's come from the device factorisation
  TriHost fwd, bwd;      // sparse part: columns < t0 of L (rows of the tail included).  On large factors with a dense tail factor_upper returns BEFORE these two are
                         // built: a thread of its own fills them in while the device already works on the tail (dev_ldl.h) -- wait_forms() before reading them
  std::shared_future<void> forms_job;
  void wait_forms() const { if (forms_job.valid()) forms_job.wait(); }
  ~LdlHost() { wait_forms(); }
  int t0 = 0, T = 0;     // head size, dense-tail size (T % 64 == 0, t0 + T = N)
  std::vector<double> S; // T x T row-major, lower triangle: Schur complement of the head onto the tail -- EMPTY when dev_schur: then K22 arrives as the triplets below
  std::vector<int> k22_row, k22_col; std::vector<double> k22_val; // K22 (tail-local indices, row >= col): what S is before L21 D1 L21' is subtracted
  // dev_schur: S holds K22 only and the product L21 D1 L21' is still to be subtracted -- by the device (dev_ldl.h: dense panels of
  // L21 + a tiled rank-k update), which takes over when its dense panels beat the host's sparse accumulation (dense data
  // blocks: 35 000 head columns with 750 tail entries each = 10^10 multiply-adds, 11 s on one host thread, 0.2 s on the device).
  // L21 by head column for that purpose: column c owns [l21_ptr[c], l21_ptr[c+1]) of (l21_row = tail row - t0, l21_val), rows ascending.
  bool dev_schur = false;
  bool schur_rows = false; // the device multiplies it out row by row from the sparse L21 (k_schur_rows: LDS accumulator, T <= 20480) instead of by dense panels
  std::vector<long> l21_ptr;
  std::vector<int> l21_row;
  std::vector<double> l21_val;
};
// S -= L21 D1 L21' on the host (what the device does when dev_schur is set); clears dev_schur.  For host_solve and as the fallback.
void complete_schur_on_host(LdlHost &F);
// -1: decide by cost (default); 0: always accumulate on the host; 1: the device's dense panels; 2: the device's row-wise sparse kernel (tests).  Also env ABIP_HIP_DEV_SCHUR.
void set_dev_schur_request(int v);
// override the tail choice (tests): -2 = environment / automatic, -1 automatic, 0 none, T > 0 forced
void set_tail_request(int t);
// An elimination order for the next factor_upper call(s) instead of the minimum-degree pass (P[k] = index of pivot k; nullptr: back to minimum degree).
// For KKT matrices whose x block is diagonal and whose Schur complement onto the y block is dense anyway (the conic path with n >> m), eliminating
// the x block first IS the good order -- it is what the reference's reduced systems do -- and the ordering pass (C5: 0.8 s) has nothing to find.
void set_order_hint(const std::vector<int> *P);
// Host-only reference solve with a factor as the device would use it (level-ordered head, Schur complement factored densely on
// the host instead of the device): b <- K^-1 b in pivot order.  For the CPU tests of the ordering / numeric / tail-split code.
int host_solve(const LdlHost &F, std::vector<double> &b);
int factor_kkt(const ABIPMatrix *A, double rho_y, LdlHost &out);
// upper triangle of K = [[rho_y I, A],[A', -I]] by columns (direct.c:49-104), as factor_kkt builds it
void kkt_upper(const ABIPMatrix *A, double rho_y, std::vector<int> &Kp, std::vector<int> &Ki, std::vector<double> &Kx);
// Set-up guard of the direct back-ends: ||K z - rhs|| / ||rhs|| for a symmetric K given by its upper triangle (host arithmetic).
// The device forms W = inv(L22) explicitly for the dense tail (dev_ldl.h); one solve of a known right-hand side at set-up, checked
// here, catches a tail whose conditioning that cannot take (the callers then re-factor without a tail).
double sym_upper_residual(int N, const std::vector<int> &Kp, const std::vector<int> &Ki, const std::vector<double> &Kx,
                          const std::vector<double> &z, const std::vector<double> &rhs);
// the deterministic right-hand side of that check
void guard_rhs(int N, std::vector<double> &rhs);
// the same for any symmetric quasi-definite matrix given by its upper triangle in CSC form (QCP KKT, qcp_config.c:699-748)
int factor_upper(int N, const std::vector<int> &Kp, const std::vector<int> &Ki, const std::vector<double> &Kx, LdlHost &out);

} // namespace host
} // namespace abip
