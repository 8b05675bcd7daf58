"""CPU: the regrouped u_t'h of the persistent launch (abip_amd/csrc/dev_xcd.h, the back-substitution block).

The reference adds u_t[i] h[i] over the whole vector (abip.c:560):   dh_ref = y'h_y + (A'y - rhs_x)'h_x.
The persistent launch forms it before the back-substitution's exchange: dh_reg = y'(h_y + A h_x) - rhs_x'h_x
(A h_x once per solve) so that the sum rides on that exchange instead of costing a rendez-vous of its own.  The two are equal in exact arithmetic; in floating
point the regrouped form subtracts two numbers of size |y|'|A||h_x| where the reference subtracts inside A'y - rhs_x first.  This test states the bound the
kernel relies on: on well and badly scaled LPs both forms agree with an extended-precision value to a few ulps of the CONDITION of the sum,
kappa = (|y|'|h_y| + |y|'|A||h_x| + |rhs_x|'|h_x|) / |dh|, and the regrouped form is never more than an order of magnitude worse than the reference's grouping.
(What the regrouping cannot keep is the reference's iteration count on the one knife-edge fixture: tests/test_gpu_parity.py KNIFE_EDGE.)"""
import numpy as np
import pytest
import scipy.sparse as sp

from abip_amd import problems


def _forms(A, y, h_y, h_x, rhs_x):
    At = A.T.tocsr()
    dh_ref = float(y @ h_y + (At @ y - rhs_x) @ h_x)
    dh_reg = float(y @ (h_y + A @ h_x) - rhs_x @ h_x)
    L = np.longdouble
    Ad = A.toarray().astype(L)
    yl, hyl, hxl, rl = y.astype(L), h_y.astype(L), h_x.astype(L), rhs_x.astype(L)
    truth = float(yl @ hyl + (Ad.T @ yl - rl) @ hxl)
    kappa = float((np.abs(y) @ np.abs(h_y) + np.abs(y) @ (abs(A) @ np.abs(h_x)) + np.abs(rhs_x) @ np.abs(h_x)) / max(abs(truth), 1e-300))
    return dh_ref, dh_reg, truth, kappa


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("spread", [0, 3, 6])     # columns and the vectors scaled by 10^U(-spread, spread)
def test_regrouped_sum_stays_within_the_condition_of_the_sum(seed, spread):
    rng = np.random.default_rng(100 * spread + seed)
    m, n = 60 + 7 * seed, 150 + 11 * seed
    A, b, c = problems.lp_random_sparse(m=m, n=n, per_col=4, seed=seed)
    A = sp.csr_matrix(A @ sp.diags(10.0 ** rng.uniform(-spread, spread, n)))
    y = rng.standard_normal(m) * 10.0 ** rng.uniform(-spread, spread, m)
    h_y = rng.standard_normal(m) * 10.0 ** rng.uniform(-spread / 2, spread / 2, m)
    h_x = rng.standard_normal(n) * 10.0 ** rng.uniform(-spread / 2, spread / 2, n)
    rhs_x = rng.standard_normal(n)
    dh_ref, dh_reg, truth, kappa = _forms(A, y, h_y, h_x, rhs_x)
    eps = np.finfo(np.float64).eps
    scale = max(abs(truth), 1e-300)
    err_ref, err_reg = abs(dh_ref - truth) / scale, abs(dh_reg - truth) / scale
    # a sum of ~(m + nnz + n) terms: a few hundred ulps of its condition at worst
    assert err_reg <= 400 * eps * kappa, (err_reg, kappa)
    assert err_ref <= 400 * eps * kappa, (err_ref, kappa)
    assert err_reg <= 10 * max(err_ref, eps * kappa), (err_reg, err_ref, kappa)


def test_the_two_groupings_differ_only_by_rounding_on_the_knife_edge_fixture():
    """lp_tiny_scale5 (the fixture whose iteration count no summation order but the reference's own reproduces): with its scaled A and a solve's (y, h) the two
    groupings agree to 1e-13 relative -- the count moves because the search that follows divides differences of nearly equal vectors, not because this sum is off."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _golden import load
    z, A, b, c = load("lp_tiny_scale5")
    A = sp.csr_matrix(A)
    rng = np.random.default_rng(5)
    m, n = A.shape
    y, h_y, h_x, rhs_x = rng.standard_normal(m), np.asarray(b, float), np.asarray(c, float), rng.standard_normal(n)
    dh_ref, dh_reg, truth, kappa = _forms(A, y, h_y, h_x, rhs_x)
    assert abs(dh_ref - dh_reg) <= 1e-13 * kappa * max(abs(truth), 1e-300)
