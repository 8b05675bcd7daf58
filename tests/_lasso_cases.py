"""Seeded LASSO instances shaped like scripts/bench-qcp/get_lasso_simu_data.m:3-14 (sprandn-like X, half of the true
coefficients N(0, 1/n), lambda = |X'y|_inf / 5).  Shared by the CPU (oracle) and GPU (product) LASSO tests."""
import numpy as np
import scipy.sparse as sp

# name: (samples m, features n, density, seed).  density >= 0.1 takes the "dense" scaling branch of lasso_config.c:36-51,
# below it the "sparse" one; m > n and m <= n take the two reduced systems of :506-556.
CASES = {
    "wide_dense": (60, 150, 0.3, 1),
    "tall_dense": (200, 80, 0.3, 2),
    "wide_sparse": (100, 300, 0.05, 3),
    "tall_sparse": (300, 100, 0.05, 4),
    "wide_sparse_big": (400, 1500, 0.02, 5),
}


def gen(name):
    m, n, dens, seed = CASES[name]
    rng = np.random.default_rng(seed)
    X = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    v = np.where(rng.random(n) < 0.5, rng.standard_normal(n) / np.sqrt(n), 0.0)
    y = X @ v + 0.01 * rng.standard_normal(m)
    lam = float(np.abs(X.T @ y).max() / 5)
    return X, y, lam


def objective(X, y, lam, beta):
    return 0.5 * float(np.sum((X @ beta - y) ** 2)) + lam * float(np.abs(beta).sum())
