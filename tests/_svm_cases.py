"""Seeded two-class data for the SVM reformulations; shared by the CPU (oracle) and GPU (product) tests."""
import numpy as np
import scipy.sparse as sp

CASES = {  # name: (samples m, features n, density, seed); m > n + 1 and m <= n + 1 take the two reduced systems of svm_qp_config.c:743-806
    "tall": (60, 20, 0.5, 1),
    "wide": (40, 100, 0.3, 2),
    "tall_sparse": (300, 30, 0.04, 3),
    "mid": (500, 200, 0.05, 4),
}


def gen(name):
    m, n, dens, seed = CASES[name]
    rng = np.random.default_rng(seed)
    X = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    wt = rng.standard_normal(n)
    y = np.sign(X @ wt + 0.3 * rng.standard_normal(m))
    y[y == 0] = 1.0
    return X, y


def hinge_objective(X, y, C, w, b):
    return 0.5 * float(w @ w) + C * float(np.maximum(0.0, 1.0 - y * (X @ w + b)).sum())
