"""CPU: host-side mirror of scripts/matlab/*.m -- defaults, validation messages, parameter conversion."""
import numpy as np
import pytest
import scipy.sparse as sp

from abip_amd import api


def test_get_params_defaults():  # abip_get_params.m:5-35
    p = api.abip_get_params()
    assert p["verbose"] == 1 and p["normalize"] == 1 and p["pcg"] == 0 and p["max_admm_iter"] == 1000000
    assert p["max_ipm_iter"] == 500 and p["timelimit"] == 3600 and p["tol"] == 1e-3 and p["solver"] == -1
    assert p["lpalg"] == dict(restart_thresh=100000, restart_freq=1000, feasopt=0, scaling_method=1, half_update=0)
    assert p["qcpalg"] == dict(rho_primal=1.0, rho_dual=1e-06, admm_tol_factor=1.0)


@pytest.mark.parametrize("field,val,msg", [
    ("max_admm_iter", 0, "Invalid max_admm_iter"), ("max_ipm_iter", -1, "Invalid max_ipm_iter"),
    ("timelimit", 0.0, "Invalid timelimit"), ("tol", 0.0, "Invalid tol")])
def test_check_params_rejects(field, val, msg):  # abip_check_params.m
    p = api.abip_get_params()
    p[field] = val
    with pytest.raises(ValueError, match=msg):
        api.abip_check_params(p)


def test_check_params_fills_missing_and_validates_lpalg():
    p = api.abip_check_params({"tol": 1e-5, "lpalg": {"scaling_method": 3}})
    assert p["tol"] == 1e-5 and p["lpalg"]["scaling_method"] == 3 and p["lpalg"]["restart_freq"] == 1000
    with pytest.raises(ValueError, match="Invalid scaling method"):
        api.abip_check_params({"lpalg": {"scaling_method": 7}})
    with pytest.raises(ValueError, match="restart_thresh"):
        api.abip_check_params({"lpalg": {"restart_thresh": 0}})


def test_lpparam_convert_keeps_reference_field_names():  # abip_lpsolve.m:34-61, incl. the two names the mex never reads
    p = api.abip_check_params({"tol": 1e-6, "lpalg": {"scaling_method": 2, "feasopt": 1}})
    lp = api._lpparam_convert(p)
    assert lp["eps"] == 1e-6 and lp["qp_rescale"] == 1 and lp["pc_ruiz_rescale"] == 0 and lp["origin_rescale"] == 0
    assert "max_admm_iter" in lp and "max_admm_iters" not in lp      # ignored by the gateway, as upstream
    assert "restart_freq" in lp and "restart_fre" not in lp
    assert "max_admm_iter" not in api._MEX_FIELDS and "restart_freq" not in api._MEX_FIELDS
    assert api._MEX_FIELDS["timelimit"] == "max_time" and api._MEX_FIELDS["feasopt"] == "pfeasopt"


def test_cone_dispatch_and_input_errors():
    A = sp.identity(2, format="csc")
    data = dict(A=A, b=np.ones(2), c=np.ones(2))
    with pytest.raises(ValueError, match="Invalid conic format for LP"):
        api.abip_lpsolve(data, {}, api.abip_get_params())
    with pytest.raises(ValueError, match="sparse format"):
        api.abip_direct(dict(A=np.eye(2), b=np.ones(2), c=np.ones(2)), {})
    with pytest.raises(ValueError, match="must contain a vector 'b'"):
        api.abip_direct(dict(A=A, c=np.ones(2)), {})
