"""abip_amd -- MI355X-native ABIP-LP inner-ADMM path behind the reference's own interface.

The product is the C-ABI shared library ``abip_amd/lib/libabip_hip.so`` (headers in ``include/``);
this package is its Python host-side mirror of the reference's Matlab surface.
"""
from .api import abip, abip_check_params, abip_direct, abip_get_params, abip_indirect, abip_lpsolve  # noqa: F401
from .solver import LINSYS_DIRECT, LINSYS_INDIRECT, Solver, default_settings  # noqa: F401

__all__ = ["abip", "abip_get_params", "abip_check_params", "abip_lpsolve", "abip_direct", "abip_indirect", "Solver",
           "default_settings", "LINSYS_DIRECT", "LINSYS_INDIRECT"]
