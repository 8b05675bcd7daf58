"""ctypes binding of libabip_hip.so (include/abip.h + include/abip_hip.h).

The library is the product; this module only declares its C ABI to Python.  There is no
CPU fallback: if the shared object is missing or no HIP device is usable the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

c_int = C.c_long       # abip_int (DLONG build, the reference's mex default)
c_flt = C.c_double
PF = C.POINTER(c_flt)
PI = C.POINTER(c_int)

# ABIP_HIP_LIBRARY: another variant of the same library (the tests' libabip_hip_hooks.so with the fault-injection hooks compiled in)
LIB_PATH = os.environ.get("ABIP_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libabip_hip.so")


class ABIPMatrix(C.Structure):  # src/abip-lp/linsys/amatrix.h:10-17
    _fields_ = [("x", PF), ("i", PI), ("p", PI), ("m", c_int), ("n", c_int)]


class ABIPSettings(C.Structure):  # src/abip-lp/include/abip.h:36-79
    _fields_ = [
        ("normalize", c_int), ("pfeasopt", c_int), ("scale", c_flt), ("rho_y", c_flt), ("sparsity_ratio", c_flt),
        ("max_ipm_iters", c_int), ("max_admm_iters", c_int), ("max_time", c_flt),
        ("eps", c_flt), ("alpha", c_flt), ("cg_rate", c_flt),
        ("adaptive", c_int), ("eps_cor", c_flt), ("eps_pen", c_flt),
        ("dynamic_sigma", c_flt), ("dynamic_x", c_flt), ("dynamic_eta", c_flt),
        ("restart_fre", c_int), ("restart_thresh", c_int),
        ("verbose", c_int), ("warm_start", c_int), ("adaptive_lookback", c_int),
        ("origin_rescale", c_int), ("pc_ruiz_rescale", c_int), ("qp_rescale", c_int), ("ruiz_iter", c_int),
        ("hybrid_mu", c_int), ("hybrid_thresh", c_flt), ("dynamic_sigma_second", c_flt),
        ("half_update", c_int), ("avg_criterion", c_int),
    ]


class ABIPData(C.Structure):  # src/abip-lp/include/abip.h:23-34
    _fields_ = [("m", c_int), ("n", c_int), ("A", C.POINTER(ABIPMatrix)), ("b", PF), ("c", PF), ("sp", c_flt),
                ("stgs", C.POINTER(ABIPSettings))]


class ABIPSolution(C.Structure):  # src/abip-lp/include/abip.h:81-86
    _fields_ = [("x", PF), ("y", PF), ("s", PF)]


class ABIPInfo(C.Structure):  # src/abip-lp/include/abip.h:88-105
    _fields_ = [("status", C.c_char * 32), ("status_val", c_int), ("ipm_iter", c_int), ("admm_iter", c_int),
                ("pobj", c_flt), ("dobj", c_flt), ("res_pri", c_flt), ("res_dual", c_flt), ("rel_gap", c_flt),
                ("res_infeas", c_flt), ("res_unbdd", c_flt), ("setup_time", c_flt), ("solve_time", c_flt)]


K_CLASSES = ("spmv_At", "spmv_A", "cg_vec", "sptrsv", "vec", "qnorm", "cg_edge", "xcd")  # include/abip_hip.h ABIP_HIP_K_*


class AbipHipProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_long * 8), ("noop_ms", C.c_double), ("noop_launches", C.c_long),
                ("admm_iters", C.c_long), ("cg_iters", C.c_long), ("kkt_solves", C.c_long),
                ("stamp_ms", C.c_double * 8), ("stamp_launches", C.c_long * 8), ("stamp_noop_launches", C.c_long),
                ("allreduce_ms", C.c_double), ("allreduce_calls", C.c_long), ("allreduce_bytes", C.c_double), ("cg_iters_skipped", C.c_long)]


def kernel_sources_sha256() -> str:
    """Hash of everything the launch path's kernels are built from: dev_kernels.h and every header it includes, plus solver.hip (launch geometry and arguments).
    Ties a committed trace / stamp ratio (scripts/trace_medians.py -> profiles/*_trace_durations.json) to the kernels it was measured on: bench.py does not apply a
    ratio taken on other sources.  One helper for bench.py and scripts/trace_medians.py."""
    import hashlib
    import os
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in ("dev_common.h", "dev_kernels.h", "dev_peer.h", "lp_scalars.h", "solver.hip"):
        h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()


# every symbol include/abip.h and include/abip_hip.h declare
EXPORTS = (
    "abip_init", "abip_solve", "abip_finish", "abip_main", "abip_version", "abip_set_default_settings",
    "abip_free_data", "abip_free_sol", "abip_hip_set_linsys", "abip_hip_get_linsys",
    "abip_hip_device_info", "abip_hip_solve_begin", "abip_hip_step", "abip_hip_solve_end",
    "abip_hip_accum_by_A", "abip_hip_accum_by_Atrans", "abip_hip_kkt_solve", "abip_hip_get_vector",
    "abip_hip_get_scalar", "abip_hip_profile_enable", "abip_hip_profile_read", "abip_hip_sync",
    "abip_hip_dist_get_unique_id", "abip_hip_dist_init_rccl", "abip_hip_dist_init_callback", "abip_hip_dist_finalize",
    "abip_hip_dist_peer_capacity", "abip_hip_dist_peer_prepare", "abip_hip_dist_init_peer",
    "abip_hip_dist_partition", "abip_hip_dist_rows", "abip_hip_host_factor_solve", "abip_hip_host_normalize_A",
    "abip_hip_dist_comm_count", "abip_hip_profile_enable_stamps", "abip_hip_set_copy_a_matrix", "abip_hip_get_copy_a_matrix", "abip_hip_ldl_solve", "abip_hip_xcd_plan", "abip_hip_tail_plan", "abip_hip_csc_to_csr",
    "abip_qcp", "abip_qcp_set_default_settings", "abip_hip_qcp_last_stats", "abip_hip_qcp_phase_times", "abip_hip_qcp_cone_prox", "abip_hip_qcp_dist_partition", "abip_hip_qcp_host_probe",
)

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_long)

_lib = None


def _preload_hip_runtime() -> None:
    """A process can drive the GPU through ONE HIP runtime only.  PyTorch wheels bundle their own libamdhip64.so and load it by
    path, so if this library were loaded first (binding /opt/rocm's copy) and torch afterwards, the process would hold two
    runtimes and the second one to touch the device fails ("no usable HIP device").  When torch is installed but not yet
    imported, load its copy first: libabip_hip.so then binds to it by SONAME and a later `import torch` reuses it."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("ABIP_HIP_SYSTEM_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  abip_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    W = C.c_void_p
    L.abip_init.restype = W
    L.abip_init.argtypes = [C.POINTER(ABIPData), C.POINTER(ABIPInfo)]
    L.abip_solve.restype = c_int
    L.abip_solve.argtypes = [W, C.POINTER(ABIPData), C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
    L.abip_finish.restype = None
    L.abip_finish.argtypes = [W]
    L.abip_main.restype = c_int
    L.abip_main.argtypes = [C.POINTER(ABIPData), C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
    L.abip_version.restype = C.c_char_p
    L.abip_set_default_settings.restype = None
    L.abip_set_default_settings.argtypes = [C.POINTER(ABIPData)]
    L.abip_hip_set_linsys.restype = None
    L.abip_hip_set_linsys.argtypes = [C.c_int]
    L.abip_hip_get_linsys.restype = C.c_int
    L.abip_hip_set_copy_a_matrix.restype = None
    L.abip_hip_set_copy_a_matrix.argtypes = [C.c_int]
    L.abip_hip_device_info.restype = C.c_int
    L.abip_hip_device_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int)]
    L.abip_hip_solve_begin.restype = c_int
    L.abip_hip_solve_begin.argtypes = [W, C.POINTER(ABIPData), C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
    L.abip_hip_step.restype = c_int
    L.abip_hip_step.argtypes = [W, c_int, PI, C.POINTER(ABIPInfo)]
    L.abip_hip_solve_end.restype = c_int
    L.abip_hip_solve_end.argtypes = [W, C.POINTER(ABIPSolution), C.POINTER(ABIPInfo)]
    L.abip_hip_accum_by_A.restype = c_int
    L.abip_hip_accum_by_A.argtypes = [W, PF, PF]
    L.abip_hip_accum_by_Atrans.restype = c_int
    L.abip_hip_accum_by_Atrans.argtypes = [W, PF, PF]
    L.abip_hip_kkt_solve.restype = c_int
    L.abip_hip_kkt_solve.argtypes = [W, PF, PF, c_int]
    L.abip_hip_get_vector.restype = c_int
    L.abip_hip_get_vector.argtypes = [W, C.c_char_p, PF, c_int]
    L.abip_hip_get_scalar.restype = c_flt
    L.abip_hip_get_scalar.argtypes = [W, C.c_char_p]
    L.abip_hip_profile_enable.restype = None
    L.abip_hip_profile_enable.argtypes = [W, C.c_uint]
    L.abip_hip_profile_read.restype = None
    L.abip_hip_profile_read.argtypes = [W, C.POINTER(AbipHipProfile), C.c_int]
    L.abip_hip_sync.restype = None
    L.abip_hip_sync.argtypes = [W]
    L.abip_hip_dist_get_unique_id.restype = C.c_int
    L.abip_hip_dist_get_unique_id.argtypes = [C.c_void_p]
    L.abip_hip_dist_init_rccl.restype = C.c_int
    L.abip_hip_dist_init_rccl.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.abip_hip_dist_init_callback.restype = C.c_int
    L.abip_hip_dist_init_callback.argtypes = [C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
    L.abip_hip_dist_peer_capacity.restype = C.c_long
    L.abip_hip_dist_peer_capacity.argtypes = [C.c_long, C.c_long]
    L.abip_hip_dist_peer_prepare.restype = C.c_int
    L.abip_hip_dist_peer_prepare.argtypes = [C.c_long, C.c_void_p]
    L.abip_hip_dist_init_peer.restype = C.c_int
    L.abip_hip_dist_init_peer.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.abip_hip_dist_finalize.restype = None
    L.abip_hip_dist_comm_count.restype = C.c_int
    L.abip_hip_profile_enable_stamps.restype = C.c_int
    L.abip_hip_profile_enable_stamps.argtypes = [W, C.c_uint]
    L.abip_hip_dist_partition.restype = C.c_int
    L.abip_hip_dist_partition.argtypes = [C.POINTER(ABIPMatrix), C.c_int, PI]
    L.abip_hip_dist_rows.restype = None
    L.abip_hip_dist_rows.argtypes = [W, PI, PI]
    _lib = L
    return L
