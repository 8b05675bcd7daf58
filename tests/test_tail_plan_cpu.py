"""CPU: the plan of the dense tail's stream (abip_amd/csrc/dev_tail.h SymPlan through the pure-host export abip_hip_tail_plan): the lower triangle of the T x T inverse,
cut into column chunks of 512 and units of four rows, dealt to the wavefronts as contiguous ranges of units.  Checked here without a GPU: the ranges tile the
unit list, no wavefront is empty, every wavefront gets the same number of units +- 1 and at least one 16-row tile, the (wavefront, chunk) slots of the
column-partial table are unique and dense, and qlo / qhi name exactly the wavefronts whose range meets a chunk (what k_tail_sym_fin adds up)."""
import ctypes as C

import numpy as np
import pytest

from abip_amd import _lib


def plan(T, waves):
    L = _lib.load()
    pi = C.POINTER(C.c_int)
    L.abip_hip_tail_plan.restype = C.c_int
    L.abip_hip_tail_plan.argtypes = [C.c_int, C.c_int, pi, pi, pi, pi]
    out4, pre, qlo, qhi = np.zeros(4, np.int32), np.zeros(65, np.int32), np.zeros(64, np.int32), np.zeros(64, np.int32)
    rc = L.abip_hip_tail_plan(T, waves, out4.ctypes.data_as(pi), pre.ctypes.data_as(pi), qlo.ctypes.data_as(pi), qhi.ctypes.data_as(pi))
    return rc, out4, pre, qlo, qhi


@pytest.mark.parametrize("T,waves", [(64, 2048), (128, 2048), (512, 2048), (576, 2048), (2048, 2048), (8960, 2048), (10048, 2048), (10048, 1024), (10048, 8192), (16384, 2048), (32768, 4096)])
def test_ranges_tile_the_triangle(T, waves):
    rc, (ncc, nu, nwv, slots), pre, qlo, qhi = plan(T, waves)
    assert rc == 0 and ncc == -(-T // 512) and nwv % 4 == 0 and 4 <= nwv <= max(waves, 4)
    # units of a chunk: its rows [512 cc, T) four at a time
    assert pre[0] == 0 and all(pre[cc + 1] - pre[cc] == (T - 512 * cc) // 4 for cc in range(ncc)) and pre[ncc] == nu
    assert nu * 4 * 512 >= T * (T + 1) // 2           # (the chunks cover the lower triangle)
    first = [q * nu // nwv for q in range(nwv + 1)]
    lens = np.diff(first)
    assert first[0] == 0 and first[-1] == nu and lens.min() >= 1 and lens.max() - lens.min() <= 1
    assert lens.min() >= min(4, nu // nwv)            # at least one 16-row tile per wavefront (unless the tail is smaller than that)
    assert slots == nwv + ncc
    seen = set()
    for cc in range(ncc):
        meets = [q for q in range(nwv) if first[q] < pre[cc + 1] and first[q + 1] > pre[cc]]
        assert meets == list(range(qlo[cc], qhi[cc] + 1)), (cc, meets[:3], meets[-3:], qlo[cc], qhi[cc])
        for q in meets:
            assert q + cc not in seen                 # one slot per (wavefront, chunk) visit ...
            seen.add(q + cc)
    assert max(seen) < slots                          # ... inside the table


def test_no_plan_where_the_kernel_has_none():
    assert plan(40, 2048)[0] == -1          # not a multiple of 16
    assert plan(0, 2048)[0] == -1
    assert plan(32768 + 512, 2048)[0] == -1  # more than 64 column chunks
