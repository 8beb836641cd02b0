"""The reference's reduction matrices (Tests/ResultSetTest.cpp `Reduce.*`, `ReduceRandomGroups.*`) through
hdk_hip_reduce_buffers on the device: same fixtures, fills and expectations as tests/test_resultset_matrices.py
(tests/rs_matrix.py), the plan filled by hand from the descriptor -- nothing of hdk_amd/plan.py in the loop."""
import ctypes as C

import numpy as np
import pytest

import rs_matrix as M
from hdk_amd._lib import check, lib
from test_resultset_matrices import (DOC, RANDOM, REDUCE, check_random_case, check_reduce_case, run_random_case,
                                     run_reduce_case)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mgr():
    from hdk_amd.hip_mgr import HipMgr
    return HipMgr()


def make_device_reducer(mgr):
    def reduce_on_device(O, lay, plan, this_buf, this_entries, that_bufs):
        d_this = mgr.to_device(this_buf, 0)
        d_that = [mgr.to_device(b, 0) for b in that_bufs]
        that = (C.c_void_p * len(d_that))(*[d.ptr for d in d_that])
        counts = (C.c_uint32 * len(d_that))(*([lay.entry_count] * len(d_that)))
        d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
        iv = np.array(lay.init_vals, dtype=np.int64)
        check(lib().hdk_hip_reduce_buffers(C.byref(plan), d_this.ptr, this_entries, that, counts, len(d_that),
                                           iv.ctypes.data, d_err.ptr, 0, None))
        mgr.synchronizeStream(0)
        assert int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0]) == 0
        out = mgr.to_host(d_this.ptr, this_buf.nbytes, 0, np.uint8)
        for d in [d_this, d_err] + d_that:
            d.free()
        return out
    return reduce_on_device


@pytest.mark.parametrize("case", REDUCE, ids=[c["name"] for c in REDUCE])
def test_reduce_matrix_device(mgr, oracle, case):
    lay = M.make_layout(DOC, case)
    if not M.supported_by_library(lay):
        plan = M.make_plan(lay)
        d = mgr.to_device(np.zeros(max(lay.buffer_bytes, 8), dtype=np.uint8), 0)
        that = (C.c_void_p * 1)(d.ptr)
        counts = (C.c_uint32 * 1)(lay.entry_count)
        iv = np.array(lay.init_vals, dtype=np.int64)
        st = lib().hdk_hip_reduce_buffers(C.byref(plan), d.ptr, lay.entry_count, that, counts, 1, iv.ctypes.data, None, 0, None)
        assert st != 0  # 1- and 2-byte slots: rejected loudly, never reduced wrongly
        return
    rl, rbuf = run_reduce_case(oracle, case, make_device_reducer(mgr))
    check_reduce_case(case, rl, rbuf)


@pytest.mark.parametrize("case", RANDOM, ids=[c["name"] for c in RANDOM])
def test_reduce_random_groups_device(mgr, oracle, case):
    for seed in (1, 2):
        rl, em, rbuf = run_random_case(oracle, case, make_device_reducer(mgr), seed)
        check_random_case(case, rl, em, rbuf)
