"""GPU half of tests/test_abi_negative.py: corrupted plans and bad arguments through the entry points that enqueue
kernels.  Every call must come back with HDK_HIP_ERR_INVALID_ARG / _UNSUPPORTED before anything reaches the device,
and the device must still run a good launch afterwards (nothing faulted)."""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import lib
from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
from hdk_amd.storage import ArrowStorage

from test_abi_negative import _plans, corruptions
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu


def test_corrupted_plans_never_reach_the_device(oracle, gpu_executor_factory):
    L = lib()
    rng = np.random.default_rng(5)
    n = 100_000
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 50, n).astype(np.int64), "v": rng.integers(-9, 9, n).astype(np.int64)},
                    fragment_size=30_000)
    q = QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v"))])
    cp, want, err = run_oracle(oracle, st, q)
    ex = gpu_executor_factory(st)
    step = ex.prepare(cp)
    ws = C.c_size_t(0)
    out = C.create_string_buffer(256)
    bad_calls = 0
    for other in [cp] + _plans(60, seed=11):
        for what, mutate in corruptions(other):
            bad = A.Plan.from_buffer_copy(other.plan)
            mutate(bad)
            codes = [
                L.hdk_hip_workspace_size(C.byref(bad), C.byref(step.ko), 0, C.byref(ws)),
                L.hdk_hip_describe_launch(C.byref(bad), C.byref(step.ko), 0, out, 256),
                L.hdk_hip_launch(C.byref(bad), step._params, C.byref(step.ko), 0, None, step.workspace.ptr, step.workspace.nbytes),
            ]
            assert all(c in (A.ERR_INVALID_ARG, A.ERR_UNSUPPORTED) for c in codes), (what, codes)
            bad_calls += len(codes)
    assert bad_calls > 1500
    # bad arguments around a good plan
    good = step.plan
    assert L.hdk_hip_launch(C.byref(good), None, C.byref(step.ko), 0, None, step.workspace.ptr, step.workspace.nbytes) == A.ERR_INVALID_ARG
    assert L.hdk_hip_launch(C.byref(good), step._params, C.byref(step.ko), 0, None, step.workspace.ptr, 16) == A.ERR_INVALID_ARG
    assert L.hdk_hip_launch(C.byref(good), step._params, C.byref(step.ko), 0, None, step.workspace.ptr + 4, step.workspace.nbytes - 4) == A.ERR_INVALID_ARG
    assert L.hdk_hip_launch(C.byref(good), step._params, C.byref(step.ko), 99, None, step.workspace.ptr, step.workspace.nbytes) == A.ERR_INVALID_ARG
    params = (C.c_void_p * A.KP_COUNT)(*[step._params[i] for i in range(A.KP_COUNT)])
    params[A.KP_ERROR_CODE] = None
    assert L.hdk_hip_launch(C.byref(good), params, C.byref(step.ko), 0, None, step.workspace.ptr, step.workspace.nbytes) == A.ERR_INVALID_ARG
    shape = A.ExchangeShape()
    assert L.hdk_hip_exchange_shape_for(C.byref(good), C.byref(step.ko), 4, 1000, 0, C.byref(shape)) == A.ERR_UNSUPPORTED  # not open addressing
    # the device is fine: the good launch still gives the oracle's buffer
    assert_buffers_equal(cp, step.run().buffer, want)
    step.free()
