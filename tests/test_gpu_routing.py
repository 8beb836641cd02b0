"""The assumed MI355X of tests/test_kernel_routing_cpu.py is the device: every plan of that file gets the same kernels from
`hdk_hip_describe_launch` on device 0 as on HDK_HIP_DEVICE_ASSUMED_MI355X, and the device's numbers are the assumed ones."""
import ctypes as C

import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import lib, sync_switches
from hdk_amd.ir import Agg, Cast, Cmp, ColRef, FP64, KeyRef, Lit, Or, QueryUnit
from hdk_amd.plan import compile_query

import test_kernel_routing_cpu as R

pytestmark = pytest.mark.gpu

storage = R.storage  # (the module-scoped fixture of the CPU file)


def _names_on(cp, device, total_rows=R.BIG, flags=0):
    sync_switches()
    ko = A.KernelOptions(0, 0, 0, flags, total_rows, 0, 0)
    out = C.create_string_buffer(256)
    assert lib().hdk_hip_describe_launch(C.byref(cp.plan), C.byref(ko), device, out, 256) == 0
    return out.value.decode()


def test_assumed_device_is_this_device(storage, gpu_executor_factory):
    ex = gpu_executor_factory(storage)
    props = ex.mgr.getDeviceProperties(0)
    assert (props.num_cu, props.wavefront_size, props.max_threads_per_block, props.grid_size) == (256, 64, 1024, 1024)
    assert props.shared_mem_per_block == 160 << 10
    K, V, C_ = ColRef("key"), ColRef("val"), ColRef("c")
    queries = [
        QueryUnit("t", groupby=[K], targets=[KeyRef(0), Agg("sum", V)]),
        QueryUnit("t", quals=[Or(Cmp(V, "<", Lit(0)), Cmp(K, "=", Lit(3)))], groupby=[K], targets=[KeyRef(0), Agg("sum", V)]),
        QueryUnit("t", groupby=[K, ColRef("k2")], targets=[KeyRef(0), KeyRef(1), Agg("count", None)]),
        R._bh("x10"), R._bh("x1k"), R._bh("x100k"), R._bh("x10", quals=[Cmp(ColRef("y10"), "<=", Lit(7))]),
        QueryUnit("syn", groupby=[ColRef("x1k") % 37], targets=[KeyRef(0), Agg("sum", ColRef("y10"))]),
        QueryUnit("syn", groupby=[ColRef("x1k") % 37], targets=[KeyRef(0), Agg("sum", ColRef("d"))]),
        QueryUnit("syn", groupby=[ColRef("sparse")], force_baseline=True, baseline_entry_count=180_001,
                  targets=[KeyRef(0), Agg("sum", ColRef("y10")), Agg("count", None)]),
        QueryUnit("t", groupby=[ColRef("wide")], force_baseline=True, baseline_entry_count=200_000_000,
                  targets=[KeyRef(0), Agg("sum", V)]),
    ]
    for q in queries:
        cp = compile_query(storage, q)
        for rows in (R.BIG, 100_000):
            assert _names_on(cp, 0, rows) == _names_on(cp, R.ASSUMED_MI355X, rows), (q, rows)


def test_assumed_device_routes_the_star_schema_shapes_like_this_device():
    from hdk_amd.ir import JoinSpec
    from hdk_amd.storage import ArrowStorage
    import numpy as np
    rng = np.random.default_rng(4)
    nd, n = 10_000_000, 200_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": np.arange(nd, dtype=np.int64), "dval": rng.integers(0, 10**6, nd).astype(np.int64),
                            "attr": rng.integers(0, 64, nd).astype(np.int64)})
    st.import_numpy("fact", {"fk": rng.integers(0, nd, n, dtype=np.int64), "val": rng.integers(-2**31, 2**31, n, dtype=np.int64),
                             "g32": rng.integers(0, 64, n).astype(np.int32)}, fragment_size=n // 2 + 1)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    V, D, At = ColRef("val"), ColRef("dval", "dim"), ColRef("attr", "dim")
    for q in (QueryUnit("fact", joins=j, targets=[Agg("sum", V + D)]),
              QueryUnit("fact", joins=j, groupby=[D / 15625], targets=[KeyRef(0), Agg("sum", V)]),
              QueryUnit("fact", joins=j, groupby=[D % 64], targets=[KeyRef(0), Agg("sum", V)]),
              QueryUnit("fact", joins=j, quals=[Cmp(D, "<", Lit(500_000))], groupby=[At], targets=[KeyRef(0), Agg("sum", V)]),
              QueryUnit("fact", joins=j, quals=[Cmp(D, "<", Lit(500_000))], groupby=[ColRef("g32")], targets=[KeyRef(0), Agg("sum", V)])):
        cp = R._fused(compile_query(st, q))
        for rows in (R.BIG, 1_000_000):
            assert _names_on(cp, 0, rows) == _names_on(cp, R.ASSUMED_MI355X, rows), (q, rows)
