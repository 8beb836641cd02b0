"""GPU parity for the general join loops: one-to-many matching sets, LEFT joins,
filters on joined columns, keyed (composite / wide key) tables built on the device, two join levels --
every case of tests/joins_cases.py against the oracle, through the LDS strategy (row-at-a-time
interpreter), the global-atomics strategy, and the scalar projection kernel."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs

from joins_cases import make_case, pyhdk_join_tables, sort_rows
from test_gpu_projection import _sorted_rows
from test_projection import run_projection_oracle
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    return make_case(seed=12, nf=120_000, nd=2_000)


def test_aggregates_over_general_joins(oracle, gpu_executor_factory, case):
    st, _, cases, _ = case
    ex = gpu_executor_factory(st)
    for name, q, _, _ in cases:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, name
        step = ex.prepare(cp)
        # (matching sets: the row-at-a-time interpreter; at most one partner per row -- one-to-one tables, perfect or keyed,
        # INNER / LEFT / SEMI / ANTI: the batched ones)
        names = step.kernel_names()
        one_to_one = all(i["kind"] in (A.JOIN_ONE_TO_ONE, A.JOIN_KEYED_ONE_TO_ONE) for i in cp.join_infos)
        many = len(cp.join_infos) == 1 and cp.join_infos[0]["kind"] == A.JOIN_ONE_TO_MANY  # replayed per match in the batched kernel
        want_k = ("hdk_scan_agg_vec_join", "hdk_scan_agg_vec_keyed") if one_to_one else ("hdk_scan_agg_vec_many" if many else "hdk_scan_agg_generic")
        assert names.startswith(want_k), (name, names)
        res = step.run()
        assert_buffers_equal(cp, res.buffer, want)
        step.free()
        res = ex.execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS)
        assert_buffers_equal(cp, res.buffer, want)


def test_projections_over_general_joins(oracle, gpu_executor_factory, case):
    st, _, _, proj_cases = case
    ex = gpu_executor_factory(st)
    for name, q, _ in proj_cases:
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0 and nrows > 0, name
        step = ex.prepare(cp)
        one_to_one = all(i["kind"] in (A.JOIN_ONE_TO_ONE, A.JOIN_KEYED_ONE_TO_ONE) for i in cp.join_infos)
        assert (step.kernel_names() == "hdk_scan_project_scalar") == (not one_to_one), (name, step.kernel_names())
        res = step.run()
        assert res.total_matched == nrows, name
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows)), name
        step.free()


def test_post_join_filter_in_the_batched_kernels(oracle, gpu_executor_factory):
    """A filter on a joined column with an inner one-to-one join stays on the batched (vec) kernels,
    fused table or not."""
    from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, QueryUnit
    from hdk_amd.storage import ArrowStorage
    rng = np.random.default_rng(3)
    nd, nf = 5_000, 400_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "v": rng.integers(0, 100, nd).astype(np.int64),
                            "g": rng.integers(0, 7, nd).astype(np.int32)})
    st.import_numpy("fact", {"fk": rng.integers(-5, nd + 5, nf).astype(np.int64), "val": rng.integers(-9, 9, nf)},
                    fragment_size=90_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                  quals=[Cmp(ColRef("v", "dim"), "<", Lit(50)), Cmp(ColRef("val"), ">", Lit(-5))],
                  groupby=[ColRef("g", "dim")], targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.quals[0].after_joins == 1 and cp.plan.quals[1].after_joins == 0
    for fuse in (True, False):
        ex = gpu_executor_factory(st)
        ex.fuse_join_tables = fuse
        step = ex.prepare(cp)
        assert step.kernel_names().startswith("hdk_scan_agg_vec_join")
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
    assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_SCALAR).buffer, want)


def test_pyhdk_api_join_pairs_on_gpu(gpu_executor_factory):
    """python/tests/test_pyhdk_api.py:609-667 (reference golden input/output pairs)."""
    st, cases = pyhdk_join_tables()
    ex = gpu_executor_factory(st)
    for q, expected in cases:
        res = ex.execute(q)
        got = res.to_columns()
        assert list(got) == list(expected)
        g, w = sort_rows(got), sort_rows(expected)
        assert len(g) == len(w)
        for a, b in zip(g, w):
            for x, y in zip(a, b):
                assert (x is None and y is None) or x == pytest.approx(y)
