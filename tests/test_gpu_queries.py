"""GPU parity on the query shapes either side of the headline kernel: taxi Q1-Q4 (reference known
answers), hash-join probe (config C3 shape), filters; all through the C ABI, checked against the oracle."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from taxi import check_taxi_results, load_taxi, taxi_queries
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu


def _run(O, make, st, q, **kw):
    cp, want, err = run_oracle(O, st, q)
    assert err == 0
    ex = make(st)
    res = ex.execute(cp, **kw)
    assert_buffers_equal(cp, res.buffer, want)
    if cp.join_infos:  # the reference-layout join table (no fused payload) and the scalar interpreter
        ex2 = make(st)
        ex2.fuse_join_tables = False
        assert_buffers_equal(cp, ex2.execute(cp, **kw).buffer, want)
        assert_buffers_equal(cp, ex2.execute(cp, flags=A.LAUNCH_FORCE_SCALAR).buffer, want)
    return cp, res


def test_taxi_q1_q4(oracle, gpu_executor_factory):
    for frag in (None, 6):
        st = ArrowStorage()
        load_taxi(st, frag)
        cols = [_run(oracle, gpu_executor_factory, st, q)[1].to_columns() for q in taxi_queries()]
        check_taxi_results(*cols)


def test_join_probe_sum_c3_shape(oracle, gpu_executor_factory):
    rng = np.random.default_rng(2026)
    nd, nf = 100_000, 1_000_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "dval": rng.integers(0, 10**6, nd).astype(np.int64)},
                    fragment_size=30_000)
    st.import_numpy("fact", {"fk": rng.integers(0, nd, nf).astype(np.int64),
                             "val": rng.integers(-2**31, 2**31, nf).astype(np.int64)}, fragment_size=250_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                  targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim"), "s")])
    _run(oracle, gpu_executor_factory, st, q)
    q2 = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") / 20000],
                   targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    cp, res = _run(oracle, gpu_executor_factory, st, q2)
    assert sum(res.to_columns()["c"]) == nf


def test_join_with_misses_and_nulls(oracle, gpu_executor_factory):
    rng = np.random.default_rng(7)
    nd, nf = 500, 50_000
    fk = rng.integers(-20, nd + 20, nf).astype(np.int64)
    fk[rng.random(nf) < 0.05] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "w": rng.integers(1, 9, nd).astype(np.int32)})
    st.import_numpy("fact", {"fk": fk, "val": rng.integers(0, 100, nf).astype(np.int32)}, fragment_size=9_999)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("val"), "<", Lit(90))],
                  groupby=[ColRef("w", "dim")], targets=[KeyRef(0, "w"), Agg("count"), Agg("avg", ColRef("val"))])
    _run(oracle, gpu_executor_factory, st, q)


def test_duplicate_dim_keys_take_the_one_to_many_table(oracle, gpu_executor_factory):
    st = ArrowStorage()
    st.import_numpy("dim", {"key": np.array([0, 1, 2, 0], dtype=np.int64), "d": np.arange(4, dtype=np.int64)})
    st.import_numpy("fact", {"fk": np.array([0, 1, 2], dtype=np.int64)})
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=[Agg("sum", ColRef("d", "dim"), "s")])
    ex = gpu_executor_factory(st)
    cp = ex.compile(q)
    assert cp.join_infos[0]["kind"] == A.JOIN_ONE_TO_MANY  # NeedsOneToManyHash (PerfectHashTableBuilder.h:134-141)
    assert ex.execute(cp).to_columns()["s"] == [0 + 3 + 1 + 2]
