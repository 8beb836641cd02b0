"""Filter/project (QueryDescriptionType::Projection): the oracle against plain numpy.  CPU only."""
import numpy as np
import pyarrow as pa

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Cmp, ColRef, JoinSpec, Lit, Proj, QueryUnit
from hdk_amd.plan import compile_query
from hdk_amd.storage import ArrowStorage

from util import host_fragments, oracle_init_buffer, oracle_join_tables


def run_projection_oracle(O, st, q):
    cp = compile_query(st, q)
    buf = oracle_init_buffer(O, cp)
    err, n = O.run_projection(cp.plan, host_fragments(O, st, cp), buf, cp.entry_count, oracle_join_tables(O, st, cp))
    return cp, buf, err, n


def _table(n=5000, seed=1):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 100, n).astype(np.int64)
    b = rng.integers(-50, 50, n).astype(np.int32)
    b[rng.random(n) < 0.1] = A.NULL_INT
    d = rng.normal(size=n)
    s = rng.integers(0, 7, n).astype(np.int16)
    st = ArrowStorage()
    st.import_numpy("t", {"a": a, "b": b, "d": d, "s": s}, fragment_size=1234)
    return st, a, b, d, s


def test_projection_rowwise_and_columnar(oracle):
    st, a, b, d, s = _table()
    for columnar in (False, True):
        q = QueryUnit("t", quals=[Cmp(ColRef("a"), ">=", Lit(90)), Cmp(ColRef("b"), "<", Lit(0))],
                      targets=[Proj(ColRef("a"), "a"), Proj(ColRef("b") * 2 + ColRef("a"), "e"), Proj(ColRef("d"), "d"),
                               Proj(ColRef("s"), "s")], output_columnar=columnar)
        cp, buf, err, n = run_projection_oracle(oracle, st, q)
        assert err == 0 and cp.plan.query_kind == A.Q_PROJECTION
        keep = (a >= 90) & (b != A.NULL_INT) & (b < 0)
        assert n == int(keep.sum())
        cols = rs.to_columns(cp, buf, cp.entry_count, n)
        assert cols["a"] == a[keep].tolist()  # scan order on the CPU
        assert cols["e"] == (b[keep].astype(np.int64) * 2 + a[keep]).tolist()
        assert cols["d"] == d[keep].tolist() and cols["s"] == s[keep].tolist()
        if columnar:
            assert cp.slot_widths == [8, 8, 8, 2]  # logical widths for plain columns
        pos, _ = rs.projection_arrays(cp, buf, n)
        assert (np.diff(pos) != 0).all()  # row positions inside each fragment


def test_projection_nulls_and_limit(oracle):
    st, a, b, d, s = _table(2000, 2)
    q = QueryUnit("t", targets=[Proj(ColRef("b"), "b"), Proj(ColRef("b") + 1, "b1")], scan_limit=100)
    cp, buf, err, n = run_projection_oracle(oracle, st, q)
    assert err < 0 and n == 2000  # ran out of slots: negative code, counter keeps counting
    cols = rs.to_columns(cp, buf, cp.entry_count, n)
    want = [None if v == A.NULL_INT else int(v) for v in b[:100]]
    assert cols["b"] == want
    assert cols["b1"] == [None if v is None else v + 1 for v in want]


def test_projection_with_join_and_strings(oracle):
    st = ArrowStorage()
    st.import_arrow(pa.table({"key": pa.array([3, 1, 2], pa.int64()), "name": pa.array(["c", "a", "b"])}), "dim")
    st.import_arrow(pa.table({"fk": pa.array([1, 2, 3, 4, None, 2], pa.int64()), "v": pa.array([10, 20, 30, 40, 50, 60])}),
                    "fact", fragment_size=4)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                  targets=[Proj(ColRef("v"), "v"), Proj(ColRef("name", "dim"), "name")])
    cp, buf, err, n = run_projection_oracle(oracle, st, q)
    assert err == 0 and n == 4
    assert rs.to_columns(cp, buf, cp.entry_count, n) == {"v": [10, 20, 30, 60], "name": ["a", "b", "c", "b"]}
