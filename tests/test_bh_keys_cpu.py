"""Group-by keys without an integer range -- cast(x as double) (the reference's BaselineHash benchmark queries,
Benchmarks/synthetic_benchmark/queries/BaselineHash/BH001-005.sql) and x % m (no modulo case in getExpressionRange,
QE/ExpressionRange.cpp:391-419) -- get the GroupByBaselineHash layout; the planner's entry count follows
RelAlgExecutor.cpp:1533-1557 and the oracle's result equals an independent numpy group-by."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Agg, Cast, ColRef, FP64, JoinSpec, KeyRef, QueryUnit
from hdk_amd.plan import BIG_GROUP_THRESHOLD, DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS, compile_query
from hdk_amd.storage import ArrowStorage

from util import run_oracle


def _bh_query(xcol):
    y = ColRef("y10")
    return QueryUnit("t", groupby=[Cast(ColRef(xcol), FP64)],
                     targets=[KeyRef(0, "key0"), Agg("count", y, "c"), Agg("sum", y, "s"), Agg("max", y, "mx"), Agg("min", y, "mn"),
                              Agg("avg", y, "a")])


def _table(n, seed=3, nulls=True):
    rng = np.random.default_rng(seed)
    x10 = rng.integers(1, 11, n).astype(np.int32)
    x1k = rng.integers(1, 1001, n).astype(np.int32)
    y10 = rng.integers(1, 11, n).astype(np.int32)
    if nulls:
        y10[rng.random(n) < 0.03] = A.NULL_INT
        x10[rng.random(n) < 0.01] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", {"x10": x10, "x1k": x1k, "y10": y10}, fragment_size=max(n // 3, 1))
    return st, x10, x1k, y10


def test_double_key_layout_and_entry_count():
    st, *_ = _table(50_000)
    cp = compile_query(st, _bh_query("x10"))
    p = cp.plan
    assert p.query_kind == A.Q_BASELINE_HASH and p.key_width == 8 and p.key_count == 1
    # a big input: 2 x the NDV bound (10 values + NULL), as 2 x the estimator's answer would be
    assert cp.entry_count == 2 * 11
    assert compile_query(st, _bh_query("x1k")).entry_count == 2 * 1000
    # a small input: the default guess, no estimation (groups_approx_upper_bound <= big_group_threshold)
    small, *_ = _table(BIG_GROUP_THRESHOLD)
    assert compile_query(small, _bh_query("x10")).entry_count == DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS
    # the projected key has no slot (target_groupby_indices), the other targets follow the row's 8-byte key word
    assert p.targets[0].slot_width == 0 and p.row_size_quad == 1 + 6


def test_oracle_double_key_equals_numpy_groupby(oracle):
    st, x10, _, y10 = _table(60_000)
    cp, buf, err = run_oracle(oracle, st, _bh_query("x10"))
    assert err == 0
    cols = rs.to_columns(cp, buf)
    got = {k: (c, s, mx, mn, a) for k, c, s, mx, mn, a in zip(cols["key0"], cols["c"], cols["s"], cols["mx"], cols["mn"], cols["a"])}
    keys = set(np.unique(x10[x10 != A.NULL_INT]).astype(float).tolist()) | {None}
    assert set(got) == keys and all(k is None or isinstance(k, float) for k in got)
    for k in keys:
        sel = (x10 == A.NULL_INT) if k is None else (x10 == int(k))
        y = y10[sel]
        y = y[y != A.NULL_INT].astype(np.int64)
        c, s, mx, mn, a = got[k]
        assert (c, s, mx, mn) == (len(y), int(y.sum()), int(y.max()), int(y.min()))
        assert abs(a - y.sum() / len(y)) < 1e-12


def test_modulo_key_after_a_join_is_baseline_hash(oracle):
    """SURVEY 8(d)'s C3 variant as written: GROUP BY dim.dval % 64 -- no expression range, so an open-addressing table."""
    rng = np.random.default_rng(8)
    nd, n = 5_000, 40_000
    st = ArrowStorage()
    dval = rng.integers(0, 10**6, nd).astype(np.int64)
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "dval": dval})
    fk = rng.integers(0, nd, n).astype(np.int64)
    val = rng.integers(-2**31, 2**31, n).astype(np.int64)
    st.import_numpy("fact", {"fk": fk, "val": val}, fragment_size=15_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") % 64],
                  targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s")])
    cp, buf, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.query_kind == A.Q_BASELINE_HASH and cp.plan.key_width == 8
    assert cp.entry_count == 2 * 64
    cols = rs.to_columns(cp, buf)
    by_key = np.empty(nd, dtype=np.int64)
    by_key[st.get("dim").columns["key"].fragments[0]] = dval
    g = by_key[fk] % 64
    want = {int(k): int(val[g == k].sum()) for k in np.unique(g)}
    assert dict(zip(cols["g"], cols["s"])) == want
