"""SQL-level pins of the oracle: the reference's taxi Q1-Q4 known answers and SQLite as an independent
second opinion (the reference's own test method, Tests/ArrowSQLRunner/SQLiteComparator.cpp:68-120)."""
import sqlite3

import numpy as np
import pyarrow as pa
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from taxi import check_taxi_results, load_taxi, taxi_queries
from util import run_oracle


def test_taxi_q1_q4_known_answers(oracle):
    for frag in (None, 7):
        st = ArrowStorage()
        load_taxi(st, frag)
        cols = []
        for q in taxi_queries():
            cp, buf, err = run_oracle(oracle, st, q)
            assert err == 0
            cols.append(rs.to_columns(cp, buf))
        check_taxi_results(*cols)


def _sqlite(tables, sql):
    con = sqlite3.connect(":memory:")
    for name, cols in tables.items():
        names = list(cols)
        con.execute(f"create table {name} ({', '.join(names)})")
        rows = list(zip(*[cols[n] for n in names]))
        con.executemany(f"insert into {name} values ({', '.join('?' * len(names))})", rows)
    return con.execute(sql).fetchall()


def _rows(cols, order):
    names = list(cols)
    rows = list(zip(*[cols[n] for n in names]))
    return sorted(rows, key=lambda r: tuple((x is None, x) for x in (r[i] for i in order)))


def test_groupby_vs_sqlite(oracle):
    rng = np.random.default_rng(42)
    n = 5000
    k = rng.integers(-3, 20, n).tolist()
    v = [None if rng.random() < 0.1 else int(x) for x in rng.integers(-1000, 1000, n)]
    f = [None if rng.random() < 0.1 else float(x) for x in rng.normal(size=n)]
    kk = [None if rng.random() < 0.05 else x for x in k]
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array(kk, pa.int32()), "v": pa.array(v, pa.int64()), "f": pa.array(f, pa.float64())}),
                    "t", fragment_size=1300)
    q = QueryUnit("t", groupby=[ColRef("k")],
                  targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("count", ColRef("v"), "cv"),
                           Agg("sum", ColRef("v"), "s"), Agg("min", ColRef("v"), "mn"), Agg("max", ColRef("v"), "mx"),
                           Agg("avg", ColRef("f"), "af")])
    cp, buf, err = run_oracle(oracle, st, q)
    assert err == 0
    got = _rows(rs.to_columns(cp, buf), [0])
    want = _sqlite({"t": {"k": kk, "v": v, "f": f}},
                   "select k, count(*), count(v), sum(v), min(v), max(v), avg(f) from t group by k")
    want = sorted(want, key=lambda r: (r[0] is None, r[0]))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:6] == w[:6]
        assert (g[6] is None and w[6] is None) or abs(g[6] - w[6]) <= 1e-9 * max(1, abs(w[6]))


def test_filter_nongrouped_vs_sqlite(oracle):
    rng = np.random.default_rng(43)
    n = 4000
    a = rng.integers(0, 100, n).tolist()
    b = [None if rng.random() < 0.2 else int(x) for x in rng.integers(-50, 50, n)]
    st = ArrowStorage()
    st.import_arrow(pa.table({"a": pa.array(a, pa.int64()), "b": pa.array(b, pa.int32())}), "t", fragment_size=999)
    q = QueryUnit("t", quals=[Cmp(ColRef("a"), ">", Lit(30)), Cmp(ColRef("b"), "<=", Lit(10))],
                  targets=[Agg("count", None, "c"), Agg("sum", ColRef("b") + ColRef("a"), "s"), Agg("min", ColRef("b"), "m")])
    cp, buf, err = run_oracle(oracle, st, q)
    assert err == 0
    got = rs.to_columns(cp, buf)
    want = _sqlite({"t": {"a": a, "b": b}}, "select count(*), sum(b + a), min(b) from t where a > 30 and b <= 10")[0]
    assert (got["c"][0], got["s"][0], got["m"][0]) == want


def test_join_vs_sqlite(oracle):
    rng = np.random.default_rng(44)
    nd, nf = 200, 6000
    dkey = rng.permutation(nd).astype(np.int64) + 1000
    dval = rng.integers(0, 1000, nd).astype(np.int64)
    fk = rng.integers(990, 1000 + nd + 10, nf).astype(np.int64)  # some keys miss the dim table
    val = rng.integers(-100, 100, nf).astype(np.int64)
    st = ArrowStorage()
    st.import_numpy("dim", {"key": dkey, "dval": dval}, fragment_size=64)
    st.import_numpy("fact", {"fk": fk, "val": val}, fragment_size=1700)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                  targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim"), "s"), Agg("count", None, "c")])
    cp, buf, err = run_oracle(oracle, st, q)
    assert err == 0
    got = rs.to_columns(cp, buf)
    want = _sqlite({"dim": {"key": dkey.tolist(), "dval": dval.tolist()}, "fact": {"fk": fk.tolist(), "val": val.tolist()}},
                   "select sum(val + dval), count(*) from fact join dim on fact.fk = dim.key")[0]
    assert (got["s"][0], got["c"][0]) == want
    # grouped by a dim column
    q2 = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") / 100],
                   targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s")])
    cp, buf, err = run_oracle(oracle, st, q2)
    assert err == 0
    got = _rows(rs.to_columns(cp, buf), [0])
    want = sorted(_sqlite({"dim": {"key": dkey.tolist(), "dval": dval.tolist()},
                           "fact": {"fk": fk.tolist(), "val": val.tolist()}},
                          "select dval / 100, sum(val) from fact join dim on fact.fk = dim.key group by dval / 100"))
    assert got == want


def test_or_not_filters_vs_sqlite(oracle):
    """OR / NOT over comparisons with NULLs: three-valued logic (logical_or / logical_not, QE/RuntimeFunctions.cpp:355-384)
    through the plan's filter program, against SQLite on the same rows."""
    from hdk_amd.ir import And, Not, Or
    rng = np.random.default_rng(44)
    n = 4000
    a = [None if rng.random() < 0.15 else int(x) for x in rng.integers(-20, 20, n)]
    b = [None if rng.random() < 0.15 else int(x) for x in rng.integers(-20, 20, n)]
    d = [None if rng.random() < 0.15 else float(x) for x in rng.normal(size=n)]
    k = rng.integers(0, 5, n).tolist()
    st = ArrowStorage()
    st.import_arrow(pa.table({"a": pa.array(a, pa.int32()), "b": pa.array(b, pa.int64()), "d": pa.array(d, pa.float64()),
                              "k": pa.array(k, pa.int16())}), "t", fragment_size=900)
    A_, B_, D_ = ColRef("a"), ColRef("b"), ColRef("d")
    cases = [
        ([Or(Cmp(A_, "<", Lit(0)), Cmp(B_, ">", Lit(5)))], "a < 0 or b > 5"),
        ([Not(Cmp(A_, "<", Lit(0)))], "not (a < 0)"),
        ([Not(Or(Cmp(A_, "<", Lit(0)), Cmp(B_, ">", Lit(5))))], "not (a < 0 or b > 5)"),
        ([Or(And(Cmp(A_, ">=", Lit(-5)), Cmp(A_, "<=", Lit(5))), Not(Cmp(D_, "<", Lit(0.25)))), Cmp(B_, "<>", Lit(3))],
         "((a >= -5 and a <= 5) or not (d < 0.25)) and b <> 3"),
        ([Or(Cmp(A_, "=", B_), Or(Cmp(A_, "<", Lit(-15)), Not(Cmp(B_, "<=", Lit(10)))))], "a = b or a < -15 or not (b <= 10)"),
    ]
    for quals, where in cases:
        q = QueryUnit("t", quals=quals, groupby=[ColRef("k")],
                      targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("sum", ColRef("b"), "s")])
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == 0 and cp.plan.num_filter_ops > 0
        got = _rows(rs.to_columns(cp, buf), [0])
        want = _sqlite({"t": {"a": a, "b": b, "d": d, "k": k}},
                       f"select k, count(*), sum(b) from t where {where} group by k order by k")
        assert got == [tuple(r) for r in want], where


def test_float_accumulators_follow_row_order_float_arithmetic(oracle):
    """SUM / MIN / MAX / AVG over a FLOAT column: the oracle applies agg_*_float[_skip_val] row by row on a float in
    the slot (QE/RuntimeFunctions.cpp:770-875); numpy float32 arithmetic in the same row order gives the same bits."""
    rng = np.random.default_rng(17)
    n = 5000
    k = rng.integers(0, 7, n).astype(np.int32)
    f = (rng.random(n) * 100 - 30).astype(np.float32)
    null = rng.random(n) < 0.1
    null[k == 3] = True                      # a group with nothing but NULLs
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array(k, pa.int32()), "f": pa.array(f, pa.float32(), mask=null)}), "t",
                    fragment_size=1300)
    q = QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0, "k"), Agg("sum", ColRef("f"), "s"),
                                                       Agg("min", ColRef("f"), "lo"), Agg("max", ColRef("f"), "hi"),
                                                       Agg("avg", ColRef("f"), "a"), Agg("count", ColRef("f"), "c")])
    cp, buf, err = run_oracle(oracle, st, q)
    assert err == 0
    cols = rs.to_columns(cp, buf)
    got = {kk: (s, lo, hi, a, c) for kk, s, lo, hi, a, c in zip(cols["k"], cols["s"], cols["lo"], cols["hi"], cols["a"], cols["c"])}
    for g in range(7):
        v = f[(k == g) & ~null]
        if len(v) == 0:
            assert got[g] == (None, None, None, None, 0)
            continue
        acc = np.float32(v[0])
        for x in v[1:]:
            acc = np.float32(acc + x)
        s, lo, hi, a, c = got[g]
        assert (np.float32(s), np.float32(lo), np.float32(hi), c) == (acc, v.min(), v.max(), len(v))
        assert a == float(acc) / len(v)


def test_jit_shaped_baseline_loop_is_bit_exact_with_the_interpreter(oracle):
    """bench.py's CPU baseline (orc_c2_jit_shaped: the row loop HDK's JIT would emit for C2, hand-inlined) against the
    oracle's plan interpreter on the same fragments: keyed and keyless layouts, NULLs, absent keys, 1 and 4 threads."""
    from hdk_amd.plan import compile_query
    from util import oracle_init_buffer, run_oracle
    rng = np.random.default_rng(12)
    n = 300_000
    for nulls, domain in ((0.02, 64), (0.0, 64), (0.3, 50)):
        key = rng.integers(3, 3 + domain, n, dtype=np.int64)
        key[key == 17] = 18  # an empty entry in the middle of the table
        val = rng.integers(-2**31, 2**31, n, dtype=np.int64)
        if nulls:
            val[rng.random(n) < nulls] = A.NULL_BIGINT
        st = ArrowStorage()
        st.import_numpy("t", {"key": key, "val": val}, fragment_size=41_000)
        q = QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s")])
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        t = st.get("t")
        for threads in (1, 4):
            for ft in (True, False):
                sec, out = oracle.c2_jit_shaped(t.columns["key"].fragments, t.columns["val"].fragments, cp.plan,
                                                oracle_init_buffer(oracle, cp), threads, first_touch=ft, reps=2)
                assert sec > 0 and np.array_equal(out, want), (nulls, domain, threads, ft, bool(cp.plan.keyless))
