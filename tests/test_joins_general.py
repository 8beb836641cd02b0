"""Oracle-level pins of the general join loops: SQLite as the second opinion (the reference's own
test method, Tests/ArrowSQLRunner/SQLiteComparator.cpp) and the pinned input/output pairs of
python/tests/test_pyhdk_api.py:609-667 (test_join).  CPU only."""
import math
import sqlite3

import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.plan import compile_query

from joins_cases import make_case, pyhdk_join_tables, sort_rows
from test_projection import run_projection_oracle
from util import run_oracle


def _sqlite(tables, sql):
    con = sqlite3.connect(":memory:")
    for name, cols in tables.items():
        names = list(cols)
        con.execute(f"create table {name} ({', '.join(names)})")
        con.executemany(f"insert into {name} values ({', '.join('?' * len(names))})", list(zip(*[cols[n] for n in names])))
    return con.execute(sql).fetchall()


def _close(a, b):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, float) or isinstance(b, float):
        return math.isclose(a, b, rel_tol=1e-9, abs_tol=1e-9)
    return a == b


def _assert_rows(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert all(_close(x, y) for x, y in zip(g, w)), (g, w)


@pytest.fixture(scope="module")
def case():
    return make_case()


def test_table_kinds_chosen_from_the_data(case):
    st, _, cases, _ = case
    kinds = {name: [j["kind"] for j in compile_query(st, q).join_infos] for name, q, _, _ in cases}
    assert kinds["otm_sum"] == [A.JOIN_ONE_TO_MANY]
    assert kinds["keyed_composite"] == [A.JOIN_KEYED_ONE_TO_ONE]
    assert kinds["keyed_otm"] == [A.JOIN_KEYED_ONE_TO_MANY]
    assert kinds["keyed_wide"] == [A.JOIN_KEYED_ONE_TO_ONE]
    assert kinds["two_levels"] == [A.JOIN_ONE_TO_MANY, A.JOIN_ONE_TO_ONE]


def test_aggregates_over_joins_vs_sqlite(oracle, case):
    st, sql_tables, cases, _ = case
    for name, q, sql, order in cases:
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == 0, name
        got = sort_rows(rs.to_columns(cp, buf))
        want = sorted(_sqlite(sql_tables, sql), key=lambda r: tuple((x is None, x) for x in r))
        if not order:  # non-grouped: SQLite returns one row even over an empty input
            assert len(got) == 1
        _assert_rows(got, want)


def test_projections_over_joins_vs_sqlite(oracle, case):
    st, sql_tables, _, proj_cases = case
    for name, q, sql in proj_cases:
        cp, buf, err, n = run_projection_oracle(oracle, st, q)
        assert err == 0, name
        got = sort_rows(rs.to_columns(cp, buf, nrows=n))
        want = sorted(_sqlite(sql_tables, sql), key=lambda r: tuple((x is None, x) for x in r))
        assert n == len(want), name
        _assert_rows(got, want)


def test_pyhdk_api_join_pairs(oracle):
    st, cases = pyhdk_join_tables()
    for q, expected in cases:
        cp, buf, err, n = run_projection_oracle(oracle, st, q)
        assert err == 0
        got = rs.to_columns(cp, buf, nrows=n)
        assert list(got) == list(expected)
        _assert_rows(sort_rows(got), sort_rows(expected))
