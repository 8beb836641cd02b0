"""CPU pins of the join variants of tests/join_variants_cases.py: the planner emits them with the reference's arguments,
the oracle (pinned symbol by symbol to the compiled reference, tests/test_oracle_vs_ref.py) runs them, SQLite is the
second opinion wherever SQL can state the same join."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import QueryMustRunOnCpu
from hdk_amd.plan import compile_query

from join_variants_cases import D, make_variants
from joins_cases import sort_rows
from test_joins_general import _assert_rows, _sqlite
from test_projection import run_projection_oracle
from util import oracle_join_tables, run_oracle


@pytest.fixture(scope="module")
def case():
    return make_variants()


def test_planner_emits_the_reference_arguments(case):
    """bucket_normalization, null modes, translated NULLs and table kinds as PerfectJoinHashTable computes them
    (QE/JoinHashTable/PerfectJoinHashTable.cpp:45-85,798-816; Builders/PerfectHashTableBuilder.h:100-130)."""
    st, _, cases, _ = case
    plans = {name: compile_query(st, q) for name, q, _ in cases}
    j = lambda n: plans[n].plan.joins[0]  # noqa: E731
    i = lambda n: plans[n].join_infos[0]  # noqa: E731
    # DATE keys: bucket 86400, range in epoch seconds, normalised slot count, SmallDate column kind
    jn, info = j("date_one_to_one"), i("date_one_to_one")
    assert jn.bucket == D and jn.kind == A.JOIN_ONE_TO_ONE and jn.null_mode == A.JOIN_NULL_NULLABLE
    assert jn.min_key % D == 0 and jn.max_key % D == 0 and jn.null_val == A.NULL_BIGINT
    assert info["bucketized"] and info["bucket"] == D and info["col_types"] == [A.JC_SMALL_DATE]
    assert info["hash_entry_count"] == jn.max_key - jn.min_key + 1
    assert jn.entry_count == info["entry_count"] == (jn.max_key - jn.min_key) // D + 1
    assert info["null_val"] == -(2**31) and info["elem_sz"] == 4  # the BUILD compares with the column's own NULL
    assert j("date_one_to_many").kind == A.JOIN_ONE_TO_MANY and j("date16_one_to_many").kind == A.JOIN_ONE_TO_MANY
    assert i("date16_one_to_many")["elem_sz"] == 2 and i("date16_one_to_many")["null_val"] == -(2**15)
    assert j("date64_bucket_collapses_a_day").kind == A.JOIN_ONE_TO_MANY  # two seconds of one day share a slot
    assert i("date64_bucket_collapses_a_day")["col_types"] == [A.JC_SIGNED]
    # IS NOT DISTINCT FROM: one more slot, NULLs filed under max + 1 and probed with max + 1
    jn, info = j("bw_eq_one_to_one"), i("bw_eq_one_to_one")
    assert jn.null_mode == A.JOIN_NULL_BITWISE and jn.translated_null == jn.max_key + 1 and jn.kind == A.JOIN_ONE_TO_ONE
    assert info["uses_bw_eq"] == 1 and info["translated_null_build"] == jn.max_key + 1
    assert jn.entry_count == jn.max_key - jn.min_key + 2
    assert j("bw_eq_one_to_many").kind == A.JOIN_ONE_TO_MANY  # two NULL rows share the translated slot
    # DATE + kBwEq: the probe gets max / bucket + 1 (PerfectJoinHashTable.cpp:805-807), the build max + 1
    jn, info = j("date_bw_eq"), i("date_bw_eq")
    assert jn.null_mode == A.JOIN_NULL_BITWISE and jn.bucket == D
    assert jn.translated_null == jn.max_key // D + 1 and info["translated_null_build"] == jn.max_key + 1
    # SEMI / ANTI: one-to-one tables with the first-row-wins fill, whatever the duplicates
    for n, typ, kind in (("semi_dups", A.JOIN_SEMI, A.JOIN_ONE_TO_ONE), ("anti_dups", A.JOIN_ANTI, A.JOIN_ONE_TO_ONE),
                         ("semi_keyed", A.JOIN_SEMI, A.JOIN_KEYED_ONE_TO_ONE), ("anti_keyed", A.JOIN_ANTI, A.JOIN_KEYED_ONE_TO_ONE),
                         ("semi_date", A.JOIN_SEMI, A.JOIN_ONE_TO_ONE)):
        assert j(n).type == typ and j(n).kind == kind and i(n)["for_semi_join"] == 1, n


def test_refused_shapes(case):
    from hdk_amd.ir import Agg, ColRef, JoinSpec, QueryUnit
    st = case[0]
    with pytest.raises(QueryMustRunOnCpu):  # a DATE against an integer
        compile_query(st, QueryUnit("fact", joins=[JoinSpec("ddate", ColRef("k"), "day")], targets=[Agg("count", None)]))
    with pytest.raises(QueryMustRunOnCpu):  # kBwEq on a keyed table
        compile_query(st, QueryUnit("fact", joins=[JoinSpec("dk", [ColRef("a"), ColRef("b")], ["a", "b"], null_safe=True)],
                                    targets=[Agg("count", None)]))


def test_join_variants_vs_sqlite(oracle, case):
    st, sql_tables, cases, _ = case
    for name, q, sql in cases:
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == 0, name
        if sql is None:
            continue
        got = sort_rows(rs.to_columns(cp, buf))
        want = sorted(_sqlite(sql_tables, sql), key=lambda r: tuple((x is None, x) for x in r))
        assert len(got) == len(want) and len(got) >= 1, name
        _assert_rows(got, want)
        assert any(v not in (None, 0) for v in got[0]), name  # (the case joins something)


def test_join_variant_projections_vs_sqlite(oracle, case):
    st, sql_tables, _, proj_cases = case
    for name, q, sql in proj_cases:
        cp, buf, err, n = run_projection_oracle(oracle, st, q)
        assert err == 0 and n > 0, name
        got = sort_rows(rs.to_columns(cp, buf, nrows=n))
        want = sorted(_sqlite(sql_tables, sql), key=lambda r: tuple((x is None, x) for x in r))
        assert n == len(want), name
        _assert_rows(got, want)


def test_date_bw_eq_follows_the_reference_formulas(oracle, case):
    """DATE + kBwEq: NULL rows are FILED in slot (max + 1 - min) / 86400 = the last day's slot, NULL keys PROBE with
    max / 86400 + 1, which fails the probe's own `key >= min_key` test for any date after 1970: NULL fact rows match
    nothing, and rows of the last day also see the dimension's NULL rows (reference behaviour, restated)."""
    st, _, cases, _ = case
    q = dict((n, q) for n, q, _ in cases)["date_bw_eq"]
    cp = compile_query(st, q)
    jn, info = cp.plan.joins[0], cp.join_infos[0]
    table = oracle_join_tables(oracle, st, cp)[0]
    n = info["entry_count"]
    dim = st.get("ddup")
    day = np.concatenate(dim.columns["day"].fragments).astype(np.int64)
    last_slot = (jn.max_key + 1 - jn.min_key) // D
    assert last_slot == n - 1 == (jn.max_key - jn.min_key) // D
    nulls = int((day == -(2**31)).sum())
    on_last_day = int((day * D == jn.max_key).sum())
    assert nulls > 0 and table[n + last_slot] == nulls + on_last_day  # counts section of [pos | count | ids]
    L = oracle.lib()
    assert L.orc_bucketized_hash_join_idx_bitwise(table.ctypes.data, A.NULL_BIGINT, jn.min_key, jn.max_key, A.NULL_BIGINT,
                                                  jn.translated_null, D) == -1
