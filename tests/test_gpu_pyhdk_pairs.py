"""The reference's own Python-level input/output pairs for the path, run through the Engine facade on the
GPU: python/tests/test_pyhdk_sql.py:52-69 (count / projection / filter on the 3-row table with fragment
size 2) and python/tests/test_pyhdk_api.py:457-496 (test_agg).  ORDER BY stays above the boundary: the
reference sorts with `.sort(...)`, the test sorts the fetched rows."""
import pytest

from hdk_amd.engine import Engine
from hdk_amd.ir import Agg, Cmp, ColRef, KeyRef, Lit, Proj, QueryMustRunOnCpu, QueryUnit

pytestmark = pytest.mark.gpu


def _sorted(table, by):
    d = table.to_pydict()
    rows = sorted(zip(*[d[n] for n in d]), key=lambda r: tuple(r[list(d).index(b)] for b in by))
    return {n: [r[i] for r in rows] for i, n in enumerate(d)}


def test_pyhdk_sql_simple_queries():
    import pandas
    import pyarrow
    eng = Engine()
    eng.import_arrow(pyarrow.Table.from_pandas(pandas.DataFrame({"a": [1, 2, 3], "b": [10, 20, 30]})), "test",
                     fragment_size=2)  # TableOptions(2)
    # SELECT COUNT(*) FROM test;
    t = eng.run(QueryUnit("test", targets=[Agg("count", None, "EXPR$0")]))
    assert t.shape == (1, 1) and t["EXPR$0"].to_pylist() == [3]
    # SELECT * FROM test;
    t = eng.run(QueryUnit("test", targets=[Proj(ColRef("a"), "a"), Proj(ColRef("b"), "b")]))
    assert t.shape == (3, 2)
    assert _sorted(t, ["a"]) == {"a": [1, 2, 3], "b": [10, 20, 30]}
    # SELECT COUNT(*) FROM test WHERE a < 3;
    t = eng.run(QueryUnit("test", quals=[Cmp(ColRef("a"), "<", Lit(3))], targets=[Agg("count", None, "EXPR$0")]))
    assert t["EXPR$0"].to_pylist() == [2]
    with pytest.raises(QueryMustRunOnCpu):
        eng.run(QueryUnit("test", targets=[Agg("count", None, "c")]), device_type="CPU")


def test_pyhdk_api_agg_pairs():
    eng = Engine()
    eng.import_pydict({"a": [1, 2, 1, 2, 1, 2, 1, 2, 1, 2], "b": [1, 1, 1, 1, 1, 2, 2, 2, 2, 2],
                       "c": [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]}, "ht")
    # ht.agg(["a", -2], "sum(c)", ht.ref("c").min(), hdk.count()).sort("a", "b")
    t = eng.run(QueryUnit("ht", groupby=[ColRef("a"), ColRef("b")],
                          targets=[KeyRef(0, "a"), KeyRef(1, "b"), Agg("sum", ColRef("c"), "c_sum"),
                                   Agg("min", ColRef("c"), "c_min"), Agg("count", None, "count")]))
    assert _sorted(t, ["a", "b"]) == {"a": [1, 1, 2, 2], "b": [1, 2, 1, 2], "c_sum": [9, 16, 6, 24],
                                      "c_min": [1, 7, 2, 6], "count": [3, 2, 2, 3]}
    # ht.agg(ht.ref("a"), aggs={"bc": "count(b)", "cmx": ht.ref("c").max()}, cmn="min(c)", cv=ht.ref("c").avg()).sort("a")
    t = eng.run(QueryUnit("ht", groupby=[ColRef("a")],
                          targets=[KeyRef(0, "a"), Agg("count", ColRef("b"), "bc"), Agg("max", ColRef("c"), "cmx"),
                                   Agg("min", ColRef("c"), "cmn"), Agg("avg", ColRef("c"), "cv")]))
    assert _sorted(t, ["a"]) == {"a": [1, 2], "bc": [5, 5], "cmx": [9, 10], "cmn": [1, 2], "cv": [5.0, 6.0]}
