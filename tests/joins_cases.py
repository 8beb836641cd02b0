"""Join shapes beyond the inner one-to-one probe: one-to-many matching
sets, LEFT joins, filters on joined columns, composite and wide keys (keyed tables), two join levels.
Each case is (storage, sqlite tables, [(QueryUnit, sql, order_by_cols)]); the CPU suite checks the
oracle against SQLite, the GPU suite checks the kernels against the oracle."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, Proj, QueryUnit
from hdk_amd.storage import ArrowStorage


def _py(arr, null):
    return [None if v == null else int(v) for v in arr.tolist()]


def make_case(seed=11, nf=8000, nd=300):
    rng = np.random.default_rng(seed)
    # dim: duplicate keys (one-to-many), a composite key (a, b) that is unique, and NULL keys
    dk = rng.integers(100, 100 + nd // 3, nd).astype(np.int64)          # ~3 rows per key
    dk[rng.random(nd) < 0.03] = A.NULL_BIGINT
    da = rng.integers(0, 40, nd).astype(np.int32)
    db = np.arange(nd, dtype=np.int32) // 40                             # (a, b): few duplicates removed below
    _, first = np.unique(np.stack([da, db], axis=1), axis=0, return_index=True)
    keep = np.sort(first)
    dk, da, db = dk[keep], da[keep], db[keep]
    nd = len(dk)
    dv = rng.integers(-500, 500, nd).astype(np.int64)
    dw = rng.integers(0, 6, nd).astype(np.int16)
    wide = (rng.permutation(nd).astype(np.int64) - nd // 2) * 30_000_000_000  # range >> 2^31: keyed, unique
    fk = rng.integers(95, 105 + nd // 3, nf).astype(np.int64)            # some keys miss
    fk[rng.random(nf) < 0.04] = A.NULL_BIGINT
    fa = rng.integers(0, 44, nf).astype(np.int32)
    fb = rng.integers(0, 9, nf).astype(np.int32)
    fwide = wide[rng.integers(0, nd, nf)].copy()
    fwide[rng.random(nf) < 0.2] += 7                                      # misses
    fv = rng.integers(-100, 100, nf).astype(np.int64)
    st = ArrowStorage()
    st.import_numpy("dim", {"k": dk, "a": da, "b": db, "v": dv, "w": dw, "wide": wide}, fragment_size=97)
    st.import_numpy("fact", {"fk": fk, "fa": fa, "fb": fb, "fwide": fwide, "val": fv}, fragment_size=2100)
    # second-level dim keyed by dim.w (unique)
    st.import_numpy("dim2", {"w": np.arange(5, dtype=np.int16), "z": np.array([10, 20, 30, 40, 50], dtype=np.int64)})
    sql_tables = {
        "dim": {"k": _py(dk, A.NULL_BIGINT), "a": da.tolist(), "b": db.tolist(), "v": dv.tolist(), "w": dw.tolist(),
                "wide": wide.tolist()},
        "fact": {"fk": _py(fk, A.NULL_BIGINT), "fa": fa.tolist(), "fb": fb.tolist(), "fwide": fwide.tolist(),
                 "val": fv.tolist()},
        "dim2": {"w": list(range(5)), "z": [10, 20, 30, 40, 50]},
    }
    D = lambda n: ColRef(n, "dim")  # noqa: E731
    cases = [
        # one-to-many inner join, non-grouped
        ("otm_sum", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k")],
                              targets=[Agg("sum", ColRef("val") + D("v"), "s"), Agg("count", None, "c")]),
         "select sum(val + v), count(*) from fact join dim on fk = k", []),
        # one-to-many grouped by a dim column, filter on a dim column (post-join qual)
        ("otm_group_filter", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k")],
                                       quals=[Cmp(D("v"), ">", Lit(-200)), Cmp(ColRef("val"), "<", Lit(60))],
                                       groupby=[D("w")],
                                       targets=[KeyRef(0, "w"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c"),
                                                Agg("min", D("v"), "mn")]),
         "select w, sum(val), count(*), min(v) from fact join dim on fk = k where v > -200 and val < 60 group by w", [0]),
        # LEFT one-to-many: unmatched fact rows survive with NULL dim columns
        ("left_otm", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k", "left")],
                               targets=[Agg("count", None, "c"), Agg("count", D("v"), "cv"), Agg("sum", D("v"), "sv"),
                                        Agg("sum", ColRef("val"), "s")]),
         "select count(*), count(v), sum(v), sum(val) from fact left join dim on fk = k", []),
        # LEFT join grouped by the (nullable) dim column
        ("left_group", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k", "left")], groupby=[D("w")],
                                 targets=[KeyRef(0, "w"), Agg("count", None, "c"), Agg("avg", D("v"), "av")]),
         "select w, count(*), avg(v) from fact left join dim on fk = k group by w", [0]),
        # composite key (a, b): keyed one-to-one table
        ("keyed_composite", QueryUnit("fact", joins=[JoinSpec("dim", [ColRef("fa"), ColRef("fb")], ["a", "b"])],
                                      groupby=[D("w")],
                                      targets=[KeyRef(0, "w"), Agg("sum", ColRef("val") * D("v"), "s"), Agg("count", None, "c")]),
         "select w, sum(val * v), count(*) from fact join dim on fa = a and fb = b group by w", [0]),
        # composite key LEFT
        ("keyed_composite_left", QueryUnit("fact", joins=[JoinSpec("dim", [ColRef("fa"), ColRef("fb")], ["a", "b"], "left")],
                                           targets=[Agg("count", None, "c"), Agg("count", D("v"), "cv"), Agg("max", D("v"), "mx")]),
         "select count(*), count(v), max(v) from fact left join dim on fa = a and fb = b", []),
        # composite key with duplicates: (a) with (w) -> keyed one-to-many
        ("keyed_otm", QueryUnit("fact", joins=[JoinSpec("dim", [ColRef("fa"), ColRef("fb")], ["a", "w"])],
                                targets=[Agg("count", None, "c"), Agg("sum", D("v"), "s")]),
         "select count(*), sum(v) from fact join dim on fa = a and fb = w", []),
        # single key, range too wide for a perfect table -> keyed
        ("keyed_wide", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fwide"), "wide")],
                                 targets=[Agg("count", None, "c"), Agg("sum", D("v") + ColRef("val"), "s")]),
         "select count(*), sum(v + val) from fact join dim on fwide = wide", []),
        # two levels: fact -> dim (one-to-many) -> dim2 (on a dim column)
        ("two_levels", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k"), JoinSpec("dim2", D("w"), "w")],
                                 groupby=[ColRef("z", "dim2")],
                                 targets=[KeyRef(0, "z"), Agg("count", None, "c"), Agg("sum", ColRef("val"), "s")]),
         "select z, count(*), sum(val) from fact join dim on fk = k join dim2 on dim.w = dim2.w group by z", [0]),
        # two levels, second LEFT (dim.w = 5 has no dim2 row)
        ("two_levels_left", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k"),
                                                      JoinSpec("dim2", D("w"), "w", "left")],
                                      targets=[Agg("count", None, "c"), Agg("count", ColRef("z", "dim2"), "cz"),
                                               Agg("sum", ColRef("z", "dim2"), "sz")]),
         "select count(*), count(z), sum(z) from fact join dim on fk = k left join dim2 on dim.w = dim2.w", []),
    ]
    proj_cases = [
        ("proj_otm", QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "k")], quals=[Cmp(ColRef("val"), ">", Lit(90))],
                               targets=[Proj(ColRef("val"), "val"), Proj(D("v"), "v"), Proj(D("w"), "w")]),
         "select val, v, w from fact join dim on fk = k where val > 90"),
        ("proj_left", QueryUnit("fact", joins=[JoinSpec("dim", [ColRef("fa"), ColRef("fb")], ["a", "b"], "left")],
                                quals=[Cmp(ColRef("val"), ">", Lit(95))],
                                targets=[Proj(ColRef("fa"), "fa"), Proj(ColRef("fb"), "fb"), Proj(D("v"), "v")]),
         "select fa, fb, v from fact left join dim on fa = a and fb = b where val > 95"),
    ]
    return st, sql_tables, cases, proj_cases


def pyhdk_join_tables():
    """Inputs of python/tests/test_pyhdk_api.py:609-667 (test_join)."""
    st = ArrowStorage()
    st.import_numpy("ht1", {"a": np.array([1, 2, 3, 4, 5], dtype=np.int64), "b": np.array([5, 4, 3, 2, 1], dtype=np.int64),
                            "x": np.array([1.1, 2.2, 3.3, 4.4, 5.5])})
    st.import_numpy("ht2", {"a": np.array([1, 2, 3, 4, 5], dtype=np.int64), "b": np.array([1, 2, 3, 4, 5], dtype=np.int64),
                            "y": np.array([5.5, 4.4, 3.3, 2.2, 1.1])})
    H2 = lambda n: ColRef(n, "ht2")  # noqa: E731
    T = lambda n: ColRef(n, "ht1")  # noqa: E731
    cases = [
        # ht1.join(ht2): natural join on (a, b)
        (QueryUnit("ht1", joins=[JoinSpec("ht2", [T("a"), T("b")], ["a", "b"])],
                   targets=[Proj(T("a"), "a"), Proj(T("b"), "b"), Proj(T("x"), "x"), Proj(H2("y"), "y")]),
         {"a": [3], "b": [3], "x": [3.3], "y": [3.3]}),
        # ht1.join(ht2, how="left")
        (QueryUnit("ht1", joins=[JoinSpec("ht2", [T("a"), T("b")], ["a", "b"], "left")],
                   targets=[Proj(T("a"), "a"), Proj(T("b"), "b"), Proj(T("x"), "x"), Proj(H2("y"), "y")]),
         {"a": [1, 2, 3, 4, 5], "b": [5, 4, 3, 2, 1], "x": [1.1, 2.2, 3.3, 4.4, 5.5], "y": [None, None, 3.3, None, None]}),
        # ht1.join(ht2, "a")
        (QueryUnit("ht1", joins=[JoinSpec("ht2", T("a"), "a")],
                   targets=[Proj(T("a"), "a"), Proj(T("b"), "b"), Proj(T("x"), "x"), Proj(H2("b"), "b_1"), Proj(H2("y"), "y")]),
         {"a": [1, 2, 3, 4, 5], "b": [5, 4, 3, 2, 1], "x": [1.1, 2.2, 3.3, 4.4, 5.5], "b_1": [1, 2, 3, 4, 5],
          "y": [5.5, 4.4, 3.3, 2.2, 1.1]}),
        # ht1.join(ht2, "a", "b"): ht1.a = ht2.b
        (QueryUnit("ht1", joins=[JoinSpec("ht2", T("a"), "b")],
                   targets=[Proj(T("a"), "a"), Proj(T("b"), "b"), Proj(T("x"), "x"), Proj(H2("a"), "a_1"), Proj(H2("y"), "y")]),
         {"a": [1, 2, 3, 4, 5], "b": [5, 4, 3, 2, 1], "x": [1.1, 2.2, 3.3, 4.4, 5.5], "a_1": [1, 2, 3, 4, 5],
          "y": [5.5, 4.4, 3.3, 2.2, 1.1]}),
        # ht1.join(ht2, ["a", "b"], ["b", "a"])
        (QueryUnit("ht1", joins=[JoinSpec("ht2", [T("a"), T("b")], ["b", "a"])],
                   targets=[Proj(T("a"), "a"), Proj(T("b"), "b"), Proj(T("x"), "x"), Proj(H2("y"), "y")]),
         {"a": [3], "b": [3], "x": [3.3], "y": [3.3]}),
    ]
    return st, cases


def sort_rows(cols, names=None):
    names = names or list(cols)
    rows = list(zip(*[cols[n] for n in names]))
    return sorted(rows, key=lambda r: tuple((x is None, x) for x in r))
