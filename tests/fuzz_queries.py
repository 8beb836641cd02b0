"""Seeded random query generator for the differential tests: random tables (mixed widths, NULLs, fp),
random filters, joins, keys and aggregates inside the fixed library's limits.  Shapes the plan compiler
rejects (QueryMustRunOnCpu) are skipped by the callers -- rejecting is a legal outcome, a wrong answer is not."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, Proj, QueryUnit
from hdk_amd.storage import ArrowStorage


def make_tables(rng, n, nd):
    def with_nulls(a, null, frac):
        a = a.copy()
        if frac:
            a[rng.random(len(a)) < frac] = null
        return a
    fact = {
        "k8": with_nulls(rng.integers(0, 40, n).astype(np.int8), A.NULL_TINYINT, 0.02),
        "k16": rng.integers(-300, 300, n).astype(np.int16),
        "k32": with_nulls(rng.integers(0, 5000, n).astype(np.int32), A.NULL_INT, 0.03),
        "k64": rng.integers(0, 900, n, dtype=np.int64) * 7_000_000_013,
        "fk": with_nulls(rng.integers(-5, nd + 5, n).astype(np.int64), A.NULL_BIGINT, 0.02),
        "fk2": rng.integers(0, 12, n).astype(np.int32),
        "v64": with_nulls(rng.integers(-10**9, 10**9, n, dtype=np.int64), A.NULL_BIGINT, 0.05),
        "v32": with_nulls(rng.integers(-1000, 1000, n).astype(np.int32), A.NULL_INT, 0.05),
        "v16": rng.integers(-100, 100, n).astype(np.int16),
        "d": with_nulls(rng.normal(size=n) * 100, np.frombuffer(np.int64(A.NULL_DOUBLE_BITS).tobytes(), np.float64)[0], 0.05),
    }
    dim = {
        "key": rng.permutation(nd).astype(np.int64),
        "dup": rng.integers(0, max(nd // 4, 1), nd).astype(np.int64),
        "g": rng.integers(0, 9, nd).astype(np.int32),
        "w": with_nulls(rng.integers(-50, 50, nd).astype(np.int64), A.NULL_BIGINT, 0.04),
        "a": (np.arange(nd) % 12).astype(np.int32),
    }
    st = ArrowStorage()
    st.import_numpy("fact", fact, fragment_size=int(rng.integers(n // 7 + 1, n // 2 + 2)))
    st.import_numpy("dim", dim, fragment_size=int(rng.integers(nd // 3 + 1, nd + 1)))
    return st


INT_COLS = ["k8", "k16", "k32", "v64", "v32", "v16"]
KEY_COLS = ["k8", "k16", "k32", "k64"]


def random_query(rng, allow_join=True, projection=False):
    joins, dim_cols = [], []
    if allow_join and rng.random() < 0.45:
        kind = rng.integers(0, 4)
        typ = "left" if rng.random() < 0.3 else "inner"
        if kind == 0:
            joins = [JoinSpec("dim", ColRef("fk"), "key", typ)]                       # one-to-one
        elif kind == 1:
            joins = [JoinSpec("dim", ColRef("fk") / 4, "dup", typ)]                    # one-to-many, expression key
        elif kind == 2:
            joins = [JoinSpec("dim", [ColRef("fk"), ColRef("fk2")], ["key", "a"], typ)]  # keyed one-to-one
        else:
            joins = [JoinSpec("dim", [ColRef("fk2"), ColRef("k8")], ["a", "g"], typ)]    # keyed one-to-many
        dim_cols = [ColRef("g", "dim"), ColRef("w", "dim")]
    quals = []
    for _ in range(int(rng.integers(0, 3))):
        if dim_cols and rng.random() < 0.4:
            quals.append(Cmp(dim_cols[1], rng.choice(["<", ">", "<>"]), Lit(int(rng.integers(-30, 30)))))
        else:
            c = str(rng.choice(INT_COLS + ["d"]))
            lhs = ColRef(c)
            if c != "d" and rng.random() < 0.3:
                lhs = lhs * int(rng.integers(1, 4)) + int(rng.integers(-5, 5))
            lit = Lit(float(rng.normal() * 50)) if c == "d" else Lit(int(rng.integers(-200, 2000)))
            quals.append(Cmp(lhs, str(rng.choice(["<", "<=", ">", ">=", "=", "<>"])), lit))
    def arg():
        r = rng.random()
        if dim_cols and r < 0.3:
            return dim_cols[1] if rng.random() < 0.7 else dim_cols[1] + ColRef("v32")
        if r < 0.5:
            return ColRef(str(rng.choice(["v64", "v32", "v16"])))
        if r < 0.65:
            return ColRef("d")
        if r < 0.8:
            return ColRef("v32") * ColRef("v16")
        return ColRef("v64") / int(rng.integers(2, 9)) - ColRef("k16")
    if projection:
        targets = [Proj(ColRef("k32"), "k32"), Proj(arg(), "e1"), Proj(ColRef("d"), "d")]
        if dim_cols:
            targets.append(Proj(dim_cols[0], "g"))
        return QueryUnit("fact", quals=quals + [Cmp(ColRef("k16"), ">", Lit(int(rng.integers(150, 280))))], joins=joins,
                         targets=targets, output_columnar=bool(rng.random() < 0.5))
    if not joins and rng.random() < 0.2:
        # the shape of the specialised open-addressing kernels: one plain key, plain arguments, plain filters
        keys = [str(k) for k in rng.choice(["k32", "k64", "k16", "k8"], size=int(rng.choice([1, 1, 2])), replace=False)]
        quals = [Cmp(ColRef(str(rng.choice(["v32", "v16", "d"]))), str(rng.choice(["<", ">", "<>"])),
                     Lit(float(rng.normal() * 40)) if rng.random() < 0.3 else Lit(int(rng.integers(-50, 50))))
                 for _ in range(int(rng.integers(0, 3)))]
        targets = [KeyRef(i, f"key{i}") for i in range(len(keys))]
        for i in range(int(rng.integers(1, 4))):
            kind = str(rng.choice(["count", "sum", "min", "max", "avg"]))
            a = None if (kind == "count" and rng.random() < 0.5) else ColRef(str(rng.choice(["v64", "v32", "d"])))
            targets.append(Agg(kind, a, f"t{i}"))
        return QueryUnit("fact", quals=quals, groupby=[ColRef(k) for k in keys], targets=targets, force_baseline=True)
    nkeys = int(rng.choice([0, 1, 1, 2]))
    groupby = []
    for _ in range(nkeys):
        if dim_cols and rng.random() < 0.35:
            groupby.append(dim_cols[0])
        else:
            groupby.append(ColRef(str(rng.choice(KEY_COLS))))
    targets = [KeyRef(i, f"key{i}") for i in range(len(groupby))]
    aggs = ["count", "sum", "min", "max", "avg"]
    for i in range(int(rng.integers(1, 5))):
        kind = str(rng.choice(aggs))
        a = None if (kind == "count" and rng.random() < 0.5) else arg()
        targets.append(Agg(kind, a, f"t{i}"))
    return QueryUnit("fact", quals=quals, joins=joins, groupby=groupby, targets=targets,
                     output_columnar=bool(groupby and rng.random() < 0.3),
                     force_baseline=bool(groupby and rng.random() < 0.25))


# ---- the wider generator of the soak runs: float32 measures, OR / NOT trees, transformed keys, 64-bit COUNT -----------

def make_tables_wide(rng, n, nd):
    """make_tables plus a float column, a timestamp and a decimal on the fact side (drawn AFTER the base columns)."""
    from hdk_amd.ir import Type
    st = make_tables(rng, n, nd)
    fact = st.get("fact")
    frag = fact.frag_rows[0]
    cols = {name: np.concatenate(fact.columns[name].fragments) for name in fact.column_order}
    null_f = np.array([A.NULL_FLOAT_BITS], dtype=np.int32).view(np.float32)[0]
    f32 = (rng.normal(size=n) * 30).astype(np.float32)
    f32[rng.random(n) < 0.04] = null_f
    ts = rng.integers(1_230_768_000, 1_483_228_800, n, dtype=np.int64)  # 2009-01-01 .. 2017-01-01
    ts[rng.random(n) < 0.02] = A.NULL_BIGINT
    dec = rng.integers(-300, 900, n, dtype=np.int64)
    dec[rng.random(n) < 0.03] = A.NULL_BIGINT
    cols.update({"f32": f32, "ts": ts, "dec": dec})
    # a DATE in days on both sides (bucketized join tables) and a dimension key with NULL rows (IS NOT DISTINCT FROM)
    from hdk_amd.ir import DATE32
    day = (17000 + rng.integers(0, nd // 2 + 20, n)).astype(np.int32)
    day[rng.random(n) < 0.03] = A.NULL_INT
    cols["day"] = day
    dday = (17010 + rng.integers(0, nd // 2, nd)).astype(np.int32)
    dday[rng.random(nd) < 0.02] = A.NULL_INT
    nkey = rng.permutation(nd).astype(np.int64)
    nkey[rng.random(nd) < 0.01] = A.NULL_BIGINT
    st.import_numpy("ddim", {"day": dday, "uday": (17010 + rng.permutation(nd)).astype(np.int32), "nkey": nkey,
                             "x": rng.integers(-9, 9, nd).astype(np.int64)},
                    fragment_size=int(rng.integers(nd // 3 + 1, nd + 1)), types={"day": DATE32, "uday": DATE32})
    # fragment shapes the base generator never draws: many fragments shorter than a tile, exactly one tile, one row more
    frag = int(rng.choice([frag, frag, n // 61 + 1, 2048, 8193]))
    st.import_numpy("fact", cols, fragment_size=frag,
                    types={"ts": Type("timestamp", 8, unit="s"), "dec": Type("decimal", 8, scale=2), "day": DATE32})
    return st


def random_query_wide(rng, projection=False):
    """random_query, then seeded rewrites towards the rows the base generator never draws."""
    from dataclasses import replace
    from hdk_amd.ir import And, Cast, ExtractYear, INT32, Not, Or
    q = random_query(rng, projection=projection)
    quals = list(q.quals)
    if len(quals) >= 2 and rng.random() < 0.6:
        a, b = quals[0], quals[1]
        tree = [Or(a, b), Or(Not(a), b), Not(And(a, b)), And(Or(a, b), Not(b))][int(rng.integers(0, 4))]
        quals = [tree] + quals[2:]
    elif quals and rng.random() < 0.3:
        quals = [Not(quals[0])] + quals[1:]
    if rng.random() < 0.25:
        quals.append(Cmp(ColRef("f32"), str(rng.choice(["<", ">", "<>"])), Lit(float(rng.normal() * 20))))
    targets = list(q.targets)
    for i, t in enumerate(targets):
        if isinstance(t, Agg) and t.arg is not None and rng.random() < 0.3:
            targets[i] = Agg(t.kind, ColRef("f32"), t.name)
        elif isinstance(t, Proj) and t.name == "d" and rng.random() < 0.5:
            targets[i] = Proj(ColRef("f32"), "d")
    groupby = list(q.groupby)
    if groupby and not q.force_baseline and rng.random() < 0.4:
        j = int(rng.integers(0, len(groupby)))
        groupby[j] = ExtractYear(ColRef("ts")) if rng.random() < 0.5 else Cast(ColRef("dec"), INT32)
    joins = list(q.joins)
    if not joins and rng.random() < 0.5:
        # the join variants of SURVEY.md 8 a11 / a12 on a query that reads no inner column (so that SEMI's "first row of
        # a key wins" stays invisible): bucketized DATE tables, IS NOT DISTINCT FROM, SEMI / ANTI
        typ = str(rng.choice(["inner", "inner", "left", "semi", "anti"]))
        v = int(rng.integers(0, 5))
        if v == 0:
            joins = [JoinSpec("ddim", ColRef("day"), "day", typ)]                                  # bucketized one-to-many
        elif v == 1:
            joins = [JoinSpec("ddim", ColRef("day"), "uday", typ)]                                 # bucketized one-to-one
        elif v == 2:
            joins = [JoinSpec("ddim", ColRef("fk"), "nkey", typ, null_safe=True)]                  # NULLs match NULLs
        elif v == 3:
            joins = [JoinSpec("ddim", ColRef("day"), "day", typ, null_safe=typ not in ("semi", "anti"))]
        else:
            joins = [JoinSpec("dim", [ColRef("fk2"), ColRef("k8")], ["a", "g"], "semi" if typ != "anti" else "anti")]  # keyed
    return replace(q, quals=quals, targets=targets, groupby=groupby, joins=joins, bigint_count=bool(rng.random() < 0.3))
