"""The reference's synthetic benchmark suite as QueryUnits (omniscidb/Benchmarks/synthetic_benchmark/queries/*): the shapes
BASELINE.md takes its BaselineHash / PerfectHashSingleCol figures from, and their neighbours in the same directory --
NonGroupedAgg (NGA01-05), MultiStep (MSBS001-003, MSPHS001-003: only the per-row step below the post-aggregate arithmetic
is the hot path's; `max(x100) + max(x10 + 1)` and `sum(x100) / sum(x10 + 1)` are computed above it from these targets) and
PerfectHashMultiCol (PHM001-006).  Table: create_table.py:118-130 -- INT columns x10 ... x1m / y10 / z10 ..., uniform in [1, N].
Shared by the routing tests (CPU), the parity tests (GPU), scripts/bench_configs.py and workloads.py."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cast, ColRef, FP32, FP64, KeyRef, QueryUnit

SYN_COLUMNS = {"x10": 10, "y10": 10, "z10": 10, "x100": 100, "y100": 100, "z100": 100, "x1k": 1000, "x10k": 10_000, "x100k": 100_000,
               "x1m": 1_000_000,
               # (not in create_table.py: group counts between the suite's decades, for the tables that just fit / just miss LDS)
               "x2k": 2000, "x4k": 4000, "x6k": 6000}


def syn_table(rng, n, columns=None, null_frac=0.0):
    """numpy columns of the benchmark table (int32, uniform in [1, N]); `null_frac` of every column's values NULL."""
    out = {}
    for c in (columns or SYN_COLUMNS):
        v = rng.integers(1, SYN_COLUMNS[c] + 1, n).astype(np.int32)
        if null_frac:
            v[rng.random(n) < null_frac] = A.NULL_INT
        out[c] = v
    return out


NGA_COLS = ("x10", "y10", "z10", "x100", "y100", "z100")


def nga(i, table="syn"):
    """NonGroupedAgg/NGA0<i>.sql"""
    cols = [ColRef(c) for c in NGA_COLS]
    if i == 1:
        return QueryUnit(table, targets=[Agg("count", None, "c")] + [Agg("count", c, f"c{k}") for k, c in enumerate(cols)])
    kind = {2: "sum", 3: "max", 4: "min", 5: "avg"}[i]
    return QueryUnit(table, targets=[Agg(kind, c, f"a{k}") for k, c in enumerate(cols)])


def _ms_targets():
    # count(*), max(x100), max(x10), max(x10 + 1), sum(x100), sum(x10 + 1): what MSBS / MSPHS aggregate per row
    x100, x10 = ColRef("x100"), ColRef("x10")
    return [Agg("count", None, "c"), Agg("max", x100, "mx100"), Agg("max", x10, "mx10"), Agg("max", x10 + 1, "mx10p"),
            Agg("sum", x100, "s100"), Agg("sum", x10 + 1, "s10p")]


MS_KEYS = {1: "x1k", 2: "x10k", 3: "x100k", 4: "x1m", "2k": "x2k", "4k": "x4k", "6k": "x6k"}


def msbs(i, table="syn", key_type=FP32):
    """MultiStep/MSBS00<i>.sql: GROUP BY cast(x AS float) -- GroupByBaselineHash"""
    return QueryUnit(table, groupby=[Cast(ColRef(MS_KEYS[i]), key_type)], targets=[KeyRef(0, "k")] + _ms_targets())


def msphs(i, table="syn"):
    """MultiStep/MSPHS00<i>.sql: GROUP BY x -- GroupByPerfectHash"""
    return QueryUnit(table, groupby=[ColRef(MS_KEYS[i])], targets=[KeyRef(0, "k")] + _ms_targets())


PHM_KEYS = {1: "x10", 2: "x100", 3: "x1k", 4: "x10k", 5: "x100k", 6: "x1m"}


def phm(i, table="syn"):
    """PerfectHashMultiCol/PHM00<i>.sql: GROUP BY x, y10 with the five aggregates of z10"""
    z = ColRef("z10")
    return QueryUnit(table, groupby=[ColRef(PHM_KEYS[i]), ColRef("y10")],
                     targets=[KeyRef(0, "k0"), KeyRef(1, "k1"), Agg("count", z, "c"), Agg("sum", z, "s"), Agg("max", z, "mx"),
                              Agg("min", z, "mn"), Agg("avg", z, "a")])


def filtered(q, column, op, literal):
    """`q` WHERE column <op> literal"""
    import dataclasses
    from hdk_amd.ir import Cmp, Lit
    return dataclasses.replace(q, quals=list(q.quals) + [Cmp(ColRef(column), op, Lit(literal))])


def filtered_or(q, first, second):
    """`q` WHERE first OR NOT (second negated): two (column, op, literal) leaves under an OR / NOT program"""
    import dataclasses
    from hdk_amd.ir import Cmp, Lit, Not, Or
    neg = {"<": ">=", "<=": ">", ">": "<=", ">=": "<", "=": "<>", "<>": "="}
    c1 = Cmp(ColRef(first[0]), first[1], Lit(first[2]))
    c2 = Not(Cmp(ColRef(second[0]), neg[second[1]], Lit(second[2])))
    return dataclasses.replace(q, quals=list(q.quals) + [Or(c1, c2)])
