"""Differential fuzz, CPU half: the oracle against SQLite on seeded random group-by / filter queries
without joins (SQL semantics of NULLs, integer division and averages need care to state portably, so the
comparison sticks to COUNT / SUM / MIN / MAX over plain columns).  The GPU half is tests/test_gpu_fuzz.py."""
import sqlite3

import numpy as np

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Agg, Cmp, ColRef, KeyRef, Lit, QueryMustRunOnCpu, QueryUnit
from hdk_amd.plan import compile_query

from fuzz_queries import make_tables
from util import run_oracle


def test_oracle_vs_sqlite_random_groupbys(oracle):
    rng = np.random.default_rng(20261002)
    st = make_tables(rng, 6000, 200)
    fact = st.get("fact")
    nulls = {"k8": A.NULL_TINYINT, "k32": A.NULL_INT, "v64": A.NULL_BIGINT, "v32": A.NULL_INT}
    cols = {}
    for c in ("k8", "k16", "k32", "v64", "v32", "v16"):
        a = np.concatenate(fact.columns[c].fragments)
        cols[c] = [None if (c in nulls and v == nulls[c]) else int(v) for v in a.tolist()]
    con = sqlite3.connect(":memory:")
    con.execute(f"create table fact ({', '.join(cols)})")
    con.executemany(f"insert into fact values ({', '.join('?' * len(cols))})", list(zip(*cols.values())))
    done = 0
    for _ in range(40):
        keys = [str(k) for k in rng.choice(["k8", "k16", "k32"], size=int(rng.integers(0, 3)), replace=False)]
        fcol = str(rng.choice(["v64", "v32", "v16", "k16"]))
        fop = str(rng.choice(["<", ">", "<=", ">=", "<>"]))
        flit = int(rng.integers(-500, 500))
        aggs = [(str(rng.choice(["count", "sum", "min", "max"])), str(rng.choice(["v64", "v32", "v16"]))) for _ in range(3)]
        q = QueryUnit("fact", quals=[Cmp(ColRef(fcol), fop, Lit(flit))], groupby=[ColRef(k) for k in keys],
                      targets=[KeyRef(i, k) for i, k in enumerate(keys)] + [Agg("count", None, "n")] +
                      [Agg(a, ColRef(c), f"a{i}") for i, (a, c) in enumerate(aggs)], bigint_count=True)
        try:
            cp, buf, err = run_oracle(oracle, st, q)
        except QueryMustRunOnCpu:
            continue
        assert err == 0
        sel = ", ".join(keys + ["count(*)"] + [f"{a}({c})" for a, c in aggs])
        sql = f"select {sel} from fact where {fcol} {fop} {flit}" + (f" group by {', '.join(keys)}" if keys else "")
        want = sorted(con.execute(sql).fetchall(), key=lambda r: tuple((x is None, x) for x in r))
        got_cols = rs.to_columns(cp, buf)
        got = sorted(zip(*got_cols.values()), key=lambda r: tuple((x is None, x) for x in r))
        if not keys and not want[0][0]:
            want = [tuple(0 if i == 0 else None for i in range(len(want[0])))]
        assert got == want, sql
        done += 1
    assert done >= 30
