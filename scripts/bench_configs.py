#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): the other BASELINE.json configs on one GPU,
at sizes that fit comfortably, reported as rows/s and algorithmic GB/s (SURVEY.md 8d byte counts).

    python scripts/bench_configs.py [--rows N] [--only c1,c2n,c3,c4,c5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(step, reps=5, warm=2):
    from hdk_amd._lib import check, lib
    L = lib()
    for _ in range(warm):
        step.enqueue()
    step.mgr.synchronizeStream(step.dev)
    n = C.c_int32(0)
    check(L.hdk_hip_collect_scan_times(step.dev, None, 0, C.byref(n)))
    t0 = time.perf_counter()
    for _ in range(reps):
        step.enqueue()
    step.mgr.synchronizeStream(step.dev)
    wall = (time.perf_counter() - t0) / reps
    buf = (C.c_float * reps)()
    check(L.hdk_hip_collect_scan_times(step.dev, buf, reps, C.byref(n)))
    k = [buf[i] for i in range(min(n.value, reps))]
    return wall, (float(np.mean(k)) * 1e-3 if k else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256_000_000)
    ap.add_argument("--only", default="c1,c2,c2n,c2f,c2x,c2cc,c3,c3g,c3f,q1,q2,q3,q4,c5,p1,p50,pj")
    ap.add_argument("--launch-env", default="", help="KEY=VAL,... set while the steps are prepared and run (A/B switches of the library)")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--dim-rows", type=int, default=10_000_000)
    ap.add_argument("--no-fuse", action="store_true")
    ap.add_argument("--hot-frac", type=float, default=0.0, help="this share of the rows gets ONE key in x10k / x100k (msbs2-3, msphs2-3, phm3-5)")
    ap.add_argument("--null-frac", type=float, default=0.0, help="NULLs in every column of the synthetic suite's table (nga*, msbs*, msphs*, phm*)")
    ap.add_argument("--flags", type=int, default=0, help="extra HDK_HIP_LAUNCH_* flags")
    args = ap.parse_args()
    only = set(args.only.split(","))

    from hdk_amd import _abi as A
    from hdk_amd.executor import Executor
    from hdk_amd.ir import Agg, Cast, Cmp, ColRef, ExtractYear, FP64, INT32, JoinSpec, KeyRef, Lit, Or, Proj, QueryUnit, Type
    from hdk_amd.storage import ArrowStorage

    n = args.rows
    rng = np.random.default_rng(20261002)
    st = ArrowStorage()
    frag = 32_000_000
    print(f"# generating {n} rows ...", file=sys.stderr)
    n_t0 = n if (only & {"c1", "c2", "c2n", "c2f", "c2x", "c2cc", "c2or", "c2m", "c3", "c3g", "c3gm", "c3f", "c3x", "c3x8", "c3x2", "ph64", "bh64", "bh64f", "c3d", "c3k", "p1", "p50", "pj", "c5"}) else 1000
    key = rng.integers(0, 64, n_t0, dtype=np.int64)
    val = rng.integers(-2**31, 2**31, n_t0, dtype=np.int64)
    valn = val.copy()
    valn[rng.random(n_t0) < 0.01] = A.NULL_BIGINT
    nd = args.dim_rows
    need_t = bool(only & {"c1", "c2", "c2n", "c2f", "c2x", "c2cc", "c2or", "c2m", "c3", "c3g", "c3gm", "c3f", "c3x", "c3x8", "c3x2", "ph64", "bh64", "bh64f", "c3d", "c3k", "p1", "p50", "pj", "c5"})
    need_trips = bool(only & {"q1", "q2", "q3", "q4", "q3v", "q3m", "q4v"})
    if not need_t:
        n_t = 1000
        key, val, valn = key[:n_t], val[:n_t], valn[:n_t]
    fk = rng.integers(0, nd, len(key), dtype=np.int64)
    tcols = {"key": key, "val": val, "valn": valn, "fk": fk, "hk": rng.integers(0, max(n // 10, 1000), len(key), dtype=np.int64)}
    if only & {"c3d", "c3k"}:  # plain key columns for the parity-path joins (no arithmetic in front of the probe)
        tcols.update({"fk10": fk // 10, "k1": key * 15, "k2": fk // 10000})
    if only & {"c3x"}:
        tcols["g32"] = key.astype(np.int32)  # a fact-side group key as an INT column
    if only & {"ph64", "bh64", "bh64f"}:
        tcols["y64"] = rng.integers(1, 11, len(key), dtype=np.int64)  # the BaselineHash benchmark's measure as a BIGINT column
    st.import_numpy("t", tcols, fragment_size=frag)
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "dval": rng.integers(0, 10**6, nd).astype(np.int64),
                            "attr": rng.integers(0, 64, nd).astype(np.int64)}, fragment_size=frag)
    if only & {"c3d", "c3k"}:
        # parity-path joins: a dimension with two rows per key (one-to-many table) and one with a two-column key (keyed table)
        nd2 = max(nd // 10, 1000)
        st.import_numpy("dim2", {"key": np.concatenate([rng.permutation(nd2), rng.permutation(nd2)]).astype(np.int64),
                                 "dval": rng.integers(0, 10**6, 2 * nd2).astype(np.int64)}, fragment_size=frag)
        st.import_numpy("dimk", {"k1": (np.arange(nd2) % 1000).astype(np.int64), "k2": (np.arange(nd2) // 1000).astype(np.int64),
                                 "dval": rng.integers(0, 10**6, nd2).astype(np.int64)}, fragment_size=frag)
    # taxi-shaped table (taxi_reduced_bench.cpp:13-24 column types)
    nt = n if need_trips else 1000
    st.import_numpy("trips", {"cab_type": rng.integers(0, 2, nt).astype(np.int32),
                              "passenger_count": rng.integers(0, 7, nt).astype(np.int16),
                              "pickup_datetime": rng.integers(1230768000, 1451606400, nt, dtype=np.int64),
                              "trip_distance": rng.integers(0, 5000, nt, dtype=np.int64),
                              "total_amount": rng.integers(0, 20000, nt, dtype=np.int64)}, fragment_size=frag,
                    types={"cab_type": Type("dict", 4), "pickup_datetime": Type("timestamp", 8, unit="s"),
                           "trip_distance": Type("decimal", 8, scale=2), "total_amount": Type("decimal", 8, scale=2)})
    # the reference's synthetic benchmark table (Benchmarks/synthetic_benchmark/create_table.py:118-130): INT columns, uniform
    need_syn = bool(only & {"bh1", "bh1f", "bhm", "bh2k", "bh4k", "bh2", "bh3", "bh4", "bh5", "ph1", "ph2", "ph3"})
    ns = n if need_syn else 1000
    syn = {f"x{nm}": rng.integers(1, hi + 1, ns).astype(np.int32) for nm, hi in (("10", 10), ("100", 100), ("1k", 1000), ("10k", 10_000), ("100k", 100_000))}
    if only & {"bh2k", "bh4k"}:  # between the 256-thread dense tables (2 K entries) and the two-pass forms
        syn["x2k"] = rng.integers(1, 2001, ns).astype(np.int32)
        syn["x4k"] = rng.integers(1, 4001, ns).astype(np.int32)
    syn["y10"] = rng.integers(1, 11, ns).astype(np.int32)
    st.import_numpy("syn", syn, fragment_size=frag)
    # the same table for the suite's other families (tests/syn_queries.py: NonGroupedAgg, MultiStep, PerfectHashMultiCol), with
    # only the columns the asked-for configs read
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import syn_queries as SQ
    SYN2 = {f"nga{i}": (SQ.nga(i, "syn2"), 24) for i in range(1, 6)}
    SYN2.update({f"msbs{i}": (SQ.msbs(i, "syn2"), 12) for i in range(1, 5)})  # (cast(x AS float) keys, as written)
    SYN2.update({f"msphs{i}": (SQ.msphs(i, "syn2"), 12) for i in range(1, 5)})
    SYN2.update({f"msphs{i}": (SQ.msphs(i, "syn2"), 12) for i in ("2k", "4k", "6k")})
    SYN2.update({f"msbs{i}": (SQ.msbs(i, "syn2"), 12) for i in ("2k", "4k", "6k")})
    SYN2.update({f"phm{i}": (SQ.phm(i, "syn2"), 12) for i in range(1, 7)})
    # three measures by one key / by two keys (not in the reference's suite: the shape next to MSBS006-007)
    _s3 = [Agg("sum", ColRef(c), "s" + c) for c in ("x10", "y10", "z10")]
    SYN2["sum3"] = (QueryUnit("syn2", groupby=[ColRef("x1k")], targets=[KeyRef(0, "k"), Agg("count", None, "n")] + _s3), 16)
    SYN2["sum3k2"] = (QueryUnit("syn2", groupby=[ColRef("x100"), ColRef("y100")], targets=[KeyRef(0, "k0"), KeyRef(1, "k1")] + _s3), 20)
    # a mix no compile-time shape covers: the "aggregates of plain columns" policy (BhmPlain) against the run-time form
    SYN2["mix3"] = (QueryUnit("syn2", groupby=[ColRef("x1k")], targets=[KeyRef(0, "k"), Agg("count", None, "n"), Agg("sum", ColRef("x10"), "s"),
                                                                        Agg("max", ColRef("y10"), "mx"), Agg("min", ColRef("z10"), "mn"),
                                                                        Agg("avg", ColRef("z10"), "av")]), 16)
    SYN2["mix3f"] = (QueryUnit("syn2", quals=[Cmp(ColRef("x100"), "<", Lit(71))], groupby=SYN2["mix3"][0].groupby, targets=SYN2["mix3"][0].targets), 20)
    want2 = [k for k in SYN2 if k in only]
    cols2 = set()
    for k in want2:
        q2 = SYN2[k][0]
        for e in list(q2.groupby) + [t.arg for t in q2.targets if getattr(t, "arg", None) is not None] + [c.lhs for c in q2.quals if hasattr(c, "lhs")]:
            stack = [e]
            while stack:
                x = stack.pop()
                if isinstance(x, ColRef):
                    cols2.add(x.name)
                for attr in ("arg", "lhs", "rhs", "left", "right"):
                    if hasattr(x, attr) and getattr(x, attr) is not None and not isinstance(getattr(x, attr), (int, float, str)):
                        stack.append(getattr(x, attr))
    if want2:
        t2 = SQ.syn_table(rng, n, sorted(cols2), null_frac=args.null_frac)
        if args.hot_frac > 0:  # one hot key in the big key columns
            for c in ("x1k", "x10k", "x100k"):
                if c in t2:
                    t2[c][rng.random(n) < args.hot_frac] = 4242
        st.import_numpy("syn2", t2, fragment_size=frag)
    ex = Executor(st, 0)
    ex.fuse_join_tables = not args.no_fuse

    def bh(xcol):  # queries/BaselineHash/BH001-005.sql
        y = ColRef("y10")
        return QueryUnit("syn", groupby=[Cast(ColRef(xcol), FP64)],
                         targets=[KeyRef(0, "key0"), Agg("count", y), Agg("sum", y), Agg("max", y), Agg("min", y), Agg("avg", y)])
    Q = {
        "c1": (QueryUnit("t", targets=[Agg("sum", ColRef("val"))]), 8),
        "c2": (QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
        "c2n": (QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0), Agg("sum", ColRef("valn"))]), 16),
        # C2 with a filter that half of the rows pass (filter column = a third 8-byte column: 24 B/row)
        "c2f": (QueryUnit("t", quals=[Cmp(ColRef("valn"), "<", Lit(0))], groupby=[ColRef("key")],
                          targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 24),
        # the streaming kernel's wider menu (round 4): an expression argument over two columns; a column-column filter
        "c2x": (QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0), Agg("sum", ColRef("val") * ColRef("hk"))]), 24),
        "c2cc": (QueryUnit("t", quals=[Cmp(ColRef("val"), "<", ColRef("hk"))], groupby=[ColRef("key")],
                           targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 24),
        "c3": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                         targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim"))]), 16),
        "c3g": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") / 15625],
                          targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
        "c3d": (QueryUnit("t", joins=[JoinSpec("dim2", ColRef("fk10"), "key")],
                          targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim2")), Agg("count")]), 16),
        "c3k": (QueryUnit("t", joins=[JoinSpec("dimk", [ColRef("k1"), ColRef("k2")], ["k1", "k2"])],
                          targets=[Agg("sum", ColRef("val") + ColRef("dval", "dimk")), Agg("count")]), 24),
        # a perfect-hash GROUP BY whose table does not fit LDS (rows / 10 groups): the global-atomics strategy
        "c2m": (QueryUnit("t", groupby=[ColRef("hk")], targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
        # an OR in the filter: the postfix filter program, in the batched interpreter since round 4
        "c2or": (QueryUnit("t", quals=[Or(Cmp(ColRef("val"), "<", Lit(0)), Cmp(ColRef("key"), "=", Lit(3)))], groupby=[ColRef("key")],
                           targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
        # star schema: filter on one dimension column, group by another (two payload words in the sliced join)
        "c3f": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("dval", "dim"), "<", Lit(500_000))],
                          groupby=[ColRef("attr", "dim")], targets=[KeyRef(0), Agg("sum", ColRef("val")), Agg("count")]), 16),
        # group by a column of the FACT table under the join, filter on the dimension (round 5: the second outer column rides in
        # the 8-byte tuple's spare bits); two fact measures
        "c3x": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("dval", "dim"), "<", Lit(500_000))],
                          groupby=[ColRef("g32")], targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 20),
        "c3x8": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("dval", "dim"), "<", Lit(500_000))],
                           groupby=[ColRef("key")], targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 24),
        "c3x2": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("dval", "dim"), "<", Lit(500_000))],
                           targets=[Agg("sum", ColRef("val")), Agg("sum", ColRef("key")), Agg("count")]), 24),
        # the benchmark's five aggregates over BIGINT columns (the general step form of the on-chip kernels): perfect hash on the
        # key, open addressing behind a cast, and the latter with a filter
        "bh1f": (QueryUnit("syn", groupby=[Cast(ColRef("x10"), FP64)], quals=[Cmp(ColRef("y10"), "<=", Lit(7))],
                           targets=[KeyRef(0, "key0")] + [Agg(k, ColRef("y10")) for k in ("count", "sum", "max", "min", "avg")]), 8),
        # a key without an expression range (x % m: GroupByBaselineHash) in front of the benchmark's aggregates
        "bh2k": (bh("x2k"), 8),
        "bh4k": (bh("x4k"), 8),
        "bhm": (QueryUnit("syn", groupby=[ColRef("x1k") % 37],
                          targets=[KeyRef(0, "key0")] + [Agg(k, ColRef("y10")) for k in ("count", "sum", "max", "min", "avg")]), 8),
        "ph64": (QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0)] + [Agg(k, ColRef("y64")) for k in ("count", "sum", "max", "min", "avg")]), 16),
        "bh64": (QueryUnit("t", groupby=[Cast(ColRef("key"), FP64)], targets=[KeyRef(0)] + [Agg(k, ColRef("y64")) for k in ("count", "sum", "max", "min", "avg")]), 16),
        "bh64f": (QueryUnit("t", groupby=[Cast(ColRef("key"), FP64)], quals=[Cmp(ColRef("y64"), "<=", Lit(7))],
                            targets=[KeyRef(0)] + [Agg(k, ColRef("y64")) for k in ("count", "sum", "max", "min", "avg")]), 16),
        "q1": (QueryUnit("trips", groupby=[ColRef("cab_type")], targets=[KeyRef(0), Agg("count")]), 4),
        "q2": (QueryUnit("trips", groupby=[ColRef("passenger_count")],
                         targets=[KeyRef(0), Agg("avg", ColRef("total_amount"))]), 10),
        "q3": (QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime"))],
                         targets=[KeyRef(0), KeyRef(1), Agg("count")]), 10),
        "q4": (QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime")),
                                           Cast(ColRef("trip_distance"), INT32)],
                         targets=[KeyRef(0), KeyRef(1), KeyRef(2), Agg("count")]), 18),
        # Q3 / Q4 with measures: the value form of the keys kernel
        "q3v": (QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime"))],
                          targets=[KeyRef(0), KeyRef(1), Agg("count"), Agg("avg", ColRef("total_amount"))]), 18),
        "q3m": (QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime"))],
                          targets=[KeyRef(0), KeyRef(1), Agg("sum", ColRef("total_amount")), Agg("min", ColRef("trip_distance")),
                                   Agg("max", ColRef("trip_distance"))]), 26),
        "q4v": (QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime")),
                                            Cast(ColRef("trip_distance"), INT32)],
                          targets=[KeyRef(0), KeyRef(1), KeyRef(2), Agg("count"), Agg("sum", ColRef("total_amount"))]), 26),
        # filter/project: SELECT key, val WHERE key < X (1 % / 50 % of the rows pass); bytes = 16 in + 24 out per passing row
        "p1": (QueryUnit("t", quals=[Cmp(ColRef("val"), "<", Lit(-2**31 + 2**32 // 100))], output_columnar=True,
                         targets=[Proj(ColRef("key"), "key"), Proj(ColRef("val"), "val")]), 16 + 0.24),
        "p50": (QueryUnit("t", quals=[Cmp(ColRef("val"), "<", Lit(0))], output_columnar=True,
                          targets=[Proj(ColRef("key"), "key"), Proj(ColRef("val"), "val")]), 16 + 12),
        "pj": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("val"), "<", Lit(-2**31 + 2**32 // 20))],
                         output_columnar=True, targets=[Proj(ColRef("val"), "val"), Proj(ColRef("dval", "dim"), "dval")]), 16 + 1.2),
        # the reference's BaselineHash benchmark queries: 10 ... 100 K groups behind a double key; 8 bytes per row
        # the same aggregates by the integer column itself: the perfect-hash twin of BH001-003 (what the dense route would run)
        "ph1": (QueryUnit("syn", groupby=[ColRef("x10")], targets=[KeyRef(0, "k"), Agg("count", ColRef("y10")), Agg("sum", ColRef("y10")), Agg("max", ColRef("y10")), Agg("min", ColRef("y10")), Agg("avg", ColRef("y10"))]), 8),
        "ph2": (QueryUnit("syn", groupby=[ColRef("x100")], targets=[KeyRef(0, "k"), Agg("count", ColRef("y10")), Agg("sum", ColRef("y10")), Agg("max", ColRef("y10")), Agg("min", ColRef("y10")), Agg("avg", ColRef("y10"))]), 8),
        "ph3": (QueryUnit("syn", groupby=[ColRef("x1k")], targets=[KeyRef(0, "k"), Agg("count", ColRef("y10")), Agg("sum", ColRef("y10")), Agg("max", ColRef("y10")), Agg("min", ColRef("y10")), Agg("avg", ColRef("y10"))]), 8),
        "bh1": (bh("x10"), 8), "bh2": (bh("x100"), 8), "bh3": (bh("x1k"), 8), "bh4": (bh("x10k"), 8), "bh5": (bh("x100k"), 8),
        # SURVEY 8(d)'s C3 variant as written: no expression range for a modulo, so an open-addressing table of 128 entries
        "c3gm": (QueryUnit("t", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") % 64],
                           targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
        "c5": (QueryUnit("t", groupby=[ColRef("hk")], force_baseline=True,
                         targets=[KeyRef(0), Agg("sum", ColRef("val"))]), 16),
    }
    Q.update({k: SYN2[k] for k in want2})
    for kv in [x for x in args.launch_env.split(",") if x]:
        os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
    for name, (q, bpr) in Q.items():
        if name not in only:
            continue
        try:
            cp = ex.compile(q)
            step = ex.prepare(cp, grid=args.grid, flags=A.LAUNCH_RECORD_EVENTS | args.flags)
            wall, kern = timed(step)
            res = step.fetch()
            t = kern or wall
            n = st.get(q.table).num_rows
            print(json.dumps({"config": name, "kernel": step.kernel_names(), "rows": n, "entries": cp.entry_count,
                              "rows_per_s": n / wall, "kernel_ms": None if kern is None else kern * 1e3,
                              "step_ms": wall * 1e3, "alg_bytes_per_row": bpr, "alg_GBps": n * bpr / t / 1e9,
                              "frac_of_8TBps": n * bpr / t / 8e12, "groups": res.row_count(),
                              "join_kinds": [int(step.plan.joins[i].kind) for i in range(step.plan.num_joins)]}))
            step.free()
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"config": name, "error": repr(e)}))


if __name__ == "__main__":
    main()
