"""hdk_amd/glue/HipPlanExtractor.h against hdk_amd/plan.py: the C++ pattern matcher RelAlgExecutionUnit ->
hdk_hip_plan (run over stand-in hdk::ir trees by tests/cpp/extract_dump.cpp, with the reference's own operator enums)
must produce the SAME expression half as the Python planner every GPU parity test goes through -- filters and the
filter program, join descriptors, group-by keys, targets (aggregate, argument chain, skip rule, fp-slot kind, key
index, skip value) -- byte for byte.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import (Agg, And, Cast, Cmp, ColRef, ExtractYear, FP32, INT32, JoinSpec, KeyRef, Lit, Not, Or, Proj,
                        QueryUnit, Type)
from hdk_amd.plan import compile_query
from hdk_amd.storage import ArrowStorage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "extract_dump")


def _tables():
    rng = np.random.default_rng(3)
    n = 4000
    st = ArrowStorage()
    I64N, I64 = Type("int", 8, True), Type("int", 8, False)
    st.import_numpy("t", {"key": rng.integers(0, 64, n, dtype=np.int64), "val": rng.integers(-9, 9, n, dtype=np.int64)},
                    types={"key": I64, "val": I64N})
    st.import_numpy("trips", {"passenger_count": rng.integers(0, 7, n).astype(np.int16),
                              "pickup_datetime": rng.integers(1230768000, 1451606400, n, dtype=np.int64),
                              "trip_distance": rng.integers(0, 5000, n, dtype=np.int64)},
                    types={"passenger_count": Type("int", 2, True), "pickup_datetime": Type("timestamp", 8, True, unit="s"),
                           "trip_distance": Type("decimal", 8, True, scale=2)})
    st.import_numpy("f", {"k": rng.integers(0, 5, n).astype(np.int32), "v": rng.integers(-50, 50, n, dtype=np.int64),
                          "w": rng.normal(size=n)},
                    types={"k": Type("int", 4, True), "v": I64N, "w": Type("fp", 8, True)})
    st.import_numpy("dim", {"dval": rng.integers(0, 1000, 1000, dtype=np.int64), "key": rng.permutation(1000).astype(np.int64)},
                    types={"dval": I64, "key": I64})
    st.import_numpy("fact", {"fk": rng.integers(0, 1200, n, dtype=np.int64), "val": rng.integers(-9, 9, n, dtype=np.int64)},
                    types={"fk": I64N, "val": I64N})
    st.import_numpy("g", {"k": rng.integers(0, 300, n).astype(np.int32), "f": rng.random(n).astype(np.float32),
                          "d": rng.normal(size=n)},
                    types={"k": Type("int", 4, False), "f": Type("fp", 4, True), "d": Type("fp", 8, True)})
    from hdk_amd.ir import DATE32
    days = (18000 + np.arange(100)).astype(np.int32)
    dday = np.concatenate([days, days[:50]])
    dday[7] = -(2**31)
    st.import_numpy("ddup", {"v": rng.integers(-9, 9, 150, dtype=np.int64), "day": dday}, types={"v": I64N, "day": DATE32})
    fday = rng.integers(17990, 18110, n).astype(np.int32)
    fday[::50] = -(2**31)
    st.import_numpy("fdate", {"day": fday}, types={"day": DATE32})
    dk = rng.permutation(1000).astype(np.int64)
    dk[int(np.nonzero((dk != 0) & (dk != 999))[0][0])] = A.NULL_BIGINT  # one NULL row, the range stays [0, 999]
    st.import_numpy("dimn", {"dval": rng.integers(0, 1000, 1000, dtype=np.int64), "key": dk}, types={"dval": I64, "key": I64N})
    st.import_numpy("h", {"k": rng.integers(0, 300, n).astype(np.int32), "s": rng.integers(-99, 99, n).astype(np.int16),
                          "i": rng.integers(-999, 999, n).astype(np.int32)},
                    types={"k": Type("int", 4, False), "s": Type("int", 2, True), "i": Type("int", 4, True)})
    return st


def _queries():
    pc, ts = ColRef("passenger_count"), ColRef("pickup_datetime")
    return {
        "c2": QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")],
                        bigint_count=True),
        "q4": QueryUnit("trips", groupby=[pc, ExtractYear(ts), Cast(ColRef("trip_distance"), INT32)],
                        targets=[KeyRef(0, "pc"), KeyRef(1, "year"), KeyRef(2, "dist"), Agg("count", None, "cnt")]),
        "filters": QueryUnit("f", quals=[And(Or(Cmp(ColRef("v"), "<", Lit(5)), Not(Cmp(ColRef("w"), ">=", Lit(2.5)))),
                                             Cmp(ColRef("k"), "=", Lit(2)))],
                             targets=[Agg("count", None, "c"), Agg("min", ColRef("v") + 3, "m"), Agg("avg", ColRef("w"), "a")]),
        "join": QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                          targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim"), "s"), Agg("count", None, "c")], bigint_count=True),
        "c5": QueryUnit("t", groupby=[ColRef("key")], force_baseline=True, baseline_entry_count=4096,
                        targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s")]),
        "projection": QueryUnit("t", quals=[Cmp(ColRef("val"), "<", Lit(100))], targets=[Proj(ColRef("key"), "key"), Proj(ColRef("val") * 2, "v2")]),
        "floats": QueryUnit("g", groupby=[ColRef("k")], targets=[KeyRef(0, "k"), Agg("sum", ColRef("f"), "sf"), Agg("avg", ColRef("f"), "af"),
                                                                   Agg("min", ColRef("d"), "md"), Agg("count", ColRef("f"), "cf")]),
        "date_bw_eq": QueryUnit("fdate", joins=[JoinSpec("ddup", ColRef("day"), "day", null_safe=True)],
                                targets=[Agg("count", None, "c"), Agg("sum", ColRef("v", "ddup"), "s")], bigint_count=True),
        "bw_eq_left": QueryUnit("fact", joins=[JoinSpec("dimn", ColRef("fk"), "key", "left", null_safe=True)],
                                targets=[Agg("count", None, "c"), Agg("sum", ColRef("dval", "dimn"), "s")], bigint_count=True),
        "semi": QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key", "semi")],
                          targets=[Agg("count", None, "c"), Agg("sum", ColRef("val"), "s")], bigint_count=True),
        "minmax_narrow": QueryUnit("h", groupby=[ColRef("k")],
                                   targets=[KeyRef(0, "k"), Agg("min", ColRef("s"), "mn"), Agg("max", ColRef("i"), "mx"),
                                            Agg("min", ColRef("i") + 1, "mn2"), Agg("sum", ColRef("s"), "sm")], bigint_count=True),
        "single": QueryUnit("g", groupby=[ColRef("k")], targets=[KeyRef(0, "k"), Agg("single_value", ColRef("f"), "s1"),
                                                                   Agg("single_value", ColRef("d"), "s2"), Agg("count", None, "c")]),
    }


def _b(x):
    return bytes(x)


def _expr_half(p):
    """The fields the extractor is responsible for (layout fields are make_plan's, filled from the descriptor)."""
    out = {"num_quals": p.num_quals, "num_joins": p.num_joins, "key_count": p.key_count, "num_targets": p.num_targets,
           "quals": [_b(p.quals[i]) for i in range(p.num_quals)],
           "filter": (p.num_filter_ops, bytes(p.filter_ops[:p.num_filter_ops]), p.filter_after_joins if p.num_filter_ops else 0),
           "keys": [_b(p.keys[k]) for k in range(p.key_count)]}
    # everything the probe is handed (getHashJoinArgs, PerfectJoinHashTable.cpp:786-824): key, range, NULL, the translated
    # NULL of a _bitwise probe, bucket_normalization; plus the loop kind and the table's slot count
    out["joins"] = [(_b(j.outer_key), j.kind, j.type, j.null_mode, j.null_val if j.null_mode else 0, j.min_key, j.max_key,
                     j.bucket, j.translated_null if j.null_mode == A.JOIN_NULL_BITWISE else 0, j.table_idx,
                     j.entry_count if j.kind != A.JOIN_ONE_TO_ONE or j.null_mode == A.JOIN_NULL_BITWISE else 0) for j in
                    [p.joins[i] for i in range(p.num_joins)]]
    tg = []
    for t in range(p.num_targets):
        x = p.targets[t]
        tg.append((x.agg, x.has_arg, _b(x.arg) if x.has_arg else b"", x.skip_null, x.arg_is_fp, x.key_idx if x.agg == A.AGG_ID else -1,
                   x.null_val if x.has_arg and x.agg != A.AGG_ID else 0))
    out["targets"] = tg
    return out


@pytest.fixture(scope="module")
def dumped(tmp_path_factory):
    if os.path.isdir("/root/reference/omniscidb"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "--no-print-directory"], stdout=subprocess.DEVNULL)
    if not os.path.exists(EXE):
        pytest.skip("tests/cpp/_build/extract_dump is missing (built where the reference tree exists)")
    d = tmp_path_factory.mktemp("plans")
    r = subprocess.run([EXE, str(d)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "refused 5 of 5" in r.stdout  # shapes outside the library throw QueryMustRunOnCpu
    return d


@pytest.mark.parametrize("name", ["c2", "q4", "filters", "join", "c5", "projection", "floats", "single", "date_bw_eq", "bw_eq_left",
                                  "semi", "minmax_narrow"])
def test_extractor_matches_the_python_planner(dumped, name):
    st = _tables()
    want = compile_query(st, _queries()[name]).plan
    raw = open(os.path.join(str(dumped), name + ".plan"), "rb").read()
    assert len(raw) == C.sizeof(A.Plan)
    got = A.Plan.from_buffer_copy(raw)
    from hdk_amd._lib import lib
    assert lib().hdk_hip_validate_plan(C.byref(got), 0) == A.OK, lib().hdk_hip_last_error()
    g, w = _expr_half(got), _expr_half(want)
    for k in w:
        assert g[k] == w[k], (name, k, g[k], w[k])
    assert got.query_kind == want.query_kind and got.key_width == want.key_width
