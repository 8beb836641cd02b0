"""Join variants the reference's planner emits beyond the plain / nullable inner probe:
  * DATE keys -> bucket_normalization = 86400: bucketized_hash_join_idx[_nullable|_bitwise] over tables built by
    fill_hash_join_buff_bucketized / fill_one_to_many_hash_table_bucketized
    (QE/JoinHashTable/PerfectJoinHashTable.cpp:45-85,798-816,1018-1031; HashJoinRuntime.cpp:197-293);
  * IS NOT DISTINCT FROM (kBwEq) -> uses_bw_eq builds + hash_join_idx_bitwise probes;
  * SEMI / ANTI joins -> first-row-wins fills (fill_hashtable_for_semi_join, JoinHashImpl.h:68-77) and the
    JoinLoop::codegen rules (QE/LoopControlFlow/JoinLoop.cpp:254-262).
Each case is (name, QueryUnit, sqlite sql or None); the CPU suite checks the oracle against SQLite where SQL can say
the same thing, the GPU suite checks every kernel family against the oracle."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.ir import DATE16, DATE32, DATE64, Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, Proj, QueryUnit
from hdk_amd.storage import ArrowStorage

D = 86400


def _py(arr, null):
    return [None if v == null else int(v) for v in arr.tolist()]


def make_variants(seed=5, nf=6000, nd=400):
    rng = np.random.default_rng(seed)
    NI32 = -(2**31)
    # ---- dims -------------------------------------------------------------------------------------------------------
    # ddate: one row per day (unique -> bucketized one-to-one), days since the epoch as int32 (Arrow date32), some NULL
    day0 = 18000
    dd_day = (day0 + rng.permutation(nd)).astype(np.int32)
    dd_day[rng.random(nd) < 0.03] = NI32
    dd_v = rng.integers(-300, 300, nd).astype(np.int64)
    dd_g = rng.integers(0, 5, nd).astype(np.int32)
    # ddup: several rows per day (-> bucketized one-to-many); and a DATE64 column in SECONDS, several values per day
    du_day = (day0 + rng.integers(0, nd // 3, nd)).astype(np.int32)
    du_day16 = (du_day - day0 + 100).astype(np.int16)  # a 2-byte day count
    du_sec = du_day.astype(np.int64) * D + rng.integers(0, D, nd)
    # the probe's range test is on SECONDS (key >= min_key && key <= max_key): let the column's range cover whole days,
    # so that "same day" and "inside the range" agree and SQLite (which sees day numbers) can be the second opinion
    i_lo, i_hi = int(np.argmin(du_day)), int(np.argmax(du_day))
    du_sec[i_lo] = int(du_day[i_lo]) * D
    du_sec[i_hi] = int(du_day[i_hi]) * D + D - 1
    nul = rng.random(nd) < 0.03
    nul[[i_lo, i_hi]] = False
    du_sec[nul] = A.NULL_BIGINT
    du_day = du_day.copy()
    du_day[rng.random(nd) < 0.02] = NI32
    du_v = rng.integers(-300, 300, nd).astype(np.int64)
    # dk: integer keys, unique except for the NULL rows (one NULL row -> one-to-one even under kBwEq; two -> one-to-many)
    dk1 = (100 + rng.permutation(nd)).astype(np.int64)
    dk1[7] = A.NULL_BIGINT
    dk2 = dk1.copy()
    dk2[11] = A.NULL_BIGINT
    dk2[23] = A.NULL_BIGINT
    dkd = (100 + rng.integers(0, nd // 4, nd)).astype(np.int64)  # duplicates: SEMI keeps the first row of a key
    dkd[rng.random(nd) < 0.02] = A.NULL_BIGINT
    dk_v = rng.integers(-300, 300, nd).astype(np.int64)
    dk_a = rng.integers(0, 30, nd).astype(np.int32)
    dk_b = rng.integers(0, 6, nd).astype(np.int32)
    st = ArrowStorage()
    st.import_numpy("ddate", {"day": dd_day, "v": dd_v, "g": dd_g}, fragment_size=97, types={"day": DATE32})
    st.import_numpy("ddup", {"day": du_day, "sec": du_sec, "day16": du_day16, "v": du_v}, fragment_size=131,
                    types={"day": DATE32, "sec": DATE64, "day16": DATE16})
    st.import_numpy("dk", {"k1": dk1, "k2": dk2, "kd": dkd, "v": dk_v, "a": dk_a, "b": dk_b}, fragment_size=89)
    # ---- fact -------------------------------------------------------------------------------------------------------
    f_day = (day0 - 5 + rng.integers(0, nd + 10, nf)).astype(np.int32)
    f_day[rng.random(nf) < 0.04] = NI32
    f_sec = f_day.astype(np.int64) * D + rng.integers(0, D, nf)
    f_sec[f_day == NI32] = A.NULL_BIGINT
    f_day16 = np.where(f_day == NI32, -(2**15), (f_day.astype(np.int64) - day0 + 100)).astype(np.int16)
    f_k = (95 + rng.integers(0, nd + 10, nf)).astype(np.int64)
    f_k[rng.random(nf) < 0.05] = A.NULL_BIGINT
    f_a = rng.integers(0, 33, nf).astype(np.int32)
    f_b = rng.integers(0, 7, nf).astype(np.int32)
    f_val = rng.integers(-100, 100, nf).astype(np.int64)
    st.import_numpy("fact", {"day": f_day, "sec": f_sec, "day16": f_day16, "k": f_k, "a": f_a, "b": f_b, "val": f_val},
                    fragment_size=1700, types={"day": DATE32, "sec": DATE64, "day16": DATE16})
    sql_tables = {
        "ddate": {"day": _py(dd_day, NI32), "v": dd_v.tolist(), "g": dd_g.tolist()},
        # SQLite sees a DATE64 as its day number: the bucketized table cannot tell two seconds of one day apart either
        "ddup": {"day": _py(du_day, NI32), "secday": [None if v == A.NULL_BIGINT else int(v) // D for v in du_sec.tolist()],
                 "day16": du_day16.tolist(), "v": du_v.tolist()},
        "dk": {"k1": _py(dk1, A.NULL_BIGINT), "k2": _py(dk2, A.NULL_BIGINT), "kd": _py(dkd, A.NULL_BIGINT),
               "v": dk_v.tolist(), "a": dk_a.tolist(), "b": dk_b.tolist()},
        "fact": {"day": _py(f_day, NI32), "secday": [None if v == A.NULL_BIGINT else int(v) // D for v in f_sec.tolist()],
                 "day16": _py(f_day16, -(2**15)), "k": _py(f_k, A.NULL_BIGINT), "a": f_a.tolist(), "b": f_b.tolist(),
                 "val": f_val.tolist()},
    }
    F = ColRef
    DD = lambda n: ColRef(n, "ddate")  # noqa: E731
    DU = lambda n: ColRef(n, "ddup")  # noqa: E731
    DK = lambda n: ColRef(n, "dk")  # noqa: E731
    cases = [
        # ---- bucketized (DATE) -------------------------------------------------------------------------------------
        ("date_one_to_one", QueryUnit("fact", joins=[JoinSpec("ddate", F("day"), "day")],
                                      targets=[Agg("count", None, "c"), Agg("sum", F("val") + DD("v"), "s")]),
         "select count(*), sum(val + v) from fact join ddate on fact.day = ddate.day"),
        ("date_one_to_one_group", QueryUnit("fact", joins=[JoinSpec("ddate", F("day"), "day")], groupby=[DD("g")],
                                            quals=[Cmp(DD("v"), ">", Lit(-100))],
                                            targets=[KeyRef(0, "g"), Agg("count", None, "c"), Agg("max", DD("v"), "mx")]),
         "select g, count(*), max(v) from fact join ddate on fact.day = ddate.day where v > -100 group by g"),
        ("date_left", QueryUnit("fact", joins=[JoinSpec("ddate", F("day"), "day", "left")],
                                targets=[Agg("count", None, "c"), Agg("count", DD("v"), "cv"), Agg("sum", DD("v"), "sv")]),
         "select count(*), count(v), sum(v) from fact left join ddate on fact.day = ddate.day"),
        ("date_one_to_many", QueryUnit("fact", joins=[JoinSpec("ddup", F("day"), "day")],
                                       targets=[Agg("count", None, "c"), Agg("sum", DU("v"), "s")]),
         "select count(*), sum(v) from fact join ddup on fact.day = ddup.day"),
        ("date16_one_to_many", QueryUnit("fact", joins=[JoinSpec("ddup", F("day16"), "day16")],
                                         targets=[Agg("count", None, "c"), Agg("sum", DU("v") - F("val"), "s")]),
         "select count(*), sum(v - val) from fact join ddup on fact.day16 = ddup.day16"),
        # a DATE held in seconds: both sides land in the slot of their DAY
        ("date64_bucket_collapses_a_day", QueryUnit("fact", joins=[JoinSpec("ddup", F("sec"), "sec")],
                                                    targets=[Agg("count", None, "c"), Agg("sum", DU("v"), "s")]),
         "select count(*), sum(v) from fact join ddup on fact.secday = ddup.secday"),
        # ---- IS NOT DISTINCT FROM --------------------------------------------------------------------------------------
        ("bw_eq_one_to_one", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "k1", null_safe=True)],
                                       targets=[Agg("count", None, "c"), Agg("sum", DK("v"), "s")]),
         "select count(*), sum(v) from fact join dk on fact.k is dk.k1"),
        ("bw_eq_one_to_many", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "k2", null_safe=True)],
                                        targets=[Agg("count", None, "c"), Agg("sum", DK("v") * F("val"), "s")]),
         "select count(*), sum(v * val) from fact join dk on fact.k is dk.k2"),
        ("bw_eq_left_group", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "k1", "left", null_safe=True)], groupby=[DK("b")],
                                       targets=[KeyRef(0, "b"), Agg("count", None, "c"), Agg("min", DK("v"), "mn")]),
         "select dk.b, count(*), min(v) from fact left join dk on fact.k is dk.k1 group by dk.b"),
        # ---- SEMI / ANTI -----------------------------------------------------------------------------------------------
        ("semi_dups", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "kd", "semi")],
                                targets=[Agg("count", None, "c"), Agg("sum", F("val"), "s")]),
         "select count(*), sum(val) from fact where exists (select 1 from dk where dk.kd = fact.k)"),
        ("anti_dups", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "kd", "anti")], groupby=[F("b")],
                                targets=[KeyRef(0, "b"), Agg("count", None, "c"), Agg("sum", F("val"), "s")]),
         "select b, count(*), sum(val) from fact where not exists (select 1 from dk where dk.kd = fact.k) group by b"),
        ("semi_keyed", QueryUnit("fact", joins=[JoinSpec("dk", [F("a"), F("b")], ["a", "b"], "semi")],
                                 targets=[Agg("count", None, "c"), Agg("sum", F("val"), "s")]),
         "select count(*), sum(val) from fact where exists (select 1 from dk where dk.a = fact.a and dk.b = fact.b)"),
        ("anti_keyed", QueryUnit("fact", joins=[JoinSpec("dk", [F("a"), F("b")], ["a", "b"], "anti")],
                                 targets=[Agg("count", None, "c"), Agg("sum", F("val"), "s")]),
         "select count(*), sum(val) from fact where not exists (select 1 from dk where dk.a = fact.a and dk.b = fact.b)"),
        ("semi_date", QueryUnit("fact", joins=[JoinSpec("ddup", F("day"), "day", "semi")],
                                targets=[Agg("count", None, "c"), Agg("sum", F("val"), "s")]),
         "select count(*), sum(val) from fact where exists (select 1 from ddup where ddup.day = fact.day)"),
        # ---- DATE + IS NOT DISTINCT FROM: the reference's probe is handed max / bucket + 1 as the translated NULL while
        # the build files the NULL rows under max + 1 (module docstring of hdk_amd/plan.py: _compile_joins_and_quals), so
        # SQL has no say here -- the case pins device == oracle == the reference's two formulas
        ("date_bw_eq", QueryUnit("fact", joins=[JoinSpec("ddup", F("day"), "day", null_safe=True)],
                                 targets=[Agg("count", None, "c"), Agg("sum", DU("v"), "s")]),
         None),
    ]
    proj_cases = [
        ("proj_date", QueryUnit("fact", joins=[JoinSpec("ddate", F("day"), "day")], quals=[Cmp(F("val"), ">", Lit(90))],
                                targets=[Proj(F("val"), "val"), Proj(DD("v"), "v"), Proj(F("day"), "day")]),
         "select val, v, fact.day * 86400 from fact join ddate on fact.day = ddate.day where val > 90"),
        ("proj_bw_eq", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "k2", null_safe=True)], quals=[Cmp(F("val"), ">", Lit(80))],
                                 targets=[Proj(F("val"), "val"), Proj(DK("v"), "v")]),
         "select val, v from fact join dk on fact.k is dk.k2 where val > 80"),
        ("proj_anti", QueryUnit("fact", joins=[JoinSpec("dk", F("k"), "kd", "anti")], quals=[Cmp(F("val"), ">", Lit(60))],
                                targets=[Proj(F("val"), "val"), Proj(F("a"), "a")]),
         "select val, a from fact where val > 60 and not exists (select 1 from dk where dk.kd = fact.k)"),
    ]
    return st, sql_tables, cases, proj_cases
