"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer / COUNT results must match bit for bit; fp64 SUM/AVG within 1e-6 relative."""
import numpy as np
import pyarrow as pa
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cast, Cmp, ColRef, ExtractYear, INT32, INT64, KeyRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu


def _check(O, make, st, q, grid=0, flags=0):
    cp, want, err = run_oracle(O, st, q)
    assert err == 0
    ex = make(st)
    res = ex.execute(cp, grid=grid, flags=flags)
    assert_buffers_equal(cp, res.buffer, want)
    # the plan-interpreter kernel must agree too (it is the fallback for every other shape)
    res_g = ex.execute(cp, grid=grid, flags=flags | A.LAUNCH_FORCE_GENERIC)
    assert_buffers_equal(cp, res_g.buffer, want)
    res_s = ex.execute(cp, grid=grid, flags=flags | A.LAUNCH_FORCE_SCALAR)
    assert_buffers_equal(cp, res_s.buffer, want)
    return cp, res


def test_nocatalog_groupby_golden(oracle, gpu_executor_factory):
    # NoCatalogRelAlgTest.cpp:100-107,211-232 (reference golden): keys/vals in two fragments
    st = ArrowStorage()
    at = pa.table({"k": pa.array([1, 2, 1, 2, 1, 2, 1, 3, 1, 3], pa.int32()),
                   "v": pa.array([10, 20, 30, 40, 50, None, 70, None, 90, 100], pa.int32())})
    st.import_arrow(at, "test_agg", fragment_size=5)
    for columnar in (False, True):
        q = QueryUnit("test_agg", groupby=[ColRef("k")],
                      targets=[KeyRef(0, "k"), Agg("count", None, "cnt"), Agg("count", ColRef("v"), "cntv"),
                               Agg("sum", ColRef("v"), "sumv"), Agg("avg", ColRef("v"), "avgv")],
                      output_columnar=columnar)
        cp, res = _check(oracle, gpu_executor_factory, st, q)
        cols = res.to_columns()
        assert cols == {"k": [1, 2, 3], "cnt": [5, 3, 2], "cntv": [5, 2, 1], "sumv": [250, 60, 100],
                        "avgv": [50.0, 30.0, 100.0]}


def test_c1_plumbing_sum(oracle, gpu_executor_factory):
    # BASELINE config 1: SELECT SUM(a) FROM 1M-row int64 table
    st = ArrowStorage()
    st.import_arrow(pa.table({"a": pa.array(np.arange(1_000_000, dtype=np.int64))}), "t", fragment_size=300_000)
    q = QueryUnit("t", targets=[Agg("sum", ColRef("a"), "s")])
    cp, res = _check(oracle, gpu_executor_factory, st, q)
    assert res.to_columns() == {"s": [499_999_500_000]}


@pytest.mark.parametrize("nkeys,nulls", [(64, False), (64, True), (3, False), (1, True), (1000, False)])
def test_c2_shape_small(oracle, gpu_executor_factory, nkeys, nulls):
    rng = np.random.default_rng(20261002 + nkeys)
    n = 300_000
    key = rng.integers(0, nkeys, n, dtype=np.int64)
    val = rng.integers(-2**31, 2**31, n, dtype=np.int64)
    if nulls:
        val[rng.random(n) < 0.01] = A.NULL_BIGINT
        key[rng.random(n) < 0.01] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("t", {"key": key, "val": val}, fragment_size=70_001)
    q = QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s")])
    _check(oracle, gpu_executor_factory, st, q)
    # odd grid sizes exercise the tile walk
    _check(oracle, gpu_executor_factory, st, q, grid=7)


def test_all_aggs_int_and_fp(oracle, gpu_executor_factory):
    rng = np.random.default_rng(7)
    n = 200_000
    key = rng.integers(-5, 40, n, dtype=np.int32)
    iv = rng.integers(-1000, 1000, n, dtype=np.int32)
    fv = rng.normal(size=n)
    iv[rng.random(n) < 0.05] = A.NULL_INT
    fv_bits = fv.view(np.int64).copy()
    fv_bits[rng.random(n) < 0.05] = A.NULL_DOUBLE_BITS
    fv = fv_bits.view(np.float64)
    st = ArrowStorage()
    st.import_numpy("t", {"k": key, "i": iv, "f": fv}, fragment_size=50_000)
    for columnar in (False, True):
        q = QueryUnit("t", groupby=[ColRef("k")], output_columnar=columnar,
                      targets=[KeyRef(0, "k"), Agg("count"), Agg("min", ColRef("i")), Agg("max", ColRef("i")),
                               Agg("sum", ColRef("f")), Agg("min", ColRef("f")), Agg("max", ColRef("f")),
                               Agg("avg", ColRef("i"))])
        _check(oracle, gpu_executor_factory, st, q)


def test_filter_and_expressions(oracle, gpu_executor_factory):
    rng = np.random.default_rng(11)
    n = 150_000
    st = ArrowStorage()
    a = rng.integers(0, 100, n, dtype=np.int64)
    b = rng.integers(-50, 50, n, dtype=np.int32)
    b[rng.random(n) < 0.1] = A.NULL_INT
    st.import_numpy("t", {"a": a, "b": b}, fragment_size=40_000)
    q = QueryUnit("t", quals=[Cmp(ColRef("a"), ">=", Lit(10)), Cmp(ColRef("b") + 3, "<", Lit(20))],
                  groupby=[ColRef("a") / 10],
                  targets=[KeyRef(0, "bucket"), Agg("count"), Agg("sum", ColRef("b") * 2 + ColRef("a"), "s")])
    _check(oracle, gpu_executor_factory, st, q)


def test_non_grouped_multi_target(oracle, gpu_executor_factory):
    rng = np.random.default_rng(3)
    n = 123_457
    st = ArrowStorage()
    x = rng.integers(-10**9, 10**9, n, dtype=np.int64)
    x[rng.random(n) < 0.02] = A.NULL_BIGINT
    d = rng.normal(size=n) * 1e3
    st.import_numpy("t", {"x": x, "d": d}, fragment_size=50_000)
    q = QueryUnit("t", targets=[Agg("count"), Agg("count", ColRef("x")), Agg("sum", ColRef("x")),
                                Agg("min", ColRef("x")), Agg("max", ColRef("x")), Agg("avg", ColRef("d")),
                                Agg("sum", ColRef("d"))])
    _check(oracle, gpu_executor_factory, st, q)


def test_empty_and_ragged(oracle, gpu_executor_factory):
    st = ArrowStorage()
    st.import_numpy("e", {"k": np.zeros(0, dtype=np.int64), "v": np.zeros(0, dtype=np.int64)})
    q = QueryUnit("e", targets=[Agg("count"), Agg("sum", ColRef("v"))])
    cp, res = _check(oracle, gpu_executor_factory, st, q)
    assert res.to_columns() == {"count_0": [0], "sum_1": [None]}
    # ragged: 1-row and odd fragments
    st.import_numpy("r", {"k": np.arange(1001, dtype=np.int64) % 5, "v": np.arange(1001, dtype=np.int64)},
                    fragment_size=333)
    q = QueryUnit("r", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v")), Agg("count")])
    _check(oracle, gpu_executor_factory, st, q)


def test_multi_key_perfect_hash_taxi_like(oracle, gpu_executor_factory):
    rng = np.random.default_rng(5)
    n = 100_000
    st = ArrowStorage()
    pc = rng.integers(0, 7, n).astype(np.int16)
    ts = rng.integers(1230768000, 1451606400, n, dtype=np.int64)  # 2009..2015
    at = pa.table({"passenger_count": pa.array(pc), "pickup_datetime": pa.array(ts, pa.timestamp("s"))})
    st.import_arrow(at, "trips", fragment_size=30_000)
    q = QueryUnit("trips", groupby=[ColRef("passenger_count"), ExtractYear(ColRef("pickup_datetime"))],
                  targets=[KeyRef(0, "passenger_count"), KeyRef(1, "pickup_year"), Agg("count", None, "cnt")])
    cp, res = _check(oracle, gpu_executor_factory, st, q)
    assert cp.plan.query_kind == A.Q_PERFECT_HASH and cp.plan.key_count == 2


def test_transformed_keys_kernel(oracle, gpu_executor_factory):
    """hdk_scan_agg_keys (taxi Q3/Q4 shape: plain / year / decimal-cast keys, COUNT(*) only): NULL keys of every
    kind, timestamps outside the 32-bit fast range of extract_year, negative decimals, a plain filter, 1-3 keys.
    Checked against the oracle and against both interpreters."""
    from hdk_amd.ir import Type
    rng = np.random.default_rng(77)
    n = 150_000
    pc = rng.integers(0, 7, n).astype(np.int16)
    pc[rng.random(n) < 0.02] = A.NULL_SMALLINT
    ts = rng.integers(1230768000, 1451606400, n, dtype=np.int64)   # 2009..2015
    ts[rng.random(n) < 0.01] = -8_640_000                           # 1969: the 64-bit path of extract_year
    ts[rng.random(n) < 0.01] = 2_090_000_000                        # 2036: beyond the 32-bit fast range
    ts[rng.random(n) < 0.02] = A.NULL_BIGINT
    dist = rng.integers(-249, 850, n, dtype=np.int64)               # decimal(…,2): -2.49 .. 8.49 -> -2 .. 8
    dist[rng.random(n) < 0.02] = A.NULL_BIGINT
    flt = rng.integers(0, 100, n).astype(np.int32)
    flt[rng.random(n) < 0.02] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("trips", {"pc": pc, "ts": ts, "dist": dist, "flt": flt}, fragment_size=40_001,
                    types={"ts": Type("timestamp", 8, unit="s"), "dist": Type("decimal", 8, scale=2)})
    cnt = Agg("count", None, "cnt")
    cases = [
        ([ColRef("pc"), ExtractYear(ColRef("ts"))], [KeyRef(0), KeyRef(1), cnt], []),
        ([ColRef("pc"), ExtractYear(ColRef("ts")), Cast(ColRef("dist"), INT32)], [KeyRef(0), KeyRef(1), KeyRef(2), cnt], []),
        ([ExtractYear(ColRef("ts"))], [KeyRef(0), cnt], [Cmp(ColRef("flt"), "<", Lit(30))]),
        ([Cast(ColRef("dist"), INT32), ColRef("pc")], [cnt, KeyRef(1)], [Cmp(ColRef("flt"), ">=", Lit(50))]),
    ]
    for groupby, targets, quals in cases:
        for columnar in (False, True):
            q = QueryUnit("trips", quals=quals, groupby=groupby, targets=targets, output_columnar=columnar)
            cp, res = _check(oracle, gpu_executor_factory, st, q)
            ex = gpu_executor_factory(st)
            step = ex.prepare(cp)
            assert step.kernel_names().split(",")[0] == "hdk_scan_agg_keys", step.kernel_names()
            step.free()
            _check(oracle, gpu_executor_factory, st, q, grid=5)


def test_mid_size_tables_stay_in_lds(oracle, gpu_executor_factory):
    """Perfect-hash tables of 32-60 KiB (unreplicated, 2 blocks per CU) still take the LDS kernels; one word
    more and the plan falls to global atomics.  Both must match the oracle."""
    rng = np.random.default_rng(61)
    n = 400_000
    st = ArrowStorage()
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    v[rng.random(n) < 0.03] = A.NULL_BIGINT
    st.import_numpy("t", {"k5": rng.integers(0, 5000, n).astype(np.int32), "k2": rng.integers(0, 2400, n).astype(np.int64),
                          "k9": rng.integers(0, 9000, n).astype(np.int32), "v": v}, fragment_size=130_000)
    cases = [("k5", [KeyRef(0), Agg("count")], "hdk_scan_agg_direct"),                        # 5000 x 1 word
             ("k2", [KeyRef(0), Agg("sum", ColRef("v")), Agg("count")], "hdk_scan_agg_direct"),  # 2400 x 3 words
             ("k2", [KeyRef(0), Agg("min", ColRef("v") / 2)], "hdk_scan_agg_vec"),            # an expression outside the streaming kernel's menu -> batched kernel
             ("k9", [KeyRef(0), Agg("count")], "hdk_scan_agg_global")]                          # 9000 words: too big
    for key, targets, kernel in cases:
        q = QueryUnit("t", groupby=[ColRef(key)], targets=targets)
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp)
        assert step.kernel_names().split(",")[0] == kernel, (key, step.kernel_names())
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()


def test_graph_replay_matches_and_is_repeatable(oracle, gpu_executor_factory):
    """PreparedStep.capture_graph(): init + scan + finalize recorded as a hipGraph; every replay must leave
    the oracle's buffer (the init kernel is part of the graph, so replays do not accumulate)."""
    rng = np.random.default_rng(9)
    n = 200_000
    st = ArrowStorage()
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    v[rng.random(n) < 0.1] = A.NULL_BIGINT
    st.import_numpy("t", {"k": rng.integers(0, 50, n).astype(np.int32), "v": v, "d": rng.normal(size=n)}, fragment_size=70_000)
    for q in (QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v")), Agg("count")]),
              QueryUnit("t", quals=[Cmp(ColRef("v"), ">", Lit(0))], groupby=[ColRef("k")],
                        targets=[KeyRef(0), Agg("avg", ColRef("d")), Agg("min", ColRef("v") * 2)]),
              QueryUnit("t", targets=[Agg("sum", ColRef("v")), Agg("max", ColRef("d"))])):
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp).capture_graph()
        assert step._graph is not None
        for _ in range(3):
            step.replay()
            assert_buffers_equal(cp, step.fetch().buffer, want)
        step.free()


def test_transformed_keys_kernel_random(oracle, gpu_executor_factory):
    """Seeded random key combinations for hdk_scan_agg_keys: narrow and wide plain keys, year and decimal-cast keys,
    NULLs everywhere, random plain filters -- against the oracle and both interpreters."""
    from hdk_amd.ir import Type
    rng = np.random.default_rng(4242)
    n = 120_000

    def with_nulls(a, null, frac=0.02):
        a = a.copy()
        a[rng.random(n) < frac] = null
        return a

    ts = rng.integers(1230768000, 1451606400, n, dtype=np.int64)  # 2009..2015
    ts_wide = ts.copy()
    ts_wide[rng.random(n) < 0.01] = -8_640_000      # 1969: the 64-bit path of extract_year
    ts_wide[rng.random(n) < 0.01] = 2_090_000_000   # 2036: beyond the 32-bit fast range
    cols = {
        "k8": with_nulls(rng.integers(-3, 5, n).astype(np.int8), -128),
        "k16": with_nulls(rng.integers(0, 7, n).astype(np.int16), A.NULL_SMALLINT),
        "k32": rng.integers(100, 106, n).astype(np.int32),
        "k64": with_nulls(rng.integers(10**12, 10**12 + 5, n, dtype=np.int64), A.NULL_BIGINT),
        "ts": with_nulls(ts, A.NULL_BIGINT),
        "tsw": with_nulls(ts_wide, A.NULL_BIGINT),
        "dec": with_nulls(rng.integers(-249, 850, n, dtype=np.int64), A.NULL_BIGINT),
        "big": rng.integers(-10**15, 10**15, n, dtype=np.int64),  # decimal whose scaled value needs the 64-bit division
        "f": with_nulls(rng.integers(0, 100, n).astype(np.int32), A.NULL_INT),
    }
    st = ArrowStorage()
    st.import_numpy("t", cols, fragment_size=33_333,
                    types={"ts": Type("timestamp", 8, unit="s"), "tsw": Type("timestamp", 8, unit="s"),
                           "dec": Type("decimal", 8, scale=2), "big": Type("decimal", 8, scale=2)})
    pool = [lambda: ColRef("k8"), lambda: ColRef("k16"), lambda: ColRef("k32"), lambda: ColRef("k64"),
            lambda: ExtractYear(ColRef("ts")), lambda: ExtractYear(ColRef("tsw")), lambda: Cast(ColRef("dec"), INT32)]
    ran = 0
    for i in range(24):
        nk = int(rng.integers(1, 4))
        picks = rng.choice(len(pool), size=nk, replace=False)
        groupby = [pool[int(p)]() for p in picks]
        quals = []
        if rng.random() < 0.5:
            quals.append(Cmp(ColRef("f"), str(rng.choice(["<", ">=", "<>"])), Lit(int(rng.integers(10, 90)))))
        if rng.random() < 0.2:
            quals.append(Cmp(ColRef("big"), ">", Lit(0)))
        targets = [KeyRef(j) for j in range(nk)] + [Agg("count", None, "cnt")]
        q = QueryUnit("t", quals=quals, groupby=groupby, targets=targets, output_columnar=bool(rng.random() < 0.3))
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, (i, q)
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp)
        name = step.kernel_names().split(",")[0]
        step.free()
        if name != "hdk_scan_agg_keys":  # table too large for LDS, or a single plain key the streaming kernel takes
            continue
        try:
            _check(oracle, gpu_executor_factory, st, q)
        except AssertionError as e:
            raise AssertionError(f"case {i}: {q}\n{e}") from e
        ran += 1
    assert ran >= 8, ran


def test_transformed_keys_kernel_with_value_aggregates(oracle, gpu_executor_factory):
    """The value form of hdk_scan_agg_keys: SUM / AVG / MIN / MAX / COUNT over one or two plain columns (every width,
    signed, decimal, double, float; nullable or not) grouped by plain / year / decimal-cast keys, with filters --
    against the oracle and both interpreters.  Seeded random combinations."""
    from hdk_amd.ir import Type
    rng = np.random.default_rng(777)
    n = 120_000

    def with_nulls(a, null, frac=0.03):
        a = a.copy()
        a[rng.random(n) < frac] = null
        return a

    ts = rng.integers(1230768000, 1451606400, n, dtype=np.int64)
    ts[rng.random(n) < 0.01] = 2_090_000_000
    d = rng.normal(size=n) * 100
    d[rng.random(n) < 0.03] = np.finfo(np.float64).tiny      # NULL_DOUBLE
    f32 = (rng.random(n) * 100).astype(np.float32)
    f32[rng.random(n) < 0.03] = np.finfo(np.float32).tiny    # NULL_FLOAT
    cols = {
        "k16": with_nulls(rng.integers(0, 7, n).astype(np.int16), A.NULL_SMALLINT),
        "k32": rng.integers(100, 106, n).astype(np.int32),
        "ts": with_nulls(ts, A.NULL_BIGINT),
        "dec": with_nulls(rng.integers(-249, 850, n, dtype=np.int64), A.NULL_BIGINT),
        "v8": with_nulls(rng.integers(-100, 100, n).astype(np.int8), -128),
        "v16": with_nulls(rng.integers(-3000, 3000, n).astype(np.int16), A.NULL_SMALLINT),
        "v32": with_nulls(rng.integers(-10**6, 10**6, n).astype(np.int32), A.NULL_INT),
        "v64": with_nulls(rng.integers(-2**40, 2**40, n, dtype=np.int64), A.NULL_BIGINT),
        "nn": rng.integers(0, 1000, n, dtype=np.int64),
        "amt": with_nulls(rng.integers(0, 20000, n, dtype=np.int64), A.NULL_BIGINT),
        "d": d, "f32": f32,
        "flt": with_nulls(rng.integers(0, 100, n).astype(np.int32), A.NULL_INT),
    }
    st = ArrowStorage()
    st.import_numpy("t", cols, fragment_size=33_333,
                    types={"ts": Type("timestamp", 8, unit="s"), "dec": Type("decimal", 8, scale=2),
                           "amt": Type("decimal", 8, scale=2), "nn": Type("int", 8, False)})
    keys = [lambda: ColRef("k16"), lambda: ColRef("k32"), lambda: ExtractYear(ColRef("ts")),
            lambda: Cast(ColRef("dec"), INT32)]
    vals = ["v8", "v16", "v32", "v64", "nn", "amt", "d", "f32"]
    ran = 0
    for i in range(28):
        nk = int(rng.integers(1, 4))
        groupby = [keys[int(p)]() for p in rng.choice(len(keys), size=nk, replace=False)]
        if nk == 1 and isinstance(groupby[0], ColRef):
            groupby.append(ExtractYear(ColRef("ts")))  # a single plain key belongs to the streaming kernel
            nk = 2
        vcols = [vals[int(p)] for p in rng.choice(len(vals), size=int(rng.integers(1, 3)), replace=False)]
        targets = [KeyRef(j) for j in range(nk)]
        for t in range(int(rng.integers(1, 5))):
            targets.append(Agg(str(rng.choice(["sum", "avg", "min", "max", "count"])), ColRef(str(rng.choice(vcols)))))
        if rng.random() < 0.5:
            targets.append(Agg("count", None))
        quals = [Cmp(ColRef("flt"), str(rng.choice(["<", ">="])), Lit(int(rng.integers(20, 80))))] if rng.random() < 0.4 else []
        q = QueryUnit("t", quals=quals, groupby=groupby, targets=targets, output_columnar=bool(rng.random() < 0.3))
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, (i, q)
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp)
        name = step.kernel_names().split(",")[0]
        step.free()
        if name != "hdk_scan_agg_keys_values":  # e.g. the table does not fit LDS with this many words per entry
            continue
        try:
            _check(oracle, gpu_executor_factory, st, q)
            _check(oracle, gpu_executor_factory, st, q, grid=7)
        except AssertionError as e:
            raise AssertionError(f"case {i}: {q}\n{e}") from e
        ran += 1
    assert ran >= 12, ran


def test_keys_value_form_with_a_table_too_large_to_replicate(oracle, gpu_executor_factory, monkeypatch):
    """Taxi Q4 with a measure (three transformed keys, ~2 000 entries x 3-4 words): the LDS table leaves room for two or
    three blocks per CU, so the value form runs with 512 threads per block (scan_agg_keys.h: kKeysWideBlock).  Both block
    sizes against the oracle, ragged fragments, NULL keys and NULL measures, a small grid."""
    from hdk_amd.ir import Type
    rng = np.random.default_rng(4242)
    n = 400_013
    ts = rng.integers(1230768000, 1451606400, n, dtype=np.int64)
    ts[rng.random(n) < 0.01] = A.NULL_BIGINT
    pc = rng.integers(0, 7, n).astype(np.int16)
    pc[rng.random(n) < 0.02] = A.NULL_SMALLINT
    dist = rng.integers(0, 2700, n, dtype=np.int64)  # 8 x 8 x 28 = 1 792 entries x 3 or 4 words: 42-56 KB of LDS
    amt = rng.integers(0, 20000, n, dtype=np.int64)
    amt[rng.random(n) < 0.03] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("trips", {"pc": pc, "ts": ts, "dist": dist, "amt": amt}, fragment_size=55_555,
                    types={"ts": Type("timestamp", 8, unit="s"), "dist": Type("decimal", 8, scale=2), "amt": Type("decimal", 8, scale=2)})
    for targets in ([Agg("count", None, "c"), Agg("sum", ColRef("amt"), "s")],
                    [Agg("avg", ColRef("amt"), "a")], [Agg("max", ColRef("amt"), "m"), Agg("count", None, "c")]):
        q = QueryUnit("trips", groupby=[ColRef("pc"), ExtractYear(ColRef("ts")), Cast(ColRef("dist"), INT32)],
                      targets=[KeyRef(0), KeyRef(1), KeyRef(2)] + targets)
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        assert cp.entry_count * 3 * 8 > 40 * 1024, cp.entry_count  # fewer than four blocks of it fit a CU's 160 KB
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp)
        assert step.kernel_names().split(",")[0] == "hdk_scan_agg_keys_values"
        step.free()
        for no_wide in ("", "1"):
            if no_wide:
                monkeypatch.setenv("HDK_HIP_KEYS_NO_WIDE_BLOCK", "1")
            else:
                monkeypatch.delenv("HDK_HIP_KEYS_NO_WIDE_BLOCK", raising=False)
            for grid in (0, 5):
                assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, grid=grid).buffer, want)
    monkeypatch.delenv("HDK_HIP_KEYS_NO_WIDE_BLOCK", raising=False)


def test_direct_kernel_expression_arguments_and_column_filters(oracle, gpu_executor_factory):
    """The streaming kernel's wider menu (round 4): aggregate arguments `a op b` / `a op literal` over two plain 8-byte
    columns and filters `column cmp column` stay on hdk_scan_agg_direct (they used to fall to the batched interpreter at a
    third of the rate).  NULLs on either operand, NULL keys, every key width, ragged fragments; an overflow is reported
    with the reference's error code on every kernel family."""
    from hdk_amd._lib import HdkHipError
    rng = np.random.default_rng(61)
    n = 300_007
    a = rng.integers(-2**31, 2**31, n, dtype=np.int64)
    b = rng.integers(-1000, 1000, n, dtype=np.int64)
    c32 = rng.integers(-50, 50, n).astype(np.int32)
    a[rng.random(n) < 0.03] = A.NULL_BIGINT
    b[rng.random(n) < 0.03] = A.NULL_BIGINT
    k64 = rng.integers(0, 64, n, dtype=np.int64)
    k16 = rng.integers(-3, 40, n).astype(np.int16)
    k16[rng.random(n) < 0.02] = A.NULL_SMALLINT
    st = ArrowStorage()
    st.import_numpy("t", {"k64": k64, "k16": k16, "a": a, "b": b, "c32": c32, "c64": c32.astype(np.int64)}, fragment_size=77_777)
    ex = gpu_executor_factory(st)
    A_, B_ = ColRef("a"), ColRef("b")
    queries = [
        QueryUnit("t", groupby=[ColRef("k64")], targets=[KeyRef(0, "k"), Agg("sum", A_ * B_, "s")]),
        QueryUnit("t", groupby=[ColRef("k16")], targets=[KeyRef(0, "k"), Agg("sum", A_ + B_, "s"), Agg("count", A_ + B_, "c")]),
        QueryUnit("t", groupby=[ColRef("k64")], targets=[KeyRef(0, "k"), Agg("avg", A_ - 7, "av"), Agg("min", A_ - 7, "mn"), Agg("max", A_ - 7, "mx")]),
        QueryUnit("t", groupby=[ColRef("k64")], quals=[Cmp(A_, "<", B_)], targets=[KeyRef(0, "k"), Agg("sum", A_, "s")]),
        QueryUnit("t", groupby=[ColRef("k16")], quals=[Cmp(B_, ">=", ColRef("c64")), Cmp(A_, ">", Lit(-10**9))],
                  targets=[KeyRef(0, "k"), Agg("sum", A_ * 3, "s"), Agg("count", None, "c")]),
        QueryUnit("t", groupby=[ColRef("k64")], quals=[Cmp(ColRef("k64"), "<>", ColRef("c64")), Cmp(B_, "<", Lit(900))],
                  targets=[KeyRef(0, "k"), Agg("max", A_ - B_, "m")]),
        # a 4-byte column in a column-column filter is outside the streamed form: the batched interpreter takes the plan
        QueryUnit("t", groupby=[ColRef("k16")], quals=[Cmp(B_, ">=", ColRef("c32"))], targets=[KeyRef(0, "k"), Agg("sum", A_ * 3, "s")]),
    ]
    for qi, q in enumerate(queries):
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        step = ex.prepare(cp)
        assert step.kernel_names().startswith("hdk_scan_agg_direct" if qi < 6 else "hdk_scan_agg_vec"), (q, step.kernel_names())
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        for flags in (A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_SCALAR):
            assert_buffers_equal(cp, ex.execute(cp, flags=flags).buffer, want)
    # overflow of a BIGINT product: ERR_OVERFLOW_OR_UNDERFLOW (7) from the streaming kernel too
    st.get("t").columns["a"].fragments[1][5] = 2**62
    st.get("t").columns["b"].fragments[1][5] = 4
    cp, want, err = run_oracle(oracle, st, queries[0])
    assert err == A.ERR_OVERFLOW_OR_UNDERFLOW
    for flags in (0, A.LAUNCH_FORCE_GENERIC):
        with pytest.raises(HdkHipError) as ei:
            gpu_executor_factory(st).execute(cp, flags=flags)
        assert ei.value.code == A.ERR_OVERFLOW_OR_UNDERFLOW


@pytest.mark.parametrize("slice_log2,key_kind", [("6", "uniform"), ("6", "hot"), ("9", "uniform"), ("", "uniform")])
def test_perfect_hash_tables_beyond_lds_by_entry_range_partitions(oracle, gpu_executor_factory, monkeypatch, slice_log2, key_kind):
    """scan_agg_perfect_part.h: a GroupByPerfectHash table too large for LDS used to mean one or two global atomics per row;
    now the rows are scattered by entry range and every slice of the table is aggregated in LDS.  Forced onto small inputs
    (HDK_HIP_PERFECT_PARTITIONS_ALWAYS), with 64- and 512-entry slices (one and two scatter levels) and the default slice;
    keys of every width with and without NULLs (translated slot), filters, one and two argument columns, fp arguments,
    COUNT / SUM / AVG / MIN / MAX, ragged fragments; a hot key overflows its slab and the armed global-atomics kernel does the
    launch; the result equals the oracle's and the global-atomics kernel's."""
    monkeypatch.setenv("HDK_HIP_PERFECT_PARTITIONS_ALWAYS", "1")
    if slice_log2:
        monkeypatch.setenv("HDK_HIP_PERFECT_SLICE_LOG2", slice_log2)
    rng = np.random.default_rng(97)
    n = 700_003
    nk = 60_000
    if key_kind == "hot":
        k64 = np.where(rng.random(n) < 0.8, 1234, rng.integers(0, nk, n)).astype(np.int64) + 1000
    else:
        k64 = rng.integers(0, nk, n).astype(np.int64) + 1000
    k64n = k64.copy()
    k64n[rng.random(n) < 0.02] = A.NULL_BIGINT
    k32 = (k64 - 30_000).astype(np.int32)
    k32[rng.random(n) < 0.02] = A.NULL_INT
    k16 = rng.integers(-20_000, 20_000, n).astype(np.int16)
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    v[rng.random(n) < 0.03] = A.NULL_BIGINT
    w = rng.integers(-100, 100, n).astype(np.int32)
    d = rng.normal(size=n)
    d[rng.random(n) < 0.03] = np.finfo(np.float64).tiny
    flt = rng.integers(0, 100, n).astype(np.int32)
    st = ArrowStorage()
    g1 = rng.integers(-5, 300, n).astype(np.int32)
    g1[rng.random(n) < 0.02] = A.NULL_INT
    g2 = rng.integers(1000, 1090, n).astype(np.int64)
    g3 = rng.integers(0, 9, n).astype(np.int16)
    st.import_numpy("t", {"k64": k64, "k64n": k64n, "k32": k32, "k16": k16, "v": v, "w": w, "d": d, "flt": flt, "g1": g1, "g2": g2, "g3": g3},
                    fragment_size=123_457)
    V, W, D = ColRef("v"), ColRef("w"), ColRef("d")
    queries = [
        QueryUnit("t", groupby=[ColRef("k64")], targets=[KeyRef(0, "k"), Agg("sum", V, "s")]),
        QueryUnit("t", groupby=[ColRef("k64n")], targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("avg", V, "a"), Agg("max", W, "mw")]),
        QueryUnit("t", groupby=[ColRef("k32")], quals=[Cmp(ColRef("flt"), "<", Lit(70))],
                  targets=[KeyRef(0, "k"), Agg("min", V, "mn"), Agg("count", V, "cv"), Agg("sum", W, "sw")]),
        QueryUnit("t", groupby=[ColRef("k16")], targets=[KeyRef(0, "k"), Agg("sum", D, "sd"), Agg("count", None, "c")]),
        QueryUnit("t", groupby=[ColRef("k64n")], targets=[Agg("count", None, "c")]),
        QueryUnit("t", groupby=[ColRef("k32")], targets=[KeyRef(0, "k"), Agg("max", D, "md"), Agg("min", W, "mw")]),
        # arguments that are one checked integer step over plain columns (computed by the scatter pass)
        QueryUnit("t", groupby=[ColRef("k64n")], targets=[KeyRef(0, "k"), Agg("sum", V * W, "vw"), Agg("count", V * W, "c")]),
        QueryUnit("t", groupby=[ColRef("k32")], targets=[KeyRef(0, "k"), Agg("max", V - 7, "m"), Agg("sum", W + ColRef("flt"), "s")]),
        # several key columns (perfect_key_hash: strides over the keys' cardinalities), NULLs in one of them
        QueryUnit("t", groupby=[ColRef("g1"), ColRef("g2")], targets=[KeyRef(0, "a"), KeyRef(1, "b"), Agg("sum", V, "s"), Agg("count", None, "c")]),
        QueryUnit("t", groupby=[ColRef("g2"), ColRef("g3"), ColRef("g1")], quals=[Cmp(ColRef("flt"), ">=", Lit(10))],
                  targets=[KeyRef(2, "a"), KeyRef(0, "b"), Agg("min", W, "mw"), Agg("avg", V, "av")]),
    ]
    took = 0
    for q in queries:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp)
        names = step.kernel_names()
        if names.startswith("hdk_pp_scatter"):
            took += 1
        else:
            assert cp.entry_count * cp.plan.row_size_quad <= 7680, (q, names)  # (fits LDS: another kernel's)
        res = step.run()
        assert_buffers_equal(cp, res.buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS).buffer, want)
    assert took >= 9, took
    # a key outside the range the layout was sized for: the reference's get_group_value_fast has no check; here it is an error
    from hdk_amd._lib import HdkHipError
    tcol = st.get("t").columns["k64"]
    saved = tcol.fragments[1][7]
    tcol.fragments[1][7] = 10**9
    cp, want, err = run_oracle(oracle, st, queries[0]) if False else (ex.compile(queries[0]), None, 0)
    with pytest.raises(HdkHipError) as ei:
        gpu_executor_factory(st).execute(cp)
    assert ei.value.code == A.ERR_OUT_OF_SLOTS
    tcol.fragments[1][7] = saved
    # an argument outside the 32 bits its statistics promised (8-byte packed tuples): the armed global-atomics kernel redoes it
    vcol = st.get("t").columns["v"]
    saved = vcol.fragments[2][11]
    vcol.fragments[2][11] = 2**40 + 5
    cp, want, err = run_oracle(oracle, st, queries[0])
    assert err == 0
    step = gpu_executor_factory(st).prepare(cp)
    assert step.kernel_names().startswith("hdk_pp_scatter")
    assert_buffers_equal(cp, step.run().buffer, want)
    step.free()
    vcol.fragments[2][11] = saved
