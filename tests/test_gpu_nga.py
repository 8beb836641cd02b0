"""Non-grouped aggregates over several columns on the column-by-column streaming kernel (hdk_scan_agg_cols,
hdk_amd/csrc/scan_agg_cols.h): the reference's NonGroupedAgg benchmark queries (NGA01-05) and their edge cases, bit-exact
against the oracle (agg_*[_skip_val], QE/RuntimeFunctions.cpp:456-476,612-660; double sums within 1e-6 relative)."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cmp, ColRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from syn_queries import NGA_COLS, nga, syn_table
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu

NULL_DOUBLE = np.frombuffer(np.uint64(A.NULL_DOUBLE_BITS).tobytes(), dtype=np.float64)[0]


def _run(oracle, make, st, q, kernel="hdk_scan_agg_cols"):
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    step = make(st).prepare(cp)
    assert step.kernel_names().split(",")[0] == kernel, step.kernel_names()
    res = step.run()
    step.free()
    assert_buffers_equal(cp, res.buffer, want)
    other = make(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC)  # the interpreter agrees
    assert_buffers_equal(cp, other.buffer, want)
    return res


@pytest.mark.parametrize("null_frac", [0.0, 0.03])
def test_nga_benchmark_queries(oracle, gpu_executor_factory, null_frac):
    rng = np.random.default_rng(11)
    n = 1_300_003  # ragged against every tile size; fragments of different lengths
    cols = syn_table(rng, n, NGA_COLS, null_frac=null_frac)
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=400_001)
    for i in range(1, 6):
        res = _run(oracle, gpu_executor_factory, st, nga(i))
        out = res.to_columns()
        if i == 1:
            assert out["c"][0] == n
            assert out["c0"][0] == int((cols["x10"] != A.NULL_INT).sum())
        if i == 2:
            live = cols["z100"][cols["z100"] != A.NULL_INT].astype(np.int64)
            assert out["a5"][0] == int(live.sum())


def test_all_null_column_and_every_width(oracle, gpu_executor_factory):
    """An all-NULL column (SUM / MIN / MAX / AVG stay NULL, COUNT is 0), NULLs in every column, 1- / 2- / 4- / 8-byte
    integers and doubles in one query, tiny and empty fragments."""
    rng = np.random.default_rng(12)
    n = 70_001
    i8 = rng.integers(-100, 100, n).astype(np.int8)
    i8[rng.random(n) < 0.1] = -128
    i16 = rng.integers(-30_000, 30_000, n).astype(np.int16)
    i16[rng.random(n) < 0.1] = A.NULL_SMALLINT
    i32 = rng.integers(-2**31 + 1, 2**31, n).astype(np.int32)
    i32[rng.random(n) < 0.1] = A.NULL_INT
    i64 = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    i64[rng.random(n) < 0.1] = A.NULL_BIGINT
    d = rng.normal(size=n) * 1e6
    d[rng.random(n) < 0.1] = NULL_DOUBLE
    allnull = np.full(n, A.NULL_INT, dtype=np.int32)
    st = ArrowStorage()
    st.import_numpy("t", {"i8": i8, "i16": i16, "i32": i32, "i64": i64, "d": d, "allnull": allnull}, fragment_size=9_973)
    names = ("i8", "i16", "i32", "i64", "d", "allnull")
    for kind in ("count", "sum", "min", "max", "avg"):
        q = QueryUnit("t", targets=[Agg("count", None, "n")] + [Agg(kind, ColRef(c), f"{kind}_{c}") for c in names])
        res = _run(oracle, gpu_executor_factory, st, q)
        out = res.to_columns()
        if kind == "count":
            assert out["count_allnull"][0] == 0 and out["n"][0] == n
        else:
            assert out[f"{kind}_allnull"][0] is None
        if kind == "min":
            assert out["min_i32"][0] == int(i32[i32 != A.NULL_INT].min()) and out["min_i8"][0] == int(i8[i8 != -128].min())
        if kind == "max":
            assert out["max_i64"][0] == int(i64[i64 != A.NULL_BIGINT].max())
    # several aggregates of the same column next to others (one pass per distinct column)
    q = QueryUnit("t", targets=[Agg("sum", ColRef("i32"), "s"), Agg("min", ColRef("i32"), "mn"), Agg("avg", ColRef("i32"), "a"),
                                Agg("max", ColRef("i16"), "mx"), Agg("count", ColRef("i16"), "c"), Agg("sum", ColRef("d"), "sd")])
    _run(oracle, gpu_executor_factory, st, q)


def test_small_inputs_and_repeated_launches_accumulate(oracle, gpu_executor_factory):
    """Fewer rows than one 16-byte chunk, one row, and a second launch into the same out_vec (the slots accumulate like
    repeated row-function calls: agg_sum adds, agg_min keeps the smaller)."""
    rng = np.random.default_rng(13)
    for n in (1, 3, 5, 257):
        st = ArrowStorage()
        st.import_numpy("syn", syn_table(rng, n, NGA_COLS, null_frac=0.2 if n > 3 else 0.0), fragment_size=max(n // 2, 1))
        for i in (1, 2, 3, 4, 5):
            _run(oracle, gpu_executor_factory, st, nga(i))
    st = ArrowStorage()
    cols = syn_table(rng, 50_000, NGA_COLS, null_frac=0.05)
    st.import_numpy("syn", cols, fragment_size=20_000)
    ex = gpu_executor_factory(st)
    cp = ex.compile(nga(2))
    step = ex.prepare(cp)
    step.init_output()
    step.launch()
    step.launch()
    out = step.fetch().to_columns()
    step.free()
    live = cols["x10"][cols["x10"] != A.NULL_INT].astype(np.int64)
    assert out["a0"][0] == 2 * int(live.sum())


def test_filtered_or_single_column_plans_keep_their_kernels(oracle, gpu_executor_factory):
    """A filter ties the columns' rows together: not this kernel's shape (the interpreter); one column alike in every
    target: the one-argument streaming kernel as before."""
    rng = np.random.default_rng(14)
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(rng, 100_000, NGA_COLS, null_frac=0.02), fragment_size=40_000)
    q = QueryUnit("syn", quals=[Cmp(ColRef("x10"), ">", Lit(3))], targets=[Agg("sum", ColRef("x100"), "a"), Agg("sum", ColRef("y100"), "b")])
    _run(oracle, gpu_executor_factory, st, q, kernel="hdk_scan_agg_vec")
    q1 = QueryUnit("syn", targets=[Agg("sum", ColRef("x100"), "a"), Agg("count", ColRef("x100"), "b")])
    _run(oracle, gpu_executor_factory, st, q1, kernel="hdk_scan_agg_direct")
