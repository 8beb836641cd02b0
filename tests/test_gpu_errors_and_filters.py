"""Error protocol and filter logic on the device: OR / NOT filters against the oracle;
ERR_DIV_BY_ZERO, ERR_OVERFLOW_OR_UNDERFLOW (checked + - *, QE/ArithmeticIR.cpp:277-520), ERR_INTERRUPTED and
ERR_OUT_OF_TIME (QE/cuda_mapd_rt.cu:105-148) reported through ERROR_CODE[0] with the reference's codes
(QE/Execute.h:1019-1031), whatever kernel runs the plan."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import HdkHipError
from hdk_amd.ir import Agg, And, Cmp, ColRef, KeyRef, Lit, Not, Or, Proj, QueryUnit
from hdk_amd.storage import ArrowStorage

from test_gpu_baseline import _check_rows
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu

STRATEGIES = (0, A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_SCALAR, A.LAUNCH_FORCE_GLOBAL_ATOMICS)


def _table(n=300_000, seed=5):
    rng = np.random.default_rng(seed)
    a = rng.integers(-50, 50, n).astype(np.int32)
    a[rng.random(n) < 0.1] = A.NULL_INT
    b = rng.integers(-50, 50, n).astype(np.int64)
    b[rng.random(n) < 0.1] = A.NULL_BIGINT
    d = rng.normal(size=n)
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 40, n).astype(np.int64), "a": a, "b": b, "d": d,
                          "big": rng.integers(0, 9000, n).astype(np.int64) * (2**34)}, fragment_size=70_000)
    return st


def test_or_not_filters_match_the_oracle(oracle, gpu_executor_factory):
    st = _table()
    A_, B_, D_ = ColRef("a"), ColRef("b"), ColRef("d")
    filters = [
        [Or(Cmp(A_, "<", Lit(0)), Cmp(B_, ">", Lit(5)))],
        [Not(Or(Cmp(A_, "<", Lit(0)), Cmp(B_, ">", Lit(5))))],
        [Or(And(Cmp(A_, ">=", Lit(-5)), Cmp(A_, "<=", Lit(5))), Not(Cmp(D_, "<", Lit(0.25)))), Cmp(B_, "<>", Lit(3))],
    ]
    seen_kernels = set()
    for quals in filters:
        # perfect hash, open addressing, non-grouped, projection
        queries = [
            QueryUnit("t", quals=quals, groupby=[ColRef("k")], targets=[KeyRef(0, "k"), Agg("count", None, "c"),
                                                                       Agg("sum", ColRef("b"), "s"), Agg("avg", D_, "ad")]),
            QueryUnit("t", quals=quals, groupby=[ColRef("big")], force_baseline=True, baseline_entry_count=30_011,
                      targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("min", ColRef("b"), "mn")]),
            QueryUnit("t", quals=quals, targets=[Agg("count", None, "c"), Agg("sum", ColRef("a"), "s")]),
        ]
        for q in queries:
            cp, want, err = run_oracle(oracle, st, q)
            assert err == 0 and cp.plan.num_filter_ops > 0
            step = gpu_executor_factory(st).prepare(cp)
            if cp.plan.query_kind != A.Q_BASELINE_HASH:
                # up to three `column cmp literal` leaves: the specialised kernels run the program themselves (plain_quals.h,
                # round 5); more leaves: the batched interpreter (filter_program_pass_v)
                names = step.kernel_names()
                assert names.startswith(("hdk_scan_agg_vec", "hdk_scan_agg_direct")), names
                seen_kernels.add(names.split(",")[0])
            res = step.run()
            step.free()
            if cp.plan.query_kind != A.Q_BASELINE_HASH:  # and through the batched interpreter whatever the pick was
                assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)
            if cp.plan.query_kind == A.Q_BASELINE_HASH:
                _check_rows(cp, res.buffer, want)
            else:
                assert_buffers_equal(cp, res.buffer, want)
                # and row at a time (filter_program_pass)
                assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_SCALAR).buffer, want)
        from test_gpu_projection import _sorted_rows
        from test_projection import run_projection_oracle
        qp = QueryUnit("t", quals=quals, targets=[Proj(ColRef("k"), "k"), Proj(ColRef("b"), "b")], output_columnar=True)
        cp, want, err, nrows = run_projection_oracle(oracle, st, qp)
        step = gpu_executor_factory(st).prepare(cp)
        assert step.kernel_names() == "hdk_scan_project", step.kernel_names()
        res = step.run()
        step.free()
        assert err == 0 and res.total_matched == nrows
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))


def test_or_filters_on_the_streaming_kernels(oracle, gpu_executor_factory):
    """`WHERE val < 0 OR key = 3` (c2or of scripts/bench_configs.py) and its relatives on the kernels that took only
    conjunctions until round 5: the streaming kernel (one key, one argument) and the on-chip open-addressing kernels (the
    multi-key shape stays with the interpreter).  NULLs in every filter column: three-valued OR / AND / NOT (a NULL leaf under OR passes when the other side is
    TRUE, NOT NULL is NULL)."""
    from hdk_amd.ir import Cast
    rng = np.random.default_rng(77)
    n = 700_000
    key = rng.integers(0, 64, n, dtype=np.int64)
    val = rng.integers(-2**31, 2**31, n, dtype=np.int64)
    val[rng.random(n) < 0.05] = A.NULL_BIGINT
    c = rng.integers(-50, 50, n).astype(np.int32)
    c[rng.random(n) < 0.1] = A.NULL_INT
    f = rng.normal(size=n)
    f[rng.random(n) < 0.1] = np.frombuffer(np.uint64(A.NULL_DOUBLE_BITS).tobytes(), dtype=np.float64)[0]
    x = rng.integers(1, 20, n).astype(np.int32)
    y = rng.integers(1, 11, n).astype(np.int32)
    y[rng.random(n) < 0.05] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", {"key": key, "val": val, "c": c, "f": f, "x": x, "y": y}, fragment_size=n // 3 + 11)
    K, V, C, F = ColRef("key"), ColRef("val"), ColRef("c"), ColRef("f")
    programs = [
        [Or(Cmp(V, "<", Lit(0)), Cmp(K, "=", Lit(3)))],
        [Not(Or(Cmp(C, "<", Lit(0)), Cmp(F, ">", Lit(0.5))))],
        [Or(And(Cmp(C, ">=", Lit(-5)), Cmp(C, "<=", Lit(5))), Not(Cmp(F, "<", Lit(0.25))))],
        [And(Or(Cmp(C, "<", Lit(10)), Cmp(V, ">", Lit(0))), Cmp(K, "<>", Lit(7)))],
    ]
    from test_gpu_baseline import _check_rows
    from hdk_amd.ir import FP64
    for quals in programs:
        shapes = [
            (QueryUnit("t", quals=quals, groupby=[K], targets=[KeyRef(0, "k"), Agg("sum", V, "s")]), "hdk_scan_agg_direct"),
            # (two keys: the keys kernel carries no program code -- it cost the unfiltered taxi queries 3-4 % -- the interpreter)
            (QueryUnit("t", quals=quals, groupby=[K, ColRef("x")], targets=[KeyRef(0, "k"), KeyRef(1, "x"), Agg("count", None, "n")]), "hdk_scan_agg_vec"),
            (QueryUnit("t", quals=quals, groupby=[Cast(ColRef("x"), FP64)],
                       targets=[KeyRef(0, "k"), Agg("count", ColRef("y"), "n"), Agg("sum", ColRef("y"), "s"), Agg("min", ColRef("y"), "mn")]),
             "hdk_scan_agg_bh_"),
        ]
        for q, kernel in shapes:
            cp, want, err = run_oracle(oracle, st, q)
            assert err == 0 and cp.plan.num_filter_ops > 0
            step = gpu_executor_factory(st).prepare(cp)
            assert step.kernel_names().startswith(kernel), (q, step.kernel_names())
            res = step.run()
            step.free()
            if cp.plan.query_kind == A.Q_BASELINE_HASH:
                _check_rows(cp, res.buffer, want)
            else:
                assert_buffers_equal(cp, res.buffer, want)
            other = gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC)
            if cp.plan.query_kind == A.Q_BASELINE_HASH:
                _check_rows(cp, other.buffer, want)
            else:
                assert_buffers_equal(cp, other.buffer, want)


def test_filter_program_deeper_than_the_streaming_stack(oracle, gpu_executor_factory):
    """`(a AND b) OR ((a AND c) OR (b AND c))`: three leaves (plan.py deduplicates them), eleven ops, a value stack of FOUR --
    one more than the streaming kernels' evaluators keep in registers (plain_quals.h, scan_agg_fast.h).  The matchers must
    hand it to the interpreter (host_match.h: filter_program_depth); the result equals the oracle's either way."""
    from hdk_amd.ir import Cast, FP64
    from test_gpu_baseline import _check_rows
    rng = np.random.default_rng(78)
    n = 400_000
    c = rng.integers(-50, 50, n).astype(np.int32)
    c[rng.random(n) < 0.1] = A.NULL_INT
    val = rng.integers(-2**31, 2**31, n, dtype=np.int64)
    val[rng.random(n) < 0.05] = A.NULL_BIGINT
    y = rng.integers(1, 11, n).astype(np.int32)
    y[rng.random(n) < 0.05] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", {"key": rng.integers(0, 64, n, dtype=np.int64), "val": val, "c": c, "x": rng.integers(1, 20, n).astype(np.int32),
                          "y": y}, fragment_size=n // 3 + 7)
    K, V, C_ = ColRef("key"), ColRef("val"), ColRef("c")
    a, b, cc = Cmp(C_, "<", Lit(10)), Cmp(V, ">", Lit(0)), Cmp(K, "<>", Lit(7))
    deep = [Or(And(a, b), Or(And(a, cc), And(b, cc)))]
    left = [Or(Or(And(a, b), And(a, cc)), And(b, cc))]  # (the same predicate, left-nested: depth 3, stays on the streaming kernels)
    for quals, streaming in ((deep, False), (left, True)):
        shapes = [
            (QueryUnit("t", quals=quals, groupby=[K], targets=[KeyRef(0, "k"), Agg("sum", V, "s")]), "hdk_scan_agg_direct"),
            (QueryUnit("t", quals=quals, groupby=[Cast(ColRef("x"), FP64)],
                       targets=[KeyRef(0, "k"), Agg("count", ColRef("y"), "n"), Agg("sum", ColRef("y"), "s")]), "hdk_scan_agg_bh_dense"),
        ]
        for q, kernel in shapes:
            cp, want, err = run_oracle(oracle, st, q)
            assert err == 0 and cp.plan.num_quals == 3 and cp.plan.num_filter_ops == 11
            step = gpu_executor_factory(st).prepare(cp)
            names = step.kernel_names()
            assert names.startswith(kernel) == streaming and ("_vec" in names) != streaming, (quals, names)
            res = step.run()
            step.free()
            if cp.plan.query_kind == A.Q_BASELINE_HASH:
                _check_rows(cp, res.buffer, want)
            else:
                assert_buffers_equal(cp, res.buffer, want)


def _expect_error(oracle, gpu_executor_factory, st, q, code):
    cp, want, err = run_oracle(oracle, st, q)
    assert err == code, err
    for flags in STRATEGIES:
        if flags == A.LAUNCH_FORCE_GLOBAL_ATOMICS and cp.plan.query_kind == A.Q_NON_GROUPED:
            continue
        with pytest.raises(HdkHipError) as ei:
            gpu_executor_factory(st).execute(cp, flags=flags)
        assert ei.value.code == code, (flags, ei.value.code)


def test_division_by_zero_is_reported(oracle, gpu_executor_factory):
    st = _table()
    for q in (QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("b") / ColRef("a"))]),
              QueryUnit("t", quals=[Cmp(ColRef("b") % ColRef("a"), "=", Lit(1))], targets=[Agg("count")]),
              QueryUnit("t", targets=[Agg("sum", ColRef("d") / ColRef("a"))])):
        _expect_error(oracle, gpu_executor_factory, st, q, A.ERR_DIV_BY_ZERO)


def test_integer_overflow_is_reported(oracle, gpu_executor_factory):
    """+ - * are checked in the operation's SQL type like the reference's generated code: int32 * int32 overflows at 32
    bits although the value would fit the int64 it is carried in; int64 at 64 bits; rows with a NULL operand are not
    checked; the same expressions on values that fit give results and no error."""
    rng = np.random.default_rng(6)
    n = 200_000
    x32 = rng.integers(-60_000, 60_000, n).astype(np.int32)
    y32 = rng.integers(-60_000, 60_000, n).astype(np.int32)
    x64 = rng.integers(-2**40, 2**40, n).astype(np.int64)
    small = rng.integers(-1000, 1000, n).astype(np.int32)
    y32n = y32.copy()
    y32n[:] = A.NULL_INT  # every product has a NULL operand: nothing to check, nothing overflows
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 10, n).astype(np.int64), "x32": x32, "y32": y32, "x64": x64, "small": small,
                          "y32n": y32n}, fragment_size=60_000)
    bad = [QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("x32") * ColRef("y32"))]),
           QueryUnit("t", targets=[Agg("max", ColRef("x64") * ColRef("x64"))]),
           QueryUnit("t", quals=[Cmp(ColRef("x32") + Lit(2_147_480_000), ">", Lit(0))], targets=[Agg("count")]),
           QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("min", ColRef("x64") * Lit(2**30))])]
    for q in bad:
        _expect_error(oracle, gpu_executor_factory, st, q, A.ERR_OVERFLOW_OR_UNDERFLOW)
    good = [QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("small") * ColRef("small"))]),
            QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("x32") * ColRef("y32n")),
                                                          Agg("count", ColRef("x32") * ColRef("y32n"))]),
            QueryUnit("t", targets=[Agg("sum", ColRef("x64") + ColRef("x64"))])]
    for q in good:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        for flags in STRATEGIES[:3]:
            assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=flags).buffer, want)


def test_interrupt_and_watchdog(oracle, gpu_executor_factory):
    """hdk_hip_set_interrupt + HDK_HIP_LAUNCH_CHECK_INTERRUPT -> ERR_INTERRUPTED; watchdog_ms -> ERR_OUT_OF_TIME; a
    launch that asks for neither is not affected by a raised flag; re-arming gives the normal result again."""
    rng = np.random.default_rng(8)
    n = 8_000_000
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 50, n).astype(np.int64), "v": rng.integers(-100, 100, n).astype(np.int64),
                          "big": rng.integers(0, 200_000, n).astype(np.int64) * (2**33)}, fragment_size=1_000_000)
    shapes = [(QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v"))]), (0, A.LAUNCH_FORCE_GENERIC,
                                                                                                     A.LAUNCH_FORCE_SCALAR)),
              (QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v") + ColRef("k"))]), (0,)),
              (QueryUnit("t", groupby=[ColRef("big")], force_baseline=True, baseline_entry_count=600_011,
                         targets=[KeyRef(0), Agg("sum", ColRef("v"))]), (0, A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_PARTITIONED)),
              (QueryUnit("t", quals=[Cmp(ColRef("v"), ">", Lit(90))], targets=[Proj(ColRef("k"), "k")], output_columnar=True),
               (0, A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_SCALAR))]
    ex = gpu_executor_factory(st)
    try:
        for q, flag_set in shapes:
            cp = ex.compile(q)
            for flags in flag_set:
                ex.interrupt(1)
                with pytest.raises(HdkHipError) as ei:
                    ex.execute(cp, flags=flags | A.LAUNCH_CHECK_INTERRUPT)
                assert ei.value.code == A.ERR_INTERRUPTED, (flags, ei.value.code)
                ex.execute(cp, flags=flags)  # not asked to look at the flag: runs to the end
                ex.interrupt(0)
                ex.execute(cp, flags=flags | A.LAUNCH_CHECK_INTERRUPT)
        # watchdog: the row-at-a-time interpreter over 8 M rows takes far longer than 1 ms of device time
        cp = ex.compile(shapes[0][0])
        with pytest.raises(HdkHipError) as ei:
            ex.execute(cp, flags=A.LAUNCH_FORCE_SCALAR, grid=8, watchdog_ms=1)
        assert ei.value.code == A.ERR_OUT_OF_TIME
        cp2, want, err = run_oracle(oracle, st, shapes[0][0], frag_ids=[0])
        res = ex.execute(cp2, frag_ids=[0], watchdog_ms=60_000)
        assert_buffers_equal(cp2, res.buffer, want)
    finally:
        ex.interrupt(0)


def test_watch_state_belongs_to_the_launch(oracle, gpu_executor_factory):
    """The interrupt / watchdog words live in the launch's own workspace (csrc/watch.h), not in a device global:
    (1) a watchdog launch that timed out does not leave a stale deadline behind for a hipGraph captured earlier
    (replays of the graph keep giving the oracle's buffer); (2) a captured graph carries its own re-arming node, so
    a replay after another step's watchdog launch on the SAME device is clean; (3) an unarmed launch on the same device
    does not disarm the watchdog of another step."""
    rng = np.random.default_rng(81)
    n = 4_000_000
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 50, n).astype(np.int64), "v": rng.integers(-100, 100, n).astype(np.int64)},
                    fragment_size=1_000_000)
    q = QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v"))])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    ex = gpu_executor_factory(st)
    graph_step = ex.prepare(cp).capture_graph()
    assert graph_step._graph is not None
    graph_step.replay()
    assert_buffers_equal(cp, graph_step.fetch().buffer, want)
    # a launch that runs out of its 1 ms budget (row-at-a-time interpreter, 8 blocks) ...
    slow = ex.prepare(cp, flags=A.LAUNCH_FORCE_SCALAR, grid=8, watchdog_ms=1)
    with pytest.raises(HdkHipError) as ei:
        slow.run()
    assert ei.value.code == A.ERR_OUT_OF_TIME
    # ... leaves nothing behind: graph replays (no host-side arming happens for them) and plain launches are clean
    for _ in range(2):
        graph_step.replay()
        assert_buffers_equal(cp, graph_step.fetch().buffer, want)
    plain = ex.prepare(cp)
    assert_buffers_equal(cp, plain.run().buffer, want)
    # the slow step still times out after unarmed launches ran on the device (they do not touch its words),
    # and its error word is its own
    with pytest.raises(HdkHipError) as ei:
        slow.run()
    assert ei.value.code == A.ERR_OUT_OF_TIME
    # the same step (its workspace last held an armed launch) relaunched WITHOUT a budget runs to the end
    slow.ko.watchdog_ms = 0
    slow.mgr.zeroDeviceMem(slow.d_err.ptr, 4, slow.dev)
    assert_buffers_equal(cp, slow.run().buffer, want)
    for s_ in (graph_step, plain):
        s_.free()
    slow.free()


# ---- float accumulators (takes_float_argument) ----------------------------------------------------------------------
def _float_table(n=200_000, seed=23):
    import pyarrow as pa
    rng = np.random.default_rng(seed)
    f = (rng.random(n) * 100).astype(np.float32)          # positive: the relative error of a float sum stays small
    g = (rng.random(n) * 50 - 25).astype(np.float32)
    null = rng.random(n) < 0.1
    k = rng.integers(0, 300, n).astype(np.int32)
    null[k == 7] = True                                   # a group with nothing but NULLs keeps the float sentinel
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array(k, pa.int32()), "f": pa.array(f, pa.float32(), mask=null),
                              "g": pa.array(g, pa.float32()),
                              "big": pa.array(rng.integers(0, 2000, n).astype(np.int64) * (2**34), pa.int64())}),
                    "t", fragment_size=45_000)
    return st


def _float_targets():
    F, G = ColRef("f"), ColRef("g")
    return [Agg("sum", F, "s"), Agg("min", F, "lo"), Agg("max", F, "hi"), Agg("avg", F, "a"), Agg("count", F, "c"),
            Agg("min", G, "glo"), Agg("max", G, "ghi")]


@pytest.mark.parametrize("columnar", [False, True])
def test_float_accumulators_match_the_oracle(oracle, gpu_executor_factory, columnar):
    """SUM / MIN / MAX / AVG over FLOAT: a float in the low 4 bytes of the slot (agg_*_float[_skip_val],
    QE/RuntimeFunctions.cpp:770-875).  MIN / MAX / COUNT and every other byte of the buffer bit-exact; float sums
    within float32 summation error (the oracle adds row by row in float; the device's order differs)."""
    st = _float_table()
    q = QueryUnit("t", groupby=[ColRef("k")], output_columnar=columnar, targets=[KeyRef(0, "k")] + _float_targets())
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.targets[1].arg_is_fp == A.FP_SLOT_FLOAT
    for flags in STRATEGIES:
        res = gpu_executor_factory(st).execute(cp, flags=flags)
        assert_buffers_equal(cp, res.buffer, want)
    cols = res.to_columns()
    i7 = cols["k"].index(7)
    assert (cols["s"][i7], cols["lo"][i7], cols["hi"][i7], cols["a"][i7], cols["c"][i7]) == (None, None, None, None, 0)


def test_float_accumulators_non_grouped_and_open_addressing(oracle, gpu_executor_factory):
    st = _float_table()
    q = QueryUnit("t", targets=_float_targets())
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    for flags in (0, A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_SCALAR):
        # 200 K rows in one float accumulator: the row-order float sum itself is only good to ~1e-3
        assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=flags).buffer, want, float32_rtol=5e-3)
    qb = QueryUnit("t", groupby=[ColRef("big")], force_baseline=True, baseline_entry_count=9001,
                   targets=[KeyRef(0, "big")] + _float_targets())
    cp, want, err = run_oracle(oracle, st, qb)
    assert err == 0 and cp.plan.query_kind == A.Q_BASELINE_HASH
    for flags in (0, A.LAUNCH_FORCE_GENERIC):
        _check_rows(cp, gpu_executor_factory(st).execute(cp, flags=flags).buffer, want, rtol=2e-4)


def test_float_accumulators_reduce_on_device(oracle, gpu_executor_factory):
    """hdk_hip_reduce_buffers on float slots == the oracle's reduction of the same partials in the same order
    (get_width_for_slot -> sizeof(float), QE/ResultSetReduction.cpp:1176-1185), bit for bit."""
    import ctypes as C
    from hdk_amd._lib import check, lib
    st = _float_table(90_000, 29)
    q = QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0, "k")] + _float_targets())
    ex = gpu_executor_factory(st)
    cp = ex.compile(q)
    mgr = ex.mgr
    partials = [ex.execute(cp, frag_ids=[fr]).buffer.copy() for fr in range(2)]
    want = partials[0].copy()
    assert oracle.reduce(cp.plan, want, cp.entry_count, partials[1], cp.entry_count, cp.init_vals) == 0
    d0, d1 = mgr.to_device(partials[0], 0), mgr.to_device(partials[1], 0)
    that = (C.c_void_p * 1)(d1.ptr)
    counts = (C.c_uint32 * 1)(cp.entry_count)
    d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
    iv = np.ascontiguousarray(cp.init_vals)
    check(lib().hdk_hip_reduce_buffers(C.byref(cp.plan), d0.ptr, cp.entry_count, that, counts, 1, iv.ctypes.data,
                                       d_err.ptr, 0, None))
    mgr.synchronizeStream(0)
    assert np.array_equal(mgr.to_host(d0.ptr, cp.buffer_bytes, 0), want[:cp.buffer_bytes // 8])
