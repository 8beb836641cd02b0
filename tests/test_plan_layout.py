"""Host logic: layout decisions mirror the reference's MemoryLayoutBuilder / QueryMemoryDescriptor rules
(file:line cited per case).  CPU only."""
import numpy as np
import pyarrow as pa
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, BinOp, Cast, ColRef, ExtractYear, INT32, KeyRef, Lit, QueryMustRunOnCpu, QueryUnit
from hdk_amd.plan import (compact_init_vals, columnar_slot_offsets, compile_query, init_buffer_host)
from hdk_amd.storage import ArrowStorage


def _st(n=1000, nulls=False, keys=64):
    rng = np.random.default_rng(1)
    val = rng.integers(-2**31, 2**31, n).astype(np.int64)
    if nulls:
        val[::17] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("t", {"key": rng.integers(0, keys, n).astype(np.int64), "val": val}, fragment_size=300)
    return st


def test_c2_layout_keyless_when_sum_arg_has_no_nulls():
    # get_keyless_info (MemoryLayoutBuilder.cpp:296-325): SUM over a nullable-typed column whose range
    # has no NULLs can never equal its init value -> keyless, slot index 1 marks empty entries
    cp = compile_query(_st(), QueryUnit("t", groupby=[ColRef("key")],
                                        targets=[KeyRef(0), Agg("sum", ColRef("val"))]))
    p = cp.plan
    assert p.query_kind == A.Q_PERFECT_HASH and p.entry_count == 64
    assert p.keyless == 1 and p.idx_target_as_key == 1
    assert p.row_size_quad == 2 and cp.buffer_bytes == 64 * 16
    assert cp.init_vals.tolist() == [0, A.NULL_BIGINT]
    assert p.targets[0].slot_off == 0 and p.targets[1].slot_off == 8


def test_c2_layout_keyed_when_nulls_present():
    cp = compile_query(_st(nulls=True), QueryUnit("t", groupby=[ColRef("key")],
                                                  targets=[KeyRef(0), Agg("sum", ColRef("val"))]))
    p = cp.plan
    assert p.keyless == 0 and p.row_size_quad == 3  # [key][key target][sum]
    assert p.targets[0].slot_off == 8 and p.targets[1].slot_off == 16
    buf = init_buffer_host(cp).reshape(64, 3)
    assert (buf[:, 0] == A.EMPTY_KEY_64).all() and (buf[:, 1] == 0).all() and (buf[:, 2] == A.NULL_BIGINT).all()


def test_null_keys_add_an_entry_and_translate():
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array([5, None, 7, 9], pa.int32()), "v": pa.array([1, 2, 3, 4], pa.int64())}), "t")
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("count")]))
    p = cp.plan
    # getBucketedCardinality: max-min+1 (+1 with nulls) (ColRangeInfo.cpp:24-27); NULL -> max+1 (RowFuncBuilder.cpp:456-461)
    assert p.entry_count == 6 and p.key_has_nulls[0] == 1 and p.key_null_translated[0] == 10 and p.key_min[0] == 5


def test_compact_count_slots():
    # pick_target_compact_width (MemoryLayoutBuilder.cpp:559-652): single key, COUNT(*) only -> 4-byte slots
    st = ArrowStorage()
    st.import_arrow(pa.table({"c": pa.array(["green", "yellow", "green"])}), "t")
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("c")], targets=[KeyRef(0), Agg("count")]))
    assert cp.slot_widths == [4, 4] and cp.plan.row_size_quad == 1 and cp.plan.keyless == 1
    cp8 = compile_query(st, QueryUnit("t", groupby=[ColRef("c")], targets=[KeyRef(0), Agg("count")], bigint_count=True))
    assert cp8.slot_widths == [8, 8]
    assert compact_init_vals(cp).tolist() == [0]


def test_perfect_vs_baseline_switch():
    st = ArrowStorage()
    n = 100
    st.import_numpy("t", {"a": np.arange(n, dtype=np.int64) * 10_000_000, "b": np.arange(n, dtype=np.int64) % 7,
                          "c": np.arange(n, dtype=np.int64) % 5})
    # single column: range >= 2^30 / ((keys+targets)*8) -> baseline (MemoryLayoutBuilder.cpp:176-179,228-236)
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("a")], targets=[KeyRef(0), Agg("count")]))
    assert cp.plan.query_kind == A.Q_BASELINE_HASH
    # multi column: product of cardinalities <= baseline_threshold -> perfect, entry_count = product (:121-162)
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("b"), ColRef("c")], targets=[KeyRef(0), KeyRef(1), Agg("count")]))
    assert cp.plan.query_kind == A.Q_PERFECT_HASH and cp.plan.entry_count == 35 and cp.plan.keyless == 0
    assert [cp.plan.key_card[i] for i in range(2)] == [7, 5]
    # modulo has no expression range -> baseline (ExpressionRange.cpp:416-419)
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("b") % 3], targets=[KeyRef(0), Agg("count")]))
    assert cp.plan.query_kind == A.Q_BASELINE_HASH and cp.plan.key_width == 8  # Invalid range -> 8-byte component


def test_baseline_key_width_and_columnar_offsets():
    st = ArrowStorage()
    st.import_numpy("t", {"a": np.arange(50, dtype=np.int32), "b": np.arange(50, dtype=np.int32) % 3, "v": np.ones(50)})
    q = QueryUnit("t", groupby=[ColRef("a"), ColRef("b")], force_baseline=True, baseline_entry_count=101,
                  targets=[KeyRef(0), KeyRef(1), Agg("avg", ColRef("v"))])
    cp = compile_query(st, q)
    assert cp.plan.key_width == 4  # pick_baseline_key_width: both ranges fit int32
    # projected keys have zero-width slots in a baseline table (target_groupby_indices,
    # MemoryLayoutBuilder.cpp:921-927 + ColSlotContext.cpp:43-48): row = align8(2*4) + AVG's two 8-byte slots
    assert cp.plan.row_size_quad == 1 + 2
    assert cp.slot_widths == [0, 0, 8, 8]
    assert [cp.plan.targets[i].slot_width for i in range(3)] == [0, 0, 8]
    q.output_columnar = True
    cpc = compile_query(st, q)
    assert cpc.plan.key_width == 8  # columnar group keys are 64-bit (QueryMemoryDescriptor.cpp:344-372)
    offs = columnar_slot_offsets(cpc)
    assert offs == [2 * 101 * 8, 2 * 101 * 8, 2 * 101 * 8, 3 * 101 * 8]
    assert cpc.buffer_bytes == 4 * 101 * 8


def test_non_grouped_init_is_null_sentinel_and_skip():
    # OutputBufferInitialization.cpp:57-60 + TargetExprBuilder.cpp:546-551
    st = ArrowStorage()
    st.import_numpy("t", {"a": np.arange(10, dtype=np.int64)})
    cp = compile_query(st, QueryUnit("t", targets=[Agg("sum", ColRef("a")), Agg("count"), Agg("min", ColRef("a")),
                                                   Agg("avg", ColRef("a"))]))
    assert cp.plan.query_kind == A.Q_NON_GROUPED
    assert cp.init_vals.tolist() == [A.NULL_BIGINT, 0, A.NULL_BIGINT, A.NULL_BIGINT, 0]
    assert [cp.plan.targets[i].skip_null for i in range(4)] == [1, 0, 1, 1]


def test_min_max_keep_argument_null():
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array([1, 1, 2], pa.int8()), "v": pa.array([3, None, 4], pa.int32())}), "t")
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("k")],
                                     targets=[KeyRef(0), Agg("min", ColRef("v")), Agg("sum", ColRef("v"))]))
    # MIN/MAX: init = the ARGUMENT type's NULL widened (get_agg_initial_val on int32 in an 8-byte slot); SUM: BIGINT NULL
    assert cp.plan.targets[1].null_val == A.NULL_INT and cp.plan.targets[2].null_val == A.NULL_BIGINT
    assert cp.init_vals.tolist() == [0, A.NULL_INT, A.NULL_BIGINT]


def test_single_value_layout():
    # checked_single_agg_id is never a *_skip_val call and always takes the ARGUMENT type's NULL
    # (QE/TargetExprBuilder.cpp:429-445,542-546); its slot starts like MAX's (get_agg_initial_val,
    # QE/OutputBufferInitialization.cpp:211-250): the type's NULL when nullable, the integer minimum of the SLOT width
    # when not -- which for a non-nullable INT32 argument in an 8-byte slot is not the NULL the function is handed
    # (the reference's own quirk, restated as it is); non-grouped queries are always nullable (:57-60)
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array([1, 1, 2], pa.int8()), "v": pa.array([3, None, 4], pa.int32()),
                              "w": pa.array([5, 5, 6], pa.int32())}), "t")
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("single_value", ColRef("v")),
                                                                        Agg("single_value", ColRef("w"))]))
    t1, t2 = cp.plan.targets[1], cp.plan.targets[2]
    assert t1.agg == t2.agg == A.AGG_SINGLE_VALUE and t1.skip_null == t2.skip_null == 0
    assert t1.null_val == t2.null_val == A.NULL_INT
    assert cp.plan.keyless == 0  # (get_keyless_info: default -> keyless = false, MemoryLayoutBuilder.cpp:398-400)
    w_nullable = st.get("t").columns["w"].type.nullable
    assert cp.init_vals.tolist() == [0, A.NULL_INT, A.NULL_INT if w_nullable else -(2**63)]
    cp = compile_query(st, QueryUnit("t", targets=[Agg("single_value", ColRef("w"))]))
    assert cp.plan.query_kind == A.Q_NON_GROUPED and cp.init_vals.tolist() == [A.NULL_INT]


def test_unsupported_shapes_raise_query_must_run_on_cpu():
    st = ArrowStorage()
    st.import_numpy("t", {"a": np.arange(10, dtype=np.int64), "f": np.ones(10, dtype=np.float32), "d": np.ones(10)})
    # (round 6: a FLOAT key column is its value widened to double in the key word, as castToTypeIn(group_key, 64) makes it; a cast
    # of an integer to FLOAT only while the argument's statistics lie inside +-2^24, where the step library's conversion to double
    # is the same value; a computed FLOAT key otherwise stays out)
    cpf = compile_query(st, QueryUnit("t", groupby=[ColRef("f")], targets=[Agg("count")]))
    assert cpf.plan.query_kind == A.Q_BASELINE_HASH and cpf.plan.key_width == 8
    from hdk_amd.ir import Cast, FP32
    ok = compile_query(st, QueryUnit("t", groupby=[Cast(ColRef("a"), FP32)], targets=[Agg("count")]))
    assert ok.plan.keys[0].nsteps == 1 and ok.plan.keys[0].null_val == A.to_i64(FP32.null_as_int64_or_double_bits())
    st.import_numpy("wide", {"a": np.array([0, 1 << 30], dtype=np.int64)})
    with pytest.raises(QueryMustRunOnCpu):
        compile_query(st, QueryUnit("wide", groupby=[Cast(ColRef("a"), FP32)], targets=[Agg("count")]))  # would need the rounding to float
    # (a double key is its bit pattern in an open-addressing table since round 5: groupByColumnCodegen, QE/IRCodegen.cpp:1219-1221)
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("d")], targets=[Agg("count")]))
    assert cp.plan.query_kind == A.Q_BASELINE_HASH and cp.plan.key_width == 8
    deep = BinOp("+", ColRef("a"), BinOp("*", ColRef("a"), Lit(2)))
    with pytest.raises(QueryMustRunOnCpu):
        compile_query(st, QueryUnit("t", targets=[Agg("sum", deep)]))  # right operand must be a leaf


def test_taxi_q4_plan_shape():
    import sys, os
    from taxi import load_taxi, taxi_queries
    st = ArrowStorage()
    load_taxi(st)
    cp = compile_query(st, taxi_queries()[3])
    p = cp.plan
    assert p.query_kind == A.Q_PERFECT_HASH and p.key_count == 3
    assert p.keys[1].steps[0].op == A.OP_EXTRACT_YEAR and p.keys[2].steps[0].op == A.OP_SCALE_DOWN
    assert p.keys[2].steps[0].rhs.ival == 100


@pytest.mark.parametrize("columnar", [False, True])
@pytest.mark.parametrize("shape", ["perfect_keyed", "perfect_keyless", "compact4", "baseline8", "baseline4", "two_keys"])
def test_host_init_image_equals_oracle_init(oracle, shape, columnar):
    """The numpy image `init_buffer_host` (a host convenience) against the oracle's restatement of the reference's
    init kernels (QE/GpuInitGroups.cu:17-166) -- every layout family; with two different background fills the
    oracle's output may differ only in padding bytes, which the image has as zero."""
    from util import oracle_init_buffer
    rng = np.random.default_rng(5)
    n = 500
    st = ArrowStorage()
    v = rng.integers(-100, 100, n).astype(np.int64)
    v[::9] = A.NULL_BIGINT
    st.import_numpy("t", {"k": rng.integers(0, 30, n).astype(np.int64), "w": rng.integers(0, 5, n).astype(np.int32) * 1000,
                          "v": v, "d": rng.normal(size=n), "big": rng.integers(0, 30, n).astype(np.int64) * (2**33)})
    tg_all = [KeyRef(0), Agg("count"), Agg("sum", ColRef("v")), Agg("min", ColRef("d")), Agg("avg", ColRef("v"))]
    Q = {
        "perfect_keyed": QueryUnit("t", groupby=[ColRef("k")], targets=tg_all),
        "perfect_keyless": QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("count"), Agg("max", ColRef("d"))]),
        "compact4": QueryUnit("t", groupby=[ColRef("w")], targets=[KeyRef(0), Agg("count")]),
        "baseline8": QueryUnit("t", groupby=[ColRef("big")], force_baseline=True, targets=tg_all),
        "baseline4": QueryUnit("t", groupby=[ColRef("k")], force_baseline=True, targets=tg_all),
        "two_keys": QueryUnit("t", groupby=[ColRef("k"), ColRef("w")], targets=[KeyRef(1), KeyRef(0)] + tg_all[1:]),
    }
    q = Q[shape]
    q.output_columnar = columnar
    if columnar and shape == "baseline4":
        q.groupby = [ColRef("big")]  # columnar tables keep 8-byte keys
    cp = compile_query(st, q)
    img = init_buffer_host(cp)
    a = oracle_init_buffer(oracle, cp, fill=0)
    b = oracle_init_buffer(oracle, cp, fill=-1)
    assert np.array_equal(img[:cp.buffer_quads], a[:cp.buffer_quads])
    pad = (a != b)
    if pad.any():  # only possible behind 4-byte keys (row-wise) or between columns (columnar)
        assert cp.plan.key_width == 4 or columnar
        ab, bb = a.view(np.uint8), b.view(np.uint8)
        assert (ab != bb).sum() <= (8 * cp.entry_count if not columnar else 8 * (len(cp.slot_widths) + 4))


def test_float_argument_targets_get_float_accumulators():
    """takes_float_argument (Shared/TargetInfo.h:170-179): SUM / MIN / MAX / AVG over FLOAT keep a float in the low 4
    bytes of an 8-byte slot; init values are the 4-byte patterns sign-extended (OutputBufferInitialization.cpp:52-65,
    get_agg_initial_val :112-258); COUNT(float) stays an integer slot."""
    st = ArrowStorage()
    st.import_arrow(pa.table({"k": pa.array([1, 1, 2], pa.int32()), "f": pa.array([1.5, None, 2.5], pa.float32())}), "t")
    cp = compile_query(st, QueryUnit("t", groupby=[ColRef("k")], targets=[
        KeyRef(0), Agg("sum", ColRef("f")), Agg("min", ColRef("f")), Agg("max", ColRef("f")), Agg("avg", ColRef("f")),
        Agg("count", ColRef("f"))]))
    tg = cp.plan.targets
    assert [tg[i].arg_is_fp for i in range(1, 6)] == [A.FP_SLOT_FLOAT] * 4 + [A.FP_SLOT_DOUBLE]
    assert [tg[i].slot_width for i in range(1, 6)] == [8] * 5
    widened = np.array([np.float32(np.finfo(np.float32).tiny)], dtype=np.float64).view(np.int64)[0]
    assert all(tg[i].null_val == widened for i in range(1, 5))
    assert cp.init_vals.tolist() == [0, A.NULL_FLOAT_BITS, A.NULL_FLOAT_BITS, A.NULL_FLOAT_BITS, A.NULL_FLOAT_BITS, 0, 0]
    st2 = ArrowStorage()
    st2.import_numpy("t", {"k": np.array([1, 2], dtype=np.int32), "f": np.array([1.0, 2.0], dtype=np.float32)},
                     types=None)
    from hdk_amd.ir import Type
    st2.get("t").columns["f"].type = Type("fp", 4, False)
    cp2 = compile_query(st2, QueryUnit("t", groupby=[ColRef("k")], targets=[
        Agg("sum", ColRef("f")), Agg("min", ColRef("f")), Agg("max", ColRef("f"))]))
    fmax = int(np.array([np.finfo(np.float32).max], dtype=np.float32).view(np.int32)[0])
    fmin = int(np.array([-np.finfo(np.float32).max], dtype=np.float32).view(np.int32)[0])
    assert cp2.init_vals.tolist() == [0, fmax, fmin] and fmin < 0   # -FLT_MAX sign-extends into the high bytes
