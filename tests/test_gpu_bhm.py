"""The multi-argument / multi-key on-chip group-by (hdk_scan_agg_bhm, hdk_amd/csrc/scan_bhm.h) against the oracle: the
reference's MultiStep (MSBS001, MSPHS001) and PerfectHashMultiCol (PHM001, PHM002) benchmark shapes and their edge cases --
NULL keys and arguments, statistics that do not hold (the armed fallback), ragged fragments, row-wise and columnar tables,
few groups (replicated LDS tables) and thousands (one 1024-thread block per CU).  Integers bit-exact; placement of an
open-addressing table asserted through the reference's probe sequence."""
import dataclasses

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cast, ColRef, FP64, KeyRef, QueryUnit
from hdk_amd.storage import ArrowStorage

from syn_queries import msbs, msphs, phm, syn_table
from test_gpu_baseline import _assert_reference_placement, _check_rows
from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu

BHM = "hdk_scan_agg_bhm"


def _run(oracle, make, st, q, kernel=BHM):
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    step = make(st).prepare(cp)
    assert kernel is None or step.kernel_names().split(",")[0] == kernel, step.kernel_names()
    res = step.run()
    step.free()
    if cp.plan.query_kind == A.Q_BASELINE_HASH:
        _check_rows(cp, res.buffer, want)
        if not cp.plan.output_columnar:
            _assert_reference_placement(oracle, cp, res.buffer)
    else:
        assert_buffers_equal(cp, res.buffer, want)
    other = make(st).execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS)  # the reference's own scheme gives the same groups
    if cp.plan.query_kind == A.Q_BASELINE_HASH:
        _check_rows(cp, other.buffer, want)
    else:
        assert_buffers_equal(cp, other.buffer, want)
    return cp, res


@pytest.mark.parametrize("null_frac", [0.0, 0.04])
def test_multistep_and_multicol_benchmark_shapes(oracle, gpu_executor_factory, null_frac):
    rng = np.random.default_rng(21)
    n = 900_007
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(rng, n, ("x10", "y10", "z10", "x100", "x1k"), null_frac=null_frac), fragment_size=300_011)
    for q in (msbs(1), msbs(1, key_type=FP64), msphs(1), phm(1), phm(2)):  # (MSBS as written: cast(x1k AS float); and its double twin)
        cp, res = _run(oracle, gpu_executor_factory, st, q)
        step = gpu_executor_factory(st).prepare(cp)
        names = step.kernel_names()
        step.free()
        assert names.endswith("hdk_bhm_fold" if cp.plan.query_kind == A.Q_BASELINE_HASH else "hdk_finalize"), names
    # columnar output tables
    for q in (msphs(1), phm(2), msbs(1, key_type=FP64)):
        _run(oracle, gpu_executor_factory, st, dataclasses.replace(q, output_columnar=True))


def test_few_groups_replicated_tables_and_every_aggregate(oracle, gpu_executor_factory):
    """Ten groups (sixteen LDS replicas), MIN and MAX and AVG and COUNT of three columns, `column - literal` and
    `column * literal` arguments, negative values."""
    rng = np.random.default_rng(22)
    n = 500_003
    a = rng.integers(-500, 500, n).astype(np.int32)
    a[rng.random(n) < 0.05] = A.NULL_INT
    b = rng.integers(0, 70_000, n).astype(np.int32)
    c = rng.integers(-3, 4, n).astype(np.int32)
    c[rng.random(n) < 0.5] = A.NULL_INT
    k = rng.integers(-4, 6, n).astype(np.int32)
    k[rng.random(n) < 0.02] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", {"k": k, "a": a, "b": b, "c": c}, fragment_size=170_001)
    A_, B_, C_ = ColRef("a"), ColRef("b"), ColRef("c")
    # (HDK_HIP_MAX_TARGETS = 8 per plan)
    t1 = [KeyRef(0, "k"), Agg("count", None, "n"), Agg("min", A_, "mna"), Agg("max", A_, "mxa"), Agg("avg", A_, "ava"),
          Agg("sum", A_ * 3, "s3a"), Agg("count", C_, "cc"), Agg("max", B_ - 7, "mxb")]
    t2 = [KeyRef(0, "k"), Agg("min", C_, "mnc"), Agg("sum", B_, "sb"), Agg("avg", C_, "avc"), Agg("min", B_, "mnb"), Agg("max", C_, "mxc")]
    for groupby in ([ColRef("k")], [Cast(ColRef("k"), FP64)]):
        for targets in (t1, t2):
            _run(oracle, gpu_executor_factory, st, QueryUnit("t", groupby=groupby, targets=targets))


def test_three_key_columns_and_three_argument_columns(oracle, gpu_executor_factory):
    rng = np.random.default_rng(23)
    n = 400_009
    st = ArrowStorage()
    cols = {"k0": rng.integers(1, 8, n).astype(np.int32), "k1": rng.integers(-3, 4, n).astype(np.int32),
            "k2": rng.integers(100, 120, n).astype(np.int32), "a": rng.integers(0, 1000, n).astype(np.int32),
            "b": rng.integers(-50, 50, n).astype(np.int32), "c": rng.integers(1, 5, n).astype(np.int32)}
    cols["k1"][rng.random(n) < 0.03] = A.NULL_INT
    cols["b"][rng.random(n) < 0.1] = A.NULL_INT
    st.import_numpy("t", cols, fragment_size=150_000)
    q = QueryUnit("t", groupby=[ColRef("k0"), ColRef("k1"), ColRef("k2")],
                  targets=[KeyRef(0, "k0"), KeyRef(1, "k1"), KeyRef(2, "k2"), Agg("sum", ColRef("a"), "sa"), Agg("max", ColRef("a"), "mxa"),
                           Agg("avg", ColRef("b"), "avb"), Agg("min", ColRef("b") + 1, "mnb"), Agg("count", ColRef("c"), "cc")])
    _run(oracle, gpu_executor_factory, st, q)


def test_statistics_that_do_not_hold_take_the_armed_fallback(oracle, gpu_executor_factory):
    """An argument outside its announced range, a NULL where the statistics announce none, a key outside the table's range:
    the on-chip kernel raises its flag, its folds skip, and the global-atomics kernel armed behind them redoes the launch --
    the result is the oracle's either way (never a wrong result from stale metadata)."""
    rng = np.random.default_rng(24)
    n = 300_000
    base = syn_table(rng, n, ("x10", "x100", "x1k"))
    for what in ("argument", "null", "key"):
        cols = {c: v.copy() for c, v in base.items()}
        st = ArrowStorage()
        st.import_numpy("syn", cols, fragment_size=100_000)
        t = st.get("syn")
        if what == "argument":
            t.columns["x100"].fragments[1][777] = 100_000  # statistics say [1, 100]
        elif what == "null":
            t.columns["x10"].fragments[2][5] = A.NULL_INT  # has_nulls = False
        for q in (msphs(1), msbs(1, key_type=FP64)):
            if what == "key":
                if q.groupby[0] == ColRef("x1k"):
                    continue  # (a perfect-hash key outside its range is OUT_OF_SLOTS by contract)
                t.columns["x1k"].fragments[0][3] = 1234  # an open-addressing key outside the dense range: still a valid group
            cp, want, err = run_oracle(oracle, st, q)
            assert err == 0
            step = gpu_executor_factory(st).prepare(cp)
            assert step.kernel_names().startswith(BHM)
            res = step.run()
            step.free()
            if cp.plan.query_kind == A.Q_BASELINE_HASH:
                _check_rows(cp, res.buffer, want)
            else:
                assert_buffers_equal(cp, res.buffer, want)


def test_routing_limits(oracle, gpu_executor_factory):
    """What stays with the other kernels: one plain argument column (the packed kernels), tables beyond LDS."""
    rng = np.random.default_rng(25)
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(rng, 200_000, ("x10", "y10", "x100", "x1k", "x100k")), fragment_size=100_000)
    y = ColRef("y10")
    one_arg = QueryUnit("syn", groupby=[Cast(ColRef("x1k"), FP64)], targets=[KeyRef(0, "k")] + [Agg(kd, y) for kd in ("count", "sum", "max", "min", "avg")])
    ex = gpu_executor_factory(st)
    s1 = ex.prepare(ex.compile(one_arg))
    assert s1.kernel_names().startswith("hdk_scan_agg_bh_dense")
    s1.free()
    s2 = ex.prepare(ex.compile(msphs(3)))
    assert not s2.kernel_names().startswith(BHM)
    s2.free()


@pytest.mark.parametrize("groups,hot", [(9_000, 0.0), (8_000, 0.7)])
def test_bh004_size_class_in_one_pass(oracle, gpu_executor_factory, groups, hot):
    """BaselineHash/BH004 and PerfectHashSingleCol/PHS004 (10 000 groups, five aggregates of one column): 24 bytes an entry
    do not fit a CU's LDS (the packed kernels took two passes: 0.19 of the roofline), 12 bytes do -- one pass, also with a hot
    key (70 % of the rows in one group: nothing overflows, the packed fields are sized for a block's rows)."""
    from test_gpu_bh_lds import _bh_query, _mid_table, _phs_query
    st = _mid_table(3_300_000, groups, 27, hot)
    for q in (_bh_query("x"), _phs_query("x")):
        cp, res = _run(oracle, gpu_executor_factory, st, q)


# ---- tables beyond a CU's LDS: the two-pass form (scan_bhm_part.h) ---------------------------------------------------------------
PART = "hdk_bhm_scatter"


@pytest.mark.parametrize("null_frac", [0.0, 0.03])
def test_two_pass_form_for_tables_beyond_lds(oracle, gpu_executor_factory, monkeypatch, null_frac):
    """MSBS002 / MSPHS002 / PHM003-004's size class (10 K - 100 K groups): 4-byte tuples [entry in the bin | argument codes] by
    key range, a bin's table in LDS, eight slabs folded like the one-pass kernel's.  Ragged fragments, NULLs, both layouts."""
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")  # (the row count here is far below where the passes pay)
    rng = np.random.default_rng(31)
    n = 1_200_011
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(rng, n, ("x10", "y10", "z10", "x100", "x1k", "x10k", "x100k"), null_frac=null_frac), fragment_size=400_003)
    for q in (msbs(2), msphs(2), msphs(3), msbs(3, key_type=FP64), phm(4)):
        cp, res = _run(oracle, gpu_executor_factory, st, q, kernel=PART)
    _run(oracle, gpu_executor_factory, st, dataclasses.replace(msphs(2), output_columnar=True), kernel=PART)
    # five aggregates of one column by 100 K groups (BH005 / PHS005's shape): entry-in-bin + code fit 16 bits -- 2-byte tuples;
    # HDK_HIP_BHM_WIDE_TUPLES keeps them at 4 bytes
    y = ColRef("y10")
    for wide in ("", "1"):
        monkeypatch.setenv("HDK_HIP_BHM_WIDE_TUPLES", wide) if wide else monkeypatch.delenv("HDK_HIP_BHM_WIDE_TUPLES", raising=False)
        for groupby in ([Cast(ColRef("x100k"), FP64)], [ColRef("x100k")]):
            q = QueryUnit("syn", groupby=groupby, targets=[KeyRef(0, "k")] + [Agg(kd, y, kd) for kd in ("count", "sum", "max", "min", "avg")])
            _run(oracle, gpu_executor_factory, st, q, kernel=PART)
            _run(oracle, gpu_executor_factory, st, dataclasses.replace(q, targets=q.targets[:3]), kernel=PART)


def test_two_pass_form_hot_key_stays_on_chip_in_generations(oracle, gpu_executor_factory, monkeypatch):
    """70 % of the rows in ONE group: the sample gives that bin sub-slabs of its size, and pass B -- whose packed fields hold 2^23
    rows of a table -- flushes its LDS table into the slab generation by generation (here: every 4 096 tuples, so that every
    block does it many times and every word kind is joined: counts, sums, MIN, MAX, NULL counts).  No fallback (the hook)."""
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    monkeypatch.setenv("HDK_HIP_BHM_FLAG_IS_ERROR", "1")
    monkeypatch.setenv("HDK_HIP_BHM_PART_GENERATION", "1")
    rng = np.random.default_rng(34)
    n = 1_000_000
    cols = syn_table(rng, n, ("x10", "y10", "z10", "x100", "x10k", "x100k"), null_frac=0.02)
    cols["x10k"][rng.random(n) < 0.7] = 4242
    cols["x100k"][rng.random(n) < 0.5] = 77_777
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=333_334)
    y = ColRef("y10")
    five = QueryUnit("syn", groupby=[Cast(ColRef("x100k"), FP64)], targets=[KeyRef(0, "k")] + [Agg(kd, y, kd) for kd in ("count", "sum", "max", "min", "avg")])
    for q in (msphs(2), msbs(2, key_type=FP64), msphs(3), five, phm(4)):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)
    # ... and the hot bin's sub-slabs cut into parts whose blocks join their tables in the slab with atomics (here: any sub-slab
    # beyond 8 192 tuples; generations of 4 096 inside the parts, and without)
    monkeypatch.setenv("HDK_HIP_BHM_PART_TUPLES", "1")
    for q in (msphs(2), msbs(2, key_type=FP64), five, phm(4)):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)
    monkeypatch.delenv("HDK_HIP_BHM_PART_GENERATION")
    for q in (msphs(3), five):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)


def test_two_pass_form_falls_back_on_a_hot_key_and_on_stale_statistics(oracle, gpu_executor_factory, monkeypatch):
    """70 % of the rows in one group (the sample sees it: no overflow any more -- the result must be the oracle's either way); an
    argument outside its statistics: the flag is raised, the folds skip and the armed global-atomics kernel gives the oracle's
    result."""
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    rng = np.random.default_rng(32)
    n = 900_000
    cols = syn_table(rng, n, ("x10", "x100", "x10k"))
    cols["x10k"][rng.random(n) < 0.7] = 4242
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=300_000)
    for q in (msphs(2), msbs(2, key_type=FP64)):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)
    st2 = ArrowStorage()
    cols2 = syn_table(rng, n, ("x10", "x100", "x10k"))
    st2.import_numpy("syn", cols2, fragment_size=300_000)
    st2.get("syn").columns["x100"].fragments[1][99] = 7_000  # statistics say [1, 100]
    _run(oracle, gpu_executor_factory, st2, msphs(2), kernel=PART)


def test_two_pass_form_sizes_its_bins_from_a_sample(oracle, gpu_executor_factory, monkeypatch):
    """Uneven bins -- 5 % of NULL keys (ONE entry), a Zipf-like key distribution, a key range that is half empty -- used to overflow
    the sub-slabs of uniform size and send the whole launch to the global-atomics fallback (1 270 ms instead of 1.3 ms per 256 M
    rows).  Now a sample of the keys sizes every bin's sub-slabs.  HDK_HIP_BHM_FLAG_IS_ERROR turns a fallback into an error:
    these inputs must stay on the fast path; and a sample that cannot see the skew (stride beyond the input) still falls back."""
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    monkeypatch.setenv("HDK_HIP_BHM_FLAG_IS_ERROR", "1")
    rng = np.random.default_rng(33)
    n = 1_500_000
    cols = syn_table(rng, n, ("x10", "x100", "x10k", "x100k"))
    cols["x10k"][rng.random(n) < 0.05] = A.NULL_INT
    zipf = np.minimum(rng.zipf(1.3, n), 100_000).astype(np.int32)  # (the first keys carry most rows: the first bins)
    cols["x100k"] = zipf
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=500_000)
    assert (zipf <= 512).mean() > 0.5
    for q in (msphs(2), msbs(2, key_type=FP64), msphs(3)):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)
    # a blind sample: the uniform sizes again -- the skewed inputs overflow, raise the flag ... and with the hook, the error
    from hdk_amd._lib import HdkHipError
    monkeypatch.setenv("HDK_HIP_BHM_PART_SAMPLE_STRIDE", "100000")
    with pytest.raises(HdkHipError):
        _run(oracle, gpu_executor_factory, st, msphs(3), kernel=PART)
    monkeypatch.delenv("HDK_HIP_BHM_FLAG_IS_ERROR")
    _run(oracle, gpu_executor_factory, st, msphs(3), kernel=PART)  # (the armed fallback: the oracle's result)


def test_float_group_keys(oracle, gpu_executor_factory):
    """FLOAT group keys (round 6): cast(<integer> AS FLOAT) as MSBS001-005 write it, and a plain FLOAT column.  The key word is the
    value widened to double, the NULL key the FLOAT sentinel widened (castToTypeIn(group_key, 64), QE/IRCodegen.cpp:1219-1221);
    the result column reads back as floats with None for the NULL group, and every group equals numpy's."""
    from hdk_amd.ir import FP32
    rng = np.random.default_rng(41)
    n = 300_000
    cols = syn_table(rng, n, ("x10", "x100", "x1k"), null_frac=0.02)
    f = rng.integers(-20, 20, n).astype(np.float32) * np.float32(0.5)
    f[rng.random(n) < 0.03] = np.frombuffer(np.uint32(A.NULL_FLOAT_BITS).tobytes(), dtype=np.float32)[0]
    cols["f"] = f
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=100_000)
    cp, res = _run(oracle, gpu_executor_factory, st, msbs(1))
    out = res.to_columns()
    x = cols["x1k"]
    assert sum(1 for k in out["k"] if k is None) == 1 and len(out["k"]) == len(np.unique(x[x != A.NULL_INT])) + 1
    i = out["k"].index(7.0)
    m = x == 7
    x100 = cols["x100"][m]
    assert out["c"][i] == int(m.sum()) and out["s100"][i] == int(x100[x100 != A.NULL_INT].astype(np.int64).sum())
    inull = out["k"].index(None)
    assert out["c"][inull] == int((x == A.NULL_INT).sum())
    # a FLOAT column as the key: the general open-addressing kernels (its values are not a dense integer range)
    q = QueryUnit("syn", groupby=[ColRef("f")], targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("sum", ColRef("x10"), "s")])
    cpf, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cpf.key_types[0] == FP32.with_nullable(True) or cpf.key_types[0].size == 4
    resf = gpu_executor_factory(st).execute(cpf)
    _check_rows(cpf, resf.buffer, want)
    outf = resf.to_columns()
    live = f[f != np.frombuffer(np.uint32(A.NULL_FLOAT_BITS).tobytes(), dtype=np.float32)[0]]
    assert sorted(k for k in outf["k"] if k is not None) == sorted(float(v) for v in np.unique(live))
    assert sum(1 for k in outf["k"] if k is None) == 1


# ---- BIGINT columns inside 32 bits, and filtered plans ---------------------------------------------------------------------------
def _bigint_table(rng, n, null_frac, fragment_size):
    cols = {}
    for name, v in syn_table(rng, n, ("x10", "y10", "z10", "x100", "x1k", "x10k"), null_frac=0.0).items():
        v = v.astype(np.int64)
        if null_frac:
            v[rng.random(n) < null_frac] = A.NULL_BIGINT
        cols[name] = v
    st = ArrowStorage()
    st.import_numpy("syn", cols, fragment_size=fragment_size)
    return st, cols


@pytest.mark.parametrize("null_frac", [0.0, 0.04])
def test_bigint_columns_whose_statistics_fit_32_bits(oracle, gpu_executor_factory, monkeypatch, null_frac):
    """8-byte key and argument columns (what an Arrow table of int64 gives every integer column): narrowed in registers, the
    NULL sentinel INT64_MIN mapped onto INT32_MIN; the same kernels at twice the bytes per row.  Compile-time shapes for the two
    commonest forms, the run-time form for the rest; the two-pass form beyond LDS."""
    rng = np.random.default_rng(51)
    st, _ = _bigint_table(rng, 700_003, null_frac, 233_337)
    y, x = ColRef("y10"), ColRef("x10")
    # (10 000 groups: BH004's size class -- with 1 000 the one-argument packed kernels take it, 8-byte columns included)
    five = QueryUnit("syn", groupby=[ColRef("x10k")], targets=[KeyRef(0, "k")] + [Agg(kd, y, kd) for kd in ("count", "sum", "max", "min", "avg")])
    # (open addressing: a perfect-hash plan of sums alone has had its own kernel since round 3, hdk_scan_agg_keys_values)
    two_sums = QueryUnit("syn", groupby=[Cast(ColRef("x100"), FP64)], targets=[KeyRef(0, "k"), Agg("sum", x, "sx"), Agg("avg", y, "ay"), Agg("count", None, "n")])
    # (with a NULL key the 10 001 entries no longer fit one CU's LDS, and 700 K rows are too few for two passes: whatever it takes)
    _run(oracle, gpu_executor_factory, st, five, kernel=None if null_frac else BHM)
    for q in (two_sums, msphs(1), phm(1), phm(2), msbs(1, key_type=FP64)):
        _run(oracle, gpu_executor_factory, st, q)
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    for q in (msphs(2), msbs(2, key_type=FP64)):
        _run(oracle, gpu_executor_factory, st, q, kernel=PART)


def test_bigint_value_outside_32_bits_takes_the_armed_fallback(oracle, gpu_executor_factory, monkeypatch):
    """The statistics say the column fits 32 bits; a fragment that holds 2^40 + 7 (low word: an in-range 7) or INT32_MIN itself
    (the narrowed NULL) must not be aggregated as if it did: the flag is raised, the armed kernel gives the oracle's result."""
    rng = np.random.default_rng(52)
    for bad, name in (((1 << 40) + 7, "x10"), (-(1 << 31), "x10"), ((1 << 33) + 5, "x1k")):
        st, _ = _bigint_table(rng, 300_000, 0.01, 100_000)
        st.get("syn").columns[name].fragments[2][4321] = bad
        q = QueryUnit("syn", groupby=[Cast(ColRef("x1k"), FP64)], targets=[KeyRef(0, "k"), Agg("sum", ColRef("x10"), "s"), Agg("max", ColRef("y10"), "m")])
        _run(oracle, gpu_executor_factory, st, q)
        if name != "x1k":  # (a perfect-hash KEY outside its range is an error in every form, the reference's included: not this test's subject)
            _run(oracle, gpu_executor_factory, st, msphs(1))
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    st, _ = _bigint_table(rng, 300_000, 0.0, 100_000)
    st.get("syn").columns["x100"].fragments[0][17] = (1 << 35) + 50
    _run(oracle, gpu_executor_factory, st, msphs(2), kernel=PART)


@pytest.mark.parametrize("wide", [False, True])
def test_filtered_multi_argument_group_bys(oracle, gpu_executor_factory, monkeypatch, wide):
    """Plain filters (column cmp literal; conjunctions; AND / OR / NOT programs with the reference's three-valued logic) in front of
    the on-chip tables: a row that fails takes no part -- not in COUNT(*), not in any argument; a NULL filter operand fails it."""
    from hdk_amd.ir import And, Cmp, Lit, Not, Or
    rng = np.random.default_rng(53)
    n = 600_011
    if wide:
        st, cols = _bigint_table(rng, n, 0.03, 200_003)
    else:
        st = ArrowStorage()
        cols = syn_table(rng, n, ("x10", "y10", "z10", "x100", "x1k", "x10k"), null_frac=0.03)
        st.import_numpy("syn", cols, fragment_size=200_003)
    X10, Y10, X100, X1K = ColRef("x10"), ColRef("y10"), ColRef("x100"), ColRef("x1k")
    filters = (
        [Cmp(X10, "<", Lit(7))],
        [Cmp(X10, ">=", Lit(3)), Cmp(X100, "<>", Lit(50))],
        [Or(Cmp(Y10, "=", Lit(2)), Cmp(X100, ">", Lit(90)))],
        [Not(Or(Cmp(X10, "<", Lit(4)), Cmp(X1K, ">", Lit(500))))],
        [Or(And(Cmp(X10, ">=", Lit(2)), Cmp(X10, "<=", Lit(5))), Not(Cmp(Y10, "<", Lit(9))))],
        [Cmp(X10, ">", Lit(100))],  # no row passes
    )
    for quals in filters:
        for q in (msphs(1), phm(2), msbs(1, key_type=FP64)):
            cp, res = _run(oracle, gpu_executor_factory, st, dataclasses.replace(q, quals=quals))
    # numpy's answer for one of them, to pin the oracle's filter semantics here too
    q = QueryUnit("syn", quals=[Cmp(X10, "<", Lit(7))], groupby=[Cast(X100, FP64)],
                  targets=[KeyRef(0, "k"), Agg("count", None, "n"), Agg("sum", Y10, "s"), Agg("max", X10, "m")])
    cp, res = _run(oracle, gpu_executor_factory, st, q)
    null = A.NULL_BIGINT if wide else A.NULL_INT
    keep = (cols["x10"] != null) & (cols["x10"] < 7)
    out = res.to_columns()
    assert sum(out["n"]) == int(keep.sum())
    y = cols["y10"][keep]
    assert sum(v for v in out["s"] if v is not None) == int(y[y != null].astype(np.int64).sum())
    assert max(v for v in out["m"] if v is not None) == 6
    monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    for quals in filters[:4]:
        _run(oracle, gpu_executor_factory, st, dataclasses.replace(msphs(2), quals=quals), kernel=PART)


def _random_case(rng):
    """A random table of INT or BIGINT columns (random ranges, some with NULLs), sometimes a plain filter, a random group-by over 1-3 of them (perfect hash, or one key
    cast to double / float: open addressing) and 1-6 aggregates over up to three columns, plain or `column op literal`."""
    from hdk_amd.ir import FP32
    n = int(rng.integers(50_000, 400_000))
    ncols = 6
    cols, spans = {}, {}
    dtype, null = (np.int64, A.NULL_BIGINT) if rng.random() < 0.3 else (np.int32, A.NULL_INT)  # (one width for the whole table)
    for c in range(ncols):
        lo = int(rng.integers(-50, 50))
        span = int(rng.choice([3, 10, 40, 100, 1000, 5000, 40_000]))
        v = rng.integers(lo, lo + span, n).astype(dtype)
        if rng.random() < 0.4:
            v[rng.random(n) < rng.choice([0.001, 0.05, 0.5])] = null
        cols[f"c{c}"] = v
        spans[f"c{c}"] = span
    names = list(cols)
    rng.shuffle(names)
    form = rng.choice(["perfect", "double", "float"])
    nkeys = int(rng.integers(1, 4)) if form == "perfect" else 1
    keys = names[:nkeys]
    while form == "perfect" and nkeys > 1 and np.prod([spans[k] + 2 for k in keys]) > 300_000:  # (the planner's multi-column limit is 1e6)
        nkeys -= 1
        keys = names[:nkeys]
    args = names[nkeys:nkeys + int(rng.integers(1, 4))]
    groupby = [ColRef(k) for k in keys] if form == "perfect" else [Cast(ColRef(keys[0]), FP64 if form == "double" else FP32)]
    targets = [KeyRef(i, f"k{i}") for i in range(nkeys)]
    if rng.random() < 0.6:
        targets.append(Agg("count", None, "n"))
    for i in range(int(rng.integers(1, 8 - len(targets) + 1))):
        a = ColRef(str(rng.choice(args)))
        r = rng.random()
        expr = a if r < 0.6 else (a + int(rng.integers(1, 9)) if r < 0.75 else (a - int(rng.integers(1, 9)) if r < 0.9 else a * int(rng.integers(2, 5))))
        targets.append(Agg(str(rng.choice(["count", "sum", "min", "max", "avg"])), expr, f"t{i}"))
    quals = []
    if rng.random() < 0.35:  # a plain filter, a conjunction of two, or an OR / NOT program over any of the columns
        from hdk_amd.ir import Cmp, Lit, Not, Or

        def leaf():
            c = str(rng.choice(list(cols)))
            live = cols[c][cols[c] != null]
            return Cmp(ColRef(c), str(rng.choice(["<", "<=", ">", ">=", "=", "<>"])), Lit(int(rng.choice(live))))
        form_f = rng.random()
        quals = [leaf()] if form_f < 0.4 else ([leaf(), leaf()] if form_f < 0.6 else ([Or(leaf(), leaf())] if form_f < 0.8 else [Not(Or(leaf(), leaf()))]))
    st = ArrowStorage()
    st.import_numpy("t", cols, fragment_size=int(rng.integers(n // 5 + 1, n + 1)))
    return st, QueryUnit("t", quals=quals, groupby=groupby, targets=targets, output_columnar=bool(rng.random() < 0.25))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["default", "dynamic", "two_pass"])
def test_random_multi_argument_group_bys(oracle, gpu_executor_factory, monkeypatch, mode):
    """Differential test over random shapes of the multi-argument group-by's domain, on the compile-time shapes where they apply,
    on the run-time descriptor form alone (HDK_HIP_BHM_DYNAMIC) and with the two-pass form forced for every table beyond LDS
    (HDK_HIP_BH_PARTITIONS_ALWAYS): whatever kernel the launch takes, the groups are the oracle's.  HDK_FUZZ_SEEDS=a:b widens it."""
    import os
    from hdk_amd.ir import QueryMustRunOnCpu
    if mode == "dynamic":
        monkeypatch.setenv("HDK_HIP_BHM_DYNAMIC", "1")
    if mode == "two_pass":
        monkeypatch.setenv("HDK_HIP_BH_PARTITIONS_ALWAYS", "1")
    lo, hi = (int(x) for x in os.environ.get("HDK_FUZZ_SEEDS", "0:14").split(":"))
    took = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(9000 + seed)
        st, q = _random_case(rng)
        try:
            cp, want, err = run_oracle(oracle, st, q)
        except QueryMustRunOnCpu:
            continue
        if err != 0:
            continue
        step = gpu_executor_factory(st).prepare(cp)
        names = step.kernel_names()
        res = step.run()
        step.free()
        took += "bhm" in names
        if cp.plan.query_kind == A.Q_BASELINE_HASH:
            _check_rows(cp, res.buffer, want)
        else:
            assert_buffers_equal(cp, res.buffer, want)
    assert took >= (hi - lo) // 3, took  # (the generator aims at this kernel family: most cases must land on it)
