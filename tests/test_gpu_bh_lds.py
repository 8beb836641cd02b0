"""GPU parity for the open-addressing group-by kept in LDS (hdk_amd/csrc/scan_bh.h): GroupByBaselineHash plans whose table
is small -- the reference's BaselineHash benchmark shape (cast(x as double) key, five aggregates), keys without an
expression range (x % m, also behind a join), few-group tables of every layout.  Compared with the oracle as {key ->
slots}; every group must sit on the reference's probe sequence, and the global-atomics kernels (the reference's own
scheme, HDK_HIP_LAUNCH_FORCE_GLOBAL_ATOMICS) must give the identical result."""
import os

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cast, Cmp, ColRef, FP64, JoinSpec, KeyRef, Lit, Or, QueryUnit
from hdk_amd.storage import ArrowStorage

from test_gpu_baseline import _assert_reference_placement, _check_rows, _rows
from util import run_oracle

pytestmark = pytest.mark.gpu

BH_KERNELS = ("hdk_scan_agg_bh_vec", "hdk_scan_agg_bh_direct", "hdk_scan_agg_bh_packed", "hdk_scan_agg_bh_dense")


def _one_pass_kernel(dense, monkeypatch):
    """the one-pass packed kernel in its two forms: entries by key - min (the key column's statistics are narrow: all of
    these tests' tables) or, with the switch, by 32-bit tags as for sparse keys"""
    if not dense:
        monkeypatch.setenv("HDK_HIP_NO_BH_DENSE", "1")
    return "hdk_scan_agg_bh_dense" if dense else "hdk_scan_agg_bh_packed"


def _bh_table(n, seed, nulls=True):
    rng = np.random.default_rng(seed)
    cols = {"x10": rng.integers(1, 11, n).astype(np.int32), "x100": rng.integers(1, 101, n).astype(np.int32),
            "x1k": rng.integers(1, 1001, n).astype(np.int32), "y10": rng.integers(1, 11, n).astype(np.int32),
            "k64": rng.integers(0, 300, n, dtype=np.int64) * 3_000_000_019 - 2**40,
            "v": rng.integers(-2**31, 2**31, n, dtype=np.int64), "d": rng.normal(size=n),
            "f": rng.normal(size=n).astype(np.float32), "k2": rng.integers(0, 3, n).astype(np.int32),
            "k32": (rng.integers(0, 90, n) * 7 - 100).astype(np.int32)}
    if nulls:
        cols["y10"][rng.random(n) < 0.03] = A.NULL_INT
        cols["x10"][rng.random(n) < 0.01] = A.NULL_INT
        cols["v"][rng.random(n) < 0.05] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("t", cols, fragment_size=n // 3 + 1)
    return st


def _bh_query(xcol, **kw):
    y = ColRef("y10")
    return QueryUnit("t", groupby=[Cast(ColRef(xcol), FP64)],
                     targets=[KeyRef(0, "key0"), Agg("count", y, "c"), Agg("sum", y, "s"), Agg("max", y, "mx"), Agg("min", y, "mn"),
                              Agg("avg", y, "a")], **kw)


def _run_and_check(oracle, ex, st, q, expect_kernel=BH_KERNELS, placement=True):
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.query_kind == A.Q_BASELINE_HASH
    step = ex.prepare(cp)
    names = step.kernel_names()
    assert any(names.startswith(k) for k in expect_kernel), names
    res = step.run()
    _check_rows(cp, res.buffer, want)
    if placement and not cp.plan.output_columnar and res.row_count() < cp.entry_count:
        _assert_reference_placement(oracle, cp, res.buffer)
    # a second launch into the same buffer accumulates like a second run of the row function would
    step.launch()
    twice = step.fetch()
    step.free()
    assert len(_rows(cp, twice.buffer)) == len(_rows(cp, want))
    c1, c2 = res.to_columns(), twice.to_columns()
    if "c" in c1:
        assert sorted(2 * x for x in c1["c"]) == sorted(c2["c"])
    # the reference's own scheme gives the identical groups
    ref = ex.execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS)
    _check_rows(cp, res.buffer, ref.buffer)
    return cp, res


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("xcol,groups", [("x10", 11), ("x100", 100), ("x1k", 1000)])
def test_reference_baseline_hash_benchmark_shape(oracle, gpu_executor_factory, xcol, groups, dense, monkeypatch):
    """BH001-003: cast(x as double) key, count / sum / max / min / avg of one int column (with NULLs; x10 has NULL keys)."""
    kernel = _one_pass_kernel(dense, monkeypatch)
    st = _bh_table(700_000, 11)
    ex = gpu_executor_factory(st)
    cp, res = _run_and_check(oracle, ex, st, _bh_query(xcol), expect_kernel=(kernel,))
    assert res.row_count() == groups
    keys = res.to_columns()["key0"]
    assert all(k is None or (isinstance(k, float) and k == int(k)) for k in keys)


@pytest.mark.parametrize("columnar", [False, True])
def test_few_groups_every_slot_kind(oracle, gpu_executor_factory, columnar):
    st = _bh_table(500_000, 12)
    ex = gpu_executor_factory(st)
    q = QueryUnit("t", groupby=[ColRef("k64")], force_baseline=True, baseline_entry_count=401, output_columnar=columnar,
                  targets=[KeyRef(0, "k"), Agg("sum", ColRef("v"), "s"), Agg("count", None, "c"), Agg("count", ColRef("v"), "cv"),
                           Agg("min", ColRef("d"), "mn"), Agg("max", ColRef("v"), "mx"), Agg("avg", ColRef("f"), "af"),
                           Agg("sum", ColRef("f"), "sf")])
    _run_and_check(oracle, ex, st, q)


def test_two_int32_keys_share_the_key_word(oracle, gpu_executor_factory):
    st = _bh_table(300_000, 13)
    ex = gpu_executor_factory(st)
    q = QueryUnit("t", groupby=[ColRef("k32"), ColRef("k2")], force_baseline=True, baseline_entry_count=601,
                  targets=[KeyRef(0, "a"), KeyRef(1, "b"), Agg("sum", ColRef("y10"), "s"), Agg("count", None, "c")])
    cp, res = _run_and_check(oracle, ex, st, q)
    assert cp.plan.key_width == 4 and cp.plan.key_count == 2


def test_compact_count_slots(oracle, gpu_executor_factory):
    """COUNT(*) alone by a 4-byte key: 4-byte slots (pick_target_compact_width)."""
    st = _bh_table(300_000, 14)
    ex = gpu_executor_factory(st)
    q = QueryUnit("t", groupby=[ColRef("k32")], force_baseline=True, baseline_entry_count=257,
                  targets=[KeyRef(0, "a"), Agg("count", None, "c")])
    cp, res = _run_and_check(oracle, ex, st, q)
    assert cp.plan.targets[1].slot_width == 4


def test_filters_and_expression_keys(oracle, gpu_executor_factory):
    st = _bh_table(400_000, 15)
    ex = gpu_executor_factory(st)
    for quals in ([Cmp(ColRef("y10"), "<=", Lit(7))], [Or(Cmp(ColRef("v"), "<", Lit(0)), Cmp(ColRef("x10"), "=", Lit(3)))]):
        q = QueryUnit("t", groupby=[ColRef("x100") % 37], quals=quals,
                      targets=[KeyRef(0, "g"), Agg("sum", ColRef("v"), "s"), Agg("avg", ColRef("d"), "a"), Agg("count", None, "c")])
        cp, res = _run_and_check(oracle, ex, st, q)
        assert cp.entry_count == 2 * 37


@pytest.mark.parametrize("xcol,m,signed", [("x1k", 37, False), ("k32", 11, True), ("x10", 7, False), ("k64s", 64, True)])
def test_modulo_key_on_the_dense_kernel(oracle, gpu_executor_factory, xcol, m, signed, monkeypatch):
    """`GROUP BY col % m` with the benchmark's aggregates: the planner has no range for a modulo (an open-addressing layout of
    2 x NDV-bound entries, or the 16 384-entry default guess), the kernel does -- (-m, m), or [0, m) when the statistics say the
    column is not negative -- and runs the plan on a dense LDS table over it (hdk_scan_agg_bh_dense, key_form 2).  Negative
    values (C's truncating remainder), NULL keys (x10: the NULL group), an 8-byte column inside 32 bits, a filter in front; the
    interpreter and the reference's own scheme give the same groups."""
    st = _bh_table(500_000, 41)
    t = st.get("t")
    # an 8-byte signed key column inside 32 bits
    import numpy as _np
    k64s = _np.concatenate(t.columns["k32"].fragments).astype(_np.int64) * 3 - 50
    st2 = ArrowStorage()
    cols = {k: _np.concatenate(c.fragments) for k, c in t.columns.items()}
    cols["k64s"] = k64s
    st2.import_numpy("t", cols, fragment_size=500_000 // 3 + 1)
    ex = gpu_executor_factory(st2)
    y = ColRef("y10")
    for quals in ([], [Cmp(ColRef("y10"), "<=", Lit(7))]):
        q = QueryUnit("t", groupby=[ColRef(xcol) % m], quals=quals,
                      targets=[KeyRef(0, "g"), Agg("count", y, "c"), Agg("sum", y, "s"), Agg("max", y, "mx"), Agg("min", y, "mn"), Agg("avg", y, "a")])
        cp, res = _run_and_check(oracle, ex, st2, q, expect_kernel=("hdk_scan_agg_bh_dense,",))
        groups = res.row_count()
        assert groups <= (2 * m - 1 if signed else m) + 1
    monkeypatch.setenv("HDK_HIP_NO_BH_MOD_KEYS", "1")
    q = QueryUnit("t", groupby=[ColRef(xcol) % m], targets=[KeyRef(0, "g"), Agg("count", y, "c"), Agg("sum", y, "s")])
    _run_and_check(oracle, ex, st2, q, expect_kernel=("hdk_scan_agg_bh_vec",))


@pytest.mark.parametrize("groups", [1_500, 3_000, 4_000])
def test_dense_tables_of_a_few_thousand_groups_stay_on_chip(oracle, gpu_executor_factory, groups):
    """1.5 K groups: 256-thread blocks; 3 K and 4 K (2 x entries beyond what tags could hold on chip): ONE 512-thread block
    per CU with a dense table of up to 4 096 entries -- still one pass."""
    rng = np.random.default_rng(groups)
    n = 1_200_000
    x = rng.integers(1, groups + 1, n).astype(np.int32)
    x[rng.random(n) < 0.002] = A.NULL_INT
    y = rng.integers(1, 11, n).astype(np.int32)
    y[rng.random(n) < 0.02] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", {"x": x, "y10": y}, fragment_size=n // 3 + 5)
    ex = gpu_executor_factory(st)
    cp, res = _run_and_check(oracle, ex, st, _bh_query("x"), expect_kernel=("hdk_scan_agg_bh_dense_plain,",))
    assert res.row_count() == groups + 1


def test_modulo_key_behind_a_join(oracle, gpu_executor_factory):
    """SURVEY 8(d)'s C3 variant as written: fact JOIN dim, GROUP BY dim.dval % 64, SUM(fact.val)."""
    rng = np.random.default_rng(16)
    nd, n = 20_000, 600_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "dval": rng.integers(0, 10**6, nd).astype(np.int64)})
    st.import_numpy("fact", {"fk": rng.integers(0, nd + 50, n).astype(np.int64), "val": rng.integers(-2**31, 2**31, n).astype(np.int64)},
                    fragment_size=190_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") % 64],
                  targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    ex = gpu_executor_factory(st)
    for fuse in (True, False):
        ex.fuse_join_tables = fuse
        cp, res = _run_and_check(oracle, ex, st, q, expect_kernel=("hdk_scan_agg_bh_vec_join", "hdk_join_"))
        assert res.row_count() == 64


def test_more_groups_than_entries_is_out_of_slots(oracle, gpu_executor_factory):
    from hdk_amd._lib import HdkHipError
    st = _bh_table(100_000, 17)
    q = QueryUnit("t", groupby=[ColRef("x1k")], force_baseline=True, baseline_entry_count=300,
                  targets=[KeyRef(0), Agg("sum", ColRef("v"))])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == A.ERR_OUT_OF_SLOTS
    ex = gpu_executor_factory(st)
    step = ex.prepare(cp)
    assert step.kernel_names().startswith(BH_KERNELS), step.kernel_names()
    step.free()
    with pytest.raises(HdkHipError) as ei:
        ex.execute(cp)
    assert ei.value.code == A.ERR_OUT_OF_SLOTS


_SOAK = os.environ.get("HDK_FUZZ_SEEDS", "")
_SOAK_SEEDS = list(range(*map(int, _SOAK.split(":")))) if _SOAK else []


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [1, 2, 3] + _SOAK_SEEDS)
def test_small_open_addressing_tables_random_shapes(oracle, gpu_executor_factory, seed):
    """Seeded random plans with small open-addressing tables: 1-2 keys of either width, plain / double-cast / modulo keys,
    1-4 aggregates over int64 / int32 / double / float columns with NULLs, optional filters, skewed keys, load factors
    up to more groups than entries (ERR_OUT_OF_SLOTS on both sides), ragged fragments, row-wise and columnar."""
    from hdk_amd._lib import HdkHipError
    rng = np.random.default_rng(7700 + seed)
    n = int(rng.choice([900, 40_000, 350_000, 1_500_000]))
    ndv = int(rng.choice([3, 40, 300, 900]))
    base = rng.integers(0, ndv, n, dtype=np.int64)
    if rng.random() < 0.4:
        base[rng.random(n) < 0.5] = 1
    cols = {"k64": base * 3_000_000_019 - 2**40, "k32": (base * 7 - 1000).astype(np.int32), "k2": rng.integers(0, 3, n).astype(np.int32),
            "v": rng.integers(-2**31, 2**31, n, dtype=np.int64), "i32": rng.integers(-1000, 1000, n).astype(np.int32),
            "d": rng.normal(size=n), "f": rng.normal(size=n).astype(np.float32)}
    cols["v"][rng.random(n) < 0.05] = A.NULL_BIGINT
    cols["i32"][rng.random(n) < 0.05] = A.NULL_INT
    cols["k32"][rng.random(n) < 0.01] = A.NULL_INT
    st = ArrowStorage()
    st.import_numpy("t", cols, fragment_size=int(rng.integers(n // 5 + 1, n + 2)))
    ex = gpu_executor_factory(st)
    for qi in range(5):
        form = str(rng.choice(["k64", "k32", "k32+k2", "double", "mod"]))
        if form == "double":
            groupby, groups = [Cast(ColRef("k32"), FP64)], ndv + 1
        elif form == "mod":
            m = int(rng.integers(2, 50))
            groupby, groups = [ColRef("i32") % m], 2 * m
        elif form == "k32+k2":
            groupby, groups = [ColRef("k32"), ColRef("k2")], 3 * (ndv + 1)
        else:
            groupby, groups = [ColRef(form)], ndv + 1
        load = float(rng.choice([0.3, 0.5, 0.9, 1.4]))
        entries = max(int(groups / load) | 1, 5)
        targets = [KeyRef(i, f"k{i}") for i in range(len(groupby))]
        for ti in range(int(rng.integers(1, 5))):
            kind = str(rng.choice(["sum", "count", "min", "max", "avg"]))
            arg = None if (kind == "count" and rng.random() < 0.5) else ColRef(str(rng.choice(["v", "i32", "d", "f"])))
            targets.append(Agg(kind, arg, f"t{ti}"))
        quals = [Cmp(ColRef("i32"), "<=", Lit(int(rng.integers(-200, 900))))] if rng.random() < 0.4 else []
        columnar = bool(rng.random() < 0.25) and len(groupby) == 1
        q = QueryUnit("t", groupby=groupby, quals=quals, force_baseline=True, baseline_entry_count=entries, targets=targets,
                      output_columnar=columnar)
        cp, want, err = run_oracle(oracle, st, q)
        what = (seed, qi, n, ndv, form, entries, columnar, q)
        if err:
            assert err == A.ERR_OUT_OF_SLOTS, what
            with pytest.raises(HdkHipError) as ei:
                ex.execute(cp)
            assert ei.value.code == A.ERR_OUT_OF_SLOTS, what
            continue
        step = ex.prepare(cp)
        names = step.kernel_names()
        res = step.run()
        step.free()
        try:
            if entries <= 300:  # (<= 512 LDS entries of at most 14 words: always within the 64 KiB the kernels take)
                assert names.startswith(BH_KERNELS), names
            # (a float SUM is added row by row in float by the reference: with zero-mean values the partial sums wander like
            # sqrt(k) and every addition rounds at their magnitude, so the noise of a group's sum grows LINEARLY with its rows --
            # eps32 x rows / 2.8: 0.016 for the 750 K rows of seed 3154's hot key, where 0.028 was seen)
            _check_rows(cp, res.buffer, want, float32_atol=2e-3 * max(1.0, n / 40_000))
            if res.row_count() < cp.entry_count and not columnar:
                _assert_reference_placement(oracle, cp, res.buffer)
        except AssertionError as e:
            raise AssertionError(f"{what}: {e}") from e


# ---- tables beyond LDS: the 256-bin passes (hdk_bh_scatter + hdk_bh_aggregate) ---------------------------------------------
def _mid_table(n, groups, seed, hot=0.0):
    rng = np.random.default_rng(seed)
    x = rng.integers(1, groups + 1, n).astype(np.int32)
    if hot:
        x[rng.random(n) < hot] = 7
    y = rng.integers(1, 11, n).astype(np.int32)
    y[rng.random(n) < 0.02] = A.NULL_INT
    x64 = x.astype(np.int64) * 3 - 100
    x[rng.random(n) < 0.002] = A.NULL_INT  # NULL keys: a group of their own
    st = ArrowStorage()
    st.import_numpy("t", {"x": x, "y10": y, "x64": x64, "w": rng.integers(-5000, 5000, n).astype(np.int64)},
                    fragment_size=n // 3 + 7)
    return st


def _two_pass_kernels(bins, monkeypatch):
    """tables beyond LDS: 256 bins by key RANGE with 4-byte tuples when the key column's statistics are dense enough (all of
    these tests' tables), or -- with the switch -- by the key's hash with 8-byte tuples, as for sparse keys"""
    # (round 6: tables up to ~12 000 groups fit LDS at the multi-argument kernel's 12 bytes an entry and run in ONE pass,
    # tests/test_gpu_bhm.py; these tests are about the two-pass forms, which still take what is larger)
    monkeypatch.setenv("HDK_HIP_NO_BHM", "1")
    if bins == "hash":
        monkeypatch.setenv("HDK_HIP_NO_BH_DENSE_PARTITIONS", "1")
    return "hdk_bh_dscatter,hdk_bh_daggregate" if bins == "range" else "hdk_bh_scatter,hdk_bh_aggregate"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("bins", ["range", "hash"])
@pytest.mark.parametrize("groups,hot", [(9_000, 0.0), (60_000, 0.0), (9_000, 0.6)])
def test_mid_sized_tables_by_hash_bins(oracle, gpu_executor_factory, groups, hot, bins, monkeypatch):
    """BH004 / BH005's size class: 10 K - 100 K groups behind a double key.  A hot key overflows its bin's slab: those rows
    take the reference's own scheme inside the scatter pass."""
    st = _mid_table(4_400_000, groups, 21, hot)
    ex = gpu_executor_factory(st)
    q = _bh_query("x")
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.entry_count in (2 * groups, 2 * groups + 2)
    kernels = _two_pass_kernels(bins, monkeypatch)
    step = ex.prepare(cp)
    assert step.kernel_names() == kernels, step.kernel_names()
    res = step.run()
    step.free()
    _check_rows(cp, res.buffer, want)
    _assert_reference_placement(oracle, cp, res.buffer)
    _check_rows(cp, ex.execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS).buffer, want)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("bins", ["range", "hash"])
def test_mid_sized_table_under_an_or_filter(oracle, gpu_executor_factory, bins, monkeypatch):
    """`WHERE y10 <= 3 OR w < 0` in front of the two-pass forms: the scatter passes run the AND / OR / NOT program themselves
    (plain_quals.h; the hash-bin scatter in its own instantiation).  NULLs in y10: a NULL leaf under OR passes when the other
    side is TRUE."""
    st = _mid_table(4_300_000, 20_000, 24)
    ex = gpu_executor_factory(st)
    q = _bh_query("x", quals=[Or(Cmp(ColRef("y10"), "<=", Lit(3)), Cmp(ColRef("w"), "<", Lit(0)))])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.num_filter_ops > 0
    kernels = _two_pass_kernels(bins, monkeypatch)
    step = ex.prepare(cp)
    assert step.kernel_names() == kernels, step.kernel_names()
    res = step.run()
    step.free()
    _check_rows(cp, res.buffer, want)
    _assert_reference_placement(oracle, cp, res.buffer)
    _check_rows(cp, ex.execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("bins", ["range", "hash"])
def test_mid_sized_table_int64_columns_filters_and_stale_statistics(oracle, gpu_executor_factory, bins, monkeypatch):
    """8-byte key and argument columns that fit 32 bits by their statistics, a filter, and -- second run -- statistics that
    do NOT hold (narrowed by hand in the launch's plan copy): the rows outside them take the exact path, same result."""
    st = _mid_table(4_300_000, 30_000, 22)
    ex = gpu_executor_factory(st)
    w = ColRef("w")
    q = QueryUnit("t", groupby=[ColRef("x64")], quals=[Cmp(ColRef("y10"), "<=", Lit(8))], force_baseline=True, baseline_entry_count=70_001,
                  targets=[KeyRef(0, "k"), Agg("sum", w, "s"), Agg("min", w, "mn"), Agg("max", w, "mx"), Agg("count", None, "c")])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    kernels = _two_pass_kernels(bins, monkeypatch)
    for stale in (False, True):
        step = ex.prepare(cp)
        assert step.kernel_names() == kernels, step.kernel_names()
        if stale:
            for ci in range(step.plan.num_cols):
                c = step.plan.cols[ci]
                if c.has_stats and c.width == 8:
                    c.min_val, c.max_val = c.min_val // 2, c.max_val // 2
        res = step.run()
        step.free()
        _check_rows(cp, res.buffer, want)
        _assert_reference_placement(oracle, cp, res.buffer)


@pytest.mark.parametrize("dense", [True, False])
def test_small_table_with_stale_statistics(oracle, gpu_executor_factory, dense, monkeypatch):
    """The one-pass packed kernel with statistics that do not hold: rows outside them bypass the packed sum (and, in the dense
    form, keys outside them the table)."""
    kernel = _one_pass_kernel(dense, monkeypatch)
    st = _mid_table(900_000, 300, 23)
    ex = gpu_executor_factory(st)
    w = ColRef("w")
    q = QueryUnit("t", groupby=[ColRef("x64")], force_baseline=True, baseline_entry_count=701,
                  targets=[KeyRef(0, "k"), Agg("sum", w, "s"), Agg("min", w, "mn"), Agg("max", w, "mx"), Agg("avg", w, "a")])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    step = ex.prepare(cp)
    assert step.kernel_names().startswith(kernel), step.kernel_names()
    for ci in range(step.plan.num_cols):
        c = step.plan.cols[ci]
        if c.has_stats and c.width == 8:
            c.min_val, c.max_val = c.min_val // 2, c.max_val // 2
    res = step.run()
    step.free()
    _check_rows(cp, res.buffer, want)
    _assert_reference_placement(oracle, cp, res.buffer)


# ---- GroupByPerfectHash plans of the same shape through the same kernels (the reference's PerfectHashSingleCol queries) -------
def _phs_query(xcol, table="t", **kw):
    y = ColRef("y10")
    return QueryUnit(table, groupby=[ColRef(xcol)],
                     targets=[KeyRef(0, "k"), Agg("count", y, "c"), Agg("sum", y, "s"), Agg("max", y, "mx"), Agg("min", y, "mn"),
                              Agg("avg", y, "a")], **kw)


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("xcol", ["x10", "x100", "x1k"])
@pytest.mark.parametrize("columnar", [False, True])
def test_perfect_hash_benchmark_shape_on_the_packed_kernel(oracle, gpu_executor_factory, xcol, columnar, dense, monkeypatch):
    """PHS001-003: count / sum / max / min / avg of one int column by an int key with a perfect-hash layout (NULL keys and
    NULL arguments present): the buffer must equal the oracle's bit for bit -- entries, stored keys, untouched slots."""
    from util import assert_buffers_equal
    st = _bh_table(600_000, 31)
    ex = gpu_executor_factory(st)
    q = _phs_query(xcol, output_columnar=columnar)
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.query_kind == A.Q_PERFECT_HASH
    kernel = _one_pass_kernel(dense, monkeypatch)
    step = ex.prepare(cp)
    assert step.kernel_names().startswith(kernel), step.kernel_names()
    res = step.run()
    assert_buffers_equal(cp, res.buffer, want)
    step.launch()  # a second launch accumulates
    twice = step.fetch()
    step.free()
    c1, c2 = res.to_columns(), twice.to_columns()
    assert [2 * x for x in c1["c"]] == c2["c"] and c1["k"] == c2["k"]
    # the interpreter and the global-atomics kernels give the identical buffer
    assert_buffers_equal(cp, ex.execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)
    assert_buffers_equal(cp, ex.execute(cp, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS).buffer, want)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("bins", ["range", "hash"])
def test_perfect_hash_table_beyond_lds_by_hash_bins(oracle, gpu_executor_factory, bins, monkeypatch):
    """PHS004's size class: 9 000 groups -- the table does not fit LDS, the rows go through the 256-bin passes and the fold
    addresses the perfect-hash table by key - min."""
    from util import assert_buffers_equal
    st = _mid_table(4_400_000, 9_000, 32)
    ex = gpu_executor_factory(st)
    q = _phs_query("x")
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.query_kind == A.Q_PERFECT_HASH
    kernels = _two_pass_kernels(bins, monkeypatch)
    step = ex.prepare(cp)
    assert step.kernel_names() == kernels, step.kernel_names()
    res = step.run()
    step.free()
    assert_buffers_equal(cp, res.buffer, want)


@pytest.mark.parametrize("signed,nulls", [(False, False), (True, True)])
def test_modulo_key_behind_the_sliced_join(oracle, gpu_executor_factory, signed, nulls):
    """C3gm on the sliced LDS path (forced on a small table: HDK_HIP_LAUNCH_CLUSTER_PROBES): an internal dense group table
    over the key's own range (-m, m), folded into the open-addressing table by hdk_bh_fold_dense; negative payloads (C's
    truncating remainder) and NULL payloads included."""
    rng = np.random.default_rng(41 + signed)
    nd, n = 30_000, 700_000
    dval = rng.integers(-10**6 if signed else 0, 10**6, nd).astype(np.int64)
    if nulls:
        dval[rng.random(nd) < 0.02] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "dval": dval})
    st.import_numpy("fact", {"fk": rng.integers(0, nd + 30, n).astype(np.int64), "val": rng.integers(-2**31, 2**31, n).astype(np.int64)},
                    fragment_size=230_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], groupby=[ColRef("dval", "dim") % 64],
                  targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0 and cp.plan.query_kind == A.Q_BASELINE_HASH
    ex = gpu_executor_factory(st)
    step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
    names = step.kernel_names()
    assert names.startswith("hdk_join_order_probe,hdk_join_scatter_slices") and names.endswith("hdk_bh_fold_dense"), names
    res = step.run()
    step.free()
    _check_rows(cp, res.buffer, want)
    _assert_reference_placement(oracle, cp, res.buffer)
    assert res.row_count() == (127 if signed else 64) + (1 if nulls else 0)
