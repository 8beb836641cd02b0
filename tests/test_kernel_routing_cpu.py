"""Which kernels a plan takes -- pinned on a CPU-only box.

Kernel routing (the `match_*` functions of hdk_amd/csrc) is host arithmetic over the plan, the kernel options and a few
device numbers; `hdk_hip_describe_launch` answers it for an assumed MI355X (`HDK_HIP_DEVICE_ASSUMED_MI355X`) without touching a
device.  The GPU suite asserts the same names on the device (and `test_gpu_abi_negative.py` that the assumed numbers are the
device's); here the driver's per-round CPU run sees a routing regression -- a benchmark shape falling back to an
interpreter or to global atomics -- without waiting for a GPU.  Tables are small (the statistics and NDV bounds are what
routing reads); `total_rows` is passed as the BASELINE size.
"""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import lib, sync_switches
from hdk_amd.ir import Agg, Cast, Cmp, ColRef, FP64, KeyRef, Lit, Or, QueryUnit
from hdk_amd.plan import compile_query
from hdk_amd.storage import ArrowStorage

ASSUMED_MI355X = -355  # include/hdk_hip.h: HDK_HIP_DEVICE_ASSUMED_MI355X
BIG = 1_000_000_000


def _names(cp, total_rows=BIG, flags=0):
    sync_switches()
    ko = A.KernelOptions(0, 0, 0, flags, total_rows, 0, 0)
    out = C.create_string_buffer(256)
    st = lib().hdk_hip_describe_launch(C.byref(cp.plan), C.byref(ko), ASSUMED_MI355X, out, 256)
    assert st == 0, lib().hdk_hip_last_error()
    return out.value.decode()


@pytest.fixture(scope="module")
def storage():
    rng = np.random.default_rng(3)
    n = 300_000
    st = ArrowStorage()
    y = rng.integers(1, 11, n).astype(np.int32)
    y[rng.random(n) < 0.02] = A.NULL_INT
    st.import_numpy("syn", {"x10": rng.integers(1, 11, n).astype(np.int32), "x1k": rng.integers(1, 1001, n).astype(np.int32),
                            "x100k": rng.integers(1, 100_001, n).astype(np.int32), "x4k": rng.integers(1, 4001, n).astype(np.int32), "y10": y,
                            "sparse": (rng.integers(0, 90_000, n) * 7919).astype(np.int32), "d": rng.normal(size=n)},
                    fragment_size=n // 3 + 1)
    st.import_numpy("t", {"key": rng.integers(0, 64, n, dtype=np.int64), "val": rng.integers(-2**31, 2**31, n, dtype=np.int64),
                          "c": rng.integers(-50, 50, n).astype(np.int32), "k2": rng.integers(0, 7, n).astype(np.int16),
                          "wide": rng.integers(0, 2**40, n, dtype=np.int64)}, fragment_size=n // 3 + 1)
    return st


def _bh(xcol, **kw):
    y = ColRef("y10")
    return QueryUnit("syn", groupby=[Cast(ColRef(xcol), FP64)],
                     targets=[KeyRef(0, "k")] + [Agg(k, y) for k in ("count", "sum", "max", "min", "avg")], **kw)


def test_headline_and_its_neighbours(storage):
    K, V, C_ = ColRef("key"), ColRef("val"), ColRef("c")
    c2 = QueryUnit("t", groupby=[K], targets=[KeyRef(0), Agg("sum", V)])
    assert _names(compile_query(storage, c2)) == "hdk_scan_agg_direct,hdk_finalize"
    c1 = QueryUnit("t", targets=[Agg("sum", V)])
    assert _names(compile_query(storage, c1)) == "hdk_scan_agg_direct,hdk_finalize"
    for quals in ([Cmp(C_, "<", Lit(0))], [Or(Cmp(V, "<", Lit(0)), Cmp(K, "=", Lit(3)))], [Or(Cmp(C_, "<", Lit(0)), Cmp(V, ">", Lit(5)))]):
        q = QueryUnit("t", quals=quals, groupby=[K], targets=[KeyRef(0), Agg("sum", V)])
        assert _names(compile_query(storage, q)) == "hdk_scan_agg_direct,hdk_finalize", quals
    two_keys = QueryUnit("t", groupby=[K, ColRef("k2")], targets=[KeyRef(0), KeyRef(1), Agg("count", None)])
    assert _names(compile_query(storage, two_keys)) == "hdk_scan_agg_keys,hdk_finalize"
    # (the keys kernel carries no filter-program code: the interpreter)
    two_keys_or = QueryUnit("t", quals=[Or(Cmp(C_, "<", Lit(0)), Cmp(V, ">", Lit(5)))], groupby=[K, ColRef("k2")],
                            targets=[KeyRef(0), KeyRef(1), Agg("count", None)])
    assert _names(compile_query(storage, two_keys_or)).startswith("hdk_scan_agg_vec")


def test_filter_program_deeper_than_three_goes_to_the_interpreter(storage):
    """Round 5's advisor finding: the streaming evaluators keep a three-value stack; `(a AND b) OR ((a AND c) OR (b AND c))`
    needs four (three deduplicated leaves, eleven ops).  It must route to the interpreter; its left-nested form stays."""
    from hdk_amd.ir import And
    K, V, C_ = ColRef("key"), ColRef("val"), ColRef("c")
    a, b, c = Cmp(C_, "<", Lit(10)), Cmp(V, ">", Lit(0)), Cmp(K, "<>", Lit(7))
    deep = [Or(And(a, b), Or(And(a, c), And(b, c)))]
    left = [Or(Or(And(a, b), And(a, c)), And(b, c))]
    for quals, streaming in ((deep, False), (left, True)):
        q = QueryUnit("t", quals=quals, groupby=[K], targets=[KeyRef(0), Agg("sum", V)])
        cp = compile_query(storage, q)
        assert cp.plan.num_quals == 3 and cp.plan.num_filter_ops == 11
        assert (_names(cp) == "hdk_scan_agg_direct,hdk_finalize") == streaming, (quals, _names(cp))
        assert _names(cp).startswith("hdk_scan_agg_vec") != streaming
        y = ColRef("y10")
        ya, yb, yc = Cmp(y, "<=", Lit(7)), Cmp(ColRef("x10"), ">", Lit(2)), Cmp(ColRef("x1k"), "<>", Lit(5))
        yq = [Or(And(ya, yb), Or(And(ya, yc), And(yb, yc)))] if not streaming else [Or(Or(And(ya, yb), And(ya, yc)), And(yb, yc))]
        names = _names(compile_query(storage, _bh("x10", quals=yq)))
        assert names.startswith("hdk_scan_agg_bh_dense,") == streaming and ("_vec" in names) != streaming, names


def test_reference_baseline_hash_benchmark_shapes(storage, monkeypatch):
    """BH001 / BH003 / BH005 (bench.py lines) and what their switches select."""
    one_pass = "hdk_scan_agg_bh_dense_plain,hdk_bh_fold_slabs"
    assert _names(compile_query(storage, _bh("x10"))) == one_pass
    assert _names(compile_query(storage, _bh("x1k"))) == one_pass
    assert _names(compile_query(storage, _bh("x4k"))) == one_pass  # (4 096 dense entries under one 512-thread block per CU)
    # (round 6: BH005's 100 K groups on the multi-argument kernels' two-pass form -- leaner scatter, 12-byte entries: 1.14 ms per
    # 256 M rows against hdk_bh_dscatter's 1.38; its switch gives round 5's passes back)
    assert _names(compile_query(storage, _bh("x100k"))) == "hdk_bhm_scatter,hdk_bhm_aggregate,hdk_bhm_reduce_slabs,hdk_bhm_fold"
    monkeypatch.setenv("HDK_HIP_NO_BHM_PARTITIONS", "1")
    assert _names(compile_query(storage, _bh("x100k"))) == "hdk_bh_dscatter,hdk_bh_daggregate"
    monkeypatch.delenv("HDK_HIP_NO_BHM_PARTITIONS")
    # BH004's 10 000 groups: 24 bytes an entry do not fit a CU's LDS, the multi-argument kernel's 12 do -- ONE pass (round 6)
    st10k = ArrowStorage()
    rng10k = np.random.default_rng(4)
    st10k.import_numpy("syn", {"x10k": rng10k.integers(1, 10_001, 50_000).astype(np.int32), "y10": rng10k.integers(1, 11, 50_000).astype(np.int32)},
                       fragment_size=20_000)
    assert _names(compile_query(st10k, _bh("x10k"))) == "hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,hdk_bhm_fold"
    ph10k = QueryUnit("syn", groupby=[ColRef("x10k")], targets=[KeyRef(0, "k")] + [Agg(k, ColRef("y10")) for k in ("count", "sum", "max", "min", "avg")])
    assert _names(compile_query(st10k, ph10k)) == "hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,hdk_finalize"
    # filtered: the general kernels; an fp argument: the word form; a modulo key: a dense table over (-m, m) in the general
    # kernels when the aggregates are the packed shape, else the interpreter with an LDS table
    assert _names(compile_query(storage, _bh("x10", quals=[Cmp(ColRef("y10"), "<=", Lit(7))]))) == "hdk_scan_agg_bh_dense,hdk_bh_fold_slabs"
    fp_arg = QueryUnit("syn", groupby=[Cast(ColRef("x10"), FP64)], targets=[KeyRef(0), Agg("sum", ColRef("d"))])
    assert _names(compile_query(storage, fp_arg)).startswith("hdk_scan_agg_bh_direct")
    expr_key = QueryUnit("syn", groupby=[ColRef("x1k") % 37], targets=[KeyRef(0), Agg("sum", ColRef("y10"))])
    assert _names(compile_query(storage, expr_key)) == "hdk_scan_agg_bh_dense,hdk_bh_fold_slabs"
    expr_key_fp = QueryUnit("syn", groupby=[ColRef("x1k") % 37], targets=[KeyRef(0), Agg("sum", ColRef("d"))])
    assert _names(compile_query(storage, expr_key_fp)).startswith("hdk_scan_agg_bh_vec")
    # sparse keys: tags on chip, hash bins beyond
    sparse = QueryUnit("syn", groupby=[ColRef("sparse")], force_baseline=True, baseline_entry_count=180_001,
                       targets=[KeyRef(0), Agg("sum", ColRef("y10")), Agg("count", None)])
    assert _names(compile_query(storage, sparse)) == "hdk_bh_scatter,hdk_bh_aggregate"
    # the reference's own scheme when asked for
    assert _names(compile_query(storage, _bh("x10")), flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS).startswith("hdk_scan_agg_global")
    monkeypatch.setenv("HDK_HIP_NO_BH_DENSE", "1")
    assert _names(compile_query(storage, _bh("x10"))) == "hdk_scan_agg_bh_packed_plain,hdk_bh_fold_slabs"
    assert _names(compile_query(storage, _bh("x4k"))) == "hdk_bh_dscatter,hdk_bh_daggregate"  # (tags would need 8 K entries)
    monkeypatch.setenv("HDK_HIP_NO_BH_DENSE_PARTITIONS", "1")
    assert _names(compile_query(storage, _bh("x100k"))) == "hdk_bh_scatter,hdk_bh_aggregate"
    monkeypatch.setenv("HDK_HIP_NO_BH_LDS", "1")
    assert _names(compile_query(storage, _bh("x10"))).startswith("hdk_scan_agg_global")


def test_huge_open_addressing_table_takes_the_radix_passes(storage):
    """C5's shape: 8-byte key, a table of 200 M entries, 1 B rows."""
    q = QueryUnit("t", groupby=[ColRef("wide")], force_baseline=True, baseline_entry_count=200_000_000,
                  targets=[KeyRef(0), Agg("sum", ColRef("val"))])
    names = _names(compile_query(storage, q))
    assert names.startswith("hdk_part_scatter,hdk_part_scatter,hdk_part_aggregate"), names
    # a small input of the same plan: the direct global-atomics kernel
    assert _names(compile_query(storage, q), total_rows=100_000).startswith("hdk_scan_agg_baseline_direct")


def _fused(cp):
    """The plan as the executor launches it when a one-to-one join table is used in its fused [row id | payloads] form
    (hdk_amd/executor.py: _fuse_join_tables -- the plan-side half of it; no table is built here)."""
    plan = A.Plan.from_buffer_copy(cp.plan)
    cols = [ci for ci, (_t, _c, slot) in enumerate(cp.input_cols) if slot == 1]
    for k, ci in enumerate(cols):
        plan.cols[ci].kind = A.COL_DOUBLE if plan.cols[ci].kind in (A.COL_FLOAT, A.COL_DOUBLE) else A.COL_INT
        plan.cols[ci].width = 8
        plan.cols[ci].table = -1
        plan.cols[ci].buf_idx = 1 + k
    plan.joins[0].kind = A.JOIN_ONE_TO_ONE_FUSED
    plan.joins[0].fused_stride = 1 + len(cols)

    class _Cp:
        pass
    out = _Cp()
    out.plan = plan
    return out


def test_star_schema_shapes_take_the_sliced_join():
    """C3 and the shapes around it (bench.py's c3 / c3g / c3gm / c3m; bench_configs' c3f / c3x) against a 10 M-key dimension:
    key-range slices in LDS, ONE scatter level -- also for two dimension columns (packed into one LDS word) and for a group key
    from the fact table (in the tuple's spare bits)."""
    from hdk_amd.ir import JoinSpec
    rng = np.random.default_rng(4)
    nd, n = 10_000_000, 200_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": np.arange(nd, dtype=np.int64), "dval": rng.integers(0, 10**6, nd).astype(np.int64),
                            "attr": rng.integers(0, 64, nd).astype(np.int64)})
    st.import_numpy("fact", {"fk": rng.integers(0, nd, n, dtype=np.int64), "val": rng.integers(-2**31, 2**31, n, dtype=np.int64),
                             "g": rng.integers(0, 64, n, dtype=np.int64), "g32": rng.integers(0, 64, n).astype(np.int32)},
                    fragment_size=n // 2 + 1)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    V, D, At = ColRef("val"), ColRef("dval", "dim"), ColRef("attr", "dim")
    one_level = "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_vec_join,hdk_finalize"
    shapes = {
        "c3": (QueryUnit("fact", joins=j, targets=[Agg("sum", V + D)]),
               "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced,hdk_join_agg_direct,hdk_finalize"),
        "c3g": (QueryUnit("fact", joins=j, groupby=[D / 15625], targets=[KeyRef(0), Agg("sum", V)]), one_level),
        "c3m": (QueryUnit("fact", joins=j, targets=[Agg("sum", V), Agg("count", None), Agg("max", D)]), one_level),
        "c3gm": (QueryUnit("fact", joins=j, groupby=[D % 64], targets=[KeyRef(0), Agg("sum", V)]),
                 "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_bh_vec_join,hdk_bh_fold_dense"),
        "c3f": (QueryUnit("fact", joins=j, quals=[Cmp(D, "<", Lit(500_000))], groupby=[At], targets=[KeyRef(0), Agg("sum", V), Agg("count", None)]),
                one_level),
        "c3x": (QueryUnit("fact", joins=j, quals=[Cmp(D, "<", Lit(500_000))], groupby=[ColRef("g32")], targets=[KeyRef(0), Agg("sum", V)]), one_level),
        "c3x2": (QueryUnit("fact", joins=j, quals=[Cmp(D, "<", Lit(500_000))], targets=[Agg("sum", V), Agg("sum", ColRef("g")), Agg("count", None)]),
                 one_level),
    }
    for name, (q, want) in shapes.items():
        cp = compile_query(st, q)
        assert _names(_fused(cp)) == want, (name, _names(_fused(cp)))
    # a small fact table: nothing to slice for -- row order (the interpreter / the direct kernel)
    assert "hdk_join_scatter_slices" not in _names(_fused(compile_query(st, shapes["c3g"][0])), total_rows=1_000_000)


def test_non_grouped_benchmark_queries_stream_column_by_column():
    """NonGroupedAgg/NGA01-05.sql (six aggregates over six INT columns) used to run on the batched interpreter: now the
    column-by-column streaming kernel (scan_agg_cols.h); a filter (rows tie the columns together) keeps the interpreter."""
    from syn_queries import NGA_COLS, nga, syn_table
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(np.random.default_rng(5), 50_000, NGA_COLS, null_frac=0.02), fragment_size=20_000)
    for i in range(1, 6):
        assert _names(compile_query(st, nga(i))) == "hdk_scan_agg_cols,hdk_finalize", i
        assert _names(compile_query(st, nga(i)), flags=A.LAUNCH_FORCE_GENERIC) == "hdk_scan_agg_vec,hdk_finalize"
    filtered = QueryUnit("syn", quals=[Cmp(ColRef("x10"), ">", Lit(3))], targets=[Agg("sum", ColRef("x100")), Agg("sum", ColRef("y100"))])
    assert _names(compile_query(st, filtered)) == "hdk_scan_agg_vec,hdk_finalize"


def test_multistep_and_multicol_benchmark_shapes_stay_on_chip():
    """MultiStep/MSBS001, MSPHS001 (six aggregates over two columns and an expression, 1 000 groups) and
    PerfectHashMultiCol/PHM001-002 (two key columns) used to fall to global atomics, the interpreter or the perfect-partitioned
    passes: now the multi-argument on-chip kernel (scan_bhm.h).  Its switch gives the old routes back."""
    from syn_queries import msbs, msphs, phm, syn_table
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(np.random.default_rng(6), 60_000, ("x10", "y10", "z10", "x100", "x1k")), fragment_size=20_000)
    for q, fold in ((msbs(1), "hdk_bhm_fold"), (msbs(1, key_type=FP64), "hdk_bhm_fold"), (msphs(1), "hdk_finalize"), (phm(1), "hdk_finalize"),
                    (phm(2), "hdk_finalize")):
        names = _names(compile_query(st, q))
        assert names == f"hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,{fold}", (q.groupby, names)
        assert "global" not in names and "hdk_pp_" not in names and "_vec" not in names
    assert _names(compile_query(st, msphs(1)), flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS).startswith("hdk_scan_agg_global")
    # unknown row bound: the packed fields cannot be sized -- the general routes
    assert "bhm" not in _names(compile_query(st, msphs(1)), total_rows=0)


def test_bigint_columns_and_filters_stay_on_chip():
    """The same shapes over 8-byte columns whose statistics fit 32 bits (Arrow int64 tables), and behind plain filters / AND-OR-NOT
    programs: still the multi-argument on-chip kernel (one pass and two), not global atomics or the interpreter.  An 8-byte
    column whose statistics do NOT fit 32 bits, and a mix of widths, keep the general routes."""
    from syn_queries import msbs, msphs, phm, syn_table
    import dataclasses
    from hdk_amd.ir import Not
    rng = np.random.default_rng(8)
    names_ = ("x10", "y10", "z10", "x100", "x1k", "x10k")
    narrow = syn_table(rng, 60_000, names_)
    wide = {k: v.astype(np.int64) for k, v in narrow.items()}
    st4, st8 = ArrowStorage(), ArrowStorage()
    st4.import_numpy("syn", narrow, fragment_size=20_000)
    st8.import_numpy("syn", wide, fragment_size=20_000)
    X10, X100 = ColRef("x10"), ColRef("x100")
    quals = [Or(Cmp(X10, "<", Lit(4)), Not(Cmp(X100, ">", Lit(50))))]
    for st in (st4, st8):
        for q, fold in ((msbs(1, key_type=FP64), "hdk_bhm_fold"), (msphs(1), "hdk_finalize"), (phm(2), "hdk_finalize")):
            for qq in (q, dataclasses.replace(q, quals=[Cmp(X10, "<", Lit(7))]), dataclasses.replace(q, quals=quals)):
                if st is st4 and not qq.quals:
                    continue  # (the test above)
                names = _names(compile_query(st, qq))
                assert names == f"hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,{fold}", (qq.groupby, qq.quals, names)
        names = _names(compile_query(st, dataclasses.replace(msphs(2), quals=quals)))
        assert names == "hdk_bhm_scatter,hdk_bhm_aggregate,hdk_bhm_reduce_slabs,hdk_finalize", names
    big = dict(wide)
    big["x10"] = big["x10"] + (1 << 40)
    st = ArrowStorage()
    st.import_numpy("syn", big, fragment_size=20_000)
    assert "bhm" not in _names(compile_query(st, msphs(1)))
    mixed = dict(wide)
    mixed["x10"] = narrow["x10"]
    st = ArrowStorage()
    st.import_numpy("syn", mixed, fragment_size=20_000)
    assert "bhm" not in _names(compile_query(st, msphs(1)))


def test_multistep_switch_gives_the_old_routes_back(monkeypatch):
    from syn_queries import msphs, phm, syn_table
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(np.random.default_rng(6), 60_000, ("x10", "y10", "z10", "x100", "x1k")), fragment_size=20_000)
    monkeypatch.setenv("HDK_HIP_NO_BHM", "1")
    assert _names(compile_query(st, msphs(1))).startswith("hdk_scan_agg_global")
    assert _names(compile_query(st, phm(1))).startswith("hdk_scan_agg_vec")


def test_multistep_shapes_beyond_lds_take_the_range_bin_passes():
    """MSBS002-003, MSPHS002-003 (10 K / 100 K groups), PHM003-005 (10 K - 1 M entries): no global atomics, no hdk_pp_* passes
    (round 5's routes) -- 4-byte tuples by key range, scan_bhm_part.h."""
    from syn_queries import msbs, msphs, phm, syn_table
    st = ArrowStorage()
    st.import_numpy("syn", syn_table(np.random.default_rng(7), 60_000, ("x10", "y10", "z10", "x100", "x1k", "x10k", "x100k")), fragment_size=20_000)
    for q, fold in ((msbs(2), "hdk_bhm_fold"), (msbs(3, key_type=FP64), "hdk_bhm_fold"), (msphs(2), "hdk_finalize"), (msphs(3), "hdk_finalize"),
                    (phm(4), "hdk_finalize"), (phm(5), "hdk_finalize")):
        names = _names(compile_query(st, q))
        assert names == f"hdk_bhm_scatter,hdk_bhm_aggregate,hdk_bhm_reduce_slabs,{fold}", (q.groupby, names)
    # (PHM003's 10 000 entries of 12 bytes fit one CU's LDS: one pass)
    assert _names(compile_query(st, phm(3))) == "hdk_scan_agg_bhm,hdk_bhm_reduce_slabs,hdk_finalize"
    # small inputs: the passes do not pay (the global-atomics kernel); HDK_HIP_NO_BHM_PARTITIONS: round 5's routes
    assert "bhm" not in _names(compile_query(st, msphs(2)), total_rows=1_000_000)
