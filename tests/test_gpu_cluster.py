"""Join probes over key-range-clustered outer rows (hdk_amd/csrc/scan_cluster.h): the pre-pass permutes the outer
columns, the plan's ordinary kernel runs over the permuted fragments.  Forced on small tables here
(LAUNCH_CLUSTER_PROBES); integer results must equal the oracle's bit for bit -- aggregates do not depend on row order."""
import os

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd.ir import Agg, Cmp, ColRef, JoinSpec, KeyRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from util import assert_buffers_equal, run_oracle

pytestmark = pytest.mark.gpu


def _tables(nf, nd, key_kind, seed):
    rng = np.random.default_rng(seed)
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64) + 1000, "dval": rng.integers(0, 10**6, nd).astype(np.int64),
                            "g": rng.integers(0, 9, nd).astype(np.int64)})
    if key_kind == "uniform":
        fk = rng.integers(1000 - 50, 1000 + nd + 50, nf).astype(np.int64)        # some keys have no partner
    elif key_kind == "hot":                                                        # 60 % of the rows share 3 keys: overflow fragment
        fk = np.where(rng.random(nf) < 0.6, rng.choice([1000 + 5, 1000 + nd // 2, 1000 + nd - 1], nf),
                      rng.integers(1000, 1000 + nd, nf)).astype(np.int64)
    else:                                                                          # already sorted
        fk = np.sort(rng.integers(1000, 1000 + nd, nf)).astype(np.int64)
    fk[rng.random(nf) < 0.03] = A.NULL_BIGINT
    val = rng.integers(-2**31, 2**31, nf).astype(np.int64)
    val[rng.random(nf) < 0.05] = A.NULL_BIGINT
    st.import_numpy("fact", {"fk": fk, "val": val, "w": rng.normal(size=nf), "k2": rng.integers(0, 5, nf).astype(np.int64)},
                    fragment_size=nf // 3 + 17)
    return st


@pytest.mark.parametrize("key_kind", ["uniform", "hot", "sorted"])
def test_clustered_probes_match_the_oracle(oracle, gpu_executor_factory, key_kind):
    st = _tables(700_000, 40_000, key_kind, 11)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    queries = [
        # C3's shape: non-grouped SUM over a fact column + a joined column
        QueryUnit("fact", joins=j, targets=[Agg("sum", ColRef("val") + ColRef("dval", "dim"), "s"), Agg("count", None, "c")]),
        # perfect-hash group-by on a joined column, filter on an outer and on a joined column, MIN / MAX / AVG
        QueryUnit("fact", joins=j, quals=[Cmp(ColRef("val"), ">", Lit(-10**9)), Cmp(ColRef("dval", "dim"), "<", Lit(900_000))],
                  groupby=[ColRef("g", "dim")], targets=[KeyRef(0, "g"), Agg("min", ColRef("val"), "lo"),
                                                         Agg("max", ColRef("dval", "dim"), "hi"), Agg("count", ColRef("val"), "c")]),
        # group-by on an OUTER column (it travels through the permutation), three outer columns in all
        QueryUnit("fact", joins=j, groupby=[ColRef("k2")], targets=[KeyRef(0, "k2"), Agg("sum", ColRef("dval", "dim"), "s"),
                                                                     Agg("count", None, "c")]),
        # a double outer column: SUM within the fp tolerance, COUNT exact
        QueryUnit("fact", joins=j, targets=[Agg("avg", ColRef("w"), "a"), Agg("max", ColRef("w"), "m")]),
    ]
    for q in queries:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        for fuse in (True, False):
            ex = gpu_executor_factory(st)
            ex.fuse_join_tables = fuse
            step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
            names = step.kernel_names()
            # the C3 shape over a fused table has a kernel of its own that reads clustered TUPLES; everything else runs the
            # batched interpreter over the permuted columns
            # (and for tables whose key-range slices fit LDS, a path of its own: scan_join_sliced.h)
            # (round 4: join + GROUP BY on the joined column / filters / other target lists over a fused table: the general
            # sliced kernel, scan_join_sliced2.h, with the batched interpreter armed behind it)
            assert names.startswith("hdk_cluster_by_key,hdk_cluster_params,hdk_scan_agg_vec_join") or \
                names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced,hdk_join_agg_direct") or \
                names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_vec_join"), names
            assert_buffers_equal(cp, step.run().buffer, want)
            # a second run of the same prepared step: the scratch of the first is gone, the result is not
            assert_buffers_equal(cp, step.run().buffer, want)
            step.free()


def test_clustering_is_off_for_shapes_it_does_not_cover(gpu_executor_factory):
    st = _tables(50_000, 2_000, "uniform", 5)
    ex = gpu_executor_factory(st)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    q = QueryUnit("fact", joins=j, targets=[Agg("count", None, "c")])
    # only on request; the flag that forbids it; LEFT joins keep every outer row
    assert "cluster" not in ex.prepare(q).kernel_names()
    assert "cluster" not in ex.prepare(q, flags=A.LAUNCH_CLUSTER_PROBES | A.LAUNCH_NO_CLUSTER_PROBES).kernel_names()
    ql = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key", type="left")], targets=[Agg("count", None, "c")])
    assert "cluster" not in ex.prepare(ql, flags=A.LAUNCH_CLUSTER_PROBES).kernel_names()


def _direct_targets():
    V, D = ColRef("val"), ColRef("dval", "dim")
    return [
        [Agg("sum", V + D, "s"), Agg("count", None, "c")],                                   # BASELINE config 3
        [Agg("sum", V * D, "p"), Agg("min", V - D, "lo"), Agg("max", D, "hi"), Agg("count", V, "cv")],
        [Agg("avg", V, "a"), Agg("sum", D + 5, "s5")],
        [Agg("max", V, "mv"), Agg("min", D - V, "md"), Agg("sum", V, "sv"), Agg("count", None, "c")],
    ]


@pytest.mark.parametrize("key_kind", ["uniform", "hot"])
def test_direct_join_kernel_matches_the_oracle(oracle, gpu_executor_factory, key_kind):
    """hdk_join_agg_direct (scan_join_direct.h): the C3 shape without the interpreter -- in row order, over clustered
    tuples, and against the interpreters; NULL keys, keys without a partner, NULL arguments, every aggregate."""
    st = _tables(600_000, 30_000, key_kind, 17)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    for targets in _direct_targets():
        q = QueryUnit("fact", joins=j, targets=targets)
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        for flags, prefix in ((0, "hdk_join_agg_direct"), (A.LAUNCH_CLUSTER_PROBES, "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced"),
                              (A.LAUNCH_FORCE_GENERIC, "hdk_scan_agg_vec_join")):
            ex = gpu_executor_factory(st)
            ex.fuse_join_tables = True
            step = ex.prepare(cp, flags=flags, grid=(0 if flags else 5))
            assert step.kernel_names().startswith(prefix) or flags == A.LAUNCH_FORCE_GENERIC, step.kernel_names()
            assert_buffers_equal(cp, step.run().buffer, want)
            step.free()


@pytest.mark.parametrize("grid", [1, 5, 64, 300])
def test_sliced_join_with_a_small_caller_grid(oracle, gpu_executor_factory, grid):
    """KernelOptions::gridDimX sizes the slab workspace and what hdk_finalize folds; every block of the sliced probe pass
    flushes into slab[blockIdx.x].  With a grid smaller than the slices need (30 000 keys / 64 = 469 slices here) the
    launch must fall back to row order, with a larger one the members per slice are clamped to it -- never a write past
    the workspace, never a slab that is not folded (round-3 advisor finding)."""
    st = _tables(400_000, 30_000, "uniform", 23)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=_direct_targets()[0])
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    ex = gpu_executor_factory(st)
    ex.fuse_join_tables = True
    step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES, grid=grid)
    names = step.kernel_names()
    # range 30 000 -> 256 slices of 118 keys: below 256 blocks the slices are not used (the row-order kernel reads
    # clustered tuples instead), from 256 on they are, with the members per slice clamped to the grid
    assert ("hdk_join_agg_sliced" in names) == (grid >= 256) and ("hdk_join_agg_direct" in names or "hdk_scan_agg_vec_join" in names), names
    for _ in range(2):
        assert_buffers_equal(cp, step.run().buffer, want)
    step.free()


def test_direct_join_kernel_reports_overflow(oracle, gpu_executor_factory):
    from hdk_amd._lib import HdkHipError
    st = _tables(200_000, 5_000, "uniform", 3)
    t = st.get("fact")
    big = t.columns["val"].fragments[0]
    big[5] = 2**62                                          # val * dval leaves BIGINT for any dval >= 2
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=[Agg("sum", ColRef("val") * ColRef("dval", "dim"), "p")])
    cp, want, err = run_oracle(oracle, st, q)
    if err == 0:
        pytest.skip("the planted row found no partner")
    assert err == A.ERR_OVERFLOW_OR_UNDERFLOW
    for flags in (0, A.LAUNCH_CLUSTER_PROBES):
        ex = gpu_executor_factory(st)
        ex.fuse_join_tables = True
        with pytest.raises(HdkHipError) as ei:
            ex.execute(cp, flags=flags)
        assert ei.value.code == A.ERR_OVERFLOW_OR_UNDERFLOW


def _sliced_tables(nf, nd, seed, x_kind, key_kind="uniform", pay_nulls=False):
    """fact(fk, x) JOIN dim(key, dval): x_kind picks what the slices' tuples look like -- "int32" (fits 32 bits, no
    NULLs: 8-byte tuples), "int32_nulls" (8-byte tuples, INT32_MIN stands for NULL), "wide" (16-byte tuples)."""
    rng = np.random.default_rng(seed)
    st = ArrowStorage()
    dval = rng.integers(-10**6, 10**6, nd).astype(np.int64)
    if pay_nulls:
        dval[rng.random(nd) < 0.1] = A.NULL_BIGINT
    attr = rng.integers(0, 48, nd).astype(np.int32)  # a second inner column: queries that read both get [row id | p0 | p1] entries
    if pay_nulls:
        attr[rng.random(nd) < 0.05] = A.NULL_INT
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64) * 3 - 77, "dval": dval, "attr": attr})  # every third key only
    lo, hi = -77, 3 * nd - 77
    if key_kind == "uniform":
        fk = rng.integers(lo - 20, hi + 20, nf).astype(np.int64)
    elif key_kind == "hot":
        fk = np.where(rng.random(nf) < 0.7, -77 + 3 * 1234, rng.integers(lo, hi, nf)).astype(np.int64)
    else:
        fk = np.sort(rng.integers(lo, hi, nf)).astype(np.int64)
    fk[rng.random(nf) < 0.02] = A.NULL_BIGINT
    if x_kind == "int32":
        x = rng.integers(-2**31, 2**31, nf).astype(np.int64)
    elif x_kind == "int32_nulls":
        x = rng.integers(-2**31 + 1, 2**31, nf).astype(np.int64)
        x[rng.random(nf) < 0.05] = A.NULL_BIGINT
    else:
        x = rng.integers(-2**45, 2**45, nf).astype(np.int64)
        x[rng.random(nf) < 0.05] = A.NULL_BIGINT
    # a second outer column with a narrow range (a group key / a second measure of the fact table: rides in the tuple's spare bits)
    g = rng.integers(-3, 60, nf).astype(np.int64)
    if x_kind != "int32":
        g[rng.random(nf) < 0.03] = A.NULL_BIGINT
    g32 = np.where(g == A.NULL_BIGINT, A.NULL_INT, g).astype(np.int32)  # (the same as an INT column)
    st.import_numpy("fact", {"fk": fk, "x": x, "g": g, "g32": g32}, fragment_size=nf // 4 + 5)
    return st


@pytest.mark.parametrize("x_kind,key_kind,pay_nulls", [("int32", "uniform", False), ("int32_nulls", "uniform", True),
                                                        ("wide", "uniform", True), ("int32", "hot", False),
                                                        ("int32_nulls", "sorted", False), ("wide", "hot", False)])
def test_sliced_join_matches_the_oracle(oracle, gpu_executor_factory, x_kind, key_kind, pay_nulls):
    """scan_join_sliced.h: the join table's key-range slices probed out of LDS -- 8- and 16-byte tuples, NULL keys, keys
    without a partner (two of three slots are empty), NULL payloads, a hot key (overflow area, probed in memory), an input
    that is already clustered (the order probe hands the launch to the row-order kernel); every aggregate of the shape."""
    st = _sliced_tables(900_000, 50_000, 23, x_kind, key_kind, pay_nulls)
    X, D = ColRef("x"), ColRef("dval", "dim")
    for targets in ([Agg("sum", X + D, "s"), Agg("count", None, "c")],
                    [Agg("sum", X, "sx"), Agg("min", X - D, "lo"), Agg("max", D, "hi"), Agg("count", D, "cd")],
                    [Agg("avg", D, "a"), Agg("count", X, "cx")]):
        q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=targets)
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        assert step.kernel_names().startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced"), step.kernel_names()
        assert_buffers_equal(cp, step.run().buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)  # (scratch and mode word are per launch)
        step.free()


def test_sliced_join_survives_stale_statistics(oracle, gpu_executor_factory):
    """Column statistics that no longer hold (an x outside the announced 32-bit range; a payload outside it) make the
    launch fall back to the row-order kernel on the device: same answer as the oracle's."""
    from hdk_amd.storage import ChunkStats
    st = _sliced_tables(600_000, 40_000, 29, "int32")
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")],
                  targets=[Agg("sum", ColRef("x") + ColRef("dval", "dim"), "s"), Agg("count", None, "c")])
    for table, col in (("fact", "x"), ("dim", "dval")):
        c = st.get(table).columns[col]
        saved = c.fragments[0][11]
        c.fragments[0][11] = 2**40 + 3  # the statistics (computed at import) still say +-2^31 / +-1e6
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        assert "hdk_join_agg_sliced" in step.kernel_names()
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        c.fragments[0][11] = saved


def _sliced2_queries():
    """the star-schema shapes around BASELINE config 3 (scan_join_sliced2.h): join + perfect-hash GROUP BY on the joined
    column, filters on either side, lists of integer aggregates over x and the payload"""
    X, P = ColRef("x"), ColRef("dval", "dim")
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    return [
        # SURVEY 8d's variant of C3: GROUP BY a function of the joined column
        QueryUnit("fact", joins=j, groupby=[P / 15625], targets=[KeyRef(0, "g"), Agg("sum", X, "s")]),
        QueryUnit("fact", joins=j, groupby=[P / 31250], targets=[KeyRef(0, "g"), Agg("sum", X + P, "s"), Agg("count", None, "c"),
                                                                  Agg("max", P, "mx"), Agg("min", X, "mn")]),
        # filters: outer column (pass 1: before the scatter) and joined column (pass 2)
        QueryUnit("fact", joins=j, quals=[Cmp(X, ">", Lit(-10**9)), Cmp(P, "<", Lit(500_000))], groupby=[P / 100_000],
                  targets=[KeyRef(0, "g"), Agg("avg", X, "a"), Agg("count", P, "cp")]),
        QueryUnit("fact", joins=j, quals=[Cmp(ColRef("fk"), "<", Lit(10**9)), Cmp(X, "<=", Lit(0)), Cmp(P, "<>", Lit(17))],
                  targets=[Agg("sum", X, "s"), Agg("count", None, "c"), Agg("max", P, "mx")]),
        # non-grouped lists the compile-time form of C3 does not cover
        QueryUnit("fact", joins=j, targets=[Agg("sum", X, "s"), Agg("count", None, "c"), Agg("max", P, "mx")]),
        QueryUnit("fact", joins=j, targets=[Agg("sum", X * P, "p"), Agg("min", X - P, "lo"), Agg("max", P - X, "hi"), Agg("count", X, "cx")]),
        QueryUnit("fact", joins=j, targets=[Agg("sum", P * 3, "p3"), Agg("min", X + 7, "x7"), Agg("avg", P, "ap")]),
    ]


@pytest.mark.parametrize("key_kind", ["uniform", "hot", "sorted"])
@pytest.mark.parametrize("x_kind,pay_nulls", [("int32", False), ("int32_nulls", True)])
def test_sliced2_matches_the_oracle(oracle, gpu_executor_factory, key_kind, x_kind, pay_nulls):
    """hdk_join_agg_sliced2: every shape against the oracle -- keys without a partner, NULL keys, NULL x, NULL payloads
    (a NULL group key takes the translated slot), hot keys (the overflow area), sorted keys (the order probe hands the
    launch to the armed interpreter); a second run of the prepared step gives the same buffer."""
    st = _sliced_tables(500_000, 30_000, 41, x_kind, key_kind, pay_nulls)
    seen = 0
    for q in _sliced2_queries():
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        ex = gpu_executor_factory(st)
        step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        names = step.kernel_names()
        assert names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_vec_join"), (q, names)
        assert_buffers_equal(cp, step.run().buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        # and the same plan on the interpreters
        assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)
        seen += 1
    assert seen == 7


@pytest.mark.parametrize("key_kind", ["uniform", "hot", "sorted"])
def test_sliced2_two_scatter_levels(oracle, gpu_executor_factory, key_kind, monkeypatch):
    """Key ranges beyond 256 LDS-sized slices take a second scatter level (hdk_join_scatter_level2): forced here on a
    90 K-key range with 128-key fine slices (704 slices in 235 coarse bins of 3), every sliced2 shape against the oracle."""
    monkeypatch.setenv("HDK_HIP_SLICE_TWO_LEVELS", "1")
    monkeypatch.setenv("HDK_HIP_SLICE_FINE_KEYS", "128")
    st = _sliced_tables(600_000, 30_000, 47, "int32_nulls", key_kind, True)
    for q in _sliced2_queries():
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        assert step.kernel_names().startswith(
            "hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_scatter_level2,hdk_join_agg_sliced2,hdk_scan_agg_vec_join"), step.kernel_names()
        assert_buffers_equal(cp, step.run().buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
    # the headline form SUM(x + payload) too: beyond 10.2 M keys it takes this path
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=[Agg("sum", ColRef("x") + ColRef("dval", "dim"), "s")])
    cp, want, err = run_oracle(oracle, st, q)
    monkeypatch.setenv("HDK_HIP_SLICED2_ALWAYS", "1")
    step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
    assert "hdk_join_scatter_level2" in step.kernel_names()
    assert_buffers_equal(cp, step.run().buffer, want)
    step.free()


def _sliced2_two_payload_queries():
    """filter on one inner column, group by / aggregate another: the fused entries carry two payload words"""
    X, P, Q = ColRef("x"), ColRef("dval", "dim"), ColRef("attr", "dim")
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    return [
        QueryUnit("fact", joins=j, quals=[Cmp(P, "<", Lit(250_000))], groupby=[Q], targets=[KeyRef(0, "g"), Agg("sum", X, "s"), Agg("count", None, "c")]),
        QueryUnit("fact", joins=j, quals=[Cmp(Q, "=", Lit(7))], targets=[Agg("sum", X + P, "s"), Agg("count", None, "c")]),
        QueryUnit("fact", joins=j, quals=[Cmp(Q, ">=", Lit(10)), Cmp(X, ">", Lit(-10**9))], groupby=[P / 50_000],
                  targets=[KeyRef(0, "g"), Agg("max", Q, "mq"), Agg("min", X - Q, "lo"), Agg("avg", P, "ap")]),
        QueryUnit("fact", joins=j, groupby=[Q / 4], targets=[KeyRef(0, "g"), Agg("sum", P * 2, "p2"), Agg("count", Q, "cq"), Agg("sum", X * Q, "xq")]),
    ]


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("key_kind,two_levels", [("uniform", False), ("hot", False), ("uniform", True), ("sorted", True)])
def test_sliced2_two_payload_words(oracle, gpu_executor_factory, key_kind, two_levels, packed, monkeypatch):
    """hdk_join_agg_sliced2<*, 2 | 3>: plans that read TWO columns of the inner table (filter on one, group by / aggregate the
    other) stay on the sliced path: entries of three words; per slice in LDS two int32 arrays, or ONE array of packed codes
    when the two columns' statistics fit 32 bits together (here: 21 + 6 bits).  NULLs in both columns, keys without a partner,
    a hot key (overflow area probed in memory), one and two scatter levels."""
    if not packed:
        monkeypatch.setenv("HDK_HIP_S2_NO_PACKED_PAIR", "1")
    if two_levels:
        monkeypatch.setenv("HDK_HIP_SLICE_TWO_LEVELS", "1")
        monkeypatch.setenv("HDK_HIP_SLICE_FINE_KEYS", "192")
    st = _sliced_tables(500_000, 30_000, 53, "int32_nulls", key_kind, True)
    for q in _sliced2_two_payload_queries():
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        names = step.kernel_names()
        assert "hdk_join_agg_sliced2" in names and (("hdk_join_scatter_level2" in names) == two_levels), (q, names)
        assert_buffers_equal(cp, step.run().buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)


def _sliced2_outer_pair_queries():
    """a second column of the OUTER table under the join: the group key (GROUP BY fact.g ... WHERE dim.d < c) or a second
    measure; it rides in the bits of the 8-byte tuple the key offset leaves free"""
    X, G, P, Q = ColRef("x"), ColRef("g"), ColRef("dval", "dim"), ColRef("attr", "dim")
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    return [
        QueryUnit("fact", joins=j, quals=[Cmp(P, "<", Lit(250_000))], groupby=[G], targets=[KeyRef(0, "g"), Agg("sum", X, "s")]),
        QueryUnit("fact", joins=j, groupby=[G / 3], targets=[KeyRef(0, "g"), Agg("sum", X + P, "s"), Agg("count", G, "cg"), Agg("max", G, "mg"),
                                                              Agg("count", None, "c")]),
        QueryUnit("fact", joins=j, quals=[Cmp(P, ">=", Lit(-400_000)), Cmp(X, "<", Lit(10**9))],
                  targets=[Agg("sum", X, "sx"), Agg("sum", G, "sg"), Agg("min", G, "mn"), Agg("avg", X * G, "a")]),
        QueryUnit("fact", joins=j, quals=[Cmp(P, "<", Lit(600_000))], groupby=[Q], targets=[KeyRef(0, "q"), Agg("sum", X, "sx"), Agg("sum", G + 5, "sg")]),
        QueryUnit("fact", joins=j, groupby=[G], targets=[KeyRef(0, "g"), Agg("count", None, "c"), Agg("min", P, "lo"), Agg("max", G, "hi")]),
        # the same column as a 4-byte INT
        QueryUnit("fact", joins=j, quals=[Cmp(P, "<", Lit(100_000))], groupby=[ColRef("g32")],
                  targets=[KeyRef(0, "g"), Agg("sum", X, "s"), Agg("count", ColRef("g32"), "c")]),
    ]


@pytest.mark.parametrize("key_kind,x_kind,two_levels", [("uniform", "int32", False), ("hot", "int32_nulls", False), ("uniform", "int32_nulls", True),
                                                        ("sorted", "int32", True)])
def test_sliced2_second_outer_column(oracle, gpu_executor_factory, key_kind, x_kind, two_levels, monkeypatch):
    """GROUP BY an outer column / two outer measures under the sliced join: the second column's code sits above the key
    offset in the tuple (scan_join_sliced.h: SliceArgs::y_shift).  NULL group keys and NULL measures, keys without a partner,
    a hot key, one and two scatter levels; the interpreters give the same buffer."""
    if two_levels:
        monkeypatch.setenv("HDK_HIP_SLICE_TWO_LEVELS", "1")
        monkeypatch.setenv("HDK_HIP_SLICE_FINE_KEYS", "192")
    st = _sliced_tables(500_000, 30_000, 59, x_kind, key_kind, x_kind != "int32")
    for q in _sliced2_outer_pair_queries():
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, q
        step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        names = step.kernel_names()
        assert "hdk_join_agg_sliced2" in names and (("hdk_join_scatter_level2" in names) == two_levels), (q, names)
        assert_buffers_equal(cp, step.run().buffer, want)
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
        assert_buffers_equal(cp, gpu_executor_factory(st).execute(cp, flags=A.LAUNCH_FORCE_GENERIC).buffer, want)


def test_sliced2_second_outer_column_limits(oracle, gpu_executor_factory, monkeypatch):
    """What does not fit the spare bits stays off the sliced path (and still equals the oracle): a second outer column whose
    range needs more bits than the key offset leaves; three outer columns.  A value outside the announced statistics of the
    riding column hands the launch to the armed interpreter."""
    st = _sliced_tables(300_000, 30_000, 61, "int32")
    rng = np.random.default_rng(5)
    f = st.get("fact")
    n = 300_000
    st2 = ArrowStorage()
    st2.import_numpy("dim", {k: np.concatenate(c.fragments) for k, c in st.get("dim").columns.items()})
    cols = {k: np.concatenate(c.fragments) for k, c in f.columns.items()}
    cols["wide"] = rng.integers(0, 2**20, n).astype(np.int64)  # 20 bits + 17 bits of key offset: does not fit
    cols["h"] = rng.integers(0, 9, n).astype(np.int64)
    st2.import_numpy("fact", cols, fragment_size=n // 3 + 1)
    X, G, W, H, P = ColRef("x"), ColRef("g"), ColRef("wide"), ColRef("h"), ColRef("dval", "dim")
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    for q, sliced in [(QueryUnit("fact", joins=j, targets=[Agg("sum", X, "a"), Agg("sum", W, "b")]), False),
                      (QueryUnit("fact", joins=j, targets=[Agg("sum", X, "a"), Agg("sum", G, "b"), Agg("sum", H, "c")]), False),
                      (QueryUnit("fact", joins=j, targets=[Agg("sum", W, "a"), Agg("sum", H + P, "b")]), False),  # (y op payload: not a form)
                      (QueryUnit("fact", joins=j, targets=[Agg("sum", W * H, "a"), Agg("max", P, "b"), Agg("min", H, "c")]), True)]:
        cp, want, err = run_oracle(oracle, st2, q)
        assert err == 0
        step = gpu_executor_factory(st2).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        assert ("hdk_join_agg_sliced2" in step.kernel_names()) == sliced, (q, step.kernel_names())
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()
    # stale statistics of the riding column
    q = _sliced2_outer_pair_queries()[4]
    c = st.get("fact").columns["g"]
    dim_keys = set(np.concatenate(st.get("dim").columns["key"].fragments).tolist())
    at = next(i for i, k in enumerate(st.get("fact").columns["fk"].fragments[0].tolist()) if k in dim_keys)  # a row with a partner
    saved = c.fragments[0][at]
    c.fragments[0][at] = 4000  # the statistics (computed at import) still say -3 .. 59; the layout has no entry for it either
    from hdk_amd._lib import HdkHipError
    ex = gpu_executor_factory(st)
    step = ex.prepare(ex.compile(q), flags=A.LAUNCH_CLUSTER_PROBES)
    assert "hdk_join_agg_sliced2" in step.kernel_names()
    with pytest.raises(HdkHipError) as ei:
        step.run()
    assert ei.value.code == A.ERR_OUT_OF_SLOTS
    step.free()
    q = _sliced2_outer_pair_queries()[2]  # as a measure: no layout to leave, the sum must simply be right
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
    assert "hdk_join_agg_sliced2" in step.kernel_names()
    assert_buffers_equal(cp, step.run().buffer, want)
    step.free()
    c.fragments[0][at] = saved


def test_sliced2_survives_stale_statistics_and_reports_errors(oracle, gpu_executor_factory):
    """an x outside the announced 32 bits -> the armed interpreter redoes the launch; a payload outside them that also
    leaves the group-key range, or a key range that no longer covers the data -> ERR_OUT_OF_SLOTS, as on the interpreter.
    (The oracle is not asked about out-of-range keys: like the reference's get_group_value_fast it would write outside
    the buffer.)"""
    from hdk_amd._lib import HdkHipError
    from hdk_amd.storage import ChunkStats
    st = _sliced_tables(400_000, 30_000, 43, "int32")
    q = _sliced2_queries()[1]
    c = st.get("fact").columns["x"]
    saved = c.fragments[0][11]
    c.fragments[0][11] = 2**40 + 3  # the statistics (computed at import) still say +-2^31
    cp, want, err = run_oracle(oracle, st, q)
    assert err == 0
    step = gpu_executor_factory(st).prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
    assert "hdk_join_agg_sliced2" in step.kernel_names()
    assert_buffers_equal(cp, step.run().buffer, want)
    step.free()
    c.fragments[0][11] = saved
    # a stale payload: outside 32 bits AND outside the group-key range the layout was sized for
    d = st.get("dim").columns["dval"]
    saved = d.fragments[0][11]
    d.fragments[0][11] = 2**40 + 3
    for flags in (A.LAUNCH_CLUSTER_PROBES, A.LAUNCH_FORCE_GENERIC):
        ex = gpu_executor_factory(st)
        step = ex.prepare(ex.compile(q), flags=flags)
        with pytest.raises(HdkHipError) as ei:
            step.run()
        assert ei.value.code == A.ERR_OUT_OF_SLOTS
        step.free()
    d.fragments[0][11] = saved
    # a key range that no longer covers the data, inside 32 bits: the sliced kernel itself reports it
    d.stats = [ChunkStats(-10**6, 10**5, False) for _ in d.stats]
    ex = gpu_executor_factory(st)
    step = ex.prepare(ex.compile(_sliced2_queries()[0]), flags=A.LAUNCH_CLUSTER_PROBES)
    assert "hdk_join_agg_sliced2" in step.kernel_names()
    with pytest.raises(HdkHipError) as ei:
        step.run()
    assert ei.value.code == A.ERR_OUT_OF_SLOTS
    step.free()


_SOAK = os.environ.get("HDK_FUZZ_SEEDS", "")
_SOAK_SEEDS = list(range(*map(int, _SOAK.split(":")))) if _SOAK else []


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [1, 2] + _SOAK_SEEDS)
def test_sliced_join_random_shapes(oracle, gpu_executor_factory, seed):
    """Seeded random shapes through the sliced join: fact sizes from one batch to millions of rows in ragged fragments,
    dimensions from 300 keys (64-key slices) to 2.4 M (9 K-key slices), dense or sparse key ranges, uniform / hot / sorted
    / block-clustered foreign keys (overflow area; the order probe), x columns that travel in 8- or 16-byte tuples, NULL
    keys / x / payloads, one to three random aggregates of the shape.  HDK_FUZZ_SEEDS adds seeds for soak runs."""
    rng = np.random.default_rng(7000 + seed)
    nf = int(rng.choice([5_000, 90_000, 700_000, 2_500_000]))
    nd = int(rng.choice([300, 20_000, 400_000, 2_400_000]))
    stride = int(rng.choice([1, 1, 3]))
    off = int(rng.integers(-1000, 1000))
    st = ArrowStorage()
    dval = rng.integers(-10**6, 10**6, nd).astype(np.int64)
    if rng.random() < 0.5:
        dval[rng.random(nd) < 0.1] = A.NULL_BIGINT
    attr = rng.integers(0, 40, nd).astype(np.int32)
    if rng.random() < 0.5:
        attr[rng.random(nd) < 0.05] = A.NULL_INT
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64) * stride + off, "dval": dval, "attr": attr})
    lo, hi = off, off + stride * (nd - 1) + 1
    kind = str(rng.choice(["uniform", "hot", "sorted", "blocks"]))
    if kind == "uniform":
        fk = rng.integers(lo - 5, hi + 5, nf)
    elif kind == "hot":
        fk = np.where(rng.random(nf) < 0.6, off + stride * int(rng.integers(0, nd)), rng.integers(lo, hi, nf))
    elif kind == "sorted":
        fk = np.sort(rng.integers(lo, hi, nf))
    else:  # runs of 3 000 rows inside a narrow key window: clustered without being sorted
        starts = rng.integers(lo, max(hi - 500, lo + 1), nf // 3000 + 1)
        fk = np.repeat(starts, 3000)[:nf] + rng.integers(0, 500, nf)
    fk = fk.astype(np.int64)
    fk[rng.random(nf) < 0.02] = A.NULL_BIGINT
    xk = str(rng.choice(["int32", "int32_nulls", "wide"]))
    if xk == "int32":
        x = rng.integers(-2**31, 2**31, nf).astype(np.int64)
    elif xk == "int32_nulls":
        x = rng.integers(-2**31 + 1, 2**31, nf).astype(np.int64)
        x[rng.random(nf) < 0.05] = A.NULL_BIGINT
    else:
        x = rng.integers(-2**45, 2**45, nf).astype(np.int64)
        x[rng.random(nf) < 0.05] = A.NULL_BIGINT
    frag = int(rng.integers(nf // 6 + 1, nf + 2))
    gcol = rng.integers(0, int(rng.choice([4, 50, 3000])), nf).astype(np.int64) - 7  # a second outer column (rides in the tuple when it fits)
    if rng.random() < 0.5:
        gcol[rng.random(nf) < 0.04] = A.NULL_BIGINT
    st.import_numpy("fact", {"fk": fk, "x": x, "g": gcol}, fragment_size=frag)
    X, D, G = ColRef("x"), ColRef("dval", "dim"), ColRef("g")
    pool = [Agg("sum", X + D, "s0"), Agg("sum", D + X, "s1"), Agg("count", None, "c"), Agg("sum", X, "sx"), Agg("min", X - D, "lo"),
            Agg("max", D, "hi"), Agg("count", D, "cd"), Agg("avg", D, "a"), Agg("count", X, "cx"), Agg("max", X, "mx")]
    At = ColRef("attr", "dim")
    pool2 = pool[2:] + [Agg("max", At, "ma"), Agg("sum", X * At, "xa"), Agg("count", At, "ca"), Agg("sum", At + 3, "a3")]
    pool3 = pool2 + [Agg("sum", G, "sg"), Agg("min", G, "ng"), Agg("count", G, "cg"), Agg("sum", X + G, "xg")]
    ex = gpu_executor_factory(st)
    for qi in range(5):
        targets = [pool[0]] if (qi == 0 and rng.random() < 0.5) else [pool[int(i)] for i in rng.choice(len(pool), int(rng.integers(1, 4)), replace=False)]
        q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], targets=targets)
        if qi >= 3:  # the general sliced form: filters on either side, GROUP BY a joined column, one or two payload words
            gb = [None, D / int(rng.choice([50_000, 125_000])), At, At / 3, G, G / 5][int(rng.integers(0, 6))]
            quals = [c for c in (Cmp(D, "<", Lit(int(rng.integers(-500_000, 900_000)))), Cmp(At, ">=", Lit(int(rng.integers(0, 30)))),
                                 Cmp(X, ">", Lit(int(rng.integers(-2**31, 0))))) if rng.random() < 0.4]
            pl = pool3 if rng.random() < 0.5 else pool2
            targets = [pl[int(i)] for i in rng.choice(len(pl), int(rng.integers(1, 4)), replace=False)]
            q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=quals, groupby=[gb] if gb is not None else [],
                          targets=([KeyRef(0, "g")] if gb is not None else []) + targets)
        cp, want, err = run_oracle(oracle, st, q)
        what = (seed, qi, nf, nd, stride, kind, xk, [t.name for t in targets])
        assert err == 0, what
        step = ex.prepare(cp, flags=A.LAUNCH_CLUSTER_PROBES)
        names = step.kernel_names()
        if os.environ.get("HDK_SOAK_LOG"):
            with open(os.environ["HDK_SOAK_LOG"], "a") as f:
                f.write(f"{what} {names.split(',')[0]}\n")
        try:
            assert_buffers_equal(cp, step.run().buffer, want)
        except AssertionError as e:
            raise AssertionError(f"{what} kernels {names}\n{e}") from e
        finally:
            step.free()
