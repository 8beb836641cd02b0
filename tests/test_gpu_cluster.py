"""Join probes over key-range-clustered outer rows (hdk_amd/csrc/scan_cluster.h): the pre-pass permutes the outer
columns, the plan's ordinary kernel runs over the permuted fragments.  Forced on small tables here
(LAUNCH_CLUSTER_PROBES); integer results must equal the oracle's bit for bit -- aggregates do not depend on row order."""
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
            assert names.startswith("hdk_cluster_by_key,hdk_cluster_params,hdk_scan_agg_vec_join") or \
                names.startswith("hdk_cluster_by_key,hdk_join_agg_direct"), names
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
        for flags, prefix in ((0, "hdk_join_agg_direct"), (A.LAUNCH_CLUSTER_PROBES, "hdk_cluster_by_key,hdk_join_agg_direct"),
                              (A.LAUNCH_FORCE_GENERIC, "hdk_scan_agg_vec_join")):
            ex = gpu_executor_factory(st)
            ex.fuse_join_tables = True
            step = ex.prepare(cp, flags=flags, grid=(0 if flags else 5))
            assert step.kernel_names().startswith(prefix) or flags == A.LAUNCH_FORCE_GENERIC, step.kernel_names()
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
