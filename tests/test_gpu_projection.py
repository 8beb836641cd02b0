"""GPU parity for the filter/project kernel (wave-level selection-vector compaction): the SET of output
rows must equal the oracle's (row order on a GPU is scheduling-dependent in the reference as well)."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Cmp, ColRef, JoinSpec, Lit, Proj, QueryUnit
from hdk_amd.storage import ArrowStorage

from test_projection import run_projection_oracle

pytestmark = pytest.mark.gpu


def _sorted_rows(cp, buf, n):
    pos, cols = rs.projection_arrays(cp, buf, n)
    m = np.stack([pos] + cols, axis=1) if len(pos) else np.zeros((0, 1 + len(cols)), dtype=np.int64)
    return m[np.lexsort(m.T[::-1])]


@pytest.mark.parametrize("columnar", [False, True])
def test_filter_project_matches_oracle(oracle, gpu_executor_factory, columnar):
    rng = np.random.default_rng(8)
    n = 700_000
    a = rng.integers(0, 1000, n).astype(np.int64)
    b = rng.integers(-500, 500, n).astype(np.int32)
    b[rng.random(n) < 0.05] = A.NULL_INT
    d = rng.normal(size=n)
    s = rng.integers(0, 100, n).astype(np.int8)
    st = ArrowStorage()
    st.import_numpy("t", {"a": a, "b": b, "d": d, "s": s}, fragment_size=199_999)
    for sel in (990, 500, 0):  # ~1 %, ~50 %, 100 % selectivity
        q = QueryUnit("t", quals=[Cmp(ColRef("a"), ">=", Lit(sel))], output_columnar=columnar,
                      targets=[Proj(ColRef("a"), "a"), Proj(ColRef("b") * 3 - ColRef("a"), "e"), Proj(ColRef("d"), "d"),
                               Proj(ColRef("s"), "s"), Proj(ColRef("b"), "b")])
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0
        res = gpu_executor_factory(st).execute(cp)
        assert res.total_matched == nrows == int((a >= sel).sum())
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))


def test_project_with_join_and_limit(oracle, gpu_executor_factory):
    rng = np.random.default_rng(9)
    nd, nf = 1000, 100_000
    st = ArrowStorage()
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64), "w": rng.integers(0, 50, nd).astype(np.int32)})
    st.import_numpy("fact", {"fk": rng.integers(-10, nd + 10, nf).astype(np.int64), "v": rng.integers(0, 10**6, nf)},
                    fragment_size=30_000)
    q = QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key")], quals=[Cmp(ColRef("v"), "<", Lit(500_000))],
                  targets=[Proj(ColRef("v"), "v"), Proj(ColRef("w", "dim"), "w"), Proj(ColRef("fk"), "fk")])
    cp, want, err, nrows = run_projection_oracle(oracle, st, q)
    res = gpu_executor_factory(st).execute(cp)
    assert res.total_matched == nrows
    assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))
    # LIMIT: the buffer fills up, the counter keeps counting, the error code is negative (benign)
    q.scan_limit = 1000
    cp, want, err, nrows = run_projection_oracle(oracle, st, q)
    res = gpu_executor_factory(st).execute(cp)
    assert err < 0 and res.error_code < 0 and res.total_matched == nrows and res.row_count() == 1000
    got = _sorted_rows(cp, res.buffer, 1000)
    full = QueryUnit("fact", joins=q.joins, quals=q.quals, targets=q.targets)
    cpf, wantf, _, nf_rows = run_projection_oracle(oracle, st, full)
    allrows = {tuple(r) for r in _sorted_rows(cpf, wantf, nf_rows).tolist()}
    assert all(tuple(r) in allrows for r in got.tolist())  # any 1000 of the qualifying rows


@pytest.mark.parametrize("columnar", [False, True])
def test_direct_filter_project_kernel(oracle, gpu_executor_factory, columnar):
    """hdk_scan_project_direct (plain-column filters and targets): mixed widths, NULLs, fp and int compares,
    ragged tails, a LIMIT that fills up -- against the oracle and against the interpreter kernel."""
    rng = np.random.default_rng(18)
    n = 900_001
    a = rng.integers(0, 1000, n).astype(np.int64)
    b = rng.integers(-500, 500, n).astype(np.int32)
    b[rng.random(n) < 0.05] = A.NULL_INT
    s = rng.integers(-100, 100, n).astype(np.int8)
    h = rng.integers(0, 30000, n).astype(np.int16)
    d = rng.normal(size=n)
    d[rng.random(n) < 0.05] = np.frombuffer(np.int64(A.NULL_DOUBLE_BITS).tobytes(), dtype=np.float64)[0]
    st = ArrowStorage()
    st.import_numpy("t", {"a": a, "b": b, "s": s, "h": h, "d": d}, fragment_size=230_003)
    cases = [
        ([Cmp(ColRef("a"), ">=", Lit(990))], ["a", "b", "d", "s", "h"]),
        ([Cmp(ColRef("b"), "<", Lit(0)), Cmp(ColRef("d"), ">", Lit(0.25))], ["b", "d"]),
        ([Cmp(ColRef("s"), "<>", Lit(7)), Cmp(ColRef("h"), "<=", Lit(20000)), Cmp(ColRef("b"), ">", Lit(1.5))], ["h", "a"]),
        ([], ["s", "b"]),
        ([Cmp(ColRef("d"), "<", Lit(-3))], ["d"]),
    ]
    ex = gpu_executor_factory(st)
    for quals, cols in cases:
        q = QueryUnit("t", quals=quals, output_columnar=columnar, targets=[Proj(ColRef(c), c) for c in cols])
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0
        step = ex.prepare(cp)
        assert step.kernel_names().endswith("hdk_scan_project_direct")
        res = step.run()
        step.free()
        assert res.total_matched == nrows
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))
        res2 = ex.execute(cp, flags=A.LAUNCH_FORCE_GENERIC)
        assert res2.total_matched == nrows
        assert np.array_equal(_sorted_rows(cp, res2.buffer, nrows), _sorted_rows(cp, want, nrows))
    q = QueryUnit("t", quals=[Cmp(ColRef("a"), "<", Lit(500))], output_columnar=columnar, scan_limit=5000,
                  targets=[Proj(ColRef("a"), "a"), Proj(ColRef("b"), "b")])
    cp, want, err, nrows = run_projection_oracle(oracle, st, q)
    res = ex.execute(cp)
    assert res.total_matched == nrows == int((a < 500).sum()) and res.error_code < 0
    got = rs.to_columns(cp, res.buffer, nrows=5000)
    assert len(got["a"]) == 5000 and all(x < 500 for x in got["a"])


def test_two_pass_forms_one_pass_and_their_fallbacks(oracle, gpu_executor_factory):
    """The two passes: the counting pass hands a selection bitmask to the writing pass when the launch states its row
    count; without one (total_rows = 0) the filter is evaluated twice; a mask that is too short (5 M rows = 1221 tiles
    against the 1024 tiles of slack) is re-evaluated past its end.  The writing pass is picked on the device from the
    share of rows that pass (sparse: row list + gathers; dense: coalesced column reads + value compaction through LDS);
    HDK_HIP_PROJECT_WRITER forces either.  HDK_HIP_PROJECT_ONE_PASS takes the one-pass kernel (decoupled look-back over
    batches of tiles), which hands the launch to the two passes armed behind it when the input has more tiles than status
    words.  Same rows always, at 3.7 % and at 60 % selectivity."""
    import os
    rng = np.random.default_rng(33)
    n = 5_000_000
    a = rng.integers(0, 1000, n).astype(np.int64)
    b = rng.integers(-500, 500, n).astype(np.int32)
    st = ArrowStorage()
    st.import_numpy("t", {"a": a, "b": b}, fragment_size=1_300_007)
    ex = gpu_executor_factory(st)
    for bound in (37, 600):
        q = QueryUnit("t", quals=[Cmp(ColRef("a"), "<", Lit(bound))], output_columnar=True,
                      targets=[Proj(ColRef("a"), "a"), Proj(ColRef("b"), "b")])
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0 and nrows == int((a < bound).sum())
        want_rows = _sorted_rows(cp, want, nrows)
        one = {"HDK_HIP_PROJECT_ONE_PASS": "1"}
        for total_rows, env in ((n, {}), (0, {}), (1, {}), (n, {"HDK_HIP_PROJECT_WRITER": "sparse"}),
                                (n, {"HDK_HIP_PROJECT_WRITER": "dense"}), (0, {"HDK_HIP_PROJECT_WRITER": "dense"}),
                                (n, one), (1, one), (1, dict(one, HDK_HIP_PROJECT_STATUS_SLACK="0")),
                                (n, dict(one, HDK_HIP_PROJECT_STATUS_SLACK="0"))):
            os.environ.update(env)
            try:
                step = ex.prepare(cp)
                assert step.kernel_names().endswith("hdk_scan_project_direct")
                assert step.kernel_names().startswith("hdk_scan_project_stream") == ("HDK_HIP_PROJECT_ONE_PASS" in env)
                step.ko.total_rows = total_rows
                res = step.run()
                step.free()
            finally:
                for k in env:
                    del os.environ[k]
            assert res.total_matched == nrows, (bound, total_rows, env)
            assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), want_rows), (bound, total_rows, env)


@pytest.mark.parametrize("columnar,fused", [(False, True), (True, True), (True, False)])
def test_direct_filter_project_through_a_join(oracle, gpu_executor_factory, columnar, fused, monkeypatch):
    """The two-pass kernels take ONE inner one-to-one join too (scan_project_fast.h: pf_join_probe): the counting pass
    probes the rows that passed the filters, a row without a partner loses its verdict bit, the sparse writing pass
    gathers the joined columns -- from the fused entries or, with the reference's table, through the row id.  NULL keys,
    keys without a partner, 2-byte unsigned-looking keys, 1 % / 60 % / 100 % selectivity, ragged fragments, a LIMIT; the
    batched interpreter (HDK_HIP_PROJECT_NO_FAST_JOIN) gives the same set of rows; LEFT joins and filters on a joined column stay
    with the interpreters."""
    rng = np.random.default_rng(31)
    nd, nf = 40_000, 600_011
    st = ArrowStorage()
    w = rng.integers(-50, 50, nd).astype(np.int32)
    w[rng.random(nd) < 0.05] = A.NULL_INT
    st.import_numpy("dim", {"key": rng.permutation(nd).astype(np.int64) * 2 + 5, "w": w, "z": rng.integers(-2**40, 2**40, nd)})
    fk = rng.integers(0, 2 * nd + 20, nf).astype(np.int64)
    fk[rng.random(nf) < 0.03] = A.NULL_BIGINT
    st.import_numpy("fact", {"fk": fk, "v": rng.integers(0, 1000, nf).astype(np.int64), "s": rng.integers(0, 100, nf).astype(np.int16)},
                    fragment_size=170_001)
    j = [JoinSpec("dim", ColRef("fk"), "key")]
    for sel in (990, 400, 0):
        q = QueryUnit("fact", joins=j, quals=[Cmp(ColRef("v"), ">=", Lit(sel)), Cmp(ColRef("s"), "<", Lit(95))], output_columnar=columnar,
                      targets=[Proj(ColRef("v"), "v"), Proj(ColRef("w", "dim"), "w"), Proj(ColRef("s"), "s"), Proj(ColRef("z", "dim"), "z")])
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0 and nrows > 0
        ex = gpu_executor_factory(st)
        ex.fuse_join_tables = fused
        step = ex.prepare(cp)
        assert step.kernel_names().startswith("hdk_scan_project_count,hdk_scan_project_offsets"), step.kernel_names()
        res = step.run()
        step.free()
        assert res.total_matched == nrows
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))
        monkeypatch.setenv("HDK_HIP_PROJECT_NO_FAST_JOIN", "1")
        ex2 = gpu_executor_factory(st)
        ex2.fuse_join_tables = fused
        step = ex2.prepare(cp)
        assert step.kernel_names() == "hdk_scan_project_join"
        res2 = step.run()
        step.free()
        monkeypatch.delenv("HDK_HIP_PROJECT_NO_FAST_JOIN")
        assert np.array_equal(_sorted_rows(cp, res2.buffer, nrows), _sorted_rows(cp, want, nrows))
    # LIMIT
    q = QueryUnit("fact", joins=j, quals=[Cmp(ColRef("v"), ">=", Lit(100))], output_columnar=columnar, scan_limit=777,
                  targets=[Proj(ColRef("v"), "v"), Proj(ColRef("w", "dim"), "w")])
    cp, want, err, nrows = run_projection_oracle(oracle, st, q)
    res = gpu_executor_factory(st).execute(cp)
    assert err < 0 and res.error_code < 0 and res.total_matched == nrows and res.row_count() == 777
    # not this kernel's: LEFT join, a filter on the joined column
    for q in (QueryUnit("fact", joins=[JoinSpec("dim", ColRef("fk"), "key", type="left")], quals=[Cmp(ColRef("v"), ">=", Lit(900))],
                        targets=[Proj(ColRef("v"), "v"), Proj(ColRef("w", "dim"), "w")]),
              QueryUnit("fact", joins=j, quals=[Cmp(ColRef("w", "dim"), ">", Lit(0))], targets=[Proj(ColRef("v"), "v")])):
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        step = gpu_executor_factory(st).prepare(cp)
        assert "hdk_scan_project_count" not in step.kernel_names(), step.kernel_names()
        res = step.run()
        step.free()
        assert res.total_matched == nrows
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows))
