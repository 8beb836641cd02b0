"""The reference's layout / reduction matrices (Tests/ResultSetTest.cpp:1658-1909 `Reduce.*`, :2239-2977
`ReduceRandomGroups.*`, helpers Tests/ResultSetTestUtils.cpp) against the ORACLE's reducer and against plan.py's
layouts.  The cases are data (tests/golden/resultset_matrices.json); layouts, fills and expectations are restated
in tests/rs_matrix.py without using hdk_amd/plan.py.  CPU only -- the same cases run through
hdk_hip_reduce_buffers in tests/test_gpu_resultset_matrices.py."""
import numpy as np
import pytest

import rs_matrix as M
from hdk_amd import _abi as A

DOC = M.load_matrices()
REDUCE = DOC["reduce_cases"]
RANDOM = DOC["random_group_cases"]


def reduce_with_oracle(O, lay, plan, this_buf, this_entries, that_bufs):
    iv = np.array(lay.init_vals, dtype=np.int64)
    this64 = this_buf.view(np.int64)
    for tb in that_bufs:
        assert O.reduce(plan, this64, this_entries, tb.view(np.int64), lay.entry_count, iv) == 0
    return this_buf


def run_reduce_case(O, case, reducer):
    """test_reduce (ResultSetTest.cpp:1051-1156) for one case; returns (layout of the result, result buffer)."""
    lay = M.make_layout(DOC, case)
    plan = M.make_plan(lay)
    b1 = M.fill_storage(O, lay, M.make_generator(case["gen1"], lay.entry_count), case["step"])
    b2 = M.fill_storage(O, lay, M.make_generator(case["gen2"], lay.entry_count), case["step"])
    if lay.kind == "perfect":
        return lay, reducer(O, lay, plan, b1, lay.entry_count, [b2])
    # baseline: the reduced set is sized for the sum of the entry counts and every partial is re-inserted
    # (ResultSetManager::reduce, QE/ResultSetReduction.cpp:905-1000; QE/Execute.cpp:1241-1275)
    rl, rbuf = M.result_storage(O, lay, 2 * lay.entry_count)
    return rl, reducer(O, lay, plan, rbuf, rl.entry_count, [b1, b2])


def check_reduce_case(case, rl, rbuf):
    step = case["step"]
    rows = {}
    for e in range(rl.entry_count):
        if rl.is_empty(rbuf, e):
            continue
        row = rl.decode_row(rbuf, e)
        # perfect hash: the row index IS the entry index; baseline: rows sorted by the first column
        idx = e if rl.kind == "perfect" else int(row[0][1])
        rows[idx] = row
    if rl.kind == "perfect":
        assert sorted(rows) == [i for i in range(rl.entry_count) if i % step == 0]
    else:
        assert sorted(rows) == list(range(2 * M.make_layout(DOC, case).entry_count))
    for idx, row in rows.items():
        want = M.expected_reduce_row(rl, idx, step)
        for (cls, got), w in zip(row, want):
            if w is None:
                continue
            if cls == "fp":
                assert got == pytest.approx(w, rel=1e-12), (case["name"], idx)
            else:
                assert got == w, (case["name"], idx)


@pytest.mark.parametrize("case", REDUCE, ids=[c["name"] for c in REDUCE])
def test_reduce_matrix_oracle(oracle, case):
    lay = M.make_layout(DOC, case)
    if not M.supported_by_library(lay):
        pytest.skip("slot widths outside the library's descriptor subset (none of the reference's matrices since round 3)")
    rl, rbuf = run_reduce_case(oracle, case, reduce_with_oracle)
    check_reduce_case(case, rl, rbuf)


def run_random_case(O, case, reducer, seed):
    lay = M.make_layout(DOC, case)
    plan = M.make_plan(lay)
    em = M.Emulator(O, lay, case["prct1"], case["prct2"], case["flow"], seed)
    if lay.kind == "perfect":
        return lay, em, reducer(O, lay, plan, em.bufs[0], lay.entry_count, [em.bufs[1]])
    rl, rbuf = M.result_storage(O, lay, 2 * lay.entry_count)
    return rl, em, reducer(O, lay, plan, rbuf, rl.entry_count, em.bufs)


def check_random_case(case, rl, em, rbuf):
    want = em.expected()
    got = {}
    n0 = em.lay.entry_count
    for e in range(rl.entry_count):
        if rl.is_empty(rbuf, e):
            continue
        # group index: the entry itself (perfect hash) or key / 2 (EvenNumberGenerator advances once per entry)
        g = e if rl.kind == "perfect" else rl.read_key(rbuf, e, 0) // 2
        assert 0 <= g < n0 and g not in got
        got[g] = rl.decode_row(rbuf, e)
    assert sorted(got) == sorted(want), case["name"]
    for g, row in got.items():
        for t, (cls, val), w in zip(rl.targets, row, want[g]):
            if t.agg == "avg":
                assert (val is None) == (w is None), (case["name"], g)
                if w is not None:
                    assert val == pytest.approx(w, abs=0.01)  # EPS of the reference's ASSERT_NEAR
            else:
                assert val == w, (case["name"], g, t.agg)


@pytest.mark.parametrize("case", RANDOM, ids=[c["name"] for c in RANDOM])
def test_reduce_random_groups_oracle(oracle, case):
    for seed in (1, 2, 3):
        rl, em, rbuf = run_random_case(oracle, case, reduce_with_oracle, seed)
        check_random_case(case, rl, em, rbuf)


# ---- plan.py against the same layout rules ------------------------------------------------------------------
def _layout_of_compiled(cp):
    """What plan.py decided, in rs_matrix terms."""
    p = cp.plan
    slots = []
    for ti in range(p.num_targets):
        tg = p.targets[ti]
        slots.append((int(tg.slot_width), int(tg.slot_off)))
        if tg.agg == A.AGG_AVG:
            slots.append((int(tg.slot2_width), int(tg.slot2_off)))
    return slots


@pytest.mark.parametrize("columnar", [False, True])
@pytest.mark.parametrize("shape", ["one_col", "one_col_keyless", "two_col", "baseline", "compact4"])
def test_plan_layout_follows_the_descriptor_rules(shape, columnar):
    """plan.py's row-wise offsets, row size, columnar offsets and buffer size against the independent restatement of
    ColSlotContext / QueryMemoryDescriptor in rs_matrix.Layout, for the descriptor families of the matrices."""
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.plan import columnar_slot_offsets, compile_query
    from hdk_amd.storage import ArrowStorage
    rng = np.random.default_rng(3)
    n = 400
    v = rng.integers(-50, 50, n).astype(np.int64)
    v[::7] = A.NULL_BIGINT
    st = ArrowStorage()
    st.import_numpy("t", {"k": rng.integers(0, 100, n).astype(np.int64), "k2": rng.integers(0, 6, n).astype(np.int64),
                          "v": v, "w": rng.integers(1, 9, n).astype(np.int32), "d": rng.normal(size=n),
                          "big": rng.integers(0, 50, n).astype(np.int64) * (2**35)})
    Q = {
        "one_col": QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("avg", ColRef("v")), Agg("sum", ColRef("v")),
                                                                 Agg("min", ColRef("d"))]),
        "one_col_keyless": QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("avg", ColRef("w")),
                                                                         Agg("sum", ColRef("w"))]),
        "two_col": QueryUnit("t", groupby=[ColRef("k"), ColRef("k2")], targets=[KeyRef(0), KeyRef(1), Agg("count"),
                                                                               Agg("max", ColRef("v"))]),
        "baseline": QueryUnit("t", groupby=[ColRef("big"), ColRef("k2")], force_baseline=True,
                              targets=[Agg("sum", ColRef("v")), Agg("avg", ColRef("d")), Agg("count")]),
        "compact4": QueryUnit("t", groupby=[ColRef("w")], targets=[KeyRef(0), Agg("count")]),
    }
    q = Q[shape]
    q.output_columnar = columnar
    cp = compile_query(st, q)
    p = cp.plan
    slots = _layout_of_compiled(cp)
    widths = [w for w, _ in slots]
    desc = {"kind": "perfect" if p.query_kind == A.Q_PERFECT_HASH else "baseline", "entry_count": int(p.entry_count),
            "group_col_widths": [8] * int(p.key_count)}
    lay = M.Layout(desc, [], 8, columnar, bool(p.keyless), int(p.idx_target_as_key))
    lay.slot_widths, lay.slot_target = widths, list(range(len(widths)))
    lay._copy_widths(lay)
    assert shape != "one_col_keyless" or p.keyless == 1
    assert shape != "compact4" or widths == [4, 4]
    if columnar:
        assert columnar_slot_offsets(cp) == lay.col_off
    else:
        assert int(p.key_width) == 8
        assert [o for _, o in slots] == [lay.key_bytes + o for o in lay.slot_off]
        assert int(p.row_size_quad) * 8 == lay.row_bytes
    assert cp.buffer_bytes == lay.buffer_bytes
