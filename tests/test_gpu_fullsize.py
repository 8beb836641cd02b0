"""Every BASELINE.json configuration at BASELINE size on one GPU (workloads.py), checked through properties that do
not need an oracle at that size -- sum of per-group sums = SUM over the column, group count and key set = the
distinct keys, row counts, idempotence, agreement between kernel strategies -- plus a bit-exact comparison with the
oracle on a 2 M-row sample of the same data.  HDK_FULLSIZE_ROWS scales the row counts down (debugging only)."""
import os

import numpy as np
import pytest

from hdk_amd import _abi as A

# (every test here imports torch for the device buffers; the first import on a fresh box can take minutes)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

SCALE = float(os.environ.get("HDK_FULLSIZE_SCALE", "1"))


@pytest.fixture(scope="module")
def mgr():
    from hdk_amd.hip_mgr import HipMgr
    return HipMgr()


@pytest.fixture(autouse=True)
def _release_device_memory():
    yield
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


def _workload(name, mgr, **kw):
    import torch
    from workloads import CONFIGS, Workload
    rows = int(CONFIGS[name][0] * SCALE)
    w = Workload(name, rows, 0, mgr, **kw)
    torch.cuda.synchronize()
    return w


def _run_into(w, out, flags=0):
    import torch
    step = w.ex.prepare(w.compiled, w.frag_ids, flags=flags, out_ptr=out.data_ptr())
    names = step.kernel_names()
    step.enqueue()
    w.ex.mgr.synchronizeStream(0)
    err = int(w.ex.mgr.to_host(step.d_err.ptr, 4, 0, np.int32)[0])
    step.free()
    assert err == 0, err
    torch.cuda.synchronize()
    return names


def _baseline_groups(w, out):
    """(sorted keys, sums in key order) of a row-wise open-addressing table [key | SUM], decoded on the device."""
    import torch
    p = w.compiled.plan
    rq, n = int(p.row_size_quad), int(p.entry_count)
    rows = out[:n * rq].view(n, rq)
    if p.key_width == 4:
        keys = (rows[:, 0] << 32) >> 32  # low half, sign-extended
        live = keys != A.EMPTY_KEY_32
    else:
        keys = rows[:, 0]
        live = keys != A.EMPTY_KEY_64
    sum_quad = int(p.targets[1].slot_off) // 8
    k = keys[live]
    s = rows[:, sum_quad][live]
    order = torch.argsort(k)
    return k[order], s[order]


def _oracle_sample(w, oracle, flags=0, rows=2_000_000):
    from hdk_amd.executor import Executor
    from test_gpu_baseline import _check_rows
    from util import assert_buffers_equal, run_oracle
    st = w.sample_storage(rows)
    cp, want, err = run_oracle(oracle, st, w.query)
    assert err == 0
    ex = Executor(st, 0, w.ex.mgr)
    res = ex.execute(cp, flags=flags)
    if cp.plan.query_kind == A.Q_BASELINE_HASH:
        _check_rows(cp, res.buffer, want)
    else:
        assert_buffers_equal(cp, res.buffer, want)


@pytest.mark.parametrize("name", ["c5", "c5s"])
def test_c5_open_addressing_group_by_at_baseline_size(name, mgr, oracle):
    """C5 (1 B rows, 100 M keys, 200 M-entry table) and the shard one GPU of eight sees (125 M rows from the same key
    domain): radix-partitioned path; group keys = torch.unique of the key column, sum of sums = column sum, two runs
    give the same {key -> sum}, and the atomics kernel agrees."""
    import torch
    w = _workload(name, mgr)
    cp = w.compiled
    assert cp.plan.query_kind == A.Q_BASELINE_HASH
    if SCALE == 1:
        assert cp.entry_count == 200_000_000  # 2 x min(rows, key domain) (QE/RelAlgExecutor.cpp:1553-1557)
    out = torch.empty(cp.buffer_quads, dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    assert names.startswith("hdk_part_scatter"), names
    k1, s1 = _baseline_groups(w, out)
    ref = w.reference_checks()
    assert int(s1.sum().item()) % (1 << 64) == ref["sum_val"]
    uniq = torch.unique(torch.cat([w.cols[w.key_col][f] for f in w.frag_ids]))
    assert k1.numel() == uniq.numel() and bool(torch.equal(k1, uniq))
    del uniq
    _run_into(w, out)  # idempotence (placement may differ, the groups may not)
    k2, s2 = _baseline_groups(w, out)
    assert bool(torch.equal(k1, k2)) and bool(torch.equal(s1, s2))
    del k2, s2
    if name == "c5s":  # the other strategy on the same input (16 ms per 256 M rows: only at the smaller size)
        _run_into(w, out, flags=A.LAUNCH_FORCE_GLOBAL_ATOMICS)
        k3, s3 = _baseline_groups(w, out)
        assert bool(torch.equal(k1, k3)) and bool(torch.equal(s1, s3))
    _oracle_sample(w, oracle, flags=A.LAUNCH_FORCE_PARTITIONED)


def test_c3_join_probe_at_baseline_size(mgr, oracle):
    """C3: 1 B-row fact JOIN 10 M-row dim, SUM(fact.val + dim.dval) = the same sum computed with torch gathers; fused
    and reference table layouts agree; the oracle agrees on a sample."""
    import torch
    w = _workload("c3", mgr)
    cp = w.compiled
    out = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    # a 160 MB table: key-range slices probed out of LDS (scan_join_sliced.h), the row-order kernel armed behind them
    assert names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced,hdk_join_agg_direct"), names
    ref = w.reference_checks()
    got = int(out[0].item()) % (1 << 64)
    assert got == ref["sum_val_plus_dval"]
    assert _run_into(w, out, flags=A.LAUNCH_NO_CLUSTER_PROBES).startswith("hdk_join_agg_direct")  # in row order
    assert int(out[0].item()) % (1 << 64) == got
    assert _run_into(w, out, flags=A.LAUNCH_FORCE_GENERIC).startswith("hdk_scan_agg_vec_join")  # the batched interpreter
    assert int(out[0].item()) % (1 << 64) == got
    w.ex.fuse_join_tables = False  # the reference's table layout (slot -> row id -> inner column)
    _run_into(w, out)
    assert int(out[0].item()) % (1 << 64) == got
    _oracle_sample(w, oracle)


def test_c3g_join_then_group_by_at_baseline_size(mgr, oracle):
    """C3g (SURVEY.md 8d's variant of C3): 1 B-row fact JOIN 10 M-row dim GROUP BY dim.dval / 15625 -> 64 groups, SUM(fact.val):
    per-group sums = torch index_add over the gathered dimension column; the sliced passes and the interpreter agree."""
    import torch
    from hdk_amd.executor import ExecutionResult
    w = _workload("c3g", mgr)
    cp = w.compiled
    out = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    assert names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2,hdk_scan_agg_vec_join"), names
    ref = w.reference_checks()
    first = out.cpu().numpy().copy()
    cols = ExecutionResult(cp, first, cp.entry_count).to_columns()
    assert sorted(cols["g"]) == list(range(64))
    for g, s_ in zip(cols["g"], cols["s"]):
        assert (s_ - ref["group_sums"][g]) % (1 << 64) == 0, g
    assert _run_into(w, out, flags=A.LAUNCH_FORCE_GENERIC).startswith("hdk_scan_agg_vec_join")
    assert np.array_equal(out.cpu().numpy(), first)
    _run_into(w, out)
    assert np.array_equal(out.cpu().numpy(), first)  # idempotent
    _oracle_sample(w, oracle)


def test_c3m_other_target_list_at_baseline_size(mgr, oracle):
    """C3 with SUM(val), COUNT(*), MAX(dval): not the compile-time form of the headline query, same sliced passes."""
    import torch
    w = _workload("c3m", mgr)
    cp = w.compiled
    out = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    assert names.startswith("hdk_join_order_probe,hdk_join_scatter_slices,hdk_join_agg_sliced2"), names
    ref = w.reference_checks()
    got = out.cpu().numpy()
    assert int(got[0]) % (1 << 64) == ref["sum_val"] and int(got[1]) == w.rows and int(got[2]) == ref["max_dval"]
    _oracle_sample(w, oracle)


def test_c2_at_baseline_size(mgr, oracle):
    import torch
    from hdk_amd.executor import ExecutionResult
    w = _workload("c2", mgr)
    cp = w.compiled
    out = torch.empty(cp.buffer_quads, dtype=torch.int64, device="cuda")
    assert _run_into(w, out).startswith("hdk_scan_agg_direct")
    cols = ExecutionResult(cp, out.cpu().numpy(), cp.entry_count).to_columns()
    assert sorted(cols["key"]) == list(range(64))
    assert sum(cols["s"]) % (1 << 64) == w.reference_checks()["sum_val"]
    first = out.clone()
    _run_into(w, out)
    assert bool(torch.equal(first, out))
    _oracle_sample(w, oracle)


@pytest.mark.parametrize("name", ["q1", "q2", "q3", "q4"])
def test_taxi_queries_at_baseline_size(name, mgr, oracle):
    """Taxi Q1-Q4 over 1 B synthetic rows: counts add up to the row count, Q1/Q2 groups equal torch.bincount /
    index_add of the key columns, the specialised kernel and the batched interpreter give the identical buffer."""
    import torch
    from hdk_amd.executor import ExecutionResult
    w = _workload(name, mgr)
    cp = w.compiled
    out = torch.empty(cp.buffer_quads, dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    assert names.split(",")[0] in ("hdk_scan_agg_direct", "hdk_scan_agg_keys"), names
    fast = out.clone()
    cols = ExecutionResult(cp, out.cpu().numpy(), cp.entry_count).to_columns()
    ref = w.reference_checks()
    if name == "q1":
        got = dict(zip(cols["cab_type"], cols["cnt"]))
        assert got == {"green": ref["key_counts"][0], "yellow": ref["key_counts"][1]}
    elif name == "q2":
        for k, avg in zip(cols["passenger_count"], cols["avg_amount"]):
            want = ref["key_sums"][k] / ref["key_counts"][k] / 100.0  # decimal(14,2)
            assert avg == pytest.approx(want, rel=1e-12)
        assert len(cols["passenger_count"]) == 7
    else:
        assert sum(cols["cnt"]) == w.local_rows
        per_pc = {}
        for k, c in zip(cols["passenger_count"], cols["cnt"]):
            per_pc[k] = per_pc.get(k, 0) + c
        assert [per_pc.get(k, 0) for k in range(8)] == ref["key_counts"]
        assert set(cols["year"]) == set(range(2009, 2016))
    _run_into(w, out, flags=A.LAUNCH_FORCE_GENERIC)
    assert bool(torch.equal(fast, out))
    _oracle_sample(w, oracle)


@pytest.mark.parametrize("name,kernel", [("nga2", "hdk_scan_agg_cols"), ("msbs1", "hdk_scan_agg_bhm"), ("msphs1", "hdk_scan_agg_bhm"),
                                         ("phm2", "hdk_scan_agg_bhm"), ("msphs1w", "hdk_scan_agg_bhm"), ("msphs1f", "hdk_scan_agg_bhm")])
def test_suite_families_at_one_billion_rows(name, kernel, mgr, oracle):
    """The reference's NonGroupedAgg / MultiStep / PerfectHashMultiCol benchmark shapes at 1 B rows on the kernels round 6 gave
    them (scan_agg_cols.h, scan_bhm.h; MSPHS001 also over BIGINT columns and behind a filter): every group and every target against torch.bincount / index_add_ / scatter_reduce_
    over the same columns, idempotence, and the oracle on a 2 M-row sample."""
    import torch
    from hdk_amd.executor import ExecutionResult
    w = _workload(name, mgr)
    cp = w.compiled
    out = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device="cuda")
    names = _run_into(w, out)
    assert names.split(",")[0] == kernel, names
    ref = w.reference_checks()["syn"]
    assert sum(ref["count"]) == w.local_rows or w.query.quals  # (msphs1f: the rows that pass its filter)
    cols = ExecutionResult(cp, out.cpu().numpy(), cp.entry_count).to_columns()
    assert w.check_syn(cols, ref)
    _run_into(w, out)
    assert w.check_syn(ExecutionResult(cp, out.cpu().numpy(), cp.entry_count).to_columns(), ref)
    _oracle_sample(w, oracle)
