"""world_size-2 (gloo, CPU) coverage of the N>1 path's host logic: fragment sharding, the all-gather of
partial tables in rank order, and the fold of the gathered partials.  The partial tables are produced
by the oracle (tests may use it), the collective is the real torch.distributed call; the GPU merge
kernel itself is covered by tests/test_gpu_primitives.py::test_device_reduce_perfect_hash."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_fragments():
    from hdk_amd.distributed import shard_fragments
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += shard_fragments(32, world, r)
        assert sorted(seen) == list(range(32))
    assert shard_fragments(5, 2, 1) == [1, 3]
    with pytest.raises(ValueError):
        shard_fragments(4, 2, 2)


def _worker(rank, world, port, result_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hdk_amd import _abi as A
        from hdk_amd.distributed import all_gather_partials, merge_gathered, shard_fragments
        from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
        from hdk_amd.plan import compile_query
        from hdk_amd.storage import ArrowStorage
        from oracle import oracle as O
        from util import host_fragments, oracle_init_buffer, run_oracle
        rng = np.random.default_rng(77)  # same table on every rank; each rank scans only its fragments
        n = 60_000
        v = rng.integers(-1000, 1000, n).astype(np.int64)
        v[rng.random(n) < 0.2] = A.NULL_BIGINT
        st = ArrowStorage()
        st.import_numpy("t", {"k": rng.integers(0, 40, n).astype(np.int64), "v": v, "f": rng.normal(size=n)},
                        fragment_size=7_000)
        q = QueryUnit("t", groupby=[ColRef("k")],
                      targets=[KeyRef(0), Agg("sum", ColRef("v")), Agg("count"), Agg("min", ColRef("v")),
                               Agg("avg", ColRef("f"))])
        cp = compile_query(st, q)
        mine = shard_fragments(st.get("t").num_fragments, world, rank)
        local = oracle_init_buffer(O, cp)
        assert O.run_plan(cp.plan, host_fragments(O, st, cp, mine), local) == 0
        gathered = all_gather_partials(torch.from_numpy(local), world).numpy()
        merged = merge_gathered(cp, gathered, world, O.reduce)
        # every rank must hold the same, complete result: compare with a single-process run
        _, full, err = run_oracle(O, st, cp)
        from util import assert_buffers_equal
        assert_buffers_equal(cp, merged, full)
        # and the fold order is rank order: merging by hand gives the identical bits
        byhand = gathered[:cp.buffer_quads].copy()
        for r in range(1, world):
            O.reduce(cp.plan, byhand, cp.entry_count, gathered[r * cp.buffer_quads:(r + 1) * cp.buffer_quads],
                     cp.entry_count, cp.init_vals)
        assert np.array_equal(byhand, merged)
        result_q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        result_q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_merge_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _baseline_worker(rank, world, port, result_q, nkeys=3000):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hdk_amd import result_set as rs
        from hdk_amd.distributed import baseline_table_quads, exchange_owner_segments, shard_fragments
        from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
        from hdk_amd.plan import compile_query
        from hdk_amd.storage import ArrowStorage
        from oracle import oracle as O
        from util import host_fragments, oracle_init_buffer, run_oracle
        rng = np.random.default_rng(78)
        n = 50_000
        st = ArrowStorage()
        st.import_numpy("t", {"k": rng.integers(0, nkeys, n).astype(np.int64) * (1_000_003 if nkeys > 100 else 2**35 + 11),
                              "v": rng.integers(-1000, 1000, n).astype(np.int64)}, fragment_size=6_000)
        q = QueryUnit("t", groupby=[ColRef("k")], force_baseline=True, baseline_entry_count=8_191,
                      targets=[KeyRef(0, "k"), Agg("sum", ColRef("v"), "s"), Agg("count", None, "c")])
        cp = compile_query(st, q)
        p = cp.plan
        assert not p.output_columnar and p.key_width == 8
        rq = p.row_size_quad
        assert baseline_table_quads(cp, 10) == 10 * rq  # host-only ABI call: works without a GPU
        mine = shard_fragments(st.get("t").num_fragments, world, rank)
        local = oracle_init_buffer(O, cp)
        assert O.run_plan(p, host_fragments(O, st, cp, mine), local) == 0
        # host stand-in for hdk_hip_partition_baseline (the HIP kernel is covered by the GPU test):
        # non-empty rows, owner = mulhi32(key_hash, G), compact row-wise segments in owner order
        rows = local.view(np.int64).reshape(p.entry_count, rq)
        rows = rows[rows[:, 0] != np.iinfo(np.int64).max]
        L = O.lib()
        owners = np.array([(L.orc_key_hash(np.array([k], dtype=np.int64).ctypes.data, 1, 8) * world) >> 32
                           for k in rows[:, 0]], dtype=np.int64)
        counts = np.array([(owners == o).sum() for o in range(world)], dtype=np.uint32)
        send = torch.from_numpy(np.concatenate([rows[owners == o].reshape(-1) for o in range(world)] +
                                               [np.zeros(1, dtype=np.int64)]))
        recv, recv_counts = exchange_owner_segments(cp, send, counts, world, rank)
        recv = recv.numpy()
        owner_table = oracle_init_buffer(O, cp)
        off = 0
        for c in recv_counts:
            seg = np.ascontiguousarray(recv[off:off + int(c) * rq])
            if c:
                assert O.reduce(p, owner_table, p.entry_count, seg, int(c), cp.init_vals) == 0
            off += int(c) * rq
        got = rs.to_columns(cp, owner_table)
        # the whole query on one process, restricted to the keys this rank owns
        _, full, err = run_oracle(O, st, cp)
        want = rs.to_columns(cp, full)
        want_mine = {}
        for k, s, c in zip(want["k"], want["s"], want["c"]):
            if (L.orc_key_hash(np.array([k], dtype=np.int64).ctypes.data, 1, 8) * world) >> 32 == rank:
                want_mine[k] = (s, c)
        assert {k: (s, c) for k, s, c in zip(got["k"], got["s"], got["c"])} == want_mine
        tot = torch.tensor([len(got["k"]), 1 if len(got["k"]) == 0 else 0])
        dist.all_reduce(tot)
        assert int(tot[0].item()) == len(want["k"])
        if nkeys < world:
            assert int(tot[1].item()) >= world - nkeys  # owners that received nothing still took part in the exchange
        result_q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        result_q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, port, *extra):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_two_rank_baseline_owner_exchange_gloo():
    _spawn(_baseline_worker, 2, 31500 + (os.getpid() % 2000))


def test_four_rank_owner_exchange_uneven_splits_and_empty_owners_gloo():
    """World size 4, 9 fragments (ranks hold 3/2/2/2), uneven all_to_all splits; with only 3 distinct keys at least
    one owner receives nothing and must still come out of the exchange with an empty table."""
    _spawn(_baseline_worker, 4, 33500 + (os.getpid() % 2000))
    _spawn(_baseline_worker, 4, 35500 + (os.getpid() % 2000), 3)


def test_four_rank_gather_and_merge_gloo():
    _spawn(_worker, 4, 37500 + (os.getpid() % 2000))


def _segments_worker(rank, world, port, result_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hdk_amd.distributed import exchange_equal_segments, owner_entry_count_for
        seg = 4096 + 256  # bytes of a segment: [header | slabs], the same on every rank
        send = torch.empty(world * seg, dtype=torch.uint8)
        for o in range(world):  # segment o carries (sender, owner) in every byte pair
            send[o * seg:(o + 1) * seg] = torch.tensor([rank, o] * (seg // 2), dtype=torch.uint8)
        recv = torch.zeros(world * seg, dtype=torch.uint8)
        exchange_equal_segments(send, recv, world)
        for r in range(world):  # what rank r scattered for THIS owner sits in segment r
            got = recv[r * seg:(r + 1) * seg].view(-1, 2)
            assert bool((got[:, 0] == r).all()) and bool((got[:, 1] == rank).all()), (rank, r)
        with pytest.raises(ValueError):
            exchange_equal_segments(send[:-1], recv[:-1], world)
        assert owner_entry_count_for(200_000_000, 8) == 25_000_000 and owner_entry_count_for(10, 4) == 1024
        result_q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        result_q.put((rank, "fail: " + traceback.format_exc()))
        raise e
    finally:
        dist.destroy_process_group()


def test_tuple_exchange_routes_equal_segments_gloo():
    """The collective of the tuple exchange at world size 2 and 4: segment o of rank r arrives as segment r of rank o."""
    _spawn(_segments_worker, 2, 39500 + (os.getpid() % 2000))
    _spawn(_segments_worker, 4, 41500 + (os.getpid() % 2000))


def test_exchange_cost_model_counts_bytes_the_right_way():
    """1 B rows over 100 M keys: a rank of TWO holds 500 M rows = 5 rows per group, so pre-aggregated entries are fewer
    bytes than tuples at G = 2 and 4 and about as many at G = 8 (round-3 verdict: DESIGN.md had this backwards); what the
    model picks also prices the local aggregation, the owner partition and the re-insert."""
    from hdk_amd.distributed import choose_open_addressing_exchange as model
    m2, m4, m8 = (model(g, 1_000_000_000 // g, 100_000_000) for g in (2, 4, 8))
    assert 4.9 < m2["rows_per_group_and_rank"] < 5.1 and 2.6 < m4["rows_per_group_and_rank"] < 2.8 and 1.7 < m8["rows_per_group_and_rank"] < 1.8
    assert m2["tables_wire_bytes"] < 0.5 * m2["tuples_wire_bytes"]
    assert m4["tables_wire_bytes"] < m4["tuples_wire_bytes"]
    assert m8["tables_wire_bytes"] > m8["tuples_wire_bytes"]
    assert m2["mode"] == "tables" and m8["mode"] == "tuples"
    # a faster wire moves the break-even: with nothing to pay for bytes the tuple exchange's cheaper compute wins everywhere
    assert model(2, 500_000_000, 100_000_000, link_gbps=1e6)["mode"] == "tuples"
    # few keys: a table exchange ships almost nothing
    assert model(8, 125_000_000, 1000)["tables_wire_bytes"] < 1e6
