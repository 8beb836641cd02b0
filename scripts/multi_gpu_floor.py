#!/usr/bin/env python3
"""What ONE rank of an 8-GPU job does per step, measured on one device (no multi-GPU box is available to the
build; the wire is the only piece left out).

  part A  perfect-hash / non-grouped configs (c2, q1..q4): rank 0 of 8 holds fragments 0, 8, 16, 24 of the 1 B-row
          table (128 M rows).  Step = init + scan + finalize + [all-gather emulated by a device copy of the partial
          into 8 slots] + fold of 7 partials.  Reported: GPU ms per step (HIP events), host enqueue time per step,
          and the same with the plan resident in the workspace (HDK_HIP_LAUNCH_PLAN_RESIDENT).
  part B  C5 (open addressing), the table-exchange merge of round 2: every rank aggregates its shard into a
          200 M-entry table, splits the non-empty entries by owner, owners re-insert.  The eight ranks are run one
          after another; owner 0's inbox is assembled from their segments.  Reported per piece: shard scan,
          partition count (+ host sync), partition scatter, owner-table init, owner re-insert.

  part C  C5, the tuple exchange (hdk_hip_scatter_to_owners / hdk_hip_aggregate_from_ranks): every rank scatters its
          shard straight into per-owner segments, the all-to-all is emulated by device copies, every owner
          aggregates its inbox into a table sized for its keys.  Reported: scatter ms per rank, aggregate ms per
          owner, bytes on the wire, and the full-size checks (sum of sums, number of groups, owners disjoint).

    python scripts/multi_gpu_floor.py [--only a,b,c] [--world 8] [--steps 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ev_ms(torch, stream, fn, reps=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def part_a(args, mgr, quiet=False):
    import torch
    from hdk_amd import _abi as A
    from hdk_amd import distributed as D
    from hdk_amd._lib import check, lib
    from workloads import CONFIGS, Workload, fragment_rows
    from bench import c2_numpy_generator
    L = lib()
    world, dev = args.world, 0
    out = {}
    for name in [c for c in ("c2", "q1", "q2", "q3", "q4") if c in args.configs]:
        rows = CONFIGS[name][0]
        nfrag = len(fragment_rows(rows))
        frag_ids = D.shard_fragments(nfrag, world, 0)
        w = Workload(name, rows, dev, mgr, frag_ids=frag_ids,
                     generators=c2_numpy_generator(torch, torch.device("cuda", dev), 0) if name == "c2" else None)
        cp = w.compiled
        quads = max(cp.buffer_quads, 1)
        ts = torch.cuda.Stream(device=dev)
        h = ts.cuda_stream
        out_t = torch.empty(quads, dtype=torch.int64, device="cuda")
        gathered = torch.empty(world * quads, dtype=torch.int64, device="cuda")
        d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
        iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
        that = (C.c_void_p * (world - 1))(*[gathered.data_ptr() + i * quads * 8 for i in range(1, world)])
        counts = (C.c_uint32 * (world - 1))(*([cp.entry_count] * (world - 1)))
        res = {}
        for label, flags in (("plan_uploaded", 0), ("plan_resident", A.LAUNCH_PLAN_RESIDENT)):
            step = w.ex.prepare(cp, w.frag_ids, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=out_t.data_ptr())
            if flags:
                step.enqueue(h)  # the first launch puts the plan into the workspace head
                torch.cuda.synchronize()
                step.ko.flags |= flags

            def one():
                step.enqueue(h)
                gathered.view(world, quads).copy_(out_t)  # stands in for all_gather_into_tensor
                check(L.hdk_hip_reduce_buffers(C.byref(cp.plan), gathered.data_ptr(), cp.entry_count, that, counts,
                                               world - 1, iv.ctypes.data, d_err.data_ptr(), dev, h))

            with torch.cuda.stream(ts):
                for _ in range(3):
                    one()
                torch.cuda.synchronize()
                n_ev = C.c_int32(0)
                check(L.hdk_hip_collect_scan_times(dev, None, 0, C.byref(n_ev)))
                t0 = time.perf_counter()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts)
                for _ in range(args.steps):
                    one()
                e1.record(ts)
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                t_wall = time.perf_counter() - t0
            buf = (C.c_float * args.steps)()
            check(L.hdk_hip_collect_scan_times(dev, buf, args.steps, C.byref(n_ev)))
            scan = float(np.mean([buf[i] for i in range(min(n_ev.value, args.steps))]))
            res[label] = {"gpu_ms_per_step": e0.elapsed_time(e1) / args.steps, "scan_kernel_ms": scan,
                          "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "wall_ms_per_step": t_wall / args.steps * 1e3}
            step.free()
        res["rows_per_rank"] = w.local_rows
        res["projected_rows_per_s_at_8_gpus_wire_excluded"] = rows / (res["plan_resident"]["wall_ms_per_step"] * 1e-3)
        out[name] = res
        if not quiet:
            print(json.dumps({name: res}), flush=True)
        del w, step
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def part_b(args, mgr):
    import torch
    from hdk_amd import _abi as A
    from hdk_amd import distributed as D
    from hdk_amd._lib import check, lib
    from workloads import Workload, fragment_rows
    L = lib()
    world, dev = args.world, 0
    rows = 1_000_000_000
    nfrag = len(fragment_rows(rows))
    ts = torch.cuda.Stream(device=dev)
    h = ts.cuda_stream
    segs, seg_counts, per_rank = [], [], []
    cp = None
    for r in range(world if not args.one_rank else 1):
        w = Workload("c5", rows, dev, mgr, frag_ids=D.shard_fragments(nfrag, world, r))
        cp = w.compiled
        quads = cp.buffer_quads
        out_t = torch.empty(quads, dtype=torch.int64, device="cuda")
        step = w.ex.prepare(cp, w.frag_ids, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=out_t.data_ptr())
        with torch.cuda.stream(ts):
            step.enqueue(h)
            torch.cuda.synchronize()
            scan = ev_ms(torch, ts, lambda: step.enqueue(h), 3)
            iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
            counts = (C.c_uint32 * world)()
            t0 = time.perf_counter()
            check(L.hdk_hip_partition_baseline_count(C.byref(cp.plan), out_t.data_ptr(), cp.entry_count, iv.ctypes.data,
                                                     world, counts, dev, h))
            t_count = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            send, cnt = D.partition_baseline_on_device(cp, out_t, world, dev, h)
            torch.cuda.synchronize()
            t_part_both = (time.perf_counter() - t0) * 1e3
        q = [D.baseline_table_quads(cp, int(c)) for c in cnt]
        offs = np.concatenate([[0], np.cumsum(q)])
        segs.append(send[int(offs[0]):int(offs[1])].clone())
        seg_counts.append(int(cnt[0]))
        per_rank.append({"rank": r, "rows": w.local_rows, "shard_scan_ms": scan, "groups": int(cnt.sum()),
                         "partition_count_with_host_sync_ms": t_count, "partition_count_plus_scatter_ms": t_part_both,
                         "bytes_to_other_owners": int(sum(q[1:])) * 8})
        print(json.dumps(per_rank[-1]), flush=True)
        step.free()
        del w, step, send, out_t
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    if args.one_rank:
        segs, seg_counts = segs * world, seg_counts * world
    recv = torch.cat(segs + [torch.zeros(1, dtype=torch.int64, device="cuda")])
    rc = np.array(seg_counts, dtype=np.uint32)
    res = {"per_rank": per_rank, "owner0_inbox_entries": int(rc.sum())}
    for label, n_owner in (("owner_table_200M_entries", None), ("owner_table_sized_to_its_keys", cp.entry_count // world)):
        with torch.cuda.stream(ts):
            D.merge_baseline_on_device(cp, recv, rc, dev, n_owner, h)  # warm-up (pool growth)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            table, ne = D.merge_baseline_on_device(cp, recv, rc, dev, n_owner, h)
            torch.cuda.synchronize()
            res[label] = {"init_plus_reinsert_ms": (time.perf_counter() - t0) * 1e3, "entries": ne}
            rq = int(cp.plan.row_size_quad)
            keys = (table[:ne * rq].view(ne, rq)[:, 0] << 32) >> 32 if cp.plan.key_width == 4 else table[:ne * rq].view(ne, rq)[:, 0]
            empty = A.EMPTY_KEY_32 if cp.plan.key_width == 4 else A.EMPTY_KEY_64
            res[label]["groups"] = int((keys != empty).sum().item())
        del table
        torch.cuda.empty_cache()
    print(json.dumps({"c5_table_exchange_merge": res}), flush=True)
    return res


def part_c(args, mgr, quiet=False):
    import torch
    from hdk_amd import _abi as A
    from hdk_amd import distributed as D
    from workloads import Workload, fragment_rows
    world, dev = args.world, 0
    rows = 1_000_000_000
    nfrag = len(fragment_rows(rows))
    frs = fragment_rows(rows)
    bound = max(sum(frs[f] for f in D.shard_fragments(nfrag, world, r)) for r in range(world))
    ts = torch.cuda.Stream(device=dev)
    h = ts.cuda_stream
    xs, tables, per_rank, want_sum, keys_seen = [], [], [], 0, []
    for r in range(world):
        w = Workload("c5", rows, dev, mgr, frag_ids=D.shard_fragments(nfrag, world, r))
        cp = w.compiled
        probe = w.ex.prepare(cp, w.frag_ids)
        x = D.TupleExchange(probe, world, bound)
        table = torch.empty(x.owner_table_quads, dtype=torch.int64, device="cuda")
        probe.free()
        step = w.ex.prepare(cp, w.frag_ids, out_ptr=table.data_ptr())
        x = D.TupleExchange(step, world, bound)
        with torch.cuda.stream(ts):
            x.scatter(h)
            torch.cuda.synchronize()
            ms = ev_ms(torch, ts, lambda: x.scatter(h), args.steps)
        want_sum = (want_sum + w.reference_checks()["sum_val"]) % (1 << 64)
        keys_seen.append(torch.unique(torch.cat([w.cols[w.key_col][f] for f in w.frag_ids])))
        per_rank.append({"rank": r, "rows": w.local_rows, "scatter_ms": ms, "tuple_bytes": int(x.shape.tuple_bytes),
                         "segment_bytes": int(x.shape.segment_bytes), "bytes_to_other_owners": x.bytes_sent_per_rank,
                         "coarse_per_owner": int(x.shape.coarse_per_owner)})
        if not quiet:
            print(json.dumps(per_rank[-1]), flush=True)
        xs.append(x)
        tables.append(table)
        # the inputs are not needed any more: only the send buffers are
        w.ex.cache.clear() if hasattr(w.ex.cache, "clear") else None
        for k in list(w.cols):
            w.cols[k] = {}
        del w
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    distinct = int(torch.unique(torch.cat(keys_seen)).numel())
    del keys_seen
    per_owner, got_sum, groups, all_keys = [], 0, 0, []
    for o in range(world):
        x = xs[o]
        with torch.cuda.stream(ts):
            for r in range(world):
                x.segment(x.recv, r).copy_(xs[r].segment(xs[r].send, o))
            x.aggregate(h)
            torch.cuda.synchronize()
            ms = ev_ms(torch, ts, lambda: x.aggregate(h), args.steps)
        err = int(x.step.mgr.to_host(x.step.d_err.ptr, 4, dev, np.int32)[0])
        cp = x.step.cp
        rq, ne = int(cp.plan.row_size_quad), x.owner_entries
        rows_t = tables[o][:ne * rq].view(ne, rq)
        keys = (rows_t[:, 0] << 32) >> 32 if cp.plan.key_width == 4 else rows_t[:, 0]
        live = keys != (A.EMPTY_KEY_32 if cp.plan.key_width == 4 else A.EMPTY_KEY_64)
        sq = int(cp.plan.targets[1].slot_off) // 8
        got_sum = (got_sum + int(rows_t[:, sq][live].sum().item())) % (1 << 64)
        groups += int(live.sum().item())
        all_keys.append(keys[live])
        per_owner.append({"owner": o, "aggregate_ms": ms, "groups": int(live.sum().item()), "entries": ne, "error": err})
        if not quiet:
            print(json.dumps(per_owner[-1]), flush=True)
    allk = torch.cat(all_keys)
    res = {"per_rank": per_rank, "per_owner": per_owner,
           "step_ms_wire_excluded": max(p["scatter_ms"] for p in per_rank) + max(p["aggregate_ms"] for p in per_owner),
           "checks": {"sum_of_sums": got_sum == want_sum, "groups": groups, "groups_equal_distinct_keys": groups == distinct,
                      "owners_disjoint": int(torch.unique(allk).numel()) == int(allk.numel())}}
    res["projected_rows_per_s_at_%d_gpus_wire_excluded" % world] = rows / (res["step_ms_wire_excluded"] * 1e-3)
    if not quiet:
        print(json.dumps({"c5_tuple_exchange": {k: v for k, v in res.items() if k not in ("per_rank", "per_owner")}}), flush=True)
    return res


def part_d(args, mgr, quiet=False, link_gbps=64.0, nchunks=4):
    """The exchange PIPELINE of one rank of an 8-GPU C5 job, emulated on one device (VERDICT r4 task 6): the rank's rows in
    `nchunks` chunks, scatter(k + 1) / wire(k) / aggregate(k - 1) on three streams (hdk_amd.distributed.ChunkedTupleExchange).
    The wire of chunk k is a device copy of its segments plus a spin of segment_bytes / `link_gbps` -- the modelled time of
    one all-to-all over point-to-point xGMI links, every owner's segment on its own link; the owner's inbox is the rank's own
    segment for owner 0 eight times over (the tuple count of the real inbox, the rank's own keys).  Reports the serial step
    (stage after stage, as bench.py's N > 1 step runs today) beside the overlapped one and max(compute, wire)."""
    import torch
    from hdk_amd import distributed as D
    from workloads import Workload, fragment_rows
    world, dev, rows = args.world, 0, 1_000_000_000
    frs = fragment_rows(rows)
    mine = D.shard_fragments(len(frs), world, 0)
    w = Workload("c5", rows, dev, mgr, frag_ids=mine)
    cp = w.compiled
    chunks = [mine[i * len(mine) // nchunks:(i + 1) * len(mine) // nchunks] for i in range(nchunks)]
    chunks = [c for c in chunks if c]
    bounds = [max(sum(frs[f] for f in D.shard_fragments(len(frs), world, r)[i * len(mine) // nchunks:(i + 1) * len(mine) // nchunks])
                  for r in range(world)) for i in range(len(chunks))]
    probe = w.ex.prepare(cp, chunks[0])
    x0 = D.TupleExchange(probe, world, bounds[0])
    table = torch.empty(x0.owner_table_quads, dtype=torch.int64, device="cuda")
    probe.free()
    steps = [w.ex.prepare(cp, c, out_ptr=table.data_ptr()) for c in chunks]
    pipe = D.ChunkedTupleExchange(steps, world, bounds)
    # cycles of torch.cuda._sleep per millisecond
    ts = torch.cuda.current_stream()
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    torch.cuda._sleep(20_000_000)
    e1.record(ts)
    torch.cuda.synchronize()
    cyc_per_ms = 20_000_000 / e0.elapsed_time(e1)
    wire_ms = [int(c.shape.segment_bytes) / (link_gbps * 1e9) * 1e3 for c in pipe.chunks]

    def wire(k, c, stream):
        with torch.cuda.stream(stream):
            torch.cuda._sleep(int(wire_ms[k] * cyc_per_ms))
            for r in range(world):
                c.segment(c.recv, r).copy_(c.segment(c.send, 0), non_blocking=True)

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None or dt < best else best
        return best

    cur = torch.cuda.current_stream()
    overlapped = wall(lambda: pipe.run(wire, after=cur), args.steps)
    errs = pipe.error_codes()

    def serial():
        h = cur.cuda_stream
        for k, c in enumerate(pipe.chunks):
            c.scatter(h)
        for k, c in enumerate(pipe.chunks):
            wire(k, c, cur)
        for k, c in enumerate(pipe.chunks):
            c.aggregate(h, accumulate=k > 0)

    serial_ms = wall(serial, args.steps)
    h = cur.cuda_stream
    scatter_ms = wall(lambda: [c.scatter(h) for c in pipe.chunks], args.steps)
    agg_ms = wall(lambda: [c.aggregate(h, accumulate=k > 0) for k, c in enumerate(pipe.chunks)], args.steps)
    res = {"chunks": len(chunks), "link_gbps_assumed": link_gbps, "rows_per_rank": w.local_rows,
           "scatter_ms": scatter_ms, "aggregate_ms": agg_ms, "compute_ms": scatter_ms + agg_ms, "wire_ms_modelled": sum(wire_ms),
           "step_ms_serial": serial_ms, "step_ms_overlapped": overlapped,
           "overlapped_over_max_of_compute_and_wire": overlapped / max(scatter_ms + agg_ms, sum(wire_ms)),
           "chunk_error_codes": errs,
           "projected_rows_per_s_at_%d_gpus" % world: rows / (overlapped * 1e-3)}
    if not quiet:
        print(json.dumps({"c5_exchange_pipeline": res}), flush=True)
    for s_ in steps:
        s_.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="a,b")
    ap.add_argument("--configs", default="c2,q1,q2,q3,q4")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--chunks", type=int, default=4, help="part D: chunks of the exchange pipeline")
    ap.add_argument("--one-rank", action="store_true", help="part B: run rank 0 only and use its segment 8 times")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    args.configs = args.configs.split(",")
    import torch
    torch.cuda.set_device(0)
    from hdk_amd.hip_mgr import HipMgr
    mgr = HipMgr()
    out = {}
    if "a" in args.only:
        out["perfect_hash_shard_floor"] = part_a(args, mgr)
    if "b" in args.only:
        out["c5_table_exchange_merge"] = part_b(args, mgr)
    if "c" in args.only:
        out["c5_tuple_exchange"] = part_c(args, mgr)
    if "d" in args.only:
        out["c5_exchange_pipeline"] = part_d(args, mgr, nchunks=args.chunks)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
