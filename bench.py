#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: rows/sec on a 1 B-row, 64-key int64
`SELECT key, SUM(val) FROM t GROUP BY key` (config C2) at 1/2/4/8 GPUs, plus % of the HBM roofline and the
HDK-semantics CPU path timed beside it.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c5|c5s|q1..q4] [--scaling strong|weak]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over the rank's resident fragments: output-buffer init, the multi-fragment
scan/aggregate launch (every pass of it), finalize and -- for N > 1 -- the partial-aggregate merge over RCCL/xGMI:
all-gather of the per-GPU tables + device fold (perfect hash / non-grouped); open addressing (C5) exchanges TUPLES
instead of tables: every rank scatters its rows into per-owner segments, one all-to-all with equal splits, every
owner aggregates what it received (partial tables + owner re-insert remain as the fallback for skewed keys).
Inputs are resident in HBM before the timed region (as HDK keeps chunks cached in GpuBufferMgr).

Scaling: STRONG by default -- ONE table of `--rows` rows in 32 M-row fragments, fragment f on rank
f mod N, value = table rows / max-over-ranks step time.  `--scaling weak` gives every rank its own table.
At N = 1 the line also carries, under "configs", the other BASELINE configurations at BASELINE size (C3 join probe,
C5 open addressing and its per-GPU shard shape, taxi Q1-Q4), each with its own roofline object, and under
"multi_gpu_emulated" the per-rank step of an 8-GPU job measured on this one device (wire excluded).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


class Comm:
    """torch.distributed with a host-staged test mode (HDK_BENCH_BACKEND=gloo) for boxes with fewer GPUs than ranks."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:  # (main() has already started the ranks itself when no launcher did)
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: refusing to print a line for another job size")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
        self.backend = os.environ.get("HDK_BENCH_BACKEND", "nccl")
        self.dev = self.local_rank if self.backend == "nccl" else self.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(self.dev)
        # HDK_BENCH_SINGLE_RANK_COLLECTIVES=1: a ONE-rank job still initialises the process group and runs the N > 1 step
        # (all-gather + fold, tuple exchange + all-to-all) -- the only way to drive the RCCL branch on a one-GPU box
        self.multi = self.world > 1 or os.environ.get("HDK_BENCH_SINGLE_RANK_COLLECTIVES") == "1"
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev))
            else:
                dist.init_process_group(self.backend)
        self.group = None if self.backend == "nccl" else "host"

    def all_gather(self, dst, src):
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(dst, src)
        else:
            parts = [self.torch.empty(src.numel(), dtype=src.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, src.cpu())
            dst.copy_(self.torch.cat(parts))

    def all_reduce(self, t, op="sum"):
        if not self.multi:
            return t
        o = self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=o)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=o)
            t.copy_(h)
        return t

    def barrier(self):
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if self.multi:
            self.dist.destroy_process_group()


def c2_numpy_generator(torch, dev, frag_offset=0):
    """C2's inputs as BASELINE.md section 2 fixes them: numpy PCG64(SEED + global fragment index) per fragment."""
    from workloads import SEED

    def gen(col):
        def g(f, n):
            rng = np.random.Generator(np.random.PCG64(SEED + frag_offset + f))
            key = rng.integers(0, 64, n, dtype=np.int64)
            if col == "key":
                return torch.from_numpy(key).to(dev)
            return torch.from_numpy(rng.integers(-2**31, 2**31, n, dtype=np.int64)).to(dev)
        return g
    return gen


def run_config(name, args, comm, mgr, steps, warmup, primary):
    """Build the workload, time `steps` steps, check the result at full size.  Returns the dict of one bench line."""
    import torch
    from hdk_amd import _abi as A
    from hdk_amd._lib import check, lib
    from hdk_amd import distributed as D
    from hdk_amd.executor import ExecutionResult
    from workloads import CONFIGS, Workload
    L = lib()
    world, rank, dev = comm.world, comm.rank, comm.dev
    multi = comm.multi  # (world > 1, or one rank told to run the N > 1 step)
    rows = int(args.rows) if (args.rows and primary) else CONFIGS[name][0]
    strong = args.scaling == "strong"
    nfrag_total = len(__import__("workloads").fragment_rows(rows))
    frag_ids = D.shard_fragments(nfrag_total, world, rank) if (strong and world > 1) else None
    t_gen = time.perf_counter()
    kw = {"dim_rows": int(args.dim_rows)} if (args.dim_rows and name.startswith("c3")) else {}
    w = Workload(name, rows, dev, mgr, frag_ids=frag_ids, seed_offset=0 if strong else rank * 100_003, **kw,
                 generators=c2_numpy_generator(torch, torch.device("cuda", dev), 0 if strong else rank * nfrag_total)
                 if name == "c2" else None)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    cp = w.compiled
    quads = max(cp.buffer_quads, 1)
    baseline = cp.plan.query_kind == A.Q_BASELINE_HASH

    # One explicit (non-default) stream carries everything: the library treats a NULL stream as "my own stream", which
    # would not be ordered with torch's default stream and hence with the RCCL collectives.
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    out_t = torch.empty(quads, dtype=torch.int64, device="cuda")
    step = w.ex.prepare(cp, w.frag_ids, grid=args.grid if primary else 0, flags=A.LAUNCH_RECORD_EVENTS | int(os.environ.get("HDK_BENCH_FLAGS", "0")),
                        out_ptr=out_t.data_ptr())
    gathered = torch.empty(world * quads, dtype=torch.int64, device="cuda") if (multi and not baseline) else None
    d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
    init_vals = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
    merge_ms, owner = [], {}
    # ---- N > 1 ----------------------------------------------------------------------------------------------------
    xch, xch_ev, mode = None, [], "single"
    if multi and not baseline:
        mode = "gather+fold"
        # (hoisted out of the step: the argument arrays of the fold; the plan stays resident in the workspace after
        # the first launch, so a step uploads nothing)
        that = (C.c_void_p * (world - 1))(*[gathered.data_ptr() + i * quads * 8 for i in range(1, world)])
        counts = (C.c_uint32 * (world - 1))(*([cp.entry_count] * (world - 1)))
    if multi and baseline:
        # open addressing: TUPLES go to their owner rank (hdk_hip_scatter_to_owners -> one all-to-all with equal
        # splits -> hdk_hip_aggregate_from_ranks); skewed keys fall back to the exchange of partial tables
        bound = torch.tensor([w.local_rows], dtype=torch.int64, device="cuda")
        comm.all_reduce(bound, "max")
        # (the shape only depends on the plan, the row bound and -- through the tuple width -- on the column statistics;
        # the ranks settle the width among themselves inside TupleExchange, and the choice of the mode is made by ALL
        # ranks together: a rank on its own in "tables" mode would issue other collectives than its peers)
        step.free()
        n_owner = D.owner_entry_count_for(cp.entry_count, world)
        owner_t = torch.empty(D.baseline_table_quads(cp, n_owner), dtype=torch.int64, device="cuda")
        step = w.ex.prepare(cp, w.frag_ids, flags=int(os.environ.get("HDK_BENCH_FLAGS", "0")), out_ptr=owner_t.data_ptr())
        why = ""
        # tuples or pre-aggregated tables: HDK_BENCH_EXCHANGE=tuples|tables forces one; "auto" (default) asks the cost model
        # (measured compute rates + an assumed link rate, hdk_amd/distributed.py: the same answer on every rank)
        pick = os.environ.get("HDK_BENCH_EXCHANGE", "auto")
        owner["model"] = D.choose_open_addressing_exchange(world, int(bound.item()), getattr(w, "key_domain", cp.entry_count // 2))
        if pick == "auto":
            pick = owner["model"]["mode"]
        if pick == "tables":
            why = "cost model / HDK_BENCH_EXCHANGE: tables"
        else:
            try:
                C_probe = A.ExchangeShape()
                ko_probe = A.KernelOptions.from_buffer_copy(step.ko)
                ko_probe.total_rows = int(bound.item())
                check(L.hdk_hip_exchange_shape_for(C.byref(step.plan), C.byref(ko_probe), world, n_owner, dev, C.byref(C_probe)))
            except Exception as e:  # plan outside the radix-partitioned shape
                why = str(e)
        failed = torch.tensor([1 if why else 0], dtype=torch.int64, device="cuda")
        comm.all_reduce(failed, "max")
        if int(failed.item()) == 0:
            xch = D.TupleExchange(step, world, int(bound.item()), n_owner)  # (collective: the ranks agree on the tuple width)
            owner["table"], owner["entries"] = owner_t, n_owner
            mode = "tuples"
        else:
            if rank == 0:
                print(f"# tuple exchange not used ({why or 'another rank cannot'}); exchanging partial tables", file=sys.stderr)
            step.free()
            step = w.ex.prepare(cp, w.frag_ids, grid=args.grid if primary else 0, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=out_t.data_ptr())
            mode = "tables"

    def _one_step():
        if mode == "tuples":
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(tstream)
            xch.scatter(stream)
            ev[1].record(tstream)
            if comm.backend == "nccl":
                xch.exchange()
            else:  # host-staged test mode
                r_h = torch.empty(xch.send.numel(), dtype=torch.uint8)
                comm.dist.all_to_all_single(r_h, xch.send.cpu())
                xch.recv.copy_(r_h)
            ev[2].record(tstream)
            xch.aggregate(stream)
            ev[3].record(tstream)
            xch_ev.append(ev)
            return
        step.enqueue(stream)  # output-buffer initialisation + launch (fused for the open-addressing tables)
        if not multi:
            return
        if mode == "gather+fold":
            # ResultSetReduction over the per-GPU partial tables: all-gather (1.5 KB each for C2) + device fold
            comm.all_gather(gathered, out_t)
            check(L.hdk_hip_reduce_buffers(C.byref(cp.plan), gathered.data_ptr(), cp.entry_count, that, counts,
                                           world - 1, init_vals.ctypes.data, d_err.data_ptr(), dev, stream))
        else:
            # partial tables: entries to their owner rank (one all-to-all over xGMI), owners re-insert
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(tstream)
            send, counts_o = D.partition_baseline_on_device(cp, out_t, world, dev, stream)
            if comm.backend == "nccl":
                recv, rc = D.exchange_owner_segments(cp, send, counts_o, world, rank)
            else:
                r_h, rc = D.exchange_owner_segments(cp, send.cpu(), counts_o, world, rank)
                recv = r_h.to("cuda")
            owner["table"], owner["entries"] = D.merge_baseline_on_device(cp, recv, rc, dev, D.owner_entry_count_for(cp.entry_count, world), stream)
            e1.record(tstream)
            owner["sent_bytes"] = int(sum(D.baseline_table_quads(cp, int(c)) for i, c in enumerate(counts_o) if i != rank)) * 8
            owner["events"] = (e0, e1)

    def one_step():
        with torch.cuda.stream(tstream):
            _one_step()
        if mode == "tables":
            torch.cuda.synchronize()
            merge_ms.append(owner["events"][0].elapsed_time(owner["events"][1]))

    if mode == "tuples":
        # one probe step decides for all ranks: an incomplete exchange (skew, stale statistics) -> partial tables
        one_step()
        torch.cuda.synchronize()
        bad = torch.tensor([int(step.mgr.to_host(step.d_err.ptr, 4, dev, np.int32)[0] == A.ERR_EXCHANGE_INCOMPLETE)],
                           dtype=torch.int64, device="cuda")
        comm.all_reduce(bad, "max")
        if int(bad.item()):
            step.free()
            step = w.ex.prepare(cp, w.frag_ids, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=out_t.data_ptr())
            mode, xch = "tables", None
        xch_ev.clear()
    if mode == "gather+fold":
        one_step()  # (puts the plan into the workspace head)
        torch.cuda.synchronize()
        step.ko.flags |= A.LAUNCH_PLAN_RESIDENT

    for _ in range(warmup):
        one_step()
    comm.barrier()
    n_ev = C.c_int32(0)
    check(L.hdk_hip_collect_scan_times(dev, None, 0, C.byref(n_ev)))  # drop warm-up events
    merge_ms.clear()

    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    if multi:
        comm.dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    elapsed = float(comm.all_reduce(tt, "max").item())
    ranks_seen = int(comm.all_reduce(torch.ones(1, dtype=torch.int64, device="cuda")).item())  # ranks that really ran

    ms_buf = (C.c_float * max(steps, 1))()
    check(L.hdk_hip_collect_scan_times(dev, ms_buf, steps, C.byref(n_ev)))
    scan_ms = [ms_buf[i] for i in range(min(n_ev.value, steps))]
    xch_ms = None
    if mode == "tuples":
        evs = xch_ev[-steps:]
        xch_ms = {"scatter": float(np.mean([e[0].elapsed_time(e[1]) for e in evs])),
                  "all_to_all": float(np.mean([e[1].elapsed_time(e[2]) for e in evs])),
                  "aggregate": float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))}
        scan_ms = [xch_ms["scatter"] + xch_ms["aggregate"]]  # this rank's kernels (its rows scattered, its keys aggregated)
    avg_scan_ms = float(np.mean(scan_ms)) if scan_ms else float("nan")

    # ---- correctness at full size: size-independent properties (every rank takes part in the collectives) ----------
    checks = {}
    ref = w.reference_checks()
    if name in ("c2", "c5", "c5s"):
        if baseline:
            tbl = owner["table"] if multi else out_t
            n_e = owner["entries"] if multi else cp.entry_count
            k, s = _baseline_groups(torch, cp, tbl, n_e)
            mine = torch.tensor([int(s.sum().item()), int(k.numel()), ref["sum_val"] if ref["sum_val"] < 2**63 else ref["sum_val"] - 2**64,
                                 w.local_rows], dtype=torch.int64, device="cuda")
            comm.all_reduce(mine)
            got_sum, groups, want_sum, all_rows = (int(x) for x in mine.cpu().tolist())
            checks["sum_of_sums"] = bool((got_sum - want_sum) % (1 << 64) == 0)
            checks["groups"] = groups
            if not multi:
                checks["groups_equal_distinct_keys"] = bool(groups == w.distinct_keys())
            first = (k.clone(), s.clone())
            one_step()
            torch.cuda.synchronize()
            tbl = owner["table"] if multi else out_t
            k2, s2 = _baseline_groups(torch, cp, tbl, n_e)
            checks["idempotent"] = bool(torch.equal(first[0], k2) and torch.equal(first[1], s2))
        else:
            final = (gathered[:quads] if multi else out_t).cpu().numpy()
            cols = ExecutionResult(cp, final, cp.entry_count).to_columns()
            mine = torch.tensor([ref["sum_val"] if ref["sum_val"] < 2**63 else ref["sum_val"] - 2**64, w.local_rows],
                                dtype=torch.int64, device="cuda")
            comm.all_reduce(mine)
            want_sum, all_rows = (int(x) for x in mine.cpu().tolist())
            checks["sum_of_sums"] = bool((sum(v for v in cols["s"] if v is not None) - want_sum) % (1 << 64) == 0)
            checks["groups"] = len(cols["key"])
            one_step()
            torch.cuda.synchronize()
            again = (gathered[:quads] if multi else out_t).cpu().numpy()
            checks["idempotent"] = bool(np.array_equal(again, final))
        checks["row_count"] = bool(all_rows == (rows if strong else rows * world))
    elif name == "c3":
        final = (gathered[:quads] if multi else out_t).cpu().numpy()
        want = ref["sum_val_plus_dval"]
        mine = torch.tensor([want if want < 2**63 else want - 2**64], dtype=torch.int64, device="cuda")
        comm.all_reduce(mine)
        checks["sum_equals_torch_gather_sum"] = bool((int(final[0]) - int(mine.item())) % (1 << 64) == 0)
    elif name in __import__("workloads").SYN_SUITE:
        # the suite's NonGroupedAgg / MultiStep / PerfectHashMultiCol shapes: every group and target against torch.bincount /
        # index_add_ / scatter_reduce_ over the same columns (workloads.Workload.check_syn)
        final = out_t.cpu().numpy()
        cols = ExecutionResult(cp, final, cp.entry_count).to_columns()
        checks["groups"] = len(next(iter(cols.values())))
        checks["every_group_and_target_equals_torch"] = bool(w.check_syn(cols, ref["syn"]))
        one_step()
        torch.cuda.synchronize()
        checks["idempotent"] = bool(w.check_syn(ExecutionResult(cp, out_t.cpu().numpy(), cp.entry_count).to_columns(), ref["syn"]))
    elif name.startswith("bh"):
        # the reference's BaselineHash benchmark shape: every group's count / sum / max / min (and avg = sum / count) against
        # torch.bincount / index_add_ / scatter_reduce_ over the same columns
        final = out_t.cpu().numpy()
        cols = ExecutionResult(cp, final, cp.entry_count).to_columns()
        want = ref["bh"]
        ok = len(cols["key0"]) == w.bh_groups
        for k, c_, s_, mx_, mn_, a_ in zip(cols["key0"], cols["c"], cols["s"], cols["mx"], cols["mn"], cols["a"]):
            g = int(k)
            ok = ok and k == float(g) and (c_, s_, mx_, mn_) == (want["count"][g], want["sum"][g], want["max"][g], want["min"][g])
            ok = ok and abs(a_ - want["sum"][g] / want["count"][g]) <= 1e-6 * abs(a_)
        checks["groups"] = len(cols["key0"])
        checks["every_group_equals_torch"] = bool(ok)
        one_step()
        torch.cuda.synchronize()
        again = ExecutionResult(cp, out_t.cpu().numpy(), cp.entry_count).to_columns()
        checks["idempotent"] = bool(sorted(zip(again["key0"], again["c"], again["s"])) == sorted(zip(cols["key0"], cols["c"], cols["s"])))
    elif name in ("c3g", "c3gm"):
        final = (gathered[:quads] if multi else out_t).cpu().numpy()
        cols = ExecutionResult(cp, final, cp.entry_count).to_columns()
        mine = torch.tensor([x if x < 2**63 else x - 2**64 for x in (v % (1 << 64) for v in ref["group_sums"])], dtype=torch.int64,
                            device="cuda")
        comm.all_reduce(mine)
        want = [int(x) for x in mine.cpu().tolist()]
        checks["groups"] = len(cols["g"])
        checks["per_group_sums_equal_torch_index_add"] = bool(
            sorted(cols["g"]) == list(range(64)) and all((s_ - want[g]) % (1 << 64) == 0 for g, s_ in zip(cols["g"], cols["s"])))
    elif name == "c3m":
        final = (gathered[:quads] if multi else out_t).cpu().numpy()
        mine = torch.tensor([ref["sum_val"] if ref["sum_val"] < 2**63 else ref["sum_val"] - 2**64, w.local_rows], dtype=torch.int64,
                            device="cuda")
        comm.all_reduce(mine)
        mx = torch.tensor([ref["max_dval"]], dtype=torch.int64, device="cuda")
        comm.all_reduce(mx, "max")
        checks["sum_count_max_equal_torch"] = bool((int(final[0]) - int(mine[0].item())) % (1 << 64) == 0 and
                                                   int(final[1]) == int(mine[1].item()) and int(final[2]) == int(mx.item()))
    else:
        final = (gathered[:quads] if multi else out_t).cpu().numpy()
        cols = ExecutionResult(cp, final, cp.entry_count).to_columns()
        mine = torch.tensor(ref["key_counts"] + [w.local_rows], dtype=torch.int64, device="cuda")
        comm.all_reduce(mine)
        kc = [int(x) for x in mine.cpu().tolist()]
        checks["counts_add_up"] = bool(name == "q2" or sum(cols["cnt"]) == kc[-1])
        keyname = "cab_type" if name == "q1" else "passenger_count"
        per = {}
        for i, k in enumerate(cols[keyname]):
            kk = {"green": 0, "yellow": 1}.get(k, k)
            per[kk] = per.get(kk, 0) + (cols["cnt"][i] if "cnt" in cols else 0)
        checks["groups"] = len(cols[keyname])
        if name != "q2":
            checks["per_key_counts_equal_torch_bincount"] = bool([per.get(i, 0) for i in range(8)] == kc[:8])

    # ---- bit-exact parity with the oracle on a sample of the same data (rank 0) ------------------------------------
    if rank == 0 and not args.no_oracle_sample:
        from oracle import oracle as O
        from hdk_amd.executor import Executor
        from util import assert_buffers_equal, run_oracle
        st = w.sample_storage(2_000_000)
        scp, want_buf, err = run_oracle(O, st, w.query)
        # (C5's sample is far too small for the radix-partitioned passes to be chosen by themselves: force them, they are what
        # the full-size run uses; the small-table strategies pick themselves)
        res = Executor(st, dev, mgr).execute(scp, flags=A.LAUNCH_FORCE_PARTITIONED if name in ("c5", "c5s") else 0)
        try:
            if baseline:
                from test_gpu_baseline import _check_rows
                _check_rows(scp, res.buffer, want_buf)
            else:
                assert_buffers_equal(scp, res.buffer, want_buf)
            checks["oracle_bit_exact_on_sample"] = err == 0
        except AssertionError:
            checks["oracle_bit_exact_on_sample"] = False

    total_rows = rows if strong else rows * world
    value = total_rows * steps / elapsed
    achieved = w.local_rows * w.alg_bytes_per_row / (avg_scan_ms * 1e-3) / 1e9 if scan_ms else None
    traffic = None
    traffic_source = None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):  # counters are collected by separate rocprofv3 --pmc passes of this same command
        pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc.json")
        if os.path.exists(pmc_path) and rows == CONFIGS[name][0] and world == 1:
            with open(pmc_path) as fpmc:
                traffic = json.load(fpmc).get("traffic_bytes_per_launch")
            traffic_source = f"profiles/{rnd}_{name}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured by this run)"
            break
    kernels = step.kernel_names()
    out = {
        "metric": "rows/sec, 1B-row int64 GROUP BY SUM" if name == "c2" else f"rows/sec, {name}",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "ranks_seen_by_collective": ranks_seen,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"name": name, "workload": w.description, "rows": rows, "rows_per_gpu": w.local_rows,
                   "fragments": nfrag_total, "fragments_per_gpu": len(w.frag_ids), "fragment_rows": 32_000_000,
                   "layout": ("open addressing" if baseline else "perfect hash" if cp.plan.key_count else "non-grouped") +
                             (", keyless" if cp.plan.keyless else "") + f", {cp.entry_count} entries",
                   "parallelism": (f"fragment f -> GPU f mod {world}; " if strong else f"one table per GPU x {world}; ") +
                                  ({"tuples": "tuples scattered to owner segments + all-to-all (equal splits) + owner aggregation",
                                    "tables": "owner partition of the partial table + all-to-all + owner re-insert",
                                    "gather+fold": "all-gather of the partial tables + device fold", "single": "one GPU"}[mode])},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": dominant_kernel(kernels), "kernels": kernels, "avg_kernel_ms": avg_scan_ms,
                     "median_kernel_ms": float(np.median(scan_ms)) if scan_ms else None,
                     "alg_bytes_per_row": w.alg_bytes_per_row,
                     "note": "per GPU: this rank's rows x algorithmic bytes / mean HIP-event time of the scan launch "
                             "(all of its passes)"},
        "checks": checks,
        "setup_s": {"generate": t_gen},
    }
    if mode == "tables" and merge_ms:
        out["merge"] = {"ms": float(np.mean(merge_ms)), "bytes_sent_over_xgmi_per_gpu": owner.get("sent_bytes"),
                        "what": "owner partition + all_to_all_single + owner re-insert, HIP events on the step's stream"}
    if owner.get("model"):
        out["exchange_model"] = owner["model"]
    if mode == "tuples":
        out["exchange"] = {"ms": xch_ms, "bytes_sent_over_xgmi_per_gpu": xch.bytes_sent_per_rank,
                           "tuple_bytes": int(xch.shape.tuple_bytes), "owner_entries": xch.owner_entries,
                           "what": "HIP events on the step's stream around hdk_hip_scatter_to_owners, all_to_all_single and "
                                   "hdk_hip_aggregate_from_ranks"}
    step.free()
    w.resident_out = None if (multi or baseline) else out_t  # (the end-to-end run compares its buffer with this one)
    return out, w


# The pass of a multi-kernel launch that takes the most time, by the rocprofv3 kernel statistics of this same command
# (profiles/r05_bench_default_kernel_stats.csv): the scatter pass of every radix strategy (C3 5.4 of 6.8 ms, C5 level 1 5.4 of
# 12.3 ms, the 256-bin pass 4.7 of 6.7 ms); single-pass launches name their scan kernel.  `roofline.frac` is computed from the
# HIP-event time of ALL passes either way.
_DOMINANT = ("hdk_join_scatter_slices", "hdk_part_scatter", "hdk_bh_dscatter", "hdk_bh_scatter", "hdk_pp_scatter")


def dominant_kernel(names):
    ks = names.split(",")
    for d in _DOMINANT:
        if d in ks:
            return d
    return ks[0]


def _baseline_groups(torch, cp, table, entry_count):
    from hdk_amd import _abi as A
    p = cp.plan
    rq, n = int(p.row_size_quad), int(entry_count)
    rows = table[:n * rq].view(n, rq)
    if p.key_width == 4:
        keys = (rows[:, 0] << 32) >> 32
        live = keys != A.EMPTY_KEY_32
    else:
        keys = rows[:, 0]
        live = keys != A.EMPTY_KEY_64
    sq = int(p.targets[1].slot_off) // 8
    k, s = keys[live], rows[:, sq][live]
    order = torch.argsort(k)
    return k[order], s[order]


def end_to_end(w, mgr, comm, resident_out, steps=3):
    """C2 with the columns in HOST memory (the reference's cold path: Executor::fetchChunks -> copyHostToDevice[Async],
    QE/Execute.cpp:2965-3068, DataMgr/GpuMgr.h:29-40): every fragment's chunks sit in pinned host memory
    (allocatePinnedHostMem) and travel through GpuMgr::copyHostToDeviceAsync on the manager's stream into one of two staging
    sets while the launch over the previous fragment runs on a second stream; per-fragment launches accumulate into ONE
    output buffer (repeated launches = repeated row-function calls), which must equal the resident run's.  Reports the H->D
    rate, rows/s and how much of the scan time the copies hide."""
    import torch
    from hdk_amd import _abi as A
    from hdk_amd.executor import Executor
    from hdk_amd.storage import ArrowStorage, ChunkStats, Column, Table
    from workloads import TensorChunk
    dev = comm.dev
    tdev = torch.device("cuda", dev)
    t_setup = time.perf_counter()
    frag_rows = [w.frag_rows[f] for f in w.frag_ids]
    big = max(frag_rows)
    names = ["key", "val"]
    # host side: one pinned array per column holding all fragments back to back
    offs = np.concatenate([[0], np.cumsum(frag_rows)]).astype(np.int64)
    host, host_ptrs = {}, []
    for c in names:
        arr, ptr = mgr.pinned_array((int(offs[-1]),), np.int64)
        host_ptrs.append(ptr)
        ht = torch.from_numpy(arr)
        for i, f in enumerate(w.frag_ids):
            ht[offs[i]:offs[i + 1]].copy_(w.cols[("t", c)][f])  # (D -> H once, untimed)
        host[c] = arr
    torch.cuda.synchronize()
    # device side: two staging sets; a two-fragment (+ ragged tail) view of them as a table the executor can prepare steps on
    stage = [{c: torch.empty(big, dtype=torch.int64, device=tdev) for c in names} for _ in range(2)]
    tails = sorted({r for r in frag_rows if r != big})
    st = ArrowStorage()
    vrows = [big, big] + tails
    ex = Executor(st, dev, mgr)
    cols = []
    outer = w.storage.get("t")
    for c in names:
        stats = outer.columns[c].stats[0]
        cols.append(Column(c, outer.columns[c].type, [None] * len(vrows), [ChunkStats(stats.min, stats.max, stats.has_nulls)] * len(vrows)))
        for vi, r in enumerate(vrows):
            src = stage[vi][c] if vi < 2 else None
            if src is not None:
                ex.cache.put(("t", c, vi), TensorChunk(src))
    st.add_table(Table("t", cols, vrows))
    # (a ragged tail fragment reads the head of whichever staging set its parity selects: one view per (tail size, set))
    cp = ex.compile(w.query)
    out_t = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device=tdev)
    step_of = {}
    for b in range(2):
        step_of[(big, b)] = ex.prepare(cp, [b], out_ptr=out_t.data_ptr())
    for ti, r in enumerate(tails):
        for b in range(2):
            for c in names:
                ex.cache.put(("t", c, 2 + ti), TensorChunk(stage[b][c][:r]))
            step_of[(r, b)] = ex.prepare(cp, [2 + ti], out_ptr=out_t.data_ptr())
    copy_s = torch.cuda.ExternalStream(mgr.getStream(dev), device=tdev)
    comp_s = torch.cuda.Stream(device=tdev)
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    t_setup = time.perf_counter() - t_setup

    def run(copies=True, launches=True):
        for e in done:
            e.record(comp_s)
        step_of[(big, 0)].init_output(comp_s.cuda_stream)
        for i in range(len(frag_rows)):
            b, r = i % 2, frag_rows[i]
            if copies:
                copy_s.wait_event(done[b])
                for c in names:
                    mgr.copyHostToDeviceAsync(stage[b][c].data_ptr(), host[c][offs[i]:offs[i + 1]], r * 8, dev)
                ready[b].record(copy_s)
                comp_s.wait_event(ready[b])
            if launches:
                step_of[(r, b)].launch(comp_s.cuda_stream)
                done[b].record(comp_s)
        torch.cuda.synchronize()

    def timed(**kw):
        run(**kw)  # warm
        best = None
        for _ in range(steps):
            t0 = time.perf_counter()
            run(**kw)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best

    t_pipe = timed()
    same = bool(torch.equal(out_t[:resident_out.numel()], resident_out))
    t_copy = timed(launches=False)
    t_scan = timed(copies=False)  # (launch per fragment over whatever the staging sets hold)
    total_bytes = int(offs[-1]) * 16
    out = {"what": "C2 from pinned host memory: copyHostToDeviceAsync of fragment f + 1 (manager stream) beside the launch over "
                   "fragment f (second stream), two staging sets, launches accumulate into one buffer; best of %d" % steps,
           "rows_per_s": int(offs[-1]) / t_pipe, "ms_per_step": t_pipe * 1e3, "h2d_GBps": total_bytes / t_pipe / 1e9,
           "copy_only_ms": t_copy * 1e3, "copy_only_GBps": total_bytes / t_copy / 1e9, "scan_only_ms": t_scan * 1e3,
           "overlap_fraction": max(0.0, min(1.0, (t_copy + t_scan - t_pipe) / t_scan)) if t_scan > 0 else None,
           "same_result_as_resident_run": same, "pinned_host_GB": total_bytes / 1e9, "setup_s": t_setup,
           "link": _pcie_link(dev)}
    for s_ in step_of.values():
        s_.free()
    del stage, out_t
    for ptr in host_ptrs:
        mgr.freePinnedHostMem(ptr)
    return out


def _pcie_link(dev):
    """Device `dev`'s host link as sysfs states it (speed x width), for the H->D figure: looked up by the device's own PCI
    address (hipDeviceGetPCIBusId through torch), so that HIP_VISIBLE_DEVICES or a multi-GPU host cannot pair the H->D rate
    with another card's link; without a resolvable address the field says which card it read."""
    def read(d):
        with open(os.path.join(d, "current_link_speed")) as f1, open(os.path.join(d, "current_link_width")) as f2:
            return f"{f1.read().strip()} x{f2.read().strip()}"
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        d = f"/sys/bus/pci/devices/{bdf}"
        if os.path.exists(os.path.join(d, "current_link_speed")):
            return read(d)
    except (OSError, AttributeError, RuntimeError):
        pass
    try:
        import glob
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.exists(os.path.join(d, "current_link_speed")) and os.path.exists(os.path.join(d, "mem_info_vram_total")):
                return read(d) + " (first GPU found in sysfs, not matched to the device)"
    except OSError:
        pass
    return None


def cpu_quota_cores():
    """CPUs' worth of time the process's cgroup may use per second (cpu.max / cfs_quota_us), or None when unlimited.  The GPU
    boxes of this pool allow 256 CPUs in the affinity mask and 16 CPUs of QUOTA: every run with more busy threads than that is
    throttled (cpu.stat: three periods in four), which is what rounds 4-5 read as NUMA placement -- more threads, less bandwidth."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
            q, per = float(f1.read()), float(f2.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(w, args):
    """The oracle's HDK-semantics CPU path on the host cores, kernel per fragment + reduction
    (QE/Execute.cpp:2776-2788, :1290-1317), on every fragment of the table, median of 5; plus the sub-task variant
    (QE/ExecutionKernel.cpp:341-358: fragments cut into sub-ranges so that every core has work)."""
    from oracle import oracle as O
    from util import oracle_init_buffer
    cp = w.compiled
    nproc = os.cpu_count() or 1
    sample = w.frag_ids[:args.cpu_sample_frags] if args.cpu_sample_frags else w.frag_ids
    frags = []
    for f in sample:
        h = w.host_fragment(w.query.table, f)
        frags.append([h["key"], h["val"]])
    rows = [w.frag_rows[f] for f in sample]
    srows = int(sum(rows))
    init_buf = oracle_init_buffer(O, cp)

    last = {}

    def timed(fr, nr, threads):
        hf = O.HostFragments(fr, nr)
        times = []
        for _ in range(5):
            tc = time.perf_counter()
            err, last["buf"] = O.run_plan_parallel(cp.plan, hf, init_buf, cp.init_vals, threads)
            times.append(time.perf_counter() - tc)
            assert err == 0
        return srows / float(np.median(times))

    quota = cpu_quota_cores()
    usable = int(min(nproc, O.lib().orc_allowed_cpu_count(), max(int(quota), 1) if quota else nproc))  # CPUs that can really run at once
    threads = int(min(usable, len(sample), O.lib().orc_max_threads()))
    per_fragment = timed(frags, rows, threads)
    variants = {"kernel_per_fragment": {"rows_per_s": per_fragment, "threads": threads}}
    pieces = max(1, -(-usable // len(sample)))
    if pieces > 1:
        sub, subrows = [], []
        for cols, n in zip(frags, rows):
            cut = [n * i // pieces for i in range(pieces + 1)]
            for a, b in zip(cut[:-1], cut[1:]):
                sub.append([c[a:b] for c in cols])
                subrows.append(b - a)
        t_all = int(min(usable, O.lib().orc_max_threads()))
        variants["sub_tasks_all_cores"] = {"rows_per_s": timed(sub, subrows, t_all), "threads": t_all}
    # the row loop HDK's JIT would emit for this query, hand-inlined (oracle/hdk_oracle.c: orc_c2_jit_shaped): decoders ->
    # get_group_value_fast -> agg_sum[_skip_val] with the plan's constants folded, one kernel per fragment on its own
    # thread with a private buffer, fragments first touched by the thread that scans them; bit-exact vs the interpreter
    want = np.array(last["buf"], copy=True)
    for label, ft in (("jit_shaped", True), ("jit_shaped_no_first_touch", False)):
        try:
            sec, out = O.c2_jit_shaped([f[0] for f in frags], [f[1] for f in frags], cp.plan, init_buf, threads, first_touch=ft, reps=5)
        except (ValueError, MemoryError) as e:
            variants[label] = {"error": str(e)}
            continue
        variants[label] = {"rows_per_s": srows / sec, "threads": threads, "host_GBps": srows * w.alg_bytes_per_row / sec / 1e9,
                           "bit_exact_vs_interpreter": bool(np.array_equal(out, want)), "timing": "best of 5, scan + reduction"}
    # every CPU busy: the fragments cut into sub-ranges (QE/ExecutionKernel.cpp:341-358), one pinned thread per sub-range --
    # per physical core and per hardware thread (SMT) -- on pages first touched by their thread
    allowed = int(O.lib().orc_allowed_cpu_count())
    # (thread counts: what the quota lets run at once is `threads` above; twice and four times that, while the affinity mask
    # allows, show what oversubscribing the quota costs)
    for t_want in sorted({min(2 * usable, allowed), min(4 * usable, allowed)} - {threads}):
        per = max(1, -(-t_want // len(sample)))
        if per <= 1 and t_want <= len(sample):
            continue
        subk, subv = [], []
        for cols, n in zip(frags, rows):
            cut = [n * i // per for i in range(per + 1)]
            for a_, b_ in zip(cut[:-1], cut[1:]):
                subk.append(cols[0][a_:b_])
                subv.append(cols[1][a_:b_])
        t_run = min(t_want, len(subk))
        label = f"jit_shaped_{t_run}_threads"
        try:
            sec, out = O.c2_jit_shaped(subk, subv, cp.plan, init_buf, t_run, first_touch=True, reps=5)
            variants[label] = {"rows_per_s": srows / sec, "threads": t_run, "host_GBps": srows * w.alg_bytes_per_row / sec / 1e9,
                               "bit_exact_vs_interpreter": bool(np.array_equal(out, want)), "timing": "best of 5, scan + reduction"}
        except (ValueError, MemoryError) as e:
            variants[label] = {"error": str(e)}
    # ONE thread over one 8 M-row piece: what the row loop costs per row when nothing else competes for memory -- the number
    # that explains the gap to the streaming-read figure below (a dependent read-modify-write per row, not a stream)
    single = None
    try:
        n1 = min(8_000_000, rows[0])
        sec1, _ = O.c2_jit_shaped([frags[0][0][:n1]], [frags[0][1][:n1]], cp.plan, init_buf, 1, first_touch=True, reps=5)
        mhz = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("cpu MHz"):
                    mhz = max(mhz or 0.0, float(ln.split(":")[1]))
        single = {"rows_per_s": n1 / sec1, "ns_per_row": sec1 / n1 * 1e9, "host_GBps": n1 * w.alg_bytes_per_row / sec1 / 1e9,
                  "cycles_per_row_at_max_cpu_MHz": (sec1 / n1 * mhz * 1e6) if mhz else None, "max_cpu_MHz_seen": mhz}
    except (ValueError, MemoryError, OSError) as e:
        single = {"error": str(e)}
    # the host: sockets / NUMA nodes, and what a pinned kernel-per-thread READ of two 8-byte columns reaches (the access
    # shape of the row loop without its dependent read-modify-write) -- the figure the baseline is held against.  The same
    # 8 GiB in all, whatever the thread count (a footprint that grows with the threads measured the host's page placement,
    # not its DRAM: 726 GB/s with 32 threads x 512 MiB, 86 with 256 x 512 MiB on the same box, profiles/r05_cpu_host_probe.txt)
    host = {"cpus_allowed": allowed, "numa_nodes": None, "sockets": None, "numa_placement": "threads pinned spread over the allowed "
            "CPUs (t -> cpu[t * ncpus / T]); every thread first-touches its own pages, so a sub-range lives on its thread's node"}
    try:
        host["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
        with open("/proc/cpuinfo") as f:
            lines = f.read().splitlines()
        host["sockets"] = len({ln.split(":")[1].strip() for ln in lines if ln.startswith("physical id")}) or None
        host["model"] = next((ln.split(":", 1)[1].strip() for ln in lines if ln.startswith("model name")), None)
    except OSError:
        pass
    # (round 6) threads placed one per PHYSICAL core first, round robin over the NUMA nodes, SMT siblings last; pages first touched
    # by their reader after the pin; all threads timed between two barriers of one parallel region; the same 8 GiB in all
    stream, per_node = {}, {}
    cores = max(int(O.lib().orc_physical_core_count()), 1)
    pn = (C.c_double * 8)()
    host["cpu_quota_cores"] = quota
    host["usable_cpus"] = usable
    t_list = sorted({max(usable // 4, 1), max(usable // 2, 1), usable})
    over = min(4 * usable, allowed)
    for t_stream in t_list + ([over] if over > usable else []):
        stream[str(t_stream)] = float(O.lib().orc_host_stream_read_gbps_placed(t_stream, (4 << 30) // t_stream, 3, pn, 8))
        per_node[str(t_stream)] = [round(x, 1) for x in list(pn)[:max(host["numa_nodes"] or 1, 1)]]
    best_t = max(stream, key=lambda k: stream[k])
    host["physical_cores"] = cores
    host["stream_read_GBps_by_threads"] = stream
    host["stream_read_GBps_by_threads_per_numa_node"] = per_node
    host["stream_read_GBps"] = stream[best_t]
    host["stream_threads"] = int(best_t)
    ks = [str(t) for t in t_list]  # (inside the quota; the oversubscribed figure is listed, not judged)
    host["stream_figures_sane"] = all(stream[b] * 2 >= stream[a] for a, b in zip(ks, ks[1:]))
    host["stream_read_GBps"] = max(stream[k] for k in ks)
    host["stream_threads"] = int(max(ks, key=lambda k: stream[k]))
    if over > usable:
        host["oversubscribed"] = {"threads": over, "stream_read_GBps": stream[str(over)],
                                  "what": "more busy threads than the cgroup's CPU quota: throttled (cpu.stat nr_throttled)"}
    host["numa_placement"] = ("threads placed one per physical core first, round robin over the L3 domains (CCDs) of alternating NUMA "
                              "nodes, SMT siblings last (oracle/hdk_oracle.c: placement_order); every thread first-touches its own pages "
                              "after the pin; thread counts inside the cgroup's CPU quota")
    for v in variants.values():
        if "host_GBps" in v and host["stream_read_GBps"] > 0:
            v["frac_of_host_stream_read"] = v["host_GBps"] / host["stream_read_GBps"]
    # the reported figure: the FASTEST of the JIT-shaped variants (what the host reached; a GPU/CPU ratio from anything slower
    # would flatter the GPU)
    jit = {k: v for k, v in variants.items() if k.startswith("jit_shaped") and "rows_per_s" in v}
    pool = jit or {k: v for k, v in variants.items() if "rows_per_s" in v}
    pick = max(pool, key=lambda k: pool[k]["rows_per_s"])
    best = fastest = variants[pick]
    return {"value": best["rows_per_s"], "unit": "rows/s", "cores": best["threads"], "host_cores": nproc, "kind": "port",
            "sample": f"{len(sample)} of {w.nfrag} fragments = {srows} rows of the same table; one kernel per fragment (or per "
                      f"sub-range) on OpenMP threads + reduction of the partials.  kernel_per_fragment / sub_tasks_all_cores: the "
                      f"oracle's plan INTERPRETER (median of 5); jit_shaped*: the row loop HDK's LLVM JIT would emit for this query, "
                      f"hand-inlined, threads placed one per physical core and L3 domain first (inside the cgroup's CPU quota), fragments in mmap'ed pages first touched by "
                      f"their thread, private buffers on their own cache lines (best of 5); reported: the fastest jit_shaped variant",
            "reported_variant": next(k for k, v in variants.items() if v is best),
            "fastest_variant": next(k for k, v in variants.items() if v is fastest),
            "host": host,
            "single_thread": single,
            "host_stream_read_GBps": host.get("stream_read_GBps"),
            "frac_of_host_stream_read": best.get("frac_of_host_stream_read"),
            "relation_to_the_stream_figure": (
                "the JIT-shaped loop is get_group_value_fast + agg_sum per row: load key -> row address -> load slot -> add -> store, a "
                "store-to-load chain through the group table with a compare on the key slot, not a streaming read; "
                + (f"one thread alone retires a row in {single['ns_per_row']:.2f} ns = {single['host_GBps']:.1f} GB/s"
                   + (f" ({single['cycles_per_row_at_max_cpu_MHz']:.1f} cycles at {single['max_cpu_MHz_seen']:.0f} MHz)" if single.get("cycles_per_row_at_max_cpu_MHz") else "")
                   + f"; {best['threads']} threads reach {best['rows_per_s'] / single['rows_per_s']:.1f} x that"
                   if single and "ns_per_row" in single else "single-thread figure unavailable")),
            "variants": variants}


def measure_traffic_live(config, rows_arg):
    """roofline.traffic measured by THIS run: two child passes of this same script under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE, then WRITE_SIZE -- one counter per pass, kernel trace only, as MI355X_MICROARCH.md's HBM section prescribes),
    summed over the kernels of one launch.  gfx950: bytes read = 2 x 1024 x FETCH_SIZE (a 128-byte request is tallied as 64),
    bytes written = 1024 x WRITE_SIZE.  The children run before this process touches the GPU; None when rocprofv3 is not there,
    a pass fails, or this process is itself being profiled (no nested profilers)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or os.environ.get("ROCPROFILER_SDK_TOOL_LIBRARIES"):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps = 3
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hdk_pmc_", dir="/tmp")
        try:
            cmd = [rocprof, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "x", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--config", config, "--steps", str(steps), "--warmup", "1", "--extra", "none",
                   "--no-cpu-baseline", "--no-oracle-sample", "--no-multi-gpu-emulation", "--no-end-to-end", "--no-live-traffic"]
            if rows_arg:
                cmd += ["--rows", str(rows_arg)]
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                print(f"# live traffic: the {ctr} pass failed (rc {r.returncode}); using the committed profile", file=sys.stderr)
                return None
            acc, calls = {}, {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    if "hdk" not in k or "k_cal" in k or row["Counter_Name"] != ctr:
                        continue
                    acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
                    calls.setdefault(k, set()).add(row["Dispatch_Id"])
            per[ctr] = {k: (acc[k] / len(calls[k]), len(calls[k])) for k in acc}
        except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
            print(f"# live traffic: {e}; using the committed profile", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fe, wr = per["FETCH_SIZE"], per["WRITE_SIZE"]
    if not fe:
        return None
    # the launch's kernels: dispatched as often as the one that reads most (warm-up + steps + the checks' extra step) -- one-off
    # work (a join-table build) is dispatched fewer times and is not part of a step
    main_k = max(fe, key=lambda k: fe[k][0] * fe[k][1])
    n_main = fe[main_k][1]
    traffic, kernels = 0.0, {}
    for k, (kb, n) in fe.items():
        if n < n_main:
            continue
        rd, wb = 2 * 1024 * kb, 1024 * wr.get(k, (0.0, 0))[0]
        traffic += (rd + wb) * n / n_main
        kernels[k.replace("void ", "").replace("hdk::", "").split("(")[0]] = {"read_bytes": round(rd), "written_bytes": round(wb), "dispatches": n}
    return {"bytes_per_launch": round(traffic), "kernels": kernels,
            "source": "measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes (one counter per pass)"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: one rank per GPU through torch.distributed.run, as a child."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--rows", type=int, default=0, help="rows of the table (default: the config's BASELINE size)")
    ap.add_argument("--dim-rows", type=int, default=0, help="c3*: rows of the dimension table (default 10 M, BASELINE's)")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--extra", default="auto", help="other configs to report under 'configs' (auto: all at N=1, none otherwise)")
    ap.add_argument("--cpu-sample-frags", type=int, default=0, help="fragments the CPU baseline runs on (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-sample", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-resident (H->D inclusive) run of C2")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child passes (N = 1 default run); use the committed profile")
    ap.add_argument("--multi-gpu-emulation", choices=["c5", "full", "none"], default="c5",
                    help="N = 1 default run: also measure one rank's step of an 8-GPU job on this device (c5: the tuple "
                         "exchange; full: plus C2's shard; none)")
    ap.add_argument("--no-multi-gpu-emulation", dest="multi_gpu_emulation", action="store_const", const="none")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # No launcher: start the N ranks ourselves, as a CHILD process -- this parent has not touched HIP (nothing above
        # imports torch) and never replaces itself.  Rank 0's JSON line is the child's stdout, relayed as it comes.
        sys.exit(launch_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to print a line for another job size")

    live = None
    if (args.gpus == 1 and not args.no_live_traffic and "RANK" not in os.environ and args.extra == "auto" and args.config == "c2"
            and not os.environ.get("HDK_BENCH_SINGLE_RANK_COLLECTIVES")):
        live = measure_traffic_live(args.config, args.rows)  # (children: before this process touches the GPU)

    comm = Comm(args)
    from hdk_amd._lib import check, lib
    from hdk_amd.hip_mgr import HipMgr
    mgr = HipMgr()
    line, w = run_config(args.config, args, comm, mgr, args.steps, args.warmup, primary=True)
    if comm.rank == 0 and live:
        line["roofline"]["traffic"] = live["bytes_per_launch"]
        line["roofline"]["traffic_source"] = live["source"]
        line["roofline"]["traffic_kernels"] = live["kernels"]
    if comm.rank == 0:
        copy_gbps, read_gbps = C.c_double(0), C.c_double(0)
        check(lib().hdk_hip_mgr_measure_hbm(comm.dev, 4 << 30, 3, C.byref(copy_gbps), C.byref(read_gbps)))
        line["roofline"]["peak_measured"] = {"read_GBps": read_gbps.value, "copy_GBps": copy_gbps.value,
                                             "what": "16 B/lane streaming read / copy (read + written bytes) of 4 GiB, best of 3"}
        if line["roofline"]["achieved"]:
            line["roofline"]["frac_of_measured_read"] = line["roofline"]["achieved"] / read_gbps.value
    if comm.rank == 0 and comm.world == 1 and not args.no_end_to_end and args.config == "c2" and w.resident_out is not None:
        line["end_to_end"] = end_to_end(w, mgr, comm, w.resident_out)
    if comm.rank == 0 and comm.world == 1 and not args.no_cpu_baseline and args.config == "c2":
        line["cpu_baseline"] = cpu_baseline(w, args)
    else:
        line["cpu_baseline"] = None
    del w
    extra = args.extra
    if extra == "auto":
        extra = "c3,c3g,c3gm,c3m,bh1,bh3,bh4,bh5,nga2,msbs1,msphs1,msphs1w,msphs1f,phm2,c5,c5s,q1,q2,q3,q4" if (comm.world == 1 and args.config == "c2" and not args.rows) else ""
    if extra == "none":
        extra = ""
    configs = []
    for name in [x for x in extra.split(",") if x]:
        import gc
        import torch
        gc.collect()
        torch.cuda.empty_cache()
        # (>= 10 timed steps where the headline has them: with 5, one hiccup doubled a 1.5 ms step in round 5's record)
        o, ww = run_config(name, args, comm, mgr, min(args.steps, 10), min(args.warmup, 2), primary=False)
        del ww
        configs.append({k: o[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline", "checks")})
    if configs:
        line["configs"] = configs
    if comm.rank == 0 and comm.world == 1 and args.config == "c2" and not args.rows and args.multi_gpu_emulation != "none":
        # What one rank of an 8-GPU job does per step, measured here on one device (everything but the wire):
        # C2's shard (4 of 32 fragments) incl. init, finalize, an emulated all-gather and the fold of 7 partials; C5's
        # tuple exchange, all 8 ranks and owners one after another (scripts/multi_gpu_floor.py)
        import gc
        import types
        import torch
        gc.collect()
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import multi_gpu_floor as F
        c = F.part_c(types.SimpleNamespace(world=8, steps=5), mgr, quiet=True)
        line["multi_gpu_emulated"] = {
            "what": "per-rank step of an 8-GPU job measured on ONE device, wire excluded (no multi-GPU box was available "
                    "to the build); the all-gather / all-to-all are device copies",
            "c5_tuple_exchange_8_ranks": {"scatter_ms_max": max(p["scatter_ms"] for p in c["per_rank"]),
                                          "aggregate_ms_max": max(p["aggregate_ms"] for p in c["per_owner"]),
                                          "merge_emulated_ms": max(p["aggregate_ms"] for p in c["per_owner"]),
                                          "step_ms": c["step_ms_wire_excluded"],
                                          "bytes_sent_over_xgmi_per_gpu": c["per_rank"][0]["bytes_to_other_owners"],
                                          "tuple_bytes": c["per_rank"][0]["tuple_bytes"], "checks": c["checks"],
                                          "projected_rows_per_s_at_8_gpus": c["projected_rows_per_s_at_8_gpus_wire_excluded"]}}
        # ... and the same rank's step as a PIPELINE: its rows in four chunks, scatter / all-to-all / aggregate of consecutive
        # chunks on three streams, the wire a device copy plus the modelled link time (scripts/multi_gpu_floor.py: part_d)
        gc.collect()
        torch.cuda.empty_cache()
        d = F.part_d(types.SimpleNamespace(world=8, steps=5), mgr, quiet=True)
        line["multi_gpu_emulated"]["c5_exchange_pipeline_rank0_of_8"] = d
        if args.multi_gpu_emulation == "full":
            # (not in the default run: its launches carry the headline kernel's name and would blur that kernel's
            # average in a rocprofv3 --stats summary of this command; profiles/r03_multi_gpu_floor_after.json has them)
            gc.collect()
            torch.cuda.empty_cache()
            fa = types.SimpleNamespace(world=8, steps=10, configs=["c2"], one_rank=False)
            a = F.part_a(fa, mgr, quiet=True)["c2"]
            line["multi_gpu_emulated"]["c2_rank0_of_8"] = {
                "rows": a["rows_per_rank"], "step_ms": a["plan_resident"]["wall_ms_per_step"],
                "scan_kernel_ms": a["plan_resident"]["scan_kernel_ms"],
                "host_enqueue_ms": a["plan_resident"]["host_enqueue_ms_per_step"],
                "projected_rows_per_s_at_8_gpus": a["projected_rows_per_s_at_8_gpus_wire_excluded"]}
    comm.close()
    if comm.rank == 0:
        emit(line)


# ---- what the driver reads ---------------------------------------------------------------------------------------------
# The LAST stdout line is a compact headline (< 4 KB: round 5's single 20 KB line could not be parsed by the driver).  The
# full objects -- every secondary config with its own roofline, the CPU baseline's variants and host probe, the end-to-end
# leg, the multi-GPU emulations -- go to a side file whose path the headline names.
HEADLINE_MAX_BYTES = 4096
_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_kernel_ms",
                  "median_kernel_ms", "alg_bytes_per_row", "peak_measured", "frac_of_measured_read")
_CPU_KEYS = ("value", "unit", "cores", "host_cores", "kind", "sample", "reported_variant", "frac_of_host_stream_read", "host_stream_read_GBps")
_E2E_KEYS = ("rows_per_s", "h2d_GBps", "overlap_fraction", "same_result_as_resident_run", "link")
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "ranks_seen_by_collective", "steps", "warmup", "ms_per_step", "higher_is_better",
             "scaling", "vs_baseline", "dtype", "data", "config")


def _sig(x, digits=5):
    """Floats to `digits` significant figures (the side file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _checks_ok(checks):
    return bool(checks) and all(v is not False for v in checks.values())


def headline(line, detail_path=None):
    """The compact object printed as the last stdout line: BASELINE.json's metric with `roofline` and `cpu_baseline`, the
    secondary configs reduced to name -> [ms_per_step, roofline.frac, every check held]."""
    h = {k: line[k] for k in _TOP_KEYS if k in line}
    h["value"] = line["value"]  # (full precision: the driver divides by it)
    r = dict(line.get("roofline") or {})
    if isinstance(r.get("traffic_source"), str):
        r["traffic_source"] = r["traffic_source"].split(" (")[0]
    if isinstance(r.get("peak_measured"), dict):
        r["peak_measured"] = {k: v for k, v in r["peak_measured"].items() if k != "what"}
    h["roofline"] = _sig({k: r[k] for k in _ROOFLINE_KEYS if k in r})
    cb = line.get("cpu_baseline")
    if cb:
        c = {k: cb[k] for k in _CPU_KEYS if k in cb}
        if isinstance(c.get("sample"), str) and len(c["sample"]) > 160:
            c["sample"] = c["sample"][:157] + "..."
        h["cpu_baseline"] = _sig(c)
    else:
        h["cpu_baseline"] = None
    h["checks"] = line.get("checks")
    if line.get("end_to_end"):
        h["end_to_end"] = _sig({k: line["end_to_end"][k] for k in _E2E_KEYS if k in line["end_to_end"]})
    for k in ("merge", "exchange", "exchange_model"):  # (N > 1 only; a handful of numbers)
        if line.get(k):
            h[k] = _sig({kk: vv for kk, vv in line[k].items() if kk != "what" and not isinstance(vv, (dict, list)) or kk == "ms"})
    if line.get("configs"):
        h["configs"] = {(c["config"].get("name") or c["metric"].split(", ")[-1]): [_sig(c["ms_per_step"], 4), _sig(c["roofline"]["frac"], 3), _checks_ok(c["checks"])]
                        for c in line["configs"]}
        h["configs_fields"] = ["ms_per_step", "roofline.frac", "checks_ok"]
    if line.get("multi_gpu_emulated"):
        m = line["multi_gpu_emulated"]
        h["multi_gpu_emulated"] = _sig({
            "measured_on": "ONE device, wire excluded / modelled (no multi-GPU box)",
            "c5_step_ms_8_ranks": m.get("c5_tuple_exchange_8_ranks", {}).get("step_ms"),
            "c5_pipeline_over_max_of_compute_and_wire":
                m.get("c5_exchange_pipeline_rank0_of_8", {}).get("overlapped_over_max_of_compute_and_wire")})
    if detail_path:
        h["detail"] = detail_path
    # (never over the limit: drop the optional parts, least important first)
    for k in ("multi_gpu_emulated", "configs_fields", "exchange_model", "end_to_end", "configs", "exchange", "merge"):
        if len(json.dumps(h)) < HEADLINE_MAX_BYTES:
            break
        h.pop(k, None)
    return h


def emit(line):
    """Side file with everything, then the headline as the last line of stdout."""
    path = os.environ.get("HDK_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    rel = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError as e:  # a read-only tree: the headline still goes out
        print(f"# bench detail not written: {e}", file=sys.stderr)
    sys.stderr.flush()
    try:
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would come out at exit --
        # AFTER this line; drain it first so that the headline is the last thing on stdout
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    s = json.dumps(headline(line, rel))
    assert len(s) < HEADLINE_MAX_BYTES, len(s)
    sys.stdout.write(s + "\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
