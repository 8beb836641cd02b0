#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: rows/sec on a 1 B-row, 64-key int64
`SELECT key, SUM(val) FROM t GROUP BY key` (config C2), plus % of the HBM roofline and the
HDK-semantics CPU path timed beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over the rank's resident fragments: output-buffer init, the
multi-fragment scan/aggregate launch, finalize and -- for N > 1 -- the partial-aggregate merge
(RCCL all-gather of the per-GPU tables over xGMI + the device reduction kernel).  Inputs are
resident in HBM before the timed region (as HDK keeps chunks cached in GpuBufferMgr).

Scaling is WEAK: every rank holds its own 1 B-row table (32 fragments x 32 M rows, regenerated
from per-fragment seeds), value = rows all ranks processed / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 20261002
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_ROW = 16  # key 8 B + val 8 B


def gen_fragment(seed_idx, rows, nkeys, null_frac):
    rng = np.random.Generator(np.random.PCG64(SEED + seed_idx))
    key = rng.integers(0, nkeys, rows, dtype=np.int64)
    val = rng.integers(-2**31, 2**31, rows, dtype=np.int64)
    if null_frac > 0:
        val[rng.random(rows) < null_frac] = -(2**63)
    return key, val


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    ap.add_argument("--fragment-size", type=int, default=32_000_000)
    ap.add_argument("--keys", type=int, default=64)
    ap.add_argument("--null-frac", type=float, default=0.0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--cpu-sample-frags", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    from hdk_amd import _abi as A
    from hdk_amd._lib import check, lib
    from hdk_amd.executor import Executor
    from hdk_amd.hip_mgr import HipMgr
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.storage import ArrowStorage, ChunkStats, Column, Table
    from hdk_amd.ir import Type

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # HDK_BENCH_BACKEND=gloo is a TEST mode for boxes with fewer GPUs than ranks (ranks share devices, the
    # collectives are staged through host memory): it exercises the N>1 control flow, not xGMI.
    backend = os.environ.get("HDK_BENCH_BACKEND", "nccl")
    dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    def all_gather_tables(dst, src):
        if backend == "nccl":
            dist.all_gather_into_tensor(dst, src)
        else:
            parts = [torch.empty(src.numel(), dtype=src.dtype) for _ in range(world)]
            dist.all_gather(parts, src.cpu())
            dst.copy_(torch.cat(parts))

    def all_reduce_sum(t):
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)

    def all_reduce_max(t):
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
    mgr = HipMgr()
    L = lib()

    # ---- synthetic data: per-fragment seeds, generated on the host, made resident in HBM ----------
    frag_rows = []
    r = args.rows
    while r > 0:
        frag_rows.append(min(args.fragment_size, r))
        r -= frag_rows[-1]
    nfrag = len(frag_rows)
    keep_host = 0 if (args.no_cpu_baseline or rank != 0) else min(args.cpu_sample_frags, nfrag)
    t_gen = time.perf_counter()
    st = ArrowStorage()
    ex = Executor(st, dev, mgr)
    ktype, vtype = Type("int", 8, True), Type("int", 8, True)
    kcol = Column("key", ktype, [None] * nfrag, [None] * nfrag)
    vcol = Column("val", vtype, [None] * nfrag, [None] * nfrag)
    table = Table("t", [kcol, vcol], frag_rows)
    st.add_table(table)

    def produce(f):
        return f, gen_fragment(rank * nfrag + f, frag_rows[f], args.keys, args.null_frac)

    host_frags = {}
    t_upload = 0.0
    # Fragments are generated on the host a few at a time: pool.map submits everything it is given at once, and
    # 32 fragments x 512 MB per rank, times 8 ranks on one node, is more host memory than the job needs to hold.
    in_flight = 4 if world > 1 else 8
    with ThreadPoolExecutor(max_workers=min(in_flight, os.cpu_count() or 1)) as pool:
        for f0 in range(0, nfrag, in_flight):
            for f, (key, val) in pool.map(produce, range(f0, min(f0 + in_flight, nfrag))):
                has_null = bool(args.null_frac > 0 and (val == -(2**63)).any())
                vv = val[val != -(2**63)] if has_null else val
                kcol.stats[f] = ChunkStats(int(key.min()), int(key.max()), False)
                vcol.stats[f] = ChunkStats(int(vv.min()), int(vv.max()), has_null)
                # device-resident chunk (DataMgr GPU_LEVEL cache); the host copy is dropped unless sampled
                tu = time.perf_counter()
                ex.cache.put(("t", "key", f), mgr.to_device(key, dev))
                ex.cache.put(("t", "val", f), mgr.to_device(val, dev))
                t_upload += time.perf_counter() - tu
                if f < keep_host:
                    host_frags[f] = (key, val)
                    kcol.fragments[f], vcol.fragments[f] = key, val
    t_gen = time.perf_counter() - t_gen

    q = QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "sum_val")])
    cp = ex.compile(q)
    quads = cp.buffer_quads

    # One explicit (non-default) stream carries everything: the library treats a NULL stream as "my own
    # stream", which would not be ordered with torch's default stream and hence with the RCCL collective.
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    assert stream != 0
    out_t = torch.empty(max(quads, 1), dtype=torch.int64, device="cuda")
    step = ex.prepare(cp, list(range(nfrag)), grid=args.grid, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=out_t.data_ptr())
    gathered = torch.empty(world * max(quads, 1), dtype=torch.int64, device="cuda") if world > 1 else None
    d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
    init_vals = np.ascontiguousarray(cp.init_vals, dtype=np.int64)

    def one_step():
        with torch.cuda.stream(tstream):
            _one_step()

    def _one_step():
        step.init_output(stream)
        step.launch(stream)
        if world > 1:
            # ResultSetReduction over the per-GPU partial tables: all-gather (1.5 KB each) + device merge
            all_gather_tables(gathered, out_t)
            that = (C.c_void_p * (world - 1))(*[gathered.data_ptr() + i * quads * 8 for i in range(1, world)])
            counts = (C.c_uint32 * (world - 1))(*([cp.entry_count] * (world - 1)))
            check(L.hdk_hip_reduce_buffers(C.byref(cp.plan), gathered.data_ptr(), cp.entry_count, that, counts,
                                           world - 1, init_vals.ctypes.data, d_err.data_ptr(), dev, stream))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    n_ev = C.c_int32(0)
    check(L.hdk_hip_collect_scan_times(dev, None, 0, C.byref(n_ev)))  # drop warm-up events

    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_reduce_max(tt)
        elapsed = float(tt.item())

    ms_buf = (C.c_float * max(args.steps, 1))()
    check(L.hdk_hip_collect_scan_times(dev, ms_buf, args.steps, C.byref(n_ev)))
    scan_ms = [ms_buf[i] for i in range(min(n_ev.value, args.steps))]
    avg_scan_ms = float(np.mean(scan_ms)) if scan_ms else float("nan")

    # ---- correctness at full size ---------------------------------------------------------------------
    final = (gathered[:quads] if world > 1 else out_t).cpu().numpy()
    from hdk_amd.executor import ExecutionResult
    res = ExecutionResult(cp, final, cp.entry_count)
    cols = res.to_columns()
    if os.environ.get("HDK_BENCH_DEBUG"):
        print(f"[rank {rank}] final[:6]={final[:6].tolist()} out_t[:6]={out_t[:6].cpu().numpy().tolist()} "
              f"err={int(d_err.item())} quads={quads} keyless={cp.plan.keyless} final.dtype={final.dtype} "
              f"cols={ {k: v[:3] for k, v in cols.items()} }", file=sys.stderr)
    checks = {}
    # (1) size-independent property: sum of per-key sums == non-grouped SUM(val) over every rank's rows, and
    # the row counts add up.  Every rank scans its own rows once more (non-grouped), the totals are combined
    # with one all-reduce (two's-complement wrap-around is the same on both sides).
    q2 = QueryUnit("t", targets=[Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    tot = ex.execute(q2, frag_ids=list(range(nfrag))).to_columns()
    local_tot = torch.tensor([np.int64(np.uint64(int(tot["s"][0] or 0) % (1 << 64))), int(tot["c"][0])],
                             dtype=torch.int64, device="cuda")
    if world > 1:
        all_reduce_sum(local_tot)
    all_sum, all_cnt = (int(x) for x in local_tot.cpu().tolist())
    # (2) idempotence: one more step (all ranks: it contains the collective) gives the identical buffer
    one_step()
    torch.cuda.synchronize()
    again = (gathered[:quads] if world > 1 else out_t).cpu().numpy()
    if rank == 0:
        key_sum = sum(v for v in cols["sum_val"] if v is not None)
        checks["sum_of_sums"] = bool((key_sum - all_sum) % (1 << 64) == 0)
        checks["row_count"] = bool(all_cnt == args.rows * world)
        checks["groups"] = len(cols["key"])
        checks["idempotent"] = bool(np.array_equal(again, final))

    # ---- CPU baseline (rank 0, N = 1 only): the oracle's HDK-semantics path on a bounded sample -----
    cpu = None
    if rank == 0 and world == 1 and keep_host > 0:
        from oracle import oracle as O
        from util import oracle_init_buffer
        sample = list(range(keep_host))
        frags = [[host_frags[f][0], host_frags[f][1]] for f in sample]
        hf = O.HostFragments(frags, [frag_rows[f] for f in sample])
        threads = int(min(O.lib().orc_max_threads(), len(sample)))
        init_buf = oracle_init_buffer(O, cp)
        times = []
        cbuf = None
        for _ in range(3):
            tc = time.perf_counter()
            err, cbuf = O.run_plan_parallel(cp.plan, hf, init_buf, cp.init_vals, threads)
            times.append(time.perf_counter() - tc)
            assert err == 0
        srows = sum(frag_rows[f] for f in sample)
        cpu = {"value": srows / float(np.median(times)), "unit": "rows/s", "cores": threads, "kind": "port",
               "sample": f"{len(sample)} fragments = {srows} rows of the same workload, kernel-per-fragment on "
                         f"{threads} OpenMP threads + reduction, median of 3"}
        # (3) bit-exact parity of the HIP path with the oracle on the same sample
        gres = ex.execute(cp, frag_ids=sample)
        checks["oracle_bit_exact_on_sample"] = bool(np.array_equal(gres.buffer, cbuf[:quads]))

    if rank == 0:
        total_rows = args.rows * world
        value = total_rows * args.steps / elapsed
        achieved = args.rows * ALG_BYTES_PER_ROW / (avg_scan_ms * 1e-3) / 1e9 if scan_ms else None
        # HBM traffic per launch from the committed PMC passes of this same workload (profiles/), taken
        # with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs and corrected as the guide says
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_c2_pmc.json")
        if os.path.exists(pmc_path) and args.rows == 1_000_000_000 and args.keys == 64 and args.null_frac == 0:
            with open(pmc_path) as fpmc:
                traffic = json.load(fpmc).get("traffic_bytes_per_launch")
        out = {
            "metric": "rows/sec, 1B-row int64 GROUP BY SUM",
            "value": value,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "C2: SELECT key, SUM(val) GROUP BY key; int64, uniform keys",
                       "rows_per_gpu": args.rows, "keys": args.keys, "fragments_per_gpu": nfrag,
                       "fragment_rows": args.fragment_size, "null_frac": args.null_frac,
                       "layout": "perfect hash, %s, %d entries" % ("keyless" if cp.plan.keyless else "keyed",
                                                                   cp.entry_count),
                       "parallelism": f"fragments sharded over {world} GPU(s), all-gather + device reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "kernel": step.kernel_names().split(",")[0], "avg_kernel_ms": avg_scan_ms,
                         "alg_bytes_per_row": ALG_BYTES_PER_ROW},
            "cpu_baseline": cpu,
            "checks": checks,
            "setup_s": {"generate_and_upload": t_gen, "upload_h2d": t_upload},
            "pcie_inclusive": {"note": "informational: one cold pass incl. pageable-host -> HBM upload of the 16 B/row inputs",
                               "h2d_GBps": args.rows * ALG_BYTES_PER_ROW / t_upload / 1e9 if t_upload else None,
                               "rows_per_s": args.rows / (t_upload + elapsed / args.steps) if t_upload else None},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
