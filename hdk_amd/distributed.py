"""Multi-GPU execution of one step: fragments sharded over ranks, partial tables merged.

The reference has no collective layer: a multi-device query produces one partial
ResultSet per device, copies them to the host and reduces them with tbb
(Executor::reduceMultiDeviceResultSets, QE/Execute.cpp:1224-1336).  Here every rank is one process
on one GPU (torch.distributed; backend "nccl" = RCCL over xGMI), and the same reduction happens on
the devices:
  * fragment f of the table belongs to rank f mod G (row-range fragments are independent units);
  * perfect-hash / non-grouped plans: partial tables have identical geometry -> one all-gather of
    the (small) tables, then `hdk_hip_reduce_buffers` folds partials 1..G-1 into partial 0 in rank
    order on every rank (deterministic, every rank ends with the full result, like an all-reduce
    but with the exact agg_*_skip_val semantics a plain ncclSum cannot express);
  * baseline-hash plans: slot positions differ per rank -> every rank splits its non-empty entries
    by owner = mulhi(key_hash, G), one all-to-all moves each entry to its owner, the owner re-inserts
    what it received (reduceOneEntryBaseline); the result is the concatenation of the owners'
    disjoint tables (reduce_baseline_multi_gpu).
  * baseline-hash plans of the radix-partitioned shape (C5) exchange TUPLES instead (TupleExchange): a rank
    never builds a table of its own -- pass 1 of the radix-partitioned group-by scatters its rows straight
    into per-owner segments of fixed size, ONE all-to-all with equal splits moves them (no counts, no host
    synchronisation), every owner runs passes 2-4 over what it received into a table sized for ITS keys.
    The table exchange stays as the fallback for skewed keys and for plans outside that shape.
The collective calls are backend-agnostic (gloo on CPU in the tests); the merge itself is the HIP
kernel and needs a device.
"""
import ctypes as C
from typing import Callable, List, Optional

import numpy as np

from . import _abi as A


_side_streams = {}


def torch_stream_for_library(device):
    """(torch stream to run under, raw handle for the C ABI).  The library reads a NULL stream as "my own
    per-device stream", which is not ordered with torch's default stream (handle 0) nor with the RCCL
    collectives issued from it.  When the caller is on the default stream, a cached side stream is used
    and fenced against it on both ends by `run_ordered`."""
    import torch
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != 0:
        return cur, cur.cuda_stream, None
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=device)
    return side, side.cuda_stream, cur


def run_ordered(device, fn):
    """Run fn(raw_stream_handle) on a torch stream the library can share, ordered after the work already
    queued on the caller's current stream and before whatever the caller queues next."""
    import torch
    ts, handle, fence = torch_stream_for_library(device)
    if fence is None:
        return fn(handle)
    ts.wait_stream(fence)
    with torch.cuda.stream(ts):
        out = fn(handle)
    fence.wait_stream(ts)
    return out


def shard_fragments(num_fragments: int, world_size: int, rank: int) -> List[int]:
    """Fragment f -> rank f mod G (cf. FragmentIDAssignmentExecutionPolicy,
    QE/CostModel/Dispatchers/DefaultExecutionPolicy.cpp:18-26)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return [f for f in range(num_fragments) if f % world_size == rank]


def all_gather_partials(local, world_size: int, group=None):
    """All-gather equally sized partial tables; returns a tensor [world_size * n] in rank order.
    `local` is a 1-D torch tensor (int64) on the backend's device."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return local.clone()
    out = torch.empty(world_size * local.numel(), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(local) for _ in range(world_size)]
        dist.all_gather(parts, local, group=group)
        out = torch.cat(parts)
    return out


def merge_gathered_on_device(cp, gathered, world_size: int, device_id: int, stream=None, d_err=None,
                             entry_counts: Optional[List[int]] = None):
    """Fold partials 1..G-1 of `gathered` (device tensor, rank order) into partial 0, in place, with
    the HIP reduction kernel.  Returns the merged table as a view of `gathered`."""
    from ._lib import check, lib
    import torch
    quads = cp.buffer_quads
    if world_size == 1:
        return gathered[:quads]
    if cp.plan.query_kind == A.Q_BASELINE_HASH:
        raise NotImplementedError("baseline-hash partials go through reduce_baseline_multi_gpu")
    that = (C.c_void_p * (world_size - 1))(*[gathered.data_ptr() + i * quads * 8 for i in range(1, world_size)])
    counts = (C.c_uint32 * (world_size - 1))(*([cp.entry_count] * (world_size - 1)))
    if d_err is None:
        d_err = torch.zeros(1, dtype=torch.int32, device=gathered.device)
    iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)

    def fold(h):
        check(lib().hdk_hip_reduce_buffers(C.byref(cp.plan), gathered.data_ptr(), cp.entry_count, that, counts,
                                           world_size - 1, iv.ctypes.data, d_err.data_ptr(), device_id, h))

    if stream is None:
        run_ordered(gathered.device, fold)  # never the library's private stream: `gathered` comes from torch
    else:
        fold(stream)
    return gathered[:quads]


def merge_gathered(cp, gathered_host: np.ndarray, world_size: int, reducer: Callable):
    """Host-side driver of the same fold for callers that hold the gathered tables on the host
    (tests on the gloo backend pass the oracle's reducer; production uses merge_gathered_on_device)."""
    quads = cp.buffer_quads
    this = gathered_host[:quads].copy()
    for r in range(1, world_size):
        rc = reducer(cp.plan, this, cp.entry_count, gathered_host[r * quads:(r + 1) * quads], cp.entry_count,
                     cp.init_vals)
        if rc:
            raise RuntimeError(f"reduction failed with code {rc}")
    return this


# ---- baseline hash: owner-partitioned all-to-all ----------------------------------------
def baseline_table_quads(cp, entry_count: int) -> int:
    """int64 words of a baseline table of `entry_count` entries in cp's layout (host-only call)."""
    from ._lib import check, lib
    q = C.c_int64(0)
    check(lib().hdk_hip_baseline_table_quads(C.byref(cp.plan), int(entry_count), C.byref(q)))
    return int(q.value)


def partition_baseline_on_device(cp, table, num_owners: int, device_id: int, stream=None):
    """Split the non-empty entries of the local table (device int64 tensor) by owner.
    Returns (send, counts): `send` holds the owners' compact tables back to back in owner order
    (owner o: baseline_table_quads(cp, counts[o]) words), `counts` is a numpy uint32 array."""
    from ._lib import check, lib
    import torch
    L = lib()
    iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
    counts = (C.c_uint32 * num_owners)()
    check(L.hdk_hip_partition_baseline_count(C.byref(cp.plan), table.data_ptr(), cp.entry_count, iv.ctypes.data,
                                             num_owners, counts, device_id, stream))
    quads = [baseline_table_quads(cp, counts[o]) for o in range(num_owners)]
    send = torch.empty(max(sum(quads), 1), dtype=torch.int64, device=table.device)
    offs = np.concatenate([[0], np.cumsum(quads)])
    segs = (C.c_void_p * num_owners)(*[send.data_ptr() + int(offs[o]) * 8 for o in range(num_owners)])
    check(L.hdk_hip_partition_baseline(C.byref(cp.plan), table.data_ptr(), cp.entry_count, iv.ctypes.data, num_owners,
                                       counts, segs, device_id, stream))
    return send, np.array(list(counts), dtype=np.uint32)


def exchange_owner_segments(cp, send, counts: np.ndarray, world_size: int, rank: int, group=None):
    """The one exchange step of the baseline merge: rank r sends its owner-o segment to rank o.
    Backend-agnostic (RCCL all-to-all over xGMI in production, gloo in the CPU tests).
    Returns (recv, recv_counts): segments received from ranks 0..G-1 back to back, and their entry counts."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return send, counts.copy()
    mine = torch.from_numpy(counts.astype(np.int64)).to(send.device)
    allc = torch.empty(world_size * world_size, dtype=torch.int64, device=send.device)
    try:
        dist.all_gather_into_tensor(allc, mine, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(mine) for _ in range(world_size)]
        dist.all_gather(parts, mine, group=group)
        allc = torch.cat(parts)
    allc = allc.cpu().numpy().reshape(world_size, world_size)  # allc[r][o]: entries rank r holds for owner o
    in_split = [baseline_table_quads(cp, int(c)) for c in counts]
    recv_counts = allc[:, rank].astype(np.uint32)
    out_split = [baseline_table_quads(cp, int(c)) for c in recv_counts]
    recv = torch.empty(max(sum(out_split), 1), dtype=torch.int64, device=send.device)
    dist.all_to_all_single(recv[:sum(out_split)], send[:sum(in_split)], out_split, in_split, group=group)
    return recv, recv_counts


def merge_baseline_on_device(cp, recv, recv_counts: np.ndarray, device_id: int, owner_entry_count: Optional[int] = None,
                             stream=None):
    """Owner side: initialise a fresh table and re-insert every received segment in rank order
    (reduceOneEntryBaseline semantics, QE/ResultSetReduction.cpp:694-731).  Returns (table, entry_count)."""
    from ._lib import check, lib
    from .plan import columnar_init_vals, compact_init_vals, eff_key_count
    import torch
    L = lib()
    p = cp.plan
    n = int(owner_entry_count or p.entry_count)
    table = torch.empty(baseline_table_quads(cp, n), dtype=torch.int64, device=recv.device)
    block, grid = 1024, 1024  # launch shape of the fill kernel only
    if p.output_columnar:
        d_init = torch.from_numpy(columnar_init_vals(cp)).to(recv.device)
        d_sizes = torch.from_numpy(np.array(cp.slot_widths, dtype=np.int8)).to(recv.device)
        check(L.hdk_hip_init_columnar_group_by_buffer(table.data_ptr(), d_init.data_ptr(), n, eff_key_count(p),
                                                      len(cp.slot_widths), d_sizes.data_ptr(), 1, p.keyless, 8, block,
                                                      grid, device_id, stream))
    else:
        d_init = torch.from_numpy(compact_init_vals(cp)).to(recv.device)
        check(L.hdk_hip_init_group_by_buffer(table.data_ptr(), d_init.data_ptr(), n, eff_key_count(p), p.key_width,
                                             p.row_size_quad, p.keyless, 1, block, grid, device_id, stream))
    offs, segs, cnts = 0, [], []
    for c in recv_counts:
        if c:
            segs.append(recv.data_ptr() + offs * 8)
            cnts.append(int(c))
        offs += baseline_table_quads(cp, int(c))
    d_err = torch.zeros(1, dtype=torch.int32, device=recv.device)
    if not stream:  # the library's own stream is not ordered with torch's default stream, which has just filled d_err
        torch.cuda.current_stream(recv.device).synchronize()
    if segs:
        that = (C.c_void_p * len(segs))(*segs)
        tc = (C.c_uint32 * len(segs))(*cnts)
        iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
        check(L.hdk_hip_reduce_buffers(C.byref(p), table.data_ptr(), n, that, tc, len(segs), iv.ctypes.data,
                                       d_err.data_ptr(), device_id, stream))
    err = int(d_err.item())
    if err:
        raise RuntimeError(f"baseline merge failed with error code {err}")
    return table, n


def reduce_baseline_multi_gpu(cp, table, world_size: int, rank: int, device_id: int, group=None, stream=None):
    """partition -> all-to-all -> owner merge.  Every rank returns its owner table: the query result
    is the concatenation of the ranks' non-empty entries (keys are disjoint across owners)."""
    def steps(h):
        send, counts = partition_baseline_on_device(cp, table, world_size, device_id, h)
        recv, recv_counts = exchange_owner_segments(cp, send, counts, world_size, rank, group)
        return merge_baseline_on_device(cp, recv, recv_counts, device_id, None, h)

    if stream is None and table.is_cuda:
        # library kernels, torch copies and the RCCL all-to-all are all ordered on ONE torch stream
        return run_ordered(table.device, steps)
    return steps(stream)


# ---- baseline hash: tuple exchange (include/hdk_hip.h: hdk_hip_exchange_shape_for & co) -----------------------------
def owner_entry_count_for(entry_count: int, world_size: int) -> int:
    """Entries of an owner's table: the plan's table (2 x the NDV estimate, QE/RelAlgExecutor.cpp:1553-1557) split
    evenly -- an owner holds 1 / G of the keys (owner = mulhi32(key_hash, G))."""
    return max(-(-int(entry_count) // int(world_size)), 1024)


def choose_open_addressing_exchange(world_size: int, rows_per_rank: int, key_domain: int, tuple_bytes: int = 8,
                                    entry_bytes: int = 16, link_gbps: float = 64.0, rates: Optional[dict] = None) -> dict:
    """Tuples or pre-aggregated tables for a multi-GPU open-addressing group-by?  A model, not a measurement of the wire:
    every piece of COMPUTE is a rate measured on one MI355X (profiles/r03_multi_gpu_floor_after.json,
    profiles/r03_c5_tuple_exchange_emulated.json), the wire is `link_gbps` per direction and GPU pair (xGMI is point to
    point: a rank's bytes for owner o cross ONE link; the figure is an ASSUMPTION -- the microarch guide lists ~153 GB/s per
    link peak, RCCL all-to-all reaches a fraction of it; no multi-GPU box was available to measure it).

    tuples: every row travels once as `tuple_bytes`; a rank sends rows x (G-1)/G of them, (G-1) links in parallel.
    tables: a rank first aggregates its rows into a table of its own, then ships its DISTINCT groups:
            d = K (1 - exp(-rows / K)) entries of `entry_bytes` -- fewer bytes than tuples while a rank sees its keys
            several times (1 B rows / 100 M keys: 5 rows per group and rank at G = 2, 2.7 at G = 4, 1.75 at G = 8), but it
            pays a local aggregation, the owner partition of the table and a re-insert at the owner.
    Returns {"mode", "tuples_ms", "tables_ms", "tuples_wire_bytes", "tables_wire_bytes", ...}."""
    import math
    G = int(world_size)
    r = {"scatter_rows_per_ms": 128e6 / 0.72,          # hdk_hip_scatter_to_owners: 128 M rows in 0.72 ms
         "aggregate_tuples_per_ms": 128e6 / 1.09,      # hdk_hip_aggregate_from_ranks: 8 x 16 M tuples in 1.04-1.14 ms
         "local_groupby_rows_per_ms": 125e6 / 1.98,    # one local table (C5 shard shape)
         "partition_entries_per_ms": 200e6 / 7.4,      # hdk_hip_partition_baseline over a 200 M-entry table
         "reinsert_entries_per_ms": 71e6 / 4.3}        # hdk_hip_reduce_buffers re-insert at the owner
    r.update(rates or {})
    rows, K = float(rows_per_rank), float(key_domain)
    distinct = K * (1.0 - math.exp(-rows / K)) if K > 0 else rows
    frac = (G - 1) / G
    link_bytes_per_ms = link_gbps * 1e6
    t_bytes = rows * frac * tuple_bytes
    e_bytes = distinct * frac * entry_bytes
    links = max(G - 1, 1)
    tuples_ms = rows / r["scatter_rows_per_ms"] + t_bytes / links / link_bytes_per_ms + rows / r["aggregate_tuples_per_ms"]
    tables_ms = (rows / r["local_groupby_rows_per_ms"] + 2 * distinct / r["partition_entries_per_ms"] +
                 e_bytes / links / link_bytes_per_ms + distinct / r["reinsert_entries_per_ms"])
    return {"mode": "tuples" if tuples_ms <= tables_ms else "tables", "tuples_ms": tuples_ms, "tables_ms": tables_ms,
            "tuples_wire_bytes": t_bytes, "tables_wire_bytes": e_bytes, "distinct_groups_per_rank": distinct,
            "rows_per_group_and_rank": rows / distinct if distinct else 0.0, "link_gbps_assumed": link_gbps}


def exchange_equal_segments(send, recv, world_size: int, group=None):
    """The one collective of the tuple exchange: `send` and `recv` are `world_size` segments of equal size; segment o of
    rank r's `send` lands as segment r of rank o's `recv`.  Sizes are static (hdk_hip_exchange_shape::segment_bytes):
    no counts travel and the host never waits for the device.  RCCL all-to-all over xGMI in production, gloo in the
    CPU tests."""
    import torch.distributed as dist
    if not dist.is_initialized():
        recv.copy_(send)
        return
    if send.numel() % world_size or recv.numel() != send.numel():
        raise ValueError("send / recv must be world_size segments of equal size")
    dist.all_to_all_single(recv, send, group=group)


class TupleExchange:
    """One rank's side of a multi-GPU open-addressing group-by: scatter -> all-to-all -> aggregate.

    `step` is the rank's PreparedStep whose output buffer (GROUPBY_BUF[0]) is the OWNER table of this rank
    (`owner_table_quads` words); `rows_bound` is the same upper bound on a rank's rows on every rank."""

    def __init__(self, step, world_size: int, rows_bound: int, owner_entry_count: Optional[int] = None, flags: int = 0,
                 group=None):
        import torch
        from ._lib import check, lib
        self.torch, self.L, self.check = torch, lib(), check
        self.step, self.world, self.dev = step, int(world_size), step.dev
        cp = step.cp
        self.owner_entries = int(owner_entry_count or owner_entry_count_for(cp.entry_count, world_size))
        self.ko = A.KernelOptions.from_buffer_copy(step.ko)
        self.ko.total_rows = int(rows_bound)
        self.ko.flags = int(flags)
        self.shape = A.ExchangeShape()
        check(self.L.hdk_hip_exchange_shape_for(C.byref(step.plan), C.byref(self.ko), self.world, self.owner_entries,
                                                self.dev, C.byref(self.shape)))
        # The tuple width comes from THIS rank's column statistics (8-byte tuples when key and argument fit 32 bits);
        # the all-to-all has equal splits and the owner lays its inbox out with its own shape, so the ranks must agree:
        # the widest tuple any rank needs is taken by all (HDK_HIP_LAUNCH_WIDE_TUPLES), then the whole shape is compared.
        self._agree_on_shape(step, group)
        device = torch.device("cuda", self.dev)
        n = self.world * int(self.shape.segment_bytes)
        self.send = torch.empty(n, dtype=torch.uint8, device=device)
        self.recv = torch.empty(n, dtype=torch.uint8, device=device)
        self.ws_scatter = torch.empty(int(self.shape.scatter_workspace_bytes), dtype=torch.uint8, device=device)
        self.ws_aggregate = torch.empty(int(self.shape.aggregate_workspace_bytes), dtype=torch.uint8, device=device)

    def _agree_on_shape(self, step, group=None):
        import torch.distributed as dist
        if self.world == 1 or not dist.is_initialized():
            return
        torch = self.torch
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

        def fields():
            sh = self.shape
            return [int(sh.tuple_bytes), int(sh.segment_bytes), int(sh.segment_header_bytes), int(sh.coarse_per_owner),
                    int(sh.regions_log2), int(sh.sub_slab_tuples), int(sh.owner_entry_count), int(sh.rows_bound)]

        def spread(vals):
            hi = torch.tensor(vals, dtype=torch.int64, device=dev)
            lo = hi.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            return hi.cpu().tolist(), lo.cpu().tolist()

        hi, lo = spread(fields())
        if hi[0] != lo[0]:  # some rank cannot narrow: nobody does (every rank takes this branch: hi / lo are global)
            if int(self.shape.tuple_bytes) < hi[0]:
                self.ko.flags |= A.LAUNCH_WIDE_TUPLES
                self.check(self.L.hdk_hip_exchange_shape_for(C.byref(step.plan), C.byref(self.ko), self.world,
                                                             self.owner_entries, self.dev, C.byref(self.shape)))
            hi, lo = spread(fields())
        if hi != lo:
            raise RuntimeError(f"ranks disagree on the exchange shape (max {hi} / min {lo}): the same plan, row bound and "
                               "owner entry count are required on every rank")

    @property
    def owner_table_quads(self) -> int:
        return baseline_table_quads(self.step.cp, self.owner_entries)

    @property
    def bytes_sent_per_rank(self) -> int:
        return (self.world - 1) * int(self.shape.segment_bytes)

    def _on_stream(self, stream, fn):
        # A NULL stream means "the library's own stream" to the C ABI, which is ordered neither with torch's default
        # stream nor with the collective issued from it: a caller on the default stream gets a fenced side stream.
        if stream:
            return fn(stream)
        return run_ordered(self.torch.device("cuda", self.dev), fn)

    def scatter(self, stream=None):
        self._on_stream(stream, lambda h: self.check(self.L.hdk_hip_scatter_to_owners(
            C.byref(self.step.plan), self.step._params, C.byref(self.ko), C.byref(self.shape), self.send.data_ptr(),
            self.dev, h, self.ws_scatter.data_ptr(), self.ws_scatter.numel())))

    def exchange(self, group=None):
        exchange_equal_segments(self.send, self.recv, self.world, group)

    def aggregate(self, stream=None, recv=None, accumulate=False):
        """Passes 2-4 over the received segments into the owner's table; `accumulate`: the table already holds an earlier
        chunk's groups (HDK_HIP_LAUNCH_ACCUMULATE) -- merge instead of writing it completely."""
        r = self.recv if recv is None else recv
        ko = self.ko
        if accumulate:
            ko = A.KernelOptions.from_buffer_copy(self.ko)
            ko.flags |= A.LAUNCH_ACCUMULATE
        self._on_stream(stream, lambda h: self.check(self.L.hdk_hip_aggregate_from_ranks(
            C.byref(self.step.plan), self.step._params, C.byref(ko), C.byref(self.shape), r.data_ptr(), self.dev, h,
            self.ws_aggregate.data_ptr(), self.ws_aggregate.numel())))

    def segment(self, buf, i):
        n = int(self.shape.segment_bytes)
        return buf[i * n:(i + 1) * n]


class ChunkedTupleExchange:
    """The tuple exchange of a rank's rows in K CHUNKS (fragment subsets), pipelined over three streams:

        scatter stream    scatter_to_owners(chunk k + 1)
        wire stream       all-to-all(chunk k)               (RCCL over xGMI; `wire(k, send, recv, stream)` is the caller's)
        aggregate stream  aggregate_from_ranks(chunk k - 1) (chunk 0 writes the owner's table, later ones accumulate into it)

    so that a step costs max(compute, wire) plus the pipeline's fill instead of their sum (the serial form
    is projected at 3.7 ms for C5 at G = 8, compute 1.8 + wire 1.9).  Every chunk is a complete exchange of its own -- its
    own shape (sized by the chunk's row bound), send / recv segments with their headers, workspaces -- so a chunk that turns
    out skewed or stale flags only itself; the owner's error word says so at the end as for one chunk.

    `steps`: one PreparedStep per chunk over that chunk's fragments, ALL with the owner's table as GROUPBY_BUF[0];
    `rows_bounds`: per chunk, the same on every rank."""

    def __init__(self, steps, world_size: int, rows_bounds, owner_entry_count: Optional[int] = None, group=None):
        import torch
        self.torch = torch
        self.world = int(world_size)
        self.chunks = [TupleExchange(st, world_size, rb, owner_entry_count, group=group) for st, rb in zip(steps, rows_bounds)]
        dev = torch.device("cuda", steps[0].dev)
        # (plain priorities: a high-priority wire stream measured WORSE on the one-device emulation -- 6.2 ms against 4.3 at
        # four chunks, profiles/r05_exchange_pipeline.txt)
        self.s_scatter, self.s_wire, self.s_agg = (torch.cuda.Stream(device=dev) for _ in range(3))
        self.ev_scattered = [torch.cuda.Event() for _ in self.chunks]
        self.ev_arrived = [torch.cuda.Event() for _ in self.chunks]
        self.ev_done = torch.cuda.Event()
        self._ran = False  # (a step has been enqueued: the next one is ordered after its ev_done)

    @property
    def owner_entries(self):
        return self.chunks[0].owner_entries

    @property
    def bytes_sent_per_rank(self):
        return sum(c.bytes_sent_per_rank for c in self.chunks)

    def run(self, wire, after=None):
        """One step.  wire(k, chunk, stream): moves chunk k's `send` segments into the peers' `recv` on `stream` (a torch
        stream: issue the collective under `with torch.cuda.stream(stream)`).  `after`: a torch stream the step starts after
        and that is made to wait for the step's end (the caller's timeline).  Consecutive run() calls are ordered among
        themselves (each starts when the previous one's last aggregate has finished); reading the owner's table is the
        caller's to order: wait for `ev_done` or pass `after`."""
        torch = self.torch
        # Step n + 1 starts after step n has ENDED on all three streams, whatever the caller does: scatter(chunk 0) rewrites
        # chunk 0's `send` while the previous step's all-to-all may still read it, the wire rewrites `recv` and chunk 0's
        # non-accumulating aggregate rewrites the owner's table under the previous step's last aggregates (round 5's advisor
        # finding: only `after=` callers that synchronised every step were safe).  ev_done is recorded on the aggregate
        # stream, which by then has waited for every wire and hence every scatter of that step.
        if self._ran:
            for s_ in (self.s_scatter, self.s_wire, self.s_agg):
                s_.wait_event(self.ev_done)
        self._ran = True
        if after is not None:
            start = torch.cuda.Event()
            start.record(after)
            for s_ in (self.s_scatter, self.s_wire, self.s_agg):
                s_.wait_event(start)
        for k, c in enumerate(self.chunks):
            c.scatter(self.s_scatter.cuda_stream)
            self.ev_scattered[k].record(self.s_scatter)
            self.s_wire.wait_event(self.ev_scattered[k])
            wire(k, c, self.s_wire)
            self.ev_arrived[k].record(self.s_wire)
            self.s_agg.wait_event(self.ev_arrived[k])
            c.aggregate(self.s_agg.cuda_stream, accumulate=k > 0)
        self.ev_done.record(self.s_agg)
        if after is not None:
            after.wait_event(self.ev_done)

    def error_codes(self):
        """The error word of every chunk's launches (synchronises): 0, or e.g. HDK_HIP_ERR_EXCHANGE_INCOMPLETE for a chunk
        whose exchange could not be used (the caller then redoes the step with partial tables)."""
        out = []
        for c in self.chunks:
            c.step.mgr.synchronizeStream(c.step.dev)
            self.torch.cuda.synchronize()
            out.append(int(c.step.mgr.to_host(c.step.d_err.ptr, 4, c.step.dev, np.int32)[0]))
        return out

    def wire_rccl(self, group=None):
        """wire() over torch.distributed (RCCL all-to-all with equal splits on the wire stream)."""
        def wire(k, c, stream):
            with self.torch.cuda.stream(stream):
                exchange_equal_segments(c.send, c.recv, self.world, group)
        return wire
