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
  * baseline-hash plans: slot positions differ per rank -> all-gather the tables and re-insert
    (reduceOneEntryBaseline) into a table sized for the union.
The collective calls are backend-agnostic (gloo on CPU in the tests); the merge itself is the HIP
kernel and needs a device.
"""
import ctypes as C
from typing import Callable, List, Optional

import numpy as np

from . import _abi as A


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
        raise NotImplementedError("baseline-hash merge goes through merge_baseline_on_device")
    that = (C.c_void_p * (world_size - 1))(*[gathered.data_ptr() + i * quads * 8 for i in range(1, world_size)])
    counts = (C.c_uint32 * (world_size - 1))(*([cp.entry_count] * (world_size - 1)))
    if d_err is None:
        d_err = torch.zeros(1, dtype=torch.int32, device=gathered.device)
    iv = np.ascontiguousarray(cp.init_vals, dtype=np.int64)
    check(lib().hdk_hip_reduce_buffers(C.byref(cp.plan), gathered.data_ptr(), cp.entry_count, that, counts,
                                       world_size - 1, iv.ctypes.data, d_err.data_ptr(), device_id, stream))
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
