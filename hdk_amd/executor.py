"""Host side of the hot path: fragments -> device buffers -> kernel params -> launch -> result set.

Mirrors the reference's call chain for one execution step:
  Executor::fetchChunks            (QE/Execute.cpp:2965-3068)   -> `BufferCache.chunk`
  buildHashTableForQualifier       (QE/Execute.cpp:3692) + PerfectJoinHashTableBuilder::
      initHashTableOnGpu           (QE/JoinHashTable/Builders/PerfectHashTableBuilder.h:53-137)
                                                                -> `Executor._build_join_table`
  prepareKernelParams              (QE/QueryExecutionContext.cpp:788-964) -> `PreparedStep._params`
  createAndInitializeGroupByBufferGpu (QE/QueryMemoryInitializer.cpp:1054-1156) -> `init_output`
  DeviceKernel::launch             (QE/QueryExecutionContext.cpp:355)     -> `hdk_hip_launch`
  copyGroupByBuffersFromGpu        (:404-426) + aggregate_error_codes (:221-234) -> `fetch`
All compute goes through the C ABI; there is no CPU execution path in this package
(`device_type="CPU"` raises).
"""
import ctypes as C
import dataclasses
from typing import Dict, List, Optional

import numpy as np

from . import _abi as A
from . import result_set
from ._lib import HdkHipError, check, lib, sync_switches
from .hip_mgr import DeviceBuffer, HipMgr
from .ir import QueryMustRunOnCpu, QueryUnit
from .plan import DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS, CompiledPlan, columnar_init_vals, compact_init_vals, compile_query, eff_key_count
from .storage import ArrowStorage


class BufferCache:
    """Device-resident chunks, keyed like DataMgr's GPU_LEVEL chunk cache (table, column, fragment)."""

    def __init__(self, mgr: HipMgr, device_id: int):
        self.mgr, self.device_id = mgr, device_id
        self._chunks: Dict[tuple, DeviceBuffer] = {}

    def chunk(self, table, col_name, frag_idx) -> DeviceBuffer:
        key = (table.name, col_name, frag_idx)
        b = self._chunks.get(key)
        if b is None:
            b = self.mgr.to_device(table.columns[col_name].fragments[frag_idx], self.device_id)
            self._chunks[key] = b
        return b

    def linearized(self, table, col_name) -> DeviceBuffer:
        """All fragments of an inner-join column as one buffer (ColumnFetcher::linearizeColumnFragments)."""
        key = (table.name, col_name, "all")
        b = self._chunks.get(key)
        if b is None:
            frags = table.columns[col_name].fragments
            arr = frags[0] if len(frags) == 1 else np.concatenate(frags)
            b = self.mgr.to_device(arr, self.device_id)
            self._chunks[key] = b
        return b

    def put(self, key, buf: DeviceBuffer):
        self._chunks[key] = buf

    def clear(self):
        for b in self._chunks.values():
            b.free()
        self._chunks.clear()

    def nbytes(self):
        return sum(b.nbytes for b in self._chunks.values())


class ExecutionResult:
    """Host ResultSet + its layout; `to_arrow()` like pyhdk's ExecutionResult (_sql.pyx:76-83)."""

    def __init__(self, cp: CompiledPlan, buf: np.ndarray, entry_count: int, error_code: int = 0,
                 total_matched: Optional[int] = None):
        self.compiled = cp
        self.buffer = buf
        self.entry_count = entry_count
        self.error_code = error_code
        self.total_matched = total_matched  # projection: rows claimed (TOTAL_MATCHED)

    def to_arrow(self):
        return result_set.to_arrow(self.compiled, self.buffer, self.entry_count, self.total_matched)

    def to_columns(self):
        return result_set.to_columns(self.compiled, self.buffer, self.entry_count, self.total_matched)

    def row_count(self):
        if self.compiled.plan.query_kind == A.Q_PROJECTION:
            return min(int(self.total_matched or 0), self.entry_count)
        return int(result_set.non_empty_mask(self.compiled, self.buffer, self.entry_count).sum())


class PreparedStep:
    """Everything one (multi-fragment) kernel launch needs, resident on the device."""

    def __init__(self, ex: "Executor", cp: CompiledPlan, frag_ids: List[int], grid=0, flags=0,
                 out_ptr: Optional[int] = None, watchdog_ms: int = 0):
        self.ex, self.cp, self.frag_ids = ex, cp, list(frag_ids)
        self.mgr, self.dev = ex.mgr, ex.device_id
        self.L = lib()
        sync_switches()  # (a test or an A/B script may have changed HDK_HIP_* since the library read them)
        self.keep: List[DeviceBuffer] = []
        self._graph = None
        p = cp.plan
        storage = ex.storage
        outer = storage.get(cp.query.table)
        nfrag = len(self.frag_ids)
        self.rows_in_step = int(sum(outer.frag_rows[f] for f in self.frag_ids))
        self.ko = A.KernelOptions(grid, 0, 0, flags, self.rows_in_step, int(watchdog_ms), 0)
        ntab = 1 + len(cp.inner_tables)

        # ---- join hash tables (built once per device, cached) ----------------------------
        # the launch uses a private copy of the plan: fusing join tables rewrites column descriptors
        self.plan = A.Plan.from_buffer_copy(cp.plan)
        fuse = bool(cp.inner_tables) and ex.fuse_join_tables and \
            not (flags & (A.LAUNCH_FORCE_SCALAR | A.LAUNCH_FORCE_GLOBAL_ATOMICS)) and \
            bool({"hdk_scan_agg_vec_join", "hdk_scan_project_join", "hdk_scan_agg_vec_keyed", "hdk_scan_project_keyed",
                  "hdk_scan_agg_bh_vec_join"} &
                 set(self.kernel_names().split(",")))
        # (a table that will be fused is built together with its fused form: one sweep over the inner rows)
        self.join_tables = []
        for ji in range(len(cp.inner_tables)):
            spec = self._fuse_spec(ji) if fuse else None
            self.join_tables.append(ex._build_join_table(cp, ji, ((spec[0], self), spec[1], spec[2]) if spec else None))
        if fuse:
            self._fuse_join_tables()
        p = self.plan

        # ---- col_buffers[frag][buf_idx] ---------------------------------------------------
        ncols = len(cp.input_cols)
        flat = np.zeros(max(nfrag * ncols, 1), dtype=np.uint64)
        for fi, f in enumerate(self.frag_ids):
            for ci, (tn, cn, slot) in enumerate(cp.input_cols):
                if slot == 0:
                    flat[fi * ncols + ci] = ex.cache.chunk(outer, cn, f).ptr
                else:
                    flat[fi * ncols + ci] = ex.cache.linearized(storage.get(tn), cn).ptr
        d_flat = self._dev(flat)
        frag_ptrs = np.array([d_flat.ptr + fi * ncols * 8 for fi in range(nfrag)] or [0], dtype=np.uint64)
        d_frag_ptrs = self._dev(frag_ptrs)
        num_rows = np.zeros(max(nfrag * ntab, 1), dtype=np.int64)
        frag_offs = np.zeros(max(nfrag * ntab, 1), dtype=np.uint64)
        row_off = 0
        starts = np.concatenate([[0], np.cumsum(outer.frag_rows)]) if outer.frag_rows else [0]
        for fi, f in enumerate(self.frag_ids):
            num_rows[fi * ntab] = outer.frag_rows[f]
            frag_offs[fi * ntab] = starts[f]
            for ti, tn in enumerate(cp.inner_tables):
                num_rows[fi * ntab + 1 + ti] = storage.get(tn).num_rows
        self.rows_in_step = int(sum(outer.frag_rows[f] for f in self.frag_ids))
        d_num_rows = self._dev(num_rows)
        d_frag_offs = self._dev(frag_offs)
        d_nfrag = self._dev(np.array([nfrag], dtype=np.uint64))
        d_ntab = self._dev(np.array([ntab], dtype=np.uint32))
        self.init_vals_host = compact_init_vals(cp)
        d_init = self._dev(self.init_vals_host)
        self.d_init = d_init
        self.d_err = self._dev(np.zeros(1, dtype=np.int32))
        d_max_matched = self._dev(np.array([p.entry_count], dtype=np.int32))
        d_total_matched = self._dev(np.zeros(2, dtype=np.int32))  # int32 counter in an 8-byte cell
        self.d_total_matched = d_total_matched
        self.d_zero = self._dev(np.zeros(1, dtype=np.int64))
        if len(self.join_tables) == 1:
            jt_param = self.join_tables[0].ptr
        elif self.join_tables:
            jt_param = self._dev(np.array([t.ptr for t in self.join_tables], dtype=np.int64)).ptr
        else:
            jt_param = 0

        # ---- output buffer + GROUPBY_BUF ---------------------------------------------------
        self.buffer_bytes = cp.buffer_bytes
        if out_ptr is None:
            self.out = self.mgr.alloc(max(self.buffer_bytes, 8), self.dev)
            self.keep.append(self.out)
            self.out_ptr = self.out.ptr
        else:
            self.out_ptr = out_ptr
        if p.query_kind == A.Q_NON_GROUPED:
            nslots = len(cp.slot_widths)
            ptrs = np.array([self.out_ptr + 8 * i for i in range(nslots)], dtype=np.uint64)
        else:
            ptrs = np.array([self.out_ptr], dtype=np.uint64)
        d_gb = self._dev(ptrs)
        if p.output_columnar:
            self.d_col_sizes = self._dev(np.array(cp.slot_widths, dtype=np.int8))
            self.d_init_raw = self._dev(columnar_init_vals(cp))

        params = (C.c_void_p * A.KP_COUNT)()
        params[A.KP_COL_BUFFERS] = d_frag_ptrs.ptr
        params[A.KP_NUM_FRAGMENTS] = d_nfrag.ptr
        params[A.KP_LITERALS] = None
        params[A.KP_NUM_ROWS] = d_num_rows.ptr
        params[A.KP_FRAG_ROW_OFFSETS] = d_frag_offs.ptr
        params[A.KP_MAX_MATCHED] = d_max_matched.ptr
        params[A.KP_TOTAL_MATCHED] = d_total_matched.ptr
        params[A.KP_INIT_AGG_VALS] = d_init.ptr
        params[A.KP_GROUPBY_BUF] = d_gb.ptr
        params[A.KP_ERROR_CODE] = self.d_err.ptr
        params[A.KP_NUM_TABLES] = d_ntab.ptr
        params[A.KP_JOIN_HASH_TABLES] = jt_param or None
        self._params = params

        ws = C.c_size_t(0)
        check(self.L.hdk_hip_workspace_size(C.byref(p), C.byref(self.ko), self.dev, C.byref(ws)))
        self.workspace = self.mgr.alloc(max(ws.value, 16), self.dev)
        self.keep.append(self.workspace)
        self.workspace_bytes = ws.value

    def _dev(self, arr) -> DeviceBuffer:
        b = self.mgr.to_device(arr, self.dev)
        self.keep.append(b)
        return b

    def kernel_names(self) -> str:
        sync_switches()
        out = C.create_string_buffer(256)
        check(self.L.hdk_hip_describe_launch(C.byref(self.plan), C.byref(self.ko), self.dev, out, 256))
        return out.value.decode()

    def _fuse_spec(self, ji, max_bytes=2 << 30):
        """What the fused form of join ji's table holds: (cache key, plan columns read through the join, stride), or
        None when the table stays as it is (not one-to-one, more than seven columns, too large)."""
        cp = self.cp
        info = cp.join_infos[ji]
        if info["kind"] != A.JOIN_ONE_TO_ONE:
            return None
        cols = [ci for ci, (tn, cn, slot) in enumerate(cp.input_cols) if slot == ji + 1]
        stride = 1 + len(cols)
        if len(cols) > 7 or info["entry_count"] * stride * 8 > max_bytes:
            return None
        key = (info["inner_table"], info["inner_col"], tuple(cp.input_cols[ci][1] for ci in cols), info["uses_bw_eq"],
               info["for_semi_join"])
        return key, cols, stride

    def _payload_arrays(self, ji, cols):
        cp, p = self.cp, self.plan
        inner = self.ex.storage.get(cp.join_infos[ji]["inner_table"])
        ptrs = (C.c_void_p * max(len(cols), 1))()
        widths = (C.c_int32 * max(len(cols), 1))()
        kinds = (C.c_int32 * max(len(cols), 1))()
        for k, ci in enumerate(cols):
            ptrs[k] = self.ex.cache.linearized(inner, cp.input_cols[ci][1]).ptr
            widths[k] = cp.plan.cols[ci].width
            kinds[k] = cp.plan.cols[ci].kind
        return ptrs, widths, kinds

    def _fuse_join_tables(self):
        """Replace each one-to-one join table by the fused [row id | payload ...] form when the batched
        interpreter (or the projection kernel) will run the plan: one gather per probing row instead of
        slot -> row id -> inner column (HDK_JOIN_ONE_TO_ONE_FUSED, include/hdk_hip.h)."""
        cp, p = self.cp, self.plan
        for ji, info in enumerate(cp.join_infos):
            spec = self._fuse_spec(ji)
            if spec is None:
                continue
            key, cols, stride = spec
            entries = info["entry_count"]
            fused = self.ex._fused_cache.get(key)
            if fused is None:  # (the plain table came out of the cache: derive the fused form from it)
                ptrs, widths, kinds = self._payload_arrays(ji, cols)
                fused = self.mgr.alloc(entries * stride * 8, self.dev)
                check(self.L.hdk_hip_build_fused_join_table(self.join_tables[ji].ptr, entries, ptrs, widths, kinds,
                                                            len(cols), fused.ptr, self.dev, None))
                self.mgr.synchronizeStream(self.dev)
                self.ex._fused_cache[key] = fused
            for k, ci in enumerate(cols):
                # payload words are int64 (ints sign-extended, float widened to double by the decoder)
                p.cols[ci].kind = A.COL_DOUBLE if p.cols[ci].kind in (A.COL_FLOAT, A.COL_DOUBLE) else A.COL_INT
                p.cols[ci].width = 8
                p.cols[ci].table = -(ji + 1)
                p.cols[ci].buf_idx = 1 + k
            p.joins[ji].kind = A.JOIN_ONE_TO_ONE_FUSED
            p.joins[ji].fused_stride = stride
            self.join_tables[ji] = fused

    # ---- the three steps of launchGpuCode ------------------------------------------------------
    def init_output(self, stream=None):
        p = self.cp.plan
        props = self.mgr.getDeviceProperties(self.dev)
        if p.query_kind == A.Q_NON_GROUPED:
            # out_vec slots start at init_agg_vals (QueryExecutionContext.cpp:452-458); the keyless
            # row-wise fill with one "entry" of nslots quads is exactly that copy, on the stream
            check(self.L.hdk_hip_init_group_by_buffer(
                self.out_ptr, self.d_init.ptr, 1, 0, 8, len(self.cp.slot_widths), 1, 1,
                props.max_threads_per_block, props.grid_size, self.dev, stream))
        elif p.query_kind == A.Q_PROJECTION and not self.ex.init_projection_buffers:
            # The reference fills projection buffers like any other (QueryMemoryInitializer.cpp:1103-1153), but
            # every claimed output row is written completely (row position + each target) and nothing past
            # TOTAL_MATCHED is ever read: the fill (6 GB for a 256 M-row scan) is skipped unless asked for.
            pass
        elif p.output_columnar:
            check(self.L.hdk_hip_init_columnar_group_by_buffer(
                self.out_ptr, self.d_init_raw.ptr, p.entry_count, eff_key_count(p), len(self.cp.slot_widths),
                self.d_col_sizes.ptr, 1, p.keyless, 8, props.max_threads_per_block, props.grid_size,
                self.dev, stream))
        else:
            check(self.L.hdk_hip_init_group_by_buffer(
                self.out_ptr, self.d_init.ptr, p.entry_count, eff_key_count(p), p.key_width, p.row_size_quad,
                p.keyless, 1, props.max_threads_per_block, props.grid_size, self.dev, stream))
        if p.query_kind == A.Q_PROJECTION:
            # TOTAL_MATCHED restarts at 0 for every launch (prepareKernelParams,
            # QueryExecutionContext.cpp:941-950); reset on the launch stream with the fill kernel
            check(self.L.hdk_hip_init_group_by_buffer(
                self.d_total_matched.ptr, self.d_zero.ptr, 1, 0, 8, 1, 1, 1,
                props.max_threads_per_block, props.grid_size, self.dev, stream))

    def launch(self, stream=None, init_output=False):
        """hdk_hip_launch into the (initialised) output buffer; init_output=True hands the initialisation to the launch
        (HDK_HIP_LAUNCH_INIT_OUTPUT: row-wise group-by buffers only, and only for a buffer that holds nothing yet)."""
        ko = self.ko
        if init_output:
            ko = A.KernelOptions.from_buffer_copy(self.ko)
            ko.flags |= A.LAUNCH_INIT_OUTPUT
        check(self.L.hdk_hip_launch(C.byref(self.plan), self._params, C.byref(ko), self.dev, stream,
                                    self.workspace.ptr, self.workspace.nbytes))

    @property
    def launch_initialises(self) -> bool:
        """A fresh row-wise open-addressing table: the launch writes the empty image itself -- the radix-partitioned
        group-by builds it region by region in LDS, which saves a write and a read of the whole table."""
        p = self.cp.plan
        return p.query_kind == A.Q_BASELINE_HASH and not p.output_columnar

    def enqueue(self, stream=None):
        """One complete step on the stream: initialise the output buffer and launch (fused where the library can)."""
        if self.launch_initialises:
            self.launch(stream, init_output=True)
        else:
            self.init_output(stream)
            self.launch(stream)

    def fetch(self, stream_synced=False) -> ExecutionResult:
        if not stream_synced:
            self.mgr.synchronizeStream(self.dev)
        err = int(self.mgr.to_host(self.d_err.ptr, 4, self.dev, np.int32)[0])
        if err > 0:
            raise HdkHipError(err, f"device error code {err} (QE/Execute.h:1019-1031)")
        buf = self.mgr.to_host(self.out_ptr, max(self.buffer_bytes, 8), self.dev, np.int64)
        total = None
        if self.cp.plan.query_kind == A.Q_PROJECTION:
            total = int(self.mgr.to_host(self.d_total_matched.ptr, 4, self.dev, np.int32)[0])
        # err < 0: a projection ran out of output rows (benign under a LIMIT; aggregate_error_codes,
        # QueryExecutionContext.cpp:221-234, surfaces it only when nothing positive happened)
        return ExecutionResult(self.cp, buf[:self.buffer_bytes // 8], self.cp.entry_count, err, total)

    def run(self, stream=None) -> ExecutionResult:
        self.enqueue(stream)
        return self.fetch()

    # ---- hipGraph: record init + launch once, replay per execution (small inputs are launch-bound) --------
    def capture_graph(self, stream=None):
        """Record `init_output(); launch()` as a hipGraph on `stream` (None = the manager's stream).  Only the
        LDS-strategy plans are captured (their launches are kernels only once the plan is resident in the
        workspace); for anything else this is a no-op and replay() falls back to the plain sequence."""
        if self._graph or not self.kernel_names().endswith("hdk_finalize"):
            return self
        self.init_output(stream)
        self.launch(stream)  # uploads the plan into the workspace head
        self.mgr.synchronizeStream(self.dev) if stream is None else None
        saved = self.ko.flags
        self.ko.flags = (saved | A.LAUNCH_PLAN_RESIDENT) & ~A.LAUNCH_RECORD_EVENTS
        try:
            check(self.L.hdk_hip_graph_begin_capture(self.dev, stream))
            self.init_output(stream)
            self.launch(stream)
            g = C.c_void_p()
            check(self.L.hdk_hip_graph_end_capture(self.dev, stream, C.byref(g)))
            self._graph = g
        finally:
            self.ko.flags = saved
        return self

    def replay(self, stream=None):
        """One execution: the recorded graph if there is one, else init_output() + launch()."""
        if self._graph:
            check(self.L.hdk_hip_graph_launch(self._graph, self.dev, stream))
        else:
            self.enqueue(stream)

    def free(self):
        if self._graph:
            self.L.hdk_hip_graph_destroy(self._graph)
            self._graph = None
        for b in self.keep:
            b.free()
        self.keep.clear()


class Executor:
    """One device's executor (cf. Executor + QueryExecutionContext for a MultifragmentKernel,
    QE/Execute.cpp:2084-2090: one kernel over all fragments assigned to the device)."""

    def __init__(self, storage: ArrowStorage, device_id: int = 0, mgr: Optional[HipMgr] = None):
        self.storage = storage
        self.mgr = mgr or HipMgr()
        self.device_id = device_id
        self.cache = BufferCache(self.mgr, device_id)
        self._join_cache: Dict[tuple, DeviceBuffer] = {}
        self._fused_cache: Dict[tuple, DeviceBuffer] = {}
        self.fuse_join_tables = True  # HDK_JOIN_ONE_TO_ONE_FUSED for the batched kernels
        self.init_projection_buffers = False  # see PreparedStep.init_output

    def compile(self, q: QueryUnit) -> CompiledPlan:
        return compile_query(self.storage, q)

    def _join_columns(self, inner, info):
        """JoinColumn / JoinColumnTypeInfo per key column: one JoinChunk per inner fragment."""
        jcs, tis, keep = [], [], []
        for k, col in enumerate(info["inner_cols"]):
            chunks = (A.JoinChunk * inner.num_fragments)()
            rid = 0
            for f in range(inner.num_fragments):
                chunks[f].col_buff = self.cache.chunk(inner, col, f).ptr
                chunks[f].num_elems = inner.frag_rows[f]
                chunks[f].row_id = rid
                rid += inner.frag_rows[f]
            raw = np.frombuffer(bytes(chunks), dtype=np.uint8)
            d_chunks = self.mgr.to_device(raw, self.device_id)
            keep.append(d_chunks)
            jcs.append(A.JoinColumn(d_chunks.ptr, raw.nbytes, inner.num_fragments, inner.num_rows, info["elem_szs"][k]))
            # JoinColumnTypeInfo as PerfectJoinHashTableBuilder fills it (Builders/PerfectHashTableBuilder.h:100-106):
            # {size, range min, range max, the column's own NULL, is_bitwise_eq, range max + 1, column kind}
            tis.append(A.JoinColumnTypeInfo(info["elem_szs"][k], info["mins"][k], info["maxs"][k], info["null_vals"][k],
                                            info["uses_bw_eq"], info["col_types"][k], info["translated_null_build"]))
        return jcs, tis, keep

    def _build_join_table(self, cp: CompiledPlan, ji: int, fuse_spec=None) -> DeviceBuffer:
        """HashJoin::getInstance / PerfectJoinHashTable::reify / BaselineJoinHashTable::reify for one
        device (QE/JoinHashTable/HashJoin.cpp:258-330): the table kind the plan names -- perfect or
        keyed, one-to-one or one-to-many -- built with the *_on_device entry points.  The plan chose
        one-to-one from the inner table's data; a duplicate showing up anyway (stale data) is reported
        like the reference's NeedsOneToManyHash."""
        info = cp.join_infos[ji]
        kind = info["kind"]
        key = (info["inner_table"], tuple(info["inner_cols"]), kind, info["uses_bw_eq"], info["for_semi_join"])
        if key in self._join_cache:
            return self._join_cache[key]
        L = lib()
        dev = self.device_id
        inner = self.storage.get(info["inner_table"])
        jcs, tis, keep = self._join_columns(inner, info)
        d_err = self.mgr.to_device(np.zeros(1, dtype=np.int32), dev)
        semi = info["for_semi_join"]
        fused_key = None
        if kind in (A.JOIN_ONE_TO_ONE, A.JOIN_ONE_TO_MANY):
            # HashEntryInfo{max - min + 1 (+ 1 for kBwEq), bucket_normalization}; the table has the NORMALISED count of
            # slots (initHashTableOnGpu, Builders/PerfectHashTableBuilder.h:82-130)
            hei = A.HashEntryInfo(info["hash_entry_count"], info["bucket"])
            entries = info["entry_count"]
            if entries <= 0 or entries > 2**31 - 1:
                raise QueryMustRunOnCpu("join key range too large for a perfect hash table (TooManyHashEntries)")
            if kind == A.JOIN_ONE_TO_ONE:
                table = self.mgr.alloc(entries * 4, dev)
                check(L.hdk_hip_init_hash_join_buff(table.ptr, entries, A.JOIN_INVALID_SLOT, dev, None))
                if fuse_spec is not None and fuse_spec[0][0] not in self._fused_cache:
                    # the table and its fused form in one sweep (hdk_hip_fill_hash_join_buff_fused: partitioned from a
                    # few million rows on, no gather through the row id)
                    (fkey, step), cols, stride = fuse_spec
                    ptrs, widths, kinds = step._payload_arrays(ji, cols)
                    fused = self.mgr.alloc(entries * stride * 8, dev)
                    sb = L.hdk_hip_join_build_scratch_bytes(inner.num_rows, entries, len(cols))
                    scratch = self.mgr.alloc(sb, dev) if sb else None
                    check(L.hdk_hip_fill_hash_join_buff_fused(table.ptr, A.JOIN_INVALID_SLOT, semi, d_err.ptr, jcs[0], tis[0],
                                                              max(info["bucket"], 1), ptrs, widths, kinds, len(cols), fused.ptr,
                                                              scratch.ptr if scratch else None, sb, dev, None))
                    self.mgr.synchronizeStream(dev)
                    if scratch:
                        scratch.free()
                    self._fused_cache[fkey] = fused
                    fused_key = fkey
                elif info["bucketized"]:
                    # (the reference always calls the bucketized fill for a one-to-one table, with bucket 1 for
                    # everything but a DATE key; the plain entry point is that call with bucket 1)
                    check(L.hdk_hip_fill_hash_join_buff_bucketized(table.ptr, A.JOIN_INVALID_SLOT, semi, d_err.ptr, jcs[0],
                                                                   tis[0], info["bucket"], dev, None))
                else:
                    check(L.hdk_hip_fill_hash_join_buff(table.ptr, A.JOIN_INVALID_SLOT, semi, d_err.ptr, jcs[0], tis[0],
                                                        dev, None))
            else:
                n32 = 2 * entries + inner.num_rows
                table = self.mgr.alloc(n32 * 4, dev)
                check(L.hdk_hip_init_hash_join_buff(table.ptr, n32, A.JOIN_INVALID_SLOT, dev, None))
                fill = L.hdk_hip_fill_one_to_many_hash_table_bucketized if info["bucketized"] else \
                    L.hdk_hip_fill_one_to_many_hash_table
                check(fill(table.ptr, hei, A.JOIN_INVALID_SLOT, jcs[0], tis[0], dev, None))
        else:
            kc, w, entries = len(info["inner_cols"]), info["key_width"], info["entry_count"]
            jc_arr = (A.JoinColumn * kc)(*jcs)
            ti_arr = (A.JoinColumnTypeInfo * kc)(*tis)
            one = kind == A.JOIN_KEYED_ONE_TO_ONE
            dict_bytes = entries * (kc + (1 if one else 0)) * w
            table = self.mgr.alloc(dict_bytes + (0 if one else (2 * entries + inner.num_rows) * 4), dev)
            check(L.hdk_hip_init_baseline_hash_join_buff(table.ptr, entries, kc, w, 1 if one else 0,
                                                         A.JOIN_INVALID_SLOT, dev, None))
            check(L.hdk_hip_fill_baseline_hash_join_buff(table.ptr, entries, A.JOIN_INVALID_SLOT, semi, kc, w,
                                                         1 if one else 0, d_err.ptr, jc_arr, ti_arr, dev, None))
            if not one:
                check(L.hdk_hip_fill_one_to_many_baseline_hash_table(table.ptr + dict_bytes, table.ptr, entries,
                                                                     A.JOIN_INVALID_SLOT, kc, w, jc_arr, ti_arr, dev, None))
        self.mgr.synchronizeStream(dev)
        err = int(self.mgr.to_host(d_err.ptr, 4, dev, np.int32)[0])
        d_err.free()
        for d in keep:
            d.free()
        if err != 0:
            table.free()
            if fused_key is not None:
                self._fused_cache.pop(fused_key).free()
            raise QueryMustRunOnCpu(f"join table build failed with code {err} (-1: duplicate key in a one-to-one "
                                    "table, PerfectHashTableBuilder.h:134-141 NeedsOneToManyHash; -2: stale metadata)")
        self._join_cache[key] = table
        return table

    def interrupt(self, value: int = 1):
        """Executor::interrupt (QE/GpuInterrupt.cpp): launches started with LAUNCH_CHECK_INTERRUPT stop with
        ERR_INTERRUPTED; interrupt(0) re-arms (DeviceKernel::initializeRuntimeInterrupter)."""
        check(lib().hdk_hip_set_interrupt(self.device_id, int(value)))

    def prepare(self, q, frag_ids: Optional[List[int]] = None, grid=0, flags=0,
                out_ptr: Optional[int] = None, watchdog_ms: int = 0) -> PreparedStep:
        cp = q if isinstance(q, CompiledPlan) else self.compile(q)
        outer = self.storage.get(cp.query.table)
        if frag_ids is None:
            frag_ids = list(range(outer.num_fragments))
        return PreparedStep(self, cp, frag_ids, grid=grid, flags=flags, out_ptr=out_ptr, watchdog_ms=watchdog_ms)

    def execute(self, q: QueryUnit, device_type: str = "GPU", frag_ids=None, **kw) -> ExecutionResult:
        if device_type != "GPU":
            raise QueryMustRunOnCpu("hdk_amd ships the GPU path only; run device_type='CPU' on HDK itself")
        # A QueryUnit whose open-addressing table the PLANNER sized (no baseline_entry_count): running out of slots means
        # the estimate was wrong (stale statistics, a key from an inner column) -- RelAlgExecutor::handleOutOfMemoryRetry
        # (QE/RelAlgExecutor.cpp:1713-1747) re-runs with a doubled max_groups_buffer_entry_guess, at most twice more, and
        # so does this.  A CompiledPlan or a caller-pinned entry count reports ERR_OUT_OF_SLOTS as before.
        retries_left = 2 if (isinstance(q, QueryUnit) and not q.baseline_entry_count) else 0
        guess = 0
        while True:
            step = self.prepare(q, frag_ids, **kw)
            try:
                return step.run()
            except HdkHipError as e:
                if e.code != A.ERR_OUT_OF_SLOTS or step.cp.plan.query_kind != A.Q_BASELINE_HASH or retries_left == 0:
                    raise
                retries_left -= 1
                guess = max(2 * max(guess, int(step.cp.entry_count)), DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS)
                q = dataclasses.replace(q, baseline_entry_count=guess)
            finally:
                step.free()
