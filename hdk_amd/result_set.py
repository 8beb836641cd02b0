"""Host copy of an output buffer -> rows / Arrow.

The reader side of the layout contract (reference omniscidb/ResultSet/): which entries are real
groups (ResultSetStorage::isEmptyEntry, ResultSetStorage.cpp:439-521), how slots become values
(ResultSetIteration.cpp: AVG = sum/count with NULL for count 0 -- load_avg_*,
RuntimeFunctions.cpp:1615-1653; in-band NULLs -> Arrow nulls) and ArrowResultSetConverter's column
naming/typing (ResultSet/ArrowResultSetConverter.cpp:825).  Pure numpy: the buffer is already on
the host (copyGroupByBuffersFromGpu, QueryMemoryInitializer.cpp:1250-1283).
"""
from typing import Dict, List

import numpy as np

from . import _abi as A
from .plan import CompiledPlan, align8, columnar_slot_offsets


def _slot_arrays(cp: CompiledPlan, buf: np.ndarray, entry_count: int) -> List[np.ndarray]:
    """One numpy array (length entry_count) per slot, sign-extended to int64."""
    p = cp.plan
    raw = buf.view(np.uint8)
    out = []
    if p.query_kind == A.Q_NON_GROUPED:
        return [buf[i:i + 1].astype(np.int64) for i in range(len(cp.slot_widths))]
    if p.output_columnar:
        for off, w in zip(columnar_slot_offsets(cp, entry_count), cp.slot_widths):
            if not w:  # zero-width slot (projected key of a baseline table): read from the key column
                out.append(None)
                continue
            dt = np.int64 if w == 8 else np.int32
            out.append(raw[off:off + entry_count * w].view(dt).astype(np.int64))
        return out
    rb = int(p.row_size_quad) * 8
    rows = raw[:entry_count * rb].reshape(entry_count, rb)
    s = 0
    for ti in range(p.num_targets):
        tg = p.targets[ti]
        for k in range(2 if tg.agg == A.AGG_AVG else 1):
            off = tg.slot_off if k == 0 else tg.slot2_off
            w = cp.slot_widths[s]
            s += 1
            if not w:
                out.append(None)
                continue
            dt = np.int64 if w == 8 else np.int32
            out.append(np.ascontiguousarray(rows[:, off:off + w]).view(dt).reshape(-1).astype(np.int64))
    return out


def _key_arrays(cp: CompiledPlan, buf: np.ndarray, entry_count: int) -> List[np.ndarray]:
    p = cp.plan
    if p.keyless or p.query_kind == A.Q_NON_GROUPED:
        return []
    raw = buf.view(np.uint8)
    keys = []
    if p.output_columnar:
        stride = align8(entry_count * 8)
        for k in range(p.key_count):
            keys.append(raw[k * stride:k * stride + entry_count * 8].view(np.int64).copy())
        return keys
    rb = int(p.row_size_quad) * 8
    rows = raw[:entry_count * rb].reshape(entry_count, rb)
    kw = int(p.key_width)
    dt = np.int64 if kw == 8 else np.int32
    for k in range(p.key_count):
        keys.append(np.ascontiguousarray(rows[:, k * kw:(k + 1) * kw]).view(dt).reshape(-1).astype(np.int64))
    return keys


def non_empty_mask(cp: CompiledPlan, buf: np.ndarray, entry_count: int) -> np.ndarray:
    """ResultSetStorage::isEmptyEntry[Columnar] negated."""
    p = cp.plan
    if p.query_kind == A.Q_NON_GROUPED:
        return np.ones(1, dtype=bool)
    if p.keyless:
        ks = int(p.idx_target_as_key)
        slot = _slot_arrays(cp, buf, entry_count)[ks]
        iv = int(cp.init_vals[ks])
        if cp.slot_widths[ks] == 4:
            iv = int(np.int64(iv).astype(np.int32))
        return slot != iv
    k0 = _key_arrays(cp, buf, entry_count)[0]
    empty = A.EMPTY_KEY_64 if (p.key_width == 8 or p.output_columnar) else A.EMPTY_KEY_32
    return k0 != empty


def projection_arrays(cp: CompiledPlan, buf: np.ndarray, nrows: int, capacity=None):
    """(row positions, [one int64/float64 numpy array per projected column]) of a Projection buffer
    holding `nrows` rows (TOTAL_MATCHED)."""
    p = cp.plan
    n = int(capacity if capacity is not None else p.entry_count)
    raw = np.ascontiguousarray(buf).view(np.uint8)
    nrows = min(int(nrows), n)
    cols = []
    if p.output_columnar:
        pos = raw[:n * 8].view(np.int64)[:nrows].copy()
        for off, w in zip(columnar_slot_offsets(cp, n), cp.slot_widths):
            dt = {8: np.int64, 4: np.int32, 2: np.int16, 1: np.int8}[w]
            cols.append(raw[off:off + n * w].view(dt)[:nrows].astype(np.int64))
    else:
        rb = int(p.row_size_quad) * 8
        rows = raw[:n * rb].reshape(n, rb)[:nrows]
        pos = np.ascontiguousarray(rows[:, :8]).view(np.int64).reshape(-1)
        for ti in range(p.num_targets):
            off = p.targets[ti].slot_off
            cols.append(np.ascontiguousarray(rows[:, off:off + 8]).view(np.int64).reshape(-1))
    return pos, cols


def _projection_columns(cp: CompiledPlan, buf: np.ndarray, nrows: int, capacity=None) -> Dict[str, list]:
    p = cp.plan
    _, cols = projection_arrays(cp, buf, nrows, capacity)
    res: Dict[str, list] = {}
    for oc, arr in zip(cp.out_cols, cols):
        tg = p.targets[oc.target_idx]
        if tg.arg_is_fp:
            nullv = tg.arg.null_val
            vals = [None if (tg.arg.nullable and v == nullv) else float(np.int64(v).view(np.float64))
                    for v in arr.tolist()]
        else:
            # a narrow columnar slot holds the value truncated to its width: compare with the truncated NULL
            w = cp.slot_widths[oc.target_idx]
            nullv = int(tg.arg.null_val)
            if w < 8:
                nullv = int(np.int64(nullv).astype({4: np.int32, 2: np.int16, 1: np.int8}[w]))
            vals = [None if (tg.arg.nullable and v == nullv) else v for v in arr.tolist()]
            if oc.dictionary is not None:
                vals = [None if v is None else oc.dictionary[v] for v in vals]
            elif oc.scale:
                vals = [None if v is None else v / 10 ** oc.scale for v in vals]
        res[oc.name] = vals
    return res


def to_columns(cp: CompiledPlan, buf: np.ndarray, entry_count=None, nrows=None) -> Dict[str, list]:
    """Materialise result rows as {column name: python list} in entry order (None = NULL)."""
    p = cp.plan
    n = int(entry_count if entry_count is not None else p.entry_count)
    buf = np.ascontiguousarray(buf)
    if p.query_kind == A.Q_PROJECTION:
        return _projection_columns(cp, buf, n if nrows is None else nrows, n)
    mask = non_empty_mask(cp, buf, n)
    slots = _slot_arrays(cp, buf, n)
    keys = _key_arrays(cp, buf, n)
    first_slot = []
    s = 0
    for ti in range(p.num_targets):
        first_slot.append(s)
        s += 2 if p.targets[ti].agg == A.AGG_AVG else 1
    res: Dict[str, list] = {}
    for oc in cp.out_cols:
        tg = p.targets[oc.target_idx]
        fs = first_slot[oc.target_idx]
        w = cp.slot_widths[fs]
        # a projected key without a slot of its own comes from the key columns (target_groupby_indices)
        vals = (slots[fs] if slots[fs] is not None else keys[oc.key_idx])[mask]
        if oc.kind == "key":
            kt = cp.key_types[oc.key_idx]
            nullv = kt.null_value()
            col = []
            if kt.is_fp:  # the key word holds the double's bits (groupByColumnCodegen's bit-cast, QE/IRCodegen.cpp:1219-1221)
                fv = np.ascontiguousarray(vals, dtype=np.int64).view(np.float64)
                # (a FLOAT key sits in the word widened to double, its NULL the widened FLOAT sentinel: makeTargetValue's case 8)
                null_bits = kt.null_as_int64_or_double_bits()
                if kt.size == 4:
                    fv = fv.astype(np.float32).astype(np.float64)
                for v, f in zip(vals.tolist(), fv.tolist()):
                    col.append(None if (kt.nullable and v == null_bits) else f)
                res[oc.name] = col
                continue
            for v in vals.tolist():
                if kt.nullable and v == nullv:
                    col.append(None)
                elif oc.dictionary is not None:
                    col.append(oc.dictionary[v])
                else:
                    col.append(v)
            res[oc.name] = col
            continue
        if oc.agg == "count":
            res[oc.name] = [int(v) for v in vals.tolist()]
            continue
        nullv = int(tg.null_val)
        if w == 4:
            nullv = int(np.int64(nullv).astype(np.int32))
        nullable = bool(tg.skip_null)
        float_slot = tg.arg_is_fp == A.FP_SLOT_FLOAT
        if float_slot:
            # takes_float_argument: the value is the float in the slot's low 4 bytes (ResultSetIteration.cpp:50-60,
            # actual_compact_sz = sizeof(float)); NULL when those bits are the float sentinel
            vals = (vals.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
            fvals = vals.view(np.float32).astype(np.float64)
            is_null = vals == np.uint32(A.NULL_FLOAT_BITS)
        if oc.agg == "avg":
            cnt = slots[fs + 1][mask]
            col = []
            for i, (sv, c) in enumerate(zip(vals.tolist(), cnt.tolist())):
                if c == 0:
                    col.append(None)  # load_avg_*: null when count == 0
                elif float_slot:
                    col.append(float(fvals[i]) / c)
                elif tg.arg_is_fp:
                    col.append(float(np.int64(sv).view(np.float64)) / c)
                else:
                    col.append(sv / c)
            # decimal arguments: AVG(decimal) is scaled back (ResultSetIteration.cpp pair_to_double)
            if oc.scale:
                col = [None if v is None else v / 10 ** oc.scale for v in col]
            res[oc.name] = col
            continue
        col = []
        for i, v in enumerate(vals.tolist()):
            if float_slot:
                col.append(None if (nullable and is_null[i]) else float(fvals[i]))
            elif nullable and v == nullv:
                col.append(None)
            elif tg.arg_is_fp:
                col.append(float(np.int64(v).view(np.float64)))
            else:
                col.append(v)
        if oc.scale and oc.agg in ("sum", "min", "max"):
            col = [None if v is None else v / 10 ** oc.scale for v in col]
        res[oc.name] = col
    return res


def to_arrow(cp: CompiledPlan, buf: np.ndarray, entry_count=None, nrows=None):
    import pyarrow as pa
    cols = to_columns(cp, buf, entry_count, nrows)
    arrays, names = [], []
    for oc in cp.out_cols:
        v = cols[oc.name]
        if oc.kind in ("key", "proj") and oc.dictionary is not None:
            arr = pa.array(v, type=pa.string())
        elif oc.agg == "count":
            arr = pa.array(v, type=pa.int32() if oc.type.size == 4 else pa.int64())
        elif oc.type.is_fp or oc.agg == "avg" or any(isinstance(x, float) for x in v):
            arr = pa.array(v, type=pa.float64())
        else:
            width = {1: pa.int8(), 2: pa.int16(), 4: pa.int32(), 8: pa.int64()}[oc.type.size]
            arr = pa.array(v, type=pa.int64() if oc.agg == "sum" else width)
        arrays.append(arr)
        names.append(oc.name)
    return pa.table(arrays, names=names)
