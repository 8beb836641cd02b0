"""ArrowStorage-shaped input side: Arrow tables -> fixed-width, in-band-null column fragments.

Mirrors the part of the reference the kernels depend on (omniscidb/ArrowStorage/):
  * `importArrowTable(table, name, fragment_size)` (ArrowStorage.h:85-91): row-range fragments,
    default 32,000,000 rows (ArrowStorage.h:40);
  * null bitmaps are rewritten to in-band sentinels (ArrowStorageUtils.cpp:176-213);
  * decimals are stored as scaled int64, timestamps as int64 ticks, dictionary-encoded strings as
    int32 ids (StringDictionary is out of scope: ids come from Arrow's own dictionary);
  * per-fragment chunk metadata (min / max / has_nulls) as ArrowStorage computes it
    (ArrowStorage.cpp computeStats) -- the inputs of ColRangeInfo / ExpressionRange.
Host buffers are numpy arrays; device residency is the executor's buffer cache.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import _abi as A
from .ir import Type

DEFAULT_FRAGMENT_SIZE = 32_000_000  # reference omniscidb/ArrowStorage/ArrowStorage.h:40


@dataclass
class ChunkStats:
    min: Optional[float]
    max: Optional[float]
    has_nulls: bool


@dataclass
class Column:
    name: str
    type: Type
    fragments: List[np.ndarray]  # one array per fragment, dtype of the physical width
    stats: List[ChunkStats]
    dictionary: Optional[list] = None  # id -> string, for 'dict' columns

    def table_stats(self) -> ChunkStats:
        mins = [s.min for s in self.stats if s.min is not None]
        maxs = [s.max for s in self.stats if s.max is not None]
        return ChunkStats(min(mins) if mins else None, max(maxs) if maxs else None,
                          any(s.has_nulls for s in self.stats))


class Table:
    def __init__(self, name, columns: List[Column], frag_rows: List[int]):
        self.name = name
        self.columns: Dict[str, Column] = {c.name: c for c in columns}
        self.column_order = [c.name for c in columns]
        self.frag_rows = frag_rows

    @property
    def num_rows(self):
        return int(sum(self.frag_rows))

    @property
    def num_fragments(self):
        return len(self.frag_rows)


_NP_OF = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def _null_sentinel_np(t: Type):
    if t.is_fp:
        if t.size == 8:
            return np.array([A.NULL_DOUBLE_BITS], dtype=np.int64).view(np.float64)[0]
        return np.array([A.NULL_FLOAT_BITS], dtype=np.int32).view(np.float32)[0]
    return _NP_OF[t.size](t.null_value())


def _stats(arr: np.ndarray, t: Type) -> ChunkStats:
    if arr.size == 0:
        return ChunkStats(None, None, False)
    sent = _null_sentinel_np(t)
    if t.is_fp:
        isnull = arr.view(np.int64 if t.size == 8 else np.int32) == (
            A.NULL_DOUBLE_BITS if t.size == 8 else A.NULL_FLOAT_BITS)
    else:
        isnull = arr == sent
    has_nulls = bool(isnull.any()) if t.nullable else False
    valid = arr[~isnull] if has_nulls else arr
    if valid.size == 0:
        return ChunkStats(None, None, has_nulls)
    if t.is_fp:
        return ChunkStats(float(valid.min()), float(valid.max()), has_nulls)
    if t.is_date_in_days:
        # the statistics of a DATE-in-days chunk are kept in epoch seconds (DateDaysEncoder::updateStatsEncoded,
        # omniscidb/DataMgr/DateDaysEncoder.h:94-120): what the expression ranges and the join tables are sized from
        return ChunkStats(int(valid.min()) * 86400, int(valid.max()) * 86400, has_nulls)
    return ChunkStats(int(valid.min()), int(valid.max()), has_nulls)


def _arrow_to_fixed(col, field_type):
    """One Arrow ChunkedArray -> (Type, numpy array with in-band nulls, dictionary|None)."""
    import pyarrow as pa
    import pyarrow.compute as pc

    t = field_type
    dictionary = None
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        col = pc.dictionary_encode(col)
        t = col.type
    if pa.types.is_dictionary(t):
        col = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
        if isinstance(col, pa.ChunkedArray):
            col = col.chunk(0) if col.num_chunks else pa.array([], type=t)
        dictionary = col.dictionary.to_pylist()
        idx = col.indices
        ht = Type("dict", 4, True)
        arr = np.asarray(idx.fill_null(ht.null_value()).cast(pa.int32()).to_numpy(zero_copy_only=False),
                         dtype=np.int32)
        return ht, np.ascontiguousarray(arr), dictionary
    if pa.types.is_boolean(t):
        ht = Type("bool", 1, True)
        arr = col.cast(pa.int8()).fill_null(ht.null_value())
    elif pa.types.is_integer(t):
        size = t.bit_width // 8
        if pa.types.is_unsigned_integer(t):
            raise TypeError("unsigned integer columns are not supported (HDK has no unsigned SQL types)")
        ht = Type("int", size, True)
        arr = col.fill_null(ht.null_value())
    elif pa.types.is_floating(t):
        size = t.bit_width // 8
        if size == 2:
            raise TypeError("float16 columns are not supported")
        ht = Type("fp", size, True)
        sent = _null_sentinel_np(ht)
        arr = col.fill_null(float(sent))
    elif pa.types.is_timestamp(t):
        ht = Type("timestamp", 8, True, unit=t.unit)
        arr = col.cast(pa.int64()).fill_null(ht.null_value())
    elif pa.types.is_date32(t):
        # date32 stays a 4-byte day count with an in-band NULL (ArrowStorageUtils.cpp:899-914: size 4 ->
        # replaceNullValuesImpl<int32_t>); the scan reads it as epoch seconds (fixed_width_small_date_decode)
        ht = Type("date", 4, True, unit="d")
        arr = col.cast(pa.int32()).fill_null(ht.null_value())
    elif pa.types.is_decimal(t):
        if t.precision > 18:
            raise TypeError("decimal precision > 18 is not supported (HDK decimals are int64)")
        ht = Type("decimal", 8, True, scale=t.scale)
        # value * 10^scale as int64
        ints = [None if v is None else int(v.scaleb(t.scale)) for v in col.to_pylist()]
        arr = pa.array(ints, type=pa.int64()).fill_null(ht.null_value())
    else:
        raise TypeError(f"unsupported Arrow type {t}")
    np_arr = arr.to_numpy(zero_copy_only=False) if not isinstance(arr, pa.ChunkedArray) else (
        np.concatenate([c.to_numpy(zero_copy_only=False) for c in arr.chunks]) if arr.num_chunks
        else np.array([], dtype=_NP_OF.get(ht.size, np.float64)))
    want = (np.float64 if ht.size == 8 else np.float32) if ht.is_fp else _NP_OF[ht.size]
    return ht, np.ascontiguousarray(np_arr.astype(want, copy=False)), dictionary


class ArrowStorage:
    """In-memory table registry with the reference's import call shape
    (python/pyhdk/hdk.py:2361 `import_arrow(at, name, fragment_size)`)."""

    def __init__(self):
        self.tables: Dict[str, Table] = {}

    def import_arrow(self, at, name: str, fragment_size: Optional[int] = None) -> Table:
        fragment_size = int(fragment_size or DEFAULT_FRAGMENT_SIZE)
        if fragment_size <= 0:
            raise ValueError("fragment_size must be positive")
        n = at.num_rows
        bounds = list(range(0, n, fragment_size)) or [0]
        frag_rows = [min(fragment_size, n - b) for b in bounds] if n else [0]
        cols = []
        for f in at.schema:
            ht, arr, dictionary = _arrow_to_fixed(at.column(f.name), f.type)
            frags = [np.ascontiguousarray(arr[b:b + r]) for b, r in zip(bounds, frag_rows)]
            cols.append(Column(f.name, ht, frags, [_stats(x, ht) for x in frags], dictionary))
        t = Table(name, cols, frag_rows)
        self.tables[name] = t
        return t

    def import_numpy(self, name: str, columns: Dict[str, np.ndarray], fragment_size=None,
                     types: Optional[Dict[str, Type]] = None) -> Table:
        """Direct import of already in-band-null fixed-width columns (used by bench.py so a 1 B-row
        table never goes through Arrow)."""
        fragment_size = int(fragment_size or DEFAULT_FRAGMENT_SIZE)
        n = len(next(iter(columns.values())))
        bounds = list(range(0, n, fragment_size)) or [0]
        frag_rows = [min(fragment_size, n - b) for b in bounds] if n else [0]
        cols = []
        for cname, arr in columns.items():
            ht = (types or {}).get(cname)
            if ht is None:
                ht = Type("fp", arr.dtype.itemsize) if arr.dtype.kind == "f" else Type("int", arr.dtype.itemsize)
            frags = [np.ascontiguousarray(arr[b:b + r]) for b, r in zip(bounds, frag_rows)]
            cols.append(Column(cname, ht, frags, [_stats(x, ht) for x in frags]))
        t = Table(name, cols, frag_rows)
        self.tables[name] = t
        return t

    def add_table(self, t: Table):
        self.tables[t.name] = t
        return t

    def drop_table(self, name):
        self.tables.pop(name, None)

    def get(self, name) -> Table:
        return self.tables[name]
