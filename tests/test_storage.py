"""ArrowStorage-shaped import: fragmenting, in-band NULL sentinels, chunk stats (reference
omniscidb/ArrowStorage/ArrowStorageUtils.cpp:176-213, ArrowStorage.h:40,85-91).  CPU only."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from hdk_amd import _abi as A
from hdk_amd.storage import DEFAULT_FRAGMENT_SIZE, ArrowStorage


def test_default_fragment_size():
    assert DEFAULT_FRAGMENT_SIZE == 32_000_000


def test_fragments_and_null_sentinels():
    st = ArrowStorage()
    at = pa.table({
        "i8": pa.array([1, None, 3, 4, 5], pa.int8()), "i16": pa.array([1, 2, None, 4, 5], pa.int16()),
        "i32": pa.array([None, 2, 3, 4, 5], pa.int32()), "i64": pa.array([1, 2, 3, None, 5], pa.int64()),
        "f32": pa.array([1.0, None, 3.0, 4.0, 5.0], pa.float32()), "f64": pa.array([1.0, 2.0, None, 4.0, 5.0], pa.float64()),
        "ts": pa.array([0, 1, None, 3, 4], pa.timestamp("s")),
        "dec": pa.array([decimal.Decimal("1.25"), None, decimal.Decimal("-3.50"), decimal.Decimal("0"), decimal.Decimal("9.99")],
                        pa.decimal128(14, 2)),
        "s": pa.array(["a", "b", None, "a", "c"]), "b": pa.array([True, False, None, True, True])})
    t = st.import_arrow(at, "t", fragment_size=2)
    assert t.frag_rows == [2, 2, 1] and t.num_rows == 5 and t.num_fragments == 3
    cat = lambda n: np.concatenate(t.columns[n].fragments)
    assert cat("i8").tolist() == [1, A.NULL_TINYINT, 3, 4, 5] and cat("i8").dtype == np.int8
    assert cat("i16").tolist() == [1, 2, A.NULL_SMALLINT, 4, 5]
    assert cat("i32").tolist() == [A.NULL_INT, 2, 3, 4, 5]
    assert cat("i64").tolist() == [1, 2, 3, A.NULL_BIGINT, 5]
    assert cat("f32").view(np.int32)[1] == A.NULL_FLOAT_BITS and cat("f64").view(np.int64)[2] == A.NULL_DOUBLE_BITS
    assert cat("ts").tolist() == [0, 1, A.NULL_BIGINT, 3, 4] and t.columns["ts"].type.kind == "timestamp"
    assert cat("dec").tolist() == [125, A.NULL_BIGINT, -350, 0, 999] and t.columns["dec"].type.scale == 2
    assert t.columns["s"].type.kind == "dict" and cat("s").dtype == np.int32 and cat("s")[2] == A.NULL_INT
    assert [t.columns["s"].dictionary[i] for i in cat("s")[[0, 1, 3, 4]]] == ["a", "b", "a", "c"]
    assert cat("b").tolist() == [1, 0, A.NULL_TINYINT, 1, 1]
    st64 = t.columns["i64"].table_stats()
    assert (st64.min, st64.max, st64.has_nulls) == (1, 5, True)
    assert t.columns["i16"].stats[1].has_nulls and not t.columns["i16"].stats[0].has_nulls


def test_empty_table_and_unsupported():
    st = ArrowStorage()
    t = st.import_arrow(pa.table({"a": pa.array([], pa.int64())}), "e")
    assert t.num_rows == 0 and t.num_fragments == 1 and t.columns["a"].table_stats().min is None
    with pytest.raises(TypeError):
        st.import_arrow(pa.table({"u": pa.array([1, 2], pa.uint32())}), "u")
