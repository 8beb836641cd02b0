"""Pins the oracle (oracle/hdk_oracle.c) against
  (a) golden vectors produced by the REFERENCE's own compiled runtime (tests/golden/gen_golden.py),
  (b) the literal known-answer tests of the reference's test-suite:
      QueryEngine/GroupByHashTest.cpp:57-266, Tests/JoinHashTableTest.cpp:133-260,
      Tests/NoCatalogRelAlgTest.cpp:100-107,211-232.
CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from hdk_amd import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
EMPTY64 = 2**63 - 1
NULL64 = -(2**63)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "ref_runtime_vectors.json")) as f:
        return json.load(f)


def test_hashes(oracle, golden):
    L = oracle.lib()
    for h in golden["hash"]:
        a = np.array([h["key"]], dtype=np.int64)
        assert L.orc_murmur_hash3(a.ctypes.data, 8, 0) == h["murmur3_8"]
        assert L.orc_murmur_hash1(a.ctypes.data, 8, 0) == h["murmur1_8"]
        assert L.orc_murmur_hash64a(a.ctypes.data, 8, 0) == h["murmur64a_8"]
        assert L.orc_murmur_hash3(a.view(np.int32).ctypes.data, 4, 0) == h["murmur3_4"]
        assert L.orc_key_hash(a.ctypes.data, 1, 8) == h["key_hash"]
    for m in golden["hash_multi"]:
        a = np.array(m["key"], dtype=np.int64)
        assert L.orc_key_hash(a.ctypes.data, len(a), 8) == m["key_hash"]
        assert L.orc_key_hash(a.astype(np.int32).ctypes.data, len(a), 4) == m["key_hash_w4"]
    # SURVEY.md 8(a4) probed KATs
    for k, want in ((0, 1669671676), (1, 1392991556), (63, 1840533663), (-1, 1651860712)):
        a = np.array([k], dtype=np.int64)
        assert L.orc_key_hash(a.ctypes.data, 1, 8) == want


def test_get_group_value_sequences(oracle, golden):
    L = oracle.lib()
    for g in golden["get_group_value"]:
        n, nk, kw, rsq = g["entry_count"], g["key_count"], g["key_width"], g["row_size_quad"]
        buf = np.zeros(n * rsq, dtype=np.int64)
        if kw == 8:
            buf.reshape(n, rsq)[:, :nk] = EMPTY64
        else:
            buf.view(np.int32).reshape(n, rsq * 2)[:, :nk] = 2**31 - 1
        for key, want_off in zip(g["keys"], g["slot_quads"]):
            k = np.array(key, dtype=np.int64 if kw == 8 else np.int32)
            p = L.orc_get_group_value(buf.ctypes.data, n, k.ctypes.data, nk, kw, rsq)
            off = -1 if not p else (p - buf.ctypes.data) // 8
            assert off == want_off
            if p:
                buf[off] += int(k.sum()) + 1
        assert buf.tolist() == g["final"]
    for g in golden["get_group_value_columnar_slot"]:
        n, nk = g["entry_count"], g["key_count"]
        buf = np.zeros(n * (nk + 1), dtype=np.int64)
        buf[:n * nk] = EMPTY64
        for key, want in zip(g["keys"], g["slots"]):
            k = np.array(key, dtype=np.int64)
            s = L.orc_get_group_value_columnar_slot(buf.ctypes.data, n, k.ctypes.data, nk, 8)
            assert s == want
            if s >= 0:
                buf[n * nk + s] += 1
        assert buf.tolist() == g["final"]
    for g in golden["get_group_value_fast"]:
        n, rsq = g["entries"], g["row_size_quad"]
        buf = np.zeros(n * rsq, dtype=np.int64)
        buf.reshape(n, rsq)[:, 0] = EMPTY64
        for k in g["keys"]:
            p = L.orc_get_group_value_fast(buf.ctypes.data, k, g["min_key"], g["bucket"], rsq)
            buf[(p - buf.ctypes.data) // 8] += 7
        assert buf.tolist() == g["final"]


def test_aggregates(oracle, golden):
    L = oracle.lib()
    vals = golden["agg"]["int_vals"]
    res = golden["agg"]["results"]
    for name in ("sum", "min", "max"):
        for init in (NULL64, 0, 2**63 - 1, -(2**63) + 1):
            acc = np.array([init], dtype=np.int64)
            for v in vals:
                getattr(L, f"orc_agg_{name}_skip_val")(acc.ctypes.data, v, NULL64)
            assert int(acc[0]) == res[f"{name}_skip_val_init_{init}"]
            acc = np.array([init], dtype=np.int64)
            for v in vals:
                if v != NULL64:
                    getattr(L, f"orc_agg_{name}")(acc.ctypes.data, v)
            assert int(acc[0]) == res[f"{name}_init_{init}"]
    acc = np.array([0], dtype=np.uint64)
    for v in vals:
        L.orc_agg_count_skip_val(acc.ctypes.data, v, NULL64)
    assert int(acc[0]) == res["count_skip_val"]
    dvals = np.array(golden["agg"]["double_vals_bits"], dtype=np.int64).view(np.float64)
    nulld = float(np.array([A.NULL_DOUBLE_BITS], dtype=np.int64).view(np.float64)[0])
    for name in ("sum", "min", "max"):
        acc = np.array([A.NULL_DOUBLE_BITS], dtype=np.int64)
        for v in dvals:
            getattr(L, f"orc_agg_{name}_double_skip_val")(acc.ctypes.data, float(v), nulld)
        assert int(acc[0]) == res[f"{name}_double_skip_val_bits"]


def test_single_value(oracle, golden):
    """checked_single_agg_id[_int32|_double|_float] (QE/RuntimeFunctions.cpp:489-506,567-583,743-760,799-816): return
    code and slot after every call of the reference's sequences."""
    L = oracle.lib()
    nulld = float(np.array([A.NULL_DOUBLE_BITS], dtype=np.int64).view(np.float64)[0])
    nullf = float(np.array([0x00800000], dtype=np.int32).view(np.float32)[0])
    for rec in golden["single_value"]:
        a64 = np.array([NULL64], dtype=np.int64)
        a32 = np.array([-(2**31)], dtype=np.int32)
        ad = np.array([nulld], dtype=np.float64).view(np.int64).copy()
        af = np.array([nullf], dtype=np.float32).view(np.int32).copy()
        for i, v in enumerate(rec["seq"]):
            isnull = v == NULL64
            assert [L.orc_checked_single_agg_id(a64.ctypes.data, v, NULL64), int(a64[0])] == rec["int64"][i]
            assert [L.orc_checked_single_agg_id_int32(a32.ctypes.data, -(2**31) if isnull else v, -(2**31)), int(a32[0])] == rec["int32"][i]
            assert [L.orc_checked_single_agg_id_double(ad.ctypes.data, nulld if isnull else v * 0.5, nulld), int(ad[0])] == rec["double"][i]
            assert [L.orc_checked_single_agg_id_float(af.ctypes.data, nullf if isnull else v * 0.25, nullf), int(af[0])] == rec["float"][i]


def test_scalar_helpers(oracle, golden):
    L = oracle.lib()
    for s in golden["scalar"]:
        assert L.orc_scale_decimal_down_not_nullable(s["x"], 100, NULL64) == s["scale_down_100"]
        assert L.orc_floor_div_lhs(s["x"], 7) == s["floor_div_7"]
    for e in golden["extract_year"]:
        assert L.orc_extract_year(e["ts"]) == e["year"], e
    for g in golden["logical"]:
        assert L.orc_logical_and(g["l"], g["r"], -128) == g["and"]
        assert L.orc_logical_or(g["l"], g["r"], -128) == g["or"]
        assert L.orc_logical_not(g["l"], -128) == g["not"]


def test_join_probe(oracle, golden):
    L = oracle.lib()
    jp = golden["join_probe"]
    table = np.array(jp["table"], dtype=np.int32)
    for c in jp["cases"]:
        k = c["key"]
        assert L.orc_hash_join_idx(table.ctypes.data, k, jp["min"], jp["max"]) == c["idx"]
        assert L.orc_hash_join_idx_nullable(table.ctypes.data, k, jp["min"], jp["max"], NULL64) == c["nullable"]
        assert L.orc_hash_join_idx_bitwise(table.ctypes.data, k, jp["min"], jp["max"], NULL64, 15) == c["bitwise"]
        assert L.orc_bucketized_hash_join_idx(table.ctypes.data, k, jp["min"], 21, 2) == c["bucketized_2"]


def test_join_probe_bucketized(oracle, golden):
    """bucketized_hash_join_idx[_nullable|_bitwise] (QE/GroupByRuntime.cpp:274-350) on a DATE-shaped key."""
    L = oracle.lib()
    jp = golden["join_probe_bucketized"]
    t = np.array(jp["table"], dtype=np.int32)
    mn, mx, D = jp["min"], jp["max"], jp["bucket"]
    for c in jp["cases"]:
        k = c["key"]
        assert L.orc_bucketized_hash_join_idx(t.ctypes.data, k, mn, mx, D) == c["plain"]
        assert L.orc_bucketized_hash_join_idx_nullable(t.ctypes.data, k, mn, mx, NULL64, D) == c["nullable"]
        assert L.orc_bucketized_hash_join_idx_bitwise(t.ctypes.data, k, mn, mx, NULL64, mx + 1, D) == c["bitwise_max_plus_1"]
        if c["bitwise_probe_arg"] is not None:
            assert L.orc_bucketized_hash_join_idx_bitwise(t.ctypes.data, k, mn, mx, NULL64, mx // D + 1, D) == \
                c["bitwise_probe_arg"]


def test_small_date_decode(oracle, golden):
    L = oracle.lib()
    g = golden["small_date_decode"]
    a4 = np.array([c["v"] for c in g["w4"]], dtype=np.int32)
    a2 = np.array([c["v"] for c in g["w2"]], dtype=np.int16)
    for i, c in enumerate(g["w4"]):
        assert L.orc_fixed_width_small_date_decode(a4.ctypes.data, 4, -(2**31), NULL64, i) == c["out"]
    for i, c in enumerate(g["w2"]):
        assert L.orc_fixed_width_small_date_decode(a2.ctypes.data, 2, -(2**15), NULL64, i) == c["out"]


# ---- QueryEngine/GroupByHashTest.cpp ports ------------------------------------------------------------
def _groups_buffer(entry_count, key_qw_count, init_val=0):
    rsq = key_qw_count + 1
    buf = np.zeros(entry_count * rsq, dtype=np.int64)
    b = buf.reshape(entry_count, rsq)
    b[:, :key_qw_count] = EMPTY64
    b[:, key_qw_count] = init_val
    return buf, rsq


def _ggv(L, buf, n, key, rsq):
    k = np.array(key, dtype=np.int64)
    p = L.orc_get_group_value(buf.ctypes.data, n, k.ctypes.data, len(k), 8, rsq)
    return None if not p else (p - buf.ctypes.data) // 8


def test_groupbyhashtest_set_get(oracle):
    L = oracle.lib()
    buf, rsq = _groups_buffer(10, 1)  # InitTest.OneKey
    assert all(buf[i] == EMPTY64 for i in range(0, buf.size, 2))
    g1 = _ggv(L, buf, 10, [31], rsq)  # SetGetTest.OneKey
    assert g1 is not None and _ggv(L, buf, 10, [31], rsq) == g1
    buf[g1] = 42
    assert buf[_ggv(L, buf, 10, [31], rsq)] == 42
    buf, rsq = _groups_buffer(10, 5)  # SetGetTest.ManyKeys
    key = [31, 32, 33, 34, 35]
    g1 = _ggv(L, buf, 10, key, rsq)
    assert g1 is not None and _ggv(L, buf, 10, key, rsq) == g1
    buf, rsq = _groups_buffer(10, 1)  # SetGetTest.OneKeyCollision
    a = _ggv(L, buf, 10, [31], rsq)
    buf[a] = 32
    b = _ggv(L, buf, 10, [41], rsq)
    buf[b] = 42
    assert buf[_ggv(L, buf, 10, [31], rsq)] == 32 and buf[_ggv(L, buf, 10, [41], rsq)] == 42


def test_groupbyhashtest_full_table(oracle):
    L = oracle.lib()
    for stride in (1, 10):  # OneKeyNoCollisions / OneKeyAllCollisions
        buf, rsq = _groups_buffer(10, 1)
        for i in range(10):
            k = 31 + stride * i
            buf[_ggv(L, buf, 10, [k], rsq)] = k + 100
        for i in range(10):
            k = 31 + stride * i
            assert buf[_ggv(L, buf, 10, [k], rsq)] == k + 100
        assert _ggv(L, buf, 10, [31 + stride * 10], rsq) is None  # full table -> nullptr


# ---- Tests/JoinHashTableTest.cpp:133-260 -----------------------------------------------------------------
def _build_one_to_one(O, keys):
    L = O.lib()
    arr = np.array(keys, dtype=np.int32)
    lo, hi = int(arr.min()), int(arr.max())
    buff = np.empty(hi - lo + 1, dtype=np.int32)
    L.orc_init_hash_join_buff(buff.ctypes.data, buff.size, -1)
    ti = A.JoinColumnTypeInfo(4, lo, hi, A.NULL_INT, 0, A.JC_SIGNED, 0)
    chunks = O.make_join_chunks([arr])
    rc = L.orc_fill_hash_join_buff(buff.ctypes.data, -1, 0, C.cast(chunks, C.c_void_p), 1, C.byref(ti), 1)
    return rc, buff, lo


def test_join_build_one_to_one_kats(oracle):
    rc, buff, lo = _build_one_to_one(oracle, list(range(10)))  # PerfectOneToOne1
    assert rc == 0 and {i + lo: int(v) for i, v in enumerate(buff) if v >= 0} == {i: i for i in range(10)}
    rc, buff, lo = _build_one_to_one(oracle, [0, 1, 2, 4, 5, 6, 7, 9])  # PerfectOneToOne2
    assert rc == 0
    assert {i + lo: int(v) for i, v in enumerate(buff) if v >= 0} == {0: 0, 1: 1, 2: 2, 4: 3, 5: 4, 6: 5, 7: 6, 9: 7}
    rc, _, _ = _build_one_to_one(oracle, [0, 1, 2, 3, 4, 0, 1, 2, 3, 4])  # duplicate -> needs one-to-many
    assert rc == -1


def _build_one_to_many(O, keys):
    L = O.lib()
    arr = np.array(keys, dtype=np.int32)
    lo, hi = int(arr.min()), int(arr.max())
    n = hi - lo + 1
    buff = np.empty(2 * n + arr.size, dtype=np.int32)
    L.orc_init_hash_join_buff(buff.ctypes.data, 2 * n, -1)
    ti = A.JoinColumnTypeInfo(4, lo, hi, A.NULL_INT, 0, A.JC_SIGNED, 0)
    chunks = O.make_join_chunks([arr])
    L.orc_fill_one_to_many_hash_table(buff.ctypes.data, n, -1, C.cast(chunks, C.c_void_p), 1, C.byref(ti), 1)
    pos, cnt, ids = buff[:n], buff[n:2 * n], buff[2 * n:]
    return {k + lo: sorted(int(x) for x in ids[pos[k]:pos[k] + cnt[k]]) for k in range(n) if pos[k] >= 0}


def test_join_build_one_to_many_kats(oracle):
    # PerfectOneToMany1 / PerfectOneToMany2 (DecodedJoinHashBufferSet literals)
    assert _build_one_to_many(oracle, [0, 1, 2, 3, 4, 0, 1, 2, 3, 4]) == {0: [0, 5], 1: [1, 6], 2: [2, 7], 3: [3, 8],
                                                                          4: [4, 9]}
    assert _build_one_to_many(oracle, [0, 2, 3, 4, 0, 2, 3, 4]) == {0: [0, 4], 2: [1, 5], 3: [2, 6], 4: [3, 7]}


# ---- Tests/JoinHashTableTest.cpp:355-440 (Build.KeyedOneToOne / Build.KeyedOneToMany) -----------------
def decode_keyed(buff, entries, kc, width, one_to_one, num_rows):
    """HashTable::toSet for a keyed table: {key tuple: sorted row ids}."""
    dt = np.int32 if width == 4 else np.int64
    empty = np.iinfo(dt).max
    comps = kc + (1 if one_to_one else 0)
    d = buff[:entries * comps * width].view(dt).reshape(entries, comps)
    out = {}
    if one_to_one:
        for e in range(entries):
            if d[e, 0] != empty:
                out[tuple(int(x) for x in d[e, :kc])] = [int(d[e, kc])]
        return out
    otm = buff[entries * comps * width:].view(np.int32)
    pos, cnt, ids = otm[:entries], otm[entries:2 * entries], otm[2 * entries:2 * entries + num_rows]
    for e in range(entries):
        if d[e, 0] != empty:
            out[tuple(int(x) for x in d[e, :kc])] = sorted(int(x) for x in ids[pos[e]:pos[e] + cnt[e]])
    return out


def keyed_columns(O, arrays, width):
    kc = len(arrays)
    jcs, tis, keep = (A.JoinColumn * kc)(), (A.JoinColumnTypeInfo * kc)(), []
    for k, a in enumerate(arrays):
        chunks = O.make_join_chunks([a])
        keep.append(chunks)
        jcs[k] = A.JoinColumn(C.cast(chunks, C.c_void_p).value, C.sizeof(chunks), 1, len(a), a.dtype.itemsize)
        tis[k] = A.JoinColumnTypeInfo(a.dtype.itemsize, int(a.min()), int(a.max()),
                                      int(np.iinfo(a.dtype).min), 0, A.JC_SIGNED, 0)
    return jcs, tis, keep


def test_join_build_keyed_kats(oracle):
    L = oracle.lib()
    # KeyedOneToOne: table2.b = 0,1,3 joined on (b, b): "| keyed one-to-one | keys * (1,1,1) (3,3,2) (0,0,0) * * |"
    b = np.array([0, 1, 3], dtype=np.int32)
    entries = 2 * len(b)
    jcs, tis, keep = keyed_columns(oracle, [b, b], 4)
    buff = np.empty(entries * 3 * 4, dtype=np.uint8)
    L.orc_init_baseline_hash_join_buff(buff.ctypes.data, entries, 2, 4, 1, -1)
    assert L.orc_fill_baseline_hash_join_buff(buff.ctypes.data, entries, -1, 2, 4, jcs, tis) == 0
    assert decode_keyed(buff, entries, 2, 4, True, 3) == {(0, 0): [0], (1, 1): [1], (3, 3): [2]}
    # KeyedOneToMany: b = 0,1,3,3: "keys (1,1) (3,3) (0,0) | offsets 0 1 3 | counts 1 2 1 | payloads 1 2 3 0"
    b = np.array([0, 1, 3, 3], dtype=np.int32)
    entries = 2 * len(b)
    jcs, tis, keep = keyed_columns(oracle, [b, b], 4)
    buff = np.empty(entries * 3 * 4, dtype=np.uint8)
    L.orc_init_baseline_hash_join_buff(buff.ctypes.data, entries, 2, 4, 1, -1)
    assert L.orc_fill_baseline_hash_join_buff(buff.ctypes.data, entries, -1, 2, 4, jcs, tis) == -1  # duplicate
    buff = np.empty(entries * 2 * 4 + (2 * entries + 4) * 4, dtype=np.uint8)
    L.orc_init_baseline_hash_join_buff(buff.ctypes.data, entries, 2, 4, 0, -1)
    assert L.orc_fill_one_to_many_baseline_hash_table(buff.ctypes.data, entries, -1, 2, 4, jcs, tis) == 0
    assert decode_keyed(buff, entries, 2, 4, False, 4) == {(0, 0): [0], (1, 1): [1], (3, 3): [2, 3]}


# ---- Tests/NoCatalogRelAlgTest.cpp:211-232 ----------------------------------------------------------------
def test_nocatalog_group_by_single_column(oracle):
    import pyarrow as pa
    from hdk_amd import result_set as rs
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.storage import ArrowStorage
    from util import run_oracle
    st = ArrowStorage()
    at = pa.table({"k": pa.array([1, 2, 1, 2, 1, 2, 1, 3, 1, 3], pa.int32()),
                   "v": pa.array([10, 20, 30, 40, 50, None, 70, None, 90, 100], pa.int32())})
    st.import_arrow(at, "test_agg", fragment_size=5)
    for columnar in (False, True):
        q = QueryUnit("test_agg", groupby=[ColRef("k")], output_columnar=columnar,
                      targets=[KeyRef(0, "k"), Agg("count", None, "c"), Agg("count", ColRef("v"), "cv"),
                               Agg("sum", ColRef("v"), "s"), Agg("avg", ColRef("v"), "a")])
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == 0
        assert rs.to_columns(cp, buf) == {"k": [1, 2, 3], "c": [5, 3, 2], "cv": [5, 2, 1], "s": [250, 60, 100],
                                          "a": [50.0, 30.0, 100.0]}
