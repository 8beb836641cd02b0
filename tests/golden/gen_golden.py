#!/usr/bin/env python3
"""Generates tests/golden/ref_runtime_vectors.json from the REFERENCE's own compiled runtime
(oracle/_ref/libhdk_ref_runtime.so = /root/reference/omniscidb/QueryEngine/RuntimeFunctions.cpp +
MurmurHash.cpp + Utils/ExtractFromTime.cpp built in place by oracle/Makefile).

Run in the build container (where /root/reference exists):  python tests/golden/gen_golden.py
The output is DATA ONLY: inputs and the reference's outputs.  It travels to the GPU box, the
reference does not.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

EMPTY64 = 2**63 - 1
NULL64 = -(2**63)


def main():
    R = O.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libhdk_ref_runtime.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(20261002)
    out = {"source": "reference runtime compiled from /root/reference (oracle/Makefile target `ref`)"}

    # ---- hashes ------------------------------------------------------------------------------
    keys = [0, 1, 63, -1, 2**31, -(2**31), EMPTY64, NULL64] + [int(x) for x in rng.integers(-2**62, 2**62, 24)]
    h = []
    for k in keys:
        a = np.array([k], dtype=np.int64)
        h.append({"key": k, "murmur3_8": R.MurmurHash3(a.ctypes.data, 8, 0), "murmur1_8": R.MurmurHash1(a.ctypes.data, 8, 0),
                  "murmur64a_8": R.MurmurHash64A(a.ctypes.data, 8, 0),
                  "murmur3_4": R.MurmurHash3(a.view(np.int32).ctypes.data, 4, 0),
                  "key_hash": R.key_hash(a.ctypes.data, 1, 8)})
    out["hash"] = h
    multi = []
    for n in (2, 3, 5):
        a = rng.integers(-1000, 1000, n).astype(np.int64)
        multi.append({"key": a.tolist(), "key_hash": R.key_hash(a.ctypes.data, n, 8),
                      "key_hash_w4": R.key_hash(a.astype(np.int32).ctypes.data, n, 4)})
    out["hash_multi"] = multi

    # ---- baseline group-by buffers: insert sequence -> final buffer ------------------------------
    gb = []
    for entry_count, nkeys, kw in ((10, 1, 8), (17, 2, 8), (16, 1, 4), (13, 3, 4)):
        key_quads = (nkeys * kw + 7) // 8
        rsq = key_quads + 1
        buf = np.zeros(entry_count * rsq, dtype=np.int64)
        if kw == 8:
            buf.reshape(entry_count, rsq)[:, :nkeys] = EMPTY64
        else:
            kb = buf.view(np.int32).reshape(entry_count, rsq * 2)
            kb[:, :nkeys] = 2**31 - 1
        seq, slots = [], []
        for i in range(entry_count + 3):
            key = rng.integers(0, 6, nkeys).astype(np.int64 if kw == 8 else np.int32)
            p = R.get_group_value(buf.ctypes.data, entry_count, key.ctypes.data, nkeys, kw, rsq)
            off = -1 if not p else (p - buf.ctypes.data) // 8
            if p:
                buf[off] += int(key.sum()) + 1
            seq.append(key.tolist())
            slots.append(int(off))
        gb.append({"entry_count": entry_count, "key_count": nkeys, "key_width": kw, "row_size_quad": rsq,
                   "keys": seq, "slot_quads": slots, "final": buf.tolist()})
    out["get_group_value"] = gb

    col = []
    for entry_count, nkeys in ((11, 1), (19, 2)):
        buf = np.zeros(entry_count * (nkeys + 1), dtype=np.int64)
        buf[:entry_count * nkeys] = EMPTY64
        seq, slots = [], []
        for i in range(entry_count + 2):
            key = rng.integers(0, 5, nkeys).astype(np.int64)
            s = R.get_group_value_columnar_slot(buf.ctypes.data, entry_count, key.ctypes.data, nkeys, 8)
            if s >= 0:
                buf[entry_count * nkeys + s] += 1
            seq.append(key.tolist())
            slots.append(int(s))
        col.append({"entry_count": entry_count, "key_count": nkeys, "keys": seq, "slots": slots, "final": buf.tolist()})
    out["get_group_value_columnar_slot"] = col

    # ---- perfect hash ---------------------------------------------------------------------------------
    fast = []
    for min_key, bucket, rsq in ((0, 0, 2), (-5, 0, 3), (100, 10, 2)):
        n = 12
        buf = np.zeros(n * rsq, dtype=np.int64)
        buf.reshape(n, rsq)[:, 0] = EMPTY64
        seq = []
        for i in range(30):
            k = int(min_key + (rng.integers(0, n) * (bucket or 1)))
            p = R.get_group_value_fast(buf.ctypes.data, k, min_key, bucket, rsq)
            buf[(p - buf.ctypes.data) // 8] += 7
            seq.append(k)
        fast.append({"min_key": min_key, "bucket": bucket, "row_size_quad": rsq, "entries": n, "keys": seq,
                     "final": buf.tolist()})
    out["get_group_value_fast"] = fast

    # ---- aggregates -------------------------------------------------------------------------------------
    agg = {}
    vals = [5, NULL64, 7, -3, NULL64, 2**40, -2**40, 0]
    for name in ("sum", "min", "max"):
        for init in (NULL64, 0, 2**63 - 1, -(2**63) + 1):
            acc = np.array([init], dtype=np.int64)
            for v in vals:
                getattr(R, f"agg_{name}_skip_val")(acc.ctypes.data, v, NULL64)
            agg[f"{name}_skip_val_init_{init}"] = int(acc[0])
            acc = np.array([init], dtype=np.int64)
            for v in vals:
                if v != NULL64:
                    getattr(R, f"agg_{name}")(acc.ctypes.data, v)
            agg[f"{name}_init_{init}"] = int(acc[0])
    acc = np.array([0], dtype=np.uint64)
    for v in vals:
        R.agg_count_skip_val(acc.ctypes.data, v, NULL64)
    agg["count_skip_val"] = int(acc[0])
    nulld = np.array([0x0010000000000000], dtype=np.int64).view(np.float64)[0]
    dvals = [1.5, float(nulld), -2.25, 1e300, -1e300, float(nulld), 3.0]
    for name in ("sum", "min", "max"):
        acc = np.array([nulld], dtype=np.float64)
        for v in dvals:
            getattr(R, f"agg_{name}_double_skip_val")(acc.view(np.int64).ctypes.data, v, float(nulld))
        agg[f"{name}_double_skip_val_bits"] = int(acc.view(np.int64)[0])
    out["agg"] = {"int_vals": vals, "double_vals_bits": [int(np.float64(x).view(np.int64)) for x in dvals], "results": agg}

    # ---- SINGLE_VALUE: checked_single_agg_id and its typed forms (return code and slot after every call) ------------
    nullf = float(np.array([0x00800000], dtype=np.int32).view(np.float32)[0])
    seqs = [[NULL64, 5, 5, NULL64, 5], [7, 8], [NULL64, NULL64], [3, NULL64, 3, 4, 3], [0, 0, NULL64, 1]]
    single = []
    for seq in seqs:
        rec = {"seq": seq, "int64": [], "int32": [], "double": [], "float": []}
        a64 = np.array([NULL64], dtype=np.int64)
        a32 = np.array([-(2**31)], dtype=np.int32)
        ad = np.array([nulld], dtype=np.float64).view(np.int64).copy()
        af = np.array([nullf], dtype=np.float32).view(np.int32).copy()
        for v in seq:
            isnull = v == NULL64
            rec["int64"].append([int(R.checked_single_agg_id(a64.ctypes.data, v, NULL64)), int(a64[0])])
            rec["int32"].append([int(R.checked_single_agg_id_int32(a32.ctypes.data, -(2**31) if isnull else v, -(2**31))), int(a32[0])])
            rec["double"].append([int(R.checked_single_agg_id_double(ad.ctypes.data, float(nulld) if isnull else v * 0.5, float(nulld))), int(ad[0])])
            rec["float"].append([int(R.checked_single_agg_id_float(af.ctypes.data, nullf if isnull else v * 0.25, nullf)), int(af[0])])
        single.append(rec)
    out["single_value"] = single

    # ---- scalar helpers -------------------------------------------------------------------------------------
    sc = []
    for x in [0, 1, 49, 50, 51, -49, -50, -51, 12345, -12345, 2**40 + 3]:
        sc.append({"x": x, "scale_down_100": R.scale_decimal_down_not_nullable(x, 100, NULL64),
                   "floor_div_7": R.floor_div_lhs(x, 7)})
    out["scalar"] = sc
    ts = [0, 1, 86399, 86400, 951782400, 1375344877, 1230768000, 1451606399, 1451606400, 4102444800, 4102444799,
          -1, -86400, -2208988800, 253402300799, 2**32, 2085978495, 2085978496]
    out["extract_year"] = [{"ts": t, "year": R.extract_year(t)} for t in ts]
    NB = -128
    out["logical"] = [{"l": l, "r": r, "and": R.logical_and(l, r, NB), "or": R.logical_or(l, r, NB),
                       "not": R.logical_not(l, NB)} for l in (0, 1, NB) for r in (0, 1, NB)]
    ar = []
    for a, b in [(3, 4), (NULL64, 4), (3, NULL64), (-7, 2), (7, -2), (2**62, 2**62)]:
        ar.append({"a": a, "b": b, "add": R.add_int64_t_nullable(a, b, NULL64), "sub": R.sub_int64_t_nullable(a, b, NULL64),
                   "mul": R.mul_int64_t_nullable(a, b, NULL64) if abs(a) < 2**31 or a == NULL64 else None,
                   "div": R.div_int64_t_nullable(a, b, NULL64), "mod": R.mod_int64_t_nullable(a, b, NULL64),
                   "lt": R.lt_int64_t_nullable(a, b, NULL64, NB), "eq": R.eq_int64_t_nullable(a, b, NULL64, NB)})
    out["arith"] = ar

    # ---- join probe ---------------------------------------------------------------------------------------------
    table = np.array([3, -1, 0, 5, -1, 9], dtype=np.int32)
    jp = []
    for k in (9, 10, 12, 15, 16, 100, NULL64):
        jp.append({"key": k, "idx": R.hash_join_idx(table.ctypes.data, k, 10, 15),
                   "nullable": R.hash_join_idx_nullable(table.ctypes.data, k, 10, 15, NULL64),
                   "bitwise": R.hash_join_idx_bitwise(table.ctypes.data, k, 10, 15, NULL64, 15),
                   "bucketized_2": R.bucketized_hash_join_idx(table.ctypes.data, k, 10, 21, 2)})
    out["join_probe"] = {"table": table.tolist(), "min": 10, "max": 15, "cases": jp}

    # ---- bucketized probes (DATE keys: bucket_normalization = 86400) and the small-date decoder -------------------
    D = 86400
    btable = np.array([4, -1, 2, 7, -1, 0, 11], dtype=np.int32)  # 5 day slots + the slot(s) a translated NULL lands in
    mn, mx = 100 * D, 104 * D
    bp = []
    for k in (99 * D, 100 * D, 100 * D + 1, 102 * D, 104 * D, 104 * D + 5, 105 * D, NULL64):
        bp.append({"key": k, "plain": R.bucketized_hash_join_idx(btable.ctypes.data, k, mn, mx, D),
                   "nullable": R.bucketized_hash_join_idx_nullable(btable.ctypes.data, k, mn, mx, NULL64, D),
                   # the two translated NULLs of the reference: what the probe passes (max / bucket + 1,
                   # PerfectJoinHashTable.cpp:805-807) and what the build fills (max + 1, PerfectHashTableBuilder.h:100-106)
                   "bitwise_probe_arg": R.bucketized_hash_join_idx_bitwise(btable.ctypes.data, k, mn, mx, NULL64, mx // D + 1, D)
                   if mx // D + 1 >= mn else None,
                   "bitwise_max_plus_1": R.bucketized_hash_join_idx_bitwise(btable.ctypes.data, k, mn, mx, NULL64, mx + 1, D)})
    out["join_probe_bucketized"] = {"table": btable.tolist(), "min": mn, "max": mx, "bucket": D, "cases": bp}
    days = np.array([0, 1, -1, 19000, -(2**31), 2**31 - 1, -25567], dtype=np.int32)
    days16 = np.array([0, 1, -1, 19000, -(2**15), 2**15 - 1], dtype=np.int16)
    out["small_date_decode"] = {
        "w4": [{"v": int(v), "out": R.fixed_width_small_date_decode(days.ctypes.data, 4, -(2**31), NULL64, i)}
               for i, v in enumerate(days)],
        "w2": [{"v": int(v), "out": R.fixed_width_small_date_decode(days16.ctypes.data, 2, -(2**15), NULL64, i)}
               for i, v in enumerate(days16)]}

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_runtime_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
