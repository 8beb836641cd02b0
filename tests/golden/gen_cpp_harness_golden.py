"""Expected output of tests/cpp/harness.cpp, computed with numpy from the harness's data definition (a splitmix64
finaliser of the row number), independent of every line of device code.
    python tests/golden/gen_cpp_harness_golden.py > tests/golden/cpp_harness_output.txt
    python tests/golden/gen_cpp_harness_golden.py multi_device > tests/golden/cpp_multi_device_output.txt"""
import numpy as np

NULL = np.iinfo(np.int64).min
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix(x):
    x = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x ^ (x >> np.uint64(31))


def gen_val(i):
    v = (mix(i + np.uint64(1 << 40)) % np.uint64(2000001)).astype(np.int64) - 1000000
    return np.where(mix(i + np.uint64(1 << 41)) % np.uint64(32) == 0, NULL, v)


def expected_lines():
    out = []
    with np.errstate(over="ignore"):
        i = np.arange(4 * 500000, dtype=np.uint64)
        key = (mix(i) % np.uint64(64)).astype(np.int64)
        val = gen_val(i)
        out.append("c2 error_code 0")
        for k in range(64):
            sel = key == k
            v = val[sel]
            v = v[v != NULL]
            s = int(v.sum()) if len(v) else NULL
            out.append(f"c2 key {k} sum {s} count {int(sel.sum())}")
        d = np.arange(1000, dtype=np.uint64)
        dkey = ((d * np.uint64(37)) % np.uint64(1000)).astype(np.int64)
        dval = (mix(d + np.uint64(1 << 42)) % np.uint64(1000)).astype(np.int64)
        dval_by_key = np.zeros(1000, dtype=np.int64)
        dval_by_key[dkey] = dval
        j = np.arange(3 * 400000, dtype=np.uint64)
        h = mix(j + np.uint64(1 << 43))
        fk = np.where((h >> np.uint64(32)) % np.uint64(64) == 0, NULL, (h % np.uint64(1200)).astype(np.int64))
        fval = gen_val(j + np.uint64(1 << 44))
        match = (fk != NULL) & (fk >= 0) & (fk < 1000)
        add = fval[match] + dval_by_key[fk[match]]
        add = add[fval[match] != NULL]
        out.append("join build_error 0")
        out.append("join fused_build 1")
        out.append("join error_code 0")
        out.append(f"join sum {int(add.sum())} count {int(match.sum())}")
        out.append("interrupt error_code 10")
        # ---- the steps whose plans come out of the extractor ----
        i = np.arange(3 * 400000, dtype=np.uint64)
        pc = (mix(i + np.uint64(1 << 45)) % np.uint64(7)).astype(np.int64)
        ts = np.int64(1230768000) + (mix(i + np.uint64(1 << 46)) % np.uint64(220838400)).astype(np.int64)
        year = ts.astype("datetime64[s]").astype("datetime64[Y]").astype(np.int64) + 1970
        out.append("q3 error_code 0")
        pairs, counts = np.unique(np.stack([pc, year], axis=1), axis=0, return_counts=True)
        for (p_, y_), c_ in zip(pairs.tolist(), counts.tolist()):
            out.append(f"q3 passenger_count {p_} year {y_} count {c_}")
        i = np.arange(8 * 1100000, dtype=np.uint64)
        key = (mix(i + np.uint64(1 << 47)) % np.uint64(3000000)).astype(np.int64)
        val = (mix(i + np.uint64(1 << 48)) % np.uint64(2000001)).astype(np.int64) - 1000000
        uk, inv = np.unique(key, return_inverse=True)
        sums = np.zeros(len(uk), dtype=np.int64)
        np.add.at(sums, inv, val)
        mixed = np.bitwise_xor.reduce(mix((uk.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + sums.astype(np.uint64)) & M64))
        out.append("c5 error_code 0")
        out.append(f"c5 groups {len(uk)} sum_of_sums {int(sums.sum())} checksum {int(mixed)}")
        i = np.arange(2 * 500000, dtype=np.uint64)
        key = (mix(i) % np.uint64(64)).astype(np.int64)
        val = gen_val(i)
        sel = (val != NULL) & (val < -900000)
        out.append(f"projection error_code 0 matched {int(sel.sum())}")
        pos = np.arange(2 * 500000, dtype=np.int64) % 500000  # the row position inside its fragment (`pos` of the row function)
        out.append(f"projection sum_pos {int(pos[sel].sum())} sum_key {int(key[sel].sum())} sum_v2 {int((val[sel] * 2).sum())}")
        i = np.arange(2 * 300000, dtype=np.uint64)
        k = (mix(i + np.uint64(1 << 49)) % np.uint64(10)).astype(np.int64)
        h = mix(i + np.uint64(1 << 50))
        isnull = h % np.uint64(16) == 0
        x = ((h >> np.uint64(8)) % np.uint64(16)).astype(np.int64)
        out.append("floats error_code 0")
        for g in range(10):
            m = (k == g) & ~isnull
            s = float(x[m].sum())
            out.append(f"floats key {g} sum {s:.1f} avg_sum {s:.1f} avg_count {int(m.sum())} count {int(m.sum())}")
        i = np.arange(4 * 500000, dtype=np.uint64)
        key = (mix(i) % np.uint64(64)).astype(np.int64)
        val = gen_val(i)
        out.append("reduce error_code 0")
        for g in range(64):
            sel = key == g
            v = val[sel]
            v = v[v != NULL]
            out.append(f"reduce key {g} sum {int(v.sum()) if len(v) else NULL} count {int(sel.sum())}")
    return out


def multi_device_lines():
    """Expected output of tests/cpp/multi_device.cpp: the c2 and c5 steps' data spread over the devices of a node and merged
    over RCCL -- the same groups whatever the number of devices."""
    out = []
    for ln in expected_lines():
        if ln.startswith("c2 ") or ln.startswith("c5 "):
            out.append("mg_" + ln)
    return out


if __name__ == "__main__":
    import sys
    print("\n".join(multi_device_lines() if sys.argv[1:] == ["multi_device"] else expected_lines()))
