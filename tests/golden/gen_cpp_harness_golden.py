"""Expected output of tests/cpp/harness.cpp, computed with numpy from the harness's data definition (a splitmix64
finaliser of the row number), independent of every line of device code.
    python tests/golden/gen_cpp_harness_golden.py > tests/golden/cpp_harness_output.txt"""
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
        out.append("join error_code 0")
        out.append(f"join sum {int(add.sum())} count {int(match.sum())}")
        out.append("interrupt error_code 10")
    return out


if __name__ == "__main__":
    print("\n".join(expected_lines()))
