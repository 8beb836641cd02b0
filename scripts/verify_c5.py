#!/usr/bin/env python3
"""Full-size cross-check of the two open-addressing strategies (no oracle at this size): the radix-partitioned
path and the global-atomics kernel must produce the same {key -> SUM, COUNT} set on the C5 shape.

    python scripts/verify_c5.py [--rows 256000000]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256_000_000)
    args = ap.parse_args()
    from hdk_amd import _abi as A
    from hdk_amd.executor import Executor
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.storage import ArrowStorage
    n = args.rows
    rng = np.random.default_rng(20261002)
    st = ArrowStorage()
    st.import_numpy("t", {"hk": rng.integers(0, max(n // 10, 1000), n, dtype=np.int64),
                          "val": rng.integers(-2**31, 2**31, n, dtype=np.int64)}, fragment_size=32_000_000)
    q = QueryUnit("t", groupby=[ColRef("hk")], force_baseline=True,
                  targets=[KeyRef(0, "k"), Agg("sum", ColRef("val"), "s"), Agg("count", None, "c")])
    ex = Executor(st, 0)
    cp = ex.compile(q)
    rq = int(cp.plan.row_size_quad)
    out = {}
    for name, flags in (("partitioned", 0), ("atomics", A.LAUNCH_FORCE_GLOBAL_ATOMICS)):
        step = ex.prepare(cp, flags=flags)
        res = step.run()
        rows = res.buffer.view(np.int64).reshape(-1, rq)
        kw = int(cp.plan.key_width)
        keys = (rows[:, 0] & 0xFFFFFFFF).astype(np.uint32).view(np.int32) if kw == 4 else rows[:, 0]
        live = keys != (np.iinfo(np.int32).max if kw == 4 else np.iinfo(np.int64).max)
        m = np.stack([keys[live].astype(np.int64)] + [rows[live, 1 + i] for i in range(rq - 1)], axis=1)
        out[name] = (step.kernel_names().split(",")[0], m[np.argsort(m[:, 0], kind="stable")])
        step.free()
    (k1, a), (k2, b) = out["partitioned"], out["atomics"]
    same = a.shape == b.shape and bool(np.array_equal(a, b))
    hk = np.concatenate(st.get("t").columns["hk"].fragments)
    val = np.concatenate(st.get("t").columns["val"].fragments)
    print({"rows": n, "kernels": [k1, k2], "groups": int(a.shape[0]), "identical": same,
           "groups_expected": int(np.unique(hk).size), "total_sum_ok": bool(int(a[:, 1].sum()) == int(val.sum())),
           "total_count_ok": bool(int(a[:, 2].sum()) == n)})
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
