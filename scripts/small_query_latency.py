#!/usr/bin/env python3
"""Step latency of small inputs (BASELINE C1: 1 M rows), kernel-by-kernel launches vs a recorded hipGraph."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from hdk_amd.executor import Executor
    from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
    from hdk_amd.storage import ArrowStorage
    n = 1_000_000
    rng = np.random.default_rng(1)
    st = ArrowStorage()
    st.import_numpy("t", {"a": np.arange(n, dtype=np.int64), "k": rng.integers(0, 64, n, dtype=np.int64)})
    ex = Executor(st, 0)
    for name, q in (("c1 SUM(a)", QueryUnit("t", targets=[Agg("sum", ColRef("a"))])),
                    ("c2 GROUP BY k SUM(a)", QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("a"))]))):
        step = ex.prepare(q)
        for label, fn in (("launches", lambda: (step.init_output(), step.launch())), ("hipGraph", None)):
            if fn is None:
                step.capture_graph()
                fn = step.replay
            for _ in range(20):
                fn()
            step.mgr.synchronizeStream(0)
            t0 = time.perf_counter()
            reps = 2000
            for _ in range(reps):
                fn()
            step.mgr.synchronizeStream(0)
            print(f"{name:24s} {label:9s} {(time.perf_counter() - t0) / reps * 1e6:7.1f} us/step")
        res = step.fetch().to_columns()
        assert (res.get("sum_0") or res.get("sum_1"))
        step.free()


if __name__ == "__main__":
    main()
