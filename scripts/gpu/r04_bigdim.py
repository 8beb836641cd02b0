#!/usr/bin/env python3
"""Round 4: the join beyond the old 10.2 M-key cliff.  C3 (SUM(val + dval)) and C3g over dimensions of 10 M / 12 M / 100 M
rows and a sparse one (5 M rows over a 50 M key range), 1 B fact rows (or --rows), kernel times from HIP events."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--dims", default="10000000,12000000,100000000,sparse")
    ap.add_argument("--queries", default="c3,c3g")
    args = ap.parse_args()
    import torch
    from hdk_amd import _abi as A
    from hdk_amd._lib import check, lib
    from hdk_amd.hip_mgr import HipMgr
    from workloads import Workload
    mgr = HipMgr()
    L = lib()
    out = []
    for d in args.dims.split(","):
        for qn in args.queries.split(","):
            kw = {"dim_rows": 5_000_000, "dim_key_stride": 10} if d == "sparse" else {"dim_rows": int(d)}
            w = Workload(qn, args.rows, 0, mgr, **kw)
            cp = w.compiled
            o = torch.empty(max(cp.buffer_quads, 1), dtype=torch.int64, device="cuda")
            step = w.ex.prepare(cp, w.frag_ids, flags=A.LAUNCH_RECORD_EVENTS, out_ptr=o.data_ptr())
            n = C.c_int32(0)
            for _ in range(2):
                step.enqueue()
            mgr.synchronizeStream(0)
            check(L.hdk_hip_collect_scan_times(0, None, 0, C.byref(n)))
            for _ in range(5):
                step.enqueue()
            mgr.synchronizeStream(0)
            ms = (C.c_float * 8)()
            check(L.hdk_hip_collect_scan_times(0, ms, 8, C.byref(n)))
            t = float(np.mean([ms[i] for i in range(n.value)]))
            err = int(mgr.to_host(step.d_err.ptr, 4, 0, np.int32)[0])
            ref = w.reference_checks()
            ok = None
            if qn == "c3":
                ok = (int(o[0].item()) - ref["sum_val_plus_dval"]) % (1 << 64) == 0
            elif qn == "c3g":
                from hdk_amd.executor import ExecutionResult
                cols = ExecutionResult(cp, o.cpu().numpy(), cp.entry_count).to_columns()
                ok = all((s_ - ref["group_sums"][g]) % (1 << 64) == 0 for g, s_ in zip(cols["g"], cols["s"])) and len(cols["g"]) == 64
            rec = {"dim": d, "query": qn, "rows": args.rows, "kernels": step.kernel_names(), "ms": t, "rows_per_s": args.rows / t * 1e3,
                   "err": err, "check": bool(ok)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            step.free()
            del w, o
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04", "bigdim.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
