#!/usr/bin/env python3
"""Round 4: one-to-one join table build, atomics (k_join_build + k_build_fused) against slot-range partitions
(join_build_part.h): a permutation dimension of --rows keys (and a sparse one, keys x 10), table alone and table + fused
form with one 8-byte payload column.  Wall time around a synchronised call, best of 3."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="10000000,100000000")
    args = ap.parse_args()
    import torch
    from hdk_amd import _abi as A
    from hdk_amd._lib import check, lib
    from hdk_amd.hip_mgr import HipMgr
    mgr = HipMgr()
    L = lib()
    out = []
    for rows in [int(r) for r in args.rows.split(",")]:
        for stride in (1, 10):
            if stride == 10 and rows > 20_000_000:
                continue
            g = torch.Generator(device="cuda")
            g.manual_seed(5)
            key = torch.randperm(rows, dtype=torch.int64, device="cuda", generator=g) * stride
            pay = torch.randint(0, 10**6, (rows,), dtype=torch.int64, device="cuda", generator=g)
            n = (rows - 1) * stride + 1
            chunk = A.JoinChunk()
            chunk.col_buff = key.data_ptr()
            chunk.num_elems = rows
            chunk.row_id = 0
            raw = np.frombuffer(bytes(chunk), dtype=np.uint8)
            d_chunks = mgr.to_device(raw, 0)
            jc = A.JoinColumn(d_chunks.ptr, raw.nbytes, 1, rows, 8)
            ti = A.JoinColumnTypeInfo(8, 0, n - 1, A.NULL_BIGINT, 0, A.JC_SIGNED, 0)
            table = torch.empty(n, dtype=torch.int32, device="cuda")
            fused = torch.empty(2 * n, dtype=torch.int64, device="cuda")
            d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
            ptrs = (C.c_void_p * 1)(pay.data_ptr())
            widths = (C.c_int32 * 1)(8)
            kinds = (C.c_int32 * 1)(A.COL_INT)
            sb = L.hdk_hip_join_build_scratch_bytes(rows, n, 1)
            scratch = torch.empty(max(sb, 8), dtype=torch.uint8, device="cuda")
            res = {}
            keep = {}
            for mode in ("atomics", "partitioned"):
                os.environ["HDK_HIP_BUILD_PARTITION_MIN_ROWS"] = "0" if mode == "atomics" else "1"
                for what in ("table", "table+fused"):
                    best = 1e9
                    for _ in range(4):
                        d_err.zero_()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        check(L.hdk_hip_init_hash_join_buff(table.data_ptr(), n, -1, 0, None))
                        if what == "table":
                            check(L.hdk_hip_fill_hash_join_buff(table.data_ptr(), -1, 0, d_err.data_ptr(), jc, ti, 0, None))
                        else:
                            check(L.hdk_hip_fill_hash_join_buff_fused(table.data_ptr(), -1, 0, d_err.data_ptr(), jc, ti, 1, ptrs, widths,
                                                                      kinds, 1, fused.data_ptr(), scratch.data_ptr(), sb, 0, None))
                        mgr.synchronizeStream(0)
                        best = min(best, (time.perf_counter() - t0) * 1e3)
                    res[f"{mode}:{what}"] = best
                    assert int(d_err.item()) == 0
                    keep[f"{mode}:{what}"] = (int(table.to(torch.int64).sum().item()), int(fused.sum().item()) if what != "table" else 0)
            same = keep["atomics:table"] == keep["partitioned:table"] and keep["atomics:table+fused"] == keep["partitioned:table+fused"]
            rec = {"rows": rows, "slots": n, "scratch_GB": sb / 1e9, "ms": res, "same_checksums": same}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del key, pay, table, fused, scratch
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04", "join_build.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
