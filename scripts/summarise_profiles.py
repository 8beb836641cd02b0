#!/usr/bin/env python3
"""gpurun_out/<round>/ (scripts/gpu/profile_<round>.sh) -> profiles/<round>_*  (round = argv[1], default r03): the kernel-stats CSV and bench line of the driver's
bench command, and one PMC summary per config with the HBM traffic of a launch.

Counters follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes (kernel trace only); both are in KB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so
bytes read = 2 x 1024 x FETCH_SIZE; WRITE_SIZE is taken as reported (1024 x).  `traffic_bytes_per_launch` sums the
kernels that run once per launch (init kernel, scan passes, finalize), weighted by their dispatch counts."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
ARGS = {"c3big": "--config c3 --dim-rows 100000000"}  # name -> bench arguments when they are not just --config <name>
SRC = os.path.join(ROOT, "gpurun_out", RND)
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from workloads import CONFIGS  # noqa: E402  (name -> (rows, algorithmic bytes per row, description))
ALG = {name: (bpr, rows) for name, (rows, bpr, _d) in CONFIGS.items()}
ALG["c3big"] = (16, 1_000_000_000)


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "hdk" not in k or "k_cal" in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return {k: ({c: v / len(calls[k]) for c, v in acc[k].items()}, len(calls[k])) for k in acc}


def short(k):
    return k.replace("void ", "").replace("hdk::", "").split("(")[0]


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "bench_default_kernel_stats.csv"), os.path.join(DST, f"{RND}_bench_default_kernel_stats.csv"))
    line = [l for l in open(os.path.join(SRC, "bench_default_bench.json")) if l.startswith("{")][-1]
    with open(os.path.join(DST, f"{RND}_bench_default.json"), "w") as f:
        f.write(line)
    for extra in ("bench_no_e2e_kernel_stats.csv", "bench_detail.json"):  # (round 6: the headline's side file travels with it)
        if os.path.exists(os.path.join(SRC, extra)):
            shutil.copy(os.path.join(SRC, extra), os.path.join(DST, f"{RND}_{extra}"))
    for name, (bpr, rows) in ALG.items():
        if not os.path.exists(os.path.join(SRC, f"{name}_fetch_counters.csv")):
            continue
        fe = counters(os.path.join(SRC, f"{name}_fetch_counters.csv"))
        wr = counters(os.path.join(SRC, f"{name}_write_counters.csv"))
        main_k = max(fe, key=lambda k: fe[k][0].get("FETCH_SIZE", 0) * fe[k][1])
        n_main = fe[main_k][1]
        kernels, traffic = {}, 0.0
        for k in sorted(fe, key=lambda k: -fe[k][0].get("FETCH_SIZE", 0)):
            f_kb, calls = fe[k][0].get("FETCH_SIZE", 0.0), fe[k][1]
            w_kb = wr.get(k, ({}, 0))[0].get("WRITE_SIZE", 0.0)
            if calls < n_main:
                continue  # one-off work (join-table build, merges of the checks), not part of a launch
            rd, wb = 2 * 1024 * f_kb, 1024 * w_kb
            kernels[short(k)] = {"dispatches": calls, "FETCH_SIZE_KB": round(f_kb), "WRITE_SIZE_KB": round(w_kb),
                                 "read_bytes": round(rd), "written_bytes": round(wb)}
            traffic += (rd + wb) * calls / n_main
        cfg = ARGS.get(name, f"--config {name}")
        out = {"what": f"bench.py {cfg} (BASELINE size: {rows} rows on one GPU), averages per dispatch",
               "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py {cfg} --steps 5 --warmup 2 "
                          "--no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --extra none   (one counter per pass)",
               "correction": "gfx950: read bytes = 2 x 1024 x FETCH_SIZE (128-byte requests tallied as 64); written bytes = 1024 x WRITE_SIZE",
               "rows": rows, "algorithmic_bytes_per_row": bpr, "algorithmic_bytes": bpr * rows, "kernels": kernels,
               "traffic_bytes_per_launch": round(traffic), "traffic_bytes_per_row": round(traffic / rows, 2),
               "traffic_over_algorithmic": round(traffic / (bpr * rows), 3)}
        for extra in ("tcc", "sq", "lds"):
            p = os.path.join(SRC, f"{name}_{extra}_counters.csv")
            if os.path.exists(p):
                c = counters(p)
                out[extra] = {short(k): {"dispatches": n, **{cn: round(v) for cn, v in vals.items()}}
                              for k, (vals, n) in c.items() if n >= n_main}
        with open(os.path.join(DST, f"{RND}_{name}_pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(name, "traffic/launch %.2f GB = %.1f B/row (%.2fx algorithmic)" % (traffic / 1e9, traffic / rows, traffic / (bpr * rows)))


if __name__ == "__main__":
    main()
