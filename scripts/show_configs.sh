#!/bin/bash
# usage: scripts/show_configs.sh [bench_configs.py args]  -- pretty-prints scripts/bench_configs.py
python scripts/bench_configs.py "$@" 2>&1 | grep -v "^#" | python -c '
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        print(l[:200]); continue
    d = json.loads(l)
    if "error" in d:
        print(d); continue
    k = ("%.3f" % d["kernel_ms"]) if d["kernel_ms"] else "-"
    print("%-4s %-22s rows/s %.3e  kern %s ms step %.3f ms  alg %.0f GB/s (%.1f%% of 8TB/s) groups %d entries %d" % (
        d["config"], d["kernel"].split(",")[0], d["rows_per_s"], k, d["step_ms"], d["alg_GBps"],
        100 * d["frac_of_8TBps"], d["groups"], d["entries"]))
'
