#!/bin/bash
# PMC counters for the secondary configs (kernel trace only; one counter group per pass).
# usage: scripts/pmc_configs.sh "<configs>" "<counters...>" <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/pmc_$3 -o pmc --output-format csv -- python3 scripts/bench_configs.py --rows ${ROWS:-128000000} --only $1 > gpurun_out/pmc_$3.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$3/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file", glob.glob("gpurun_out/pmc_$3/**", recursive=True)[:10])
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); calls[k] += 1
    for k in acc:
        if any(w in k for w in "${KFILTER:-scan}".split(",")):
            print(k, "calls", calls[k], {c: round(v / calls[k]) for c, v in acc[k].items()})
PY
