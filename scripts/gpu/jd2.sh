#!/bin/bash
python -m pytest tests/test_gpu_cluster.py -m gpu -x -q 2>&1 | tail -5
echo "small dim (L2-resident), row order"; python scripts/bench_configs.py --rows 256000000 --only c3 --dim-rows 100000 --flags 512 2>/dev/null | cut -c1-230
echo "small dim, clustered"; python scripts/bench_configs.py --rows 256000000 --only c3 --dim-rows 100000 --flags 256 2>/dev/null | cut -c1-230
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/jd; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $O/p -o c3 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c3 > $O/log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/jd/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:50]
    if "hdk" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k]); print(k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
