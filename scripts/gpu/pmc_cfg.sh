#!/bin/bash
# SQ counters of one bench config's kernels: scripts/gpu/pmc_cfg.sh <config> "<counters>" [pass name]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
c=$1; ctr=$2; tag=${3:-sq}
timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${c}_$tag -o $c --output-format csv -- python3 bench.py --config $c --steps 3 --warmup 1 --extra none --no-cpu-baseline --no-oracle-sample --no-end-to-end --no-live-traffic --multi-gpu-emulation none > $O/pmc_${c}_$tag.log 2>&1
find $O/pmc_${c}_$tag -name "*counter_collection.csv" -exec cp {} $O/${c}_${tag}_counters.csv \;
rm -rf $O/pmc_${c}_$tag
python3 - "$O/${c}_${tag}_counters.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "hdk" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: "%.4g" % (v / n[(k, c)]) for c, v in d.items()})
PY
