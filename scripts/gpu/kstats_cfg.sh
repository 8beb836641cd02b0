#!/bin/bash
# kernel-trace stats of scripts/bench_configs.py configs: scripts/gpu/kstats_cfg.sh <only> <rows> [extra args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
only=$1; rows=$2; shift 2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_cfg -o cfg --output-format csv -- python3 scripts/bench_configs.py --rows $rows --only $only "$@" > $O/kstats_cfg.log 2>&1
find $O/prof_cfg -name "*kernel_stats.csv" -exec cp {} $O/cfg_kernel_stats.csv \;
rm -rf $O/prof_cfg
python3 - "$O/cfg_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if "hdk" in r["Name"]:
        print(r["Name"][:110], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
