#!/bin/bash
# kernel-trace stats of one bench config: scripts/gpu/kstats.sh <config> [env assignments...]; prints the top kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
c=$1; shift
for kv in "$@"; do export "$kv"; done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 bench.py --config $c --steps 10 --warmup 2 --extra none --no-cpu-baseline --no-oracle-sample --no-end-to-end --no-live-traffic --multi-gpu-emulation none > $O/kstats_$c.log 2>&1
find $O/prof_$c -name "*kernel_stats.csv" -exec cp {} $O/${c}_kernel_stats.csv \;
rm -rf $O/prof_$c
python3 - "$O/${c}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r["Name"][:100], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
