#!/bin/bash
# per-kernel averages of the C5 passes at BASELINE size (1 B rows)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c5q; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof1b -o c5 --output-format csv -- python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --extra none > $O/c5_1b.log 2>&1
find $O/prof1b -name "*kernel_stats.csv" -exec cp {} $O/c5_1b_kernel_stats.csv \;
rm -rf $O/prof1b
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c5q/c5_1b_kernel_stats.csv')):
    n=r['Name']
    if 'part' in n or 'baseline' in n or 'init' in n:
        print(n[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
tail -1 $O/c5_1b.log | cut -c1-200
