#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c3q; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c3 > $O/c3_prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/c3_kernel_stats.csv \;
rm -rf $O/prof
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c3q/c3_kernel_stats.csv')):
    n=r['Name']
    if 'hdk' in n:
        print(n[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
grep rows_per_s $O/c3_prof.log | cut -c1-300
