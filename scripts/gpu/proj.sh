#!/bin/bash
python -m pytest tests/test_gpu_projection.py tests/test_gpu_errors_and_filters.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/proj; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only p1,p50,c2f > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/proj/kernel_stats.csv')):
    n=r['Name']
    if 'hdk' in n:
        print(n[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
grep rows_per_s $O/prof.log | cut -c1-330
