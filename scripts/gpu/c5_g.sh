#!/bin/bash
# per-kernel times of the partitioned group-by at 256 M rows for several flush granules
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c5g; mkdir -p $O; rm -f $O/summary.txt
for g in 0 3; do
  export HDK_HIP_PART_G_LOG2=$g
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$g -o c5 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c5 > $O/c5_prof$g.log 2>&1
  find $O/prof$g -name "*kernel_stats.csv" -exec cp {} $O/c5_kernel_stats$g.csv \;
  rm -rf $O/prof$g
  echo "== g_log2=$g" >> $O/summary.txt
  python3 - $O/c5_kernel_stats$g.csv >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'part' in n or 'baseline' in n:
        print(n[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
  grep -o '"rows_per_s": [0-9.e+]*' $O/c5_prof$g.log >> $O/summary.txt
done
