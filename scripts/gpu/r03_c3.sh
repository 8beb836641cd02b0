#!/bin/bash
# C3 after a change to the join kernels: parity tests, then the bench line with per-kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_queries.py tests/test_gpu_joins.py -m gpu -x -q 2>&1 | tail -15
for v in default "$@"; do
  if [ "$v" = nocluster ]; then export HDK_BENCH_FLAGS=512; else unset HDK_BENCH_FLAGS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3_$v -- python3 bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r03/c3_$v.json 2> gpurun_out/r03/c3_$v.err
  f=$(find /tmp/prof_c3_$v -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/r03/c3_${v}_kernel_stats.csv
  python3 - $v <<'PY'
import csv, json, sys
v=sys.argv[1]
for l in open(f'gpurun_out/r03/c3_{v}.json'):
    if l.startswith('{'):
        d=json.loads(l); print(v, 'ms_per_step', d['ms_per_step'], 'value %.3g' % d['value'], d['checks'], d['roofline']['kernel'])
for r in csv.DictReader(open(f'gpurun_out/r03/c3_{v}_kernel_stats.csv')):
    n=r['Name']
    if 'hdk_' in n:
        print('  ', n.replace('void hdk::','')[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
  tail -3 gpurun_out/r03/c3_$v.err | cut -c1-300
done
