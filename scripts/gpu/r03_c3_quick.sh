#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_cluster.py -m gpu -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r03/c3_q.json 2> gpurun_out/r03/c3_q.err
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, json, sys
for l in open('gpurun_out/r03/c3_q.json'):
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'value %.3g' % d['value'], d['checks'])
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'hdk_' in n:
        print('  ', n.replace('void hdk::','')[:70], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
if [ "$1" = pmc ]; then bash scripts/gpu/pmc_quick.sh c3 x 2>&1 | grep -E "==|join_" | cut -c1-420; fi
