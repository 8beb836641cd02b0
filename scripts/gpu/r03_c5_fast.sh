#!/bin/bash
# after a change to the radix-partitioned kernels: baseline parity tests, then per-kernel times (rocprofv3) of c5 / c5s
timeout 900 python -m pytest tests/test_gpu_baseline.py -m gpu -x -q 2>&1 | tail -5
bash scripts/gpu/r03_c5_prof.sh "$@" 2>&1 | grep -E "^==|ms_per_step"
python3 - "$@" <<'PY'
import csv, sys
for lib in ['default'] + sys.argv[1:]:
  for cfg in ('c5','c5s'):
    print('==',lib,cfg)
    for r in csv.DictReader(open(f'gpurun_out/r03/{cfg}_{lib}_kernel_stats.csv')):
        n=r['Name']
        if 'hdk_part' in n or 'baseline' in n:
            print('  ',n.replace('void hdk::','')[:75], r['Calls'], 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
