#!/bin/bash
for g in 256 512 768 1024 1536 2048; do
  echo "C2 grid $g"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-oracle-sample --extra none --grid $g 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"
done
for g in 0 512 768 1024 1536 2048; do
  echo "grid $g"; python scripts/bench_configs.py --rows 256000000 --only q1,q2,q3,q4,c1 --grid $g 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['config'], round(d['kernel_ms'],3), round(d['alg_GBps']))"
done
