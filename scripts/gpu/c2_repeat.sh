#!/bin/bash
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extra none 2>/dev/null | python -c "import sys,json; x=json.loads(sys.stdin.readline()); print(x['value'], x['ms_per_step'], x['roofline']['avg_kernel_ms'], x['roofline']['peak_measured']['read_GBps'])"; done
