#!/bin/bash
echo "row order"; python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-140
echo "clustered"; HDK_BENCH_FLAGS=256 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['checks'])"
