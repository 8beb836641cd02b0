#!/bin/bash
python -m pytest tests/test_gpu_groupby.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "prefetch $v"
  HDK_HIP_DIRECT_PF=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extra none 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'], d['checks'])"
done
python scripts/bench_configs.py --rows 256000000 --only c1,c2,c2n,c2f,q1,q2 2>/dev/null | cut -c1-330
