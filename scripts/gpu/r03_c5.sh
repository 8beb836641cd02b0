#!/bin/bash
# C5 after a change to the radix-partitioned kernels: parity tests, the single-GPU bench lines (narrow / wide tuples),
# the 8-rank tuple exchange emulated on one device
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_baseline.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15
for cfg in c5 c5s; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null > gpurun_out/r03/${cfg}_narrow.json; cut -c1-400 gpurun_out/r03/${cfg}_narrow.json
  HDK_HIP_PART_WIDE=1 timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null > gpurun_out/r03/${cfg}_wide.json; cut -c1-400 gpurun_out/r03/${cfg}_wide.json
done
timeout 900 python scripts/multi_gpu_floor.py --only c --steps 5 --out gpurun_out/r03/floor_c.json 2>&1 | tail -25 | cut -c1-500
