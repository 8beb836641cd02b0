#!/bin/bash
# C3 (join probe) sensitivity to the dimension-table size: separates cache/TLB capacity effects from per-row cost.
mkdir -p gpurun_out
for nd in 100000 1000000 10000000; do
  for f in "" "--no-fuse"; do
    echo "## dim-rows=$nd $f"
    python scripts/bench_configs.py --rows 128000000 --only c3,c3g --dim-rows $nd $f 2>/dev/null | cut -c1-330
  done
done | tee gpurun_out/c3_sweep.txt
