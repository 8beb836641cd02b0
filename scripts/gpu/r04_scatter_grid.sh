#!/bin/bash
# Round 4: blocks per CU of the two radix scatters (HDK_HIP_SCATTER_BLOCKS_PER_CU) against the microbenchmark's finding
# that two resident blocks per CU beat three; and the microbenchmark itself.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 300 scripts/microbench/scatter_runs 2>&1 | tee gpurun_out/r04/scatter_runs.txt
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --extra none"
for cap in 0 2 1; do
  for c in c3 c5; do
    if [ $cap = 0 ]; then unset HDK_HIP_SCATTER_BLOCKS_PER_CU; else export HDK_HIP_SCATTER_BLOCKS_PER_CU=$cap; fi
    python3 bench.py --config $c $Q 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cap $cap', '$c', '%.3f ms/step' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_kernel_ms'], all(d['checks'].values()))
"
  done
done
