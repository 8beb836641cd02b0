#!/bin/bash
# round 4: per-kernel times of the sliced2 path (rocprofv3 kernel trace) at 1 B rows; CONFIGS="c3g c3m" by default
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for c in ${CONFIGS:-c3g c3m}; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 bench.py --config $c --steps 5 --warmup 2 --extra none --no-cpu-baseline --no-multi-gpu-emulation --no-oracle-sample > $O/prof_$c.json 2> $O/prof_$c.err
  find $O/prof_$c -name "*kernel_stats.csv" -exec cp {} $O/${c}_kernel_stats.csv \;
  rm -rf $O/prof_$c
  head -8 $O/${c}_kernel_stats.csv | cut -c1-150
done
