#!/bin/bash
# Refresh of the round-2 profiles after a change to the C5 passes only: the driver's bench command (kernel stats) and
# the C5 / C5-shard counter passes.  Same outputs as profile_r02.sh (gpurun_out/r02/), the other configs' files stay.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
run_stats() {
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_bench.json 2> $O/${name}_bench.err
  find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/prof_$name
}
run_pmc() {
  local name=$1 tag=$2 ctr=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${name}_$tag -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_${tag}.log 2>&1
  find $O/pmc_${name}_$tag -name "*counter_collection.csv" -exec cp {} $O/${name}_${tag}_counters.csv \;
  rm -rf $O/pmc_${name}_$tag
}
run_stats bench_default
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --extra none"
for c in c5 c5s; do
  run_pmc $c fetch "FETCH_SIZE" --config $c $Q
  run_pmc $c write "WRITE_SIZE" --config $c $Q
done
run_pmc c5 sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --config c5 $Q
ls $O | wc -l
tail -1 $O/bench_default_bench.json | cut -c1-600
