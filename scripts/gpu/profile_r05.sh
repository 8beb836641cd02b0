#!/bin/bash
# Round-5 profiles: kernel-trace stats of the driver's bench command, and PMC passes (one counter per pass, kernel trace
# only) for every bench config plus the 100 M-row-dimension C3.  Outputs under gpurun_out/r05/, summarised by
# scripts/summarise_profiles.py r05.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
run_stats() {  # name, bench args...
  local name=$1; shift
  timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_bench.json 2> $O/${name}_bench.err
  find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/prof_$name
}
run_pmc() {  # name, tag, counters, bench args...
  local name=$1 tag=$2 ctr=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${name}_$tag -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_${tag}.log 2>&1
  find $O/pmc_${name}_$tag -name "*counter_collection.csv" -exec cp {} $O/${name}_${tag}_counters.csv \;
  rm -rf $O/pmc_${name}_$tag
}
run_stats bench_default
run_stats bench_no_e2e --no-end-to-end
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --no-end-to-end --extra none"
for c in $([ -n "$SKIP_PMC" ] || echo ${CONFIGS:-c2 c3 c3g c3gm c3m bh1 bh3 bh5 c5 c5s q1 q2 q3 q4}); do
  run_pmc $c fetch "FETCH_SIZE" --config $c $Q
  run_pmc $c write "WRITE_SIZE" --config $c $Q
done
# the Infinity-Cache experiment (scripts/microbench/mall_pingpong.hip): per-kernel times of one pass over everything against
# 512 MiB chunks, streamed and scattered tuples
for cfg in ${MALL_CFGS:-"0 0 0" "0 0 512" "1 0 0" "1 0 512"}; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_mall -o mall --output-format csv -- scripts/microbench/mall_pingpong 536870912 $1 $2 $3 > $O/mall_$1_$3.txt 2>&1
  find $O/prof_mall -name "*kernel_stats.csv" -exec cp {} $O/mall_$1_$3_kernel_stats.csv \;
  rm -rf $O/prof_mall
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r05/bench_default_bench.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('C2 %.4g rows/s %.3f ms frac %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic')))
        cb=d['cpu_baseline']; print('cpu_baseline', json.dumps(cb)[:900])
        for c in d.get('configs',[]): print(' ', c['metric'], '%.3g'%c['value'], '%.3f ms'%c['ms_per_step'], all(v for v in c['checks'].values()), c['roofline']['kernel'][:60])
        print(json.dumps(d.get('multi_gpu_emulated'))[:900])
PY
tail -3 $O/bench_default_bench.err | cut -c1-300
ls $O | head -80
