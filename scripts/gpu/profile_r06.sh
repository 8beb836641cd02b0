#!/bin/bash
# Round-6 profiles: kernel-trace stats of the driver's bench command, and PMC passes (one counter per pass, kernel trace
# only) for every bench config plus the 100 M-row-dimension C3.  Outputs under gpurun_out/r06/, summarised by
# scripts/summarise_profiles.py r06.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
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
run_stats bench_default --no-live-traffic
run_stats bench_no_e2e --no-end-to-end --no-live-traffic
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --no-end-to-end --no-live-traffic --extra none"
for c in $([ -n "$SKIP_PMC" ] || echo ${CONFIGS:-c2 c3 c3g c3gm c3m bh1 bh3 bh4 bh5 nga2 msbs1 msphs1 msphs1w msphs1f phm2 c5 c5s q1 q2 q3 q4}); do
  run_pmc $c fetch "FETCH_SIZE" --config $c $Q
  run_pmc $c write "WRITE_SIZE" --config $c $Q
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r06/bench_default_bench.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('C2 %.4g rows/s %.3f ms frac %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic')))
        cb=d['cpu_baseline']; print('cpu_baseline', json.dumps(cb)[:900])
        print(d.get('configs'))
        print(json.dumps(d.get('multi_gpu_emulated'))[:900])
PY
tail -3 $O/bench_default_bench.err | cut -c1-300
ls $O | head -80
