#!/bin/bash
# the round's closing run: the whole -m gpu suite, the driver's bench command under rocprofv3 --kernel-trace --stats
# (-> profiles/r03_bench_default.* through scripts/summarise_profiles.py r03), the filter/project configs with their
# per-kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench_default -o bench_default --output-format csv -- python3 bench.py > $O/bench_default_bench.json 2> $O/bench_default_bench.err
find $O/prof_bench_default -name "*kernel_stats.csv" -exec cp {} $O/bench_default_kernel_stats.csv \;
rm -rf $O/prof_bench_default
tail -3 $O/bench_default_bench.err | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_proj -o proj --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only p1,p50 > $O/projection_configs.json 2> $O/projection_configs.err
find $O/prof_proj -name "*kernel_stats.csv" -exec cp {} $O/projection_kernel_stats.csv \;
rm -rf $O/prof_proj
python3 - <<'PY'
import json
for l in open('gpurun_out/r03/bench_default_bench.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('C2 %.4g rows/s %.3f ms frac %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic')))
        cb=d['cpu_baseline']; print('cpu_baseline', cb['value'], cb['cores'], cb.get('best_variant'))
        for c in d.get('configs',[]): print(' ', c['metric'], '%.3g'%c['value'], '%.3f ms'%c['ms_per_step'], all(v for v in c['checks'].values()))
        print(json.dumps(d.get('multi_gpu_emulated'))[:700])
for l in open('gpurun_out/r03/projection_configs.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config'], d['kernel'].split(',')[0], 'kernel_ms %.3f' % d['kernel_ms'], 'rows/s %.3g' % d['rows_per_s'])
PY
grep -E "project" $O/projection_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
