#!/bin/bash
# the round's standing check: the whole -m gpu suite, the driver's bench line, C2's PMC passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r03/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('C2 %.4g rows/s %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
        cb=d['cpu_baseline']; print('cpu_baseline', cb['value'], cb['cores'], cb.get('best_variant')); 
        for k,v in cb['variants'].items(): print('   ', k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
        for c in d.get('configs',[]): print(' ', c['metric'], '%.3g'%c['value'], '%.3f ms'%c['ms_per_step'], all(v for v in c['checks'].values()))
        print(json.dumps(d.get('multi_gpu_emulated'))[:900])
PY
