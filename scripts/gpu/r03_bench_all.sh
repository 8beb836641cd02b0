#!/bin/bash
# the driver's bench line at N = 1, the N = 2 / 4 control flow host-staged over gloo on one GPU, the multi-GPU floors
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/bench; mkdir -p $O
( time timeout 1200 python bench.py --steps 10 --warmup 3 ) > $O/n1.json 2> $O/n1.err
echo "rc=$?" >> $O/n1.err; tail -4 $O/n1.err
export HDK_BENCH_BACKEND=gloo
run() { # name, nproc, port, args...
  local name=$1 np=$2 port=$3; shift 3
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port $port bench.py --gpus $np "$@" > $O/$name.json 2> $O/$name.err
  echo "$name rc=$?"; grep -E "^\{" $O/$name.json | cut -c1-300; grep -E "Error|error|Traceback" $O/$name.err | head -5
}
run n2_c2 2 29511 --steps 3 --warmup 1 --rows 256000000
run n2_c5 2 29512 --steps 3 --warmup 1 --rows 128000000 --config c5 --no-cpu-baseline
run n4_c5 4 29514 --steps 3 --warmup 1 --rows 256000000 --config c5 --no-cpu-baseline
run n2_q3 2 29513 --steps 3 --warmup 1 --rows 128000000 --config q3 --scaling weak
unset HDK_BENCH_BACKEND
timeout 900 python scripts/multi_gpu_floor.py --only a,b --out gpurun_out/r03/floor_after.json 2>&1 | tail -3 | cut -c1-600
python3 - <<'PY'
import json
for l in open('gpurun_out/r03/bench/n1.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'] and d['cpu_baseline']['value'])
        for c in d.get('configs',[]): print(' ', c['metric'], '%.3g'%c['value'], '%.3f ms'%c['ms_per_step'], c['checks'])
        print(json.dumps(d.get('multi_gpu_emulated'))[:1500])
PY
