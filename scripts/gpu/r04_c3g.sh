#!/bin/bash
# round 4: the sliced2 path -- parity tests, then C3g / C3m / C3 at 1 B rows
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_cluster.py -x -q -m gpu -k "sliced2 or small_caller" > gpurun_out/r04/sliced2_tests.log 2>&1; tail -5 gpurun_out/r04/sliced2_tests.log
for c in c3g c3m c3; do
  python bench.py --config $c --steps 5 --warmup 2 --extra none --no-cpu-baseline --no-multi-gpu-emulation > gpurun_out/r04/bench_$c.json 2> gpurun_out/r04/bench_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_$c.json"))
print("$c", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["checks"])
PY
done
