#!/bin/bash
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/suite.log 2>&1
tail -12 gpurun_out/suite.log
python scripts/bench_configs.py --rows 256000000 --only p1,p50 2>/dev/null | cut -c1-300
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/bench_default.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print("C2", "%.4e"%d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("peak_measured"), d["roofline"].get("traffic"), d["checks"])
print(d["cpu_baseline"]["value"])
for c in d["configs"]:
    print(c["metric"], "%.3e"%c["value"], "%.3f ms"%c["ms_per_step"], "%.0f GB/s %.3f"%(c["roofline"]["achieved"], c["roofline"]["frac"]), c["roofline"].get("traffic"), all(v for v in c["checks"].values()))
PY
