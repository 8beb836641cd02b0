mkdir -p gpurun_out/r05
S=$(date +%s); python bench.py > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err; E=$(date +%s); echo "bench.py wall $((E-S)) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r05/bench_final.json").read().strip().split(chr(10))[-1])
print(d["metric"], d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["n_gpus"], d["ranks_seen_by_collective"])
print("checks", d["checks"])
print("e2e", d["end_to_end"]["h2d_GBps"], d["end_to_end"]["same_result_as_resident_run"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
for c in d["configs"]: print(" ", c["config"]["workload"][:40], c["ms_per_step"], round(c["roofline"]["frac"],3), all(c["checks"].values()))
PY
