for g in 512 768 1024 1280 1536 2048 3072; do
  timeout 300 python bench.py --no-cpu-baseline --grid $g --steps 10 2>/dev/null | tail -1 > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print("grid", $g, "%.3e rows/s" % d["value"], "%.0f GB/s" % d["roofline"]["achieved"], "kernel %.3f ms" % d["roofline"]["avg_kernel_ms"], "step %.3f ms" % d["ms_per_step"])
PY
done
