#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_baseline.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not multi_gpu" 2>&1 | tail -6
python scripts/bench_configs.py --rows 256000000 --only c5 2>/dev/null | tee gpurun_out/c5b.jsonl | cut -c1-420
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/c5b_bench.json | cut -c1-900
python bench.py --config c5s --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/c5sb_bench.json | cut -c1-900
