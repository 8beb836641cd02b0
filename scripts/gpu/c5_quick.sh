#!/bin/bash
bash scripts/gpu/c5_stats.sh 2>&1 | tail -7 | cut -c1-300
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-260
python bench.py --config c5s --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-260
python -m pytest tests/test_gpu_baseline.py -m gpu -x -q -k "not multi_gpu" 2>&1 | tail -2
