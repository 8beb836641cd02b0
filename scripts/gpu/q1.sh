#!/bin/bash
python -m pytest tests/test_gpu_groupby.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python scripts/bench_configs.py --rows 256000000 --only q1,q2,c1 2>/dev/null | cut -c1-300
python bench.py --config q1 --steps 10 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-330
