#!/bin/bash
# streaming kernel after a change: the tests that reach it, then the configs that run through it
python -m pytest tests/test_gpu_groupby.py tests/test_gpu_queries.py tests/test_gpu_errors_and_filters.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
python scripts/bench_configs.py --rows 256000000 --only c1,c2,q1,q2 2>/dev/null | cut -c1-250
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-200
