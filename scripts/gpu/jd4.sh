#!/bin/bash
python -m pytest tests/test_gpu_cluster.py tests/test_gpu_joins.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_cpp_harness.py -m gpu -x -q 2>&1 | tail -6
python scripts/bench_configs.py --rows 256000000 --only c3,c3g 2>/dev/null | cut -c1-260
python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --extra none 2>/dev/null | cut -c1-330
