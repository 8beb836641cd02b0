#!/bin/bash
python -m pytest tests/test_gpu_cluster.py tests/test_gpu_joins.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15
bash scripts/gpu/c3_stats.sh 2>&1 | tail -8 | cut -c1-260
python scripts/bench_configs.py --rows 256000000 --only c3 --flags 512 2>/dev/null | cut -c1-260
