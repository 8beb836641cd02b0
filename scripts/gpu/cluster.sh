#!/bin/bash
python -m pytest tests/test_gpu_cluster.py tests/test_gpu_joins.py -m gpu -x -q 2>&1 | tail -15
python scripts/bench_configs.py --rows 256000000 --only c3,c3g 2>/dev/null | cut -c1-330
