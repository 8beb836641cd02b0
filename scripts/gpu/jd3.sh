#!/bin/bash
python -m pytest tests/test_gpu_cluster.py -m gpu -x -q 2>&1 | grep -E "Error|error|assert|passed|failed" | head -12
for g in 256 512 1024; do echo "clustered, consumer grid $g"; HDK_HIP_JD_GRID=$g python scripts/bench_configs.py --rows 256000000 --only c3 2>/dev/null | cut -c100-230; done
