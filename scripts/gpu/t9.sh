#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/t9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_errors_and_filters.py tests/test_gpu_fuzz.py tests/test_gpu_groupby.py tests/test_gpu_queries.py -m gpu -x -q --durations=5 > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --extra "" --no-cpu-baseline > $O/c2.json 2> $O/c2.err
