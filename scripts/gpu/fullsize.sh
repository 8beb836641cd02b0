#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fs; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=12 -x ${FS_ARGS:-} > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
