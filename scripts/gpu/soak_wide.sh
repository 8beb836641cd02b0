#!/bin/bash
# re-run given wide-generator fuzz seeds (comma list) and keep only the assertion lines
mkdir -p gpurun_out
for s in ${1//,/ }; do
  HDK_FUZZ_SEEDS="$s:$((s+1))" python -m pytest "tests/test_gpu_fuzz.py::test_random_wide_aggregate_plans[$s]" "tests/test_gpu_fuzz.py::test_random_projection_plans[-$s]" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed" | cut -c1-1800
done | tee gpurun_out/soak_wide.txt
