#!/bin/bash
# wide-generator aggregate fuzz at a scaled-up fact table, given seeds (comma list); assertion lines only
mkdir -p gpurun_out
for s in ${1//,/ }; do
  HDK_FUZZ_ROWS="${2:-2000000}" HDK_FUZZ_SEEDS="$s:$((s+1))" python -m pytest "tests/test_gpu_fuzz.py::test_random_wide_aggregate_plans[$s]" -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | grep -E "^E  |passed|failed" | cut -c1-1800
done | tee gpurun_out/soak_wide_big.txt
