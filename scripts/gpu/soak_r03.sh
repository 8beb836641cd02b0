#!/bin/bash
# round-3 soak: random shapes through the radix-partitioned group-by (every tuple / pass-3 form, and the same queries as
# an emulated multi-GPU tuple exchange) and through the sliced join.  bash scripts/gpu/soak_r03.sh [first:last seeds]
mkdir -p gpurun_out/r03
rm -f gpurun_out/r03/soak_cases.txt
export HDK_SOAK_LOG=gpurun_out/r03/soak_cases.txt HDK_FUZZ_SEEDS="${1:-10:40}"
python -m pytest tests/test_gpu_baseline.py tests/test_gpu_cluster.py -m gpu -q -p no:cacheprovider --timeout 2400 -k random_shapes 2>&1 | tail -30 | cut -c1-1800 | tee gpurun_out/r03/soak.txt
wc -l gpurun_out/r03/soak_cases.txt
