#!/bin/bash
# soak of the radix-partitioned path: random shapes (tests/test_gpu_baseline.py::test_radix_partitioned_random_shapes)
mkdir -p gpurun_out
rm -f gpurun_out/soak_part_cases.txt; HDK_SOAK_LOG=gpurun_out/soak_part_cases.txt HDK_FUZZ_SEEDS="${1:-10:40}" python -m pytest tests/test_gpu_baseline.py -m gpu -q -p no:cacheprovider --timeout 1500 -k random_shapes 2>&1 | tail -30 | cut -c1-1800 | tee gpurun_out/soak_part.txt
