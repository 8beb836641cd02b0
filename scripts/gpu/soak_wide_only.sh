#!/bin/bash
# soak of the wider generator only (aggregate plans and projections), seeds "a:b"
mkdir -p gpurun_out
HDK_FUZZ_ROWS="${2:-60000}" HDK_FUZZ_SEEDS="${1:-100:150}" python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 1500 -k "wide or projection_plans[-" 2>&1 | tail -30 | cut -c1-1500 | tee gpurun_out/soak_wide_only.txt
