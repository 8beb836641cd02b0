#!/bin/bash
# differential soak: many more seeded random plans than the default suite runs (no -x: report every failing seed)
mkdir -p gpurun_out
HDK_FUZZ_ROWS="${2:-60000}" HDK_FUZZ_SEEDS="${1:-100:150}" python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | tail -40 | tee gpurun_out/soak.txt
