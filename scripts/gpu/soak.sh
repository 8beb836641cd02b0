#!/bin/bash
# differential soak: many more seeded random plans than the default suite runs (no -x: report every failing seed).
# $3 = "pb": every one-to-one join table of the run is built by slot-range partitions (join_build_part.h), two levels forced.
mkdir -p gpurun_out
if [ "$3" = "pb" ]; then export HDK_HIP_BUILD_PARTITION_MIN_ROWS=1 HDK_HIP_BUILD_TWO_LEVELS=2; fi
HDK_FUZZ_ROWS="${2:-60000}" HDK_FUZZ_SEEDS="${1:-100:150}" python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | tail -40 | tee gpurun_out/soak.txt
