#!/bin/bash
# further soaks with the closing build (results: gpurun_out/r05/soak_more_*.txt)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05; mkdir -p $O
HDK_FUZZ_ROWS=60000 HDK_FUZZ_SEEDS=${S1:-4000:4080} python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 2800 2>&1 | tail -3 > $O/soak_more_fuzz.txt
HDK_HIP_BUILD_PARTITION_MIN_ROWS=1 HDK_HIP_BUILD_TWO_LEVELS=2 HDK_HIP_PERFECT_PARTITIONS_ALWAYS=1 HDK_FUZZ_ROWS=60000 HDK_FUZZ_SEEDS=${S2:-4100:4130} python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 2800 2>&1 | tail -3 > $O/soak_more_fuzz_pb.txt
HDK_FUZZ_SEEDS=${S3:-5000:5600} python -m pytest tests/test_gpu_bh_lds.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -3 > $O/soak_more_bh.txt
HDK_HIP_NO_BH_PLAIN=1 HDK_FUZZ_SEEDS=${S4:-5600:5800} python -m pytest tests/test_gpu_bh_lds.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -3 > $O/soak_more_bh_general.txt
HDK_FUZZ_SEEDS=${S5:-5000:5300} python -m pytest tests/test_gpu_cluster.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -3 > $O/soak_more_sliced.txt
HDK_HIP_S2_NO_PACKED_PAIR=1 HDK_HIP_SLICE_TWO_LEVELS=1 HDK_HIP_SLICE_FINE_KEYS=192 HDK_FUZZ_SEEDS=${S6:-5300:5400} python -m pytest tests/test_gpu_cluster.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -3 > $O/soak_more_sliced_2lvl.txt
tail -2 $O/soak_more_*.txt
