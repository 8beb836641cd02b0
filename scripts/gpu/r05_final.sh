#!/bin/bash
# Round 5 closing run: the whole -m gpu suite, then the soaks with the closing build (results: gpurun_out/r05/final_*.txt)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/final_gpu_suite.txt
HDK_FUZZ_SEEDS=1000:1400 python -m pytest tests/test_gpu_bh_lds.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -4 > $O/final_soak_bh.txt
HDK_FUZZ_SEEDS=1000:1200 python -m pytest tests/test_gpu_cluster.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -4 > $O/final_soak_sliced.txt
HDK_FUZZ_ROWS=60000 HDK_FUZZ_SEEDS=1000:1040 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | tail -4 > $O/final_soak_fuzz.txt
HDK_HIP_NO_BH_DENSE=1 HDK_HIP_NO_BH_DENSE_PARTITIONS=1 HDK_FUZZ_SEEDS=1400:1500 python -m pytest tests/test_gpu_bh_lds.py -q -m gpu -k random_shapes -p no:cacheprovider 2>&1 | tail -4 > $O/final_soak_bh_tags.txt
tail -3 $O/final_*.txt
