#!/bin/bash
python -m pytest tests/test_gpu_groupby.py tests/test_gpu_fullsize.py -m gpu -x -q -k "keys or taxi or q3 or q4 or multi_key" 2>&1 | tail -8
python scripts/bench_configs.py --rows 256000000 --only q3,q4,q3v,q3m,q4v 2>/dev/null | tee gpurun_out/keys_bench3.jsonl | cut -c1-250
