#!/bin/bash
python scripts/bench_configs.py --rows 256000000 --only q3,q4,q3v,q3m,q4v 2>/dev/null | tee gpurun_out/keys_bench.jsonl
