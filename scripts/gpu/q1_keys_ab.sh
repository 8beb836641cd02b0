#!/bin/bash
# A/B: taxi Q1 / Q2 through the streaming kernel (default) and through hdk_scan_agg_keys (HDK_HIP_PREFER_KEYS)
python scripts/bench_configs.py --rows 256000000 --only q1,q2 2>/dev/null | cut -c1-260
HDK_HIP_PREFER_KEYS=1 python scripts/bench_configs.py --rows 256000000 --only q1,q2 2>/dev/null | cut -c1-260
