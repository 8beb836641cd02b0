#!/bin/bash
# the tests that reach the partitioned group-by (pass 1-4) plus the full-size configurations
python -m pytest tests/test_gpu_baseline.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -k "not multi_gpu" 2>&1 | tail -6
