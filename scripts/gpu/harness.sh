#!/bin/bash
mkdir -p gpurun_out
tests/cpp/_build/harness > gpurun_out/harness.out 2> gpurun_out/harness.err; echo "rc=$?" >> gpurun_out/harness.err
python -m pytest tests/test_gpu_cpp_harness.py tests/test_gpu_errors_and_filters.py -m gpu -x -q 2>&1 | tail -5
