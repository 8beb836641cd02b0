#!/bin/bash
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/suite.log 2>&1
tail -16 gpurun_out/suite.log
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/bench_default.err
cut -c1-1500 gpurun_out/bench_default.json
