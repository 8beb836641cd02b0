#!/bin/bash
mkdir -p gpurun_out
timeout 300 scripts/microbench/stream2 | tee gpurun_out/stream2.txt
