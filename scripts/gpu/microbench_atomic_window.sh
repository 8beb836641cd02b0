#!/bin/bash
mkdir -p gpurun_out
timeout 600 scripts/microbench/atomic_window | tee gpurun_out/atomic_window.txt
