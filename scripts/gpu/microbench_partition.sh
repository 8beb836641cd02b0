#!/bin/bash
mkdir -p gpurun_out
timeout 600 scripts/microbench/partition | tee gpurun_out/partition.txt | tail -12
