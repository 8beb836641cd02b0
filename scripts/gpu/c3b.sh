#!/bin/bash
echo "clustered (default)"; python scripts/bench_configs.py --rows 256000000 --only c3 2>/dev/null | cut -c1-260
echo "not clustered"; python scripts/bench_configs.py --rows 256000000 --only c3 --flags 512 2>/dev/null | cut -c1-260
echo "small dim (100 K rows: L2-resident), not clustered"; python scripts/bench_configs.py --rows 256000000 --only c3 --dim-rows 100000 --flags 512 2>/dev/null | cut -c1-260
echo "small dim, clustered"; python scripts/bench_configs.py --rows 256000000 --only c3 --dim-rows 100000 --flags 256 2>/dev/null | cut -c1-260
