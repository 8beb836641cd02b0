#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
timeout 600 bash scripts/show_configs.sh --rows 256000000 --only c5 > $O/c5.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c5 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c5 > $O/c5_prof.log 2>&1
cp $O/prof/*kernel_stats.csv $O/c5_kernel_stats.csv 2>/dev/null || find $O/prof -name "*stats*" -exec cp {} $O/ \;
rm -rf $O/prof
