cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04
for b in 2 3 4; do HDK_HIP_PP_BLOCKS_PER_CU=$b timeout 600 python scripts/bench_configs.py --rows 256000000 --only c2m 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print('blocks/CU $b', 'kernel_ms %.3f' % d['kernel_ms'])
"; done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/pp -o pp --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c2m > /dev/null 2>&1; find gpurun_out/r04/pp -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/c2m_kernel_stats.csv \; ; rm -rf gpurun_out/r04/pp; grep -E "k_pp|global|init" gpurun_out/r04/c2m_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
