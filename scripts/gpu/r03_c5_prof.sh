#!/bin/bash
# per-kernel times of the C5 passes (rocprofv3 kernel trace) for the default build and, if present, A/B variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
for lib in default $@; do
  if [ "$lib" != default ]; then export HDK_HIP_LIB=$R/hdk_amd/libhdk_hip_$lib.so; else unset HDK_HIP_LIB; fi
  for cfg in c5 c5s; do
    rm -rf /tmp/prof_$lib_$cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${lib}_$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --extra none > /tmp/prof_${lib}_$cfg.log 2>&1
    f=$(find /tmp/prof_${lib}_$cfg -name '*kernel_stats.csv' | head -1)
    echo "== $lib $cfg"; grep -E "hdk_part|baseline_direct" $f | cut -d, -f1-5 | cut -c1-160
    cp $f $R/gpurun_out/r03/${cfg}_${lib}_kernel_stats.csv
    python3 -c "
import json,sys
for l in open('/tmp/prof_${lib}_$cfg.log'):
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['checks'])
"
  done
done
