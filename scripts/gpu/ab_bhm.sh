#!/bin/bash
# A/B of the multi-argument on-chip group-by: blocks per CU (HDK_HIP_BHM_BLOCKS_PER_CU) and variant builds
ROWS=${ROWS:-256000000}
CFG=${CFG:-msbs1,msphs1,phm2}
run() { python3 scripts/bench_configs.py --rows $ROWS --only $CFG 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ',d['config'], d.get('kernel_ms'), round(d.get('frac_of_8TBps',0),3), d.get('error',''))"; }
for b in ${BLOCKS:-1 2 3 4}; do echo "blocks/CU $b"; HDK_HIP_BHM_BLOCKS_PER_CU=$b run; done
for v in ${VARIANTS:-np u4}; do for b in ${VBLOCKS:-2 3}; do echo "variant $v blocks/CU $b"; HDK_HIP_LIB=hdk_amd/libhdk_hip_$v.so HDK_HIP_BHM_BLOCKS_PER_CU=$b run; done; done
