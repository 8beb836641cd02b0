#!/bin/bash
# A/B: SQ counters of the partitioned group-by kernels, round-1 build against the current one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; mkdir -p $O
for which in old new; do
  if [ $which = old ]; then export HDK_HIP_LIB=$GRAFT_REPO_ROOT/scripts/microbench/old/libhdk_hip.so; else unset HDK_HIP_LIB; fi
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $O/p$which$i -o c5 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c5 > $O/p$which$i.log 2>&1
    echo "== $which: $grp" >> $O/summary.txt
    python3 - $O/p$which$i >> $O/summary.txt 2>&1 <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:44]
    if "part" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k]); print(k, {c: round(v / n / 1e6, 1) for c, v in acc[k].items()})
PY
    rm -rf $O/p$which$i
  done
done
