#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_c5b; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o c5 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c5 > $O/p$i.log 2>&1
  echo "== $grp" >> $O/summary.txt
  python3 - $O/p$i >> $O/summary.txt 2>&1 <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    if "part" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k]); print(k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
