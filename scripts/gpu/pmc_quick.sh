#!/bin/bash
# quick PMC passes for one bench config: bash scripts/gpu/pmc_quick.sh <config> <tag> [HDK_HIP_LIB variant]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=${1:-c5}; tag=${2:-q}; O=gpurun_out/r03/pmc_${cfg}_$tag; mkdir -p $O
Q="--config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-oracle-sample --extra none"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o x --output-format csv -- python3 bench.py $Q > $O/p$i.log 2>&1
  python3 - $O/p$i "$grp" >> $O/summary.txt 2>&1 <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
print("==", sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void hdk::", "")[:60]
    if "hdk" not in r["Kernel_Name"]: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k]); print(" ", k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
