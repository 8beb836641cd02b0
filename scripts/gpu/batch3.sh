#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b3; mkdir -p $O
timeout 300 ./scripts/microbench/partition > $O/partition.txt 2>&1
timeout 300 ./scripts/microbench/gather2 10000000 uc > $O/gather2_uc.txt 2>&1
timeout 300 ./scripts/microbench/gather2 10000000 fg > $O/gather2_fg.txt 2>&1
timeout 300 ./scripts/microbench/gather2 30000000 uc >> $O/gather2_uc.txt 2>&1
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_$tag -o g2 --output-format csv -- ./scripts/microbench/gather2 10000000 uc > $O/pmc_$tag.log 2>&1
  echo "== gather2 nd=10M uc: $grp" >> $O/gather2_uc_pmc.txt
  python3 - $O/pmc_$tag >> $O/gather2_uc_pmc.txt 2>&1 <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k]); print(k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
  rm -rf $O/pmc_$tag
done
