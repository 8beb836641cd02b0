#!/bin/bash
# GPU batch 1 (round 2): microbenchmarks that price the join probe (C3) and the radix scatter (C5).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b1; mkdir -p $O
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -o "TCC_[A-Z0-9_]*" $O/counters_all.txt | sort -u > $O/counters_tcc.txt
timeout 300 ./scripts/microbench/gather2 > $O/gather2.txt 2>&1
timeout 300 ./scripts/microbench/partition > $O/partition.txt 2>&1
summ() {  # $1 = dir
python3 - "$1" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file under", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k])
    print(k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
}
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_$tag -o g2 --output-format csv -- ./scripts/microbench/gather2 > $O/pmc_$tag.log 2>&1
  echo "== $grp" >> $O/gather2_pmc.txt
  summ $O/pmc_$tag >> $O/gather2_pmc.txt 2>&1
  rm -rf $O/pmc_$tag
done
