#!/bin/bash
# GPU batch 2: scatter ablations; per-size probe counters; counters of the real C3 kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b2; mkdir -p $O
timeout 300 ./scripts/microbench/partition > $O/partition.txt 2>&1
summ() {
python3 - "$1" "$2" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file under", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:int(sys.argv[2])]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(calls[k])
    print(k, "calls", n, {c: round(v / n) for c, v in acc[k].items()})
PY
}
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_$tag -o g2 --output-format csv -- ./scripts/microbench/gather2 10000000 > $O/pmc_$tag.log 2>&1
  echo "== gather2 nd=10M: $grp" >> $O/gather2_pmc.txt
  summ $O/pmc_$tag 40 >> $O/gather2_pmc.txt 2>&1
  rm -rf $O/pmc_$tag
done
# the real kernel: C3 and C3g through bench_configs at 256 M rows
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_c3_$tag -o c3 --output-format csv -- python3 scripts/bench_configs.py --rows 256000000 --only c3 > $O/pmc_c3_$tag.log 2>&1
  echo "== c3 256M rows: $grp" >> $O/c3_pmc.txt
  summ $O/pmc_c3_$tag 60 | grep -i "scan\|join" >> $O/c3_pmc.txt 2>&1
  rm -rf $O/pmc_c3_$tag
done
