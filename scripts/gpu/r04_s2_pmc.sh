#!/bin/bash
# SQ / LDS counters of pass 2 of the sliced join: the general kernel (c3g, c3m) beside the compile-time form (c3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --extra none"
run_pmc() {  # name, tag, counters, bench args...
  local name=$1 tag=$2 ctr=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${name}_$tag -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_${tag}.log 2>&1
  find $O/pmc_${name}_$tag -name "*counter_collection.csv" -exec cp {} $O/${name}_${tag}_counters.csv \;
  rm -rf $O/pmc_${name}_$tag
}
for c in c3 c3g c3m; do
  run_pmc $c sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --config $c $Q
  run_pmc $c lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --config $c $Q
done
python3 - <<'PY'
import csv, collections
for c in ("c3", "c3g", "c3m"):
    for tag in ("sq", "lds"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f"gpurun_out/r04/{c}_{tag}_counters.csv")):
            k = r["Kernel_Name"]
            if "join_agg_sliced" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k in acc:
            n = len(calls[k])
            print(c, tag, k[:48], {cn: round(v / n / 1e6, 1) for cn, v in acc[k].items()}, "(millions per dispatch)")
PY
