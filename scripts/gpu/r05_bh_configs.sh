#!/bin/bash
# profiles/r05_bh_configs.txt: the reference's BaselineHash / PerfectHashSingleCol benchmark shapes and C3gm at 256 M rows, with
# this round's kernels and with each of their layers switched off, and the reference's own scheme (global atomics) at 64 M rows.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05; mkdir -p $O
F='"config": "[a-z0-9]*", "kernel": "[a-z_0-9,]*"\|"kernel_ms": [0-9.]*\|"rows_per_s": [0-9.e+]*'
run() {  # label, rows, extra args; env from the caller
  echo "== $1"
  python3 scripts/bench_configs.py --only ${ONLY:-bh1,bh2,bh3,bh4,bh5,ph1,ph2,ph3,c3gm} --rows $2 $3 2>/dev/null | grep -o "$F" | paste - - - | sed 's/"//g'
}
{
run "default build, 256 M rows" 268435456
HDK_HIP_NO_BH_DENSE=1 HDK_HIP_NO_BH_DENSE_PARTITIONS=1 run "keys by 32-bit tags / hash bins (HDK_HIP_NO_BH_DENSE=1 HDK_HIP_NO_BH_DENSE_PARTITIONS=1): what sparse keys get" 268435456
ONLY=bh1,bh2,bh3,ph1,ph3 HDK_HIP_BH_FOLD_GROUPS=8 run "one-kernel slab fold, 8 groups (HDK_HIP_BH_FOLD_GROUPS=8)" 268435456
ONLY=bh1,bh3 HDK_HIP_BH_DIRECT_FOLD=1 run "every scan block folds into the output table (HDK_HIP_BH_DIRECT_FOLD=1)" 268435456
ONLY=bh1,bh3,bh5,ph1,ph3 HDK_HIP_NO_BH_PACKED=1 run "word-form LDS kernels (HDK_HIP_NO_BH_PACKED=1)" 268435456
ONLY=bh1,bh3,bh5,c3gm run "the reference's scheme: global atomics on the final table (--flags 1), 64 M rows" 67108864 "--flags 1"
} > $O/bh_configs.txt 2>&1
cat $O/bh_configs.txt
