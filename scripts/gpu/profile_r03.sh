#!/bin/bash
# Round-3 profiles: kernel-trace stats of the driver's bench command, and PMC passes (one counter group per pass,
# kernel trace only) for the bench configs.  Outputs under gpurun_out/r03/, summarised by scripts/summarise_r03.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
run_stats() {  # name, bench args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_bench.json 2> $O/${name}_bench.err
  find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/prof_$name
}
run_pmc() {  # name, tag, counters, bench args...
  local name=$1 tag=$2 ctr=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${name}_$tag -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}_${tag}.log 2>&1
  find $O/pmc_${name}_$tag -name "*counter_collection.csv" -exec cp {} $O/${name}_${tag}_counters.csv \;
  rm -rf $O/pmc_${name}_$tag
}
run_stats bench_default
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --extra none"
for c in c2 c3 c5 c5s q1 q2 q3 q4; do
  run_pmc $c fetch "FETCH_SIZE" --config $c $Q
  run_pmc $c write "WRITE_SIZE" --config $c $Q
done
run_pmc c3 tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" --config c3 $Q
run_pmc c3 sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --config c3 $Q
run_pmc q3 sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --config q3 $Q
run_pmc c5 sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --config c5 $Q
run_pmc c5 lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --config c5 $Q
ls -la $O | head -60
