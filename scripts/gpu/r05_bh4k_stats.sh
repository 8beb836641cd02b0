mkdir -p gpurun_out/r05; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in bh2k bh4k; do
  rm -rf gpurun_out/r05/bh_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/r05/bh_stats -o s --output-format csv -- python3 scripts/bench_configs.py --only $c --rows 268435456 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r05/bh_stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "bh_" in r["Name"]: print("$c", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1000,1), "us")
PY
done
