# per-kernel times of the sliced-join shapes of round 5 (256 M rows, 10 M-key dimension)
mkdir -p gpurun_out/r05; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${CFGS:-c3g c3f c3x c3x2}; do
  rm -rf gpurun_out/r05/c3_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/r05/c3_stats -o s --output-format csv -- python3 scripts/bench_configs.py --only $c --rows 268435456 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r05/c3_stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "join_" in r["Name"] and float(r["AverageNs"]) > 20000: print("$c", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1000,1), "us")
PY
done
