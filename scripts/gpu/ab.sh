#!/bin/bash
# A/B of two builds of the library in ONE call (boxes differ by several per cent): ab.sh <variant name> <configs...>
# runs bench.py for each config with the default library and with hdk_amd/libhdk_hip_<variant>.so, alternating, twice.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; shift
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-oracle-sample --no-multi-gpu-emulation --extra none"
for rep in 1 2; do
  for c in "$@"; do
    for lib in default $V; do
      if [ $lib = default ]; then unset HDK_HIP_LIB; else export HDK_HIP_LIB=$GRAFT_REPO_ROOT/hdk_amd/libhdk_hip_$lib.so; fi
      python3 bench.py --config $c $Q 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib', '$c', '%.3f ms/step' % d['ms_per_step'], all(d['checks'].values()))
"
    done
  done
done
