#!/bin/bash
# Print VGPR / spill / scratch of the scan kernels (cross-compile only; no GPU needed).
cd "$(dirname "$0")/../hdk_amd/csrc" || exit 1
tmp=$(mktemp -d)
hipcc -std=c++17 -O3 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off --cuda-device-only -S scan_agg.hip -o $tmp/k.s || exit 1
python3 - $tmp/k.s "$@" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for b in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    n = g("name")
    if "direct" in n and "--all" not in sys.argv:
        continue
    print(f"{n[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size')}")
PY
rm -rf $tmp
