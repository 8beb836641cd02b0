#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a translation unit (cross-compiles to ISA; no GPU needed).

    python scripts/kernel_resources.py scan_baseline.hip [name-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hdk_amd", "csrc")


def main():
    src = sys.argv[1]
    want = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                               "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out], cwd=CSRC,
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    if len(sys.argv) > 2 and sys.argv[-1].endswith(".s"):
        open(sys.argv[-1], "w").write(text)
        want = want[:-1]
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        def g(k):
            mm = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
            return mm.group(1) if mm else "?"
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{dem[:110]:110s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} "
              f"scratch {g('private_segment_fixed_size'):>5s}")


if __name__ == "__main__":
    main()
