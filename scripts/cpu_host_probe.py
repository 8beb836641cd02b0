#!/usr/bin/env python3
"""What the GPU box's HOST gives the CPU baseline (bench.py: cpu_baseline): sockets, NUMA nodes, CPUs allowed, and the
streaming-read rate of the oracle's pinned kernel-per-thread loop for several thread counts.  CPU only; run through gpurun to
see the box the driver uses."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

L = O.lib()
print("cpu_count", os.cpu_count(), "allowed", L.orc_allowed_cpu_count(), "omp_max_threads", L.orc_max_threads())
for cmd in (["lscpu"], ["numactl", "-H"]):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout
        print("$", " ".join(cmd))
        print("\n".join(ln for ln in out.splitlines() if any(k in ln for k in ("Model name", "Socket", "Core(s)", "Thread(s)", "NUMA", "node", "MHz", "L3"))))
    except (OSError, subprocess.TimeoutExpired) as e:
        print(cmd, "unavailable:", e)
for t in (16, 32, 64, 128, 192, 256):
    if t > L.orc_allowed_cpu_count():
        break
    g = L.orc_host_stream_read_gbps(t, 256 << 20, 3)
    print(f"stream read, {t:3d} threads x 2 x 256 MiB: {g:7.1f} GB/s")
