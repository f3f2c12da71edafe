#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel in one HIP source (development tool):
tools/resusage.py k_gemm.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-retrieval---thesis-2026_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wall",
       "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-c", sys.argv[1], "-o", "/dev/null"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur.replace("mirx::(anonymous namespace)::", "").replace("void ", ""))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|TotalSGPRs|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    elif "error" in line or "warning" in line:
        print(line)
for k, v in rows.items():
    print(f"{k:60s} vgpr {v.get('VGPRs', -1):3d} agpr {v.get('AGPRs', 0):3d} sgpr {v.get('TotalSGPRs', -1):3d} scratch {v.get('ScratchSize', 0):4d} "
          f"spill {v.get('VGPRs Spill', 0):3d} occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0)}")
