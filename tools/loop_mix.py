#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel (development tool):
tools/loop_mix.py k_attention_h2.hip 14k_attention_h2E [extra hipcc flags]  -- the second argument is a substring of the mangled name."""
import collections
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-retrieval---thesis-2026_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                "-Wno-unused-function", "--cuda-device-only", "-S", *sys.argv[3:], sys.argv[1], "-o", "/tmp/loop_mix.s"],
               cwd=csrc, check=True, capture_output=True)
src = open("/tmp/loop_mix.s").read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):.*?s_endpgm", src, re.S | re.M)
lines = m.group(0).split("\n")
labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
best = None
for i, l in enumerate(lines):
    mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        span = (labels[mm.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
c = collections.Counter()
for l in lines[best[0]:best[1] + 1]:
    mm = re.match(r"^\s+([a-z_0-9]+)", l)
    if mm:
        c[mm.group(1)] += 1
grp = lambda f: sum(v for k, v in c.items() if f(k))  # noqa: E731
print(f"{m.group(1)[:70]}: largest loop = {sum(c.values())} instructions: VALU {grp(lambda k: k.startswith('v_') and 'mfma' not in k)}, "
      f"MFMA {grp(lambda k: 'mfma' in k)}, SALU {grp(lambda k: k.startswith('s_'))}, LDS {grp(lambda k: k.startswith('ds_'))}, "
      f"VMEM {grp(lambda k: k.startswith(('global_', 'buffer_')))}")
for k, v in c.most_common(int(os.environ.get("TOP", "18"))):
    print(f"  {k:28s}{v}")
