#!/usr/bin/env python3
"""Tier-1 statistics of a random N x 1024 gallery for one gallery size (development tool):
python tools/probe_scale.py 24000000"""
import sys, torch
sys.path.insert(0, '.')
from mirx.index import FlatIndex
dev = torch.device("cuda:0")
n, d, nq = int(sys.argv[1]), 1024, 256
g = torch.Generator(device=dev).manual_seed(99)
ix = FlatIndex(d, "COSINE", 0); ix.reserve(n)
chunk = 1 << 19
for s in range(0, n, chunk):
    m = min(chunk, n - s)
    ix.add(torch.nn.functional.normalize(torch.randn(m, d, generator=g, device=dev), dim=1))
q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=dev), dim=1)
sc, ids = ix.search(q, 10, return_f64=True)
print(n, ix.last_stats())
