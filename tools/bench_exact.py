#!/usr/bin/env python3
"""Exact-tier micro benchmark (development tool): fp64 scan q/s."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.index import FlatIndex
from mirx import _lib

def main():
    dev = torch.device("cuda:0")
    for n, d, nq in ((500_000, 1024, 512), (500_000, 256, 512), (30_000, 1024, 2048)):
        g = torch.Generator(device=dev).manual_seed(1)
        ix = FlatIndex(d, "COSINE", 0)
        ix.add(torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=dev), dim=1))
        q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=dev), dim=1)
        ix.set_option(_lib.OPT_TIERS, _lib.TIER_EXACT_ONLY)
        ix.set_option(_lib.OPT_PROFILE, 1)
        ix.search(q, 10); torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.search(q, 10); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"exact N={n} D={d} Q={nq}: {dt*1e3:.1f} ms, {nq/dt:.0f} q/s, {nq*n/dt/1e9:.2f} G pairs/s  {ix.last_timings()}", flush=True)
        del ix

if __name__ == "__main__":
    main()
