#!/usr/bin/env python3
"""How does tier 1 behave on clustered (realistic) embeddings?  Prints tier statistics."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.index import FlatIndex

def main():
    dev = torch.device("cuda:0")
    n, d, nq = 500_000, 1024, 1024
    for ncls, spread in ((3, 0.5), (3, 1.0), (100, 0.5), (1000, 0.3)):
        g = torch.Generator(device=dev).manual_seed(7)
        centers = torch.nn.functional.normalize(torch.randn(ncls, d, generator=g, device=dev), dim=1)
        lab = torch.randint(0, ncls, (n,), generator=g, device=dev)
        x = torch.nn.functional.normalize(centers[lab] + spread / (d ** 0.5) * torch.randn(n, d, generator=g, device=dev), dim=1)
        ix = FlatIndex(d, "COSINE", 0)
        ix.add(x)
        q = x[torch.randint(0, n, (nq,), generator=g, device=dev)] + 0.01 * torch.randn(nq, d, generator=g, device=dev)
        q = torch.nn.functional.normalize(q, dim=1)
        ix.search(q, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, i = ix.search(q, 10)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = ix.last_stats()
        print(f"classes={ncls} spread={spread}: top1 cos={float(s[:,0].mean()):.3f} 10th={float(s[:,9].mean()):.3f} "
              f"{dt*1e3:.1f} ms  stats={st}", flush=True)
        del ix

if __name__ == "__main__":
    main()
