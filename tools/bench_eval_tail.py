#!/usr/bin/env python3
"""Metric tail of evaluate() (test.py:1080-1110) at growing N: device full ranking + device AP
(mirx_rank_metrics) vs the host numpy tail fed with the same ranking (development tool)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import metrics as mm  # noqa: E402
from mirx.evaluate import rank_self  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="3000,10000,30000")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--host-max", type=int, default=10000, help="largest N the numpy tail is timed at")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for n in [int(v) for v in a.sizes.split(",")]:
        e = torch.nn.functional.normalize(torch.randn(n, a.dim, generator=torch.Generator().manual_seed(n)), dim=1).to(dev)
        labels = np.arange(n) % 3
        rank_self(e[:256], "cdist")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ranks, _ = rank_self(e, "cdist")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        got = mm.compute_map(ranks.t(), labels, [1, 5, 10])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        line = f"N={n}: full ranking {1e3*(t1-t0):.1f} ms, device AP/precision {1e3*(t2-t1):.1f} ms (mAP {got[0]:.6f})"
        if n <= a.host_max:
            r_np = ranks.cpu().numpy().T
            t3 = time.perf_counter()
            host = mm.compute_map(r_np, labels, [1, 5, 10])
            t4 = time.perf_counter()
            line += f"; host numpy tail {1e3*(t4-t3):.0f} ms (+ {n*n*8/1e9:.2f} GB D2H), |dmAP| {abs(host[0]-got[0]):.1e}"
        print(line, flush=True)
        del ranks


if __name__ == "__main__":
    main()
