#!/usr/bin/env python3
"""Sweep of the small-launch limits (mirx_set_tuning) over the reference's batch sizes (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mirx import _lib  # noqa: E402
from mirx.model import DenseNet121  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DenseNet121().eval().to(dev)
lib = _lib.load()
x = bench.synthetic_images(256, 224, 1, dev)


def ms(b):
    xb = x[:b].contiguous()
    with torch.no_grad():
        for _ in range(3):
            m(xb)
        torch.cuda.synchronize()
        it = max(5, 400 // b)
        t0 = time.perf_counter()
        for _ in range(it):
            m(xb)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


batches = (8, 16, 32, 64, 128, 256)
print("c1x1 c3x3 | " + " ".join(f"B={b:<6d}" for b in batches))
for t1 in (0, 64, 128, 256, 512, 1024):
    for t3 in (0, 48, 96, 192, 384):
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV1X1_SMALL_MAX_WG, t1), "t")
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, t3), "t")
        print(f"{t1:4d} {t3:4d} | " + " ".join(f"{ms(b):8.3f}" for b in batches), flush=True)
