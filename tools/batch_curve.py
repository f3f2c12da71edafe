#!/usr/bin/env python3
"""DenseNet-121 forward at the reference's batch sizes (development tool; bench.py puts the same curve on the line)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mirx.model import DenseNet121  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="1,2,4,8,16,32,64,128,256,1024,2048")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DenseNet121().eval().to(dev)
args = argparse.Namespace(image_size=224)
for r in bench.densenet_batch_curve(m, args, dev, tuple(int(v) for v in a.batches.split(","))):
    print(f"B={r['batch']:5d}  {r['ms_per_forward']:8.3f} ms  {r['images_per_s']:9.1f} img/s", flush=True)
x = bench.synthetic_images(1, 224, 1, dev)
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m._cache()
    print(f"cache validity check: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per forward")
