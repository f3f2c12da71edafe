#!/usr/bin/env python3
"""Insertion/deletion curve of one query-hit pair (224x224, step 1000 -> 52 forwards): the reference's
sequential B=1 loop on the device embedder vs mirx.xai.CausalMetric's batched construction."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.model import DenseNet121  # noqa: E402
from mirx.xai import CausalMetric  # noqa: E402


def sequential(model, q, r, expl, step):
    hw = 224 * 224
    n_steps = (hw + step - 1) // step
    with torch.no_grad():
        qf = model(q)
        start, finish = r.clone().reshape(1, 3, hw), torch.zeros_like(r).reshape(1, 3, hw)
        order = torch.from_numpy(np.flip(np.argsort(expl.flatten())).copy()).to(q.device)
        out = np.empty(n_steps + 1)
        for i in range(n_steps + 1):
            out[i] = float(F.cosine_similarity(qf, model(start.reshape(1, 3, 224, 224)))[0].clamp(min=0))
            if i < n_steps:
                c = order[step * i: step * (i + 1)]
                start[0, :, c] = finish[0, :, c]
    return out


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = DenseNet121().eval().to(dev)
    q, r = torch.randn(1, 3, 224, 224, device=dev), torch.randn(1, 3, 224, 224, device=dev)
    expl = np.random.default_rng(0).random((224, 224))
    cm = CausalMetric(m, "del", 1000, torch.zeros_like, 224)
    for _ in range(2):
        a = sequential(m, q, r, expl, 1000)
        b = cm.evaluate(q, r, expl)[1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = sequential(m, q, r, expl, 1000)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    b = cm.evaluate(q, r, expl)[1]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"52-step deletion curve: sequential B=1 loop {1e3*(t1-t0):.1f} ms, batched {1e3*(t2-t1):.1f} ms, "
          f"max |diff| {np.abs(a-b).max():.2e}")


if __name__ == "__main__":
    main()
