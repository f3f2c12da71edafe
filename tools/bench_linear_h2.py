"""Times mirx_linear_split2h (k_linear_h2) on the token-major backbones' layer shapes: ms and fp32-equivalent TFLOP/s
(3 fp16 MFMAs per product: the bar is 838.9).  MIRX_LIB_PATH selects a diagnostic build."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib                                   # noqa: E402
from mirx import model as mm                            # noqa: E402

SHAPES = {
    "dinov2": (32 * 1370, (("qkv", 768, 2304, 0, False), ("proj+res", 768, 768, 0, True), ("fc1+gelu", 768, 3072, 1, False),
                           ("fc2+res", 3072, 768, 0, True))),
    "medsiglip": (16 * 1024, (("qkv", 1152, 3456, 0, False), ("proj+res", 1152, 1152, 0, True),
                              ("fc1+gelu_tanh", 1152, 4304, 2, False), ("fc2+res", 4304, 1152, 0, True))),
    "convnextv2": (64 * 24 * 24, (("s3 fc1+gelu", 512, 2048, 1, False),)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinov2", choices=sorted(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    m, shapes = SHAPES[a.model]
    tot = 0.0
    for name, k, n, act, res in shapes:
        if k % 16:
            k = (k + 15) // 16 * 16
        torch.manual_seed(0)
        x = torch.randn(m, k, device=dev).clamp_(-8, 8)
        lin = torch.nn.Linear(k, n).to(dev)
        w2, ws = mm._linear_h2_weights(lin)
        xs = 2.0 ** 11
        r = torch.randn(m, n, device=dev) if res else None
        y = torch.empty(m, n, device=dev)

        def ours():
            _lib.check(lib.mirx_linear_split2h(vp(x), m, k, vp(w2), vp(lin.bias.detach()), n, act, vp(r) if res else None, None,
                                               xs, 1.0 / (xs * ws), vp(y), None), "lin")

        for _ in range(3):
            ours()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            ours()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        tot += dt
        fl = 2.0 * m * k * n
        print(f"{name:14s} m={m} k={k} n={n}: {dt*1e3:7.3f} ms {fl/dt/1e12:6.1f} TF-eq = {fl/dt/838.9e12:.3f} of 838.9", flush=True)
    print(f"total {tot*1e3:.3f} ms")


if __name__ == "__main__":
    main()
