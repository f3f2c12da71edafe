"""Times mirx_linear_split3 against torch F.linear (rocBLAS fp32) on the DINOv2 / ConvNeXtV2 layer shapes."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib                                   # noqa: E402
from mirx.model import _split3_weights                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=32 * 1370)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    for name, k, n, act in (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1+gelu", 768, 3072, 1),
                            ("fc2", 3072, 768, 0), ("cnx fc1 s3", 512, 2048, 1), ("cnx fc2 s3", 2048, 512, 0)):
        m = a.tokens
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        b = torch.randn(n, device=dev)
        w3 = _split3_weights(w)
        y = torch.empty(m, n, device=dev)

        def ours():
            _lib.check(lib.mirx_linear_split3(vp(x), m, k, vp(w3), vp(b), n, act, None, None, vp(y), None), "lin")

        def theirs():
            o = torch.nn.functional.linear(x, w, b)
            return torch.nn.functional.gelu(o) if act else o

        res = []
        for fn in (ours, theirs):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / a.iters)
        fl = 2.0 * m * k * n
        print(f"{name:12s} m={m} k={k} n={n}: split3 {res[0]*1e3:7.3f} ms {fl/res[0]/1e12:6.1f} TF-eq | "
              f"rocBLAS fp32 {res[1]*1e3:7.3f} ms {fl/res[1]/1e12:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
