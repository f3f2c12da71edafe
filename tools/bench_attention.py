"""Times the attention kernels (fp32 MFMA / three-term bf16 MFMA / two-term fp16 MFMA) and torch SDPA on a ViT shape."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=1370)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--head-dim", type=int, default=64)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    qkv = torch.randn(a.batch, a.tokens, 3, a.heads, a.head_dim, device=dev)
    out = torch.empty(a.batch, a.tokens, a.heads * a.head_dim, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    flop = 4.0 * a.batch * a.heads * a.tokens * a.tokens * a.head_dim
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    for name, fn in (("fp32 MFMA", lambda: _lib.check(lib.mirx_attention_qkv_f32(vp(qkv), a.batch, a.tokens, a.heads, a.head_dim, a.head_dim ** -0.5, vp(out), None), "a")),
                     ("split-3 bf16 MFMA", lambda: _lib.check(lib.mirx_attention_qkv_f32_split3(vp(qkv), a.batch, a.tokens, a.heads, a.head_dim, a.head_dim ** -0.5, vp(out), None), "a")),
                     ("split-2 fp16 MFMA", lambda: _lib.check(lib.mirx_attention_qkv_f32_split2h(vp(qkv), a.batch, a.tokens, a.heads, a.head_dim, a.head_dim ** -0.5, 6.0, 6.0, vp(out), None), "a")),
                     ("torch SDPA fp32", lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        print(f"{name:20s} {dt*1e3:7.3f} ms  {flop/dt/1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
