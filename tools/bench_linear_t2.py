"""Checks and times mirx_linear_terms (k_linear_t2) on the token-major backbones' layer shapes against mirx_linear_split2h
(k_linear_h2): error vs a float64 product, ms, fp32-equivalent TFLOP/s (bar: 838.9 = fp16 MFMA peak / 3)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import model as mm                            # noqa: E402

SHAPES = {
    "dinov2": (32 * 1370, (("qkv", 768, 2304, 0, False), ("proj+res", 768, 768, 0, True), ("fc1+gelu", 768, 3072, 1, False),
                           ("fc2+res", 3072, 768, 0, True))),
    "medsiglip": (16 * 1024, (("qkv", 1152, 3456, 0, False), ("proj+res", 1152, 1152, 0, True),
                              ("fc1+gelu_tanh", 1152, 4304, 2, False), ("fc2+res", 4304, 1152, 0, True))),
    "convnextv2": (64 * 24 * 24, (("s3 fc1+gelu", 512, 2048, 1, False),)),
    "store": (42470, (("k32 qkv-wide", 32, 2304, 0, False), ("k32 proj-wide+res", 32, 768, 0, True), ("k32 fc1-wide gelu", 32, 3072, 1, False))),
    "small": (300, (("ragged", 96, 200, 1, False), ("ragged+res", 200, 96, 0, True))),
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def stamps(name):
    """Per-workgroup cycle stamps of the LAST launch (steady state of the timing loop), wave 0 of every workgroup.  The library
    clears the buffer on every read-back, so a shape only ever shows its own workgroups."""
    import ctypes
    import numpy as np
    from mirx import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    torch.cuda.synchronize()
    assert lib.mirx_debug_lt2_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.reshape(4096, 8).astype(np.float64)
    s = s[s[:, 3] > 0]
    nk = s[:, 3]
    med = lambda v: float(np.median(v))      # noqa: E731
    print(f"   stamps[{name}] {len(s)} wgs: K loop {med(s[:, 0] / nk):.0f} cycles/stage = part1 {med(s[:, 4] / nk):.0f} + wait {med(s[:, 5] / nk):.0f} "
          f"+ barrier {med(s[:, 6] / nk):.0f} + part2 {med(s[:, 7] / nk):.0f}; epilogue median {med(s[:, 1]):.0f} mean {float(s[:, 1].mean()):.0f} cycles; "
          f"clock {med(s[:, 0] / s[:, 2] * 100):.0f} MHz", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinov2", choices=sorted(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="diagnostic library (MIRX_LT2_EXP & 32): in-kernel cycle stamps of the last launch")
    ap.add_argument("--tokens", type=int, default=0, help="override the number of token rows")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m, shapes = SHAPES[a.model]
    m = a.tokens or m
    tot = [0.0, 0.0, 0.0]
    for name, k, n, act, res in shapes:
        torch.manual_seed(0)
        x = torch.randn(m, k, device=dev).clamp_(-6, 6)
        lin = torch.nn.Linear(k, n).to(dev)
        r = torch.randn(m, n, device=dev) if res else None
        g = torch.rand(n, device=dev) + 0.5 if res else None
        bound = 6.0
        xt, xs = mm._rows_to_terms(x, bound)
        y = mm._linear_terms(lin, xt, xs, (m,), act=act, res=r, gamma=g)
        err = float("nan")
        if not a.no_check:
            rows = slice(0, min(m, 4096))
            ref = x[rows].double() @ lin.weight.double().t() + lin.bias.double()
            if act == 1:
                ref = torch.nn.functional.gelu(ref)
            elif act == 2:
                ref = torch.nn.functional.gelu(ref, approximate="tanh")
            if res:
                ref = r[rows].double() + g.double() * ref
            err = float((y[rows].double() - ref).abs().max() / ref.abs().max())
            tail = slice(max(0, m - 200), m)                                    # the ragged last tile
            ref2 = x[tail].double() @ lin.weight.double().t() + lin.bias.double()
            if act == 0 and not res:
                err = max(err, float((y[tail].double() - ref2).abs().max() / ref2.abs().max()))
        t_new = timed(lambda: mm._linear_terms(lin, xt, xs, (m,), act=act, res=r, gamma=g, out=y if not res else None), a.iters)
        t_cvt = timed(lambda: mm._rows_to_terms(x, bound), a.iters)
        t_old = float("nan")
        if k % 16 == 0:
            t_old = timed(lambda: mm._linear_h2(lin, x, bound, act=act, res=r, gamma=g), a.iters)
        if a.stamps:
            stamps(name)
        fl = 2.0 * m * k * n
        tot[0] += t_new
        tot[1] += t_old
        tot[2] += t_cvt
        print(f"{name:14s} m={m} k={k} n={n}: terms {t_new*1e3:7.3f} ms {fl/t_new/1e12:6.1f} TF-eq = {fl/t_new/838.9e12:.3f} | "
              f"split2h {t_old*1e3:7.3f} ms | rows_to_terms {t_cvt*1e3:6.3f} ms | rel err {err:.2e}", flush=True)
    print(f"total: terms {tot[0]*1e3:.3f} ms, split2h {tot[1]*1e3:.3f} ms, conversions {tot[2]*1e3:.3f} ms")


if __name__ == "__main__":
    main()
