#!/usr/bin/env python3
"""Layer-level A/B of the DenseNet 3x3-conv kernels (development tool): ms per 1024-image layer for every kernel that
supports the map side."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib  # noqa: E402
from mirx.model import (_conv3x3_weights_split2h, _conv3x3_weights_split3, _winograd_weights,  # noqa: E402
                        _winograd_weights_split3)

vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    u, u3, w3 = _winograd_weights(w), _winograd_weights_split3(w), _conv3x3_weights_split3(w)
    w2, osc = _conv3x3_weights_split2h(w)
    for side in (56, 28, 14, 7):
        x = torch.relu(torch.randn(a.batch, 128, side, side, generator=g, device=dev))
        out = torch.empty(a.batch, 32, side, side, device=dev)
        rout = torch.zeros(a.batch, device=dev)                 # range row: one float per image
        bs = 32 * side * side
        kinds = {"wino": lambda: lib.mirx_conv3x3_winograd_nchw(vp(x), vp(u), a.batch, side, vp(out), bs, None)}
        if side != 7:
            kinds["wino3"] = lambda: lib.mirx_conv3x3_winograd_split3_nchw(vp(x), vp(u3), a.batch, side, vp(out), bs, None)
            kinds["direct3"] = lambda: lib.mirx_conv3x3_direct_split3_nchw(vp(x), vp(w3), a.batch, side, vp(out), bs, None)
        if True:
            # terms path: the bottleneck pre-split into fp16 terms [n][8][2][hw][16] (timing only: random planes)
            yt = (torch.randn(a.batch, 8, 2, side * side, 16, generator=g, device=dev) * 100).half()
            yinv = torch.full((a.batch,), 2.0 ** -7, device=dev)
            kinds["terms"] = lambda: lib.mirx_conv3x3_direct_terms_nchw(vp(yt), vp(w2), vp(osc), a.batch, side, vp(out), bs,
                                                                        vp(yinv), vp(rout), 0, None)
        for name, fn in kinds.items():
            for it in range(a.iters + 2):
                if it == 2:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(fn(), name)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            tf = 2.0 * a.batch * side * side * 128 * 32 * 9 / ms / 1e9
            print(f"side {side:2d} {name:9s}: {ms:7.3f} ms/layer  {tf:6.1f} TFLOP/s direct-equivalent", flush=True)


if __name__ == "__main__":
    main()
