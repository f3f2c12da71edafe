#!/usr/bin/env python3
"""Times the two-fp16-term DenseNet stem (development tool): ms per 1024 images; with MIRX_LIB_PATH pointing at a diagnostic
build (-DMIRX_STEM_EXP=1 no K loop, 2 no conv-tile epilogue, 4 no pooling) it shows what each phase costs."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib  # noqa: E402
from mirx.model import _stem_weights_split2h  # noqa: E402

vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(b, 3, 224, 224, generator=g, device=dev)
    w2, osc = _stem_weights_split2h(torch.randn(64, 3, 7, 7, generator=g, device=dev) * 0.05)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    y = torch.empty(b, 64, 56, 56, device=dev)
    rin, rout = x.abs().amax(dim=(1, 2, 3)).contiguous(), torch.zeros(b, device=dev)      # ranges are per image
    for it in range(12):
        if it == 2:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.mirx_stem_conv7_bn_relu_pool_split2h_into(vp(x), vp(w2), vp(osc), vp(sc), vp(sh), b, 224, 224, vp(y),
                                                                 64 * 56 * 56, vp(rin), vp(rout), None), "stem")
    e1.record()
    torch.cuda.synchronize()
    print(f"stem B={b}: {e0.elapsed_time(e1) / 10 * 1024 / b:.3f} ms per 1024 images", flush=True)


if __name__ == "__main__":
    main()
