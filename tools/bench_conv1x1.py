#!/usr/bin/env python3
"""Layer-level A/B of the DenseNet 1x1-conv kernels (development tool): three-bf16-term (split3) vs two-fp16-term
(split2h) on the shapes of every dense layer, at a given batch; prints ms per layer group and the error of each
against a float64 restatement on one small case."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib  # noqa: E402
from mirx.model import _split2h_weights, _split3_weights  # noqa: E402

vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def run(lib, kind, buf, ctot, cin, sc, sh, w3, w2, osc, bias, n, hw, y, slots_in, slots_out, ks, kb):
    if kind == "s3":
        _lib.check(lib.mirx_conv1x1_bn_relu_split3(vp(buf), ctot * hw, cin, vp(sc), vp(sh), vp(w3), vp(bias), n, hw, 128, 1,
                                                   vp(y), 128 * hw, None), "s3")
    else:
        _lib.check(lib.mirx_conv1x1_bn_relu_split2h(vp(buf), ctot * hw, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(bias), n, hw,
                                                    128, 1, vp(y), 128 * hw, vp(slots_in), ks, kb, vp(slots_out), 0, 0, None), "h2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stamps", action="store_true", help="diagnostic library (-DMIRX_C1H2_STAMPS): cycle stamps of the last h2 launch per block")
    ap.add_argument("--hw", type=int, nargs="*", default=None, help="experiment: block-3 layer set (256..992 channels) at these pixel counts per image")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    blocks = ((56, 64, 6), (28, 128, 12), (14, 256, 24), (7, 512, 16))
    if a.hw:
        blocks = tuple((-h, 256, 24) for h in a.hw)
    tot = {"s3": 0.0, "h2": 0.0}
    for side, c0, nl in blocks:
        hw = side * side if side > 0 else -side
        ctot = c0 + 32 * nl
        buf = torch.randn(a.batch, ctot, hw, generator=g, device=dev)
        y = torch.empty(a.batch, 128, hw, device=dev)
        slots_in = torch.full((a.batch,), float(buf.abs().max()), device=dev)      # range rows: one float per image
        slots_out = torch.zeros(a.batch, device=dev)
        layers = []
        for i in range(nl):
            cin = c0 + 32 * i
            w = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
            sc = torch.rand(cin, generator=g, device=dev) + 0.5
            sh = torch.randn(cin, generator=g, device=dev) * 0.3
            bias = torch.randn(128, generator=g, device=dev)
            w2, osc = _split2h_weights(w)
            layers.append((cin, sc, sh, _split3_weights(w), w2, osc, bias, float(sc.abs().max()), float(sh.abs().max())))
        for kind in ("s3", "h2"):
            for it in range(a.iters + 1):
                if it == 1:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                for (cin, sc, sh, w3, w2, osc, bias, ks, kb) in layers:
                    run(lib, kind, buf, ctot, cin, sc, sh, w3, w2, osc, bias, a.batch, hw, y, slots_in, slots_out, ks, kb)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            tot[kind] += ms
            gb = sum((l[0] + 128) * 4 * hw * a.batch for l in layers) / 1e9
            print(f"side {side:2d} ({nl:2d} layers) {kind}: {ms:7.3f} ms   {gb / ms:6.2f} TB/s algorithmic", flush=True)
            if kind == "h2" and a.stamps:
                import numpy as np
                dbg = ctypes.CDLL(_lib.LIB_PATH)
                sb = np.zeros(8192 * 8, dtype=np.uint64)
                assert dbg.mirx_debug_c1_stamps(sb.ctypes.data_as(ctypes.c_void_p)) == 0
                t8 = sb.reshape(8192, 8)
                valu = (t8[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
                t = t8.astype(np.float64)
                t[:, 7] = (t8[:, 7] >> np.uint64(32)).astype(np.float64)
                keep = t[:, 2] > 0
                t, valu = t[keep], valu[keep]
                nk_ = t[:, 2]
                md = lambda v: float(np.median(v))      # noqa: E731
                print(f"   stamps (last layer, cin {layers[-1][0]}, {len(t)} wgs): K loop {md(t[:, 0] / nk_):.0f} cycles/stage = wait {md(t[:, 3] / nk_):.0f} "
                      f"+ barrier {md(t[:, 4] / nk_):.0f} + issue {md(t[:, 5] / nk_):.0f} + frags/mfma {md(t[:, 6] / nk_):.0f} + BN / ReLU / split {md(valu / nk_):.0f} + LDS stores and the rest {md(t[:, 7] / nk_):.0f}; "
                      f"clock {md(t[:, 0] / t[:, 1] * 100):.0f} MHz", flush=True)
        del buf, y
    print(f"total s3 {tot['s3']:.2f} ms, h2 {tot['h2']:.2f} ms per {a.batch} images")

    # error of both against float64 on one layer
    n, cin, hw = 4, 512, 196
    buf = torch.randn(n, cin + 32, hw, generator=g, device=dev) * 3
    w = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3
    bias = torch.randn(128, generator=g, device=dev)
    w2, osc = _split2h_weights(w)
    w3 = _split3_weights(w)
    slots_in = torch.full((n,), float(buf.abs().max()), device=dev)
    slots_out = torch.zeros(n, device=dev)
    want = torch.relu(torch.einsum("oc,bcp->bop", w.double(),
                                   torch.relu(buf[:, :cin].double() * sc.double()[None, :, None] + sh.double()[None, :, None]))
                      + bias.double()[None, :, None])
    for kind in ("s3", "h2"):
        y = torch.empty(n, 128, hw, device=dev)
        run(lib, kind, buf, cin + 32, cin, sc, sh, w3, w2, osc, bias, n, hw, y, slots_in, slots_out, float(sc.abs().max()),
            float(sh.abs().max()))
        torch.cuda.synchronize()
        print(f"{kind}: max err vs float64 {float((y.double() - want).abs().max()):.3e} (max |y| {float(want.abs().max()):.2f})")
    print("published out range", float(slots_out.max()), "true", float(want.abs().max()))


if __name__ == "__main__":
    main()
