#!/usr/bin/env python3
"""Embed-only micro benchmark (development tool): DenseNet-121 images/s on one GPU."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.model import ConvNeXtV2, DenseNet121, DinoV2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--no-hip-stem", action="store_true")
    ap.add_argument("--no-hip-conv1x1", action="store_true")
    ap.add_argument("--model", default="densenet121", choices=["densenet121", "convnextv2", "dinov2", "medsiglip"])
    ap.add_argument("--no-split3-linear", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over this many HIP streams (tail overlap)")
    ap.add_argument("--no-split2h", action="store_true", help="DenseNet: the three-bf16-term path of round 1")
    ap.add_argument("--no-grn-kernel", action="store_true", help="ConvNeXtV2: GRN scale vector through ATen")
    ap.add_argument("--no-split2h-attention", action="store_true", help="ViT / SigLIP: three-bf16-term flash attention")
    ap.add_argument("--plane-stride", default="", help="DenseNet: padded channel planes, e.g. 28:800,14:224")
    ap.add_argument("--no-linear-terms", action="store_true", help="ViT / SigLIP: Linears on mirx_linear_split2h (A/B arm)")
    ap.add_argument("--fused-small", action="store_true", help="DenseNet: 14 / 7 maps on the one-launch dense layer (A/B arm)")
    a = ap.parse_args()
    changes = {}
    if a.no_split3_linear:
        changes["linear_three_bf16"] = False
    if a.no_split2h:
        changes["densenet_two_fp16"] = False
    if a.no_grn_kernel:
        changes["grn_scale_kernel"] = False
    if a.no_split2h_attention:
        changes["attention_two_fp16"] = False
    if a.plane_stride:
        changes["plane_stride"] = tuple((int(kv.split(":")[0]), int(kv.split(":")[1])) for kv in a.plane_stride.split(","))
    if a.no_hip_stem:
        changes["hip_stem"] = False
    if a.no_hip_conv1x1:
        changes["hip_conv1x1"] = False
    if a.no_linear_terms:
        changes["linear_terms_min_rows"] = 0
    if a.fused_small:
        changes["fused_small_maps"] = True
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if a.model == "convnextv2":
        m = ConvNeXtV2(embedding_dim=256).eval().to(dev)
    elif a.model == "dinov2":
        m = DinoV2(embedding_dim=256).eval().to(dev)
    elif a.model == "medsiglip":
        from mirx.model import MedSigLIP
        m = MedSigLIP().eval().to(dev)
    else:
        m = DenseNet121().eval().to(dev)
    m.configure(**changes)
    if a.channels_last:
        m = m.to(memory_format=torch.channels_last)
    x = torch.randn(a.batch, 3, a.size, a.size, device=dev)
    streams = [torch.cuda.Stream() for _ in range(a.streams)] if a.streams > 1 else None
    parts = list(x.chunk(a.streams)) if streams else None

    def fwd():
        if not streams:
            return m(x)
        cur = torch.cuda.current_stream()
        outs = []
        for s_, p_ in zip(streams, parts):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs.append(m(p_))
        for s_ in streams:
            cur.wait_stream(s_)
        return outs

    with torch.no_grad():
        for _ in range(a.warmup):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            y = fwd()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"B={a.batch} {a.size}x{a.size}: {dt*1e3:.1f} ms/batch, {a.batch/dt:.0f} img/s")


if __name__ == "__main__":
    main()
