#!/usr/bin/env python3
"""Development probe: per-image embedding error of the DenseNet paths (two-fp16-term / three-bf16-term) against the CPU
restatement, for inputs of mixed magnitudes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirx.model as mm  # noqa: E402
from oracle import densenet as OD  # noqa: E402

torch.manual_seed(0)
m = mm.DenseNet121().eval()
sd = OD.randomize_bn_stats(m.state_dict(), seed=1)
m.load_state_dict(sd)
sd = {k: v.cpu() for k, v in sd.items()}
m = m.cuda()
x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(77))
for scales in ((1, 1, 1, 1, 1), (1, 30, 1e-3, 1, 1)):
    xs = x * torch.tensor(scales).view(-1, 1, 1, 1)
    with torch.no_grad():
        ref = OD.embed(xs, sd)
        ref64 = OD.embed(xs.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}) \
            if os.environ.get("PROBE_F64") else None
        e2 = m(xs.cuda()).cpu()
        old = m.configure(densenet_two_fp16=False)
        e3 = m(xs.cuda()).cpu()
        m.configure(**old.__dict__)
    print("scales", scales)
    print("  h2 vs cpu fp32 per image:", [f"{v:.2e}" for v in (e2 - ref).abs().amax(1).tolist()])
    print("  s3 vs cpu fp32 per image:", [f"{v:.2e}" for v in (e3 - ref).abs().amax(1).tolist()])
    print("  h2 vs s3        per image:", [f"{v:.2e}" for v in (e2 - e3).abs().amax(1).tolist()])
    if ref64 is not None:
        print("  h2 vs cpu fp64:", [f"{v:.2e}" for v in (e2.double() - ref64).abs().amax(1).tolist()])
        print("  s3 vs cpu fp64:", [f"{v:.2e}" for v in (e3.double() - ref64).abs().amax(1).tolist()])
        print("  cpu fp32 vs fp64:", [f"{v:.2e}" for v in (ref.double() - ref64).abs().amax(1).tolist()])
