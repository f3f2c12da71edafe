#!/usr/bin/env python3
"""Development probe: is the two-fp16-term DenseNet path sensitive to what ran before it in the process?"""
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
x5 = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(77))
x5[1] *= 30.0
x5[2] *= 1e-3
x6 = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(7))
ref = OD.embed(x5, sd)


def run(tag):
    with torch.no_grad():
        e = m(x5.cuda()).cpu()
    r = m._mirx_last_ranges.clone().cpu()
    print(f"{tag}: err {float((e - ref).abs().max()):.2e}  ranges max per row[:8] {r.amax(1)[:8].tolist()}")
    return e, r


e1, r1 = run("fresh")
steps = os.environ.get("PROBE_STEPS", "b6,eager,m2").split(",")
if "b6" in steps:
    with torch.no_grad():
        m(x6.cuda())
    e, r = run("after B=6 forward")
if "eager" in steps:
    y2 = m(x6[:2].cuda()).detach()
    e, r = run("after eager forward")
if "m2" in steps:
    torch.manual_seed(1)
    m2 = mm.DenseNet121(embedding_dim=256, num_labels=3).eval().cuda()
    with torch.no_grad():
        m2(x6[:2].cuda())
    e, r = run("after m2 forward")
d = (r - r1).abs().amax(1)
print("range rows that differ from the fresh run:", [(i, float(r1[i].max()), float(r[i].max())) for i in torch.nonzero(d > 0).flatten().tolist()][:20])
