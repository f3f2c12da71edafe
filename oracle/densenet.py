"""CPU restatement of the DenseNet-121 embedder (test infrastructure only).

Follows the reference wrapper model.py:50-84 around torchvision's densenet121 `features`
(a third-party dependency that is neither vendored under /root/reference nor installed:
backbone parity is UNPINNED -- see oracle/__init__.py).  Architecture restated from the
published definition (Huang et al. 2017, Table 1, DenseNet-121, growth 32, bn_size 4):
conv0 7x7/2 -> BN -> ReLU -> maxpool 3x3/2 -> 4 dense blocks (6,12,24,16) with transitions
(BN-ReLU-conv1x1 halve - avgpool2) -> norm5 -> [reference adds] ReLU -> global avgpool ->
flatten -> optional fc -> L2 normalise.

Pure functional code over a state dict with the REFERENCE's key layout
(`densenet121.0.<torchvision features name>`, `fc.*`), so it is independent of the
product's module classes.  Known answers it is checked by (tests/test_model_cpu.py):
6 953 856 feature parameters, 1024 x H/32 x W/32 final map, 364 feature tensors.
"""
import torch
import torch.nn.functional as F

BLOCKS = (6, 12, 24, 16)
PFX = "densenet121.0."


def _bn(x, sd, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


def feature_map(x, sd):
    """[B,3,H,W] -> [B,1024,H/32,W/32] after norm5 and the reference's extra ReLU."""
    x = F.conv2d(x, sd[PFX + "conv0.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, PFX + "norm0"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for bi, nl in enumerate(BLOCKS, start=1):
        feats = [x]
        for li in range(1, nl + 1):
            p = f"{PFX}denseblock{bi}.denselayer{li}."
            cat = torch.cat(feats, 1)
            y = F.conv2d(F.relu(_bn(cat, sd, p + "norm1")), sd[p + "conv1.weight"])
            y = F.conv2d(F.relu(_bn(y, sd, p + "norm2")), sd[p + "conv2.weight"], padding=1)
            feats.append(y)
        x = torch.cat(feats, 1)
        if bi != len(BLOCKS):
            p = f"{PFX}transition{bi}."
            x = F.conv2d(F.relu(_bn(x, sd, p + "norm")), sd[p + "conv.weight"])
            x = F.avg_pool2d(x, kernel_size=2, stride=2)
    return F.relu(_bn(x, sd, PFX + "norm5"))


def embed(x, sd):
    """The reference forward (model.py:71-84) without a classification head."""
    x = feature_map(x, sd)
    x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    if "fc.weight" in sd:
        x = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    return F.normalize(x, dim=1)


def stem(x, sd):
    """conv0 -> norm0 -> relu0 -> pool0 only (for the HIP stem kernel's parity test)."""
    x = F.conv2d(x, sd[PFX + "conv0.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, PFX + "norm0"))
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def randomize_bn_stats(sd, seed=0):
    """Give BatchNorm non-trivial running statistics/affine so folding errors are visible."""
    g = torch.Generator().manual_seed(seed)
    out = dict(sd)
    for k in sd:
        if k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".weight") and sd[k].dim() == 1:
            out[k] = 0.75 + 0.5 * torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".bias") and sd[k].dim() == 1 and "fc." not in k:
            out[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return out
