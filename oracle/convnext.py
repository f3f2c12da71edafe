"""CPU restatement of the ConvNeXtV2-base embedder (test infrastructure only).

Follows the reference wrapper model.py:87-117 around timm's `convnextv2_base` created with
num_classes=0 (model.py:96-100; timm==0.9.7 pinned in requirements.txt:13, NOT vendored and not
installed here -> backbone parity UNPINNED by the reference).  Architecture restated from the
published ConvNeXt V2 definition (Woo et al. 2023): stem conv 4x4/4 + LayerNorm2d; stages of
(3, 3, 27, 3) blocks with widths (128, 256, 512, 1024), each later stage opened by LayerNorm2d +
conv 2x2/2; block = depthwise 7x7 -> LayerNorm -> Linear(4x) -> GELU -> GRN -> Linear -> + x;
head = global average pool -> LayerNorm -> flatten.  Known answers: 87 692 800 parameters,
[B,1024] features; secondary executable cross-check: transformers.ConvNextV2Model built from a
local config (different key names, same arithmetic) -- tests/test_convnext_cpu.py.

State-dict key layout = the reference wrapper's (`convnext.` + timm names), `fc.*`.
"""
import torch
import torch.nn.functional as F

DEPTHS = (3, 3, 27, 3)
DIMS = (128, 256, 512, 1024)
EPS = 1e-6
P = "convnext."


def _ln2d(x, w, b):
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, EPS).permute(0, 3, 1, 2)


def features(x, sd):
    x = F.conv2d(x, sd[P + "stem.0.weight"], sd[P + "stem.0.bias"], stride=4)
    x = _ln2d(x, sd[P + "stem.1.weight"], sd[P + "stem.1.bias"])
    for si, depth in enumerate(DEPTHS):
        sp = f"{P}stages.{si}."
        if si > 0:
            x = _ln2d(x, sd[sp + "downsample.0.weight"], sd[sp + "downsample.0.bias"])
            x = F.conv2d(x, sd[sp + "downsample.1.weight"], sd[sp + "downsample.1.bias"], stride=2)
        for bi in range(depth):
            bp = f"{sp}blocks.{bi}."
            c = x.shape[1]
            y = F.conv2d(x, sd[bp + "conv_dw.weight"], sd[bp + "conv_dw.bias"], padding=3, groups=c)
            y = y.permute(0, 2, 3, 1)
            y = F.layer_norm(y, (c,), sd[bp + "norm.weight"], sd[bp + "norm.bias"], EPS)
            y = F.gelu(F.linear(y, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"]))
            g = torch.linalg.vector_norm(y, ord=2, dim=(1, 2), keepdim=True)
            n = g / (g.mean(dim=-1, keepdim=True) + 1e-6)
            y = y + torch.addcmul(sd[bp + "mlp.grn.bias"], sd[bp + "mlp.grn.weight"], y * n)
            y = F.linear(y, sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"])
            x = y.permute(0, 3, 1, 2) + x
    x = x.mean(dim=(2, 3), keepdim=True)
    x = _ln2d(x, sd[P + "head.norm.weight"], sd[P + "head.norm.bias"])
    return torch.flatten(x, 1)


def embed(x, sd):
    x = features(x, sd)
    if "fc.weight" in sd:
        x = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    return F.normalize(x, dim=1)


def to_hf_state_dict(sd):
    """Rename the timm-layout keys to transformers.ConvNextV2Model's (for the cross-check)."""
    out = {}
    for k, v in sd.items():
        if not k.startswith(P):
            continue
        k2 = k[len(P):]
        k2 = (k2.replace("stem.0.", "embeddings.patch_embeddings.").replace("stem.1.", "embeddings.layernorm.")
              .replace("head.norm.", "layernorm."))
        if k2.startswith("stages."):
            k2 = "encoder." + k2
            k2 = (k2.replace(".downsample.0.", ".downsampling_layer.0.").replace(".downsample.1.", ".downsampling_layer.1.")
                  .replace(".blocks.", ".layers.").replace(".conv_dw.", ".dwconv.").replace(".norm.", ".layernorm.")
                  .replace(".mlp.fc1.", ".pwconv1.").replace(".mlp.fc2.", ".pwconv2.").replace(".mlp.grn.", ".grn."))
            if ".grn." in k2:
                v = v.view(1, 1, 1, -1)
        out[k2] = v
    return out
