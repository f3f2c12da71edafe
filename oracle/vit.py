"""CPU restatement of the DINOv2 ViT-B/14 embedder (test infrastructure only).

Follows the reference wrappers model.py:448-494 (`DinoV2`) and
nih_multilabel_retrieval.py:170-221 (`DINOv2MultiLabelRetrievalModel`) around timm's
`vit_base_patch14_dinov2.lvd142m` (timm==0.9.7, requirements.txt:13; not vendored, not installed
-> backbone parity UNPINNED by the reference).  Architecture restated from the published DINOv2 /
ViT definition: 14x14 patch embedding, [CLS] token, learned position embedding, 12 pre-norm blocks
(LayerNorm eps 1e-6, 12-head attention with qkv bias, LayerScale, GELU MLP x4), final LayerNorm,
CLS pooling.  Known answers: 86 579 712 parameters at 518 px (1370 tokens); cross-check:
transformers.Dinov2Model from a local config (tests/test_vit_cpu.py).

Key layout = the reference wrappers': `backbone.` + timm names.
"""
import torch
import torch.nn.functional as F

EPS = 1e-6
P = "backbone."


def tokens(x, sd, heads=12):
    """[B,3,H,W] -> [B, 1 + (H/14)(W/14), C] after the final norm (timm forward_features)."""
    w = sd[P + "patch_embed.proj.weight"]
    x = F.conv2d(x, w, sd[P + "patch_embed.proj.bias"], stride=w.shape[-1])
    b, c, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[P + "cls_token"].expand(b, -1, -1), x], dim=1) + sd[P + "pos_embed"]
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith(P + "blocks."))
    hd = c // heads
    for i in range(depth):
        bp = f"{P}blocks.{i}."
        y = F.layer_norm(x, (c,), sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], EPS)
        qkv = F.linear(y, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"])
        qkv = qkv.reshape(b, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        a = torch.softmax(q @ k.transpose(-2, -1), dim=-1) @ v
        a = a.transpose(1, 2).reshape(b, -1, c)
        a = F.linear(a, sd[bp + "attn.proj.weight"], sd[bp + "attn.proj.bias"])
        x = x + a * sd[bp + "ls1.gamma"]
        y = F.layer_norm(x, (c,), sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], EPS)
        y = F.linear(F.gelu(F.linear(y, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"])),
                     sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"])
        x = x + y * sd[bp + "ls2.gamma"]
    return F.layer_norm(x, (c,), sd[P + "norm.weight"], sd[P + "norm.bias"], EPS)


def embed(x, sd, heads=12):
    """model.py:488-494: CLS feature -> optional fc -> L2 normalise."""
    f = tokens(x, sd, heads)[:, 0]
    if "fc.weight" in sd:
        f = F.linear(f, sd["fc.weight"], sd["fc.bias"])
    return F.normalize(f, dim=1)


def nih_forward(x, sd, heads=12):
    """nih_multilabel_retrieval.py:209-221 -> dict(cls_embedding, projection, embedding, logits)."""
    cls = tokens(x, sd, heads)[:, 0]
    p = F.linear(cls, sd["projection_head.0.weight"], sd["projection_head.0.bias"])
    p = F.linear(F.gelu(p), sd["projection_head.2.weight"], sd["projection_head.2.bias"])
    return {"cls_embedding": cls, "projection": p, "embedding": F.normalize(p, dim=1),
            "logits": F.linear(p, sd["classification_head.weight"], sd["classification_head.bias"])}


def to_hf_dinov2(sd):
    """timm-layout keys -> transformers.Dinov2Model keys (qkv split into query/key/value)."""
    out = {}
    c = sd[P + "cls_token"].shape[-1]
    for k, v in sd.items():
        if not k.startswith(P):
            continue
        k2 = k[len(P):]
        if k2 == "cls_token":
            out["embeddings.cls_token"] = v
        elif k2 == "pos_embed":
            out["embeddings.position_embeddings"] = v
        elif k2.startswith("patch_embed.proj."):
            out["embeddings.patch_embeddings.projection." + k2.split(".")[-1]] = v
        elif k2.startswith("norm."):
            out["layernorm." + k2.split(".")[-1]] = v
        elif k2.startswith("blocks."):
            _, i, rest = k2.split(".", 2)
            lp = f"encoder.layer.{i}."
            if rest.startswith("attn.qkv."):
                kind = rest.split(".")[-1]
                for j, name in enumerate(("query", "key", "value")):
                    out[f"{lp}attention.attention.{name}.{kind}"] = v[j * c:(j + 1) * c]
            elif rest.startswith("attn.proj."):
                out[lp + "attention.output.dense." + rest.split(".")[-1]] = v
            elif rest == "ls1.gamma":
                out[lp + "layer_scale1.lambda1"] = v
            elif rest == "ls2.gamma":
                out[lp + "layer_scale2.lambda1"] = v
            else:
                out[lp + rest] = v
    return out
