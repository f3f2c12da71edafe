"""DINOv2 ViT-B/14 wrappers: known answers, oracle agreement, and an independent executable
cross-check against transformers.Dinov2Model built from a local config (no download)."""
import pytest
import torch

from oracle import vit as OV


def _perturb(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("gamma"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))          # LayerScale away from 1e-5
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "cls_token" in n:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m


def test_known_answers_and_contract():
    from mirx.model import DinoV2
    m = DinoV2(embedding_dim=512)
    assert sum(p.numel() for p in m.backbone.parameters()) == 86_579_712
    sd = m.state_dict()
    assert sd["backbone.pos_embed"].shape == (1, 1370, 768)
    assert sd["backbone.patch_embed.proj.weight"].shape == (768, 3, 14, 14)
    assert sd["backbone.blocks.11.attn.qkv.weight"].shape == (2304, 768)
    assert sd["backbone.blocks.0.ls1.gamma"].shape == (768,) and sd["fc.weight"].shape == (512, 768)
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert "backbone.blocks.9.attn.qkv.weight" in trainable and "backbone.blocks.8.attn.qkv.weight" not in trainable
    assert "backbone.norm.weight" in trainable and "backbone.pos_embed" not in trainable
    with pytest.raises(RuntimeError):
        DinoV2(pretrained=True)
    with pytest.raises(ValueError):
        DinoV2(model_name="vit_huge")


def test_forward_matches_oracle_and_hf():
    from mirx.model import DinoV2, DINOv2MultiLabelRetrievalModel
    torch.manual_seed(0)
    m = _perturb(DinoV2(embedding_dim=64, img_size=70), 1).eval()        # 5x5 patches + CLS
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 70, 70, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = m(x)
        toks = m.backbone.forward_features(x)
    assert y.shape == (2, 64) and toks.shape == (2, 26, 768)
    torch.testing.assert_close(y, OV.embed(x, sd), atol=2e-6, rtol=0)
    from transformers import Dinov2Config, Dinov2Model
    hf = Dinov2Model(Dinov2Config(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, patch_size=14,
                                  image_size=70, layer_norm_eps=1e-6)).eval()
    missing, unexpected = hf.load_state_dict(OV.to_hf_dinov2(sd), strict=False)
    assert set(missing) <= {"embeddings.mask_token"} and not unexpected, (missing, unexpected)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state
    torch.testing.assert_close(toks, ref, atol=3e-5, rtol=1e-4)
    # NIH multi-label variant: dict contract
    torch.manual_seed(3)
    n = _perturb(DINOv2MultiLabelRetrievalModel(num_labels=14, img_size=70), 4).eval()
    with torch.no_grad():
        out = n(x)
    ref = OV.nih_forward(x, {k: v.detach() for k, v in n.state_dict().items()})
    assert set(out) == {"cls_embedding", "projection", "embedding", "logits"}
    for k in out:
        torch.testing.assert_close(out[k], ref[k], atol=2e-6, rtol=1e-5)
    assert out["embedding"].shape == (2, 256) and out["logits"].shape == (2, 14)


def test_medsiglip_wrapper_contract():
    """MedSigLIP (model.py:536-634): tiny local vision config, output contract + key layout."""
    from mirx.model import MedSigLIP, build_model
    torch.manual_seed(0)
    cfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56,
               patch_size=14)
    m = MedSigLIP(embed_dim=32, unfreeze_layers=1, vision_config=cfg).eval()
    x = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (2, 32)
    torch.testing.assert_close(y.norm(dim=1), torch.ones(2), atol=1e-6, rtol=0)
    sd = m.state_dict()
    assert "backbone.embeddings.patch_embedding.weight" in sd and "backbone.post_layernorm.weight" in sd
    assert "backbone.head.probe" in sd and "projection.0.weight" in sd and "projection.3.bias" in sd
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert "backbone.encoder.layers.2.mlp.fc1.weight" in trainable
    assert "backbone.encoder.layers.1.mlp.fc1.weight" not in trainable
    assert "projection.0.weight" in trainable
    # reference forward: projection(pooler_output) then normalise
    with torch.no_grad():
        ref = torch.nn.functional.normalize(m.projection(m.backbone(pixel_values=x).pooler_output), dim=1)
    torch.testing.assert_close(y, ref)
    m2 = MedSigLIP(embed_dim=32, vision_config=cfg, weights={"state-dict": sd}).eval()
    with torch.no_grad():
        torch.testing.assert_close(m2(x), y)
    assert m.verify_attention_output("cpu")
    m.ensure_eager_attention()
    with pytest.raises(ValueError):
        build_model("clip-vit")
