"""Own SigLIP towers (mirx.siglip) against transformers.SiglipModel built from a LOCAL config: identical parameter
names (checkpoints load unchanged) and identical outputs on CPU, with and without a key-padding mask.  CPU only."""
import pytest
import torch

V = dict(hidden_size=64, intermediate_size=112, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14)
T = dict(hidden_size=64, intermediate_size=112, num_hidden_layers=2, num_attention_heads=4, vocab_size=100,
         max_position_embeddings=16, projection_size=64)


def _feat(o):
    return o if torch.is_tensor(o) else o.pooler_output


def test_dual_encoder_matches_transformers():
    tr = pytest.importorskip("transformers")
    from mirx.siglip import SiglipDualEncoder
    torch.manual_seed(0)
    cfg = tr.SiglipConfig(text_config=dict(T, bos_token_id=1, eos_token_id=2, pad_token_id=0), vision_config=dict(V))
    hf = tr.SiglipModel(cfg).eval()
    m = SiglipDualEncoder(V, T).eval()
    missing, unexpected = m.load_state_dict(hf.state_dict(), strict=False)
    assert not missing and not unexpected                    # same key layout: reference checkpoints load unchanged
    assert set(m.state_dict()) == set(hf.state_dict())
    ids = torch.randint(0, 100, (3, 16))
    mask = torch.ones(3, 16, dtype=torch.long)
    mask[0, 5:] = 0
    mask[2, 11:] = 0
    px = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        for am in (mask, None):
            a = _feat(hf.get_text_features(input_ids=ids, attention_mask=am))
            b = m.get_text_features(ids, am)
            assert a.shape == b.shape == (3, 64)
            assert float((a - b).abs().max()) <= 2e-6
        c, d = _feat(hf.get_image_features(pixel_values=px)), m.get_image_features(px)
        assert float((c - d).abs().max()) <= 2e-6
        o_hf = hf(input_ids=ids, pixel_values=px, attention_mask=mask)
        o = m(input_ids=ids, pixel_values=px, attention_mask=mask)
        assert float((o_hf.logits_per_image - o.logits_per_image).abs().max()) <= 2e-5
        assert float((o_hf.text_embeds - o.text_embeds).abs().max()) <= 2e-6
        # attention maps for the reference's rollout explainer (model.py:546-551)
        out = m.vision_model(pixel_values=px, output_attentions=True)
        assert len(out.attentions) == 2 and out.attentions[0].shape == (2, 4, 16, 16)
        assert float((out.attentions[0].sum(-1) - 1).abs().max()) < 1e-5
    with pytest.raises(ValueError):
        m.text_model(input_ids=torch.zeros(1, 17, dtype=torch.long))
    with pytest.raises(ValueError):
        m.vision_model(pixel_values=torch.zeros(1, 3, 70, 70))


def test_full_geometry_parameter_counts():
    """MedSigLIP = SigLIP so400m: 428 565 440 vision-tower parameters (the count SURVEY 8c quotes)."""
    from mirx.siglip import MEDSIGLIP_TEXT, MEDSIGLIP_VISION, SiglipTextTower, SiglipVisionTower
    with torch.device("meta"):
        v = SiglipVisionTower(**MEDSIGLIP_VISION)
        t = SiglipTextTower(**MEDSIGLIP_TEXT)
    assert sum(p.numel() for p in v.parameters()) == 428_565_440
    c, f = 1152, 4304
    layer = 4 * (c * c + c) + (c * f + f) + (f * c + c) + 4 * c
    assert sum(p.numel() for p in t.parameters()) == 27 * layer + 32000 * c + 64 * c + 2 * c + c * c + c
