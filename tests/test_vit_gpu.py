"""DINOv2 / MedSigLIP wrappers on the GPU against the CPU restatement (fp32, 1e-5 on unit-norm
embeddings)."""
import pytest
import torch

from oracle import vit as OV

pytestmark = pytest.mark.gpu


def test_dinov2_embeddings_match_cpu_restatement():
    from mirx.model import DinoV2
    torch.manual_seed(0)
    m = DinoV2(embedding_dim=256).eval()                        # native 518 px, 1370 tokens
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("gamma"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 518, 518, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = OV.embed(x[:1], sd)
        y = m.cuda()(x.cuda()).cpu()
    assert y.shape == (2, 256)
    assert float((y.norm(dim=1) - 1).abs().max()) < 1e-6
    assert float((y[:1] - ref).abs().max()) <= 1e-5, float((y[:1] - ref).abs().max())
    # a non-native size goes through the position-embedding resampling
    with torch.no_grad():
        z = m(torch.randn(1, 3, 224, 224, device="cuda"))
    assert z.shape == (1, 256)


def test_medsiglip_tiny_on_gpu():
    from mirx.model import MedSigLIP
    torch.manual_seed(0)
    cfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56,
               patch_size=14)
    m = MedSigLIP(embed_dim=32, vision_config=cfg).eval()
    x = torch.randn(3, 3, 56, 56)
    with torch.no_grad():
        ref = m(x)
        y = m.cuda()(x.cuda()).cpu()
    assert float((y - ref).abs().max()) <= 1e-5
