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


def test_medsiglip_head_dim_72_uses_fused_tower_path():
    """SigLIP-So400m geometry in small: head_dim 72 -> packed-qkv Linear + the flash-attention kernel on two fp16 terms
    (mirx_attention_qkv_f32_split2h, q / k / v bounded through the LayerNorm), MLP width not a multiple of 128 (padded output
    tile).  CPU (the module graph) vs the routed GPU path."""
    import mirx.model as mm
    torch.manual_seed(1)
    cfg = dict(hidden_size=144, intermediate_size=208, num_hidden_layers=2, num_attention_heads=2, image_size=84,
               patch_size=14)
    m = mm.MedSigLIP(embed_dim=32, vision_config=cfg).eval()
    x = torch.randn(3, 3, 84, 84)
    with torch.no_grad():
        ref = m(x)
        m = m.cuda()
        calls = []
        lib = mm._lib.load()
        orig = lib.mirx_attention_qkv_f32_split2h

        class _Spy:
            def __call__(self, *a):
                calls.append(a[4])
                return orig(*a)
        try:
            lib.mirx_attention_qkv_f32_split2h = _Spy()
            y = m(x.cuda()).cpu()
        finally:
            lib.mirx_attention_qkv_f32_split2h = orig
    assert calls == [72, 72]                                       # one launch per encoder layer, head_dim 72
    assert float((y - ref).abs().max()) <= 1e-5
    assert m.verify_attention_output("cuda")                       # attention maps still come from the module path


def test_dinov2_rows_do_not_depend_on_the_batch():
    """the bench's batch (32 at 518 x 518, 1370 tokens) against the same images embedded two at a time
    (<= 1e-6) and against the CPU restatement (1e-5)."""
    from mirx.model import DinoV2
    torch.manual_seed(0)
    m = DinoV2(embedding_dim=256).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("gamma"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(32, 3, 518, 518, generator=torch.Generator().manual_seed(5))
    m = m.cuda()
    with torch.no_grad():
        big = m(x.cuda()).cpu()
        small = torch.cat([m(x[i:i + 2].cuda()).cpu() for i in (0, 14, 30)])
        ref = OV.embed(x[31:32], sd)
    assert float((big[[0, 1, 14, 15, 30, 31]] - small).abs().max()) <= 1e-6
    assert float((big[31:32] - ref).abs().max()) <= 1e-5
