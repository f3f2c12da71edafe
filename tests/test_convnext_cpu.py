"""ConvNeXtV2-base wrapper: known answers, oracle agreement and an independent executable
cross-check against transformers.ConvNextV2Model built from a local config (no download)."""
import pytest
import torch

from oracle import convnext as OC


@pytest.fixture(scope="module")
def model():
    from mirx.model import ConvNeXtV2
    torch.manual_seed(0)
    m = ConvNeXtV2(embedding_dim=256).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                         # GRN / LayerNorm parameters away from their trivial init
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(0.75 + 0.5 * torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m


def test_known_answers(model):
    nparam = sum(p.numel() for p in model.convnext.parameters())
    assert nparam == 87_692_800                                   # SURVEY 8c known answer
    sd = model.state_dict()
    for k, shape in (("convnext.stem.0.weight", (128, 3, 4, 4)), ("convnext.stem.1.bias", (128,)),
                     ("convnext.stages.1.downsample.1.weight", (256, 128, 2, 2)),
                     ("convnext.stages.2.blocks.26.conv_dw.weight", (512, 1, 7, 7)),
                     ("convnext.stages.3.blocks.2.mlp.fc1.weight", (4096, 1024)),
                     ("convnext.stages.0.blocks.0.mlp.grn.weight", (512,)),
                     ("convnext.head.norm.weight", (1024,)), ("fc.weight", (256, 1024))):
        assert tuple(sd[k].shape) == shape, k
    assert model.convnext.num_features == 1024
    with pytest.raises(RuntimeError):
        from mirx.model import ConvNeXtV2
        ConvNeXtV2(pretrained=True)


def test_forward_matches_oracle_and_hf(model):
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = model(x)
        feats = model.convnext(x)
    assert y.shape == (2, 256) and feats.shape == (2, 1024)
    torch.testing.assert_close(y.norm(dim=1), torch.ones(2), atol=1e-6, rtol=0)
    torch.testing.assert_close(y, OC.embed(x, sd), atol=1e-6, rtol=0)
    from transformers import ConvNextV2Config, ConvNextV2Model
    hf = ConvNextV2Model(ConvNextV2Config(depths=[3, 3, 27, 3], hidden_sizes=[128, 256, 512, 1024],
                                          layer_norm_eps=1e-6)).eval()
    missing, unexpected = hf.load_state_dict(OC.to_hf_state_dict(sd), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    with torch.no_grad():
        pooled = hf(pixel_values=x).pooler_output
    torch.testing.assert_close(feats, pooled, atol=2e-5, rtol=1e-4)
