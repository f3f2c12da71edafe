"""DenseNet-121 wrapper: known answers + eager-vs-oracle agreement on CPU."""
import torch

from oracle import densenet as OD


def _model(**kw):
    from mirx.model import DenseNet121
    torch.manual_seed(0)
    return DenseNet121(**kw).eval()


def test_known_answers_parameter_count_and_keys():
    m = _model()
    feats = m.densenet121[0]
    nparam = sum(p.numel() for p in feats.parameters())
    assert nparam == 6_953_856                                   # SURVEY 8c known answer
    sd = m.state_dict()
    for k in ("densenet121.0.conv0.weight", "densenet121.0.norm0.running_mean",
              "densenet121.0.denseblock1.denselayer1.norm1.weight",
              "densenet121.0.denseblock3.denselayer24.conv2.weight",
              "densenet121.0.transition2.conv.weight", "densenet121.0.norm5.bias"):
        assert k in sd, k
    assert sd["densenet121.0.conv0.weight"].shape == (64, 3, 7, 7)
    assert sd["densenet121.0.denseblock4.denselayer16.conv1.weight"].shape == (128, 992, 1, 1)
    assert sd["densenet121.0.norm5.weight"].shape == (1024,)
    assert len([k for k in sd if k.startswith("densenet121.0.") and "num_batches" not in k]) == 604
    assert [n for n, _ in m.densenet121.named_children()] == ["0", "avgpool"]
    assert m.fc is None and m.classification_head is None


def test_forward_contract_and_oracle_agreement():
    m = _model()
    sd = OD.randomize_bn_stats(m.state_dict(), seed=1)
    m.load_state_dict(sd)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = m(x)
        fmap = m.densenet121[0](x)
    assert y.shape == (2, 1024) and fmap.shape == (2, 1024, 2, 2)
    torch.testing.assert_close(y.norm(dim=1), torch.ones(2), atol=1e-6, rtol=0)
    ref = OD.embed(x, sd)
    torch.testing.assert_close(y, ref, atol=1e-6, rtol=0)


def test_embedding_dim_num_labels_and_checkpoint_wrappers(tmp_path):
    m = _model(embedding_dim=256, num_labels=14)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        out = m(x)
    assert set(out) == {"embedding", "logits"}
    assert out["embedding"].shape == (1, 256) and out["logits"].shape == (1, 14)
    sd = m.state_dict()
    p = tmp_path / "ckpt.pth"
    torch.save({"state-dict": sd}, p)
    from mirx.model import DenseNet121, build_model
    m2 = DenseNet121(embedding_dim=256, num_labels=14, weights=str(p)).eval()
    with torch.no_grad():
        torch.testing.assert_close(m2(x)["embedding"], out["embedding"])
    import pytest
    with pytest.raises(RuntimeError):
        DenseNet121(pretrained=True)
    with pytest.raises(ValueError):
        build_model("nope")
