"""mirx.xai.SBSMBatch on the CPU with a small torch model against the restatement that keeps the reference's
tensors (oracle/xai.py:sbsm_batch, explanations.py:105-152): same masks, same saliency."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("pair", [False, True])
def test_sbsm_batch_cpu(pair, tmp_path):
    from mirx.xai import SBSMBatch, sliding_window_masks
    from oracle import xai as ox
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                              torch.nn.Flatten()).eval()
    masks = sliding_window_masks((32, 40), 12, 8)
    assert masks.shape == (5 * 6, 1, 32, 40)            # starts -4, 4, ..., 28 and -4, ..., 36
    xq, xr = torch.randn(2, 3, 32, 40), torch.randn(2, 3, 32, 40)
    with torch.no_grad():
        want = ox.sbsm_batch(net, masks, xq, xr if pair else None, gpu_batch=5)
    ex = SBSMBatch(net, (32, 40), gpu_batch=5)
    ex.generate_masks(12, 8, savepath=str(tmp_path / "m.npy"))
    assert np.array_equal(np.load(tmp_path / "m.npy"), masks)
    got = ex(xq, xr if pair else None)
    assert got.shape == want.shape == ((4 if pair else 2), 32, 40)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-7)
