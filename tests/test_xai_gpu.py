"""Batched insertion / deletion metric (mirx.xai.CausalMetric) against the sequential restatement
(oracle/xai.py) run with the CPU DenseNet oracle on the same weights.  Tolerance 2e-5 on the cosine
curve: fp32 embeddings of two independent implementations (device fused path vs CPU restatement,
1e-5 each on unit-norm rows, tests/test_model_gpu.py)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["del", "ins"])
def test_causal_metric_matches_sequential_loop(mode):
    from mirx.model import DenseNet121
    from mirx.xai import CausalMetric, gkern
    from oracle import xai as ox
    from oracle import densenet as OD
    torch.manual_seed(11)
    size, step = 64, 300                                   # 14 steps + 1; last step partial (4096 % 300)
    m = DenseNet121().eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, 3, size, size, generator=g)
    r = torch.randn(1, 3, size, size, generator=g)
    expl = torch.rand(size, size, generator=g).numpy()
    expl[3, :7] = expl[3, 7]                               # saliency ties follow argsort's order
    kern = gkern(11, math.sqrt(5))
    if mode == "del":
        sub_cpu = sub_gpu = torch.zeros_like
    else:
        sub_cpu = lambda x: torch.nn.functional.conv2d(x, kern, padding=5)               # noqa: E731
        sub_gpu = lambda x: torch.nn.functional.conv2d(x, kern.to(x.device), padding=5)  # noqa: E731
    want_auc, want_scores, want_zero = ox.causal_metric(lambda x: OD.embed(x, sd), mode, step, sub_cpu, q, r, expl)
    cm = CausalMetric(m.to(dev), mode, step, sub_gpu, input_size=size, max_batch=6)      # 15 images in 3 chunks
    got_auc, got_scores, got_zero = cm.evaluate(q.to(dev), r.to(dev), expl)
    assert got_scores.shape == want_scores.shape == (math.ceil(size * size / step) + 1,)
    np.testing.assert_allclose(got_scores, want_scores, rtol=0, atol=2e-5)
    assert abs(got_auc - want_auc) < 2e-5 and got_zero == want_zero
    # endpoints: image 0 is `start`, the last image is `finish`
    one = CausalMetric(m, mode, step, sub_gpu, input_size=size, max_batch=1024).evaluate(q.to(dev), r.to(dev), expl)
    np.testing.assert_allclose(one[1], got_scores, rtol=0, atol=2e-6)


def test_gkern_and_auc():
    from mirx.xai import auc, gkern
    k = gkern(51, math.sqrt(50))
    assert k.shape == (3, 3, 51, 51) and k.dtype == torch.float32
    assert abs(float(k[0, 0].sum()) - 1.0) < 1e-4 and float(k[0, 1].abs().sum()) == 0.0
    assert float(k[1, 1, 25, 25]) == float(k[1, 1].max())
    assert auc(np.array([1.0, 0.5, 0.0])) == 0.5


@pytest.mark.parametrize("self_sim", [True, False])
def test_sbsm_batch_matches_reference_tensors(self_sim, tmp_path):
    """mirx.xai.SBSMBatch (chunked masked forwards + one matrix product) against the restatement that keeps
    the reference's [B*N] stack and [H, W, B, N] tensor, on the CPU DenseNet oracle.  Tolerance: distances of
    fp32 embeddings from two implementations (1e-5 each) averaged over <= 16 windows."""
    from mirx.model import DenseNet121
    from mirx.xai import SBSMBatch, sliding_window_masks
    from oracle import xai as ox
    from oracle import densenet as OD
    torch.manual_seed(3)
    size = 64
    m = DenseNet121().eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    xq = torch.randn(2, 3, size, size, generator=g)
    xr = None if self_sim else torch.randn(2, 3, size, size, generator=g)
    masks = sliding_window_masks((size, size), 24, 16)
    assert masks.shape == (25, 1, size, size) and masks.dtype == np.uint8
    assert masks[0, 0, :16, :16].sum() == 0 and masks[0, 0, 16:, :].all()       # first window (-8 .. 16) clipped
    want = ox.sbsm_batch(lambda t: OD.embed(t, sd), masks, xq, xr, gpu_batch=7)
    ex = SBSMBatch(m.to(dev), (size, size), gpu_batch=7)
    ex.generate_masks(24, 16, savepath=str(tmp_path / "masks.npy"))
    got = ex(xq.to(dev), None if self_sim else xr.to(dev))
    assert got.shape == want.shape == ((2 if self_sim else 4), size, size)      # [Q * B, H, W] for pairs
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=3e-5)
    ex2 = SBSMBatch(m, (size, size), gpu_batch=64)
    ex2.load_masks(str(tmp_path / "masks.npy"))
    np.testing.assert_allclose(ex2(xq.to(dev), None if self_sim else xr.to(dev)).cpu().numpy(), got.cpu().numpy(),
                               rtol=0, atol=5e-6)
