"""ConvNeXtV2 on the GPU: the HIP depthwise-7x7 kernel and the whole embedder against the CPU
oracle restatement (fp32; tolerance 1e-5 absolute on unit-norm embeddings)."""
import ctypes

import pytest
import torch

from oracle import convnext as OC

pytestmark = pytest.mark.gpu


def test_dwconv7_kernel():
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for (b, c, h, w) in ((2, 128, 96, 96), (3, 40, 17, 23), (2, 1024, 12, 12), (1, 33, 5, 4),
                         (2, 64, 24, 24), (1, 32, 48, 20), (1, 96, 32, 16)):        # 12- and 16-wide tile variants
        x = torch.randn(b, c, h, w, generator=g)
        wt = 0.2 * torch.randn(c, 1, 7, 7, generator=g)
        bias = torch.randn(c, generator=g)
        want = torch.nn.functional.conv2d(x, wt, bias, padding=3, groups=c).permute(0, 2, 3, 1).contiguous()
        xg, wg, bg = x.cuda(), wt.cuda(), bias.cuda()
        y = torch.empty((b, h, w, c), device="cuda")
        rc = lib.mirx_dwconv7x7_nchw_to_nhwc(ctypes.c_void_p(xg.data_ptr()), ctypes.c_void_p(wg.data_ptr()),
                                             ctypes.c_void_p(bg.data_ptr()), b, c, h, w,
                                             ctypes.c_void_p(y.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        torch.testing.assert_close(y.cpu(), want, atol=2e-5, rtol=1e-5)


def test_embeddings_match_cpu_restatement():
    from mirx.model import ConvNeXtV2
    torch.manual_seed(0)
    m = ConvNeXtV2(embedding_dim=256).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = OC.embed(x[:1], sd)
        y = m.cuda()(x.cuda()).cpu()
    assert y.shape == (2, 256)
    assert float((y.norm(dim=1) - 1).abs().max()) < 1e-6
    assert float((y[:1] - ref).abs().max()) <= 1e-5, float((y[:1] - ref).abs().max())


def test_rows_do_not_depend_on_the_batch():
    """the bench's batch (64 at 384 x 384) against the same images embedded two at a time (<= 1e-6), and a
    row of it against the CPU restatement (1e-5)."""
    from mirx.model import ConvNeXtV2
    torch.manual_seed(0)
    m = ConvNeXtV2(embedding_dim=256).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(64, 3, 384, 384, generator=torch.Generator().manual_seed(4))
    m = m.cuda()
    with torch.no_grad():
        big = m(x.cuda()).cpu()
        small = torch.cat([m(x[i:i + 2].cuda()).cpu() for i in (0, 30, 62)])
        ref = OC.embed(x[63:64], sd)
    assert float((big[[0, 1, 30, 31, 62, 63]] - small).abs().max()) <= 1e-6
    assert float((big[63:64] - ref).abs().max()) <= 1e-5
