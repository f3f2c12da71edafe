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


def test_dwconv7_channels_last_kernel():
    """mirx_dwconv7x7_nhwc (no LDS, a lane = a channel, 4-row strips with a register window) against conv2d in float64:
    ragged heights / widths / channel counts, maps narrower than the window."""
    from mirx import _lib
    lib = _lib.load()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    g = torch.Generator().manual_seed(3)
    for (b, c, h, w) in ((2, 128, 96, 96), (3, 40, 17, 23), (2, 1024, 12, 12), (1, 33, 5, 4), (2, 512, 24, 24), (1, 70, 3, 50),
                         (1, 64, 7, 7), (2, 200, 9, 1)):
        x = torch.randn(b, c, h, w, generator=g)
        wt = 0.2 * torch.randn(c, 1, 7, 7, generator=g)
        bias = torch.randn(c, generator=g)
        want = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=3, groups=c).permute(0, 2, 3, 1)
        xg = x.permute(0, 2, 3, 1).contiguous().cuda()
        wg = wt.reshape(c, 49).t().contiguous().cuda()
        y = torch.full((b, h, w, c), float("nan"), device="cuda")
        assert lib.mirx_dwconv7x7_nhwc(vp(xg), vp(wg), vp(bias.cuda()), b, c, h, w, vp(y), None) == 0
        torch.cuda.synchronize()
        assert float((y.cpu().double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_layernorm_patch2_kernel():
    """mirx_layernorm_patch2_nhwc: per-pixel LayerNorm over channels written as the (ky, kx, c) patch rows of a 2 x 2 / 2 conv."""
    from mirx import _lib
    lib = _lib.load()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    g = torch.Generator().manual_seed(4)
    for (b, h, w, c) in ((2, 8, 6, 128), (1, 24, 24, 512), (3, 2, 2, 36)):
        x = torch.randn(b, h, w, c, generator=g) * 2 + 0.5
        gam, bet = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        ln = torch.nn.functional.layer_norm(x.double(), (c,), gam.double(), bet.double(), 1e-6)
        want = ln.view(b, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(b * (h // 2) * (w // 2), 4 * c)
        y = torch.full((b * (h // 2) * (w // 2), 4 * c), float("nan"), device="cuda")
        assert lib.mirx_layernorm_patch2_nhwc(vp(x.cuda()), b, h, w, c, vp(gam.cuda()), vp(bet.cuda()), 1e-6, vp(y), None) == 0
        torch.cuda.synchronize()
        assert float((y.cpu().double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    assert lib.mirx_layernorm_patch2_nhwc(vp(y), 1, 3, 2, 128, None, None, 1e-6, vp(y), None) != 0      # odd height


def test_channels_last_path_matches_the_nchw_path():
    """The ConvNeXtV2 fast path with a channels-last residual stream (no-LDS depthwise conv, LayerNorm into patch rows,
    row-major block tail) against the NCHW path of the same model: the same arithmetic per element up to the summation order
    of the depthwise taps and of the final average pool."""
    from mirx.model import ConvNeXtV2
    torch.manual_seed(0)
    m = ConvNeXtV2(embedding_dim=256).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    m = m.cuda()
    x = torch.randn(3, 3, 384, 384, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        assert m.convnext._nhwc_ok(x)
        a = m(x)
        old = m.configure(cnx_channels_last=False)
        try:
            assert not m.convnext._nhwc_ok(x)
            b = m(x)
        finally:
            m.configure(**old.__dict__)
        odd = m(x[:, :, :352, :352].contiguous())                    # 352 / 32 = 11: still the fast path
        assert m.convnext._nhwc_ok(x[:, :, :352, :352]) and torch.isfinite(odd).all()
    assert float((a - b).abs().max()) < 2e-6


@pytest.mark.parametrize("n_img,tpi,k,n", [(3, 144, 128, 512), (2, 576, 512, 2048), (5, 130, 64, 200), (1, 9216, 128, 512)])
def test_fc1_gelu_with_grn_partials(n_img, tpi, k, n):
    """mirx_linear_split2h_gelu_grn: the fc1 + GELU output equals mirx_linear_split2h's bit for bit, and gx is the L2 norm over
    each image's tokens of that output (float64 restatement; tiles that straddle two images, ragged last tile)."""
    import mirx.model as mm
    from mirx import _lib
    lib = _lib.load()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    dev = torch.device("cuda:0")
    torch.manual_seed(n_img + tpi)
    m = n_img * tpi
    x = torch.randn(m, k, device=dev).clamp_(-5, 5)
    lin = torch.nn.Linear(k, n).to(dev)
    w2, ws = mm._linear_h2_weights(lin)
    xs = mm._terms_scale(5.0)
    with torch.no_grad():
        want = mm._linear_h2(lin, x, 5.0, act=1)
    y = torch.full((m, n), float("nan"), device=dev)
    parts = torch.full((((m + 127) // 128) * 2 * n,), float("nan"), device=dev)
    gx = torch.full((n_img, n), float("nan"), device=dev)
    assert lib.mirx_linear_split2h_gelu_grn(vp(x), n_img, tpi, k, vp(w2), vp(lin.bias.detach()), n, xs, 1.0 / (xs * ws), vp(y),
                                            vp(parts), vp(gx), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    ref = want.double().view(n_img, tpi, n).pow(2).sum(1).sqrt()
    assert float((gx.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    gx2 = torch.empty_like(gx)
    assert lib.mirx_linear_split2h_gelu_grn(vp(x), n_img, tpi, k, vp(w2), vp(lin.bias.detach()), n, xs, 1.0 / (xs * ws), vp(y),
                                            vp(parts), vp(gx2), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(gx, gx2)                                                   # fixed summation order
    assert lib.mirx_linear_split2h_gelu_grn(vp(x), n_img, 100, k, vp(w2), None, n, xs, 1.0, vp(y), vp(parts), vp(gx), None) != 0
